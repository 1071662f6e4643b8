// Compile-and-run check of include/snail_adapter.hpp against MOCK types that expose the same member names
// as the reference's BVH / Context / ShadowContext / Camera / TreeStats (src/bvh/tree.h, src/ray_group.h,
// src/camera.h, src/tree_stats.h).  This is a test of the adapter template, not a build of the reference.
//   adapter_mock <nodes.bin> <tris.bin> <depth> <resx> <resy> <cam13.bin> <out.bin>
// traces a frame through HipBVH::BeginFrame + per-packet TraversePrimary(Context<1,0>) copies, then one shadow
// packet and one <0,1> packet through the immediate path, and dumps the results for the Python side to compare.
#include <cstdio>
#include <vector>
#include <cstdint>
#include "../../include/snail_adapter.hpp"

struct Vec3f { float x, y, z; };
struct Camera { float plane_dist; Vec3f pos, right, up, front; };
struct TreeStats {
	unsigned in = 0, it = 0, sk = 0;
	void Intersection(unsigned v) { in += v; }
	void LoopIteration(unsigned v) { it += v; }
	void Skip(unsigned v) { sk += v; }
};
struct Vec3q { float x[4], y[4], z[4]; };
struct floatq { float v[4]; };
struct i32x4 { int v[4]; };
struct Vec2q { float x[4], y[4]; };
template <bool so, bool mask> struct RayGroup {
	enum { sharedOrigin = so, hasMask = mask };
	const Vec3q *origin, *dir, *idir; int size; char *maskp;
	const Vec3q *OriginPtr() const { return origin; }
	const Vec3q *DirPtr() const { return dir; }
	const Vec3q *IDirPtr() const { return idir; }
};
template <bool so, bool mask> struct Context {
	RayGroup<so, mask> rays; floatq *distance; i32x4 *object; i32x4 *element; Vec2q *barycentric; TreeStats *stats;
	int Size() const { return rays.size; }
	char *MaskPtr() { return rays.maskp; }
};
struct ShadowContext {
	RayGroup<1, 0> rays; floatq *distance; TreeStats *stats;
	int Size() const { return rays.size; }
};
struct Node { float b[6]; unsigned sub; int aux; };
struct Triangle { float f[16]; Vec3f Nrm() const { return Vec3f{f[12], f[13], f[14]}; } };
struct ShTriangle { float f[16]; };
struct BBox { Vec3f min, max; };
struct MockBVH {
	typedef Triangle CElement; typedef ShTriangle SElement;
	enum { isctFlags = 1, maxDepth = 64 };
	std::vector<Node> nodes; std::vector<Triangle> tris; std::vector<ShTriangle> shTris; int depth = 0;
	bool HasShadingData() const { return !shTris.empty(); }
	const ShTriangle &GetSElement(int e, int) const { return shTris[e]; }
	Vec3f GetNormal(int e, int) const { return tris[e].Nrm(); }
	int GetMaterialId(int, int) const { return 0; }
	BBox GetBBox() const { return BBox{{nodes[0].b[0], nodes[0].b[1], nodes[0].b[2]}, {nodes[0].b[3], nodes[0].b[4], nodes[0].b[5]}}; }
};

template <class T> static std::vector<T> slurp(const char *path) {
	FILE *f = std::fopen(path, "rb"); if(!f) { std::perror(path); std::exit(2); }
	std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
	std::vector<T> v(n / sizeof(T)); if(std::fread(v.data(), 1, n, f) != (size_t)n) std::exit(2); std::fclose(f); return v;
}

int main(int argc, char **argv) {
	if(argc < 8) { std::puts("compiled and linked"); return 0; }
	MockBVH bvh;
	bvh.nodes = slurp<Node>(argv[1]); bvh.tris = slurp<Triangle>(argv[2]); bvh.depth = std::atoi(argv[3]);
	const int resx = std::atoi(argv[4]), resy = std::atoi(argv[5]);
	std::vector<float> c = slurp<float>(argv[6]);
	Camera cam{c[12], {c[0], c[1], c[2]}, {c[3], c[4], c[5]}, {c[6], c[7], c[8]}, {c[9], c[10], c[11]}};
	snail::HipBVH<MockBVH> acc;
	acc.Upload(bvh, 0);
	acc.BeginFrame(cam, resx, resy);
	std::vector<float> t((size_t)resx * resy);
	std::vector<int> id((size_t)resx * resy);
	Vec3q origin; for(int l = 0; l < 4; l++) { origin.x[l] = cam.pos.x; origin.y[l] = cam.pos.y; origin.z[l] = cam.pos.z; }
	for(int y = 0; y < resy; y += 16) for(int x = 0; x < resx; x += 16) {
		floatq dist[64]; i32x4 obj[64]; Vec2q bary[64]; TreeStats st;
		Context<1, 0> ctx{{&origin, nullptr, nullptr, 64, nullptr}, dist, obj, nullptr, bary, &st};
		acc.SetPacket(x, y);
		acc.TraversePrimary(ctx);
		for(int q = 0; q < 64; q++) for(int l = 0; l < 4; l++) {
			int xx = x + (q & 3) * 4 + l, yy = y + (q >> 2);
			if(xx < resx && yy < resy) { t[(size_t)yy * resx + xx] = dist[q].v[l]; id[(size_t)yy * resx + xx] = obj[q].v[l]; }
		}
	}
	acc.EndFrame();
	FILE *f = std::fopen(argv[7], "wb");
	std::fwrite(t.data(), 4, t.size(), f); std::fwrite(id.data(), 4, id.size(), f);
	std::fclose(f);
	std::printf("adapter ok: %d x %d, normal of tri 0 = %g %g %g\n", resx, resy, acc.GetNormal(0, 0).x, acc.GetNormal(0, 0).y, acc.GetNormal(0, 0).z);
	return 0;
}
