// Compile-and-run check of include/snail_adapter.hpp against MOCK types that expose the same member names as the reference's
// BVH / Scene / Context / ShadowContext / Camera / TreeStats / Options / MipmapTexture / Light (src/bvh/tree.h, src/scene.h,
// src/ray_group.h, src/camera.h, src/tree_stats.h, src/render.h, src/mipmap_texture.h, src/light.h).  This is a test of the adapter
// templates, not a build of the reference.
//   adapter_mock <dir>
// reads its inputs from <dir> (written by tests/test_gpu_parity.py::test_cpp_adapter_end_to_end) and writes, for the Python side to
// compare with the oracle:
//   out_primary.bin   a frame through HipBVH::BeginFrame + per-packet TraversePrimary(Context<1,0>) copies (prefetched path)
//   out_sh_imm.bin    shadow packet 0 through HipBVH::TraverseShadow (immediate path: one synchronous call per packet)
//   out_sh_batch.bin  all shadow packets through snail::ShadowBatch (one call)
//   out_ry_imm.bin    secondary packet 0 through HipBVH::TraversePrimary(Context<0,1>) (immediate path)
//   out_ry_batch.bin  all secondary packets through snail::RayBatch (one call)
//   out_tiles.bin     Render(scene, camera, resx, resy, data, coords, offsets, options, rank, threads)   -- the reference's signature
//   out_image.bin     Render(scene, camera, image, options, threads)                                      -- the reference's signature
//   out_sh_thr.bin    all shadow packets, one HipBVH::TraverseShadow call each, from 8 std::threads at once on the ONE scene
//   out_ry_thr.bin    all secondary packets, one HipBVH::TraversePrimary(Context<0,1>) call each, from the same 8 threads (the reference's
//                     Render(..., threads) shares one const Scene over its pthread workers: src/render.cpp:214-267, src/thread_pool.cpp:151-180)
//   stats.txt         the TreeStats each of them returned / accumulated
#include <cstdint>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

// ---- mock reference types (names and members as in the reference) ----
using std::vector;
typedef unsigned int uint;
int gVals[16] = {0};
struct Vec3f { float x, y, z; };
struct Camera { float plane_dist; Vec3f pos, right, up, front; };
struct TreeStats {
	unsigned in = 0, it = 0, sk = 0, rays = 0;
	void Intersection(unsigned v = 1) { in += v; }
	void LoopIteration(unsigned v = 1) { it += v; }
	void Skip(unsigned v = 1) { sk += v; }
	void TracingRays(unsigned v = 1) { rays += v; }
};
struct Options { Options(bool refl, bool rdtsc) : reflections(refl), rdtscShader(rdtsc) {} Options() { reflections = rdtscShader = 0; } bool reflections, rdtscShader; };   // src/render.h:9-14
struct Light { Vec3f pos, color; float radius, radSq, iRadius; };
struct MipmapTexture {
	int w = 0, h = 0, pitch = 0; std::vector<unsigned char> bytes;
	int Width() const { return w; } int Height() const { return h; } int Pitch() const { return pitch; }
	void *DataPointer() { return bytes.data(); }
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
template <class AccStruct> struct Scene {       // src/scene.h:26-58: the members the device pipeline reads
	AccStruct geometry;
	Vec3f ambientLight{0.1f, 0.1f, 0.1f};
	vector<Light> lights;
};
// the reference's own generic Render templates (src/render.h:16-23): must LOSE overload resolution against the adapter's
// ... for a call the device pipeline implements; a call with a switch it does not implement (gVals[5], [6], [8], [9]) must ARRIVE here,
// with the frame the reference's RenderTask::Work is going to ask for already prefetched (twice the resolution under 4x antialiasing)
static int g_expectHostRender = 0;
template <class AccStruct>
TreeStats Render(const Scene<AccStruct> &scene, const Camera &, uint, uint, unsigned char *, const vector<int> &, const vector<int> &, const Options, uint, uint) {
	if(!g_expectHostRender) { std::puts("generic tile Render called"); std::exit(3); }
	std::printf("host tile Render: prefetched %d x %d %d\n", scene.geometry.Frame().resx, scene.geometry.Frame().resy, (int)scene.geometry.HaveFrame());
	TreeStats st; st.it = 777; return st;
}
template <class AccStruct> TreeStats Render(const Scene<AccStruct> &scene, const Camera &, MipmapTexture &, const Options, uint) {
	if(!g_expectHostRender) { std::puts("generic image Render called"); std::exit(3); }
	std::printf("host image Render: prefetched %d x %d %d\n", scene.geometry.Frame().resx, scene.geometry.Frame().resy, (int)scene.geometry.HaveFrame());
	TreeStats st; st.it = 778; return st;
}

#define SNAIL_ADAPTER_RENDER_OVERLOADS
#include "../../include/snail_adapter.hpp"

template <class T> static std::vector<T> slurp(const std::string &path, bool optional = false) {
	FILE *f = std::fopen(path.c_str(), "rb");
	if(!f) { if(optional) return {}; std::perror(path.c_str()); std::exit(2); }
	std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
	std::vector<T> v(n / sizeof(T)); if(n && std::fread(v.data(), 1, n, f) != (size_t)n) std::exit(2); std::fclose(f); return v;
}
template <class T> static void dump(FILE *f, const std::vector<T> &v) { std::fwrite(v.data(), sizeof(T), v.size(), f); }

int main(int argc, char **argv) {
	if(argc < 2) { std::puts("compiled and linked"); return 0; }
	const std::string d = std::string(argv[1]) + "/";
	const std::vector<int> meta = slurp<int>(d + "meta.bin");   // depth, resx, resy, nShadow, nRays, reflections, depthShading
	const int depth = meta[0], resx = meta[1], resy = meta[2], nSh = meta[3], nRy = meta[4];
	Scene<snail::HipBVH<MockBVH>> scene;
	MockBVH bvh;
	bvh.nodes = slurp<Node>(d + "nodes.bin"); bvh.tris = slurp<Triangle>(d + "tris.bin"); bvh.depth = depth;
	const std::vector<float> c = slurp<float>(d + "cam.bin");
	const Camera cam{c[12], {c[0], c[1], c[2]}, {c[3], c[4], c[5]}, {c[6], c[7], c[8]}, {c[9], c[10], c[11]}};
	{
		const std::vector<float> L = slurp<float>(d + "lights.bin", true);
		for(size_t k = 0; k + 6 < L.size(); k += 7) scene.lights.push_back(Light{{L[k], L[k + 1], L[k + 2]}, {L[k + 3], L[k + 4], L[k + 5]}, L[k + 6], L[k + 6] * L[k + 6], 1.0f / L[k + 6]});
	}
	snail::HipBVH<MockBVH> &acc = scene.geometry;
	acc.Upload(bvh, 0);
	// meta[8] != 0: the arithmetic the reference's SSE build executes on this host ("Arithmetic"): every expectation of the
	// test then comes from the oracle's ORC_MODE_SSE, which runs this CPU's rcpps / rsqrtps
	const bool hostSse = meta.size() > 8 && meta[8] != 0;
	if(hostSse && !acc.SetArith(SNAIL_ARITH_HOST_SSE)) { std::fprintf(stderr, "SetArith(HOST_SSE): %s\n", snail_last_error()); return 3; }
	FILE *fs = std::fopen((d + "stats.txt").c_str(), "w");

	{ // ---- prefetched primary path ----
		acc.BeginFrame(cam, resx, resy);
		std::vector<float> t((size_t)resx * resy);
		std::vector<int> id((size_t)resx * resy);
		Vec3q origin; for(int l = 0; l < 4; l++) { origin.x[l] = cam.pos.x; origin.y[l] = cam.pos.y; origin.z[l] = cam.pos.z; }
		TreeStats total;
		for(int y = 0; y < resy; y += 16) for(int x = 0; x < resx; x += 16) {
			floatq dist[64]; i32x4 obj[64]; Vec2q bary[64]; TreeStats st;
			Context<1, 0> ctx{{&origin, nullptr, nullptr, 64, nullptr}, dist, obj, nullptr, bary, &st};
			acc.SetPacket(x, y);
			acc.TraversePrimary(ctx);
			total.in += st.in; total.it += st.it; total.sk += st.sk;
			for(int q = 0; q < 64; q++) for(int l = 0; l < 4; l++) {
				int xx = x + (q & 3) * 4 + l, yy = y + (q >> 2);
				if(xx < resx && yy < resy) { t[(size_t)yy * resx + xx] = dist[q].v[l]; id[(size_t)yy * resx + xx] = obj[q].v[l]; }
			}
		}
		acc.EndFrame();
		FILE *f = std::fopen((d + "out_primary.bin").c_str(), "wb"); dump(f, t); dump(f, id); std::fclose(f);
		std::fprintf(fs, "primary %u %u %u %u\n", total.in, total.it, 0u, total.sk);
	}
	if(nSh > 0) { // ---- shadow packets: immediate (packet 0) and batched (all) ----
		const std::vector<float> o3 = slurp<float>(d + "sh_origin.bin");
		const std::vector<Vec3q> dir = slurp<Vec3q>(d + "sh_dir.bin"), idir = slurp<Vec3q>(d + "sh_idir.bin");
		std::vector<floatq> dist = slurp<floatq>(d + "sh_dist.bin");
		std::vector<floatq> imm(dist.begin(), dist.begin() + 64);
		std::vector<Vec3q> org((size_t)nSh);
		for(int p = 0; p < nSh; p++) for(int l = 0; l < 4; l++) { org[p].x[l] = o3[p * 3]; org[p].y[l] = o3[p * 3 + 1]; org[p].z[l] = o3[p * 3 + 2]; }
		TreeStats st0, st1;
		ShadowContext c0{{&org[0], dir.data(), idir.data(), 64, nullptr}, imm.data(), &st0};
		acc.TraverseShadow(c0);
		FILE *f = std::fopen((d + "out_sh_imm.bin").c_str(), "wb"); dump(f, imm); std::fclose(f);
		snail::ShadowBatch batch;
		for(int p = 0; p < nSh; p++) { ShadowContext cp{{&org[p], dir.data() + p * 64, idir.data() + p * 64, 64, nullptr}, dist.data() + p * 64, nullptr}; batch.Add(cp); }
		batch.Flush(acc, &st1);
		f = std::fopen((d + "out_sh_batch.bin").c_str(), "wb"); dump(f, dist); std::fclose(f);
		std::fprintf(fs, "shadow_imm %u %u %u %u\nshadow_batch %u %u %u %u\n", st0.in, st0.it, 0u, st0.sk, st1.in, st1.it, 0u, st1.sk);
	}
	if(nRy > 0) { // ---- secondary packets RayGroup<0,1>: immediate (packet 0) and batched (all) ----
		const std::vector<Vec3q> org = slurp<Vec3q>(d + "ry_origin.bin"), dir = slurp<Vec3q>(d + "ry_dir.bin"), idir = slurp<Vec3q>(d + "ry_idir.bin");
		std::vector<char> mask = slurp<char>(d + "ry_mask.bin");
		std::vector<floatq> dist = slurp<floatq>(d + "ry_dist.bin");
		std::vector<i32x4> obj((size_t)nRy * 64, i32x4{{0, 0, 0, 0}});
		std::vector<Vec2q> bary((size_t)nRy * 64, Vec2q{{0, 0, 0, 0}, {0, 0, 0, 0}});
		std::vector<floatq> d0(dist.begin(), dist.begin() + 64); std::vector<i32x4> o0(64, i32x4{{0, 0, 0, 0}}); std::vector<Vec2q> b0(64, Vec2q{{0, 0, 0, 0}, {0, 0, 0, 0}});
		TreeStats st0, st1;
		Context<0, 1> c0{{org.data(), dir.data(), idir.data(), 64, mask.data()}, d0.data(), o0.data(), nullptr, b0.data(), &st0};
		acc.TraversePrimary(c0);
		FILE *f = std::fopen((d + "out_ry_imm.bin").c_str(), "wb"); dump(f, d0); dump(f, o0); dump(f, b0); std::fclose(f);
		snail::RayBatch batch(false, true);
		for(int p = 0; p < nRy; p++) {
			Context<0, 1> cp{{org.data() + p * 64, dir.data() + p * 64, idir.data() + p * 64, 64, mask.data() + p * 64}, dist.data() + p * 64, obj.data() + p * 64, nullptr,
							 bary.data() + p * 64, nullptr};
			batch.Add(cp);
		}
		batch.Flush(acc, &st1);
		f = std::fopen((d + "out_ry_batch.bin").c_str(), "wb"); dump(f, dist); dump(f, obj); dump(f, bary); std::fclose(f);
		std::fprintf(fs, "rays_imm %u %u %u %u\nrays_batch %u %u %u %u\n", st0.in, st0.it, 0u, st0.sk, st1.in, st1.it, 0u, st1.sk);
	}
	if(nSh > 0 && nRy > 0) { // ---- the immediate path from 8 threads at once on one scene: every thread its own packets, results and TreeStats ----
		const std::vector<float> o3 = slurp<float>(d + "sh_origin.bin");
		const std::vector<Vec3q> sdir = slurp<Vec3q>(d + "sh_dir.bin"), sidir = slurp<Vec3q>(d + "sh_idir.bin");
		std::vector<floatq> sdist = slurp<floatq>(d + "sh_dist.bin");
		std::vector<Vec3q> sorg((size_t)nSh);
		for(int p = 0; p < nSh; p++) for(int l = 0; l < 4; l++) { sorg[p].x[l] = o3[p * 3]; sorg[p].y[l] = o3[p * 3 + 1]; sorg[p].z[l] = o3[p * 3 + 2]; }
		const std::vector<Vec3q> rorg = slurp<Vec3q>(d + "ry_origin.bin"), rdir = slurp<Vec3q>(d + "ry_dir.bin"), ridir = slurp<Vec3q>(d + "ry_idir.bin");
		std::vector<char> rmask = slurp<char>(d + "ry_mask.bin");
		std::vector<floatq> rdist = slurp<floatq>(d + "ry_dist.bin");
		std::vector<i32x4> robj((size_t)nRy * 64, i32x4{{0, 0, 0, 0}});
		std::vector<Vec2q> rbary((size_t)nRy * 64, Vec2q{{0, 0, 0, 0}, {0, 0, 0, 0}});
		constexpr int kThreads = 8, kRounds = 3;   // (rounds: the same packets again from a fresh copy of the inputs -- more calls in flight per thread)
		std::vector<TreeStats> shStats(kThreads), ryStats(kThreads);
		const std::vector<floatq> sdist0 = sdist, rdist0 = rdist;
		const snail::HipBVH<MockBVH> &cacc = acc;
		std::vector<std::thread> pool;
		for(int k = 0; k < kThreads; k++)
			pool.emplace_back([&, k] {
				for(int round = 0; round < kRounds; round++) {
					for(int p = k; p < nSh; p += kThreads) {
						std::copy(sdist0.begin() + p * 64, sdist0.begin() + (p + 1) * 64, sdist.begin() + p * 64);
						TreeStats st;
						ShadowContext c{{&sorg[p], sdir.data() + p * 64, sidir.data() + p * 64, 64, nullptr}, sdist.data() + p * 64, &st};
						cacc.TraverseShadow(c);
						if(round == 0) { shStats[k].in += st.in; shStats[k].it += st.it; shStats[k].sk += st.sk; }
					}
					for(int p = k; p < nRy; p += kThreads) {
						std::copy(rdist0.begin() + p * 64, rdist0.begin() + (p + 1) * 64, rdist.begin() + p * 64);
						std::fill(robj.begin() + p * 64, robj.begin() + (p + 1) * 64, i32x4{{0, 0, 0, 0}});
						std::fill(rbary.begin() + p * 64, rbary.begin() + (p + 1) * 64, Vec2q{{0, 0, 0, 0}, {0, 0, 0, 0}});
						TreeStats st;
						Context<0, 1> c{{rorg.data() + p * 64, rdir.data() + p * 64, ridir.data() + p * 64, 64, rmask.data() + p * 64}, rdist.data() + p * 64, robj.data() + p * 64,
										 nullptr, rbary.data() + p * 64, &st};
						cacc.TraversePrimary(c);
						if(round == 0) { ryStats[k].in += st.in; ryStats[k].it += st.it; ryStats[k].sk += st.sk; }
					}
				}
			});
		for(std::thread &t : pool) t.join();
		TreeStats sh, ry;
		for(int k = 0; k < kThreads; k++) { sh.in += shStats[k].in; sh.it += shStats[k].it; sh.sk += shStats[k].sk; ry.in += ryStats[k].in; ry.it += ryStats[k].it; ry.sk += ryStats[k].sk; }
		FILE *f = std::fopen((d + "out_sh_thr.bin").c_str(), "wb"); dump(f, sdist); std::fclose(f);
		f = std::fopen((d + "out_ry_thr.bin").c_str(), "wb"); dump(f, rdist); dump(f, robj); dump(f, rbary); std::fclose(f);
		std::fprintf(fs, "shadow_thr %u %u %u %u\nrays_thr %u %u %u %u\n", sh.in, sh.it, 0u, sh.sk, ry.in, ry.it, 0u, ry.sk);
	}
	{ // ---- the tile API with the reference's signatures ----
		gVals[7] = meta[5]; gVals[1] = meta[6];
		const std::vector<int> coords = slurp<int>(d + "tiles.bin"), offsets = slurp<int>(d + "offsets.bin");
		size_t total = 0;
		for(size_t k = 0; k < offsets.size(); k++) total = std::max(total, (size_t)offsets[k] + (size_t)3 * coords[k * 4 + 2] * coords[k * 4 + 3]);
		std::vector<unsigned char> data(total, 0xAB);
		const TreeStats st = Render(scene, cam, (uint)resx, (uint)resy, data.data(), coords, offsets, Options(), 0u, 4u);
		FILE *f = std::fopen((d + "out_tiles.bin").c_str(), "wb"); dump(f, data); std::fclose(f);
		std::fprintf(fs, "tiles %u %u %u %u\n", st.in, st.it, st.rays, st.sk);
		{ // the same call with the tree uploaded THREE times (here: to the one device there is): the tile list is dealt over the handles
			// (snail_render_tiles_multi); bytes and counters must not change
			Scene<snail::HipBVH<MockBVH>> scene3;
			scene3.lights = scene.lights; scene3.ambientLight = scene.ambientLight;
			scene3.geometry.Upload(bvh, std::vector<int>{0, 0, 0});
			if(hostSse && !scene3.geometry.SetArith(SNAIL_ARITH_HOST_SSE)) return 3;
			std::vector<unsigned char> data3(total, 0xAB);
			const TreeStats s3 = Render(scene3, cam, (uint)resx, (uint)resy, data3.data(), coords, offsets, Options(), 0u, 4u);
			const bool same = data3 == data && s3.in == st.in && s3.it == st.it && s3.rays == st.rays && s3.sk == st.sk;
			std::fprintf(fs, "tiles_multi %d %d\n", (int)same, scene3.geometry.DeviceCount());
		}
		MipmapTexture img; img.w = resx; img.h = resy; img.pitch = (resx * 3 + 63) / 64 * 64; img.bytes.assign((size_t)img.pitch * resy, 0xCD);
		// Options(true, ...) must change NOTHING: the reference stores its Options and never reads them (src/render.cpp:24,37); the bounce is
		// gVals[7]'s alone (src/scene_trace.cpp:454).  With gVals[7] == 0 this call must give the no-bounce picture.
		const TreeStats si = Render(scene, cam, img, Options(true, false), 4u);
		f = std::fopen((d + "out_image.bin").c_str(), "wb"); dump(f, img.bytes); std::fclose(f);
		std::fprintf(fs, "image %u %u %u %u %d\n", si.in, si.it, si.rays, si.sk, img.pitch);
	}
	// ---- a switch the device pipeline does not implement: the call must reach the reference's own renderer (the generic templates above),
	// prefetched; meta[7] = index into gVals (5, 6, 8; 59 = gVals[5] AND gVals[9]: the stats heat-map under 4x antialiasing), 0 = skip ----
	if(meta.size() > 7 && meta[7] > 0) {
		const int sw = meta[7];
		gVals[7] = 0; gVals[1] = 0;
		if(sw == 59) { gVals[5] = 1; gVals[9] = 1; } else gVals[sw] = 1;
		if(sw == 6) bvh.shTris.resize(bvh.tris.size());        // full shading needs shading data (src/scene_trace.cpp:145)
		g_expectHostRender = 1;
		const std::vector<int> coords = slurp<int>(d + "tiles.bin"), offsets = slurp<int>(d + "offsets.bin");
		std::vector<unsigned char> data(1 << 20, 0);
		const TreeStats st = Render(scene, cam, (uint)resx, (uint)resy, data.data(), coords, offsets, Options(), 0u, 4u);
		MipmapTexture img; img.w = resx; img.h = resy; img.pitch = resx * 3; img.bytes.assign((size_t)img.pitch * resy, 0);
		// (gVals[8], the tint, belongs to the tile-list renderer only -- colorizeNodes: with it set the image call stays on the device)
		g_expectHostRender = sw == 8 ? 0 : 1;
		const TreeStats si = Render(scene, cam, img, Options(), 4u);
		std::printf("switch %d: tile stats %u image stats %u frame left %d\n", sw, st.it, si.it, (int)acc.HaveFrame());
		if(sw == 59) { gVals[5] = 0; gVals[9] = 0; } else gVals[sw] = 0;
		g_expectHostRender = 0;
	}
	std::fclose(fs);
	std::printf("adapter ok: %d x %d, normal of tri 0 = %g %g %g\n", resx, resy, acc.GetNormal(0, 0).x, acc.GetNormal(0, 0).y, acc.GetNormal(0, 0).z);
	return 0;
}
