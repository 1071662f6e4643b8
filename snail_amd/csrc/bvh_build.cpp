// bvh_build.cpp -- host-side scene construction for libsnailhip.so.
//
// The reference builds its BVH on the CPU once per scene (BVH::Construct, src/bvh/tree.cpp:293-328) and
// the build stays on the CPU here too.  triId is an index into the triangle array AS
// PERMUTED BY THE BUILDER (src/bvh/tree.cpp:118-120), so drop-in parity needs the very same tree:
// SAH full sweep over the three axes with std::sort on the centroid key 3a+ba+ca, cost
// SA_l*n + SA_r*(count-n), std::nth_element re-partition unless the winning axis was the last one
// sorted, leaves at <= 4 triangles / depth 63 / no-split-cheaper (src/bvh/tree.cpp:51-159).
// All arithmetic fp32 with one rounding per operation (build with -ffp-contract=off, no -mfma).
//
// Implementation notes (this is not a transcription): triangle bounds and the three sort keys are
// computed once and carried along with the permutation; recursion is replaced by an explicit stack
// that reproduces the reference's pre-order node numbering (both children are appended when the parent
// is split, the left subtree is completed before the right one).
#include "../../include/snail_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace snail {

struct Tri64 { float a[3], ba[3], ca[3]; float t0, it0; int32_t pad; float plane[4]; };
struct Node32 { float bmin[3], bmax[3]; uint32_t sub; int32_t aux; };
static_assert(sizeof(Tri64) == 64 && sizeof(Node32) == 32, "record sizes are part of the ABI");

static inline float fmin2(float a, float b) { return a < b ? a : b; } // veclib/vecbase.h:75-76
static inline float fmax2(float a, float b) { return a > b ? a : b; }

struct Bounds {
	float lo[3], hi[3];
	void merge(const Bounds &o) {
		for(int k = 0; k < 3; k++) { lo[k] = fmin2(lo[k], o.lo[k]); hi[k] = fmax2(hi[k], o.hi[k]); }
	}
	// BoxSA, src/bvh/tree.cpp:45-47: (W*(D+H) + D*H)*2 with W=x, H=y, D=z extents
	float area() const {
		float w = hi[0] - lo[0], h = hi[1] - lo[1], d = hi[2] - lo[2];
		return (w * (d + h) + d * h) * 2.0f;
	}
};

// Triangle::GetBBox (src/triangle.h:35-37,62-70): the corners are re-derived as a, ba+a, ca+a
static Bounds boundsOf(const Tri64 &t) {
	Bounds b;
	for(int k = 0; k < 3; k++) {
		float p1 = t.a[k], p2 = t.ba[k] + t.a[k], p3 = t.ca[k] + t.a[k];
		b.lo[k] = fmin2(p1, fmin2(p2, p3));
		b.hi[k] = fmax2(p1, fmax2(p2, p3));
	}
	return b;
}

struct Item { Bounds box; float key[3]; int32_t src; };  // per triangle, permuted together with it

struct KeyLess {
	const Item *items; int axis;
	bool operator()(int i, int j) const { return items[i].key[axis] < items[j].key[axis]; }
};

struct Job { int node, first, count, depth; };

int buildSweep(Tri64 *tris, int nTris, Node32 *nodes, int *outNodes, int *outDepth, int32_t *perm) {
	if(nTris <= 0) { *outNodes = 0; *outDepth = 0; return 0; }
	const int maxDepth = SNAIL_MAX_DEPTH;

	std::vector<Item> items(nTris);
	for(int i = 0; i < nTris; i++) {
		items[i].box = boundsOf(tris[i]);
		for(int k = 0; k < 3; k++) items[i].key[k] = tris[i].a[k] * 3.0f + tris[i].ba[k] + tris[i].ca[k]; // OrderTris, tree.cpp:12-18
		items[i].src = i;
	}

	auto putBox = [&](int n, const Bounds &b) {
		for(int k = 0; k < 3; k++) { nodes[n].bmin[k] = b.lo[k]; nodes[n].bmax[k] = b.hi[k]; }
	};

	Bounds root = items[0].box;
	for(int i = 1; i < nTris; i++) root.merge(items[i].box);
	putBox(0, root);
	nodes[0].sub = 0; nodes[0].aux = 0;
	int nNodes = 1, depth = 0;

	std::vector<Job> todo;
	todo.push_back(Job{0, 0, nTris, 0});
	std::vector<int> order;
	std::vector<float> leftSA, rightSA;
	std::vector<Tri64> triTmp;
	std::vector<Item> itemTmp;

	while(!todo.empty()) {
		Job j = todo.back();
		todo.pop_back();
		const int first = j.first, count = j.count;
		bool leaf = count <= 4 || j.depth == maxDepth - 1;
		int splitAt = count / 2, splitAxis = 0;

		if(!leaf) {
			Bounds nb;
			for(int k = 0; k < 3; k++) { nb.lo[k] = nodes[j.node].bmin[k]; nb.hi[k] = nodes[j.node].bmax[k]; }
			{ // MaxAxis(bbox.Size()), src/rtbase.h:136-138 -- only survives if no finite cost is found
				float sx = nb.hi[0] - nb.lo[0], sy = nb.hi[1] - nb.lo[1], sz = nb.hi[2] - nb.lo[2];
				splitAxis = sy > sx ? (sz > sy ? 2 : 1) : (sz > sx ? 2 : 0);
			}
			float best = std::numeric_limits<float>::infinity();
			const float noSplit = 1.0f * count * nb.area();
			order.resize(count); leftSA.resize(count); rightSA.resize(count);

			for(int axis = 0; axis < 3; axis++) {
				for(int n = 0; n < count; n++) order[n] = first + n;
				std::sort(order.begin(), order.begin() + count, KeyLess{items.data(), axis});

				Bounds acc = items[order[0]].box;
				leftSA[0] = acc.area();
				for(int n = 1; n < count; n++) { acc.merge(items[order[n]].box); leftSA[n] = acc.area(); }
				acc = items[order[count - 1]].box;
				rightSA[count - 1] = acc.area();
				for(int n = count - 2; n >= 0; n--) { acc.merge(items[order[n]].box); rightSA[n] = acc.area(); }

				for(size_t n = 1; n < (size_t)count; n++) {
					float cost = leftSA[n - 1] * n + rightSA[n] * (count - n);   // size_t -> float, tree.cpp:97
					if(cost < best) { best = cost; splitAt = (int)n; splitAxis = axis; }
				}
			}
			best = 0.0f + 1.0f * best;                                           // traverseCost + intersectCost * minCost
			if(noSplit < best) leaf = true;

			if(!leaf) {
				if(splitAxis != 2) {                                             // tree.cpp:111-116
					for(int n = 0; n < count; n++) order[n] = first + n;
					std::nth_element(order.begin(), order.begin() + splitAt, order.begin() + count, KeyLess{items.data(), splitAxis});
				}
				triTmp.resize(count); itemTmp.resize(count);
				for(int n = 0; n < count; n++) { triTmp[n] = tris[order[n]]; itemTmp[n] = items[order[n]]; }
				for(int n = 0; n < count; n++) { tris[first + n] = triTmp[n]; items[first + n] = itemTmp[n]; }
			}
		}

		if(leaf) {                                                               // tree.cpp:54-62
			Bounds b = items[first].box;
			for(int n = 1; n < count; n++) b.merge(items[first + n].box);
			putBox(j.node, b);
			depth = std::max(depth, j.depth);
			nodes[j.node].sub = (uint32_t)first | 0x80000000u;
			nodes[j.node].aux = count;
			continue;
		}

		Bounds lb = items[first].box, rb = items[first + count - 1].box;
		for(int n = 1; n < splitAt; n++) lb.merge(items[first + n].box);
		for(int n = splitAt; n < count; n++) rb.merge(items[first + n].box);

		const int child = nNodes;
		nodes[j.node].sub = (uint32_t)child;
		// tree.cpp:148-151: the second assignment to firstNode is the one that counts
		int firstNode = lb.lo[splitAxis] == rb.lo[splitAxis] ? (lb.hi[splitAxis] < rb.hi[splitAxis] ? 0 : 1) : 0;
		nodes[j.node].aux = (splitAxis & 0xffff) | (firstNode << 16);
		putBox(child, lb); putBox(child + 1, rb);
		nodes[child].sub = nodes[child + 1].sub = 0;
		nodes[child].aux = nodes[child + 1].aux = 0;
		nNodes += 2;

		todo.push_back(Job{child + 1, first + splitAt, count - splitAt, j.depth + 1});   // right: after the whole left subtree
		todo.push_back(Job{child + 0, first, splitAt, j.depth + 1});
	}

	if(perm) for(int i = 0; i < nTris; i++) perm[i] = items[i].src;
	*outNodes = nNodes;
	*outDepth = depth;
	return 0;
}

void trisFromVerts(const float *v, int n, Tri64 *out) {
	for(int i = 0; i < n; i++) {
		const float *p = v + (size_t)i * 9;
		Tri64 &t = out[i];
		for(int k = 0; k < 3; k++) { t.a[k] = p[k]; t.ba[k] = p[3 + k] - p[k]; t.ca[k] = p[6 + k] - p[k]; }
		float nx = t.ba[1] * t.ca[2] - t.ba[2] * t.ca[1];
		float ny = t.ba[2] * t.ca[0] - t.ba[0] * t.ca[2];
		float nz = t.ba[0] * t.ca[1] - t.ba[1] * t.ca[0];
		float len = sqrtf(nx * nx + ny * ny + nz * nz);
		float inv = 1.0f / len;                      // Vec3::operator/=(base): Inv(s) then 3 multiplies (veclib/vec3.h:47-51)
		nx *= inv; ny *= inv; nz *= inv;
		t.t0 = len; t.it0 = 1.0f / len; t.pad = 0;
		t.plane[0] = nx; t.plane[1] = ny; t.plane[2] = nz;
		t.plane[3] = nx * t.a[0] + ny * t.a[1] + nz * t.a[2];
	}
}

} // namespace snail

extern void snail_set_error(const char *fmt, ...);

// The reference's builder runs under the default floating-point environment; a host process may not (a shared library built with
// -ffast-math switches flush-to-zero / denormals-are-zero on for the thread that loads it).  The two host-side entry points compute under the
// default MXCSR and restore the caller's, so that the tree and the triangle records do not depend on what else the process has loaded.
#include <xmmintrin.h>
namespace {
struct FpEnvGuard {
	unsigned old;
	FpEnvGuard() : old(_mm_getcsr()) { _mm_setcsr(0x1f80u); }
	~FpEnvGuard() { _mm_setcsr(old); }
};
} // namespace

extern "C" int snail_tris_from_verts(const float *verts9, int n, void *tris64) {
	if(n < 0 || (n > 0 && (!verts9 || !tris64))) { snail_set_error("snail_tris_from_verts: bad arguments"); return 1; }
	FpEnvGuard fpEnv;
	snail::trisFromVerts(verts9, n, (snail::Tri64 *)tris64);
	return 0;
}

extern "C" int snail_bvh_build(void *tris64, int nTris, void *nodes32, int *nNodes, int *depth, int32_t *perm) {
	if(nTris <= 0 || !tris64 || !nodes32 || !nNodes || !depth) { snail_set_error("snail_bvh_build: bad arguments"); return 1; }
	FpEnvGuard fpEnv;
	int rc = snail::buildSweep((snail::Tri64 *)tris64, nTris, (snail::Node32 *)nodes32, nNodes, depth, perm);
	if(rc == 0 && *depth > SNAIL_MAX_DEPTH) { snail_set_error("snail_bvh_build: depth %d exceeds %d", *depth, SNAIL_MAX_DEPTH); return 2; }
	return rc;
}
