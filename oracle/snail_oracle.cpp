// snail_oracle.cpp -- CPU ORACLE for the Snail hot path.  TEST INFRASTRUCTURE ONLY (see snail_oracle.h).
//
// A scalar-per-lane restatement of the reference's packet algorithm.  Every function cites the
// reference file:line it follows (paths relative to /root/reference).  All arithmetic is IEEE fp32
// with one rounding per operation: build with -ffp-contract=off and WITHOUT -mfma/-march=native.
// Min/Max follow veclib: Min(a,b) = a<b ? a : b, Max(a,b) = a>b ? a : b (veclib/vecbase.h:75-76; the
// SSE minps/maxps used for f32x4 have the same "second operand on NaN" behaviour, veclib/sse/f32.h:104-105).
//
// parity unpinned (strict sense): see the header comment of snail_oracle.h.

#include "snail_oracle.h"

#include <emmintrin.h>
#include <xmmintrin.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

const float kInf = std::numeric_limits<float>::infinity();

inline float Min(float a, float b) { return a < b ? a : b; }
inline float Max(float a, float b) { return a > b ? a : b; }

// ---- the two approximate operations -------------------------------------------------------------
// IEEE: veclib/vecbase.h:53-55.  SSE: veclib/sse/base.h:84-92 (rcpps / rsqrtps + one Newton step).
template <int MODE> inline float Inv(float x);
template <> inline float Inv<ORC_MODE_IEEE>(float x) { return 1.0f / x; }
template <> inline float Inv<ORC_MODE_SSE>(float x) {
	float t = _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(x)));
	return (t + t) - ((x * t) * t);
}
template <int MODE> inline float RSqrt(float x);
template <> inline float RSqrt<ORC_MODE_IEEE>(float x) { return 1.0f / sqrtf(x); }
template <> inline float RSqrt<ORC_MODE_SSE>(float x) {
	float t = _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(x)));
	return (0.5f * t) * (3.0f - ((x * t) * t));
}
// ORC_MODE_TABLE: the SSE definitions over rcpps / rsqrtps of a NAMED CPU, given as data (orc_set_tables) -- what the reference computes on that CPU,
// evaluated on any host.  What every x86 CPU probed so far does (profiles/r4_rcp_probe_*.txt): the result's sign is the input's,
// its 12 significant mantissa bits depend on the top 12 mantissa bits of the input only (rsqrtps: and on the parity of the exponent), the exponent
// moves exactly; denormal inputs read as zero, results below the normal range are zero, NaNs come back quieted.  The oracle's own restatement of that
// rule; pinned against the INSTRUCTIONS of this host with tables extracted here (tests/test_oracle_pins.py::test_table_mode_*).
uint32_t g_rcpTab[4096], g_rsqTab[2][4096];   // bits of rcpps(1.m) | rsqrtps(1.m) for [1, 2) | rsqrtps(2 x 1.m) for [2, 4), index = m >> 11
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float rcpTable(float x) {
	const uint32_t b = f2u(x), sign = b & 0x80000000u, mant = b & 0x007fffffu;
	const int ex = (int)((b >> 23) & 0xffu);
	if(ex == 0xff) return u2f(mant ? (b | 0x00400000u) : sign);       // NaN -> quiet NaN, same payload; +-inf -> +-0
	if(ex == 0) return u2f(sign | 0x7f800000u);                       // +-0 and +-denormals -> +-inf
	const uint32_t t = g_rcpTab[mant >> 11];                          // in (0.5, 1]: biased exponent 126, or 127 for an exact 1.0
	const int rex = (int)(t >> 23) + (127 - ex);                      // 1 / (1.m 2^(ex-127)) = T 2^(127-ex)
	if(rex < 1) return u2f(sign);                                     // would be denormal: flushed
	return u2f(sign | ((uint32_t)rex << 23) | (t & 0x007fffffu));
}
inline float rsqrtTable(float x) {
	const uint32_t b = f2u(x), mant = b & 0x007fffffu;
	const int ex = (int)((b >> 23) & 0xffu);
	if(ex == 0xff && mant) return u2f(b | 0x00400000u);               // NaN -> quiet NaN
	if(ex == 0) return u2f((b & 0x80000000u) | 0x7f800000u);          // +-0, +-denormals -> +-inf
	if(b & 0x80000000u) return u2f(0xffc00000u);                      // negative numbers and -inf: the default NaN
	if(ex == 0xff) return 0.0f;                                       // +inf -> +0
	const int e = ex - 127;                                           // x = 1.m 2^e = (1.m or 2 x 1.m) 4^k
	const int par = e & 1, k = (e - par) / 2;
	const uint32_t t = g_rsqTab[par][mant >> 11];
	return u2f((uint32_t)((int)(t >> 23) - k) << 23 | (t & 0x007fffffu));
}
template <> inline float Inv<ORC_MODE_TABLE>(float x) {
	float t = rcpTable(x);
	return (t + t) - ((x * t) * t);
}
template <> inline float RSqrt<ORC_MODE_TABLE>(float x) {
	float t = rsqrtTable(x);
	return (0.5f * t) * (3.0f - ((x * t) * t));
}
// one call per arithmetic mode (a run-time `mode` picks the instantiation)
#define ORC_BY_MODE(mode, F, ...)                                                                                                            \
	do {                                                                                                                                   \
		if((mode) == ORC_MODE_SSE) F<ORC_MODE_SSE>(__VA_ARGS__);                                                                           \
		else if((mode) == ORC_MODE_TABLE) F<ORC_MODE_TABLE>(__VA_ARGS__);                                                                  \
		else F<ORC_MODE_IEEE>(__VA_ARGS__);                                                                                                \
	} while(0)

struct V3 {
	float x, y, z;
	float operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
};
inline V3 mk(const float *p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
// veclib/vec3.h:92-106: dot is ((x*x' + y*y') + z*z'); cross as written there
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 b, V3 c) { return V3{b.y * c.z - b.z * c.y, b.z * c.x - b.x * c.z, b.x * c.y - b.y * c.x}; }
inline V3 VMin(V3 a, V3 b) { return V3{Min(a.x, b.x), Min(a.y, b.y), Min(a.z, b.z)}; }
inline V3 VMax(V3 a, V3 b) { return V3{Max(a.x, b.x), Max(a.y, b.y), Max(a.z, b.z)}; }

// ---- small expressions of the shading path, named so that tests/test_oracle_pins.py can pin each of them against the SAME expression
// written with the reference's own veclib types (oracle/veclib_probe.cpp `exprs`, built from /root/reference/veclib) ----
// Abs(f32x4) clears the sign bit (veclib/sse/f32.h:105): -0 -> +0, NaN stays NaN
inline float AbsQ(float v) { return fabsf(v); }
// Reflect (src/rtbase_math.h:54-58): dot = nrm | ray; ray - nrm * (dot + dot)
inline V3 reflect(V3 ray, V3 nrm) { const float dt = dot(nrm, ray); return ray - nrm * (dt + dt); }
// SafeInv (src/rtbase.h:117-120): Inv(v + 1e-8) per component
template <int MODE> inline float safeInv(float d) { return Inv<MODE>(d + 0.00000001f); }
// FastInv(f32x4) = raw rcpps (veclib/sse/f32.h:101); the scalar definition is Inv (veclib/vecbase.h:57)
template <int MODE> inline float FastInv(float x) { return MODE == ORC_MODE_SSE ? _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(x))) : MODE == ORC_MODE_TABLE ? rcpTable(x) : 1.0f / x; }
// the light's attenuation (src/scene_trace.cpp:585-587): Max(0, ((1 - a) * 0.2 + FastInv(16 * a * a)) - 0.0625), a = distance * iRadius
template <int MODE> inline float attenuation(float distance, float iRadius) {
	const float atten = distance * iRadius;
	return Max(0.0f, ((1.0f - atten) * 0.2f + FastInv<MODE>(16.0f * atten * atten)) - 0.0625f);
}
// one channel of ConvColor (src/render.cpp:11-17): Trunc(Clamp(c * 255, 0, 255)), Clamp(o, lo, hi) = Min(Max(o, lo), hi) (veclib/vecbase.h:77)
inline int convChannel(float c) { return (int)Min(Max(c * 255.0f, 0.0f), 255.0f); }
// ForWhich(a < b) (veclib/sse/f32.h:84): bit l = lane l of the ordered compare
inline int forWhichLess(const float (&a)[4], const float (&b)[4]) { int m = 0; for(int l = 0; l < 4; l++) m |= (a[l] < b[l] ? 1 : 0) << l; return m; }

struct Box { V3 mn, mx; };
// src/triangle.h:35-37,62-70: P2 = ba + a, P3 = ca + a; BoundMin = VMin(P1, VMin(P2, P3))
inline Box triBox(const OrcTri &t) {
	V3 p1 = mk(t.a), p2 = mk(t.ba) + mk(t.a), p3 = mk(t.ca) + mk(t.a);
	return Box{VMin(p1, VMin(p2, p3)), VMax(p1, VMax(p2, p3))};
}
inline void grow(Box &b, const Box &o) { b.mn = VMin(b.mn, o.mn); b.mx = VMax(b.mx, o.mx); } // bounding_box.h:21-25

// ---- BVH builder: src/bvh/tree.cpp:7-19 (OrderTris), :45-47 (BoxSA), :51-159 (FindSplitSweep) ----
struct OrderTris {
	const OrcTri *tris; int axis;
	bool operator()(int i1, int i2) const {
		const OrcTri &a = tris[i1], &b = tris[i2];
		return a.a[axis] * 3.0f + a.ba[axis] + a.ca[axis] < b.a[axis] * 3.0f + b.ba[axis] + b.ca[axis];
	}
};
inline float BoxSA(const Box &b) {
	float w = b.mx.x - b.mn.x, h = b.mx.y - b.mn.y, d = b.mx.z - b.mn.z;
	return (w * (d + h) + d * h) * 2.0f;
}
inline int MaxAxis(V3 v) { return v.y > v.x ? (v.z > v.y ? 2 : 1) : (v.z > v.x ? 2 : 0); } // src/rtbase.h:136-138

struct Builder {
	OrcTri *tris; int32_t *perm; OrcNode *nodes; int nNodes = 0; int depth = 0;
	enum { maxDepth = 64 };

	void setBox(int n, const Box &b) {
		nodes[n].bmin[0] = b.mn.x; nodes[n].bmin[1] = b.mn.y; nodes[n].bmin[2] = b.mn.z;
		nodes[n].bmax[0] = b.mx.x; nodes[n].bmax[1] = b.mx.y; nodes[n].bmax[2] = b.mx.z;
	}
	Box getBox(int n) const { return Box{mk(nodes[n].bmin), mk(nodes[n].bmax)}; }

	void split(int nNode, int first, int count, int sdepth) {
		Box bbox = getBox(nNode);
		bool leaf = count <= 4 || sdepth == maxDepth - 1;
		int minIdx = count / 2, minAxis = 0;

		if(!leaf) {
			minAxis = MaxAxis(bbox.mx - bbox.mn);
			std::vector<int> indices(count);
			float minCost = kInf;
			float noSplitCost = 1.0f * count * BoxSA(bbox);

			for(int axis = 0; axis <= 2; axis++) {
				for(int n = 0; n < count; n++) indices[n] = first + n;
				std::sort(indices.begin(), indices.begin() + count, OrderTris{tris, axis});

				std::vector<float> leftSA(count), rightSA(count);
				rightSA[count - 1] = BoxSA(triBox(tris[indices[count - 1]]));
				leftSA[0] = BoxSA(triBox(tris[indices[0]]));
				Box last = triBox(tris[indices[0]]);
				for(size_t n = 1; n < (size_t)count; n++) { grow(last, triBox(tris[indices[n]])); leftSA[n] = BoxSA(last); }
				last = triBox(tris[indices[count - 1]]);
				for(int n = count - 2; n >= 0; n--) { grow(last, triBox(tris[indices[n]])); rightSA[n] = BoxSA(last); }

				for(size_t n = 1; n < (size_t)count; n++) {
					float cost = leftSA[n - 1] * n + rightSA[n] * (count - n);
					if(cost < minCost) { minCost = cost; minIdx = (int)n; minAxis = axis; }
				}
			}
			minCost = 0.0f + 1.0f * minCost;
			if(noSplitCost < minCost) leaf = true;

			if(!leaf) {
				if(minAxis != 2) {
					for(int n = 0; n < count; n++) indices[n] = first + n;
					std::nth_element(indices.begin(), indices.begin() + minIdx, indices.end(), OrderTris{tris, minAxis});
				}
				std::vector<OrcTri> ttemp(count);
				for(int n = 0; n < count; n++) ttemp[n] = tris[indices[n]];
				for(int n = 0; n < count; n++) tris[first + n] = ttemp[n];
				if(perm) {
					std::vector<int32_t> ptemp(count);
					for(int n = 0; n < count; n++) ptemp[n] = perm[indices[n]];
					for(int n = 0; n < count; n++) perm[first + n] = ptemp[n];
				}
			}
		}

		if(leaf) { // tree.cpp:54-62
			Box b = triBox(tris[first]);
			for(int n = 1; n < count; n++) grow(b, triBox(tris[first + n]));
			setBox(nNode, b);
			depth = std::max(depth, sdepth);
			nodes[nNode].sub = (uint32_t)first | 0x80000000u;
			nodes[nNode].aux = count;
			return;
		}

		Box leftBox = triBox(tris[first]), rightBox = triBox(tris[first + count - 1]);
		for(int n = 1; n < minIdx; n++) grow(leftBox, triBox(tris[first + n]));
		for(int n = minIdx; n < count; n++) grow(rightBox, triBox(tris[first + n]));

		int subNode = nNodes;
		nodes[nNode].sub = (uint32_t)subNode;
		// tree.cpp:148-151 -- the second assignment of firstNode wins
		int firstNode = leftBox.mn[minAxis] == rightBox.mn[minAxis] ? (leftBox.mx[minAxis] < rightBox.mx[minAxis] ? 0 : 1) : 0;
		nodes[nNode].aux = (minAxis & 0xffff) | (firstNode << 16);
		setBox(nNodes++, leftBox);
		setBox(nNodes++, rightBox);

		split(subNode + 0, first, minIdx, sdepth + 1);
		split(subNode + 1, first + minIdx, count - minIdx, sdepth + 1);
	}
};

// ---- packet views ---------------------------------------------------------------------------------
// Vec3q memory layout: per quad {x[4], y[4], z[4]} (12 floats).
struct Rays {
	int size; bool shared;
	const float *origin, *dir, *idir; const uint8_t *mask;
	float O(int q, int c, int l) const { return origin[(shared ? 0 : q) * 12 + c * 4 + l]; }
	float D(int q, int c, int l) const { return dir[q * 12 + c * 4 + l]; }
	float I(int q, int c, int l) const { return idir[q * 12 + c * 4 + l]; }
	V3 org0() const { return V3{origin[0], origin[4], origin[8]}; } // ExtractN(Origin(0), 0)
};

struct Interval { V3 minDir, maxDir, minIDir, maxIDir, minOrigin, maxOrigin; };

// src/rtbase.cpp:61-121 -- three ComputeMinMax overloads; `active(q,l)` selects the variant.
template <class Active>
void computeMinMax(const float *vec, int size, bool anyMask, Active active, V3 *outMin, V3 *outMax) {
	float mn[3][4], mx[3][4];
	int q0 = 0;
	if(!anyMask) {
		for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) mn[c][l] = mx[c][l] = vec[c * 4 + l];
		q0 = 1;
	} else {
		auto anyLane = [&](int q) { return active(q, 0) || active(q, 1) || active(q, 2) || active(q, 3); };
		while(q0 < size && !anyLane(q0)) q0++;
		if(q0 == size) { *outMin = *outMax = V3{0.0f, 0.0f, 0.0f}; return; }
		for(int k = 0; k < 4; k++) if(active(q0, k)) {
			for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) mn[c][l] = mx[c][l] = vec[q0 * 12 + c * 4 + k];
			break;
		}
	}
	for(int q = q0; q < size; q++)
		for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) {
			if(anyMask && !active(q, l)) continue;
			float v = vec[q * 12 + c * 4 + l];
			mn[c][l] = Min(mn[c][l], v);
			mx[c][l] = Max(mx[c][l], v);
		}
	float omn[3], omx[3];
	for(int c = 0; c < 3; c++) { // Minimize/Maximize, src/rtbase_math.h:63-64
		omn[c] = Min(Min(mn[c][0], mn[c][1]), Min(mn[c][2], mn[c][3]));
		omx[c] = Max(Max(mx[c][0], mx[c][1]), Max(mx[c][2], mx[c][3]));
	}
	*outMin = mk(omn); *outMax = mk(omx);
}

// src/ray_group.h:296-333 (RayInterval ctors)
Interval makeInterval(const Rays &r, const float *distMask) {
	Interval i;
	if(distMask) {
		auto act = [&](int q, int l) { return distMask[q * 4 + l] >= 0.0f; };
		computeMinMax(r.dir, r.size, true, act, &i.minDir, &i.maxDir);
		computeMinMax(r.idir, r.size, true, act, &i.minIDir, &i.maxIDir);
		i.minOrigin = i.maxOrigin = r.org0();
		return i;
	}
	auto act = [&](int q, int l) { return r.mask ? ((r.mask[q] >> l) & 1) != 0 : true; };
	bool m = r.mask != nullptr;
	computeMinMax(r.dir, r.size, m, act, &i.minDir, &i.maxDir);
	computeMinMax(r.idir, r.size, m, act, &i.minIDir, &i.maxIDir);
	if(r.shared) i.minOrigin = i.maxOrigin = r.org0();
	else computeMinMax(r.origin, r.size, m, act, &i.minOrigin, &i.maxOrigin);
	return i;
}

// src/bounding_box.cpp:208-236
bool boxTestInterval(const OrcNode &n, const Interval &i) {
	float lmin = 0, lmax = 0;
	for(int k = 0; k < 3; k++) {
		float l1 = i.minIDir[k] * (n.bmin[k] - i.maxOrigin[k]);
		float l2 = i.maxIDir[k] * (n.bmin[k] - i.maxOrigin[k]);
		float l3 = i.minIDir[k] * (n.bmax[k] - i.minOrigin[k]);
		float l4 = i.maxIDir[k] * (n.bmax[k] - i.minOrigin[k]);
		float lo = Min(Min(l1, l2), Min(l3, l4)), hi = Max(Max(l1, l2), Max(l3, l4));
		if(k == 0) { lmin = lo; lmax = hi; }
		else { lmin = Max(lmin, lo); lmax = Min(lmax, hi); }
	}
	return lmax >= 0.0f && lmin <= lmax;
}

// one lane of the slab test; src/bounding_box.cpp:75-99 (primary) and :151-169 (shadow operand order)
template <bool SHADOW>
inline bool lanePasses(const OrcNode &n, const Rays &r, const float *tmin, const float *tmax, int q, int l, float dist) {
	float lmin = 0, lmax = 0;
	for(int k = 0; k < 3; k++) {
		float id = r.I(q, k, l);
		float l1 = id * (r.shared ? tmin[k] : n.bmin[k] - r.O(q, k, l));
		float l2 = id * (r.shared ? tmax[k] : n.bmax[k] - r.O(q, k, l));
		if(k == 0) { lmin = Min(l1, l2); lmax = Max(l1, l2); }
		else if(SHADOW) { lmin = Max(Min(l1, l2), lmin); lmax = Min(Max(l1, l2), lmax); }
		else { lmin = Max(lmin, Min(l1, l2)); lmax = Min(lmax, Max(l1, l2)); }
	}
	if(SHADOW) return lmax >= 0.0f && lmin <= Min(lmax, dist);
	return !(lmax < 0.0f || lmin > Min(lmax, dist));
}

// src/bounding_box.cpp:61-142 and :144-200 -- shrinks [first,last] in place
template <bool SHADOW>
bool boxTest(const OrcNode &n, const Rays &r, const float *dist, int &first, int &last) {
	bool ret = false;
	float tmin[3] = {0, 0, 0}, tmax[3] = {0, 0, 0};
	if(r.shared) {
		V3 o = r.org0();
		for(int k = 0; k < 3; k++) { tmin[k] = n.bmin[k] - o[k]; tmax[k] = n.bmax[k] - o[k]; }
	}
	auto quadPasses = [&](int q) {
		bool any = false;
		for(int l = 0; l < 4; l++) any |= lanePasses<SHADOW>(n, r, tmin, tmax, q, l, dist[q * 4 + l]);
		return any;
	};
	for(int q = first; q <= last; q++) if(quadPasses(q)) { first = q; ret = true; break; }
	for(int q = last; q >= first; q--) if(quadPasses(q)) { last = q; ret = true; break; }
	return ret;
}

// src/triangle.cpp:110-167 with the hard-wired `enum { sharedOrigin = 1 }` branch (:122-129)
bool triTestInterval(const OrcTri &t, const Interval &i) {
	V3 nrm = mk(t.plane);
	float det = (nrm.x < 0.0f ? i.minDir.x : i.maxDir.x) * nrm.x + (nrm.y < 0.0f ? i.minDir.y : i.maxDir.y) * nrm.y +
				(nrm.z < 0.0f ? i.minDir.z : i.maxDir.z) * nrm.z;
	if(det < 0.0f) return true;
	V3 tvec = i.minOrigin - mk(t.a);
	V3 c1 = cross(mk(t.ba), tvec), c2 = cross(tvec, mk(t.ca));
	V3 c1a = i.minDir * c1, c1b = i.maxDir * c1, c2a = i.minDir * c2, c2b = i.maxDir * c2;
	float u0 = Min(c1a.x, c1b.x) + Min(c1a.y, c1b.y) + Min(c1a.z, c1b.z);
	float u1 = Max(c1a.x, c1b.x) + Max(c1a.y, c1b.y) + Max(c1a.z, c1b.z);
	float v0 = Min(c2a.x, c2b.x) + Min(c2a.y, c2b.y) + Min(c2a.z, c2b.z);
	float v1 = Max(c2a.x, c2b.x) + Max(c2a.y, c2b.y) + Max(c2a.z, c2b.z);
	return Min(u1, v1) >= 0.0f && u0 + v0 <= det * t.t0;
}

// src/triangle.cpp:3-63 (closest hit).  Per-lane; the quad-level ForAny() there only skips work.
template <int MODE>
void collidePrimary(const OrcTri &t, const Rays &r, float *dist, int32_t *obj, float *bary, int idx, int first, int last) {
	V3 nrm = mk(t.plane), a = mk(t.a), ba = mk(t.ba), ca = mk(t.ca);
	V3 tvec0{0, 0, 0}, tvec1{0, 0, 0};
	float tmulS = 0;
	if(r.shared) {
		V3 tvec = r.org0() - a;
		tvec0 = cross(ba, tvec) * t.it0;
		tvec1 = cross(tvec, ca) * t.it0;
		tmulS = -dot(tvec, nrm);
	}
	for(int q = first; q <= last; q++)
		for(int l = 0; l < 4; l++) {
			V3 d{r.D(q, 0, l), r.D(q, 1, l), r.D(q, 2, l)};
			float det = dot(d, nrm), u, v, tmul;
			if(r.shared) { v = dot(d, tvec0); u = dot(d, tvec1); tmul = tmulS; }
			else {
				V3 tvec = V3{r.O(q, 0, l), r.O(q, 1, l), r.O(q, 2, l)} - a;
				V3 tv0 = cross(ba, tvec), tv1 = cross(tvec, ca);
				tmul = -dot(tvec, nrm);
				v = dot(d, tv0) * t.it0;
				u = dot(d, tv1) * t.it0;
			}
			float duv = det - u - v;
			float uvmin = Min(u, Min(v, duv)), uvmax = Max(u, Max(v, duv));
			bool test = uvmax <= 0.0f || uvmin >= 0.0f;
			if(r.mask) test = test && ((r.mask[q] >> l) & 1);
			if(!test) continue;
			float idet = Inv<MODE>(det);
			float d2 = idet * tmul;
			if(d2 < dist[q * 4 + l] && d2 > 0.0f) {
				dist[q * 4 + l] = d2;
				obj[q * 4 + l] = idx;
				bary[q * 8 + l] = u * idet;
				bary[q * 8 + 4 + l] = v * idet;
			}
		}
}

// src/triangle.cpp:65-102 (any hit)
bool collideShadow(const OrcTri &t, const Rays &r, float *dist, int first, int last) {
	bool full = (last - first + 1) == r.size;
	V3 nrm = mk(t.plane), a = mk(t.a);
	V3 tvec = r.org0() - a;
	V3 tvec0 = cross(mk(t.ba), tvec) * t.it0, tvec1 = cross(tvec, mk(t.ca)) * t.it0;
	float tmul = -dot(tvec, nrm);
	for(int q = first; q <= last; q++)
		for(int l = 0; l < 4; l++) {
			V3 d{r.D(q, 0, l), r.D(q, 1, l), r.D(q, 2, l)};
			float det = dot(d, nrm), v = dot(d, tvec0), u = dot(d, tvec1);
			bool test = Min(u, v) >= 0.0f && u + v <= det;
			test = test && tmul > 0.0f && tmul < dist[q * 4 + l] * det;
			full &= test;
			if(test) dist[q * 4 + l] = -kInf;
		}
	return full;
}

struct Stats { uint64_t intersects = 0, iters = 0, rays = 0, skips = 0; };

struct StackElem { int node; short first, last; short failsAtPush = 0; /* instrumentation only (g_farHist) */ };

// optional instrumentation (tests/range_hist.py): [0,64) inner visits by (last-first) on entry, [64,128) leaf visits likewise,
// [128,192) leaf visits by (last-first) after the box test.  Single-threaded use only.
uint64_t *g_rangeHist = nullptr;
// optional instrumentation (tests/far_child_hist.py; round 5's question: how many visits would a reject-only test of the FAR child at push time save?):
// per packet kind k (0 = shared origin, 1 = per-ray origins) and width w = last - first of the pushed range, four counters at [(k * 64 + w) * 4 + ..]:
// +0 far-child entries popped, +1 of those whose box test fails at the pop, +2 of those that fail with the distances of the PUSH already (no lane of the
// pushed range passes then: it cannot pass later, distances only shrink), +3 all node visits that start with that range width.  Single-threaded use only.
uint64_t *g_farHist = nullptr;

// src/bvh/traverse.cpp:14-80
template <int MODE>
void traversePrimary(const OrcNode *nodes, const OrcTri *tris, const Rays &r, float *dist, int32_t *obj, float *bary, Stats &st) {
	StackElem stack[64 + 2]; int sp = 0;
	stack[sp++] = StackElem{0, 0, (short)(r.size - 1)};
	int sign[3] = {r.D(0, 0, 0) < 0.0f, r.D(0, 1, 0) < 0.0f, r.D(0, 2, 0) < 0.0f};
	Interval iv = makeInterval(r, nullptr);

	bool root = true;
	while(sp) {
		int nNode = stack[--sp].node, first = stack[sp].first, last = stack[sp].last;
		uint64_t *fh = (g_farHist && !root) ? g_farHist + ((r.shared ? 0 : 64) + (last - first)) * 4 : nullptr;   // this entry is a popped far child
		const bool failedAtPush = stack[sp].failsAtPush != 0;
		root = false;
		if(fh) fh[0]++;
		for(;;) {
			st.iters++;
			const OrcNode &n = nodes[nNode];
			if(g_rangeHist) g_rangeHist[((n.sub & 0x80000000u) ? 64 : 0) + (last - first)]++;
			if(g_farHist) g_farHist[((r.shared ? 0 : 64) + (last - first)) * 4 + 3]++;
			if(fh) {   // the popped node's own test, repeated without side effects
				int f2 = first, l2 = last;
				if(!(boxTestInterval(n, iv) && boxTest<false>(n, r, dist, f2, l2))) { fh[1]++; if(failedAtPush) fh[2]++; }
				fh = nullptr;
			}
			if(n.sub & 0x80000000u) {
				int count = n.aux, firstTri = (int)(n.sub & 0x7fffffffu);
				if(!boxTestInterval(n, iv)) break;
				if(boxTest<false>(n, r, dist, first, last))
					for(int k = 0; k < count; k++) {
						if(g_rangeHist && k == 0) g_rangeHist[128 + (last - first)]++;
						const OrcTri &t = tris[firstTri + k];
						if(r.shared ? triTestInterval(t, iv) : true) {
							collidePrimary<MODE>(t, r, dist, obj, bary, firstTri + k, first, last);
							st.intersects += (uint64_t)(last - first + 1);
						}
					}
				break;
			}
			if(!boxTestInterval(n, iv)) break;
			if(!boxTest<false>(n, r, dist, first, last)) break;
			int child = (int)n.sub, axis = n.aux & 0xffff, firstNode = (n.aux >> 16) ^ sign[axis];
			stack[sp++] = StackElem{child + (firstNode ^ 1), (short)first, (short)last};
			if(g_farHist) {
				const OrcNode &far = nodes[child + (firstNode ^ 1)];
				int f2 = first, l2 = last;
				stack[sp - 1].failsAtPush = !(boxTestInterval(far, iv) && boxTest<false>(far, r, dist, f2, l2));
			}
			nNode = child + firstNode;
		}
	}
}

// src/bvh/traverse.cpp:82-149
void traverseShadow(const OrcNode *nodes, const OrcTri *tris, const Rays &r, float *dist, Stats &st) {
	StackElem stack[64 + 2]; int sp = 0;
	stack[sp++] = StackElem{0, 0, (short)(r.size - 1)};
	int sign[3] = {r.D(0, 0, 0) < 0.0f, r.D(0, 1, 0) < 0.0f, r.D(0, 2, 0) < 0.0f};
	Interval iv = makeInterval(r, dist);

	while(sp) {
		int nNode = stack[--sp].node, first = stack[sp].first, last = stack[sp].last;
		for(;;) {
			st.iters++;
			const OrcNode &n = nodes[nNode];
			if(n.sub & 0x80000000u) {
				int count = n.aux, firstTri = (int)(n.sub & 0x7fffffffu);
				if(!boxTestInterval(n, iv)) break;
				if(boxTest<true>(n, r, dist, first, last))
					for(int k = 0; k < count; k++) {
						const OrcTri &t = tris[firstTri + k];
						if(triTestInterval(t, iv)) {
							if(collideShadow(t, r, dist, first, last)) { st.skips++; return; }
							st.intersects += (uint64_t)(last - first + 1);
						}
					}
				break;
			}
			if(!boxTestInterval(n, iv)) break;
			if(!boxTest<true>(n, r, dist, first, last)) break;
			int child = (int)n.sub, axis = n.aux & 0xffff, firstNode = (n.aux >> 16) ^ sign[axis];
			stack[sp++] = StackElem{child + (firstNode ^ 1), (short)first, (short)last};
			nNode = child + firstNode;
		}
	}
}

// ---- primary ray generator: src/ray_generator.cpp:4-15 (ctor), :23-47 (Generate, level 3) ---------
struct RayGen { V3 tright, tup; float txyz[3][4]; };

RayGen makeRayGen(const OrcCamera &cam, int w, int h) {
	RayGen g;
	float invW = 1.0f / float(w), invH = 1.0f / float(h);
	invW *= float(w) / float(h);
	const float ax[4] = {0.0f, 1.0f, 0.0f, 1.0f}, ay[4] = {0.0f, 0.0f, 1.0f, 1.0f};
	g.tright = mk(cam.right) * invW;
	g.tup = mk(cam.up) * invH;
	V3 fp = mk(cam.front) * cam.plane_dist;
	for(int l = 0; l < 4; l++) {
		float taddx = ax[l] - w * 0.5f, taddy = ay[l] - h * 0.5f;
		g.txyz[0][l] = g.tright.x * taddx + g.tup.x * taddy + fp.x;
		g.txyz[1][l] = g.tright.y * taddx + g.tup.y * taddy + fp.y;
		g.txyz[2][l] = g.tright.z * taddx + g.tup.z * taddy + fp.z;
	}
	return g;
}

template <int MODE>
void genPacket(const RayGen &g, int x, int y, float *dir, float *idir) {
	const float xoff[4] = {float(x + 0), float(x + 0), float(x + 2), float(x + 2)};
	const float yoff[4] = {float(y + 0), float(y + 0), float(y - 1), float(y - 1)};
	for(int ty = 0; ty < 16; ty++)
		for(int k = 0; k < 4; k++) {
			int q = ty * 4 + k;
			for(int l = 0; l < 4; l++) {
				float tposx = float(4 * k) + xoff[l];
				float tposy = float(ty) + yoff[l];
				float px = g.tright.x * tposx + (g.tup.x * tposy + g.txyz[0][l]);
				float py = g.tright.y * tposx + (g.tup.y * tposy + g.txyz[1][l]);
				float pz = g.tright.z * tposx + (g.tup.z * tposy + g.txyz[2][l]);
				float rs = RSqrt<MODE>(px * px + py * py + pz * pz);
				float d[3] = {px * rs, py * rs, pz * rs};
				for(int c = 0; c < 3; c++) {
					dir[q * 12 + c * 4 + l] = d[c];
					idir[q * 12 + c * 4 + l] = safeInv<MODE>(d[c]); // SafeInv, src/rtbase.h:117-120
				}
			}
		}
}

// ---- single-ray accounting walk ------------------------------------------------
template <int MODE>
void accountRay(const OrcNode *nodes, const OrcTri *tris, V3 o, V3 d, V3 id, uint64_t &vn, uint64_t &vt, uint64_t &hits) {
	int stack[66]; int sp = 0;
	stack[sp++] = 0;
	int sign[3] = {d.x < 0.0f, d.y < 0.0f, d.z < 0.0f};
	float dist = kInf;
	while(sp) {
		int nNode = stack[--sp];
		for(;;) {
			const OrcNode &n = nodes[nNode];
			vn++;
			float lmin = 0, lmax = 0;
			for(int k = 0; k < 3; k++) {
				float l1 = id[k] * (n.bmin[k] - o[k]), l2 = id[k] * (n.bmax[k] - o[k]);
				if(k == 0) { lmin = Min(l1, l2); lmax = Max(l1, l2); }
				else { lmin = Max(lmin, Min(l1, l2)); lmax = Min(lmax, Max(l1, l2)); }
			}
			if(lmax < 0.0f || lmin > Min(lmax, dist)) break;
			if(n.sub & 0x80000000u) {
				int count = n.aux, firstTri = (int)(n.sub & 0x7fffffffu);
				for(int k = 0; k < count; k++) {
					const OrcTri &t = tris[firstTri + k];
					vt++;
					V3 nrm = mk(t.plane), tvec = o - mk(t.a);
					V3 tvec0 = cross(mk(t.ba), tvec) * t.it0, tvec1 = cross(tvec, mk(t.ca)) * t.it0;
					float tmul = -dot(tvec, nrm);
					float det = dot(d, nrm), v = dot(d, tvec0), u = dot(d, tvec1), duv = det - u - v;
					float uvmin = Min(u, Min(v, duv)), uvmax = Max(u, Max(v, duv));
					if(!(uvmax <= 0.0f || uvmin >= 0.0f)) continue;
					float t2 = Inv<MODE>(det) * tmul;
					if(t2 < dist && t2 > 0.0f) dist = t2;
				}
				break;
			}
			int child = (int)n.sub, axis = n.aux & 0xffff, firstNode = (n.aux >> 16) ^ sign[axis];
			stack[sp++] = child + (firstNode ^ 1);
			nNode = child + firstNode;
		}
	}
	if(dist < kInf) hits++;
}

template <class F>
void parallelFor(int n, int threads, F f) {
	if(threads <= 1) { for(int i = 0; i < n; i++) f(i, 0); return; }
	std::atomic<int> next{0};
	std::vector<std::thread> pool;
	for(int t = 0; t < threads; t++)
		pool.emplace_back([&, t] { for(int i; (i = next.fetch_add(1)) < n;) f(i, t); });
	for(auto &th : pool) th.join();
}

template <int MODE>
void renderPrimary(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, int x0, int y0, int w, int h,
				   float *ot, float *ou, float *ov, int32_t *oid, uint64_t *stats, int threads) {
	RayGen g = makeRayGen(*cam, resx, resy);
	int pw = (w + 15) / 16, ph = (h + 15) / 16;
	threads = std::max(threads, 1);
	std::vector<Stats> tstats(threads);
	float origin[12];
	for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) origin[c * 4 + l] = cam->pos[c];

	parallelFor(pw * ph, threads, [&](int p, int tid) {
		int px = x0 + (p % pw) * 16, py = y0 + (p / pw) * 16;
		float dir[768], idir[768], dist[256], bary[512];
		int32_t obj[256];
		genPacket<MODE>(g, px, py, dir, idir);
		for(int i = 0; i < 256; i++) { dist[i] = kInf; obj[i] = 0; }   // src/scene_trace.cpp:112-115
		memset(bary, 0, sizeof(bary));
		Rays r{64, true, origin, dir, idir, nullptr};
		Stats st;                                                       // (per packet: neighbouring workers' counters share cache lines)
		st.rays += 256;                                                 // src/scene_trace.cpp:116-117
		traversePrimary<MODE>(nodes, tris, r, dist, obj, bary, st);
		tstats[tid].intersects += st.intersects; tstats[tid].iters += st.iters; tstats[tid].rays += st.rays; tstats[tid].skips += st.skips;
		for(int q = 0; q < 64; q++) {
			int yy = py + (q >> 2);
			if(yy >= resy || yy >= y0 + h) continue;
			for(int l = 0; l < 4; l++) {
				int xx = px + (q & 3) * 4 + l;
				if(xx >= resx || xx >= x0 + w) continue;
				size_t o = (size_t)yy * resx + xx;
				if(ot) ot[o] = dist[q * 4 + l];
				if(ou) ou[o] = bary[q * 8 + l];
				if(ov) ov[o] = bary[q * 8 + 4 + l];
				if(oid) oid[o] = obj[q * 4 + l];
			}
		}
	});
	if(stats) for(auto &s : tstats) { stats[0] += s.intersects; stats[1] += s.iters; stats[2] += s.rays; stats[3] += s.skips; }
}

template <int MODE>
void accountPrimary(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, int x0, int y0, int w, int h,
					uint64_t *out, int threads) {
	RayGen g = makeRayGen(*cam, resx, resy);
	int pw = (w + 15) / 16, ph = (h + 15) / 16;
	threads = std::max(threads, 1);
	std::vector<uint64_t> acc((size_t)threads * 4, 0);
	parallelFor(pw * ph, threads, [&](int p, int tid) {
		int px = x0 + (p % pw) * 16, py = y0 + (p / pw) * 16;
		float dir[768], idir[768];
		genPacket<MODE>(g, px, py, dir, idir);
		uint64_t *a = &acc[(size_t)tid * 4];
		for(int q = 0; q < 64; q++) for(int l = 0; l < 4; l++) {
			V3 d{dir[q * 12 + l], dir[q * 12 + 4 + l], dir[q * 12 + 8 + l]};
			V3 id{idir[q * 12 + l], idir[q * 12 + 4 + l], idir[q * 12 + 8 + l]};
			a[0]++;
			accountRay<MODE>(nodes, tris, mk(cam->pos), d, id, a[1], a[2], a[3]);
		}
	});
	for(int t = 0; t < threads; t++) for(int k = 0; k < 4; k++) out[k] += acc[(size_t)t * 4 + k];
}

// src/funcs.cpp:8-49
float boxPointDistanceSq(V3 mn, V3 mx, V3 p) {
	float sq = 0.0f, delta;
	for(int k = 0; k < 3; k++) {
		if(p[k] < mn[k]) { delta = p[k] - mn[k]; sq += delta * delta; }
		else if(p[k] > mx[k]) { delta = p[k] - mx[k]; sq += delta * delta; }
	}
	return sq;
}

// Scene::RayTrace (src/scene_trace.cpp:86-520), simple-shading configuration, for one packet of 64 quads: shared origin
// (primary) or per-ray origins with lane masks (reflection packets).  `reflections` = gVals[7]; `depth` = cache.reflections.
// Lanes the reference leaves uninitialised or non-finite (direction / origin of lanes that are masked off) are zeros here:
// a masked lane has distance = -inf, so any FINITE value in it is culled by every box test and cannot influence a result.
struct Lighting { const float *lights7; int nLights; const float *ambient, *color; bool reflections; };

template <int MODE>
void rayTracePacket(const OrcNode *nodes, const OrcTri *tris, const Rays &r, const Lighting &L, int depth, float (*outColor)[3] /*256*/, Stats &st) {
	float dist[256], bary[512];
	int32_t obj[256];
	auto maskBit = [&](int q, int l) { return r.mask ? ((r.mask[q] >> l) & 1) != 0 : true; };
	for(int q = 0; q < 64; q++) for(int l = 0; l < 4; l++) {
		dist[q * 4 + l] = maskBit(q, l) ? kInf : -kInf;       // src/scene_trace.cpp:112-115
		obj[q * 4 + l] = 0;
		if(maskBit(q, l)) st.rays++;                          // stats.TracingRays(CountMaskBits(...)), :116-117
	}
	memset(bary, 0, sizeof(bary));
	traversePrimary<MODE>(nodes, tris, r, dist, obj, bary, st);

	// samples (src/scene_trace.cpp:366-379, 397-452; simple_material.h:19-28)
	float pos[256][3], nrm[256][3], sdiff[256][3], sspec[256][3];
	bool hit[256];
	float mnP[3][4], mxP[3][4];
	for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) { mnP[c][l] = kInf; mxP[c][l] = -kInf; }
	for(int q = 0; q < 64; q++) for(int l = 0; l < 4; l++) {
		const int i = q * 4 + l;
		hit[i] = dist[i] < kInf && maskBit(q, l);
		for(int c = 0; c < 3; c++) pos[i][c] = r.D(q, c, l) * dist[i] + r.O(q, c, l);
		if(hit[i]) {
			for(int c = 0; c < 3; c++) { mnP[c][l] = Min(mnP[c][l], pos[i][c]); mxP[c][l] = Max(mxP[c][l], pos[i][c]); }
			const OrcTri &t = tris[obj[i]];
			for(int c = 0; c < 3; c++) nrm[i][c] = t.plane[c];
			V3 d{r.D(q, 0, l), r.D(q, 1, l), r.D(q, 2, l)};
			const float dn = AbsQ(dot(d, mk(nrm[i])));        // Abs(rays.Dir | normal)
			for(int c = 0; c < 3; c++) sdiff[i][c] = sspec[i][c] = L.color[c] * dn;
		} else for(int c = 0; c < 3; c++) { nrm[i][c] = 0.0f; sdiff[i][c] = sspec[i][c] = 0.0f; }
	}

	// reflections (src/scene_trace.cpp:454-466, TraceReflection :603-618, Reflect src/rtbase_math.h:54-58)
	if(L.reflections && depth < 1) {
		std::vector<float> rorg(768, 0.0f), rdir(768, 0.0f), ridir(768, 0.0f);
		uint8_t sel[64];
		bool all = true;
		for(int q = 0; q < 64; q++) {
			sel[q] = 0;
			for(int l = 0; l < 4; l++) {
				const int i = q * 4 + l;
				if(!hit[i]) continue;
				sel[q] |= (uint8_t)(1 << l);
				V3 d{r.D(q, 0, l), r.D(q, 1, l), r.D(q, 2, l)}, n = mk(nrm[i]);
				const V3 rd = reflect(d, n);
				const float f[3] = {rd.x, rd.y, rd.z};
				for(int c = 0; c < 3; c++) {
					rdir[q * 12 + c * 4 + l] = f[c];
					rorg[q * 12 + c * 4 + l] = pos[i][c] + f[c] * 0.001f;
					ridir[q * 12 + c * 4 + l] = safeInv<MODE>(f[c]);
				}
			}
			for(int l = 0; l < 4; l++) if(!hit[q * 4 + l]) for(int c = 0; c < 3; c++) ridir[q * 12 + c * 4 + l] = safeInv<MODE>(0.0f);
			all = all && sel[q] == 15;
		}
		// selector.All() ? RayGroup<0,0> : RayGroup<0,1> -- the same walk either way; a full mask selects nothing away
		Rays rr{64, false, rorg.data(), rdir.data(), ridir.data(), all ? nullptr : sel};
		float (*reflColor)[3] = new float[256][3];
		rayTracePacket<MODE>(nodes, tris, rr, L, depth + 1, reflColor, st);
		for(int i = 0; i < 256; i++)
			if(hit[i]) for(int c = 0; c < 3; c++) sdiff[i][c] = sdiff[i][c] + (reflColor[i][c] - sdiff[i][c]) * 0.3f;
		delete[] reflColor;
	}

	V3 tMin{Min(Min(mnP[0][0], mnP[0][1]), Min(mnP[0][2], mnP[0][3])), Min(Min(mnP[1][0], mnP[1][1]), Min(mnP[1][2], mnP[1][3])),
		   Min(Min(mnP[2][0], mnP[2][1]), Min(mnP[2][2], mnP[2][3]))};
	V3 tMax{Max(Max(mxP[0][0], mxP[0][1]), Max(mxP[0][2], mxP[0][3])), Max(Max(mxP[1][0], mxP[1][1]), Max(mxP[1][2], mxP[1][3])),
		   Max(Max(mxP[2][0], mxP[2][1]), Max(mxP[2][2], mxP[2][3]))};

	float lDiff[256][3], lSpec[256][3];
	for(int i = 0; i < 256; i++) for(int c = 0; c < 3; c++) { lDiff[i][c] = L.ambient[c]; lSpec[i][c] = 0.0f; }

	for(int n = 0; n < L.nLights; n++) {
		const float *LL = L.lights7 + n * 7;
		const V3 lpos = mk(LL), lcol = mk(LL + 3);
		const float radius = LL[6], iRadius = 1.0f / radius, radSq = radius * radius;   // src/light.h:9-13
		if(boxPointDistanceSq(tMin, tMax, lpos) > radSq) continue;

		// Scene::TraceLight (src/scene_trace.cpp:523-601)
		float sdir[768], sidir[768], sdist[256], distance[256], dotv[256];
		memset(sdir, 0, sizeof(sdir)); memset(sidir, 0, sizeof(sidir));
		for(int q = 0; q < 64; q++) {
			bool any = hit[q * 4] || hit[q * 4 + 1] || hit[q * 4 + 2] || hit[q * 4 + 3];
			for(int l = 0; l < 4; l++) {
				const int i = q * 4 + l;
				sdist[i] = -kInf; distance[i] = 0.0f; dotv[i] = 0.0f;
				if(!any || !hit[i]) continue;         // reference: uninitialised / masked; zeros here (see header)
				V3 lv = mk(pos[i]) - lpos;
				if(dot(lv, lv) < 0.0001f) lv = V3{0.0f, 1.0f, 0.0f};
				distance[i] = sqrtf(dot(lv, lv));
				const float inv = Inv<MODE>(distance[i]);
				V3 fl = lv * inv;
				const float f[3] = {fl.x, fl.y, fl.z};
				for(int c = 0; c < 3; c++) { sdir[q * 12 + c * 4 + l] = f[c]; sidir[q * 12 + c * 4 + l] = safeInv<MODE>(f[c]); }
				dotv[i] = dot(mk(nrm[i]), fl);
				if(dotv[i] > 0.0f) { sdist[i] = distance[i] * 0.9999f; st.rays++; }
			}
		}
		float lorg[12];
		for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) lorg[c * 4 + l] = lpos[c];
		Rays sr{64, true, lorg, sdir, sidir, nullptr};
		traverseShadow(nodes, tris, sr, sdist, st);

		for(int i = 0; i < 256; i++) {
			if(!(sdist[i] > 0.0f)) continue;          // += Condition(msk, ...) adds +0 otherwise
			const float atten = attenuation<MODE>(distance[i], iRadius);
			const float diffMul = dotv[i] * atten;
			float specMul = dotv[i];
			specMul *= specMul; specMul *= specMul; specMul *= specMul; specMul *= specMul;
			specMul *= atten;
			const float lc[3] = {lcol.x, lcol.y, lcol.z};
			for(int c = 0; c < 3; c++) { lDiff[i][c] += lc[c] * diffMul; lSpec[i][c] += lc[c] * specMul; }
		}
	}
	for(int i = 0; i < 256; i++)
		for(int c = 0; c < 3; c++)      // diffuse = specular = color * |d.n| (0 for missed lanes); diffuse possibly blended with the reflection
			outColor[i][c] = L.nLights ? sdiff[i][c] * lDiff[i][c] + sspec[i][c] * lSpec[i][c] : sdiff[i][c];
}

template <int MODE>
void whittedPacket(const OrcNode *nodes, const OrcTri *tris, const OrcCamera &cam, const RayGen &g, int px, int py, const Lighting &L,
				   float (*outColor)[3] /*256*/, Stats &st) {
	float origin[12], dir[768], idir[768];
	for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) origin[c * 4 + l] = cam.pos[c];
	genPacket<MODE>(g, px, py, dir, idir);
	Rays r{64, true, origin, dir, idir, nullptr};
	rayTracePacket<MODE>(nodes, tris, r, L, 0, outColor, st);
}

template <int MODE>
void renderWhitted(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, const float *lights7, int nLights,
				   const float *ambient, const float *color, int flags, uint8_t *frame, int pitch, uint64_t *stats, int threads) {
	const Lighting L{lights7, nLights, ambient, color, (flags & 1) != 0};
	const bool aa = (flags & 2) != 0, depthShading = (flags & 4) != 0;
	// gVals[9]: the generator works at twice the resolution (src/render.cpp:60-62)
	RayGen g = makeRayGen(*cam, aa ? resx * 2 : resx, aa ? resy * 2 : resy);
	int pw = (resx + 15) / 16, ph = (resy + 15) / 16;
	threads = std::max(threads, 1);
	std::vector<Stats> tstats(threads);
	// one 16x16 packet through Scene::RayTrace: the light pipeline, or gVals[1]'s depth shading (src/scene_trace.cpp:128-137)
	auto tracePacket = [&](int x, int y, float (*out)[3], Stats &st) {
		if(!depthShading) { whittedPacket<MODE>(nodes, tris, *cam, g, x, y, L, out, st); return; }
		float origin[12], dir[768], idir[768], dist[256], bary[512];
		int32_t obj[256];
		for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) origin[c * 4 + l] = cam->pos[c];
		genPacket<MODE>(g, x, y, dir, idir);
		for(int i = 0; i < 256; i++) { dist[i] = kInf; obj[i] = 0; }
		Rays r{64, true, origin, dir, idir, nullptr};
		st.rays += 256;
		traversePrimary<MODE>(nodes, tris, r, dist, obj, bary, st);
		for(int i = 0; i < 256; i++) { const float d = Inv<MODE>(dist[i]); out[i][0] = d * 20.0f; out[i][1] = d * 250.0f; out[i][2] = d * 2.0f; }
	};
	parallelFor(pw * ph, threads, [&](int p, int tid) {
		int px = (p % pw) * 16, py = (p / pw) * 16;
		float col[256][3];
		if(!aa) tracePacket(px, py, col, tstats[tid]);
		else {
			// 4x antialiasing (src/render.cpp:71-110): four packets of the double-resolution frame, each reduced 2x2 into one quarter of
			// this packet: (row 2r + row 2r+1) * 0.25 per lane, then lane 0 + lane 1 and lane 2 + lane 3 -- in that order
			const int offx[4] = {0, 16, 0, 16}, offy[4] = {0, 0, 16, 16}, coff[4] = {0, 2, 32, 34};
			for(int k = 0; k < 4; k++) {
				float tcol[256][3];
				tracePacket(px * 2 + offx[k], py * 2 + offy[k], tcol, tstats[tid]);
				for(int r = 0; r < 8; r++) for(int h = 0; h < 2; h++) {
					const int q = 8 * r + 2 * h, dq = coff[k] + 4 * r + h;
					for(int i = 0; i < 2; i++) for(int c = 0; c < 3; c++) {
						float v[4];
						for(int l = 0; l < 4; l++) v[l] = (tcol[(q + i) * 4 + l][c] + tcol[(q + i + 4) * 4 + l][c]) * 0.25f;
						col[dq * 4 + i * 2 + 0][c] = v[0] + v[1];
						col[dq * 4 + i * 2 + 1][c] = v[2] + v[3];
					}
				}
			}
		}
		for(int q = 0; q < 64; q++) {
			int yy = py + (q >> 2);
			if(yy >= resy) continue;
			for(int l = 0; l < 4; l++) {
				int xx = px + (q & 3) * 4 + l;
				if(xx >= resx) continue;
				uint8_t *d = frame + (size_t)yy * pitch + (size_t)xx * 3;
				for(int c = 0; c < 3; c++) d[2 - c] = (uint8_t)convChannel(col[q * 4 + l][c]); // r,g,b -> B,G,R
			}
		}
	});
	if(stats) for(auto &s : tstats) { stats[0] += s.intersects; stats[1] += s.iters; stats[2] += s.rays; stats[3] += s.skips; }
}

// Scene::TraceTransparency (src/scene_trace.cpp:620-634) for one primary packet whose hit distances and transparency selector the
// caller supplies: origin[q] = rays.Dir(q) * (distance[q] + 0.001) + rays.Origin(q); the packet goes on as RayGroup<0,0> / <0,1> with the
// SAME dir / idir arrays through RayTrace (:631-633).  Selector lanes without a hit are dropped (only a shaded hit can set transSel,
// :190,306,349); lanes outside the selector carry zeros (see rayTracePacket: masked lanes cannot influence a result).
template <int MODE>
void transparencyPacket(const OrcNode *nodes, const OrcTri *tris, const OrcCamera &cam, const RayGen &g, int px, int py, const float *t /*256*/,
						const uint8_t *sel /*64*/, const Lighting &L, float (*outColor)[3] /*256*/, Stats &st) {
	float dir[768], idir[768];
	genPacket<MODE>(g, px, py, dir, idir);
	std::vector<float> org(768, 0.0f), d2(768, 0.0f), i2(768, 0.0f);
	uint8_t m[64];
	bool all = true;
	for(int q = 0; q < 64; q++) {
		m[q] = 0;
		for(int l = 0; l < 4; l++) {
			const bool on = ((sel[q] >> l) & 1) != 0 && t[q * 4 + l] < kInf;
			if(on) m[q] |= (uint8_t)(1 << l);
			for(int c = 0; c < 3; c++) {
				const float dv = on ? dir[q * 12 + c * 4 + l] : 0.0f;
				d2[q * 12 + c * 4 + l] = dv;
				org[q * 12 + c * 4 + l] = on ? dv * (t[q * 4 + l] + 0.001f) + cam.pos[c] : 0.0f;
				i2[q * 12 + c * 4 + l] = on ? idir[q * 12 + c * 4 + l] : safeInv<MODE>(0.0f);
			}
		}
		all = all && m[q] == 15;
	}
	Rays rr{64, false, org.data(), d2.data(), i2.data(), all ? nullptr : m};
	rayTracePacket<MODE>(nodes, tris, rr, L, 1, outColor, st);      // depth 1: the nested call of the simple-shading configuration
}

uint64_t fnv1a(uint64_t h, const void *p, size_t n) {
	const unsigned char *b = (const unsigned char *)p;
	for(size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
	return h;
}

#include "snail_sse4.inc"

} // namespace

// The checker must not inherit a caller's floating-point control state: a shared library built with -ffast-math sets flush-to-zero /
// denormals-are-zero in MXCSR for the thread that loads it (crtfastmath), and which libraries a Python process loads depends on the host
// CPU.  The path's arithmetic keeps denormals (the device does, and so does the reference under its default environment): every entry point
// runs with the default MXCSR (round to nearest, exceptions masked, no FTZ / DAZ) and restores the caller's on return; worker threads
// inherit the entry point's.
struct FpEnvGuard {
	unsigned old;
	FpEnvGuard() : old(_mm_getcsr()) { _mm_setcsr(0x1f80u); }
	~FpEnvGuard() { _mm_setcsr(old); }
};
extern "C" {

void orc_tris_from_verts(const float *verts, int n, OrcTri *out) {
	FpEnvGuard fpEnv;
	for(int i = 0; i < n; i++) {
		V3 v0 = mk(verts + i * 9), v1 = mk(verts + i * 9 + 3), v2 = mk(verts + i * 9 + 6);
		V3 ba = v1 - v0, ca = v2 - v0;
		V3 nrm = cross(ba, ca);
		float len = sqrtf(dot(nrm, nrm));
		float inv = 1.0f / len;                        // Vec3::operator/=(base), veclib/vec3.h:47-51
		nrm = nrm * inv;
		OrcTri &t = out[i];
		t.a[0] = v0.x; t.a[1] = v0.y; t.a[2] = v0.z;
		t.ba[0] = ba.x; t.ba[1] = ba.y; t.ba[2] = ba.z;
		t.ca[0] = ca.x; t.ca[1] = ca.y; t.ca[2] = ca.z;
		t.t0 = len; t.it0 = 1.0f / len; t.pad = 0;
		t.plane[0] = nrm.x; t.plane[1] = nrm.y; t.plane[2] = nrm.z; t.plane[3] = dot(nrm, v0);
	}
}

int orc_bvh_build(OrcTri *tris, int n, OrcNode *nodes, int *depth, int32_t *perm) {
	FpEnvGuard fpEnv;
	if(n <= 0) return 0;
	if(perm) for(int i = 0; i < n; i++) perm[i] = i;
	Builder b{tris, perm, nodes};
	Box bbox = triBox(tris[0]);
	for(int i = 1; i < n; i++) grow(bbox, triBox(tris[i]));
	b.setBox(0, bbox);
	b.nNodes = 1;
	nodes[0].sub = 0; nodes[0].aux = 0;
	b.split(0, 0, n, 0);
	if(depth) *depth = b.depth;
	return b.nNodes;
}

uint64_t orc_fnv_nodes(const OrcNode *nodes, int n) { return fnv1a(0xcbf29ce484222325ull, nodes, (size_t)n * sizeof(OrcNode)); }
uint64_t orc_fnv_tris(const OrcTri *tris, int n) {
	uint64_t h = 0xcbf29ce484222325ull;
	for(int i = 0; i < n; i++) {
		h = fnv1a(h, &tris[i], 44);
		h = fnv1a(h, (const char *)&tris[i] + 48, 16);
	}
	return h;
}

void orc_gen_packet(const OrcCamera *cam, int resx, int resy, int px, int py, int mode, float *dir, float *idir) {
	FpEnvGuard fpEnv;
	RayGen g = makeRayGen(*cam, resx, resy);
	ORC_BY_MODE(mode, genPacket, g, px, py, dir, idir);
}

void orc_trace_rays(const OrcNode *nodes, const OrcTri *tris, int npackets, int size, int sharedOrigin, const float *origin,
					const float *dir, const float *idir, const uint8_t *mask, float *distance, int32_t *object, float *bary,
					uint64_t *stats, int mode) {
	FpEnvGuard fpEnv;
	Stats st;
	for(int p = 0; p < npackets; p++) {
		size_t qo = (size_t)p * size;
		Rays r{size, sharedOrigin != 0, origin + (sharedOrigin ? (size_t)p * 12 : qo * 12), dir + qo * 12, idir + qo * 12,
			   mask ? mask + qo : nullptr};
		ORC_BY_MODE(mode, traversePrimary, nodes, tris, r, distance + qo * 4, object + qo * 4, bary + qo * 8, st);
	}
	if(stats) { stats[0] += st.intersects; stats[1] += st.iters; stats[2] += st.rays; stats[3] += st.skips; }
}

void orc_trace_shadow(const OrcNode *nodes, const OrcTri *tris, int npackets, int size, const float *origin, const float *dir,
					  const float *idir, float *distance, uint64_t *stats, int mode) {
	FpEnvGuard fpEnv;
	(void)mode; // no approximate operation on the any-hit path (src/triangle.cpp:94-95: no division)
	Stats st;
	for(int p = 0; p < npackets; p++) {
		size_t qo = (size_t)p * size;
		float org[12];
		for(int c = 0; c < 3; c++) for(int l = 0; l < 4; l++) org[c * 4 + l] = origin[p * 3 + c];
		Rays r{size, true, org, dir + qo * 12, idir + qo * 12, nullptr};
		traverseShadow(nodes, tris, r, distance + qo * 4, st);
	}
	if(stats) { stats[0] += st.intersects; stats[1] += st.iters; stats[2] += st.rays; stats[3] += st.skips; }
}

void orc_render_primary(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, int x0, int y0, int w,
						int h, float *t, float *u, float *v, int32_t *triId, uint64_t *stats, int mode, int threads) {
	FpEnvGuard fpEnv;
	ORC_BY_MODE(mode, renderPrimary, nodes, tris, cam, resx, resy, x0, y0, w, h, t, u, v, triId, stats, threads);
}

void orc_render_primary_sse4(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, int x0, int y0, int w,
							 int h, float *t, float *u, float *v, int32_t *triId, uint64_t *stats, int threads) {
	FpEnvGuard fpEnv;
	sse4::renderPrimary(nodes, tris, cam, resx, resy, x0, y0, w, h, t, u, v, triId, stats, threads);
}

void orc_account_primary(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, int x0, int y0, int w,
						 int h, uint64_t *out, int mode, int threads) {
	FpEnvGuard fpEnv;
	ORC_BY_MODE(mode, accountPrimary, nodes, tris, cam, resx, resy, x0, y0, w, h, out, threads);
}

void orc_shade_depth(const float *t, int n, uint8_t *bgr, int mode) {
	FpEnvGuard fpEnv;
	for(int i = 0; i < n; i++) {
		// Condition(tDistance > maxDist(+inf), 0, Inv(tDistance)): the comparison is never true (src/scene_trace.cpp:130)
		float dist = orc_inv(t[i], mode);
		float c[3] = {dist * 20.0f, dist * 250.0f, dist * 2.0f}; // r, g, b
		int q[3];
		for(int k = 0; k < 3; k++) q[k] = convChannel(c[k]); // Trunc(Clamp(..)), src/render.cpp:11-17
		bgr[i * 3 + 0] = (uint8_t)q[2]; bgr[i * 3 + 1] = (uint8_t)q[1]; bgr[i * 3 + 2] = (uint8_t)q[0];
	}
}

void orc_render_whitted(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, const float *lights7, int nLights,
						const float ambient[3], const float color[3], int flags, uint8_t *frame_bgr, int pitch, uint64_t *stats, int mode, int threads) {
	FpEnvGuard fpEnv;
	ORC_BY_MODE(mode, renderWhitted, nodes, tris, cam, resx, resy, lights7, nLights, ambient, color, flags, frame_bgr, pitch, stats, threads);
}

// the `compress` store of RenderTask::Work (src/render.cpp:140-163): planes R, G-R, B-R of a tile, from the interleaved B,G,R frame
void orc_trace_transparency(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, const int32_t *packet_xy, int nPackets,
							const float *t, const uint8_t *sel, const float *lights7, int nLights, const float ambient[3], const float color[3],
							float *out_color, uint64_t *stats, int mode) {
	FpEnvGuard fpEnv;
	const Lighting L{lights7, nLights, ambient, color, false};
	RayGen g = makeRayGen(*cam, resx, resy);
	Stats st;
	for(int p = 0; p < nPackets; p++) {
		float (*col)[3] = (float (*)[3])(out_color + (size_t)p * 768);
		ORC_BY_MODE(mode, transparencyPacket, nodes, tris, *cam, g, packet_xy[p * 2], packet_xy[p * 2 + 1], t + (size_t)p * 256, sel + (size_t)p * 64, L, col, st);
	}
	if(stats) { stats[0] += st.intersects; stats[1] += st.iters; stats[2] += st.rays; stats[3] += st.skips; }
}

void orc_planar_encode_tile(const uint8_t *frame_bgr, int pitch, int x, int y, int w, int h, uint8_t *out) {
	for(int ty = 0; ty < h; ty++)
		for(int tx = 0; tx < w; tx++) {
			const uint8_t *s = frame_bgr + (size_t)(y + ty) * pitch + (size_t)(x + tx) * 3;
			uint8_t *dr = out + (size_t)ty * w + tx, *dg = dr + (size_t)w * h, *db = dg + (size_t)w * h;
			*dr = s[2]; *dg = s[1]; *db = s[0];
			*dg -= *dr; *db -= *dr; // src/render.cpp:157-160
		}
}
// DecompressTask::Work, plane loop (src/compression.cpp:112-141)
void orc_planar_decode_tile(const uint8_t *planes, int x, int y, int w, int h, uint8_t *frame_bgr, int pitch) {
	const uint8_t *srcr = planes, *srcg = planes + (size_t)w * h, *srcb = planes + (size_t)2 * w * h;
	for(int ty = 0; ty < h; ty++) {
		uint8_t *tdst = frame_bgr + (size_t)(y + ty) * pitch + (size_t)x * 3;
		for(int tx = 0; tx < w; tx++) {
			const uint8_t red = srcr[(size_t)ty * w + tx];
			tdst[2] = red; tdst[1] = (uint8_t)(srcg[(size_t)ty * w + tx] + red); tdst[0] = (uint8_t)(srcb[(size_t)ty * w + tx] + red);
			tdst += 3;
		}
	}
}
void orc_debug_range_hist(uint64_t *hist) { g_rangeHist = hist; }
void orc_debug_far_hist(uint64_t *hist512) { g_farHist = hist512; }
unsigned orc_caller_mxcsr(void) { return _mm_getcsr(); } // diagnostics: what the calling thread runs with (0x1f80 = default)
void orc_debug_set_mxcsr(unsigned v) { _mm_setcsr(v); } // tests: put the calling thread into flush-to-zero mode (0x9fc0) and back (0x1f80)
// The named shading expressions on one row of eight floats (qa = in[0..3], qb = in[4..7] as two SSE quads; v1 = (qa, qa<<<1, qa<<<2),
// v2 = (qb, qb<<<1, qb<<<2) as two Vec3q, <<< = lane rotation) -- the same row the reference's veclib evaluates in
// oracle/veclib_probe.cpp `exprs`; word layout documented there.  Words 45.. use the approximate operations (mode).
void orc_veclib_exprs(const float *in, uint32_t *out, int mode) {
	FpEnvGuard fpEnv;
	auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
	float qa[4], qb[4];
	V3 v1[4], v2[4];
	for(int l = 0; l < 4; l++) { qa[l] = in[l]; qb[l] = in[4 + l]; }
	for(int l = 0; l < 4; l++) {
		v1[l] = V3{qa[l], qa[(l + 1) & 3], qa[(l + 2) & 3]};
		v2[l] = V3{qb[l], qb[(l + 1) & 3], qb[(l + 2) & 3]};
	}
	const int m = forWhichLess(qa, qb);
	out[0] = (uint32_t)m | (m ? 16u : 0u) | (m == 15 ? 32u : 0u);       // ForWhich | ForAny << 4 | ForAll << 5
	for(int l = 0; l < 4; l++) {
		out[1 + l] = bits(sqrtf(qa[l]));
		out[5 + l] = bits(AbsQ(qa[l]));
		out[9 + l] = bits(dot(v1[l], v2[l]));
		const V3 cr = cross(v1[l], v2[l]), rf = reflect(v1[l], v2[l]);
		out[13 + l] = bits(cr.x); out[17 + l] = bits(cr.y); out[21 + l] = bits(cr.z);
		out[25 + l] = bits(rf.x); out[29 + l] = bits(rf.y); out[33 + l] = bits(rf.z);
		out[37 + l] = bits(((m >> l) & 1) ? v1[l].x : v2[l].x);             // Condition(qa < qb, v1, v2).x
		out[41 + l] = (uint32_t)convChannel(qa[l]);
		if(mode == ORC_MODE_SSE) {
			out[45 + l] = bits(FastInv<ORC_MODE_SSE>(qa[l]));
			out[49 + l] = bits(attenuation<ORC_MODE_SSE>(qa[l], qb[l]));
			out[53 + l] = bits(safeInv<ORC_MODE_SSE>(qa[l]));
		} else if(mode == ORC_MODE_TABLE) {
			out[45 + l] = bits(FastInv<ORC_MODE_TABLE>(qa[l]));
			out[49 + l] = bits(attenuation<ORC_MODE_TABLE>(qa[l], qb[l]));
			out[53 + l] = bits(safeInv<ORC_MODE_TABLE>(qa[l]));
		} else {
			out[45 + l] = bits(FastInv<ORC_MODE_IEEE>(qa[l]));
			out[49 + l] = bits(attenuation<ORC_MODE_IEEE>(qa[l], qb[l]));
			out[53 + l] = bits(safeInv<ORC_MODE_IEEE>(qa[l]));
		}
	}
}

float orc_inv(float x, int mode) { return mode == ORC_MODE_SSE ? Inv<ORC_MODE_SSE>(x) : mode == ORC_MODE_TABLE ? Inv<ORC_MODE_TABLE>(x) : Inv<ORC_MODE_IEEE>(x); }
float orc_rsqrt(float x, int mode) { return mode == ORC_MODE_SSE ? RSqrt<ORC_MODE_SSE>(x) : mode == ORC_MODE_TABLE ? RSqrt<ORC_MODE_TABLE>(x) : RSqrt<ORC_MODE_IEEE>(x); }
// ORC_MODE_TABLE's data: 3 x 4096 words -- bits of rcpps(1.m), of rsqrtps(1.m) and of rsqrtps(2 x 1.m), index m >> 11 (the layout of the product's
// snail_host_sse_tables / tests/golden/rcp_tables.npz)
void orc_set_tables(const uint32_t *tables12288) {
	memcpy(g_rcpTab, tables12288, sizeof g_rcpTab);
	memcpy(g_rsqTab, tables12288 + 4096, sizeof g_rsqTab);
}
// ... taken from THIS host's instructions (no check of the block structure here: the tests compare rule and instruction)
void orc_tables_of_this_cpu(uint32_t *tables12288) {
	for(uint32_t i = 0; i < 4096; i++) {
		tables12288[i] = f2u(_mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(u2f(0x3f800000u | (i << 11))))));
		tables12288[4096 + i] = f2u(_mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(u2f(0x3f800000u | (i << 11))))));
		tables12288[8192 + i] = f2u(_mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(u2f(0x40000000u | (i << 11))))));
	}
}
// raw look-ups, bit patterns in and out: fn 0 = rcpps, 1 = rsqrtps; by the table rule (table != 0) or by this host's instruction
void orc_raw_approx(int fn, int table, const uint32_t *in, uint32_t *out, int n) {
	for(int i = 0; i < n; i++) {
		const float x = u2f(in[i]);
		out[i] = f2u(table ? (fn == 0 ? rcpTable(x) : rsqrtTable(x)) : (fn == 0 ? _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(x))) : _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(x)))));
	}
}
float orc_min(float a, float b) { return Min(a, b); }
float orc_max(float a, float b) { return Max(a, b); }

} // extern "C"
