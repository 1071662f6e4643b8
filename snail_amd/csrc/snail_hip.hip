// snail_hip.hip -- gfx950 (MI355X / CDNA4) kernels and C-ABI of libsnailhip.so.
//
// Hot path of nadult/Snail re-designed for wave64:
//
//   ONE 16x16 RAY PACKET = ONE WAVEFRONT, ONE SSE QUAD (4 rays) = ONE LANE.
//
// The reference walks the BVH once per packet of 64 quads and keeps packet-wide state: a stack of
// (node, firstActive, lastActive), a child order taken from lane 0 of quad 0, whole-packet interval
// culls, and a [first,last] quad range that every box test shrinks (src/bvh/traverse.cpp:14-80,
// src/bounding_box.cpp:61-142).  Those semantics decide which triangle wins an exact-t tie, so they are
// kept exactly -- and they map onto a CDNA4 wavefront without any divergence:
//   * control flow is wave-uniform: node index, stack pointer, first/last live in SGPRs;
//   * a 32-B node record is ONE scalar load (s_load_dwordx8) through the scalar cache, requested ahead of
//     its use from a copy of the tree laid out for that (child pairs on one 64-B line, see SNAIL_PF_VISIT);
//   * the traversal stack is a VGPR pair (lane i = stack slot i): pop = v_readlane, push = lane-predicated move;
//   * "scan for the first/last quad with a surviving lane" is one __ballot + s_ff1/s_flbit;
//   * per-leaf, the packet-level Triangle::TestInterval cull and the shared-origin terms tvec0/tvec1/tmul
//     are computed with FOUR LANES PER TRIANGLE (one per vector component, the other components read from
//     the quad neighbours through DPP); survivors are broadcast with ds_bpermute and intersected by all
//     lanes -- or, under a narrow quad range, by the range's rays spread one or two per lane.
// No MFMA: this is branchy slab / Moeller-Trumbore work.  No FMA contraction either: every mul/add is
// rounded separately, in the reference's operand order (build with -ffp-contract=off), IEEE divide and
// sqrt (Inv(x)=1/x, RSqrt(x)=1/sqrt(x): veclib's scalar definitions, veclib/vecbase.h:53-55).
//
// Two instantiations of the walk live in each kernel and are selected per packet (wave-uniform):
//   FAST  : all inputs finite -> no NaN can arise in a slab test, so veclib's Min/Max (second operand
//           on NaN) equal v_min_f32/v_max_f32 up to the sign of zero (which no comparison sees), and
//           BBox::TestInterval is implied by the per-lane test (monotonic rounding) and is skipped;
//   EXACT : select-based Min/Max in the reference's operand order + the interval culls, for packets
//           containing non-finite reciprocals (e.g. dir == -1e-8 exactly, src/rtbase.h:117-120).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/snail_hip.h"
#ifdef SNAIL_DEBUG_API
#include "../../include/snail_hip_debug.h"
#endif

// ---------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void snail_set_error(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

#define HIP_TRY(expr)                                                                                     \
	do {                                                                                                  \
		hipError_t e_ = (expr);                                                                           \
		if(e_ != hipSuccess) {                                                                            \
			snail_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
			return 100 + (int)e_;                                                                         \
		}                                                                                                 \
	} while(0)

// ---------------------------------------------------------------------------------------------------
// device code
// ---------------------------------------------------------------------------------------------------
namespace dev {

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4 *scalar_ptr; // constant address space -> s_load

struct Node {
	float bmin[3], bmax[3];
	unsigned sub;
	int aux;
};

__device__ __forceinline__ float asf(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ float asf(int u) { return __int_as_float(u); }

// one 32-B record, wave-uniform index -> s_load_dwordx8.  The byte offset is formed in 32 bits (the node array is
// < 4 GiB) so that the scalar unit spends one shift, not a 64-bit shift/add chain, per fetch.
typedef const __attribute__((address_space(4))) char *scalar_bytes;
__device__ __forceinline__ Node loadNode(const uint4 *nodes, int idx) {
	const unsigned off = (unsigned)idx << 5;
	scalar_ptr p = (scalar_ptr)((scalar_bytes)(unsigned long long)nodes + off);
	u32x4 a = p[0], b = p[1];
	Node n;
	n.bmin[0] = asf(a.x); n.bmin[1] = asf(a.y); n.bmin[2] = asf(a.z);
	n.bmax[0] = asf(a.w); n.bmax[1] = asf(b.x); n.bmax[2] = asf(b.y);
	n.sub = b.z; n.aux = (int)b.w;
	return n;
}

struct Tri {
	float a[3], ba[3], ca[3], t0, it0, n[3];
};
template <class V4> __device__ __forceinline__ Tri unpackTri(V4 r0, V4 r1, V4 r2, V4 r3) {
	Tri t;
	t.a[0] = asf(r0.x); t.a[1] = asf(r0.y); t.a[2] = asf(r0.z);
	t.ba[0] = asf(r0.w); t.ba[1] = asf(r1.x); t.ba[2] = asf(r1.y);
	t.ca[0] = asf(r1.z); t.ca[1] = asf(r1.w); t.ca[2] = asf(r2.x);
	t.t0 = asf(r2.y); t.it0 = asf(r2.z);
	t.n[0] = asf(r3.x); t.n[1] = asf(r3.y); t.n[2] = asf(r3.z);
	return t;
}
__device__ __forceinline__ Tri loadTriVector(const uint4 *tris, int idx) { // per-lane gather, 4 x dwordx4
	const uint4 *p = tris + (size_t)idx * 4;
	return unpackTri(p[0], p[1], p[2], p[3]);
}
__device__ __forceinline__ Tri loadTriScalar(const uint4 *tris, int idx) { // wave-uniform, s_load_dwordx16
	scalar_ptr p = (scalar_ptr)(unsigned long long)(tris + (size_t)idx * 4);
	u32x4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
	return unpackTri(r0, r1, r2, r3);
}

// Arithmetic modes of the walk (selected per packet, wave-uniform):
//   M_EXACT: veclib Min/Max as selects, a<b?a:b / a>b?a:b (veclib/vecbase.h:75-76; minps/maxps of
//            veclib/sse/f32.h:104-105 have the same "second operand on NaN" behaviour), reference operand
//            order, interval culls included.  Always correct; used when a packet holds non-finite values.
//   M_FAST : every input finite => no NaN can arise, Min/Max == v_min_f32/v_max_f32 up to the sign of zero
//            (never observed by a comparison); BBox::TestInterval is implied by the per-lane test
//            (monotonic rounding) and skipped.
//   M_COH  : M_FAST + every ray of the packet has the same idir sign bit per axis, so min(l1,l2)/max(l1,l2)
//            of a slab are known without comparing: near/far planes are picked once per node (scalar XOR-swap).
enum { M_EXACT = 0, M_FAST = 1, M_COH = 2 };

// raw VALU min/max: clang would wrap llvm.minnum in sNaN-quieting canonicalisations (v_max_f32 x,x,x)
// whenever an operand crosses a basic block; the FAST paths only ever see finite values.
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Inv(det) of the hit test (src/triangle.cpp:55; exact IEEE division here and in the oracle).  For every x with biased exponent 1..252
// (2^-126 <= |x| < 2^126) v_rcp_f32 followed by ONE Newton step in two FMAs is bit-identical to the 11-instruction correctly rounded
// division sequence -- checked over all 2^32 inputs on the device (tools/micro/recip_check.hip, snail_debug_recip_check): 0 differences
// inside that range; outside it (denormal or 0 inputs, denormal results, +-inf) the wave takes the full division.
#ifndef SNAIL_FAST_RECIP
#define SNAIL_FAST_RECIP 1
#endif
__device__ __forceinline__ float recipExact(float x) {
#if SNAIL_FAST_RECIP
	const unsigned e = (__float_as_uint(x) & 0x7f800000u) - 0x00800000u;
	if(__builtin_expect(__builtin_amdgcn_ballot_w64(e >= 0x7e000000u) == 0, 1)) {
		float r = __builtin_amdgcn_rcpf(x);
		const float err = __builtin_fmaf(-x, r, 1.0f);
		return __builtin_fmaf(err, r, r);
	}
#endif
	return 1.0f / x;
}

template <int M> __device__ __forceinline__ float Min(float a, float b) {
	if(M == M_EXACT) return a < b ? a : b;
	return vmin(a, b);
}
template <int M> __device__ __forceinline__ float Max(float a, float b) {
	if(M == M_EXACT) return a > b ? a : b;
	return vmax(a, b);
}
// Min(a, Min(b, c)) / Max(a, Max(b, c))
template <int M> __device__ __forceinline__ float Min3(float a, float b, float c) {
	if(M == M_EXACT) { float m = b < c ? b : c; return a < m ? a : m; }
	return vmin3(a, b, c);
}
template <int M> __device__ __forceinline__ float Max3(float a, float b, float c) {
	if(M == M_EXACT) { float m = b > c ? b : c; return a > m ? a : m; }
	return vmax3(a, b, c);
}

// two stack words into lane `laneSel`: the lane select goes through M0 (gfx9 allows one SGPR on the constant
// bus, so value and select cannot both be ordinary SGPRs)
__device__ __forceinline__ void writeLane2(int &vregA, int valueA, int &vregB, int valueB, int laneSel) {
	asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
				 : "+v"(vregA), "+v"(vregB) : "s"(valueA), "s"(valueB), "s"(laneSel) : "m0");
}
__device__ __forceinline__ float readlanef(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float firstlanef(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// wave64 min/max reductions on the VALU with DPP (no LDS round trips): row_shr 1,2,4,8 inside each row of 16,
// then row_bcast:15 / row_bcast:31 across rows; the full result lands in lane 63.  Inputs are finite (FAST paths).
template <bool MAX, int CTRL, int ROWMASK> __device__ __forceinline__ float dppStep(float v, float ident) {
	const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
	return MAX ? vmax(v, o) : vmin(v, o);
}
template <bool MAX> __device__ __forceinline__ float waveReduce(float v) {
	const float ident = MAX ? -__builtin_inff() : __builtin_inff();
	v = dppStep<MAX, 0x111, 0xf>(v, ident); // row_shr:1
	v = dppStep<MAX, 0x112, 0xf>(v, ident); // row_shr:2
	v = dppStep<MAX, 0x114, 0xf>(v, ident); // row_shr:4
	v = dppStep<MAX, 0x118, 0xf>(v, ident); // row_shr:8
	v = dppStep<MAX, 0x142, 0xa>(v, ident); // row_bcast:15 -> rows 1,3
	v = dppStep<MAX, 0x143, 0xc>(v, ident); // row_bcast:31 -> rows 2,3
	return readlanef(v, 63);
}
__device__ __forceinline__ float waveMin(float v) { return waveReduce<false>(v); }
__device__ __forceinline__ float waveMax(float v) { return waveReduce<true>(v); }
// the six reductions of a packet's direction interval (min and max of three components) at once: the same six DPP steps, written as ONE
// instruction each -- v_min_f32_dpp d, d(shifted), d: a lane without a source keeps its value, which is what the identity operand of the
// two-instruction form achieves -- and interleaved over the six registers, so that no step waits for the DPP read-after-write hazard
// (the compiler's form: mov identity, nop, mov_dpp, min = 4 instructions per step and register; 36 instead of 144 per packet)
#ifndef SNAIL_REDUCE6_ASM
#define SNAIL_REDUCE6_ASM 1
#endif
__device__ __forceinline__ void waveReduce6(float (&mn)[3], float (&mx)[3]) {
#if SNAIL_REDUCE6_ASM
#define SNAIL_R6_STEP(CTRL)                                                                                                                 \
	"v_min_f32_dpp %0, %0, %0 " CTRL "\n v_min_f32_dpp %1, %1, %1 " CTRL "\n v_min_f32_dpp %2, %2, %2 " CTRL "\n"                           \
	"v_max_f32_dpp %3, %3, %3 " CTRL "\n v_max_f32_dpp %4, %4, %4 " CTRL "\n v_max_f32_dpp %5, %5, %5 " CTRL "\n"
	asm("s_nop 1\n" SNAIL_R6_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") SNAIL_R6_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
		SNAIL_R6_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") SNAIL_R6_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
		SNAIL_R6_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") SNAIL_R6_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
		: "+v"(mn[0]), "+v"(mn[1]), "+v"(mn[2]), "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]));
#undef SNAIL_R6_STEP
#pragma unroll
	for(int c = 0; c < 3; c++) { mn[c] = readlanef(mn[c], 63); mx[c] = readlanef(mx[c], 63); }
#else
#pragma unroll
	for(int c = 0; c < 3; c++) { mn[c] = waveMin(mn[c]); mx[c] = waveMax(mx[c]); }
#endif
}

__device__ __forceinline__ u64 rangeMask(int first, int last) { return ((2ull << last) - 1ull) & ~((1ull << first) - 1ull); }

// RayInterval (src/ray_group.h:293-338): packet bounds of dir, idir and origin
struct Interval {
	float minDir[3], maxDir[3], minIDir[3], maxIDir[3], minOrg[3], maxOrg[3];
};

struct Counters {
	unsigned intersects, iters, skips;
	unsigned fetched, leaves; // diagnostic (k_primary<.., DIAG>): triangle records fetched, leaf bodies entered; dead code elsewhere
};

// per-lane quad state: 4 rays
struct Quad {
	float d[3][4], id[3][4];
	float dist[4];
};

// ---- ComputeMinMax (src/rtbase.cpp:61-121) ---------------------------------------------------------
// FAST: all values finite -> plain wave reduction (min/max are order independent without NaN).
// EXACT: the reference's sequential fold, per SSE slot, through LDS (NaN makes the fold order visible).
template <bool EXACT, bool MASKED>
__device__ void computeMinMax(const float (&v)[3][4], unsigned act4, int size, int lane, float *lds /*64*12+64 floats*/,
							  float (&outMin)[3], float (&outMax)[3]) {
	const float inf = __builtin_inff();
	if(!EXACT) {
		u64 anyAct = __ballot(act4 != 0);
		float mn[3], mx[3];
#pragma unroll
		for(int c = 0; c < 3; c++) {
			mn[c] = inf; mx[c] = -inf;
#pragma unroll
			for(int l = 0; l < 4; l++)
				if(act4 & (1u << l)) { mn[c] = vmin(mn[c], v[c][l]); mx[c] = vmax(mx[c], v[c][l]); }
		}
		waveReduce6(mn, mx);
#pragma unroll
		for(int c = 0; c < 3; c++) { outMin[c] = anyAct ? mn[c] : 0.0f; outMax[c] = anyAct ? mx[c] : 0.0f; }
		return;
	}
	// EXACT: stage the packet in LDS, lanes 0..11 fold (component c = lane>>2, slot l = lane&3)
	unsigned *ldsMask = (unsigned *)(lds + 64 * 12);
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	if(lane < size) {
#pragma unroll
		for(int c = 0; c < 3; c++)
#pragma unroll
			for(int l = 0; l < 4; l++) lds[lane * 12 + c * 4 + l] = v[c][l];
		ldsMask[lane] = act4;
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	float mn = 0.0f, mx = 0.0f;
	bool none = false;
	if(lane < 12) {
		int c = lane >> 2, l = lane & 3;
		int q = 0;
		if(!MASKED) { mn = mx = lds[c * 4 + l]; q = 1; }
		else {
			while(q < size && ldsMask[q] == 0) q++;
			if(q == size) none = true;
			else {
				int k = __builtin_ctz(ldsMask[q]);
				mn = mx = lds[q * 12 + c * 4 + k];
			}
		}
		if(!none)
			for(; q < size; q++) {
				if(MASKED && !(ldsMask[q] & (1u << l))) continue;
				float x = lds[q * 12 + c * 4 + l];
				mn = mn < x ? mn : x;
				mx = mx > x ? mx : x;
			}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	// Minimize / Maximize across the 4 slots: Min(Min(t0,t1),Min(t2,t3)) (src/rtbase_math.h:63-64)
	bool noneU = __builtin_amdgcn_readfirstlane((int)none) != 0;
	for(int c = 0; c < 3; c++) {
		float a0 = readlanef(mn, c * 4 + 0), a1 = readlanef(mn, c * 4 + 1), a2 = readlanef(mn, c * 4 + 2), a3 = readlanef(mn, c * 4 + 3);
		float b0 = readlanef(mx, c * 4 + 0), b1 = readlanef(mx, c * 4 + 1), b2 = readlanef(mx, c * 4 + 2), b3 = readlanef(mx, c * 4 + 3);
		float m01 = a0 < a1 ? a0 : a1, m23 = a2 < a3 ? a2 : a3;
		float x01 = b0 > b1 ? b0 : b1, x23 = b2 > b3 ? b2 : b3;
		outMin[c] = noneU ? 0.0f : (m01 < m23 ? m01 : m23);
		outMax[c] = noneU ? 0.0f : (x01 > x23 ? x01 : x23);
	}
}

// BBox::TestInterval (src/bounding_box.cpp:208-236): wave-uniform, only needed on the EXACT path
__device__ __forceinline__ bool boxTestInterval(const Node &n, const Interval &i) {
	float lmin = 0.0f, lmax = 0.0f;
#pragma unroll
	for(int k = 0; k < 3; k++) {
		float l1 = i.minIDir[k] * (n.bmin[k] - i.maxOrg[k]);
		float l2 = i.maxIDir[k] * (n.bmin[k] - i.maxOrg[k]);
		float l3 = i.minIDir[k] * (n.bmax[k] - i.minOrg[k]);
		float l4 = i.maxIDir[k] * (n.bmax[k] - i.minOrg[k]);
		float lo = Min<M_EXACT>(Min<M_EXACT>(l1, l2), Min<M_EXACT>(l3, l4));
		float hi = Max<M_EXACT>(Max<M_EXACT>(l1, l2), Max<M_EXACT>(l3, l4));
		if(k == 0) { lmin = lo; lmax = hi; }
		else { lmin = Max<M_EXACT>(lmin, lo); lmax = Min<M_EXACT>(lmax, hi); }
	}
	return lmax >= 0.0f && lmin <= lmax;
}

// Triangle::TestInterval (src/triangle.cpp:110-167, shared-origin branch :122-129); each lane its own triangle.
// Branch-free: (det < 0) | (...) has the same truth value as the reference's early return.
template <int M>
__device__ __forceinline__ bool triTestInterval(const Tri &t, const Interval &i) {
	float det;
	if(M == M_EXACT)
		det = (t.n[0] < 0.0f ? i.minDir[0] : i.maxDir[0]) * t.n[0] + (t.n[1] < 0.0f ? i.minDir[1] : i.maxDir[1]) * t.n[1] +
			  (t.n[2] < 0.0f ? i.minDir[2] : i.maxDir[2]) * t.n[2];
	else // finite operands, minDir <= maxDir: the selected product is the larger of the two (n < 0 flips the order; n == 0 gives zeros
		 // whose sign no comparison below observes) -- two multiplies and a max instead of compare -> SGPR -> select -> multiply
		det = vmax(i.minDir[0] * t.n[0], i.maxDir[0] * t.n[0]) + vmax(i.minDir[1] * t.n[1], i.maxDir[1] * t.n[1]) +
			  vmax(i.minDir[2] * t.n[2], i.maxDir[2] * t.n[2]);
	float tv[3] = {i.minOrg[0] - t.a[0], i.minOrg[1] - t.a[1], i.minOrg[2] - t.a[2]};
	float c1[3] = {t.ba[1] * tv[2] - t.ba[2] * tv[1], t.ba[2] * tv[0] - t.ba[0] * tv[2], t.ba[0] * tv[1] - t.ba[1] * tv[0]};
	float c2[3] = {tv[1] * t.ca[2] - tv[2] * t.ca[1], tv[2] * t.ca[0] - tv[0] * t.ca[2], tv[0] * t.ca[1] - tv[1] * t.ca[0]};
	float c1a[3], c1b[3], c2a[3], c2b[3];
#pragma unroll
	for(int k = 0; k < 3; k++) {
		c1a[k] = i.minDir[k] * c1[k]; c1b[k] = i.maxDir[k] * c1[k];
		c2a[k] = i.minDir[k] * c2[k]; c2b[k] = i.maxDir[k] * c2[k];
	}
	float u0 = Min<M>(c1a[0], c1b[0]) + Min<M>(c1a[1], c1b[1]) + Min<M>(c1a[2], c1b[2]);
	float u1 = Max<M>(c1a[0], c1b[0]) + Max<M>(c1a[1], c1b[1]) + Max<M>(c1a[2], c1b[2]);
	float v0 = Min<M>(c2a[0], c2b[0]) + Min<M>(c2a[1], c2b[1]) + Min<M>(c2a[2], c2b[2]);
	float v1 = Max<M>(c2a[0], c2b[0]) + Max<M>(c2a[1], c2b[1]) + Max<M>(c2a[2], c2b[2]);
	return (det < 0.0f) | ((Min<M>(u1, v1) >= 0.0f) & (u0 + v0 <= det * t.t0));
}

// shared-origin terms of Triangle::Collide (src/triangle.cpp:13-18 / :76-80)
struct TriTerms {
	float t0v[3], t1v[3], tmul;
};
__device__ __forceinline__ TriTerms triTerms(const Tri &t, float ox, float oy, float oz) {
	TriTerms r;
	float tv[3] = {ox - t.a[0], oy - t.a[1], oz - t.a[2]};
	r.t0v[0] = (t.ba[1] * tv[2] - t.ba[2] * tv[1]) * t.it0;
	r.t0v[1] = (t.ba[2] * tv[0] - t.ba[0] * tv[2]) * t.it0;
	r.t0v[2] = (t.ba[0] * tv[1] - t.ba[1] * tv[0]) * t.it0;
	r.t1v[0] = (tv[1] * t.ca[2] - tv[2] * t.ca[1]) * t.it0;
	r.t1v[1] = (tv[2] * t.ca[0] - tv[0] * t.ca[2]) * t.it0;
	r.t1v[2] = (tv[0] * t.ca[1] - tv[1] * t.ca[0]) * t.it0;
	r.tmul = -(tv[0] * t.n[0] + tv[1] * t.n[1] + tv[2] * t.n[2]);
	return r;
}

__device__ __forceinline__ float selLanes(float a, float b, u64 lanesOfB) {
	float r;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(lanesOfB));
	return r;
}
__device__ __forceinline__ float xbar(int byteAddr, float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byteAddr, __float_as_int(x))); }
// ---- the packet-level triangle cull + shared-origin terms, FOUR LANES PER TRIANGLE -----------------------------------------------------
// triTestInterval() and triTerms() above are ~95 VALU instructions that run with one lane per triangle -- at most 4 of 64 lanes in an
// ordinary leaf.  Here lane 4q + c (c = 0, 1, 2) works on COMPONENT c of triangle q (leaves of at most 16 triangles): it loads
// a[c], ba[c], ca[c], n[c] and t0 / it0, forms tv[c] = o[c] - a[c], reads the other two components of tv, ba, ca from its quad neighbours
// through DPP (quad_perm: no LDS, no extra instruction where the read folds into the multiply), and computes component c of both cross
// products, of the scaled terms and of every per-axis product of the interval test; the three-term sums (dot products, u / v bounds)
// are added up in lane 4q in the reference's order ((x + y) + z).  Same operations on the same operands as the one-lane form -- the
// same bits -- at ~50 instructions per leaf instead of ~95, and a lane keeps 4 registers of triangle data (n[c], tvec0[c], tvec1[c];
// tmul in lane 4q) instead of 10.  Returns the lanes 4q whose triangle passes the cull (bit 4q); the survivor's terms are read by
// crossbar from lanes 4q, 4q + 1, 4q + 2.
#ifndef SNAIL_CULL_QUAD
#define SNAIL_CULL_QUAD 7 // bit 0: narrow closest-hit leaves, bit 1: narrow any-hit leaves, bit 2: wide leaves; 0 = one lane per triangle everywhere (A/B measurements)
#endif
template <int CTRL> __device__ __forceinline__ float quadRot(float v) { // lane c of a quad <- lane (c + 1) % 3 [0xC9] or (c + 2) % 3 [0xD2]; lane 3 keeps its own
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
struct QuadTerms {
	float n, t0v, t1v, tmul; // component c = lane & 3 of the triangle (lane >> 2): plane normal, tvec0, tvec1; tmul valid in lane 4q
};
template <int M>
__device__ __forceinline__ u64 cullQuad(const uint4 *__restrict__ tris, int count /* <= 16, uniform */, int firstTri, int lane, const float (&org)[3][4],
										const Interval &iv, QuadTerms &T) {
	const int c = lane & 3;
	const bool active = lane < 4 * count;
	T.n = T.t0v = T.t1v = T.tmul = 0.0f;
	bool pass = false;
	// component c of the packet constants (lane 4q + 3 idles on component 0's)
	const float oc = selLanes(selLanes(org[0][0], org[1][0], 0x2222222222222222ull), org[2][0], 0x4444444444444444ull);
	const float dmn = selLanes(selLanes(iv.minDir[0], iv.minDir[1], 0x2222222222222222ull), iv.minDir[2], 0x4444444444444444ull);
	const float dmx = selLanes(selLanes(iv.maxDir[0], iv.maxDir[1], 0x2222222222222222ull), iv.maxDir[2], 0x4444444444444444ull);
	if(active) {
		const float *rec = (const float *)(tris + (size_t)(firstTri + (lane >> 2)) * 4);
		const float *rc = rec + (c == 3 ? 0 : c);
		const float ac = rc[0], bac = rc[3], cac = rc[6], nc = rc[12];
		const float t0 = rec[9], it0 = rec[10];
		const float tv = oc - ac;
		const float tv1 = quadRot<0xC9>(tv), tv2 = quadRot<0xD2>(tv);
		const float ba1 = quadRot<0xC9>(bac), ba2 = quadRot<0xD2>(bac), ca1 = quadRot<0xC9>(cac), ca2 = quadRot<0xD2>(cac);
		const float c1 = ba1 * tv2 - ba2 * tv1; // (ba x tv)[c]   src/triangle.cpp:13-15, :124-125
		const float c2 = tv1 * ca2 - tv2 * ca1; // (tv x ca)[c]
		T.n = nc; T.t0v = c1 * it0; T.t1v = c2 * it0;
		const float p = tv * nc;
		T.tmul = -((p + quadRot<0xC9>(p)) + quadRot<0xD2>(p)); // lane 4q: -((tv.x n.x + tv.y n.y) + tv.z n.z)
		// Triangle::TestInterval (src/triangle.cpp:110-167), per component, sums in lane 4q
		float m;
		if(M == M_EXACT) m = (nc < 0.0f ? dmn : dmx) * nc;
		else m = vmax(dmn * nc, dmx * nc);
		const float det = (m + quadRot<0xC9>(m)) + quadRot<0xD2>(m);
		const float c1a = dmn * c1, c1b = dmx * c1, c2a = dmn * c2, c2b = dmx * c2;
		const float u0p = Min<M>(c1a, c1b), u1p = Max<M>(c1a, c1b), v0p = Min<M>(c2a, c2b), v1p = Max<M>(c2a, c2b);
		const float u0 = (u0p + quadRot<0xC9>(u0p)) + quadRot<0xD2>(u0p), u1 = (u1p + quadRot<0xC9>(u1p)) + quadRot<0xD2>(u1p);
		const float v0 = (v0p + quadRot<0xC9>(v0p)) + quadRot<0xD2>(v0p), v1 = (v1p + quadRot<0xC9>(v1p)) + quadRot<0xD2>(v1p);
		pass = (c == 0) & ((det < 0.0f) | ((Min<M>(u1, v1) >= 0.0f) & (u0 + v0 <= det * t0)));
	}
	return __builtin_amdgcn_ballot_w64(pass);
}
// ---- leaf of a NARROW closest-hit packet range --------------------------------------------------------------------------------
// A lane holds one SSE quad = 4 rays, so intersecting a triangle costs 4 x 23 VALU instructions whatever the width of [first, last] --
// and at the leaves that range is narrow (atrium frame: 43 % of the leaf bodies see <= 16 quads, 70 % <= 32; stress-1M: 77 % / 94 %;
// tests/range_hist.py).  For width <= 64 / (4 / R) the rays of the range are spread over the wave, R rays per lane (R = 1: lane j <- quad
// first + j / 4, ray j % 4;  R = 2: lane j <- quad first + j / 2, rays 2 (j % 2), 2 (j % 2) + 1), through the LDS crossbar (ds_bpermute_b32: no
// VALU issue, no LDS memory): 16 crossbar reads + 12 (8) selects in, 8 reads + 8 selects out.  Every ray sees exactly the operations of
// the wide form in the same order; the triangles are taken in the same order; lanes past the range's last ray never accept; a leaf
// whose triangles all fail the packet-level cull returns before anything moves.  Not for masked / any-hit / barycentric-tracking
// packets and not for leaves of more than 64 triangles (they keep the wide form).
// (Control flow: the early return and the unconditional write-back are what this compiler can place beside the hand-written node
// loop; a flag-guarded gather or a "nothing was hit" early-out end in "illegal VGPR to SGPR copy" on the loop's operands.)
#ifndef SNAIL_LEAF_COMPACT_SHADOW
#define SNAIL_LEAF_COMPACT_SHADOW 1 // the narrow-range form for any-hit packets too
#endif
#ifndef SNAIL_LEAF_COMPACT_PERRAY
#define SNAIL_LEAF_COMPACT_PERRAY 1 // the narrow-range form for per-ray-origin packets (mirrored / continuation rays)
#endif
#ifndef SNAIL_LEAF_COMPACT
#define SNAIL_LEAF_COMPACT 1 // 0 = every leaf in the wide form (A/B measurements)
#endif
// a hit's triangle index, stored into the caller's per-lane record: an int, or -- inside the hand-written walks -- the same bits in a float: a
// 32-bit INTEGER VGPR value that lives across the loop statements can end up sharing its undefined register (the instruction selector keeps one
// per type and path) with the statements' scalar in / out operands, which this compiler reports as "illegal VGPR to SGPR copy"
__device__ __forceinline__ void setId(int &d, int v) { d = v; }
__device__ __forceinline__ void setId(float &d, int v) { d = __int_as_float(v); }
template <int R> struct NarrowRays {
	float d[3][R], dist[R];
	int tid[R];
};
// lane j's R rays out of the quad lanes (all lanes active)
template <int R> __device__ __forceinline__ void narrowGather(const float (&q)[4], int srcAddr, float (&out)[R]) {
	const float a0 = xbar(srcAddr, q[0]), a1 = xbar(srcAddr, q[1]), a2 = xbar(srcAddr, q[2]), a3 = xbar(srcAddr, q[3]);
	if(R == 1) out[0] = selLanes(selLanes(a0, a1, 0xaaaaaaaaaaaaaaaaull), selLanes(a2, a3, 0xaaaaaaaaaaaaaaaaull), 0xccccccccccccccccull);
	else { out[0] = selLanes(a0, a2, 0xaaaaaaaaaaaaaaaaull); out[R - 1] = selLanes(a1, a3, 0xaaaaaaaaaaaaaaaaull); }
}
template <int R, int M, class TID>
__device__ __forceinline__ void leafSharedNarrow(const uint4 *__restrict__ tris, int count, int firstTri, int lane, int first, int last,
												 const float (&org)[3][4], Quad &Q, TID (&tid)[4], const Interval &iv, Counters &st) {
	constexpr int LPQ = 4 / R;                      // lanes per quad
	const int width = last - first + 1;             // count <= 64: one chunk
	const bool inRange = lane >= first && lane <= last;
	const bool live = lane < width * LPQ;           // this lane holds rays of the range
	st.leaves++;
	st.fetched += (unsigned)count;
	constexpr bool quadCull = (SNAIL_CULL_QUAD & 1) != 0;   // (the caller sends leaves of more than 16 triangles to the wide form then)
	Tri t = {};
	TriTerms tt = {};
	QuadTerms qt = {};
	u64 keep;
	if(quadCull) keep = cullQuad<M>(tris, count, firstTri, lane, org, iv, qt);
	else {
		bool pass = false;
		if(lane < count) {
			t = loadTriVector(tris, firstTri + lane);
			tt = triTerms(t, org[0][0], org[1][0], org[2][0]);
			pass = triTestInterval<M>(t, iv);
		}
		keep = __builtin_amdgcn_ballot_w64(pass);
	}
	if(keep == 0) return;
	NarrowRays<R> N;
	const int srcAddr = (first + lane / LPQ) * 4;   // lanes past the range read some quad's rays and never accept
#pragma unroll
	for(int c = 0; c < 3; c++) narrowGather<R>(Q.d[c], srcAddr, N.d[c]);
	narrowGather<R>(Q.dist, srcAddr, N.dist);
#pragma unroll
	for(int i = 0; i < R; i++) N.tid[i] = -1;
	do {
		const int kb = __builtin_ctzll(keep);
		keep &= keep - 1;
		float nx, ny, nz, ax, ay, az, bx, by, bz, tmul;
		int k;
		if(quadCull) {
			k = kb >> 2;
			nx = xbar(kb * 4, qt.n); ny = xbar(kb * 4 + 4, qt.n); nz = xbar(kb * 4 + 8, qt.n);
			ax = xbar(kb * 4, qt.t0v); ay = xbar(kb * 4 + 4, qt.t0v); az = xbar(kb * 4 + 8, qt.t0v);
			bx = xbar(kb * 4, qt.t1v); by = xbar(kb * 4 + 4, qt.t1v); bz = xbar(kb * 4 + 8, qt.t1v);
			tmul = xbar(kb * 4, qt.tmul);
		} else {
			k = kb;
			nx = xbar(k * 4, t.n[0]); ny = xbar(k * 4, t.n[1]); nz = xbar(k * 4, t.n[2]);
			ax = xbar(k * 4, tt.t0v[0]); ay = xbar(k * 4, tt.t0v[1]); az = xbar(k * 4, tt.t0v[2]);
			bx = xbar(k * 4, tt.t1v[0]); by = xbar(k * 4, tt.t1v[1]); bz = xbar(k * 4, tt.t1v[2]);
			tmul = xbar(k * 4, tt.tmul);
		}
		const int idx = firstTri + k;
		if(live)
#pragma unroll
			for(int i = 0; i < R; i++) { // src/triangle.cpp:44-60
				const float det = N.d[0][i] * nx + N.d[1][i] * ny + N.d[2][i] * nz;
				const float v = N.d[0][i] * ax + N.d[1][i] * ay + N.d[2][i] * az;
				const float u = N.d[0][i] * bx + N.d[1][i] * by + N.d[2][i] * bz;
				const float duv = det - u - v;
				const float uvmin = Min3<M>(u, v, duv), uvmax = Max3<M>(u, v, duv);
				if((uvmax <= 0.0f) | (uvmin >= 0.0f)) {
					const float dd = recipExact(det) * tmul;
					if(dd < N.dist[i] && dd > 0.0f) { N.dist[i] = dd; N.tid[i] = idx; }
				}
			}
		st.intersects += width;
	} while(keep);
	// back to the quad lanes (lanes outside the range read garbage and keep their own values)
	const int q = lane - first;
#pragma unroll
	for(int l = 0; l < 4; l++) {
		const int src = (q * LPQ + l / R) * 4;
		const float nd = xbar(src, N.dist[l % R]);
		const int nt = __builtin_amdgcn_ds_bpermute(src, N.tid[l % R]);
		if(inRange && nt >= 0) { Q.dist[l] = nd; setId(tid[l], nt); }
	}
}

// the same for an any-hit (shadow) packet: a lane's rays carry their distance only (negative = masked, -inf once occluded); never taken when
// the range is the whole packet (the "every quad occluded" early-out of src/bvh/traverse.cpp:117-121 needs the wide form's bookkeeping)
template <int R, int M>
__device__ __forceinline__ void leafSharedNarrowShadow(const uint4 *__restrict__ tris, int count, int firstTri, int lane, int first, int last,
													   const float (&org)[3][4], Quad &Q, const Interval &iv, Counters &st) {
	constexpr int LPQ = 4 / R;
	const float inf = __builtin_inff();
	const int width = last - first + 1;             // count <= 64: one chunk
	const bool inRange = lane >= first && lane <= last;
	const bool live = lane < width * LPQ;
	st.leaves++;
	st.fetched += (unsigned)count;
	constexpr bool quadCull = (SNAIL_CULL_QUAD & 2) != 0;
	Tri t = {};
	TriTerms tt = {};
	QuadTerms qt = {};
	u64 keep;
	if(quadCull) keep = cullQuad<M>(tris, count, firstTri, lane, org, iv, qt);
	else {
		const bool mine = lane < count;
		t = loadTriVector(tris, firstTri + (mine ? lane : 0));
		tt = triTerms(t, org[0][0], org[1][0], org[2][0]);
		keep = __builtin_amdgcn_ballot_w64(mine & triTestInterval<M>(t, iv));
	}
	if(keep == 0) return;
	float nd[3][R], ndist[R];
	const int srcAddr = (first + lane / LPQ) * 4;
#pragma unroll
	for(int c = 0; c < 3; c++) narrowGather<R>(Q.d[c], srcAddr, nd[c]);
	narrowGather<R>(Q.dist, srcAddr, ndist);
	do {
		const int kb = __builtin_ctzll(keep);
		keep &= keep - 1;
		float nx, ny, nz, ax, ay, az, bx, by, bz, tmul;
		if(quadCull) {
			nx = xbar(kb * 4, qt.n); ny = xbar(kb * 4 + 4, qt.n); nz = xbar(kb * 4 + 8, qt.n);
			ax = xbar(kb * 4, qt.t0v); ay = xbar(kb * 4 + 4, qt.t0v); az = xbar(kb * 4 + 8, qt.t0v);
			bx = xbar(kb * 4, qt.t1v); by = xbar(kb * 4 + 4, qt.t1v); bz = xbar(kb * 4 + 8, qt.t1v);
			tmul = xbar(kb * 4, qt.tmul);
		} else {
			const int k = kb;
			nx = xbar(k * 4, t.n[0]); ny = xbar(k * 4, t.n[1]); nz = xbar(k * 4, t.n[2]);
			ax = xbar(k * 4, tt.t0v[0]); ay = xbar(k * 4, tt.t0v[1]); az = xbar(k * 4, tt.t0v[2]);
			bx = xbar(k * 4, tt.t1v[0]); by = xbar(k * 4, tt.t1v[1]); bz = xbar(k * 4, tt.t1v[2]);
			tmul = xbar(k * 4, tt.tmul);
		}
#pragma unroll
		for(int i = 0; i < R; i++) { // src/triangle.cpp:91-98
			const float det = nd[0][i] * nx + nd[1][i] * ny + nd[2][i] * nz;
			const float v = nd[0][i] * ax + nd[1][i] * ay + nd[2][i] * az;
			const float u = nd[0][i] * bx + nd[1][i] * by + nd[2][i] * bz;
			bool test = (Min<M>(u, v) >= 0.0f) & (u + v <= det);
			test = test & (tmul > 0.0f) & (tmul < ndist[i] * det);
			if(live && test) ndist[i] = -inf;
		}
		st.intersects += width;
	} while(keep);
	const int q = lane - first;
#pragma unroll
	for(int l = 0; l < 4; l++) {
		const float d = xbar((q * LPQ + l / R) * 4, ndist[l % R]);
		if(inRange) Q.dist[l] = d;
	}
}

// ---- leaf, shared origin (src/bvh/traverse.cpp:34-56 / :98-124): lanes 0..chunk-1 each take one triangle (packet-level
// cull + shared-origin terms in parallel), survivors are broadcast one by one to the whole packet.  Returns true when a
// shadow packet is fully occluded (the walk ends, src/bvh/traverse.cpp:117-121).
#ifndef SNAIL_LEAF_MASK
#define SNAIL_LEAF_MASK 1 // 0 = every lane computes everything in the leaf (A/B measurements)
#endif
template <bool MASK, bool SHADOW, int M, bool BARY, class TID>
__device__ __forceinline__ bool leafShared(const uint4 *__restrict__ tris, int count, int firstTri, int size, int lane, int first, int last,
										   const float (&org)[3][4], Quad &Q, unsigned mask4, TID (&tid)[4], float (&bu)[4], float (&bv)[4],
										   const Interval &iv, Counters &st) {
	const float inf = __builtin_inff();
	const bool inRange = lane >= first && lane <= last;
	const int width = last - first + 1;
	if(SNAIL_LEAF_COMPACT && !SHADOW && !MASK && !BARY) {
		// (first / last come out of an asm statement with vector outputs too, which makes them divergent in the compiler's eyes: a branch
		// on them would drag every counter into VGPRs)
		const int widthU = __builtin_amdgcn_readfirstlane(width);
		const int countU = __builtin_amdgcn_readfirstlane(count);
		if(widthU <= 32 && countU <= ((SNAIL_CULL_QUAD & 1) ? 16 : 64)) {
			const int firstU = __builtin_amdgcn_readfirstlane(first);
			if(widthU <= 16) leafSharedNarrow<1, M>(tris, countU, firstTri, lane, firstU, firstU + widthU - 1, org, Q, tid, iv, st);
			else leafSharedNarrow<2, M>(tris, countU, firstTri, lane, firstU, firstU + widthU - 1, org, Q, tid, iv, st);
			return false;
		}
	}
	if(SNAIL_LEAF_COMPACT_SHADOW && SHADOW && !MASK && !BARY) {
		const int widthU = __builtin_amdgcn_readfirstlane(width);
		const int countU = __builtin_amdgcn_readfirstlane(count);
#ifndef SNAIL_SHADOW_NARROW2
#define SNAIL_SHADOW_NARROW2 1 // ranges of 17..32 quads with two rays per lane (fits since the four-lane cull: 76 VGPRs)
#endif
		if(widthU <= (SNAIL_SHADOW_NARROW2 ? 32 : 16) && widthU < size && countU <= ((SNAIL_CULL_QUAD & 2) ? 16 : 64)) {
			const int firstU = __builtin_amdgcn_readfirstlane(first);
			if(widthU <= 16) leafSharedNarrowShadow<1, M>(tris, countU, firstTri, lane, firstU, firstU + widthU - 1, org, Q, iv, st);
			else leafSharedNarrowShadow<2, M>(tris, countU, firstTri, lane, firstU, firstU + widthU - 1, org, Q, iv, st);
			return false;
		}
	}
	const u64 curRange = rangeMask(first, last);
	st.leaves++;
	constexpr bool LANE_MASK = SNAIL_LEAF_MASK && !SHADOW;
	// one surviving triangle against the packet's quads (src/triangle.cpp:44-60 / :91-98); returns true when a shadow packet is fully occluded
	auto collide = [&](const float nx, const float ny, const float nz, const float ax, const float ay, const float az, const float bx, const float by, const float bz,
					   const float tmul, const int idx) -> bool {
		bool all4 = true;
		if(!LANE_MASK || inRange)
#pragma unroll
		for(int l = 0; l < 4; l++) {
			const float det = Q.d[0][l] * nx + Q.d[1][l] * ny + Q.d[2][l] * nz;
			const float v = Q.d[0][l] * ax + Q.d[1][l] * ay + Q.d[2][l] * az;
			const float u = Q.d[0][l] * bx + Q.d[1][l] * by + Q.d[2][l] * bz;
			if(SHADOW) { // src/triangle.cpp:91-98
				bool test = (Min<M>(u, v) >= 0.0f) & (u + v <= det);
				test = test & (tmul > 0.0f) & (tmul < Q.dist[l] * det);
				all4 = all4 & test;
				if(inRange && test) Q.dist[l] = -inf;
			} else { // src/triangle.cpp:44-60
				const float duv = det - u - v;
				const float uvmin = Min3<M>(u, v, duv), uvmax = Max3<M>(u, v, duv);
				bool test = ((uvmax <= 0.0f) | (uvmin >= 0.0f)) & inRange;
				if(MASK) test = test & (((mask4 >> l) & 1u) != 0);
				if(test) {
					const float idet = recipExact(det);
					const float dd = idet * tmul;
					if(dd < Q.dist[l] && dd > 0.0f) {
						Q.dist[l] = dd; setId(tid[l], idx);
						if(BARY) { bu[l] = u * idet; bv[l] = v * idet; }
					}
				}
			}
		}
		if(SHADOW) {
			const bool full = width == size && (__builtin_amdgcn_ballot_w64(all4) & curRange) == curRange;
			if(full) { st.skips++; return true; }
		}
		st.intersects += width;
		return false;
	};
	if(SNAIL_CULL_QUAD & 4) {
		// cull and shared-origin terms with four lanes per triangle (cullQuad), 16 triangles at a time (an ordinary leaf holds <= 4);
		// survivors broadcast from lanes 4k .. 4k + 2, in triangle order
		const int countU = __builtin_amdgcn_readfirstlane(count);
		st.fetched += (unsigned)countU;
		for(int base = 0; base < countU; base += 16) {
			QuadTerms qt;
			u64 keep = cullQuad<M>(tris, countU - base < 16 ? countU - base : 16, firstTri + base, lane, org, iv, qt);
			while(keep) {
				const int kb = __builtin_ctzll(keep);
				keep &= keep - 1;
				const float nx = xbar(kb * 4, qt.n), ny = xbar(kb * 4 + 4, qt.n), nz = xbar(kb * 4 + 8, qt.n);
				const float ax = xbar(kb * 4, qt.t0v), ay = xbar(kb * 4 + 4, qt.t0v), az = xbar(kb * 4 + 8, qt.t0v);
				const float bx = xbar(kb * 4, qt.t1v), by = xbar(kb * 4 + 4, qt.t1v), bz = xbar(kb * 4 + 8, qt.t1v);
				const float tmul = xbar(kb * 4, qt.tmul);
				if(collide(nx, ny, nz, ax, ay, az, bx, by, bz, tmul, firstTri + base + (kb >> 2))) return true;
			}
		}
		return false;
	}
	for(int base = 0; base < count; base += 64) {
		const int chunk = count - base < 64 ? count - base : 64;
		st.fetched += (unsigned)chunk;
		const bool mine = lane < chunk;
		// Leaves of more than 16 triangles: one lane per triangle.  Closest-hit packets: only the lanes that own a triangle fetch it and
		// evaluate the cull and the shared-origin terms, and only the quads of the range [first, last] intersect a survivor (EXEC off: the
		// instruction count is the same, the switched lanes are not -- this part is power-limited, profiles/README.md).  Any-hit packets
		// keep every lane on: their test is three compares shorter and the masks cost more than they save.
		Tri t = {};
		TriTerms tt = {};
		bool pass = false;
		if(LANE_MASK) {
			if(mine) {
				t = loadTriVector(tris, firstTri + base + lane);
				tt = triTerms(t, org[0][0], org[1][0], org[2][0]);
				pass = triTestInterval<M>(t, iv);
			}
		} else {
			t = loadTriVector(tris, firstTri + base + (mine ? lane : 0));
			tt = triTerms(t, org[0][0], org[1][0], org[2][0]);
			pass = mine & triTestInterval<M>(t, iv);
		}
		u64 keep = __builtin_amdgcn_ballot_w64(pass);

		while(keep) {
			const int k = __builtin_ctzll(keep);
			keep &= keep - 1;
			// broadcast of lane k's triangle through the LDS crossbar (ds_bpermute_b32, no LDS memory) into VGPRs: ten v_readlane_b32
			// into SGPRs cost 4 VALU issue cycles each plus the SGPR-write -> VALU-read hazard.  Same box, tools/exp_leaf.sh: atrium
			// 23.5 vs 22.1 Grays/s, stress-1M 10.2 vs 10.4 (there the ~100 cycles of crossbar latency per survivor show); issuing the
			// next survivor's broadcast ahead of the current intersection (two register sets) costs more than it hides: 21.0 / 9.5.
#ifdef SNAIL_EXP_LEAF_READLANE // experiment hook (tools/exp_leaf.sh)
#define BCAST(x) readlanef(x, k)
#else
#define BCAST(x) __int_as_float(__builtin_amdgcn_ds_bpermute(k * 4, __float_as_int(x)))
#endif
			const float nx = BCAST(t.n[0]), ny = BCAST(t.n[1]), nz = BCAST(t.n[2]);
			const float ax = BCAST(tt.t0v[0]), ay = BCAST(tt.t0v[1]), az = BCAST(tt.t0v[2]);
			const float bx = BCAST(tt.t1v[0]), by = BCAST(tt.t1v[1]), bz = BCAST(tt.t1v[2]);
			const float tmul = BCAST(tt.tmul);
#undef BCAST
			if(collide(nx, ny, nz, ax, ay, az, bx, by, bz, tmul, firstTri + base + k)) return true;
		}
	}
	return false;
}

// ---- leaf, per-ray origins (src/triangle.cpp:30-38; no packet-level triangle cull: src/bvh/traverse.cpp:44): wave-uniform
// scalar triangle fetch, every lane does the full Collide arithmetic for its 4 rays
// Measured and not taken (round 2, profiles/README.md): computing det, tv, tmul of all four rays first and skipping u, v and the inside test
// when no live lane of the packet can accept (det, tmul of equal sign and |tmul| <= dist |det| (1 + 1e-5) + 1e-30: exact by construction)
// -- mirrored bounce 1.005 vs 0.940 ms per frame: too few leaf triangles are rejected by every ray, and the kernel goes from 79 to 95 VGPRs.
// the narrow-range form of the per-ray-origin leaf (one ray per lane): here EVERY triangle of the leaf is intersected by every ray of the
// range (no packet-level cull), ~47 VALU instructions per ray and triangle -- 4 x 47 per lane in the wide form whatever the range's width
template <int R, bool MASK, int M>
__device__ __forceinline__ void leafPerRayNarrow(const uint4 *__restrict__ tris, int count, int firstTri, int lane, int first, int last,
												 const float (&org)[3][4], Quad &Q, unsigned mask4, int (&tid)[4], Counters &st) {
	constexpr int LPQ = 4 / R;                       // lanes per quad: R = 1: lane j <- quad first + j / 4, ray j % 4;  R = 2: quad first + j / 2, rays 2 (j % 2), + 1
	const int width = last - first + 1;
	const bool inRange = lane >= first && lane <= last;
	const int srcAddr = (first + lane / LPQ) * 4;    // lanes past the range read some quad's rays and never accept
	// (the crossbar read first, by EVERY lane: behind `live &&` it would run with the other lanes switched off, and a switched-off source
	// lane reads as 0)
	const unsigned quadMask = MASK ? (unsigned)__builtin_amdgcn_ds_bpermute(srcAddr, (int)mask4) : 15u;
	bool live[R];
#pragma unroll
	for(int i = 0; i < R; i++) live[i] = (lane < width * LPQ) & (((quadMask >> ((lane % LPQ) * R + i)) & 1u) != 0);
	float no[3][R], nd[3][R], ndist[R];
	int ntid[R];
#pragma unroll
	for(int c = 0; c < 3; c++) { narrowGather<R>(org[c], srcAddr, no[c]); narrowGather<R>(Q.d[c], srcAddr, nd[c]); }
	narrowGather<R>(Q.dist, srcAddr, ndist);
#pragma unroll
	for(int i = 0; i < R; i++) ntid[i] = -1;
	st.leaves++; st.fetched += (unsigned)count;
	for(int k = 0; k < count; k++) {
		const Tri t = loadTriScalar(tris, firstTri + k);
#pragma unroll
		for(int i = 0; i < R; i++) {
			const float det = nd[0][i] * t.n[0] + nd[1][i] * t.n[1] + nd[2][i] * t.n[2];
			float tv[3] = {no[0][i] - t.a[0], no[1][i] - t.a[1], no[2][i] - t.a[2]};
			float c0[3] = {t.ba[1] * tv[2] - t.ba[2] * tv[1], t.ba[2] * tv[0] - t.ba[0] * tv[2], t.ba[0] * tv[1] - t.ba[1] * tv[0]};
			float c1[3] = {tv[1] * t.ca[2] - tv[2] * t.ca[1], tv[2] * t.ca[0] - tv[0] * t.ca[2], tv[0] * t.ca[1] - tv[1] * t.ca[0]};
			const float tmul = -(tv[0] * t.n[0] + tv[1] * t.n[1] + tv[2] * t.n[2]);
			const float v = (nd[0][i] * c0[0] + nd[1][i] * c0[1] + nd[2][i] * c0[2]) * t.it0;
			const float u = (nd[0][i] * c1[0] + nd[1][i] * c1[1] + nd[2][i] * c1[2]) * t.it0;
			const float duv = det - u - v;
			const float uvmin = Min3<M>(u, v, duv), uvmax = Max3<M>(u, v, duv);
			if(((uvmax <= 0.0f) | (uvmin >= 0.0f)) & live[i]) {
				const float dd = recipExact(det) * tmul;
				if(dd < ndist[i] && dd > 0.0f) { ndist[i] = dd; ntid[i] = firstTri + k; }
			}
		}
		st.intersects += width;
	}
	const int q = lane - first;
#pragma unroll
	for(int l = 0; l < 4; l++) {
		const int src = (q * LPQ + l / R) * 4;
		const float d = xbar(src, ndist[l % R]);
		const int nt = __builtin_amdgcn_ds_bpermute(src, ntid[l % R]);
		if(inRange && nt >= 0) { Q.dist[l] = d; tid[l] = nt; }
	}
}
#ifndef SNAIL_PERRAY_NARROW2
#define SNAIL_PERRAY_NARROW2 1 // per-ray-origin leaves of ranges of 17..32 quads with two rays per lane (94 instead of 188 vector instructions per triangle)
#endif
template <bool MASK, int M, bool BARY>
__device__ __forceinline__ void leafPerRay(const uint4 *__restrict__ tris, int count, int firstTri, int lane, int first, int last, const float (&org)[3][4],
										   Quad &Q, unsigned mask4, int (&tid)[4], float (&bu)[4], float (&bv)[4], Counters &st) {
	const bool inRange = lane >= first && lane <= last;
	const int width = last - first + 1;
	if(SNAIL_LEAF_COMPACT_PERRAY && !BARY) {
		const int widthU = __builtin_amdgcn_readfirstlane(width);
		if(widthU <= (SNAIL_PERRAY_NARROW2 ? 32 : 16)) {
			const int firstU = __builtin_amdgcn_readfirstlane(first);
			if(widthU <= 16) leafPerRayNarrow<1, MASK, M>(tris, __builtin_amdgcn_readfirstlane(count), firstTri, lane, firstU, firstU + widthU - 1, org, Q, mask4, tid, st);
			else leafPerRayNarrow<2, MASK, M>(tris, __builtin_amdgcn_readfirstlane(count), firstTri, lane, firstU, firstU + widthU - 1, org, Q, mask4, tid, st);
			return;
		}
	}
	st.leaves++; st.fetched += (unsigned)count;
	for(int k = 0; k < count; k++) {
		const Tri t = loadTriScalar(tris, firstTri + k);
#pragma unroll
		for(int l = 0; l < 4; l++) {
			const float det = Q.d[0][l] * t.n[0] + Q.d[1][l] * t.n[1] + Q.d[2][l] * t.n[2];
			float tv[3] = {org[0][l] - t.a[0], org[1][l] - t.a[1], org[2][l] - t.a[2]};
			float c0[3] = {t.ba[1] * tv[2] - t.ba[2] * tv[1], t.ba[2] * tv[0] - t.ba[0] * tv[2], t.ba[0] * tv[1] - t.ba[1] * tv[0]};
			float c1[3] = {tv[1] * t.ca[2] - tv[2] * t.ca[1], tv[2] * t.ca[0] - tv[0] * t.ca[2], tv[0] * t.ca[1] - tv[1] * t.ca[0]};
			const float tmul = -(tv[0] * t.n[0] + tv[1] * t.n[1] + tv[2] * t.n[2]);
			const float v = (Q.d[0][l] * c0[0] + Q.d[1][l] * c0[1] + Q.d[2][l] * c0[2]) * t.it0;
			const float u = (Q.d[0][l] * c1[0] + Q.d[1][l] * c1[1] + Q.d[2][l] * c1[2]) * t.it0;
			const float duv = det - u - v;
			const float uvmin = Min3<M>(u, v, duv), uvmax = Max3<M>(u, v, duv);
			bool test = ((uvmax <= 0.0f) | (uvmin >= 0.0f)) & inRange;
			if(MASK) test = test & (((mask4 >> l) & 1u) != 0);
			if(test) {
				const float idet = recipExact(det);
				const float dd = idet * tmul;
				if(dd < Q.dist[l] && dd > 0.0f) {
					Q.dist[l] = dd; tid[l] = firstTri + k;
					if(BARY) { bu[l] = u * idet; bv[l] = v * idet; }
				}
			}
		}
		st.intersects += width;
	}
}

// ---- the packet walk -------------------------------------------------------------------------------
// SHARED : one origin per packet (primary / shadow)   MASK : per-lane 4-bit masks (secondary rays)
// SHADOW : any-hit TraverseShadow                      M    : arithmetic mode (above)
// BARY   : keep barycentrics up to date in registers (else the caller derives them from the final triId)
// DEEP   : scene depth > 62, stack slots >= 64 live in a second VGPR pair
// DISTPOS: every lane's distance is >= 0 on entry (primary packets) -> single-compare slab test
// oct    : M_COH only: the packet's sign octant from classify() (bit k = idir negative on axis k)
// Stack: lane i of (stkNode, stkFL) is slot i.
template <bool SHARED, bool MASK, bool SHADOW, int M, bool BARY, bool DEEP, bool DISTPOS>
__device__ __forceinline__ void walk(const uint4 *__restrict__ nodes, const uint4 *__restrict__ tris, int size, int lane,
									 const float (&org)[3][4] /* SHARED: [c][0] uniform */, Quad &Q, unsigned mask4, int (&tid)[4],
									 float (&bu)[4], float (&bv)[4], float *lds, Counters &st, const int oct = 0) {
	constexpr bool EXACT = M == M_EXACT;
	Interval iv;
	{ // RayInterval ctor (src/ray_group.h:296-333)
		unsigned act4 = lane < size ? 15u : 0u;
		if(SHADOW) {
			act4 = 0;
			if(lane < size)
#pragma unroll
				for(int l = 0; l < 4; l++) act4 |= (Q.dist[l] >= 0.0f ? 1u : 0u) << l;
		} else if(MASK) act4 = lane < size ? (mask4 & 15u) : 0u;
		computeMinMax<EXACT, (MASK || SHADOW)>(Q.d, act4, size, lane, lds, iv.minDir, iv.maxDir);
		if(EXACT) computeMinMax<EXACT, (MASK || SHADOW)>(Q.id, act4, size, lane, lds, iv.minIDir, iv.maxIDir);   // only BBox::TestInterval reads it
		else {
#pragma unroll
			for(int k = 0; k < 3; k++) iv.minIDir[k] = iv.maxIDir[k] = 0.0f;
		}
		if(SHARED) {
#pragma unroll
			for(int k = 0; k < 3; k++) iv.minOrg[k] = iv.maxOrg[k] = org[k][0];
		} else computeMinMax<EXACT, MASK>(org, act4, size, lane, lds, iv.minOrg, iv.maxOrg);
	}

	// child order from lane 0 of quad 0 (src/bvh/traverse.cpp:21)
	const int signBits = __builtin_amdgcn_readfirstlane((Q.d[0][0] < 0.0f ? 1 : 0) | (Q.d[1][0] < 0.0f ? 2 : 0) | (Q.d[2][0] < 0.0f ? 4 : 0));
	// M_COH: the packet's sign octant picks the near/far slab plane per axis (0 or -1 per axis, for the scalar XOR-swap below)
	const int octMask[3] = {-(oct & 1), -((oct >> 1) & 1), -((oct >> 2) & 1)};

	int stkNode = 0, stkFL = 0, stkNode2 = 0, stkFL2 = 0;
	int sp = 0;
	int first = 0, last = size - 1;
	Node n = loadNode(nodes, 0);

	for(;;) {
		st.iters++;
		const bool isLeaf = (n.sub & 0x80000000u) != 0;
		// children are adjacent (src/bvh/tree.cpp:153-157); near = firstNode ^ sign[axis] (src/bvh/traverse.cpp:71-74).
		// The near child's record is requested here (for a leaf: index 0, the root, always valid); the compiler is free to sink
		// the load into the descend branch and does -- measured either way, the waves wait for the VALU pipe, not for this load.
		const int axis = n.aux & 0xffff;
		const int firstNode = ((n.aux >> 16) ^ (signBits >> axis)) & 1;
		const int nearIdx = isLeaf ? 0 : (int)n.sub + firstNode;
		const int farIdx = (int)n.sub + (firstNode ^ 1);
		const Node nn = loadNode(nodes, nearIdx);

		// ---- BBox::TestInterval + BBox::Test (src/bounding_box.cpp:208-236, :61-142 / :144-200) ----
		u64 passMask = 0; // quads with a surviving lane (before clipping to [first,last])
		if(EXACT) {
			bool anyPass = false;
			if(boxTestInterval(n, iv)) {
				float tmn[3], tmx[3];
				if(SHARED) {
#pragma unroll
					for(int k = 0; k < 3; k++) { tmn[k] = n.bmin[k] - org[k][0]; tmx[k] = n.bmax[k] - org[k][0]; }
				}
#pragma unroll
				for(int l = 0; l < 4; l++) {
					float lmin = 0.0f, lmax = 0.0f;
#pragma unroll
					for(int k = 0; k < 3; k++) {
						float l1 = Q.id[k][l] * (SHARED ? tmn[k] : n.bmin[k] - org[k][l]);
						float l2 = Q.id[k][l] * (SHARED ? tmx[k] : n.bmax[k] - org[k][l]);
						float lo = Min<M_EXACT>(l1, l2), hi = Max<M_EXACT>(l1, l2);
						if(k == 0) { lmin = lo; lmax = hi; }
						else if(SHADOW) { lmin = Max<M_EXACT>(lo, lmin); lmax = Min<M_EXACT>(hi, lmax); }
						else { lmin = Max<M_EXACT>(lmin, lo); lmax = Min<M_EXACT>(lmax, hi); }
					}
					bool pass = SHADOW ? (lmax >= 0.0f && lmin <= Min<M_EXACT>(lmax, Q.dist[l])) : !(lmax < 0.0f || lmin > Min<M_EXACT>(lmax, Q.dist[l]));
					anyPass |= pass;
				}
			}
			passMask = __builtin_amdgcn_ballot_w64(anyPass);
		} else {
			// finite inputs: lane passes  <=>  lmax >= 0  &&  lmin <= lmax  &&  lmin <= dist   (both flavours)
			float tn[4], tf[4];
			if(M == M_COH) {
				// near/far plane per axis by the packet's sign octant, as a scalar XOR-swap on the bit patterns
				// (stays on the SALU: no VALU select, no branch): near = neg ? bmax : bmin, far = the other one
				float pn[3], pf[3];
#pragma unroll
				for(int k = 0; k < 3; k++) {
					const int lo = __float_as_int(n.bmin[k]), hi = __float_as_int(n.bmax[k]);
					const int sw = (lo ^ hi) & octMask[k];
					pn[k] = __int_as_float(lo ^ sw); pf[k] = __int_as_float(hi ^ sw);
					if(SHARED) { pn[k] = pn[k] - org[k][0]; pf[k] = pf[k] - org[k][0]; }
				}
#pragma unroll
				for(int l = 0; l < 4; l++) {
					float lo[3], hi[3];
#pragma unroll
					for(int k = 0; k < 3; k++) {
						lo[k] = Q.id[k][l] * (SHARED ? pn[k] : pn[k] - org[k][l]);
						hi[k] = Q.id[k][l] * (SHARED ? pf[k] : pf[k] - org[k][l]);
					}
					tn[l] = vmax3(lo[0], lo[1], lo[2]);
					tf[l] = vmin3(hi[0], hi[1], hi[2]);
				}
			} else {
				float pn[3], pf[3];
#pragma unroll
				for(int k = 0; k < 3; k++) { pn[k] = SHARED ? n.bmin[k] - org[k][0] : n.bmin[k]; pf[k] = SHARED ? n.bmax[k] - org[k][0] : n.bmax[k]; }
#pragma unroll
				for(int l = 0; l < 4; l++) {
					float lo[3], hi[3];
#pragma unroll
					for(int k = 0; k < 3; k++) {
						const float a = Q.id[k][l] * (SHARED ? pn[k] : pn[k] - org[k][l]);
						const float b = Q.id[k][l] * (SHARED ? pf[k] : pf[k] - org[k][l]);
						lo[k] = vmin(a, b); hi[k] = vmax(a, b);
					}
					tn[l] = vmax3(lo[0], lo[1], lo[2]);
					tf[l] = vmin3(hi[0], hi[1], hi[2]);
				}
			}
			// all compares last: one VALU->SALU hand-over per node instead of twelve
			if(DISTPOS) { // dist >= 0 on every lane (primary packets): lmax>=0 && lmin<=lmax && lmin<=dist  <=>  max(lmin,0) <= min(lmax,dist)
#pragma unroll
				for(int l = 0; l < 4; l++) { tn[l] = vmax(tn[l], 0.0f); tf[l] = vmin(tf[l], Q.dist[l]); }
				// any lane with tn <= tf  <=>  max_l (tf_l - tn_l) >= 0: both operands are finite here and fp32 denormals are on
				// (.amdhsa_float_denorm_mode_32 3), so the sign of the difference is exact -- ONE compare, nothing for the scalar unit to OR
				const float slack = vmax3(tf[0] - tn[0], tf[1] - tn[1], vmax(tf[2] - tn[2], tf[3] - tn[3]));
				passMask = __builtin_amdgcn_ballot_w64(slack >= 0.0f);
			} else {
				// any distance (masked lanes: -inf): lmax>=0 && lmin<=lmax && lmin<=dist  <=>  min(min(lmax,dist) - lmin, lmax) >= 0
				// (lmin, lmax finite; min(lmax,dist) finite or -inf, so the difference is exact in sign or -inf)
				float sl[4];
#pragma unroll
				for(int l = 0; l < 4; l++) sl[l] = vmin(vmin(tf[l], Q.dist[l]) - tn[l], tf[l]);
				const float slack = vmax3(sl[0], sl[1], vmax(sl[2], sl[3]));
				passMask = __builtin_amdgcn_ballot_w64(slack >= 0.0f);
			}
		}
		// clip to the quad range [first,last]: (lane - first) <= (last - first) as ONE unsigned VALU compare -- the scalar unit
		// (shared by the CU's four SIMDs, and the busiest unit of this kernel) would need five instructions for the same mask
		const u64 alive = passMask & __builtin_amdgcn_ballot_w64((unsigned)(lane - first) <= (unsigned)(last - first));
		if(alive != 0) {
			first = __builtin_ctzll(alive);
			last = 63 - __builtin_clzll(alive);
			if(!isLeaf) {
				const int fl = first | (last << 8);
				// push: v_writelane_b32 (no clang builtin).  Both the value and the lane select are SALU-produced SGPRs,
				// so none of the VALU->v_writelane hazards of the ISA applies.
				if(!DEEP || sp < 64) writeLane2(stkNode, farIdx, stkFL, fl, sp);
				else writeLane2(stkNode2, farIdx, stkFL2, fl, sp - 64);
				sp++;
				n = nn;
				continue;
			}
		// ---- leaf (src/bvh/traverse.cpp:34-56 / :98-124) ----
		const int count = n.aux, firstTri = (int)(n.sub & 0x7fffffffu);
		if(SHARED) {
			if(leafShared<MASK, SHADOW, M, BARY>(tris, count, firstTri, size, lane, first, last, org, Q, mask4, tid, bu, bv, iv, st)) return;
		} else {
			leafPerRay<MASK, M, BARY>(tris, count, firstTri, lane, first, last, org, Q, mask4, tid, bu, bv, st);
		}
		} // alive != 0
		// ---- pop (src/bvh/traverse.cpp:26-30) ----
		if(sp == 0) break;
		sp--;
		int cur, fl;
		if(!DEEP || sp < 64) { cur = __builtin_amdgcn_readlane(stkNode, sp); fl = __builtin_amdgcn_readlane(stkFL, sp); }
		else { cur = __builtin_amdgcn_readlane(stkNode2, sp - 64); fl = __builtin_amdgcn_readlane(stkFL2, sp - 64); }
		first = fl & 0xff; last = fl >> 8;
		n = loadNode(nodes, cur);
	}
}

// ---- the node loop in assembly -------------------------------------------------------------------------------------------
// Same algorithm and the same IEEE operations as dev::walk in its M_COH / M_FAST modes for trees of depth <= 62; only the descend /
// cull / push / pop loop is written by hand, because the compiler's version of it carries register shuffling (loop-carried copies
// around the stack VGPRs and the node SGPRs, hazard padding around its inline v_min/v_max): 56 VALU + ~24 scalar instructions per
// inner node of a coherent shared-origin packet, against ~103 for the compiler at the time (profiles/README.md).  One asm statement
// = "pop, then descend until a leaf survives its box test (-> the C++ leaf code) or the stack is empty".  Inside the loop EXEC is the
// quad range [first,last] (s_bfm_b64 + s_bitset1_b64 whenever the range changes): the compare of the slab test then yields the
// clipped mask directly (VALU compares and lane reads/writes that target SGPRs are the expensive instructions here, 2.4-4 cycles
// against 1.6 for a multiply: profiles/README.md); v_readlane / v_writelane ignore EXEC; every exit restores EXEC = all lanes.  Variants are assembled
// from string macros: slab products (coherent: near/far planes by sign octant, one statement per octant; non-coherent: min/max per
// axis), shared or per-ray origins, the slack formula (distances >= 0, or any distance with -inf = masked), what is counted.
//   node record  s[84:91] = bmin.xyz, bmax.xyz, sub, aux            stack: lane i of (stkN, stkF) = slot i
//   slab test    tn = max3_k(id_k * (near_k - o_k)), tf = min3_k(id_k * (far_k - o_k));  lane passes <=> max(tn,0) <= min(tf,dist)
//                quad passes <=> max_l(min(tf,dist) - max(tn,0)) >= 0   (finite operands, fp32 denormals on: exact sign)
//   range clip   (unsigned)(lane - first) <= (unsigned)(last - first)
//   child order  near = sub + (firstNode ^ sign[axis]) with sign from lane 0 of quad 0 (src/bvh/traverse.cpp:21,71-74):
//                sign16 = signBits << 16, so bit 16 of (sign16 >> axis) ^ aux is that XOR (aux = axis | firstNode << 16)
//   iters        every chain of visits starts with a pop (the root is pushed by the caller) and every push is popped, so
//                visits = 2 * pops - 1: only the pops are counted
#ifndef SNAIL_EXP_PAD
#define SNAIL_EXP_PAD "" // experiment hook: extra instructions per node visit (tools/exp_pad.sh)
#endif
// slab products of ray L -> tn in t0, tf in t3.  COH: near/far planes known (pn*, pf*); FAST: planes bmin-o / bmax-o, min/max per axis
#define SNAIL_SLAB_COH(L, NX, FX, NY, FY, NZ, FZ)                                                                                           \
	"v_mul_f32 %[t0], %[ix" L "], %[pnx]\n v_mul_f32 %[t1], %[iy" L "], %[pny]\n v_mul_f32 %[t2], %[iz" L "], %[pnz]\n"                    \
	"v_mul_f32 %[t3], %[ix" L "], %[pfx]\n v_mul_f32 %[t4], %[iy" L "], %[pfy]\n v_mul_f32 %[t5], %[iz" L "], %[pfz]\n"                    \
	"v_max3_f32 %[t0], %[t0], %[t1], %[t2]\n v_min3_f32 %[t3], %[t3], %[t4], %[t5]\n"
#define SNAIL_SLAB_FAST(L, NX, FX, NY, FY, NZ, FZ)                                                                                          \
	"v_mul_f32 %[t0], %[ix" L "], %[pnx]\n v_mul_f32 %[t3], %[ix" L "], %[pfx]\n v_min_f32 %[u0], %[t0], %[t3]\n v_max_f32 %[t3], %[t0], %[t3]\n" \
	"v_mul_f32 %[t1], %[iy" L "], %[pny]\n v_mul_f32 %[t4], %[iy" L "], %[pfy]\n v_min_f32 %[t0], %[t1], %[t4]\n v_max_f32 %[t4], %[t1], %[t4]\n" \
	"v_mul_f32 %[t2], %[iz" L "], %[pnz]\n v_mul_f32 %[t5], %[iz" L "], %[pfz]\n v_min_f32 %[t1], %[t2], %[t5]\n v_max_f32 %[t5], %[t2], %[t5]\n" \
	"v_max3_f32 %[t0], %[u0], %[t0], %[t1]\n v_min3_f32 %[t3], %[t3], %[t4], %[t5]\n"
// shared origin with CAMERA-RELATIVE node records (SnailScene::relFor: bmin - o, bmax - o stored per node, once per origin): the plane
// offsets of SNAIL_PRE_SHARED are the record's own words, multiplied straight out of the scalar registers -- six vector instructions
// less per visit, and they were the head of the visit's dependency chain.  Same subtraction, same products: same bits.
#define SNAIL_SLAB_COH_R(L, NX, FX, NY, FY, NZ, FZ)                                                                                         \
	"v_mul_f32 %[t0], " NX ", %[ix" L "]\n v_mul_f32 %[t1], " NY ", %[iy" L "]\n v_mul_f32 %[t2], " NZ ", %[iz" L "]\n"                    \
	"v_mul_f32 %[t3], " FX ", %[ix" L "]\n v_mul_f32 %[t4], " FY ", %[iy" L "]\n v_mul_f32 %[t5], " FZ ", %[iz" L "]\n"                    \
	"v_max3_f32 %[t0], %[t0], %[t1], %[t2]\n v_min3_f32 %[t3], %[t3], %[t4], %[t5]\n"
#define SNAIL_SLAB_FAST_R(L, NX, FX, NY, FY, NZ, FZ)                                                                                        \
	"v_mul_f32 %[t0], " NX ", %[ix" L "]\n v_mul_f32 %[t3], " FX ", %[ix" L "]\n v_min_f32 %[u0], %[t0], %[t3]\n v_max_f32 %[t3], %[t0], %[t3]\n" \
	"v_mul_f32 %[t1], " NY ", %[iy" L "]\n v_mul_f32 %[t4], " FY ", %[iy" L "]\n v_min_f32 %[t0], %[t1], %[t4]\n v_max_f32 %[t4], %[t1], %[t4]\n" \
	"v_mul_f32 %[t2], " NZ ", %[iz" L "]\n v_mul_f32 %[t5], " FZ ", %[iz" L "]\n v_min_f32 %[t1], %[t2], %[t5]\n v_max_f32 %[t5], %[t2], %[t5]\n" \
	"v_max3_f32 %[t0], %[u0], %[t0], %[t1]\n v_min3_f32 %[t3], %[t3], %[t4], %[t5]\n"
// per-ray origins (mirrored / transparency packets): the plane offsets are per ray too
#define SNAIL_SLABO_COH(L, NX, FX, NY, FY, NZ, FZ)                                                                                          \
	"v_sub_f32 %[t0], " NX ", %[ox" L "]\n v_sub_f32 %[t1], " NY ", %[oy" L "]\n v_sub_f32 %[t2], " NZ ", %[oz" L "]\n"                    \
	"v_sub_f32 %[t3], " FX ", %[ox" L "]\n v_sub_f32 %[t4], " FY ", %[oy" L "]\n v_sub_f32 %[t5], " FZ ", %[oz" L "]\n"                    \
	"v_mul_f32 %[t0], %[ix" L "], %[t0]\n v_mul_f32 %[t1], %[iy" L "], %[t1]\n v_mul_f32 %[t2], %[iz" L "], %[t2]\n"                       \
	"v_mul_f32 %[t3], %[ix" L "], %[t3]\n v_mul_f32 %[t4], %[iy" L "], %[t4]\n v_mul_f32 %[t5], %[iz" L "], %[t5]\n"                       \
	"v_max3_f32 %[t0], %[t0], %[t1], %[t2]\n v_min3_f32 %[t3], %[t3], %[t4], %[t5]\n"
#define SNAIL_SLABO_FAST(L, NX, FX, NY, FY, NZ, FZ)                                                                                         \
	"v_sub_f32 %[t0], " NX ", %[ox" L "]\n v_sub_f32 %[t3], " FX ", %[ox" L "]\n v_mul_f32 %[t0], %[ix" L "], %[t0]\n v_mul_f32 %[t3], %[ix" L "], %[t3]\n" \
	"v_min_f32 %[u0], %[t0], %[t3]\n v_max_f32 %[t3], %[t0], %[t3]\n"                                                                      \
	"v_sub_f32 %[t1], " NY ", %[oy" L "]\n v_sub_f32 %[t4], " FY ", %[oy" L "]\n v_mul_f32 %[t1], %[iy" L "], %[t1]\n v_mul_f32 %[t4], %[iy" L "], %[t4]\n" \
	"v_min_f32 %[t0], %[t1], %[t4]\n v_max_f32 %[t4], %[t1], %[t4]\n"                                                                      \
	"v_sub_f32 %[t2], " NZ ", %[oz" L "]\n v_sub_f32 %[t5], " FZ ", %[oz" L "]\n v_mul_f32 %[t2], %[iz" L "], %[t2]\n v_mul_f32 %[t5], %[iz" L "], %[t5]\n" \
	"v_min_f32 %[t1], %[t2], %[t5]\n v_max_f32 %[t5], %[t2], %[t5]\n"                                                                      \
	"v_max3_f32 %[t0], %[u0], %[t0], %[t1]\n v_min3_f32 %[t3], %[t3], %[t4], %[t5]\n"
// shared origin: the six plane offsets once per node, ahead of the rays; origin operands of the asm statement
#define SNAIL_PRE_SHARED(NX, FX, NY, FY, NZ, FZ)                                                                                            \
	" v_sub_f32 %[pnx], " NX ", %[ox]\n v_sub_f32 %[pny], " NY ", %[oy]\n v_sub_f32 %[pnz], " NZ ", %[oz]\n"                               \
	" v_sub_f32 %[pfx], " FX ", %[ox]\n v_sub_f32 %[pfy], " FY ", %[oy]\n v_sub_f32 %[pfz], " FZ ", %[oz]\n"
#define SNAIL_PRE_NONE(NX, FX, NY, FY, NZ, FZ) ""
#define SNAIL_ORG_SHARED() [ox] "v"(org[0][0]), [oy] "v"(org[1][0]), [oz] "v"(org[2][0])
#define SNAIL_ORG_PERRAY()                                                                                                                 \
	[ox0] "v"(org[0][0]), [ox1] "v"(org[0][1]), [ox2] "v"(org[0][2]), [ox3] "v"(org[0][3]), [oy0] "v"(org[1][0]), [oy1] "v"(org[1][1]),     \
		[oy2] "v"(org[1][2]), [oy3] "v"(org[1][3]), [oz0] "v"(org[2][0]), [oz1] "v"(org[2][1]), [oz2] "v"(org[2][2]), [oz3] "v"(org[2][3])
// the ray's slack -> S.  POS: distances >= 0 (primary): min(tf,dist) - max(tn,0);  ANY: any distance, -inf = masked (shadow):
// min(min(tf,dist) - tn, tf)
#define SNAIL_TAIL_POS(L, S) "v_max_f32 %[t0], 0, %[t0]\n v_min_f32 %[t3], %[t3], %[d" L "]\n v_sub_f32 %[" S "], %[t3], %[t0]\n"
#define SNAIL_TAIL_ANY(L, S) "v_min_f32 %[t4], %[t3], %[d" L "]\n v_sub_f32 %[t4], %[t4], %[t0]\n v_min_f32 %[" S "], %[t4], %[t3]\n"
// the traversal stack inside the loop.  2W: two words per entry (node; first | last << 8) in two VGPRs.  1W: one word, node | first << 20
// | last << 26, for trees of at most 2^20 node slots: one lane read per pop and one lane write per push instead of two (these are
// the most expensive instructions of the loop, ~4 cycles each against 1.6 for a multiply)
#define SNAIL_POP_2W                                                                                                                       \
	" v_readlane_b32 %[cur], %[stkN], %[sp]\n v_readlane_b32 %[fl], %[stkF], %[sp]\n"                                                      \
	" s_and_b32 %[first], %[fl], 0xff\n s_lshr_b32 %[last], %[fl], 8\n"
#define SNAIL_POP_1W                                                                                                                       \
	" v_readlane_b32 %[fl], %[stkN], %[sp]\n"                                                                                              \
	" s_and_b32 %[cur], %[fl], 0xfffff\n s_bfe_u32 %[first], %[fl], 0x60014\n s_lshr_b32 %[last], %[fl], 26\n"
// push (far child in %[fl]; %[off] is free)
#define SNAIL_PUSH_2W                                                                                                                      \
	" s_lshl_b32 %[off], %[last], 8\n s_or_b32 %[off], %[off], %[first]\n"                                                                 \
	" s_mov_b32 m0, %[sp]\n v_writelane_b32 %[stkN], %[fl], m0\n v_writelane_b32 %[stkF], %[off], m0\n"
#define SNAIL_PUSH_1W                                                                                                                      \
	" s_lshl_b32 %[off], %[last], 6\n s_or_b32 %[off], %[off], %[first]\n s_lshl_b32 %[off], %[off], 20\n s_or_b32 %[off], %[off], %[fl]\n" \
	" s_mov_b32 m0, %[sp]\n v_writelane_b32 %[stkN], %[off], m0\n"
#define SNAIL_COUNT " s_add_u32 %[cnt], %[cnt], 1\n"
// Every statement of the loop starts by waiting for the scalar loads the COMPILER may still have in flight: it does not wait for a load whose
// result turned out dead (the per-ray-origin leaf fetches a triangle record with s_load_dwordx16 and may leave the loop before using it), such a
// load may target the very registers the statement pins (s68..s91 are free between two statements), and scalar loads return out of order --
// a record requested here could be overwritten by the stale one.  (A precaution, one instruction per statement: in today's builds the compiler's
// own scalar loads target s4..s67, and no such overwrite has been observed.)
#define SNAIL_DRAIN_SMEM " s_waitcnt lgkmcnt(0)\n"
#define SNAIL_DESCEND_ASM(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, NX, FX, NY, FY, NZ, FZ)                                                 \
	SNAIL_DESCEND_ASM_S(SNAIL_POP_2W, SNAIL_PUSH_2W, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, NX, FX, NY, FY, NZ, FZ)
#define SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, NX, FX, NY, FY, NZ, FZ)                                                            \
	asm volatile(SNAIL_DRAIN_SMEM                                                                                                          \
				 "L_pop_%=:\n"                                                                                                             \
				 " s_cmp_eq_u32 %[sp], 0\n s_cbranch_scc1 L_done_%=\n"                                                                     \
				 " s_sub_u32 %[sp], %[sp], 1\n" CNTPOP                                                                                     \
				 POP                                                                                                                       \
				 " s_lshl_b32 %[off], %[cur], 5\n s_load_dwordx8 s[84:91], %[base], %[off]\n"                                              \
				 " s_sub_u32 %[width], %[last], %[first]\n"                                                                                \
				 " s_bfm_b64 exec, %[width], %[first]\n s_bitset1_b64 exec, %[last]\n"                                                     \
				 " s_waitcnt lgkmcnt(0)\n"                                                                                                 \
				 "L_visit_%=:\n" CNTVISIT                                                                                                  \
				 PRE(NX, FX, NY, FY, NZ, FZ)                                                                                               \
				 SLAB("0", NX, FX, NY, FY, NZ, FZ) TAIL("0", "s0") SLAB("1", NX, FX, NY, FY, NZ, FZ) TAIL("1", "s1")                       \
				 SLAB("2", NX, FX, NY, FY, NZ, FZ) TAIL("2", "s2") SLAB("3", NX, FX, NY, FY, NZ, FZ) TAIL("3", "s3")                       \
				 " v_max_f32 %[s2], %[s2], %[s3]\n v_max3_f32 %[s0], %[s0], %[s1], %[s2]\n"                                                \
				 SNAIL_EXP_PAD                                                                                                             \
				 " v_cmp_le_f32 vcc, 0, %[s0]\n"                                                                                           \
				 " s_and_b64 %[alive], vcc, exec\n s_cbranch_scc0 L_pop_%=\n"                                                              \
				 " s_ff1_i32_b64 %[first], %[alive]\n s_flbit_i32_b64 %[last], %[alive]\n s_xor_b32 %[last], %[last], 63\n"                \
				 " s_sub_u32 %[width], %[last], %[first]\n"                                                                                \
				 " s_bfm_b64 exec, %[width], %[first]\n s_bitset1_b64 exec, %[last]\n"                                                     \
				 " s_cmp_lt_i32 s90, 0\n s_cbranch_scc1 L_leaf_%=\n"                                                                       \
				 " s_lshr_b32 %[cur], %[sign16], s91\n s_xor_b32 %[cur], %[cur], s91\n s_bfe_u32 %[cur], %[cur], 0x10010\n"                \
				 " s_add_u32 %[fl], s90, 1\n s_sub_u32 %[fl], %[fl], %[cur]\n"                                                             \
				 " s_add_u32 %[cur], s90, %[cur]\n s_lshl_b32 %[off], %[cur], 5\n"                                                         \
				 " s_load_dwordx8 s[84:91], %[base], %[off]\n"                                                                             \
				 PUSH                                                                                                                      \
				 " s_add_u32 %[sp], %[sp], 1\n"                                                                                            \
				 " s_waitcnt lgkmcnt(0)\n s_branch L_visit_%=\n"                                                                           \
				 "L_leaf_%=:\n s_mov_b32 %[leafSub], s90\n s_mov_b32 %[leafAux], s91\n s_branch L_end_%=\n"                                \
				 "L_done_%=:\n s_mov_b32 %[leafSub], 0\n s_mov_b32 %[leafAux], 0\n"                                                        \
				 "L_end_%=:\n s_mov_b64 exec, -1\n"                                                                                        \
				 : [sp] "+s"(sp), [first] "+s"(first), [last] "+s"(last), [cnt] "+s"(cnt), [stkN] "+v"(stkN), [stkF] "+v"(stkF),           \
				   [leafSub] "=&s"(leafSub), [leafAux] "=&s"(leafAux), [cur] "=&s"(sCur), [fl] "=&s"(sFl), [off] "=&s"(sOff),              \
				   [width] "=&s"(sWidth), [rng] "=&s"(sRng), [alive] "=&s"(sAlive), [pnx] "=&v"(vt[0]), [pny] "=&v"(vt[1]),                \
				   [pnz] "=&v"(vt[2]), [pfx] "=&v"(vt[3]), [pfy] "=&v"(vt[4]), [pfz] "=&v"(vt[5]), [t0] "=&v"(vt[6]), [t1] "=&v"(vt[7]),   \
				   [t2] "=&v"(vt[8]), [t3] "=&v"(vt[9]), [t4] "=&v"(vt[10]), [t5] "=&v"(vt[11]), [s0] "=&v"(vt[12]), [s1] "=&v"(vt[13]),   \
				   [s2] "=&v"(vt[14]), [s3] "=&v"(vt[15]), [u0] "=&v"(vt[16])                                                              \
				 : [base] "s"(nodeBase), [sign16] "s"(sign16), [lane] "v"(lane), ORGOPS(), [ix0] "v"(Q.id[0][0]), [ix1] "v"(Q.id[0][1]), [ix2] "v"(Q.id[0][2]), [ix3] "v"(Q.id[0][3]),        \
				   [iy0] "v"(Q.id[1][0]), [iy1] "v"(Q.id[1][1]), [iy2] "v"(Q.id[1][2]), [iy3] "v"(Q.id[1][3]), [iz0] "v"(Q.id[2][0]),      \
				   [iz1] "v"(Q.id[2][1]), [iz2] "v"(Q.id[2][2]), [iz3] "v"(Q.id[2][3]), [d0] "v"(Q.dist[0]), [d1] "v"(Q.dist[1]),          \
				   [d2] "v"(Q.dist[2]), [d3] "v"(Q.dist[3])                                                                                \
				 : "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "vcc", "scc", "m0");                                            \
	/* every result is consumed HERE, in the block of the asm statement (no code is emitted for this): LLVM classes the whole result     \
	   tuple of an asm with mixed SGPR/VGPR outputs as scalar once the tuple itself is live across a block boundary                      \
	   (SITargetLowering::requiresUniformRegister), i.e. once the optimiser sinks one of the extractions into a successor */              \
	asm volatile("" ::"s"(sp), "s"(first), "s"(last), "s"(cnt), "v"(stkN), "v"(stkF), "s"(leafSub), "s"(leafAux), "s"(sCur), "s"(sFl), "s"(sOff),    \
				 "s"(sWidth), "s"(sRng), "s"(sAlive), "v"(vt[0]), "v"(vt[1]), "v"(vt[2]), "v"(vt[3]), "v"(vt[4]), "v"(vt[5]), "v"(vt[6]), "v"(vt[7]), \
				 "v"(vt[8]), "v"(vt[9]), "v"(vt[10]), "v"(vt[11]), "v"(vt[12]), "v"(vt[13]), "v"(vt[14]), "v"(vt[15]), "v"(vt[16]))
// ---- the same loop with the node records fetched AHEAD of their use (one-word stack entries only) -------------------------------
// A packet's walk is a chain of dependent record fetches: visit -> (slab test) -> fetch the child -> visit ..., and a pop fetches the
// popped node.  Beside four other waves per SIMD that latency is hidden; in a frame's tail -- its heaviest packets, alone on the
// machine -- it is the frame time: the heaviest packet of the atrium frame traced ALONE takes 0.18 ms = ~1000 cycles per visit of which
// ~390 issue (tools/heavy_alone.py).  Here three records are resident or in flight:
//   C = s[84:91]  the node being tested
//   N = s[76:83]  its near child, requested as soon as C has arrived (for a leaf: the root, a harmless touch), i.e. BEFORE the slab test
//   T = s[68:75]  the record of the stack's top entry: requested when that entry becomes the top (at a push: the far child, the same
//                 64-B line as the near one; at a pop: the entry below), so that a pop finds its node already here
// A register set never has two requests in flight (scalar loads return out of order): every request into N or T follows an
// s_waitcnt lgkmcnt(0) that covers the previous one, and the statement is left with nothing in flight.  T does not survive the C++
// leaf code between two statements: L_entry requests it again (its line was fetched a leaf body ago).
//   topw = the stack word (record slot | first << 20 | last << 26) of the top entry, kept in an SGPR: one lane read per pop as before
// The loop reads its own copy of the tree (SnailScene::dPF, built by snail_scene_create / the LBVH builder; dev::pfEncode):
//   [slot 0: unused][slot i + 1: node i] ... [triangle records], ONE allocation, so that one base register addresses both, and
//   a child pair (sub, sub + 1; sub is odd) shares ONE 64-B line: the near child's request brings the far child's record as well;
//   words 6, 7 of an inner record = byte offset of the child that is visited first when sign[axis] is clear, 1 << axis: the near child
//   is that offset ^ (sign[axis] ? 32 : 0), the far child near ^ 32 -- five scalar instructions where the reference's encoding took eight,
//   and no shift on the far child's request; of a leaf record = 0x80000000 | byte offset of its first triangle record, count: the
//   "near child" request of a leaf -- issued BEFORE its box test, as for any node -- fetches that triangle's line, which the leaf code
//   would otherwise wait for from cold (a leaf's triangle loads are the longest stall of a packet's walk).
#define SNAIL_MOV_REC(D0, D1, D2, D3, S0, S1, S2, S3)                                                                                       \
	" s_mov_b64 " D0 ", " S0 "\n s_mov_b64 " D1 ", " S1 "\n s_mov_b64 " D2 ", " S2 "\n s_mov_b64 " D3 ", " S3 "\n"
#define SNAIL_A_FROM_T SNAIL_MOV_REC("s[84:85]", "s[86:87]", "s[88:89]", "s[90:91]", "s[68:69]", "s[70:71]", "s[72:73]", "s[74:75]")
// One visit of the loop with the tested record in register set X (planes NX..FZ, SUB = subNode | leaf bit, AUX) and the near child
// requested into the OTHER set: a descent is a jump to the other copy of the body, not a copy of eight registers.  Inside the
// body EXEC = the lanes that survived the node (a lane that fails a box fails every box inside it -- each operation of the slab test
// rounds monotonically -- so first / last come out as with the whole range); a pop rebuilds EXEC from the popped range.
// SNAIL_PF_LEAFREQ: what a LEAF record's "near child" request fetches.  The loop's own copy of the tree holds the triangle records behind
// the nodes: the leaf flag is cleared and the request is the leaf's first triangle record (ahead of the leaf code).  A camera-relative
// node array (below) holds nodes only: the request becomes slot 0.
#define SNAIL_PF_LEAFREQ_TRI " s_bitset0_b32 %[cur], 31\n"
#define SNAIL_PF_LEAFREQ_SLOT0 " s_max_i32 %[cur], %[cur], 0\n"
#define SNAIL_PF_VISIT(X, Y, OTHERSET, SUB, AUX, PRE, SLAB, TAIL, CNTVISIT, LEAFREQ, NX, FX, NY, FY, NZ, FZ)                                          \
				 "L_visit" X "_%=:\n" CNTVISIT                                                                                              \
				 " s_and_b32 %[cur], " AUX ", %[sign16]\n s_cselect_b32 %[cur], 32, 0\n" /* sign[axis] of lane 0 -> 32 or 0 */               \
				 " s_xor_b32 %[cur], " SUB ", %[cur]\n" /* near child's byte offset (a leaf: its first triangle's, maybe + 32) */             \
				 " s_xor_b32 %[fl], %[cur], 32\n" /* far child's: the other half of the pair's 64-B line */                                  \
				 LEAFREQ                                                                                                                   \
				 " s_load_dwordx8 " OTHERSET ", %[base], %[cur]\n"                                                                         \
				 PRE(NX, FX, NY, FY, NZ, FZ)                                                                                               \
				 SLAB("0", NX, FX, NY, FY, NZ, FZ) TAIL("0", "s0") SLAB("1", NX, FX, NY, FY, NZ, FZ) TAIL("1", "s1")                       \
				 SLAB("2", NX, FX, NY, FY, NZ, FZ) TAIL("2", "s2") SLAB("3", NX, FX, NY, FY, NZ, FZ) TAIL("3", "s3")                       \
				 " v_max_f32 %[s2], %[s2], %[s3]\n v_max3_f32 %[s0], %[s0], %[s1], %[s2]\n"                                                \
				 SNAIL_EXP_PAD                                                                                                             \
				 " v_cmp_le_f32 vcc, 0, %[s0]\n" /* EXEC = the parent's survivors: VCC has no bit outside them, VCC IS the new survivor set */ \
				 " s_cbranch_vccz L_fail_%=\n"                                                                                             \
				 " s_ff1_i32_b64 %[first], vcc\n s_flbit_i32_b64 %[last], vcc\n s_xor_b32 %[last], %[last], 63\n"                          \
				 " s_mov_b64 exec, vcc\n"                                                                                                  \
				 " s_cmp_lt_i32 " SUB ", 0\n s_cbranch_scc1 L_leaf" X "_%=\n"                                                               \
				 " s_lshl_b32 %[off], %[last], 6\n s_or_b32 %[off], %[off], %[first]\n s_lshl_b32 %[off], %[off], 20\n"                    \
				 " s_lshr_b32 %[topw], %[fl], 5\n s_or_b32 %[topw], %[topw], %[off]\n" /* stack word: record slot | first << 20 | last << 26 */ \
				 " v_writelane_b32 %[stkN], %[topw], m0\n" /* m0 = sp throughout this statement */                                                           \
				 " s_add_u32 m0, m0, 1\n"                                                                                            \
				 " s_waitcnt lgkmcnt(0)\n"                                                                                                 \
				 " s_load_dwordx8 s[68:75], %[base], %[fl]\n" /* the far child is the new top entry */                                    \
				 " s_branch L_visit" Y "_%=\n"                                                                                              \
				 "L_leaf" X "_%=:\n s_bfe_u32 %[leafSub], " SUB ", 0x190006\n s_sub_u32 %[leafSub], %[leafSub], 0x80000\n s_bitset1_b32 %[leafSub], 31\n" /* 0x80000000 | first triangle: (offset - 2^25) / 64 */ \
				 " s_mov_b32 %[leafAux], " AUX "\n s_waitcnt lgkmcnt(0)\n s_branch L_end_%=\n"
// set A = s[84:91] (planes NXA.., sub s90, aux s91), set B = s[76:83] (planes NXB.., sub s82, aux s83)
#define SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, NXA, FXA, NYA, FYA, NZA, FZA, NXB, FXB, NYB, FYB, NZB, FZB)                 \
	asm volatile(SNAIL_DRAIN_SMEM " s_mov_b32 m0, %[sp]\n"                                                                                \
				 "L_entry_%=:\n"                                                                                                           \
				 " s_cmp_eq_u32 m0, 0\n s_cbranch_scc1 L_done_%=\n"                                                                     \
				 " s_sub_u32 %[off], m0, 1\n"                                                                                           \
				 " v_readlane_b32 %[topw], %[stkN], %[off]\n"                                                                              \
				 " s_and_b32 %[cur], %[topw], 0xfffff\n s_lshl_b32 %[off], %[cur], 5\n"                                                    \
				 " s_load_dwordx8 s[68:75], %[base], %[off]\n"                                                                             \
				 "L_pop_%=:\n" /* sp > 0, topw = the top entry, T = its record (requested) */                                              \
				 " s_sub_u32 m0, m0, 1\n" CNTPOP                                                                                     \
				 " s_bfe_u32 %[first], %[topw], 0x60014\n s_lshr_b32 %[last], %[topw], 26\n"                                               \
				 " s_sub_u32 %[width], %[last], %[first]\n"                                                                                \
				 " s_bfm_b64 exec, %[width], %[first]\n s_bitset1_b64 exec, %[last]\n"                                                     \
				 " s_cmp_eq_u32 m0, 0\n s_cbranch_scc1 L_last_%=\n"                                                                     \
				 " s_sub_u32 %[off], m0, 1\n"                                                                                           \
				 " v_readlane_b32 %[topw], %[stkN], %[off]\n"                                                                              \
				 " s_and_b32 %[cur], %[topw], 0xfffff\n s_lshl_b32 %[off], %[cur], 5\n"                                                    \
				 " s_waitcnt lgkmcnt(0)\n" SNAIL_A_FROM_T                                                                                  \
				 " s_load_dwordx8 s[68:75], %[base], %[off]\n" /* the new top entry's record */                                            \
				 " s_branch L_visitA_%=\n"                                                                                                 \
				 "L_last_%=:\n"                                                                                                            \
				 " s_waitcnt lgkmcnt(0)\n" SNAIL_A_FROM_T                                                                                  \
				 SNAIL_PF_VISIT("A", "B", "s[76:83]", "s90", "s91", PRE, SLAB, TAIL, CNTVISIT, LEAFREQ, NXA, FXA, NYA, FYA, NZA, FZA)                \
				 SNAIL_PF_VISIT("B", "A", "s[84:91]", "s82", "s83", PRE, SLAB, TAIL, CNTVISIT, LEAFREQ, NXB, FXB, NYB, FYB, NZB, FZB)                \
				 "L_fail_%=:\n"                                                                                                            \
				 " s_cmp_eq_u32 m0, 0\n s_cbranch_scc0 L_pop_%=\n"                                                                      \
				 "L_done_%=:\n s_mov_b32 %[leafSub], 0\n s_mov_b32 %[leafAux], 0\n s_waitcnt lgkmcnt(0)\n"                                  \
				 "L_end_%=:\n s_mov_b64 exec, -1\n s_mov_b32 %[sp], m0\n"                                                                                        \
				 : [sp] "+s"(sp), [first] "+s"(first), [last] "+s"(last), [cnt] "+s"(cnt), [stkN] "+v"(stkN), [stkF] "+v"(stkF),           \
				   [leafSub] "=&s"(leafSub), [leafAux] "=&s"(leafAux), [cur] "=&s"(sCur), [fl] "=&s"(sFl), [off] "=&s"(sOff),              \
				   [width] "=&s"(sWidth), [topw] "=&s"(sTopw), [alive] "=&s"(sAlive), [pnx] "=&v"(vt[0]), [pny] "=&v"(vt[1]),              \
				   [pnz] "=&v"(vt[2]), [pfx] "=&v"(vt[3]), [pfy] "=&v"(vt[4]), [pfz] "=&v"(vt[5]), [t0] "=&v"(vt[6]), [t1] "=&v"(vt[7]),   \
				   [t2] "=&v"(vt[8]), [t3] "=&v"(vt[9]), [t4] "=&v"(vt[10]), [t5] "=&v"(vt[11]), [s0] "=&v"(vt[12]), [s1] "=&v"(vt[13]),   \
				   [s2] "=&v"(vt[14]), [s3] "=&v"(vt[15]), [u0] "=&v"(vt[16])                                                              \
				 : [base] "s"(nodeBase), [sign16] "s"(sign16), [lane] "v"(lane), ORGOPS(), [ix0] "v"(Q.id[0][0]), [ix1] "v"(Q.id[0][1]), [ix2] "v"(Q.id[0][2]), [ix3] "v"(Q.id[0][3]),        \
				   [iy0] "v"(Q.id[1][0]), [iy1] "v"(Q.id[1][1]), [iy2] "v"(Q.id[1][2]), [iy3] "v"(Q.id[1][3]), [iz0] "v"(Q.id[2][0]),      \
				   [iz1] "v"(Q.id[2][1]), [iz2] "v"(Q.id[2][2]), [iz3] "v"(Q.id[2][3]), [d0] "v"(Q.dist[0]), [d1] "v"(Q.dist[1]),          \
				   [d2] "v"(Q.dist[2]), [d3] "v"(Q.dist[3])                                                                                \
				 : "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", \
				   "s88", "s89", "s90", "s91", "vcc", "scc", "m0");                                                                        \
	asm volatile("" ::"s"(sp), "s"(first), "s"(last), "s"(cnt), "v"(stkN), "v"(stkF), "s"(leafSub), "s"(leafAux), "s"(sCur), "s"(sFl), "s"(sOff),    \
				 "s"(sWidth), "s"(sTopw), "s"(sAlive), "v"(vt[0]), "v"(vt[1]), "v"(vt[2]), "v"(vt[3]), "v"(vt[4]), "v"(vt[5]), "v"(vt[6]), "v"(vt[7]), \
				 "v"(vt[8]), "v"(vt[9]), "v"(vt[10]), "v"(vt[11]), "v"(vt[12]), "v"(vt[13]), "v"(vt[14]), "v"(vt[15]), "v"(vt[16]))
#define SNAIL_DESCEND_PF_PLAIN(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ)                                                                     \
	SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s85", "s88", "s86", "s89", "s76", "s79", "s77", "s80", "s78", "s81")
#define SNAIL_DESCEND_PF_OCT(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, OCT)                                                                \
	switch(OCT) {                                                                                                                          \
	case 0: SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s85", "s88", "s86", "s89", "s76", "s79", "s77", "s80", "s78", "s81"); break; \
	case 1: SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s87", "s84", "s85", "s88", "s86", "s89", "s79", "s76", "s77", "s80", "s78", "s81"); break; \
	case 2: SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s88", "s85", "s86", "s89", "s76", "s79", "s80", "s77", "s78", "s81"); break; \
	case 3: SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s87", "s84", "s88", "s85", "s86", "s89", "s79", "s76", "s80", "s77", "s78", "s81"); break; \
	case 4: SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s85", "s88", "s89", "s86", "s76", "s79", "s77", "s80", "s81", "s78"); break; \
	case 5: SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s87", "s84", "s85", "s88", "s89", "s86", "s79", "s76", "s77", "s80", "s81", "s78"); break; \
	case 6: SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s88", "s85", "s89", "s86", "s76", "s79", "s80", "s77", "s81", "s78"); break; \
	default: SNAIL_DESCEND_PF(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s87", "s84", "s88", "s85", "s89", "s86", "s79", "s76", "s80", "s77", "s81", "s78"); break; \
	}

// ---- the prefetching loop with the PUSH DEFERRED into the next visit (primary packets over camera-relative records) -----------------
// A wave issues in order, so the scalar tail of a visit -- survivors -> first / last, pack the stack word, write the lane, bump the stack
// pointer, wait, request the far child's record, branch: ~17 instructions that depend on each other -- is time in which the wave issues
// nothing to the vector pipe (a fifth of a visit; the scalar and vector pipes run side by side only for DIFFERENT instructions of a wave's
// stream).  Here a visit that descends only sets EXEC to its survivors and falls (or jumps) into the NEXT visit's "pending" copy, which does
// the push -- first / last out of EXEC, the stack word, the lane write, the far child's record request -- and its own near / far child
// computation BETWEEN its slab products, where scalar instructions issue beside vector ones; a pop likewise leaves the fetch of the new
// top entry's word and record to the popped node's visit.  Three copies of the visit: after a pop (record set A), pending in B, pending
// in A; the far child's offset lives in a register of its own per set (A: %[fl], B: %[width]),
// so that a visit's own near / far computation does not overwrite the pending one.  Invariants are those of SNAIL_DESCEND_PF
// (a register set never has two requests in flight; m0 = sp; topw = the top entry's word, T = its record).
#define SNAIL_PF2_NEARFAR(SUB, AUX, FARX, OTHERSET, LEAFREQ)                                                                                          \
				 " s_and_b32 %[cur], " AUX ", %[sign16]\n s_cselect_b32 %[cur], 32, 0\n"                                                       \
				 " s_xor_b32 %[cur], " SUB ", %[cur]\n s_xor_b32 " FARX ", %[cur], 32\n"                                                       \
				 LEAFREQ /* what a leaf's "near child" request fetches: SNAIL_PF_LEAFREQ_* */                                                \
				 " s_load_dwordx8 " OTHERSET ", %[base], %[cur]\n"
// the end of a visit: EXEC <- the survivors; first / last are taken from EXEC where they are needed (the next visit's push, or the leaf)
#define SNAIL_PF2_TAIL(X, SUB) /* SCC = "this node is a leaf", set by SNAIL_PF2_ISLEAF after the visit's last other scalar instruction */ \
				 " v_max_f32 %[s2], %[s2], %[s3]\n v_max3_f32 %[s0], %[s0], %[s1], %[s2]\n"                                                \
				 " v_cmpx_le_f32 vcc, 0, %[s0]\n"                                                                                           \
				 " s_cbranch_execz L_fail_%=\n"                                                                                            \
				 " s_cbranch_scc1 L_leaf" X "_%=\n"
#define SNAIL_PF2_ISLEAF(SUB) " s_cmp_lt_i32 " SUB ", 0\n"
// the lane range of a stack word as an EXEC mask, kept for the top entry in %[alive] so that a pop only moves it
#define SNAIL_PF2_ALIVE " s_sub_u32 %[cur], %[last], %[first]\n s_bfm_b64 %[alive], %[cur], %[first]\n s_bitset1_b64 %[alive], %[last]\n"
#define SNAIL_PF2_FIRSTLAST " s_ff1_i32_b64 %[first], exec\n s_flbit_i32_b64 %[last], exec\n s_xor_b32 %[last], %[last], 63\n"
// a visit entered from a descent: the push of (FARY, survivors' first / last) happens here, between the slab products
#define SNAIL_PF2_PENDING(X, OTHERSET, SUB, AUX, FARX, FARY, PRE, SLAB, TAIL, CNTVISIT, LEAFREQ, NX, FX, NY, FY, NZ, FZ)                                                   \
				 "L_visit" X "p_%=:\n"                                                                                                      \
				 " s_waitcnt lgkmcnt(0)\n" /* this record has arrived; T's last request too */                                                \
				 " s_load_dwordx8 s[68:75], %[base], " FARY "\n" /* the pusher's far child is the new top entry */                             \
				 CNTVISIT PRE(NX, FX, NY, FY, NZ, FZ)                                                                                        \
				 SLAB("0", NX, FX, NY, FY, NZ, FZ) TAIL("0", "s0")                                                                           \
				 SNAIL_PF2_NEARFAR(SUB, AUX, FARX, OTHERSET, LEAFREQ)                                                                        \
				 SLAB("1", NX, FX, NY, FY, NZ, FZ) TAIL("1", "s1")                                                                           \
				 SNAIL_PF2_FIRSTLAST /* of the pusher: EXEC is still its survivor set */                                                     \
				 " s_lshl_b32 %[off], %[last], 6\n s_or_b32 %[off], %[off], %[first]\n s_lshl_b32 %[off], %[off], 20\n"                    \
				 SNAIL_PF2_ALIVE                                                                                                            \
				 SLAB("2", NX, FX, NY, FY, NZ, FZ) TAIL("2", "s2")                                                                           \
				 " s_lshr_b32 %[topw], " FARY ", 5\n s_or_b32 %[topw], %[topw], %[off]\n"                                                    \
				 " v_writelane_b32 %[stkN], %[topw], m0\n s_add_u32 m0, m0, 1\n" SNAIL_PF2_ISLEAF(SUB)                                       \
				 SLAB("3", NX, FX, NY, FY, NZ, FZ) TAIL("3", "s3")                                                                           \
				 SNAIL_PF2_TAIL(X, SUB)
#define SNAIL_PF2_LEAF(X, SUB, AUX)                                                                                                         \
				 "L_leaf" X "_%=:\n" SNAIL_PF2_FIRSTLAST                                                                                    \
				 " s_bfe_u32 %[leafSub], " SUB ", 0x190006\n s_sub_u32 %[leafSub], %[leafSub], 0x80000\n s_bitset1_b32 %[leafSub], 31\n"   \
				 " s_mov_b32 %[leafAux], " AUX "\n s_waitcnt lgkmcnt(0)\n s_branch L_end_%=\n"
#define SNAIL_DESCEND_PF2X(EXTRACLOB, PREVARS, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, NXA, FXA, NYA, FYA, NZA, FZA, NXB, FXB, NYB, FYB, NZB, FZB)                                                  \
	asm volatile(SNAIL_DRAIN_SMEM " s_mov_b32 m0, %[sp]\n"                                                                                \
				 "L_entry_%=:\n"                                                                                                           \
				 " s_cmp_eq_u32 m0, 0\n s_cbranch_scc1 L_done_%=\n"                                                                     \
				 " s_sub_u32 %[off], m0, 1\n"                                                                                           \
				 " v_readlane_b32 %[topw], %[stkN], %[off]\n"                                                                              \
				 " s_and_b32 %[cur], %[topw], 0xfffff\n s_lshl_b32 %[off], %[cur], 5\n"                                                    \
				 " s_load_dwordx8 s[68:75], %[base], %[off]\n"                                                                             \
				 " s_bfe_u32 %[first], %[topw], 0x60014\n s_lshr_b32 %[last], %[topw], 26\n" SNAIL_PF2_ALIVE                               \
				 "L_pop_%=:\n" /* sp > 0, topw = the top entry, alive = its lanes, T = its record (requested) */                            \
				 " s_sub_u32 m0, m0, 1\n" CNTPOP                                                                                     \
				 " s_mov_b64 exec, %[alive]\n"                                                                                             \
				 " s_waitcnt lgkmcnt(0)\n" SNAIL_A_FROM_T                                                                                  \
				 /* the popped node's visit (record set A); the NEW top entry's word and record are fetched inside it */                  \
				 " s_sub_u32 %[off], m0, 1\n s_max_i32 %[off], %[off], 0\n" /* (an empty stack re-reads entry 0: harmless, never used) */  \
				 " v_readlane_b32 %[topw], %[stkN], %[off]\n"                                                                              \
				 CNTVISIT PRE(NXA, FXA, NYA, FYA, NZA, FZA)                                                                                 \
				 SLAB("0", NXA, FXA, NYA, FYA, NZA, FZA) TAIL("0", "s0")                                                          \
				 " s_and_b32 %[cur], %[topw], 0xfffff\n s_lshl_b32 %[off], %[cur], 5\n"                                                    \
				 " s_load_dwordx8 s[68:75], %[base], %[off]\n"                                                                             \
				 SLAB("1", NXA, FXA, NYA, FYA, NZA, FZA) TAIL("1", "s1")                                                          \
				 SNAIL_PF2_NEARFAR("s90", "s91", "%[fl]", "s[76:83]", LEAFREQ)                                                                       \
				 SLAB("2", NXA, FXA, NYA, FYA, NZA, FZA) TAIL("2", "s2")                                                          \
				 " s_bfe_u32 %[first], %[topw], 0x60014\n s_lshr_b32 %[last], %[topw], 26\n" SNAIL_PF2_ALIVE SNAIL_PF2_ISLEAF("s90")        \
				 SLAB("3", NXA, FXA, NYA, FYA, NZA, FZA) TAIL("3", "s3")                                                          \
				 SNAIL_PF2_TAIL("A", "s90")                                                                                                 \
				 /* falls through: A descends into B, its push pending */                                                                   \
				 SNAIL_PF2_PENDING("B", "s[84:91]", "s82", "s83", "%[width]", "%[fl]", PRE, SLAB, TAIL, CNTVISIT, LEAFREQ, NXB, FXB, NYB, FYB, NZB, FZB)                   \
				 /* falls through: B descends into A, its push pending */                                                                   \
				 SNAIL_PF2_PENDING("A", "s[76:83]", "s90", "s91", "%[fl]", "%[width]", PRE, SLAB, TAIL, CNTVISIT, LEAFREQ, NXA, FXA, NYA, FYA, NZA, FZA)                   \
				 " s_branch L_visitBp_%=\n"                                                                                                \
				 SNAIL_PF2_LEAF("A", "s90", "s91") SNAIL_PF2_LEAF("B", "s82", "s83")                                                       \
				 "L_fail_%=:\n"                                                                                                            \
				 " s_cmp_eq_u32 m0, 0\n s_cbranch_scc0 L_pop_%=\n"                                                                      \
				 "L_done_%=:\n s_mov_b32 %[leafSub], 0\n s_mov_b32 %[leafAux], 0\n s_waitcnt lgkmcnt(0)\n"                                  \
				 "L_end_%=:\n s_mov_b64 exec, -1\n s_mov_b32 %[sp], m0\n"                                                                                        \
				 : [sp] "+s"(sp), [first] "+s"(first), [last] "+s"(last), [cnt] "+s"(cnt), [stkN] "+v"(stkN), [stkF] "+v"(stkF),           \
				   [leafSub] "=&s"(leafSub), [leafAux] "=&s"(leafAux), [cur] "=&s"(sCur), [fl] "=&s"(sFl), [off] "=&s"(sOff),              \
				   [width] "=&s"(sWidth), [topw] "=&s"(sTopw), [alive] "=&s"(sAlive), PREVARS() [t0] "=&v"(vt[6]), [t1] "=&v"(vt[7]),   \
				   [t2] "=&v"(vt[8]), [t3] "=&v"(vt[9]), [t4] "=&v"(vt[10]), [t5] "=&v"(vt[11]), [s0] "=&v"(vt[12]), [s1] "=&v"(vt[13]),   \
				   [s2] "=&v"(vt[14]), [s3] "=&v"(vt[15]), [u0] "=&v"(vt[16])                                                              \
				 : [base] "s"(nodeBase), [sign16] "s"(sign16), [lane] "v"(lane), ORGOPS(), [ix0] "v"(Q.id[0][0]), [ix1] "v"(Q.id[0][1]), [ix2] "v"(Q.id[0][2]), [ix3] "v"(Q.id[0][3]),        \
				   [iy0] "v"(Q.id[1][0]), [iy1] "v"(Q.id[1][1]), [iy2] "v"(Q.id[1][2]), [iy3] "v"(Q.id[1][3]), [iz0] "v"(Q.id[2][0]),      \
				   [iz1] "v"(Q.id[2][1]), [iz2] "v"(Q.id[2][2]), [iz3] "v"(Q.id[2][3]), [d0] "v"(Q.dist[0]), [d1] "v"(Q.dist[1]),          \
				   [d2] "v"(Q.dist[2]), [d3] "v"(Q.dist[3])                                                                                \
				 : "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", \
				   "s88", "s89", "s90", "s91", "vcc", "scc", "m0" EXTRACLOB);                                                                        \
	asm volatile("" ::"s"(sp), "s"(first), "s"(last), "s"(cnt), "v"(stkN), "v"(stkF), "s"(leafSub), "s"(leafAux), "s"(sCur), "s"(sFl), "s"(sOff),    \
				 "s"(sWidth), "s"(sTopw), "s"(sAlive), "v"(vt[6]), "v"(vt[7]), \
				 "v"(vt[8]), "v"(vt[9]), "v"(vt[10]), "v"(vt[11]), "v"(vt[12]), "v"(vt[13]), "v"(vt[14]), "v"(vt[15]), "v"(vt[16]))
// the six plane-offset registers of SNAIL_PRE_SHARED are operands only where a visit forms them (PREVARS = SNAIL_PREVARS_of(PRE))
#define SNAIL_PREVARS_SHARED() [pnx] "=&v"(vt[0]), [pny] "=&v"(vt[1]), [pnz] "=&v"(vt[2]), [pfx] "=&v"(vt[3]), [pfy] "=&v"(vt[4]), [pfz] "=&v"(vt[5]),
#define SNAIL_PREVARS_NONE() [pnx] "=&v"(vt[0]), [pny] "=&v"(vt[1]), [pnz] "=&v"(vt[2]), [pfx] "=&v"(vt[3]), [pfy] "=&v"(vt[4]), [pfz] "=&v"(vt[5]),   /* (SNAIL_PRE_NONE keeps the operands; SNAIL_PRE_NONE_X drops them) */
#define SNAIL_PREVARS_SNAIL_PRE_SHARED SNAIL_PREVARS_SHARED
#define SNAIL_PREVARS_SNAIL_PRE_NONE SNAIL_PREVARS_NONE
#define SNAIL_PREVARS_SNAIL_PRE_SEL SNAIL_PREVARS_NONE
#define SNAIL_PREVARS_EMPTY()
#define SNAIL_PRE_NONE_X(NX, FX, NY, FY, NZ, FZ) "" /* = SNAIL_PRE_NONE, in walks that compile without the six unused operands */
#define SNAIL_PREVARS_SNAIL_PRE_NONE_X SNAIL_PREVARS_EMPTY
#define SNAIL_PREVARS_of(PRE) SNAIL_PREVARS_##PRE
#define SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, NXA, FXA, NYA, FYA, NZA, FZA, NXB, FXB, NYB, FYB, NZB, FZB)                 \
	SNAIL_DESCEND_PF2X(, SNAIL_PREVARS_of(PRE), PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, NXA, FXA, NYA, FYA, NZA, FZA, NXB, FXB, NYB, FYB, NZB, FZB)
// coherent packets WITHOUT one statement per sign octant: the near / far plane of each axis is picked on the scalar side, per visit, into
// s62..s67 (nine scalar instructions that issue beside the visit's vector ones) -- for walks whose leaf code leaves the compiler no room
// for eight copies of this loop (per-ray origins)
#define SNAIL_PRE_SEL(NX, FX, NY, FY, NZ, FZ)                                                                                               \
				 " s_bitcmp1_b32 %[sign16], 8\n s_cselect_b32 s62, " FX ", " NX "\n s_cselect_b32 s65, " NX ", " FX "\n"                       \
				 " s_bitcmp1_b32 %[sign16], 9\n s_cselect_b32 s63, " FY ", " NY "\n s_cselect_b32 s66, " NY ", " FY "\n"                       \
				 " s_bitcmp1_b32 %[sign16], 10\n s_cselect_b32 s64, " FZ ", " NZ "\n s_cselect_b32 s67, " NZ ", " FZ "\n"
#define SNAIL_SLABO_SEL(L, NX, FX, NY, FY, NZ, FZ) SNAIL_SLABO_COH(L, "s62", "s65", "s63", "s66", "s64", "s67")
#define SNAIL_SEL_CLOB , "s62", "s63", "s64", "s65", "s66", "s67"
#define SNAIL_DESCEND_PF2_SEL(ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ)                                                                \
	SNAIL_DESCEND_PF2X(SNAIL_SEL_CLOB, SNAIL_PREVARS_NONE, SNAIL_PRE_SEL, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s85", "s88", "s86", "s89", "s76", "s79", "s77", "s80", "s78", "s81")
#define SNAIL_DESCEND_PF2_PLAIN(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ) SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s85", "s88", "s86", "s89", "s76", "s79", "s77", "s80", "s78", "s81")
#define SNAIL_DESCEND_PF2_OCT(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, OCT)                                                                                                    \
	switch(OCT) {                                                                                                                          \
	case 0: SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s85", "s88", "s86", "s89", "s76", "s79", "s77", "s80", "s78", "s81"); break;            \
	case 1: SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s87", "s84", "s85", "s88", "s86", "s89", "s79", "s76", "s77", "s80", "s78", "s81"); break;            \
	case 2: SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s88", "s85", "s86", "s89", "s76", "s79", "s80", "s77", "s78", "s81"); break;            \
	case 3: SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s87", "s84", "s88", "s85", "s86", "s89", "s79", "s76", "s80", "s77", "s78", "s81"); break;            \
	case 4: SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s85", "s88", "s89", "s86", "s76", "s79", "s77", "s80", "s81", "s78"); break;            \
	case 5: SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s87", "s84", "s85", "s88", "s89", "s86", "s79", "s76", "s77", "s80", "s81", "s78"); break;            \
	case 6: SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s84", "s87", "s88", "s85", "s89", "s86", "s76", "s79", "s80", "s77", "s81", "s78"); break;            \
	default: SNAIL_DESCEND_PF2(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, LEAFREQ, "s87", "s84", "s88", "s85", "s89", "s86", "s79", "s76", "s80", "s77", "s81", "s78"); break;           \
	}

// near/far plane registers by sign octant (bit k set = idir negative on axis k: near plane = bmax[k]); s[84:86] = bmin, s[87:89] = bmax
#define SNAIL_DESCEND_OCT(PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, OCT) SNAIL_DESCEND_OCT_S(SNAIL_POP_2W, SNAIL_PUSH_2W, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, OCT)
#define SNAIL_DESCEND_OCT_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, OCT)                                                                               \
	switch(OCT) {                                                                                                                          \
	case 0: SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, "s84", "s87", "s85", "s88", "s86", "s89"); break;                              \
	case 1: SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, "s87", "s84", "s85", "s88", "s86", "s89"); break;                              \
	case 2: SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, "s84", "s87", "s88", "s85", "s86", "s89"); break;                              \
	case 3: SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, "s87", "s84", "s88", "s85", "s86", "s89"); break;                              \
	case 4: SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, "s84", "s87", "s85", "s88", "s89", "s86"); break;                              \
	case 5: SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, "s87", "s84", "s85", "s88", "s89", "s86"); break;                              \
	case 6: SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, "s84", "s87", "s88", "s85", "s89", "s86"); break;                              \
	default: SNAIL_DESCEND_ASM_S(POP, PUSH, PRE, ORGOPS, SLAB, TAIL, CNTPOP, CNTVISIT, "s87", "s84", "s88", "s85", "s89", "s86"); break;                             \
	}

// Issue priority by work done (wave-uniform, at every leaf): packet costs are heavy-tailed and a frame ends with its heaviest
// packets; a wave that has already popped more entries than most packets ever do is one of them.  s_setprio makes the SIMD's
// arbiter prefer it over its co-resident waves (priority, then age: MI355X_MICROARCH.md, "Two waves per SIMD"), so the long
// packets run at the speed of a lone wave while the short ones fill the gaps -- work-conserving, results untouched.
#ifndef SNAIL_PRIO_T1
#define SNAIL_PRIO_T1 0 // pops; 0 = off
#define SNAIL_PRIO_T2 0
#define SNAIL_PRIO_T3 0
#endif
#if SNAIL_PRIO_T1 > 0
#define SNAIL_PRIO_BY_WORK(cnt)                                                                                                            \
	do {                                                                                                                                   \
		if((cnt) >= SNAIL_PRIO_T3) __builtin_amdgcn_s_setprio(3);                                                                          \
		else if((cnt) >= SNAIL_PRIO_T2) __builtin_amdgcn_s_setprio(2);                                                                     \
		else if((cnt) >= SNAIL_PRIO_T1) __builtin_amdgcn_s_setprio(1);                                                                     \
	} while(0)
#else
#define SNAIL_PRIO_BY_WORK(cnt) do { } while(0)
#endif
// SHADOW=false: closest hit of a primary packet (distances >= 0; visits = 2 * pops - 1: every chain of visits starts with a pop,
// the root is pushed here, and every push is popped).  SHADOW=true: any hit of a shadow packet (masked lanes -inf; the walk ends
// when a triangle occludes the whole packet, so every visit is counted).  COH: one asm statement per sign octant, picked by a
// wave-uniform switch at every (re-)entry, i.e. once per leaf; the leaf code exists once.
#ifndef SNAIL_NODE_PREFETCH
#define SNAIL_NODE_PREFETCH 1 // 0 = the loop without record prefetch for one-word stacks too (A/B measurements)
#endif
#ifndef SNAIL_DEFER_PUSH
#define SNAIL_DEFER_PUSH 1 // the prefetching loop with the push of a descent done inside the next visit (SNAIL_DESCEND_PF2); 0 = SNAIL_DESCEND_PF (A/B measurements)
#endif
#if SNAIL_DEFER_PUSH
#define SNAIL_WALK_PF_OCT SNAIL_DESCEND_PF2_OCT
#define SNAIL_WALK_PF_PLAIN SNAIL_DESCEND_PF2_PLAIN
#else
#define SNAIL_WALK_PF_OCT SNAIL_DESCEND_PF_OCT
#define SNAIL_WALK_PF_PLAIN SNAIL_DESCEND_PF_PLAIN
#endif
#ifndef SNAIL_REL_NODES
#define SNAIL_REL_NODES 1 // primary packets walk camera-relative node records (no plane offsets to compute per visit); 0 = the loop's plain copy
#endif
#ifndef SNAIL_REL_SHADOW
#define SNAIL_REL_SHADOW 1 // shadow packets of k_light walk records relative to their light's position (ShadeArgs::relLight), as primary packets do for the camera
#endif
// the node array a PACK instantiation of the hand-written walks is given: the prefetching loop's own copy of the tree
#define SNAIL_PACK_NODES(A) (SNAIL_NODE_PREFETCH ? (A).pf : (A).nodes)
template <bool SHADOW, bool COH, bool PACK, bool MASK, bool BARY, bool POSDIST, bool REL = true /* PACK: `nodes` holds records relative to the packet's origin (else the loop's plain copy) */>
__device__ __forceinline__ void walkSharedAsm(const uint4 *__restrict__ nodes /* PACK: the prefetching loop's copy, SnailScene::dPF */, const uint4 *__restrict__ tris, int size, int lane,
											  const float (&org)[3][4], Quad &Q, unsigned mask4, int (&tid)[4], float (&bu)[4], float (&bv)[4], float *lds,
											  Counters &st, const int oct) {
	Interval iv;
	{ // RayInterval ctor (src/ray_group.h:296-333), as in dev::walk
		unsigned act4 = lane < size ? 15u : 0u;
		if(SHADOW) {
			act4 = 0;
			if(lane < size)
#pragma unroll
				for(int l = 0; l < 4; l++) act4 |= (Q.dist[l] >= 0.0f ? 1u : 0u) << l;
		} else if(MASK) act4 = lane < size ? (mask4 & 15u) : 0u;
		computeMinMax<false, (MASK || SHADOW)>(Q.d, act4, size, lane, lds, iv.minDir, iv.maxDir);
	}
#pragma unroll
	for(int k = 0; k < 3; k++) { iv.minIDir[k] = iv.maxIDir[k] = 0.0f; iv.minOrg[k] = iv.maxOrg[k] = org[k][0]; }
	const int signBits = __builtin_amdgcn_readfirstlane((Q.d[0][0] < 0.0f ? 1 : 0) | (Q.d[1][0] < 0.0f ? 2 : 0) | (Q.d[2][0] < 0.0f ? 4 : 0));
	float tidBits[4];   // the caller's tid[] as float bits while the loop statements are around (setId)
#pragma unroll
	for(int l = 0; l < 4; l++) tidBits[l] = __int_as_float(tid[l]);
	constexpr bool PF = PACK && SNAIL_NODE_PREFETCH;   // the record-prefetching loop over its own copy of the tree
	const int sign16 = PF ? signBits : signBits << 16; // (PF: sign bit k against an inner record's 1 << axis)
	const u64 nodeBase = (u64)nodes;
	// stack slot 0 = the root (PF: record slot 1) with the full quad range.  (float-typed: the only 32-bit INTEGER values that go in and out of the loop
	// statements are then scalar ones -- the instruction selector shares one undefined register among all undefined values of a type on a path, and an
	// undefined VGPR feeding a scalar operand's PHI is this compiler's "illegal VGPR to SGPR copy")
	float stkN = __int_as_float(PACK ? (int)((unsigned)(size - 1) << 26) | (PF ? 1 : 0) : 0), stkF = __int_as_float((size - 1) << 8);
	int sp = 1, first = 0, last = size - 1, cnt = 0;
	for(;;) {
		int leafSub, leafAux, sCur, sFl, sOff, sWidth;
		u64 sRng, sAlive;
		float vt[17];
#define SNAIL_SHARED_VARIANTS(POP, PUSH)                                                                                                   \
		if(COH) {                                                                                                                          \
			if(SHADOW) { SNAIL_DESCEND_OCT_S(POP, PUSH, SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_COH, SNAIL_TAIL_ANY, "", SNAIL_COUNT, oct) } \
			else if(POSDIST) { SNAIL_DESCEND_OCT_S(POP, PUSH, SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_COH, SNAIL_TAIL_POS, SNAIL_COUNT, "", oct) } \
			else { SNAIL_DESCEND_OCT_S(POP, PUSH, SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_COH, SNAIL_TAIL_ANY, SNAIL_COUNT, "", oct) }      \
		} else {                                                                                                                           \
			if(SHADOW) { SNAIL_DESCEND_ASM_S(POP, PUSH, SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_FAST, SNAIL_TAIL_ANY, "", SNAIL_COUNT, "s84", "s87", "s85", "s88", "s86", "s89"); } \
			else if(POSDIST) { SNAIL_DESCEND_ASM_S(POP, PUSH, SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_FAST, SNAIL_TAIL_POS, SNAIL_COUNT, "", "s84", "s87", "s85", "s88", "s86", "s89"); } \
			else { SNAIL_DESCEND_ASM_S(POP, PUSH, SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_FAST, SNAIL_TAIL_ANY, SNAIL_COUNT, "", "s84", "s87", "s85", "s88", "s86", "s89"); } \
		}
		if(PF) {
			int sTopw;
			// primary packets (POSDIST) read camera-relative records: no plane offsets to form, a leaf's request is slot 0
			if(COH) {
				if(SHADOW && REL && SNAIL_REL_SHADOW) { SNAIL_WALK_PF_OCT(SNAIL_PRE_NONE_X, SNAIL_ORG_SHARED, SNAIL_SLAB_COH_R, SNAIL_TAIL_ANY, "", SNAIL_COUNT, SNAIL_PF_LEAFREQ_SLOT0, oct) }
				else if(SHADOW) { SNAIL_WALK_PF_OCT(SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_COH, SNAIL_TAIL_ANY, "", SNAIL_COUNT, SNAIL_PF_LEAFREQ_TRI, oct) }
#if SNAIL_REL_NODES
				else if(POSDIST) { SNAIL_WALK_PF_OCT(SNAIL_PRE_NONE_X, SNAIL_ORG_SHARED, SNAIL_SLAB_COH_R, SNAIL_TAIL_POS, SNAIL_COUNT, "", SNAIL_PF_LEAFREQ_SLOT0, oct) }
#else
				else if(POSDIST) { SNAIL_WALK_PF_OCT(SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_COH, SNAIL_TAIL_POS, SNAIL_COUNT, "", SNAIL_PF_LEAFREQ_TRI, oct) }
#endif
				else { SNAIL_WALK_PF_OCT(SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_COH, SNAIL_TAIL_ANY, SNAIL_COUNT, "", SNAIL_PF_LEAFREQ_TRI, oct) }
			} else {
				if(SHADOW && REL && SNAIL_REL_SHADOW) { SNAIL_WALK_PF_PLAIN(SNAIL_PRE_NONE_X, SNAIL_ORG_SHARED, SNAIL_SLAB_FAST_R, SNAIL_TAIL_ANY, "", SNAIL_COUNT, SNAIL_PF_LEAFREQ_SLOT0); }
				else if(SHADOW) { SNAIL_WALK_PF_PLAIN(SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_FAST, SNAIL_TAIL_ANY, "", SNAIL_COUNT, SNAIL_PF_LEAFREQ_TRI); }
#if SNAIL_REL_NODES
				else if(POSDIST) { SNAIL_WALK_PF_PLAIN(SNAIL_PRE_NONE_X, SNAIL_ORG_SHARED, SNAIL_SLAB_FAST_R, SNAIL_TAIL_POS, SNAIL_COUNT, "", SNAIL_PF_LEAFREQ_SLOT0); }
#else
				else if(POSDIST) { SNAIL_WALK_PF_PLAIN(SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_FAST, SNAIL_TAIL_POS, SNAIL_COUNT, "", SNAIL_PF_LEAFREQ_TRI); }
#endif
				else { SNAIL_WALK_PF_PLAIN(SNAIL_PRE_SHARED, SNAIL_ORG_SHARED, SNAIL_SLAB_FAST, SNAIL_TAIL_ANY, SNAIL_COUNT, "", SNAIL_PF_LEAFREQ_TRI); }
			}
		} else if(PACK) { SNAIL_SHARED_VARIANTS(SNAIL_POP_1W, SNAIL_PUSH_1W) }
		else { SNAIL_SHARED_VARIANTS(SNAIL_POP_2W, SNAIL_PUSH_2W) }
#undef SNAIL_SHARED_VARIANTS
		if(leafSub == 0) break;
		SNAIL_PRIO_BY_WORK(cnt);
		if(leafShared<MASK, SHADOW, COH ? M_COH : M_FAST, BARY>(tris, leafAux, (int)((unsigned)leafSub & 0x7fffffffu), size, lane, first, last, org, Q, mask4,
																 tidBits, bu, bv, iv, st))
			break;
	}
#pragma unroll
	for(int l = 0; l < 4; l++) tid[l] = __float_as_int(tidBits[l]);
	st.iters += SHADOW ? (unsigned)cnt : 2u * (unsigned)cnt - 1u;
}

// closest hit of a packet with per-ray origins (TraversePrimaryN<0,mask>), node loop in assembly as in walkSharedAsm; any
// distance on entry (masked lanes -inf), `size` quads
#ifndef SNAIL_SHADOW_ENTRY_PF
#define SNAIL_SHADOW_ENTRY_PF 1 // the packets of snail_trace_shadow and of snail_trace_rays with shared origins through the prefetching loop (0 = the plain two-word loop, as before round 3)
#endif
#ifndef SNAIL_PERRAY_COH_PF
#define SNAIL_PERRAY_COH_PF SNAIL_DEFER_PUSH // coherent per-ray-origin packets (most mirrored packets) through the prefetching loop as well (SNAIL_DESCEND_PF2_SEL)
#endif
template <bool MASK, bool COH, bool BARY, bool PACK>
__device__ __forceinline__ void walkPerRayAsm(const uint4 *__restrict__ nodes /* PACK and not COH: the prefetching loop's copy */, const uint4 *__restrict__ tris, int size, int lane, const float (&org)[3][4],
											  Quad &Q, unsigned mask4, int (&tid)[4], float (&bu)[4], float (&bv)[4], Counters &st, const int oct) {
	const int signBits = __builtin_amdgcn_readfirstlane((Q.d[0][0] < 0.0f ? 1 : 0) | (Q.d[1][0] < 0.0f ? 2 : 0) | (Q.d[2][0] < 0.0f ? 4 : 0));
	constexpr bool PF = PACK && SNAIL_NODE_PREFETCH && (SNAIL_PERRAY_COH_PF || !COH);   // (SNAIL_PERRAY_COH_PF 0: coherent packets keep the plain two-word loop over the caller's records)
	// PF: bits 0..2 = the signs of lane 0's first ray (child order, as the reference takes it); bits 8..10 = the packet's sign octant (plane selection of SNAIL_PRE_SEL)
	const int sign16 = PF ? (COH ? signBits | __builtin_amdgcn_readfirstlane(oct) << 8 : signBits) : signBits << 16;
	const u64 nodeBase = (u64)nodes;
	float stkN = __int_as_float(PF ? (int)((unsigned)(size - 1) << 26) | 1 : 0), stkF = __int_as_float((size - 1) << 8); // stack slot 0 = the root (PF: record slot 1) with the full quad range; float-typed as in walkSharedAsm
	int sp = 1, first = 0, last = size - 1, cnt = 0;
	for(;;) {
		int leafSub, leafAux, sCur, sFl, sOff, sWidth;
		u64 sRng, sAlive;
		float vt[17];
		if(PF) {
			// one-word stack entries + node records fetched ahead.  Non-coherent packets: the plain form of the loop.  Coherent packets would need the
			// loop once per sign octant, and this compiler cannot place eight (or even two) copies of it beside the per-ray leaf code ("illegal VGPR
			// to SGPR copy": the scalar-register pressure of the 16-SGPR triangle record plus three node record sets): they take ONE statement in
			// which the near / far planes are picked per visit on the scalar side (SNAIL_DESCEND_PF2_SEL).
			int sTopw;
			if(COH) { SNAIL_DESCEND_PF2_SEL(SNAIL_ORG_PERRAY, SNAIL_SLABO_SEL, SNAIL_TAIL_ANY, SNAIL_COUNT, "", SNAIL_PF_LEAFREQ_TRI); }
			else { SNAIL_WALK_PF_PLAIN(SNAIL_PRE_NONE, SNAIL_ORG_PERRAY, SNAIL_SLABO_FAST, SNAIL_TAIL_ANY, SNAIL_COUNT, "", SNAIL_PF_LEAFREQ_TRI); }
		} else if(COH) { SNAIL_DESCEND_OCT(SNAIL_PRE_NONE, SNAIL_ORG_PERRAY, SNAIL_SLABO_COH, SNAIL_TAIL_ANY, SNAIL_COUNT, "", oct) }
		else { SNAIL_DESCEND_ASM(SNAIL_PRE_NONE, SNAIL_ORG_PERRAY, SNAIL_SLABO_FAST, SNAIL_TAIL_ANY, SNAIL_COUNT, "", "s84", "s87", "s85", "s88", "s86", "s89"); }
		if(leafSub == 0) break;
		leafPerRay<MASK, COH ? M_COH : M_FAST, BARY>(tris, leafAux, (int)((unsigned)leafSub & 0x7fffffffu), lane, first, last, org, Q, mask4, tid, bu, bv, st);
	}
	st.iters += 2u * (unsigned)cnt - 1u;
}

// barycentrics of the final hits, derived after the walk (primary kernel): the same operations on the same
// operands as src/triangle.cpp:13-18,26-28,55,60 -> the same bits as updating them on every accepted hit
__device__ __forceinline__ void finalBarycentrics(const uint4 *__restrict__ tris, const float (&org)[3][4], const Quad &Q, const int (&tid)[4],
												  float (&bu)[4], float (&bv)[4]) {
#pragma unroll
	for(int l = 0; l < 4; l++) {
		bu[l] = 0.0f; bv[l] = 0.0f;
		if(Q.dist[l] < __builtin_inff()) {
			const Tri t = loadTriVector(tris, tid[l]);
			const TriTerms tt = triTerms(t, org[0][0], org[1][0], org[2][0]);
			const float det = Q.d[0][l] * t.n[0] + Q.d[1][l] * t.n[1] + Q.d[2][l] * t.n[2];
			const float v = Q.d[0][l] * tt.t0v[0] + Q.d[1][l] * tt.t0v[1] + Q.d[2][l] * tt.t0v[2];
			const float u = Q.d[0][l] * tt.t1v[0] + Q.d[1][l] * tt.t1v[1] + Q.d[2][l] * tt.t1v[2];
			const float idet = recipExact(det);   // = 1.0f / det bit for bit (see recipExact)
			bu[l] = u * idet; bv[l] = v * idet;
		}
	}
}

// packet classification (wave-uniform): M_EXACT unless everything is finite; M_COH if additionally every ray THAT MATTERS has
// the same idir sign on each axis -- `oct` then holds those signs (bit k = negative on axis k).  A ray whose distance is -inf on
// entry (a masked lane as Scene::RayTrace / TraceLight set it up, src/scene_trace.cpp:112-115,:551-557) fails every slab test
// whatever planes it is given, so its signs are ignored: mirrored and shadow packets stay coherent although their masked lanes
// carry placeholder directions.
__device__ __forceinline__ int classify(bool fastOK, bool laneFinite, bool live, const float (&id)[3][4], const float (&dist)[4], int &oct) {
	oct = 0;
	if(!(fastOK && __all(laneFinite || !live))) return M_EXACT;
	bool coh = true;
#pragma unroll
	for(int k = 0; k < 3; k++) {
		bool anyNeg = false, anyPos = false;
#pragma unroll
		for(int l = 0; l < 4; l++) {
			const bool matters = dist[l] > -__builtin_inff();
			anyNeg |= matters && __float_as_int(id[k][l]) < 0;
			anyPos |= matters && __float_as_int(id[k][l]) >= 0;
		}
		const bool n = __any(anyNeg && live), p = __any(anyPos && live);
		coh = coh && !(n && p);
		oct |= n ? (1 << k) : 0;
	}
	return coh ? M_COH : M_FAST;
}

__device__ __forceinline__ bool originSaneDev(const float (&o)[3]) {
	return __builtin_fabsf(o[0]) <= 1.0e9f && __builtin_fabsf(o[1]) <= 1.0e9f && __builtin_fabsf(o[2]) <= 1.0e9f;
}
__device__ __forceinline__ bool finite4(const float (&v)[3][4]) {
	bool f = true;
#pragma unroll
	for(int c = 0; c < 3; c++)
#pragma unroll
		for(int l = 0; l < 4; l++) f = f && (__builtin_fabsf(v[c][l]) < __builtin_inff());
	return f;
}

__device__ __forceinline__ void flushStats(u64 *stats, const Counters &st, unsigned rays, int lane) {
	if(stats && lane == 0) {
		atomicAdd(&stats[0], (u64)st.intersects);
		atomicAdd(&stats[1], (u64)st.iters);
		atomicAdd(&stats[2], (u64)rays);
		atomicAdd(&stats[3], (u64)st.skips);
	}
}

// Trunc(Clamp(c * 255, 0, 255)) of ConvColor (src/render.cpp:11-17); Clamp = Min(Max(v, lo), hi), veclib/vecbase.h:75-77
__device__ __forceinline__ int convChannelW(float c) {
	float v = c * 255.0f;
	v = v > 0.0f ? v : 0.0f;
	v = v < 255.0f ? v : 255.0f;
	return (int)v;
}

// ---- primary kernel: RayGenerator::Generate + SafeInv + TraversePrimary<1,0> ----------------------
struct GenConst {
	float tright[3], tup[3], txyz[3][4], org[3];
};

// Multi-frame launches: ONE launch may trace up to SNAIL_MAX_BATCH frames of the same packet set (each with its own camera and output
// planes): block b takes frame b % nFrames at dispatch rank b / nFrames, so that the heaviest packets of ALL its frames start first.  A
// frame's tail -- its heaviest packets, ~0.2 ms whatever the launch holds -- and the launch overheads are then paid once per nFrames
// frames; what it costs is latency: a frame is complete when its launch is.
struct FrameOut {
	float *t, *u, *v;
	int *id;
	unsigned char *bgr; // packet-major B,G,R of the gVals[1] depth shading (src/scene_trace.cpp:128-137), 3 B/ray, or null
};
struct PrimaryArgs {
	const uint4 *nodes, *tris;
	const uint4 *pf; // the record-prefetching loop's copy of the tree (SnailScene::dPF; used when `pack` is set)
	int nFrames;
	GenConst g[SNAIL_MAX_BATCH];
	FrameOut out[SNAIL_MAX_BATCH];
	const uint4 *rel[SNAIL_MAX_BATCH]; // per frame: the node records relative to that frame's camera position (SnailScene::relFor; used when `pack` is set)
	int resx, resy, x0, y0, w, h; // rect (frame layout) ...
	const int2 *packetXY;		  // ... or explicit packet list (packet-major layout)
	int nPackets, pw, ph;		  // packet grid of the rect
	int nBlocks;
	int pack;                     // at most 2^20 node slots: one-word stack entries in the hand-written walks
	int packetMajor;              // rect mode: store packet-major ([cy*pw+cx][256], the reference's quad order) instead of frame layout
	int fastOK;
	u64 *stats;
	unsigned *cost; // diagnostic (k_primary_diag only): per packet 8 words {iters, intersects, shader cycles, start time >> 6, triangle records fetched, leaf bodies, 0, 0}
	const int *order; // dispatch order (block -> slot index, a permutation of [0, nSlots)) or null = the built-in interleave
	int *slotCost;	  // out, per slot: node visits of its packet (0 for a slot without a packet) or null
	int nSlots;		  // rect mode: nBlocks; list mode: nPackets
	int *defer;		// [0] = count, [1] = finished blocks of the M_EXACT pass, [2..] = frame * nSlots + logical index of deferred packets
};

typedef const PrimaryArgs __attribute__((address_space(4))) *PrimaryArgsK;
typedef const FrameOut __attribute__((address_space(4))) *FrameOutK;
// the kernels that run primaryPacket take ONE argument, the PrimaryArgs by value: it sits at offset 0 of the kernel-argument segment
__device__ __forceinline__ PrimaryArgsK lateArgs() {
	PrimaryArgsK p = (PrimaryArgsK)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(p));
	return p;
}

#define LDS_FLOATS_PER_WAVE (64 * 12 + 64)

// Block -> packet mapping of the primary kernel.  ONE WAVE PER BLOCK: packet costs vary ~10x (p5 66 K .. max
// 570 K cycles on the atrium frame), and a multi-wave block holds all its slots until its slowest wave ends.
// Blocks are dealt round-robin over the 8 XCDs (each with a private L2): XCD x receives blocks x, x+8, ...
// Give each XCD whole 4x4-packet REGIONS (64x64 px; its 16 consecutive blocks), regions interleaved over the
// image: neighbouring packets (same BVH subtrees) share an L2, and every XCD samples the whole frame, so a
// heavy image band does not land on one XCD.
__device__ __forceinline__ int interleave16(int b) { // -> logical index; 16 consecutive logical indices per XCD turn
	const int xcd = b & 7, j = b >> 3;
	return (((j >> 4) << 3) + xcd) * 16 + (j & 15);
}

// One primary packet.  EXACTPASS=false: the main kernel -- M_COH (one specialised walk per sign octant) and M_FAST;
// a packet that needs M_EXACT (a non-finite reciprocal: practically never for camera rays) is appended to A.defer and
// left to the second, tiny kernel (EXACTPASS=true).  Keeping the select-based M_EXACT walk out of the main kernel
// takes its register allocation from 128 to 84-96 VGPRs, i.e. from 4 to 5 waves per SIMD.
template <bool DEEP, bool EXACTPASS, bool DIAG = false>
__device__ __forceinline__ void primaryPacket(const PrimaryArgs &A, const int li, const int fi, float *lds) {
	const int lane = threadIdx.x & 63;
	const GenConst &G = A.g[fi];

	int px, py, pidx;
	if(A.packetXY) {
		pidx = li;
		if((unsigned)pidx >= (unsigned)A.nPackets) return;
		int2 xy = A.packetXY[pidx];
		px = __builtin_amdgcn_readfirstlane(xy.x);
		py = __builtin_amdgcn_readfirstlane(xy.y);
	} else {
		const int nrx = (A.pw + 3) >> 2;
		const int region = li >> 4, k = li & 15;
		const int rx = region % nrx, ry = region / nrx;
		const int cx = rx * 4 + (k & 3), cy = ry * 4 + (k >> 2);
		if(cx >= A.pw || cy >= A.ph) {
			if(A.slotCost && fi == 0 && lane == 0 && (unsigned)li < (unsigned)A.nSlots) A.slotCost[li] = 0;
			return;
		}
		px = A.x0 + cx * 16;
		py = A.y0 + cy * 16;
		pidx = cy * A.pw + cx;
	}

	const u64 tStart = DIAG ? __builtin_amdgcn_s_memtime() : 0;
	// ---- RayGenerator::Generate, level 3 (src/ray_generator.cpp:23-47): quad ty*4+k, lane j -> pixel (x+4k+j, y+ty)
	Quad Q;
	const int ty = lane >> 2, k4 = lane & 3;
#pragma unroll
	for(int l = 0; l < 4; l++) {
		const float xoff = (float)(px + (l >= 2 ? 2 : 0));
		const float yoff = (float)(py - (l >= 2 ? 1 : 0));
		const float tposx = (float)(4 * k4) + xoff;
		const float tposy = (float)ty + yoff;
		const float p0 = G.tright[0] * tposx + (G.tup[0] * tposy + G.txyz[0][l]);
		const float p1 = G.tright[1] * tposx + (G.tup[1] * tposy + G.txyz[1][l]);
		const float p2 = G.tright[2] * tposx + (G.tup[2] * tposy + G.txyz[2][l]);
		const float rs = recipExact(__builtin_sqrtf(p0 * p0 + p1 * p1 + p2 * p2));
		Q.d[0][l] = p0 * rs; Q.d[1][l] = p1 * rs; Q.d[2][l] = p2 * rs;
#pragma unroll
		for(int c = 0; c < 3; c++) Q.id[c][l] = recipExact(Q.d[c][l] + 0.00000001f); // SafeInv (src/rtbase.h:117-120)
		Q.dist[l] = __builtin_inff();											   // src/scene_trace.cpp:112-115
	}
	int tid[4] = {0, 0, 0, 0};
	float bu[4] = {0, 0, 0, 0}, bv[4] = {0, 0, 0, 0};
	float org[3][4];
#pragma unroll
	for(int c = 0; c < 3; c++)
#pragma unroll
		for(int l = 0; l < 4; l++) org[c][l] = G.org[c];

	Counters st = {0, 0, 0, 0, 0};
	int oct;
	const int mode = classify(A.fastOK != 0, finite4(Q.id) && finite4(Q.d), true, Q.id, Q.dist, oct);
	if(EXACTPASS) walk<true, false, false, M_EXACT, false, DEEP, true>(A.nodes, A.tris, 64, lane, org, Q, 15u, tid, bu, bv, lds, st);
	else if(mode == M_EXACT) {
		if(lane == 0) A.defer[2 + atomicAdd(&A.defer[0], 1)] = fi * A.nSlots + li;
		return;
	} else if(DEEP) { // depth > 62: the C++ walk with its second stack register pair
		if(mode == M_COH) walk<true, false, false, M_COH, false, DEEP, true>(A.nodes, A.tris, 64, lane, org, Q, 15u, tid, bu, bv, lds, st, oct);
		else walk<true, false, false, M_FAST, false, DEEP, true>(A.nodes, A.tris, 64, lane, org, Q, 15u, tid, bu, bv, lds, st);
	} else if(A.pack && !DIAG) { // (the counting build walks with the plain loop: same visits, same tests -- and this compiler cannot place the
		// record-prefetching loop's three record sets beside the extra counters: "illegal VGPR to SGPR copy")
		const uint4 *pn = (SNAIL_NODE_PREFETCH && SNAIL_REL_NODES) ? A.rel[fi] : SNAIL_PACK_NODES(A);   // this frame's camera-relative records
		if(mode == M_COH) walkSharedAsm<false, true, true, false, false, true>(pn, A.tris, 64, lane, org, Q, 15u, tid, bu, bv, lds, st, oct);
		else walkSharedAsm<false, false, true, false, false, true>(pn, A.tris, 64, lane, org, Q, 15u, tid, bu, bv, lds, st, 0);
	} else if(mode == M_COH) walkSharedAsm<false, true, false, false, false, true>(A.nodes, A.tris, 64, lane, org, Q, 15u, tid, bu, bv, lds, st, oct);
	else walkSharedAsm<false, false, false, false, false, true>(A.nodes, A.tris, 64, lane, org, Q, 15u, tid, bu, bv, lds, st, 0);
	// The epilogue reads its arguments (output planes, layout) through an opaque copy of the kernel-argument pointer: otherwise their loads
	// are hoisted to the top of the kernel and ~20 SGPRs stay live across the walk, whose hand-written loop already pins 24.
	const PrimaryArgsK E = lateArgs();
	const FrameOutK F = &E->out[fi];
	if(F->u || F->v) finalBarycentrics(E->tris, org, Q, tid, bu, bv); // (the staged shading pipeline asks for t and triId only)

	flushStats(E->stats, st, 256u, lane);
	if(E->slotCost && fi == 0 && lane == 0) E->slotCost[li] = (int)st.iters;
	if(DIAG && E->cost && lane == 0) {
		const u64 tEnd = __builtin_amdgcn_s_memtime();
		unsigned *c = E->cost + (size_t)pidx * 8;
		c[0] = st.iters; c[1] = st.intersects; c[2] = (unsigned)(tEnd - tStart); c[3] = (unsigned)(tStart >> 6);
		c[4] = st.fetched; c[5] = st.leaves; c[6] = 0; c[7] = 0;
	}

	if(F->bgr) { // fused gVals[1] depth shading + ConvColor: c = Inv(t) * (20, 250, 2), bytes B,G,R (same operations as k_shade_depth)
		unsigned bytes[12];
#pragma unroll
		for(int l = 0; l < 4; l++) {
			const float dist = 1.0f / Q.dist[l];
			bytes[l * 3 + 0] = (unsigned)convChannelW(dist * 2.0f); bytes[l * 3 + 1] = (unsigned)convChannelW(dist * 250.0f); bytes[l * 3 + 2] = (unsigned)convChannelW(dist * 20.0f);
		}
		unsigned *o = (unsigned *)(F->bgr + ((size_t)pidx * 256 + (size_t)lane * 4) * 3);
#pragma unroll
		for(int k = 0; k < 3; k++) o[k] = bytes[4 * k] | (bytes[4 * k + 1] << 8) | (bytes[4 * k + 2] << 16) | (bytes[4 * k + 3] << 24);
	}
	if(E->packetXY || E->packetMajor) { // packet-major (Context layout)
		const size_t o = (size_t)pidx * 256 + (size_t)lane * 4;
		if(F->t) *(float4 *)(F->t + o) = make_float4(Q.dist[0], Q.dist[1], Q.dist[2], Q.dist[3]);
		if(F->u) *(float4 *)(F->u + o) = make_float4(bu[0], bu[1], bu[2], bu[3]);
		if(F->v) *(float4 *)(F->v + o) = make_float4(bv[0], bv[1], bv[2], bv[3]);
		if(F->id) *(int4 *)(F->id + o) = make_int4(tid[0], tid[1], tid[2], tid[3]);
	} else {
		const int yy = py + ty, xx = px + k4 * 4;
		const int xlim = min(E->resx, E->x0 + E->w), ylim = min(E->resy, E->y0 + E->h);
		if(yy < ylim) {
			const size_t o = (size_t)yy * E->resx + xx;
			if(xx + 3 < xlim && (E->resx & 3) == 0) {
				if(F->t) *(float4 *)(F->t + o) = make_float4(Q.dist[0], Q.dist[1], Q.dist[2], Q.dist[3]);
				if(F->u) *(float4 *)(F->u + o) = make_float4(bu[0], bu[1], bu[2], bu[3]);
				if(F->v) *(float4 *)(F->v + o) = make_float4(bv[0], bv[1], bv[2], bv[3]);
				if(F->id) *(int4 *)(F->id + o) = make_int4(tid[0], tid[1], tid[2], tid[3]);
			} else {
#pragma unroll
				for(int l = 0; l < 4; l++)
					if(xx + l < xlim) {
						if(F->t) F->t[o + l] = Q.dist[l];
						if(F->u) F->u[o + l] = bu[l];
						if(F->v) F->v[o + l] = bv[l];
						if(F->id) F->id[o + l] = tid[l];
					}
			}
		}
	}
}

#ifndef SNAIL_PRIMARY_WAVES
#define SNAIL_PRIMARY_WAVES 6 // occupancy target of the primary kernel (76 VGPRs by itself; 7 = 72 VGPRs measured separately: profiles/README.md)
#endif
// SNAIL_BLOCK_WAVES packets per workgroup (one per wave; waves end independently, nothing of the block is shared): the XCD's
// region turn is kept -- wave w of hardware block B takes entry (B >> 3) * W + w of XCD (B & 7)'s list.
#ifndef SNAIL_BLOCK_WAVES
#define SNAIL_BLOCK_WAVES 1 // 2 and 4 measured slower (22.65 / 21.80 vs 23.40 Grays/s): residency is not limited by workgroup slots
#endif
#ifndef SNAIL_PRIO_RANK
#define SNAIL_PRIO_RANK 1024
#endif
template <bool DEEP>
__global__ __launch_bounds__(64 * SNAIL_BLOCK_WAVES) __attribute__((amdgpu_waves_per_eu(SNAIL_PRIMARY_WAVES))) void k_primary(PrimaryArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	const int wv = SNAIL_BLOCK_WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
	int b = (int)blockIdx.x * SNAIL_BLOCK_WAVES + wv; // dispatch index of this wave's packet
	int fi = 0;
	if(A.nFrames > 1) { fi = b % A.nFrames; b = b / A.nFrames; } // several frames in one launch: frame fi at dispatch rank b
	int li;
	if(SNAIL_BLOCK_WAVES > 1) {
		const int xcd = (int)blockIdx.x & 7, j = ((int)blockIdx.x >> 3) * SNAIL_BLOCK_WAVES + wv;   // (single-frame launches only)
		li = (((j >> 4) << 3) + xcd) * 16 + (j & 15);
	} else li = interleave16(b);
	if(A.order) { // fed-back dispatch order (snail_order_from_cost_dev): heaviest packets of the previous frame first
		if(b >= A.nSlots) return;
		li = __builtin_amdgcn_readfirstlane(A.order[b]);
		if((unsigned)li >= (unsigned)A.nSlots) return;
		// issue priority by rank in the fed-back order: the heaviest 1024 packets (one per SIMD) outrank whatever shares their SIMD, the
		// next 1024 come second, the next 2048 third -- the frame ends with its heaviest packets, so they should never wait for an issue
		// slot (priority, then age: MI355X_MICROARCH.md "Two waves per SIMD").  Same box, 4 frames in flight: 23.64 vs 23.10 Grays/s,
		// lone frame 0.231 vs 0.234 ms (profiles/README.md, round 2); 0 = off
#if SNAIL_PRIO_RANK > 0
		if(b < SNAIL_PRIO_RANK) __builtin_amdgcn_s_setprio(3);
		else if(b < 2 * SNAIL_PRIO_RANK) __builtin_amdgcn_s_setprio(2);
		else if(b < 4 * SNAIL_PRIO_RANK) __builtin_amdgcn_s_setprio(1);
#endif
	}
	primaryPacket<DEEP, false>(A, li, fi, lds);
}
// the diagnostic build of the same packet code (snail_account_packets): per-packet cost records; never on a product path
template <bool DEEP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SNAIL_PRIMARY_WAVES))) void k_primary_diag(PrimaryArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	primaryPacket<DEEP, false, true>(A, interleave16((int)blockIdx.x), 0, lds);
}

// Dispatch order from per-slot costs: slots in (approximately) descending cost -- a counting sort over 4096 cost classes, one
// workgroup, no scratch.  The hardware dispatcher hands blocks to CUs in index order, so this is longest-processing-time-first
// scheduling of the packets; ties and the order inside a class are arbitrary (results never depend on the dispatch order).
// 512 threads = two waves per SIMD, so the workgroup fits beside frames that hold 6 of the 8 wave slots of every SIMD.  Alone it
// takes 20 us; beside four frames in flight its waves used to get a sixth of their SIMDs' issue slots (53-110 us with 256 or 1024
// threads, round 2): it now runs at s_setprio 3 -- above every traversal wave of its SIMDs (priority, then age) -- on its slot's
// stream once every `order_refresh` frames of a moving camera, while the other streams keep the machine full.
#define ORDER_THREADS 512
__global__ __launch_bounds__(ORDER_THREADS) void k_order_from_cost(const int *__restrict__ cost, int n, int *__restrict__ order) {
	__builtin_amdgcn_s_setprio(3);
	constexpr int T = ORDER_THREADS, PER = 4096 / T;
	__shared__ int bins[4096];
	__shared__ int part[T];
	__shared__ int maxCost;
	const int tid = (int)threadIdx.x;
	if(tid == 0) maxCost = 0;
	for(int i = tid; i < 4096; i += T) bins[i] = 0;
	__syncthreads();
	int m = 0;
	for(int i = tid; i < n; i += T) m = max(m, cost[i]);
	atomicMax(&maxCost, m);
	__syncthreads();
	int shift = 0;
	while((maxCost >> shift) > 4095) shift++;
	for(int i = tid; i < n; i += T) atomicAdd(&bins[4095 - min(max(cost[i], 0) >> shift, 4095)], 1); // class 0 = heaviest
	__syncthreads();
	// exclusive scan of the 4096 class counts: PER consecutive classes per thread, then a Hillis-Steele scan of the partial sums
	int sum = 0;
	for(int k = 0; k < PER; k++) sum += bins[tid * PER + k];
	part[tid] = sum;
	__syncthreads();
	for(int d = 1; d < T; d <<= 1) {
		const int add = tid >= d ? part[tid - d] : 0;
		__syncthreads();
		part[tid] += add;
		__syncthreads();
	}
	int run = part[tid] - sum;
	for(int k = 0; k < PER; k++) {
		const int c = bins[tid * PER + k];
		bins[tid * PER + k] = run;
		run += c;
	}
	__syncthreads();
	for(int i = tid; i < n; i += T) order[atomicAdd(&bins[4095 - min(max(cost[i], 0) >> shift, 4095)], 1)] = i;
}

// the deferred M_EXACT packets (grid-stride over the list; the last block to finish re-arms the list for its next use)
template <bool DEEP>
__global__ __launch_bounds__(64) void k_primary_exact(PrimaryArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	const int n = __builtin_amdgcn_readfirstlane(A.defer[0]);
	if(n == 0) return; // nothing was deferred (the rule): the list is armed as it stands, no fence, no counter
	for(int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
		const int e = __builtin_amdgcn_readfirstlane(A.defer[2 + i]);
		primaryPacket<DEEP, true>(A, e % A.nSlots, e / A.nSlots, lds);
	}
	__threadfence();
	if((threadIdx.x & 63) == 0 && atomicAdd(&A.defer[1], 1) == (int)gridDim.x - 1) { A.defer[0] = 0; A.defer[1] = 0; }
}

// ---- Scene::RayTrace, simple-shading configuration, fused per packet (primary walk + one shadow walk per light) ----
__device__ __forceinline__ void loadQuad3(const float *base, size_t quad, float (&v)[3][4]) {
	const float4 *p = (const float4 *)(base + quad * 12);
	float4 x = p[0], y = p[1], z = p[2];
	v[0][0] = x.x; v[0][1] = x.y; v[0][2] = x.z; v[0][3] = x.w;
	v[1][0] = y.x; v[1][1] = y.y; v[1][2] = y.z; v[1][3] = y.w;
	v[2][0] = z.x; v[2][1] = z.y; v[2][2] = z.z; v[2][3] = z.w;
}

// ---- Scene::RayTrace after the traversal: samples, reflection rays, lights (BASELINE config 3) --------------------------
// The frame is shaded in stages over packet-major buffers that stay in HBM (288 GB: a 1080p frame's intermediate state is
// ~150 MB), each stage with its own register budget instead of one kernel holding two walks and all samples at once:
//   k_primary                        hits of the primary packets                               (the bench kernel, 5 waves/SIMD)
//   k_light<SRC_PRIMARY>             one wave per (packet, light): samples -> shadow packet -> TraverseShadow -> the
//                                    surviving distances, 4 B/ray/light            Scene::TraceLight, src/scene_trace.cpp:523-566
//   k_final<SRC_PRIMARY, DST_FRAME>  samples, attenuation and accumulation per light from those distances, B,G,R store
//                                                                                  src/scene_trace.cpp:567-601, 484-512
// and with gVals[7] (one mirrored bounce), between k_primary and k_light:
//   k_final<SRC_PRIMARY, DST_MIRROR> samples -> mirrored rays + lane masks          Scene::TraceReflection, :603-618
//   k_rays<false,true>               TraversePrimary<0,1> of the mirrored packets
//   k_light<SRC_MIRROR>, k_final<SRC_MIRROR, DST_COLOR>   the nested RayTrace of the mirrored packets -> colour per ray (float)
// k_final<SRC_PRIMARY, DST_FRAME> then blends diffuse += (colour - diffuse) * 0.3 (:462-465).  Samples and shadow rays are
// recomputed (the same operations on the same operands, hence the same bits) wherever they are needed: a few hundred VALU
// instructions per packet against the thousands of a walk, and no kernel carries state across a walk that the walk does not use.
enum { SRC_PRIMARY = 0, SRC_MIRROR = 1 };
enum { DST_FRAME = 0, DST_MIRROR = 1, DST_COLOR = 2, DST_CONTINUE = 3 };
struct ShadeArgs {
	const uint4 *nodes, *tris;
	const uint4 *pf; // the record-prefetching loop's copy of the tree (SnailScene::dPF; used when `pack` is set)
	GenConst g;
	int resx, resy, pw, ph, fastOK;
	const int2 *packetXY; // explicit packet list (tile sharding): packet li = packetXY[li], intermediates and output indexed by li; or null = the frame's grid
	int nPackets;        // packet slots of the intermediates: list length, or pw * ph
	unsigned char *bgrPackets; // list mode: packet-major B,G,R output [nPackets][256][3] instead of the frame
	float *colPackets;         // list mode: the packets' colours as FLOATS [nPackets][256][3] (r, g, b) instead of bytes: the input of dev::k_aa_reduce
	int nBlocks;         // grid.x of the per-packet kernels (packets padded to whole XCD regions)
	int pack;            // at most 2^20 node slots: one-word stack entries in the hand-written walks
	int nLights;
	float lights[SNAIL_MAX_LIGHTS][7];
	const uint4 *relLight[SNAIL_MAX_LIGHTS]; // per light: the node records relative to its position (SnailScene::relFor; set by launchLights when `pack` is set)
	float ambient[3], color[3];
	const float *hitT;   // primary hits, packet-major
	const int *hitId;
	float *rOrg, *rDir, *rIDir; // mirrored packets (Context layout: per quad x[4], y[4], z[4])
	unsigned char *rMask;
	float *rDist;
	int *rObj;
	float *rCol;         // colour of the mirrored rays, [packet][256][3]
	float *sDist;        // shadow distances after TraverseShadow, [light][packet][256]
	int blend;           // DST_FRAME: diffuse += (rCol - diffuse) * 0.3 on hit lanes
	const unsigned char *selIn; // DST_CONTINUE: the caller's transparency selector, 1 byte per quad (low 4 bits = lanes), packet-major
	int *defer;          // [0] = count, [1] = finished blocks of the M_EXACT pass, [16..] = light * nBlocks + grid index of deferred shadow packets
	unsigned char *frame;
	int pitch;
	u64 *stats;
};


struct PacketPos {
	int px, py;
	size_t pidx;
	bool valid;
};
__device__ __forceinline__ PacketPos packetOf(const ShadeArgs &A, int li) {
	PacketPos P;
	if(A.packetXY) {
		P.valid = li < A.nPackets;
		const int2 xy = A.packetXY[P.valid ? li : 0];
		P.px = __builtin_amdgcn_readfirstlane(xy.x); P.py = __builtin_amdgcn_readfirstlane(xy.y);
		P.pidx = (size_t)li;
		return P;
	}
	const int nrx = (A.pw + 3) >> 2;
	const int region = li >> 4, kk = li & 15;
	const int cx = (region % nrx) * 4 + (kk & 3), cy = (region / nrx) * 4 + (kk >> 2);
	P.valid = cx < A.pw && cy < A.ph;
	P.px = cx * 16; P.py = cy * 16;
	P.pidx = (size_t)cy * A.pw + cx;
	return P;
}

// the packet's rays and hits, then its samples: src/scene_trace.cpp:366-379,397-452 + SimpleMaterial::Shade_
// (src/shading/simple_material.h:19-28).  sdn = Abs(rays.Dir | normal) (the colour is applied at the end).
struct Samples {
	bool hit[4];
	float pos[3][4], nrm[3][4], sdn[4];
};
template <int SRC>
__device__ __forceinline__ void loadSamples(const ShadeArgs &A, const PacketPos &P, int lane, float (&d)[3][4], Samples &S) {
	const size_t quad = P.pidx * 64 + lane;
	const float inf = __builtin_inff();
	float org[3][4], dist[4];
	int tid[4];
	unsigned mask4 = 15u;
	if(SRC == SRC_MIRROR) {
		loadQuad3(A.rDir, quad, d);
		loadQuad3(A.rOrg, quad, org);
		mask4 = A.rMask[quad] & 15u;
		const float4 dv = *(const float4 *)(A.rDist + quad * 4);
		const int4 ov = *(const int4 *)(A.rObj + quad * 4);
		dist[0] = dv.x; dist[1] = dv.y; dist[2] = dv.z; dist[3] = dv.w;
		tid[0] = ov.x; tid[1] = ov.y; tid[2] = ov.z; tid[3] = ov.w;
	} else {
		const int ty = lane >> 2, k4 = lane & 3;
#pragma unroll
		for(int l = 0; l < 4; l++) { // RayGenerator::Generate, exactly as in primaryPacket
			const float xoff = (float)(P.px + (l >= 2 ? 2 : 0)), yoff = (float)(P.py - (l >= 2 ? 1 : 0));
			const float tposx = (float)(4 * k4) + xoff, tposy = (float)ty + yoff;
			const float p0 = A.g.tright[0] * tposx + (A.g.tup[0] * tposy + A.g.txyz[0][l]);
			const float p1 = A.g.tright[1] * tposx + (A.g.tup[1] * tposy + A.g.txyz[1][l]);
			const float p2 = A.g.tright[2] * tposx + (A.g.tup[2] * tposy + A.g.txyz[2][l]);
			const float rs = recipExact(__builtin_sqrtf(p0 * p0 + p1 * p1 + p2 * p2));
			d[0][l] = p0 * rs; d[1][l] = p1 * rs; d[2][l] = p2 * rs;
#pragma unroll
			for(int c = 0; c < 3; c++) org[c][l] = A.g.org[c];
		}
		const float4 dv = *(const float4 *)(A.hitT + quad * 4);
		const int4 ov = *(const int4 *)(A.hitId + quad * 4);
		dist[0] = dv.x; dist[1] = dv.y; dist[2] = dv.z; dist[3] = dv.w;
		tid[0] = ov.x; tid[1] = ov.y; tid[2] = ov.z; tid[3] = ov.w;
	}
#pragma unroll
	for(int l = 0; l < 4; l++) {
		S.hit[l] = dist[l] < inf && ((mask4 >> l) & 1u) != 0;
#pragma unroll
		for(int c = 0; c < 3; c++) S.pos[c][l] = d[c][l] * dist[l] + org[c][l];
		const float4 pl = *(const float4 *)((const float *)(A.tris + (size_t)(S.hit[l] ? tid[l] : 0) * 4) + 12); // GetNormal = plane.xyz (src/bvh/tree.h:40-42)
		S.nrm[0][l] = S.hit[l] ? pl.x : 0.0f; S.nrm[1][l] = S.hit[l] ? pl.y : 0.0f; S.nrm[2][l] = S.hit[l] ? pl.z : 0.0f;
		const float dn = d[0][l] * S.nrm[0][l] + d[1][l] * S.nrm[1][l] + d[2][l] * S.nrm[2][l];
		S.sdn[l] = S.hit[l] ? __builtin_fabsf(dn) : 0.0f;
	}
}

// bbox of the packet's hit points: per SSE slot over the quads, then Minimize / Maximize (src/scene_trace.cpp:375-376,
// src/rtbase_math.h:63-64)
__device__ __forceinline__ void hitBounds(const Samples &S, float (&tMin)[3], float (&tMax)[3]) {
	// min / max of finite values (+-inf for lanes without a hit) are exact and order-independent up to the sign of a zero, which
	// neither the comparisons nor the squared differences of BoxPointDistanceSq observe: fold the 4 SSE slots first, then ONE
	// wave reduction per component instead of four
	const float inf = __builtin_inff();
#pragma unroll
	for(int c = 0; c < 3; c++) {
		float mn = inf, mx = -inf;
#pragma unroll
		for(int l = 0; l < 4; l++) {
			mn = vmin(mn, S.hit[l] ? S.pos[c][l] : inf);
			mx = vmax(mx, S.hit[l] ? S.pos[c][l] : -inf);
		}
		tMin[c] = waveMin(mn);
		tMax[c] = waveMax(mx);
	}
}
// the packet-level light cull: BoxPointDistanceSq(bbox, light) > radSq (src/scene_trace.cpp:494-501, src/funcs.cpp:8-49); wave-uniform
__device__ __forceinline__ bool lightCulled(const float (&tMin)[3], const float (&tMax)[3], const float (&lp)[3], float radSq) {
	float sq = 0.0f;
#pragma unroll
	for(int c = 0; c < 3; c++) {
		if(lp[c] < tMin[c]) { const float dl = lp[c] - tMin[c]; sq += dl * dl; }
		else if(lp[c] > tMax[c]) { const float dl = lp[c] - tMax[c]; sq += dl * dl; }
	}
	return sq > radSq;
}
// one lane of the shadow packet (src/scene_trace.cpp:538-558): fromLight = (position - light) / |position - light|, its N.L, and
// the ray length 0.9999 |..| when N.L > 0 (-inf = masked otherwise).  Lanes without a hit: zeros, masked.
__device__ __forceinline__ void shadowLane(const Samples &S, int l, const float (&lp)[3], float (&sd)[3], float &distance, float &dotv, float &sdist) {
	sd[0] = sd[1] = sd[2] = 0.0f;
	sdist = -__builtin_inff(); distance = 0.0f; dotv = 0.0f;
	if(S.hit[l]) {
		float lv[3] = {S.pos[0][l] - lp[0], S.pos[1][l] - lp[1], S.pos[2][l] - lp[2]};
		if(lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2] < 0.0001f) { lv[0] = 0.0f; lv[1] = 1.0f; lv[2] = 0.0f; }
		distance = __builtin_sqrtf(lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2]);
		const float inv = 1.0f / distance;
#pragma unroll
		for(int c = 0; c < 3; c++) sd[c] = lv[c] * inv;
		dotv = S.nrm[0][l] * sd[0] + S.nrm[1][l] * sd[1] + S.nrm[2][l] * sd[2];
		if(dotv > 0.0f) sdist = distance * 0.9999f;
	}
}

// ---- one (packet, light): the shadow packet and its walk ----
// EXACTPASS=false: the main kernel, walks in M_COH / M_FAST; a shadow packet that needs M_EXACT (a non-finite value: practically
// never) is appended to A.defer untouched -- nothing has been written or counted for it -- and traced by the second, tiny
// launch (EXACTPASS=true; M_EXACT is valid for any packet).  As in the primary kernel this keeps the select-based walk out
// of the main kernel's register allocation.
template <bool DEEP, int SRC, bool EXACTPASS>
__device__ __forceinline__ void lightPacket(const ShadeArgs &A, const int li, const int n, float *lds) {
	const int lane = threadIdx.x & 63;
	const PacketPos P = packetOf(A, li);
	if(!P.valid) return;
	const float lp[3] = {A.lights[n][0], A.lights[n][1], A.lights[n][2]};
	const float radius = A.lights[n][6], radSq = radius * radius;
	Quad Q;
	{
		float d[3][4];
		Samples S;
		loadSamples<SRC>(A, P, lane, d, S);
		float tMin[3], tMax[3];
		hitBounds(S, tMin, tMax);
		if(lightCulled(tMin, tMax, lp, radSq)) return;
#pragma unroll
		for(int l = 0; l < 4; l++) {
			float sd[3], distance, dotv;
			shadowLane(S, l, lp, sd, distance, dotv, Q.dist[l]);
#pragma unroll
			for(int c = 0; c < 3; c++) { Q.d[c][l] = sd[c]; Q.id[c][l] = S.hit[l] ? 1.0f / (sd[c] + 0.00000001f) : 0.0f; }
		}
	}
	unsigned rays = 0;
#pragma unroll
	for(int l = 0; l < 4; l++) rays += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(Q.dist[l] > 0.0f));
	float lorg[3][4];
#pragma unroll
	for(int c = 0; c < 3; c++)
#pragma unroll
		for(int l = 0; l < 4; l++) lorg[c][l] = lp[c];
	Counters st = {0, 0, 0, 0, 0};
	int stid[4];
	float bu[4], bv[4];
	if(EXACTPASS) walk<true, false, true, M_EXACT, false, DEEP, false>(A.nodes, A.tris, 64, lane, lorg, Q, 15u, stid, bu, bv, lds, st);
	else {
		const bool fin = finite4(Q.id) && finite4(Q.d);
		int oct;
		const int mode = classify(A.fastOK != 0 && originSaneDev(lp), fin, true, Q.id, Q.dist, oct);
		if(mode == M_EXACT) {
			if(lane == 0) A.defer[16 + atomicAdd(&A.defer[0], 1)] = n * A.nBlocks + li;
			return;
		}
		if(DEEP) {
			if(mode == M_COH) walk<true, false, true, M_COH, false, DEEP, false>(A.nodes, A.tris, 64, lane, lorg, Q, 15u, stid, bu, bv, lds, st, oct);
			else walk<true, false, true, M_FAST, false, DEEP, false>(A.nodes, A.tris, 64, lane, lorg, Q, 15u, stid, bu, bv, lds, st);
		} else if(A.pack) {
			const uint4 *pn = (SNAIL_NODE_PREFETCH && SNAIL_REL_SHADOW) ? A.relLight[n] : SNAIL_PACK_NODES(A);   // this light's relative records
			if(mode == M_COH) walkSharedAsm<true, true, true, false, false, false>(pn, A.tris, 64, lane, lorg, Q, 15u, stid, bu, bv, lds, st, oct);
			else walkSharedAsm<true, false, true, false, false, false>(pn, A.tris, 64, lane, lorg, Q, 15u, stid, bu, bv, lds, st, 0);
		} else if(mode == M_COH) walkSharedAsm<true, true, false, false, false, false>(A.nodes, A.tris, 64, lane, lorg, Q, 15u, stid, bu, bv, lds, st, oct);
		else walkSharedAsm<true, false, false, false, false, false>(A.nodes, A.tris, 64, lane, lorg, Q, 15u, stid, bu, bv, lds, st, 0);
	}
	flushStats(A.stats, st, rays, lane);
	const size_t packets = (size_t)A.nPackets;
	*(float4 *)(A.sDist + ((size_t)n * packets + P.pidx) * 256 + (size_t)lane * 4) = make_float4(Q.dist[0], Q.dist[1], Q.dist[2], Q.dist[3]);
}

template <bool DEEP, int SRC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6))) void k_light(ShadeArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	lightPacket<DEEP, SRC, false>(A, interleave16((int)blockIdx.x), (int)blockIdx.y, lds);
}
template <bool DEEP, int SRC>
__global__ __launch_bounds__(64) void k_light_exact(ShadeArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	const int cnt = __builtin_amdgcn_readfirstlane(A.defer[0]);
	if(cnt == 0) return;
	for(int i = (int)blockIdx.x; i < cnt; i += (int)gridDim.x) {
		const int e = __builtin_amdgcn_readfirstlane(A.defer[16 + i]);
		lightPacket<DEEP, SRC, true>(A, e % A.nBlocks, e / A.nBlocks, lds);
	}
	__threadfence();
	if((threadIdx.x & 63) == 0 && atomicAdd(&A.defer[1], 1) == (int)gridDim.x - 1) { A.defer[0] = 0; A.defer[1] = 0; }
}

// ---- one packet: samples -> mirrored rays (DST_MIRROR), or samples + the lights' contributions -> colour ----
template <int SRC, int DST>
__global__ __launch_bounds__(64) void k_final(ShadeArgs A) {
	const int lane = threadIdx.x & 63;
	const PacketPos P = packetOf(A, interleave16((int)blockIdx.x));
	if(!P.valid) return;
	const size_t quad = P.pidx * 64 + lane;
	const float inf = __builtin_inff();
	float d[3][4];
	Samples S;
	loadSamples<SRC>(A, P, lane, d, S);

	if(DST == DST_CONTINUE) {
		// Scene::TraceTransparency (src/scene_trace.cpp:620-634): the packet's rays continue behind their hits -- origin = dir * (t + 0.001)
		// + origin, dir and idir (= SafeInv(dir), as the caller's RayGroup carries them) unchanged -- for the lanes of the caller's selector
		// (transSel: lanes whose material is transparent, src/scene_trace.cpp:190,306,349,472; only lanes with a hit can carry one).
		// Lanes outside the selector: zeros, distance -inf, exactly as the mirrored packets of DST_MIRROR.
		float ro[3][4], rd[3][4], ri[3][4], rdist[4];
		const float4 tv4 = *(const float4 *)(A.hitT + quad * 4);
		const float tt[4] = {tv4.x, tv4.y, tv4.z, tv4.w};
		const unsigned selq = A.selIn[quad] & 15u;
		unsigned sel = 0;
#pragma unroll
		for(int l = 0; l < 4; l++) {
			const bool on = S.hit[l] && ((selq >> l) & 1u) != 0;
			const float tl = tt[l] + 0.001f;
#pragma unroll
			for(int c = 0; c < 3; c++) {
				rd[c][l] = on ? d[c][l] : 0.0f;
				ro[c][l] = on ? d[c][l] * tl + A.g.org[c] : 0.0f;
				ri[c][l] = recipExact(rd[c][l] + 0.00000001f);
			}
			rdist[l] = on ? inf : -inf;
			sel |= on ? (1u << l) : 0u;
		}
		float4 *po = (float4 *)(A.rOrg + quad * 12), *pd = (float4 *)(A.rDir + quad * 12), *pi = (float4 *)(A.rIDir + quad * 12);
#pragma unroll
		for(int c = 0; c < 3; c++) {
			po[c] = make_float4(ro[c][0], ro[c][1], ro[c][2], ro[c][3]);
			pd[c] = make_float4(rd[c][0], rd[c][1], rd[c][2], rd[c][3]);
			pi[c] = make_float4(ri[c][0], ri[c][1], ri[c][2], ri[c][3]);
		}
		A.rMask[quad] = (unsigned char)sel;
		*(float4 *)(A.rDist + quad * 4) = make_float4(rdist[0], rdist[1], rdist[2], rdist[3]);
		*(int4 *)(A.rObj + quad * 4) = make_int4(0, 0, 0, 0);
		unsigned cnt = 0;     // stats.TracingRays(CountMaskBits(mask)) of the nested RayTrace (src/scene_trace.cpp:116-117)
#pragma unroll
		for(int l = 0; l < 4; l++) cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64((sel >> l) & 1u));
		const Counters none = {0, 0, 0, 0, 0};
		flushStats(A.stats, none, cnt, lane);
		return;
	}
	if(DST == DST_MIRROR) {
		// Scene::TraceReflection (src/scene_trace.cpp:603-618): Reflect (src/rtbase_math.h:54-58), origin = position + 0.001 dir,
		// SafeInv; selector = hit lanes.  Masked lanes: zeros (see include/snail_hip.h), distance -inf.
		float rd[3][4], ro[3][4], ri[3][4], rdist[4];
		unsigned sel = 0;
#pragma unroll
		for(int l = 0; l < 4; l++) {
			const float dt = S.nrm[0][l] * d[0][l] + S.nrm[1][l] * d[1][l] + S.nrm[2][l] * d[2][l];
			const float dt2 = dt + dt;
#pragma unroll
			for(int c = 0; c < 3; c++) {
				const float r = d[c][l] - S.nrm[c][l] * dt2;
				rd[c][l] = S.hit[l] ? r : 0.0f;
				ro[c][l] = S.hit[l] ? S.pos[c][l] + r * 0.001f : 0.0f;
				ri[c][l] = recipExact(rd[c][l] + 0.00000001f);
			}
			rdist[l] = S.hit[l] ? inf : -inf; // src/scene_trace.cpp:112-115
			sel |= S.hit[l] ? (1u << l) : 0u;
		}
		float4 *po = (float4 *)(A.rOrg + quad * 12), *pd = (float4 *)(A.rDir + quad * 12), *pi = (float4 *)(A.rIDir + quad * 12);
#pragma unroll
		for(int c = 0; c < 3; c++) {
			po[c] = make_float4(ro[c][0], ro[c][1], ro[c][2], ro[c][3]);
			pd[c] = make_float4(rd[c][0], rd[c][1], rd[c][2], rd[c][3]);
			pi[c] = make_float4(ri[c][0], ri[c][1], ri[c][2], ri[c][3]);
		}
		A.rMask[quad] = (unsigned char)sel;
		*(float4 *)(A.rDist + quad * 4) = make_float4(rdist[0], rdist[1], rdist[2], rdist[3]);
		*(int4 *)(A.rObj + quad * 4) = make_int4(0, 0, 0, 0);
		// stats.TracingRays(CountMaskBits(mask)) of the nested RayTrace (src/scene_trace.cpp:116-117)
		unsigned cnt = 0;
#pragma unroll
		for(int l = 0; l < 4; l++) cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(S.hit[l]));
		const Counters none = {0, 0, 0, 0, 0};
		flushStats(A.stats, none, cnt, lane);
		return;
	}

	// ---- lights: src/scene_trace.cpp:484-512; per light the tail of Scene::TraceLight (:567-601) ----
	float lDiff[3][4], lSpec[3][4];
#pragma unroll
	for(int c = 0; c < 3; c++)
#pragma unroll
		for(int l = 0; l < 4; l++) { lDiff[c][l] = A.ambient[c]; lSpec[c][l] = 0.0f; }
	if(A.nLights) {
		float tMin[3], tMax[3];
		hitBounds(S, tMin, tMax);
		const size_t packets = (size_t)A.nPackets;
		for(int n = 0; n < A.nLights; n++) {
			const float lp[3] = {A.lights[n][0], A.lights[n][1], A.lights[n][2]};
			const float lc[3] = {A.lights[n][3], A.lights[n][4], A.lights[n][5]};
			const float radius = A.lights[n][6], iRadius = 1.0f / radius, radSq = radius * radius;
			if(lightCulled(tMin, tMax, lp, radSq)) continue;
			const float4 sv = *(const float4 *)(A.sDist + ((size_t)n * packets + P.pidx) * 256 + (size_t)lane * 4);
			const float sdist[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
			for(int l = 0; l < 4; l++) {
				float sd[3], distance, dotv, unused;
				shadowLane(S, l, lp, sd, distance, dotv, unused);
				if(sdist[l] > 0.0f) {
					float atten = distance * iRadius;
					atten = Max<M_EXACT>(0.0f, ((1.0f - atten) * 0.2f + 1.0f / (16.0f * atten * atten)) - 0.0625f);
					const float diffMul = dotv * atten;
					float specMul = dotv;
					specMul *= specMul; specMul *= specMul; specMul *= specMul; specMul *= specMul;
					specMul *= atten;
#pragma unroll
					for(int c = 0; c < 3; c++) { lDiff[c][l] += lc[c] * diffMul; lSpec[c][l] += lc[c] * specMul; }
				}
			}
		}
	}

	// ---- outColor = diffuse * lDiffuse + specular * lSpecular (src/scene_trace.cpp:504-512) ----
	float col[3][4];
#pragma unroll
	for(int l = 0; l < 4; l++) {
		float refl[3] = {0.0f, 0.0f, 0.0f};
		if(DST == DST_FRAME && A.blend) {
			const float *rc = A.rCol + (quad * 4 + l) * 3;
			refl[0] = rc[0]; refl[1] = rc[1]; refl[2] = rc[2];
		}
#pragma unroll
		for(int c = 0; c < 3; c++) {
			const float spec = A.color[c] * S.sdn[l];
			float diff = spec;
			if(DST == DST_FRAME && A.blend && S.hit[l]) diff = diff + (refl[c] - diff) * 0.3f; // src/scene_trace.cpp:462-465
			col[c][l] = A.nLights ? diff * lDiff[c][l] + spec * lSpec[c][l] : diff;
		}
	}
	if(DST == DST_COLOR) {
#pragma unroll
		for(int l = 0; l < 4; l++) {
			float *rc = A.rCol + (quad * 4 + l) * 3;
			rc[0] = col[0][l]; rc[1] = col[1][l]; rc[2] = col[2][l];
		}
		return;
	}
	if(DST == DST_FRAME && A.colPackets) { // 4x antialiasing: the double-resolution packets' colours go on to dev::k_aa_reduce as floats
#pragma unroll
		for(int l = 0; l < 4; l++) {
			float *rc = A.colPackets + (quad * 4 + l) * 3;
			rc[0] = col[0][l]; rc[1] = col[1][l]; rc[2] = col[2][l];
		}
		return;
	}
	unsigned bytes[12];
#pragma unroll
	for(int l = 0; l < 4; l++) { bytes[l * 3 + 0] = (unsigned)convChannelW(col[2][l]); bytes[l * 3 + 1] = (unsigned)convChannelW(col[1][l]); bytes[l * 3 + 2] = (unsigned)convChannelW(col[0][l]); }
	if(A.bgrPackets) { // tile sharding: packet-major bytes, what a render node returns (scattered by snail_packets_bgr_to_frame_dev)
		unsigned *o = (unsigned *)(A.bgrPackets + (quad * 4) * 3);
#pragma unroll
		for(int k = 0; k < 3; k++) o[k] = bytes[4 * k] | (bytes[4 * k + 1] << 8) | (bytes[4 * k + 2] << 16) | (bytes[4 * k + 3] << 24);
		return;
	}
	const int yy = P.py + (lane >> 2), xx = P.px + (lane & 3) * 4;
	if(yy < A.resy) {
		unsigned char *dd = A.frame + (size_t)yy * A.pitch + (size_t)xx * 3;
		if(xx + 3 < A.resx && (A.pitch & 3) == 0 && ((unsigned long long)A.frame & 3) == 0) { // 4 pixels = 12 bytes = three aligned dwords (xx is a multiple of 4)
			unsigned w[3];
#pragma unroll
			for(int k = 0; k < 3; k++) w[k] = bytes[4 * k] | (bytes[4 * k + 1] << 8) | (bytes[4 * k + 2] << 16) | (bytes[4 * k + 3] << 24);
			unsigned *dw = (unsigned *)dd;
			dw[0] = w[0]; dw[1] = w[1]; dw[2] = w[2];
		} else {
#pragma unroll
			for(int l = 0; l < 4; l++)
				if(xx + l < A.resx) { dd[l * 3 + 0] = (unsigned char)bytes[l * 3 + 0]; dd[l * 3 + 1] = (unsigned char)bytes[l * 3 + 1]; dd[l * 3 + 2] = (unsigned char)bytes[l * 3 + 2]; }
		}
	}
}

// ---- 4x antialiasing of the tile renderer (gVals[9]; src/render.cpp:60-62, :71-110) -----------------------------------------------
// Every 16x16 packet of the image is the 2x2 reduction of FOUR packets of the double-resolution frame (sub-packet k at (2x + 16 (k & 1),
// 2y + 16 (k >> 1)), stored at 4 p + k): out = ((top-left + bottom-left) * 0.25) + ((top-right + bottom-right) * 0.25), the reference's
// operation order (row 2r plus row 2r + 1 per SSE lane, times 0.25, then lane 0 + lane 1 and lane 2 + lane 3), then ConvColor.
// One wave per image packet, lane = output quad.  DEPTH: the input holds hit distances [4n][256] and the colour is gVals[1]'s depth
// shading Inv(t) * (20, 250, 2) (src/scene_trace.cpp:128-137); else float colours [4n][256][3] written by k_final.  Output: packet-major B,G,R.
template <bool DEPTH>
__global__ __launch_bounds__(64) void k_aa_reduce(const float *__restrict__ in, int nPackets, unsigned char *__restrict__ bgrPackets) {
	const int p = (int)blockIdx.x, lane = (int)threadIdx.x;
	if(p >= nPackets) return;
	const int row = lane >> 2, qc = lane & 3;                     // output quad: row, quad column
	const int k = (row >= 8 ? 2 : 0) + (qc >= 2 ? 1 : 0);         // the quarter of the packet = the sub-packet it comes from
	const int r = row & 7, h = qc & 1;
	unsigned bytes[12];
#pragma unroll
	for(int j = 0; j < 4; j++) {
		const int i = j >> 1, s2 = (j & 1) * 2;
		const int qa = 8 * r + 2 * h + i, qb = qa + 4;              // the input quads of rows 2r and 2r + 1
		const size_t ia = ((size_t)(4 * p + k) * 64 + qa) * 4 + s2, ib = ((size_t)(4 * p + k) * 64 + qb) * 4 + s2;
		float c[3];
#pragma unroll
		for(int ch = 0; ch < 3; ch++) {
			float a0, a1, b0, b1;
			if(DEPTH) {
				const float scale = ch == 0 ? 20.0f : ch == 1 ? 250.0f : 2.0f;
				a0 = (1.0f / in[ia]) * scale; a1 = (1.0f / in[ia + 1]) * scale; b0 = (1.0f / in[ib]) * scale; b1 = (1.0f / in[ib + 1]) * scale;
			} else { a0 = in[ia * 3 + ch]; a1 = in[(ia + 1) * 3 + ch]; b0 = in[ib * 3 + ch]; b1 = in[(ib + 1) * 3 + ch]; }
			c[ch] = (a0 + b0) * 0.25f + (a1 + b1) * 0.25f;
		}
		bytes[j * 3 + 0] = (unsigned)convChannelW(c[2]); bytes[j * 3 + 1] = (unsigned)convChannelW(c[1]); bytes[j * 3 + 2] = (unsigned)convChannelW(c[0]);
	}
	unsigned *o = (unsigned *)(bgrPackets + ((size_t)p * 256 + (size_t)lane * 4) * 3);
#pragma unroll
	for(int w = 0; w < 3; w++) o[w] = bytes[4 * w] | (bytes[4 * w + 1] << 8) | (bytes[4 * w + 2] << 16) | (bytes[4 * w + 3] << 24);
}

// ---- generic packets: TraversePrimary<SHARED,MASK>(Context&) --------------------------------------
struct RaysArgs {
	const uint4 *nodes, *tris;
	const uint4 *pf; // the record-prefetching loop's copy of the tree (SnailScene::dPF; used when `pack` is set)
	int nPackets, size, fastOK;
	int pack; // at most 2^20 node slots: one-word stack entries + record prefetch in the per-ray-origin walk
	const float *origin, *dir, *idir;
	const unsigned char *mask;
	float *distance;
	int *object;
	float *bary;
	u64 *stats;
	int *defer; // k_rays: [0] = count, [1] = finished blocks of the M_EXACT pass, [16..] = deferred packet indices
};


// One wave per block (packet costs are heavy-tailed, see k_primary), blocks dealt to the XCDs 16 consecutive packets at a time.
// EXACTPASS as in the primary kernel: the main launch walks in M_COH / M_FAST and appends a packet that needs M_EXACT to A.defer
// untouched (its distances / objects in memory are still the caller's); the second, small launch walks those in M_EXACT.
template <bool SHARED, bool MASK, bool DEEP, bool BARY, bool EXACTPASS>
__device__ __forceinline__ void raysPacket(const RaysArgs &A, const int p, float *lds) {
	const int lane = threadIdx.x & 63;
	if(p >= A.nPackets) return;
	const int size = A.size;
	const size_t q0 = (size_t)p * size;
	const bool live = lane < size;
	const size_t q = q0 + (live ? lane : 0);

	Quad Q;
	float org[3][4];
	loadQuad3(A.dir, q, Q.d);
	loadQuad3(A.idir, q, Q.id);
	if(SHARED) {
		scalar_ptr op = (scalar_ptr)(unsigned long long)(A.origin + (size_t)p * 12);
		u32x4 ox = op[0], oy = op[1], oz = op[2];
#pragma unroll
		for(int l = 0; l < 4; l++) { org[0][l] = asf(ox.x); org[1][l] = asf(oy.x); org[2][l] = asf(oz.x); } // ExtractN(Origin(0), 0)
	} else loadQuad3(A.origin, q, org);
	unsigned mask4 = 15u;
	if(MASK) mask4 = A.mask[q] & 15u;
	float4 dv = *(const float4 *)(A.distance + q * 4);
	int4 ov = *(const int4 *)(A.object + q * 4);
	float4 b0 = make_float4(0, 0, 0, 0), b1 = b0;
	if(BARY) { b0 = *(const float4 *)(A.bary + q * 8); b1 = *(const float4 *)(A.bary + q * 8 + 4); }
	Q.dist[0] = dv.x; Q.dist[1] = dv.y; Q.dist[2] = dv.z; Q.dist[3] = dv.w;
	int tid[4] = {ov.x, ov.y, ov.z, ov.w};
	float bu[4] = {b0.x, b0.y, b0.z, b0.w}, bv[4] = {b1.x, b1.y, b1.z, b1.w};

	Counters st = {0, 0, 0, 0, 0};
	if(EXACTPASS) walk<SHARED, MASK, false, M_EXACT, BARY, DEEP, false>(A.nodes, A.tris, size, lane, org, Q, mask4, tid, bu, bv, lds, st);
	else {
		bool fin = finite4(Q.id) && finite4(Q.d) && finite4(org);
#pragma unroll
		for(int l = 0; l < 4; l++) fin = fin && !(Q.dist[l] != Q.dist[l]);
		int oct;
		const int mode = classify(A.fastOK != 0, fin, live, Q.id, Q.dist, oct);
		if(mode == M_EXACT) {
			if(lane == 0) A.defer[16 + atomicAdd(&A.defer[0], 1)] = p;
			return;
		}
		if(!SHARED && !DEEP) { // per-ray origins: the hand-written node loop
			if(A.pack) {
				if(mode == M_COH) walkPerRayAsm<MASK, true, BARY, true>(SNAIL_PERRAY_COH_PF ? SNAIL_PACK_NODES(A) : A.nodes, A.tris, size, lane, org, Q, mask4, tid, bu, bv, st, oct);
				else walkPerRayAsm<MASK, false, BARY, true>(SNAIL_PACK_NODES(A), A.tris, size, lane, org, Q, mask4, tid, bu, bv, st, 0);
			} else if(mode == M_COH) walkPerRayAsm<MASK, true, BARY, false>(A.nodes, A.tris, size, lane, org, Q, mask4, tid, bu, bv, st, oct);
			else walkPerRayAsm<MASK, false, BARY, false>(A.nodes, A.tris, size, lane, org, Q, mask4, tid, bu, bv, st, 0);
		} else if(SHARED && !DEEP && A.pack && SNAIL_SHADOW_ENTRY_PF) { // shared origin, any distances on entry: the prefetching loop over its plain copy of the tree
			if(mode == M_COH) walkSharedAsm<false, true, true, MASK, BARY, false, false>(SNAIL_PACK_NODES(A), A.tris, size, lane, org, Q, mask4, tid, bu, bv, lds, st, oct);
			else walkSharedAsm<false, false, true, MASK, BARY, false, false>(SNAIL_PACK_NODES(A), A.tris, size, lane, org, Q, mask4, tid, bu, bv, lds, st, 0);
		} else if(SHARED && !DEEP) {
			if(mode == M_COH) walkSharedAsm<false, true, false, MASK, BARY, false>(A.nodes, A.tris, size, lane, org, Q, mask4, tid, bu, bv, lds, st, oct);
			else walkSharedAsm<false, false, false, MASK, BARY, false>(A.nodes, A.tris, size, lane, org, Q, mask4, tid, bu, bv, lds, st, 0);
		} else if(mode == M_COH) walk<SHARED, MASK, false, M_COH, BARY, DEEP, false>(A.nodes, A.tris, size, lane, org, Q, mask4, tid, bu, bv, lds, st, oct);
		else walk<SHARED, MASK, false, M_FAST, BARY, DEEP, false>(A.nodes, A.tris, size, lane, org, Q, mask4, tid, bu, bv, lds, st);
	}
	flushStats(A.stats, st, 0u, lane);

	if(live) {
		*(float4 *)(A.distance + q * 4) = make_float4(Q.dist[0], Q.dist[1], Q.dist[2], Q.dist[3]);
		*(int4 *)(A.object + q * 4) = make_int4(tid[0], tid[1], tid[2], tid[3]);
		if(BARY) {
			*(float4 *)(A.bary + q * 8) = make_float4(bu[0], bu[1], bu[2], bu[3]);
			*(float4 *)(A.bary + q * 8 + 4) = make_float4(bv[0], bv[1], bv[2], bv[3]);
		}
	}
}
#ifndef SNAIL_RAYS_WAVES
#define SNAIL_RAYS_WAVES 0 // occupancy target of the generic-packet kernels; 0 = the compiler's own allocation (no spills: these kernels keep lane-indexed state
						   // in VGPRs; a forced six-wave budget -- 80 VGPRs, up to 88 spilled into scratch -- measured no gain)
#endif
#if SNAIL_RAYS_WAVES > 0
#define SNAIL_RAYS_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SNAIL_RAYS_WAVES)))
#else
#define SNAIL_RAYS_OCCUPANCY
#endif
template <bool SHARED, bool MASK, bool DEEP, bool BARY>
__global__ __launch_bounds__(64) SNAIL_RAYS_OCCUPANCY void k_rays(RaysArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	raysPacket<SHARED, MASK, DEEP, BARY, false>(A, interleave16((int)blockIdx.x), lds);
}
template <bool SHARED, bool MASK, bool DEEP, bool BARY>
__global__ __launch_bounds__(64) void k_rays_exact(RaysArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	const int n = __builtin_amdgcn_readfirstlane(A.defer[0]);
	if(n == 0) return;
	for(int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) raysPacket<SHARED, MASK, DEEP, BARY, true>(A, __builtin_amdgcn_readfirstlane(A.defer[16 + i]), lds);
	__threadfence();
	if((threadIdx.x & 63) == 0 && atomicAdd(&A.defer[1], 1) == (int)gridDim.x - 1) { A.defer[0] = 0; A.defer[1] = 0; }
}

// ---- shadow packets: TraverseShadow(ShadowContext&) -----------------------------------------------
// same launch structure as k_rays: one wave per block, M_EXACT packets deferred to the second launch
template <bool DEEP, bool EXACTPASS>
__device__ __forceinline__ void shadowPacket(const RaysArgs &A, const int p, float *lds) {
	const int lane = threadIdx.x & 63;
	if(p >= A.nPackets) return;
	const int size = A.size;
	const bool live = lane < size;
	const size_t q = (size_t)p * size + (live ? lane : 0);

	Quad Q;
	float org[3][4];
	loadQuad3(A.dir, q, Q.d);
	loadQuad3(A.idir, q, Q.id);
	{
		const float *op = A.origin + (size_t)p * 3;
		const float o0 = firstlanef(op[0]), o1 = firstlanef(op[1]), o2 = firstlanef(op[2]);
#pragma unroll
		for(int l = 0; l < 4; l++) { org[0][l] = o0; org[1][l] = o1; org[2][l] = o2; }
	}
	float4 dv = *(const float4 *)(A.distance + q * 4);
	Q.dist[0] = dv.x; Q.dist[1] = dv.y; Q.dist[2] = dv.z; Q.dist[3] = dv.w;
	int tid[4] = {0, 0, 0, 0};
	float bu[4] = {0, 0, 0, 0}, bv[4] = {0, 0, 0, 0};

	Counters st = {0, 0, 0, 0, 0};
	if(EXACTPASS) walk<true, false, true, M_EXACT, false, DEEP, false>(A.nodes, A.tris, size, lane, org, Q, 15u, tid, bu, bv, lds, st);
	else {
		bool fin = finite4(Q.id) && finite4(Q.d) && finite4(org);
#pragma unroll
		for(int l = 0; l < 4; l++) fin = fin && !(Q.dist[l] != Q.dist[l]);
		int oct;
		const int mode = classify(A.fastOK != 0, fin, live, Q.id, Q.dist, oct);
		if(mode == M_EXACT) {
			if(lane == 0) A.defer[16 + atomicAdd(&A.defer[0], 1)] = p;
			return;
		}
		if(DEEP) {
			if(mode == M_COH) walk<true, false, true, M_COH, false, DEEP, false>(A.nodes, A.tris, size, lane, org, Q, 15u, tid, bu, bv, lds, st, oct);
			else walk<true, false, true, M_FAST, false, DEEP, false>(A.nodes, A.tris, size, lane, org, Q, 15u, tid, bu, bv, lds, st);
		} else if(A.pack && SNAIL_SHADOW_ENTRY_PF) { // the prefetching loop over its plain copy of the tree (origins differ from packet to packet: no relative records)
			if(mode == M_COH) walkSharedAsm<true, true, true, false, false, false, false>(SNAIL_PACK_NODES(A), A.tris, size, lane, org, Q, 15u, tid, bu, bv, lds, st, oct);
			else walkSharedAsm<true, false, true, false, false, false, false>(SNAIL_PACK_NODES(A), A.tris, size, lane, org, Q, 15u, tid, bu, bv, lds, st, 0);
		} else if(mode == M_COH) walkSharedAsm<true, true, false, false, false, false>(A.nodes, A.tris, size, lane, org, Q, 15u, tid, bu, bv, lds, st, oct);
		else walkSharedAsm<true, false, false, false, false, false>(A.nodes, A.tris, size, lane, org, Q, 15u, tid, bu, bv, lds, st, 0);
	}
	flushStats(A.stats, st, 0u, lane);
	if(live) *(float4 *)(A.distance + q * 4) = make_float4(Q.dist[0], Q.dist[1], Q.dist[2], Q.dist[3]);
}
template <bool DEEP>
__global__ __launch_bounds__(64) SNAIL_RAYS_OCCUPANCY void k_shadow(RaysArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	shadowPacket<DEEP, false>(A, interleave16((int)blockIdx.x), lds);
}
template <bool DEEP>
__global__ __launch_bounds__(64) void k_shadow_exact(RaysArgs A) {
	__shared__ float lds[LDS_FLOATS_PER_WAVE];
	const int n = __builtin_amdgcn_readfirstlane(A.defer[0]);
	if(n == 0) return;
	for(int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) shadowPacket<DEEP, true>(A, __builtin_amdgcn_readfirstlane(A.defer[16 + i]), lds);
	__threadfence();
	if((threadIdx.x & 63) == 0 && atomicAdd(&A.defer[1], 1) == (int)gridDim.x - 1) { A.defer[0] = 0; A.defer[1] = 0; }
}

// ---- packet-major -> frame scatter ------------------------------------------------------------------
struct ScatterArgs {
	const int2 *packetXY;
	int nPackets, resx, resy;
	const float *pt, *pu, *pv;
	const int *pid;
	float *t, *u, *v;
	int *id;
};
__global__ __launch_bounds__(256) void k_packets_to_frame(ScatterArgs A) {
	const int lane = threadIdx.x & 63;
	const int p = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
	if(p >= A.nPackets) return;
	const int2 xy = A.packetXY[p];
	const int yy = xy.y + (lane >> 2), xx = xy.x + (lane & 3) * 4;
	if(yy >= A.resy) return;
	const size_t src = (size_t)p * 256 + (size_t)lane * 4, dst = (size_t)yy * A.resx + xx;
#pragma unroll
	for(int l = 0; l < 4; l++)
		if(xx + l < A.resx) {
			if(A.t && A.pt) A.t[dst + l] = A.pt[src + l];
			if(A.u && A.pu) A.u[dst + l] = A.pu[src + l];
			if(A.v && A.pv) A.v[dst + l] = A.pv[src + l];
			if(A.id && A.pid) A.id[dst + l] = A.pid[src + l];
		}
}

// ---- gVals[1] depth shading + ConvColor (src/scene_trace.cpp:128-137, src/render.cpp:11-17) ---------
__device__ __forceinline__ int convChannel(float c) { // Trunc(Clamp(c * 255, 0, 255)); Clamp = Min(Max(v, lo), hi), veclib/vecbase.h:75-77
	float v = c * 255.0f;
	v = v > 0.0f ? v : 0.0f;
	v = v < 255.0f ? v : 255.0f;
	return (int)v;
}
__global__ __launch_bounds__(256) void k_shade_depth(const float *t, int nRays, unsigned char *bgr) {
	const int i = (int)(blockIdx.x * 256 + threadIdx.x) * 4; // 4 rays per thread: one 16-B load, three 4-B stores (nRays is a multiple of 256)
	if(i >= nRays) return;
	const float4 tv = *(const float4 *)(t + i);
	const float tt[4] = {tv.x, tv.y, tv.z, tv.w};
	unsigned bytes[12];
#pragma unroll
	for(int l = 0; l < 4; l++) {
		const float dist = 1.0f / tt[l];         // Condition(t > inf, 0, Inv(t)): the condition is never true
		bytes[l * 3 + 0] = (unsigned)convChannel(dist * 2.0f); bytes[l * 3 + 1] = (unsigned)convChannel(dist * 250.0f); bytes[l * 3 + 2] = (unsigned)convChannel(dist * 20.0f);
	}
	if(((unsigned long long)bgr & 3) == 0) {
		unsigned *o = (unsigned *)(bgr + (size_t)i * 3);
#pragma unroll
		for(int k = 0; k < 3; k++) o[k] = bytes[4 * k] | (bytes[4 * k + 1] << 8) | (bytes[4 * k + 2] << 16) | (bytes[4 * k + 3] << 24);
	} else {
#pragma unroll
		for(int k = 0; k < 12; k++) bgr[(size_t)i * 3 + k] = (unsigned char)bytes[k];
	}
}
// nPerChunk > 0: the source is cut into chunks of nPerChunk packets that lie chunkStride bytes apart (rank r's shard of a gathered buffer that
// holds several frames per rank: packet p = entry p % nPerChunk of chunk p / nPerChunk); 0 = one contiguous array
__global__ __launch_bounds__(256) void k_bgr_to_frame(const int2 *packetXY, int nPackets, int resx, int resy, const unsigned char *src,
														unsigned char *frame, int pitch, int nPerChunk, long long chunkStride) {
	const int lane = threadIdx.x & 63;
	const int p = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
	if(p >= nPackets) return;
	const int2 xy = packetXY[p];
	const int yy = xy.y + (lane >> 2), xx = xy.x + (lane & 3) * 4;
	if(yy >= resy) return;
	const unsigned char *s = nPerChunk > 0 ? src + (size_t)(p / nPerChunk) * (size_t)chunkStride + ((size_t)(p % nPerChunk) * 256 + (size_t)lane * 4) * 3
										   : src + ((size_t)p * 256 + (size_t)lane * 4) * 3;
	unsigned char *d = frame + (size_t)yy * pitch + (size_t)xx * 3;
	if(xx + 3 < resx && (pitch & 3) == 0 && (((unsigned long long)frame | (unsigned long long)src) & 3) == 0) { // 4 pixels = three aligned dwords
		const unsigned *sw = (const unsigned *)s;
		unsigned *dw = (unsigned *)d;
		const unsigned a = sw[0], b = sw[1], c = sw[2];
		dw[0] = a; dw[1] = b; dw[2] = c;
		return;
	}
	for(int l = 0; l < 4; l++)
		if(xx + l < resx) { d[l * 3 + 0] = s[l * 3 + 0]; d[l * 3 + 1] = s[l * 3 + 1]; d[l * 3 + 2] = s[l * 3 + 2]; }
}

// ---- the render node's tile wire format (src/render.cpp:140-163) and its inverse (src/compression.cpp:112-141) ----------
// A tile (x, y, w, h) travels as three w*h byte planes: R, G-R, B-R (mod 256), R = the channel ConvColor shifts by 16, i.e. byte 2
// of a stored pixel.  Source: packet-major BGR bytes (in-packet index = 16*row + column, the reference's quad order); the
// packets of a tile are consecutive, row bands outer, columns inner (RenderTask::Work loop order).
__global__ __launch_bounds__(256) void k_bgr_to_planar(const int4 *tiles, const int *firstPacket, const long long *outOff, int nTiles,
														 const unsigned char *src, unsigned char *out) {
	const int tile = (int)blockIdx.y;
	if(tile >= nTiles) return;
	const int4 T = tiles[tile];
	const int n = T.z * T.w;
	const int ppr = (T.z + 15) >> 4;
	unsigned char *o = out + outOff[tile];
	for(int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x)) {
		const int ty = i / T.z, tx = i - ty * T.z;
		const size_t pk = (size_t)firstPacket[tile] + (size_t)(ty >> 4) * ppr + (size_t)(tx >> 4);
		const unsigned char *s = src + (pk * 256 + (size_t)((ty & 15) * 16 + (tx & 15))) * 3;
		const unsigned char b = s[0], g = s[1], r = s[2];
		o[i] = r; o[(size_t)n + i] = (unsigned char)(g - r); o[(size_t)2 * n + i] = (unsigned char)(b - r);
	}
}
__global__ __launch_bounds__(256) void k_planar_to_frame(const int4 *tiles, const long long *inOff, int nTiles, const unsigned char *in, unsigned char *frame,
														   int pitch, int resx, int resy) {
	const int tile = (int)blockIdx.y;
	if(tile >= nTiles) return;
	const int4 T = tiles[tile];
	const int n = T.z * T.w;
	const unsigned char *p = in + inOff[tile];
	for(int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x)) {
		const int ty = i / T.z, tx = i - ty * T.z;
		const int xx = T.x + tx, yy = T.y + ty;
		if(xx >= resx || yy >= resy) continue;
		const unsigned char red = p[i];
		unsigned char *d = frame + (size_t)yy * pitch + (size_t)xx * 3;
		d[2] = red; d[1] = (unsigned char)(p[(size_t)n + i] + red); d[0] = (unsigned char)(p[(size_t)2 * n + i] + red);
	}
}

#ifdef SNAIL_DEBUG_API // the workbench build (libsnailhip_debug.so, include/snail_hip_debug.h): diagnostics and experiments only
// A stream-ordered pause of `ticks` periods of the 100 MHz constant clock (s_memrealtime): one wave that sleeps in 64-cycle naps.
// (snail_debug_delay_dev: de-phasing experiments of pipelined frame streams).  Ends after `ticks` whatever happens.
__global__ __launch_bounds__(64) void k_delay(unsigned ticks) {
	const u64 t0 = __builtin_amdgcn_s_memrealtime();
	while((unsigned)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
}

// diagnostic: the shader clock as the guide's DVFS check reads it (MI355X_MICROARCH.md, "DVFS give-back" item 6): delta s_memtime (shader
// cycles) over delta s_memrealtime (100 MHz) across `ticks` periods of the constant clock, one sleeping wave; out[0] = cycles, out[1] = ticks
__global__ __launch_bounds__(64) void k_clock(unsigned ticks, unsigned long long *out) {
	const u64 r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
	while((unsigned)(__builtin_amdgcn_s_memrealtime() - r0) < ticks) __builtin_amdgcn_s_sleep(1);
	const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	if(threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

// exhaustive check of recipExact() against the correctly rounded division: thread = one float bit pattern (a wave holds 64 consecutive
// patterns, i.e. one exponent, so every in-range input goes through the short sequence); out[0] = results that differ bitwise (NaN = one
// class), out[1] = inputs inside the short sequence's range
__global__ __launch_bounds__(256) void k_recip_check(unsigned base, unsigned long long *out) {
	const unsigned bits = base + blockIdx.x * 256u + threadIdx.x;
	const float x = __uint_as_float(bits);
	const float ref = 1.0f / x, got = recipExact(x);
	const bool same = (ref != ref) ? (got != got) : __float_as_uint(ref) == __float_as_uint(got);
	const bool inside = ((bits & 0x7f800000u) - 0x00800000u) < 0x7e000000u;
	const u64 bad = __builtin_amdgcn_ballot_w64(!same), in = __builtin_amdgcn_ballot_w64(inside);
	if((threadIdx.x & 63) == 0) {
		if(bad) atomicAdd(&out[0], (unsigned long long)__builtin_popcountll(bad));
		if(in) atomicAdd(&out[1], (unsigned long long)__builtin_popcountll(in));
	}
}

// diagnostic: what the workgroup dispatcher alone sustains (tools/dispatch_rate.py)
__global__ void k_nop(int *sink) { if(sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 1; }
#endif // SNAIL_DEBUG_API

// ---- the record-prefetching loop's copy of the node records (SNAIL_PF_VISIT): slot i + 1 <- node i, words 6 / 7 re-encoded ----
// (one thread per node; used for trees that were built on the device -- snail_scene_create encodes on the host, same function)
__host__ __device__ inline void pfEncode(const unsigned (&in)[8], unsigned (&out)[8], unsigned trisOff) {
	for(int k = 0; k < 6; k++) out[k] = in[k];
	if(in[6] & 0x80000000u) { out[6] = 0x80000000u | (trisOff + ((in[6] & 0x7fffffffu) << 6)); out[7] = in[7]; }   // leaf: first triangle's byte offset, count
	else { out[6] = (in[6] + 1u + (in[7] >> 16)) << 5; out[7] = 1u << (in[7] & 3u); }   // inner: the child taken first when sign[axis] is clear; 1 << axis
}
__global__ __launch_bounds__(256) void k_pf_encode(const uint4 *__restrict__ nodes, int nNodes, unsigned trisOff, uint4 *__restrict__ pf) {
	const int i = (int)(blockIdx.x * 256 + threadIdx.x);
	if(i >= nNodes) return;
	const uint4 a = nodes[(size_t)i * 2], b = nodes[(size_t)i * 2 + 1];
	const unsigned in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	unsigned out[8];
	pfEncode(in, out, trisOff);
	pf[(size_t)(i + 1) * 2] = make_uint4(out[0], out[1], out[2], out[3]);
	pf[(size_t)(i + 1) * 2 + 1] = make_uint4(out[4], out[5], out[6], out[7]);
}

// slot i of the prefetching loop's copy -> slot i of a camera-relative array: bmin - o, bmax - o (the subtraction SNAIL_PRE_SHARED makes at
// every visit, made once per node and origin), link words unchanged
__global__ __launch_bounds__(256) void k_rel_nodes(const uint4 *__restrict__ pf, int nSlots, float ox, float oy, float oz, uint4 *__restrict__ rel) {
	const int i = (int)(blockIdx.x * 256 + threadIdx.x);
	if(i >= nSlots) return;
	const uint4 a = pf[(size_t)i * 2], b = pf[(size_t)i * 2 + 1];
	const float o[3] = {ox, oy, oz};
	const unsigned in[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
	unsigned out[6];
	for(int k = 0; k < 6; k++) out[k] = __float_as_uint(__uint_as_float(in[k]) - o[k % 3]);
	rel[(size_t)i * 2] = make_uint4(out[0], out[1], out[2], out[3]);
	rel[(size_t)i * 2 + 1] = make_uint4(out[4], out[5], b.z, b.w);
}

// ---- single-ray accounting walk: V_n, V_t per ray ----------------------------
struct AccountArgs {
	const uint4 *nodes, *tris;
	GenConst g;
	int x0, y0, pw, ph;
	u64 *out;
};
__global__ __launch_bounds__(256) void k_account(AccountArgs A) {
	const int lane = threadIdx.x & 63;
	const int p = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
	if(p >= A.pw * A.ph) return;
	const int px = A.x0 + (p % A.pw) * 16, py = A.y0 + (p / A.pw) * 16;
	const int ty = lane >> 2, k4 = lane & 3;
	const float inf = __builtin_inff();
	unsigned vn = 0, vt = 0, hits = 0;
	for(int l = 0; l < 4; l++) {
		const float xoff = (float)(px + (l >= 2 ? 2 : 0)), yoff = (float)(py - (l >= 2 ? 1 : 0));
		const float tposx = (float)(4 * k4) + xoff, tposy = (float)ty + yoff;
		float pp[3], d[3], id[3];
		for(int c = 0; c < 3; c++) pp[c] = A.g.tright[c] * tposx + (A.g.tup[c] * tposy + A.g.txyz[c][l]);
		const float rs = 1.0f / __builtin_sqrtf(pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2]);
		for(int c = 0; c < 3; c++) { d[c] = pp[c] * rs; id[c] = 1.0f / (d[c] + 0.00000001f); }
		const int sg[3] = {d[0] < 0.0f, d[1] < 0.0f, d[2] < 0.0f};
		int stack[SNAIL_MAX_DEPTH + 2], sp = 0;
		stack[sp++] = 0;
		float dist = inf;
		while(sp) {
			int cur = stack[--sp];
			for(;;) {
				const uint4 *np = A.nodes + (size_t)cur * 2;
				const uint4 a = np[0], b = np[1];
				const float bmin[3] = {asf(a.x), asf(a.y), asf(a.z)}, bmax[3] = {asf(a.w), asf(b.x), asf(b.y)};
				vn++;
				float lmin = 0.0f, lmax = 0.0f;
				for(int k = 0; k < 3; k++) {
					float l1 = id[k] * (bmin[k] - A.g.org[k]), l2 = id[k] * (bmax[k] - A.g.org[k]);
					float lo = Min<M_EXACT>(l1, l2), hi = Max<M_EXACT>(l1, l2);
					if(k == 0) { lmin = lo; lmax = hi; }
					else { lmin = Max<M_EXACT>(lmin, lo); lmax = Min<M_EXACT>(lmax, hi); }
				}
				if(lmax < 0.0f || lmin > Min<M_EXACT>(lmax, dist)) break;
				if(b.z & 0x80000000u) {
					const int count = (int)b.w, firstTri = (int)(b.z & 0x7fffffffu);
					for(int k = 0; k < count; k++) {
						const Tri t = loadTriVector(A.tris, firstTri + k);
						vt++;
						float tv[3] = {A.g.org[0] - t.a[0], A.g.org[1] - t.a[1], A.g.org[2] - t.a[2]};
						float t0v[3] = {(t.ba[1] * tv[2] - t.ba[2] * tv[1]) * t.it0, (t.ba[2] * tv[0] - t.ba[0] * tv[2]) * t.it0,
										(t.ba[0] * tv[1] - t.ba[1] * tv[0]) * t.it0};
						float t1v[3] = {(tv[1] * t.ca[2] - tv[2] * t.ca[1]) * t.it0, (tv[2] * t.ca[0] - tv[0] * t.ca[2]) * t.it0,
										(tv[0] * t.ca[1] - tv[1] * t.ca[0]) * t.it0};
						const float tmul = -(tv[0] * t.n[0] + tv[1] * t.n[1] + tv[2] * t.n[2]);
						const float det = d[0] * t.n[0] + d[1] * t.n[1] + d[2] * t.n[2];
						const float v = d[0] * t0v[0] + d[1] * t0v[1] + d[2] * t0v[2];
						const float u = d[0] * t1v[0] + d[1] * t1v[1] + d[2] * t1v[2];
						const float duv = det - u - v;
						const float uvmin = Min<M_EXACT>(u, Min<M_EXACT>(v, duv)), uvmax = Max<M_EXACT>(u, Max<M_EXACT>(v, duv));
						if(!(uvmax <= 0.0f || uvmin >= 0.0f)) continue;
						const float t2 = (1.0f / det) * tmul;
						if(t2 < dist && t2 > 0.0f) dist = t2;
					}
					break;
				}
				const int child = (int)b.z, axis = (int)(b.w & 0xffff), fn = (int)((b.w >> 16) & 0xffff) ^ sg[axis];
				stack[sp++] = child + (fn ^ 1);
				cur = child + fn;
			}
		}
		if(dist < inf) hits++;
	}
	// wave reduction then one atomic per wave
	u64 r0 = 4, r1 = vn, r2 = vt, r3 = hits;
	for(int o = 32; o > 0; o >>= 1) {
		r0 += __shfl_xor(r0, o); r1 += __shfl_xor(r1, o); r2 += __shfl_xor(r2, o); r3 += __shfl_xor(r3, o);
	}
	if(lane == 0) { atomicAdd(&A.out[0], r0); atomicAdd(&A.out[1], r1); atomicAdd(&A.out[2], r2); atomicAdd(&A.out[3], r3); }
}

} // namespace dev

// ---------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------------------------------
namespace { struct TileJob; void freeTileJobs(SnailScene *); } // render_host.inc: cached lists of snail_render_tiles / snail_render_image
struct SnailScene {
	int device = 0;
	TileJob *tileJob = nullptr, *frameJob = nullptr;
	int nNodes = 0, nTris = 0, depth = 0;
	uint4 *dNodes = nullptr, *dTris = nullptr;   // the caller's records; dTris points INTO dPF (one allocation, see below)
	// [slot 0: unused][slot i + 1: node i, re-encoded for the record-prefetching loop (dev::pfEncode)] ... [triangle records at trisOff]
	char *dPF = nullptr;
	int trisOff = 0;
	// camera-relative copies of the node slots (dev::k_rel_nodes), one per distinct origin, least-recently-used first out: a static or
	// turning camera (and a light) costs one pass over the nodes EVER; a moving one costs one per new position
	struct RelNodes {
		float org[3] = {0, 0, 0};
		uint4 *d = nullptr;
		bool valid = false;
		unsigned long long stamp = 0;
		hipEvent_t filled = nullptr;
		enum { kStreams = 8 };
		hipStream_t usedOn[kStreams] = {};
		hipEvent_t used[kStreams] = {};
		int nUsed = 0;
	};
	enum { kRelSlots = 16 };
	RelNodes rel[kRelSlots];
	unsigned long long relClock = 0;
	int pfOK = 0;     // the prefetching loop may walk this tree: nested, every child pair starts at an odd index, offsets fit (stackPack)
	int fastOK = 0; // every triangle record finite and of sane magnitude (see file header)
	int nestedOK = 1; // every child box lies inside its parent's (stackPack)
	int pfOKButNesting = 0;
	int lastBlocks = 0, lastThreads = 0;
	unsigned long long *dStats = nullptr; // 4 x u64 scratch for the host-pointer entry points
	enum { kDeferSlots = 8 };
	int *dDefer[kDeferSlots] = {};        // deferred-packet lists, one per launch in flight (round-robin)
	// a slot's buffers are reused by the 8th launch after it, possibly on another stream and possibly while the host runs far
	// ahead of the device: each slot carries an event recorded after its last kernel, and the next user's stream waits for it
	hipEvent_t deferDone[kDeferSlots] = {};
	bool deferUsed[kDeferSlots] = {};
	int deferCap = 0;
	unsigned launchCount = 0;
	// intermediate state of snail_render_whitted_dev, one set per launch in flight (round-robin, like dDefer)
	struct ShadeScratch {
		size_t packets = 0;
		bool refl = false;
		float *hitT = nullptr; int *hitId = nullptr;
		float *rOrg = nullptr, *rDir = nullptr, *rIDir = nullptr, *rDist = nullptr, *rCol = nullptr;
		int *rObj = nullptr;
		unsigned char *rMask = nullptr;
		int *defer = nullptr;
		float *sDist = nullptr;
		hipEvent_t done = nullptr;
		bool used = false;
	} shade[kDeferSlots];
	unsigned shadeCount = 0;
	// deferred-packet lists of snail_trace_rays*, same slot discipline
	struct RayDefer { int *p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; } rayDefer[kDeferSlots];
	unsigned rayCount = 0;
};

namespace {

struct DeviceGuard {
	int prev = -1;
	bool ok = true;
	explicit DeviceGuard(int dev) {
		if(hipGetDevice(&prev) != hipSuccess) prev = -1;
		if(prev != dev) ok = hipSetDevice(dev) == hipSuccess;
	}
	~DeviceGuard() {
		int cur = -1;
		if(prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
	}
};

// RayGenerator ctor (src/ray_generator.cpp:4-15); plain fp32, this TU is built with -ffp-contract=off
dev::GenConst makeGen(const float cam[13], int w, int h) {
	dev::GenConst g;
	const float *pos = cam, *right = cam + 3, *up = cam + 6, *front = cam + 9;
	const float pd = cam[12];
	float invW = 1.0f / float(w), invH = 1.0f / float(h);
	invW *= float(w) / float(h);
	const float ax[4] = {0.0f, 1.0f, 0.0f, 1.0f}, ay[4] = {0.0f, 0.0f, 1.0f, 1.0f};
	for(int c = 0; c < 3; c++) { g.tright[c] = right[c] * invW; g.tup[c] = up[c] * invH; g.org[c] = pos[c]; }
	for(int l = 0; l < 4; l++) {
		const float taddx = ax[l] - w * 0.5f, taddy = ay[l] - h * 0.5f;
		for(int c = 0; c < 3; c++) {
			const float fp = front[c] * pd;
			g.txyz[c][l] = g.tright[c] * taddx + g.tup[c] * taddy + fp;
		}
	}
	return g;
}

bool originSane(const float *o) {
	for(int c = 0; c < 3; c++)
		if(!(std::fabs(o[c]) <= 1.0e9f)) return false;
	return true;
}

// Environment switches exist in the workbench build only (-DSNAIL_DEBUG_API): the product library reads no environment.
#ifdef SNAIL_DEBUG_API
int debugEnvInt(const char *name) { const char *v = getenv(name); return v ? atoi(v) : 0; }
#else
constexpr int debugEnvInt(const char *) { return 0; }
#endif

// depth > 62 needs the second stack register pair (DEEP instantiations).  Workbench build: SNAIL_DEBUG_FORCE_DEEP=1 selects them for
// any scene (the deep-BVH stack study of BASELINE config 5, tools/quick_time.py; the test of the walks ordinary scenes never select)
bool useDeep(const SnailScene *s) {
	static const bool force = debugEnvInt("SNAIL_DEBUG_FORCE_DEEP") != 0;
	return force || s->depth > 62;
}

// One-word stack entries (and with them the record-prefetching node loop over its own copy of the tree, SnailScene::dPF) need record
// slots below 2^20, child pairs at odd indices (the reference's builders and the LBVH allocate children in pairs from index 1 on, so a
// pair shares one 64-B line of the copy and "the other child" is offset ^ 32), byte offsets below 2^31 -- and a NESTED tree: inside
// the prefetching loop EXEC is the set of quads that survived the parent (a quad that fails a box fails every box inside it), so a
// child's first / last come from the survivors only, whereas the reference rescans the whole inherited range
// (src/bounding_box.cpp:71-139).  Trees of the reference's builders and of the LBVH refit are nested by construction; a caller's
// tree that is not (snail_scene_create checks every child box against its parent's) takes the two-word loop, whose EXEC is the
// inherited range [first, last] -- the reference's rescan exactly.  Workbench build: SNAIL_DEBUG_NO_PACK=1 keeps the two-word form.
int stackPack(const SnailScene *s) {
	static const bool off = debugEnvInt("SNAIL_DEBUG_NO_PACK") != 0;
	static const bool assumeNested = debugEnvInt("SNAIL_DEBUG_ASSUME_NESTED") != 0;   // (tests/nonnested_env.py: shows that the check below matters)
	return !off && (s->pfOK || (assumeNested && s->pfOKButNesting)) ? 1 : 0;
}

// The node slots relative to `org` for a launch on `stream` (stream-ordered: filled on this stream on a miss; a later user on another stream
// waits for the fill; a slot that is recycled waits for its last users).  *which = the cache entry, for relUsed() after the consumer's launch.
int relFor(SnailScene *s, const float org[3], hipStream_t stream, const uint4 **out, int *which) {
	SnailScene::RelNodes *hit = nullptr, *victim = nullptr;
	for(auto &e : s->rel) {
		if(e.valid && memcmp(e.org, org, 12) == 0) { hit = &e; break; }
		if(!victim || (!e.valid && victim->valid) || (e.valid == victim->valid && e.stamp < victim->stamp)) victim = &e;
	}
	if(hit) {
		HIP_TRY(hipStreamWaitEvent(stream, hit->filled, 0));
		hit->stamp = ++s->relClock;
		*out = hit->d; *which = (int)(hit - s->rel);
		return 0;
	}
	SnailScene::RelNodes &e = *victim;
	const int nSlots = s->nNodes + 1;
	if(!e.d) HIP_TRY(hipMalloc((void **)&e.d, (size_t)nSlots * 32));
	if(!e.filled) HIP_TRY(hipEventCreateWithFlags(&e.filled, hipEventDisableTiming));
	for(int k = 0; k < e.nUsed; k++) HIP_TRY(hipStreamWaitEvent(stream, e.used[k], 0));   // its last readers, on whatever streams
	e.nUsed = 0;
	hipLaunchKernelGGL(dev::k_rel_nodes, dim3((unsigned)((nSlots + 255) / 256)), dim3(256), 0, stream, (const uint4 *)s->dPF, nSlots, org[0], org[1], org[2], e.d);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e.filled, stream));
	memcpy(e.org, org, 12); e.valid = true; e.stamp = ++s->relClock;
	*out = e.d; *which = (int)(&e - s->rel);
	return 0;
}
// after the launch that reads entry `which` was enqueued on `stream`
int relUsed(SnailScene *s, int which, hipStream_t stream) {
	SnailScene::RelNodes &e = s->rel[which];
	int k = 0;
	while(k < e.nUsed && e.usedOn[k] != stream) k++;
	if(k == e.nUsed) {
		if(e.nUsed == SnailScene::RelNodes::kStreams) { // more streams than slots (never with the renderers of this repo): fold the oldest into this one
			HIP_TRY(hipStreamWaitEvent(stream, e.used[0], 0));
			k = 0;
		} else e.nUsed++;
		e.usedOn[k] = stream;
		if(!e.used[k]) HIP_TRY(hipEventCreateWithFlags(&e.used[k], hipEventDisableTiming));
	}
	HIP_TRY(hipEventRecord(e.used[k], stream));
	return 0;
}

int checkScene(const SnailScene *s, const char *fn) {
	if(!s || !s->dNodes || !s->dPF || !s->dTris) { snail_set_error("%s: invalid scene handle", fn); return 1; }
	return 0;
}

// the frames of one launch: cameras (13 floats each) and output planes per frame (any plane may be null)
struct FrameSet {
	int n = 0;
	const float *cam[SNAIL_MAX_BATCH] = {};
	dev::FrameOut out[SNAIL_MAX_BATCH] = {};
};

int launchPrimaryFrames(SnailScene *s, const FrameSet &FS, int resx, int resy, int x0, int y0, int w, int h, const int32_t *dPacketXY, int nPackets, uint64_t *dStats,
						hipStream_t stream, unsigned *dCost = nullptr, bool packetMajor = false, const int32_t *dOrder = nullptr, int32_t *dSlotCost = nullptr) {
	if(resx <= 0 || resy <= 0) { snail_set_error("snail_trace_primary: bad resolution %dx%d", resx, resy); return 1; }
	if(FS.n < 1 || FS.n > SNAIL_MAX_BATCH || (SNAIL_BLOCK_WAVES > 1 && FS.n > 1)) { snail_set_error("snail_trace_primary: 1..%d frames per launch (got %d)", SNAIL_MAX_BATCH, FS.n); return 1; }
	dev::PrimaryArgs A;
	memset(&A, 0, sizeof(A));
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.nFrames = FS.n;
	A.fastOK = s->fastOK;
	for(int k = 0; k < FS.n; k++) {
		A.g[k] = makeGen(FS.cam[k], resx, resy);
		A.out[k] = FS.out[k];
		A.fastOK = A.fastOK && originSane(FS.cam[k]);
	}
	A.resx = resx; A.resy = resy;
	A.stats = (dev::u64 *)dStats;
	A.cost = dCost;
	A.packetMajor = packetMajor ? 1 : 0;
	A.pack = stackPack(s);
	int relWhich[SNAIL_MAX_BATCH];
	for(int k = 0; k < FS.n; k++) {
		relWhich[k] = -1; A.rel[k] = nullptr;
		if(A.pack && SNAIL_REL_NODES && SNAIL_NODE_PREFETCH)
			if(int rc = relFor(s, FS.cam[k], stream, &A.rel[k], &relWhich[k])) return rc;   // cam[0..2] = the camera position
	}
	int blocks;
	if(dPacketXY) {
		if(nPackets <= 0) return 0;
		A.packetXY = (const int2 *)dPacketXY;
		A.nPackets = nPackets;
		blocks = ((nPackets + 127) / 128) * 128;
	} else {
		if((x0 & 15) || (y0 & 15) || w <= 0 || h <= 0 || x0 < 0 || y0 < 0) {
			snail_set_error("snail_trace_primary: rect origin must be a non-negative multiple of 16 and the size positive (got %d,%d %dx%d)", x0, y0, w, h);
			return 1;
		}
		A.x0 = x0; A.y0 = y0; A.w = w; A.h = h;
		A.pw = (w + 15) / 16; A.ph = (h + 15) / 16;
		A.nPackets = A.pw * A.ph;
		const int nRegions = ((A.pw + 3) / 4) * ((A.ph + 3) / 4);
		blocks = ((nRegions + 7) / 8) * 8 * 16;
	}
	A.nBlocks = blocks;
	A.nSlots = dPacketXY ? nPackets : blocks;
	A.order = dOrder; A.slotCost = dSlotCost;
	const int gridBlocks = blocks * FS.n;
	s->lastBlocks = gridBlocks; s->lastThreads = 64;
	// deferred-packet list of this launch (re-allocated, synchronously, only when a larger launch than ever before arrives)
	if(gridBlocks + 2 > s->deferCap) {
		HIP_TRY(hipDeviceSynchronize());
		for(int k = 0; k < SnailScene::kDeferSlots; k++) {
			if(s->dDefer[k]) (void)hipFree(s->dDefer[k]);
			s->dDefer[k] = nullptr;
			HIP_TRY(hipMalloc((void **)&s->dDefer[k], (size_t)(gridBlocks + 2) * sizeof(int)));
			HIP_TRY(hipMemset(s->dDefer[k], 0, 2 * sizeof(int)));
		}
		s->deferCap = gridBlocks + 2;
	}
	const int slot = (int)(s->launchCount++ % SnailScene::kDeferSlots);
	A.defer = s->dDefer[slot];
	if(!s->deferDone[slot]) HIP_TRY(hipEventCreateWithFlags(&s->deferDone[slot], hipEventDisableTiming));
	if(s->deferUsed[slot]) HIP_TRY(hipStreamWaitEvent(stream, s->deferDone[slot], 0));
	// workbench build, SNAIL_DEBUG_DYNLDS=<bytes>: occupancy experiments only (unused dynamic LDS limits waves per CU)
	static const int dynLds = debugEnvInt("SNAIL_DEBUG_DYNLDS");
	// a scene with sane records defers (practically) nothing: a handful of blocks suffices; an unsafe scene defers every packet
	const int exactBlocks = A.fastOK ? (blocks < 8 ? blocks : 8) : (blocks < 2048 ? blocks : 2048);
	static_assert(128 % SNAIL_BLOCK_WAVES == 0, "the slot count is a multiple of 128");
	const dim3 grid(gridBlocks / SNAIL_BLOCK_WAVES), block(64 * SNAIL_BLOCK_WAVES);
	if(dCost) { // diagnostic launch (snail_account_packets)
		if(useDeep(s)) hipLaunchKernelGGL(dev::k_primary_diag<true>, dim3(blocks), dim3(64), 0, stream, A);
		else hipLaunchKernelGGL(dev::k_primary_diag<false>, dim3(blocks), dim3(64), 0, stream, A);
	} else if(useDeep(s)) hipLaunchKernelGGL(dev::k_primary<true>, grid, block, dynLds, stream, A);
	else hipLaunchKernelGGL(dev::k_primary<false>, grid, block, dynLds, stream, A);
	if(useDeep(s)) hipLaunchKernelGGL(dev::k_primary_exact<true>, dim3(exactBlocks), dim3(64), 0, stream, A);
	else hipLaunchKernelGGL(dev::k_primary_exact<false>, dim3(exactBlocks), dim3(64), 0, stream, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(s->deferDone[slot], stream));
	s->deferUsed[slot] = true;
	for(int k = 0; k < FS.n; k++)
		if(relWhich[k] >= 0) { if(int rc = relUsed(s, relWhich[k], stream)) return rc; }
	return 0;
}

// one frame per launch (every entry point but the *_batch_dev ones)
int launchPrimary(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, const int32_t *dPacketXY,
				  int nPackets, float *t, float *u, float *v, int32_t *id, uint64_t *dStats, hipStream_t stream, unsigned *dCost = nullptr,
				  bool packetMajor = false, uint8_t *dBgr = nullptr, const int32_t *dOrder = nullptr, int32_t *dSlotCost = nullptr) {
	FrameSet FS;
	FS.n = 1; FS.cam[0] = cam;
	FS.out[0].t = t; FS.out[0].u = u; FS.out[0].v = v; FS.out[0].id = (int *)id; FS.out[0].bgr = dBgr;
	return launchPrimaryFrames(s, FS, resx, resy, x0, y0, w, h, dPacketXY, nPackets, dStats, stream, dCost, packetMajor, dOrder, dSlotCost);
}

// scratch of one staged frame: ONE allocation, carved (hitT is its base); grown synchronously when a larger frame or the first
// reflection frame arrives
int shadeScratch(SnailScene *s, SnailScene::ShadeScratch &W, size_t packets, size_t blocks, bool refl) {
	if(W.hitT && W.packets >= packets && (W.refl || !refl)) return 0;
	HIP_TRY(hipDeviceSynchronize());
	if(W.hitT) (void)hipFree(W.hitT);
	const hipEvent_t keep = W.done;
	W = SnailScene::ShadeScratch();
	W.done = keep;
	const size_t rays = packets * 256, quads = packets * 64;
	const size_t deferInts = blocks * SNAIL_MAX_LIGHTS + 16;
	size_t bytes = rays * 8 + deferInts * 4 + rays * 4 * SNAIL_MAX_LIGHTS; // hitT, hitId, defer list, sDist
	if(refl) bytes += quads * 12 * 4 * 3 + rays * 4 * 2 + rays * 12 + quads; // rOrg, rDir, rIDir; rDist, rObj; rCol; rMask
	char *base = nullptr;
	HIP_TRY(hipMalloc((void **)&base, bytes));
	W.packets = packets; W.refl = refl;
	W.hitT = (float *)base; base += rays * 4;
	W.hitId = (int *)base; base += rays * 4;
	W.defer = (int *)base; base += deferInts * 4;
	HIP_TRY(hipMemset(W.defer, 0, 16 * sizeof(int)));
	W.sDist = (float *)base; base += rays * 4 * SNAIL_MAX_LIGHTS;
	if(refl) {
		W.rOrg = (float *)base; base += quads * 48;
		W.rDir = (float *)base; base += quads * 48;
		W.rIDir = (float *)base; base += quads * 48;
		W.rDist = (float *)base; base += rays * 4;
		W.rObj = (int *)base; base += rays * 4;
		W.rCol = (float *)base; base += rays * 12;
		W.rMask = (unsigned char *)base;
	}
	return 0;
}

template <bool SHARED, bool MASK>
void launchRaysKernels(const SnailScene *s, const dev::RaysArgs &A, int blocks, int exactBlocks, hipStream_t stream) {
	const bool deep = useDeep(s), bary = A.bary != nullptr;
#define SNAIL_RAYS_LAUNCH(D, B)                                                                                                            \
	do {                                                                                                                                   \
		hipLaunchKernelGGL((dev::k_rays<SHARED, MASK, D, B>), dim3(blocks), dim3(64), 0, stream, A);                                        \
		hipLaunchKernelGGL((dev::k_rays_exact<SHARED, MASK, D, B>), dim3(exactBlocks), dim3(64), 0, stream, A);                             \
	} while(0)
	if(deep && bary) SNAIL_RAYS_LAUNCH(true, true);
	else if(deep) SNAIL_RAYS_LAUNCH(true, false);
	else if(bary) SNAIL_RAYS_LAUNCH(false, true);
	else SNAIL_RAYS_LAUNCH(false, false);
#undef SNAIL_RAYS_LAUNCH
}

int launchRays(SnailScene *s, bool shadow, int nPackets, int size, int sharedOrigin, const float *origin, const float *dir,
					  const float *idir, const uint8_t *mask, float *distance, int32_t *object, float *bary, uint64_t *dStats, hipStream_t stream) {
	if(nPackets <= 0) return 0;
	if(size < 1 || size > SNAIL_PACKET_QUADS) { snail_set_error("packet size %d outside 1..%d quads", size, SNAIL_PACKET_QUADS); return 1; }
	if(!origin || !dir || !idir || !distance || (!shadow && !object)) { snail_set_error("null ray array"); return 1; }
	dev::RaysArgs A;
	memset(&A, 0, sizeof(A));
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.nPackets = nPackets; A.size = size; A.fastOK = s->fastOK; A.pack = stackPack(s);
	A.origin = origin; A.dir = dir; A.idir = idir; A.mask = mask;
	A.distance = distance; A.object = object; A.bary = bary;
	A.stats = (dev::u64 *)dStats;
	const int blocks = ((nPackets + 127) / 128) * 128;
	SnailScene::RayDefer &R = s->rayDefer[s->rayCount++ % SnailScene::kDeferSlots];
	if((size_t)nPackets + 16 > R.cap) { // grown synchronously when a larger batch than ever before arrives
		HIP_TRY(hipDeviceSynchronize());
		if(R.p) (void)hipFree(R.p);
		R.p = nullptr; R.cap = 0;
		HIP_TRY(hipMalloc((void **)&R.p, ((size_t)nPackets + 16) * sizeof(int)));
		HIP_TRY(hipMemset(R.p, 0, 16 * sizeof(int)));
		R.cap = (size_t)nPackets + 16;
	}
	if(!R.done) HIP_TRY(hipEventCreateWithFlags(&R.done, hipEventDisableTiming));
	if(R.used) HIP_TRY(hipStreamWaitEvent(stream, R.done, 0));
	A.defer = R.p;
	const int exactBlocks = A.fastOK ? (blocks < 8 ? blocks : 8) : (blocks < 2048 ? blocks : 2048);
	if(shadow) {
		if(useDeep(s)) {
			hipLaunchKernelGGL(dev::k_shadow<true>, dim3(blocks), dim3(64), 0, stream, A);
			hipLaunchKernelGGL(dev::k_shadow_exact<true>, dim3(exactBlocks), dim3(64), 0, stream, A);
		} else {
			hipLaunchKernelGGL(dev::k_shadow<false>, dim3(blocks), dim3(64), 0, stream, A);
			hipLaunchKernelGGL(dev::k_shadow_exact<false>, dim3(exactBlocks), dim3(64), 0, stream, A);
		}
	} else if(sharedOrigin && mask) launchRaysKernels<true, true>(s, A, blocks, exactBlocks, stream);
	else if(sharedOrigin) launchRaysKernels<true, false>(s, A, blocks, exactBlocks, stream);
	else if(mask) launchRaysKernels<false, true>(s, A, blocks, exactBlocks, stream);
	else launchRaysKernels<false, false>(s, A, blocks, exactBlocks, stream);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(R.done, stream));
	R.used = true;
	return 0;
}

template <int SRC>
int launchLights(SnailScene *s, dev::ShadeArgs A /* a copy: relLight is filled in here */, hipStream_t stream) {
	if(A.nLights <= 0) return 0;
	int relWhich[SNAIL_MAX_LIGHTS];
	for(int n = 0; n < SNAIL_MAX_LIGHTS; n++) { relWhich[n] = -1; A.relLight[n] = nullptr; }
	if(SNAIL_NODE_PREFETCH && SNAIL_REL_SHADOW && A.pack && !useDeep(s))
		for(int n = 0; n < A.nLights; n++)
			if(int rc = relFor(s, A.lights[n], stream, &A.relLight[n], &relWhich[n])) return rc;   // lights[n][0..2] = the light's position
	const dim3 grid(A.nBlocks, A.nLights);
	const int total = A.nBlocks * A.nLights;
	const int exactBlocks = A.fastOK ? (total < 8 ? total : 8) : (total < 2048 ? total : 2048);
	if(useDeep(s)) {
		hipLaunchKernelGGL((dev::k_light<true, SRC>), grid, dim3(64), 0, stream, A);
		hipLaunchKernelGGL((dev::k_light_exact<true, SRC>), dim3(exactBlocks), dim3(64), 0, stream, A);
	} else {
		hipLaunchKernelGGL((dev::k_light<false, SRC>), grid, dim3(64), 0, stream, A);
		hipLaunchKernelGGL((dev::k_light_exact<false, SRC>), dim3(exactBlocks), dim3(64), 0, stream, A);
	}
	HIP_TRY(hipGetLastError());
	for(int n = 0; n < A.nLights; n++)
		if(relWhich[n] >= 0) { if(int rc = relUsed(s, relWhich[n], stream)) return rc; }
	return 0;
}

} // namespace

#include "lbvh.inc"

extern "C" {

const char *snail_last_error(void) { return g_err; }

int snail_device_count(void) {
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// byte offset of the triangle records in SnailScene::dPF: a CONSTANT (2^25 = room for 2^20 record slots), so that the prefetching loop
// turns a leaf record's triangle offset back into an index with immediates (SNAIL_PF_VISIT, L_leaf)
static size_t pfTrisOffset(int) { return (size_t)1 << 25; }

SnailScene *snail_scene_create(const void *nodes32, int nNodes, const void *tris64, int nTris, int depth, int device) {
	if(!nodes32 || !tris64 || nNodes <= 0 || nTris <= 0) { snail_set_error("snail_scene_create: empty scene"); return nullptr; }
	if(depth < 0 || depth > SNAIL_MAX_DEPTH) { snail_set_error("snail_scene_create: depth %d outside 0..BVH::maxDepth = %d", depth, SNAIL_MAX_DEPTH); return nullptr; }
	// validate topology on the host so that no kernel can index out of bounds (a GPU fault resets the node)
	const uint32_t *nw = (const uint32_t *)nodes32;
	int pairsOdd = 1; // every child pair starts at an odd index (children are allocated in pairs from index 1 on: src/bvh/tree.cpp:153-157)
	for(int i = 0; i < nNodes; i++) {
		uint32_t sub = nw[i * 8 + 6], aux = nw[i * 8 + 7];
		if(sub & 0x80000000u) {
			uint32_t first = sub & 0x7fffffffu;
			if((int32_t)aux < 0 || (uint64_t)first + aux > (uint64_t)nTris) { snail_set_error("snail_scene_create: leaf %d references triangles [%u,%u) of %d", i, first, first + aux, nTris); return nullptr; }
		} else {
			if((uint64_t)sub + 1 >= (uint64_t)nNodes || sub == 0) { snail_set_error("snail_scene_create: node %d has child %u of %d", i, sub, nNodes); return nullptr; }
			if((aux & 0xffff) > 2 || (aux >> 16) > 1) { snail_set_error("snail_scene_create: node %d has axis/firstNode %u/%u", i, aux & 0xffff, aux >> 16); return nullptr; }
			if(!(sub & 1u)) pairsOdd = 0;
		}
	}
	// ... and that no walk can run away: every node is reached at most once from the root (no cycle, no shared subtree: a back-edge
	// would keep every wave in its loop for ever) and no leaf lies deeper than the caller says (the traversal stack is sized by
	// `depth`: lane i of a VGPR pair = slot i, a second pair beyond 62 levels; a deeper tree would wrap the lane select)
	int realDepth = 0, nestedOK = 1;
	{
		std::vector<uint8_t> seen((size_t)nNodes, 0);
		std::vector<std::pair<int, int>> todo; // (node, level), root = level 0 as BVH::depth counts (src/bvh/tree.cpp:54-59)
		todo.emplace_back(0, 0);
		while(!todo.empty()) {
			const auto [i, level] = todo.back();
			todo.pop_back();
			if(seen[i]) { snail_set_error("snail_scene_create: node %d is reachable twice from the root (cycle or shared subtree)", i); return nullptr; }
			seen[i] = 1;
			if(level > realDepth) realDepth = level;
			if(level > depth) { snail_set_error("snail_scene_create: node %d lies at level %d, deeper than the declared depth %d", i, level, depth); return nullptr; }
			const uint32_t sub = nw[(size_t)i * 8 + 6];
			if(!(sub & 0x80000000u)) {
				todo.emplace_back((int)sub + 1, level + 1); todo.emplace_back((int)sub, level + 1);
				// child box inside the parent's box, per axis (a NaN bound fails the comparison and counts as not nested)
				const float *pb = (const float *)nodes32 + (size_t)i * 8;
				for(int c = 0; c < 2; c++) {
					const float *cb = (const float *)nodes32 + ((size_t)sub + c) * 8;
					for(int k = 0; k < 3; k++)
						if(!(cb[k] >= pb[k]) || !(cb[3 + k] <= pb[3 + k])) nestedOK = 0;
				}
			}
		}
	}
	int fastOK = 1;
	const float *tf = (const float *)tris64;
	for(int i = 0; i < nTris && fastOK; i++) {
		const float *r = tf + (size_t)i * 16;
		for(int k = 0; k < 9; k++) if(!(std::fabs(r[k]) <= 1.0e9f)) fastOK = 0;         // a, ba, ca
		if(!(r[9] > 0.0f) || !(r[10] <= 1.0e12f) || !(r[10] > 0.0f)) fastOK = 0;        // t0, it0
		for(int k = 12; k < 16; k++) if(!(std::fabs(r[k]) <= 1.0e18f)) fastOK = 0;     // plane
	}
	const float *nf = (const float *)nodes32;
	for(int i = 0; i < nNodes && fastOK; i++) {
		for(int k = 0; k < 6; k++) if(!(std::fabs(nf[(size_t)i * 8 + k]) <= 1.0e9f)) fastOK = 0;
		for(int k = 0; k < 3; k++) if(!(nf[(size_t)i * 8 + k] <= nf[(size_t)i * 8 + 3 + k])) fastOK = 0;   // M_COH needs min <= max
	}

	DeviceGuard guard(device);
	if(!guard.ok) { snail_set_error("snail_scene_create: hipSetDevice(%d) failed", device); return nullptr; }
	SnailScene *s = new SnailScene();
	s->device = device; s->nNodes = nNodes; s->nTris = nTris; s->depth = realDepth; s->fastOK = fastOK; s->nestedOK = nestedOK; // the measured depth (<= declared) picks the stack form
	// ONE allocation: [slot 0][the node records re-encoded for the record-prefetching loop][triangle records]; the caller's node
	// records, verbatim, in a second one (every other walk reads those)
	const size_t trisOff = pfTrisOffset(nNodes), pfBytes = trisOff + (size_t)nTris * 64;
	const bool fits = (size_t)nNodes + 1 < ((size_t)1 << 20) && pfBytes < ((size_t)1 << 31);
	s->pfOKButNesting = pairsOdd && fits;
	s->pfOK = s->pfOKButNesting && nestedOK;
	s->trisOff = (int)(fits ? trisOff : 0);
	std::vector<uint32_t> pf;
	if(s->pfOKButNesting) {
		pf.assign(((size_t)nNodes + 1) * 8, 0u);
		for(int i = 0; i < nNodes; i++) {
			unsigned in[8], out[8];
			for(int k = 0; k < 8; k++) in[k] = nw[(size_t)i * 8 + k];
			dev::pfEncode(in, out, (unsigned)trisOff);
			for(int k = 0; k < 8; k++) pf[((size_t)i + 1) * 8 + k] = out[k];
		}
	}
	hipError_t e;
	if((e = hipMalloc((void **)&s->dNodes, (size_t)nNodes * 32)) != hipSuccess || (e = hipMalloc((void **)&s->dPF, fits ? pfBytes : (size_t)nTris * 64)) != hipSuccess ||
	   (e = hipMalloc((void **)&s->dStats, 4 * sizeof(unsigned long long))) != hipSuccess ||
	   (e = hipMemcpy(s->dNodes, nodes32, (size_t)nNodes * 32, hipMemcpyHostToDevice)) != hipSuccess ||
	   (e = hipMemcpy(s->dPF + s->trisOff, tris64, (size_t)nTris * 64, hipMemcpyHostToDevice)) != hipSuccess ||
	   (!pf.empty() && (e = hipMemcpy(s->dPF, pf.data(), pf.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)) {
		snail_set_error("snail_scene_create: %s", hipGetErrorString(e));
		snail_scene_destroy(s);
		return nullptr;
	}
	s->dTris = (uint4 *)(s->dPF + s->trisOff);
	return s;
}

SnailScene *snail_scene_create_lbvh(const float *tri_verts, int nTris, int device, int maxLeafTris, int32_t *perm, float *build_ms) {
	if(!tri_verts || nTris <= 0 || maxLeafTris < 1 || maxLeafTris > 64) { snail_set_error("snail_scene_create_lbvh: bad arguments (1 <= maxLeafTris <= 64)"); return nullptr; }
	DeviceGuard guard(device);
	if(!guard.ok) { snail_set_error("snail_scene_create_lbvh: hipSetDevice(%d) failed", device); return nullptr; }
	uint4 *dNodes = nullptr, *dTris = nullptr;
	int depth = 0, fastOK = 0;
	if(buildLbvhDevice(tri_verts, nTris, maxLeafTris, &dNodes, &dTris, perm, &depth, &fastOK, build_ms)) return nullptr;
	if(depth > SNAIL_MAX_DEPTH) {
		(void)hipFree(dNodes); (void)hipFree(dTris);
		snail_set_error("snail_scene_create_lbvh: depth %d exceeds BVH::maxDepth %d", depth, SNAIL_MAX_DEPTH);
		return nullptr;
	}
	SnailScene *s = new SnailScene();
	s->device = device; s->nNodes = 2 * nTris - 1; s->nTris = nTris; s->depth = depth; s->fastOK = fastOK;
	s->dNodes = dNodes;
	// the LBVH is nested by construction (bottom-up refit: a parent's box is the exact union of its children's) and its children live in
	// slots 1 + 2i, 2 + 2i: the record-prefetching loop's copy is encoded on the device, the triangle records move behind it
	const size_t trisOff = pfTrisOffset(s->nNodes), pfBytes = trisOff + (size_t)nTris * 64;
	const bool fits = (size_t)s->nNodes + 1 < ((size_t)1 << 20) && pfBytes < ((size_t)1 << 31);
	s->trisOff = (int)(fits ? trisOff : 0);
	s->nestedOK = 1; s->pfOKButNesting = s->pfOK = fits ? 1 : 0;
	hipError_t e;
	if((e = hipMalloc((void **)&s->dPF, fits ? pfBytes : (size_t)nTris * 64)) != hipSuccess ||
	   (e = hipMemcpy(s->dPF + s->trisOff, dTris, (size_t)nTris * 64, hipMemcpyDeviceToDevice)) != hipSuccess ||
	   (e = hipMalloc((void **)&s->dStats, 4 * sizeof(unsigned long long))) != hipSuccess) {
		(void)hipFree(dTris);
		snail_set_error("snail_scene_create_lbvh: %s", hipGetErrorString(e));
		snail_scene_destroy(s);
		return nullptr;
	}
	(void)hipFree(dTris);
	s->dTris = (uint4 *)(s->dPF + s->trisOff);
	if(fits) {
		(void)hipMemset(s->dPF, 0, 32);
		hipLaunchKernelGGL(dev::k_pf_encode, dim3((unsigned)((s->nNodes + 255) / 256)), dim3(256), 0, 0, s->dNodes, s->nNodes, (unsigned)trisOff, (uint4 *)s->dPF);
		if((e = hipDeviceSynchronize()) != hipSuccess) {
			snail_set_error("snail_scene_create_lbvh: %s", hipGetErrorString(e));
			snail_scene_destroy(s);
			return nullptr;
		}
	}
	return s;
}

int snail_scene_download(const SnailScene *s, void *nodes32, void *tris64) {
	if(int rc = checkScene(s, "snail_scene_download")) return rc;
	DeviceGuard guard(s->device);
	if(nodes32) HIP_TRY(hipMemcpy(nodes32, s->dNodes, (size_t)s->nNodes * 32, hipMemcpyDeviceToHost));
	if(tris64) HIP_TRY(hipMemcpy(tris64, s->dTris, (size_t)s->nTris * 64, hipMemcpyDeviceToHost));
	return 0;
}

void snail_scene_destroy(SnailScene *s) {
	if(!s) return;
	DeviceGuard guard(s->device);
	freeTileJobs(s);
	if(s->dNodes) (void)hipFree(s->dNodes);
	if(s->dPF) (void)hipFree(s->dPF);   // (dTris points into it)
	for(auto &e : s->rel) {
		if(e.d) (void)hipFree(e.d);
		if(e.filled) (void)hipEventDestroy(e.filled);
		for(hipEvent_t ev : e.used) if(ev) (void)hipEventDestroy(ev);
	}
	if(s->dStats) (void)hipFree(s->dStats);
	for(int k = 0; k < SnailScene::kDeferSlots; k++) if(s->dDefer[k]) (void)hipFree(s->dDefer[k]);
	for(int k = 0; k < SnailScene::kDeferSlots; k++) if(s->shade[k].hitT) (void)hipFree(s->shade[k].hitT);
	for(int k = 0; k < SnailScene::kDeferSlots; k++) {
		if(s->rayDefer[k].p) (void)hipFree(s->rayDefer[k].p);
		if(s->rayDefer[k].done) (void)hipEventDestroy(s->rayDefer[k].done);
		if(s->deferDone[k]) (void)hipEventDestroy(s->deferDone[k]);
		if(s->shade[k].done) (void)hipEventDestroy(s->shade[k].done);
	}
	delete s;
}

int snail_scene_info(const SnailScene *s, int *nNodes, int *nTris, int *depth, int *device) {
	if(!s) { snail_set_error("snail_scene_info: null scene"); return 1; }
	if(nNodes) *nNodes = s->nNodes;
	if(nTris) *nTris = s->nTris;
	if(depth) *depth = s->depth;
	if(device) *device = s->device;
	return 0;
}

int snail_scene_flags(const SnailScene *s, int *fastOK, int *nestedOK) {
	if(!s) { snail_set_error("snail_scene_flags: null scene"); return 1; }
	if(fastOK) *fastOK = s->fastOK;
	if(nestedOK) *nestedOK = s->nestedOK;
	return 0;
}

int snail_last_launch(const SnailScene *s, int *blocks, int *threads) {
	if(!s) { snail_set_error("snail_last_launch: null scene"); return 1; }
	if(blocks) *blocks = s->lastBlocks;
	if(threads) *threads = s->lastThreads;
	return 0;
}

int snail_trace_primary_dev(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, float *t, float *u,
							float *v, int32_t *id, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_primary_dev")) return rc;
	DeviceGuard guard(s->device);
	return launchPrimary(s, cam, resx, resy, x0, y0, w, h, nullptr, 0, t, u, v, id, dStats, (hipStream_t)stream);
}

int snail_primary_slots(int w, int h) {
	if(w <= 0 || h <= 0) return 0;
	const int pw = (w + 15) / 16, ph = (h + 15) / 16;
	return (((pw + 3) / 4) * ((ph + 3) / 4) + 7) / 8 * 8 * 16;
}

int snail_trace_primary_ordered_dev(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, float *t, float *u,
									float *v, int32_t *id, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, void *stream) {
	if(int rc = checkScene(s, "snail_trace_primary_ordered_dev")) return rc;
	DeviceGuard guard(s->device);
	return launchPrimary(s, cam, resx, resy, x0, y0, w, h, nullptr, 0, t, u, v, id, dStats, (hipStream_t)stream, nullptr, false, nullptr, dOrder, dSlotCost);
}

int snail_trace_packets_ordered_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, float *t,
									float *u, float *v, int32_t *id, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, void *stream) {
	if(int rc = checkScene(s, "snail_trace_packets_ordered_dev")) return rc;
	if(!dPacketXY && nPackets > 0) { snail_set_error("snail_trace_packets_ordered_dev: null packet list"); return 1; }
	if(nPackets <= 0) return 0; // a rank without tiles (the reference's server renders nothing)
	DeviceGuard guard(s->device);
	return launchPrimary(s, cam, resx, resy, 0, 0, 0, 0, dPacketXY, nPackets, t, u, v, id, dStats, (hipStream_t)stream, nullptr, false, nullptr, dOrder, dSlotCost);
}

int snail_order_from_cost_dev(const int32_t *dSlotCost, int nSlots, int32_t *dOrder, void *stream) {
	if(nSlots <= 0) return 0;
	if(!dSlotCost || !dOrder) { snail_set_error("snail_order_from_cost_dev: null buffer"); return 1; }
	hipLaunchKernelGGL(dev::k_order_from_cost, dim3(1), dim3(ORDER_THREADS), 0, (hipStream_t)stream, dSlotCost, nSlots, dOrder);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_trace_packets_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, float *t,
							float *u, float *v, int32_t *id, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_packets_dev")) return rc;
	if(!dPacketXY && nPackets > 0) { snail_set_error("snail_trace_packets_dev: null packet list"); return 1; }
	if(nPackets <= 0) return 0; // a rank without tiles (the reference's server renders nothing)
	DeviceGuard guard(s->device);
	return launchPrimary(s, cam, resx, resy, 0, 0, 0, 0, dPacketXY, nPackets, t, u, v, id, dStats, (hipStream_t)stream);
}

static int frameSetFrom(const char *fn, FrameSet &FS, int nFrames, const float *cams13) {
	if(nFrames < 1 || nFrames > SNAIL_MAX_BATCH || !cams13) { snail_set_error("%s: 1..%d frames per launch and their cameras are required (got %d)", fn, SNAIL_MAX_BATCH, nFrames); return 1; }
	FS.n = nFrames;
	for(int k = 0; k < nFrames; k++) FS.cam[k] = cams13 + (size_t)k * 13;
	return 0;
}

int snail_trace_primary_batch_dev(SnailScene *s, int nFrames, const float *cams13, int resx, int resy, float *const *t, float *const *u, float *const *v,
								  int32_t *const *id, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, void *stream) {
	if(int rc = checkScene(s, "snail_trace_primary_batch_dev")) return rc;
	FrameSet FS;
	if(int rc = frameSetFrom("snail_trace_primary_batch_dev", FS, nFrames, cams13)) return rc;
	for(int k = 0; k < nFrames; k++) {
		FS.out[k].t = t ? t[k] : nullptr; FS.out[k].u = u ? u[k] : nullptr; FS.out[k].v = v ? v[k] : nullptr; FS.out[k].id = id ? (int *)id[k] : nullptr;
	}
	DeviceGuard guard(s->device);
	return launchPrimaryFrames(s, FS, resx, resy, 0, 0, resx, resy, nullptr, 0, dStats, (hipStream_t)stream, nullptr, false, dOrder, dSlotCost);
}

int snail_trace_packets_shaded_batch_dev(SnailScene *s, int nFrames, const float *cams13, int resx, int resy, const int32_t *dPacketXY, int nPackets,
										 uint8_t *const *bgr, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_packets_shaded_batch_dev")) return rc;
	if((!dPacketXY || !bgr) && nPackets > 0) { snail_set_error("snail_trace_packets_shaded_batch_dev: null buffer"); return 1; }
	if(nPackets <= 0) return 0;
	FrameSet FS;
	if(int rc = frameSetFrom("snail_trace_packets_shaded_batch_dev", FS, nFrames, cams13)) return rc;
	for(int k = 0; k < nFrames; k++) {
		if(!bgr[k] || ((unsigned long long)bgr[k] & 3)) { snail_set_error("snail_trace_packets_shaded_batch_dev: d_bgr[%d] must be a 4-byte aligned device pointer", k); return 1; }
		FS.out[k].bgr = bgr[k];
	}
	DeviceGuard guard(s->device);
	return launchPrimaryFrames(s, FS, resx, resy, 0, 0, 0, 0, dPacketXY, nPackets, dStats, (hipStream_t)stream);
}

int snail_trace_packets_shaded_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, uint8_t *bgr,
								   uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_packets_shaded_dev")) return rc;
	if((!dPacketXY || !bgr) && nPackets > 0) { snail_set_error("snail_trace_packets_shaded_dev: null buffer"); return 1; }
	if(nPackets <= 0) return 0;
	if((unsigned long long)bgr & 3) { snail_set_error("snail_trace_packets_shaded_dev: d_bgr must be 4-byte aligned"); return 1; }
	DeviceGuard guard(s->device);
	return launchPrimary(s, cam, resx, resy, 0, 0, 0, 0, dPacketXY, nPackets, nullptr, nullptr, nullptr, nullptr, dStats, (hipStream_t)stream, nullptr, false, bgr);
}

int snail_packets_to_frame_dev(const int32_t *dPacketXY, int nPackets, int resx, int resy, const float *pt, const float *pu, const float *pv,
							   const int32_t *pid, float *t, float *u, float *v, int32_t *id, void *stream) {
	if(nPackets <= 0) return 0;
	dev::ScatterArgs A{(const int2 *)dPacketXY, nPackets, resx, resy, pt, pu, pv, pid, t, u, v, id};
	hipLaunchKernelGGL(dev::k_packets_to_frame, dim3((nPackets + 3) / 4), dim3(256), 0, (hipStream_t)stream, A);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_trace_primary(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, float *t, float *u, float *v,
						int32_t *id, uint64_t stats[4]) {
	if(int rc = checkScene(s, "snail_trace_primary")) return rc;
	DeviceGuard guard(s->device);
	const size_t n = (size_t)resx * resy;
	float *dt = nullptr, *du = nullptr, *dv = nullptr;
	int32_t *did = nullptr;
	int rc = 0;
	auto freeAll = [&] { if(dt) (void)hipFree(dt); if(du) (void)hipFree(du); if(dv) (void)hipFree(dv); if(did) (void)hipFree(did); };
#define TRY_OR_FREE(e) do { hipError_t e_ = (e); if(e_ != hipSuccess) { snail_set_error("%s: %s", #e, hipGetErrorString(e_)); freeAll(); return 100 + (int)e_; } } while(0)
	// the host buffers are full frames of which only the rect is defined: stage through device frames
	if(t) { TRY_OR_FREE(hipMalloc((void **)&dt, n * 4)); TRY_OR_FREE(hipMemcpy(dt, t, n * 4, hipMemcpyHostToDevice)); }
	if(u) { TRY_OR_FREE(hipMalloc((void **)&du, n * 4)); TRY_OR_FREE(hipMemcpy(du, u, n * 4, hipMemcpyHostToDevice)); }
	if(v) { TRY_OR_FREE(hipMalloc((void **)&dv, n * 4)); TRY_OR_FREE(hipMemcpy(dv, v, n * 4, hipMemcpyHostToDevice)); }
	if(id) { TRY_OR_FREE(hipMalloc((void **)&did, n * 4)); TRY_OR_FREE(hipMemcpy(did, id, n * 4, hipMemcpyHostToDevice)); }
	if(stats) TRY_OR_FREE(hipMemset(s->dStats, 0, 32));
	rc = launchPrimary(s, cam, resx, resy, x0, y0, w, h, nullptr, 0, dt, du, dv, did, stats ? (uint64_t *)s->dStats : nullptr, 0);
	if(rc) { freeAll(); return rc; }
	TRY_OR_FREE(hipDeviceSynchronize());
	if(t) TRY_OR_FREE(hipMemcpy(t, dt, n * 4, hipMemcpyDeviceToHost));
	if(u) TRY_OR_FREE(hipMemcpy(u, du, n * 4, hipMemcpyDeviceToHost));
	if(v) TRY_OR_FREE(hipMemcpy(v, dv, n * 4, hipMemcpyDeviceToHost));
	if(id) TRY_OR_FREE(hipMemcpy(id, did, n * 4, hipMemcpyDeviceToHost));
	if(stats) {
		unsigned long long hs[4];
		TRY_OR_FREE(hipMemcpy(hs, s->dStats, 32, hipMemcpyDeviceToHost));
		for(int k = 0; k < 4; k++) stats[k] += hs[k];
	}
	freeAll();
	return 0;
}


int snail_trace_rays_dev(SnailScene *s, int nPackets, int size, int sharedOrigin, const float *origin, const float *dir, const float *idir,
						 const uint8_t *mask, float *distance, int32_t *object, float *bary, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_rays_dev")) return rc;
	DeviceGuard guard(s->device);
	return launchRays(s, false, nPackets, size, sharedOrigin, origin, dir, idir, mask, distance, object, bary, dStats, (hipStream_t)stream);
}

int snail_trace_shadow_dev(SnailScene *s, int nPackets, int size, const float *origin3, const float *dir, const float *idir, float *distance,
						   uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_shadow_dev")) return rc;
	DeviceGuard guard(s->device);
	return launchRays(s, true, nPackets, size, 1, origin3, dir, idir, nullptr, distance, nullptr, nullptr, dStats, (hipStream_t)stream);
}

namespace {
struct DevBuf {
	void *p = nullptr;
	~DevBuf() { if(p) (void)hipFree(p); }
	int upload(const void *src, size_t bytes) {
		if(hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return 1;
		if(src && bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
		return 0;
	}
	int download(void *dst, size_t bytes) { return hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost) != hipSuccess; }
};
} // namespace

int snail_trace_rays(SnailScene *s, int nPackets, int size, int sharedOrigin, const float *origin, const float *dir, const float *idir,
					 const uint8_t *mask, float *distance, int32_t *object, float *bary, uint64_t stats[4]) {
	if(int rc = checkScene(s, "snail_trace_rays")) return rc;
	if(nPackets <= 0) return 0;
	DeviceGuard guard(s->device);
	const size_t nq = (size_t)nPackets * size;
	DevBuf o, d, i, m, ds, ob, ba;
	if(o.upload(origin, (sharedOrigin ? (size_t)nPackets : nq) * 48) || d.upload(dir, nq * 48) || i.upload(idir, nq * 48) ||
	   (mask && m.upload(mask, nq)) || ds.upload(distance, nq * 16) || ob.upload(object, nq * 16) || (bary && ba.upload(bary, nq * 32))) {
		snail_set_error("snail_trace_rays: device staging failed");
		return 2;
	}
	if(stats) HIP_TRY(hipMemset(s->dStats, 0, 32));
	int rc = launchRays(s, false, nPackets, size, sharedOrigin, (float *)o.p, (float *)d.p, (float *)i.p, mask ? (uint8_t *)m.p : nullptr, (float *)ds.p,
						(int32_t *)ob.p, bary ? (float *)ba.p : nullptr, stats ? (uint64_t *)s->dStats : nullptr, 0);
	if(rc) return rc;
	HIP_TRY(hipDeviceSynchronize());
	if(ds.download(distance, nq * 16) || ob.download(object, nq * 16) || (bary && ba.download(bary, nq * 32))) { snail_set_error("snail_trace_rays: download failed"); return 2; }
	if(stats) {
		unsigned long long hs[4];
		HIP_TRY(hipMemcpy(hs, s->dStats, 32, hipMemcpyDeviceToHost));
		for(int k = 0; k < 4; k++) stats[k] += hs[k];
	}
	return 0;
}

int snail_trace_shadow(SnailScene *s, int nPackets, int size, const float *origin3, const float *dir, const float *idir, float *distance,
					   uint64_t stats[4]) {
	if(int rc = checkScene(s, "snail_trace_shadow")) return rc;
	if(nPackets <= 0) return 0;
	DeviceGuard guard(s->device);
	const size_t nq = (size_t)nPackets * size;
	DevBuf o, d, i, ds;
	if(o.upload(origin3, (size_t)nPackets * 12) || d.upload(dir, nq * 48) || i.upload(idir, nq * 48) || ds.upload(distance, nq * 16)) {
		snail_set_error("snail_trace_shadow: device staging failed");
		return 2;
	}
	if(stats) HIP_TRY(hipMemset(s->dStats, 0, 32));
	int rc = launchRays(s, true, nPackets, size, 1, (float *)o.p, (float *)d.p, (float *)i.p, nullptr, (float *)ds.p, nullptr, nullptr,
						stats ? (uint64_t *)s->dStats : nullptr, 0);
	if(rc) return rc;
	HIP_TRY(hipDeviceSynchronize());
	if(ds.download(distance, nq * 16)) { snail_set_error("snail_trace_shadow: download failed"); return 2; }
	if(stats) {
		unsigned long long hs[4];
		HIP_TRY(hipMemcpy(hs, s->dStats, 32, hipMemcpyDeviceToHost));
		for(int k = 0; k < 4; k++) stats[k] += hs[k];
	}
	return 0;
}

static int renderWhitted(const char *fn, SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPacketsList, const float *lights7,
						 int nLights, const float ambient[3], const float color[3], int flags, uint8_t *frame, int pitch, uint8_t *bgrPackets, uint64_t *dStats,
						 void *stream, float *colPackets = nullptr) {
	if(int rc = checkScene(s, fn)) return rc;
	if(resx <= 0 || resy <= 0 || nLights < 0 || nLights > SNAIL_MAX_LIGHTS || (nLights && !lights7) || !ambient || !color || (flags & ~SNAIL_WHITTED_REFLECTIONS) ||
	   (dPacketXY ? (colPackets ? false : (!bgrPackets || ((unsigned long long)bgrPackets & 3))) : (!frame || pitch < resx * 3 || colPackets))) {
		snail_set_error("%s: bad arguments (at most %d lights; flags = SNAIL_WHITTED_REFLECTIONS or 0; 4-byte aligned output)", fn, SNAIL_MAX_LIGHTS);
		return 1;
	}
	if(dPacketXY && nPacketsList <= 0) return 0;
	DeviceGuard guard(s->device);
	const bool refl = (flags & SNAIL_WHITTED_REFLECTIONS) != 0;
	dev::ShadeArgs A;
	memset(&A, 0, sizeof(A));
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.g = makeGen(cam, resx, resy);
	A.resx = resx; A.resy = resy; A.pw = (resx + 15) / 16; A.ph = (resy + 15) / 16;
	A.fastOK = s->fastOK && originSane(cam);
	A.pack = stackPack(s);
	A.nLights = nLights;
	for(int n = 0; n < nLights; n++) for(int k = 0; k < 7; k++) A.lights[n][k] = lights7[n * 7 + k];
	for(int c = 0; c < 3; c++) { A.ambient[c] = ambient[c]; A.color[c] = color[c]; }
	A.frame = frame; A.pitch = pitch; A.stats = (dev::u64 *)dStats;
	A.packetXY = (const int2 *)dPacketXY; A.bgrPackets = bgrPackets; A.colPackets = colPackets;
	int packets, blocks;
	if(dPacketXY) {
		packets = nPacketsList;
		blocks = ((packets + 127) / 128) * 128;
	} else {
		packets = A.pw * A.ph;
		const int nRegions = ((A.pw + 3) / 4) * ((A.ph + 3) / 4);
		blocks = ((nRegions + 7) / 8) * 8 * 16;
	}
	A.nPackets = packets;
	A.nBlocks = blocks;
	SnailScene::ShadeScratch &W = s->shade[s->shadeCount++ % SnailScene::kDeferSlots];
	if(int rc = shadeScratch(s, W, (size_t)packets, (size_t)blocks, refl)) return rc;
	if(!W.done) HIP_TRY(hipEventCreateWithFlags(&W.done, hipEventDisableTiming));
	if(W.used) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, W.done, 0));
	A.hitT = W.hitT; A.hitId = W.hitId;
	A.rOrg = W.rOrg; A.rDir = W.rDir; A.rIDir = W.rIDir; A.rMask = W.rMask; A.rDist = W.rDist; A.rObj = W.rObj; A.rCol = W.rCol;
	A.sDist = W.sDist;
	A.blend = refl ? 1 : 0;
	A.defer = W.defer;
	const hipStream_t st = (hipStream_t)stream;
	const dim3 grid(blocks), wave(64);
	// the primary packets (the bench kernel), hit records packet-major
	if(dPacketXY) {
		if(int rc = launchPrimary(s, cam, resx, resy, 0, 0, 0, 0, dPacketXY, packets, W.hitT, nullptr, nullptr, W.hitId, dStats, st)) return rc;
	} else if(int rc = launchPrimary(s, cam, resx, resy, 0, 0, resx, resy, nullptr, 0, W.hitT, nullptr, nullptr, W.hitId, dStats, st, nullptr, true)) return rc;
	if(refl) { // the nested RayTrace of the mirrored packets
		hipLaunchKernelGGL((dev::k_final<dev::SRC_PRIMARY, dev::DST_MIRROR>), grid, wave, 0, st, A);
		HIP_TRY(hipGetLastError());
		if(int rc = launchRays(s, false, packets, 64, 0, W.rOrg, W.rDir, W.rIDir, W.rMask, W.rDist, W.rObj, nullptr, dStats, st)) return rc;
		if(int rc = launchLights<dev::SRC_MIRROR>(s, A, st)) return rc;
		hipLaunchKernelGGL((dev::k_final<dev::SRC_MIRROR, dev::DST_COLOR>), grid, wave, 0, st, A);
		HIP_TRY(hipGetLastError());
	}
	if(int rc = launchLights<dev::SRC_PRIMARY>(s, A, st)) return rc;
	hipLaunchKernelGGL((dev::k_final<dev::SRC_PRIMARY, dev::DST_FRAME>), grid, wave, 0, st, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(W.done, st));
	W.used = true;
	return 0;
}

int snail_render_whitted_dev(SnailScene *s, const float cam[13], int resx, int resy, const float *lights7, int nLights, const float ambient[3],
							 const float color[3], int flags, uint8_t *frame, int pitch, uint64_t *dStats, void *stream) {
	return renderWhitted("snail_render_whitted_dev", s, cam, resx, resy, nullptr, 0, lights7, nLights, ambient, color, flags, frame, pitch, nullptr, dStats, stream);
}

int snail_render_whitted_packets_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, const float *lights7,
									 int nLights, const float ambient[3], const float color[3], int flags, uint8_t *bgrPackets, uint64_t *dStats, void *stream) {
	if(!dPacketXY && nPackets > 0) { snail_set_error("snail_render_whitted_packets_dev: null packet list"); return 1; }
	if(nPackets <= 0) return 0;
	return renderWhitted("snail_render_whitted_packets_dev", s, cam, resx, resy, dPacketXY, nPackets, lights7, nLights, ambient, color, flags, nullptr, 0, bgrPackets,
						 dStats, stream);
}

int snail_trace_transparency_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, const float *dT, const int32_t *dTriId,
								 const uint8_t *dSel, const float *lights7, int nLights, const float ambient[3], const float color[3], float *dColor, uint64_t *dStats,
								 void *stream) {
	if(int rc = checkScene(s, "snail_trace_transparency_dev")) return rc;
	if(nPackets <= 0) return 0;
	if(!dPacketXY || !dT || !dTriId || !dSel || !dColor || resx <= 0 || resy <= 0 || nLights < 0 || nLights > SNAIL_MAX_LIGHTS || (nLights && !lights7) || !ambient || !color ||
	   ((unsigned long long)dColor & 3)) {
		snail_set_error("snail_trace_transparency_dev: bad arguments (null buffer, or more than %d lights)", SNAIL_MAX_LIGHTS);
		return 1;
	}
	DeviceGuard guard(s->device);
	dev::ShadeArgs A;
	memset(&A, 0, sizeof(A));
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.g = makeGen(cam, resx, resy);
	A.resx = resx; A.resy = resy; A.pw = (resx + 15) / 16; A.ph = (resy + 15) / 16;
	A.fastOK = s->fastOK && originSane(cam);
	A.pack = stackPack(s);
	A.nLights = nLights;
	for(int n = 0; n < nLights; n++) for(int k = 0; k < 7; k++) A.lights[n][k] = lights7[n * 7 + k];
	for(int c = 0; c < 3; c++) { A.ambient[c] = ambient[c]; A.color[c] = color[c]; }
	A.stats = (dev::u64 *)dStats;
	A.packetXY = (const int2 *)dPacketXY;
	const int blocks = ((nPackets + 127) / 128) * 128;
	A.nPackets = nPackets; A.nBlocks = blocks;
	SnailScene::ShadeScratch &W = s->shade[s->shadeCount++ % SnailScene::kDeferSlots];
	if(int rc = shadeScratch(s, W, (size_t)nPackets, (size_t)blocks, true)) return rc;
	if(!W.done) HIP_TRY(hipEventCreateWithFlags(&W.done, hipEventDisableTiming));
	if(W.used) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, W.done, 0));
	A.hitT = dT; A.hitId = dTriId; A.selIn = dSel;   // the caller's hit records (Context::distance / object of the packets)
	A.rOrg = W.rOrg; A.rDir = W.rDir; A.rIDir = W.rIDir; A.rMask = W.rMask; A.rDist = W.rDist; A.rObj = W.rObj;
	A.rCol = dColor;                                  // the nested RayTrace's colours go straight to the caller
	A.sDist = W.sDist; A.defer = W.defer;
	const hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL((dev::k_final<dev::SRC_PRIMARY, dev::DST_CONTINUE>), dim3(blocks), dim3(64), 0, st, A);
	HIP_TRY(hipGetLastError());
	if(int rc = launchRays(s, false, nPackets, 64, 0, W.rOrg, W.rDir, W.rIDir, W.rMask, W.rDist, W.rObj, nullptr, dStats, st)) return rc;
	if(int rc = launchLights<dev::SRC_MIRROR>(s, A, st)) return rc;
	hipLaunchKernelGGL((dev::k_final<dev::SRC_MIRROR, dev::DST_COLOR>), dim3(blocks), dim3(64), 0, st, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(W.done, st));
	W.used = true;
	return 0;
}

int snail_shade_depth_dev(const float *t, int nPackets, uint8_t *bgr, void *stream) {
	if(nPackets <= 0) return 0;
	if(!t || !bgr) { snail_set_error("snail_shade_depth_dev: null buffer"); return 1; }
	const int n = nPackets * 256;
	hipLaunchKernelGGL(dev::k_shade_depth, dim3((n / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, n, bgr);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_packets_bgr_to_frame_dev(const int32_t *dPacketXY, int nPackets, int resx, int resy, const uint8_t *bgr, uint8_t *frame, int pitch,
								   void *stream) {
	if(nPackets <= 0) return 0;
	if(!dPacketXY || !bgr || !frame || pitch < resx * 3) { snail_set_error("snail_packets_bgr_to_frame_dev: bad arguments"); return 1; }
	hipLaunchKernelGGL(dev::k_bgr_to_frame, dim3((nPackets + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const int2 *)dPacketXY, nPackets, resx,
					   resy, bgr, frame, pitch, 0, 0ll);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_packets_bgr_to_frame_chunked_dev(const int32_t *dPacketXY, int nPackets, int nPerChunk, int64_t chunkStrideBytes, int resx, int resy, const uint8_t *bgr,
										   uint8_t *frame, int pitch, void *stream) {
	if(nPackets <= 0) return 0;
	if(!dPacketXY || !bgr || !frame || pitch < resx * 3 || nPerChunk <= 0 || chunkStrideBytes < (int64_t)nPerChunk * 768 || (chunkStrideBytes & 3)) {
		snail_set_error("snail_packets_bgr_to_frame_chunked_dev: bad arguments");
		return 1;
	}
	hipLaunchKernelGGL(dev::k_bgr_to_frame, dim3((nPackets + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const int2 *)dPacketXY, nPackets, resx,
					   resy, bgr, frame, pitch, nPerChunk, (long long)chunkStrideBytes);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_packets_bgr_to_planar_dev(const int32_t *dTiles, const int32_t *dFirstPacket, const int64_t *dOutOffsets, int nTiles, const uint8_t *bgr,
									uint8_t *out, void *stream) {
	if(nTiles <= 0) return 0;
	if(!dTiles || !dFirstPacket || !dOutOffsets || !bgr || !out) { snail_set_error("snail_packets_bgr_to_planar_dev: bad arguments"); return 1; }
	if(nTiles > 65535) { snail_set_error("snail_packets_bgr_to_planar_dev: at most 65535 tiles per call (got %d)", nTiles); return 1; }
	hipLaunchKernelGGL(dev::k_bgr_to_planar, dim3(4, nTiles), dim3(256), 0, (hipStream_t)stream, (const int4 *)dTiles, dFirstPacket,
					   (const long long *)dOutOffsets, nTiles, bgr, out);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_planar_to_frame_dev(const int32_t *dTiles, const int64_t *dInOffsets, int nTiles, const uint8_t *planar, uint8_t *frame, int pitch, int resx,
							  int resy, void *stream) {
	if(nTiles <= 0) return 0;
	if(!dTiles || !dInOffsets || !planar || !frame || pitch < resx * 3) { snail_set_error("snail_planar_to_frame_dev: bad arguments"); return 1; }
	if(nTiles > 65535) { snail_set_error("snail_planar_to_frame_dev: at most 65535 tiles per call (got %d)", nTiles); return 1; }
	hipLaunchKernelGGL(dev::k_planar_to_frame, dim3(4, nTiles), dim3(256), 0, (hipStream_t)stream, (const int4 *)dTiles, (const long long *)dInOffsets,
					   nTiles, planar, frame, pitch, resx, resy);
	HIP_TRY(hipGetLastError());
	return 0;
}

#ifdef SNAIL_DEBUG_API
int snail_debug_delay_dev(float microseconds, void *stream) {
	if(!(microseconds >= 0.0f) || microseconds > 10000.0f) { snail_set_error("snail_debug_delay_dev: delay outside 0..10000 us"); return 1; }
	const unsigned ticks = (unsigned)(microseconds * 100.0f);
	if(ticks == 0) return 0;
	hipLaunchKernelGGL(dev::k_delay, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_debug_clock_dev(float microseconds, uint64_t *dOut2, void *stream) {
	if(!(microseconds > 0.0f) || microseconds > 10000.0f || !dOut2) { snail_set_error("snail_debug_clock_dev: bad arguments"); return 1; }
	hipLaunchKernelGGL(dev::k_clock, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned)(microseconds * 100.0f), (unsigned long long *)dOut2);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_debug_recip_check(uint64_t out2[2]) {
	if(!out2) { snail_set_error("snail_debug_recip_check: bad arguments"); return 1; }
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc(&d, 16));
	HIP_TRY(hipMemset(d, 0, 16));
	for(unsigned part = 0; part < 256; part++) hipLaunchKernelGGL(dev::k_recip_check, dim3(1u << 16), dim3(256), 0, 0, part << 24, d);
	hipError_t e = hipDeviceSynchronize();
	if(e == hipSuccess) e = hipMemcpy(out2, d, 16, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	HIP_TRY(e);
	return 0;
}

int snail_debug_dispatch_rate(int blocks, int threads, int reps, float *ms_per_launch) {
	if(blocks <= 0 || threads <= 0 || threads > 1024 || reps <= 0 || !ms_per_launch) { snail_set_error("snail_debug_dispatch_rate: bad arguments"); return 1; }
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
	hipLaunchKernelGGL(dev::k_nop, dim3(blocks), dim3(threads), 0, 0, (int *)nullptr);
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipEventRecord(e0, 0));
	for(int r = 0; r < reps; r++) hipLaunchKernelGGL(dev::k_nop, dim3(blocks), dim3(threads), 0, 0, (int *)nullptr);
	HIP_TRY(hipEventRecord(e1, 0));
	HIP_TRY(hipDeviceSynchronize());
	float ms = 0.0f;
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	*ms_per_launch = ms / reps;
	return 0;
}

// Experiment: `frames` full-frame primary launches back to back on ONE stream, with or without hipExtAnyOrderLaunch (which clears the
// AQL barrier bit so that consecutive dispatches of a queue may overlap -- hip_ext.h says "not supported on GFX9xx"; measured, not
// assumed).  No outputs are stored (null planes), nothing is deferred on a sane scene.  tools/anyorder.py.
int snail_debug_anyorder(SnailScene *s, const float cam[13], int resx, int resy, int frames, int flags, float *ms_total) {
	if(int rc = checkScene(s, "snail_debug_anyorder")) return rc;
	if(frames <= 0 || !ms_total || useDeep(s)) { snail_set_error("snail_debug_anyorder: bad arguments"); return 1; }
	DeviceGuard guard(s->device);
	dev::PrimaryArgs A;
	memset(&A, 0, sizeof(A));
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.nFrames = 1;
	A.g[0] = makeGen(cam, resx, resy);
	A.resx = resx; A.resy = resy; A.w = resx; A.h = resy;
	A.fastOK = s->fastOK && originSane(cam);
	A.pack = stackPack(s);
	A.pw = (resx + 15) / 16; A.ph = (resy + 15) / 16; A.nPackets = A.pw * A.ph;
	const int nRegions = ((A.pw + 3) / 4) * ((A.ph + 3) / 4);
	const int blocks = ((nRegions + 7) / 8) * 8 * 16;
	A.nBlocks = blocks; A.nSlots = blocks;
	if(blocks + 2 > s->deferCap) { snail_set_error("snail_debug_anyorder: trace one ordinary frame of this size first"); return 1; }
	A.defer = s->dDefer[0];
	hipStream_t st;
	HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	int relWhich = -1;
	if(A.pack && SNAIL_REL_NODES && SNAIL_NODE_PREFETCH) {
		if(int rc = relFor(s, cam, st, &A.rel[0], &relWhich)) return rc;
	}
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipEventRecord(e0, st));
	for(int f = 0; f < frames; f++)
		hipExtLaunchKernelGGL(dev::k_primary<false>, dim3(blocks / SNAIL_BLOCK_WAVES), dim3(64 * SNAIL_BLOCK_WAVES), 0, st, nullptr, nullptr, (unsigned)flags, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e1, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipEventElapsedTime(ms_total, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
	return 0;
}

int snail_debug_occupancy(int out[4]) {
	if(!out) { snail_set_error("snail_debug_occupancy: null output"); return 1; }
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	int perCU = 0, maxBlocks = 0, cus = 0;
	HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void *)dev::k_primary<false>, 64 * SNAIL_BLOCK_WAVES, 0));
	HIP_TRY(hipDeviceGetAttribute(&maxBlocks, hipDeviceAttributeMaxBlocksPerMultiProcessor, dev));
	HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
	out[0] = perCU; out[1] = maxBlocks; out[2] = cus; out[3] = SNAIL_BLOCK_WAVES;
	return 0;
}

#endif // SNAIL_DEBUG_API

int snail_account_packets(SnailScene *s, const float cam[13], int resx, int resy, uint32_t *out8) {
	if(int rc = checkScene(s, "snail_account_packets")) return rc;
	if(!out8) { snail_set_error("snail_account_packets: null output"); return 1; }
	DeviceGuard guard(s->device);
	const int np = ((resx + 15) / 16) * ((resy + 15) / 16);
	DevBuf c;
	if(c.upload(nullptr, (size_t)np * 32)) { snail_set_error("snail_account_packets: allocation failed"); return 2; }
	HIP_TRY(hipMemset(c.p, 0, (size_t)np * 32));
	int rc = launchPrimary(s, cam, resx, resy, 0, 0, resx, resy, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, (unsigned *)c.p);
	if(rc) return rc;
	HIP_TRY(hipDeviceSynchronize());
	if(c.download(out8, (size_t)np * 32)) { snail_set_error("snail_account_packets: download failed"); return 2; }
	return 0;
}

int snail_account_primary(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, uint64_t out[4]) {
	if(int rc = checkScene(s, "snail_account_primary")) return rc;
	if((x0 & 15) || (y0 & 15) || w <= 0 || h <= 0 || !out) { snail_set_error("snail_account_primary: bad rect"); return 1; }
	DeviceGuard guard(s->device);
	dev::AccountArgs A;
	A.nodes = s->dNodes; A.tris = s->dTris;
	A.g = makeGen(cam, resx, resy);
	A.x0 = x0; A.y0 = y0; A.pw = (w + 15) / 16; A.ph = (h + 15) / 16;
	A.out = (dev::u64 *)s->dStats;
	HIP_TRY(hipMemset(s->dStats, 0, 32));
	const int np = A.pw * A.ph;
	hipLaunchKernelGGL(dev::k_account, dim3((np + 3) / 4), dim3(256), 0, 0, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipDeviceSynchronize());
	unsigned long long hs[4];
	HIP_TRY(hipMemcpy(hs, s->dStats, 32, hipMemcpyDeviceToHost));
	for(int k = 0; k < 4; k++) out[k] += hs[k];
	return 0;
}

} // extern "C"

#include "render_host.inc"
