// host_sse.h -- the host CPU's rcpps / rsqrtps as data (internal to libsnailhip.so; the C-ABI side is include/snail_hip.h, "arithmetic").
//
// The reference's SSE build normalises primary directions with RSqrt = rsqrtps + one Newton step and inverts with Inv = rcpps + one
// Newton step (veclib/sse/base.h:84-92, veclib/sse/f32.h:98-102; callers src/ray_generator.cpp:41-44, src/rtbase.h:117-120,
// src/triangle.cpp:55, src/scene_trace.cpp:128-137,:538-558,:585-587).  rcpps / rsqrtps are table look-ups whose tables differ between
// CPU vendors and generations, so "what the reference computes" is defined by the CPU it runs on.  Every x86 CPU measured so far
// (tools/probe/rcp_probe.c: Intel Xeon / Skylake-SP: 2048 + 2 x 1024 segments; AMD EPYC 9575F / Zen 5: 3382 + 2387 + 1693 segments) returns a
// result that depends on the sign, the exponent and the top 12 mantissa bits of the input only, with 12 significant mantissa bits:
//     rcpps(+-1.m x 2^e)  = +-T_rcp[m >> 11] x 2^-e,              results below 2^-126 flushed to +-0, denormal inputs read as +-0 -> +-inf
//     rsqrtps(1.m x 2^e)  = T_rsq[e odd][m >> 11] x 2^-floor(e/2), negative inputs -> the default NaN (0xffc00000), -0 / -denormal -> -inf
// The tables (3 x 4096 words) are taken from the CPU the library runs on, at first use, and the block structure is verified over all 2^23
// mantissas before anything relies on it (hostSseTables).  The same inline functions run on the host (verification against the
// instructions over all 2^32 inputs: snail_host_sse_check) and on the device (dev_sse::rcpHost / rsqrtHost in snail_dev.inc).
#pragma once

#if defined(__HIPCC__)
#define SNAIL_HD __host__ __device__
#else
#define SNAIL_HD
#endif

enum { kHostSseShift = 11, kHostSseEntries = 1 << (23 - kHostSseShift) };   // 4096 entries per table; T_rcp, T_rsq[even exponent], T_rsq[odd exponent]

// rcpps of the float with bit pattern b; tab = T_rcp: the bits of rcpps(1.m), in (0.5, 1]
SNAIL_HD inline unsigned sseRcpBits(const unsigned *tab, unsigned b) {
	const unsigned s = b & 0x80000000u, e = (b >> 23) & 255u, m = b & 0x7fffffu;
	if(e == 255u) return m ? (b | 0x00400000u) : s;   // NaN: quieted, payload kept; +-inf -> +-0
	if(e == 0u) return s | 0x7f800000u;               // +-0, and denormals (read as zero) -> +-inf
	const unsigned base = tab[m >> kHostSseShift];
	const int oe = (int)(base >> 23) + 127 - (int)e;
	return oe <= 0 ? s : (s | ((unsigned)oe << 23) | (base & 0x7fffffu));   // no denormal results: flushed to +-0
}
// rsqrtps; tab = T_rsq: [0..4095] the bits of rsqrtps(1.m) for [1, 2), [4096..8191] those of rsqrtps(2 x 1.m) for [2, 4)
SNAIL_HD inline unsigned sseRsqrtBits(const unsigned *tab, unsigned b) {
	const unsigned s = b & 0x80000000u, e = (b >> 23) & 255u, m = b & 0x7fffffu;
	if(e == 255u && m) return b | 0x00400000u;        // NaN: quieted
	if(e == 0u) return s | 0x7f800000u;               // +-0 and +-denormals -> +-inf
	if(s) return 0xffc00000u;                         // negative, -inf included: the default NaN
	if(e == 255u) return 0u;                          // +inf -> +0
	const unsigned odd = (e & 1u) ^ 1u;               // unbiased exponent odd: x = (2 x 1.m) x 4^k
	const unsigned base = tab[odd * (unsigned)kHostSseEntries + (m >> kHostSseShift)];
	const int half = ((int)e - 127 - (int)odd) >> 1;  // k (an even number halved: exact for either sign)
	return ((unsigned)((int)(base >> 23) - half) << 23) | (base & 0x7fffffu);
}

// The tables in force, COPIED OUT under the lock together with their generation (out: 3 x 4096 words, may be null; generation: may be null): this
// CPU's own (taken at first use; returns 2 -- with the reason in *why -- when its rcpps / rsqrtps do not have the block structure above: results that
// change inside an aligned block of 2^11 mantissas, or an exponent / special-case rule that differs; or when this is not an x86 host) unless
// hostSseSetTables gave others.  The generation changes with every hostSseSetTables call: a device copy made from one snapshot is current for as long
// as the generation it was made at is.
int hostSseSnapshot(unsigned *out, unsigned *generation, const char **why);
// Tables of ANOTHER CPU instead (tab = 3 x 4096 words as hostSseSnapshot returns them; nullptr = this CPU's again): every machine of a render farm
// then computes what the reference computes on the CPU the tables came from, whatever its own CPU is.  Returns non-zero (and changes nothing)
// for tables that are not tables of reciprocals.
int hostSseSetTables(const unsigned *tab, const char **why);
// Compare the emulation with the instruction over the inputs [first, first + count) on `threads` host threads; fn 0 = rcpps, 1 = rsqrtps.
// Returns the number of inputs whose result bits differ (NaN results compare by their bits too: x86 returns the quieted input).
unsigned long long hostSseMismatches(int fn, unsigned long long first, unsigned long long count, int threads, unsigned *firstBad);
// per-chunk checksums of the INSTRUCTION's results: chunk c = inputs [c << 16, (c + 1) << 16); sum over the chunk of mix(x, f(x)) (hostSseMix)
void hostSseChunkSums(int fn, unsigned firstChunk, unsigned nChunks, int threads, unsigned long long *sums);
SNAIL_HD inline unsigned long long hostSseMix(unsigned x, unsigned y) { return ((unsigned long long)y * 0x9e3779b97f4a7c15ull) ^ ((unsigned long long)x * 0xc2b2ae3d27d4eb4full); }
