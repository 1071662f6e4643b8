// host_sse.cpp -- takes the rcpps / rsqrtps tables from the CPU this library runs on and proves that they describe it (host_sse.h).
// Plain host C++ (built without -march flags: _mm_rcp_ps / _mm_rsqrt_ps must stay the SSE instructions the reference's veclib issues,
// not AVX-512's vrcp14ps, whose tables differ).
#include "host_sse.h"

#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__) || defined(__i386__)
#include <xmmintrin.h>
#define SNAIL_HAVE_SSE 1
#else
#define SNAIL_HAVE_SSE 0   // not an x86 host: no rcpps / rsqrtps to take tables from -- SNAIL_ARITH_HOST_SSE then needs GIVEN tables (snail_arith_set_tables)
#endif

namespace {

inline unsigned bitsOf(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float floatOf(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
// the instructions exactly as veclib reaches them: _mm_rcp_ps / _mm_rsqrt_ps on a broadcast value (veclib/sse/base.h:84-92)
#if SNAIL_HAVE_SSE
inline unsigned rcpInsn(unsigned x) { return bitsOf(_mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(floatOf(x))))); }
inline unsigned rsqrtInsn(unsigned x) { return bitsOf(_mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(floatOf(x))))); }
#else
inline unsigned rcpInsn(unsigned) { return 0; }
inline unsigned rsqrtInsn(unsigned) { return 0; }
#endif

struct Tables {
	unsigned tab[3 * kHostSseEntries];
	bool ok = false;
	char why[256] = "";
};
Tables g_tables;
std::once_flag g_once;
// tables of ANOTHER CPU, given by the caller (hostSseSetTables): what hostSseTables() returns while set
Tables g_given;
std::mutex g_givenMu;
bool g_haveGiven = false;
unsigned g_generation = 1;

void buildTables() {
	Tables &T = g_tables;
	if(!SNAIL_HAVE_SSE) { snprintf(T.why, sizeof T.why, "this host is not an x86 CPU: it has no rcpps / rsqrtps (give the tables of the CPU to reproduce: snail_arith_set_tables)"); return; }
	const unsigned base[3] = {0x3f800000u, 0x3f800000u, 0x40000000u};   // [1, 2) for rcpps and the even-exponent rsqrtps table, [2, 4) for the odd one
	for(int f = 0; f < 3; f++) {
		unsigned *tab = T.tab + f * kHostSseEntries;
		for(unsigned i = 0; i < (unsigned)kHostSseEntries; i++) tab[i] = f == 0 ? rcpInsn(base[f] | (i << kHostSseShift)) : rsqrtInsn(base[f] | (i << kHostSseShift));
		// every mantissa of the block gives the block's value
		for(unsigned m = 0; m < (1u << 23); m++) {
			const unsigned y = f == 0 ? rcpInsn(base[f] | m) : rsqrtInsn(base[f] | m);
			if(y != tab[m >> kHostSseShift]) {
				snprintf(T.why, sizeof T.why, "%s of this CPU changes inside an aligned block of 2^%d mantissas (input %08x gives %08x, the block's first input %08x): not reproducible from a %d-entry table",
						 f == 0 ? "rcpps" : "rsqrtps", kHostSseShift, base[f] | m, y, tab[m >> kHostSseShift], kHostSseEntries);
				return;
			}
		}
		for(unsigned i = 0; i < (unsigned)kHostSseEntries; i++) {
			const unsigned e = tab[i] >> 23;   // sign clear, exponent 126 (or 127 for an exact 1.0)
			if(e != 126u && e != 127u) { snprintf(T.why, sizeof T.why, "unexpected table value %08x", tab[i]); return; }
		}
	}
	// exponent scaling, flushing and the special inputs: both ends of every block at every exponent and sign
	for(unsigned hi = 0; hi < 512u; hi++)
		for(unsigned i = 0; i < (unsigned)kHostSseEntries; i++)
			for(unsigned lo = 0; lo < 2u; lo++) {
				const unsigned x = (hi << 23) | (i << kHostSseShift) | (lo ? (1u << kHostSseShift) - 1u : 0u);
				const unsigned a = rcpInsn(x), b = sseRcpBits(T.tab, x), c = rsqrtInsn(x), d = sseRsqrtBits(T.tab + kHostSseEntries, x);
				if(a != b || c != d) {
					snprintf(T.why, sizeof T.why, "this CPU's %s(%08x) = %08x, the table rule gives %08x", a != b ? "rcpps" : "rsqrtps", x, a != b ? a : c, a != b ? b : d);
					return;
				}
			}
	T.ok = true;
}

template <class F> void parallelFor(unsigned long long n, int threads, F fn) {
	if(threads < 1) threads = 1;
	std::vector<std::thread> pool;
	const unsigned long long per = (n + (unsigned long long)threads - 1) / (unsigned long long)threads;
	for(int t = 0; t < threads; t++) {
		const unsigned long long a = per * (unsigned long long)t, b = a + per < n ? a + per : n;
		if(a >= b) break;
		pool.emplace_back([=] { fn(t, a, b); });
	}
	for(auto &th : pool) th.join();
}

} // namespace

// (this CPU's own tables never change once built: the pointer may be read without the lock; GIVEN tables are only ever handed out as a copy)
static const unsigned *ownTables(const char **why) {
	std::call_once(g_once, buildTables);
	if(why) *why = g_tables.why;
	return g_tables.ok ? g_tables.tab : nullptr;
}

int hostSseSnapshot(unsigned *out, unsigned *generation, const char **why) {
	if(why) *why = "";
	{
		std::lock_guard<std::mutex> lock(g_givenMu);
		if(g_haveGiven) {
			if(out) memcpy(out, g_given.tab, sizeof g_given.tab);
			if(generation) *generation = g_generation;
			return 0;
		}
	}
	const unsigned *own = ownTables(why);
	if(!own) return 2;
	std::lock_guard<std::mutex> lock(g_givenMu);
	if(g_haveGiven) {   // (given between the two looks: the given ones are in force)
		if(out) memcpy(out, g_given.tab, sizeof g_given.tab);
	} else if(out) memcpy(out, own, sizeof g_tables.tab);
	if(generation) *generation = g_generation;
	return 0;
}


int hostSseSetTables(const unsigned *tab, const char **why) {
	std::lock_guard<std::mutex> lock(g_givenMu);
	if(why) *why = "";
	if(tab) {
		// what the rule of host_sse.h assumes of a table: every entry in (0.5, 1] (sign clear, biased exponent 126, or 127 with a zero mantissa for an
		// exact 1.0) and non-increasing with the index (1 / x and 1 / sqrt(x) fall)
		for(int f = 0; f < 3; f++)
			for(int i = 0; i < kHostSseEntries; i++) {
				const unsigned v = tab[f * kHostSseEntries + i], e = v >> 23;
				if(!(e == 126u || v == 0x3f800000u) || (i && v > tab[f * kHostSseEntries + i - 1])) {
					if(why) *why = "entries must lie in (0.5, 1] and must not increase with the index";
					return 1;
				}
			}
		memcpy(g_given.tab, tab, sizeof g_given.tab);
		g_haveGiven = true;
	} else g_haveGiven = false;
	g_generation++;
	return 0;
}

unsigned long long hostSseMismatches(int fn, unsigned long long first, unsigned long long count, int threads, unsigned *firstBad) {
	std::vector<unsigned> snap((size_t)3 * kHostSseEntries);
	if(hostSseSnapshot(snap.data(), nullptr, nullptr)) return ~0ull;
	const unsigned *tab = snap.data();
	std::vector<unsigned long long> bad((size_t)(threads < 1 ? 1 : threads), 0ull);
	std::vector<unsigned long long> where(bad.size(), ~0ull);
	parallelFor(count, threads, [&](int t, unsigned long long a, unsigned long long b) {
		unsigned long long n = 0, w = ~0ull;
		for(unsigned long long i = a; i < b; i++) {
			const unsigned x = (unsigned)(first + i);
			const bool same = fn == 0 ? rcpInsn(x) == sseRcpBits(tab, x) : rsqrtInsn(x) == sseRsqrtBits(tab + kHostSseEntries, x);
			if(!same) { if(!n) w = first + i; n++; }
		}
		bad[(size_t)t] = n; where[(size_t)t] = w;
	});
	unsigned long long total = 0, w = ~0ull;
	for(size_t t = 0; t < bad.size(); t++) { total += bad[t]; if(where[t] < w) w = where[t]; }
	if(firstBad) *firstBad = (unsigned)w;
	return total;
}

void hostSseChunkSums(int fn, unsigned firstChunk, unsigned nChunks, int threads, unsigned long long *sums) {
	parallelFor(nChunks, threads, [&](int, unsigned long long a, unsigned long long b) {
		for(unsigned long long c = a; c < b; c++) {
			const unsigned x0 = (firstChunk + (unsigned)c) << 16;
			unsigned long long s = 0;
			for(unsigned k = 0; k < 65536u; k++) { const unsigned x = x0 + k; s += hostSseMix(x, fn == 0 ? rcpInsn(x) : rsqrtInsn(x)); }
			sums[c] = s;
		}
	});
}
