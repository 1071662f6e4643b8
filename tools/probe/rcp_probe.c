/* Enumerates the host CPU's rcpps / rsqrtps (the instructions behind veclib's SSE Inv / RSqrt,
 * /root/reference/veclib/sse/base.h:84-92) and reports their structure: how many mantissa segments of constant output there
 * are, whether the segments are aligned power-of-two blocks, how exponents scale, and what the special inputs give.
 * Build: gcc -O2 -msse2 rcp_probe.c -o rcp_probe ; run: ./rcp_probe [dump-prefix] */
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t rcp_u(uint32_t x) { return f2u(_mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(u2f(x))))); }
static inline uint32_t rsq_u(uint32_t x) { return f2u(_mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(u2f(x))))); }

typedef uint32_t (*fn_t)(uint32_t);

static void analyse(const char *name, fn_t f, uint32_t expBits, const char *dump) {
	uint32_t n = 1u << 23, segs = 0, prev = 0, runStart = 0, minRun = n, maxRun = 0;
	uint32_t lowOr = 0;      /* OR of the output's mantissa bits: which bits are ever set */
	int alignedPow = 23;     /* largest k such that every segment boundary is a multiple of 2^k */
	uint32_t *starts = malloc(sizeof(uint32_t) * n), *vals = malloc(sizeof(uint32_t) * n);
	for(uint32_t m = 0; m < n; m++) {
		uint32_t y = f(expBits | m);
		lowOr |= y;
		if(m == 0 || y != prev) {
			if(m) { uint32_t run = m - runStart; if(run < minRun) minRun = run; if(run > maxRun) maxRun = run;
				while(alignedPow > 0 && (m & ((1u << alignedPow) - 1))) alignedPow--; }
			starts[segs] = m; vals[segs] = y; segs++; runStart = m; prev = y;
		}
	}
	{ uint32_t run = n - runStart; if(run < minRun) minRun = run; if(run > maxRun) maxRun = run; }
	int tz = 0; while(tz < 23 && !((lowOr >> tz) & 1)) tz++;
	printf("%s exp=%08x: segments %u, run min %u max %u, boundaries aligned to 2^%d, output low zero bits %d, first %08x last %08x\n",
		name, expBits, segs, minRun, maxRun, alignedPow, tz, vals[0], vals[segs - 1]);
	/* monotone? */
	int mono = 1; for(uint32_t i = 1; i < segs; i++) if(vals[i] >= vals[i - 1]) mono = 0;
	printf("  strictly decreasing over segments: %d\n", mono);
	/* differences between consecutive outputs */
	uint32_t dmin = ~0u, dmax = 0; for(uint32_t i = 1; i < segs; i++) { uint32_t d = vals[i - 1] - vals[i]; if(d < dmin) dmin = d; if(d > dmax) dmax = d; }
	printf("  output step between segments: min %u max %u (ulps of the output)\n", dmin, dmax);
	if(dump) {
		char path[512]; snprintf(path, sizeof path, "%s_%s_%08x.bin", dump, name, expBits);
		FILE *fp = fopen(path, "wb");
		if(fp) { fwrite(&segs, 4, 1, fp); fwrite(starts, 4, segs, fp); fwrite(vals, 4, segs, fp); fclose(fp); printf("  wrote %s (%u segments)\n", path, segs); }
	}
	free(starts); free(vals);
}

int main(int argc, char **argv) {
	const char *dump = argc > 1 ? argv[1] : NULL;
	FILE *ci = fopen("/proc/cpuinfo", "r");
	if(ci) { char line[512]; while(fgets(line, sizeof line, ci)) if(!strncmp(line, "model name", 10)) { printf("%s", line); break; } fclose(ci); }
	analyse("rcp", rcp_u, 0x3f800000u, dump);      /* [1, 2) */
	analyse("rsq", rsq_u, 0x3f800000u, dump);      /* [1, 2): even exponent */
	analyse("rsq", rsq_u, 0x40000000u, dump);      /* [2, 4): odd exponent */

	/* exponent scaling: rcp(m 2^e) == rcp(m) with the exponent moved, for every e, on a mantissa sample; list the exponents where not */
	printf("rcp exponent scaling exceptions (biased exponent of x: count of 4096 sampled mantissas that differ):\n");
	for(uint32_t e = 0; e < 256; e++) {
		uint32_t bad = 0, ex = 0;
		for(uint32_t k = 0; k < 4096; k++) {
			uint32_t m = (k * 2049u + 1u) & 0x7fffffu;
			uint32_t base = rcp_u(0x3f800000u | m);          /* in (0.5, 1]: exponent 126 or 127 */
			int32_t be = (int32_t)(base >> 23) - 127;         /* -1 or 0 */
			int32_t oe = be - ((int32_t)e - 127) + 127;
			uint32_t want = (oe <= 0 || oe >= 255) ? 0xffffffffu : ((uint32_t)oe << 23) | (base & 0x7fffffu);
			uint32_t got = rcp_u((e << 23) | m);
			if(got != want) { bad++; ex = got; }
		}
		if(bad) printf("  e=%3u: %u differ (e.g. got %08x for m sample)\n", e, bad, ex);
	}
	printf("rsq exponent scaling exceptions:\n");
	for(uint32_t e = 0; e < 256; e++) {
		uint32_t bad = 0, ex = 0;
		for(uint32_t k = 0; k < 4096; k++) {
			uint32_t m = (k * 2049u + 1u) & 0x7fffffu;
			uint32_t par = (e & 1) ? 0x3f800000u : 0x40000000u;   /* e odd: unbiased even -> [1,2); e even: unbiased odd -> [2,4) */
			uint32_t base = rsq_u(par | m);
			int32_t be = (int32_t)(base >> 23) - 127;
			int32_t ue = (int32_t)e - 127;                          /* unbiased exponent of x */
			int32_t half = (e & 1) ? ue / 2 : (ue - 1) / 2;         /* x = m 2^ue = (m or 2m) 2^(2 half) */
			if(!(e & 1) && ue < 0) half = (ue - 1) / 2;
			int32_t oe = be - half + 127;
			uint32_t want = (oe <= 0 || oe >= 255) ? 0xffffffffu : ((uint32_t)oe << 23) | (base & 0x7fffffu);
			uint32_t got = rsq_u((e << 23) | m);
			if(got != want) { bad++; ex = got; }
		}
		if(bad) printf("  e=%3u: %u differ (e.g. got %08x)\n", e, bad, ex);
	}
	/* sign and specials */
	uint32_t sp[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x007fffffu, 0x80000001u, 0x00800000u, 0x7f7fffffu, 0x7f800000u, 0xff800000u,
		0x7fc00000u, 0x7f800001u, 0xffc00000u, 0x7e800000u, 0x7e800001u, 0x7effffffu, 0x7f000000u, 0x7f000001u, 0x7e7fffffu, 0xbf800000u, 0xc0000000u};
	for(unsigned i = 0; i < sizeof sp / sizeof sp[0]; i++)
		printf("special x=%08x rcp=%08x rsq=%08x\n", sp[i], rcp_u(sp[i]), rsq_u(sp[i]));
	uint32_t sbad = 0; for(uint32_t k = 0; k < (1u << 20); k++) { uint32_t x = 0x3f800000u | ((k * 8u + 3u) & 0x7fffffu); if(rcp_u(x | 0x80000000u) != (rcp_u(x) | 0x80000000u)) sbad++; }
	printf("rcp sign symmetry failures: %u\n", sbad);
	return 0;
}
