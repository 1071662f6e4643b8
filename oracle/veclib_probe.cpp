// veclib_probe.cpp -- ORACLE INFRASTRUCTURE.  Our own driver around the reference's header-only veclib
// (compiled where it lies: -I/root/reference/veclib; nothing of it is copied here).
//
// veclib_probe            rows of four u32 bit patterns (a, b, c, d) on stdin; per row the bit patterns produced by the
//                         reference's SSE primitives that the hot path uses:
//   Inv(a) RSqrt(a) Min(a,b) Max(a,b) Condition(a<b,c,d) [lane 0 of f32x4]          veclib/sse/f32.h:98-119
//   scalar Inv(a) RSqrt(a) Min(a,b) Max(a,b)                                          veclib/vecbase.h:53-76
//   Vec3<float>(a,b,c) | Vec3<float>(b,c,d)   and the x component of their ^ (cross)  veclib/vec3.h:92-106
//
// veclib_probe exprs      rows of EIGHT u32 bit patterns: qa = f32x4(in[0..3]), qb = f32x4(in[4..7]); v1 = Vec3q(qa, qa<<<1, qa<<<2),
//                         v2 = Vec3q(qb, qb<<<1, qb<<<2) (<<< = lane rotation).  Per row 57 words -- the expressions of the shading path
//                         (src/scene_trace.cpp, src/shading/simple_material.h, src/render.cpp) written with the reference's own types and
//                         operators, which tests/test_oracle_pins.py compares with the oracle's restatement (orc_veclib_exprs):
//   [0]      ForWhich(qa < qb) | ForAny << 4 | ForAll << 5                            veclib/sse/f32.h:82-84
//   [1..4]   Sqrt(qa)            [5..8]   Abs(qa)                                     veclib/sse/f32.h:98,105
//   [9..12]  v1 | v2             [13..24] (v1 ^ v2).x, .y, .z                         veclib/vec3.h:92-106 on Vec3q
//   [25..36] v1 - v2 * (dot + dot), dot = v2 | v1  (Reflect, src/rtbase_math.h:54-58: ray = v1, nrm = v2) .x, .y, .z
//   [37..40] Condition(qa < qb, v1, v2).x                                              veclib/vec3.h Condition on Vec3<f32x4>
//   [41..44] Trunc(Clamp(qa * 255, 0, 255))  (one channel of ConvColor, src/render.cpp:11-17)
//   CPU-specific (rcpps), compared live only:
//   [45..48] FastInv(qa)         [49..52] Max(0, ((1 - a) * 0.2 + FastInv(16 * a * a)) - 0.0625), a = qa * qb  (src/scene_trace.cpp:585-587)
//   [53..56] Inv(qa + 1e-8)  (SafeInv, src/rtbase.h:117-120)
#include <veclib.h>
#include <cstdio>
#include <cstring>
using namespace veclib;

static float f(unsigned u) { float x; memcpy(&x, &u, 4); return x; }
static unsigned b(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
typedef Vec3<f32x4> Vec3q;

static void put4(f32x4 v) { for(int l = 0; l < 4; l++) printf(" %08x", b(v[l])); }

static int exprs() {
	unsigned u[8];
	while(scanf("%x %x %x %x %x %x %x %x", &u[0], &u[1], &u[2], &u[3], &u[4], &u[5], &u[6], &u[7]) == 8) {
		float a[4], c[4];
		for(int l = 0; l < 4; l++) { a[l] = f(u[l]); c[l] = f(u[4 + l]); }
		f32x4 qa(a[0], a[1], a[2], a[3]), qb(c[0], c[1], c[2], c[3]);
		Vec3q v1(qa, f32x4(a[1], a[2], a[3], a[0]), f32x4(a[2], a[3], a[0], a[1]));
		Vec3q v2(qb, f32x4(c[1], c[2], c[3], c[0]), f32x4(c[2], c[3], c[0], c[1]));
		f32x4b lt = qa < qb;
		printf("%08x", (unsigned)ForWhich(lt) | (ForAny(lt) ? 16u : 0u) | (ForAll(lt) ? 32u : 0u));
		put4(Sqrt(qa)); put4(Abs(qa));
		put4(v1 | v2);
		Vec3q cr = v1 ^ v2; put4(cr.x); put4(cr.y); put4(cr.z);
		f32x4 dot = v2 | v1;
		Vec3q rf = v1 - v2 * (dot + dot); put4(rf.x); put4(rf.y); put4(rf.z);
		Vec3q cd = Condition(lt, v1, v2); put4(cd.x);
		i32x4 tr = Trunc(Clamp(qa * 255.0f, f32x4(0.0f), f32x4(255.0f)));
		for(int l = 0; l < 4; l++) printf(" %08x", (unsigned)tr[l]);
		put4(FastInv(qa));
		f32x4 atten = qa * qb;
		atten = Max(f32x4(0.0f), ((f32x4(1.0f) - atten) * 0.2f + FastInv(f32x4(16.0f) * atten * atten)) - f32x4(0.0625f));
		put4(atten);
		put4(Inv(qa + f32x4(0.00000001f)));
		printf("\n");
	}
	return 0;
}

int main(int argc, char **argv) {
	if(argc > 1 && !strcmp(argv[1], "exprs")) return exprs();
	unsigned ua, ub, uc, ud;
	while(scanf("%x %x %x %x", &ua, &ub, &uc, &ud) == 4) {
		float a = f(ua), bb = f(ub), c = f(uc), d = f(ud);
		f32x4 qa(a), qb(bb), qc(c), qd(d);
		Vec3<float> v1(a, bb, c), v2(bb, c, d);
		Vec3<float> cr = v1 ^ v2;
		printf("%08x %08x %08x %08x %08x %08x %08x %08x %08x %08x %08x\n",
			b(Inv(qa)[0]), b(RSqrt(qa)[0]), b(Min(qa, qb)[0]), b(Max(qa, qb)[0]), b(Condition(qa < qb, qc, qd)[0]),
			b(Inv(a)), b(RSqrt(a)), b(Min(a, bb)), b(Max(a, bb)), b(v1 | v2), b(cr.x));
	}
	return 0;
}
