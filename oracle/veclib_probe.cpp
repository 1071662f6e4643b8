// veclib_probe.cpp -- ORACLE INFRASTRUCTURE.  Our own driver around the reference's header-only veclib
// (compiled where it lies: -I/root/reference/veclib; nothing of it is copied here).  Reads rows of four
// u32 bit patterns (a, b, c, d) from stdin and prints, for each row, the bit patterns produced by the
// reference's SSE primitives that the hot path uses:
//   Inv(a) RSqrt(a) Min(a,b) Max(a,b) Condition(a<b,c,d) [lane 0 of f32x4]          veclib/sse/f32.h:98-119
//   scalar Inv(a) RSqrt(a) Min(a,b) Max(a,b)                                          veclib/vecbase.h:53-76
//   Vec3<float>(a,b,c) | Vec3<float>(b,c,d)   and the x component of their ^ (cross)  veclib/vec3.h:92-106
#include <veclib.h>
#include <cstdio>
#include <cstring>
using namespace veclib;

static float f(unsigned u) { float x; memcpy(&x, &u, 4); return x; }
static unsigned b(float x) { unsigned u; memcpy(&u, &x, 4); return u; }

int main() {
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
