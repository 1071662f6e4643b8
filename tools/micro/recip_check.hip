// Exhaustive check (all 2^32 float bit patterns) of short reciprocal sequences against the correctly rounded 1.0f / x the kernels use
// (compiler's v_div_scale / v_rcp / v_fma / v_div_fmas / v_div_fixup expansion, -fhip-fp32-correctly-rounded-divide-sqrt):
//   variant 1: r = v_rcp_f32(x); e = fma(-x, r, 1); r = fma(e, r, r)
//   variant 2: variant 1 followed by a second step  e = fma(-x, r, 1); r = fma(e, r, r)
// Prints, per variant, the number of inputs whose result differs bitwise (NaN results compared as a class) and the exponent range of the
// mismatching inputs, so that the fast sequence can be guarded by a range test.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/micro/recip_check.hip -o gpurun_out/recip_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct Res {
	unsigned long long bad[2];      // mismatches per variant
	unsigned long long badExp[2][256]; // per biased exponent of the input
	unsigned firstBad[2][8];
};
template <int V> __device__ __forceinline__ float fastRecip(float x) {
	float r = __builtin_amdgcn_rcpf(x);
	float e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	if(V == 2) {
		e = __builtin_fmaf(-x, r, 1.0f);
		r = __builtin_fmaf(e, r, r);
	}
	return r;
}
__global__ __launch_bounds__(256) void k(Res *res, unsigned base) {
	const unsigned bits = base + blockIdx.x * 256u + threadIdx.x;
	const float x = __uint_as_float(bits);
	const float ref = 1.0f / x;
	const float a[2] = {fastRecip<1>(x), fastRecip<2>(x)};
	for(int v = 0; v < 2; v++) {
		const bool same = (ref != ref) ? (a[v] != a[v]) : __float_as_uint(ref) == __float_as_uint(a[v]);
		if(!same) {
			unsigned long long n = atomicAdd(&res->bad[v], 1ull);
			atomicAdd(&res->badExp[v][(bits >> 23) & 255], 1ull);
			if(n < 8) res->firstBad[v][n] = bits;
		}
	}
}
int main() {
	Res *d; hipMalloc(&d, sizeof(Res)); hipMemset(d, 0, sizeof(Res));
	for(unsigned part = 0; part < 256; part++) hipLaunchKernelGGL(k, dim3(1u << 16), dim3(256), 0, 0, d, part << 24);
	if(hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
	Res h; hipMemcpy(&h, d, sizeof(Res), hipMemcpyDeviceToHost);
	for(int v = 0; v < 2; v++) {
		printf("variant %d: %llu of 4294967296 inputs differ from 1.0f / x\n", v + 1, h.bad[v]);
		int lo = -1, hi = -1;
		for(int e = 0; e < 256; e++) if(h.badExp[v][e]) { if(lo < 0) lo = e; hi = e; }
		printf("  biased input exponents with a mismatch:");
		for(int e = 0; e < 256; e++) if(h.badExp[v][e]) printf(" %d:%llu", e, h.badExp[v][e]);
		printf("\n  range [%d, %d]; first:", lo, hi);
		for(int i = 0; i < 8 && i < (int)h.bad[v]; i++) printf(" %08x", h.firstBad[v][i]);
		printf("\n");
	}
	return 0;
}
