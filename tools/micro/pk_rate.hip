// Microbenchmark: issue cost of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32 on gfx950 (is packed fp32 full rate?).
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/pk_rate.hip -o gpurun_out/pk_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND> __global__ __launch_bounds__(64) void k(float *out, int iters) {
	float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001f;
	float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0}, q = {b, b};
	for(int i = 0; i < iters; i++) {
		if(KIND == 0) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
		if(KIND == 1) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }
		if(KIND == 2) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }
		if(KIND == 3) { REP8(asm volatile("v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }
		if(KIND == 4) { REP8(asm volatile("v_max3_f32 %0, %0, %4, %1\n v_min3_f32 %1, %1, %4, %2\n v_max3_f32 %2, %2, %4, %3\n v_min3_f32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
		if(KIND == 5) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_max3_f32 %5, %5, %6, %7\n v_pk_mul_f32 %1, %1, %4\n v_min3_f32 %6, %6, %7, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(a0), "v"(a1), "v"(a2));) }
	}
	out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND> void run(const char *name, float *d, int blocks) {
	const int iters = 20000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 100);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double instr = (double)iters * 32.0, wavesPerSimd = blocks / 1024.0; // 256 CUs x 4 SIMDs
	printf("%-46s %8.3f ms  -> %.2f ns per instruction per SIMD (%.1f waves/SIMD)\n", name, ms, ms * 1e6 / (instr * wavesPerSimd), wavesPerSimd);
}
int main() {
	float *d; hipMalloc(&d, 1 << 24);
	for(int blocks : {1024, 5120}) {
		run<0>("v_mul_f32", d, blocks);
		run<1>("v_pk_mul_f32", d, blocks);
		run<2>("v_pk_mul_f32 op_sel broadcast", d, blocks);
		run<3>("v_pk_add_f32 (neg)", d, blocks);
		run<4>("v_max3/min3_f32", d, blocks);
		run<5>("pk_mul + max3 interleaved", d, blocks);
	}
	return 0;
}
