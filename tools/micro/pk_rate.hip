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
template <int KIND> double run(float *d, int blocks, int iters) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1);
	return ms * 1e6 / ((double)iters * 32.0 * (blocks / 1024.0)); // ns per instruction per SIMD (256 CUs x 4 SIMDs)
}
int main() {
	float *d; hipMalloc(&d, 1 << 24);
	// the shader clock ramps up over the first second of load (105 MHz idle -> 2.4 GHz): warm up, then take the minimum of rounds
	for(int i = 0; i < 40; i++) run<0>(d, 5120, 20000);
	const char *names[6] = {"v_mul_f32", "v_pk_mul_f32", "v_pk_mul_f32 op_sel broadcast", "v_pk_add_f32 (neg)", "v_max3/min3_f32", "pk_mul + max3 interleaved"};
	for(int blocks : {1024, 5120}) {
		double best[6] = {1e9, 1e9, 1e9, 1e9, 1e9, 1e9};
		for(int round = 0; round < 5; round++) {
			double r[6] = {run<0>(d, blocks, 20000), run<1>(d, blocks, 20000), run<2>(d, blocks, 20000), run<3>(d, blocks, 20000), run<4>(d, blocks, 20000), run<5>(d, blocks, 20000)};
			for(int k = 0; k < 6; k++) best[k] = r[k] < best[k] ? r[k] : best[k];
		}
		for(int k = 0; k < 6; k++) printf("%-34s %.3f ns per instruction per SIMD = %.2f cycles at 2.4 GHz (%.0f waves/SIMD)\n", names[k], best[k], best[k] * 2.4, blocks / 1024.0);
	}
	return 0;
}
