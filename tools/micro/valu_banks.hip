// Microbenchmark: does the cost of a three-source VALU instruction (v_max3_f32 / v_min3_f32: 1.6x a two-source one, tools/micro/pk_rate.hip)
// depend on WHICH registers it reads -- VGPR banks (register number mod 4), an SGPR or a constant among the sources?
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/valu_banks.hip -o gpurun_out/valu_banks ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define CLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "s40", "s41", "s42", "s43"
template <int KIND> __global__ __launch_bounds__(64) void k(float *out, int iters) {
	asm volatile("v_cvt_f32_u32 v4, %0\n v_add_f32 v5, 1.0, v4\n v_add_f32 v6, 2.0, v4\n v_add_f32 v7, 4.0, v4\n"
				 "v_add_f32 v8, 0.5, v4\n v_add_f32 v9, 0.5, v5\n v_add_f32 v10, 0.5, v6\n v_add_f32 v11, 0.5, v7\n"
				 "v_add_f32 v12, 0.5, v8\n v_add_f32 v13, 0.5, v9\n v_add_f32 v14, 0.5, v10\n v_add_f32 v15, 0.5, v11\n"
				 "v_add_f32 v16, 0.5, v12\n v_add_f32 v17, 0.5, v13\n v_add_f32 v18, 0.5, v14\n v_add_f32 v19, 0.5, v15\n"
				 "s_mov_b32 s40, 1.0\n s_mov_b32 s41, 2.0\n s_mov_b32 s42, 4.0\n s_mov_b32 s43, 0.5\n" ::"v"(threadIdx.x) : CLOB);
	for(int i = 0; i < iters; i++) {
		// 0: two sources, distinct banks, VOP2
		if(KIND == 0) { REP8(asm volatile("v_max_f32 v4, v4, v5\n v_max_f32 v8, v8, v9\n v_max_f32 v12, v12, v13\n v_max_f32 v16, v16, v17" ::: CLOB);) }
		// 1: two sources, same bank
		if(KIND == 1) { REP8(asm volatile("v_max_f32 v4, v4, v8\n v_max_f32 v5, v5, v9\n v_max_f32 v6, v6, v10\n v_max_f32 v7, v7, v11" ::: CLOB);) }
		// 2: three sources, three banks
		if(KIND == 2) { REP8(asm volatile("v_max3_f32 v4, v4, v5, v6\n v_min3_f32 v8, v8, v9, v10\n v_max3_f32 v12, v12, v13, v14\n v_min3_f32 v16, v16, v17, v18" ::: CLOB);) }
		// 3: three sources, one bank
		if(KIND == 3) { REP8(asm volatile("v_max3_f32 v4, v4, v8, v12\n v_min3_f32 v5, v5, v9, v13\n v_max3_f32 v6, v6, v10, v14\n v_min3_f32 v7, v7, v11, v15" ::: CLOB);) }
		// 4: three sources, two of them in one bank
		if(KIND == 4) { REP8(asm volatile("v_max3_f32 v4, v4, v8, v5\n v_min3_f32 v9, v9, v13, v6\n v_max3_f32 v14, v14, v10, v7\n v_min3_f32 v15, v15, v11, v12" ::: CLOB);) }
		// 5: three sources, one an inline constant
		if(KIND == 5) { REP8(asm volatile("v_max3_f32 v4, 0, v4, v5\n v_min3_f32 v8, 0, v8, v9\n v_max3_f32 v12, 0, v12, v13\n v_min3_f32 v16, 0, v16, v17" ::: CLOB);) }
		// 6: three sources, one an SGPR
		if(KIND == 6) { REP8(asm volatile("v_max3_f32 v4, s40, v4, v5\n v_min3_f32 v8, s41, v8, v9\n v_max3_f32 v12, s42, v12, v13\n v_min3_f32 v16, s43, v16, v17" ::: CLOB);) }
		// 7: VOP2 multiply with an SGPR source (the slab products of the camera-relative loop)
		if(KIND == 7) { REP8(asm volatile("v_mul_f32 v4, s40, v5\n v_mul_f32 v8, s41, v9\n v_mul_f32 v12, s42, v13\n v_mul_f32 v16, s43, v17" ::: CLOB);) }
		// 8: two-source VOP3 encoding (e64) of the same multiply
		if(KIND == 8) { REP8(asm volatile("v_mul_f32_e64 v4, v5, s40\n v_mul_f32_e64 v8, v9, s41\n v_mul_f32_e64 v12, v13, s42\n v_mul_f32_e64 v16, v17, s43" ::: CLOB);) }
		// 9: v_fma_f32 (three sources), three banks
		if(KIND == 9) { REP8(asm volatile("v_fma_f32 v4, v4, v5, v6\n v_fma_f32 v8, v8, v9, v10\n v_fma_f32 v12, v12, v13, v14\n v_fma_f32 v16, v16, v17, v18" ::: CLOB);) }
		// 10: dependent chain of two-source instructions (each reads the previous result)
		if(KIND == 10) { REP8(asm volatile("v_max_f32 v4, v4, v5\n v_max_f32 v4, v4, v6\n v_max_f32 v4, v4, v7\n v_max_f32 v4, v4, v9" ::: CLOB);) }
		// 11: v_max3 whose result feeds the next instruction (the slab test's pattern: max3 -> max -> sub)
		if(KIND == 11) { REP8(asm volatile("v_max3_f32 v4, v4, v5, v6\n v_max_f32 v4, 0, v4\n v_min3_f32 v8, v8, v9, v10\n v_sub_f32 v12, v8, v4" ::: CLOB);) }
		// 12: v_mul_f32 VGPR x VGPR (VOP2)
		if(KIND == 12) { REP8(asm volatile("v_mul_f32 v4, v4, v5\n v_mul_f32 v8, v8, v9\n v_mul_f32 v12, v12, v13\n v_mul_f32 v16, v16, v17" ::: CLOB);) }
		// 13: v_mul_f32 inline constant x VGPR
		if(KIND == 13) { REP8(asm volatile("v_mul_f32 v4, 1.0, v4\n v_mul_f32 v8, 1.0, v8\n v_mul_f32 v12, 1.0, v12\n v_mul_f32 v16, 1.0, v16" ::: CLOB);) }
		// 14: v_sub_f32 SGPR - VGPR
		if(KIND == 14) { REP8(asm volatile("v_sub_f32 v4, s40, v5\n v_sub_f32 v8, s41, v9\n v_sub_f32 v12, s42, v13\n v_sub_f32 v16, s43, v17" ::: CLOB);) }
		// 15: v_add_f32 VGPR + VGPR
		if(KIND == 15) { REP8(asm volatile("v_add_f32 v4, v4, v5\n v_add_f32 v8, v8, v9\n v_add_f32 v12, v12, v13\n v_add_f32 v16, v16, v17" ::: CLOB);) }
		// 16: v_cmp_le_f32 (VOPC -> vcc)
		if(KIND == 16) { REP8(asm volatile("v_cmp_le_f32 vcc, v4, v5\n v_cmp_le_f32 vcc, v8, v9\n v_cmp_le_f32 vcc, v12, v13\n v_cmp_le_f32 vcc, v16, v17" ::: CLOB, "vcc");) }
		// 17: v_mov_b32 from an SGPR
		if(KIND == 17) { REP8(asm volatile("v_mov_b32 v4, s40\n v_mov_b32 v8, s41\n v_mov_b32 v12, s42\n v_mov_b32 v16, s43" ::: CLOB);) }
		// 18: v_mov_b32 VGPR
		if(KIND == 18) { REP8(asm volatile("v_mov_b32 v4, v5\n v_mov_b32 v8, v9\n v_mov_b32 v12, v13\n v_mov_b32 v16, v17" ::: CLOB);) }
		// 19: fast and slow alternating (v_mul VGPR, v_max)
		if(KIND == 19) { REP8(asm volatile("v_mul_f32 v4, v4, v5\n v_max_f32 v8, v8, v9\n v_mul_f32 v12, v12, v13\n v_max_f32 v16, v16, v17" ::: CLOB);) }
		// 20: v_fma_f32 with an SGPR source
		if(KIND == 20) { REP8(asm volatile("v_fma_f32 v4, s40, v5, v6\n v_fma_f32 v8, s41, v9, v10\n v_fma_f32 v12, s42, v13, v14\n v_fma_f32 v16, s43, v17, v18" ::: CLOB);) }
		// 21: v_pk_mul_f32 with an SGPR pair source
		if(KIND == 21) { REP8(asm volatile("v_pk_mul_f32 v[4:5], s[40:41], v[4:5]\n v_pk_mul_f32 v[8:9], s[42:43], v[8:9]\n v_pk_mul_f32 v[12:13], s[40:41], v[12:13]\n v_pk_mul_f32 v[16:17], s[42:43], v[16:17]" ::: CLOB);) }
		// 22: v_pk_mul_f32 VGPR pairs
		if(KIND == 22) { REP8(asm volatile("v_pk_mul_f32 v[4:5], v[6:7], v[4:5]\n v_pk_mul_f32 v[8:9], v[10:11], v[8:9]\n v_pk_mul_f32 v[12:13], v[14:15], v[12:13]\n v_pk_mul_f32 v[16:17], v[18:19], v[16:17]" ::: CLOB);) }
		// 23: v_mul_f32 VGPR x VGPR, the result feeding the next (dependent chain)
		if(KIND == 23) { REP8(asm volatile("v_mul_f32 v4, v4, v5\n v_mul_f32 v4, v4, v6\n v_mul_f32 v4, v4, v7\n v_mul_f32 v4, v4, v9" ::: CLOB);) }
		// 24: v_mul_legacy / v_mul_f32 with DPP row_shr (a DPP read folded into the multiply)
		if(KIND == 24) { REP8(asm volatile("v_mul_f32_dpp v4, v5, v4 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp v8, v9, v8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp v12, v13, v12 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp v16, v17, v16 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf" ::: CLOB);) }
		// 25: v_cndmask_b32
		if(KIND == 25) { REP8(asm volatile("v_cndmask_b32 v4, v4, v5, vcc\n v_cndmask_b32 v8, v8, v9, vcc\n v_cndmask_b32 v12, v12, v13, vcc\n v_cndmask_b32 v16, v16, v17, vcc" ::: CLOB);) }
		// 26: v_mul_f32 with EXEC half off (32 lanes)
		if(KIND == 26) { asm volatile("s_mov_b64 exec, 0xffffffff" ::: "exec"); REP8(asm volatile("v_max_f32 v4, v4, v5\n v_max_f32 v8, v8, v9\n v_max_f32 v12, v12, v13\n v_max_f32 v16, v16, v17" ::: CLOB);) asm volatile("s_mov_b64 exec, -1" ::: "exec"); }
		// 27: integer add
		if(KIND == 27) { REP8(asm volatile("v_add_u32 v4, v4, v5\n v_add_u32 v8, v8, v9\n v_add_u32 v12, v12, v13\n v_add_u32 v16, v16, v17" ::: CLOB);) }
	}
	float r;
	asm volatile("v_add_f32 %0, v4, v8\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v5" : "=v"(r)::CLOB);
	out[blockIdx.x * 64 + threadIdx.x] = r;
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
template <int K> void all(float *d, int blocks, double *best) {
	if constexpr(K < 28) { double r = run<K>(d, blocks, 20000); best[K] = r < best[K] ? r : best[K]; all<K + 1>(d, blocks, best); }
}
int main() {
	float *d; hipMalloc(&d, 1 << 24);
	for(int i = 0; i < 40; i++) run<0>(d, 5120, 20000); // clock ramp-up
	const char *names[28] = {"v_max_f32 2 src, 2 banks", "v_max_f32 2 src, 1 bank", "v_max3/min3 3 banks", "v_max3/min3 1 bank", "v_max3/min3 2 of 3 in one bank",
							 "v_max3/min3 const + 2", "v_max3/min3 sgpr + 2", "v_mul_f32 sgpr, v (VOP2)", "v_mul_f32_e64 v, sgpr", "v_fma_f32 3 banks",
							 "v_max_f32 dependent chain", "max3 -> max -> min3 -> sub", "v_mul_f32 v, v", "v_mul_f32 const, v", "v_sub_f32 sgpr, v", "v_add_f32 v, v", "v_cmp_le_f32", "v_mov_b32 sgpr", "v_mov_b32 v", "v_mul / v_max alternating", "v_fma_f32 sgpr", "v_pk_mul_f32 sgpr pair", "v_pk_mul_f32 v pairs", "v_mul_f32 dependent chain", "v_mul_f32_dpp quad_perm", "v_cndmask_b32", "v_max_f32 with 32 lanes of EXEC", "v_add_u32"};
	for(int blocks : {6144}) {
		double best[28]; for(double &b : best) b = 1e9;
		for(int round = 0; round < 5; round++) all<0>(d, blocks, best);
		for(int k = 0; k < 28; k++) printf("%-34s %.3f ns per instruction per SIMD = %.2f cycles at 2.4 GHz (%.0f waves/SIMD)\n", names[k], best[k], best[k] * 2.4, blocks / 1024.0);
	}
	return 0;
}
