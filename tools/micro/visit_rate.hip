// Microbenchmark: VALU cost of ONE node visit of the packet walk (4 rays per lane) in its candidate formulations, at 6 waves per SIMD --
// tools/micro/valu_banks.hip shows two classes of vector instructions on gfx950: plain VGPR-only fp32 mul / add / fma / mov at ~2.5 cycles per
// wave64, and everything that reads an SGPR, compares, takes min / max or uses DPP at ~4.3 cycles, the two classes overlapping when mixed.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -w tools/micro/visit_rate.hip -o /tmp/visit_rate && /tmp/visit_rate
#include <hip/hip_runtime.h>
#include <cstdio>
// registers: t0..t5 = v4..v9, s0..s3 = v10..v13, u0 = v14, idir x/y/z of ray L = v(20+L), v(24+L), v(28+L), dist = v(32+L), origin v36..v38,
// plane offsets v40..v45, second set v46..v51; node planes s40..s45
#define CLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "vcc"
#define TAILPOS(D, S) "v_max_f32 v4, 0, v4\n v_min_f32 v7, v7, " D "\n v_sub_f32 " S ", v7, v4\n"
#define MINMAX "v_max3_f32 v4, v4, v5, v6\n v_min3_f32 v7, v7, v8, v9\n"
#define SLABG(P0, P1, P2, P3, P4, P5, X, Y, Z)                                                                                              \
	"v_mul_f32 v4, " P0 ", " X "\n v_mul_f32 v5, " P1 ", " Y "\n v_mul_f32 v6, " P2 ", " Z "\n v_mul_f32 v7, " P3 ", " X "\n v_mul_f32 v8, " P4 ", " Y "\n v_mul_f32 v9, " P5 ", " Z "\n" MINMAX
// camera-relative records: the products straight out of the scalar registers;  plane offsets in VGPRs v40..v45
#define SLAB_R(X, Y, Z) SLABG("s40", "s41", "s42", "s43", "s44", "s45", X, Y, Z)
#define SLAB_V(X, Y, Z) SLABG("v40", "v41", "v42", "v43", "v44", "v45", X, Y, Z)
#define END "v_max_f32 v12, v12, v13\n v_max3_f32 v10, v10, v11, v12\n v_cmp_le_f32 vcc, 0, v10\n"
#define RAY0(SL) SL("v20", "v24", "v28") TAILPOS("v32", "v10")
#define RAY1(SL) SL("v21", "v25", "v29") TAILPOS("v33", "v11")
#define RAY2(SL) SL("v22", "v26", "v30") TAILPOS("v34", "v12")
#define RAY3(SL) SL("v23", "v27", "v31") TAILPOS("v35", "v13")
#define VISIT_R RAY0(SLAB_R) RAY1(SLAB_R) RAY2(SLAB_R) RAY3(SLAB_R) END
#define VISIT_V RAY0(SLAB_V) RAY1(SLAB_V) RAY2(SLAB_V) RAY3(SLAB_V) END
// two rays per packed multiply: products of rays (0,1) then (2,3) in register pairs v[40:51], planes broadcast out of the SGPR pairs by op_sel
#define PKPROD(X, Y, Z)                                                                                                                     \
	"v_pk_mul_f32 v[40:41], s[40:41], " X " op_sel_hi:[0,1]\n v_pk_mul_f32 v[42:43], s[40:41], " Y " op_sel:[1,0] op_sel_hi:[1,1]\n"         \
	"v_pk_mul_f32 v[44:45], s[42:43], " Z " op_sel_hi:[0,1]\n v_pk_mul_f32 v[46:47], s[42:43], " X " op_sel:[1,0] op_sel_hi:[1,1]\n"         \
	"v_pk_mul_f32 v[48:49], s[44:45], " Y " op_sel_hi:[0,1]\n v_pk_mul_f32 v[50:51], s[44:45], " Z " op_sel:[1,0] op_sel_hi:[1,1]\n"
#define PKRAY(H0, H1, H2, H3, H4, H5, D, S) "v_max3_f32 v4, " H0 ", " H1 ", " H2 "\n v_min3_f32 v7, " H3 ", " H4 ", " H5 "\n" TAILPOS(D, S)
#define VISIT_PK PKPROD("v[20:21]", "v[24:25]", "v[28:29]") PKRAY("v40", "v42", "v44", "v46", "v48", "v50", "v32", "v10") PKRAY("v41", "v43", "v45", "v47", "v49", "v51", "v33", "v11") \
	PKPROD("v[22:23]", "v[26:27]", "v[30:31]") PKRAY("v40", "v42", "v44", "v46", "v48", "v50", "v34", "v12") PKRAY("v41", "v43", "v45", "v47", "v49", "v51", "v35", "v13") END
// ... and the two rays' slack by one packed subtraction: (min(tf, d) pair v[8:9]) - (max(tn, 0) pair v[14:15])
#define PKRAY2(H0, H1, H2, H3, H4, H5, D, TN, TF) "v_max3_f32 v4, " H0 ", " H1 ", " H2 "\n v_min3_f32 v7, " H3 ", " H4 ", " H5 "\n v_max_f32 " TN ", 0, v4\n v_min_f32 " TF ", v7, " D "\n"
#define VISIT_PK2 PKPROD("v[20:21]", "v[24:25]", "v[28:29]") PKRAY2("v40", "v42", "v44", "v46", "v48", "v50", "v32", "v14", "v8") PKRAY2("v41", "v43", "v45", "v47", "v49", "v51", "v33", "v15", "v9") \
	"v_pk_add_f32 v[10:11], v[8:9], v[14:15] neg_lo:[0,1] neg_hi:[0,1]\n"                                                                   \
	PKPROD("v[22:23]", "v[26:27]", "v[30:31]") PKRAY2("v40", "v42", "v44", "v46", "v48", "v50", "v34", "v14", "v8") PKRAY2("v41", "v43", "v45", "v47", "v49", "v51", "v35", "v15", "v9") \
	"v_pk_add_f32 v[12:13], v[8:9], v[14:15] neg_lo:[0,1] neg_hi:[0,1]\n" END
#define PRE_SUB "v_sub_f32 v40, s40, v36\n v_sub_f32 v41, s41, v37\n v_sub_f32 v42, s42, v38\n v_sub_f32 v43, s43, v36\n v_sub_f32 v44, s44, v37\n v_sub_f32 v45, s45, v38\n"
#define PRE_MOV "v_mov_b32 v40, s40\n v_mov_b32 v41, s41\n v_mov_b32 v42, s42\n v_mov_b32 v43, s43\n v_mov_b32 v44, s44\n v_mov_b32 v45, s45\n"
// plane offsets of the NEXT record moved to the second VGPR set during this visit, one move between the rays' blocks, sets swapped by a VGPR move
#define PRE_VMOV "v_mov_b32 v40, v46\n v_mov_b32 v41, v47\n v_mov_b32 v42, v48\n v_mov_b32 v43, v49\n v_mov_b32 v44, v50\n v_mov_b32 v45, v51\n"
template <int KIND> __global__ __launch_bounds__(64) void k(float *out, int iters) {
	asm volatile("v_cvt_f32_u32 v20, %0\n v_add_f32 v20, 1.0, v20\n v_add_f32 v21, 1.0, v20\n v_add_f32 v22, 2.0, v20\n v_add_f32 v23, 4.0, v20\n"
				 "v_add_f32 v24, 0.5, v20\n v_add_f32 v25, 0.5, v21\n v_add_f32 v26, 0.5, v22\n v_add_f32 v27, 0.5, v23\n"
				 "v_add_f32 v28, 0.5, v24\n v_add_f32 v29, 0.5, v25\n v_add_f32 v30, 0.5, v26\n v_add_f32 v31, 0.5, v27\n"
				 "v_mov_b32 v32, 0x7f800000\n v_mov_b32 v33, 0x7f800000\n v_mov_b32 v34, 0x7f800000\n v_mov_b32 v35, 0x7f800000\n"
				 "v_mov_b32 v36, 1.0\n v_mov_b32 v37, 2.0\n v_mov_b32 v38, 4.0\n v_mov_b32 v46, 1.0\n v_mov_b32 v47, 2.0\n v_mov_b32 v48, 4.0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, 2.0\n v_mov_b32 v51, 4.0\n"
				 "s_mov_b32 s40, 1.0\n s_mov_b32 s41, 2.0\n s_mov_b32 s42, 4.0\n s_mov_b32 s43, 0.5\n s_mov_b32 s44, 4.0\n s_mov_b32 s45, 2.0\n" ::"v"(threadIdx.x)
				 : CLOB, "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "s40", "s41", "s42", "s43", "s44", "s45");
	for(int i = 0; i < iters; i++) {
		if(KIND == 0) { asm volatile(VISIT_R VISIT_R ::: CLOB); }                        // camera-relative, products from SGPRs (47 per visit)
		if(KIND == 1) { asm volatile(PRE_SUB VISIT_V PRE_SUB VISIT_V ::: CLOB); }        // plane offsets subtracted per visit (53)
		if(KIND == 2) { asm volatile(PRE_MOV VISIT_V PRE_MOV VISIT_V ::: CLOB); }        // camera-relative, planes moved to VGPRs first (53)
		if(KIND == 3) { asm volatile(PRE_VMOV VISIT_V PRE_VMOV VISIT_V ::: CLOB); }      // planes arrive in VGPRs (vector load), VGPR move (53)
		if(KIND == 4) { asm volatile(VISIT_V VISIT_V ::: CLOB); }                        // planes in place in VGPRs, no move (47)
		// 5: as 2, the six moves spread between the ray blocks of the PREVIOUS visit (software pipelined into the second set, then 6 VGPR moves)
		if(KIND == 5) { asm volatile("v_mov_b32 v46, s40\n v_mov_b32 v47, s41\n" RAY0(SLAB_V) "v_mov_b32 v48, s42\n v_mov_b32 v49, s43\n" RAY1(SLAB_V)
									 "v_mov_b32 v50, s44\n v_mov_b32 v51, s45\n" RAY2(SLAB_V) RAY3(SLAB_V) END PRE_VMOV ::: CLOB); }
		if(KIND == 6) { asm volatile(VISIT_PK VISIT_PK ::: CLOB); }                      // packed products, two rays each (35)
		if(KIND == 7) { asm volatile(VISIT_PK2 VISIT_PK2 ::: CLOB); }                    // ... and packed slack subtraction (33)
	}
	float r;
	asm volatile("v_add_f32 %0, v10, v11\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v13\n v_add_f32 %0, %0, v4" : "=v"(r)::CLOB);
	out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int KIND> double run(float *d, int blocks, int iters, int visitsPerIter) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1);
	return ms * 1e6 / ((double)iters * visitsPerIter * (blocks / 1024.0)); // ns per visit per SIMD
}
int main() {
	float *d; hipMalloc(&d, 1 << 24);
	for(int i = 0; i < 40; i++) run<0>(d, 6144, 4000, 2); // clock ramp-up
	const char *names[8] = {"relative records, SGPR products (47)", "v_sub per visit + VGPR products (53)", "v_mov sgpr->vgpr + VGPR products (53)",
							"VGPR move + VGPR products (53)", "VGPR products only (47)", "sgpr moves spread + VGPR move (59)", "packed products from SGPR pairs (35)", "packed products + packed slack (33)"};
	const int vpi[8] = {2, 2, 2, 2, 2, 1, 2, 2};
	for(int blocks : {1024, 5120, 6144}) {
		double best[8]; for(double &b : best) b = 1e9;
		for(int round = 0; round < 5; round++) {
			double r[8] = {run<0>(d, blocks, 4000, 2), run<1>(d, blocks, 4000, 2), run<2>(d, blocks, 4000, 2), run<3>(d, blocks, 4000, 2), run<4>(d, blocks, 4000, 2), run<5>(d, blocks, 4000, 1), run<6>(d, blocks, 4000, 2), run<7>(d, blocks, 4000, 2)};
			for(int k = 0; k < 8; k++) best[k] = r[k] < best[k] ? r[k] : best[k];
		}
		(void)vpi;
		for(int k = 0; k < 8; k++) printf("%-44s %.1f ns per visit per SIMD = %.0f cycles at 2.4 GHz (%.0f waves/SIMD)\n", names[k], best[k], best[k] * 2.4, blocks / 1024.0);
	}
	return 0;
}
