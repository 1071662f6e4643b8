/* include/snail_hip.h from PLAIN C (gcc -std=c99): the header is the drop-in boundary, and the reference's hosts are C++ only by habit -- any FFI (cgo, JNI, ctypes)
 * binds a C ABI.  Compiles the header as C, takes the address of every declared function (a missing export fails the link), and runs the entry points that need
 * no GPU: the host-side builder on the cube of scenes/box.obj, the table probe, error reporting. */
#include <stdio.h>
#include <string.h>
#include "../../include/snail_hip.h"

typedef void (*anyfn)(void);
static volatile anyfn sink;
#define ADDR(f) sink = (anyfn)(f); if(sink == (anyfn)0) return 90;
int main(void) {
	ADDR(snail_last_error) ADDR(snail_device_count) ADDR(snail_tris_from_verts) ADDR(snail_bvh_build) ADDR(snail_scene_create) ADDR(snail_scene_destroy)
	ADDR(snail_scene_create_lbvh) ADDR(snail_scene_download) ADDR(snail_scene_info) ADDR(snail_scene_flags) ADDR(snail_scene_set_arith) ADDR(snail_scene_arith)
	ADDR(snail_host_sse_tables) ADDR(snail_arith_set_tables) ADDR(snail_arith_prepare_device) ADDR(snail_host_sse_check) ADDR(snail_trace_primary)
	ADDR(snail_trace_primary_dev) ADDR(snail_trace_frame_packets) ADDR(snail_trace_packets_dev) ADDR(snail_trace_packets_shaded_dev) ADDR(snail_primary_slots)
	ADDR(snail_trace_primary_ordered_dev) ADDR(snail_trace_packets_ordered_dev) ADDR(snail_order_from_cost_dev) ADDR(snail_order_from_cost_hint_dev) ADDR(snail_trace_primary_batch_dev)
	ADDR(snail_trace_primary_batch_reorder_dev) ADDR(snail_trace_packets_shaded_batch_dev) ADDR(snail_packets_to_frame_dev) ADDR(snail_trace_rays) ADDR(snail_trace_rays_dev)
	ADDR(snail_trace_shadow) ADDR(snail_trace_shadow_dev) ADDR(snail_shade_depth_dev) ADDR(snail_shade_depth_arith_dev) ADDR(snail_packets_bgr_to_frame_dev)
	ADDR(snail_packets_bgr_to_frame_chunked_dev) ADDR(snail_packets_bgr_to_planar_dev) ADDR(snail_planar_to_frame_dev) ADDR(snail_render_whitted_dev)
	ADDR(snail_render_whitted_ordered_dev) ADDR(snail_render_whitted_reorder_dev) ADDR(snail_render_whitted_packets_dev) ADDR(snail_trace_transparency_dev)
	ADDR(snail_render_tiles) ADDR(snail_render_tiles_multi) ADDR(snail_render_image) ADDR(snail_account_primary) ADDR(snail_account_packets) ADDR(snail_last_launch)
	/* the cube of scenes/box.obj:15-36, faces flipped as the reference's loader does by default */
	static const float V[8][3] = {{1, -1, -1}, {1, -1, 1}, {-1, -1, 1}, {-1, -1, -1}, {1, 1, -1}, {0.999999f, 1, 1.000001f}, {-1, 1, 1}, {-1, 1, -1}};
	static const int F[12][3] = {{5, 1, 4}, {5, 4, 8}, {3, 7, 8}, {3, 8, 4}, {2, 6, 3}, {6, 7, 3}, {1, 5, 2}, {5, 6, 2}, {5, 8, 6}, {8, 7, 6}, {1, 2, 3}, {1, 3, 4}};
	float verts[12][9];
	unsigned char tris[12 * 64], nodes[24 * 32 + 64];
	int32_t perm[12];
	int nNodes = 0, depth = 0, i, k;
	for(i = 0; i < 12; i++) {
		const int idx[3] = {F[i][1] - 1, F[i][0] - 1, F[i][2] - 1};
		for(k = 0; k < 3; k++) memcpy(&verts[i][k * 3], V[idx[k]], 12);
	}
	if(snail_tris_from_verts(&verts[0][0], 12, tris) != 0) { printf("tris: %s\n", snail_last_error()); return 1; }
	if(snail_bvh_build(tris, 12, nodes, &nNodes, &depth, perm) != 0) { printf("build: %s\n", snail_last_error()); return 2; }
	if(nNodes != 9 || depth != 4)   /* what HostBVH.build gives for this cube (and the oracle: tests/test_oracle_pins.py) */ { printf("unexpected tree: %d nodes, depth %d\n", nNodes, depth); return 3; }
	{
		int seen[12] = {0};
		for(i = 0; i < 12; i++) { if(perm[i] < 0 || perm[i] > 11 || seen[perm[i]]) return 4; seen[perm[i]] = 1; }
	}
	if(snail_scene_create(NULL, 0, NULL, 0, 0, 0) != NULL || strlen(snail_last_error()) == 0) return 5;   /* an error is a status + a text, never an abort */
	if(snail_primary_slots(1920, 1080) != 8192 || SNAIL_PACKET_QUADS != 64 || SNAIL_MAX_DEPTH != 64) return 6;
	{
		static uint32_t tab[3 * 4096];
		const int rc = snail_host_sse_tables(tab);   /* 0 on an x86 host whose rcpps / rsqrtps have the table structure, 2 (with a reason) elsewhere */
		if(rc != 0 && rc != 2) return 7;
		if(rc == 0 && (tab[0] >> 23) != 126u && tab[0] != 0x3f800000u) return 8;
	}
	printf("C ABI ok: %d nodes, depth %d, devices %d\n", nNodes, depth, snail_device_count());
	return 0;
}
