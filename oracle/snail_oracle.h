/*
 * snail_oracle.h -- C-ABI of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * This library is a CPU restatement of the hot path of nadult/Snail (packetised SAH-BVH
 * traversal + ray/triangle intersection).  It exists only so that tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg can CHECK the HIP path; nothing under snail_amd/ may import,
 * link or call it.
 *
 * PARITY STATUS: "parity unpinned" in the strict sense of the build contract -- the reference has no
 * tests/golden vectors for this path and its translation units cannot be
 * compiled here without a stand-in for the absent libfwk submodule (which the contract forbids).
 * What IS pinned: (1) the arithmetic primitives (Inv/RSqrt/Min/Max/Condition, Vec3 dot/cross) against
 * the reference's own header-only veclib compiled from /root/reference/veclib (oracle/_ref/veclib_probe);
 * (2) the digests SURVEY.md section 8(c) recorded from the reference during the survey session
 * (box / lancia / feline / barracks tree hashes and hit sums), checked by tests/test_oracle_pins.py.
 *
 * Two arithmetic modes for the two approximate operations of the path (Inv, RSqrt):
 *   ORC_MODE_IEEE (0): veclib's scalar definitions  Inv(x)=1.0f/x, RSqrt(x)=1.0f/sqrtf(x)
 *                      (veclib/vecbase.h:53-55) -- bit-reproducible on any IEEE machine incl. gfx950.
 *   ORC_MODE_SSE  (1): veclib's SSE definitions     rcpps/rsqrtps + one Newton step
 *                      (veclib/sse/base.h:84-92)    -- what the reference executes on x86; the raw
 *                      approximations are CPU-vendor specific.
 *   ORC_MODE_TABLE (2): the SSE definitions with rcpps / rsqrtps of a NAMED CPU given as tables (orc_set_tables; 3 x 4096 words, the layout of
 *                      tests/golden/rcp_tables.npz): what the reference computes on that CPU, on any host.  With this host's own tables
 *                      (orc_tables_of_this_cpu) it equals ORC_MODE_SSE bit for bit (tests/test_oracle_pins.py::test_table_mode_*).
 */
#ifndef SNAIL_ORACLE_H
#define SNAIL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODE_IEEE = 0, ORC_MODE_SSE = 1, ORC_MODE_TABLE = 2 };

/* 64-byte triangle record, identical to the reference's `Triangle` (src/triangle.h:133-135):
 * a, ba, ca, t0, it0, pad, plane(nx,ny,nz,n.a) */
typedef struct OrcTri { float a[3], ba[3], ca[3]; float t0, it0; int32_t pad; float plane[4]; } OrcTri;

/* 32-byte node record, identical to `BVH::Node` (src/bvh/tree.h:60-72), little-endian variant:
 * bbox min,max; subNode|first (bit31 = leaf); {short axis, short firstNode} union int count */
typedef struct OrcNode { float bmin[3], bmax[3]; uint32_t sub; int32_t aux; } OrcNode;

/* camera as the reference's `Camera` (src/camera.h:7-14) flattened: pos, right, up, front, plane_dist */
typedef struct OrcCamera { float pos[3], right[3], up[3], front[3], plane_dist; } OrcCamera;

/* Triangle::Triangle + ComputeData (src/triangle.h:16-21,123-131). verts = n*9 floats (v0,v1,v2). */
void orc_tris_from_verts(const float *verts, int n, OrcTri *out);

/* BVH::Construct with flags = useSah (src/bvh/tree.cpp:293-328 -> FindSplitSweep :51-159).
 * Permutes `tris` in place (this defines triId). `perm[i]` = original index of the triangle now at i
 * (may be NULL). `nodes` must have room for 2*n entries. Returns node count; *depth as BVH::depth. */
int orc_bvh_build(OrcTri *tris, int n, OrcNode *nodes, int *depth, int32_t *perm);

/* FNV-1a 64 over node bytes / over triangle bytes skipping the uninitialised pad word (bytes 44..47) */
uint64_t orc_fnv_nodes(const OrcNode *nodes, int n);
uint64_t orc_fnv_tris(const OrcTri *tris, int n);

/* RayGenerator ctor + Generate(level 3) + SafeInv (src/ray_generator.cpp:4-47, src/rtbase.h:117-120).
 * dir/idir: 64 quads x {x[4],y[4],z[4]} = 768 floats each (the Vec3q memory layout). */
void orc_gen_packet(const OrcCamera *cam, int resx, int resy, int px, int py, int mode,
                    float *dir, float *idir);

/* BVH::TraversePrimaryN<sharedOrigin,hasMask> (src/bvh/traverse.cpp:14-80) over `npackets` packets of
 * `size` quads each.  Layouts are the reference's Context arrays: origin {x[4],y[4],z[4]} per quad
 * (one quad total per packet when shared), dir/idir likewise, mask = 1 byte per quad (low 4 bits) or
 * NULL, distance[4]/object[4] per quad, bary {u[4],v[4]} per quad.  In/out: distance, object, bary
 * must be initialised by the caller as Scene::RayTrace does (src/scene_trace.cpp:112-115).
 * stats[4] += {intersects, iters, rays(unchanged), skips}. */
void orc_trace_rays(const OrcNode *nodes, const OrcTri *tris, int npackets, int size, int sharedOrigin,
                    const float *origin, const float *dir, const float *idir, const uint8_t *mask,
                    float *distance, int32_t *object, float *bary, uint64_t *stats, int mode);

/* BVH::TraverseShadow (src/bvh/traverse.cpp:82-149). origin = 3 floats per packet (light position). */
void orc_trace_shadow(const OrcNode *nodes, const OrcTri *tris, int npackets, int size,
                      const float *origin, const float *dir, const float *idir,
                      float *distance, uint64_t *stats, int mode);

/* RenderTask::Work restricted to traversal (src/render.cpp:58-62,67-68,112-115 +
 * src/scene_trace.cpp:106-120): every 16x16 packet whose top-left lies in [x0,x0+w) x [y0,y0+h)
 * (x0,y0,w,h multiples of 16 except at the image edge). Outputs row-major resx*resy planes (only the
 * rect is written); pixels outside the image are traced (they are part of their packet) but not
 * stored. miss = (+inf, 0, 0, 0). stats[4] += {intersects, iters, rays, skips}. threads>=1. */
void orc_render_primary(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam,
                        int resx, int resy, int x0, int y0, int w, int h,
                        float *t, float *u, float *v, int32_t *triId, uint64_t *stats,
                        int mode, int threads);

/* orc_render_primary in ORC_MODE_SSE written FOUR LANES WIDE with SSE intrinsics (oracle/snail_sse4.inc: one SSE quad per __m128, as the
 * reference's f32x4 code runs: src/bounding_box.cpp:61-142, src/triangle.cpp:3-63): bit-identical outputs and TreeStats
 * (tests/test_oracle_semantics.py::test_sse4_port_equals_scalar_oracle); this is the timed CPU baseline of bench.py ("port-sse4"). */
void orc_render_primary_sse4(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam,
                             int resx, int resy, int x0, int y0, int w, int h,
                             float *t, float *u, float *v, int32_t *triId, uint64_t *stats, int threads);

/* Single-ray, cache-less accounting walk of SURVEY.md section 8(d): for every pixel of the rect (padded
 * to whole packets as above) walk the tree with that ray alone (child order from its own direction
 * signs, strict-< closest-hit pruning) and count V_n (node boxes tested) and V_t (triangles tested).
 * out[0] += rays, out[1] += sum V_n, out[2] += sum V_t, out[3] += hits. */
void orc_account_primary(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam,
                         int resx, int resy, int x0, int y0, int w, int h, uint64_t *out,
                         int mode, int threads);

/* gVals[1] depth shading + ConvColor -> 3 bytes (B,G,R) per ray (src/scene_trace.cpp:128-137, src/render.cpp:11-17,171-198) */
void orc_shade_depth(const float *t, int n, uint8_t *bgr, int mode);

/* Scene::RayTrace for primary packets in the reference's "simple shading" configuration (no shading data, gVals all
 * zero): TraversePrimary, samples (position = d*t + o, normal = triangle plane normal, diffuse = specular =
 * color*|d.n|; src/scene_trace.cpp:359-452, src/shading/simple_material.h:14-30), per light the packet-level cull
 * BoxPointDistanceSq(bbox of hit points) > radSq (src/scene_trace.cpp:494-501, src/funcs.cpp:8-49) and
 * Scene::TraceLight = shadow-ray generation + TraverseShadow + attenuation/accumulation
 * (src/scene_trace.cpp:523-601), outColor = diffuse*lDiffuse + specular*lSpecular, ConvColor -> B,G,R bytes
 * (src/render.cpp:11-17,171-198).  lights = nLights x {pos[3], color[3], radius}; frame_bgr = resy rows of `pitch`
 * bytes.  Lanes/quads the reference leaves UNINITIALISED (shadow dir/idir of missed lanes and of quads without any
 * hit, src/scene_trace.cpp:538-541) are zeros here; they are masked (distance = -inf) and cannot influence a result.
 * flags bit 0 = gVals[7], one reflection bounce (src/scene_trace.cpp:454-466): Scene::TraceReflection (:603-618) mirrors
 * every hit ray about its normal (Reflect, src/rtbase_math.h:54-58), origin = hit point + 0.001 * direction, traces the
 * packet as RayGroup<0,1> (per-ray origins, lane masks = hit lanes) through the same RayTrace -- samples, lights, shadow
 * packets, no further bounce -- and blends diffuse += (reflected colour - diffuse) * 0.3 BEFORE the primary's own lights.
 * Lanes that are masked off in the reflected packet (no primary hit) carry zeros here (see above).
 * stats[4] += {intersects, iters, rays (primary + reflected lanes + shadow lanes with N.L > 0), skips}.
 * flags bit 1 = gVals[9], 4x antialiasing of the tile renderer (src/render.cpp:60-62, :71-110): every 16x16 packet of the image is
 * the 2x2 reduction of four packets of the double-resolution frame, in the reference's operation order.
 * flags bit 2 = gVals[1], depth shading instead of the light pipeline (src/scene_trace.cpp:128-137): colour = Inv(t) * (20, 250, 2). */
void orc_render_whitted(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy,
                        const float *lights7, int nLights, const float ambient[3], const float color[3], int flags,
                        uint8_t *frame_bgr, int pitch, uint64_t *stats, int mode, int threads);

/* Scene::TraceTransparency (src/scene_trace.cpp:620-634) for primary packets: the rays of the listed packets (top-left corners packet_xy,
 * regenerated from the camera) continue behind their hits -- origin = dir * (t + 0.001) + origin, dir / idir unchanged -- for the lanes
 * of the caller's selector `sel` (1 byte per quad, the reference's transSel; lanes without a hit are dropped) as RayGroup<0,1>, through
 * the nested RayTrace of the simple-shading configuration (no reflections, no further transparency).  t = the packets' hit distances
 * (packet-major, 256 per packet); out_color = 3 floats per ray (`transColor`); stats += the nested call's counters. */
void orc_trace_transparency(const OrcNode *nodes, const OrcTri *tris, const OrcCamera *cam, int resx, int resy, const int32_t *packet_xy, int nPackets,
                            const float *t, const uint8_t *sel, const float *lights7, int nLights, const float ambient[3], const float color[3],
                            float *out_color, uint64_t *stats, int mode);

/* The render node's tile wire format: the `compress` store of RenderTask::Work (src/render.cpp:140-163) -- planes R, G-R, B-R
 * (mod 256) of tile (x, y, w, h) taken from an interleaved B,G,R frame -- and its inverse, DecompressTask::Work's plane loop
 * (src/compression.cpp:112-141). */
void orc_planar_encode_tile(const uint8_t *frame_bgr, int pitch, int x, int y, int w, int h, uint8_t *out);
void orc_planar_decode_tile(const uint8_t *planes, int x, int y, int w, int h, uint8_t *frame_bgr, int pitch);

/* instrumentation for tests/range_hist.py: hist[192] (see snail_oracle.cpp); NULL switches it off. Single-threaded only. */
void orc_debug_range_hist(uint64_t *hist);
void orc_debug_far_hist(uint64_t *hist512); /* tests/far_child_hist.py: [kind(2)][width(64)][4] counters, see snail_oracle.cpp g_farHist */
void orc_debug_set_mxcsr(unsigned v); /* tests only */
unsigned orc_caller_mxcsr(void); /* diagnostics: MXCSR of the calling thread (0x1f80 = default; every entry point above computes under the default) */

/* The shading path's small expressions (Abs, Reflect, SafeInv, FastInv, the light attenuation, ConvColor's channel, ForWhich / ForAny /
 * ForAll of a compare, Condition on a Vec3q, Vec3q dot / cross, Sqrt) evaluated by the SAME inline helpers the oracle's hot path calls, on
 * one row of eight floats; out = 57 words, laid out as oracle/veclib_probe.cpp `exprs` prints them.  tests/test_oracle_pins.py compares
 * this with the reference's veclib evaluating the same expressions. */
void orc_veclib_exprs(const float *in8, uint32_t *out57, int mode);

/* arithmetic primitives exposed for the veclib pin test */
float orc_inv(float x, int mode);
float orc_rsqrt(float x, int mode);
/* ORC_MODE_TABLE: the rcpps / rsqrtps tables every later call in that mode computes with (process-wide: set once, before the calls) -- bits of
 * rcpps(1.m), rsqrtps(1.m), rsqrtps(2 x 1.m), index = m >> 11; orc_tables_of_this_cpu reads them out of this host's instructions;
 * orc_raw_approx: n raw look-ups (fn 0 = rcpps, 1 = rsqrtps; bit patterns) by the table rule (table != 0) or by this host's instruction. */
void orc_set_tables(const uint32_t *tables12288);
void orc_tables_of_this_cpu(uint32_t *tables12288);
void orc_raw_approx(int fn, int table, const uint32_t *in, uint32_t *out, int n);
float orc_min(float a, float b);
float orc_max(float a, float b);

#ifdef __cplusplus
}
#endif
#endif
