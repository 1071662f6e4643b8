/*
 * snail_hip.h -- C-ABI of libsnailhip.so: the MI355X (gfx950) replacement for the hot path of
 * nadult/Snail: packetised SAH-BVH traversal + ray/triangle intersection.
 *
 * The reference has no FFI layer; its seam is the compile-time `AccStruct` concept of
 * `template <class AccStruct> class Scene` (src/scene.h:26-58) and, one level up, the tile API of
 * src/render.h:16-28.  A per-packet GPU call (256 rays) would be launch-latency bound, so this
 * boundary sits at frame / tile / packet-batch granularity; include/snail_adapter.hpp re-exposes the
 * reference's own shapes on top of it (an AccStruct-conforming class, with which the reference's own
 * Render<AccStruct> / Scene<AccStruct> templates instantiate unchanged, and a frame-granular TraceFrame).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a non-zero
 * status on failure with the text available from snail_last_error() (the reference aborts through
 * FATAL/ASSERT, src/rtbase.h:13 -- the adapter turns non-zero into that behaviour).  Buffers are
 * caller-owned.  `*_dev` entry points take DEVICE pointers and a hipStream_t (as void*), never
 * synchronise and are graph-capturable (once a first call with that frame size has allocated the handle's scratch); the un-suffixed entry points take HOST pointers, run on the
 * scene's device and return after the results are in the host buffers.
 *
 * Concurrency: a SnailScene is THREAD-SAFE.  The reference hands one `const Scene<AccStruct>` to `threads` pthread workers, each of which calls
 * TraversePrimary / TraverseShadow on it (src/render.cpp:214-267, src/thread_pool.cpp:151-180, src/scene_trace.cpp:119-120,:560-563); any number
 * of host threads may likewise be inside any entry points of ONE handle at once:
 *   - the un-suffixed (host-pointer) entry points give every call a stream, counter words and a staging arena of its own, so concurrent calls
 *     run side by side on the device and each returns its own results and its own stats[4];
 *   - the `*_dev` entry points may be issued from several threads on several streams: what a launch books in the handle (its slot of the 8
 *     round-robin scratch sets -- guarded by events, so a slot's next user waits, on the device, for its previous one --, the cache of
 *     origin-relative node records) is booked under a per-handle lock that is held while a call ENQUEUES, never while it waits for the device;
 *   - snail_render_tiles / snail_render_image calls on one handle render out of that handle's ONE cached tile job and therefore take turns
 *     (calls on different handles do not); snail_render_tiles_multi takes its handles' turns in address order.
 * snail_scene_destroy and snail_scene_set_arith are the caller's to order against calls in flight (set_arith changes what LATER launches compute).
 * snail_last_error() is per host thread.
 *
 * Record layouts are the reference's own:
 *   node  = 32 B  `BVH::Node`  (src/bvh/tree.h:60-72): bbox min[3], max[3]; subNode|first (bit 31 =
 *                 leaf); {short axis, short firstNode} (inner) or int count (leaf)
 *   tri   = 64 B  `Triangle`   (src/triangle.h:133-135): a, ba, ca, t0, it0, pad, plane(n, n.a)
 *   quad arrays  = the memory layout of `Vec3q[]`, `floatq[]`, `i32x4[]`, `Vec2q[]`:
 *                 dir/idir/origin: per quad {x[4], y[4], z[4]} (12 floats); distance/object: 4 per quad;
 *                 barycentric: per quad {u[4], v[4]} (8 floats); mask: 1 byte per quad, low 4 bits.
 *   stats[4]     = {intersects, loop iterations, traced rays, skips}: TreeStats data[0], [1], [2], [9]
 *                 (src/tree_stats.h:86-113); counters are ADDED to.
 */
#ifndef SNAIL_HIP_H
#define SNAIL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SnailScene SnailScene;

#define SNAIL_PACKET_QUADS 64   /* 16x16 pixels = 64 SSE quads = 256 rays (src/render.cpp:50-53)   */
#define SNAIL_MAX_DEPTH    64   /* BVH::maxDepth (src/bvh/tree.h:33)                               */

/* ---- errors -------------------------------------------------------------------------------------- */
const char *snail_last_error(void);
/* Non-zero when the library was built with device code for the running GPU and a HIP device exists. */
int snail_device_count(void);

/* ---- host-side scene construction (stays on the CPU, exactly as in the reference) --------------- */
/* Triangle::Triangle + ComputeData (src/triangle.h:16-21,123-131). verts9: n x (v0,v1,v2). */
int snail_tris_from_verts(const float *verts9, int n, void *tris64);
/* BVH::Construct(scene, BVH::useSah) -> FindSplitSweep (src/bvh/tree.cpp:293-328, 51-159).
 * Permutes tris64 in place (this ordering DEFINES triId).  nodes32 must hold 2*nTris records.
 * perm (optional) receives, per final slot, the index the triangle had on input.
 * Both host-side functions compute under the default floating-point environment (round to nearest, denormals kept) whatever the
 * calling thread's MXCSR says, and restore it: the tree does not depend on what else the process has loaded. */
int snail_bvh_build(void *tris64, int nTris, void *nodes32, int *nNodes, int *depth, int32_t *perm);

/* ---- device scene -------------------------------------------------------------------------------- */
/* Mirrors the public BVH::nodes / BVH::tris / BVH::depth (src/bvh/tree.h:86-92), i.e. what the
 * reference ships to a render node in SendBVH (src/server.cpp:144-164).  Copies to `device`.
 * Device memory per scene: the caller's records (32 B x nNodes + 64 B x nTris), the node loop's copy of them (a fixed 32 MiB region +
 * 64 B x nTris) and, created on demand, one 32 B x (nNodes + 1) array of node boxes relative to each distinct shared origin the scene is
 * traced from -- camera positions, light positions; at most 16 are kept, least recently used first out. */
SnailScene *snail_scene_create(const void *nodes32, int nNodes, const void *tris64, int nTris, int depth, int device);
void snail_scene_destroy(SnailScene *);
/* GPU tree builder option (NOT the parity tree, reported separately): a linear BVH -- Morton order of the
 * triangle-box centres, one device radix sort, Karras' binary radix tree, bottom-up refit, subtrees of <= maxLeafTris (1..64)
 * triangles collapsed into leaves -- in the reference's node / triangle record formats, built and kept on `device`; the traversal
 * entry points below work on it unchanged.  Triangle order (triId) and tree shape differ from BVH::Construct: hit distances agree
 * with the SAH tree's except where the reference's order-dependent packet culls differ; perm (optional, nTris) maps triId ->
 * input triangle; *build_ms (optional) = device time of the build.  tri_verts: host, nTris x (v0,v1,v2). */
SnailScene *snail_scene_create_lbvh(const float *tri_verts, int nTris, int device, int maxLeafTris, int32_t *perm, float *build_ms);
/* Copy a scene's node (nNodes x 32 B) and / or triangle (nTris x 64 B) records back to the host (either may be NULL). */
int snail_scene_download(const SnailScene *, void *nodes32, void *tris64);
int snail_scene_info(const SnailScene *, int *nNodes, int *nTris, int *depth, int *device);
/* What snail_scene_create found: fastOK = every record finite and of sane magnitude (else every packet takes the M_EXACT walk);
 * nestedOK = every child box lies inside its parent's box (true for the reference's builders; a tree that is not nested is walked by
 * the node loop that rescans the whole inherited quad range at every box, exactly as src/bounding_box.cpp:71-139 does). */
int snail_scene_flags(const SnailScene *, int *fastOK, int *nestedOK);

/* ---- arithmetic of the path's approximate operations ---------------------------------------------- */
/* The reference's veclib has two definitions of Inv / RSqrt / FastInv: the scalar one (1 / x, 1 / sqrt(x), correctly rounded:
 * veclib/vecbase.h:53-57) and the one its x86 build executes -- rcpps / rsqrtps + one Newton step (veclib/sse/base.h:84-92,
 * veclib/sse/f32.h:98-102), whose low bits depend on the CPU's look-up tables.  They differ by ~1e-7 relative; the operations reach
 * the hit records through the normalisation and inversion of ray directions (src/ray_generator.cpp:41-44, src/rtbase.h:117-120,
 * src/scene_trace.cpp:547,:615), through 1 / det of an accepted hit (src/triangle.cpp:55), and the pictures additionally through
 * the depth shading (src/scene_trace.cpp:128-137) and the light's attenuation (:585-587).
 *   SNAIL_ARITH_IEEE      (default) the scalar definitions: results do not depend on the host.
 *   SNAIL_ARITH_HOST_SSE  the SSE definitions AS THE HOST'S CPU EXECUTES THEM: its rcpps / rsqrtps tables (3 x 4096 words, taken from
 *                         the CPU at first use and verified against the instructions) are reproduced on the device, followed by
 *                         veclib's Newton steps, each operation rounded separately.  Every entry point that takes the scene then returns
 *                         what the reference's SSE build computes on this machine, bit for bit (hit records, TreeStats, pictures).
 * snail_scene_set_arith returns 2 -- nothing changes -- when the host's instructions do not have the table structure (host_sse.h);
 * rays handed in by the caller (snail_trace_rays / _shadow) are used as they are: only 1 / det depends on the mode there. */
#define SNAIL_ARITH_IEEE     0
#define SNAIL_ARITH_HOST_SSE 1
int snail_scene_set_arith(SnailScene *, int arith);
int snail_scene_arith(const SnailScene *, int *arith);
/* The host CPU's tables: T_rcp[4096] (bits of rcpps(1.m), index m >> 11), T_rsqrt[4096] for [1, 2), T_rsqrt[4096] for [2, 4); NULL = only
 * the status.  No device needed. */
int snail_host_sse_tables(uint32_t *tables12288);
/* SNAIL_ARITH_HOST_SSE with the tables of ANOTHER CPU (3 x 4096 words in the layout above, e.g. what snail_host_sse_tables returned on that
 * machine; NULL = this host's own again): every machine of a render farm then computes what the reference computes on the CPU the tables came
 * from, whatever its own CPU is -- the reference's own cluster mixed CPU families (readme_distributed.txt), whose rcpps / rsqrtps differ.
 * Process-wide; a device's copy is refreshed by the next snail_scene_set_arith(scene, SNAIL_ARITH_HOST_SSE) of a scene on it (which waits for
 * that device's work in flight): set the tables first, then the scenes' arithmetic.  snail_host_sse_tables returns the tables in force;
 * snail_host_sse_check keeps comparing with THIS host's instructions.  Returns 1 for tables that are not tables of reciprocals. */
int snail_arith_set_tables(const uint32_t *tables12288);
/* The CURRENT device's copy of the tables in force, for the one scene-less entry point that computes in this arithmetic (snail_shade_depth_arith_dev):
 * made or refreshed now (synchronises the device), so that later calls -- inside a stream capture, say -- find it current. */
int snail_arith_prepare_device(void);
/* The table rule against the instruction itself over the float bit patterns [first, first + count) (fn 0 = rcpps, 1 = rsqrtps), on
 * `threads` host threads: *mismatches = inputs whose result bits differ, *firstBad = the lowest of them.  No device needed. */
int snail_host_sse_check(int fn, uint64_t first, uint64_t count, int threads, uint64_t *mismatches, uint32_t *firstBad);

/* ---- primary packets: RayGenerator + SafeInv + TraversePrimary<1,0> ----------------------------- */
/* Replaces the per-packet body of RenderTask::Work (src/render.cpp:58-62,67-68,112-115) plus the
 * hit-array initialisation of Scene::RayTrace (src/scene_trace.cpp:106-120) for every 16x16 packet
 * whose top-left corner lies in the rect [x0,x0+w) x [y0,y0+h) (x0,y0 multiples of 16).
 * cam = pos[3], right[3], up[3], front[3], plane_dist (src/camera.h:7-14).
 * Outputs are row-major resx*resy planes; only pixels inside both the rect and the image are written;
 * rays of a packet that fall outside the image are still traced (they are part of the packet).
 * miss = (t=+inf, u=0, v=0, triId=0) as src/scene_trace.cpp:112-115.  Any output plane may be NULL. */
int snail_trace_primary(SnailScene *, const float cam[13], int resx, int resy, int x0, int y0, int w, int h,
                        float *t, float *u, float *v, int32_t *triId, uint64_t stats[4]);
int snail_trace_primary_dev(SnailScene *, const float cam[13], int resx, int resy, int x0, int y0, int w, int h,
                            float *d_t, float *d_u, float *d_v, int32_t *d_triId, uint64_t *d_stats, void *stream);

/* The whole frame's primary packets with HOST outputs in PACKET-MAJOR order (packet (cx, cy) of the ceil(resx/16) x ceil(resy/16) grid at
 * index cy * pw + cx; element [p*256 + q*4 + lane] in the reference's quad order): exactly the Context arrays a host's per-packet loop
 * reads (src/ray_generator.cpp:29-45), so that no per-pixel re-layout is left to the CPU.  Rays of edge packets that fall outside the
 * image carry what the walk found for them (they are part of their packet, src/render.cpp:67-68).  Buffers: pw * ph * 256 elements each;
 * any of them may be NULL. */
int snail_trace_frame_packets(SnailScene *, const float cam[13], int resx, int resy, float *t, float *u, float *v, int32_t *triId,
                              uint64_t stats[4]);

/* Same, for an explicit list of packets (tile sharding, src/render.cpp:244-267 / src/node.cpp:336-338):
 * d_packet_xy = nPackets x (x,y) top-left pixel.  Outputs are PACKET-MAJOR in the reference's quad
 * order (quad ty*4+k = pixels x+4k..x+4k+3 of row y+ty, src/ray_generator.cpp:29-45): element
 * [p*256 + q*4 + lane], i.e. exactly the Context arrays host shading reads (bary as two planes). */
int snail_trace_packets_dev(SnailScene *, const float cam[13], int resx, int resy, const int32_t *d_packet_xy, int nPackets,
                            float *d_t, float *d_u, float *d_v, int32_t *d_triId, uint64_t *d_stats, void *stream);
/* The same launch with the gVals[1] depth shading + ConvColor (src/scene_trace.cpp:128-137, src/render.cpp:11-17) fused into the
 * kernel's epilogue: the only output is d_bgr, packet-major [nPackets][256][3] bytes (4-byte aligned) -- what a render node returns
 * per tile.  Equal byte for byte to snail_trace_packets_dev followed by snail_shade_depth_dev. */
int snail_trace_packets_shaded_dev(SnailScene *, const float cam[13], int resx, int resy, const int32_t *d_packet_xy, int nPackets,
                                   uint8_t *d_bgr, uint64_t *d_stats, void *stream);
/* Dispatch-order feedback (no counterpart in the reference, whose thread pool pulls tiles dynamically, src/render.cpp:258-267 /
 * src/thread_pool.cpp): packet costs are heavy-tailed, and a frame ends with its heaviest packets.  The *_ordered_dev launches are
 * the launches above plus two optional device arrays of nSlots int32 -- nSlots = snail_primary_slots(w, h) for a rect (>= the
 * number of its packets; some slots hold no packet) and nPackets for a list:
 *   d_slot_cost (out): node visits of each slot's packet in THIS launch (0 for an empty slot);
 *   d_order     (in) : the slot each workgroup takes, a permutation of [0, nSlots); NULL = the built-in order.
 * snail_order_from_cost_dev turns the costs of one frame into the order of the next (stream-ordered, one small kernel on the CURRENT
 * device, no scratch; costs below zero count as zero and, for inputs of up to 49152 slots, costs above 65535 as 65535).  The order it
 * writes is ALWAYS a permutation of [0, nSlots); which one is the library's choice (round 5): for HEAVY-TAILED costs -- the cost of the
 * slot at the 99th percentile at least 3x the mean cost -- heaviest first (a counting sort over 4096 cost classes, the order inside a
 * class arbitrary), so that the launch does not end on a few long packets; for flatter costs the built-in order (neighbouring packets on
 * the same XCD, whose L2 then holds their subtrees), which is what measures faster there and does not age when the camera moves.  Inputs
 * of more than 49152 slots or not 16-byte aligned take a multi-pass kernel that always sorts.  The *_reorder_dev launches below apply the
 * same rule.  A caller who KNOWS the costs to be exact for the launches that will use the order -- the same camera, a still view -- says so
 * with SNAIL_ORDER_SORTED (snail_order_from_cost_hint_dev, the order_flags of the *_reorder_dev launches): then the order is the sorted one
 * whatever the shape of the costs, which ends a short run of frames ~4 % sooner (the last frames' tails are their heaviest packets) and costs
 * a long one nothing; with costs of an earlier VIEW it would be the wrong choice for flat costs, hence the default.
 * Results never depend on the order: hit records and counters are identical.
 * The caller keeps d_order unchanged while a launch reading it is in flight. */
int snail_primary_slots(int w, int h);
int snail_trace_primary_ordered_dev(SnailScene *, const float cam[13], int resx, int resy, int x0, int y0, int w, int h,
                                    float *d_t, float *d_u, float *d_v, int32_t *d_triId, uint64_t *d_stats,
                                    const int32_t *d_order, int32_t *d_slot_cost, void *stream);
int snail_trace_packets_ordered_dev(SnailScene *, const float cam[13], int resx, int resy, const int32_t *d_packet_xy, int nPackets,
                                    float *d_t, float *d_u, float *d_v, int32_t *d_triId, uint64_t *d_stats,
                                    const int32_t *d_order, int32_t *d_slot_cost, void *stream);
#define SNAIL_ORDER_AUTO 0   /* order_flags: the library's rule above */
#define SNAIL_ORDER_SORTED 1 /*              heaviest first whatever the shape of the costs (the caller knows them to be exact) */
int snail_order_from_cost_dev(const int32_t *d_slot_cost, int nSlots, int32_t *d_order, void *stream);   /* = ..._hint_dev(.., SNAIL_ORDER_AUTO, ..) */
int snail_order_from_cost_hint_dev(const int32_t *d_slot_cost, int nSlots, int32_t *d_order, int order_flags, void *stream);
/* Dispatch-order feedback WITHOUT a launch of its own (round 5): the *_reorder_dev forms of the multi-frame primary launch (below) and of the staged frame
 * (snail_render_whitted_reorder_dev) are their *_ordered_dev forms plus d_next_order (out; same shape as d_order; needs d_slot_cost): the order(s) the NEXT
 * such launch should use, derived from THIS launch's costs inside the launch -- by one extra workgroup of the small deferred-packet pass that follows every
 * traversal kernel anyway -- so that a moving camera's order refresh costs no kernel launch, no stream round trip and no second pass over the costs.  d_next_order
 * may be d_order itself (the traversal kernel that reads d_order has finished when it is rewritten).  The order snail_order_from_cost_hint_dev would
 * give for those costs and order_flags (SNAIL_ORDER_AUTO / SNAIL_ORDER_SORTED; the order inside a class is arbitrary in both); frames with more than
 * 20480 slots fall back to that kernel inside the call. */
/* Multi-frame launches: ONE launch traces nFrames (1..SNAIL_MAX_BATCH) frames of the same rect / packet list, each with its own camera
 * (cams13: HOST array nFrames x 13) and its own output planes (HOST arrays of nFrames DEVICE pointers; a NULL array = that plane is not
 * wanted).  The heaviest packets of all the frames are dispatched first; a frame's tail -- ~0.2 ms whatever the launch holds -- and the
 * launch overheads are paid once per nFrames frames.  What it costs is latency (a frame is complete when its launch is): the choice of
 * a host that renders a camera path for throughput, or whose launches are small (one rank's share of a frame).  Hit records, shaded
 * bytes and counters are those of nFrames single-frame launches; d_slot_cost receives the first frame's costs. */
#define SNAIL_MAX_BATCH 8
int snail_trace_primary_batch_dev(SnailScene *, int nFrames, const float *cams13, int resx, int resy, float *const *d_t, float *const *d_u,
                                  float *const *d_v, int32_t *const *d_triId, uint64_t *d_stats, const int32_t *d_order,
                                  int32_t *d_slot_cost, void *stream);
int snail_trace_primary_batch_reorder_dev(SnailScene *, int nFrames, const float *cams13, int resx, int resy, float *const *d_t, float *const *d_u,
                                          float *const *d_v, int32_t *const *d_triId, uint64_t *d_stats, const int32_t *d_order,
                                          int32_t *d_slot_cost, int32_t *d_next_order, int order_flags, void *stream);
int snail_trace_packets_shaded_batch_dev(SnailScene *, int nFrames, const float *cams13, int resx, int resy, const int32_t *d_packet_xy,
                                         int nPackets, uint8_t *const *d_bgr, uint64_t *d_stats, void *stream);
/* Scatter packet-major planes into row-major resx*resy frame planes (clipped to the image). */
int snail_packets_to_frame_dev(const int32_t *d_packet_xy, int nPackets, int resx, int resy,
                               const float *d_pt, const float *d_pu, const float *d_pv, const int32_t *d_pid,
                               float *d_t, float *d_u, float *d_v, int32_t *d_triId, void *stream);

/* ---- arbitrary packets: TraversePrimary<sharedOrigin,hasMask>(Context&) ------------------------- */
/* src/bvh/traverse.cpp:14-80, callers src/scene_trace.cpp:119-120 (primary), :615-617 (reflection),
 * :631-633 (transparency).  nPackets packets of `size` quads (1..64).  origin: one quad per packet if
 * sharedOrigin else `size` quads per packet.  mask: NULL (hasMask=0) or 1 byte per quad.
 * distance/object/bary are IN/OUT and must be initialised by the caller as Scene::RayTrace does
 * (+inf / -inf for masked lanes, 0).  `element` of Context is not written by BVH and is not passed.
 * bary may be NULL (both forms): barycentrics are then not tracked, which is the cheaper walk (what the staged shading pipeline uses). */
int snail_trace_rays(SnailScene *, int nPackets, int size, int sharedOrigin, const float *origin, const float *dir,
                     const float *idir, const uint8_t *mask, float *distance, int32_t *object, float *bary, uint64_t stats[4]);
int snail_trace_rays_dev(SnailScene *, int nPackets, int size, int sharedOrigin, const float *d_origin, const float *d_dir,
                         const float *d_idir, const uint8_t *d_mask, float *d_distance, int32_t *d_object, float *d_bary,
                         uint64_t *d_stats, void *stream);

/* ---- shadow packets: TraverseShadow(ShadowContext&) --------------------------------------------- */
/* src/bvh/traverse.cpp:82-149, caller src/scene_trace.cpp:560-563.  origin = 3 floats per packet (the
 * light position).  distance is IN/OUT: in = max t per lane, negative = lane masked; out = -inf where
 * occluded (src/triangle.cpp:94-98). */
int snail_trace_shadow(SnailScene *, int nPackets, int size, const float *origin3, const float *dir, const float *idir,
                       float *distance, uint64_t stats[4]);
int snail_trace_shadow_dev(SnailScene *, int nPackets, int size, const float *d_origin3, const float *d_dir, const float *d_idir,
                           float *d_distance, uint64_t *d_stats, void *stream);

/* ---- framebuffer (the reference's simplest shading mode + RGB8 store) ----------------------------- */
/* gVals[1] "very simple shading" (src/scene_trace.cpp:128-137): c = Inv(t) * (20, 250, 2) per pixel, then
 * ConvColor (src/render.cpp:11-17): trunc(clamp(c*255, 0, 255)) packed b | g<<8 | r<<16 and stored as 3 bytes
 * (B,G,R) per pixel (src/render.cpp:171-198).  d_t: packet-major distances [nPackets][256];
 * d_bgr: packet-major bytes [nPackets][256][3].  This is what a render node returns per tile in the reference
 * (RGB8, src/node.cpp:336-349) and what the multi-GPU path gathers over xGMI. */
int snail_shade_depth_dev(const float *d_t, int nPackets, uint8_t *d_bgr, void *stream);
/* (the same with the arithmetic of Inv(t) named: SNAIL_ARITH_*; the call above = SNAIL_ARITH_IEEE.  This entry point takes no scene: in SNAIL_ARITH_HOST_SSE it
 * computes with a copy of the tables in force kept per DEVICE, made at the first such call on that device and again at the first one after
 * snail_arith_set_tables -- a synchronising refresh, which a stream capture cannot contain: there the call returns 3 unless snail_arith_prepare_device()
 * has been called on that device since the tables last changed.  Scene-bound launches never refresh anything: their tables are the scene's.) */
int snail_shade_depth_arith_dev(const float *d_t, int nPackets, uint8_t *d_bgr, int arith, void *stream);
/* Scatter packet-major BGR bytes into an interleaved rgb8 frame (pitch bytes per row), clipped to the image. */
int snail_packets_bgr_to_frame_dev(const int32_t *d_packet_xy, int nPackets, int resx, int resy, const uint8_t *d_bgr,
                                   uint8_t *d_frame, int pitch, void *stream);

/* The same scatter from a source cut into chunks: packet p of the list = entry p % nPerChunk of the chunk that starts (p / nPerChunk) *
 * chunkStrideBytes into d_bgr -- rank r's shard of a gathered buffer that holds several frames per rank (multi-frame launches). */
int snail_packets_bgr_to_frame_chunked_dev(const int32_t *d_packet_xy, int nPackets, int nPerChunk, int64_t chunkStrideBytes, int resx, int resy,
                                           const uint8_t *d_bgr, uint8_t *d_frame, int pitch, void *stream);

/* The render node's wire format for a tile and its inverse.  Encode = the `compress` store of RenderTask::Work
 * (src/render.cpp:140-163): tile (x, y, w, h) -> three w*h byte planes R, G-R, B-R (mod 256) at d_out + d_out_offsets[tile]
 * (the layout the reference hands to its LZ compressor; the compressor and the sockets are out of scope).  The packets of
 * tile k are d_first_packet[k], +1, ... in RenderTask::Work loop order (16-row bands outer, 16-column steps inner); d_bgr is
 * packet-major as produced by snail_shade_depth_dev / snail_render_whitted_dev.  The reference leaves tiles other than
 * 16x64 unwritten (a TODO, src/render.cpp:142-145); here every tile is written with the same formula.
 * Decode = DecompressTask::Work's plane loop (src/compression.cpp:112-141): pixel bytes (B,G,R) = (b'+r, g'+r, r) into an
 * interleaved frame of `pitch` bytes per row, clipped to resx x resy.  d_tiles: int32 [nTiles][4] = x, y, w, h. */
int snail_packets_bgr_to_planar_dev(const int32_t *d_tiles, const int32_t *d_first_packet, const int64_t *d_out_offsets, int nTiles,
                                    const uint8_t *d_bgr, uint8_t *d_out, void *stream);
int snail_planar_to_frame_dev(const int32_t *d_tiles, const int64_t *d_in_offsets, int nTiles, const uint8_t *d_planar, uint8_t *d_frame,
                              int pitch, int resx, int resy, void *stream);

/* Scene::RayTrace for primary packets in the reference's "simple shading" configuration, entirely on the device
 * (BASELINE config 3: primary + one shadow packet per light): TraversePrimary<1,0>; samples (position = d*t + o,
 * normal = triangle plane normal, diffuse = specular = color*|d.n|: src/scene_trace.cpp:359-452,
 * src/shading/simple_material.h:14-30); per light the packet-level cull BoxPointDistanceSq(bbox of the packet's hit
 * points) > radSq (src/scene_trace.cpp:494-501) and Scene::TraceLight (src/scene_trace.cpp:523-601): shadow-ray
 * generation, TraverseShadow, attenuation max(0, (1-d/r)*0.2 + Inv(16 (d/r)^2) - 0.0625), diffuse += N.L*atten,
 * specular += (N.L)^16*atten; outColor = diffuse*lDiffuse + specular*lSpecular; ConvColor -> B,G,R bytes.
 * lights7 (HOST pointer) = nLights (<= 8) x {pos[3], color[3], radius}; d_frame_bgr = resy rows of `pitch` bytes.
 * Shadow dir/idir of lanes the reference leaves uninitialised (misses; src/scene_trace.cpp:538-541) are zeros; they are
 * masked (distance = -inf) and cannot influence a result.  d_stats[2] += primary rays + shadow lanes with N.L > 0. */
#define SNAIL_MAX_LIGHTS 8
/* flags: SNAIL_WHITTED_REFLECTIONS = gVals[7], one bounce (src/scene_trace.cpp:454-466): Scene::TraceReflection (:603-618)
 * mirrors every hit ray about its normal (Reflect, src/rtbase_math.h:54-58; origin = hit point + 0.001 direction), traces the
 * packet as RayGroup<0,1> (per-ray origins, lane masks = hit lanes) through the same RayTrace -- samples, lights, shadow packets,
 * no further bounce -- and blends diffuse += (reflected colour - diffuse) * 0.3 before the primary's own lights.  Direction and
 * origin of lanes the mirrored packet masks off (no primary hit; uninitialised or non-finite in the reference) are zeros: such
 * a lane has distance -inf and is culled by every box test, so a finite value there cannot influence a result.
 * d_stats[2] += primary rays + mirrored lanes + shadow lanes with N.L > 0.  The frame is shaded in stages over packet-major
 * intermediates held by the scene handle (8 frames may be in flight on different streams).
 * Store: every pixel inside resx x resy gets its own colour.  (For a tile width that is not a multiple of 4 the reference's
 * interleaved store copies the FIRST pixel's colour into the other 1-2 pixels of the row's last quad and writes one byte past
 * them, src/render.cpp:183-188 -- an artefact of its 4-byte writes that is not reproduced; widths that are multiples of 4, which
 * is what its server demands (src/server.cpp:227-231), are byte-identical.) */
#define SNAIL_WHITTED_REFLECTIONS 1
int snail_render_whitted_dev(SnailScene *, const float cam[13], int resx, int resy, const float *lights7, int nLights,
                             const float ambient[3], const float color[3], int flags, uint8_t *d_frame_bgr, int pitch, uint64_t *d_stats,
                             void *stream);

/* The same frame with the dispatch-order feedback of the *_ordered_dev primary launches for EVERY walking stage of the pipeline (the
 * reference's tile task runs the whole Scene::RayTrace -- primary, reflection, shadow packets -- behind one dynamic queue,
 * src/render.cpp:258-267, src/thread_pool.cpp:151-180, so every stage of it is balanced): d_order (in) / d_slot_cost (out), either may be
 * NULL, are SNAIL_WHITTED_STAGES arrays of nSlots = snail_primary_slots(resx, resy) int32 each, back to back:
 *   stage 0  the primary packets                     (slot = the primary launch's slot)
 *   stage 1  the shadow packets of the primary hits  (same slots; the FIRST light's node visits)
 *   stage 2  the mirrored packets                    (slot = packet index cy * ceil(resx / 16) + cx; unused without SNAIL_WHITTED_REFLECTIONS)
 *   stage 3  the shadow packets of the mirrored hits (slots as stage 0; likewise)
 * Turn stage k's costs into its next order with snail_order_from_cost_dev(d_slot_cost + k * nSlots, nSlots, d_order + k * nSlots, stream).
 * Frame bytes and counters never depend on the orders. */
#define SNAIL_WHITTED_STAGES 4
int snail_render_whitted_ordered_dev(SnailScene *, const float cam[13], int resx, int resy, const float *lights7, int nLights,
                                     const float ambient[3], const float color[3], int flags, uint8_t *d_frame_bgr, int pitch, uint64_t *d_stats,
                                     const int32_t *d_order, int32_t *d_slot_cost, void *stream);

/* ... and with the next orders of all four stages derived inside the launch (see snail_trace_primary_batch_reorder_dev): d_next_order = SNAIL_WHITTED_STAGES
 * arrays of nSlots int32, back to back, as d_order; may be d_order itself. */
int snail_render_whitted_reorder_dev(SnailScene *, const float cam[13], int resx, int resy, const float *lights7, int nLights,
                                     const float ambient[3], const float color[3], int flags, uint8_t *d_frame_bgr, int pitch, uint64_t *d_stats,
                                     const int32_t *d_order, int32_t *d_slot_cost, int32_t *d_next_order, int order_flags, void *stream);

/* The same for an explicit list of packets (a rank's tiles): output = packet-major B,G,R bytes [nPackets][256][3] (4-byte aligned),
 * to be gathered and scattered with snail_packets_bgr_to_frame_dev -- a render node with the reference's simple shading on. */
int snail_render_whitted_packets_dev(SnailScene *, const float cam[13], int resx, int resy, const int32_t *d_packet_xy, int nPackets,
                                     const float *lights7, int nLights, const float ambient[3], const float color[3], int flags,
                                     uint8_t *d_bgr_packets, uint64_t *d_stats, void *stream);

/* Scene::TraceTransparency (src/scene_trace.cpp:620-634) for primary packets, staged on the device: the rays of the listed packets
 * continue behind their hits -- origin = dir * (t + 0.001) + origin, direction and reciprocal unchanged -- for the lanes of the
 * caller's selector (d_sel: 1 byte per quad, low 4 bits = lanes, packet-major; the reference's `transSel`, which its material system
 * fills: lanes whose hit material has fTransparency and opacity < 1, src/scene_trace.cpp:190,306,349,468-473; a lane without a hit is
 * dropped), are traced as RayGroup<0,1> and shaded by the nested RayTrace in the simple-shading configuration (samples, lights, shadow
 * packets; no reflection, no further transparency: with no shading data the nested call's own transSel is empty, :122-126,359-452).
 * d_t / d_triId = the packets' hit records (packet-major, e.g. from snail_trace_packets_dev); d_color = [nPackets][256][3] floats, the
 * reference's `transColor`, which the caller blends: diffuse = VLerp(transColor, diffuse, opacity) (:479-482).  d_stats[2] += selected
 * lanes + their shadow lanes with N.L > 0. */
int snail_trace_transparency_dev(SnailScene *, const float cam[13], int resx, int resy, const int32_t *d_packet_xy, int nPackets,
                                 const float *d_t, const int32_t *d_triId, const uint8_t *d_sel, const float *lights7, int nLights,
                                 const float ambient[3], const float color[3], float *d_color, uint64_t *d_stats, void *stream);

/* ---- the tile API (src/render.h:16-27) with host buffers ------------------------------------------------------------------ */
/* The two Render(...) shapes of the reference as plain-pointer entry points; include/snail_adapter.hpp wraps them in overloads with
 * the reference's exact signatures.  One call renders the WHOLE tile list / image on the device (same kernels as the *_dev entry
 * points) and returns after the bytes are in the caller's host buffer.
 *   snail_render_tiles = Render(scene, camera, resx, resy, data, coords, offsets, options, rank, threads) (src/render.h:16-19): per
 *     tile k = (x, y, w, h) = coords[4k..4k+3] every 16x16 packet (RenderTask::Work order, src/render.cpp:67-68) through
 *     Scene::RayTrace, stored as the three w*h byte planes R, G-R, B-R (mod 256) at data + offsets[k] (the `compress` store,
 *     src/render.cpp:140-163; the reference leaves tiles other than 16x64 unwritten -- a TODO at :142-145 -- here every tile is
 *     written).  `rank` (debug tint, gVals[8]) and `threads` have no device counterpart.
 *   snail_render_image = Render(scene, camera, image, options, threads) (src/render.h:21-23): the interleaved B,G,R store into an
 *     rgb8 image of `pitch` bytes per row (src/render.cpp:165-190, 214-240).
 * Shading: flags & SNAIL_RENDER_DEPTH = gVals[1] "very simple shading" (src/scene_trace.cpp:128-137; lights ignored); otherwise the
 * simple-shading configuration of snail_render_whitted_dev with the given lights (0..8), SNAIL_RENDER_REFLECTIONS = gVals[7].
 * stats (optional) += {intersects, iterations, traced rays, skips} of the call -- the TreeStats the reference's Render returns.
 * The device-side packet / tile lists are cached in the scene handle and rebuilt only when the tile list changes. */
#define SNAIL_RENDER_REFLECTIONS 1
#define SNAIL_RENDER_DEPTH       2
#define SNAIL_RENDER_AA4         4   /* gVals[9]: 4x antialiasing (src/render.cpp:60-62, :71-110): four double-resolution packets per packet, reduced 2x2 */
int snail_render_tiles(SnailScene *, const float cam[13], int resx, int resy, const int32_t *coords, const int64_t *offsets, int nTiles,
                       const float *lights7, int nLights, const float ambient[3], const float color[3], int flags, uint8_t *data,
                       uint64_t stats[4]);
/* snail_render_tiles over SEVERAL devices of this process: scenes[d] = the same tree uploaded to device d (snail_scene_create once per
 * device).  The reference's server deals a frame's tiles to its render nodes by shuffled round-robin and collects their buffers
 * (src/server.cpp:233-265, :389-401); here the tiles are dealt the same way to the devices, every device renders its share at the same
 * time, and the shares are written to `data` (host memory, as for the reference's server) -- bytes and summed counters identical to one
 * snail_render_tiles call over the whole list.  The deal depends on (nTiles, nScenes) only, so each device's lists stay cached. */
int snail_render_tiles_multi(SnailScene *const *scenes, int nScenes, const float cam[13], int resx, int resy, const int32_t *coords,
                             const int64_t *offsets, int nTiles, const float *lights7, int nLights, const float ambient[3],
                             const float color[3], int flags, uint8_t *data, uint64_t stats[4]);
int snail_render_image(SnailScene *, const float cam[13], int resx, int resy, const float *lights7, int nLights, const float ambient[3],
                       const float color[3], int flags, uint8_t *image_bgr, int pitch, uint64_t stats[4]);

/* ---- measurement support ------------------------------------------------------------------------- */
/* Single-ray, cache-less accounting walk of SURVEY.md section 8(d) over the same padded packet set as
 * snail_trace_primary: d_out[0] += rays, [1] += sum V_n (node boxes tested), [2] += sum V_t
 * (triangles tested), [3] += hits.  Used by bench.py to price algorithmic bytes:
 * B_alg(ray) = 32*V_n + 64*V_t + 16. */
int snail_account_primary(SnailScene *, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, uint64_t out[4]);

/* Per-packet accounting of one full-frame primary launch (the same packet code built with counters, dev::k_primary_diag; a measurement
 * pass, never on a product path), row-major over the packet grid, 8 words per packet:
 * out8[p*8 + {0..5}] = {loop iterations (node visits), quad x triangle tests, shader-clock cycles of that wavefront, start time >> 6,
 * triangle records fetched (every triangle of every leaf body entered), leaf bodies entered}; words 6, 7 = 0.
 * bench.py's packet-level algorithmic bytes come from it (32 B x node visits + 64 B x triangle records fetched + 16 B x 256 per
 * packet), and the load-balance studies of tools/packet_costs.py. */
int snail_account_packets(SnailScene *, const float cam[13], int resx, int resy, uint32_t *out8);

/* Diagnostics, experiments and environment switches (snail_debug_*) are NOT part of this contract: they live in
 * include/snail_hip_debug.h and exist only in the workbench build libsnailhip_debug.so (-DSNAIL_DEBUG_API). */

/* Launch geometry of the last primary launch on this scene (for profiles): waves, blocks, VGPR-independent. */
int snail_last_launch(const SnailScene *, int *blocks, int *threadsPerBlock);

#ifdef __cplusplus
}
#endif
#endif
