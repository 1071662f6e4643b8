// snail_adapter.hpp -- header-only C++ adapter: the reference's own shapes on top of the C-ABI.
//
// Include this from a translation unit of the reference AFTER its own headers ("bvh/tree.h",
// "scene.h", "render.h", "camera.h") and link with -lsnailhip.  It provides
//
//   class snail::HipBVH            -- a model of the `AccStruct` concept that `template <class AccStruct>
//                                     class Scene` requires (src/scene.h:26-58; the concept's members are
//                                     listed at src/bvh/tree.h:27-50,82-84): TraversePrimary<so,mask>,
//                                     TraverseShadow, GetNormal, GetSElement, GetMaterialId, HasShadingData,
//                                     GetBBox, isctFlags, CElement/SElement.
//   snail::TraceFrame(...)         -- frame-granular primary tracing into packet-major hit records, the
//                                     accelerator seam the Cell port uses (`TaskInfo`, src/spu/trace.h:34-63).
//   snail::RenderTiles / RenderImage -- the tile API of src/render.h:16-27 on the device pipeline (snail_render_tiles /
//                                     snail_render_image): tile list -> planar R, G-R, B-R bytes at data + offsets[k]; image -> rgb8.
//                                     With SNAIL_ADAPTER_RENDER_OVERLOADS defined before this header is included (in a translation
//                                     unit of the reference, after "render.h"), `Render(...)` overloads with EXACTLY the reference's two
//                                     signatures are declared for Scene<snail::HipBVH<...>>: more specialised than the reference's
//                                     templates, so existing call sites (src/node.cpp:336-338, src/rtracer.cpp:385-386) pick them.
//   snail::ShadowBatch / RayBatch  -- the batched form of the immediate path: collect the shadow / secondary packets of a tile,
//                                     trace them with ONE snail_trace_shadow / snail_trace_rays call (nPackets), results copied back.
//
// Why two levels: one TraversePrimary(Context&) call carries 256 rays; a GPU launch per packet would be
// launch-latency bound.  HipBVH therefore works in two modes:
//   * prefetched: the host calls BeginFrame(camera, resx, resy) once; the whole frame's primary packets are
//     traced by ONE kernel launch; TraversePrimary<1,0>(ctx) of a packet registered with SetPacket(x, y)
//     only copies that packet's 256 hit records into ctx (the reference's quad order is the library's
//     packet-major order, so this is four memcpy calls);
//   * immediate: every other call (secondary packets <0,*>, un-prefetched primaries, shadow packets) goes
//     to snail_trace_rays / snail_trace_shadow synchronously -- correct, and intended to be batched by a
//     host that cares (collect the packets of a tile, call once; the C-ABI takes nPackets).
// Threads: the reference's Render(..., threads) runs RenderTask::Work on `threads` pthread workers over ONE const scene
// (src/render.cpp:214-267, src/thread_pool.cpp:151-180); every const member below may be called from any number of them at once -- the
// C-ABI handle is thread-safe (include/snail_hip.h, "Concurrency"), the announced packet is per thread, the frame's TreeStats go to exactly one
// caller.  BeginFrame / EndFrame / Upload / SetArith belong to the thread that owns the frame loop.
//
// Errors: the C-ABI returns status codes; the reference aborts (FATAL -> FWK_FATAL, src/rtbase.h:13).
// SNAIL_CHECK keeps the reference's behaviour.
//
// The adapter is written against the member names of the reference's types only (it is a template over
// them), so that this repository does not need -- and does not contain -- any reference header.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "snail_hip.h"

#ifndef SNAIL_CHECK
#define SNAIL_CHECK(expr)                                                                              \
	do {                                                                                               \
		int rc_ = (expr);                                                                              \
		if(rc_ != 0) {                                                                                 \
			std::fprintf(stderr, "FATAL: %s -> %d: %s\n", #expr, rc_, snail_last_error());            \
			std::abort();                                                                              \
		}                                                                                              \
	} while(0)
#endif

namespace snail {

// Packet-major hit records of one frame: element [packet*256 + quad*4 + lane].
struct FrameHits {
	int resx = 0, resy = 0, pw = 0, ph = 0; // packet grid
	std::vector<float> t, u, v;
	std::vector<int32_t> triId;
	uint64_t stats[4] = {0, 0, 0, 0};
	size_t packetIndex(int x, int y) const { return (size_t)(y / 16) * pw + (size_t)(x / 16); }
};

// Trace all primary packets of a frame.  `CameraT` is the reference's `Camera` (src/camera.h:7-14).
template <class CameraT>
inline void TraceFrame(SnailScene *scene, const CameraT &cam, int resx, int resy, FrameHits &out) {
	const float c[13] = {cam.pos.x, cam.pos.y, cam.pos.z, cam.right.x, cam.right.y, cam.right.z, cam.up.x, cam.up.y, cam.up.z,
						 cam.front.x, cam.front.y, cam.front.z, cam.plane_dist};
	out.resx = resx; out.resy = resy;
	out.pw = (resx + 15) / 16; out.ph = (resy + 15) / 16;
	// ONE launch, outputs already in the reference's packet-major quad order (src/ray_generator.cpp:29-45): no per-pixel work on the CPU
	const size_t np = (size_t)out.pw * out.ph;
	out.t.resize(np * 256); out.u.resize(np * 256); out.v.resize(np * 256); out.triId.resize(np * 256);
	out.stats[0] = out.stats[1] = out.stats[2] = out.stats[3] = 0;
	SNAIL_CHECK(snail_trace_frame_packets(scene, c, resx, resy, out.t.data(), out.u.data(), out.v.data(), out.triId.data(), out.stats));
}

// A model of the AccStruct concept backed by libsnailhip.  `RefBVH` is the reference's `BVH`
// (src/bvh/tree.h:27-93): its public `nodes`, `tris`, `shTris`, `materials`, `depth` are used as they are.
template <class RefBVH>
class HipBVH {
public:
	typedef typename RefBVH::CElement CElement;
	typedef typename RefBVH::SElement SElement;
	enum { isComplex = 0 };
	enum { isctFlags = RefBVH::isctFlags };
	enum { maxDepth = RefBVH::maxDepth };

	HipBVH() = default;
	~HipBVH() { Release(); }
	HipBVH(const HipBVH &) = delete;
	HipBVH &operator=(const HipBVH &) = delete;

	// Call after RefBVH::Construct (the SAH build stays on the host, src/bvh/tree.cpp:293-328).
	void Upload(const RefBVH &bvh, int device = 0) { Upload(bvh, std::vector<int>(1, device)); }
	// One copy of the tree per listed device (the reference's server ships the same arrays to every render node, src/server.cpp:144-164):
	// the tile-list renderer then deals a frame's tiles over all of them (snail::RenderTiles -> snail_render_tiles_multi); every other
	// path (prefetched frame, immediate and batched packets, image renderer) runs on the first device.
	void Upload(const RefBVH &bvh, const std::vector<int> &devices) {
		if(devices.empty()) { std::fprintf(stderr, "FATAL: HipBVH::Upload: no device\n"); std::abort(); }
		ref = &bvh;
		Release();
		static_assert(sizeof(bvh.nodes[0]) == 32 && sizeof(bvh.tris[0]) == 64, "record sizes are part of the ABI");
		for(int device : devices) {
			SnailScene *h = snail_scene_create(bvh.nodes.data(), (int)bvh.nodes.size(), bvh.tris.data(), (int)bvh.tris.size(), bvh.depth, device);
			if(!h) { std::fprintf(stderr, "FATAL: snail_scene_create(device %d): %s\n", device, snail_last_error()); std::abort(); }
			handles.push_back(h);
		}
		scene = handles[0];
		// a re-upload keeps the arithmetic the host had selected (fresh handles start in SNAIL_ARITH_IEEE)
		if(arith != SNAIL_ARITH_IEEE && !SetArith(arith)) { std::fprintf(stderr, "FATAL: HipBVH::Upload: SetArith(%d): %s\n", arith, snail_last_error()); std::abort(); }
	}
	int DeviceCount() const { return (int)handles.size(); }
	SnailScene *const *Handles() const { return handles.data(); }
	// Arithmetic of Inv / RSqrt / FastInv on every device copy (include/snail_hip.h): SNAIL_ARITH_HOST_SSE = what the reference's SSE build
	// computes on THIS host (its rcpps / rsqrtps + veclib's Newton steps), bit for bit; returns false -- every handle's arithmetic stays what it
	// was before the call -- when the host's instructions cannot be reproduced from tables (snail_last_error() says why).  Remembered: a later
	// Upload() applies it to the new handles.
	bool SetArith(int a) {
		std::vector<int> before(handles.size(), SNAIL_ARITH_IEEE);
		for(size_t k = 0; k < handles.size(); k++) (void)snail_scene_arith(handles[k], &before[k]);
		for(size_t k = 0; k < handles.size(); k++)
			if(snail_scene_set_arith(handles[k], a) != 0) {
				for(size_t g = 0; g < k; g++) (void)snail_scene_set_arith(handles[g], before[g]);   // (handles[k] itself was not changed)
				return false;
			}
		arith = a;
		return true;
	}
	int Arith() const { return arith; }

	// ---- frame prefetch ----
	// (const over mutable state: the prefetched frame is a cache of what TraversePrimary would compute, and the reference's renderers
	// take the scene by const reference, src/render.h:16-23)
	template <class CameraT> void BeginFrame(const CameraT &cam, int resx, int resy) const {
		TraceFrame(scene, cam, resx, resy, frame);
		haveFrame = true;
		statsPending.store(true);   // the frame's TreeStats (one launch = one total) go to the first packet that is copied out
	}
	void EndFrame() const { haveFrame = false; }
	bool HaveFrame() const { return haveFrame; }
	// the host's packet loop (RenderTask::Work, src/render.cpp:67-68) announces the packet it is about to trace
	void SetPacket(int x, int y) const { curPacket = haveFrame ? (long)frame.packetIndex(x, y) : -1; }

	// ---- AccStruct concept ----
	bool HasShadingData() const { return ref->HasShadingData(); }
	const SElement &GetSElement(int elem, int sub) const { return ref->GetSElement(elem, sub); }
	auto GetNormal(int elem, int sub) const { return ref->GetNormal(elem, sub); }
	int GetMaterialId(int idx, int sub) const { return ref->GetMaterialId(idx, sub); }
	auto GetBBox() const { return ref->GetBBox(); }

	template <class ContextT> void TraversePrimary(ContextT &c) const {
		using RayGroupT = decltype(c.rays);                 // RayGroup<sharedOrigin, hasMask>, src/ray_group.h:74-160
		constexpr bool shared = RayGroupT::sharedOrigin != 0, masked = RayGroupT::hasMask != 0;
		const int size = c.Size();
		if(shared && !masked && size == SNAIL_PACKET_QUADS && curPacket >= 0) {
			const size_t o = (size_t)curPacket * 256;
			std::memcpy(c.distance, &frame.t[o], 256 * 4);
			std::memcpy(c.object, &frame.triId[o], 256 * 4);
			float *b = (float *)c.barycentric; // Vec2q = {u[4], v[4]} per quad
			for(int q = 0; q < 64; q++) { std::memcpy(b + q * 8, &frame.u[o + q * 4], 16); std::memcpy(b + q * 8 + 4, &frame.v[o + q * 4], 16); }
			curPacket = -1; // one primary traversal per announced packet; later calls are secondary packets
			// the reference only ever SUMS per-packet TreeStats (src/render.cpp:236-238); the launch counted the whole frame at once
			if(c.stats && statsPending.exchange(false)) {
				c.stats->Intersection((unsigned)frame.stats[0]); c.stats->LoopIteration((unsigned)frame.stats[1]); c.stats->Skip((unsigned)frame.stats[3]);
			}
			return;
		}
		uint64_t st[4] = {0, 0, 0, 0};
		SNAIL_CHECK(snail_trace_rays(scene, 1, size, shared ? 1 : 0, (const float *)c.rays.OriginPtr(), (const float *)c.rays.DirPtr(),
									 (const float *)c.rays.IDirPtr(), (const uint8_t *)c.MaskPtr(), (float *)c.distance, (int32_t *)c.object,
									 (float *)c.barycentric, st));
		if(c.stats) { c.stats->Intersection((unsigned)st[0]); c.stats->LoopIteration((unsigned)st[1]); c.stats->Skip((unsigned)st[3]); }
	}

	template <class ShadowContextT> void TraverseShadow(ShadowContextT &c) const {
		const float *o = (const float *)c.rays.OriginPtr(); // Vec3q: x[4], y[4], z[4]
		const float org[3] = {o[0], o[4], o[8]};
		uint64_t st[4] = {0, 0, 0, 0};
		SNAIL_CHECK(snail_trace_shadow(scene, 1, c.Size(), org, (const float *)c.rays.DirPtr(), (const float *)c.rays.IDirPtr(), (float *)c.distance, st));
		if(c.stats) { c.stats->Intersection((unsigned)st[0]); c.stats->LoopIteration((unsigned)st[1]); c.stats->Skip((unsigned)st[3]); }
	}

	SnailScene *Handle() const { return scene; }
	const FrameHits &Frame() const { return frame; }

private:
	void Release() {
		for(SnailScene *h : handles) snail_scene_destroy(h);
		handles.clear(); scene = nullptr;
	}
	const RefBVH *ref = nullptr;
	int arith = SNAIL_ARITH_IEEE;         // what SetArith selected last (re-applied by Upload)
	SnailScene *scene = nullptr;          // = handles[0]
	std::vector<SnailScene *> handles;    // one per device
	mutable FrameHits frame;
	mutable bool haveFrame = false;
	mutable std::atomic<bool> statsPending{false};
	static inline thread_local long curPacket = -1;   // per render thread (thread_pool workers, src/thread_pool.cpp)
};

// ---- batched immediate path --------------------------------------------------------------------------------------------------
// One synchronous launch per 256-ray packet is launch-latency bound (and slower than the CPU the moment a light is on): a host
// collects the shadow packets of a tile (one per primary packet and light, src/scene_trace.cpp:560-563) and traces them at once.
// Add() copies the packet's inputs; Flush() makes ONE snail_trace_shadow call over all of them and writes every packet's
// distances back through the pointer its context held (the contexts' arrays must stay alive until Flush()).
class ShadowBatch {
public:
	explicit ShadowBatch(int packetQuads = SNAIL_PACKET_QUADS) : size(packetQuads) {}
	template <class ShadowContextT> void Add(ShadowContextT &c) {
		if(c.Size() != size) { std::fprintf(stderr, "FATAL: ShadowBatch: packet of %d quads in a batch of %d-quad packets\n", c.Size(), size); std::abort(); }
		const float *o = (const float *)c.rays.OriginPtr();
		origin.insert(origin.end(), {o[0], o[4], o[8]});
		append(dir, (const float *)c.rays.DirPtr(), (size_t)size * 12);
		append(idir, (const float *)c.rays.IDirPtr(), (size_t)size * 12);
		append(dist, (const float *)c.distance, (size_t)size * 4);
		out.push_back((float *)c.distance);
	}
	// returns {intersects, iterations, 0, skips} of the batch
	template <class AccT, class StatsT> void Flush(const AccT &acc, StatsT *stats) {
		if(out.empty()) return;
		uint64_t st[4] = {0, 0, 0, 0};
		SNAIL_CHECK(snail_trace_shadow(acc.Handle(), (int)out.size(), size, origin.data(), dir.data(), idir.data(), dist.data(), st));
		for(size_t p = 0; p < out.size(); p++) std::memcpy(out[p], &dist[p * (size_t)size * 4], (size_t)size * 16);
		if(stats) { stats->Intersection((unsigned)st[0]); stats->LoopIteration((unsigned)st[1]); stats->Skip((unsigned)st[3]); }
		origin.clear(); dir.clear(); idir.clear(); dist.clear(); out.clear();
	}
	size_t Packets() const { return out.size(); }

private:
	static void append(std::vector<float> &v, const float *p, size_t n) { v.insert(v.end(), p, p + n); }
	int size;
	std::vector<float> origin, dir, idir, dist;
	std::vector<float *> out;
};

// The same for secondary packets with per-ray origins and lane masks -- RayGroup<0,1>: reflections and transparency
// (src/scene_trace.cpp:615-617, :631-633) -- or any other <sharedOrigin, hasMask> combination (one combination per batch).
class RayBatch {
public:
	RayBatch(bool sharedOrigin, bool hasMask, int packetQuads = SNAIL_PACKET_QUADS) : shared(sharedOrigin), masked(hasMask), size(packetQuads) {}
	template <class ContextT> void Add(ContextT &c) {
		using RayGroupT = decltype(c.rays);
		if((RayGroupT::sharedOrigin != 0) != shared || (RayGroupT::hasMask != 0) != masked || c.Size() != size) {
			std::fprintf(stderr, "FATAL: RayBatch: context does not match the batch's <sharedOrigin, hasMask, size>\n"); std::abort();
		}
		append(origin, (const float *)c.rays.OriginPtr(), shared ? 12 : (size_t)size * 12);
		append(dir, (const float *)c.rays.DirPtr(), (size_t)size * 12);
		append(idir, (const float *)c.rays.IDirPtr(), (size_t)size * 12);
		if(masked) { const uint8_t *m = (const uint8_t *)c.MaskPtr(); mask.insert(mask.end(), m, m + size); }
		append(dist, (const float *)c.distance, (size_t)size * 4);
		const int32_t *ob = (const int32_t *)c.object; obj.insert(obj.end(), ob, ob + (size_t)size * 4);
		append(bary, (const float *)c.barycentric, (size_t)size * 8);
		outDist.push_back((float *)c.distance); outObj.push_back((int32_t *)c.object); outBary.push_back((float *)c.barycentric);
	}
	template <class AccT, class StatsT> void Flush(const AccT &acc, StatsT *stats) {
		if(outDist.empty()) return;
		uint64_t st[4] = {0, 0, 0, 0};
		SNAIL_CHECK(snail_trace_rays(acc.Handle(), (int)outDist.size(), size, shared ? 1 : 0, origin.data(), dir.data(), idir.data(), masked ? mask.data() : nullptr,
									 dist.data(), obj.data(), bary.data(), st));
		for(size_t p = 0; p < outDist.size(); p++) {
			std::memcpy(outDist[p], &dist[p * (size_t)size * 4], (size_t)size * 16);
			std::memcpy(outObj[p], &obj[p * (size_t)size * 4], (size_t)size * 16);
			std::memcpy(outBary[p], &bary[p * (size_t)size * 8], (size_t)size * 32);
		}
		if(stats) { stats->Intersection((unsigned)st[0]); stats->LoopIteration((unsigned)st[1]); stats->Skip((unsigned)st[3]); }
		origin.clear(); dir.clear(); idir.clear(); mask.clear(); dist.clear(); obj.clear(); bary.clear(); outDist.clear(); outObj.clear(); outBary.clear();
	}
	size_t Packets() const { return outDist.size(); }

private:
	static void append(std::vector<float> &v, const float *p, size_t n) { v.insert(v.end(), p, p + n); }
	bool shared, masked;
	int size;
	std::vector<float> origin, dir, idir, dist, bary;
	std::vector<uint8_t> mask;
	std::vector<int32_t> obj;
	std::vector<float *> outDist, outBary;
	std::vector<int32_t *> outObj;
};

// ---- the tile API of src/render.h:16-27 on the device pipeline -----------------------------------------------------------------
// What of Scene<AccStruct> / the global switches the device pipeline honours: scene.lights (pos, color, radius: src/light.h:5-16),
// scene.ambientLight, the default material's colour (Scene::Scene sets (1,1,1), src/scene.cpp:6-10; SimpleMaterial::color is private,
// hence a parameter), gVals[1] (depth shading), gVals[7] (one mirrored bounce) and gVals[9] (4x antialiasing, src/render.cpp:60-62, :71-110).  Textured materials / full shading data (gVals[6])
// stay with the host path (HipBVH under the reference's own Render).
// The reference's renderer reads more global switches than that (src/rtbase.cpp:16, toggled by F-keys, broadcast to the render nodes
// every frame: src/server.cpp:372-376).  Those the device pipeline does NOT implement -- a call with one of them set must not come back
// as a plain frame with status 0: the Render(...) overloads below hand such a call to the reference's own renderer over the
// prefetched HipBVH path (the reference's RenderTask::Work then does its 4x antialiasing, full shading, tints ... itself, and its
// TraversePrimary calls copy pre-traced packets), or abort with this message when SNAIL_ADAPTER_NO_HOST_RENDERER is defined.
//   gVals[6]  full shading (materials, textures, transparency selection) when the scene carries shading data   src/scene_trace.cpp:145
//   gVals[5]  TreeStats visualisation   src/scene_trace.cpp:513-517
//   gVals[8]  per-rank tint of the render nodes' tiles (colorizeNodes: the tile-list Render only)   src/render.cpp:118-132
inline const char *UnsupportedSwitch(const int *gv, bool hasShadingData, bool tileList) {
	if(gv[6] && hasShadingData) return "gVals[6] (full shading: materials / textures, src/scene_trace.cpp:145)";
	if(gv[5]) return "gVals[5] (TreeStats visualisation, src/scene_trace.cpp:513-517)";
	if(gv[8] && tileList) return "gVals[8] (per-rank tint of the tiles, src/render.cpp:118-132)";
	return nullptr;
}

struct RenderMode {
	bool depthShading = false;  // gVals[1]
	bool reflections = false;   // gVals[7]
	bool antialias = false;     // gVals[9]
	float color[3] = {1.0f, 1.0f, 1.0f};
};

namespace detail {
template <class CameraT> inline void cam13(const CameraT &cam, float (&c)[13]) {
	const float v[13] = {cam.pos.x, cam.pos.y, cam.pos.z, cam.right.x, cam.right.y, cam.right.z, cam.up.x, cam.up.y, cam.up.z,
						 cam.front.x, cam.front.y, cam.front.z, cam.plane_dist};
	std::memcpy(c, v, sizeof(v));
}
template <class SceneT> inline std::vector<float> lights7(const SceneT &scene) {
	std::vector<float> l;
	for(const auto &li : scene.lights) l.insert(l.end(), {li.pos.x, li.pos.y, li.pos.z, li.color.x, li.color.y, li.color.z, li.radius});
	if(l.size() > (size_t)SNAIL_MAX_LIGHTS * 7) { std::fprintf(stderr, "FATAL: snail::Render: %zu lights, the device pipeline takes %d\n", l.size() / 7, SNAIL_MAX_LIGHTS); std::abort(); }
	return l;
}
template <class StatsT> inline StatsT toStats(const uint64_t (&st)[4]) {
	StatsT s;
	s.Intersection((unsigned)st[0]); s.LoopIteration((unsigned)st[1]); s.TracingRays((unsigned)st[2]); s.Skip((unsigned)st[3]);
	return s;
}
} // namespace detail

// Render(scene, camera, resx, resy, data, coords, offsets, ...) of src/render.h:16-19: coords = x, y, w, h per tile, offsets = byte
// offset of each tile's three planes (R, G-R, B-R; 3*w*h bytes) in `data`.  Returns the summed TreeStats.
template <class StatsT, class SceneT, class CameraT>
inline StatsT RenderTiles(const SceneT &scene, const CameraT &camera, unsigned resx, unsigned resy, unsigned char *data, const std::vector<int> &coords,
						  const std::vector<int> &offsets, const RenderMode &mode = RenderMode()) {
	float c[13];
	detail::cam13(camera, c);
	const std::vector<float> l = detail::lights7(scene);
	std::vector<int64_t> off(offsets.begin(), offsets.end());
	const float amb[3] = {scene.ambientLight.x, scene.ambientLight.y, scene.ambientLight.z};
	uint64_t st[4] = {0, 0, 0, 0};
	const int flags = (mode.depthShading ? SNAIL_RENDER_DEPTH : 0) | (mode.reflections ? SNAIL_RENDER_REFLECTIONS : 0) | (mode.antialias ? SNAIL_RENDER_AA4 : 0);
	if(scene.geometry.DeviceCount() > 1)   // the tiles dealt over every device the tree was uploaded to
		SNAIL_CHECK(snail_render_tiles_multi(scene.geometry.Handles(), scene.geometry.DeviceCount(), c, (int)resx, (int)resy, coords.data(), off.data(), (int)(coords.size() / 4),
											 l.data(), (int)(l.size() / 7), amb, mode.color, flags, data, st));
	else
		SNAIL_CHECK(snail_render_tiles(scene.geometry.Handle(), c, (int)resx, (int)resy, coords.data(), off.data(), (int)(coords.size() / 4), l.data(), (int)(l.size() / 7), amb,
									   mode.color, flags, data, st));
	return detail::toStats<StatsT>(st);
}

// Render(scene, camera, image, ...) of src/render.h:21-23; ImageT = MipmapTexture (Width, Height, Pitch, DataPointer; rgb8).
template <class StatsT, class SceneT, class CameraT, class ImageT>
inline StatsT RenderImage(const SceneT &scene, const CameraT &camera, ImageT &image, const RenderMode &mode = RenderMode()) {
	float c[13];
	detail::cam13(camera, c);
	const std::vector<float> l = detail::lights7(scene);
	const float amb[3] = {scene.ambientLight.x, scene.ambientLight.y, scene.ambientLight.z};
	uint64_t st[4] = {0, 0, 0, 0};
	SNAIL_CHECK(snail_render_image(scene.geometry.Handle(), c, (int)image.Width(), (int)image.Height(), l.data(), (int)(l.size() / 7), amb, mode.color,
								   (mode.depthShading ? SNAIL_RENDER_DEPTH : 0) | (mode.reflections ? SNAIL_RENDER_REFLECTIONS : 0) | (mode.antialias ? SNAIL_RENDER_AA4 : 0),
								   (unsigned char *)image.DataPointer(), (int)image.Pitch(), st));
	return detail::toStats<StatsT>(st);
}

} // namespace snail

// ---- overloads with the reference's own signatures (src/render.h:16-23) ---------------------------------------------------------
// Define SNAIL_ADAPTER_RENDER_OVERLOADS before including this header in a translation unit that has seen the reference's "render.h"
// (Scene, Camera, TreeStats, Options, MipmapTexture, vector, uint, gVals are the reference's names).  Both are more specialised than
// the reference's `template <class AccStruct> TreeStats Render(const Scene<AccStruct>&, ...)`, so a call with a
// Scene<snail::HipBVH<BVH>> resolves here and the whole tile list / image is rendered by the device pipeline.
#ifdef SNAIL_ADAPTER_RENDER_OVERLOADS
// A call the device pipeline cannot honour (snail::UnsupportedSwitch): the reference's own renderer does the frame, fed by ONE prefetch
// launch -- Render<snail::HipBVH<RefBVH>>(...) with the template argument spelled out names the reference's generic template
// (src/render.h:16-23; instantiate it for snail::HipBVH<BVH> beside src/render.cpp:283-287), whose RenderTask::Work
// announces each packet with geometry.SetPacket(x, y).  With 4x antialiasing the rays are generated for twice the resolution
// (src/render.cpp:60-62), so that is the frame that is prefetched.
#ifdef SNAIL_ADAPTER_NO_HOST_RENDERER
#define SNAIL_HOST_RENDER(why, scale, w, h, call)                                                                                           \
	do { std::fprintf(stderr, "FATAL: snail Render(): %s is set and the device pipeline does not implement it (and SNAIL_ADAPTER_NO_HOST_RENDERER)\n", why); std::abort(); } while(0)
#else
#define SNAIL_HOST_RENDER(why, scale, w, h, call)                                                                                           \
	do {                                                                                                                                   \
		scene.geometry.BeginFrame(camera, (int)(w) * (scale), (int)(h) * (scale));                                                         \
		const TreeStats st_ = call;                                                                                                        \
		scene.geometry.EndFrame();                                                                                                         \
		return st_;                                                                                                                        \
	} while(0)
#endif
template <class RefBVH>
inline TreeStats Render(const Scene<snail::HipBVH<RefBVH>> &scene, const Camera &camera, uint resx, uint resy, unsigned char *data, const vector<int> &coords,
						const vector<int> &offsets, const Options options, uint rank, uint threads) {
	if(const char *why = snail::UnsupportedSwitch(gVals, scene.geometry.HasShadingData(), true))
		SNAIL_HOST_RENDER(why, gVals[9] ? 2 : 1, resx, resy, Render<snail::HipBVH<RefBVH>>(scene, camera, resx, resy, data, coords, offsets, options, rank, threads));
	(void)rank; (void)threads; // (no tint requested; the host thread pool has no device counterpart)
	snail::RenderMode mode;
	mode.depthShading = gVals[1] != 0;
	mode.reflections = gVals[7] != 0;   // the bounce is gated by gVals[7] alone (src/scene_trace.cpp:454); Options::reflections is stored and never read (src/render.cpp:24,37)
	(void)options;
	mode.antialias = gVals[9] != 0;
	return snail::RenderTiles<TreeStats>(scene, camera, resx, resy, data, coords, offsets, mode);
}
template <class RefBVH>
inline TreeStats Render(const Scene<snail::HipBVH<RefBVH>> &scene, const Camera &camera, MipmapTexture &image, const Options options, uint threads) {
	if(const char *why = snail::UnsupportedSwitch(gVals, scene.geometry.HasShadingData(), false))
		SNAIL_HOST_RENDER(why, gVals[9] ? 2 : 1, image.Width(), image.Height(), Render<snail::HipBVH<RefBVH>>(scene, camera, image, options, threads));
	(void)threads;
	snail::RenderMode mode;
	mode.depthShading = gVals[1] != 0;
	mode.reflections = gVals[7] != 0;   // the bounce is gated by gVals[7] alone (src/scene_trace.cpp:454); Options::reflections is stored and never read (src/render.cpp:24,37)
	(void)options;
	mode.antialias = gVals[9] != 0;
	return snail::RenderImage<TreeStats>(scene, camera, image, mode);
}
#undef SNAIL_HOST_RENDER
#endif
