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
//
// Errors: the C-ABI returns status codes; the reference aborts (FATAL -> FWK_FATAL, src/rtbase.h:13).
// SNAIL_CHECK keeps the reference's behaviour.
//
// The adapter is written against the member names of the reference's types only (it is a template over
// them), so that this repository does not need -- and does not contain -- any reference header.
#pragma once

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
	// rays of edge packets that fall outside the image are traced on the device (they are part of their
	// packet, src/render.cpp:67-68) but never stored by the reference; they are handed back as misses
	std::vector<float> rt((size_t)resx * resy), ru((size_t)resx * resy), rv((size_t)resx * resy);
	std::vector<int32_t> ri((size_t)resx * resy);
	out.stats[0] = out.stats[1] = out.stats[2] = out.stats[3] = 0;
	SNAIL_CHECK(snail_trace_primary(scene, c, resx, resy, 0, 0, resx, resy, rt.data(), ru.data(), rv.data(), ri.data(), out.stats));
	const size_t np = (size_t)out.pw * out.ph;
	out.t.assign(np * 256, 1.0f / 0.0f); out.u.assign(np * 256, 0.0f); out.v.assign(np * 256, 0.0f); out.triId.assign(np * 256, 0);
	for(int y = 0; y < resy; y++)
		for(int x = 0; x < resx; x++) {
			// quad ty*4+k, lane j  <->  pixel (px + 4k + j, py + ty)   (src/ray_generator.cpp:29-45)
			const size_t p = out.packetIndex(x, y), q = (size_t)(y & 15) * 4 + (size_t)((x & 15) >> 2), l = (size_t)(x & 3);
			const size_t d = p * 256 + q * 4 + l, s = (size_t)y * resx + x;
			out.t[d] = rt[s]; out.u[d] = ru[s]; out.v[d] = rv[s]; out.triId[d] = ri[s];
		}
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
	~HipBVH() { if(scene) snail_scene_destroy(scene); }
	HipBVH(const HipBVH &) = delete;
	HipBVH &operator=(const HipBVH &) = delete;

	// Call after RefBVH::Construct (the SAH build stays on the host, src/bvh/tree.cpp:293-328).
	void Upload(const RefBVH &bvh, int device = 0) {
		ref = &bvh;
		if(scene) snail_scene_destroy(scene);
		static_assert(sizeof(bvh.nodes[0]) == 32 && sizeof(bvh.tris[0]) == 64, "record sizes are part of the ABI");
		scene = snail_scene_create(bvh.nodes.data(), (int)bvh.nodes.size(), bvh.tris.data(), (int)bvh.tris.size(), bvh.depth, device);
		if(!scene) { std::fprintf(stderr, "FATAL: snail_scene_create: %s\n", snail_last_error()); std::abort(); }
	}

	// ---- frame prefetch ----
	template <class CameraT> void BeginFrame(const CameraT &cam, int resx, int resy) { TraceFrame(scene, cam, resx, resy, frame); haveFrame = true; }
	void EndFrame() { haveFrame = false; }
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
	const RefBVH *ref = nullptr;
	SnailScene *scene = nullptr;
	FrameHits frame;
	bool haveFrame = false;
	static inline thread_local long curPacket = -1;   // per render thread (thread_pool workers, src/thread_pool.cpp)
};

} // namespace snail
