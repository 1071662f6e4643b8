// snail_hip.hip -- gfx950 (MI355X / CDNA4) kernels and C-ABI of libsnailhip.so.
//
// Hot path of nadult/Snail re-designed for wave64:
//
//   ONE 16x16 RAY PACKET = ONE WAVEFRONT, ONE SSE QUAD (4 rays) = ONE LANE.
//
// The reference walks the BVH once per packet of 64 quads and keeps packet-wide state: a stack of
// (node, firstActive, lastActive), a child order taken from lane 0 of quad 0, whole-packet interval
// culls, and a [first,last] quad range that every box test shrinks (src/bvh/traverse.cpp:14-80,
// src/bounding_box.cpp:61-142).  Those semantics decide which triangle wins an exact-t tie, so they are
// kept exactly -- and they map onto a CDNA4 wavefront without any divergence:
//   * control flow is wave-uniform: node index, stack pointer, first/last live in SGPRs;
//   * a 32-B node record is ONE scalar load (s_load_dwordx8) through the scalar cache, requested ahead of
//     its use from a copy of the tree laid out for that (child pairs on one 64-B line, see SNAIL_PF_VISIT);
//   * the traversal stack is a VGPR pair (lane i = stack slot i): pop = v_readlane, push = lane-predicated move;
//   * "scan for the first/last quad with a surviving lane" is one __ballot + s_ff1/s_flbit;
//   * per-leaf, the packet-level Triangle::TestInterval cull and the shared-origin terms tvec0/tvec1/tmul
//     are computed with FOUR LANES PER TRIANGLE (one per vector component, the other components read from
//     the quad neighbours through DPP); survivors are broadcast with ds_bpermute and intersected by all
//     lanes -- or, under a narrow quad range, by the range's rays spread one or two per lane.
// No MFMA: this is branchy slab / Moeller-Trumbore work.  No FMA contraction either: every mul/add is
// rounded separately, in the reference's operand order (build with -ffp-contract=off), IEEE divide and
// sqrt (Inv(x)=1/x, RSqrt(x)=1/sqrt(x): veclib's scalar definitions, veclib/vecbase.h:53-55).
//
// Two instantiations of the walk live in each kernel and are selected per packet (wave-uniform):
//   FAST  : all inputs finite -> no NaN can arise in a slab test, so veclib's Min/Max (second operand
//           on NaN) equal v_min_f32/v_max_f32 up to the sign of zero (which no comparison sees), and
//           BBox::TestInterval is implied by the per-lane test (monotonic rounding) and is skipped;
//   EXACT : select-based Min/Max in the reference's operand order + the interval culls, for packets
//           containing non-finite reciprocals (e.g. dir == -1e-8 exactly, src/rtbase.h:117-120).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/snail_hip.h"
#include "host_sse.h"
#ifdef SNAIL_DEBUG_API
#include "../../include/snail_hip_debug.h"
#endif

// ---------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void snail_set_error(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

#define HIP_TRY(expr)                                                                                     \
	do {                                                                                                  \
		hipError_t e_ = (expr);                                                                           \
		if(e_ != hipSuccess) {                                                                            \
			snail_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
			return 100 + (int)e_;                                                                         \
		}                                                                                                 \
	} while(0)

// ---------------------------------------------------------------------------------------------------
// device code
// ---------------------------------------------------------------------------------------------------
#define SNAIL_DEV_NS dev
#define SNAIL_ARITH_SSE 0
#include "snail_dev.inc"
#undef SNAIL_DEV_NS
#undef SNAIL_ARITH_SSE
#define SNAIL_DEV_NS dev_sse
#define SNAIL_ARITH_SSE 1
#include "snail_dev.inc"
#undef SNAIL_DEV_NS
#undef SNAIL_ARITH_SSE

// A launch in the scene's arithmetic: dev::K, or dev_sse::K with the same argument record (the two namespaces are the same text, so the
// records have the same layout; they are distinct types, hence the copy).  K comes last because its template arguments hold commas.
#define SNAIL_LAUNCH(SSE, ARGS_T, GRID, BLOCK, LDS, STREAM, A, ...)                                                                          \
	do {                                                                                                                                   \
		if(SSE) {                                                                                                                          \
			dev_sse::ARGS_T a_;                                                                                                            \
			static_assert(sizeof(a_) == sizeof(A), "argument records of the two arithmetics differ");                                      \
			memcpy((void *)&a_, (const void *)&(A), sizeof(a_));                                                                           \
			hipLaunchKernelGGL((dev_sse::__VA_ARGS__), GRID, BLOCK, LDS, STREAM, a_);                                                      \
		} else hipLaunchKernelGGL((dev::__VA_ARGS__), GRID, BLOCK, LDS, STREAM, A);                                                        \
	} while(0)

// ---------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------------------------------
namespace { struct TileJob; void freeTileJobs(SnailScene *); } // render_host.inc: cached lists of snail_render_tiles / snail_render_image
struct SnailScene {
	int device = 0;
	int arith = SNAIL_ARITH_IEEE; // snail_scene_set_arith: which of the two kernel sets (dev / dev_sse) every launch of this scene takes
	// SNAIL_ARITH_HOST_SSE: THIS scene's copy of the rcpps / rsqrtps tables (3 x 4096 words; handed to every launch as the first kernel argument word,
	// dev_sse::hostTab()) and the generation of the process-wide tables it was filled from (host_sse.h).  Per scene, not per device: a second scene on
	// the same device may compute with another CPU's tables, frames interleaved.
	unsigned *dTab = nullptr;
	unsigned tabGen = 0;
	TileJob *tileJob = nullptr, *frameJob = nullptr;
	int nNodes = 0, nTris = 0, depth = 0;
	uint4 *dNodes = nullptr, *dTris = nullptr;   // the caller's records; dTris points INTO dPF (one allocation, see below)
	// [slot 0: unused][slot i + 1: node i, re-encoded for the record-prefetching loop (dev::pfEncode)] ... [triangle records at trisOff]
	char *dPF = nullptr;
	int trisOff = 0;
	// camera-relative copies of the node slots (dev::k_rel_nodes), one per distinct origin, least-recently-used first out: a static or
	// turning camera (and a light) costs one pass over the nodes EVER; a moving one costs one per new position
	struct RelNodes {
		float org[3] = {0, 0, 0};
		uint4 *d = nullptr;
		bool valid = false;
		unsigned long long stamp = 0;
		hipEvent_t filled = nullptr;
		enum { kStreams = 8 };
		hipStream_t usedOn[kStreams] = {};
		hipEvent_t used[kStreams] = {};
		int nUsed = 0;
	};
	enum { kRelSlots = 40 };   // (allocated on first use.  More than the frames a renderer keeps in flight -- 4 launches x 8 frames -- so that a camera whose POSITION
	                           // moves every frame never recycles an array a running launch still reads; 16 until round 5)
	RelNodes rel[kRelSlots];
	unsigned long long relClock = 0;
	int pfOK = 0;     // the prefetching loop may walk this tree: nested, every child pair starts at an odd index, offsets fit (stackPack)
	int fastOK = 0; // every triangle record finite and of sane magnitude (see file header)
	int nestedOK = 1; // every child box lies inside its parent's (stackPack)
	int pfOKButNesting = 0;
	int lastBlocks = 0, lastThreads = 0;
#ifndef SNAIL_DEFER_SLOTS
#define SNAIL_DEFER_SLOTS 8 // launches in flight per scene handle before a scratch slot is recycled (a recycled slot waits for its previous user)
#endif
	enum { kDeferSlots = SNAIL_DEFER_SLOTS };
	int *dDefer[kDeferSlots] = {};        // deferred-packet lists, one per launch in flight (round-robin)
	// a slot's buffers are reused by the 8th launch after it, possibly on another stream and possibly while the host runs far
	// ahead of the device: each slot carries an event recorded after its last kernel, and the next user's stream waits for it
	hipEvent_t deferDone[kDeferSlots] = {};
	bool deferUsed[kDeferSlots] = {};
	int deferCap = 0;
	unsigned launchCount = 0;
	// intermediate state of snail_render_whitted_dev, one set per launch in flight (round-robin, like dDefer)
	struct ShadeScratch {
		size_t packets = 0;
		bool refl = false;
		float *hitT = nullptr; int *hitId = nullptr;
		float *rOrg = nullptr, *rDir = nullptr, *rIDir = nullptr, *rDist = nullptr, *rCol = nullptr;
		int *rObj = nullptr;
		unsigned char *rMask = nullptr;
		int *defer = nullptr;
		float *sDist = nullptr;
		hipEvent_t done = nullptr;
		bool used = false;
	} shade[kDeferSlots];
	unsigned shadeCount = 0;
	// deferred-packet lists of snail_trace_rays*, same slot discipline
	struct RayDefer { int *p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; } rayDefer[kDeferSlots];
	unsigned rayCount = 0;
	// ---- concurrency (include/snail_hip.h, "Concurrency"): the reference hands ONE const Scene<AccStruct> to `threads` pthread workers, each of which
	// calls TraversePrimary / TraverseShadow on it (src/render.cpp:214-267, src/thread_pool.cpp:151-180, src/scene_trace.cpp:119-120,:560-563) ----
	// mu: everything a launch books in this handle -- the slot rotations above, the origin-relative node cache, scratch growth, lastBlocks, arith.
	// Held while a call ENQUEUES (microseconds), never while it waits for the device.
	std::mutex mu;
	// renderMu: a whole snail_render_tiles / _image call (they render out of ONE cached job per scene); taken before mu.
	std::mutex renderMu;
	// What a host-pointer call owns for its duration: a stream of its own, its own counter words and a staging arena (grown, never shrunk).  Taken
	// from / returned to this free list under mu; as many exist as calls were ever in flight at once.
	struct HostCall {
		hipStream_t stream = nullptr;
		unsigned long long *dStats = nullptr;
		char *arena = nullptr;
		size_t cap = 0, used = 0;
	};
	std::vector<HostCall *> hostFree;
};

namespace {

struct DeviceGuard {
	int prev = -1;
	bool ok = true;
	explicit DeviceGuard(int dev) {
		if(hipGetDevice(&prev) != hipSuccess) prev = -1;
		if(prev != dev) ok = hipSetDevice(dev) == hipSuccess;
	}
	~DeviceGuard() {
		int cur = -1;
		if(prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
	}
};

// A host-pointer call's private stream, counter words and staging arena (SnailScene::HostCall) for the duration of the call: two host threads
// inside snail_trace_shadow / _rays / _primary / _frame_packets / snail_render_* on ONE scene share nothing but the handle's bookkeeping, which they
// touch under SnailScene::mu.  Everything the call moves goes through its own stream in order: staged inputs -> kernels -> results -> ONE wait.
struct HostCallScope {
	SnailScene *s;
	SnailScene::HostCall *c = nullptr;
	int rc = 0;
	bool finished = false;
	HostCallScope(SnailScene *scene, const char *fn) : s(scene) {
		{
			std::lock_guard<std::mutex> lock(s->mu);
			if(!s->hostFree.empty()) { c = s->hostFree.back(); s->hostFree.pop_back(); }
		}
		if(!c) {
			c = new SnailScene::HostCall();
			hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
			if(e == hipSuccess) e = hipMalloc((void **)&c->dStats, 4 * sizeof(unsigned long long));
			if(e != hipSuccess) {
				snail_set_error("%s: per-call stream / counters: %s", fn, hipGetErrorString(e));
				if(c->stream) (void)hipStreamDestroy(c->stream);
				delete c; c = nullptr; rc = 100 + (int)e;
			}
		}
		if(c) c->used = 0;
	}
	~HostCallScope() {
		if(!c) return;
		// a call that returns early (an error after something was enqueued) must not leave copies into the CALLER's memory in flight
		if(!finished) (void)hipStreamSynchronize(c->stream);
		std::lock_guard<std::mutex> lock(s->mu);
		s->hostFree.push_back(c);
	}
	HostCallScope(const HostCallScope &) = delete;
	HostCallScope &operator=(const HostCallScope &) = delete;
	hipStream_t stream() const { return c->stream; }
	uint64_t *stats() const { return (uint64_t *)c->dStats; }
	static size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
	// the bytes the call is going to carve, all at once (the arena is re-allocated only when a larger call than ever before arrives; its previous
	// user has returned -- calls are synchronous -- so nothing in flight reads it)
	int reserve(size_t bytes) {
		if(bytes <= c->cap) return 0;
		if(c->arena) (void)hipFree(c->arena);
		c->arena = nullptr; c->cap = 0;
		const size_t want = bytes + bytes / 4 + 4096;
		HIP_TRY(hipMalloc((void **)&c->arena, want));
		c->cap = want;
		return 0;
	}
	void *carve(size_t bytes) { void *p = c->arena + c->used; c->used += pad(bytes); return p; }
	// carve + copy in (src may be null: carve only)
	int put(void **d, const void *src, size_t bytes) {
		*d = carve(bytes);
		if(src && bytes) HIP_TRY(hipMemcpyAsync(*d, src, bytes, hipMemcpyHostToDevice, c->stream));
		return 0;
	}
	int get(void *dst, const void *d, size_t bytes) {
		if(dst && bytes) HIP_TRY(hipMemcpyAsync(dst, d, bytes, hipMemcpyDeviceToHost, c->stream));
		return 0;
	}
	int zeroStats() { HIP_TRY(hipMemsetAsync(c->dStats, 0, 4 * sizeof(unsigned long long), c->stream)); return 0; }
	// wait for everything the call enqueued, then add its counters to the caller's
	int finish(uint64_t *stats) {
		unsigned long long hs[4] = {0, 0, 0, 0};
		if(stats) HIP_TRY(hipMemcpyAsync(hs, c->dStats, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		finished = true;
		if(stats) for(int k = 0; k < 4; k++) stats[k] += hs[k];
		return 0;
	}
};

// RayGenerator ctor (src/ray_generator.cpp:4-15); plain fp32, this TU is built with -ffp-contract=off
dev::GenConst makeGen(const float cam[13], int w, int h) {
	dev::GenConst g;
	const float *pos = cam, *right = cam + 3, *up = cam + 6, *front = cam + 9;
	const float pd = cam[12];
	float invW = 1.0f / float(w), invH = 1.0f / float(h);
	invW *= float(w) / float(h);
	const float ax[4] = {0.0f, 1.0f, 0.0f, 1.0f}, ay[4] = {0.0f, 0.0f, 1.0f, 1.0f};
	for(int c = 0; c < 3; c++) { g.tright[c] = right[c] * invW; g.tup[c] = up[c] * invH; g.org[c] = pos[c]; }
	for(int l = 0; l < 4; l++) {
		const float taddx = ax[l] - w * 0.5f, taddy = ay[l] - h * 0.5f;
		for(int c = 0; c < 3; c++) {
			const float fp = front[c] * pd;
			g.txyz[c][l] = g.tright[c] * taddx + g.tup[c] * taddy + fp;
		}
	}
	return g;
}

// Host-SSE arithmetic (host_sse.h): the tables in force into a device buffer.  *dTab is allocated on first use (current device); when *tabGen is not
// the generation of the tables in force, the device is drained first -- launches in flight read the buffer's old contents -- and the buffer rewritten.
int hostSseFill(const char *fn, unsigned **dTab, unsigned *tabGen) {
	const char *why = "";
	std::vector<unsigned> snap((size_t)3 * kHostSseEntries);
	unsigned gen = 0;
	if(hostSseSnapshot(snap.data(), &gen, &why)) { snail_set_error("%s: SNAIL_ARITH_HOST_SSE is not available on this host: %s", fn, why); return 2; }
	if(*dTab && *tabGen == gen) return 0;
	if(!*dTab) HIP_TRY(hipMalloc((void **)dTab, snap.size() * sizeof(unsigned)));
	HIP_TRY(hipDeviceSynchronize());   // (tables are replaced between frames, not under them: snail_arith_set_tables)
	HIP_TRY(hipMemcpy(*dTab, snap.data(), snap.size() * sizeof(unsigned), hipMemcpyHostToDevice));
	*tabGen = gen;
	return 0;
}
// ... for the entry points that take no scene (snail_shade_depth_arith_dev; the workbench's device check): one buffer per device, refreshed -- lazily, at
// the call that finds it stale -- like a scene's
// `capturing`: the caller's stream is being captured into a graph -- a stale buffer cannot be refreshed there (the refresh drains the device): an error that
// names the remedy (snail_arith_prepare_device before the capture) instead of a broken capture.
int hostSseDeviceTables(const char *fn, const unsigned **out, bool capturing = false) {
	static std::mutex mu;
	static unsigned *tab[64] = {};
	static unsigned gen[64] = {};
	int devId = 0;
	HIP_TRY(hipGetDevice(&devId));
	if(devId < 0 || devId >= 64) { snail_set_error("%s: device index %d", fn, devId); return 1; }
	std::lock_guard<std::mutex> lock(mu);
	if(capturing) {
		unsigned now = 0;
		const char *why = "";
		if(hostSseSnapshot(nullptr, &now, &why)) { snail_set_error("%s: SNAIL_ARITH_HOST_SSE is not available on this host: %s", fn, why); return 2; }
		if(!tab[devId] || gen[devId] != now) {
			snail_set_error("%s: this device's copy of the rcpps / rsqrtps tables is missing or stale and the stream is being captured: call snail_arith_prepare_device() before the capture", fn);
			return 3;
		}
	} else if(int rc = hostSseFill(fn, &tab[devId], &gen[devId])) return rc;
	*out = tab[devId];
	return 0;
}

bool originSane(const float *o) {
	for(int c = 0; c < 3; c++)
		if(!(std::fabs(o[c]) <= 1.0e9f)) return false;
	return true;
}

// Environment switches exist in the workbench build only (-DSNAIL_DEBUG_API): the product library reads no environment.
#ifdef SNAIL_DEBUG_API
int debugEnvInt(const char *name) { const char *v = getenv(name); return v ? atoi(v) : 0; }
#else
constexpr int debugEnvInt(const char *) { return 0; }
#endif

// depth > 62 needs the second stack register pair (DEEP instantiations).  Workbench build: SNAIL_DEBUG_FORCE_DEEP=1 selects them for
// any scene (the deep-BVH stack study of BASELINE config 5, tools/quick_time.py; the test of the walks ordinary scenes never select)
bool useDeep(const SnailScene *s) {
	static const bool force = debugEnvInt("SNAIL_DEBUG_FORCE_DEEP") != 0;
	return force || s->depth > 62;
}

// One-word stack entries (and with them the record-prefetching node loop over its own copy of the tree, SnailScene::dPF) need record
// slots below 2^20, child pairs at odd indices (the reference's builders and the LBVH allocate children in pairs from index 1 on, so a
// pair shares one 64-B line of the copy and "the other child" is offset ^ 32), byte offsets below 2^31 -- and a NESTED tree: inside
// the prefetching loop EXEC is the set of quads that survived the parent (a quad that fails a box fails every box inside it), so a
// child's first / last come from the survivors only, whereas the reference rescans the whole inherited range
// (src/bounding_box.cpp:71-139).  Trees of the reference's builders and of the LBVH refit are nested by construction; a caller's
// tree that is not (snail_scene_create checks every child box against its parent's) takes the two-word loop, whose EXEC is the
// inherited range [first, last] -- the reference's rescan exactly.  Workbench build: SNAIL_DEBUG_NO_PACK=1 keeps the two-word form.
int stackPack(const SnailScene *s) {
	static const bool off = debugEnvInt("SNAIL_DEBUG_NO_PACK") != 0;
	static const bool assumeNested = debugEnvInt("SNAIL_DEBUG_ASSUME_NESTED") != 0;   // (tests/nonnested_env.py: shows that the check below matters)
	return !off && (s->pfOK || (assumeNested && s->pfOKButNesting)) ? 1 : 0;
}

// The node slots relative to `org` for a launch on `stream` (stream-ordered: filled on this stream on a miss; a later user on another stream
// waits for the fill; a slot that is recycled waits for its last users).  *which = the cache entry, for relUsed() after the consumer's launch.
// `pend` given: a miss is not filled here but appended to *pend, and relFlush() fills all of a launch's misses with ONE kernel (the frames of a
// multi-frame launch of a moving camera each have an origin of their own).
struct RelPending {
	dev::RelFillArgs fill;
	int entry[SNAIL_MAX_BATCH];
};
int relFor(SnailScene *s, const float org[3], hipStream_t stream, const uint4 **out, int *which, RelPending *pend = nullptr) {
	SnailScene::RelNodes *hit = nullptr, *victim = nullptr;
	for(auto &e : s->rel) {
		if(e.valid && memcmp(e.org, org, 12) == 0) { hit = &e; break; }
		if(!victim || (!e.valid && victim->valid) || (e.valid == victim->valid && e.stamp < victim->stamp)) victim = &e;
	}
	if(hit) {
		HIP_TRY(hipStreamWaitEvent(stream, hit->filled, 0));
		hit->stamp = ++s->relClock;
		*out = hit->d; *which = (int)(hit - s->rel);
		return 0;
	}
	SnailScene::RelNodes &e = *victim;
	const int nSlots = s->nNodes + 1;
	if(!e.d) HIP_TRY(hipMalloc((void **)&e.d, (size_t)nSlots * 32));
	if(!e.filled) HIP_TRY(hipEventCreateWithFlags(&e.filled, hipEventDisableTiming));
	HIP_TRY(hipStreamWaitEvent(stream, e.filled, 0));                                       // its previous fill (a fill whose launch never happened has no reader event)
	for(int k = 0; k < e.nUsed; k++) HIP_TRY(hipStreamWaitEvent(stream, e.used[k], 0));   // its last readers, on whatever streams
	e.nUsed = 0;
	if(pend && pend->fill.n < SNAIL_MAX_BATCH) {
		const int k = pend->fill.n++;
		memcpy(pend->fill.org[k], org, 12); pend->fill.dst[k] = e.d; pend->entry[k] = (int)(&e - s->rel);
	} else {
		hipLaunchKernelGGL(dev::k_rel_nodes, dim3((unsigned)((nSlots + 255) / 256)), dim3(256), 0, stream, (const uint4 *)s->dPF, nSlots, org[0], org[1], org[2], e.d);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(e.filled, stream));
	}
	memcpy(e.org, org, 12); e.valid = true; e.stamp = ++s->relClock;
	*out = e.d; *which = (int)(&e - s->rel);
	return 0;
}
// the pending fills of one launch, on its stream, before the launch that reads them
// (an entry is marked valid when it is appended: should the fill never be enqueued -- an error between relFor and relFlush, or in it -- relDrop
// takes the entries back, so that no later launch finds an array that was never written)
void relDrop(SnailScene *s, RelPending &pend) {
	for(int k = 0; k < pend.fill.n; k++) s->rel[pend.entry[k]].valid = false;
	pend.fill.n = 0;
}
int relFlush(SnailScene *s, RelPending &pend, hipStream_t stream) {
	if(pend.fill.n == 0) return 0;
	const int nSlots = s->nNodes + 1;
	hipLaunchKernelGGL(dev::k_rel_nodes_multi, dim3((unsigned)((nSlots + 255) / 256)), dim3(256), 0, stream, (const uint4 *)s->dPF, nSlots, pend.fill);
	hipError_t e = hipGetLastError();
	for(int k = 0; k < pend.fill.n && e == hipSuccess; k++) e = hipEventRecord(s->rel[pend.entry[k]].filled, stream);
	if(e != hipSuccess) {
		relDrop(s, pend);
		snail_set_error("origin-relative node records: %s", hipGetErrorString(e));
		return 100 + (int)e;
	}
	pend.fill.n = 0;
	return 0;
}
// after the launch that reads entry `which` was enqueued on `stream`
int relUsed(SnailScene *s, int which, hipStream_t stream) {
	SnailScene::RelNodes &e = s->rel[which];
	int k = 0;
	while(k < e.nUsed && e.usedOn[k] != stream) k++;
	if(k == e.nUsed) {
		if(e.nUsed == SnailScene::RelNodes::kStreams) { // more streams than slots (never with the renderers of this repo): fold the oldest into this one
			HIP_TRY(hipStreamWaitEvent(stream, e.used[0], 0));
			k = 0;
		} else e.nUsed++;
		e.usedOn[k] = stream;
		if(!e.used[k]) HIP_TRY(hipEventCreateWithFlags(&e.used[k], hipEventDisableTiming));
	}
	HIP_TRY(hipEventRecord(e.used[k], stream));
	return 0;
}

int checkScene(const SnailScene *s, const char *fn) {
	if(!s || !s->dNodes || !s->dPF || !s->dTris) { snail_set_error("%s: invalid scene handle", fn); return 1; }
	return 0;
}

// every entry point that books a launch in the handle holds SnailScene::mu while it ENQUEUES (launchPrimary*, launchRays, launchLights, renderWhitted,
// relFor / relUsed and the slot rotations assume it is held); nothing waits for the device under it except the rare synchronous scratch growth
#define SNAIL_LOCK(s) std::lock_guard<std::mutex> snailLock_((s)->mu)

// the frames of one launch: cameras (13 floats each) and output planes per frame (any plane may be null)
struct FrameSet {
	int n = 0;
	const float *cam[SNAIL_MAX_BATCH] = {};
	dev::FrameOut out[SNAIL_MAX_BATCH] = {};
};

// The deferred-packet launch that follows a traversal kernel, with or without the next dispatch order derived inside it (dev::orderSortBlock as its
// last workgroup: *_reorder_dev).  In-launch sort: up to SORT_IN_LAUNCH_MAX slots (their 16-bit costs + the sort's 16.4 KB + the pass's own 3.3 KB fit
// the 64 KB of LDS a workgroup gets without further ado) with 16-byte aligned costs; beyond that the order is derived by the stand-alone kernel right
// after the pass (sortAfter) -- same result for the caller, one launch more.
enum { SORT_IN_LAUNCH_MAX = 20480 };
struct ExactPass {
	dim3 grid, block;
	size_t lds = 0;
	int *nextOrder = nullptr;   // what the kernel's argument record gets
	int orderFlags = 0;         // SNAIL_ORDER_* of the caller
	bool sortAfter = false;
};
ExactPass exactPassFor(int exactBlocks, const int32_t *dSlotCost, int nSlots, int32_t *dNextOrder, int orderFlags) {
	ExactPass e;
	e.orderFlags = orderFlags;
	e.grid = dim3((unsigned)exactBlocks); e.block = dim3(64);
	if(!dNextOrder || !dSlotCost || nSlots <= 0) return e;
	if(nSlots <= SORT_IN_LAUNCH_MAX && ((uintptr_t)dSlotCost & 15) == 0) {
		e.grid = dim3((unsigned)exactBlocks + 1); e.block = dim3(ORDER_THREADS_LDS);
		e.lds = (size_t)((nSlots + 3) / 4 * 4) * 2;
		e.nextOrder = (int *)dNextOrder;
	} else e.sortAfter = true;
	return e;
}

int launchPrimaryFrames(SnailScene *s, const FrameSet &FS, int resx, int resy, int x0, int y0, int w, int h, const int32_t *dPacketXY, int nPackets, uint64_t *dStats,
						hipStream_t stream, unsigned *dCost = nullptr, bool packetMajor = false, const int32_t *dOrder = nullptr, int32_t *dSlotCost = nullptr,
						int32_t *dNextOrder = nullptr, int orderFlags = 0) {
	if(resx <= 0 || resy <= 0) { snail_set_error("snail_trace_primary: bad resolution %dx%d", resx, resy); return 1; }
	if(dNextOrder && !dSlotCost) { snail_set_error("snail_trace_primary: the next dispatch order is derived from d_slot_cost, which is null"); return 1; }
	if(FS.n < 1 || FS.n > SNAIL_MAX_BATCH || (SNAIL_BLOCK_WAVES > 1 && FS.n > 1)) { snail_set_error("snail_trace_primary: 1..%d frames per launch (got %d)", SNAIL_MAX_BATCH, FS.n); return 1; }
	dev::PrimaryArgs A;
	memset(&A, 0, sizeof(A));
	A.hostTab = s->arith == SNAIL_ARITH_HOST_SSE ? s->dTab : nullptr;
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.nFrames = FS.n;
	A.fastOK = s->fastOK;
	for(int k = 0; k < FS.n; k++) {
		A.g[k] = makeGen(FS.cam[k], resx, resy);
		A.out[k] = FS.out[k];
		A.fastOK = A.fastOK && originSane(FS.cam[k]);
	}
	A.resx = resx; A.resy = resy;
	A.stats = (dev::u64 *)dStats;
	A.cost = dCost;
	A.packetMajor = packetMajor ? 1 : 0;
	A.pack = stackPack(s);
	int relWhich[SNAIL_MAX_BATCH];
	for(int k = 0; k < FS.n; k++) { relWhich[k] = -1; A.rel[k] = nullptr; }
	int blocks;
	if(dPacketXY) {
		if(nPackets <= 0) return 0;
		A.packetXY = (const int2 *)dPacketXY;
		A.nPackets = nPackets;
		blocks = ((nPackets + 127) / 128) * 128;
	} else {
		if((x0 & 15) || (y0 & 15) || w <= 0 || h <= 0 || x0 < 0 || y0 < 0) {
			snail_set_error("snail_trace_primary: rect origin must be a non-negative multiple of 16 and the size positive (got %d,%d %dx%d)", x0, y0, w, h);
			return 1;
		}
		A.x0 = x0; A.y0 = y0; A.w = w; A.h = h;
		A.pw = (w + 15) / 16; A.ph = (h + 15) / 16;
		A.nPackets = A.pw * A.ph;
		const int nRegions = ((A.pw + 3) / 4) * ((A.ph + 3) / 4);
		blocks = ((nRegions + 7) / 8) * 8 * 16;
	}
	A.nBlocks = blocks;
	A.nSlots = dPacketXY ? nPackets : blocks;
	A.order = dOrder; A.slotCost = dSlotCost;
	// the frames' origin-relative node records (only the record-prefetching loop reads them: not the DEEP kernels' C++ walk); after every
	// early return above, so that a fill is always followed by its reader's event (relUsed below)
	if(A.pack && SNAIL_REL_NODES && SNAIL_NODE_PREFETCH && !useDeep(s)) {
		RelPending pend;
		pend.fill.n = 0;
		for(int k = 0; k < FS.n; k++)
			if(int rc = relFor(s, FS.cam[k], stream, &A.rel[k], &relWhich[k], &pend)) { relDrop(s, pend); return rc; }   // cam[0..2] = the camera position
		if(int rc = relFlush(s, pend, stream)) return rc;   // the launch's new origins (a moving camera: every frame's) in ONE pass over the records
	}
	const int gridBlocks = blocks * FS.n;
	s->lastBlocks = gridBlocks; s->lastThreads = 64;
	// deferred-packet list of this launch (re-allocated, synchronously, only when a larger launch than ever before arrives)
	if(gridBlocks + 2 > s->deferCap) {
		HIP_TRY(hipDeviceSynchronize());
		for(int k = 0; k < SnailScene::kDeferSlots; k++) {
			if(s->dDefer[k]) (void)hipFree(s->dDefer[k]);
			s->dDefer[k] = nullptr;
			HIP_TRY(hipMalloc((void **)&s->dDefer[k], (size_t)(gridBlocks + 2) * sizeof(int)));
			HIP_TRY(hipMemset(s->dDefer[k], 0, 2 * sizeof(int)));
		}
		HIP_TRY(hipDeviceSynchronize());   // (a memset is ordered on the null stream only: the launches may run on streams that do not wait for it)
		s->deferCap = gridBlocks + 2;
	}
	const int slot = (int)(s->launchCount++ % SnailScene::kDeferSlots);
	A.defer = s->dDefer[slot];
	if(!s->deferDone[slot]) HIP_TRY(hipEventCreateWithFlags(&s->deferDone[slot], hipEventDisableTiming));
	if(s->deferUsed[slot]) HIP_TRY(hipStreamWaitEvent(stream, s->deferDone[slot], 0));
	// workbench build, SNAIL_DEBUG_DYNLDS=<bytes>: occupancy experiments only (unused dynamic LDS limits waves per CU)
	static const int dynLds = debugEnvInt("SNAIL_DEBUG_DYNLDS");
	// a scene with sane records defers (practically) nothing: a handful of blocks suffices; an unsafe scene defers every packet
	const int exactBlocks = A.fastOK ? (blocks < 8 ? blocks : 8) : (blocks < 2048 ? blocks : 2048);
	static_assert(128 % SNAIL_BLOCK_WAVES == 0, "the slot count is a multiple of 128");
	const dim3 grid(gridBlocks / SNAIL_BLOCK_WAVES), block(64 * SNAIL_BLOCK_WAVES);
	const ExactPass EP = exactPassFor(exactBlocks, dSlotCost, A.nSlots, dCost ? nullptr : dNextOrder, orderFlags);
	A.nextOrder = EP.nextOrder; A.nextOrderFlags = EP.orderFlags;
	const bool sse = s->arith == SNAIL_ARITH_HOST_SSE;
	if(dCost) { // diagnostic launch (snail_account_packets)
		if(useDeep(s)) SNAIL_LAUNCH(sse, PrimaryArgs, dim3(blocks), dim3(64), 0, stream, A, k_primary_diag<true>);
		else SNAIL_LAUNCH(sse, PrimaryArgs, dim3(blocks), dim3(64), 0, stream, A, k_primary_diag<false>);
	} else if(useDeep(s)) SNAIL_LAUNCH(sse, PrimaryArgs, grid, block, dynLds, stream, A, k_primary<true>);
	else SNAIL_LAUNCH(sse, PrimaryArgs, grid, block, dynLds, stream, A, k_primary<false>);
	// workbench build, SNAIL_DEBUG_NO_EXACT_PASS=1: what the dependent second launch costs (an experiment: deferred packets are then never traced and their
	// list is never re-armed -- so only in the IEEE arithmetic, whose primary packets of an ordinary scene defer nothing, and never when the pass carries the
	// order sort; in the table arithmetic the list would grow frame by frame and overflow: round 5 read "2 x the frame rate" off such a run)
	static const bool noExact = debugEnvInt("SNAIL_DEBUG_NO_EXACT_PASS") != 0;
	if(noExact && s->arith == SNAIL_ARITH_IEEE && !EP.nextOrder && !EP.sortAfter) { }
	else if(useDeep(s)) SNAIL_LAUNCH(sse, PrimaryArgs, EP.grid, EP.block, EP.lds, stream, A, k_primary_exact<true>);
	else SNAIL_LAUNCH(sse, PrimaryArgs, EP.grid, EP.block, EP.lds, stream, A, k_primary_exact<false>);
	HIP_TRY(hipGetLastError());
	if(EP.sortAfter) { if(int rc = snail_order_from_cost_hint_dev(dSlotCost, A.nSlots, dNextOrder, EP.orderFlags, stream)) return rc; }
	HIP_TRY(hipEventRecord(s->deferDone[slot], stream));
	s->deferUsed[slot] = true;
	for(int k = 0; k < FS.n; k++)
		if(relWhich[k] >= 0) { if(int rc = relUsed(s, relWhich[k], stream)) return rc; }
	return 0;
}

// one frame per launch (every entry point but the *_batch_dev ones)
int launchPrimary(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, const int32_t *dPacketXY,
				  int nPackets, float *t, float *u, float *v, int32_t *id, uint64_t *dStats, hipStream_t stream, unsigned *dCost = nullptr,
				  bool packetMajor = false, uint8_t *dBgr = nullptr, const int32_t *dOrder = nullptr, int32_t *dSlotCost = nullptr, int32_t *dNextOrder = nullptr, int orderFlags = 0) {
	FrameSet FS;
	FS.n = 1; FS.cam[0] = cam;
	FS.out[0].t = t; FS.out[0].u = u; FS.out[0].v = v; FS.out[0].id = (int *)id; FS.out[0].bgr = dBgr;
	return launchPrimaryFrames(s, FS, resx, resy, x0, y0, w, h, dPacketXY, nPackets, dStats, stream, dCost, packetMajor, dOrder, dSlotCost, dNextOrder, orderFlags);
}

// scratch of one staged frame: ONE allocation, carved (hitT is its base); grown synchronously when a larger frame or the first
// reflection frame arrives
int shadeScratch(SnailScene *s, SnailScene::ShadeScratch &W, size_t packets, size_t blocks, bool refl) {
	if(W.hitT && W.packets >= packets && (W.refl || !refl)) return 0;
	HIP_TRY(hipDeviceSynchronize());
	if(W.hitT) (void)hipFree(W.hitT);
	const hipEvent_t keep = W.done;
	W = SnailScene::ShadeScratch();
	W.done = keep;
	const size_t rays = packets * 256, quads = packets * 64;
	const size_t deferInts = blocks * SNAIL_MAX_LIGHTS + 16;
	size_t bytes = rays * 8 + deferInts * 4 + rays * 4 * SNAIL_MAX_LIGHTS; // hitT, hitId, defer list, sDist
	if(refl) bytes += quads * 12 * 4 * 3 + rays * 4 * 2 + rays * 12 + quads; // rOrg, rDir, rIDir; rDist, rObj; rCol; rMask
	char *base = nullptr;
	HIP_TRY(hipMalloc((void **)&base, bytes));
	W.packets = packets; W.refl = refl;
	W.hitT = (float *)base; base += rays * 4;
	W.hitId = (int *)base; base += rays * 4;
	W.defer = (int *)base; base += deferInts * 4;
	HIP_TRY(hipMemset(W.defer, 0, 16 * sizeof(int)));
	HIP_TRY(hipDeviceSynchronize());   // (ordered on the null stream only, see launchPrimaryFrames)
	W.sDist = (float *)base; base += rays * 4 * SNAIL_MAX_LIGHTS;
	if(refl) {
		W.rOrg = (float *)base; base += quads * 48;
		W.rDir = (float *)base; base += quads * 48;
		W.rIDir = (float *)base; base += quads * 48;
		W.rDist = (float *)base; base += rays * 4;
		W.rObj = (int *)base; base += rays * 4;
		W.rCol = (float *)base; base += rays * 12;
		W.rMask = (unsigned char *)base;
	}
	return 0;
}

template <bool SHARED, bool MASK>
void launchRaysKernels(const SnailScene *s, const dev::RaysArgs &A, int blocks, const ExactPass &EP, hipStream_t stream) {
	const bool deep = useDeep(s), bary = A.bary != nullptr, sse = s->arith == SNAIL_ARITH_HOST_SSE;
#define SNAIL_RAYS_LAUNCH(D, B)                                                                                                            \
	do {                                                                                                                                   \
		SNAIL_LAUNCH(sse, RaysArgs, dim3(blocks), dim3(64), 0, stream, A, k_rays<SHARED, MASK, D, B>);                                      \
		SNAIL_LAUNCH(sse, RaysArgs, EP.grid, EP.block, EP.lds, stream, A, k_rays_exact<SHARED, MASK, D, B>);                                \
	} while(0)
	if(deep && bary) SNAIL_RAYS_LAUNCH(true, true);
	else if(deep) SNAIL_RAYS_LAUNCH(true, false);
	else if(bary) SNAIL_RAYS_LAUNCH(false, true);
	else SNAIL_RAYS_LAUNCH(false, false);
#undef SNAIL_RAYS_LAUNCH
}

int launchRays(SnailScene *s, bool shadow, int nPackets, int size, int sharedOrigin, const float *origin, const float *dir,
					  const float *idir, const uint8_t *mask, float *distance, int32_t *object, float *bary, uint64_t *dStats, hipStream_t stream,
					  const int32_t *dOrder = nullptr, int32_t *dSlotCost = nullptr, int nSlots = 0, int32_t *dNextOrder = nullptr, int orderFlags = 0) {
	if(nPackets <= 0) return 0;
	if(size < 1 || size > SNAIL_PACKET_QUADS) { snail_set_error("packet size %d outside 1..%d quads", size, SNAIL_PACKET_QUADS); return 1; }
	if(!origin || !dir || !idir || !distance || (!shadow && !object)) { snail_set_error("null ray array"); return 1; }
	dev::RaysArgs A;
	memset(&A, 0, sizeof(A));
	A.hostTab = s->arith == SNAIL_ARITH_HOST_SSE ? s->dTab : nullptr;
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.nPackets = nPackets; A.size = size; A.fastOK = s->fastOK; A.pack = stackPack(s);
	A.origin = origin; A.dir = dir; A.idir = idir; A.mask = mask;
	A.distance = distance; A.object = object; A.bary = bary;
	A.stats = (dev::u64 *)dStats;
	int blocks = ((nPackets + 127) / 128) * 128;
	if(!shadow && (dOrder || dSlotCost) && nSlots >= nPackets) { // dispatch-order feedback of the staged pipeline (k_rays): one block per slot
		A.order = dOrder; A.slotCost = dSlotCost; A.nSlots = nSlots;
		if(nSlots > blocks) blocks = ((nSlots + 127) / 128) * 128;
	}
	SnailScene::RayDefer &R = s->rayDefer[s->rayCount++ % SnailScene::kDeferSlots];
	if((size_t)nPackets + 16 > R.cap) { // grown synchronously when a larger batch than ever before arrives
		HIP_TRY(hipDeviceSynchronize());
		if(R.p) (void)hipFree(R.p);
		R.p = nullptr; R.cap = 0;
		HIP_TRY(hipMalloc((void **)&R.p, ((size_t)nPackets + 16) * sizeof(int)));
		HIP_TRY(hipMemset(R.p, 0, 16 * sizeof(int)));
		HIP_TRY(hipDeviceSynchronize());   // (ordered on the null stream only, see launchPrimaryFrames)
		R.cap = (size_t)nPackets + 16;
	}
	if(!R.done) HIP_TRY(hipEventCreateWithFlags(&R.done, hipEventDisableTiming));
	if(R.used) HIP_TRY(hipStreamWaitEvent(stream, R.done, 0));
	A.defer = R.p;
	const int exactBlocks = A.fastOK ? (blocks < 8 ? blocks : 8) : (blocks < 2048 ? blocks : 2048);
	const ExactPass EP = exactPassFor(exactBlocks, A.slotCost, A.nSlots, shadow ? nullptr : dNextOrder, orderFlags);
	A.nextOrder = EP.nextOrder; A.nextOrderFlags = EP.orderFlags;
	if(shadow) {
		if(useDeep(s)) {
			hipLaunchKernelGGL(dev::k_shadow<true>, dim3(blocks), dim3(64), 0, stream, A);
			hipLaunchKernelGGL(dev::k_shadow_exact<true>, dim3(exactBlocks), dim3(64), 0, stream, A);
		} else {
			hipLaunchKernelGGL(dev::k_shadow<false>, dim3(blocks), dim3(64), 0, stream, A);
			hipLaunchKernelGGL(dev::k_shadow_exact<false>, dim3(exactBlocks), dim3(64), 0, stream, A);
		}
	} else if(sharedOrigin && mask) launchRaysKernels<true, true>(s, A, blocks, EP, stream);
	else if(sharedOrigin) launchRaysKernels<true, false>(s, A, blocks, EP, stream);
	else if(mask) launchRaysKernels<false, true>(s, A, blocks, EP, stream);
	else launchRaysKernels<false, false>(s, A, blocks, EP, stream);
	HIP_TRY(hipGetLastError());
	if(EP.sortAfter) { if(int rc = snail_order_from_cost_hint_dev(A.slotCost, A.nSlots, dNextOrder, EP.orderFlags, stream)) return rc; }
	HIP_TRY(hipEventRecord(R.done, stream));
	R.used = true;
	return 0;
}

template <int SRC>
int launchLights(SnailScene *s, dev::ShadeArgs A /* a copy: relLight is filled in here */, hipStream_t stream, int32_t *dNextOrder = nullptr, int orderFlags = 0) {
	if(A.nLights <= 0) return 0;
	int relWhich[SNAIL_MAX_LIGHTS];
	for(int n = 0; n < SNAIL_MAX_LIGHTS; n++) { relWhich[n] = -1; A.relLight[n] = nullptr; }
	if(SNAIL_NODE_PREFETCH && SNAIL_REL_SHADOW && A.pack && !useDeep(s))
		for(int n = 0; n < A.nLights; n++)
			if(int rc = relFor(s, A.lights[n], stream, &A.relLight[n], &relWhich[n])) return rc;   // lights[n][0..2] = the light's position
	const dim3 grid(A.nBlocks, A.nLights);
	const int total = A.nBlocks * A.nLights;
	const int exactBlocks = A.fastOK ? (total < 8 ? total : 8) : (total < 2048 ? total : 2048);
	const bool sse = s->arith == SNAIL_ARITH_HOST_SSE;
	const ExactPass EP = exactPassFor(exactBlocks, A.slotCost, A.nSlots, dNextOrder, orderFlags);
	A.nextOrder = EP.nextOrder; A.nextOrderFlags = EP.orderFlags;
	if(useDeep(s)) {
		SNAIL_LAUNCH(sse, ShadeArgs, grid, dim3(64), 0, stream, A, k_light<true, SRC>);
		SNAIL_LAUNCH(sse, ShadeArgs, EP.grid, EP.block, EP.lds, stream, A, k_light_exact<true, SRC>);
	} else {
		SNAIL_LAUNCH(sse, ShadeArgs, grid, dim3(64), 0, stream, A, k_light<false, SRC>);
		SNAIL_LAUNCH(sse, ShadeArgs, EP.grid, EP.block, EP.lds, stream, A, k_light_exact<false, SRC>);
	}
	HIP_TRY(hipGetLastError());
	if(EP.sortAfter) { if(int rc = snail_order_from_cost_hint_dev(A.slotCost, A.nSlots, dNextOrder, EP.orderFlags, stream)) return rc; }
	for(int n = 0; n < A.nLights; n++)
		if(relWhich[n] >= 0) { if(int rc = relUsed(s, relWhich[n], stream)) return rc; }
	return 0;
}

} // namespace

#include "lbvh.inc"

extern "C" {

const char *snail_last_error(void) { return g_err; }

int snail_device_count(void) {
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// byte offset of the triangle records in SnailScene::dPF: a CONSTANT (2^25 = room for 2^20 record slots), so that the prefetching loop
// turns a leaf record's triangle offset back into an index with immediates (SNAIL_PF_VISIT, L_leaf)
static size_t pfTrisOffset(int) { return (size_t)1 << 25; }

SnailScene *snail_scene_create(const void *nodes32, int nNodes, const void *tris64, int nTris, int depth, int device) {
	if(!nodes32 || !tris64 || nNodes <= 0 || nTris <= 0) { snail_set_error("snail_scene_create: empty scene"); return nullptr; }
	if(depth < 0 || depth > SNAIL_MAX_DEPTH) { snail_set_error("snail_scene_create: depth %d outside 0..BVH::maxDepth = %d", depth, SNAIL_MAX_DEPTH); return nullptr; }
	// validate topology on the host so that no kernel can index out of bounds (a GPU fault resets the node)
	const uint32_t *nw = (const uint32_t *)nodes32;
	int pairsOdd = 1; // every child pair starts at an odd index (children are allocated in pairs from index 1 on: src/bvh/tree.cpp:153-157)
	for(int i = 0; i < nNodes; i++) {
		uint32_t sub = nw[i * 8 + 6], aux = nw[i * 8 + 7];
		if(sub & 0x80000000u) {
			uint32_t first = sub & 0x7fffffffu;
			if((int32_t)aux < 0 || (uint64_t)first + aux > (uint64_t)nTris) { snail_set_error("snail_scene_create: leaf %d references triangles [%u,%u) of %d", i, first, first + aux, nTris); return nullptr; }
		} else {
			if((uint64_t)sub + 1 >= (uint64_t)nNodes || sub == 0) { snail_set_error("snail_scene_create: node %d has child %u of %d", i, sub, nNodes); return nullptr; }
			if((aux & 0xffff) > 2 || (aux >> 16) > 1) { snail_set_error("snail_scene_create: node %d has axis/firstNode %u/%u", i, aux & 0xffff, aux >> 16); return nullptr; }
			if(!(sub & 1u)) pairsOdd = 0;
		}
	}
	// ... and that no walk can run away: every node is reached at most once from the root (no cycle, no shared subtree: a back-edge
	// would keep every wave in its loop for ever) and no leaf lies deeper than the caller says (the traversal stack is sized by
	// `depth`: lane i of a VGPR pair = slot i, a second pair beyond 62 levels; a deeper tree would wrap the lane select)
	int realDepth = 0, nestedOK = 1;
	{
		std::vector<uint8_t> seen((size_t)nNodes, 0);
		std::vector<std::pair<int, int>> todo; // (node, level), root = level 0 as BVH::depth counts (src/bvh/tree.cpp:54-59)
		todo.emplace_back(0, 0);
		while(!todo.empty()) {
			const auto [i, level] = todo.back();
			todo.pop_back();
			if(seen[i]) { snail_set_error("snail_scene_create: node %d is reachable twice from the root (cycle or shared subtree)", i); return nullptr; }
			seen[i] = 1;
			if(level > realDepth) realDepth = level;
			if(level > depth) { snail_set_error("snail_scene_create: node %d lies at level %d, deeper than the declared depth %d", i, level, depth); return nullptr; }
			const uint32_t sub = nw[(size_t)i * 8 + 6];
			if(!(sub & 0x80000000u)) {
				todo.emplace_back((int)sub + 1, level + 1); todo.emplace_back((int)sub, level + 1);
				// child box inside the parent's box, per axis (a NaN bound fails the comparison and counts as not nested)
				const float *pb = (const float *)nodes32 + (size_t)i * 8;
				for(int c = 0; c < 2; c++) {
					const float *cb = (const float *)nodes32 + ((size_t)sub + c) * 8;
					for(int k = 0; k < 3; k++)
						if(!(cb[k] >= pb[k]) || !(cb[3 + k] <= pb[3 + k])) nestedOK = 0;
				}
			}
		}
	}
	int fastOK = 1;
	const float *tf = (const float *)tris64;
	for(int i = 0; i < nTris && fastOK; i++) {
		const float *r = tf + (size_t)i * 16;
		for(int k = 0; k < 9; k++) if(!(std::fabs(r[k]) <= 1.0e9f)) fastOK = 0;         // a, ba, ca
		if(!(r[9] > 0.0f) || !(r[10] <= 1.0e12f) || !(r[10] > 0.0f)) fastOK = 0;        // t0, it0
		for(int k = 12; k < 16; k++) if(!(std::fabs(r[k]) <= 1.0e18f)) fastOK = 0;     // plane
	}
	const float *nf = (const float *)nodes32;
	for(int i = 0; i < nNodes && fastOK; i++) {
		for(int k = 0; k < 6; k++) if(!(std::fabs(nf[(size_t)i * 8 + k]) <= 1.0e9f)) fastOK = 0;
		for(int k = 0; k < 3; k++) if(!(nf[(size_t)i * 8 + k] <= nf[(size_t)i * 8 + 3 + k])) fastOK = 0;   // M_COH needs min <= max
	}

	DeviceGuard guard(device);
	if(!guard.ok) { snail_set_error("snail_scene_create: hipSetDevice(%d) failed", device); return nullptr; }
	SnailScene *s = new SnailScene();
	s->device = device; s->nNodes = nNodes; s->nTris = nTris; s->depth = realDepth; s->fastOK = fastOK; s->nestedOK = nestedOK; // the measured depth (<= declared) picks the stack form
	// ONE allocation: [slot 0][the node records re-encoded for the record-prefetching loop][triangle records]; the caller's node
	// records, verbatim, in a second one (every other walk reads those)
	const size_t trisOff = pfTrisOffset(nNodes), pfBytes = trisOff + (size_t)nTris * 64;
	const bool fits = (size_t)nNodes + 1 < ((size_t)1 << 20) && pfBytes < ((size_t)1 << 31);
	s->pfOKButNesting = pairsOdd && fits;
	s->pfOK = s->pfOKButNesting && nestedOK;
	s->trisOff = (int)(fits ? trisOff : 0);
	std::vector<uint32_t> pf;
	if(s->pfOKButNesting) {
		pf.assign(((size_t)nNodes + 1) * 8, 0u);
		for(int i = 0; i < nNodes; i++) {
			unsigned in[8], out[8];
			for(int k = 0; k < 8; k++) in[k] = nw[(size_t)i * 8 + k];
			dev::pfEncode(in, out, (unsigned)trisOff);
			for(int k = 0; k < 8; k++) pf[((size_t)i + 1) * 8 + k] = out[k];
		}
	}
	hipError_t e;
	if((e = hipMalloc((void **)&s->dNodes, (size_t)nNodes * 32)) != hipSuccess || (e = hipMalloc((void **)&s->dPF, fits ? pfBytes : (size_t)nTris * 64)) != hipSuccess ||
	   (e = hipMemcpy(s->dNodes, nodes32, (size_t)nNodes * 32, hipMemcpyHostToDevice)) != hipSuccess ||
	   (e = hipMemcpy(s->dPF + s->trisOff, tris64, (size_t)nTris * 64, hipMemcpyHostToDevice)) != hipSuccess ||
	   (!pf.empty() && (e = hipMemcpy(s->dPF, pf.data(), pf.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)) {
		snail_set_error("snail_scene_create: %s", hipGetErrorString(e));
		snail_scene_destroy(s);
		return nullptr;
	}
	s->dTris = (uint4 *)(s->dPF + s->trisOff);
	return s;
}

SnailScene *snail_scene_create_lbvh(const float *tri_verts, int nTris, int device, int maxLeafTris, int32_t *perm, float *build_ms) {
	if(!tri_verts || nTris <= 0 || maxLeafTris < 1 || maxLeafTris > 64) { snail_set_error("snail_scene_create_lbvh: bad arguments (1 <= maxLeafTris <= 64)"); return nullptr; }
	DeviceGuard guard(device);
	if(!guard.ok) { snail_set_error("snail_scene_create_lbvh: hipSetDevice(%d) failed", device); return nullptr; }
	uint4 *dNodes = nullptr, *dTris = nullptr;
	int depth = 0, fastOK = 0;
	if(buildLbvhDevice(tri_verts, nTris, maxLeafTris, &dNodes, &dTris, perm, &depth, &fastOK, build_ms)) return nullptr;
	if(depth > SNAIL_MAX_DEPTH) {
		(void)hipFree(dNodes); (void)hipFree(dTris);
		snail_set_error("snail_scene_create_lbvh: depth %d exceeds BVH::maxDepth %d", depth, SNAIL_MAX_DEPTH);
		return nullptr;
	}
	SnailScene *s = new SnailScene();
	s->device = device; s->nNodes = 2 * nTris - 1; s->nTris = nTris; s->depth = depth; s->fastOK = fastOK;
	s->dNodes = dNodes;
	// the LBVH is nested by construction (bottom-up refit: a parent's box is the exact union of its children's) and its children live in
	// slots 1 + 2i, 2 + 2i: the record-prefetching loop's copy is encoded on the device, the triangle records move behind it
	const size_t trisOff = pfTrisOffset(s->nNodes), pfBytes = trisOff + (size_t)nTris * 64;
	const bool fits = (size_t)s->nNodes + 1 < ((size_t)1 << 20) && pfBytes < ((size_t)1 << 31);
	s->trisOff = (int)(fits ? trisOff : 0);
	s->nestedOK = 1; s->pfOKButNesting = s->pfOK = fits ? 1 : 0;
	hipError_t e;
	if((e = hipMalloc((void **)&s->dPF, fits ? pfBytes : (size_t)nTris * 64)) != hipSuccess ||
	   (e = hipMemcpy(s->dPF + s->trisOff, dTris, (size_t)nTris * 64, hipMemcpyDeviceToDevice)) != hipSuccess) {
		(void)hipFree(dTris);
		snail_set_error("snail_scene_create_lbvh: %s", hipGetErrorString(e));
		snail_scene_destroy(s);
		return nullptr;
	}
	(void)hipFree(dTris);
	s->dTris = (uint4 *)(s->dPF + s->trisOff);
	if(fits) {
		(void)hipMemset(s->dPF, 0, 32);
		hipLaunchKernelGGL(dev::k_pf_encode, dim3((unsigned)((s->nNodes + 255) / 256)), dim3(256), 0, 0, s->dNodes, s->nNodes, (unsigned)trisOff, (uint4 *)s->dPF);
		if((e = hipDeviceSynchronize()) != hipSuccess) {
			snail_set_error("snail_scene_create_lbvh: %s", hipGetErrorString(e));
			snail_scene_destroy(s);
			return nullptr;
		}
	}
	return s;
}

int snail_scene_download(const SnailScene *s, void *nodes32, void *tris64) {
	if(int rc = checkScene(s, "snail_scene_download")) return rc;
	DeviceGuard guard(s->device);
	if(nodes32) HIP_TRY(hipMemcpy(nodes32, s->dNodes, (size_t)s->nNodes * 32, hipMemcpyDeviceToHost));
	if(tris64) HIP_TRY(hipMemcpy(tris64, s->dTris, (size_t)s->nTris * 64, hipMemcpyDeviceToHost));
	return 0;
}

void snail_scene_destroy(SnailScene *s) {
	if(!s) return;
	DeviceGuard guard(s->device);
	freeTileJobs(s);
	if(s->dNodes) (void)hipFree(s->dNodes);
	if(s->dPF) (void)hipFree(s->dPF);   // (dTris points into it)
	if(s->dTab) (void)hipFree(s->dTab);
	for(auto &e : s->rel) {
		if(e.d) (void)hipFree(e.d);
		if(e.filled) (void)hipEventDestroy(e.filled);
		for(hipEvent_t ev : e.used) if(ev) (void)hipEventDestroy(ev);
	}
	for(SnailScene::HostCall *c : s->hostFree) {
		if(c->arena) (void)hipFree(c->arena);
		if(c->dStats) (void)hipFree(c->dStats);
		if(c->stream) (void)hipStreamDestroy(c->stream);
		delete c;
	}
	for(int k = 0; k < SnailScene::kDeferSlots; k++) if(s->dDefer[k]) (void)hipFree(s->dDefer[k]);
	for(int k = 0; k < SnailScene::kDeferSlots; k++) if(s->shade[k].hitT) (void)hipFree(s->shade[k].hitT);
	for(int k = 0; k < SnailScene::kDeferSlots; k++) {
		if(s->rayDefer[k].p) (void)hipFree(s->rayDefer[k].p);
		if(s->rayDefer[k].done) (void)hipEventDestroy(s->rayDefer[k].done);
		if(s->deferDone[k]) (void)hipEventDestroy(s->deferDone[k]);
		if(s->shade[k].done) (void)hipEventDestroy(s->shade[k].done);
	}
	delete s;
}

int snail_scene_info(const SnailScene *s, int *nNodes, int *nTris, int *depth, int *device) {
	if(!s) { snail_set_error("snail_scene_info: null scene"); return 1; }
	if(nNodes) *nNodes = s->nNodes;
	if(nTris) *nTris = s->nTris;
	if(depth) *depth = s->depth;
	if(device) *device = s->device;
	return 0;
}

int snail_scene_flags(const SnailScene *s, int *fastOK, int *nestedOK) {
	if(!s) { snail_set_error("snail_scene_flags: null scene"); return 1; }
	if(fastOK) *fastOK = s->fastOK;
	if(nestedOK) *nestedOK = s->nestedOK;
	return 0;
}

// ---- arithmetic of the approximate operations (include/snail_hip.h) ----
int snail_scene_set_arith(SnailScene *s, int arith) {
	if(int rc = checkScene(s, "snail_scene_set_arith")) return rc;
	if(arith != SNAIL_ARITH_IEEE && arith != SNAIL_ARITH_HOST_SSE) { snail_set_error("snail_scene_set_arith: unknown arithmetic %d", arith); return 1; }
	SNAIL_LOCK(s);
	if(arith == SNAIL_ARITH_HOST_SSE) {
		DeviceGuard guard(s->device);
		if(!guard.ok) { snail_set_error("snail_scene_set_arith: hipSetDevice(%d) failed", s->device); return 1; }
		if(int rc = hostSseFill("snail_scene_set_arith", &s->dTab, &s->tabGen)) return rc;   // the tables in force NOW: this scene's from here on
	}
	s->arith = arith;
	return 0;
}
int snail_scene_arith(const SnailScene *s, int *arith) {
	if(int rc = checkScene(s, "snail_scene_arith")) return rc;
	if(arith) *arith = s->arith;
	return 0;
}
int snail_arith_set_tables(const uint32_t *tables12288) {
	const char *why = "";
	if(hostSseSetTables(tables12288, &why)) { snail_set_error("snail_arith_set_tables: %s", why); return 1; }
	return 0;
}
int snail_arith_prepare_device(void) {
	const unsigned *tab = nullptr;
	return hostSseDeviceTables("snail_arith_prepare_device", &tab);
}
int snail_host_sse_tables(uint32_t *tables12288) {
	const char *why = "";
	if(hostSseSnapshot((unsigned *)tables12288, nullptr, &why)) { snail_set_error("snail_host_sse_tables: %s", why); return 2; }
	return 0;
}
int snail_host_sse_check(int fn, uint64_t first, uint64_t count, int threads, uint64_t *mismatches, uint32_t *firstBad) {
	if((fn != 0 && fn != 1) || first > (1ull << 32) || count > (1ull << 32) - first || !mismatches) { snail_set_error("snail_host_sse_check: bad arguments"); return 1; }
	const char *why = "";
	if(hostSseSnapshot(nullptr, nullptr, &why)) { snail_set_error("snail_host_sse_check: %s", why); return 2; }
	unsigned bad = 0;
	*mismatches = hostSseMismatches(fn, first, count, threads, &bad);
	if(firstBad) *firstBad = bad;
	return 0;
}

int snail_last_launch(const SnailScene *s, int *blocks, int *threads) {
	if(!s) { snail_set_error("snail_last_launch: null scene"); return 1; }
	std::lock_guard<std::mutex> lock(const_cast<SnailScene *>(s)->mu);
	if(blocks) *blocks = s->lastBlocks;
	if(threads) *threads = s->lastThreads;
	return 0;
}

int snail_trace_primary_dev(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, float *t, float *u,
							float *v, int32_t *id, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_primary_dev")) return rc;
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchPrimary(s, cam, resx, resy, x0, y0, w, h, nullptr, 0, t, u, v, id, dStats, (hipStream_t)stream);
}

int snail_primary_slots(int w, int h) {
	if(w <= 0 || h <= 0) return 0;
	const int pw = (w + 15) / 16, ph = (h + 15) / 16;
	return (((pw + 3) / 4) * ((ph + 3) / 4) + 7) / 8 * 8 * 16;
}

int snail_trace_primary_ordered_dev(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, float *t, float *u,
									float *v, int32_t *id, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, void *stream) {
	if(int rc = checkScene(s, "snail_trace_primary_ordered_dev")) return rc;
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchPrimary(s, cam, resx, resy, x0, y0, w, h, nullptr, 0, t, u, v, id, dStats, (hipStream_t)stream, nullptr, false, nullptr, dOrder, dSlotCost);
}

int snail_trace_packets_ordered_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, float *t,
									float *u, float *v, int32_t *id, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, void *stream) {
	if(int rc = checkScene(s, "snail_trace_packets_ordered_dev")) return rc;
	if(!dPacketXY && nPackets > 0) { snail_set_error("snail_trace_packets_ordered_dev: null packet list"); return 1; }
	if(nPackets <= 0) return 0; // a rank without tiles (the reference's server renders nothing)
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchPrimary(s, cam, resx, resy, 0, 0, 0, 0, dPacketXY, nPackets, t, u, v, id, dStats, (hipStream_t)stream, nullptr, false, nullptr, dOrder, dSlotCost);
}

int snail_order_from_cost_dev(const int32_t *dSlotCost, int nSlots, int32_t *dOrder, void *stream) { return snail_order_from_cost_hint_dev(dSlotCost, nSlots, dOrder, SNAIL_ORDER_AUTO, stream); }

int snail_order_from_cost_hint_dev(const int32_t *dSlotCost, int nSlots, int32_t *dOrder, int orderFlags, void *stream) {
	if(nSlots <= 0) return 0;
	if(!dSlotCost || !dOrder) { snail_set_error("snail_order_from_cost_dev: null buffer"); return 1; }
	if(orderFlags & ~SNAIL_ORDER_SORTED) { snail_set_error("snail_order_from_cost_hint_dev: unknown order flags 0x%x", orderFlags); return 1; }
	const hipStream_t st = (hipStream_t)stream;
	const size_t stashBytes = (size_t)((nSlots + 3) / 4 * 4) * 2;
	bool ldsForm = nSlots <= ORDER_LDS_MAX_SLOTS && ((uintptr_t)dSlotCost & 15) == 0;   // (a caller's array that is not 16-byte aligned, or a frame beyond ~6K x 4K: the multi-pass kernel)
	if(ldsForm && stashBytes > 32768) {   // more dynamic LDS than a kernel gets by default (static 16.4 KB + this): raised once per device
		static std::mutex mu;
		static signed char raised[64] = {};   // 0 = not tried, 1 = raised, -1 = refused
		int devId = 0;
		HIP_TRY(hipGetDevice(&devId));
		std::lock_guard<std::mutex> lock(mu);
		signed char &r = raised[devId & 63];
		if(r == 0) r = hipFuncSetAttribute((const void *)dev::k_order_from_cost_lds, hipFuncAttributeMaxDynamicSharedMemorySize, ORDER_LDS_MAX_SLOTS * 2) == hipSuccess ? 1 : -1;
		ldsForm = r == 1;
	}
	if(ldsForm)
		hipLaunchKernelGGL(dev::k_order_from_cost_lds, dim3(1), dim3(ORDER_THREADS_LDS), stashBytes, st, dSlotCost, nSlots, dOrder, orderFlags);
	else hipLaunchKernelGGL(dev::k_order_from_cost, dim3(1), dim3(ORDER_THREADS), 0, st, dSlotCost, nSlots, dOrder);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_trace_packets_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, float *t,
							float *u, float *v, int32_t *id, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_packets_dev")) return rc;
	if(!dPacketXY && nPackets > 0) { snail_set_error("snail_trace_packets_dev: null packet list"); return 1; }
	if(nPackets <= 0) return 0; // a rank without tiles (the reference's server renders nothing)
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchPrimary(s, cam, resx, resy, 0, 0, 0, 0, dPacketXY, nPackets, t, u, v, id, dStats, (hipStream_t)stream);
}

static int frameSetFrom(const char *fn, FrameSet &FS, int nFrames, const float *cams13) {
	if(nFrames < 1 || nFrames > SNAIL_MAX_BATCH || !cams13) { snail_set_error("%s: 1..%d frames per launch and their cameras are required (got %d)", fn, SNAIL_MAX_BATCH, nFrames); return 1; }
	FS.n = nFrames;
	for(int k = 0; k < nFrames; k++) FS.cam[k] = cams13 + (size_t)k * 13;
	return 0;
}

int snail_trace_primary_batch_dev(SnailScene *s, int nFrames, const float *cams13, int resx, int resy, float *const *t, float *const *u, float *const *v,
								  int32_t *const *id, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, void *stream) {
	if(int rc = checkScene(s, "snail_trace_primary_batch_dev")) return rc;
	FrameSet FS;
	if(int rc = frameSetFrom("snail_trace_primary_batch_dev", FS, nFrames, cams13)) return rc;
	for(int k = 0; k < nFrames; k++) {
		FS.out[k].t = t ? t[k] : nullptr; FS.out[k].u = u ? u[k] : nullptr; FS.out[k].v = v ? v[k] : nullptr; FS.out[k].id = id ? (int *)id[k] : nullptr;
	}
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchPrimaryFrames(s, FS, resx, resy, 0, 0, resx, resy, nullptr, 0, dStats, (hipStream_t)stream, nullptr, false, dOrder, dSlotCost);
}

int snail_trace_primary_batch_reorder_dev(SnailScene *s, int nFrames, const float *cams13, int resx, int resy, float *const *t, float *const *u, float *const *v,
										  int32_t *const *id, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, int32_t *dNextOrder, int orderFlags, void *stream) {
	if(int rc = checkScene(s, "snail_trace_primary_batch_reorder_dev")) return rc;
	if(orderFlags & ~SNAIL_ORDER_SORTED) { snail_set_error("snail_trace_primary_batch_reorder_dev: unknown order flags 0x%x", orderFlags); return 1; }
	FrameSet FS;
	if(int rc = frameSetFrom("snail_trace_primary_batch_reorder_dev", FS, nFrames, cams13)) return rc;
	for(int k = 0; k < nFrames; k++) {
		FS.out[k].t = t ? t[k] : nullptr; FS.out[k].u = u ? u[k] : nullptr; FS.out[k].v = v ? v[k] : nullptr; FS.out[k].id = id ? (int *)id[k] : nullptr;
	}
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchPrimaryFrames(s, FS, resx, resy, 0, 0, resx, resy, nullptr, 0, dStats, (hipStream_t)stream, nullptr, false, dOrder, dSlotCost, dNextOrder, orderFlags);
}

int snail_trace_packets_shaded_batch_dev(SnailScene *s, int nFrames, const float *cams13, int resx, int resy, const int32_t *dPacketXY, int nPackets,
										 uint8_t *const *bgr, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_packets_shaded_batch_dev")) return rc;
	if((!dPacketXY || !bgr) && nPackets > 0) { snail_set_error("snail_trace_packets_shaded_batch_dev: null buffer"); return 1; }
	if(nPackets <= 0) return 0;
	FrameSet FS;
	if(int rc = frameSetFrom("snail_trace_packets_shaded_batch_dev", FS, nFrames, cams13)) return rc;
	for(int k = 0; k < nFrames; k++) {
		if(!bgr[k] || ((unsigned long long)bgr[k] & 3)) { snail_set_error("snail_trace_packets_shaded_batch_dev: d_bgr[%d] must be a 4-byte aligned device pointer", k); return 1; }
		FS.out[k].bgr = bgr[k];
	}
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchPrimaryFrames(s, FS, resx, resy, 0, 0, 0, 0, dPacketXY, nPackets, dStats, (hipStream_t)stream);
}

int snail_trace_packets_shaded_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, uint8_t *bgr,
								   uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_packets_shaded_dev")) return rc;
	if((!dPacketXY || !bgr) && nPackets > 0) { snail_set_error("snail_trace_packets_shaded_dev: null buffer"); return 1; }
	if(nPackets <= 0) return 0;
	if((unsigned long long)bgr & 3) { snail_set_error("snail_trace_packets_shaded_dev: d_bgr must be 4-byte aligned"); return 1; }
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchPrimary(s, cam, resx, resy, 0, 0, 0, 0, dPacketXY, nPackets, nullptr, nullptr, nullptr, nullptr, dStats, (hipStream_t)stream, nullptr, false, bgr);
}

int snail_packets_to_frame_dev(const int32_t *dPacketXY, int nPackets, int resx, int resy, const float *pt, const float *pu, const float *pv,
							   const int32_t *pid, float *t, float *u, float *v, int32_t *id, void *stream) {
	if(nPackets <= 0) return 0;
	dev::ScatterArgs A{(const int2 *)dPacketXY, nPackets, resx, resy, pt, pu, pv, pid, t, u, v, id};
	hipLaunchKernelGGL(dev::k_packets_to_frame, dim3((nPackets + 3) / 4), dim3(256), 0, (hipStream_t)stream, A);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_trace_primary(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, float *t, float *u, float *v,
						int32_t *id, uint64_t stats[4]) {
	if(int rc = checkScene(s, "snail_trace_primary")) return rc;
	if(resx <= 0 || resy <= 0) { snail_set_error("snail_trace_primary: bad resolution %dx%d", resx, resy); return 1; }
	DeviceGuard guard(s->device);
	HostCallScope hc(s, "snail_trace_primary");
	if(hc.rc) return hc.rc;
	const size_t n = (size_t)resx * resy * 4;
	// the host buffers are full frames of which only the rect is defined: stage through device frames
	if(int rc = hc.reserve(4 * HostCallScope::pad(n))) return rc;
	void *dt = nullptr, *du = nullptr, *dv = nullptr, *did = nullptr;
	if(t) { if(int rc = hc.put(&dt, t, n)) return rc; }
	if(u) { if(int rc = hc.put(&du, u, n)) return rc; }
	if(v) { if(int rc = hc.put(&dv, v, n)) return rc; }
	if(id) { if(int rc = hc.put(&did, id, n)) return rc; }
	if(stats) { if(int rc = hc.zeroStats()) return rc; }
	{
		SNAIL_LOCK(s);
		if(int rc = launchPrimary(s, cam, resx, resy, x0, y0, w, h, nullptr, 0, (float *)dt, (float *)du, (float *)dv, (int32_t *)did, stats ? hc.stats() : nullptr, hc.stream())) return rc;
	}
	if(int rc = hc.get(t, dt, n)) return rc;
	if(int rc = hc.get(u, du, n)) return rc;
	if(int rc = hc.get(v, dv, n)) return rc;
	if(int rc = hc.get(id, did, n)) return rc;
	return hc.finish(stats);
}


int snail_trace_rays_dev(SnailScene *s, int nPackets, int size, int sharedOrigin, const float *origin, const float *dir, const float *idir,
						 const uint8_t *mask, float *distance, int32_t *object, float *bary, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_rays_dev")) return rc;
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchRays(s, false, nPackets, size, sharedOrigin, origin, dir, idir, mask, distance, object, bary, dStats, (hipStream_t)stream);
}

int snail_trace_shadow_dev(SnailScene *s, int nPackets, int size, const float *origin3, const float *dir, const float *idir, float *distance,
						   uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_trace_shadow_dev")) return rc;
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	return launchRays(s, true, nPackets, size, 1, origin3, dir, idir, nullptr, distance, nullptr, nullptr, dStats, (hipStream_t)stream);
}

int snail_trace_rays(SnailScene *s, int nPackets, int size, int sharedOrigin, const float *origin, const float *dir, const float *idir,
					 const uint8_t *mask, float *distance, int32_t *object, float *bary, uint64_t stats[4]) {
	if(int rc = checkScene(s, "snail_trace_rays")) return rc;
	if(nPackets <= 0) return 0;
	if(size < 1 || size > SNAIL_PACKET_QUADS) { snail_set_error("packet size %d outside 1..%d quads", size, SNAIL_PACKET_QUADS); return 1; }
	if(!origin || !dir || !idir || !distance || !object) { snail_set_error("null ray array"); return 1; }
	DeviceGuard guard(s->device);
	HostCallScope hc(s, "snail_trace_rays");
	if(hc.rc) return hc.rc;
	const size_t nq = (size_t)nPackets * size;
	const size_t nOrg = (sharedOrigin ? (size_t)nPackets : nq) * 48;
	typedef HostCallScope H;
	if(int rc = hc.reserve(H::pad(nOrg) + 2 * H::pad(nq * 48) + H::pad(nq) + 2 * H::pad(nq * 16) + H::pad(nq * 32))) return rc;
	void *o = nullptr, *d = nullptr, *i = nullptr, *m = nullptr, *ds = nullptr, *ob = nullptr, *ba = nullptr;
	if(int rc = hc.put(&o, origin, nOrg)) return rc;
	if(int rc = hc.put(&d, dir, nq * 48)) return rc;
	if(int rc = hc.put(&i, idir, nq * 48)) return rc;
	if(mask) { if(int rc = hc.put(&m, mask, nq)) return rc; }
	if(int rc = hc.put(&ds, distance, nq * 16)) return rc;
	if(int rc = hc.put(&ob, object, nq * 16)) return rc;
	if(bary) { if(int rc = hc.put(&ba, bary, nq * 32)) return rc; }
	if(stats) { if(int rc = hc.zeroStats()) return rc; }
	{
		SNAIL_LOCK(s);
		if(int rc = launchRays(s, false, nPackets, size, sharedOrigin, (float *)o, (float *)d, (float *)i, (uint8_t *)m, (float *)ds, (int32_t *)ob, (float *)ba,
							   stats ? hc.stats() : nullptr, hc.stream()))
			return rc;
	}
	if(int rc = hc.get(distance, ds, nq * 16)) return rc;
	if(int rc = hc.get(object, ob, nq * 16)) return rc;
	if(int rc = hc.get(bary, ba, nq * 32)) return rc;
	return hc.finish(stats);
}

int snail_trace_shadow(SnailScene *s, int nPackets, int size, const float *origin3, const float *dir, const float *idir, float *distance,
					   uint64_t stats[4]) {
	if(int rc = checkScene(s, "snail_trace_shadow")) return rc;
	if(nPackets <= 0) return 0;
	if(size < 1 || size > SNAIL_PACKET_QUADS) { snail_set_error("packet size %d outside 1..%d quads", size, SNAIL_PACKET_QUADS); return 1; }
	if(!origin3 || !dir || !idir || !distance) { snail_set_error("null ray array"); return 1; }
	DeviceGuard guard(s->device);
	HostCallScope hc(s, "snail_trace_shadow");
	if(hc.rc) return hc.rc;
	const size_t nq = (size_t)nPackets * size;
	typedef HostCallScope H;
	if(int rc = hc.reserve(H::pad((size_t)nPackets * 12) + 2 * H::pad(nq * 48) + H::pad(nq * 16))) return rc;
	void *o = nullptr, *d = nullptr, *i = nullptr, *ds = nullptr;
	if(int rc = hc.put(&o, origin3, (size_t)nPackets * 12)) return rc;
	if(int rc = hc.put(&d, dir, nq * 48)) return rc;
	if(int rc = hc.put(&i, idir, nq * 48)) return rc;
	if(int rc = hc.put(&ds, distance, nq * 16)) return rc;
	if(stats) { if(int rc = hc.zeroStats()) return rc; }
	{
		SNAIL_LOCK(s);
		if(int rc = launchRays(s, true, nPackets, size, 1, (float *)o, (float *)d, (float *)i, nullptr, (float *)ds, nullptr, nullptr, stats ? hc.stats() : nullptr, hc.stream())) return rc;
	}
	if(int rc = hc.get(distance, ds, nq * 16)) return rc;
	return hc.finish(stats);
}

static int renderWhitted(const char *fn, SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPacketsList, const float *lights7,
						 int nLights, const float ambient[3], const float color[3], int flags, uint8_t *frame, int pitch, uint8_t *bgrPackets, uint64_t *dStats,
						 void *stream, float *colPackets = nullptr, const int32_t *dOrder = nullptr, int32_t *dSlotCost = nullptr, int32_t *dNextOrder = nullptr, int orderFlags = 0) {
	if(int rc = checkScene(s, fn)) return rc;
	if(dNextOrder && !dSlotCost) { snail_set_error("%s: the next dispatch orders are derived from d_slot_cost, which is null", fn); return 1; }
	if(resx <= 0 || resy <= 0 || nLights < 0 || nLights > SNAIL_MAX_LIGHTS || (nLights && !lights7) || !ambient || !color || (flags & ~SNAIL_WHITTED_REFLECTIONS) ||
	   (dPacketXY ? (colPackets ? false : (!bgrPackets || ((unsigned long long)bgrPackets & 3))) : (!frame || pitch < resx * 3 || colPackets))) {
		snail_set_error("%s: bad arguments (at most %d lights; flags = SNAIL_WHITTED_REFLECTIONS or 0; 4-byte aligned output)", fn, SNAIL_MAX_LIGHTS);
		return 1;
	}
	if(dPacketXY && nPacketsList <= 0) return 0;
	DeviceGuard guard(s->device);
	const bool refl = (flags & SNAIL_WHITTED_REFLECTIONS) != 0;
	dev::ShadeArgs A;
	memset(&A, 0, sizeof(A));
	A.hostTab = s->arith == SNAIL_ARITH_HOST_SSE ? s->dTab : nullptr;
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.g = makeGen(cam, resx, resy);
	A.resx = resx; A.resy = resy; A.pw = (resx + 15) / 16; A.ph = (resy + 15) / 16;
	A.fastOK = s->fastOK && originSane(cam);
	A.pack = stackPack(s);
	A.nLights = nLights;
	for(int n = 0; n < nLights; n++) for(int k = 0; k < 7; k++) A.lights[n][k] = lights7[n * 7 + k];
	for(int c = 0; c < 3; c++) { A.ambient[c] = ambient[c]; A.color[c] = color[c]; }
	A.frame = frame; A.pitch = pitch; A.stats = (dev::u64 *)dStats;
	A.packetXY = (const int2 *)dPacketXY; A.bgrPackets = bgrPackets; A.colPackets = colPackets;
	int packets, blocks;
	if(dPacketXY) {
		packets = nPacketsList;
		blocks = ((packets + 127) / 128) * 128;
	} else {
		packets = A.pw * A.ph;
		const int nRegions = ((A.pw + 3) / 4) * ((A.ph + 3) / 4);
		blocks = ((nRegions + 7) / 8) * 8 * 16;
	}
	A.nPackets = packets;
	A.nBlocks = blocks;
	A.fuse = nLights == 1 ? 1 : 0;   // one light: its k_light waves finish the packets themselves (no sDist round trip, no k_final launch)
	// dispatch-order feedback, frame grid only: SNAIL_WHITTED_STAGES arrays of `blocks` entries, back to back -- the primary packets, the shadow
	// packets of the primary hits (first light), the mirrored packets, the shadow packets of the mirrored hits (include/snail_hip.h)
	const int32_t *ord[SNAIL_WHITTED_STAGES] = {};
	int32_t *cst[SNAIL_WHITTED_STAGES] = {}, *nxt[SNAIL_WHITTED_STAGES] = {};
	if(!dPacketXY)
		for(int k = 0; k < SNAIL_WHITTED_STAGES; k++) {
			ord[k] = dOrder ? dOrder + (size_t)k * blocks : nullptr; cst[k] = dSlotCost ? dSlotCost + (size_t)k * blocks : nullptr;
			nxt[k] = dNextOrder ? dNextOrder + (size_t)k * blocks : nullptr;
		}
	SnailScene::ShadeScratch &W = s->shade[s->shadeCount++ % SnailScene::kDeferSlots];
	if(int rc = shadeScratch(s, W, (size_t)packets, (size_t)blocks, refl)) return rc;
	if(!W.done) HIP_TRY(hipEventCreateWithFlags(&W.done, hipEventDisableTiming));
	if(W.used) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, W.done, 0));
	A.hitT = W.hitT; A.hitId = W.hitId;
	A.rOrg = W.rOrg; A.rDir = W.rDir; A.rIDir = W.rIDir; A.rMask = W.rMask; A.rDist = W.rDist; A.rObj = W.rObj; A.rCol = W.rCol;
	A.sDist = W.sDist;
	A.blend = refl ? 1 : 0;
	A.defer = W.defer;
	const hipStream_t st = (hipStream_t)stream;
	const dim3 grid(blocks), wave(64);
	const bool sse = s->arith == SNAIL_ARITH_HOST_SSE;
	// the primary packets (the bench kernel), hit records packet-major
	if(dPacketXY) {
		if(int rc = launchPrimary(s, cam, resx, resy, 0, 0, 0, 0, dPacketXY, packets, W.hitT, nullptr, nullptr, W.hitId, dStats, st)) return rc;
	} else if(int rc = launchPrimary(s, cam, resx, resy, 0, 0, resx, resy, nullptr, 0, W.hitT, nullptr, nullptr, W.hitId, dStats, st, nullptr, true, nullptr, ord[0], cst[0], nxt[0], orderFlags)) return rc;
	if(refl) { // the nested RayTrace of the mirrored packets
		SNAIL_LAUNCH(sse, ShadeArgs, grid, wave, 0, st, A, k_final<dev::SRC_PRIMARY, dev::DST_MIRROR>);
		HIP_TRY(hipGetLastError());
		if(int rc = launchRays(s, false, packets, 64, 0, W.rOrg, W.rDir, W.rIDir, W.rMask, W.rDist, W.rObj, nullptr, dStats, st, ord[2], cst[2], dPacketXY ? 0 : blocks, nxt[2], orderFlags)) return rc;
		A.order = ord[3]; A.slotCost = cst[3]; A.nSlots = blocks;
		if(int rc = launchLights<dev::SRC_MIRROR>(s, A, st, nxt[3], orderFlags)) return rc;
		if(!A.fuse) SNAIL_LAUNCH(sse, ShadeArgs, grid, wave, 0, st, A, k_final<dev::SRC_MIRROR, dev::DST_COLOR>);
		HIP_TRY(hipGetLastError());
	}
	A.order = ord[1]; A.slotCost = cst[1]; A.nSlots = blocks;
	if(int rc = launchLights<dev::SRC_PRIMARY>(s, A, st, nxt[1], orderFlags)) return rc;
	if(!A.fuse) SNAIL_LAUNCH(sse, ShadeArgs, grid, wave, 0, st, A, k_final<dev::SRC_PRIMARY, dev::DST_FRAME>);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(W.done, st));
	W.used = true;
	return 0;
}

int snail_render_whitted_dev(SnailScene *s, const float cam[13], int resx, int resy, const float *lights7, int nLights, const float ambient[3],
							 const float color[3], int flags, uint8_t *frame, int pitch, uint64_t *dStats, void *stream) {
	if(int rc = checkScene(s, "snail_render_whitted_dev")) return rc;
	SNAIL_LOCK(s);
	return renderWhitted("snail_render_whitted_dev", s, cam, resx, resy, nullptr, 0, lights7, nLights, ambient, color, flags, frame, pitch, nullptr, dStats, stream);
}

int snail_render_whitted_ordered_dev(SnailScene *s, const float cam[13], int resx, int resy, const float *lights7, int nLights, const float ambient[3],
									 const float color[3], int flags, uint8_t *frame, int pitch, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, void *stream) {
	if(int rc = checkScene(s, "snail_render_whitted_ordered_dev")) return rc;
	SNAIL_LOCK(s);
	return renderWhitted("snail_render_whitted_ordered_dev", s, cam, resx, resy, nullptr, 0, lights7, nLights, ambient, color, flags, frame, pitch, nullptr, dStats, stream,
						 nullptr, dOrder, dSlotCost);
}

int snail_render_whitted_reorder_dev(SnailScene *s, const float cam[13], int resx, int resy, const float *lights7, int nLights, const float ambient[3],
									 const float color[3], int flags, uint8_t *frame, int pitch, uint64_t *dStats, const int32_t *dOrder, int32_t *dSlotCost, int32_t *dNextOrder,
									 int orderFlags, void *stream) {
	if(int rc = checkScene(s, "snail_render_whitted_reorder_dev")) return rc;
	if(orderFlags & ~SNAIL_ORDER_SORTED) { snail_set_error("snail_render_whitted_reorder_dev: unknown order flags 0x%x", orderFlags); return 1; }
	SNAIL_LOCK(s);
	return renderWhitted("snail_render_whitted_reorder_dev", s, cam, resx, resy, nullptr, 0, lights7, nLights, ambient, color, flags, frame, pitch, nullptr, dStats, stream,
						 nullptr, dOrder, dSlotCost, dNextOrder, orderFlags);
}

int snail_render_whitted_packets_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, const float *lights7,
									 int nLights, const float ambient[3], const float color[3], int flags, uint8_t *bgrPackets, uint64_t *dStats, void *stream) {
	if(!dPacketXY && nPackets > 0) { snail_set_error("snail_render_whitted_packets_dev: null packet list"); return 1; }
	if(nPackets <= 0) return 0;
	if(int rc = checkScene(s, "snail_render_whitted_packets_dev")) return rc;
	SNAIL_LOCK(s);
	return renderWhitted("snail_render_whitted_packets_dev", s, cam, resx, resy, dPacketXY, nPackets, lights7, nLights, ambient, color, flags, nullptr, 0, bgrPackets,
						 dStats, stream);
}

int snail_trace_transparency_dev(SnailScene *s, const float cam[13], int resx, int resy, const int32_t *dPacketXY, int nPackets, const float *dT, const int32_t *dTriId,
								 const uint8_t *dSel, const float *lights7, int nLights, const float ambient[3], const float color[3], float *dColor, uint64_t *dStats,
								 void *stream) {
	if(int rc = checkScene(s, "snail_trace_transparency_dev")) return rc;
	if(nPackets <= 0) return 0;
	if(!dPacketXY || !dT || !dTriId || !dSel || !dColor || resx <= 0 || resy <= 0 || nLights < 0 || nLights > SNAIL_MAX_LIGHTS || (nLights && !lights7) || !ambient || !color ||
	   ((unsigned long long)dColor & 3)) {
		snail_set_error("snail_trace_transparency_dev: bad arguments (null buffer, or more than %d lights)", SNAIL_MAX_LIGHTS);
		return 1;
	}
	DeviceGuard guard(s->device);
	SNAIL_LOCK(s);
	dev::ShadeArgs A;
	memset(&A, 0, sizeof(A));
	A.hostTab = s->arith == SNAIL_ARITH_HOST_SSE ? s->dTab : nullptr;
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.g = makeGen(cam, resx, resy);
	A.resx = resx; A.resy = resy; A.pw = (resx + 15) / 16; A.ph = (resy + 15) / 16;
	A.fastOK = s->fastOK && originSane(cam);
	A.pack = stackPack(s);
	A.nLights = nLights;
	for(int n = 0; n < nLights; n++) for(int k = 0; k < 7; k++) A.lights[n][k] = lights7[n * 7 + k];
	for(int c = 0; c < 3; c++) { A.ambient[c] = ambient[c]; A.color[c] = color[c]; }
	A.stats = (dev::u64 *)dStats;
	A.packetXY = (const int2 *)dPacketXY;
	const int blocks = ((nPackets + 127) / 128) * 128;
	A.nPackets = nPackets; A.nBlocks = blocks;
	SnailScene::ShadeScratch &W = s->shade[s->shadeCount++ % SnailScene::kDeferSlots];
	if(int rc = shadeScratch(s, W, (size_t)nPackets, (size_t)blocks, true)) return rc;
	if(!W.done) HIP_TRY(hipEventCreateWithFlags(&W.done, hipEventDisableTiming));
	if(W.used) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, W.done, 0));
	A.hitT = dT; A.hitId = dTriId; A.selIn = dSel;   // the caller's hit records (Context::distance / object of the packets)
	A.rOrg = W.rOrg; A.rDir = W.rDir; A.rIDir = W.rIDir; A.rMask = W.rMask; A.rDist = W.rDist; A.rObj = W.rObj;
	A.rCol = dColor;                                  // the nested RayTrace's colours go straight to the caller
	A.sDist = W.sDist; A.defer = W.defer;
	const hipStream_t st = (hipStream_t)stream;
	const bool sse = s->arith == SNAIL_ARITH_HOST_SSE;
	SNAIL_LAUNCH(sse, ShadeArgs, dim3(blocks), dim3(64), 0, st, A, k_final<dev::SRC_PRIMARY, dev::DST_CONTINUE>);
	HIP_TRY(hipGetLastError());
	if(int rc = launchRays(s, false, nPackets, 64, 0, W.rOrg, W.rDir, W.rIDir, W.rMask, W.rDist, W.rObj, nullptr, dStats, st)) return rc;
	A.fuse = nLights == 1 ? 1 : 0;   // (as in renderWhitted: the only light's waves write the colours themselves)
	if(int rc = launchLights<dev::SRC_MIRROR>(s, A, st)) return rc;
	if(!A.fuse) SNAIL_LAUNCH(sse, ShadeArgs, dim3(blocks), dim3(64), 0, st, A, k_final<dev::SRC_MIRROR, dev::DST_COLOR>);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(W.done, st));
	W.used = true;
	return 0;
}

int snail_shade_depth_arith_dev(const float *t, int nPackets, uint8_t *bgr, int arith, void *stream) {
	if(nPackets <= 0) return 0;
	if(!t || !bgr) { snail_set_error("snail_shade_depth_dev: null buffer"); return 1; }
	if(arith != SNAIL_ARITH_IEEE && arith != SNAIL_ARITH_HOST_SSE) { snail_set_error("snail_shade_depth_arith_dev: unknown arithmetic %d", arith); return 1; }
	const unsigned *tab = nullptr;
	if(arith == SNAIL_ARITH_HOST_SSE) {
		hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
		if(stream) (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
		if(int rc = hostSseDeviceTables("snail_shade_depth_arith_dev", &tab, cap == hipStreamCaptureStatusActive)) return rc;
	}
	const int n = nPackets * 256;
	if(arith == SNAIL_ARITH_HOST_SSE) hipLaunchKernelGGL(dev_sse::k_shade_depth, dim3((n / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, tab, t, n, bgr);
	else hipLaunchKernelGGL(dev::k_shade_depth, dim3((n / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, tab, t, n, bgr);
	HIP_TRY(hipGetLastError());
	return 0;
}
int snail_shade_depth_dev(const float *t, int nPackets, uint8_t *bgr, void *stream) { return snail_shade_depth_arith_dev(t, nPackets, bgr, SNAIL_ARITH_IEEE, stream); }

int snail_packets_bgr_to_frame_dev(const int32_t *dPacketXY, int nPackets, int resx, int resy, const uint8_t *bgr, uint8_t *frame, int pitch,
								   void *stream) {
	if(nPackets <= 0) return 0;
	if(!dPacketXY || !bgr || !frame || pitch < resx * 3) { snail_set_error("snail_packets_bgr_to_frame_dev: bad arguments"); return 1; }
	hipLaunchKernelGGL(dev::k_bgr_to_frame, dim3((nPackets + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const int2 *)dPacketXY, nPackets, resx,
					   resy, bgr, frame, pitch, 0, 0ll);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_packets_bgr_to_frame_chunked_dev(const int32_t *dPacketXY, int nPackets, int nPerChunk, int64_t chunkStrideBytes, int resx, int resy, const uint8_t *bgr,
										   uint8_t *frame, int pitch, void *stream) {
	if(nPackets <= 0) return 0;
	if(!dPacketXY || !bgr || !frame || pitch < resx * 3 || nPerChunk <= 0 || chunkStrideBytes < (int64_t)nPerChunk * 768 || (chunkStrideBytes & 3)) {
		snail_set_error("snail_packets_bgr_to_frame_chunked_dev: bad arguments");
		return 1;
	}
	hipLaunchKernelGGL(dev::k_bgr_to_frame, dim3((nPackets + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const int2 *)dPacketXY, nPackets, resx,
					   resy, bgr, frame, pitch, nPerChunk, (long long)chunkStrideBytes);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_packets_bgr_to_planar_dev(const int32_t *dTiles, const int32_t *dFirstPacket, const int64_t *dOutOffsets, int nTiles, const uint8_t *bgr,
									uint8_t *out, void *stream) {
	if(nTiles <= 0) return 0;
	if(!dTiles || !dFirstPacket || !dOutOffsets || !bgr || !out) { snail_set_error("snail_packets_bgr_to_planar_dev: bad arguments"); return 1; }
	if(nTiles > 65535) { snail_set_error("snail_packets_bgr_to_planar_dev: at most 65535 tiles per call (got %d)", nTiles); return 1; }
	hipLaunchKernelGGL(dev::k_bgr_to_planar, dim3(4, nTiles), dim3(256), 0, (hipStream_t)stream, (const int4 *)dTiles, dFirstPacket,
					   (const long long *)dOutOffsets, nTiles, bgr, out);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_planar_to_frame_dev(const int32_t *dTiles, const int64_t *dInOffsets, int nTiles, const uint8_t *planar, uint8_t *frame, int pitch, int resx,
							  int resy, void *stream) {
	if(nTiles <= 0) return 0;
	if(!dTiles || !dInOffsets || !planar || !frame || pitch < resx * 3) { snail_set_error("snail_planar_to_frame_dev: bad arguments"); return 1; }
	if(nTiles > 65535) { snail_set_error("snail_planar_to_frame_dev: at most 65535 tiles per call (got %d)", nTiles); return 1; }
	hipLaunchKernelGGL(dev::k_planar_to_frame, dim3(4, nTiles), dim3(256), 0, (hipStream_t)stream, (const int4 *)dTiles, (const long long *)dInOffsets,
					   nTiles, planar, frame, pitch, resx, resy);
	HIP_TRY(hipGetLastError());
	return 0;
}

#ifdef SNAIL_DEBUG_API
int snail_debug_delay_dev(float microseconds, void *stream) {
	if(!(microseconds >= 0.0f) || microseconds > 10000.0f) { snail_set_error("snail_debug_delay_dev: delay outside 0..10000 us"); return 1; }
	const unsigned ticks = (unsigned)(microseconds * 100.0f);
	if(ticks == 0) return 0;
	hipLaunchKernelGGL(dev::k_delay, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_debug_clock_dev(float microseconds, uint64_t *dOut2, void *stream) {
	if(!(microseconds > 0.0f) || microseconds > 10000.0f || !dOut2) { snail_set_error("snail_debug_clock_dev: bad arguments"); return 1; }
	hipLaunchKernelGGL(dev::k_clock, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned)(microseconds * 100.0f), (unsigned long long *)dOut2);
	HIP_TRY(hipGetLastError());
	return 0;
}

int snail_debug_recip_check(uint64_t out2[2]) {
	if(!out2) { snail_set_error("snail_debug_recip_check: bad arguments"); return 1; }
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc(&d, 16));
	HIP_TRY(hipMemset(d, 0, 16));
	for(unsigned part = 0; part < 256; part++) hipLaunchKernelGGL(dev::k_recip_check, dim3(1u << 16), dim3(256), 0, 0, part << 24, d);
	hipError_t e = hipDeviceSynchronize();
	if(e == hipSuccess) e = hipMemcpy(out2, d, 16, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	HIP_TRY(e);
	return 0;
}

int snail_debug_hostsse_device_check(int fn, int threads, uint64_t *badChunks, uint32_t *firstBadChunk) {
	if((fn != 0 && fn != 1) || !badChunks) { snail_set_error("snail_debug_hostsse_device_check: bad arguments"); return 1; }
	const unsigned *tab = nullptr;
	if(int rc = hostSseDeviceTables("snail_debug_hostsse_device_check", &tab)) return rc;
	constexpr unsigned kBatch = 4096;   // chunks per launch (2^28 inputs)
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, kBatch * sizeof(unsigned long long)));
	std::vector<unsigned long long> got(kBatch), want(kBatch);
	uint64_t bad = 0;
	uint32_t first = 0xffffffffu;
	for(unsigned c0 = 0; c0 < 65536u; c0 += kBatch) {
		hipError_t e = hipMemset(d, 0, kBatch * sizeof(unsigned long long));
		if(e == hipSuccess) { hipLaunchKernelGGL(dev_sse::k_hostsse_sums, dim3(kBatch), dim3(256), 0, 0, tab, fn, c0, d); e = hipGetLastError(); }
		hostSseChunkSums(fn, c0, kBatch, threads, want.data());   // (the host's sums while the kernel runs)
		if(e == hipSuccess) e = hipMemcpy(got.data(), d, kBatch * sizeof(unsigned long long), hipMemcpyDeviceToHost);
		if(e != hipSuccess) { (void)hipFree(d); snail_set_error("snail_debug_hostsse_device_check: %s", hipGetErrorString(e)); return 100 + (int)e; }
		for(unsigned k = 0; k < kBatch; k++)
			if(got[k] != want[k]) { if(!bad) first = c0 + k; bad++; }
	}
	(void)hipFree(d);
	*badChunks = bad;
	if(firstBadChunk) *firstBadChunk = first;
	return 0;
}

int snail_debug_dispatch_rate(int blocks, int threads, int reps, float *ms_per_launch) {
	if(blocks <= 0 || threads <= 0 || threads > 1024 || reps <= 0 || !ms_per_launch) { snail_set_error("snail_debug_dispatch_rate: bad arguments"); return 1; }
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
	hipLaunchKernelGGL(dev::k_nop, dim3(blocks), dim3(threads), 0, 0, (int *)nullptr);
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipEventRecord(e0, 0));
	for(int r = 0; r < reps; r++) hipLaunchKernelGGL(dev::k_nop, dim3(blocks), dim3(threads), 0, 0, (int *)nullptr);
	HIP_TRY(hipEventRecord(e1, 0));
	HIP_TRY(hipDeviceSynchronize());
	float ms = 0.0f;
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	*ms_per_launch = ms / reps;
	return 0;
}

// Experiment: `frames` full-frame primary launches back to back on ONE stream, with or without hipExtAnyOrderLaunch (which clears the
// AQL barrier bit so that consecutive dispatches of a queue may overlap -- hip_ext.h says "not supported on GFX9xx"; measured, not
// assumed).  No outputs are stored (null planes), nothing is deferred on a sane scene.  tools/anyorder.py.
int snail_debug_anyorder(SnailScene *s, const float cam[13], int resx, int resy, int frames, int flags, float *ms_total) {
	if(int rc = checkScene(s, "snail_debug_anyorder")) return rc;
	if(frames <= 0 || !ms_total || useDeep(s)) { snail_set_error("snail_debug_anyorder: bad arguments"); return 1; }
	DeviceGuard guard(s->device);
	dev::PrimaryArgs A;
	memset(&A, 0, sizeof(A));
	A.hostTab = nullptr;
	A.nodes = s->dNodes; A.tris = s->dTris; A.pf = (const uint4 *)s->dPF;
	A.nFrames = 1;
	A.g[0] = makeGen(cam, resx, resy);
	A.resx = resx; A.resy = resy; A.w = resx; A.h = resy;
	A.fastOK = s->fastOK && originSane(cam);
	A.pack = stackPack(s);
	A.pw = (resx + 15) / 16; A.ph = (resy + 15) / 16; A.nPackets = A.pw * A.ph;
	const int nRegions = ((A.pw + 3) / 4) * ((A.ph + 3) / 4);
	const int blocks = ((nRegions + 7) / 8) * 8 * 16;
	A.nBlocks = blocks; A.nSlots = blocks;
	if(blocks + 2 > s->deferCap) { snail_set_error("snail_debug_anyorder: trace one ordinary frame of this size first"); return 1; }
	A.defer = s->dDefer[0];
	hipStream_t st;
	HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	int relWhich = -1;
	if(A.pack && SNAIL_REL_NODES && SNAIL_NODE_PREFETCH && !useDeep(s)) {
		if(int rc = relFor(s, cam, st, &A.rel[0], &relWhich)) return rc;
	}
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipEventRecord(e0, st));
	for(int f = 0; f < frames; f++)
		hipExtLaunchKernelGGL(dev::k_primary<false>, dim3(blocks / SNAIL_BLOCK_WAVES), dim3(64 * SNAIL_BLOCK_WAVES), 0, st, nullptr, nullptr, (unsigned)flags, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e1, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipEventElapsedTime(ms_total, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
	return 0;
}

int snail_debug_occupancy(int out[4]) {
	if(!out) { snail_set_error("snail_debug_occupancy: null output"); return 1; }
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	int perCU = 0, maxBlocks = 0, cus = 0;
	HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void *)dev::k_primary<false>, 64 * SNAIL_BLOCK_WAVES, 0));
	HIP_TRY(hipDeviceGetAttribute(&maxBlocks, hipDeviceAttributeMaxBlocksPerMultiProcessor, dev));
	HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
	out[0] = perCU; out[1] = maxBlocks; out[2] = cus; out[3] = SNAIL_BLOCK_WAVES;
	return 0;
}

#endif // SNAIL_DEBUG_API

int snail_account_packets(SnailScene *s, const float cam[13], int resx, int resy, uint32_t *out8) {
	if(int rc = checkScene(s, "snail_account_packets")) return rc;
	if(!out8 || resx <= 0 || resy <= 0) { snail_set_error("snail_account_packets: null output or bad resolution"); return 1; }
	DeviceGuard guard(s->device);
	HostCallScope hc(s, "snail_account_packets");
	if(hc.rc) return hc.rc;
	const size_t bytes = (size_t)((resx + 15) / 16) * ((resy + 15) / 16) * 32;
	if(int rc = hc.reserve(HostCallScope::pad(bytes))) return rc;
	void *c = hc.carve(bytes);
	HIP_TRY(hipMemsetAsync(c, 0, bytes, hc.stream()));
	{
		SNAIL_LOCK(s);
		if(int rc = launchPrimary(s, cam, resx, resy, 0, 0, resx, resy, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, hc.stream(), (unsigned *)c)) return rc;
	}
	if(int rc = hc.get(out8, c, bytes)) return rc;
	return hc.finish(nullptr);
}

int snail_account_primary(SnailScene *s, const float cam[13], int resx, int resy, int x0, int y0, int w, int h, uint64_t out[4]) {
	if(int rc = checkScene(s, "snail_account_primary")) return rc;
	if((x0 & 15) || (y0 & 15) || w <= 0 || h <= 0 || !out) { snail_set_error("snail_account_primary: bad rect"); return 1; }
	DeviceGuard guard(s->device);
	HostCallScope hc(s, "snail_account_primary");
	if(hc.rc) return hc.rc;
	dev::AccountArgs A;
	A.nodes = s->dNodes; A.tris = s->dTris;
	A.g = makeGen(cam, resx, resy);
	A.x0 = x0; A.y0 = y0; A.pw = (w + 15) / 16; A.ph = (h + 15) / 16;
	A.out = (dev::u64 *)hc.stats();
	if(int rc = hc.zeroStats()) return rc;
	const int np = A.pw * A.ph;
	hipLaunchKernelGGL(dev::k_account, dim3((np + 3) / 4), dim3(256), 0, hc.stream(), A);
	HIP_TRY(hipGetLastError());
	return hc.finish(out);
}

} // extern "C"

#include "render_host.inc"
