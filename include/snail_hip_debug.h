/* snail_hip_debug.h -- the WORKBENCH of libsnailhip: diagnostics, experiments and environment switches.
 *
 * Not part of the drop-in contract (include/snail_hip.h).  These entry points exist only in libsnailhip_debug.so, the same sources built
 * with -DSNAIL_DEBUG_API (`make -C snail_amd/csrc debug`); tests and tools that need them load that library (snail_amd._lib.debug_lib()).
 * The workbench build also reads four environment variables, once, which the product library never does:
 *   SNAIL_DEBUG_FORCE_DEEP=1   every scene takes the DEEP kernel instantiations (second stack register pair, C++ node loop)
 *   SNAIL_DEBUG_NO_PACK=1      two-word stack entries and the node loop without record prefetch for every scene
 *   SNAIL_DEBUG_ASSUME_NESTED=1  the record-prefetching node loop also for a tree that snail_scene_create found NOT nested (wrong results
 *                              by design: tests/nonnested_env.py uses it to show that the nesting check discriminates)
 *   SNAIL_DEBUG_DYNLDS=<bytes> unused dynamic LDS per workgroup of dev::k_primary (occupancy experiments)
 */
#ifndef SNAIL_HIP_DEBUG_H
#define SNAIL_HIP_DEBUG_H
#include "snail_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A stream-ordered pause (one sleeping wave; 0..10000 us) on the current device: de-phasing experiments of pipelined frame streams
 * (measured: no gain, profiles/README.md). */
int snail_debug_delay_dev(float microseconds, void *stream);

/* One sleeping wave that reads the shader-cycle counter and the 100 MHz constant clock `microseconds` apart: d_out2[0] = shader
 * cycles, d_out2[1] = constant-clock ticks; clock = d_out2[0] / d_out2[1] x 100 MHz.  tools/ramp.py samples it beside the frames. */
int snail_debug_clock_dev(float microseconds, uint64_t *d_out2, void *stream);

/* Runs the kernels' reciprocal (v_rcp_f32 + one Newton step inside 2^-126 <= |x| < 2^126, the full division outside) on all 2^32 float bit
 * patterns against the correctly rounded 1.0f / x: out2[0] = results that differ (must be 0), out2[1] = inputs inside that range
 * (2 * 252 * 2^23).  Blocks the device for a few tens of milliseconds. */
int snail_debug_recip_check(uint64_t out2[2]);

/* SNAIL_ARITH_HOST_SSE: the DEVICE's reproduction of the host CPU's rcpps (fn 0) / rsqrtps (fn 1) over all 2^32 float bit patterns against
 * the instructions themselves, by checksums over chunks of 65536 consecutive patterns (the host's sums on `threads` threads):
 * *badChunks = chunks that differ (must be 0), *firstBadChunk = the lowest of them. */
int snail_debug_hostsse_device_check(int fn, int threads, uint64_t *badChunks, uint32_t *firstBadChunk);

/* Time per launch of an EMPTY kernel of `blocks` x `threads` (what the workgroup dispatcher alone sustains), averaged
 * over `reps` back-to-back launches on the default stream of the current device.  tools/dispatch_rate.py. */
int snail_debug_dispatch_rate(int blocks, int threads, int reps, float *ms_per_launch);

/* Experiment (tools/anyorder.py): `frames` full-frame primary launches back to back on one fresh stream with launch flags `flags`
 * (0, or hipExtAnyOrderLaunch = 1); *ms_total = HIP-event time around all of them.  No outputs are stored. */
int snail_debug_anyorder(SnailScene *, const float cam[13], int resx, int resy, int frames, int flags, float *ms_total);

/* out = {blocks of dev::k_primary the occupancy API admits per CU, the device's block limit per CU, CUs, waves per block
 * of this build}.  tools/occupancy.py. */
int snail_debug_occupancy(int out[4]);

#ifdef __cplusplus
}
#endif
#endif
