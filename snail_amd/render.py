"""Tile-level rendering: the mirror of src/render.h:16-28 restricted to the traversal path, plus the
image-space data parallelism of the reference's MPI server/node pair re-done for one process per GPU.

Reference scheme (src/server.cpp:233-265, src/node.cpp:245-258,336-349): the frame is cut into
blockWidth x blockHeight = 16 x 64 pixel tiles (src/rtbase_math.h:31-34), tiles are dealt to the render
nodes by shuffled round-robin at connect time, every node renders its tiles, and each frame the
per-tile buffers travel back to rank 0.  Here: the tile map is computed identically on every rank
(seeded shuffle, no message), each rank traces the 16x16 packets of its tiles with ONE kernel launch
into packet-major buffers, and ONE collective per frame (torch.distributed gather -> RCCL send/recv over
xGMI, all 7 peer links of GPU0 in parallel; gloo on CPU for tests) brings them to rank 0, which scatters
them into the frame.  There is no exchange step inside traversal, hence no other collective."""
from __future__ import annotations

import os
import time

from dataclasses import dataclass

import numpy as np

BLOCK_WIDTH = 16     # src/rtbase_math.h:31-34
BLOCK_HEIGHT = 64
PACKET = 16


def divide_image(resx: int, resy: int, bw: int = BLOCK_WIDTH, bh: int = BLOCK_HEIGHT) -> np.ndarray:
    """DivideImage as used at src/server.cpp:235: row-major list of tiles (x, y, w, h).  The reference
    demands resx % 16 == 0 and resy % 64 == 0 (src/server.cpp:227-231); partial edge tiles are allowed
    here because packets are traced whole anyway (src/render.cpp:67-68)."""
    tiles = []
    for y in range(0, resy, bh):
        for x in range(0, resx, bw):
            tiles.append((x, y, min(bw, resx - x), min(bh, resy - y)))
    return np.asarray(tiles, dtype=np.int32).reshape(-1, 4)


def assign_tiles(n_tiles: int, n_ranks: int, seed: int = 20090501, rank0_share: float = 1.0) -> np.ndarray:
    """Shuffled round-robin of src/server.cpp:239-248: every group of consecutive tiles gets a fresh permutation of the ranks
    that take part in that round.  (The reference uses std::random_shuffle with the C library's unseeded rand(); any permutation
    sequence balances equally well, so a seeded numpy generator is used and the map is reproducible on every rank without
    communication.)

    rank0_share: the fraction of a fair share that rank 0 renders.  In the reference the server renders nothing -- it receives,
    decompresses and displays (src/server.cpp:233-265, :389-414) -- which is rank0_share = 0; 1.0 deals rank 0 in like everybody
    else.  Rank 0 sits out of round g unless floor((g + 1) * share) > floor(g * share) (an even spread of its rounds)."""
    rng = np.random.RandomState(seed)
    owner = np.empty(n_tiles, dtype=np.int32)
    share = 1.0 if n_ranks == 1 else min(1.0, max(0.0, float(rank0_share)))
    n, g = 0, 0
    while n < n_tiles:
        takes_part = int(np.floor((g + 1) * share + 1e-9)) > int(np.floor(g * share + 1e-9))
        ranks = np.arange(0 if takes_part else 1, n_ranks)
        g += 1
        order = ranks[rng.permutation(len(ranks))]
        k = min(len(order), n_tiles - n)
        owner[n:n + k] = order[:k]
        n += k
    return owner


def tile_packets(tiles: np.ndarray) -> np.ndarray:
    """Top-left corners of the 16x16 packets of each tile, tile after tile (RenderTask::Work loop order,
    src/render.cpp:67-68: y outer, x inner)."""
    out = []
    for x, y, w, h in tiles.tolist():
        for py in range(y, y + h, PACKET):
            for px in range(x, x + w, PACKET):
                out.append((px, py))
    return np.asarray(out, dtype=np.int32).reshape(-1, 2)


# y of a pad entry of a padded packet list: below every image, so the scatter kernels (which clip to the image) skip it
PAD_Y = 1 << 24


@dataclass
class ShardPlan:
    """Everything a rank needs, computed identically everywhere."""
    resx: int
    resy: int
    n_ranks: int
    tiles: np.ndarray            # [nTiles,4]
    owner: np.ndarray            # [nTiles]
    packets: list                # per rank: int32 [n_r,2]
    padded: int                  # packets per rank after padding to equal size (the collective wants equal shards)

    @staticmethod
    def make(resx: int, resy: int, n_ranks: int, seed: int = 20090501, rank0_share: float = 1.0) -> "ShardPlan":
        tiles = divide_image(resx, resy)
        owner = assign_tiles(len(tiles), n_ranks, seed, rank0_share)
        packets = [tile_packets(tiles[owner == r]) for r in range(n_ranks)]
        padded = max(1, max(len(p) for p in packets))
        return ShardPlan(resx, resy, n_ranks, tiles, owner, packets, padded)

    def padded_packets(self, rank: int) -> np.ndarray:
        """The rank's packet list at the collective's shard size.  Pad entries lie outside every image (y = PAD_Y): a rank TRACES
        only its len(packets[rank]) real packets (each packet once, so the ranks' TreeStats add up to the frame's), the pad region of
        its payload buffer stays zero, and rank 0's scatter skips it."""
        p = self.packets[rank]
        if len(p) == self.padded:
            return p
        pad = np.empty((self.padded - len(p), 2), dtype=np.int32)
        pad[:, 0] = 0
        pad[:, 1] = PAD_Y
        return np.concatenate([p.reshape(-1, 2), pad], axis=0)

    def total_rays(self) -> int:
        return sum(len(p) for p in self.packets) * 256


def gather_planes(local, rank: int, world: int, group=None, gathered=None):
    """The per-frame collective: every rank's equal-sized [4, n, 256] packet-major buffer to rank 0
    (src/node.cpp:346-349 MPI_Send / src/server.cpp:389-401 MPI_Recv in the reference).  Returns the list of
    per-rank buffers on rank 0, None elsewhere.  Backend-agnostic (nccl = RCCL on GPU, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if rank == 0 and gathered is None:
        gathered = [torch.empty_like(local) for _ in range(world)]
    dist.gather(local, gathered if rank == 0 else None, dst=0, group=group)
    return gathered if rank == 0 else None


def reduce_stats(stats, rank: int, world: int, group=None):
    """TreeStats of a frame summed onto rank 0 (src/node.cpp:358-359 MPI_Send of each node's TreeStats, src/server.cpp:406-414):
    one tiny SUM-reduce of the {intersects, iters, rays, skips} counter vector.  In place; backend-agnostic."""
    import torch.distributed as dist
    if world > 1:
        dist.reduce(stats, dst=0, op=dist.ReduceOp.SUM, group=group)
    return stats if rank == 0 else None


_STREAMS = {}


def _stream_pool(torch, dev, n):
    """The HIP streams frames are pipelined over, created once per device and shared by every renderer of the process: the runtime
    deals streams to its hardware queues round-robin as they are created, so a second set of streams would share queues with the
    first one (and with each other) and serialise."""
    key = (dev.type, dev.index)
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


class DistributedRenderer:
    """One process per GPU.  With world_size 1 the frame's hit records are traced directly in frame layout (no
    collective).  With world_size > 1 every rank traces its tiles' packets (hit records stay in its HBM, where a
    device-side shader consumes them), shades them to RGB8 with the reference's depth shading
    (Scene.shade_depth) and rank 0 gathers the RGB8 tiles -- what a render node returns in the reference
    (src/node.cpp:336-349: rgb8 per tile) -- and scatters them into the frame.  `payload="hits"` gathers the
    16-B/px hit records instead (5.3x the bytes, synchronous).

    Frames are pipelined over `slots` HIP streams (default 4 frames in flight; 3 when the gather runs asynchronously on the process group's own stream; give the process at least as many hardware queues,
    GPU_MAX_HW_QUEUES >= 6 in the environment before HIP initialises, or streams share queues and serialise): the traversal of frame i+1 fills the
    CUs that frame i's heaviest packets leave idle, and the gather of frame i (on its slot's stream, or RCCL's own)
    overlaps both.  flush() completes the frames still in flight.

    rank0_share: rank 0's fraction of a fair tile share (assign_tiles; the reference's server renders nothing = 0).
    plan_ranks: cut the frame for this many ranks although the process group has `world_size` (default: world_size) -- with
    force_collective and one rank this runs ONE share (`plan_rank`, default 0) of an N-rank frame through the whole route on a
    single-GPU box (tools/host_rate.py: what the host can issue per second, the bound of the strong-scaling points)."""

    def __init__(self, scene, resx: int, resy: int, rank: int = 0, world_size: int = 1, group=None, seed: int = 20090501,
                 payload: str = "rgb8", slots: int | None = None, stage_cpu: bool = False, force_collective: bool = False, lights7=None,
                 ambient=(0.1, 0.1, 0.1), color=(1.0, 1.0, 1.0), reflections: bool = False, feedback_order: bool = True, order_refresh: int = 16,
                 inline_collective: bool | None = None, rank0_share: float = 1.0, plan_ranks: int | None = None, plan_rank: int | None = None,
                 frames_per_launch: int = 1):
        import torch
        self.torch = torch
        self.scene = scene
        self.rank, self.world = rank, world_size
        self.group = group
        self.payload = payload
        # stage_cpu: move the payload through host memory so that a CPU backend (gloo) can carry the collective --
        # used to rehearse the multi-rank path on a box whose ranks share one GPU; the product path is RCCL on device buffers
        self.stage_cpu = stage_cpu
        # lights7 given: the rgb8 tiles are shaded with the reference's simple shading + point lights (Scene.render_whitted_packets,
        # BASELINE config 3, optionally with the mirrored bounce) instead of the depth shading
        self.lights7, self.ambient, self.color, self.reflections = lights7, ambient, color, reflections
        # force_collective: run the tile plan + shade + gather + scatter route even with ONE rank (the collective then moves rank 0's
        # buffer to itself) -- lets a single-GPU box exercise the RCCL code path of the multi-GPU bench
        self.multi = world_size > 1 or force_collective
        # inline_collective: issue the per-frame gather synchronously with respect to the SLOT's stream (torch >= 2.8 runs a
        # collective with async_op=False on the current stream instead of the process group's own): trace -> gather -> scatter are
        # then one in-order chain per slot and no fifth stream competes for a hardware queue, so 4 slots can be kept (one rank on one
        # GPU, tools/collective_overhead.py: 0.093 ms/frame against 0.106 with the asynchronous gather and 3 slots).  RCCL orders the
        # operations of one communicator across user streams itself, and if a torch build still used the group's own stream the
        # call would only be a stream-level wait: correct either way.  Default on for the rgb8 payload; SNAIL_INLINE_COLLECTIVE=0 = off.
        self.inline = (bool(inline_collective) if inline_collective is not None else os.environ.get("SNAIL_INLINE_COLLECTIVE", "1") == "1") and payload == "rgb8" and not stage_cpu
        n_plan = int(plan_ranks) if plan_ranks is not None else world_size
        if n_plan != world_size and not (force_collective and world_size == 1):
            raise ValueError("plan_ranks differs from world_size: only with force_collective and one rank (a one-rank rehearsal of an N-rank share)")
        self.plan = ShardPlan.make(resx, resy, n_plan, seed, rank0_share)
        share = rank if plan_rank is None else int(plan_rank)      # which share of the plan this process renders (rehearsals: any of them)
        self.share = share
        dev = scene._dev()
        # four ACTIVE streams is the sweet spot on this part (profiles/README.md): 4 frames in flight, or 3 where the collective's own stream
        # is the fourth (asynchronous gather)
        self.nslots = max(1, slots) if slots is not None else (3 if self.multi and not self.inline else 4)
        self.streams = _stream_pool(torch, dev, self.nslots)
        self._raw_stream_switch = self._probe_raw_stream_switch(self.streams[0]) if self.multi else False
        self.step = 0
        # frames_per_launch = B > 1 (single-GPU hit-record route): render() collects B cameras and traces them with ONE launch
        # (Scene.trace_primary_batch): the heaviest packets of all B frames start first, and a frame's tail -- its heaviest packets, ~0.2 ms
        # whatever the launch holds -- is paid once per B frames.  It buys throughput where the pipeline cannot reach its steady state (a short
        # burst of frames from idle) at the price of latency: a frame is complete when its launch is.  flush() launches what is pending.
        # On the tile-sharded route (rgb8 payload, depth shading, gather on the slot's stream) the B frames also travel in ONE collective and
        # rank 0 scatters each of them with one launch (Scene.packets_bgr_to_frame_chunked): at N ranks a rank's launch holds only 8160 / N
        # packets, so launches, collectives and the ~0.2 ms latency of a frame's heaviest packets are what bound the rate there.
        multi_now = world_size > 1 or force_collective
        batch_ok = lights7 is None and (not multi_now or (payload == "rgb8" and (self.inline or stage_cpu)))
        self.batch = max(1, min(8, int(frames_per_launch))) if batch_ok else 1
        self.pending_cams, self.pending_events, self.pending_stats = [], None, None
        self.idle = True
        # one frame buffer per slot, for both payloads: the scatters of consecutive frames run on different streams and may overlap;
        # frame / frame_rgb8 name the buffer of the frame enqueued last (complete after flush(), or once the slot's stream has drained)
        # one rank with lights7: the staged config-3 pipeline writes the rgb8 frame directly (Scene.render_whitted), no hit-record frame
        self.whitted_single = (not self.multi) and lights7 is not None
        want_hits = rank == 0 and ((not self.multi and not self.whitted_single) or (self.multi and payload == "hits"))
        self.frames = [scene.alloc_frame(resx, resy) for _ in range(self.nslots)] if want_hits else []
        self.frame = self.frames[0] if self.frames else None
        # multi-frame launches: B frame buffers per slot (slot k's buffer 0 is self.frames[k])
        self.batch_frames = [[self.frames[k]] + [scene.alloc_frame(resx, resy) for _ in range(self.batch - 1)] for k in range(self.nslots)] if (want_hits and self.batch > 1) else []
        self.frames_rgb8 = ([torch.zeros((resy, resx, 3), dtype=torch.uint8, device=dev) for _ in range(self.nslots)]
                            if (rank == 0 and ((self.multi and payload == "rgb8") or self.whitted_single)) else [])
        self.frame_rgb8 = self.frames_rgb8[0] if self.frames_rgb8 else None
        self.pending = [None] * self.nslots
        self._outputs = {}      # (slot, index in the launch) -> the buffer a launch of this renderer last wrote there (output_buffers())
        # feedback_order: dispatch the packets of a frame heaviest first, by the node visits counted in an earlier frame of the same
        # slot (round 5: derived INSIDE the launch whose costs it comes from -- Scene.trace_primary_batch / render_whitted next_order, the *_reorder_dev entry
        # points -- in place, after that launch's traversal kernel has read the old order; explicit packet lists: snail_order_from_cost_dev on the slot's stream).  The
        # order of a slot is derived after its first frame and re-derived only when the camera has moved since AND `order_refresh`
        # frames of the slot have passed: node visits of a nearby view predict the heavy packets just as well, and the one-workgroup
        # sort (20 us alone, ~100 us beside four frames in flight) then never sits in a short timed region of a fixed view.
        # Worth 10-12 % on heavy-tailed scenes (stress-1M 9.76 -> 10.79 Grays/s), neutral on the atrium (profiles/README.md).
        # On the tile-sharded rgb8 route a rank's packets all start at once (fewer packets than wave slots): the order cannot matter there.
        self.feedback = bool(feedback_order) and (not self.multi or payload == "hits")
        self.order_refresh = max(1, int(order_refresh))
        n_real = len(self.plan.packets[share]) if self.multi else 0
        self.n_real = n_real
        if self.feedback:
            n = scene.primary_slots(resx, resy) if not self.multi else n_real
            if self.whitted_single:   # the staged pipeline: one order / cost array per walking stage (Scene.render_whitted)
                self.slot_cost = [torch.zeros((scene.WHITTED_STAGES, n), dtype=torch.int32, device=dev) for _ in range(self.nslots)]
                self.order_buf = [torch.empty((scene.WHITTED_STAGES, n), dtype=torch.int32, device=dev) for _ in range(self.nslots)]
            else:
                self.slot_cost = [torch.zeros(max(1, n), dtype=torch.int32, device=dev)[:n] for _ in range(self.nslots)]
                self.order_buf = [torch.empty(max(1, n), dtype=torch.int32, device=dev)[:n] for _ in range(self.nslots)]
            self.order_valid = [False] * self.nslots
            self.order_cam = [None] * self.nslots       # camera the slot's order was derived from
            self.order_age = [0] * self.nslots          # frames of the slot since then
            self.order_exact = [False] * self.nslots    # the slot's order was derived (sorted) from the costs of the camera it is used for
            if n == 0:
                self.feedback = False
        if self.multi:
            n = self.plan.padded
            # the launch list holds the rank's REAL packets only (every packet traced and counted once); buffers have the shard size
            self.packet_xy = torch.from_numpy(np.ascontiguousarray(self.plan.packets[share].reshape(-1, 2))).to(dev)
            if payload == "hits":
                # per slot one buffer [4, n, 256] (t, u, v, triId) so that the four planes can travel in ONE collective
                self.local = [torch.zeros((4, n, 256), dtype=torch.float32, device=dev) for _ in range(self.nslots)]
                self.planes = [(b[0][:n_real], b[1][:n_real], b[2][:n_real], b[3][:n_real].view(torch.int32)) for b in self.local]
            self.bgr = [torch.zeros((n, 256, 3), dtype=torch.uint8, device=dev) for _ in range(self.nslots)]
            self.bgr_real = [b[:n_real] for b in self.bgr]
            if self.batch > 1:   # multi-frame launches: B frames per slot, one payload buffer [B, n, 256, 3] = one collective
                self.bgrB = [torch.zeros((self.batch, n, 256, 3), dtype=torch.uint8, device=dev) for _ in range(self.nslots)]
                self.bgrB_real = [[b[k][:n_real] for k in range(self.batch)] for b in self.bgrB]
            if rank == 0:
                nw = world_size
                # rank r's shard of the receive buffer is scattered with ITS padded list; pad entries (y = PAD_Y) are skipped
                ranks_in_group = list(range(nw)) if n_plan == nw else [share]
                self.all_xy = [torch.from_numpy(self.plan.padded_packets(r)).to(dev) for r in ranks_in_group]
                if payload == "hits":
                    self.gathered = [[torch.empty_like(self.local[0]) for _ in range(nw)] for _ in range(self.nslots)]
                else:
                    # ONE contiguous receive buffer per slot, handed to the collective as per-rank views, so that rank 0
                    # scatters all ranks' tiles with a single launch over the concatenated packet list
                    self.gathered_all = [torch.empty((nw, n, 256, 3), dtype=torch.uint8, device=dev) for _ in range(self.nslots)]
                    self.gathered = [[g[r] for r in range(nw)] for g in self.gathered_all]
                    self.all_xy_cat = torch.cat(self.all_xy, dim=0).contiguous()
                    if self.batch > 1:
                        self.gatheredB_all = [torch.empty((nw, self.batch, n, 256, 3), dtype=torch.uint8, device=dev) for _ in range(self.nslots)]
                        self.gatheredB = [[g[r] for r in range(nw)] for g in self.gatheredB_all]
                        self.framesB_rgb8 = [[torch.zeros((resy, resx, 3), dtype=torch.uint8, device=dev) for _ in range(self.batch)] for _ in range(self.nslots)]
            else:
                self.gathered = [None] * self.nslots

    def rays_per_frame(self) -> int:
        return self.plan.total_rays()

    def local_rays(self) -> int:
        return len(self.plan.packets[self.share]) * 256

    def _finish(self, slot):
        """complete the gather issued from `slot` and (rank 0) scatter it; runs on the slot's stream"""
        work = self.pending[slot]
        if work is None:
            return
        self.pending[slot] = None
        work.wait()
        if self.rank == 0:
            self.frame_rgb8 = self.frames_rgb8[slot]
            self._outputs[(slot, 0)] = self.frame_rgb8
            self.scene.packets_bgr_to_frame(self.all_xy_cat, self.gathered_all[slot].view(-1, 256, 3), self.frame_rgb8)

    def output_buffers(self):
        """The buffers this renderer's launches have written (rank 0; complete after flush()): HitFrames on the single-GPU hit-record
        route, [resy, resx, 3] uint8 frames otherwise -- one per slot and per frame of a multi-frame launch.  bench.py hashes them after its
        timed region."""
        return [self._outputs[k] for k in sorted(self._outputs)]

    def poison_outputs(self):
        """Overwrite every buffer output_buffers() names with a poison pattern (all bits set) and forget them: what output_buffers() returns afterwards was
        written by launches issued AFTER this call.  Call between flush() and the next render(); synchronises the device.  bench.py does this in front of its
        timed region, so that a launch that silently skipped work inside it could not pass on a buffer an earlier (warm-up) frame had filled."""
        torch = self.torch
        for b in self.output_buffers():
            for t in ((b.t, b.u, b.v, b.tri_id) if hasattr(b, "tri_id") else (b,)):
                t.view(torch.uint8).fill_(0xff) if t.dtype != torch.uint8 else t.fill_(0xff)
        torch.cuda.synchronize()
        self._outputs = {}

    def _order_for(self, slot, cam):
        """the dispatch order this frame uses (or None), and whether to re-derive the slot's order from this frame's costs (key or None).  self.order_exact[slot]
        then says what kind the derivation is to be: a camera that has NOT moved since the slot's order was derived gets -- once -- the order of its own exact
        costs, sorted whatever their shape (SNAIL_ORDER_SORTED: the last frames of a run end on their heaviest packets, which then started first); a moving
        camera's orders are predictions and take the library's rule (sorted only for heavy-tailed costs)."""
        key = cam.as_array13().tobytes()
        if not self.order_valid[slot]:
            self.order_exact[slot] = False
            return None, key
        self.order_age[slot] += 1
        if self.order_cam[slot] == key:
            if self.order_exact[slot]:
                return self.order_buf[slot], None
            self.order_exact[slot] = True
            return self.order_buf[slot], key
        refresh = self.order_age[slot] >= self.order_refresh
        if refresh:
            self.order_exact[slot] = False
        return self.order_buf[slot], (key if refresh else None)

    def _order_refreshed(self, slot, key):
        """the launch that was just enqueued derives the slot's next order itself (next_order = the slot's order buffer: the *_reorder_dev entry
        points -- no kernel launch of its own, round 5)"""
        self.order_valid[slot], self.order_cam[slot], self.order_age[slot] = True, key, 0

    def _refresh_order(self, slot, st, key):
        """the stand-alone sort (snail_order_from_cost_dev), for the routes without a *_reorder_dev form: explicit packet lists"""
        if self.whitted_single:
            for k in range(self.scene.WHITTED_STAGES if self.reflections else 2):     # stages 2, 3 exist with the mirrored bounce only
                self.scene.order_from_cost(self.slot_cost[slot][k], self.order_buf[slot][k], stream=st, exact=self.order_exact[slot])
        else:
            self.scene.order_from_cost(self.slot_cost[slot], self.order_buf[slot], stream=st, exact=self.order_exact[slot])
        self.order_valid[slot], self.order_cam[slot], self.order_age[slot] = True, key, 0

    def render(self, cam, stats=None, events=None):
        """Enqueue one frame; returns immediately.  `events` = optional (start, end) torch events recorded around the
        traversal launch on the stream it is launched on (bench.py)."""
        torch = self.torch
        sc, p = self.scene, self.plan
        if self.batch > 1:       # multi-frame launches: collect, launch when the batch is full (or at flush())
            self.pending_cams.append(cam)
            if events and self.pending_events is None: self.pending_events = events
            if stats is not None: self.pending_stats = stats
            if len(self.pending_cams) == self.batch: self._launch_batch()
            return self.frame_rgb8 if self.multi else self.frame
        slot = self.step % self.nslots
        self.step += 1
        st = self.streams[slot]
        self.idle = False
        if not self.multi:      # every call below takes the stream explicitly: no stream context to enter (host time per frame matters)
            if events: events[0].record(st)
            if self.whitted_single:
                order, key = self._order_for(slot, cam) if self.feedback else (None, None)
                self.frame_rgb8 = sc.render_whitted(cam, p.resx, p.resy, self.lights7, self.ambient, self.color, out=self.frames_rgb8[slot], stats=stats, stream=st,
                                                    reflections=self.reflections, order=order, slot_cost=self.slot_cost[slot] if self.feedback else None,
                                                    next_order=self.order_buf[slot] if key is not None else None, order_exact=self.feedback and self.order_exact[slot])   # (a refresh: derived inside this launch, in place)
                self._outputs[(slot, 0)] = self.frame_rgb8
                if events: events[1].record(st)
                if key is not None: self._order_refreshed(slot, key)
                return self.frame_rgb8
            if self.feedback:
                order, key = self._order_for(slot, cam)
                if key is not None:      # a refresh: the one-frame form of the multi-frame launch derives the next order inside the launch, in place
                    out = sc.trace_primary_batch([cam], p.resx, p.resy, [self.frames[slot]], stats=stats, stream=st, order=order, slot_cost=self.slot_cost[slot],
                                                 next_order=self.order_buf[slot], order_exact=self.order_exact[slot])[0]
                    self._order_refreshed(slot, key)
                else:
                    out = sc.trace_primary(cam, p.resx, p.resy, out=self.frames[slot], stats=stats, stream=st, order=order, slot_cost=self.slot_cost[slot])
                if events: events[1].record(st)
            else:
                out = sc.trace_primary(cam, p.resx, p.resy, out=self.frames[slot], stats=stats, stream=st)
                if events: events[1].record(st)
            self.frame = out
            self._outputs[(slot, 0)] = out
            return out
        # the collective runs on the CURRENT stream: make the slot's stream current for the rest of the call.  (torch.cuda.stream(st) as
        # a context manager costs the host ~16 us per frame -- it looks the device up through is_available() twice -- of the ~45 us a frame
        # costs it on this route, tools/host_profile.py; at 4-8 ranks the host's issue rate is what bounds the frame rate.)
        return self._on_stream(st, self._render_multi, cam, stats, events, slot, st)

    def _on_stream(self, st, fn, *args):
        """fn(*args) with `st` as the current stream (what torch.distributed launches its collectives on), the caller's stream restored after"""
        torch = self.torch
        if not self._raw_stream_switch:                                  # another torch build: the documented (slower) way
            with torch.cuda.stream(st):
                return fn(*args)
        prev = torch._C._cuda_getCurrentStream(st.device_index)          # (stream_id, device_index, device_type) of the caller's stream
        set_raw = torch._C._cuda_setStream
        set_raw(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)
        try:
            return fn(*args)
        finally:
            set_raw(stream_id=prev[0], device_index=prev[1], device_type=prev[2])

    def _probe_raw_stream_switch(self, st) -> bool:
        """Once per renderer: do the private accessors behave as _on_stream assumes (a 3-tuple, keyword arguments, the switch observable
        through the public API and undone afterwards)?  Anything else -> torch.cuda.stream(st)."""
        torch = self.torch
        try:
            prev = torch._C._cuda_getCurrentStream(st.device_index)
            if len(prev) != 3:
                return False
            torch._C._cuda_setStream(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)
            try:
                ok = torch.cuda.current_stream(st.device_index).cuda_stream == st.cuda_stream
            finally:
                torch._C._cuda_setStream(stream_id=prev[0], device_index=prev[1], device_type=prev[2])
            return bool(ok)
        except Exception:
            return False

    def _render_multi(self, cam, stats, events, slot, st):
        torch = self.torch
        import torch.distributed as dist
        sc, p = self.scene, self.plan
        if True:
            self._finish(slot)                       # the slot's buffers are free again after this
            if events: events[0].record(st)
            if self.payload == "hits":
                if self.feedback:
                    order, key = self._order_for(slot, cam)
                    sc.trace_packets(cam, p.resx, p.resy, self.packet_xy, out=self.planes[slot], stats=stats, stream=st, order=order, slot_cost=self.slot_cost[slot])
                    if key is not None: self._refresh_order(slot, st, key)
                else:
                    sc.trace_packets(cam, p.resx, p.resy, self.packet_xy, out=self.planes[slot], stats=stats, stream=st)
            elif self.lights7 is not None:   # rgb8 payload, config-3 shading
                sc.render_whitted_packets(cam, p.resx, p.resy, self.packet_xy, self.lights7, self.ambient, self.color, out=self.bgr_real[slot], stats=stats,
                                          stream=st, reflections=self.reflections)
            else:   # rgb8 payload: the depth shading is fused into the traversal kernel's epilogue
                sc.trace_packets_shaded(cam, p.resx, p.resy, self.packet_xy, out=self.bgr_real[slot], stats=stats, stream=st)
            if events: events[1].record(st)
            if self.payload == "hits":
                gather_planes(self.local[slot], self.rank, self.world, self.group, self.gathered[slot] if self.rank == 0 else None)
                if self.rank == 0:
                    self.frame = self.frames[slot]
                    self._outputs[(slot, 0)] = self.frame
                    for k, xy in enumerate(self.all_xy):
                        g = self.gathered[slot][k]
                        sc.packets_to_frame(xy, (g[0], g[1], g[2], g[3].view(torch.int32)), self.frame, stream=st)
                return self.frame
            if self.stage_cpu:
                # rehearsal transport: same buffers and the same completion path as the RCCL route, bytes moved through the host
                host = self.bgr[slot].cpu()                      # synchronises the slot stream
                glist = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(host, glist, dst=0, group=self.group)
                if self.rank == 0:
                    for r in range(self.world):
                        self.gathered[slot][r].copy_(glist[r], non_blocking=False)

                class _Done:
                    def wait(self):
                        return True
                self.pending[slot] = _Done()
                return self.frame_rgb8
            if self.inline:
                dist.gather(self.bgr[slot], self.gathered[slot] if self.rank == 0 else None, dst=0, group=self.group, async_op=False)
                if self.rank == 0:
                    self.frame_rgb8 = self.frames_rgb8[slot]
                    self._outputs[(slot, 0)] = self.frame_rgb8
                    sc.packets_bgr_to_frame(self.all_xy_cat, self.gathered_all[slot].view(-1, 256, 3), self.frame_rgb8)
                return self.frame_rgb8
            self.pending[slot] = dist.gather(self.bgr[slot], self.gathered[slot] if self.rank == 0 else None, dst=0, group=self.group, async_op=True)
        return self.frame_rgb8

    def _launch_batch(self):
        """trace the pending cameras with ONE launch on the next slot's stream"""
        cams, events, stats = self.pending_cams, self.pending_events, self.pending_stats
        self.pending_cams, self.pending_events, self.pending_stats = [], None, None
        if not cams:
            return
        sc, p = self.scene, self.plan
        slot = self.step % self.nslots
        self.step += 1
        st = self.streams[slot]
        self.idle = False
        if self.multi:
            return self._on_stream(st, self._launch_batch_multi, cams, events, stats, slot, st)
        outs = self.batch_frames[slot][:len(cams)]
        if events: events[0].record(st)
        if self.feedback:
            order, key = self._order_for(slot, cams[0])
            sc.trace_primary_batch(cams, p.resx, p.resy, outs, stats=stats, stream=st, order=order, slot_cost=self.slot_cost[slot],
                                   next_order=self.order_buf[slot] if key is not None else None, order_exact=self.order_exact[slot])      # (a refresh: derived inside this launch, in place)
            if events: events[1].record(st)
            if key is not None: self._order_refreshed(slot, key)
        else:
            sc.trace_primary_batch(cams, p.resx, p.resy, outs, stats=stats, stream=st)
            if events: events[1].record(st)
        self.frame = outs[-1]
        for k, o in enumerate(outs): self._outputs[(slot, k)] = o

    def _launch_batch_multi(self, cams, events, stats, slot, st):
        """the tile-sharded route for a batch: one launch (this rank's packets x the batch's cameras), one collective, one scatter per frame"""
        import torch.distributed as dist
        torch = self.torch
        sc, p = self.scene, self.plan
        if events: events[0].record(st)
        sc.trace_packets_shaded_batch(cams, p.resx, p.resy, self.packet_xy, self.bgrB_real[slot][:len(cams)], stats=stats, stream=st)
        if events: events[1].record(st)
        if self.stage_cpu:    # rehearsal transport (gloo): the payload moves through the host
            host = self.bgrB[slot].cpu()
            glist = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, glist, dst=0, group=self.group)
            if self.rank == 0:
                for r in range(self.world):
                    self.gatheredB[slot][r].copy_(glist[r], non_blocking=False)
        else:
            dist.gather(self.bgrB[slot], self.gatheredB[slot] if self.rank == 0 else None, dst=0, group=self.group, async_op=False)
        if self.rank == 0:
            n = self.plan.padded
            for k in range(len(cams)):
                self.frame_rgb8 = self.framesB_rgb8[slot][k]
                self._outputs[(slot, k)] = self.frame_rgb8
                sc.packets_bgr_to_frame_chunked(self.all_xy_cat, n, self.batch * n * 768, self.gatheredB_all[slot][0, k], self.frame_rgb8, stream=st)
        return self.frame_rgb8

    def measure_gather(self, iters: int = 20):
        """The collective ALONE (every rank calls it, after flush()): `iters` gathers of slot 0's payload buffer with nothing else in flight;
        returns, on rank 0, {"bytes_per_collective": what rank 0 receives from the OTHER ranks, "ms": mean time of one collective, "GBps"}
        -- the inbound rate of rank 0's links when the backend is RCCL, of the host staging when it is the gloo rehearsal -- elsewhere None."""
        import time
        import torch.distributed as dist
        torch = self.torch
        if not self.multi or self.payload != "rgb8":
            return None
        send = self.bgrB[0] if self.batch > 1 else self.bgr[0]
        recv = (self.gatheredB[0] if self.batch > 1 else self.gathered[0]) if self.rank == 0 else None
        def one():
            if self.stage_cpu:
                host = send.cpu()
                dist.gather(host, [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None, dst=0, group=self.group)
            else:
                dist.gather(send, recv, dst=0, group=self.group, async_op=False)
        one()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier(group=self.group)
        t0 = time.perf_counter()
        for _ in range(iters):
            one()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / iters
        if self.rank != 0:
            return None
        nbytes = int(send.numel()) * int(send.element_size()) * (self.world - 1)
        return {"bytes_per_collective": nbytes, "ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 2) if ms > 0 else None,
                "frames_per_collective": self.batch, "transport": "host staging (gloo rehearsal)" if self.stage_cpu else "device buffers"}

    def reduce_stats(self, stats):
        """Sum the ranks' TreeStats accumulators (device int64[4], Scene.new_stats()) onto rank 0; call after flush()."""
        if self.stage_cpu and self.world > 1:
            host = stats.cpu()
            out = reduce_stats(host, self.rank, self.world, self.group)
            if out is not None:
                stats.copy_(out)
            return stats if self.rank == 0 else None
        return reduce_stats(stats, self.rank, self.world, self.group)

    def flush(self):
        """Complete every frame still in flight (call after the last render() of a sequence)."""
        torch = self.torch
        if self.batch > 1:
            self._launch_batch()
        for k in range(self.nslots):     # oldest frame first, so that frame_rgb8 ends up naming the newest
            slot = (self.step + k) % self.nslots
            with torch.cuda.stream(self.streams[slot]):
                self._finish(slot)
        for st in self.streams:
            st.synchronize()
        self.idle = True
        return self.frame_rgb8 if self.frame_rgb8 is not None else self.frame
