"""One frame over several GPUs of a node: one process per GPU, torch.distributed over RCCL.

Every (pixel, sample) path is independent and keyed on (seed, pixel, sample), and per-pixel results
are integer sums, so a frame may be split either way and reassembled with one exchange step:

  "samples"  rank r traces samples [begin_r, begin_r + count_r) of every pixel into its own
             full-frame int32 sums; the sums are SUM-REDUCED to rank 0 (12 bytes per pixel), which
             divides by spp and packs RGB8;
  "tiles"    rank r traces all samples of the 8x8 tiles r, r + world, r + 2*world, ... (interleaved:
             sky and ground cost very different amounts); every pixel belongs to one rank with all
             its samples, so each rank divides and packs its own tiles into a tile-major block
             (192 bytes per tile) and the blocks are GATHERED to rank 0 (3/world bytes per pixel and
             rank), which writes them into the frame.

The frame is bit-identical to the single-GPU one.  FramePipeline double-buffers the exchange: frame
k's collective runs while frame k + 1 is traced.

The tracer is duck-typed (`accumulate`, `resolve`, `pack_tiles`, `unpack_tiles`) so the CPU tests can
drive this logic over gloo with a CPU tracer standing in for the device.
"""
import rtamd


def sample_slice(total_spp, rank, world):
    """Contiguous, as-even-as-possible split of [0, total_spp) -> (begin, count)."""
    base, extra = divmod(total_spp, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def rank_params(width, height, total_spp, max_depth, seed, partition, rank, world):
    """rt_params of `rank`'s share of the frame."""
    if world == 1:
        return rtamd.params(width, height, total_spp, max_depth, seed=seed)
    if partition == "samples":
        begin, count = sample_slice(total_spp, rank, world)
        if count == 0:
            return None  # more ranks than samples: this rank has nothing to trace
        return rtamd.params(width, height, total_spp, max_depth, seed=seed, sample_begin=begin, sample_count=count)
    if partition == "tiles":
        return rtamd.params(width, height, total_spp, max_depth, seed=seed, tile_first=rank, tile_stride=world)
    raise ValueError(f"unknown partition {partition!r}")


def tile_block_bytes(width, height, first, stride):
    """Bytes of the tile-major block of the tiles first, first + stride, ... (192 per tile)."""
    n = ((width + 7) // 8) * ((height + 7) // 8)
    return 192 * ((n - first + stride - 1) // stride if first < n else 0)


class DeviceTracer:
    """Tracer over an rtamd.Context: tensors are CUDA tensors, work goes to torch's current stream."""

    def __init__(self, ctx, cam):
        import torch

        self.ctx, self.cam, self._torch = ctx, cam, torch

    def _stream(self):
        return self._torch.cuda.current_stream().cuda_stream

    def accumulate(self, params, sums):
        self.ctx.accumulate_device(self.cam, params, sums.data_ptr(), stream=self._stream())

    def accumulate_batch(self, jobs):
        """jobs: [(params, sums tensor), ...] -- one launch for all of them (rt_accumulate_batch_device)."""
        self.ctx.accumulate_batch_device([(self.cam, p, sums.data_ptr()) for p, sums in jobs], stream=self._stream())

    def resolve(self, sums, width, height, spp, rgb):
        self.ctx.resolve_device(sums.data_ptr(), width, height, spp, rgb.data_ptr(), stream=self._stream())

    def pack_tiles(self, sums, width, height, spp, first, stride, block):
        assert block.numel() >= rtamd.tile_block_bytes(width, height, first, stride)
        self.ctx.pack_tiles_device(sums.data_ptr(), width, height, spp, first, stride, block.data_ptr(),
                                   stream=self._stream())

    def unpack_tiles(self, blocks, n_blocks, block_bytes, width, height, rgb):
        assert blocks.numel() >= n_blocks * block_bytes and rgb.numel() >= width * height * 3
        self.ctx.unpack_tiles_device(blocks.data_ptr(), n_blocks, block_bytes, width, height, rgb.data_ptr(),
                                     stream=self._stream())


class _Done:
    """Stand-in for a collective's work handle when nothing is in flight."""

    def wait(self):
        return True


class FramePipeline:
    """Traces frames of one fixed shape on `world` ranks and leaves each in rank 0's RGB8 buffer.

    submit() enqueues this rank's share of the next frame and starts its exchange; the frame is
    completed (exchange waited for, rank 0 divides / writes the tiles) by the NEXT submit() or by
    finish(), so the collective of frame k overlaps the tracing of frame k + 1.  `slots` buffers
    rotate; slot s's RGB8 frame is `rgb[s]` on rank 0.  With world == 1 there is no exchange.

    host_exchange=True stages the exchanged tensors through host memory (for backends that cannot
    move device tensors, i.e. the gloo tests on one GPU).  always_exchange=True runs the pack /
    collective / unpack steps with world == 1 too (a one-rank communicator): the N > 1 code path on a
    one-GPU machine.

    frames_per_launch > 1: a rank's shares of that many consecutive frames are traced by ONE launch
    (tracer.accumulate_batch): a launch ends with the device draining its last paths (about 0.2 ms
    whatever its size), which weighs on a rank's small share of a frame; in a launch of several frames
    the waves go on with the next frame's work instead.  The frames of a group are exchanged one by one
    behind the launch and completed, like single frames, while the next group is traced.  Slots are
    then 2 x frames_per_launch.
    """

    def __init__(self, tracer, params, shape, partition, rank, world, dist, torch, device, slots=2,
                 host_exchange=False, always_exchange=False, frames_per_launch=1):
        """`params` = rank_params(...) of this rank (None: nothing to trace); `shape` = (width, height,
        frame spp); `partition` = "samples" or "tiles"."""
        p = params
        self.tracer, self.p, self.rank, self.world, self.dist, self.torch = tracer, p, rank, world, dist, torch
        self.host_exchange = host_exchange
        self.exchange = world > 1 or always_exchange
        self.tiles = self.exchange and partition == "tiles"
        self.group = max(1, int(frames_per_launch))
        if self.group > 1:
            slots = 2 * self.group
        self.slots = slots
        self.submitted = 0
        self.pending = None  # (slot, work handle) of the frame not yet completed
        self.waiting = []    # slots of the group being collected (frames_per_launch > 1)
        self.in_flight = []  # [(slot, work handle)] of the launched group not yet completed
        self.shape = tuple(shape)
        n = shape[0] * shape[1] * 3
        self.sums = [torch.zeros(n, dtype=torch.int32, device=device) for _ in range(slots)]
        self.rgb = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(slots)]
        if self.tiles:
            # every rank's block padded to the largest (rank 0's), so the gather is regular
            self.block_bytes = tile_block_bytes(shape[0], shape[1], 0, world)
            self.block = [torch.zeros(self.block_bytes, dtype=torch.uint8, device=device) for _ in range(slots)]
            self.blocks = ([torch.zeros(world * self.block_bytes, dtype=torch.uint8, device=device) for _ in range(slots)]
                           if rank == 0 else [None] * slots)

    # -- exchange primitives (device tensors, or staged through the host)

    def _reduce_sums(self, s):
        if not self.host_exchange:
            return self.dist.reduce(self.sums[s], dst=0, op=self.dist.ReduceOp.SUM, async_op=True), None
        host = self.sums[s].cpu()
        self.dist.reduce(host, dst=0, op=self.dist.ReduceOp.SUM)
        if self.rank == 0:
            self.sums[s].copy_(host)
        return _Done(), None

    def _gather_blocks(self, s):
        block = self.block[s]
        if not self.host_exchange:
            into = list(self.blocks[s].view(self.world, self.block_bytes).unbind(0)) if self.rank == 0 else None
            return self.dist.gather(block, gather_list=into, dst=0, async_op=True), None
        host = block.cpu()
        into = [self.torch.zeros_like(host) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(host, gather_list=into, dst=0)
        if self.rank == 0:
            self.blocks[s].copy_(self.torch.cat(into))
        return _Done(), None

    # -- frames

    def submit(self):
        if self.group > 1:
            return self._submit_grouped()
        s = self.submitted % self.slots
        if self.pending is not None and self.pending[0] == s:
            self._complete()  # the slot is still in flight (slots == 1)
        w, h, spp = self.shape
        self.sums[s].zero_()
        if self.p is not None:
            self.tracer.accumulate(self.p, self.sums[s])
        work = _Done()
        if self.exchange:
            if self.tiles:
                self.tracer.pack_tiles(self.sums[s], w, h, spp, self.rank, self.world, self.block[s])
                work, _ = self._gather_blocks(s)
            else:
                work, _ = self._reduce_sums(s)
        prev, self.pending = self.pending, (s, work)
        self.submitted += 1
        if prev is not None:
            self._complete(prev)
        return s

    def _submit_grouped(self):
        s = self.submitted % self.slots
        self.waiting.append(s)
        self.submitted += 1
        if len(self.waiting) == self.group:
            self._launch_group()
        return s

    def _launch_group(self):
        """One trace launch for the collected frames, their exchanges started behind it; then the group
        launched before is completed (its exchanges ran beside this launch's tracing)."""
        group, self.waiting = self.waiting, []
        if not group:
            return
        w, h, spp = self.shape
        for s in group:
            self.sums[s].zero_()
        if self.p is not None:
            self.tracer.accumulate_batch([(self.p, self.sums[s]) for s in group])
        started = []
        for s in group:
            work = _Done()
            if self.exchange:
                if self.tiles:
                    self.tracer.pack_tiles(self.sums[s], w, h, spp, self.rank, self.world, self.block[s])
                    work, _ = self._gather_blocks(s)
                else:
                    work, _ = self._reduce_sums(s)
            started.append((s, work))
        prev, self.in_flight = self.in_flight, started
        for item in prev:
            self._complete(item)

    def _complete(self, item=None):
        if item is None:
            item, self.pending = self.pending, None
        if item is None:
            return
        s, work = item
        work.wait()
        if self.rank != 0:
            return
        w, h, spp = self.shape
        if self.tiles:
            self.tracer.unpack_tiles(self.blocks[s], self.world, self.block_bytes, w, h, self.rgb[s])
        else:
            self.tracer.resolve(self.sums[s], w, h, spp, self.rgb[s])

    def finish(self):
        """Completes the frame still in flight; returns the slot of the last frame (rank 0's rgb[slot])."""
        if self.group > 1:
            self._launch_group()  # a last, shorter group
            prev, self.in_flight = self.in_flight, []
            for item in prev:
                self._complete(item)
        self._complete()
        return (self.submitted - 1) % self.slots if self.submitted else None


def trace_frame(tracer, params, sums, rgb, rank, world, dist=None, host_exchange=False):
    """One frame, start to end: zero, trace this rank's share, exchange, and leave the RGB8 frame in
    rank 0's `rgb`.

    `params` is rank_params(...) (None = nothing to trace); `sums` an int32 tensor of width*height*3,
    `rgb` a uint8 one.  Enqueues only (device tensors); the caller synchronises.
    """
    import torch

    sums.zero_()
    if params is not None:
        tracer.accumulate(params, sums)
    p = params
    if world > 1 and p is not None and p.tile_stride > 1:
        # interleaved tiles: disjoint pixels, so each rank's resolved tiles are gathered
        bb = tile_block_bytes(p.width, p.height, 0, world)
        block = torch.zeros(bb, dtype=torch.uint8, device=sums.device)
        tracer.pack_tiles(sums, p.width, p.height, p.spp, rank, world, block)
        if host_exchange:
            host = block.cpu()
            into = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, gather_list=into, dst=0)
            blocks = torch.cat(into).to(sums.device) if rank == 0 else None
        else:
            blocks = torch.zeros(world * bb, dtype=torch.uint8, device=sums.device) if rank == 0 else None
            dist.gather(block, gather_list=list(blocks.view(world, bb).unbind(0)) if rank == 0 else None, dst=0)
        if rank == 0:
            tracer.unpack_tiles(blocks, world, bb, p.width, p.height, rgb)
        return
    if world > 1:
        if host_exchange:
            host = sums.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                sums.copy_(host)
        else:
            dist.reduce(sums, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        tracer.resolve(sums, p.width, p.height, p.spp, rgb)
