"""One frame over several GPUs of a node: one process per GPU, torch.distributed over RCCL.

Every (pixel, sample) path is independent and keyed on (seed, pixel, sample), and per-pixel results
are integer sums, so a frame may be split either way and reassembled with one sum-reduce:

  "samples"  rank r traces samples [begin_r, begin_r + count_r) of every pixel;
  "tiles"    rank r traces all samples of the 8x8 tiles r, r + world, r + 2*world, ...

Either way each rank adds into its own full-frame int32 sums tensor and the frame has one exchange
step.  Sample slices: the sums are sum-reduced to rank 0 (12 bytes per pixel), which divides by spp
and packs RGB8.  Tiles: every pixel belongs to one rank with all its samples, so each rank resolves
its own sums -- pixels of other ranks' tiles stay 0 -- and the RGB8 frames are sum-reduced to rank 0
(3 bytes per pixel).  The frame is bit-identical to the single-GPU one.

The tracer is duck-typed (`accumulate(params, sums)`, `resolve(sums, spp, rgb)`) so the CPU tests can
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


class DeviceTracer:
    """Tracer over an rtamd.Context: tensors are CUDA tensors, work goes to torch's current stream."""

    def __init__(self, ctx, cam):
        import torch

        self.ctx, self.cam, self._torch = ctx, cam, torch

    def accumulate(self, params, sums):
        stream = self._torch.cuda.current_stream().cuda_stream
        self.ctx.accumulate_device(self.cam, params, sums.data_ptr(), stream=stream)

    def resolve(self, sums, width, height, spp, rgb):
        stream = self._torch.cuda.current_stream().cuda_stream
        self.ctx.resolve_device(sums.data_ptr(), width, height, spp, rgb.data_ptr(), stream=stream)


def trace_frame(tracer, params, sums, rgb, rank, world, dist=None):
    """One frame: zero, trace this rank's share, exchange, and leave the RGB8 frame in rank 0's `rgb`.

    `params` is rank_params(...) (None = nothing to trace); `sums` an int32 tensor of width*height*3,
    `rgb` a uint8 one (the frame on rank 0; scratch on the other ranks of a tile partition).
    Enqueues only; the caller synchronises.
    """
    sums.zero_()
    if params is not None:
        tracer.accumulate(params, sums)
    if world > 1 and params is not None and params.tile_stride > 1:
        # interleaved tiles: disjoint pixels, so the resolved bytes can be exchanged instead of the sums
        tracer.resolve(sums, params.width, params.height, params.spp, rgb)
        dist.reduce(rgb, dst=0, op=dist.ReduceOp.SUM)
        return
    if world > 1:
        dist.reduce(sums, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        p = params
        tracer.resolve(sums, p.width, p.height, p.spp, rgb)
