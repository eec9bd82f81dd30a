"""The multi-GPU frame assembly (partition -> trace -> sum-reduce -> resolve) over gloo, world size 2,
with the CPU oracle standing in for the device tracer.  Checks that both partitionings reproduce the
single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dist_render
import oracle_lib as O
import rtamd

W, H, SPP, DEPTH, SEED = 40, 26, 5, 6, 11


class OracleTracer:
    """accumulate/resolve on CPU tensors with the oracle (test stand-in for dist_render.DeviceTracer)."""

    def __init__(self, scene, cam):
        self.scene, self.cam = scene, cam

    def accumulate(self, p, sums):
        count = p.sample_count or p.spp
        stride = p.tile_stride or 1
        tiles_x, tiles_y = (p.width + 7) // 8, (p.height + 7) // 8
        acc = sums.numpy().view(np.uint32).reshape(p.height, p.width, 3)
        for t in range(p.tile_first, tiles_x * tiles_y, stride):
            ty, tx = divmod(t, tiles_x)
            tile = (tx * 8, ty * 8, min(8, p.width - tx * 8), min(8, p.height - ty * 8))
            _, s, _ = O.trace_tile(self.scene, self.cam, p.width, p.height, p.spp, p.max_depth, O.RNG_COUNTER,
                                   seed=p.seed, tile=tile, s0=p.sample_begin, s1=p.sample_begin + count,
                                   want_sums=True)
            acc += s

    def accumulate_batch(self, jobs):
        for p, sums in jobs:
            self.accumulate(p, sums)

    def resolve(self, sums, width, height, spp, rgb):
        q = sums.numpy().view(np.uint32) // spp
        rgb.numpy()[:] = np.minimum(q, 255).astype(np.uint8)

    def pack_tiles(self, sums, width, height, spp, first, stride, block):
        """Tile-major block of the tiles first, first + stride, ...: 192 bytes each, (ly*8 + lx)*3."""
        q = np.minimum(sums.numpy().view(np.uint32).reshape(height, width, 3) // spp, 255).astype(np.uint8)
        tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
        out = block.numpy()
        out[:] = 0
        for k, t in enumerate(range(first, tiles_x * tiles_y, stride)):
            ty, tx = divmod(t, tiles_x)
            tile = np.zeros((8, 8, 3), dtype=np.uint8)
            part = q[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
            tile[:part.shape[0], :part.shape[1]] = part
            out[k * 192:(k + 1) * 192] = tile.reshape(-1)

    def unpack_tiles(self, blocks, n_blocks, block_bytes, width, height, rgb):
        img = rgb.numpy().reshape(height, width, 3)
        tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
        src = blocks.numpy()
        for t in range(tiles_x * tiles_y):
            ty, tx = divmod(t, tiles_x)
            b, k = t % n_blocks, t // n_blocks
            tile = src[b * block_bytes + k * 192:b * block_bytes + (k + 1) * 192].reshape(8, 8, 3)
            h, w = min(8, height - ty * 8), min(8, width - tx * 8)
            img[ty * 8:ty * 8 + h, tx * 8:tx * 8 + w] = tile[:h, :w]


def _worker(rank, world, port, partition, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, cam = O.default_scene(), O.default_camera()
    tracer = OracleTracer(scene, cam)
    p = dist_render.rank_params(W, H, SPP, DEPTH, SEED, partition, rank, world)
    sums = torch.zeros(W * H * 3, dtype=torch.int32)
    rgb = torch.zeros(W * H * 3, dtype=torch.uint8)
    for _ in range(2):  # twice: the frame must not depend on what the buffers held
        dist_render.trace_frame(tracer, p, sums, rgb, rank, world, dist)
    dist.barrier()
    # the pipelined form: three frames through two rotating slots, exchange of one overlapping the next
    pipe = dist_render.FramePipeline(tracer, p, (W, H, SPP), partition, rank, world, dist, torch, "cpu", slots=2)
    for _ in range(3):
        pipe.submit()
    last = pipe.finish()
    dist.barrier()
    # several frames per launch: five frames in groups of two (the last group is short), four slots
    grouped = dist_render.FramePipeline(tracer, p, (W, H, SPP), partition, rank, world, dist, torch, "cpu",
                                        frames_per_launch=2)
    for _ in range(5):
        grouped.submit()
    last_grouped = grouped.finish()
    dist.barrier()
    if rank == 0:
        np.save(out_path, rgb.numpy().reshape(H, W, 3))
        np.save(out_path + ".pipe.npy", torch.stack(pipe.rgb).numpy().reshape(2, H, W, 3))
        np.save(out_path + ".grouped.npy", torch.stack(grouped.rgb).numpy().reshape(4, H, W, 3))
        assert last == 0 and last_grouped == 0 and grouped.slots == 4
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("partition", ["samples", "tiles"])
def test_two_rank_frame_equals_single_process_frame(tmp_path, partition):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), partition, out), nprocs=2, join=True)
    got = np.load(out)
    want, _, _ = O.trace_tile(O.default_scene(), O.default_camera(), W, H, SPP, DEPTH, O.RNG_COUNTER, seed=SEED)
    assert np.array_equal(got, want)
    piped = np.load(out + ".pipe.npy")
    assert np.array_equal(piped[0], want) and np.array_equal(piped[1], want)
    grouped = np.load(out + ".grouped.npy")
    assert all(np.array_equal(g, want) for g in grouped)


def test_tile_blocks_cover_every_tile_once():
    for w, h in ((40, 26), (8, 8), (1, 1), (1920, 1080), (61, 43)):
        n = ((w + 7) // 8) * ((h + 7) // 8)
        for world in (1, 2, 3, 8):
            sizes = [dist_render.tile_block_bytes(w, h, r, world) for r in range(world)]
            assert sum(sizes) == 192 * n and max(sizes) == sizes[0]
            assert sizes == [rtamd.tile_block_bytes(w, h, r, world) for r in range(world)]


def test_sample_slices_cover_the_range_exactly():
    for total in (1, 2, 5, 64, 100, 1024):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                b, c = dist_render.sample_slice(total, r, world)
                seen.extend(range(b, b + c))
            assert seen == list(range(total))


def test_rank_params():
    p = dist_render.rank_params(W, H, 10, 3, 7, "samples", 1, 4)
    assert (p.sample_begin, p.sample_count, p.spp) == (3, 3, 10)
    p = dist_render.rank_params(W, H, 10, 3, 7, "tiles", 3, 4)
    assert (p.tile_first, p.tile_stride, p.sample_count) == (3, 4, 0)
    assert dist_render.rank_params(W, H, 2, 3, 7, "samples", 3, 4) is None  # more ranks than samples
    with pytest.raises(ValueError):
        dist_render.rank_params(W, H, 2, 3, 7, "rows", 0, 2)
