"""Two processes, one device each in production; here both on GPU 0 with gloo carrying the reduce
(RCCL needs one device per rank).  The real library traces each rank's share; the assembled frame
must equal the single-process frame and the oracle's, for both partitionings.  RCCL itself runs as a
one-rank communicator: the same pack / collective / unpack calls on device tensors."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
import scenes

W, H, SPP, DEPTH, SEED = 96, 54, 6, 8, 21


def _worker(rank, world, port, partition, out_path):
    import torch
    import torch.distributed as dist

    import dist_render
    import rtamd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = rtamd.Context(0)
    ctx.set_scene(scenes.random_spheres(500))
    tracer = dist_render.DeviceTracer(ctx, rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA))
    p = dist_render.rank_params(W, H, SPP, DEPTH, SEED, partition, rank, world)
    sums = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
    rgb = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
    # gloo moves host tensors: the exchanged buffers are staged through the host here (over RCCL,
    # one device per rank, they stay on the devices)
    for _ in range(2):
        dist_render.trace_frame(tracer, p, sums, rgb, rank, world, dist, host_exchange=True)
    torch.cuda.synchronize()
    dist.barrier()
    pipe = dist_render.FramePipeline(tracer, p, (W, H, SPP), partition, rank, world, dist, torch, "cuda", slots=2,
                                     host_exchange=True)
    for _ in range(3):
        pipe.submit()
    pipe.finish()
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        np.save(out_path, rgb.cpu().numpy().reshape(H, W, 3))
        np.save(out_path + ".pipe.npy", torch.stack(pipe.rgb).cpu().numpy().reshape(2, H, W, 3))
    ctx.close()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("partition", ["samples", "tiles"])
def test_two_processes_on_the_device_assemble_the_oracle_frame(tmp_path, partition):
    import torch.multiprocessing as mp

    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), partition, out), nprocs=2, join=True)
    got = np.load(out)
    sc = scenes.random_spheres(500)
    want, _, _ = O.trace_tile(sc, O.camera(*scenes.RANDOM_SPHERES_CAMERA), W, H, SPP, DEPTH, O.RNG_COUNTER, seed=SEED)
    assert np.array_equal(got, want)
    piped = np.load(out + ".pipe.npy")
    assert np.array_equal(piped[0], want) and np.array_equal(piped[1], want)


@pytest.mark.parametrize("extra,rays", [(["--partition", "tiles"], 301321686), (["--partition", "samples"], 301321686),
                                        (["--scaling", "weak"], None)])
def test_bench_multi_rank_path_rehearsed_on_one_gpu(extra, rays):
    """bench.py's N > 1 path (rank shares, pipelined exchange, max-over-ranks timing, the JSON line) with
    two ranks sharing the GPU over gloo: not a measurement, but every line of that path runs.  The
    strong-scaling frame must trace exactly the single-GPU frame's rays."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", *extra]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=280, cwd=root).stdout
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["unit"] == "Mrays/s"
    assert "REHEARSAL" in line["data"]
    if rays is not None:
        assert line["scaling"] == "strong" and line["rays_per_step"] == rays and line["config"]["frame_spp"] == 64
    else:
        assert line["scaling"] == "weak" and line["config"]["frame_spp"] == 128 and line["rays_per_step"] > 2 * 301000000
    assert line["value"] > 0 and line["ms_per_step"] > 0


def _rccl_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    import dist_render
    import rtamd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    ctx = rtamd.Context(0)
    ctx.set_scene(scenes.random_spheres(500))
    tracer = dist_render.DeviceTracer(ctx, rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA))
    frames = []
    for partition in ("samples", "tiles"):
        p = dist_render.rank_params(W, H, SPP, DEPTH, SEED, partition, rank, world)
        pipe = dist_render.FramePipeline(tracer, p, (W, H, SPP), partition, rank, world, dist, torch, "cuda", slots=2,
                                         always_exchange=True)
        for _ in range(3):
            pipe.submit()
        pipe.finish()
        torch.cuda.synchronize()
        dist.barrier()
        frames.append(torch.stack(pipe.rgb).cpu().numpy().reshape(2, H, W, 3))
        # the same with three frames to a launch: seven frames = two whole groups and a short one, six slots
        grouped = dist_render.FramePipeline(tracer, p, (W, H, SPP), partition, rank, world, dist, torch, "cuda",
                                            always_exchange=True, frames_per_launch=3)
        for _ in range(7):
            grouped.submit()
        grouped.finish()
        torch.cuda.synchronize()
        dist.barrier()
        for slot in range(grouped.slots):
            assert torch.equal(grouped.rgb[slot], pipe.rgb[0]), (partition, slot)
    np.save(out_path, np.stack(frames))
    ctx.close()
    dist.destroy_process_group()


def test_exchange_path_over_a_one_rank_rccl_communicator(tmp_path):
    """The device-tensor exchange (async reduce of the int32 sums, async gather of the tile blocks,
    barrier) through RCCL itself, with the one rank a one-GPU machine allows."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "rccl.npy")
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    got = np.load(out)
    sc = scenes.random_spheres(500)
    want, _, _ = O.trace_tile(sc, O.camera(*scenes.RANDOM_SPHERES_CAMERA), W, H, SPP, DEPTH, O.RNG_COUNTER, seed=SEED)
    for part in range(2):
        for slot in range(2):
            assert np.array_equal(got[part, slot], want)


@pytest.mark.parametrize("partition,per_launch,steps", [("tiles", 1, 2), ("samples", 1, 2), ("tiles", 2, 5), ("samples", 4, 5)])
def test_bench_exchange_path_alone_over_rccl(partition, per_launch, steps):
    """bench.py's N > 1 steps (pack, collective, unpack / reduce, resolve) over a one-rank RCCL communicator,
    one frame per launch and several (whole groups and a last, shorter one)."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", str(steps), "--warmup", "1", "--exchange-alone",
           "--partition", partition, "--no-cpu-baseline", "--frames-per-launch", str(per_launch)]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=280, cwd=root, env=env).stdout
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rays_per_step"] == 301321686 and line["steps"] == steps
    assert line["config"]["frames_per_launch"] == per_launch
    assert line["config"]["partition"] == partition and "not the N = 1 measurement" in line["data"]
    assert ("gather" if partition == "tiles" else "reduce") in line["config"]["collective"]
