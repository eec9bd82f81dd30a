"""Random calls (frame sizes, regions, sample slices, interleaved tile sets, scans) traced with whole-tile
chunks and with random cuts of the work queue: sums and ray counts must agree.  Usage: fuzz_queue.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import numpy as np
import torch, rtamd, scenes

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = rtamd.Context(0)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
bad = 0
for case in range(cases):
    if case % 25 == 0:
        ctx.set_scene(scenes.random_spheres(int(rs.choice([3, 40, 200, 700])), grid_half=int(rs.randint(3, 9)), seed=case))
    W, H = int(rs.randint(1, 400)), int(rs.randint(1, 250))
    spp = int(rs.choice([1, 3, 7, 8, 9, 16, 33, 64, 130, 300]))
    s0 = int(rs.randint(0, spp))
    cnt = int(rs.randint(1, spp - s0 + 1))
    stride = int(rs.choice([0, 0, 2, 5, 8]))
    first = int(rs.randint(0, stride)) if stride else 0
    tile = None
    if rs.rand() < 0.5:
        tile = (int(rs.randint(-10, W)), int(rs.randint(-10, H)), int(rs.randint(1, W + 20)), int(rs.randint(1, H + 20)))
    p = rtamd.params(W, H, spp, int(rs.randint(1, 9)), seed=case, sample_begin=s0, sample_count=cnt, tile_first=first, tile_stride=stride)
    ctx.set_option(rtamd.OPT_FILTER, int(rs.choice([rtamd.FILTER_AUTO, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH])))
    out = []
    for per_wave, subtiles in ((0, 4), (int(rs.choice([1, 4, 16, 64, 256])), int(rs.choice([1, 2, 4, 8])))):
        ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, per_wave)
        ctx.set_option(rtamd.OPT_SUBTILES, subtiles)
        d = torch.full((H * W * 3,), 7, dtype=torch.int32, device="cuda")
        ctx.reset_stats()
        ctx.accumulate_device(cam, p, d.data_ptr(), tile=tile)
        torch.cuda.synchronize()
        out.append((d.cpu().numpy(), ctx.stats().rays))
    ok = np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    bad += not ok
    if not ok or case % 100 == 0:
        print(f"case {case}: {W}x{H} spp {spp} [{s0},{s0 + cnt}) stride {stride} first {first} tile {tile} cut ({per_wave},{subtiles}) rays {out[0][1]} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
