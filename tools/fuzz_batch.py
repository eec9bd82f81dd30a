"""Random launches of several jobs (rt_accumulate_batch_device: frames of different sizes, cameras, regions,
sample slices, interleaved tile sets and depths, each into sums of its own or into sums it shares with another
job, random cuts of the work queue, every scan) against the same jobs traced by calls of their own: sums
and ray counts must agree.  Usage: fuzz_batch.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import numpy as np
import torch, rtamd, scenes

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = rtamd.Context(0)
bad = 0
for case in range(cases):
    if case % 20 == 0:
        ctx.set_scene(scenes.random_spheres(int(rs.choice([3, 40, 200, 700])), grid_half=int(rs.randint(3, 9)), seed=case))
    ctx.set_option(rtamd.OPT_FILTER, int(rs.choice([rtamd.FILTER_AUTO, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH])))
    ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, int(rs.choice([0, 1, 4, 32, 256])))
    ctx.set_option(rtamd.OPT_SUBTILES, int(rs.choice([1, 2, 4, 8])))
    n_jobs = int(rs.randint(1, rtamd.MAX_JOBS + 1))
    jobs, owners = [], []
    for k in range(n_jobs):
        share = k > 0 and rs.rand() < 0.3
        if share:  # same frame as an earlier job, adding into the same sums
            o = int(rs.randint(0, k))
            W, H = jobs[o][1].width, jobs[o][1].height
            owner = owners[o]
        else:
            W, H = int(rs.randint(1, 300)), int(rs.randint(1, 200))
            owner = k
        spp = int(rs.choice([1, 2, 5, 8, 16, 40, 130]))
        s0 = int(rs.randint(0, spp))
        cnt = int(rs.randint(1, spp - s0 + 1))
        stride = int(rs.choice([0, 0, 3, 8]))
        first = int(rs.randint(0, stride)) if stride else 0
        tile = None
        if rs.rand() < 0.4:
            tile = (int(rs.randint(-10, W)), int(rs.randint(-10, H)), int(rs.randint(1, W + 20)), int(rs.randint(1, H + 20)))
        cam = rtamd.make_camera((13 - rs.rand() * 4, 2 + rs.rand(), 3 + rs.rand() * 2), (0, 0, 0), (0, 1, 0), 15 + rs.rand() * 30)
        p = rtamd.params(W, H, spp, int(rs.randint(0, 9)), seed=case, sample_begin=s0, sample_count=cnt,
                         tile_first=first, tile_stride=stride)
        jobs.append((cam, p, None, tile))
        owners.append(owner)
    results = []
    for batched in (False, True):
        bufs = {o: torch.full((jobs[o][1].height * jobs[o][1].width * 3,), 5, dtype=torch.int32, device="cuda") for o in set(owners)}
        full = [(c, p, bufs[owners[k]].data_ptr(), t) for k, (c, p, _, t) in enumerate(jobs)]
        ctx.reset_stats()
        if batched:
            ctx.accumulate_batch_device(full)
        else:
            for c, p, d, t in full:
                ctx.accumulate_device(c, p, d, tile=t)
        torch.cuda.synchronize()
        results.append(({o: b.cpu().numpy() for o, b in bufs.items()}, ctx.stats().rays))
    ok = results[0][1] == results[1][1] and all(np.array_equal(results[0][0][o], results[1][0][o]) for o in results[0][0])
    bad += not ok
    if not ok or case % 50 == 0:
        print(f"case {case}: {n_jobs} jobs, owners {owners}, rays {results[0][1]} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
