"""Two small frames through the hierarchy scan against the oracle: the first thing to run on a
device after touching the walk (python tests/check_bvh_smoke.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import numpy as np, torch, rtamd, scenes, oracle_lib as O
ctx = rtamd.Context(0)
for sc, cam in ((rtamd.default_scene(), scenes.DEFAULT_CAMERA), (scenes.random_spheres(500), scenes.RANDOM_SPHERES_CAMERA)):
    ctx.set_scene(sc)
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
    ctx.reset_stats()
    got = ctx.render(rtamd.make_camera(*cam), rtamd.params(64, 36, 2, 8, seed=3))
    want, _, rays = O.trace_tile(sc, O.camera(*cam), 64, 36, 2, 8, O.RNG_COUNTER, seed=3)
    print(len(sc), "equal:", np.array_equal(got, want), "rays", ctx.stats().rays, rays, flush=True)
ctx.close()
