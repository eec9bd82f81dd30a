"""First-light check on a GPU box: parity of small cases against the oracle, then a timed C3 frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import rtamd, scenes
import oracle_lib as O

ctx = rtamd.Context(0)
cam = rtamd.default_camera()
ocam = O.default_camera()
sc = rtamd.default_scene()
ctx.set_scene(sc)

h = ctx.primary_hits(cam, 1920, 1080)
ho = O.primary_hits(sc, ocam, 1920, 1080)
print("C2 default hits equal:", (h == ho).all(), hex(O.fnv1a64(h)), flush=True)

p = rtamd.params(128, 72, 8, 10, seed=7)
img = ctx.render(cam, p)
ref, _, rays = O.trace_tile(sc, ocam, 128, 72, 8, 10, O.RNG_COUNTER, seed=7)
print("default 128x72x8 equal:", (img == ref).all(), "mismatch px:", int((img != ref).any(axis=2).sum()), "rays gpu", ctx.stats().rays, "oracle", rays, flush=True)

c3 = scenes.random_spheres(500)
ctx.set_scene(c3)
ccam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
occam = O.camera(*scenes.RANDOM_SPHERES_CAMERA)
h = ctx.primary_hits(ccam, 1920, 1080)
t = time.time(); ho = O.primary_hits(c3, occam, 1920, 1080); print("oracle c3 hits s", time.time() - t)
print("C3 hits equal:", (h == ho).all(), flush=True)
ctx.reset_stats()
p = rtamd.params(96, 54, 4, 8, seed=1)
img = ctx.render(ccam, p)
ref, _, rays = O.trace_tile(c3, occam, 96, 54, 4, 8, O.RNG_COUNTER, seed=1)
print("c3 96x54x4 equal:", (img == ref).all(), "mismatch px:", int((img != ref).any(axis=2).sum()), "rays gpu", ctx.stats().rays, "oracle", rays, flush=True)

# exact vs conservative filter on the same frame
p = rtamd.params(480, 270, 8, 8, seed=5)
ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_EXACT); ctx.reset_stats(); a = ctx.render(ccam, p); ra = ctx.stats().rays
ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_CONSERVATIVE); ctx.reset_stats(); b = ctx.render(ccam, p); rb = ctx.stats().rays
ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_MATRIX); ctx.reset_stats(); c = ctx.render(ccam, p); rc = ctx.stats().rays
print("filters agree:", (a == b).all(), (a == c).all(), "mismatch px", int((a != c).any(axis=2).sum()), ra, rb, rc, flush=True)

# timed C3 frame
W, H, SPP, D = 1920, 1080, 64, 8
p = rtamd.params(W, H, SPP, D, seed=1)
d_rgb = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for it in range(9):
    ctx.set_option(rtamd.OPT_FILTER, it // 3)
    ctx.reset_stats()
    torch.cuda.synchronize()
    t0 = time.time()
    ctx.render_device(ccam, p, d_rgb.data_ptr(), stream=s)
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = ctx.stats(); rays = st.rays
    print("lane utilisation %.3f" % (rays / (64.0 * st.wave_passes)), end=" ")
    print(f"C3 frame {dt*1e3:.2f} ms rays {rays} Mrays/s {rays/dt/1e6:.1f} frac_fp32 {rays*500*25/dt/157.3e12:.3f}", flush=True)
