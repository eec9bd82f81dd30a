"""Synthetic scenes and cameras of the benchmark configurations.

The sphere records use the reference's conventions (Source/main.cpp:520-526):
diffuse colours on a 0..255 scale, metal colours on 0..1, Glass rendered as flat
white by the reference's GetRayColor (Source/main.cpp:594).

`random_spheres` is deterministic: it draws from numpy's legacy
RandomState(seed) stream (frozen by numpy's compatibility policy) in a fixed
order, and the 500-sphere instance is additionally committed as
tests/golden/scene_c3_500.bin.
"""
import numpy as np

from rtamd import MAT_DIFFUSE, MAT_GLASS, MAT_METAL, SPHERE_DTYPE, default_scene  # noqa: F401

# camera of the random-spheres configurations: lookfrom, lookat, up, fov
RANDOM_SPHERES_CAMERA = ((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0)
DEFAULT_CAMERA = ((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 90.0)


def random_spheres(n=500, grid_half=11, seed=12345):
    """Ground + three large spheres + a (2*grid_half)^2 grid of small ones, cut or padded to n."""
    rs = np.random.RandomState(seed)
    U = lambda: np.float32(rs.random_sample())
    out = []

    def add(center, color, radius, material, fuzz=0.0):
        out.append((tuple(np.float32(c) for c in center), tuple(np.float32(c) for c in color),
                    np.float32(radius), material, np.float32(fuzz)))

    add((0.0, -1000.0, 0.0), (1.0, 1.0, 1.0), 1000.0, MAT_DIFFUSE)
    add((0.0, 1.0, 0.0), (1.0, 1.0, 1.0), 1.0, MAT_GLASS)
    add((-4.0, 1.0, 0.0), (0.8, 0.4, 0.2), 1.0, MAT_DIFFUSE)
    add((4.0, 1.0, 0.0), (0.7, 0.6, 0.5), 1.0, MAT_METAL, 0.0)

    def small(cx, cz):
        choose = U()
        if choose < 0.6:
            add((cx, 0.2, cz), (255.0 * U(), 255.0 * U(), 255.0 * U()), 0.2, MAT_DIFFUSE)
        elif choose < 0.9:
            add((cx, 0.2, cz), (0.5 + 0.5 * U(), 0.5 + 0.5 * U(), 0.5 + 0.5 * U()), 0.2, MAT_METAL, 0.5 * U())
        else:
            add((cx, 0.2, cz), (1.0, 1.0, 1.0), 0.2, MAT_GLASS)

    for a in range(-grid_half, grid_half):
        for b in range(-grid_half, grid_half):
            if len(out) >= n:
                break
            cx = np.float32(a) + np.float32(0.9) * U()
            cz = np.float32(b) + np.float32(0.9) * U()
            if np.hypot(float(cx) - 4.0, float(cz)) <= 0.9:
                continue
            small(cx, cz)
    while len(out) < n:
        g = np.float32(grid_half)
        small(g * (np.float32(2.0) * U() - np.float32(1.0)), g * (np.float32(2.0) * U() - np.float32(1.0)))
    return np.array(out[:n], dtype=SPHERE_DTYPE)
