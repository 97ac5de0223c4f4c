"""Shared by the GPU filter tests and the CPU model of the filter: points that project onto / next to pixel boundaries."""
import numpy as np


def boundary_points(fx, fy, W, H, zs, eps):
    """float32 points whose projection through the identity camera lands at integer + e pixels, for every e in eps"""
    pts = []
    kx = np.arange(-3, W + 3, dtype=np.float64)
    for z in zs:
        for e in eps:
            ky = (kx * 7) % (H + 4) - 2
            x = (kx + e - W / 2) * z / fx
            y = (ky + e * 0.5 + 0.37 - H / 2) * z / fy
            pts.append(np.stack([x, y, np.full_like(kx, z)], 1))
            pts.append(np.stack([(ky * 3 % W + 0.41 - W / 2) * z / fx, (ky + e - H / 2) * z / fy, np.full_like(kx, z)], 1))
    return np.concatenate(pts).astype(np.float32)
