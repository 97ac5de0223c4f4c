"""Seeded synthetic scenes of SURVEY.md §8(d): Gaussian-blob point clouds, Fibonacci-sphere
cameras (cameras.json conventions) and piecewise-constant segmentation maps.  Pure numpy/scipy,
identical on every rank and on the CPU-baseline side.  Data only — no reference code involved."""
import numpy as np

BASE_SEED = 0xC0FFEE


def make_positions(n, seed):
    """Mixture of 64 Gaussian blobs, centres U([-4,4]^3), sigma U(0.2,1.0); float32 (n,3)."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-4.0, 4.0, size=(64, 3))
    sigmas = rng.uniform(0.2, 1.0, size=64)
    which = rng.integers(0, 64, size=n)
    pos = centres[which] + rng.standard_normal((n, 3)) * sigmas[which, None]
    return pos.astype(np.float32)


def make_splat_attributes(n, seed, sh_degree=3):
    """scale (log), rotation quaternion, opacity (logit), f_dc, f_rest — 3DGS PLY conventions."""
    rng = np.random.default_rng(seed + 1)
    scale = np.log(rng.uniform(0.005, 0.05, size=(n, 3))).astype(np.float32)
    rot = rng.standard_normal((n, 4))
    rot = (rot / np.linalg.norm(rot, axis=1, keepdims=True)).astype(np.float32)
    op = rng.uniform(0.05, 0.99, size=n)
    opacity = np.log(op / (1.0 - op)).astype(np.float32)
    f_dc = rng.standard_normal((n, 3)).astype(np.float32)
    n_rest = 3 * ((sh_degree + 1) ** 2 - 1)
    f_rest = (0.1 * rng.standard_normal((n, n_rest))).astype(np.float32)
    return dict(scale=scale, rot=rot, opacity=opacity, f_dc=f_dc, f_rest=f_rest)


def make_cameras(n_views, width, height, radius=8.0, convention="c2w"):
    """Poses on a Fibonacci sphere looking at the origin, fx = fy = 0.9*width.

    convention "c2w": `rotation` columns are the camera axes in world space (what cameras.json
    holds and the viewer's getViewMatrix expects, gs.js:81-107).  "w2c": rows are the camera axes,
    i.e. the matrix for which the labeler's R @ (x - p) (dls.py:66-69) looks at the origin."""
    cams = []
    golden = np.pi * (3.0 - np.sqrt(5.0))
    for i in range(n_views):
        yy = 1.0 - 2.0 * (i + 0.5) / n_views
        r = np.sqrt(max(0.0, 1.0 - yy * yy))
        th = golden * i
        d = np.array([np.cos(th) * r, yy, np.sin(th) * r])
        p = d * radius
        zc = -d
        up = np.array([0.0, 1.0, 0.0]) if abs(yy) < 0.99 else np.array([1.0, 0.0, 0.0])
        xc = np.cross(up, zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        w2c = np.stack([xc, yc, zc])
        R = w2c.T if convention == "c2w" else w2c
        cams.append({
            "id": i, "img_name": f"view_{i:05d}", "width": int(width), "height": int(height),
            "position": [float(v) for v in p], "rotation": [[float(v) for v in row] for row in R],
            "fx": 0.9 * width, "fy": 0.9 * width,
        })
    return cams


def make_segmap(height, width, n_classes, seed, n_sites=400, cell=4):
    """Voronoi diagram of n_sites random sites (evaluated on a cell-px grid, pixel-replicated),
    each region a class in [-1, n_classes-1]; int32 (height, width)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    sites = rng.uniform(0.0, 1.0, size=(n_sites, 2)) * np.array([width, height])
    cls = rng.integers(-1, n_classes, size=n_sites, dtype=np.int32)
    gh, gw = (height + cell - 1) // cell, (width + cell - 1) // cell
    gy, gx = np.mgrid[0:gh, 0:gw]
    pts = np.stack([(gx.ravel() + 0.5) * cell, (gy.ravel() + 0.5) * cell], 1)
    _, idx = cKDTree(sites).query(pts)
    coarse = cls[idx].reshape(gh, gw)
    return np.repeat(np.repeat(coarse, cell, axis=0), cell, axis=1)[:height, :width].copy()


def make_segmap_gpu(torch, device, height, width, n_classes, seed, n_sites=400, fast=False):
    """make_segmap's construction with pixel-accurate boundaries (cell = 1: the Voronoi diagram evaluated at every pixel
    centre), on the GPU - 2 M pixels x 400 sites are seconds of KD-tree queries per map on the host - returned as a HOST
    int32 array: what the labeler under test and the CPU oracle both consume.  fast: float32 distances in 1 M-pixel pieces
    (4K maps by the hundred); the default float64 form is the benchmark's."""
    dev = torch.device("cuda", device)
    rng = np.random.default_rng(seed)
    sites = rng.uniform(0.0, 1.0, size=(n_sites, 2)) * np.array([width, height])
    cls = rng.integers(-1, n_classes, size=n_sites, dtype=np.int32)
    ft = torch.float32 if fast else torch.float64
    ys = torch.arange(height, device=dev, dtype=ft) + 0.5
    xs = torch.arange(width, device=dev, dtype=ft) + 0.5
    sx = torch.from_numpy(sites[:, 0]).to(dev).to(ft)
    sy = torch.from_numpy(sites[:, 1]).to(dev).to(ft)
    tcls = torch.from_numpy(cls).to(dev)
    seg = torch.empty((height, width), dtype=torch.int32, device=dev)
    rows = max(1, ((1 << 20) if fast else (1 << 18)) // width)
    for y0 in range(0, height, rows):
        y1 = min(height, y0 + rows)
        d = (ys[y0:y1, None, None] - sy) ** 2 + (xs[None, :, None] - sx) ** 2
        seg[y0:y1] = tcls[d.argmin(dim=2)]
    return seg.cpu().numpy()


def make_scene(n, n_views, width, height, n_classes=150, config_id=0, convention="c2w", first_view=0,
               total_views=None):
    """positions, cameras, segmaps for views [first_view, first_view+n_views) of a total_views-camera rig."""
    seed = BASE_SEED + config_id
    total = total_views if total_views is not None else n_views
    cams = make_cameras(total, width, height, convention=convention)[first_view:first_view + n_views]
    segs = [make_segmap(height, width, n_classes, seed * 1000 + first_view + v) for v in range(n_views)]
    return make_positions(n, seed), cams, segs
