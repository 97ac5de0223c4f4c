import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    return importlib.import_module("3d_gaussian_splatting_project_amd")


@pytest.fixture(scope="session")
def gsx():
    return load_pkg()


@pytest.fixture(scope="session")
def ctx(gsx):
    """One libgsx context on GPU 0 for the whole session (gpu tests only)."""
    c = gsx.Context(0)
    yield c
    c.close()


def cam_dict(fx, fy, wh, R, p, name="cam"):
    return {"img_name": name, "fx": float(fx), "fy": float(fy), "width": int(wh[0]), "height": int(wh[1]),
            "rotation": np.asarray(R, dtype=np.float64).tolist(), "position": np.asarray(p, dtype=np.float64).tolist()}


def golden_project():
    g = np.load(os.path.join(GOLDEN, "vote_project.npz"))
    cams = [cam_dict(g["cam_fx"][v], g["cam_fy"][v], g["cam_wh"][v], g["cam_R"][v], g["cam_p"][v], f"c{v}")
            for v in range(len(g["cam_fx"]))]
    return g["positions"], cams, g["x"], g["y"]


def golden_assign_cases():
    """Yields (name, positions, cams, segs, img_sizes, labels) with missing-image views already
    dropped, exactly as the reference skips them (dls.py:256-259)."""
    a = np.load(os.path.join(GOLDEN, "vote_assign.npz"))
    out = []
    for case in a["cases"]:
        P = lambda k: a[str(case) + k]
        cams, segs, sizes = [], [], []
        off = 0
        for v in range(len(P("cam_fx"))):
            h, w = P("seg_shapes")[v]
            seg = P("seg_flat")[off:off + h * w].reshape(h, w).astype(np.int32)
            off += h * w
            if not P("present")[v]:
                continue
            cams.append(cam_dict(P("cam_fx")[v], P("cam_fy")[v], P("cam_wh")[v], P("cam_R")[v], P("cam_p")[v], f"{case}{v}"))
            segs.append(seg)
            sizes.append((int(P("img_wh")[v][0]), int(P("img_wh")[v][1])))
        out.append((str(case), P("positions"), cams, segs, sizes, P("labels")))
    return out
