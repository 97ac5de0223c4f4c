"""CPU: the viewer-path oracle (oracle/render_oracle.c) against the golden vectors produced by running
the reference's own worker code under node (tools/make_golden_js.js), plus known-answer tests of the
shader restatement (the GLSL itself cannot run here: parity unpinned, see the oracle header)."""
import os

import numpy as np
import pytest

import oracle
import render_cases
from conftest import GOLDEN, cam_dict


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "render_js.npz"))


def test_pack_matches_node(g):
    buf, order = oracle.pack_splats(g["xyz"], g["scale"], g["rot"], g["opacity"], g["f_dc"])
    assert np.array_equal(buf, g["buffer"])
    # importance order is descending and stable
    pos = buf[:, :12].copy().view(np.float32)
    assert np.array_equal(pos, g["xyz"][order])


def test_texture_matches_node(g):
    _, order = oracle.pack_splats(g["xyz"], g["scale"], g["rot"], g["opacity"], g["f_dc"])
    tex = oracle.texture(g["buffer"], g["labels"][order])
    assert np.array_equal(tex, g["texdata"])


def test_matrices_and_depth_order_match_node(g):
    for v in range(len(g["cam_fx"])):
        cam = cam_dict(g["cam_fx"][v], g["cam_fy"][v], (0, 0), g["cam_R"][v], g["cam_p"][v])
        W, H = g["cam_wh"][v]
        view = oracle.view_matrix(cam)
        proj = oracle.proj_matrix(cam["fx"], cam["fy"], W, H)
        vp = oracle.multiply4(proj, view)
        assert np.array_equal(view, g["view"][v]) and np.array_equal(proj, g["proj"][v]) and np.array_equal(vp, g["viewproj"][v])
        di, dropped = oracle.depth_order(g["buffer"], vp)
        assert np.array_equal(di, g["depth_index"][v])
        assert dropped == len(di) - len(np.unique(di)) or dropped == 0


def oracle_render(xyz, scale, rot, opacity, f_dc, cam, W, H):
    return oracle.render_scene(np.asarray(xyz, np.float32), np.asarray(scale, np.float32), np.asarray(rot, np.float32),
                               np.asarray(opacity, np.float32), np.asarray(f_dc, np.float32), cam, W, H)


@pytest.mark.parametrize("case", render_cases.ALL_CASES, ids=lambda c: c.__name__)
def test_known_answers(case):
    case(oracle_render)


def test_golden_scene_renders_finite_and_front_to_back(g):
    cam = cam_dict(g["cam_fx"][0], g["cam_fy"][0], (0, 0), g["cam_R"][0], g["cam_p"][0])
    img = oracle.render_view(g["texdata"], g["depth_index"][0], cam, 640, 360)
    assert np.isfinite(img).all() and img[..., 3].max() <= 1.0 + 1e-6 and img[..., 3].max() > 0.5
    assert (img[..., :3] <= img[..., 3:4] + 1e-6).all()          # premultiplied colour never exceeds alpha
