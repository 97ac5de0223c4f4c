"""Known-answer cases for the viewer's shader maths (gs.js:696-799), shared by the oracle tests (CPU)
and the HIP tests (GPU).  `render(xyz, scale, rot, opacity, f_dc, cam, W, H) -> (H, W, 4) float32`."""
import numpy as np

SH_C0 = 0.28209479177387814


def cam_identity(W, H, f=None):
    """Camera at the origin looking down +z, viewer convention (rotation = camera-to-world = I)."""
    f = float(f if f is not None else 0.9 * W)
    return {"fx": f, "fy": f, "width": W, "height": H, "rotation": np.eye(3).tolist(), "position": [0.0, 0.0, 0.0]}


def logit(p):
    return float(np.log(p / (1 - p)))


def fdc_for(rgb):
    return [(c - 0.5) / SH_C0 for c in rgb]


def u8(x):
    return float(np.clip(np.rint(x * 255.0), 0, 255)) / 255.0


def expected_single(W, H, cam, pos, sigma, alpha8, rgb8):
    """Closed form for ONE isotropic splat (scale sigma) seen by an identity-rotation camera:
    premultiplied alpha*exp(-1/2 d^T S^-1 d) inside d^T S^-1 d <= 8, S = J (sigma^2 I) J^T, fp64."""
    x, y, z = pos
    fx, fy = cam["fx"], cam["fy"]
    J = np.array([[fx / z, 0, -fx * x / z ** 2], [0, fy / z, -fy * y / z ** 2]])
    S = J @ (sigma ** 2 * np.eye(3)) @ J.T
    Si = np.linalg.inv(S)
    cx, cy = fx * x / z + W / 2, fy * y / z + H / 2          # image coordinates, y down
    yy, xx = np.mgrid[0:H, 0:W]
    d = np.stack([xx + 0.5 - cx, yy + 0.5 - cy], -1)
    q = np.einsum("...i,ij,...j->...", d, Si, d)
    w = np.where(q <= 8.0, np.exp(-0.5 * q), 0.0) * alpha8
    fade = 1.0                                                 # z >= 0.2: clamp(z_ndc/w + 1, 0, 1) = 1
    img = np.zeros((H, W, 4))
    for k in range(3):
        img[..., k] = fade * rgb8[k] * w * fade
    img[..., 3] = fade * w
    return img, q


def case_single_splat(render):
    W, H = 96, 64
    cam = cam_identity(W, H)
    pos = (0.31, -0.17, 4.0)
    sigma = 0.25
    op, rgb = 0.8, (0.9, 0.4, 0.1)
    img = render([pos], [[np.log(sigma)] * 3], [[1, 0, 0, 0]], [logit(op)], [fdc_for(rgb)], cam, W, H)
    a8, rgb8 = u8(op), [u8(c) for c in rgb]
    # the reference stores 4*Sigma as TRUNCATED fp16 (<= 2^-10 relative) and the unit quaternion as
    # 255/128-128/128 = 0.9921875 (gs.js:324-327): Sigma is scaled by r^2 with r = 1 - 2(...)=1 exactly
    # for q=(w,0,0,0) -> M = diag(1,1,1)*scale; only the fp16 truncation remains
    want, q = expected_single(W, H, cam, pos, sigma, a8, rgb8)
    # fp16 truncation shrinks Sigma by < 0.1 %: compare away from the support edge, loosely
    inner = q < 7.5
    assert img.shape == (H, W, 4) and np.isfinite(img).all()
    assert np.abs(img[inner] - want[inner]).max() < 4e-3
    assert (img[q > 8.5] == 0).all()                      # nothing outside |vPosition| = 2
    peak = img[..., 3].max()
    assert abs(peak - a8) < 0.02 and peak <= a8 + 1e-6
    return img


def case_two_splats_under_operator(render):
    """Front splat A, back splat B on the same ray: out = A + (1 - A.a) * B  (gs.js:1037)."""
    W, H = 64, 64
    cam = cam_identity(W, H)
    pa, pb = (0.05, 0.02, 3.0), (0.05 * 5 / 3, 0.02 * 5 / 3, 5.0)       # same pixel centre
    args = lambda pos, op, rgb, s: ([pos], [[np.log(s)] * 3], [[1, 0, 0, 0]], [logit(op)], [fdc_for(rgb)])
    A = render(*args(pa, 0.6, (1.0, 0.0, 0.0), 0.2), cam, W, H)
    B = render(*args(pb, 0.9, (0.0, 0.0, 1.0), 0.5), cam, W, H)
    both = render([pa, pb], [[np.log(0.2)] * 3, [np.log(0.5)] * 3], [[1, 0, 0, 0]] * 2, [logit(0.6), logit(0.9)],
                  [fdc_for((1.0, 0.0, 0.0)), fdc_for((0.0, 0.0, 1.0))], cam, W, H)
    # A lone splat has maxDepth == minDepth: depthInv = Infinity, 0*Infinity = NaN, NaN|0 = 0 -> bucket 0,
    # drawn once.  With two splats the far one (B) lands in bucket 65536, which runSort's typed arrays
    # silently drop; its slot of depthIndex stays 0, so index 0 (importance order) is drawn again LAST:
    imp = lambda s, op: (s ** 3) * op
    first_is_a = imp(0.2, u8(0.6)) >= imp(0.5, u8(0.9))
    tail = A if first_is_a else B
    want = A + (1 - A[..., 3:4]) * tail
    assert np.abs(both - want).max() < 2e-6
    return both


def case_depth_fade(render):
    """vColor fades with clamp(z_ndc/w + 1, 0, 1): 0 at cam.z = 0.1, 1 from cam.z = 0.2 (gs.js:741)."""
    W, H = 64, 64
    cam = cam_identity(W, H, f=40.0)
    peaks = []
    for z in (0.09, 0.1, 0.125, 0.15, 0.2, 0.4):
        # centre exactly on the pixel centre (32.5, 32.5); sigma = 3 px
        img = render([(0.5 * z / 40.0, 0.5 * z / 40.0, z)], [[np.log(3.0 * z / 40.0)] * 3], [[1, 0, 0, 0]], [logit(0.9)],
                     [fdc_for((1, 1, 1))], cam, W, H)
        peaks.append(float(img[..., 3].max()))
    zf = 200.0 / 199.8
    want = [np.clip(zf * (1 - 0.2 / z) + 1, 0, 1) * u8(0.9) for z in (0.09, 0.1, 0.125, 0.15, 0.2, 0.4)]
    assert np.allclose(peaks, want, atol=0.01), (peaks, want)
    assert peaks[0] == 0.0 and peaks[-1] > 0.85
    return peaks


def case_frustum_cull(render):
    """Culled iff |x_clip| > 1.2 w (gs.js:709-713): centre at +-1.2 * half-width of the frame."""
    W, H = 80, 40
    cam = cam_identity(W, H, f=40.0)
    z = 2.0
    half = (W / 2) / cam["fx"] * z          # world x at the right image edge
    out = []
    for k in (1.15, 1.25):
        # y != 0: an exactly axis-aligned cov2d (b = 0, a >= d) makes the shader normalise a zero vector
        img = render([(k * half, 0.03, z)], [[np.log(0.6)] * 3], [[1, 0, 0, 0]], [logit(0.9)], [fdc_for((1, 1, 1))], cam, W, H)
        out.append(float(img[..., 3].max()))
    assert out[0] > 0.01 and out[1] == 0.0
    return out


def case_axis_clamp(render):
    """A huge, close splat: axes are clamped to 1024 px (gs.js:738-739); the frame stays finite."""
    W, H = 128, 96
    cam = cam_identity(W, H)
    img = render([(0.01, 0.02, 1.0)], [[np.log(30.0)] * 3], [[1, 0, 0, 0]], [logit(0.5)], [fdc_for((0.2, 0.5, 0.7))], cam, W, H)
    assert np.isfinite(img).all()
    # with a 1024 px axis, |vPosition| over a 128x96 frame is < 0.2: every pixel ~ alpha * exp(-small)
    a = u8(0.5)
    assert img[..., 3].min() > a * np.exp(-2 * (2 * 80 / 1024) ** 2) - 1e-3 and img[..., 3].max() <= a + 1e-6
    return img


def case_axis_aligned_degenerate(render):
    """cov2d[0][1] == 0 with cov2d[0][0] >= cov2d[1][1]: diagonalVector = normalize(vec2(0, 0)) is NaN
    (gs.js:737), the quad has no valid position and nothing is drawn.  Restated, not "fixed"."""
    W, H = 64, 64
    cam = cam_identity(W, H)
    img = render([(0.3, 0.0, 3.0)], [[np.log(0.3)] * 3], [[1, 0, 0, 0]], [logit(0.9)], [fdc_for((1, 1, 1))], cam, W, H)
    assert (img == 0).all()
    return img


ALL_CASES = [case_axis_aligned_degenerate, case_single_splat, case_two_splats_under_operator, case_depth_fade, case_frustum_cull, case_axis_clamp]
