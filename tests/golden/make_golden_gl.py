#!/usr/bin/env python3
"""Generate tests/golden/render_gl_cases.npz and render_gl_scenes.npz: frames of the reference's OWN vertex + fragment shaders and blend state
(gaussians_selection.js:661-800, 1033-1038, 1608-1609), executed by Mesa llvmpipe (the software OpenGL ES 3.2 the image
ships), fed with what the reference's OWN worker code computes under node (32-byte rows, RGBA32UI texture, depthIndex, view
and projection matrices: tools/make_golden_js.js).  This pins the GLSL half of the rasterizer, which
rounds 1-2 could only check against the builder's C restatement.

BUILD CONTAINER ONLY (needs /root/reference, node, gcc, swrast_dri.so).  The shader text is cut out of the reference at run
time into a temporary directory and handed to tools/gl_reference/gl_frames.c; it is never written into this repository.  The
fixture holds DATA only: the scenes (raw 3DGS attributes, cameras, frame sizes) and the frames.
One deliberate difference from the browser, recorded in the fixture: the colour buffer is RGBA32F, not the canvas's RGBA8
(the contract is the fragment output within 1e-4; an 8-bit target quantises every blend step).

Usage: python tests/golden/make_golden_gl.py            writes the two fixtures
       python tests/golden/make_golden_gl.py --big K F  an UNCOMMITTED fixture F of K random scenes x 3 cameras for tools/gl_soak_gpu.py
       python tests/golden/make_golden_gl.py --soak K   no fixture: K random scenes x 3 cameras (and the four cameras of
                                                        render_js.npz's scene), oracle against GL, statistics only
"""
import base64
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
OUT = os.path.join(ROOT, "tests", "golden")
REF_JS = "/root/reference/Web_Viewer_Gaussians_Selection/gaussians_selection.js"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, OUT)

from make_golden_render import write_3dgs_ply  # noqa: E402  (tools-only PLY writer)


class GlReference:
    """render(xyz, scale, rot, opacity, f_dc, cam, W, H) -> (H, W, 4) float32 through node + llvmpipe; keeps every call."""

    def __init__(self, tmp):
        self.tmp = tmp
        src = open(REF_JS).read()
        vs = re.search(r"const vertexShaderSource = `(.*?)`\.trim\(\)", src, re.S).group(1).strip()
        fs = re.search(r"const fragmentShaderSource = `(.*?)`\.trim\(\)", src, re.S).group(1).strip()
        self.vs, self.fs = os.path.join(tmp, "vertex.glsl"), os.path.join(tmp, "fragment.glsl")
        open(self.vs, "w").write(vs)
        open(self.fs, "w").write(fs)
        self.exe = os.path.join(tmp, "gl_frames")
        subprocess.check_call(["gcc", "-O2", "-Wall", "-o", self.exe, os.path.join(ROOT, "tools", "gl_reference", "gl_frames.c"), "-ldl"])
        self.calls = []
        self.scenes = []      # raw attributes, one entry per node run
        self.gl_strings = None

    def node(self, attrs, cams):
        ply, cj, oj = (os.path.join(self.tmp, f) for f in ("in.ply", "cams.json", "out.json"))
        write_3dgs_ply(ply, attrs)
        json.dump(cams, open(cj, "w"))
        subprocess.check_call(["node", os.path.join(ROOT, "tools", "make_golden_js.js"), ply, cj, oj])
        return json.load(open(oj))

    def frames(self, xyz, scale, rot, opacity, f_dc, cams, W, H):
        """All cameras of one scene (one node run) -> list of frames."""
        n = len(xyz)
        attrs = dict(xyz=np.asarray(xyz, np.float32).reshape(n, 3), scale=np.asarray(scale, np.float32).reshape(n, 3),
                     rot=np.asarray(rot, np.float32).reshape(n, 4), opacity=np.asarray(opacity, np.float32).reshape(n),
                     f_dc=np.asarray(f_dc, np.float32).reshape(n, 3), f_rest=np.zeros((n, 0), np.float32))
        jcams = []
        for c in cams:
            jcams.append({"fx": float(c["fx"]), "fy": float(c["fy"]), "width": int(c.get("width", W)), "height": int(c.get("height", H)),
                          "rotation": np.asarray(c["rotation"], np.float64).reshape(3, 3).tolist(),
                          "position": np.asarray(c["position"], np.float64).reshape(3).tolist(), "render_width": W, "render_height": H,
                          "clicks": []})
        out = self.node(attrs, jcams)
        assert out["vertexCount"] == n
        texw, texh = int(out["texwidth"]), int(out["texheight"])
        tex = np.frombuffer(base64.b64decode(out["texdata"]), np.uint32)
        assert texw == 2048 and len(tex) == texw * texh * 4
        res = []
        self.scenes.append(attrs)
        for c, jc, oc in zip(cams, jcams, out["cameras"]):
            di = np.frombuffer(base64.b64decode(oc["depthIndex"]), np.uint32)
            assert len(di) == n
            inp, outp = os.path.join(self.tmp, "in.bin"), os.path.join(self.tmp, "out.f32")
            with open(inp, "wb") as f:
                f.write(np.array([W, H, n, texw, texh], np.int32).tobytes())
                f.write(np.asarray(oc["view"], np.float32).tobytes())     # gl.uniformMatrix4fv takes the JS numbers as f32
                f.write(np.asarray(oc["proj"], np.float32).tobytes())
                f.write(np.array([jc["fx"], jc["fy"]], np.float32).tobytes())
                f.write(np.array([W, H], np.float32).tobytes())
                f.write(tex.tobytes())
                f.write(di.astype(np.int32).tobytes())
            p = subprocess.run([self.exe, self.vs, self.fs, inp, outp], check=True, capture_output=True, text=True)
            self.gl_strings = p.stderr.strip().splitlines()[0].replace("gl_frames: ", "")
            frame = np.fromfile(outp, np.float32).reshape(H, W, 4)
            self.calls.append(dict(scene=len(self.scenes) - 1, xyz=attrs["xyz"], scale=attrs["scale"], rot=attrs["rot"],
                                   opacity=attrs["opacity"], f_dc=attrs["f_dc"], fx=jc["fx"], fy=jc["fy"], R=np.asarray(jc["rotation"]),
                                   p=np.asarray(jc["position"]), W=W, H=H, frame=frame))
            res.append(frame)
        return res

    def render(self, xyz, scale, rot, opacity, f_dc, cam, W, H):
        return self.frames(xyz, scale, rot, opacity, f_dc, [cam], W, H)[0]


BIG = (40_000, 150_000)


def soak(K):
    """Random scenes: how far is oracle/render_oracle.c from the reference's shaders, and how often does a threshold fragment flip?"""
    import importlib
    import oracle
    scene = importlib.import_module("3d_gaussian_splatting_project_amd.scene")
    rng = np.random.default_rng(20261005)
    worst, flips, covered, frames, nearest = 0.0, 0, 0, 0, []
    with tempfile.TemporaryDirectory() as tmp:
        gl = GlReference(tmp)
        jobs = []
        g = np.load(os.path.join(OUT, "render_js.npz"))
        for v in range(4):   # the node-golden scene at its own frame sizes (640x360, 1557x1037)
            W, H = (int(x) for x in g["cam_wh"][v])
            jobs.append((g["xyz"], g["scale"], g["rot"], g["opacity"], g["f_dc"],
                         [{"fx": g["cam_fx"][v], "fy": g["cam_fy"][v], "rotation": g["cam_R"][v], "position": g["cam_p"][v]}], W, H))
        for k in range(K):
            n = int(rng.integers(200, 8000))
            W, H = int(rng.integers(64, 400)), int(rng.integers(48, 300))
            seed = 0xD00D00 + k
            xyz = scene.make_positions(n, seed)
            a = scene.make_splat_attributes(n, seed, sh_degree=0)
            a["scale"] += np.float32(rng.uniform(-1.0, 1.5))            # from needle-thin to fat splats
            cams = scene.make_cameras(16, W, H, convention="c2w")
            cams = [cams[int(i)] for i in rng.choice(16, 3, replace=False)]
            for c in cams[1:]:                                         # pull two cameras towards / into the cloud
                c["position"] = (np.asarray(c["position"]) * rng.uniform(0.05, 0.8)).tolist()
            jobs.append((xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"], cams, W, H))
        for k, n in enumerate(BIG):   # heavy overdraw: tens of thousands of splats on a small frame
            W, H = 480, 270
            seed = 0xB16B00 + k
            xyz = scene.make_positions(n, seed)
            a = scene.make_splat_attributes(n, seed, sh_degree=0)
            jobs.append((xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"], scene.make_cameras(5, W, H, convention="c2w")[k:k + 2], W, H))
        for xyz, sc, rot, op, fdc, cams, W, H in jobs:
            fr = gl.frames(xyz, sc, rot, op, fdc, cams, W, H)
            for cam, f in zip(cams, fr):
                o = oracle.render_scene(xyz, sc, rot, op, fdc, cam, W, H)
                d = np.abs(o.astype(np.float64) - f).max(axis=2)
                over = d > 1e-4
                frames += 1
                flips += int(over.sum())
                covered += int((f[..., 3] > 0).sum())
                worst = max(worst, float(d[~over].max()))
                assert d.max() <= np.exp(-4.0) + 1e-4, (len(xyz), W, H, float(d.max()))
                if over.any():   # is it really a fragment on the discard threshold?  |A + 4| of the nearest fragment, oracle arithmetic
                    buf, _ = oracle.pack_splats(xyz, sc, rot, op, fdc)
                    tex = oracle.texture(buf)
                    view, proj = oracle.view_matrix(cam), oracle.proj_matrix(cam["fx"], cam["fy"], W, H)
                    vs_ = [oracle.vertex(tex[8 * i:8 * i + 8], view, proj, np.float32(cam["fx"]), np.float32(cam["fy"]), W, H) for i in range(len(xyz))]
                    for r, x in zip(*np.nonzero(over)):
                        best = (np.inf, np.inf)
                        for v in vs_:
                            if not v.drawn:
                                continue
                            dx = np.float32(np.float32(x) + np.float32(0.5)) - np.float32(v.cx)
                            dy = np.float32(np.float32(H) - np.float32(np.float32(r) + np.float32(0.5))) - np.float32(v.cy)
                            vx = np.float32(dx * np.float32(v.g0[0])) + np.float32(dy * np.float32(v.g0[1]))
                            vy = np.float32(dx * np.float32(v.g1[0])) + np.float32(dy * np.float32(v.g1[1]))
                            A = -(np.float32(vx * vx) + np.float32(vy * vy))
                            # GL snaps the quad's corners to 1/256 pixel (GL_SUBPIXEL_BITS = 8, as GPUs do): the interpolated
                            # vPosition of a fragment moves by up to |g| / 256, A = -|v|^2 by up to 2 |v| |g| / 256 with |v| = 2
                            room = 4.0 * max(float(np.hypot(v.g0[0], v.g0[1])), float(np.hypot(v.g1[0], v.g1[1]))) / 256.0 + 4e-6
                            if abs(float(A) + 4.0) / room < best[0]:
                                best = (abs(float(A) + 4.0) / room, abs(float(A) + 4.0))
                        nearest.append(best)
                        print(f'      pixel ({x},{r}): differs by {d[r, x]:.3e}; nearest fragment to the threshold: |A + 4| = {best[1]:.3e} = {best[0]:.2f} x what the 1/256-pixel vertex snapping can move it')
                print(f"{len(xyz):5d} splats {W:4d}x{H:<4d} covered {int((f[..., 3] > 0).sum()):7d}  max|oracle - GL| {d[~over].max():.3e}"
                      + (f"  outside the {int(over.sum())} pixel(s) listed above (max {d.max():.3e})" if over.any() else ""), flush=True)
        print(f"{frames} frames, {covered} covered pixels: max |oracle - GL| = {worst:.3e} outside {flips} pixels (1 in {covered // max(flips, 1)}) "
              f"that differ by more than 1e-4 (none by more than e^-4 = one fragment at the discard threshold); GL = {gl.gl_strings}")
        if nearest:
            rel = np.array([b[0] for b in nearest])
            print(f"those {len(nearest)} pixels: {int((rel <= 1.0).sum())} are covered by a fragment whose A = -|vPosition|^2 lies within the reach of GL's "
                  f"1/256-pixel vertex snapping (+ 4e-6 of fp32 rounding) of the discard threshold -4 - the rasteriser's fixed-point quad decides "
                  f"whether that fragment exists; {int((rel > 1.0).sum())} are not (accumulated rounding in heavily overdrawn pixels)")


def big_fixture(K, path):
    """K random scenes x 3 cameras -> one uncommitted .npz (git-ignored name) for tools/gl_soak_gpu.py: the HIP rasterizer against
    the reference's shaders on far more frames than the committed fixtures hold (the GPU box has no reference to run)."""
    import importlib
    scene = importlib.import_module("3d_gaussian_splatting_project_amd.scene")
    rng = np.random.default_rng(20261006)
    store = {}
    with tempfile.TemporaryDirectory() as tmp:
        gl = GlReference(tmp)
        for k in range(K):
            n = int(rng.integers(200, 20000))
            W, H = int(rng.integers(64, 480)), int(rng.integers(48, 300))
            seed = 0xFEED00 + k
            xyz = scene.make_positions(n, seed)
            a = scene.make_splat_attributes(n, seed, sh_degree=0)
            a["scale"] += np.float32(rng.uniform(-1.0, 1.5))
            cams = scene.make_cameras(16, W, H, convention="c2w")
            cams = [cams[int(i)] for i in rng.choice(16, 3, replace=False)]
            for c in cams[1:]:
                c["position"] = (np.asarray(c["position"]) * rng.uniform(0.05, 0.8)).tolist()
            gl.frames(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"], cams, W, H)
            print(f"scene {k}: {n} splats {W}x{H}", flush=True)
        for j, sc in enumerate(gl.scenes):
            for key in ("xyz", "scale", "rot", "opacity", "f_dc"):
                store[f"s{j}_{key}"] = sc[key]
        for i, c in enumerate(gl.calls):
            store[f"c{i}_scene"] = np.int64(c["scene"])
            store[f"c{i}_cam"] = np.array([c["fx"], c["fy"], c["W"], c["H"]], np.float64)
            store[f"c{i}_R"], store[f"c{i}_p"], store[f"c{i}_frame"] = c["R"], c["p"], c["frame"]
        store["calls"] = np.arange(len(gl.calls))
        store["gl"] = np.array(gl.gl_strings)
    np.savez(path, **store)
    print(f"{path}: {os.path.getsize(path) >> 20} MB, {len(gl.calls)} frames")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--soak":
        return soak(int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[1] == "--big":
        return big_fixture(int(sys.argv[2]), sys.argv[3])
    import importlib
    import oracle
    import render_cases
    scene = importlib.import_module("3d_gaussian_splatting_project_amd.scene")
    notes = []
    with tempfile.TemporaryDirectory() as tmp:
        gl = GlReference(tmp)
        # ---- the known-answer cases (tests/render_cases.py): their closed forms are asserted on the REAL shaders' frames here
        for case in render_cases.ALL_CASES:
            first = len(gl.calls)
            try:
                case(gl.render)
                verdict = "closed form holds on the GL frames"
            except AssertionError as e:
                verdict = "closed-form assertion does NOT hold on the GL frames: " + str(e)[:200]
            notes.append(f"{case.__name__}: calls {first}..{len(gl.calls) - 1}; {verdict}")
        n_cases = len(gl.calls)
        # ---- dense synthetic scenes: heavy overdraw, anisotropic splats, the quantisation edge cases of render_js.npz
        for sid, (n, V, W, H, seed) in enumerate([(4000, 2, 224, 128, 0xC0FFEE51), (1500, 1, 160, 96, 0xC0FFEE52)]):
            rng = np.random.default_rng(seed)
            xyz = scene.make_positions(n, seed)
            a = scene.make_splat_attributes(n, seed, sh_degree=0)
            a["rot"][:50] *= rng.uniform(0.1, 10.0, size=(50, 1)).astype(np.float32)
            a["rot"][50] = [1, 0, 0, 0]
            a["opacity"][51:55] = [-20.0, 20.0, 0.0, -0.0]
            a["f_dc"][55:59] = [[-5, 5, 0], [1.7724, -1.7724, 0.5], [10, -10, 2], [0.001, -0.001, 1e-9]]
            a["scale"][59] = [-12, -12, -12]
            a["scale"][60] = [0.3, 0.3, 0.3] if sid == 0 else [2.5, 2.5, 2.5]   # a large splat / one whose 4*Sigma overflows fp16
            a["scale"][61] = [-20, -20, -19]
            if sid == 1:   # non-finite attributes as a broken PLY would carry them: what the JS packer and the shaders make of them
                xyz[100:105, 0] = np.nan
                xyz[105:110, 1] = np.inf
                a["scale"][110:115, 2] = np.nan
                a["scale"][115:120, 0] = np.inf
                a["scale"][120:125, 0] = -np.inf
                a["rot"][125:130, 1] = np.nan
                a["rot"][130:135] = 0.0
                a["opacity"][135:140] = np.nan
                a["f_dc"][140:145, 0] = np.nan
                a["f_dc"][145:150, 2] = np.inf
            cams = scene.make_cameras(V, W, H, convention="c2w")
            if sid == 1:   # a camera INSIDE the cloud: near-plane fade, the 1.2 w cull, the 1024 px axis clamp on real data
                c = dict(cams[0])
                c["position"] = [0.3, -0.2, 0.1]
                cams.append(c)
            first = len(gl.calls)
            gl.frames(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"], cams, W, H)
            notes.append(f"dense scene {sid}: {n} splats, calls {first}..{len(gl.calls) - 1}, {W}x{H}")
        # ---- how far is the builder's C restatement (oracle/render_oracle.c) from the reference's shaders?
        worst = 0.0
        for i, c in enumerate(gl.calls):
            cam = {"fx": c["fx"], "fy": c["fy"], "rotation": c["R"], "position": c["p"]}
            o = oracle.render_scene(c["xyz"], c["scale"], c["rot"], c["opacity"], c["f_dc"], cam, c["W"], c["H"])
            d = np.abs(o - c["frame"])
            bad = ~np.isfinite(c["frame"]).all(axis=2)
            d[bad] = 0
            worst = max(worst, float(d.max()))
            print(f"call {i:2d}: {len(c['xyz']):5d} splats {c['W']}x{c['H']}  covered px GL {int((c['frame'][..., 3] > 0).sum()):6d} "
                  f"oracle {int((o[..., 3] > 0).sum()):6d}  max|oracle - GL| {d.max():.3e}  px > 1e-4: {int((d.max(axis=2) > 1e-4).sum())}"
                  f"  non-finite GL px {int(bad.sum())}")
        print("\n".join(notes))
        # two files of about a megabyte: the known-answer cases, the dense scenes
        for name, lo, hi in (("render_gl_cases.npz", 0, n_cases), ("render_gl_scenes.npz", n_cases, len(gl.calls))):
            store = {}
            used = sorted({gl.calls[i]["scene"] for i in range(lo, hi)})
            for j in used:
                for k in ("xyz", "scale", "rot", "opacity", "f_dc"):
                    store[f"s{j}_{k}"] = gl.scenes[j][k]
            for i in range(lo, hi):
                c = gl.calls[i]
                store[f"c{i}_scene"] = np.int64(c["scene"])
                store[f"c{i}_cam"] = np.array([c["fx"], c["fy"], c["W"], c["H"]], np.float64)
                store[f"c{i}_R"], store[f"c{i}_p"], store[f"c{i}_frame"] = c["R"], c["p"], c["frame"]
            store["calls"] = np.arange(lo, hi)
            store["gl"] = np.array(gl.gl_strings)
            store["notes"] = np.array(notes)
            store["colour_buffer"] = np.array("RGBA32F (the browser canvas is RGBA8)")
            path = os.path.join(OUT, name)
            np.savez_compressed(path, **store)
            print(f"{path}: {os.path.getsize(path)} bytes, frames {lo}..{hi - 1}")
        print(f"GL = {gl.gl_strings}; worst |oracle - GL| = {worst:.3e}")


if __name__ == "__main__":
    main()
