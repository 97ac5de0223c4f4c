// Runs the CPU half of the reference viewer (gaussians_selection.js: createWorker, getViewMatrix,
// calculateProjectionMatrix, multiply4) under node, in a vm sandbox, on a PLY file and a list of
// cameras, and dumps what the worker posts: the 32-byte splat buffer, the RGBA32UI texture words,
// and depthIndex per camera.  Build-container only (needs /root/reference); emits DATA only.
//   node tools/make_golden_js.js <in.ply> <cameras.json> <out.json>
const fs = require("fs");
const vm = require("vm");
const util = require("util");

const REF = "/root/reference/Web_Viewer_Gaussians_Selection/gaussians_selection.js";
const [plyPath, camsPath, outPath] = process.argv.slice(2);
const src = fs.readFileSync(REF, "utf8");

function sandbox() {
    const quiet = {log() {}, error() {}, time() {}, timeEnd() {}, warn() {}};
    const ctx = {
        console: quiet, TextDecoder: util.TextDecoder, setTimeout, clearTimeout,
        fetch: () => new Promise(() => {}),          // inert local stub: main() never starts, nothing is fetched
        document: {getElementById: () => ({style: {}})}, window: {}, location: {hash: ""},
    };
    vm.createContext(ctx);
    vm.runInContext(src + "\n;globalThis.__ref = {createWorker, multiply4, getViewMatrix, calculateProjectionMatrix};", ctx);
    return ctx;
}

const b64 = (typed) => Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength).toString("base64");
const ply = fs.readFileSync(plyPath);
const cams = JSON.parse(fs.readFileSync(camsPath, "utf8"));
const out = {cameras: []};

for (let ci = 0; ci < cams.length; ci++) {
    const cam = cams[ci];
    const ctx = sandbox();                 // fresh worker per camera: no "view barely moved" sort skip (gs.js:421-425)
    const ref = ctx.__ref;
    const posted = [];
    const self = {postMessage: (m) => posted.push(m)};
    ctx.postMessage = (m) => posted.push(m);   // the bare postMessage at gs.js:609
    ref.createWorker(self);
    const ab = ply.buffer.slice(ply.byteOffset, ply.byteOffset + ply.byteLength);
    self.onmessage({data: {ply: ab}});
    const view = ref.getViewMatrix(cam);
    const proj = ref.calculateProjectionMatrix(cam.fx, cam.fy, cam.render_width, cam.render_height);
    const viewProj = ref.multiply4(proj, view);
    self.onmessage({data: {view: viewProj}});
    const buf = posted.find((m) => m.buffer);
    const tex = posted.find((m) => m.texdata);
    const srt = posted.find((m) => m.depthIndex);
    if (ci === 0) {
        out.vertexCount = buf.vertexCount;
        out.buffer = b64(new Uint8Array(buf.buffer));
        out.texdata = b64(tex.texdata);
        out.texwidth = tex.texwidth;
        out.texheight = tex.texheight;
    }
    // performHitTesting (gs.js:361-395) through the worker's own 'select' message
    const hits = [];
    for (const [hx, hy] of (cam.clicks || [])) {
        posted.length = 0;
        self.onmessage({data: {type: "select", x: hx, y: hy, viewMatrix: view, projectionMatrix: proj,
                               viewport: [cam.render_width, cam.render_height]}});
        hits.push(posted.find((m) => m.type === "selection").label);
    }
    out.cameras.push({view: Array.from(view), proj: Array.from(proj), viewProj: Array.from(viewProj),
                      depthIndex: b64(srt.depthIndex), hits});
}
fs.writeFileSync(outPath, JSON.stringify(out));
