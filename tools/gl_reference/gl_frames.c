// gl_frames.c - BUILD-CONTAINER TOOL (fixture generation only; nothing here is shipped, imported by the product or run on the
// GPU box).  Runs the REFERENCE's own GLSL (the vertex and fragment shader text of gaussians_selection.js:661-800, handed over
// as two files by tools/make_golden_gl.py, which cuts them out of the reference at run time - they are never copied into this
// repository) on Mesa's llvmpipe, the software OpenGL ES 3.2 implementation this image ships (libgl1-mesa-dri 23.2.1,
// swrast_dri.so), and writes the frame the reference's fragment shader + blend state produce.
//
// There is no X server, EGL, GBM or OSMesa in the image, so the context comes straight from the driver's DRI "swrast"
// interface (GL/internal/dri_interface.h, the one GLX's software path uses): createNewScreen2 / createContextAttribs(GLES3) /
// createNewDrawable with loader callbacks that never show anything; every frame is rendered into an FBO.
//
// The GL calls restate the viewer's host side (file = Web_Viewer_Gaussians_Selection/gaussians_selection.js):
//   program, uniforms .......... :1006-1053   (uSelectionMode 0, no displacement, no custom colours: all other uniforms stay 0)
//   blend state ................. :1033-1038   disable DEPTH_TEST; BLEND; blendFuncSeparate(ONE_MINUS_DST_ALPHA, ONE, same);
//                                              FUNC_ADD
//   quad (-2,-2 2,-2 2,2 -2,2) .. :1056-1063
//   index attribute ............. :1072-1077   vertexAttribIPointer(INT), divisor 1 <- depthIndex (:1128)
//   texture RGBA32UI 2048 x h ... :1117-1124   NEAREST, CLAMP_TO_EDGE
//   focal / viewport / projection :1081-1091, view :1590
//   clear + drawArraysInstanced(TRIANGLE_FAN, 0, 4, vertexCount) :1608-1609
// One deliberate difference, stated in the fixture: the colour buffer is RGBA32F instead of the canvas's RGBA8, because the
// contract (BASELINE.json north_star) is the FRAGMENT output within 1e-4, and an 8-bit target would quantise every blend.
//
// usage: gl_frames <vertex.glsl> <fragment.glsl> <in.bin> <out.f32>
//   in.bin : int32 W, H, n, texw, texh; f32 view[16], proj[16], focal[2], viewport[2]; u32 tex[texw*texh*4]; i32 index[n]
//   out.f32: f32 [H][W][4], row 0 = TOP row of the image (GL's bottom-up rows flipped)
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

static void get_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* p) { (void)d; (void)p; *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* p) {
    (void)d; (void)x; (void)y; (void)p;
    memset(data, 0, (size_t)w * (size_t)h * 4);
}
static void put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static void get_image2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* p) {
    (void)d; (void)x; (void)y; (void)w; (void)p;
    memset(data, 0, (size_t)stride * (size_t)h);
}
static const __DRIswrastLoaderExtension loader = {.base = {__DRI_SWRAST_LOADER, 3},
                                                  .getDrawableInfo = get_info,
                                                  .putImage = put_image,
                                                  .getImage = get_image,
                                                  .putImage2 = put_image2,
                                                  .getImage2 = get_image2};
static const __DRIextension* loader_exts[] = {&loader.base, NULL};

static void* (*gpa)(const char*);
#define GLF(type, name) type name = (type)gpa(#name); if (!name) die("no " #name)

static void die(const char* what) {
    fprintf(stderr, "gl_frames: %s\n", what);
    exit(1);
}

static char* slurp(const char* path, size_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) die(path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* b = malloc((size_t)n + 1);
    if (!b || fread(b, 1, (size_t)n, f) != (size_t)n) die("read failed");
    b[n] = 0;
    fclose(f);
    if (len) *len = (size_t)n;
    return b;
}

int main(int argc, char** argv) {
    if (argc != 5) die("usage: gl_frames vertex.glsl fragment.glsl in.bin out.f32");
    const char* vs_src = slurp(argv[1], NULL);
    const char* fs_src = slurp(argv[2], NULL);
    size_t in_len = 0;
    const char* in = slurp(argv[3], &in_len);
    const int32_t* hdr = (const int32_t*)in;
    const int W = hdr[0], H = hdr[1], n = hdr[2], texw = hdr[3], texh = hdr[4];
    const float* fl = (const float*)(in + 20);
    const float *view = fl, *proj = fl + 16, *focal = fl + 32, *viewport = fl + 34;
    const uint32_t* tex = (const uint32_t*)(fl + 36);
    const int32_t* index = (const int32_t*)(tex + (size_t)texw * texh * 4);
    if (W < 1 || H < 1 || n < 0 || texw != 2048 || texh < 1 ||
        in_len != 20 + 36 * 4 + (size_t)texw * texh * 16 + (size_t)n * 4)
        die("in.bin: bad header or size");

    void* drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) die(dlerror());
    const __DRIextension** (*get_exts)(void) = (const __DRIextension** (*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) die("swrast_dri.so has no __driDriverGetExtensions_swrast");
    const __DRIextension** exts = get_exts();
    const __DRIcoreExtension* core = NULL;
    const __DRIswrastExtension* sw = NULL;
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension*)exts[i];
    }
    if (!core || !sw || sw->base.version < 4) die("driver lacks DRI_Core / DRI_SWRast v4");
    const __DRIconfig** configs = NULL;
    __DRIscreen* scr = sw->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!scr || !configs || !configs[0]) die("createNewScreen2 failed");
    unsigned err = 0;
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 0};  // WebGL2 = OpenGL ES 3.0
    __DRIcontext* ctx = sw->createContextAttribs(scr, __DRI_API_GLES3, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) die("no OpenGL ES 3 context");
    __DRIdrawable* dr = sw->createNewDrawable(scr, configs[0], NULL);
    if (!dr || !core->bindContext(ctx, dr, dr)) die("bindContext failed");
    void* ga = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!ga) die(dlerror());
    gpa = (void* (*)(const char*))dlsym(ga, "_glapi_get_proc_address");
    if (!gpa) die("no _glapi_get_proc_address");

    GLF(PFNGLGETSTRINGPROC, glGetString);
    GLF(PFNGLGETERRORPROC, glGetError);
    GLF(PFNGLCREATESHADERPROC, glCreateShader);
    GLF(PFNGLSHADERSOURCEPROC, glShaderSource);
    GLF(PFNGLCOMPILESHADERPROC, glCompileShader);
    GLF(PFNGLGETSHADERIVPROC, glGetShaderiv);
    GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog);
    GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram);
    GLF(PFNGLATTACHSHADERPROC, glAttachShader);
    GLF(PFNGLLINKPROGRAMPROC, glLinkProgram);
    GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv);
    GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog);
    GLF(PFNGLUSEPROGRAMPROC, glUseProgram);
    GLF(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
    GLF(PFNGLGETATTRIBLOCATIONPROC, glGetAttribLocation);
    GLF(PFNGLUNIFORM1IPROC, glUniform1i);
    GLF(PFNGLUNIFORM2FVPROC, glUniform2fv);
    GLF(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv);
    GLF(PFNGLGENBUFFERSPROC, glGenBuffers);
    GLF(PFNGLBINDBUFFERPROC, glBindBuffer);
    GLF(PFNGLBUFFERDATAPROC, glBufferData);
    GLF(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays);
    GLF(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray);
    GLF(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray);
    GLF(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer);
    GLF(PFNGLVERTEXATTRIBIPOINTERPROC, glVertexAttribIPointer);
    GLF(PFNGLVERTEXATTRIBDIVISORPROC, glVertexAttribDivisor);
    GLF(PFNGLGENTEXTURESPROC, glGenTextures);
    GLF(PFNGLBINDTEXTUREPROC, glBindTexture);
    GLF(PFNGLACTIVETEXTUREPROC, glActiveTexture);
    GLF(PFNGLTEXPARAMETERIPROC, glTexParameteri);
    GLF(PFNGLTEXIMAGE2DPROC, glTexImage2D);
    GLF(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers);
    GLF(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer);
    GLF(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D);
    GLF(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus);
    GLF(PFNGLVIEWPORTPROC, glViewport);
    GLF(PFNGLDISABLEPROC, glDisable);
    GLF(PFNGLENABLEPROC, glEnable);
    GLF(PFNGLBLENDFUNCSEPARATEPROC, glBlendFuncSeparate);
    GLF(PFNGLBLENDEQUATIONSEPARATEPROC, glBlendEquationSeparate);
    GLF(PFNGLCLEARCOLORPROC, glClearColor);
    GLF(PFNGLCLEARPROC, glClear);
    GLF(PFNGLDRAWARRAYSINSTANCEDPROC, glDrawArraysInstanced);
    GLF(PFNGLREADPIXELSPROC, glReadPixels);
    GLF(PFNGLPIXELSTOREIPROC, glPixelStorei);
    GLF(PFNGLFINISHPROC, glFinish);

    fprintf(stderr, "gl_frames: %s | %s | GLSL %s\n", (const char*)glGetString(GL_VERSION), (const char*)glGetString(GL_RENDERER),
            (const char*)glGetString(GL_SHADING_LANGUAGE_VERSION));
    const char* ext = (const char*)glGetString(GL_EXTENSIONS);
    if (!ext || !strstr(ext, "GL_EXT_color_buffer_float") || !strstr(ext, "GL_EXT_float_blend"))
        die("the GL lacks EXT_color_buffer_float / EXT_float_blend (fp32 colour buffer with blending)");

    GLuint sh[2];
    const GLenum kinds[2] = {GL_VERTEX_SHADER, GL_FRAGMENT_SHADER};
    const char* srcs[2] = {vs_src, fs_src};
    char log[4096];
    for (int i = 0; i < 2; ++i) {
        sh[i] = glCreateShader(kinds[i]);
        glShaderSource(sh[i], 1, &srcs[i], NULL);
        glCompileShader(sh[i]);
        GLint ok = 0;
        glGetShaderiv(sh[i], GL_COMPILE_STATUS, &ok);
        if (!ok) {
            glGetShaderInfoLog(sh[i], sizeof log, NULL, log);
            fprintf(stderr, "%s\n", log);
            die("shader does not compile");
        }
    }
    const GLuint prog = glCreateProgram();
    glAttachShader(prog, sh[0]);
    glAttachShader(prog, sh[1]);
    glLinkProgram(prog);
    GLint linked = 0;
    glGetProgramiv(prog, GL_LINK_STATUS, &linked);
    if (!linked) {
        glGetProgramInfoLog(prog, sizeof log, NULL, log);
        fprintf(stderr, "%s\n", log);
        die("program does not link");
    }
    glUseProgram(prog);
    glDisable(GL_DEPTH_TEST);
    glEnable(GL_BLEND);
    glBlendFuncSeparate(GL_ONE_MINUS_DST_ALPHA, GL_ONE, GL_ONE_MINUS_DST_ALPHA, GL_ONE);
    glBlendEquationSeparate(GL_FUNC_ADD, GL_FUNC_ADD);
    glUniform1i(glGetUniformLocation(prog, "uSelectionMode"), 0);
    glUniform1i(glGetUniformLocation(prog, "uSelectedLabel"), -2);  // NO_SELECTION: matches no label (uSelectionMode is 0 anyway)

    GLuint vao;
    glGenVertexArrays(1, &vao);
    glBindVertexArray(vao);
    const float quad[8] = {-2, -2, 2, -2, 2, 2, -2, 2};
    GLuint vbuf, ibuf;
    glGenBuffers(1, &vbuf);
    glBindBuffer(GL_ARRAY_BUFFER, vbuf);
    glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
    const GLint a_position = glGetAttribLocation(prog, "position");
    const GLint a_index = glGetAttribLocation(prog, "index");
    if (a_position < 0 || a_index < 0) die("attributes not found");
    glEnableVertexAttribArray((GLuint)a_position);
    glVertexAttribPointer((GLuint)a_position, 2, GL_FLOAT, GL_FALSE, 0, 0);

    GLuint texture;
    glGenTextures(1, &texture);
    glActiveTexture(GL_TEXTURE0);
    glBindTexture(GL_TEXTURE_2D, texture);
    glUniform1i(glGetUniformLocation(prog, "u_texture"), 0);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32UI, texw, texh, 0, GL_RGBA_INTEGER, GL_UNSIGNED_INT, tex);

    glGenBuffers(1, &ibuf);
    glEnableVertexAttribArray((GLuint)a_index);
    glBindBuffer(GL_ARRAY_BUFFER, ibuf);
    glVertexAttribIPointer((GLuint)a_index, 1, GL_INT, 0, 0);
    glVertexAttribDivisor((GLuint)a_index, 1);
    glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)n * 4, index, GL_DYNAMIC_DRAW);

    glUniform2fv(glGetUniformLocation(prog, "focal"), 1, focal);
    glUniform2fv(glGetUniformLocation(prog, "viewport"), 1, viewport);
    glUniformMatrix4fv(glGetUniformLocation(prog, "projection"), 1, GL_FALSE, proj);
    glUniformMatrix4fv(glGetUniformLocation(prog, "view"), 1, GL_FALSE, view);

    GLuint fbo, color;
    glGenTextures(1, &color);
    glActiveTexture(GL_TEXTURE1);
    glBindTexture(GL_TEXTURE_2D, color);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, GL_RGBA, GL_FLOAT, NULL);
    glActiveTexture(GL_TEXTURE0);
    glGenFramebuffers(1, &fbo);
    glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, color, 0);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("fp32 framebuffer incomplete");
    glViewport(0, 0, W, H);
    glClearColor(0.f, 0.f, 0.f, 0.f);
    glClear(GL_COLOR_BUFFER_BIT);
    if (n > 0) glDrawArraysInstanced(GL_TRIANGLE_FAN, 0, 4, n);
    glFinish();
    float* px = malloc((size_t)W * H * 16);
    if (!px) die("out of memory");
    glPixelStorei(GL_PACK_ALIGNMENT, 1);
    glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, px);
    const GLenum e = glGetError();
    if (e != GL_NO_ERROR) {
        fprintf(stderr, "GL error 0x%x\n", e);
        die("GL error");
    }
    FILE* out = fopen(argv[4], "wb");
    if (!out) die(argv[4]);
    for (int row = H - 1; row >= 0; --row)
        if (fwrite(px + (size_t)row * W * 4, 16, (size_t)W, out) != (size_t)W) die("write failed");
    fclose(out);
    return 0;
}
