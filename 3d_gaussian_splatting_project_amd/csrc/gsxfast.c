/* _gsxfast - the per-view hand-over without the interpreter's share of it.
 *
 * Context.vote_view runs once per view: 200 times in the 8 ms of a labelling run, and through ctypes one call costs ~2.4 us
 * of Python (type checks, the array's address, eight argument conversions) next to ~35 us of packing.  This CPython module
 * does the same checks in C through the buffer protocol (no numpy C API) and calls gsx_vote_view through the function
 * pointer it is given; the GIL is released for the call, as ctypes does.  It holds no state and contains no logic of the
 * labeler: labeler.py falls back to its ctypes path when the module is missing or declines an argument. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef int (*vote_view_fn)(void*, const void*, const void*, int32_t, int32_t, int32_t, int32_t, int32_t);

enum { SEG_I32 = 0, SEG_I64 = 1, SEG_U8 = 2, SEG_U8_LABELS = 3 }; /* include/gsx.h */
#define DECLINED (-1000) /* not a contiguous 2-D int32 / int64 / uint8 buffer: the caller takes the general path */

/* vote_view(fn_addr, ctx_addr, camera_addr, seg, packed_u8, img_w, img_h) -> status
 * img_w < 0: the image has the map's own size */
static PyObject* fast_vote_view(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 7) {
        PyErr_SetString(PyExc_TypeError, "vote_view(fn, ctx, camera, seg, packed_u8, img_w, img_h)");
        return NULL;
    }
    void* fn = PyLong_AsVoidPtr(args[0]);
    void* ctx = PyLong_AsVoidPtr(args[1]);
    void* cam = PyLong_AsVoidPtr(args[2]);
    const int packed = PyObject_IsTrue(args[4]);
    long iw = PyLong_AsLong(args[5]), ih = PyLong_AsLong(args[6]);
    if (PyErr_Occurred()) return NULL;
    Py_buffer view;
    if (PyObject_GetBuffer(args[3], &view, PyBUF_RECORDS_RO) != 0) {
        PyErr_Clear();
        return PyLong_FromLong(DECLINED);
    }
    int dt = -1;
    const char* f = view.format ? view.format : "B";
    if (f[0] == '<' || f[0] == '=' || f[0] == '@') ++f; /* native little-endian only (this library is x86-64 + gfx950) */
    if (f[0] && !f[1]) {
        if (f[0] == 'i' && view.itemsize == 4) dt = SEG_I32;
        else if ((f[0] == 'l' || f[0] == 'q') && view.itemsize == 8) dt = SEG_I64;
        else if (f[0] == 'B' && view.itemsize == 1) dt = packed ? SEG_U8 : SEG_U8_LABELS;
    }
    long rc = DECLINED;
    if (dt >= 0 && view.ndim == 2 && PyBuffer_IsContiguous(&view, 'C') && view.shape[0] > 0 && view.shape[1] > 0 &&
        view.shape[0] <= INT32_MAX && view.shape[1] <= INT32_MAX && fn && ctx && cam) {
        const int32_t h = (int32_t)view.shape[0], w = (int32_t)view.shape[1];
        if (iw < 0) iw = w, ih = h;
        if (iw <= INT32_MAX && ih <= INT32_MAX) {
            int r;
            Py_BEGIN_ALLOW_THREADS
            r = ((vote_view_fn)fn)(ctx, cam, view.buf, dt, w, h, (int32_t)iw, (int32_t)ih);
            Py_END_ALLOW_THREADS
            rc = r;
        }
    }
    PyBuffer_Release(&view);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {{"vote_view", (PyCFunction)(void (*)(void))fast_vote_view, METH_FASTCALL,
                                 "gsx_vote_view for a contiguous 2-D int32 / int64 / uint8 buffer; -1000 if the argument is anything else"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_gsxfast", "fast path of Context.vote_view", -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__gsxfast(void) {
    PyObject* m = PyModule_Create(&module);
    if (m) PyModule_AddIntConstant(m, "DECLINED", DECLINED);
    return m;
}
