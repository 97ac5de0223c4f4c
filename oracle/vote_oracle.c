/*
 * ORACLE (test infrastructure, NOT the product): CPU restatement of the reference's
 * majority-vote labeler.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product path (libgsx.so) never links or calls it.
 *
 * Restates, operation for operation and in fp64:
 *   project_gaussian   /root/reference/deep_learning_segmentation.py:43-82
 *   assign_labels      /root/reference/deep_learning_segmentation.py:252-308 (vote loop + arg-max)
 *
 * Third-party arithmetic on the path: numpy's `R @ v` (deep_learning_segmentation.py:66,69) is
 * OpenBLAS dgemv (NumPy 2.2.6 / OpenBLAS 0.3.29, unpinned by the reference's environment.yml).
 * For a C-contiguous 3x3 R it evaluates each row as
 *        fma(R[r][2], v2, fma(R[r][0], v0, R[r][1]*v1))
 * (established empirically against the reference in the build container, 9000/9000 rows, and
 * pinned by tests/golden/vote_project.npz + vote_assign.npz, which were produced by importing
 * the reference itself: tools/make_golden.py).  Parity status: PINNED by those fixtures.
 *
 * Build: see oracle/Makefile  (-ffp-contract=off is mandatory; fma() must stay explicit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    double fx, fy;
    int32_t width, height;
    double R[9];   /* row-major, cameras.json "rotation" */
    double p[3];   /* cameras.json "position" */
} gsxo_camera;

typedef struct {
    gsxo_camera cam;
    const int32_t* seg; /* seg_h x seg_w, row-major, any int label */
    int32_t seg_w, seg_h;
    int32_t img_w, img_h; /* PIL image size (deep_learning_segmentation.py:261-263) */
} gsxo_view;

static inline double row_dot(const double* Rr, double v0, double v1, double v2) {
    /* OpenBLAS dgemv_t association for n = 3 (see header) */
    return fma(Rr[2], v2, fma(Rr[0], v0, Rr[1] * v1));
}

/* t = -R @ p  (deep_learning_segmentation.py:66): unary minus binds first, then matmul */
static void cam_translation(const gsxo_camera* c, double t[3]) {
    double nR[9];
    for (int i = 0; i < 9; ++i) nR[i] = -c->R[i];
    for (int r = 0; r < 3; ++r) t[r] = row_dot(nR + 3 * r, c->p[0], c->p[1], c->p[2]);
}

/* returns 1 and writes (*x,*y) if the reference returns a tuple, 0 if it returns None */
static inline int project_with_t(const float pos[3], const gsxo_camera* c, const double t[3], int* x, int* y) {
    const double v0 = (double)pos[0], v1 = (double)pos[1], v2 = (double)pos[2]; /* f32 promoted, :69 */
    const double pc0 = row_dot(c->R + 0, v0, v1, v2) + t[0];
    const double pc1 = row_dot(c->R + 3, v0, v1, v2) + t[1];
    const double pc2 = row_dot(c->R + 6, v0, v1, v2) + t[2];
    if (pc2 <= 0.0) return 0;                                  /* :72 (NaN falls through, as in Python) */
    const double px = (c->fx * pc0) / pc2 + (double)c->width / 2.0;   /* :76 */
    const double py = (c->fy * pc1) / pc2 + (double)c->height / 2.0;  /* :77 */
    if (0.0 <= px && px < (double)c->width && 0.0 <= py && py < (double)c->height) { /* :80 */
        *x = (int)px; /* int() truncates, :81 */
        *y = (int)py;
        return 1;
    }
    return 0;
}

int gsxo_project(const float pos[3], const gsxo_camera* cam, int* x, int* y) {
    double t[3];
    cam_translation(cam, t);
    return project_with_t(pos, cam, t, x, y);
}

/* batch form used by the tests: positions AoS N x 3; x,y get -1 where not visible */
void gsxo_project_many(const float* positions, int64_t n, const gsxo_camera* cam, int32_t* x, int32_t* y) {
    double t[3];
    cam_translation(cam, t);
    for (int64_t i = 0; i < n; ++i) {
        int xi, yi;
        if (project_with_t(positions + 3 * i, cam, t, &xi, &yi)) { x[i] = xi; y[i] = yi; }
        else { x[i] = -1; y[i] = -1; }
    }
}

/* seg-map lookup for one (Gaussian, view): deep_learning_segmentation.py:281-288. 1 = voted. */
static inline int lookup(const gsxo_view* v, const double t[3], const float pos[3], int32_t* label) {
    int x, y;
    if (!project_with_t(pos, &v->cam, t, &x, &y)) return 0;
    const double width_scale = (double)v->seg_w / (double)v->img_w;    /* :271 */
    const double height_scale = (double)v->seg_h / (double)v->img_h;   /* :270 */
    double xs = trunc((double)x * width_scale);                        /* :281 int() */
    double ys = trunc((double)y * height_scale);                       /* :282 */
    int64_t xi = xs > (double)(v->seg_w - 1) ? v->seg_w - 1 : (int64_t)xs; /* :285 min(max(0,.),w-1) */
    int64_t yi = ys > (double)(v->seg_h - 1) ? v->seg_h - 1 : (int64_t)ys;
    if (xi < 0) xi = 0;
    if (yi < 0) yi = 0;
    *label = v->seg[yi * (int64_t)v->seg_w + xi];                      /* :288 */
    return 1;
}

/*
 * Majority vote. `views` are the cameras the reference actually processes, in order (cameras
 * whose PNG is missing are skipped by the caller, :256-259).  Votes are kept per Gaussian as an
 * insertion-ordered (label,count) list = the reference's dict; arg-max returns the FIRST entry
 * holding the maximum (Python max(), :303).  No vote at all -> -1 (:306).
 * threads <= 0: use all OpenMP threads.  Returns the number of threads used.
 */
int gsxo_assign_labels(const float* positions, int64_t n, const gsxo_view* views, int32_t n_views,
                       int32_t* labels_out, int threads) {
    double* t = (double*)malloc(sizeof(double) * 3 * (size_t)(n_views > 0 ? n_views : 1));
    for (int v = 0; v < n_views; ++v) cam_translation(&views[v].cam, t + 3 * v);
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel num_threads(threads)
#endif
    {
        int32_t* lab = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_views > 0 ? n_views : 1));
        int32_t* cnt = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_views > 0 ? n_views : 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < n; ++i) {
            int m = 0;
            for (int v = 0; v < n_views; ++v) {
                int32_t label;
                if (!lookup(&views[v], t + 3 * v, positions + 3 * i, &label)) continue;
                int k = 0;
                while (k < m && lab[k] != label) ++k;
                if (k == m) { lab[m] = label; cnt[m] = 0; ++m; }       /* :293-294 */
                cnt[k] += 1;                                            /* :295 */
            }
            int32_t best = -1;
            int32_t bestc = 0;
            for (int k = 0; k < m; ++k)
                if (cnt[k] > bestc) { bestc = cnt[k]; best = lab[k]; } /* strict > keeps the first max */
            labels_out[i] = best;
        }
        free(lab);
        free(cnt);
    }
    free(t);
    return used;
}


/* per-Gaussian vote of ONE view: out[i] = label+1 (the histogram bin) or -1 when the reference
 * casts no vote (dls.py:276-288).  Lets the tests rebuild per-rank histogram planes. */
void gsxo_view_bins(const float* positions, int64_t n, const gsxo_view* view, int32_t* out) {
    double t[3];
    cam_translation(&view->cam, t);
    for (int64_t i = 0; i < n; ++i) {
        int32_t label;
        out[i] = lookup(view, t, positions + 3 * i, &label) ? label + 1 : -1;
    }
}

int gsxo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
