/*
 * oracle/y7t_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C CPU restatement of the THIRD-PARTY native arithmetic the reference's
 * tracker hot path calls but does not vendor:
 *
 *   (1) lap.lapjv            (PyPI "lap", un-pinned by the reference; 0.4.0 semantics)
 *         call site: /root/reference/tracker/matching.py:34
 *   (2) cython_bbox.bbox_overlaps (PyPI "cython_bbox", un-pinned; Fast-R-CNN bbox.pyx)
 *         call site: /root/reference/tracker/matching.py:56-59
 *   (3) torchvision.ops.nms  (README.md:80 pins torchvision 0.8.0)
 *         call site: /root/reference/utils/general.py:679
 *
 * None of these packages is installed in the build container and the reference
 * ships no tests or golden vectors for them, so the parity of THESE THREE functions
 * with the real packages is "parity unpinned": the algorithms are restated from
 * their published sources (Jonker & Volgenant 1987 as implemented by lap's
 * lapjv.cpp; the Fast-R-CNN "+1" box-overlap convention; torchvision's greedy NMS)
 * and cross-checked in tests/ against scipy.optimize.linear_sum_assignment and
 * brute-force restatements.  Everything ELSE on the path (Kalman filters, tracker
 * state machines, detector graph) is pinned against the reference's own Python
 * sources imported in the build container (oracle/ref_harness.py).
 *
 * Build: make -C oracle   ->  oracle/liby7t_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LARGE 1000000.0

/* ------------------------------------------------------------------ */
/* (1) Jonker-Volgenant dense LAP, following lap's lapjv.cpp           */
/*     phases: column reduction + reduction transfer, augmenting row   */
/*     reduction (x2), augmentation by modified Dijkstra.              */
/* ------------------------------------------------------------------ */

static int ccrrt_dense(int n, const double *cost, int *free_rows, int *x, int *y, double *v)
{
    int n_free_rows = 0;
    char *unique = (char *)malloc((size_t)n);
    for (int i = 0; i < n; i++) { x[i] = -1; v[i] = LARGE; y[i] = 0; }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const double c = cost[(size_t)i * n + j];
            if (c < v[j]) { v[j] = c; y[j] = i; }
        }
    memset(unique, 1, (size_t)n);
    {
        int j = n;
        do {
            j--;
            const int i = y[j];
            if (x[i] < 0) x[i] = j;
            else { unique[i] = 0; y[j] = -1; }
        } while (j > 0);
    }
    for (int i = 0; i < n; i++) {
        if (x[i] < 0) {
            free_rows[n_free_rows++] = i;
        } else if (unique[i]) {
            const int j = x[i];
            double min = LARGE;
            for (int j2 = 0; j2 < n; j2++) {
                if (j2 == j) continue;
                const double c = cost[(size_t)i * n + j2] - v[j2];
                if (c < min) min = c;
            }
            v[j] -= min;
        }
    }
    free(unique);
    return n_free_rows;
}

static int carr_dense(int n, const double *cost, int n_free_rows, int *free_rows, int *x, int *y, double *v)
{
    unsigned current = 0, rr_cnt = 0;
    int new_free_rows = 0;
    while (current < (unsigned)n_free_rows) {
        int i0, j1, j2;
        double v1, v2, v1_new;
        int v1_lowers;
        rr_cnt++;
        const int free_i = free_rows[current++];
        const double *row = cost + (size_t)free_i * n;
        j1 = 0; v1 = row[0] - v[0];
        j2 = -1; v2 = LARGE;
        for (int j = 1; j < n; j++) {
            const double c = row[j] - v[j];
            if (c < v2) {
                if (c >= v1) { v2 = c; j2 = j; }
                else { v2 = v1; v1 = c; j2 = j1; j1 = j; }
            }
        }
        i0 = y[j1];
        v1_new = v[j1] - (v2 - v1);
        v1_lowers = v1_new < v[j1];
        if (rr_cnt < current * (unsigned)n) {
            if (v1_lowers) v[j1] = v1_new;
            else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = y[j2]; }
            if (i0 >= 0) {
                if (v1_lowers) free_rows[--current] = i0;
                else free_rows[new_free_rows++] = i0;
            }
        } else {
            if (i0 >= 0) free_rows[new_free_rows++] = i0;
        }
        x[free_i] = j1;
        y[j1] = free_i;
    }
    return new_free_rows;
}

static unsigned find_dense(int n, unsigned lo, const double *d, int *cols)
{
    unsigned hi = lo + 1;
    double mind = d[cols[lo]];
    for (unsigned k = hi; k < (unsigned)n; k++) {
        int j = cols[k];
        if (d[j] <= mind) {
            if (d[j] < mind) { hi = lo; mind = d[j]; }
            cols[k] = cols[hi];
            cols[hi++] = j;
        }
    }
    return hi;
}

static int scan_dense(int n, const double *cost, unsigned *plo, unsigned *phi,
                      double *d, int *cols, int *pred, const int *y, const double *v)
{
    unsigned lo = *plo, hi = *phi;
    while (lo != hi) {
        int j = cols[lo++];
        const int i = y[j];
        const double mind = d[j];
        const double *row = cost + (size_t)i * n;
        const double h = row[j] - v[j] - mind;
        for (unsigned k = hi; k < (unsigned)n; k++) {
            j = cols[k];
            const double cred_ij = row[j] - v[j] - h;
            if (cred_ij < d[j]) {
                d[j] = cred_ij;
                pred[j] = i;
                if (cred_ij == mind) {
                    if (y[j] < 0) return j;
                    cols[k] = cols[hi];
                    cols[hi++] = j;
                }
            }
        }
    }
    *plo = lo; *phi = hi;
    return -1;
}

static int find_path_dense(int n, const double *cost, int start_i, const int *y, double *v,
                           int *pred, int *cols, double *d)
{
    unsigned lo = 0, hi = 0, n_ready = 0;
    int final_j = -1;
    for (int i = 0; i < n; i++) {
        cols[i] = i; pred[i] = start_i;
        d[i] = cost[(size_t)start_i * n + i] - v[i];
    }
    while (final_j == -1) {
        if (lo == hi) {
            n_ready = lo;
            hi = find_dense(n, lo, d, cols);
            for (unsigned k = lo; k < hi; k++) {
                const int j = cols[k];
                if (y[j] < 0) final_j = j;
            }
        }
        if (final_j == -1) final_j = scan_dense(n, cost, &lo, &hi, d, cols, pred, y, v);
    }
    {
        const double mind = d[cols[lo]];
        for (unsigned k = 0; k < n_ready; k++) {
            const int j = cols[k];
            v[j] += d[j] - mind;
        }
    }
    return final_j;
}

static void ca_dense(int n, const double *cost, int n_free_rows, const int *free_rows, int *x, int *y, double *v)
{
    int *pred = (int *)malloc(sizeof(int) * (size_t)n);
    int *cols = (int *)malloc(sizeof(int) * (size_t)n);
    double *d = (double *)malloc(sizeof(double) * (size_t)n);
    for (int f = 0; f < n_free_rows; f++) {
        int i = -1, j;
        j = find_path_dense(n, cost, free_rows[f], y, v, pred, cols, d);
        while (i != free_rows[f]) {
            i = pred[j];
            y[j] = i;
            const int t = j; j = x[i]; x[i] = t;
        }
    }
    free(pred); free(cols); free(d);
}

/* square n x n problem; x[i] = column of row i, y[j] = row of column j */
void y7o_lapjv_square(int n, const double *cost, int *x, int *y)
{
    if (n <= 0) return;
    int *free_rows = (int *)malloc(sizeof(int) * (size_t)n);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    int ret = ccrrt_dense(n, cost, free_rows, x, y, v);
    int i = 0;
    while (ret > 0 && i < 2) { ret = carr_dense(n, cost, ret, free_rows, x, y, v); i++; }
    if (ret > 0) ca_dense(n, cost, ret, free_rows, x, y, v);
    free(v); free(free_rows);
}

/*
 * lap.lapjv(cost (nr x nc), extend_cost=True, cost_limit=limit) wrapper semantics
 * (lap/_lapjv.pyx): if cost_limit < inf the problem is embedded in an
 * (nr+nc)^2 matrix filled with limit/2, bottom-right block 0, top-left = cost;
 * assignments to dummy rows/cols are reported as -1.  Returns opt = sum of the
 * kept costs.  x has nr entries, y has nc entries.
 */
double y7o_lapjv_extend(const double *cost, int nr, int nc, double limit, int *x, int *y)
{
    const int n = nr + nc;
    if (n == 0) return 0.0;
    double *ext = (double *)malloc(sizeof(double) * (size_t)n * n);
    int *xe = (int *)malloc(sizeof(int) * (size_t)n);
    int *ye = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double c;
            if (i < nr && j < nc) c = cost[(size_t)i * nc + j];
            else if (i >= nr && j >= nc) c = 0.0;
            else c = limit / 2.0;
            ext[(size_t)i * n + j] = c;
        }
    y7o_lapjv_square(n, ext, xe, ye);
    double opt = 0.0;
    for (int i = 0; i < nr; i++) {
        x[i] = (xe[i] >= nc) ? -1 : xe[i];
        if (x[i] >= 0) opt += cost[(size_t)i * nc + x[i]];
    }
    for (int j = 0; j < nc; j++) y[j] = (ye[j] >= nr) ? -1 : ye[j];
    free(ext); free(xe); free(ye);
    return opt;
}

/* ------------------------------------------------------------------ */
/* (2) cython_bbox.bbox_overlaps: float64, "+1" pixel convention        */
/* ------------------------------------------------------------------ */
void y7o_bbox_overlaps(const double *boxes, int n, const double *query, int k, double *out)
{
    for (int kk = 0; kk < k; kk++) {
        const double *q = query + 4 * (size_t)kk;
        const double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
        for (int nn = 0; nn < n; nn++) {
            const double *b = boxes + 4 * (size_t)nn;
            double ov = 0.0;
            const double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]) + 1;
            if (iw > 0) {
                const double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]) + 1;
                if (ih > 0) {
                    const double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
                    ov = iw * ih / ua;
                }
            }
            out[(size_t)nn * k + kk] = ov;
        }
    }
}

/* ------------------------------------------------------------------ */
/* (3) torchvision.ops.nms greedy semantics, float32                    */
/*     order[] must already hold indices sorted by score descending.    */
/*     Returns number kept; keep[] receives original indices in order.  */
/* ------------------------------------------------------------------ */
int y7o_nms_f32(const float *boxes, const int *order, int n, float iou_thr, int *keep)
{
    char *sup = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int nk = 0;
    for (int _i = 0; _i < n; _i++) {
        const int i = order[_i];
        if (sup[i]) continue;
        keep[nk++] = i;
        const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        const float iarea = (ix2 - ix1) * (iy2 - iy1);
        for (int _j = _i + 1; _j < n; _j++) {
            const int j = order[_j];
            if (sup[j]) continue;
            const float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
            const float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;
            const float jarea = (boxes[4 * j + 2] - boxes[4 * j]) * (boxes[4 * j + 3] - boxes[4 * j + 1]);
            const float ovr = inter / (iarea + jarea - inter);
            if (ovr > iou_thr) sup[j] = 1;
        }
    }
    free(sup);
    return nk;
}
