// tests/_hostsim/y7t_hostsim.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the portable workgroup programs of yolov7-tracker_amd/csrc/y7t_track_*.h for the
// CPU (one "thread", nt = 1) so the control flow of the device tracker can be exercised by the
// `-m "not gpu"` suite where no GPU exists.  The product package never loads this library.
#define Y7T_HOSTSIM 1
static int g_literal_calls = 0;      // how many assignments met a tie and were re-solved by the literal lapjv
#define Y7T_COUNT_LITERAL() (++g_literal_calls)
static int g_tie_reason[8] = {0};
#define Y7T_TIE_REASON(k) (++g_tie_reason[k])
static int g_next_stat[4] = {0};
#define Y7T_NEXT_STAT(k) (++g_next_stat[k])
#include "../../yolov7-tracker_amd/csrc/y7t_track_step.h"
#include "../../yolov7-tracker_amd/csrc/y7t_track_deepsort.h"
#include <stdlib.h>
#include <string.h>

// the workgroup's fast scratch (LDS on the device): off by default (everything in the state blob); hs_set_fast_bytes(n) gives the programs n bytes of it,
// so the placement logic (work arrays / cost matrix / candidate lists in fast scratch when they fit) runs on the host too
static char g_fast[160 * 1024] __attribute__((aligned(64)));
static unsigned g_fast_bytes = 0;
// the launch-long home of the index lists (LDS on the device, k_tracker_step_frames): hs_arena_begin(blob) loads the lists into it and every step until
// hs_arena_end(blob) runs on them, like the frames of one device launch
static char g_arena[160 * 1024] __attribute__((aligned(64)));
static unsigned g_arena_bytes = 0;
static Y7TExec hs_ex() {
    Y7TExec e; e.tid = 0; e.nt = 1; e.rv = 0; e.ri = 0; e.fast = g_fast_bytes ? g_fast : 0; e.fast_bytes = g_fast_bytes;
    e.arena = g_arena_bytes ? g_arena : 0; e.arena_bytes = g_arena_bytes;
    return e;
}

extern "C" {
size_t hs_tracker_bytes(int cap_t, int cap_d) { return y7t_trk_layout(cap_t, cap_d).total; }

void hs_tracker_init(void* blob, int tracker, int kf, int cap_t, int cap_d, int max_time_lost, int f32_quirk,
                     double det_thresh, double low_thresh, double iou_thresh, int* id_counter) {
    Y7TTrkCfg c;
    c.tracker = tracker; c.kf = kf; c.cap_t = cap_t; c.cap_d = cap_d; c.max_time_lost = max_time_lost;
    c.f32_quirk = f32_quirk; c.det_thresh = det_thresh; c.low_thresh = low_thresh; c.iou_thresh = iou_thresh;
    y7t_tracker_init(hs_ex(), blob, c, (unsigned long long)(uintptr_t)id_counter);
}

int hs_tracker_step(void* blob, const float* dets, int n, double* out_rows, int out_cap, const double* warp) {
    int cnt = 0;
    y7t_tracker_step(hs_ex(), blob, dets, n, out_rows, out_cap, &cnt, warp);
    return cnt;
}
void hs_kf_gmc(const double* H, double* mean, double* cov) { y7t_kf_gmc(y7t_warp_load(H), mean, cov); }

int hs_arena_begin(void* blob) {
    const Y7TTrkHdr* h = (const Y7TTrkHdr*)blob;
    const size_t need = y7t_arena_bytes(h->cfg.cap_t, h->cfg.cap_d);
    if (need > sizeof(g_arena)) return 0;
    memset(g_arena, 0xA5, sizeof(g_arena));      // (nothing may depend on what the arena held before the load)
    g_arena_bytes = (unsigned)sizeof(g_arena);
    y7t_arena_load(hs_ex(), blob);
    return 1;
}
void hs_arena_end(void* blob) {
    if (!g_arena_bytes) return;
    y7t_arena_store(hs_ex(), blob);
    g_arena_bytes = 0;
}
int hs_tracker_status(void* blob) { return ((Y7TTrkHdr*)blob)->status; }
int hs_literal_calls() { return g_literal_calls; }
void hs_set_fast_bytes(int n) { g_fast_bytes = n < 0 ? 0 : (n > (int)sizeof(g_fast) ? (unsigned)sizeof(g_fast) : (unsigned)n); }
int hs_next_stat(int k) { return g_next_stat[k]; }
int hs_next_tracker() { return 1; }      // (the run-time candidate stride is the shipped path since round 3)
int hs_tie_reason(int k) { return g_tie_reason[k]; }

void hs_lapjv(const double* cost, int nr, int nc, double limit, int* x, int* y) {
    Y7TLap L;
    L.c = cost; L.nr = nr; L.nc = nc; L.ld = nc; L.n = nr + nc; L.half = limit / 2.0; L.prof = nullptr;
    void* ws = malloc(y7t_lap_ws_bytes(L.n) + 64);
    y7t_lap_bind(L, ws, L.n);
    y7t_lap_solve(hs_ex(), L);
    for (int i = 0; i < nr; ++i) x[i] = L.x[i] >= nc ? -1 : L.x[i];
    for (int j = 0; j < nc; ++j) y[j] = L.y[j] >= nr ? -1 : L.y[j];
    free(ws);
}

void hs_lapsap(const double* cost, int nr, int nc, double limit, int* x, int* y) {
    Y7TLap L;
    L.c = cost; L.nr = nr; L.nc = nc; L.ld = nc; L.n = nr + nc; L.half = limit / 2.0; L.prof = nullptr;
    void* ws = malloc(y7t_lap_ws_bytes(L.n + 1) + 64);
    y7t_lap_bind(L, ws, L.n + 1);
    if (y7t_lap_solve_sap(hs_ex(), L)) y7t_lap_solve_literal(hs_ex(), L);
    for (int i = 0; i < nr; ++i) x[i] = L.x[i] >= nc ? -1 : L.x[i];
    for (int j = 0; j < nc; ++j) y[j] = L.y[j] >= nr ? -1 : L.y[j];
    free(ws);
}

void hs_laplit(const double* cost, int nr, int nc, double limit, int* x, int* y) {      // lapjv.cpp run literally (the re-solve of problems with ties)
    Y7TLap L;
    L.c = cost; L.nr = nr; L.nc = nc; L.ld = nc; L.n = nr + nc; L.half = limit / 2.0; L.prof = nullptr;
    void* ws = malloc(y7t_lap_ws_bytes(L.n + 1) + 64);
    y7t_lap_bind(L, ws, L.n + 1);
    y7t_lap_solve_literal(hs_ex(), L);
    for (int i = 0; i < nr; ++i) x[i] = L.x[i] >= nc ? -1 : L.x[i];
    for (int j = 0; j < nc; ++j) y[j] = L.y[j] >= nr ? -1 : L.y[j];
    free(ws);
}

// the float32 pre-test of the IoU pair passes (y7t_box_apart): 1 = "cannot overlap"; and the distance the passes use (pre-test + exact formula)
int hs_box_apart(const double* b, const double* q) { return y7t_box_apart(y7t_box_row(b), -1, y7t_box_col(q)) ? 1 : 0; }
double hs_box_iou_dist(const double* b, const double* q) { return y7t_box_iou_dist(y7t_box_row(b), -1, y7t_box_col(q)); }
void hs_iou_cost(const double* a, int n, const double* b, int m, double* cost) {
    for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) cost[(size_t)i * m + j] = y7t_iou_dist(a + 4 * i, b + 4 * j);
}
void hs_kf_initiate(int kind, const double* z, int f32_std, double* mean, double* cov) { y7t_kf_initiate(kind, z, f32_std, mean, cov); }
void hs_kf_predict(int kind, double* mean, double* cov) { y7t_kf_predict(kind, mean, cov); }
void hs_kf_update(int kind, double* mean, double* cov, const double* z, double conf) { y7t_kf_update(kind, mean, cov, z, conf); }
void hs_kf_project(int kind, const double* mean, const double* cov, double conf, double* pm, double* S) { y7t_kf_project(kind, mean, cov, conf, pm, S); }
double hs_kf_gating(int kind, const double* mean, const double* cov, const double* z, int only_pos) { return y7t_kf_gating(kind, mean, cov, z, only_pos); }

// ---- DeepSORT (y7t_track_deepsort.h) ----
size_t hs_feat_bytes(int cap_t, int cap_d, int dim, int budget) { return y7t_feat_layout(cap_t, cap_d, dim, budget).total; }
void hs_feat_init(void* fblob, int cap_t, int cap_d, int dim, int budget) { y7t_feat_init(hs_ex(), fblob, cap_t, cap_d, dim, budget); }
int hs_deepsort_step(void* blob, void* fblob, const float* dets, int n, const float* feats, double* out_rows, int out_cap) {
    int cnt = 0;
    const Y7TExec ex = hs_ex();
    Y7TTrkHdr* h = (Y7TTrkHdr*)blob;
    const Y7TTrk s = y7t_trk_bind(blob, h->cfg.cap_t, h->cfg.cap_d);
    const Y7TFeat f = y7t_feat_bind(fblob);
    y7t_feat_normalize_dets(ex, f, feats, n);
    for (int k = 0; k < h->n_tracked; ++k) y7t_embed_slot(ex, f, s.tracked[k], n, s.tsu[s.tracked[k]]);
    for (int k = 0; k < h->n_lost; ++k) y7t_embed_slot(ex, f, s.lost[k], n, s.tsu[s.lost[k]]);
    y7t_tracker_step_deepsort(ex, blob, fblob, dets, n, feats, out_rows, out_cap, &cnt);
    y7t_feat_store_pending(ex, f, feats);
    return cnt;
}
int hs_feat_status(void* fblob) { return ((Y7TFeatHdr*)fblob)->status; }
int hs_pyset_difference(int n, const int* member, int n_other, int* out) {
    int* tab = (int*)malloc(sizeof(int) * (2 * Y7T_PYSET_CAP + (size_t)n + 1));
    int* unm = tab + 2 * Y7T_PYSET_CAP;          // the keys of range(n) outside `other`, ascending (the step builds this list by compaction)
    int st = 0, n_unm = 0;
    for (int v = 0; v < n; ++v) if (!member[v]) unm[n_unm++] = v;
    const int c = y7t_pyset_difference_list(n, unm, n_unm, n_other, out, tab, &st);
    free(tab);
    return st ? -1 : c;
}
}
