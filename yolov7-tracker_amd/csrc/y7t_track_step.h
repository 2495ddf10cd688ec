// y7t_track_step.h -- the ByteTrack / SORT per-frame association state machine as ONE workgroup
// program over a device-resident struct-of-arrays track pool.  Portable text (see
// y7t_track_core.h for the two ways it is compiled).
//
// Restates /root/reference/tracker/bytetrack.py:41-204 (ByteTrack.update),
// tracker/basetrack.py:368-487 (BaseTracker.update == SORT), :489-537
// (update_without_detection), :222-339 (STrack.activate/re_activate/update/multi_predict),
// :183-219 (tlwh/tlbr), :540-576 (joint/sub/remove_duplicate_stracks).
//
// Lists (tracked / lost) are ordered arrays of pool slots and keep the reference's list order,
// because list position is what the assignment indices refer to and what the returned track
// order is.  The reference's ever-growing `removed_stracks` list is represented by the per-slot
// flag `inrem` (the list itself is only ever used for id-membership tests); a slot is recycled
// once it is in neither list.
#pragma once
#include "y7t_track_core.h"

struct Y7TTrkCfg {
    int tracker;        // Y7T_SORT / Y7T_BYTETRACK / Y7T_BOTSORT
    int kf;             // Kalman kind
    int cap_t, cap_d;   // capacities: live tracks (tracked+lost), detections per frame
    int max_time_lost;  // int(frame_rate / 30 * track_buffer)
    int f32_quirk;      // 1: reproduce numpy>=2 float32 flow of freshly initiated tracks
    double det_thresh;  // opts.conf_thresh
    double low_thresh;  // max(0.15, conf_thresh - 0.3)
    double iou_thresh;  // opts.iou_thresh (SORT)
};

struct Y7TTrkHdr {
    int magic, frame_id, n_tracked, n_lost, n_free, status, n_out, n_removed_total;
    int n_act_last, n_refind_last, n_lostn_last, n_removed_last;
    int pad0, pad1;
    unsigned long long id_counter_ptr;  // device int* shared by every tracker of the process
    Y7TTrkCfg cfg;
    long long prof[32];                 // shader-clock stamps of the last step's phases (thread 0; diagnostics)
};

// byte offsets of every array inside the state blob
struct Y7TTrkLayout {
    size_t mean, cov, box, score, cls, tid, start, frame, tsu, state, act, len, inrem, f32m;
    size_t tracked, lost, freel, mark;
    size_t pool, unconf, rem, actl, refind, lostn, removedl, tmpa, tmpb;
    size_t dhi, dlo, left, dbox, dtlbr, ttlbr, cost, lapws, xrow, ycol;
    size_t total;
};

Y7T_HD size_t y7t_al(size_t x) { return (x + 63) & ~(size_t)63; }

Y7T_HD Y7TTrkLayout y7t_trk_layout(int cap_t, int cap_d) {
    Y7TTrkLayout L;
    size_t o = y7t_al(sizeof(Y7TTrkHdr));
    const size_t T = (size_t)cap_t, D = (size_t)cap_d;
#define Y7T_TAKE(f, bytes) L.f = o; o = y7t_al(o + (bytes));
    Y7T_TAKE(mean, T * 8 * 8) Y7T_TAKE(cov, T * 64 * 8)
    Y7T_TAKE(box, T * 4 * 4) Y7T_TAKE(score, T * 4) Y7T_TAKE(cls, T * 4)
    Y7T_TAKE(tid, T * 4) Y7T_TAKE(start, T * 4) Y7T_TAKE(frame, T * 4) Y7T_TAKE(tsu, T * 4)
    Y7T_TAKE(state, T * 4) Y7T_TAKE(act, T * 4) Y7T_TAKE(len, T * 4) Y7T_TAKE(inrem, T * 4) Y7T_TAKE(f32m, T * 4)
    Y7T_TAKE(tracked, T * 4) Y7T_TAKE(lost, T * 4) Y7T_TAKE(freel, T * 4) Y7T_TAKE(mark, T * 4)
    Y7T_TAKE(pool, T * 4) Y7T_TAKE(unconf, T * 4) Y7T_TAKE(rem, T * 4) Y7T_TAKE(actl, T * 4)
    Y7T_TAKE(refind, T * 4) Y7T_TAKE(lostn, T * 4) Y7T_TAKE(removedl, T * 4) Y7T_TAKE(tmpa, T * 4) Y7T_TAKE(tmpb, T * 4)
    Y7T_TAKE(dhi, D * 4) Y7T_TAKE(dlo, D * 4) Y7T_TAKE(left, D * 4)
    Y7T_TAKE(dbox, D * 4 * 4) Y7T_TAKE(dtlbr, D * 4 * 8) Y7T_TAKE(ttlbr, T * 4 * 8)
    Y7T_TAKE(cost, T * (T > D ? T : D) * 8)
    Y7T_TAKE(lapws, y7t_lap_ws_bytes(cap_t + (cap_t > cap_d ? cap_t : cap_d)))
    Y7T_TAKE(xrow, T * 4) Y7T_TAKE(ycol, (T > D ? T : D) * 4)
#undef Y7T_TAKE
    L.total = o;
    return L;
}

struct Y7TTrk {
    Y7TTrkHdr* h;
    double *mean, *cov;
    float *box, *score, *cls;
    int *tid, *start, *frame, *tsu, *state, *act, *len, *inrem, *f32m;
    int *tracked, *lost, *freel, *mark;
    int *pool, *unconf, *rem, *actl, *refind, *lostn, *removedl, *tmpa, *tmpb;
    int *dhi, *dlo, *left;
    float* dbox;
    double *dtlbr, *ttlbr, *cost;
    void* lapws;
    int *xrow, *ycol;
};

Y7T_FN Y7TTrk y7t_trk_bind(void* blob, int cap_t, int cap_d) {
    const Y7TTrkLayout L = y7t_trk_layout(cap_t, cap_d);
    char* b = (char*)blob;
    Y7TTrk s;
    s.h = (Y7TTrkHdr*)b;
    s.mean = (double*)(b + L.mean); s.cov = (double*)(b + L.cov);
    s.box = (float*)(b + L.box); s.score = (float*)(b + L.score); s.cls = (float*)(b + L.cls);
    s.tid = (int*)(b + L.tid); s.start = (int*)(b + L.start); s.frame = (int*)(b + L.frame); s.tsu = (int*)(b + L.tsu);
    s.state = (int*)(b + L.state); s.act = (int*)(b + L.act); s.len = (int*)(b + L.len); s.inrem = (int*)(b + L.inrem);
    s.f32m = (int*)(b + L.f32m);
    s.tracked = (int*)(b + L.tracked); s.lost = (int*)(b + L.lost); s.freel = (int*)(b + L.freel); s.mark = (int*)(b + L.mark);
    s.pool = (int*)(b + L.pool); s.unconf = (int*)(b + L.unconf); s.rem = (int*)(b + L.rem); s.actl = (int*)(b + L.actl);
    s.refind = (int*)(b + L.refind); s.lostn = (int*)(b + L.lostn); s.removedl = (int*)(b + L.removedl);
    s.tmpa = (int*)(b + L.tmpa); s.tmpb = (int*)(b + L.tmpb);
    s.dhi = (int*)(b + L.dhi); s.dlo = (int*)(b + L.dlo); s.left = (int*)(b + L.left);
    s.dbox = (float*)(b + L.dbox); s.dtlbr = (double*)(b + L.dtlbr); s.ttlbr = (double*)(b + L.ttlbr);
    s.cost = (double*)(b + L.cost); s.lapws = (void*)(b + L.lapws);
    s.xrow = (int*)(b + L.xrow); s.ycol = (int*)(b + L.ycol);
    return s;
}

// ---- the index lists in workgroup memory for the length of a launch -------------------------------------------------------------------------
// A frame step is ~60 barrier-separated passes over short integer lists (tracked / lost / per-frame work lists, marks, assignment vectors); every barrier
// waits for the pass's global stores to be acknowledged (~1-2 us beside a running detector), which is most of the step's time at 80 objects.  A kernel that
// runs many frames of one tracker (k_tracker_step_frames) therefore keeps those lists in LDS: y7t_arena_load copies them in once, the steps run on a
// Y7TTrk whose list pointers lead into the arena (same code, same order of operations: results are bit-identical), y7t_arena_store copies them back.
// Between launches the state blob in global memory is the only truth (the host-side STrack accessors read it).
#define Y7T_ARENA_T_LISTS 13      // tracked, lost, mark, pool, unconf, rem, actl, refind, lostn, removedl, tmpa, tmpb, xrow
#define Y7T_ARENA_D_LISTS 3       // dhi, dlo, left
Y7T_HD size_t y7t_arena_bytes(int cap_t, int cap_d) {
    return ((size_t)Y7T_ARENA_T_LISTS * cap_t + (size_t)Y7T_ARENA_D_LISTS * cap_d + (size_t)(cap_t > cap_d ? cap_t : cap_d)) * sizeof(int);
}

// the arrays that move, as (pointer-to-member, length) pairs in arena order
#define Y7T_ARENA_FOREACH(X, T, D, M) \
    X(tracked, T) X(lost, T) X(mark, T) X(pool, T) X(unconf, T) X(rem, T) X(actl, T) X(refind, T) X(lostn, T) X(removedl, T) X(tmpa, T) X(tmpb, T) X(xrow, T) \
    X(dhi, D) X(dlo, D) X(left, D) X(ycol, M)

Y7T_FN bool y7t_arena_fits(const Y7TExec& ex, int cap_t, int cap_d) { return ex.arena && ex.arena_bytes >= y7t_arena_bytes(cap_t, cap_d); }

// bind a state blob; with an arena in `ex` the index lists point into it (the caller has loaded it)
Y7T_FN Y7TTrk y7t_trk_bind_ex(const Y7TExec& ex, void* blob, int cap_t, int cap_d) {
    Y7TTrk s = y7t_trk_bind(blob, cap_t, cap_d);
    if (!y7t_arena_fits(ex, cap_t, cap_d)) return s;
    int* a = (int*)ex.arena;
    const int T = cap_t, D = cap_d, M = cap_t > cap_d ? cap_t : cap_d;
#define Y7T_X(f, n) s.f = a; a += (n);
    Y7T_ARENA_FOREACH(Y7T_X, T, D, M)
#undef Y7T_X
    return s;
}

Y7T_FN void y7t_arena_copy(const Y7TExec& ex, void* blob, bool load) {
    Y7TTrkHdr* h = (Y7TTrkHdr*)blob;
    const int T = h->cfg.cap_t, D = h->cfg.cap_d, M = T > D ? T : D;
    if (!y7t_arena_fits(ex, T, D)) return;
    const Y7TTrk g = y7t_trk_bind(blob, T, D), l = y7t_trk_bind_ex(ex, blob, T, D);
    y7t_sync(ex);
#define Y7T_X(f, n) for (int k = ex.tid; k < (n); k += ex.nt) { if (load) l.f[k] = g.f[k]; else g.f[k] = l.f[k]; }
    Y7T_ARENA_FOREACH(Y7T_X, T, D, M)
#undef Y7T_X
    y7t_sync(ex);
}
Y7T_FN void y7t_arena_load(const Y7TExec& ex, void* blob) { y7t_arena_copy(ex, blob, true); }
Y7T_FN void y7t_arena_store(const Y7TExec& ex, void* blob) { y7t_arena_copy(ex, blob, false); }

enum { Y7T_ERR_CAP_T = 1, Y7T_ERR_CAP_D = 2, Y7T_ERR_OUT = 4, Y7T_ERR_KIND = 8 /* a DeepSORT pool stepped with detections by the plain step */ };

#if !Y7T_DEVICE
static inline long long clock64() { return 0; }
#endif
#if Y7T_DEVICE
#define Y7T_PROF(h, i) do { if (ex.tid == 0) (h)->prof[i] = clock64(); } while (0)
#else
#define Y7T_PROF(h, i) do { } while (0)
#endif

// ---- STrack geometry -----------------------------------------------------------------------
// tlwh of a pool track from its Kalman mean, basetrack.py:183-211 (xyah: w = a*h; tl = c - wh/2).
// f32m: the mean still has the float32 dtype it gets from initiate under numpy>=2, so the
// reference evaluates this property in float32.
Y7T_FN void y7t_track_tlwh(int kf, const double* m, int f32m, double* o) {
    if (f32m) {
        float x = (float)m[0], y = (float)m[1], w = (float)m[2], h = (float)m[3];
        if (kf != Y7T_KF_XYWH) w = w * h;
        x = x - w / 2; y = y - h / 2;
        o[0] = x; o[1] = y; o[2] = w; o[3] = h;
    } else {
        double x = m[0], y = m[1], w = m[2], h = m[3];
        if (kf != Y7T_KF_XYWH) w = w * h;
        x = x - w / 2; y = y - h / 2;
        o[0] = x; o[1] = y; o[2] = w; o[3] = h;
    }
}

Y7T_FN void y7t_track_tlbr(int kf, const double* m, int f32m, double* o) {
    y7t_track_tlwh(kf, m, f32m, o);
    if (f32m) { o[2] = (float)o[2] + (float)o[0]; o[3] = (float)o[3] + (float)o[1]; }
    else { o[2] = o[2] + o[0]; o[3] = o[3] + o[1]; }
}

// measurement from a detection's float32 tlwh: tlwh2xyah (basetrack.py:122-129) or tlwh2xywh
// with its floor division (basetrack.py:144-150); float32 arithmetic like numpy on a float32 array
Y7T_FN void y7t_meas(int kf, const float* tlwh, double* z) {
    float x = tlwh[0], y = tlwh[1], w = tlwh[2], h = tlwh[3];
    if (kf == Y7T_KF_XYWH) { x = x + floorf(w / 2); y = y + floorf(h / 2); }
    else { x = x + w / 2; y = y + h / 2; w = w / h; }
    z[0] = x; z[1] = y; z[2] = w; z[3] = h;
}

// ---- building blocks -----------------------------------------------------------------------
// cost[i*ld + j] = 1 - IoU+1(track tlbr i, det tlbr j)
Y7T_FN void y7t_cost_matrix(const Y7TExec& ex, const double* a, int na, const double* b, int nb, double* cost, int ld) {
    // a lane owns a column (its detection box stays in registers), a wave a residue class of rows (the track box is wave-uniform:
    // scalar loads); consecutive lanes store consecutive doubles.  No integer division, one box load per element instead of two.
    const int lanes = ex.nt < 64 ? ex.nt : 64, nw = ex.nt / lanes, wave = ex.tid / lanes, lane = ex.tid - wave * lanes;
    for (int j = lane; j < nb; j += lanes) {
        const double q[4] = {b[4 * j], b[4 * j + 1], b[4 * j + 2], b[4 * j + 3]};
        for (int i = wave; i < na; i += nw) cost[(size_t)i * ld + j] = y7t_iou_dist(a + 4 * i, q);
    }
    y7t_sync(ex);
}

// ---------------------------------------------------------------------------------------------
// Sparse form of y7t_assoc for crowded frames (hundreds of tracks x hundreds of detections).  An IoU cost matrix is sparse: a track
// overlaps a handful of detections, and only edges with cost <= thresh ("candidates") can be part of the optimum of
// lap.lapjv(extend_cost=True, cost_limit=thresh) -- a dearer pair is beaten by leaving both unmatched, and in the shortest-augmenting-path
// search the null column is always reached before a non-candidate column (its distance through any row is smaller), so such columns
// are never scanned and never get a price.  Therefore:
//   1. the cost pass appends candidate edges to per-row lists instead of storing na x nb doubles (rows sorted by column afterwards: the
//      atomics make the append order arbitrary);
//   2. rows without candidates are unmatched, isolated pairs are matched (as in the dense census);
//   3. the rest splits into the connected components of the candidate graph (label propagation over the edge lists), and the assignment
//      problem separates over components: ONE LANE solves one component with a serial sparse Dijkstra / augment loop, all components
//      in parallel.  (The dense solver made ~10 block-wide passes over all columns per unsettled row: 65 % of a 500-object frame.)
// Same optimum as the dense solver; identical assignment whenever it is unique.  Falls back (returns false) when a row has more than
// Y7T_MAXC candidates or the scratch does not fit.
// ---------------------------------------------------------------------------------------------
#define Y7T_MAXC 24
#ifndef Y7T_SPARSE_MIN
#define Y7T_SPARSE_MIN 4096      // na * nb from which the sparse path is taken (64 x 64)
#endif

// colctx(j) / rowctx(i): whatever of column j / row i the cost needs (y7t_pairs); cost(rl, r, cj) -> double, the row's values taken with y7t_row_at(rl.., r)
// Returns 1: solved (s.xrow / s.ycol written); 0: not applicable (candidate overflow, a pair exactly at the limit) -> dense path; 2: two candidate edges of one
// connected component cost EXACTLY the same (costs are float32 distances or IoUs of integer boxes: it happens) -> the optimum may not be unique and the
// caller solves the dense problem with lapjv.cpp run literally (y7t_lap_solve_literal).
// The row stride MC of the candidate lists is chosen at run time (y7t_assoc_sparse_fn below) so that the lists fit in the fast scratch next to the work
// arrays -- 24 entries per row keep a 500-object frame's lists (121 KB) in global memory, where every step of the per-component solves is a dependent L2
// round trip; 16 per row (80 KB) fit in LDS and the largest row of that scene has 9 candidates.  A row that overflows the shorter stride repeats the
// association with the full one.  (Round 3, measured: the 500-object frame step 932 -> 902 us, 80 objects unchanged; profiles/r03_tracker_phases.txt)
#ifndef Y7T_NEXT_STAT
#define Y7T_NEXT_STAT(k) do { } while (0)
#endif
template <class ColFn, class RowFn, class CostFn, class Geo>
Y7T_FN int y7t_assoc_sparse_try(const Y7TExec& ex, const Y7TTrk& s, int na, int nb, double thresh, ColFn colctx, RowFn rowctx, CostFn cost, const int MC, Geo geo) {
    const int tid = ex.tid, nt = ex.nt;
#ifdef Y7T_ALWAYS_LITERAL
    return 2;
#endif
#if Y7T_DEVICE
#define Y7T_SPROF(i) do { if (thresh == 0.9 && tid == 0) s.h->prof[16 + (i)] = clock64(); } while (0)
#else
#define Y7T_SPROF(i) do { } while (0)
#endif
    Y7T_SPROF(0);
    const double thresh_hi = thresh + Y7T_TIE_EPS;      // (a pair exactly at the limit: see y7t_lap_solve_sap; the in-search watch is for <= Y7T_TIE_FULL_N
                                                        //  rows + columns, which never come here: na * nb >= Y7T_SPARSE_MIN)
    // scratch: the work arrays and, when they fit, the candidate lists live in the workgroup's fast scratch (LDS) -- the per-component
    // solves are chains of dependent reads -- otherwise in the (unused) dense cost matrix of the state blob
    const size_t work_bytes = (size_t)(2 * nb + 2) * sizeof(double) + (size_t)(4 * na + 6 * nb + 16) * sizeof(int);
    const size_t list_bytes = (size_t)na * MC * (sizeof(int) + sizeof(double));
    const size_t T = (size_t)s.h->cfg.cap_t, D = (size_t)s.h->cfg.cap_d, blob_bytes = T * (T > D ? T : D) * sizeof(double);
    char* wbase = (char*)s.cost;
    char* lbase = (char*)s.cost + ((work_bytes + 63) & ~(size_t)63);
    if (ex.fast && work_bytes + 64 <= ex.fast_bytes) {
        wbase = ex.fast;
        lbase = (work_bytes + 64 + list_bytes <= ex.fast_bytes) ? ex.fast + ((work_bytes + 63) & ~(size_t)63) : (char*)s.cost;
    }
    if (((work_bytes + 63) & ~(size_t)63) + list_bytes > blob_bytes) return 0;
    double* v = (double*)wbase;                               // [nb] column prices
    double* dd = v + nb;                                      // [nb] tentative distances
    int* rowcnt = (int*)(dd + nb + 2);                        // [na]
    int* rowlab = rowcnt + na;                                // [na]
    int* x = rowlab + na;                                     // [na]
    int* csz = x + na;                                        // [na] 1: the component led by row i has two or more rows (set in step 3)
    int* colcnt = csz + na;                                   // [nb]; behind the forced decisions (step 2): the column lists of the components larger than a wave (step 4a)
    int* collab = colcnt + nb;                                // [nb]
    int* y = collab + nb;                                     // [nb]
    int* pred = y + nb;                                       // [nb]
    int* st = pred + nb;                                      // [nb] 0 untouched, 1 touched (in the frontier), 2 scanned
    int* nextcol = st + nb;                                   // [nb] linked list of the touched columns
    int* flag = nextcol + nb;                                 // [6] overflow / at-limit pair, changed, duplicate cost inside a component, the next component of step 4a, "a row is left for steps 3 and 4", how much of colcnt[] the column lists of the components larger than a wave have taken
    double* ccost = (double*)lbase;                           // [na][MAXC] candidate costs
    int* ccol = (int*)(ccost + (size_t)na * MC);        // [na][MAXC] candidate columns
    for (int i = tid; i < na; i += nt) { rowcnt[i] = 0; x[i] = -1; rowlab[i] = i; csz[i] = 0; }
    for (int j = tid; j < nb; j += nt) { colcnt[j] = 0; y[j] = -1; v[j] = 0.0; st[j] = 0; collab[j] = 0x7fffffff; }
    if (tid == 0) { flag[0] = 0; flag[1] = 0; flag[2] = 0; flag[3] = 0; flag[4] = 0; flag[5] = 0; }
    y7t_sync(ex);
    // ---- 1. cost pass (y7t_pairs: a lane per column, a wave per row residue, the rows' contexts handed out through the scalar registers).  Round 6: with a geometry
    // (IoU costs: Y7TBoxGeo) and more than two waves of columns, the columns are walked in the order of 64 bins of their left edge (y7t_bin_perm), so a wave's 64 columns are a strip of
    // the image and a row outside the strip is skipped for all 64 at once (y7t_pairs: group rejection; the keys borrow dd[], the order nextcol[] -- both idle until step 4) ----
    const int* colperm = nullptr;
#if Y7T_DEVICE
    if (Geo::on && nb > 128 && nt >= 64) {
        for (int j = tid; j < nb; j += nt) dd[j] = geo.key(colctx(j));
        y7t_sync(ex);
        y7t_bin_perm(ex, nb, dd, nextcol, pred);            // (pred[]: nb > 128 ints, idle until step 4)
        colperm = nextcol;
    }
#endif
    y7t_pairs(ex, na, nb, colctx, rowctx, [&](int i, int j, const auto& rl, int r, const auto& cj) {
        const double c = cost(rl, r, cj);
        if (c <= thresh_hi) {                                 // (one compare on the common path: most pairs do not overlap at all)
            if (c >= thresh - Y7T_TIE_EPS) { flag[0] = 1; Y7T_TIE_REASON(4); }      // exactly at the limit: the optimum is not unique -> dense path, lapjv's own order
            if (c <= thresh) {
                const int k = Y7T_FETCH_ADD(rowcnt + i, 1);
                if (k < MC) { ccol[(size_t)i * MC + k] = j; ccost[(size_t)i * MC + k] = c; }
                else flag[0] = 1;
                Y7T_FETCH_ADD(colcnt + j, 1);
            }
        }
    }, colperm, geo);
    y7t_sync(ex);
    Y7T_SPROF(1);
    if (flag[0]) return 0;
    for (int i = tid; i < na; i += nt) {                      // sort each row's candidates by column (insertion sort, <= MAXC entries)
        int* cc = ccol + (size_t)i * MC;
        double* cw = ccost + (size_t)i * MC;
        const int n = rowcnt[i];
        for (int a = 1; a < n; ++a) {
            const int cj = cc[a]; const double cv = cw[a];
            int b = a - 1;
            while (b >= 0 && cc[b] > cj) { cc[b + 1] = cc[b]; cw[b + 1] = cw[b]; --b; }
            cc[b + 1] = cj; cw[b + 1] = cv;
        }
        for (int a = 1; a < n; ++a)                           // tie watch: one row equally far from two columns (identical boxes / appearance vectors)
            for (int b = 0; b < a; ++b) if (cw[a] == cw[b]) { flag[2] = 1; Y7T_TIE_REASON(6); }
    }
    y7t_sync(ex);
    Y7T_SPROF(2);
    // ---- 2. forced decisions ----
    for (int i = tid; i < na; i += nt) {
        if (rowcnt[i] == 0) x[i] = nb;                        // null column
        else if (rowcnt[i] == 1 && colcnt[ccol[(size_t)i * MC]] == 1) {
            const int j = ccol[(size_t)i * MC];
            const double red = ccost[(size_t)i * MC] - thresh;
            if (red < 0.0) { x[i] = j; y[j] = i; v[j] = red; }
        }
        if (x[i] == -1) flag[4] = 1;
    }
    y7t_sync(ex);
    Y7T_SPROF(3);
    // (the levels of DeepSORT's cascade are a dozen small problems per frame, most of them settled by now: steps 3 and 4 are ~10 barriers each)
    const bool rows_left = flag[4] != 0;
    if (rows_left) {
    // ---- 3. components of the candidate graph among the unsettled rows: label = smallest row index ----
    for (int it = 0; it < na + 2; ++it) {
        for (int i = tid; i < na; i += nt)
            if (x[i] == -1)
                for (int k = 0; k < rowcnt[i]; ++k) Y7T_ATOMIC_MIN_I(collab + ccol[(size_t)i * MC + k], rowlab[i]);
        if (tid == 0) flag[1] = 0;
        y7t_sync(ex);
        for (int i = tid; i < na; i += nt)
            if (x[i] == -1) {
                int m = rowlab[i];
                for (int k = 0; k < rowcnt[i]; ++k) { const int l = collab[ccol[(size_t)i * MC + k]]; m = l < m ? l : m; }
                if (m != rowlab[i]) { rowlab[i] = m; flag[1] = 1; csz[m] = 1; }      // (csz: "this leader's component has a second row"; a mark on a row that stops being a leader is never read)
            }
        y7t_sync(ex);
        if (!flag[1]) break;
        y7t_sync(ex);
    }
    Y7T_SPROF(4);
    // ---- 4. the assignment problem separates over components: serial sparse shortest augmenting paths over a component's rows in ascending order ----
    // tie watch: two candidate edges of a component with exactly the same cost (components of up to 8 rows; larger ones are not watched)
    auto tie_watch = [&](int lead) {
        int crow[8], ncr = 0;
        const int maxr = csz[lead] ? 9 : 1;                   // (a single row: no scan to the end of the matrix for its companions)
        for (int a = lead; a < na && ncr < maxr; ++a) if (rowlab[a] == lead) { if (ncr < 8) crow[ncr] = a; ++ncr; }
        if (ncr <= 8) {
            bool dup = false;
            for (int ia = 0; ia < ncr && !dup; ++ia)
                for (int pq = 0; pq < rowcnt[crow[ia]] && !dup; ++pq) {
                    const double ca = ccost[(size_t)crow[ia] * MC + pq];
                    for (int q = pq + 1; q < rowcnt[crow[ia]]; ++q) dup |= (ccost[(size_t)crow[ia] * MC + q] == ca);
                    for (int ib = ia + 1; ib < ncr; ++ib)
                        for (int q = 0; q < rowcnt[crow[ib]]; ++q) dup |= (ccost[(size_t)crow[ib] * MC + q] == ca);
                }
            if (dup) { flag[2] = 1; Y7T_TIE_REASON(5); }
        }
    };
    // one LANE walks a whole component: every step a chain of dependent reads of the work arrays
    auto solve_by_lane = [&](int lead) {
        for (int start = lead, left = csz[lead] ? na : 1; start < na && left > 0; ++start) {
            if (rowlab[start] != lead || x[start] != -1) continue;
            --left;
            // Dijkstra from `start`; the null column lives in registers (every component has its own)
            double d_null = 0.0; int pred_null = start;
            int touched = -1;
            for (int k = 0; k < rowcnt[start]; ++k) {
                const int j = ccol[(size_t)start * MC + k];
                dd[j] = ccost[(size_t)start * MC + k] - thresh - v[j]; pred[j] = start; st[j] = 1; nextcol[j] = touched; touched = j;
            }
            int final_j = -2; double mind = 0.0;
            for (;;) {
                double mv = d_null; int mj = nb;               // frontier minimum, ties to the lowest column index (the null column is index nb)
                for (int j = touched; j >= 0; j = nextcol[j])
                    if (st[j] == 1 && (dd[j] < mv || (dd[j] == mv && j < mj))) { mv = dd[j]; mj = j; }
                mind = mv;
                if (mj == nb || y[mj] < 0) { final_j = mj; break; }
                st[mj] = 2;
                const int i = y[mj];
                double cij = 0.0;
                for (int k = 0; k < rowcnt[i]; ++k) if (ccol[(size_t)i * MC + k] == mj) cij = ccost[(size_t)i * MC + k];
                const double hh = cij - thresh - v[mj] - mind;
                for (int k = 0; k < rowcnt[i]; ++k) {
                    const int j = ccol[(size_t)i * MC + k];
                    if (st[j] == 2) continue;
                    const double cred = ccost[(size_t)i * MC + k] - thresh - v[j] - hh;
                    if (st[j] == 0) { dd[j] = cred; pred[j] = i; st[j] = 1; nextcol[j] = touched; touched = j; }
                    else if (cred < dd[j]) { dd[j] = cred; pred[j] = i; }
                }
                if (-hh < d_null) { d_null = -hh; pred_null = i; }
            }
            for (int j = touched; j >= 0; j = nextcol[j]) { if (st[j] == 2) v[j] += dd[j] - mind; }
            // augment
            {
                int i = -1, j = final_j;
                while (i != start) {
                    i = (j == nb) ? pred_null : pred[j];
                    if (j != nb) y[j] = i;
                    const int t = j;
                    j = x[i];
                    x[i] = t;
                }
            }
            for (int j = touched; j >= 0; ) { const int nx = nextcol[j]; st[j] = 0; j = nx; }
        }
    };
    // ---- 4a. components of two or more rows (and at most 64 rows and columns): ONE WAVE each, the component's state in registers.
    // A 500-object frame has ~200 components, ~80 of them with two or more rows, the largest 10-20.  One lane per component (rounds 2-4) had two costs: the
    // largest component's walk (rows x searches x candidates of dependent LDS reads, ~2000 clocks per search step), and -- the larger one, measured in round 5 --
    // the lanes of a wave walking 64 different components in lockstep: every nested loop of the tie watch and of the searches runs for the longest trip count
    // among them (639 kcycles of per-component solves with only the components of more than 8 rows on waves; profiles/r05_tracker_association.txt).  Round 4
    // gave a large component a wave but kept its state in LDS: a search step then costs a wave reduction plus the same four dependent round trips (no gain,
    // profiles/r04_tracker_coop_experiment.txt).  Here lane l owns the component's l-th column (price, distance, mark, predecessor, the row matched to it) and
    // its l-th row (the column matched to it), everything in slot numbers; a search step is one DPP minimum + a ballot (ties to the lowest lane = the lowest
    // column, as the lane's scan), one read of the scanned row's candidate list, which the lanes hand to the owners of those columns through the scalar
    // registers, and the relaxation in registers; control flow is uniform over the wave.  Same arithmetic in the same order as solve_by_lane, so the same prices
    // and the same assignment.  The host build (one thread) runs the same text with 64-element arrays (Y7T_WV_*). ----
#ifndef Y7T_COOP_DIAG
#define Y7T_COOP_DIAG 1      // wave 0's share of step 4 in the header's prof[] (scripts/time_tracker.py)
#endif
    bool coop = false;
    {
#if Y7T_DEVICE
        const int wv_lane = tid & 63;
        coop = nt >= 64 && na <= 4096;
#else
        coop = na <= 4096;
#endif
        if (coop) {
            // the leaders of the components of two or more rows: every wave finds them for itself -- a ballot per 64 rows, the masks a lane each -- so that there is
            // no list to build and no barrier to wait at (the levels of DeepSORT's cascade are a dozen small problems per frame)
            int lm_lo[Y7T_WVN], lm_hi[Y7T_WVN];
            int nbig = 0;
            const int nch = (na + 63) >> 6;
#if Y7T_DEVICE
            lm_lo[0] = 0; lm_hi[0] = 0;
            for (int c = 0; c < nch; ++c) {
                const int i = c * 64 + wv_lane;
                const unsigned long long b = __ballot(i < na && rowlab[i] == i && csz[i] != 0);      // (NOT x[i] == -1: a faster wave is already writing the x of the components it has solved, and every wave must see the same leader list behind the shared ticket counter; rowlab / csz are final since step 3's last barrier, and a leader with csz set is unsettled by construction)
                if (wv_lane == c) { lm_lo[0] = (int)(unsigned)b; lm_hi[0] = (int)(unsigned)(b >> 32); }
                nbig += __popcll(b);
            }
#else
            for (int c = 0; c < 64; ++c) { lm_lo[c] = 0; lm_hi[c] = 0; }
            for (int i = 0; i < na; ++i)
                if (rowlab[i] == i && csz[i] != 0) { if ((i & 63) < 32) lm_lo[i >> 6] |= (int)(1u << (i & 31)); else lm_hi[i >> 6] |= (int)(1u << (i & 31)); ++nbig; }
#endif
            for (;;) {                                          // the waves take the components off a counter: the largest one (10-20 rows) costs as much as ten small ones
                int bi;
#if Y7T_DEVICE
                bi = 0;
                if (wv_lane == 0) bi = Y7T_FETCH_ADD(flag + 3, 1);
                bi = __builtin_amdgcn_readfirstlane(bi);
#else
                bi = flag[3]++;
#endif
                if (bi >= nbig) break;
                int lead = -1;
                for (int c = 0, k = bi; c < nch; ++c) {
                    unsigned long long m = (unsigned long long)(unsigned)Y7T_WV_AT_I(lm_lo, c) | ((unsigned long long)(unsigned)Y7T_WV_AT_I(lm_hi, c) << 32);
                    const int n = y7t_popc64(m);
                    if (k < n) { for (int t = 0; t < k; ++t) m &= m - 1; lead = c * 64 + y7t_ctz64(m); break; }
                    k -= n;
                }
                // slots: lane l <- the component's l-th column and l-th row (ascending)
                int myj[Y7T_WVN], rid[Y7T_WVN];
                int ncl = 0, nrw = 0;
#if Y7T_DEVICE
                {
                    myj[0] = -1; rid[0] = -1;
                    for (int base = 0; base < nb; base += 64) {
                        const int j = base + wv_lane;
                        const unsigned long long b = __ballot(j < nb && collab[j] == lead);
                        const int cnt = __popcll(b), k = wv_lane - ncl;
                        if (k >= 0 && k < cnt) { unsigned long long m = b; for (int t = 0; t < k; ++t) m &= m - 1; myj[0] = base + __ffsll((long long)m) - 1; }
                        ncl += cnt;
                    }
                    for (int base = lead; base < na; base += 64) {
                        const int i = base + wv_lane;
                        const unsigned long long b = __ballot(i < na && rowlab[i] == lead && x[i] == -1);
                        const int cnt = __popcll(b), k = wv_lane - nrw;
                        if (k >= 0 && k < cnt) { unsigned long long m = b; for (int t = 0; t < k; ++t) m &= m - 1; rid[0] = base + __ffsll((long long)m) - 1; }
                        nrw += cnt;
                    }
                }
#else
                for (int l = 0; l < 64; ++l) { myj[l] = -1; rid[l] = -1; }
                for (int j = 0; j < nb; ++j) if (collab[j] == lead) { if (ncl < 64) myj[ncl] = j; ++ncl; }
                for (int i = lead; i < na; ++i) if (rowlab[i] == lead && x[i] == -1) { if (nrw < 64) rid[nrw] = i; ++nrw; }
#endif
                if (ncl > 64 || nrw > 64) {
                    // ---- more rows or columns than a wave has lanes (crowds: 250+ objects on a 640-px frame).  The same wave solves the component with its state in
                    // the work arrays instead of in registers -- v / dd / st / pred / x / y by real row and column index, exactly solve_by_lane's variables and
                    // arithmetic -- and the lanes share what that walk does serially: the frontier minimum (a stride over the component's columns, DPP minimum of
                    // the distance, then of the column index among the lanes at that distance: ties to the lowest column), the scanned row's candidates a lane
                    // each (they are distinct columns: no two lanes touch one word).  Any size; control flow uniform over the wave.  (History: "lane 0 walks it,
                    // then `continue`" never returned on the device -- lanes that skip a divergent region in front of a back edge went round without lane 0 --;
                    // round 5 closed with "return 3, the caller solves the whole problem densely", which ran such frames at round 2's speed.) ----
                    Y7T_NEXT_STAT(3);
                    int off;                                    // the component's columns, ascending, in a range of colcnt[] (free since step 2; the components' column sets are disjoint)
#if Y7T_DEVICE
                    off = 0;
                    if (wv_lane == 0) off = Y7T_FETCH_ADD(flag + 5, ncl);
                    off = __builtin_amdgcn_readfirstlane(off);
                    int* cl = colcnt + off;
                    for (int base = 0, n0 = 0; base < nb; base += 64) {
                        const int j = base + wv_lane;
                        const bool mine = j < nb && collab[j] == lead;
                        const unsigned long long b = __ballot(mine);
                        if (mine) cl[n0 + __popcll(b & ((1ull << wv_lane) - 1ull))] = j;
                        n0 += __popcll(b);
                    }
#else
                    off = flag[5]; flag[5] += ncl;
                    int* cl = colcnt + off;
                    for (int j = 0, n0 = 0; j < nb; ++j) if (collab[j] == lead) cl[n0++] = j;
#endif
                    Y7T_WV_FENCE();
                    for (int rbase = lead; rbase < na; rbase += 64) {      // the component's rows in ascending order, 64 at a time (all unsettled: a search settles exactly its start row)
                        unsigned long long rm;
#if Y7T_DEVICE
                        rm = __ballot(rbase + wv_lane < na && rowlab[rbase + wv_lane] == lead);
#else
                        rm = 0ull;
                        for (int l = 0; l < 64 && rbase + l < na; ++l) if (rowlab[rbase + l] == lead) rm |= 1ull << l;
#endif
                        while (rm) {
                            const int start = rbase + y7t_ctz64(rm);
                            rm &= rm - 1ull;
                            double d_null = 0.0; int pred_null = start;
                            {
                                const int n = rowcnt[start];
                                Y7T_WV_STRIDE(k, n) {
                                    const int j = ccol[(size_t)start * MC + k];
                                    dd[j] = ccost[(size_t)start * MC + k] - thresh - v[j]; pred[j] = start; st[j] = 1;
                                }
                            }
                            Y7T_WV_FENCE();
                            int final_j = -2; double mind = 0.0;
                            for (int guard = 0; ; ++guard) {
                                if (guard > ncl + 2) { flag[2] = 1; final_j = nb; pred_null = start; break; }      // (a search scans every column at most once; never seen -- the caller would re-solve the problem densely)
                                double mv = HUGE_VAL; int mj = 0x7fffffff;
                                Y7T_WV_STRIDE(p, ncl) {                     // (a lane's columns ascend and the compare is strict: its lowest column among equals)
                                    const int j = cl[p];
                                    if (st[j] == 1) { const double t = dd[j]; if (t < mv) { mv = t; mj = j; } }
                                }
#if Y7T_DEVICE
                                {
                                    const double m = y7t_wave_min_d(mv);
                                    mj = y7t_wave_min_i(mv == m ? mj : 0x7fffffff);
                                    mv = m;
                                }
#endif
                                if (mj == 0x7fffffff || d_null < mv) { mv = d_null; mj = nb; }      // the null column (index nb) loses a tie against a real one
                                mind = mv;
                                if (mj == nb) { final_j = nb; break; }
                                const int i = y[mj];
                                if (i < 0) { final_j = mj; break; }
                                const int n = rowcnt[i];
                                double cij = 0.0;
#if Y7T_DEVICE
                                int kc = -1; double kv = 0.0;
                                if (wv_lane < n) { kc = ccol[(size_t)i * MC + wv_lane]; kv = ccost[(size_t)i * MC + wv_lane]; }
                                {
                                    const unsigned long long bm = __ballot(kc == mj);      // (the matched edge is a candidate of its row)
                                    cij = y7t_readlane_d(kv, bm ? y7t_ctz64(bm) : 0);
                                }
#else
                                for (int k = 0; k < n; ++k) if (ccol[(size_t)i * MC + k] == mj) cij = ccost[(size_t)i * MC + k];
#endif
                                const double hh = cij - thresh - v[mj] - mind;
                                Y7T_WV_STRIDE(k, n) {
                                    const int j = ccol[(size_t)i * MC + k];
                                    if (j == mj) st[j] = 2;
                                    else if (st[j] != 2) {
                                        const double cred = ccost[(size_t)i * MC + k] - thresh - v[j] - hh;
                                        if (st[j] == 0) { dd[j] = cred; pred[j] = i; st[j] = 1; }
                                        else if (cred < dd[j]) { dd[j] = cred; pred[j] = i; }
                                    }
                                }
                                if (-hh < d_null) { d_null = -hh; pred_null = i; }
                                Y7T_WV_FENCE();
                            }
                            Y7T_WV_STRIDE(p, ncl) { const int j = cl[p]; if (st[j] == 2) v[j] += dd[j] - mind; st[j] = 0; }
                            Y7T_WV_FENCE();
                            {   // augment: every lane walks the same path and writes the same words
                                int i = -1, j = final_j;
                                for (int guard = 0; i != start && guard < na + 2; ++guard) {
                                    i = (j == nb) ? pred_null : pred[j];
                                    if (j != nb) y[j] = i;
                                    const int t = j;
                                    j = x[i];
                                    x[i] = t;
                                }
                            }
                            Y7T_WV_FENCE();
                        }
                    }
                    continue;
                }
                Y7T_NEXT_STAT(2);                               // (host build: components solved on this path)
                int xs[Y7T_WVN], rc[Y7T_WVN];                   // my row: the column slot matched to it (-1 none, 64 the null column), its number of candidates
                Y7T_WV_EACH(l) { Y7T_WV(xs, l) = -1; Y7T_WV(rc, l) = Y7T_WV(rid, l) >= 0 ? rowcnt[Y7T_WV(rid, l)] : 0; }
                if (nrw <= 8) {                                 // tie watch (see tie_watch): the component's candidate edges a lane each, every edge's cost against the lanes above it
                    int er[Y7T_WVN]; double ev[Y7T_WVN];
                    int ne = 0;
                    Y7T_WV_EACH(l) { Y7T_WV(er, l) = -1; Y7T_WV(ev, l) = 0.0; }
                    for (int r = 0; r < nrw; ++r) {
                        const int i = Y7T_WV_AT_I(rid, r), n = Y7T_WV_AT_I(rc, r);
                        Y7T_WV_EACH(l) { const int k = l - ne; if (k >= 0 && k < n) { Y7T_WV(er, l) = r; Y7T_WV(ev, l) = ccost[(size_t)i * MC + k]; } }
                        ne += n;
                    }
                    if (ne <= 64) {
                        bool dup = false;
                        for (int e = 0; e + 1 < ne && !dup; ++e) {
                            const double c = Y7T_WV_AT_D(ev, e);
                            unsigned long long bm;
                            Y7T_WV_BALLOT(bm, l, l > e && Y7T_WV(er, l) >= 0 && Y7T_WV(ev, l) == c);
                            dup = bm != 0ull;
                        }
                        if (dup) { flag[2] = 1; Y7T_TIE_REASON(5); }
                    } else {
#if Y7T_DEVICE
                        if (wv_lane == 0)
#endif
                        tie_watch(lead);
                    }
                }
                double vj[Y7T_WVN], dj[Y7T_WVN], cc[Y7T_WVN];     // my column: price, tentative distance; the scanned row's cost to it
                int yr[Y7T_WVN], stj[Y7T_WVN], pr[Y7T_WVN], has[Y7T_WVN];      // my column: the row slot matched to it, mark, predecessor row slot, "the scanned row has an edge to it"
                Y7T_WV_EACH(l) {
                    Y7T_WV(vj, l) = Y7T_WV(myj, l) >= 0 ? v[Y7T_WV(myj, l)] : 0.0;
                    Y7T_WV(yr, l) = -1;
                    Y7T_WV(dj, l) = 0.0; Y7T_WV(cc, l) = 0.0; Y7T_WV(stj, l) = 0; Y7T_WV(pr, l) = -1; Y7T_WV(has, l) = 0;
                }
                // the candidates of row slot `rs` to the lanes that own their columns: cc / has
                auto hand_out = [&](int rs) {
                    const int i = Y7T_WV_AT_I(rid, rs), n = Y7T_WV_AT_I(rc, rs);
#if Y7T_DEVICE
                    int kc = 0; double kv = 0.0;
                    if (wv_lane < n) { kc = ccol[(size_t)i * MC + wv_lane]; kv = ccost[(size_t)i * MC + wv_lane]; }
                    has[0] = 0;
                    for (int k = 0; k < n; ++k) {
                        const int cj = __builtin_amdgcn_readlane(kc, k);
                        const double cv = y7t_readlane_d(kv, k);
                        if (myj[0] == cj) { cc[0] = cv; has[0] = 1; }
                    }
#else
                    for (int l = 0; l < 64; ++l) {
                        has[l] = 0;
                        for (int k = 0; k < n; ++k) if (ccol[(size_t)i * MC + k] == myj[l]) { cc[l] = ccost[(size_t)i * MC + k]; has[l] = 1; }
                    }
#endif
                };
#if Y7T_DEVICE && Y7T_COOP_DIAG
                if (thresh == 0.9 && tid == 0) { s.h->prof[23] = (long long)nbig | ((long long)nrw << 16) | ((long long)ncl << 24) | ((long long)(bi + 1) << 32); s.h->prof[29] = 0; }
#endif
                for (int start = 0; start < nrw; ++start) {    // (every row of the component is unsettled here, and a search settles exactly its start row)
                    double d_null = 0.0; int pred_null = start;
                    hand_out(start);
                    Y7T_WV_EACH(l) {
                        Y7T_WV(stj, l) = 0;
                        if (Y7T_WV(has, l)) { Y7T_WV(dj, l) = Y7T_WV(cc, l) - thresh - Y7T_WV(vj, l); Y7T_WV(pr, l) = start; Y7T_WV(stj, l) = 1; }
                    }
                    int final_s = -2; double mind = 0.0;
                    for (int guard = 0; ; ++guard) {
                        double mv; unsigned long long bm;
                        if (guard > 66) { flag[2] = 1; final_s = 64; pred_null = start; break; }      // (a search scans every column at most once; never seen -- the caller would re-solve the problem densely)
                        Y7T_WV_MIN_D(mv, l, Y7T_WV(stj, l) == 1 ? Y7T_WV(dj, l) : HUGE_VAL);
                        Y7T_WV_BALLOT(bm, l, Y7T_WV(stj, l) == 1 && Y7T_WV(dj, l) == mv);
                        int ms = bm ? y7t_ctz64(bm) : 64;           // ties to the lowest column; the null column (slot 64) loses a tie against a real one
                        if (!bm || d_null < mv) { mv = d_null; ms = 64; }
                        mind = mv;
                        if (ms == 64) { final_s = 64; break; }
                        const int is = Y7T_WV_AT_I(yr, ms);
                        if (is < 0) { final_s = ms; break; }
                        Y7T_WV_SET(stj, ms, 2);
                        hand_out(is);
                        const double hh = Y7T_WV_AT_D(cc, ms) - thresh - Y7T_WV_AT_D(vj, ms) - mind;
                        Y7T_WV_EACH(l) {
                            if (Y7T_WV(has, l) && Y7T_WV(stj, l) != 2) {
                                const double cred = Y7T_WV(cc, l) - thresh - Y7T_WV(vj, l) - hh;
                                if (Y7T_WV(stj, l) == 0) { Y7T_WV(dj, l) = cred; Y7T_WV(pr, l) = is; Y7T_WV(stj, l) = 1; }
                                else if (cred < Y7T_WV(dj, l)) { Y7T_WV(dj, l) = cred; Y7T_WV(pr, l) = is; }
                            }
                        }
                        if (-hh < d_null) { d_null = -hh; pred_null = is; }
                    }
                    Y7T_WV_EACH(l) { if (Y7T_WV(stj, l) == 2) Y7T_WV(vj, l) += Y7T_WV(dj, l) - mind; }
#if Y7T_DEVICE && Y7T_COOP_DIAG
                    if (thresh == 0.9 && tid == 0) { unsigned long long bs; Y7T_WV_BALLOT(bs, l, Y7T_WV(stj, l) == 2); s.h->prof[29] += __popcll(bs) + 1; }      // diagnostics: search steps of wave 0's last component
#endif
                    {   // augment (slots; uniform scalars)
                        int i = -1, j = final_s;
                        for (int guard = 0; i != start && guard < 66; ++guard) {
                            i = (j == 64) ? pred_null : Y7T_WV_AT_I(pr, j);
                            if (j != 64) Y7T_WV_SET(yr, j, i);
                            const int t = j;
                            j = Y7T_WV_AT_I(xs, i);
                            Y7T_WV_SET(xs, i, t);
                        }
                    }
                }
                Y7T_WV_EACH(l) {                               // slots back to indices
                    const int r = Y7T_WV(yr, l);
                    const int ri = Y7T_WV_GATHER_I(rid, r < 0 ? 0 : r);      // (the gather is a ds_bpermute: outside the branch, or a lane without a column -- more rows than columns -- is an inactive source and hands out 0)
                    if (Y7T_WV(myj, l) >= 0) {
                        y[Y7T_WV(myj, l)] = r < 0 ? -1 : ri;
                        v[Y7T_WV(myj, l)] = Y7T_WV(vj, l);
                    }
                }
                Y7T_WV_EACH(l) {
                    const int c = Y7T_WV(xs, l);
                    const int cj = Y7T_WV_GATHER_I(myj, (c < 0 || c >= 64) ? 0 : c);
                    if (Y7T_WV(rid, l) >= 0) x[Y7T_WV(rid, l)] = c == 64 ? nb : cj;
                }
            }
        }
#if Y7T_DEVICE && Y7T_COOP_DIAG
        if (thresh == 0.9 && tid == 0) { s.h->prof[22] = clock64(); if (!coop) s.h->prof[23] = 0; }      // diagnostics: when wave 0 was done with its components
#endif
    }
    // ---- 4b. one lane per remaining component (single rows; every component when the workgroup is less than a wave or the matrix has more than 4096 rows) ----
    for (int lead = tid; lead < na; lead += nt) {
        if (x[lead] != -1 || rowlab[lead] != lead) continue;
        if (coop && csz[lead]) continue;
        tie_watch(lead);
        solve_by_lane(lead);
    }
    y7t_sync(ex);
    } else Y7T_SPROF(4);
    Y7T_SPROF(5);
    if (flag[2]) return 2;
    for (int i = tid; i < na; i += nt) s.xrow[i] = (x[i] >= nb || x[i] < 0) ? -1 : x[i];
    for (int j = tid; j < nb; j += nt) s.ycol[j] = y[j];
    y7t_sync(ex);
    return 1;
}

template <class ColFn, class RowFn, class CostFn, class Geo = Y7TNoGeo>
Y7T_FN int y7t_assoc_sparse_fn(const Y7TExec& ex, const Y7TTrk& s, int na, int nb, double thresh, ColFn colctx, RowFn rowctx, CostFn cost, Geo geo = Geo()) {
    const size_t work_bytes = (size_t)(2 * nb + 2) * sizeof(double) + (size_t)(4 * na + 6 * nb + 16) * sizeof(int);
    int mc = Y7T_MAXC;
    if (ex.fast && work_bytes + 64 <= ex.fast_bytes)
        while (mc > 8 && work_bytes + 64 + (size_t)na * mc * (sizeof(int) + sizeof(double)) > ex.fast_bytes) mc -= 4;      // 24, 20, 16, 12, 8
    if (mc < Y7T_MAXC && work_bytes + 64 + (size_t)na * mc * (sizeof(int) + sizeof(double)) > ex.fast_bytes) mc = Y7T_MAXC;      // nothing fits: as before
    if (mc < Y7T_MAXC) Y7T_NEXT_STAT(0);                    // (host build: how often the short stride is used / has to be repeated)
    int r = 0;
    for (int pass = 0; pass < 2; ++pass) {                    // (ONE inlined copy of the solver: the frame step's code is 200 KB as it is)
        r = y7t_assoc_sparse_try(ex, s, na, nb, thresh, colctx, rowctx, cost, mc, geo);
        if (r != 0 || mc == Y7T_MAXC) break;
        Y7T_NEXT_STAT(1); y7t_sync(ex); mc = Y7T_MAXC;
    }
    return r;
}

// the IoU instance (matching.iou_distance on the boxes gathered in ttlbr / dtlbr)
Y7T_FN int y7t_assoc_sparse(const Y7TExec& ex, const Y7TTrk& s, int na, int nb, double thresh) {
    return y7t_assoc_sparse_fn(ex, s, na, nb, thresh,
                               [&](int j) { return y7t_box_col(s.dtlbr + 4 * j); },
                               [&](int i) { return y7t_box_row(s.ttlbr + 4 * i); },
                               [&](const Y7TBoxR& rl, int r, const Y7TBoxC& q) { return y7t_box_iou_dist(rl, r, q); }, Y7TBoxGeo());
}

// iou_distance + matching.linear_assignment(cost, thresh) for the boxes gathered in ttlbr[0..na) /
// dtlbr[0..nb)  ->  xrow[na] (det index or -1), ycol[nb] (track index or -1).
// The LAP work arrays and, when it fits, the cost matrix are placed in the workgroup's fast
// scratch (LDS on the device); otherwise they stay in the state blob (HBM/L2).
Y7T_FN void y7t_assoc(const Y7TExec& ex, const Y7TTrk& s, int na, int nb, double thresh) {
    if (na == 0 || nb == 0) {  // empty cost matrix: everything unmatched (matching.py:31-32)
        for (int i = ex.tid; i < na; i += ex.nt) s.xrow[i] = -1;
        for (int j = ex.tid; j < nb; j += ex.nt) s.ycol[j] = -1;
        y7t_sync(ex);
        return;
    }
    int sp = 0;
    if ((long long)na * nb >= Y7T_SPARSE_MIN && (sp = y7t_assoc_sparse(ex, s, na, nb, thresh)) == 1) return;
    Y7TLap L;
    L.nr = na; L.nc = nb; L.ld = nb; L.n = na + nb; L.half = thresh / 2.0;
    L.prof = (thresh == 0.9) ? s.h->prof + 16 : nullptr;   // stamp the first association
    const size_t ws = y7t_al(y7t_lap_ws_bytes(L.n)), cb = (size_t)na * nb * sizeof(double);
    void* lapws = s.lapws;
    double* cost = s.cost;
    size_t off = 0;
    if (ex.fast && ws <= ex.fast_bytes) { lapws = ex.fast; off = ws; }
    if (ex.fast && off + cb <= ex.fast_bytes) cost = (double*)(ex.fast + off);
    if (L.prof && ex.tid == 0) L.prof[8] = clock64();
    y7t_cost_matrix(ex, s.ttlbr, na, s.dtlbr, nb, cost, nb);
    if (L.prof && ex.tid == 0) L.prof[9] = clock64();
    L.c = cost;
    y7t_lap_bind(L, lapws, L.n);
#ifdef Y7T_ASSOC_JV       // experiments: lapjv's own algorithm on the implicit extended matrix (also what the SAP solver is checked against)
    y7t_lap_solve(ex, L);
#else
    if (sp == 2 || y7t_lap_solve_sap(ex, L)) y7t_lap_solve_literal(ex, L);      // ties: lapjv's own order decides
#endif
    if (L.prof && ex.tid == 0) L.prof[10] = clock64();
    for (int i = ex.tid; i < na; i += ex.nt) s.xrow[i] = (L.x[i] >= nb) ? -1 : L.x[i];
    for (int j = ex.tid; j < nb; j += ex.nt) s.ycol[j] = (L.y[j] >= na) ? -1 : L.y[j];
    y7t_sync(ex);
}


// gather tlbr of listed pool tracks
Y7T_FN void y7t_gather_track_tlbr(const Y7TExec& ex, const Y7TTrk& s, const int* list, int n) {
    const int kf = s.h->cfg.kf;
    for (int i = ex.tid; i < n; i += ex.nt) {
        const int sl = list[i];
        y7t_track_tlbr(kf, s.mean + 8 * (size_t)sl, s.f32m[sl], s.ttlbr + 4 * (size_t)i);
    }
}

// gather tlbr of listed detections
Y7T_FN void y7t_gather_det_tlbr(const Y7TExec& ex, const Y7TTrk& s, const int* list, int n) {
    for (int i = ex.tid; i < n; i += ex.nt) {
        const float* b = s.dbox + 4 * (size_t)list[i];
        // STrack.tlbr of a detection: float32 tlwh, then ret[2:] += ret[:2] in float32
        s.dtlbr[4 * (size_t)i + 0] = b[0];
        s.dtlbr[4 * (size_t)i + 1] = b[1];
        s.dtlbr[4 * (size_t)i + 2] = b[2] + b[0];
        s.dtlbr[4 * (size_t)i + 3] = b[3] + b[1];
    }
}

// apply matches of one association (parallel over pairs: distinct tracks), then append to the
// act / refind lists in row order.  mode 0: ByteTrack pool semantics (Tracked->update,
// Lost->re_activate, anything else untouched); 1: SORT (Tracked->update, else re_activate);
// 2: update only.
Y7T_FN void y7t_apply_matches(const Y7TExec& ex, const Y7TTrk& s, const int* tracks, int na, const int* dets,
                              const float* det_rows, int mode, int& n_act, int& n_refind) {
    const int kf = s.h->cfg.kf, frame_id = s.h->frame_id;
    for (int i = ex.tid; i < na; i += ex.nt) {
        const int jd = s.xrow[i];
        if (jd < 0) continue;
        const int sl = tracks[i], dj = dets[jd];
        const int st = s.state[sl];
        int what = 0;  // 1 update, 2 re_activate
        if (mode == 2 || st == Y7T_TRACKED) what = 1;
        else if (mode == 1 || st == Y7T_LOST) what = 2;
        s.tmpa[i] = what;
        if (!what) continue;
        double z[4];
        y7t_meas(kf, s.dbox + 4 * (size_t)dj, z);
        const float sc = det_rows[6 * (size_t)dj + 4];
        // basetrack.py:317-321: only `update` hands the confidence to the NSA filter
        const double conf = (kf == Y7T_KF_NSA && what == 1) ? (double)sc : 0.0;
        y7t_kf_update(kf, s.mean + 8 * (size_t)sl, s.cov + 64 * (size_t)sl, z, conf);
        s.f32m[sl] = 0;
        s.frame[sl] = frame_id;
        s.score[sl] = sc;
        s.state[sl] = Y7T_TRACKED;
        s.act[sl] = 1;
        s.tsu[sl] = 0;
        s.len[sl] = (what == 1) ? s.len[sl] + 1 : 0;
    }
    y7t_sync(ex);
    n_act = y7t_compact(ex, na, [&](int i) { return s.xrow[i] >= 0 && s.tmpa[i] == 1; }, s.tmpb, 0);
    // tmpb holds row indices; translate into slots appended to actl
    {
        const int base = s.h->n_act_last;
        for (int k = ex.tid; k < n_act; k += ex.nt) s.actl[base + k] = tracks[s.tmpb[k]];
        y7t_sync(ex);
        if (ex.tid == 0) s.h->n_act_last = base + n_act;
        y7t_sync(ex);
    }
    n_refind = y7t_compact(ex, na, [&](int i) { return s.xrow[i] >= 0 && s.tmpa[i] == 2; }, s.tmpb, 0);
    {
        const int base = s.h->n_refind_last;
        for (int k = ex.tid; k < n_refind; k += ex.nt) s.refind[base + k] = tracks[s.tmpb[k]];
        y7t_sync(ex);
        if (ex.tid == 0) s.h->n_refind_last = base + n_refind;
        y7t_sync(ex);
    }
}

// STrack.multi_predict over a slot list (basetrack.py:253-271)
Y7T_FN void y7t_multi_predict(const Y7TExec& ex, const Y7TTrk& s, const int* list, int n) {
    const int kf = s.h->cfg.kf;
    for (int i = ex.tid; i < n; i += ex.nt) {
        const int sl = list[i];
        double* m = s.mean + 8 * (size_t)sl;
        if (s.state[sl] != Y7T_TRACKED) m[7] = 0.0;
        y7t_kf_predict(kf, m, s.cov + 64 * (size_t)sl);
        s.f32m[sl] = 0;
        s.tsu[sl] += 1;
    }
    y7t_sync(ex);
}

// the end-of-frame list bookkeeping shared by update / update_without_detection
// (bytetrack.py:185-194).  act/refind/lostn/removedl hold this frame's lists.
Y7T_FN void y7t_finish(const Y7TExec& ex, const Y7TTrk& s, double* out_rows, int out_cap, int* out_count) {
    Y7TTrkHdr* h = s.h;
    const int kf = h->cfg.kf;
    int nt_ = h->n_tracked, nl = h->n_lost;
    const int n_act = h->n_act_last, n_ref = h->n_refind_last, n_lostn = h->n_lostn_last, n_rem = h->n_removed_last;
    // tracked = [t for t in tracked if t.state == Tracked]
    int n1 = y7t_compact(ex, nt_, [&](int i) { return s.state[s.tracked[i]] == Y7T_TRACKED; }, s.tmpa, 0);
    for (int k = ex.tid; k < n1; k += ex.nt) s.tmpb[k] = s.tracked[s.tmpa[k]];
    y7t_sync(ex);
    for (int k = ex.tid; k < h->cfg.cap_t; k += ex.nt) s.mark[k] = 0;
    y7t_sync(ex);
    for (int k = ex.tid; k < n1; k += ex.nt) { s.tracked[k] = s.tmpb[k]; s.mark[s.tmpb[k]] = 1; }
    y7t_sync(ex);
    // joint_stracks(tracked, activated) then (.., refind): append slots not yet present, in order.
    // Every slot occurs at most once in act / refind (a track is matched at most once per frame).
    {
        const int a = y7t_compact(ex, n_act, [&](int k) { return !s.mark[s.actl[k]]; }, s.tmpa, 0);
        for (int k = ex.tid; k < a; k += ex.nt) s.tracked[n1 + k] = s.actl[s.tmpa[k]];
        n1 += a;
        const int r = y7t_compact(ex, n_ref, [&](int k) { return !s.mark[s.refind[k]]; }, s.tmpa, 0);
        for (int k = ex.tid; k < r; k += ex.nt) s.tracked[n1 + k] = s.refind[s.tmpa[k]];
        n1 += r;
        y7t_sync(ex);
        for (int k = ex.tid; k < n1; k += ex.nt) s.mark[s.tracked[k]] = 1;
        if (ex.tid == 0) h->n_tracked = n1;
        y7t_sync(ex);
    }
    // lost = sub_stracks(lost, tracked); lost.extend(lost_new); lost = sub_stracks(lost, self.removed_stracks)
    // (membership in removed_stracks as of BEFORE this frame's removals are appended == inrem flag;
    //  lost_new holds tracks that were Tracked, so it is disjoint from the old lost list)
    int n2 = y7t_compact(ex, nl, [&](int i) { return !s.mark[s.lost[i]] && !s.inrem[s.lost[i]]; }, s.tmpa, 0);
    for (int k = ex.tid; k < n2; k += ex.nt) s.tmpb[k] = s.lost[s.tmpa[k]];
    y7t_sync(ex);
    for (int k = ex.tid; k < n2; k += ex.nt) { s.lost[k] = s.tmpb[k]; s.mark[s.tmpb[k]] |= 2; }
    y7t_sync(ex);
    {
        // sub_stracks builds a dict keyed by id, so a track that is already in `lost` (BoT-SORT re-marks unmatched Lost tracks as
        // lost, botsort.py:432-435) is not appended twice
        const int a = y7t_compact(ex, n_lostn, [&](int k) { return !s.inrem[s.lostn[k]] && !(s.mark[s.lostn[k]] & 2); }, s.tmpa, 0);
        for (int k = ex.tid; k < a; k += ex.nt) s.lost[n2 + k] = s.lostn[s.tmpa[k]];
        n2 += a;
        y7t_sync(ex);
    }
    // self.removed_stracks.extend(removed)
    for (int k = ex.tid; k < n_rem; k += ex.nt) s.inrem[s.removedl[k]] = 1;
    if (ex.tid == 0) { h->n_removed_total += n_rem; h->n_lost = n2; }
    y7t_sync(ex);
    // remove_duplicate_stracks(tracked, lost): pairs with IoU distance < 0.15 (basetrack.py:563-576)
    if (n1 > 0 && n2 > 0) {
        y7t_gather_track_tlbr(ex, s, s.tracked, n1);
        y7t_sync(ex);
        // lost boxes go through dtlbr-sized scratch? use cost tail: keep a second tlbr array in `cost`
        double* lb = s.cost;  // n2*4 doubles, consumed before cost is needed again
        for (int i = ex.tid; i < n2; i += ex.nt) {
            const int sl = s.lost[i];
            y7t_track_tlbr(kf, s.mean + 8 * (size_t)sl, s.f32m[sl], lb + 4 * (size_t)i);
        }
        for (int i = ex.tid; i < n1; i += ex.nt) s.tmpa[i] = 0;
        for (int i = ex.tid; i < n2; i += ex.nt) s.tmpb[i] = 0;
        y7t_sync(ex);
        auto duplicate = [&](int p, int q) {
            const int a = s.tracked[p], b = s.lost[q];
            const int timep = s.frame[a] - s.start[a], timeq = s.frame[b] - s.start[b];
            if (timep > timeq) s.tmpb[q] = 1; else s.tmpa[p] = 1;  // benign same-value races
        };
        const int tot = n1 * n2;
        if (tot < 16384) {                                    // a thread per pair while that keeps every thread at a handful of pairs (80-object frames: 1-2)
            for (int k = ex.tid; k < tot; k += ex.nt) {
                const int p = k / n2, q = k - p * n2;
                if (y7t_iou_dist(s.ttlbr + 4 * (size_t)p, lb + 4 * (size_t)q) < 0.15) duplicate(p, q);
            }
        } else {                                              // crowded frames (500 tracked x a few hundred lost): a lane per lost box, the tracked boxes through the scalar registers
            // (round 6: the lost boxes in ascending order of their left edge, a tracked box outside a wave's strip of 64 skipped at once -- y7t_pairs' group rejection;
            //  keys behind the boxes in `cost`, the order in rem[], which is rewritten below)
            const int* perm = nullptr;
#if Y7T_DEVICE
            if (n2 > 128 && ex.nt >= 64) {
                double* key = lb + 4 * (size_t)n2;
                for (int q = ex.tid; q < n2; q += ex.nt) key[q] = lb[4 * (size_t)q];
                y7t_sync(ex);
                y7t_bin_perm(ex, n2, key, s.rem, s.pool);      // (pool[]: rewritten by the compaction below)
                perm = s.rem;
            }
#endif
            y7t_pairs(ex, n1, n2, [&](int q) { return y7t_box_col(lb + 4 * (size_t)q); }, [&](int p) { return y7t_box_row(s.ttlbr + 4 * (size_t)p); },
                      [&](int p, int q, const Y7TBoxR& rl, int r, const Y7TBoxC& cq) { if (y7t_box_iou_dist(rl, r, cq) < 0.15) duplicate(p, q); }, perm, Y7TBoxGeo());
        }
        y7t_sync(ex);
        const int m1 = y7t_compact(ex, n1, [&](int i) { return !s.tmpa[i]; }, s.pool, 0);
        const int m2 = y7t_compact(ex, n2, [&](int i) { return !s.tmpb[i]; }, s.unconf, 0);
        for (int k = ex.tid; k < m1; k += ex.nt) s.rem[k] = s.tracked[s.pool[k]];
        for (int k = ex.tid; k < m2; k += ex.nt) s.actl[k] = s.lost[s.unconf[k]];
        y7t_sync(ex);
        for (int k = ex.tid; k < m1; k += ex.nt) s.tracked[k] = s.rem[k];
        for (int k = ex.tid; k < m2; k += ex.nt) s.lost[k] = s.actl[k];
        y7t_sync(ex);
        if (ex.tid == 0) { h->n_tracked = m1; h->n_lost = m2; }
        y7t_sync(ex);
        n1 = m1; n2 = m2;
    }
    // recycle slots that are in neither list
    for (int k = ex.tid; k < h->cfg.cap_t; k += ex.nt) s.mark[k] = 0;
    y7t_sync(ex);
    for (int k = ex.tid; k < n1; k += ex.nt) s.mark[s.tracked[k]] = 1;
    for (int k = ex.tid; k < n2; k += ex.nt) s.mark[s.lost[k]] = 1;
    y7t_sync(ex);
    {
        const int nf = y7t_compact(ex, h->cfg.cap_t, [&](int i) { return !s.mark[i]; }, s.freel, 0);
        if (ex.tid == 0) h->n_free = nf;
        for (int k = ex.tid; k < h->cfg.cap_t; k += ex.nt) if (!s.mark[k]) { s.state[k] = Y7T_NEW; s.inrem[k] = 0; s.tid[k] = 0; }
    }
    // return [t for t in tracked if t.is_activated]: rows (id, x, y, w, h, cls, score, slot)
    const int n_out = y7t_compact(ex, n1, [&](int i) { return s.act[s.tracked[i]] != 0; }, s.tmpa, 0);
    if (ex.tid == 0) {
        h->n_out = n_out;
        if (out_count) *out_count = n_out;
        if (n_out > out_cap) h->status |= Y7T_ERR_OUT;
    }
    if (out_rows) {
        for (int k = ex.tid; k < n_out && k < out_cap; k += ex.nt) {
            const int sl = s.tracked[s.tmpa[k]];
            double* o = out_rows + 8 * (size_t)k;
            o[0] = s.tid[sl];
            y7t_track_tlwh(kf, s.mean + 8 * (size_t)sl, s.f32m[sl], o + 1);
            o[5] = s.cls[sl]; o[6] = s.score[sl]; o[7] = sl;
        }
    }
    y7t_sync(ex);
}

// One frame.  dets: n x 6 float32 rows [x1, y1, x2, y2, conf, cls] (n < 0: update_without_detection)
// The body is inlined where it is named: y7t_tracker_step (below) is the CALLED copy every launch of more than 256 threads shares; the <= 256-thread kernels
// (csrc/y7t_tracker.hip) inline it, because a called function does not inherit its kernel's __launch_bounds__ -- compiled for the default 1024 threads a lane has 128
// registers, and what the frame step keeps live around its Kalman updates spills to scratch memory; under __launch_bounds__(256) a lane may use 512 (round 6)
Y7T_FN void y7t_tracker_step_body(const Y7TExec& ex, void* blob, const float* dets, int n, double* out_rows, int out_cap,
                                  int* out_count, const double* gmc_warp) {
    Y7TTrkHdr* h = (Y7TTrkHdr*)blob;
    const Y7TTrkCfg cfg = h->cfg;
    const Y7TTrk s = y7t_trk_bind_ex(ex, blob, cfg.cap_t, cfg.cap_d);
    const int kf = cfg.kf;
    if (cfg.tracker == Y7T_DEEPSORT && n >= 0) {      // frames with detections of a DeepSORT pool belong to y7t_tracker_step_deepsort (appearance rings); refuse, loudly
        if (ex.tid == 0) { h->status |= Y7T_ERR_KIND; if (out_count) *out_count = 0; }
        return;
    }
    y7t_sync(ex);
    Y7T_PROF(h, 0);
    if (ex.tid == 0) {
        h->frame_id += 1;
        h->n_act_last = h->n_refind_last = h->n_lostn_last = h->n_removed_last = 0;
        if (n > cfg.cap_d) h->status |= Y7T_ERR_CAP_D;
    }
    y7t_sync(ex);
    if (n > cfg.cap_d) n = cfg.cap_d;
    const int frame_id = h->frame_id;
    const int nt0 = h->n_tracked, nl0 = h->n_lost;
    // unconfirmed / confirmed split of tracked (bytetrack.py:95-100)
    const int n_unc = y7t_compact(ex, nt0, [&](int i) { return !s.act[s.tracked[i]]; }, s.tmpa, 0);
    for (int k = ex.tid; k < n_unc; k += ex.nt) s.unconf[k] = s.tracked[s.tmpa[k]];
    const int n_conf = y7t_compact(ex, nt0, [&](int i) { return s.act[s.tracked[i]] != 0; }, s.tmpb, 0);
    // strack_pool = joint_stracks(confirmed, lost): the two lists are disjoint by construction
    for (int k = ex.tid; k < n_conf; k += ex.nt) s.pool[k] = s.tracked[s.tmpb[k]];
    for (int k = ex.tid; k < nl0; k += ex.nt) s.pool[n_conf + k] = s.lost[k];
    y7t_sync(ex);
    const int n_pool = n_conf + nl0;
    Y7T_PROF(h, 1);
    y7t_multi_predict(ex, s, s.pool, n_pool);
    if (cfg.tracker == Y7T_BOTSORT && gmc_warp && n >= 0) {   // botsort.py:383-386: multi_gmc(strack_pool), multi_gmc(unconfirmed)
        const Y7TWarp Hm = y7t_warp_load(gmc_warp);
        for (int i = ex.tid; i < n_pool + n_unc; i += ex.nt) {
            const int sl = i < n_pool ? s.pool[i] : s.unconf[i - n_pool];
            y7t_kf_gmc(Hm, s.mean + 8 * (size_t)sl, s.cov + 64 * (size_t)sl);
            s.f32m[sl] = 0;
        }
        y7t_sync(ex);
    }
    Y7T_PROF(h, 2);
    if (n < 0) {  // update_without_detection (basetrack.py:489-537)
        y7t_finish(ex, s, out_rows, out_cap, out_count);
        return;
    }
    // detections -> STrack(cls, tlbr2tlwh(tlbr), score): float32 tlwh
    for (int j = ex.tid; j < n; j += ex.nt) {
        const float* r = dets + 6 * (size_t)j;
        s.dbox[4 * (size_t)j + 0] = r[0];
        s.dbox[4 * (size_t)j + 1] = r[1];
        s.dbox[4 * (size_t)j + 2] = r[2] - r[0];
        s.dbox[4 * (size_t)j + 3] = r[3] - r[1];
    }
    y7t_sync(ex);
    const float det_t = (float)cfg.det_thresh, low_t = (float)cfg.low_thresh;
    const float new_gate = (float)(cfg.det_thresh + 0.1);
    int n_hi, n_lo = 0;
    if (cfg.tracker == Y7T_SORT)
        n_hi = y7t_compact(ex, n, [&](int j) { return dets[6 * (size_t)j + 4] > det_t; }, s.dhi, 0);
    else {
        n_hi = y7t_compact(ex, n, [&](int j) { return dets[6 * (size_t)j + 4] >= det_t; }, s.dhi, 0);
        n_lo = y7t_compact(ex, n, [&](int j) { const float c = dets[6 * (size_t)j + 4]; return !(c >= det_t) && c > low_t; }, s.dlo, 0);
    }
    int na, nr;
    // ---- the three associations (bytetrack.py:104-160 / botsort.py:388-460 / basetrack.py SORT) as ONE loop: gather, assign, apply are the same text for each, and
    // one inlined copy of the solvers and of the Kalman updates keeps this function's code at 100 KB (three copies: 234 KB; a CALLED copy costs the caller ~10 % in
    // every phase -- what lives across a call goes through scratch: profiles/r05_tracker_association.txt section 3) ----
    //   0: pool vs high-score detections     1: remaining pool tracks vs low-score detections (not SORT)     2: unconfirmed tracks vs leftover high detections
    int n_left = 0;
    const bool is_sort = cfg.tracker == Y7T_SORT;
#if Y7T_DEVICE
#pragma clang loop unroll(disable)
#endif
    for (int ph = 0; ph < 3; ++ph) {
        const int* la; const int* ld;
        int nA, nD, mode;
        double th;
        if (ph == 0) { la = s.pool; nA = n_pool; ld = s.dhi; nD = n_hi; th = is_sort ? cfg.iou_thresh : 0.9; mode = is_sort ? 1 : 0; }
        else if (ph == 1) {
            if (is_sort) {
                // unmatched Tracked pool tracks -> Lost
                const int nl_new = y7t_compact(ex, n_pool, [&](int i) { return s.xrow[i] < 0 && s.state[s.pool[i]] == Y7T_TRACKED; }, s.tmpa, 0);
                for (int k = ex.tid; k < nl_new; k += ex.nt) { const int sl = s.pool[s.tmpa[k]]; s.lostn[k] = sl; }
                y7t_sync(ex);
                for (int k = ex.tid; k < nl_new; k += ex.nt) s.state[s.lostn[k]] = Y7T_LOST;
                if (ex.tid == 0) h->n_lostn_last = nl_new;
                y7t_sync(ex);
                continue;
            }
            // ByteTrack keeps only the still-Tracked leftovers (bytetrack.py:131); BoT-SORT takes every unmatched pool track (botsort.py:411)
            const bool only_tracked = cfg.tracker != Y7T_BOTSORT;
            const int n_rem = y7t_compact(ex, n_pool, [&](int i) { return s.xrow[i] < 0 && (!only_tracked || s.state[s.pool[i]] == Y7T_TRACKED); }, s.tmpa, 0);
            for (int k = ex.tid; k < n_rem; k += ex.nt) s.rem[k] = s.pool[s.tmpa[k]];
            y7t_sync(ex);
            la = s.rem; nA = n_rem; ld = s.dlo; nD = n_lo; th = 0.5; mode = 0;
        } else { la = s.unconf; nA = n_unc; ld = s.left; nD = n_left; th = is_sort ? cfg.iou_thresh + 0.1 : 0.7; mode = is_sort ? 1 : 2; }
        y7t_gather_track_tlbr(ex, s, la, nA);
        y7t_gather_det_tlbr(ex, s, ld, nD);
        y7t_sync(ex);
        const int stamp = ph == 0 ? 3 : ph == 1 ? 6 : 8;
        Y7T_PROF(h, stamp);
        y7t_assoc(ex, s, nA, nD, th);
        Y7T_PROF(h, stamp + 1);
        y7t_apply_matches(ex, s, la, nA, ld, dets, mode, na, nr);
        if (ph == 0) {
            Y7T_PROF(h, 5);
            // unmatched detections, in detection order: left = [D_high[i] for i in u_dets]
            n_left = y7t_compact(ex, n_hi, [&](int j) { return s.ycol[j] < 0; }, s.tmpa, 0);
            for (int k = ex.tid; k < n_left; k += ex.nt) s.left[k] = s.dhi[s.tmpa[k]];
            y7t_sync(ex);
        } else if (ph == 1) {
            const int nl_new = y7t_compact(ex, nA, [&](int i) { return s.xrow[i] < 0; }, s.tmpa, 0);
            for (int k = ex.tid; k < nl_new; k += ex.nt) { const int sl = s.rem[s.tmpa[k]]; s.lostn[k] = sl; s.state[sl] = Y7T_LOST; }
            if (ex.tid == 0) h->n_lostn_last = nl_new;
            y7t_sync(ex);
        } else {
            const int n_rm = y7t_compact(ex, n_unc, [&](int i) { return s.xrow[i] < 0; }, s.tmpa, 0);
            for (int k = ex.tid; k < n_rm; k += ex.nt) { const int sl = s.unconf[s.tmpa[k]]; s.removedl[k] = sl; s.state[sl] = Y7T_REMOVED; }
            if (ex.tid == 0) h->n_removed_last = n_rm;
            y7t_sync(ex);
        }
    }
    // ---- new tracks from still-unmatched detections above the gate (activate; ids in order) ----
    {
        // botsort.py:462-466 spawns from u_dets0_idx (every detection left after the FIRST association) -- reference quirk, kept
        const bool any_left = cfg.tracker == Y7T_BOTSORT;
        const int n_new = y7t_compact(ex, n_left, [&](int j) { return (any_left || s.ycol[j] < 0) && dets[6 * (size_t)s.left[j] + 4] > new_gate; }, s.tmpa, 0);
        int* idc = (int*)(uintptr_t)h->id_counter_ptr;
        if (ex.tid == 0) {
            int nf = h->n_free;
            const int base = h->n_act_last, made = n_new < nf ? n_new : nf;
            if (n_new > nf) h->status |= Y7T_ERR_CAP_T;
            // BaseTrack.next_id (basetrack.py:43-46): the counter is shared by every tracker of the process and y7t_tracker_step_batch
            // steps several trackers concurrently, so the frame's ids are reserved with ONE atomic add -- consecutive inside a
            // tracker (== the reference's order of activate() calls), never duplicated or lost across trackers
            const int id0 = made > 0 ? Y7T_FETCH_ADD(idc, made) : 0;
            for (int k = 0; k < made; ++k) {
                const int sl = s.freel[--nf];
                s.tmpb[k] = sl;
                s.tid[sl] = id0 + 1 + k;
                s.actl[base + k] = sl;
            }
            h->n_free = nf;
            h->n_act_last = base + made;
            s.xrow[0] = made;
        }
        y7t_sync(ex);
        const int made = s.xrow[0];
        for (int k = ex.tid; k < made; k += ex.nt) {
            const int sl = s.tmpb[k], dj = s.left[s.tmpa[k]];
            double z[4];
            for (int c = 0; c < 4; ++c) s.box[4 * (size_t)sl + c] = s.dbox[4 * (size_t)dj + c];
            y7t_meas(kf, s.dbox + 4 * (size_t)dj, z);
            y7t_kf_initiate(kf, z, cfg.f32_quirk, s.mean + 8 * (size_t)sl, s.cov + 64 * (size_t)sl);
            s.f32m[sl] = cfg.f32_quirk;
            s.score[sl] = dets[6 * (size_t)dj + 4];
            s.cls[sl] = dets[6 * (size_t)dj + 5];
            s.state[sl] = Y7T_TRACKED;
            s.act[sl] = (frame_id == 1) ? 1 : 0;
            s.frame[sl] = frame_id; s.start[sl] = frame_id;
            s.tsu[sl] = 0; s.len[sl] = 0; s.inrem[sl] = 0;
        }
        y7t_sync(ex);
    }
    // ---- age out long-lost tracks (bytetrack.py:180-183) ----
    {
        const int n_old = y7t_compact(ex, nl0, [&](int i) { return frame_id - s.frame[s.lost[i]] > cfg.max_time_lost; }, s.tmpa, 0);
        const int base = h->n_removed_last;
        for (int k = ex.tid; k < n_old; k += ex.nt) { const int sl = s.lost[s.tmpa[k]]; s.removedl[base + k] = sl; s.state[sl] = Y7T_REMOVED; }
        y7t_sync(ex);
        if (ex.tid == 0) h->n_removed_last = base + n_old;
        y7t_sync(ex);
    }
    Y7T_PROF(h, 10);
    y7t_finish(ex, s, out_rows, out_cap, out_count);
    Y7T_PROF(h, 11);
}

// initialise a state blob (single thread is enough; called once)
Y7T_FN void y7t_tracker_init(const Y7TExec& ex, void* blob, const Y7TTrkCfg& cfg, unsigned long long idc) {
    Y7TTrkHdr* h = (Y7TTrkHdr*)blob;
    const Y7TTrk s = y7t_trk_bind(blob, cfg.cap_t, cfg.cap_d);
    if (ex.tid == 0) {
        h->magic = 0x59375431;
        h->frame_id = 0; h->n_tracked = 0; h->n_lost = 0; h->n_free = cfg.cap_t; h->status = 0; h->n_out = 0;
        h->n_removed_total = 0;
        h->n_act_last = h->n_refind_last = h->n_lostn_last = h->n_removed_last = 0;
        h->id_counter_ptr = idc;
        h->cfg = cfg;
    }
    for (int k = ex.tid; k < cfg.cap_t; k += ex.nt) {
        s.freel[k] = cfg.cap_t - 1 - k;  // pop order 0,1,2,...
        s.state[k] = Y7T_NEW; s.act[k] = 0; s.inrem[k] = 0; s.tid[k] = 0; s.f32m[k] = 0; s.mark[k] = 0;
    }
    y7t_sync(ex);
}

Y7T_NOINL void y7t_tracker_step(const Y7TExec& ex, void* blob, const float* dets, int n, double* out_rows, int out_cap, int* out_count, const double* gmc_warp) {
    y7t_tracker_step_body(ex, blob, dets, n, out_rows, out_cap, out_count, gmc_warp);
}
