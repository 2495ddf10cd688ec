// y7t_track_deepsort.h -- DeepSORT's per-frame association (appearance + motion cascade) as a workgroup program over the same
// device-resident track pool as ByteTrack / SORT (y7t_track_step.h).  Portable text (device: hipcc; CPU tests: -DY7T_HOSTSIM).
//
// Restates /root/reference/tracker/deepsort.py:43-224 (gate_cost_matrix, gated_metric, DeepSORT.update),
// tracker/matching.py:105-127 (nearest_embedding_distance), :165-178 (cal_cosine_distance), :216-277 (matching_cascade),
// tracker/basetrack.py:296-339 (STrack.update with use_avg_of_feature=False: append the normalised feature, keep the last
// store_features_budget = 100), tracker/kalman_filter.py:365-411 (gating_distance, metric 'maha').
//
// Two reference behaviours that decide WHICH tracks are touched are reproduced literally:
//   * matching_cascade returns `list(set(track_indices) - set(matched))` (matching.py:275): the order of that list is CPython's
//     set-table order, not ascending -- y7t_pyset_difference emulates Objects/setobject.c (3.10) for small non-negative ints;
//   * deepsort.py:171-173 marks `strack_pool[idx]` lost for idx in the unmatched ROWS of u_tracks0 (an index into the filtered
//     list applied to the unfiltered pool).
#pragma once
#include "y7t_track_step.h"

// (Y7T_DEEPSORT = 3: y7t_track_core.h)
#define Y7T_PYSET_CAP 8192      // table entries of the emulated CPython set (enough for 1228 unmatched tracks)

// feature state of one DeepSORT tracker: caller-owned device memory next to the track-pool blob
struct Y7TFeatHdr { int magic, dim, budget, cap_t, cap_d, status, n_pend, pad1; };
struct Y7TFeatLayout { size_t ring, nfeat, fpos, app, detn, pend, cmin, cmax, casc_tr, casc_det, u0, tomatch, tmpd, pyset, total; };
struct Y7TFeat {
    Y7TFeatHdr* h;
    float* ring;        // [cap_t][budget][dim]   STrack.features, a ring of the last `budget` -- each stored as the row of cal_cosine_distance's mat1 it
                        //                        becomes: (raw first feature | f / |f| later ones) divided by its np.linalg.norm (y7t_feat_store)
    int *nfeat, *fpos;  // [cap_t] stored features, next write position
    float* app;         // [cap_t][cap_d]         nearest_embedding_distance(slot, detection row) of this frame
    float* detn;        // [cap_d][dim]           this frame's detection features, normalised (cal_cosine_distance's mat2)
    int* pend;          // [cap_d][3]             appearance vectors this frame's step decided to store: (slot, detection row, 1 = update / 0 = new track);
                        //                        written out after the step by y7t_feat_store_pending (grid-wide: a wave per vector)
    int *cmin, *cmax;   // [cap_d]                youngest / oldest time_since_update among the live slots whose nearest-embedding distance to the detection
                        //                        passes the appearance test (<= 0.15): reset by y7t_feat_normalize_dets, filled with the distances
    int *casc_tr, *casc_det, *u0;   // [cap_t]    cascade matches in match order (pool index, position in the detection list); unmatched pool indices
    int *tomatch, *tmpd;            // [cap_d]    detections_to_match of the current cascade level (+ scratch)
    int* pyset;         // [2][Y7T_PYSET_CAP]
};

Y7T_HD Y7TFeatLayout y7t_feat_layout(int cap_t, int cap_d, int dim, int budget) {
    Y7TFeatLayout L;
    size_t o = y7t_al(sizeof(Y7TFeatHdr));
    const size_t T = (size_t)cap_t, D = (size_t)cap_d;
#define Y7T_TAKE(f, bytes) L.f = o; o = y7t_al(o + (bytes));
    Y7T_TAKE(ring, T * budget * dim * 4) Y7T_TAKE(nfeat, T * 4) Y7T_TAKE(fpos, T * 4) Y7T_TAKE(app, T * D * 4)
    Y7T_TAKE(detn, D * dim * 4) Y7T_TAKE(pend, D * 3 * 4) Y7T_TAKE(cmin, D * 4) Y7T_TAKE(cmax, D * 4)
    Y7T_TAKE(casc_tr, T * 4) Y7T_TAKE(casc_det, T * 4) Y7T_TAKE(u0, T * 4) Y7T_TAKE(tomatch, D * 4) Y7T_TAKE(tmpd, D * 4)
    Y7T_TAKE(pyset, (size_t)2 * Y7T_PYSET_CAP * 4)
#undef Y7T_TAKE
    L.total = o;
    return L;
}

Y7T_FN Y7TFeat y7t_feat_bind(void* blob) {
    Y7TFeatHdr* h = (Y7TFeatHdr*)blob;
    const Y7TFeatLayout L = y7t_feat_layout(h->cap_t, h->cap_d, h->dim, h->budget);
    char* b = (char*)blob;
    Y7TFeat f;
    f.h = h;
    f.ring = (float*)(b + L.ring); f.nfeat = (int*)(b + L.nfeat); f.fpos = (int*)(b + L.fpos); f.app = (float*)(b + L.app);
    f.detn = (float*)(b + L.detn); f.pend = (int*)(b + L.pend); f.cmin = (int*)(b + L.cmin); f.cmax = (int*)(b + L.cmax);
    f.casc_tr = (int*)(b + L.casc_tr); f.casc_det = (int*)(b + L.casc_det); f.u0 = (int*)(b + L.u0);
    f.tomatch = (int*)(b + L.tomatch); f.tmpd = (int*)(b + L.tmpd); f.pyset = (int*)(b + L.pyset);
    return f;
}

Y7T_FN void y7t_feat_init(const Y7TExec& ex, void* blob, int cap_t, int cap_d, int dim, int budget) {
    Y7TFeatHdr* h = (Y7TFeatHdr*)blob;
    if (ex.tid == 0) { h->magic = 0x59374631; h->dim = dim; h->budget = budget; h->cap_t = cap_t; h->cap_d = cap_d; h->status = 0; h->n_pend = 0; }
    y7t_sync(ex);
    const Y7TFeat f = y7t_feat_bind(blob);
    for (int k = ex.tid; k < cap_t; k += ex.nt) { f.nfeat[k] = 0; f.fpos[k] = 0; }
    y7t_sync(ex);
}

// ---------------------------------------------------------------------------------------------
// CPython 3.10 set semantics for `list(set(range(n)) - set(matched))` (matching.py:275), keys = small non-negative ints
// (hash(i) == i).  Objects/setobject.c: set_difference takes one of two routes --
//   * len(so) >> 2 > len(other): copy `so` (one clean re-insert into a table of the next power of two above 2 n: every key lands in
//     slot == key) and discard the members of `other`  ->  iteration order = ascending;
//   * otherwise: a NEW set grows from 8 slots while `so` is walked in table order (ascending: range(n) never wraps its own
//     table) -- set_add_entry's linear probing (LINEAR_PROBES 9, perturb >> 5) with keys wrapping modulo the table size, resize to the
//     first power of two above 4 * used whenever fill * 5 >= mask * 3 (set_insert_clean in old-table order).
// Sequential; thread 0 only.  member[k] != 0: k is in `other`.  Returns the count, order in out[].  tab: 2 * Y7T_PYSET_CAP ints.
// ---------------------------------------------------------------------------------------------
Y7T_FN void y7t_pyset_insert_clean(int* t, int mask, int v) {
    unsigned perturb = (unsigned)v;
    int i = v & mask;
    for (;;) {
        if (t[i] < 0) { t[i] = v; return; }
        if (i + 9 <= mask) {
            for (int j = 1; j <= 9; ++j)
                if (t[i + j] < 0) { t[i + j] = v; return; }
        }
        perturb >>= 5;
        i = (int)(((unsigned)i * 5u + 1u + perturb) & (unsigned)mask);
    }
}

// `unm`: the keys of range(n) that are NOT in `other`, ascending (what walking `so` in table order and skipping members visits)
Y7T_FN int y7t_pyset_difference_list(int n, const int* unm, int n_unm, int n_other, int* out, int* tab, int* status, int cap = Y7T_PYSET_CAP) {
    int cnt = 0;
    if ((n >> 2) > n_other) {                     // set_copy_and_difference
        for (int k = 0; k < n_unm; ++k) out[cnt++] = unm[k];
        return cnt;
    }
    int* cur = tab;
    int* nxt = tab + cap;                         // two tables of `cap` entries (cap >= the power of two above 4 * n_unm)
    int mask = 7, fill = 0;
    for (int i = 0; i <= mask; ++i) cur[i] = -1;
    for (int q = 0; q < n_unm; ++q) {
        const int v = unm[q];
        // set_add_entry
        unsigned perturb = (unsigned)v;
        int i = v & mask, e = -1;
        for (;;) {
            const int probes = (i + 9 <= mask) ? 9 : 0;
            for (int j = 0; j <= probes; ++j)
                if (cur[i + j] < 0) { e = i + j; break; }
            if (e >= 0) break;
            perturb >>= 5;
            i = (int)(((unsigned)i * 5u + 1u + perturb) & (unsigned)mask);
        }
        cur[e] = v;
        ++fill;
        if (fill * 5 >= mask * 3) {               // set_table_resize(so, used * 4)   (used <= 50000)
            int newsize = 8;
            while (newsize <= fill * 4) newsize <<= 1;
            if (newsize > cap) { if (status) *status |= 8; break; }
            for (int k = 0; k < newsize; ++k) nxt[k] = -1;
            for (int k = 0; k <= mask; ++k) if (cur[k] >= 0) y7t_pyset_insert_clean(nxt, newsize - 1, cur[k]);
            int* sw = cur; cur = nxt; nxt = sw;
            mask = newsize - 1;
        }
    }
    for (int k = 0; k <= mask; ++k) if (cur[k] >= 0) out[cnt++] = cur[k];
    return cnt;
}

// ---------------------------------------------------------------------------------------------
// float32 arithmetic of the reference's appearance cost, as numpy 2.2 / OpenBLAS 0.3.29 (AVX-512 kernels) evaluate it in the environment
// the golden sequences were recorded in -- the costs of similar objects differ by a few float32 ulps, and the assignment follows them:
//   * np.linalg.norm(mat, axis=1): x * x rounded to float32, then add.reduce's PAIRWISE sum (numpy/_core/src/umath/loops_utils.h.src:
//     8 running sums for blocks <= 128, recursive halving above), sqrt;
//   * np.dot(mat1, mat2.T) (sgemm): one sequential fused-multiply-add chain over k per output;
//   * np.linalg.norm(vec) of a 1-D feature (STrack.update, basetrack.py:325) = sqrt(sdot(x, x)): OpenBLAS's SkylakeX sdot kernel --
//     four 16-lane accumulators over 64 elements per step (FMA), folded 16 -> 8 lanes, summed ((a0 + a1) + a2) + a3, halves added,
//     two horizontal adds.
// ---------------------------------------------------------------------------------------------
Y7T_NOINL float y7t_np_pairwise_sumsq(const float* x, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r = r + x[i] * x[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = x[j] * x[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + x[i + j] * x[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + x[i] * x[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return y7t_np_pairwise_sumsq(x, n2) + y7t_np_pairwise_sumsq(x + n2, n - n2);
}

Y7T_FN float y7t_fmaf(float a, float b, float c) {
#if Y7T_DEVICE
    return __builtin_fmaf(a, b, c);
#else
    return fmaf(a, b, c);
#endif
}

Y7T_FN float y7t_blas_sdot_self(const float* x, int n) {
    const int n1 = n & -32, n64 = n1 & ~63;
    float a5[4][16], a[4][8];
    for (int v = 0; v < 4; ++v) for (int l = 0; l < 16; ++l) a5[v][l] = 0.f;
    int i = 0;
    for (; i < n64; i += 64)
        for (int v = 0; v < 4; ++v) for (int l = 0; l < 16; ++l) { const float t = x[i + 16 * v + l]; a5[v][l] = y7t_fmaf(t, t, a5[v][l]); }
    for (int v = 0; v < 4; ++v) for (int l = 0; l < 8; ++l) a[v][l] = a5[v][l] + a5[v][l + 8];
    for (; i < n1; i += 32)
        for (int v = 0; v < 4; ++v) for (int l = 0; l < 8; ++l) { const float t = x[i + 8 * v + l]; a[v][l] = y7t_fmaf(t, t, a[v][l]); }
    float sv[8], hv[4];
    for (int l = 0; l < 8; ++l) sv[l] = ((a[0][l] + a[1][l]) + a[2][l]) + a[3][l];
    for (int l = 0; l < 4; ++l) hv[l] = sv[l] + sv[l + 4];
    float d = (hv[0] + hv[1]) + (hv[2] + hv[3]);
    for (int k = n1; k < n; ++k) d = d + x[k] * x[k];
    return d;
}

#if Y7T_DEVICE
// The same sums with the 64 lanes of a wave sharing one vector (every lane calls; every lane gets the result) -- bit-identical to the serial forms:
// each partial sum is built from the same operands in the same order, only by different lanes.
Y7T_FN bool y7t_dim_wave_ok(int dim) { return dim == 128 || dim == 256 || dim == 512 || dim == 1024; }

// y7t_np_pairwise_sumsq of the vector x[i] / div (div = 1: x itself): lane (block, j) owns running sum j of its 128-element block
Y7T_FN float y7t_wave_pairwise_sumsq(const float* x, int dim, float div, int lane) {
    const int nblk = dim >> 7, blk = lane >> 3, j = lane & 7;
    float r = 0.f;
    if (blk < nblk) {
        const float* xb = x + blk * 128;
        const float t0 = xb[j] / div;
        r = t0 * t0;
        for (int i = 8; i < 128; i += 8) { const float t = xb[i + j] / div; r = r + t * t; }
    }
    r = r + __shfl_xor(r, 1, 64); r = r + __shfl_xor(r, 2, 64); r = r + __shfl_xor(r, 4, 64);      // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
    for (int m = 1; m < nblk; m <<= 1) r = r + __shfl_xor(r, 8 * m, 64);                          // the recursive halving over blocks
    return __shfl(r, 0, 64);
}

// y7t_blas_sdot_self: lane 16 v + l owns accumulator (v, l)
Y7T_FN float y7t_wave_sdot_self(const float* x, int n, int lane) {
    const int n1 = n & -32, n64 = n1 & ~63;
    const int v = lane >> 4, l = lane & 15;
    float a5 = 0.f;
    for (int i = 0; i < n64; i += 64) { const float t = x[i + lane]; a5 = y7t_fmaf(t, t, a5); }
    float a = a5 + __shfl(a5, (lane + 8) & 63, 64);                                             // meaningful on l < 8
    if (n1 > n64 && l < 8) { const float t = x[n64 + 8 * v + l]; a = y7t_fmaf(t, t, a); }
    const float a0 = __shfl(a, l & 7, 64), a1 = __shfl(a, 16 + (l & 7), 64), a2 = __shfl(a, 32 + (l & 7), 64), a3 = __shfl(a, 48 + (l & 7), 64);
    const float sv = ((a0 + a1) + a2) + a3;                                                     // sv[l & 7] on every lane
    const float hv = sv + __shfl(sv, (lane & 48) + (((l & 7) + 4) & 7), 64);                    // l & 7 < 4: sv[l] + sv[l + 4]
    const float h0 = __shfl(hv, 0, 64), h1 = __shfl(hv, 1, 64), h2 = __shfl(hv, 2, 64), h3 = __shfl(hv, 3, 64);
    float d = (h0 + h1) + (h2 + h3);
    for (int k = n1; k < n; ++k) d = d + x[k] * x[k];
    return d;
}
#endif

// What STrack keeps of an appearance vector, in the form the distance uses it.  STrack stores the raw vector at creation (basetrack.py:97-103) and
// f / np.linalg.norm(f) on every update (basetrack.py:324-332, 1-D norm = sqrt(sdot)); nearest_embedding_distance then divides every stored row by
// its np.linalg.norm(axis=1) (pairwise sum) again.  Both divisions happen here, once, when the vector is stored.  Serial (one thread).
Y7T_FN void y7t_feat_store(float* dst, const float* src, int dim, int update) {
    const float nb = update ? sqrtf(y7t_blas_sdot_self(src, dim)) : 1.0f;
    for (int d = 0; d < dim; ++d) dst[d] = src[d] / nb;
    const float na = sqrtf(y7t_np_pairwise_sumsq(dst, dim));
    for (int d = 0; d < dim; ++d) dst[d] = dst[d] / na;
}

// Queue the vectors a step decides to store; y7t_feat_store_pending writes them after the step (nothing in the step reads the ring: this frame's
// distances were taken before it).  A detection row is stored at most once per frame and a slot receives at most one vector.
template <class PairFn>
Y7T_FN void y7t_feat_store_many(const Y7TExec& ex, const Y7TFeat& f, int count, int update, PairFn pair /* (i, slot&, detection row&) -> bool */) {
    for (int i = ex.tid; i < count; i += ex.nt) {
        int sl, row;
        if (!pair(i, sl, row)) continue;
        const int k = Y7T_FETCH_ADD(&f.h->n_pend, 1);
        if (k < f.h->cap_d) { f.pend[3 * k] = sl; f.pend[3 * k + 1] = row; f.pend[3 * k + 2] = update; }
        else f.h->status |= 16;
    }
    y7t_sync(ex);
}

// write the queued vectors (ex may span a whole grid: on the device a wave per vector); resets the queue
Y7T_FN void y7t_feat_store_pending(const Y7TExec& ex, const Y7TFeat& f, const float* det_feats) {
    const int dim = f.h->dim, budget = f.h->budget;
    const int count = f.h->n_pend < f.h->cap_d ? f.h->n_pend : f.h->cap_d;
#if Y7T_DEVICE
    if (y7t_dim_wave_ok(dim) && (ex.nt & 63) == 0) {
        const int lane = ex.tid & 63;
        for (int i = ex.tid >> 6; i < count; i += ex.nt >> 6) {          // wave-uniform
            const int sl = f.pend[3 * i], update = f.pend[3 * i + 2];
            const float* b = det_feats + (size_t)f.pend[3 * i + 1] * dim;
            const int pos = update ? f.fpos[sl] : 0;
            float* dst = f.ring + ((size_t)sl * budget + pos) * dim;
            const float nb = update ? sqrtf(y7t_wave_sdot_self(b, dim, lane)) : 1.0f;
            const float na = sqrtf(y7t_wave_pairwise_sumsq(b, dim, nb, lane));
            for (int d = lane; d < dim; d += 64) dst[d] = (b[d] / nb) / na;
            if (lane == 0) {
                if (update) { f.fpos[sl] = (pos + 1 == budget) ? 0 : pos + 1; if (f.nfeat[sl] < budget) f.nfeat[sl] += 1; }
                else { f.nfeat[sl] = 1; f.fpos[sl] = 1 % budget; }
            }
        }
        return;
    }
#endif
    for (int i = ex.tid; i < count; i += ex.nt) {
        const int sl = f.pend[3 * i], update = f.pend[3 * i + 2];
        const int pos = update ? f.fpos[sl] : 0;
        y7t_feat_store(f.ring + ((size_t)sl * budget + pos) * dim, det_feats + (size_t)f.pend[3 * i + 1] * dim, dim, update);
        if (update) { f.fpos[sl] = (pos + 1 == budget) ? 0 : pos + 1; if (f.nfeat[sl] < budget) f.nfeat[sl] += 1; }
        else { f.nfeat[sl] = 1; f.fpos[sl] = 1 % budget; }
    }
}

// mat2 of cal_cosine_distance: every detection feature of the frame divided by its norm (device: a wave per detection; else a lane each)
Y7T_FN void y7t_feat_normalize_dets(const Y7TExec& ex, const Y7TFeat& f, const float* det_feats, int n) {
    const int dim = f.h->dim;
    for (int j = ex.tid; j < n; j += ex.nt) { f.cmin[j] = 0x7fffffff; f.cmax[j] = -1; }
#if Y7T_DEVICE
    if (y7t_dim_wave_ok(dim) && (ex.nt & 63) == 0) {
        const int lane = ex.tid & 63;
        for (int j = ex.tid >> 6; j < n; j += ex.nt >> 6) {
            const float* b = det_feats + (size_t)j * dim;
            const float nb = sqrtf(y7t_wave_pairwise_sumsq(b, dim, 1.0f, lane));
            float* o = f.detn + (size_t)j * dim;
            for (int d = lane; d < dim; d += 64) o[d] = b[d] / nb;
        }
        y7t_sync(ex);
        return;
    }
#endif
    for (int j = ex.tid; j < n; j += ex.nt) {
        const float* b = det_feats + (size_t)j * dim;
        const float nb = sqrtf(y7t_np_pairwise_sumsq(b, dim));
        float* o = f.detn + (size_t)j * dim;
        for (int d = 0; d < dim; ++d) o[d] = b[d] / nb;
    }
    y7t_sync(ex);
}

// nearest_embedding_distance (matching.py:105-127) for ONE pool slot against every detection feature of the frame:
//   app[slot][j] = min over the slot's stored rows a' of  1 - a' . (b_j / |b_j|)        (float32: one sequential FMA chain over k per product)
// Plain form (a lane per detection); the device runs the tiled k_embed_dist (y7t_tracker.hip) with the same chains instead.
Y7T_FN void y7t_embed_slot(const Y7TExec& ex, const Y7TFeat& f, int slot, int n, int tsu /* the slot's time_since_update: ages of a detection's appearance candidates */) {
    const int nf = f.nfeat[slot], dim = f.h->dim;
    if (nf <= 0) return;
    const float* hist = f.ring + (size_t)slot * f.h->budget * dim;
    float* row = f.app + (size_t)slot * f.h->cap_d;
    for (int j = ex.tid; j < n; j += ex.nt) {
        const float* b = f.detn + (size_t)j * dim;
        float best = 3.0e38f;
        for (int hI = 0; hI < nf; ++hI) {
            const float* a = hist + (size_t)hI * dim;
            float acc = 0.f;
            for (int k = 0; k < dim; ++k) acc = y7t_fmaf(a[k], b[k], acc);
            const float c = 1.0f - acc;
            best = c < best ? c : best;
        }
        row[j] = best;
        if ((double)best <= 0.15) { Y7T_ATOMIC_MIN_I(f.cmin + j, tsu); Y7T_ATOMIC_MAX(f.cmax + j, tsu); }
    }
    y7t_sync(ex);
}

// STrack.update's feature bookkeeping (basetrack.py:324-332, use_avg_of_feature = False) for the rows of `tracks` that apply_matches
// UPDATED (tmpa[i] == 1): features.append(f / np.linalg.norm(f)); features = features[-budget:]
Y7T_FN void y7t_ds_append_features(const Y7TExec& ex, const Y7TTrk& s, const Y7TFeat& f, const int* tracks, int na, const int* dets,
                                   const float* det_feats) {
    y7t_feat_store_many(ex, f, na, 1, [&](int i, int& sl, int& row) {
        if (s.xrow[i] < 0 || s.tmpa[i] != 1) return false;
        sl = tracks[i];
        row = dets[s.xrow[i]];
        return true;
    });
}

// linear_assignment(cost, thresh) on a cost matrix already in s.cost (na x nb, row stride nb) -> s.xrow / s.ycol
Y7T_FN void y7t_assign_on_cost(const Y7TExec& ex, const Y7TTrk& s, int na, int nb, double thresh, bool literal = false) {
    Y7TLap L;
    L.nr = na; L.nc = nb; L.ld = nb; L.n = na + nb; L.half = thresh / 2.0; L.prof = nullptr;
    const size_t ws = y7t_al(y7t_lap_ws_bytes(L.n));
    void* lapws = s.lapws;
    if (ex.fast && ws <= ex.fast_bytes) lapws = ex.fast;
    L.c = s.cost;
    y7t_lap_bind(L, lapws, L.n);
    if (literal || y7t_lap_solve_sap(ex, L)) y7t_lap_solve_literal(ex, L);      // ties: lapjv's own order decides
    for (int i = ex.tid; i < na; i += ex.nt) s.xrow[i] = (L.x[i] >= nb) ? -1 : L.x[i];
    for (int j = ex.tid; j < nb; j += ex.nt) s.ycol[j] = (L.y[j] >= na) ? -1 : L.y[j];
    y7t_sync(ex);
}

// One DeepSORT frame (deepsort.py:79-227).  dets: n x 6 float32 rows; det_feats: n x dim float32, row j = appearance feature of
// detection row j (rows with conf <= det_thresh are never read).  f.app must hold this frame's nearest-embedding distances
// (y7t_feat_normalize_dets, then y7t_embed_slot for every slot of the tracked / lost lists) -- they do not depend on the Kalman state.
// (inlined into its one kernel, k_tracker_step_deepsort<MAXT>: a called function would not inherit the kernel's __launch_bounds__ -- see y7t_tracker_step_body)
Y7T_FN void y7t_tracker_step_deepsort(const Y7TExec& ex, void* blob, void* fblob, const float* dets, int n, const float* det_feats,
                                      double* out_rows, int out_cap, int* out_count) {
    Y7TTrkHdr* h = (Y7TTrkHdr*)blob;
    const Y7TTrkCfg cfg = h->cfg;
    const Y7TTrk s = y7t_trk_bind(blob, cfg.cap_t, cfg.cap_d);
    const Y7TFeat f = y7t_feat_bind(fblob);
    const int kf = cfg.kf;
    y7t_sync(ex);
    if (ex.tid == 0) {
        h->frame_id += 1;
        f.h->n_pend = 0;
        h->n_act_last = h->n_refind_last = h->n_lostn_last = h->n_removed_last = 0;
        if (n > cfg.cap_d) h->status |= Y7T_ERR_CAP_D;
    }
    y7t_sync(ex);
    if (n > cfg.cap_d) n = cfg.cap_d;
    const int frame_id = h->frame_id;
    const int nt0 = h->n_tracked, nl0 = h->n_lost;
    Y7T_PROF(h, 0);
    const int n_unc = y7t_compact(ex, nt0, [&](int i) { return !s.act[s.tracked[i]]; }, s.tmpa, 0);
    for (int k = ex.tid; k < n_unc; k += ex.nt) s.unconf[k] = s.tracked[s.tmpa[k]];
    const int n_conf = y7t_compact(ex, nt0, [&](int i) { return s.act[s.tracked[i]] != 0; }, s.tmpb, 0);
    for (int k = ex.tid; k < n_conf; k += ex.nt) s.pool[k] = s.tracked[s.tmpb[k]];
    for (int k = ex.tid; k < nl0; k += ex.nt) s.pool[n_conf + k] = s.lost[k];
    y7t_sync(ex);
    const int n_pool = n_conf + nl0;
    y7t_multi_predict(ex, s, s.pool, n_pool);
    // detections with conf > det_thresh (deepsort.py:98), float32 tlwh
    for (int j = ex.tid; j < n; j += ex.nt) {
        const float* r = dets + 6 * (size_t)j;
        s.dbox[4 * (size_t)j + 0] = r[0]; s.dbox[4 * (size_t)j + 1] = r[1];
        s.dbox[4 * (size_t)j + 2] = r[2] - r[0]; s.dbox[4 * (size_t)j + 3] = r[3] - r[1];
    }
    y7t_sync(ex);
    const float det_t = (float)cfg.det_thresh;
    const int n_hi = y7t_compact(ex, n, [&](int j) { return dets[6 * (size_t)j + 4] > det_t; }, s.dhi, 0);
    Y7T_PROF(h, 1);
    // ---- matching_cascade(gated_metric, 0.9, max_time_lost, strack_pool, detections) ----
    int n_to = n_hi, nm = 0;
    for (int k = ex.tid; k < n_hi; k += ex.nt) f.tomatch[k] = k;      // positions in the detection list (s.dhi)
    y7t_sync(ex);
#if Y7T_DEVICE
#define Y7T_CPROF(i) do { if (ex.tid == 0) { const long long t_ = clock64(); h->prof[i] += t_ - cprof_t; cprof_t = t_; } } while (0)
    long long cprof_t = clock64();
    if (ex.tid == 0) h->prof[16] = h->prof[17] = h->prof[18] = h->prof[19] = 0;
#else
#define Y7T_CPROF(i) do { } while (0)
#endif
    // which ages occur at all: one pass instead of a compaction per level (time_since_update of a pool track is 1 .. max_time_lost)
    int* lvl = s.ycol;                         // [max_time_lost] flags (ycol is free until the first assignment)
    for (int k = ex.tid; k < cfg.max_time_lost; k += ex.nt) lvl[k] = 0;
    y7t_sync(ex);
    for (int i = ex.tid; i < n_pool; i += ex.nt) { const int a = s.tsu[s.pool[i]] - 1; if (a >= 0 && a < cfg.max_time_lost) lvl[a] = 1; }
    y7t_sync(ex);
    unsigned long long lvl_mask = 0;           // max_time_lost <= 64 levels in the mask, anything above is scanned the slow way
    for (int k = 0; k < cfg.max_time_lost && k < 64; ++k) if (lvl[k]) lvl_mask |= 1ull << k;
    y7t_sync(ex);
    auto gated_at = [&](int sl, int dj) {
        // gated_metric: appearance cost, > 0.15 -> 1e5; squared Mahalanobis distance to the predicted state > chi2inv95[4] -> 1e5.  The gate is only
        // evaluated where the appearance test leaves a finite cost (the result is 1e5 either way otherwise).
        double cost = (double)f.app[(size_t)sl * cfg.cap_d + dj];
        if (cost > 0.15) return 1e5;
        double z[4];
        y7t_meas(kf, s.dbox + 4 * (size_t)dj, z);
        if (y7t_kf_gating(kf, s.mean + 8 * (size_t)sl, s.cov + 64 * (size_t)sl, z, 0) > 9.4877) cost = 1e5;
        return cost;
    };
    // ---- all levels in ONE assignment when no detection is wanted by tracks of two different ages.  The cascade hands level L only the detections the
    // younger levels left over; if every detection's appearance candidates (a superset of its candidates) share one age, those left-overs are exactly the
    // columns level L could use anyway: the candidate graph of the joint problem is the disjoint union of the levels' graphs, the solver works per
    // connected component, and rows / columns keep their relative order -- same matches, one set of passes instead of one per age. ----
    bool joint_done = false;
#ifdef Y7T_NO_JOINT      // (experiments: the level-by-level cascade only)
    if (false) {
#else
    if (cfg.max_time_lost <= 64 && (lvl_mask & (lvl_mask - 1)) != 0 && n_to > 0) {
#endif
        const int n_rows = y7t_compact(ex, n_pool, [&](int i) { const int a = s.tsu[s.pool[i]] - 1; return a >= 0 && a < cfg.max_time_lost; }, s.rem, 0);
        for (int r = ex.tid; r < n_rows; r += ex.nt) s.tmpb[r] = s.pool[s.rem[r]];
        for (int c = ex.tid; c < n_hi; c += ex.nt) s.left[c] = s.dhi[c];
        if (ex.tid == 0) s.xrow[0] = 0;
        y7t_sync(ex);
        // (f.cmin / f.cmax: ages of every detection's appearance candidates among ALL live slots, recorded with the distances -- a superset of the pool's)
        for (int c = ex.tid; c < n_hi; c += ex.nt) if (f.cmax[s.left[c]] > f.cmin[s.left[c]]) s.xrow[0] = 1;
        y7t_sync(ex);
        const bool mixed = s.xrow[0] != 0;
        y7t_sync(ex);
#if Y7T_DEVICE
        if (ex.tid == 0) { h->prof[27] += 1; if (mixed) h->prof[28] += 1; }      // diagnostics: frames with several ages / with a contested detection
#endif
        if (!mixed && y7t_assoc_sparse_fn(ex, s, n_rows, n_hi, 0.9, [&](int c) { return s.left[c]; }, [&](int r) { return s.tmpb[r]; }, [&](int sl, int r, int dj) { return gated_at(y7t_row_at(sl, r), dj); }) == 1) {
            // the cascade's match order: by age, inside an age by row
            const int nm2 = y7t_compact(ex, n_rows, [&](int r) { return s.xrow[r] >= 0; }, s.tmpa, 0);
            int* age = f.tmpd;
            for (int k = ex.tid; k < nm2; k += ex.nt) age[k] = s.tsu[s.tmpb[s.tmpa[k]]];
            y7t_sync(ex);
            for (int k = ex.tid; k < nm2; k += ex.nt) {
                const int r = s.tmpa[k], a = age[k];
                int rank = 0;
                for (int k2 = 0; k2 < nm2; ++k2) rank += (age[k2] < a) || (age[k2] == a && s.tmpa[k2] < r);
                f.casc_tr[rank] = s.rem[r];
                f.casc_det[rank] = s.xrow[r];              // (detections_to_match was the whole list: position == column)
            }
            nm = nm2;
            const int n_left = y7t_compact(ex, n_hi, [&](int c) { return s.ycol[c] < 0; }, f.tomatch, 0);
            y7t_sync(ex);
            n_to = n_left;
            joint_done = true;
#if Y7T_DEVICE
            if (ex.tid == 0) h->prof[29] += 1;
#endif
            Y7T_TIE_REASON(7);                         // (counted by the CPU test build: how often the joint form applies)
        }
        y7t_sync(ex);
    }
    for (int level = 0; !joint_done && level < cfg.max_time_lost && n_to > 0; ++level) {
        if (level < 64 && !((lvl_mask >> level) & 1ull)) continue;
        const int n_tl = y7t_compact(ex, n_pool, [&](int i) { return s.tsu[s.pool[i]] == 1 + level; }, s.rem, 0);   // pool indices of this age
        Y7T_CPROF(16);
        if (n_tl == 0) continue;
        for (int r = ex.tid; r < n_tl; r += ex.nt) s.tmpb[r] = s.pool[s.rem[r]];             // pool slot of every row
        for (int c = ex.tid; c < n_to; c += ex.nt) s.left[c] = s.dhi[f.tomatch[c]];           // detection row of every column
        y7t_sync(ex);
        // linear_assignment(cost, 0.9): entries above the limit can never be matched, so the candidate-list solver sees the same problem
        const int sp = y7t_assoc_sparse_fn(ex, s, n_tl, n_to, 0.9, [&](int c) { return s.left[c]; }, [&](int r) { return s.tmpb[r]; }, [&](int sl, int r, int dj) { return gated_at(y7t_row_at(sl, r), dj); });
        if (sp == 1) {      // (also for the small levels: a handful of candidates, no dense matrix to fill)
            Y7T_CPROF(17);
        } else {
        const int tot = n_tl * n_to;
        for (int k = ex.tid; k < tot; k += ex.nt) {
            const int r = k / n_to, c = k - r * n_to;
            s.cost[(size_t)r * n_to + c] = gated_at(s.tmpb[r], s.left[c]);
        }
        y7t_sync(ex);
        Y7T_CPROF(17);
        y7t_assign_on_cost(ex, s, n_tl, n_to, 0.9, sp == 2);
        }
        Y7T_CPROF(18);
        const int nmatch = y7t_compact(ex, n_tl, [&](int r) { return s.xrow[r] >= 0; }, s.tmpa, 0);
        for (int k = ex.tid; k < nmatch; k += ex.nt) {
            const int r = s.tmpa[k];
            f.casc_tr[nm + k] = s.rem[r];
            f.casc_det[nm + k] = f.tomatch[s.xrow[r]];
        }
        nm += nmatch;
        const int n_left = y7t_compact(ex, n_to, [&](int c) { return s.ycol[c] < 0; }, s.tmpb, 0);
        for (int k = ex.tid; k < n_left; k += ex.nt) f.tmpd[k] = f.tomatch[s.tmpb[k]];
        y7t_sync(ex);
        for (int k = ex.tid; k < n_left; k += ex.nt) f.tomatch[k] = f.tmpd[k];
        y7t_sync(ex);
        n_to = n_left;
        Y7T_CPROF(19);
    }
    // unmatched_tracks = list(set(track_indices) - set(k for k, _ in matches)): CPython set order
    Y7T_PROF(h, 2);
    for (int i = ex.tid; i < n_pool; i += ex.nt) s.mark[i] = 0;
    y7t_sync(ex);
    for (int k = ex.tid; k < nm; k += ex.nt) s.mark[f.casc_tr[k]] = 1;
    y7t_sync(ex);
    {
        const int n_unm = y7t_compact(ex, n_pool, [&](int i) { return !s.mark[i]; }, s.tmpa, 0);      // ascending, built by everyone
        if ((n_pool >> 2) > nm) {                  // set_copy_and_difference: ascending order, nothing serial about it
            for (int k = ex.tid; k < n_unm; k += ex.nt) f.u0[k] = s.tmpa[k];
            if (ex.tid == 0) s.ycol[0] = n_unm;
        } else {
            int need = 8;                          // table entries this call can reach: the power of two above 4 * n_unm
            while (need <= 4 * n_unm) need <<= 1;
            // the serial walk runs out of LDS when it fits: two tables + the list itself
            const bool in_lds = ex.fast && need <= Y7T_PYSET_CAP && ex.fast_bytes >= (size_t)(2 * need + n_unm) * sizeof(int);
            int* tab = in_lds ? (int*)ex.fast : f.pyset;
            const int* unm = s.tmpa;
            if (in_lds) {
                int* l = tab + 2 * need;
                for (int k = ex.tid; k < n_unm; k += ex.nt) l[k] = s.tmpa[k];
                unm = l;
                y7t_sync(ex);
            }
            if (ex.tid == 0) s.ycol[0] = in_lds ? y7t_pyset_difference_list(n_pool, unm, n_unm, nm, f.u0, tab, &f.h->status, need)
                                                : y7t_pyset_difference_list(n_pool, unm, n_unm, nm, f.u0, tab, &f.h->status);
        }
    }
    y7t_sync(ex);
    const int n_u0 = s.ycol[0];
    y7t_sync(ex);
    Y7T_PROF(h, 3);
    // apply the cascade's matches in match order (Tracked -> update, Lost -> re_activate)
    for (int k = ex.tid; k < nm; k += ex.nt) { s.rem[k] = s.pool[f.casc_tr[k]]; s.xrow[k] = f.casc_det[k]; }
    y7t_sync(ex);
    int na, nr;
    y7t_apply_matches(ex, s, s.rem, nm, s.dhi, dets, 0, na, nr);
    y7t_ds_append_features(ex, s, f, s.rem, nm, s.dhi, det_feats);
    Y7T_PROF(h, 4);
    // ---- Step 3: IoU association of the still-Tracked leftovers (in u0 order) with the leftover detections, thresh 0.5 ----
    const int n_t0 = y7t_compact(ex, n_u0, [&](int k) { return s.state[s.pool[f.u0[k]]] == Y7T_TRACKED; }, s.tmpa, 0);
    for (int k = ex.tid; k < n_t0; k += ex.nt) s.rem[k] = s.pool[f.u0[s.tmpa[k]]];
    for (int k = ex.tid; k < n_to; k += ex.nt) s.left[k] = s.dhi[f.tomatch[k]];           // u_dets0 as detection rows
    y7t_sync(ex);
    y7t_gather_track_tlbr(ex, s, s.rem, n_t0);
    y7t_gather_det_tlbr(ex, s, s.left, n_to);
    y7t_sync(ex);
    y7t_assoc(ex, s, n_t0, n_to, 0.5);
    y7t_apply_matches(ex, s, s.rem, n_t0, s.left, dets, 0, na, nr);
    y7t_ds_append_features(ex, s, f, s.rem, n_t0, s.left, det_feats);
    Y7T_PROF(h, 5);
    // u_det1 (detection rows), in column order
    const int n_d1 = y7t_compact(ex, n_to, [&](int c) { return s.ycol[c] < 0; }, s.tmpb, 0);
    for (int k = ex.tid; k < n_d1; k += ex.nt) s.dlo[k] = s.left[s.tmpb[k]];
    // ---- Step 4: `for idx in u_tracks1_idx: track = strack_pool[idx]` (sic, deepsort.py:171-173): the unmatched ROW NUMBERS of u_tracks0
    // index the pool ----
    {
        const int nl_new = y7t_compact(ex, n_t0, [&](int r) { return s.xrow[r] < 0; }, s.tmpa, 0);
        for (int k = ex.tid; k < nl_new; k += ex.nt) s.lostn[k] = s.pool[s.tmpa[k]];
        y7t_sync(ex);
        for (int k = ex.tid; k < nl_new; k += ex.nt) s.state[s.lostn[k]] = Y7T_LOST;
        if (ex.tid == 0) h->n_lostn_last = nl_new;
        y7t_sync(ex);
    }
    // unconfirmed tracks vs u_det1, IoU, thresh 0.9: update only
    y7t_gather_track_tlbr(ex, s, s.unconf, n_unc);
    y7t_gather_det_tlbr(ex, s, s.dlo, n_d1);
    y7t_sync(ex);
    y7t_assoc(ex, s, n_unc, n_d1, 0.9);
    y7t_apply_matches(ex, s, s.unconf, n_unc, s.dlo, dets, 2, na, nr);
    y7t_ds_append_features(ex, s, f, s.unconf, n_unc, s.dlo, det_feats);
    Y7T_PROF(h, 6);
    {
        const int n_rm = y7t_compact(ex, n_unc, [&](int i) { return s.xrow[i] < 0; }, s.tmpa, 0);
        for (int k = ex.tid; k < n_rm; k += ex.nt) { const int sl = s.unconf[s.tmpa[k]]; s.removedl[k] = sl; s.state[sl] = Y7T_REMOVED; }
        if (ex.tid == 0) h->n_removed_last = n_rm;
        y7t_sync(ex);
    }
    // new tracks from u_det2 with score > det_thresh (deepsort.py:197; every detection of the list passed that filter already)
    {
        const int n_new = y7t_compact(ex, n_d1, [&](int j) { return s.ycol[j] < 0 && dets[6 * (size_t)s.dlo[j] + 4] > det_t; }, s.tmpa, 0);
        int* idc = (int*)(uintptr_t)h->id_counter_ptr;
        if (ex.tid == 0) {
            int nf = h->n_free;
            const int base = h->n_act_last, made = n_new < nf ? n_new : nf;
            if (n_new > nf) h->status |= Y7T_ERR_CAP_T;
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
            const int sl = s.tmpb[k], dj = s.dlo[s.tmpa[k]];
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
        // STrack(..., feature=f): features = [f] (the raw vector, basetrack.py:97-103)
        y7t_feat_store_many(ex, f, made, 0, [&](int k, int& sl, int& row) {
            sl = s.tmpb[k];
            row = s.dlo[s.tmpa[k]];
            return true;
        });
    }
    // age out long-lost tracks
    Y7T_PROF(h, 7);
    {
        const int n_old = y7t_compact(ex, nl0, [&](int i) { return frame_id - s.frame[s.lost[i]] > cfg.max_time_lost; }, s.tmpa, 0);
        const int base = h->n_removed_last;
        for (int k = ex.tid; k < n_old; k += ex.nt) { const int sl = s.lost[s.tmpa[k]]; s.removedl[base + k] = sl; s.state[sl] = Y7T_REMOVED; }
        y7t_sync(ex);
        if (ex.tid == 0) h->n_removed_last = base + n_old;
        y7t_sync(ex);
    }
    Y7T_PROF(h, 8);
    y7t_finish(ex, s, out_rows, out_cap, out_count);
    Y7T_PROF(h, 9);
}
