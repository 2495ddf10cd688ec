"""HOTA / CLEAR / Identity tracking metrics on the pre-processed per-sequence `data` dict
(num_timesteps, num_gt_ids, num_tracker_ids, num_gt_dets, num_tracker_dets, gt_ids[t], tracker_ids[t], similarity_scores[t]).

Restates what the reference's vendored TrackEval computes for `track.py --track_eval`
(/root/reference/tracker/trackeval/metrics/hota.py:24-118,183-196, clear.py:38-132,168-186, identity.py:31-84,117-135;
call site track.py:196-227) -- same matchings (scipy `linear_sum_assignment` on the same score matrices), same field names.
Pinned against the reference's classes on seeded sequences: tests/golden/trackeval_metrics.json (tests/golden/make_golden.py)."""
import numpy as np
from scipy.optimize import linear_sum_assignment

EPS = np.finfo(float).eps
ALPHAS = np.arange(0.05, 0.99, 0.05)     # the 19 localisation thresholds of HOTA


def _sum(all_res, field):
    return sum(r[field] for r in all_res.values())


class _Metric:
    @classmethod
    def get_name(cls):
        return cls.__name__


class HOTA(_Metric):
    integer_array_fields = ['HOTA_TP', 'HOTA_FN', 'HOTA_FP']
    float_array_fields = ['HOTA', 'DetA', 'AssA', 'DetRe', 'DetPr', 'AssRe', 'AssPr', 'LocA', 'OWTA']
    float_fields = ['HOTA(0)', 'LocA(0)', 'HOTALocA(0)']

    def __init__(self, config=None):
        self.array_labels = ALPHAS
        self.fields = self.float_array_fields + self.integer_array_fields + self.float_fields
        self.summary_fields = self.float_array_fields + self.float_fields

    def eval_sequence(self, data):
        A = len(ALPHAS)
        res = {f: np.zeros(A) for f in self.float_array_fields + self.integer_array_fields}
        res.update({f: 0 for f in self.float_fields})
        if data['num_tracker_dets'] == 0 or data['num_gt_dets'] == 0:
            key, n = ('HOTA_FN', data['num_gt_dets']) if data['num_tracker_dets'] == 0 else ('HOTA_FP', data['num_tracker_dets'])
            res[key] = n * np.ones(A)
            res['LocA'] = np.ones(A)
            res['LocA(0)'] = 1.0
            return res
        G, T = data['num_gt_ids'], data['num_tracker_ids']
        # pass 1: how well could each (gt id, tracker id) pair be aligned over the whole sequence
        potential = np.zeros((G, T))
        n_gt = np.zeros((G, 1))
        n_tr = np.zeros((1, T))
        for g, k, sim in zip(data['gt_ids'], data['tracker_ids'], data['similarity_scores']):
            denom = sim.sum(0)[None, :] + sim.sum(1)[:, None] - sim
            share = np.zeros_like(sim)
            ok = denom > EPS
            share[ok] = sim[ok] / denom[ok]
            potential[g[:, None], k[None, :]] += share
            n_gt[g] += 1
            n_tr[0, k] += 1
        alignment = potential / (n_gt + n_tr - potential)
        matched = np.zeros((A, G, T))
        # pass 2: per-frame assignment that maximises alignment x similarity, thresholded at every alpha
        for g, k, sim in zip(data['gt_ids'], data['tracker_ids'], data['similarity_scores']):
            if len(g) == 0:
                res['HOTA_FP'] += len(k)
                continue
            if len(k) == 0:
                res['HOTA_FN'] += len(g)
                continue
            rows, cols = linear_sum_assignment(-(alignment[g[:, None], k[None, :]] * sim))
            s = sim[rows, cols]
            for a, alpha in enumerate(ALPHAS):
                hit = s >= alpha - EPS
                n = int(hit.sum())
                res['HOTA_TP'][a] += n
                res['HOTA_FN'][a] += len(g) - n
                res['HOTA_FP'][a] += len(k) - n
                if n:
                    res['LocA'][a] += sum(s[hit])
                    matched[a, g[rows[hit]], k[cols[hit]]] += 1
        for a in range(A):
            m = matched[a]
            tp = np.maximum(1, res['HOTA_TP'][a])
            res['AssA'][a] = np.sum(m * (m / np.maximum(1, n_gt + n_tr - m))) / tp
            res['AssRe'][a] = np.sum(m * (m / np.maximum(1, n_gt))) / tp
            res['AssPr'][a] = np.sum(m * (m / np.maximum(1, n_tr))) / tp
        res['LocA'] = np.maximum(1e-10, res['LocA']) / np.maximum(1e-10, res['HOTA_TP'])
        return self._final(res)

    def combine_sequences(self, all_res):
        res = {f: _sum(all_res, f) for f in self.integer_array_fields}
        for f in ['AssRe', 'AssPr', 'AssA']:
            res[f] = sum(r[f] * r['HOTA_TP'] for r in all_res.values()) / np.maximum(1.0, res['HOTA_TP'])
        res['LocA'] = np.maximum(1e-10, sum(r['LocA'] * r['HOTA_TP'] for r in all_res.values())) / np.maximum(1e-10, res['HOTA_TP'])
        return self._final(res)

    @staticmethod
    def _final(res):
        tp, fn, fp = res['HOTA_TP'], res['HOTA_FN'], res['HOTA_FP']
        res['DetRe'] = tp / np.maximum(1, tp + fn)
        res['DetPr'] = tp / np.maximum(1, tp + fp)
        res['DetA'] = tp / np.maximum(1, tp + fn + fp)
        res['HOTA'] = np.sqrt(res['DetA'] * res['AssA'])
        res['OWTA'] = np.sqrt(res['DetRe'] * res['AssA'])
        res['HOTA(0)'] = res['HOTA'][0]
        res['LocA(0)'] = res['LocA'][0]
        res['HOTALocA(0)'] = res['HOTA(0)'] * res['LocA(0)']
        return res


class CLEAR(_Metric):
    integer_fields = ['CLR_TP', 'CLR_FN', 'CLR_FP', 'IDSW', 'MT', 'PT', 'ML', 'Frag', 'CLR_Frames']
    float_fields = ['MOTA', 'MOTP', 'MODA', 'CLR_Re', 'CLR_Pr', 'MTR', 'PTR', 'MLR', 'sMOTA', 'CLR_F1', 'FP_per_frame', 'MOTAL', 'MOTP_sum']

    def __init__(self, config=None):
        self.threshold = float((config or {}).get('THRESHOLD', 0.5))
        self.fields = self.float_fields + self.integer_fields
        self.summary_fields = self.float_fields[:9] + self.integer_fields[:8]

    def eval_sequence(self, data):
        res = {f: 0 for f in self.fields}
        if data['num_tracker_dets'] == 0:
            res.update(CLR_FN=data['num_gt_dets'], ML=data['num_gt_ids'], MLR=1.0)
            return res
        if data['num_gt_dets'] == 0:
            res.update(CLR_FP=data['num_tracker_dets'], MLR=1.0)
            return res
        G = data['num_gt_ids']
        seen = np.zeros(G)
        hit = np.zeros(G)
        spells = np.zeros(G)                   # number of separate tracked stretches of each gt id
        last_id = np.full(G, np.nan)           # tracker id a gt id was last matched to (any time) -> ID switches
        prev_id = np.full(G, np.nan)           # ... in the previous frame only -> continuity bonus, fragmentation
        for g, k, sim in zip(data['gt_ids'], data['tracker_ids'], data['similarity_scores']):
            if len(g) == 0:
                res['CLR_FP'] += len(k)
                continue
            if len(k) == 0:
                res['CLR_FN'] += len(g)
                seen[g] += 1
                continue
            score = 1000 * (k[None, :] == prev_id[g[:, None]]) + sim     # keep last frame's pairing when it is still valid
            score[sim < self.threshold - EPS] = 0
            rows, cols = linear_sum_assignment(-score)
            ok = score[rows, cols] > EPS
            rows, cols = rows[ok], cols[ok]
            mg, mk = g[rows], k[cols]
            before = last_id[mg]
            res['IDSW'] += np.sum(~np.isnan(before) & (mk != before))
            seen[g] += 1
            hit[mg] += 1
            was_untracked = np.isnan(prev_id)
            last_id[mg] = mk
            prev_id[:] = np.nan
            prev_id[mg] = mk
            spells += was_untracked & ~np.isnan(prev_id)
            res['CLR_TP'] += len(mg)
            res['CLR_FN'] += len(g) - len(mg)
            res['CLR_FP'] += len(k) - len(mg)
            if len(mg):
                res['MOTP_sum'] += sum(sim[rows, cols])
        ratio = hit[seen > 0] / seen[seen > 0]
        res['MT'] = np.sum(ratio > 0.8)
        res['PT'] = np.sum(ratio >= 0.2) - res['MT']
        res['ML'] = G - res['MT'] - res['PT']
        res['Frag'] = np.sum(spells[spells > 0] - 1)
        res['CLR_Frames'] = data['num_timesteps']
        return self._final(res)

    def combine_sequences(self, all_res):
        return self._final({f: _sum(all_res, f) for f in self.integer_fields + ['MOTP_sum']})

    @staticmethod
    def _final(res):
        tp, fn, fp, sw = res['CLR_TP'], res['CLR_FN'], res['CLR_FP'], res['IDSW']
        ids = res['MT'] + res['ML'] + res['PT']
        gt = np.maximum(1.0, tp + fn)
        res['MTR'], res['MLR'], res['PTR'] = res['MT'] / np.maximum(1.0, ids), res['ML'] / np.maximum(1.0, ids), res['PT'] / np.maximum(1.0, ids)
        res['CLR_Re'] = tp / gt
        res['CLR_Pr'] = tp / np.maximum(1.0, tp + fp)
        res['MODA'] = (tp - fp) / gt
        res['MOTA'] = (tp - fp - sw) / gt
        res['MOTP'] = res['MOTP_sum'] / np.maximum(1.0, tp)
        res['sMOTA'] = (res['MOTP_sum'] - fp - sw) / gt
        res['CLR_F1'] = tp / np.maximum(1.0, tp + 0.5 * fn + 0.5 * fp)
        res['FP_per_frame'] = fp / np.maximum(1.0, res['CLR_Frames'])
        res['MOTAL'] = (tp - fp - (np.log10(sw) if sw > 0 else sw)) / gt
        return res


class Identity(_Metric):
    integer_fields = ['IDTP', 'IDFN', 'IDFP']
    float_fields = ['IDF1', 'IDR', 'IDP']

    def __init__(self, config=None):
        self.threshold = float((config or {}).get('THRESHOLD', 0.5))
        self.fields = self.float_fields + self.integer_fields
        self.summary_fields = self.fields

    def eval_sequence(self, data):
        res = {f: 0 for f in self.fields}
        if data['num_tracker_dets'] == 0:
            res['IDFN'] = data['num_gt_dets']
            return res
        if data['num_gt_dets'] == 0:
            res['IDFP'] = data['num_tracker_dets']
            return res
        G, T = data['num_gt_ids'], data['num_tracker_ids']
        overlap = np.zeros((G, T))        # frames in which gt id i and tracker id j overlap by >= threshold
        n_gt, n_tr = np.zeros(G), np.zeros(T)
        for g, k, sim in zip(data['gt_ids'], data['tracker_ids'], data['similarity_scores']):
            i, j = np.nonzero(sim >= self.threshold)
            overlap[g[i], k[j]] += 1
            n_gt[g] += 1
            n_tr[k] += 1
        # one global id-to-id assignment; a gt id may instead take its private "unmatched" column (all its boxes are IDFN), a
        # tracker id its private row (all IDFP); the off-diagonal dummies are forbidden
        n = G + T
        fn = np.zeros((n, n))
        fp = np.zeros((n, n))
        fp[G:, :T] = 1e10
        fn[:G, T:] = 1e10
        fn[:G, :T] = n_gt[:, None]
        fn[np.arange(G), T + np.arange(G)] = n_gt
        fp[:G, :T] = n_tr[None, :]
        fp[G + np.arange(T), np.arange(T)] = n_tr
        fn[:G, :T] -= overlap
        fp[:G, :T] -= overlap
        rows, cols = linear_sum_assignment(fn + fp)
        res['IDFN'] = int(fn[rows, cols].sum())
        res['IDFP'] = int(fp[rows, cols].sum())
        res['IDTP'] = int(n_gt.sum() - res['IDFN'])
        return self._final(res)

    def combine_sequences(self, all_res):
        return self._final({f: _sum(all_res, f) for f in self.integer_fields})

    @staticmethod
    def _final(res):
        tp, fn, fp = res['IDTP'], res['IDFN'], res['IDFP']
        res['IDR'] = tp / np.maximum(1.0, tp + fn)
        res['IDP'] = tp / np.maximum(1.0, tp + fp)
        res['IDF1'] = tp / np.maximum(1.0, tp + 0.5 * fp + 0.5 * fn)
        return res


class Count(_Metric):
    integer_fields = ['Dets', 'GT_Dets', 'IDs', 'GT_IDs', 'Frames']

    def __init__(self, config=None):
        self.fields = self.integer_fields
        self.summary_fields = self.fields

    def eval_sequence(self, data):
        return {'Dets': data['num_tracker_dets'], 'GT_Dets': data['num_gt_dets'], 'IDs': data['num_tracker_ids'],
                'GT_IDs': data['num_gt_ids'], 'Frames': data['num_timesteps']}

    def combine_sequences(self, all_res):
        return {f: _sum(all_res, f) for f in self.integer_fields}
