"""oracle/kalman_np.py -- TEST INFRASTRUCTURE ONLY.

numpy (float64) restatement of the reference's four constant-velocity Kalman filters
(/root/reference/tracker/kalman_filter.py): KalmanFilter :158-411 ('default', xyah),
NaiveKalmanFilter :23-155 ('naive', 7-d), BoTSORTKalmanFilter :414-605 ('botsort',
xywh), NSAKalmanFilter :607-646 ('strongsort').  One table-driven class instead of
four; formulas per SURVEY.md appendix A.  Pinned against the reference's own module in
tests/test_oracle_pinned.py (runs where /root/reference exists) and against
tests/golden/kalman_*.npz everywhere.

dtype note: under numpy>=2 the reference's `initiate` computes its
std terms in float32 when handed a float32 measurement (NEP 50 weak python floats) and
returns a float32 mean; `initiate(..., f32_std=True)` reproduces that.
"""
import numpy as np

SP, SV = 1.0 / 20, 1.0 / 160
KINDS = {"default": 0, "naive": 1, "botsort": 2, "strongsort": 3}


def _F(kind):
    if kind == "naive":
        F = np.eye(7)
        F[0, 4] = F[1, 5] = F[3, 6] = 1.0  # kalman_filter.py:34 couples r (idx 3), not area (idx 2): reference quirk, kept
        return F
    F = np.eye(8)
    for i in range(4):
        F[i, 4 + i] = 1.0
    return F


class KalmanNP:
    def __init__(self, kind="default", motion_mat=None):
        self.kind = kind
        self.d = 7 if kind == "naive" else 8
        self.F = _F(kind) if motion_mat is None else np.asarray(motion_mat, dtype=np.float64)
        self.H = np.eye(4, self.d)

    # ---- noise models -------------------------------------------------
    def _std_init(self, z):
        if self.kind in ("default", "strongsort"):
            h = z[3]
            return [2 * SP * h, 2 * SP * h, 1e-2, 2 * SP * h, 10 * SV * h, 10 * SV * h, 1e-5, 10 * SV * h]
        if self.kind == "botsort":
            w, h = z[2], z[3]
            return [2 * SP * w, 2 * SP * h, 2 * SP * w, 2 * SP * h, 10 * SV * w, 10 * SV * h, 10 * SV * w, 10 * SV * h]
        s = np.sqrt(z[2] * z[3])  # naive: sqrt(area * ratio) = h
        return [2 * SP * s, 2 * SP * s, 2 * SP * s, 1e-5, 10 * SV * s, 10 * SV * s, 10 * SV * s]

    def _std_q(self, mean):
        if self.kind in ("default", "strongsort"):
            h = mean[..., 3]
            o = np.ones_like(h)
            return [SP * h, SP * h, 1e-2 * o, SP * h, SV * h, SV * h, 1e-5 * o, SV * h]
        if self.kind == "botsort":
            w, h = mean[..., 2], mean[..., 3]
            return [SP * w, SP * h, SP * w, SP * h, SV * w, SV * h, SV * w, SV * h]
        s = np.sqrt(mean[..., 2] * mean[..., 3])
        o = np.ones_like(s)
        return [SP * s, SP * s, SP * s, 1e-5 * o, 10 * SV * s, 10 * SV * s, 10 * SV * s]

    def _std_r(self, mean, confidence=0.0):
        if self.kind == "default":
            h = mean[3]
            return [SP * h, SP * h, 1e-1, SP * h]
        if self.kind == "strongsort":
            h = mean[3]
            return [(1 - confidence) * x for x in [SP * h, SP * h, 1e-1, SP * h]]
        if self.kind == "botsort":
            w, h = mean[2], mean[3]
            return [SP * w, SP * h, SP * w, SP * h]
        s = np.sqrt(mean[2] * mean[3])
        return [SP * s, SP * s, 1e-1, SP * s]

    # ---- API mirroring kalman_filter.py ------------------------------------
    def initiate(self, z, f32_std=False):
        z = np.asarray(z)
        if f32_std:
            z32 = z.astype(np.float32)
            std = self._std_init_f32(z32)
            mean = np.r_[z32, np.zeros_like(z32)][: self.d].astype(np.float32)
            if self.kind == "naive":
                mean = np.r_[z32, np.zeros(3, np.float32)].astype(np.float32)
        else:
            std = self._std_init(z.astype(np.float64))
            mean = np.zeros(self.d)
            mean[:4] = z
        if f32_std and self.kind == "botsort":
            # every std entry is np.float32 there, so the reference squares in float32 (cov comes out float32)
            return mean, np.diag(np.square(np.asarray(std, dtype=np.float32)))
        return mean, np.diag(np.square(np.asarray(std, dtype=np.float64)))

    def _std_init_f32(self, z):
        f = np.float32
        if self.kind in ("default", "strongsort"):
            h = z[3]
            a, b = f(f(2 * SP) * h), f(f(10 * SV) * h)
            return [a, a, 1e-2, a, b, b, 1e-5, b]
        if self.kind == "botsort":
            w, h = z[2], z[3]
            return [f(f(2 * SP) * w), f(f(2 * SP) * h), f(f(2 * SP) * w), f(f(2 * SP) * h),
                    f(f(10 * SV) * w), f(f(10 * SV) * h), f(f(10 * SV) * w), f(f(10 * SV) * h)]
        s = np.sqrt(f(z[2] * z[3]))
        a, b = f(f(2 * SP) * s), f(f(10 * SV) * s)
        return [a, a, a, 1e-5, b, b, b]

    def multi_predict(self, means, covs):
        means = np.asarray(means, dtype=np.float64)
        covs = np.asarray(covs, dtype=np.float64)
        q = np.square(np.stack(self._std_q(means), -1))  # (N, d)
        m = means @ self.F.T
        P = self.F @ covs @ self.F.T
        P = P + np.einsum("ni,ij->nij", q, np.eye(self.d))
        return m, P

    def predict(self, mean, cov):
        m, P = self.multi_predict(np.asarray(mean)[None], np.asarray(cov)[None])
        return m[0], P[0]

    def project(self, mean, cov, confidence=0.0):
        mean = np.asarray(mean, dtype=np.float64)
        R = np.diag(np.square(np.asarray(self._std_r(mean, confidence), dtype=np.float64)))
        return self.H @ mean, self.H @ cov @ self.H.T + R

    def update(self, mean, cov, z, confidence=0.0):
        mean = np.asarray(mean, dtype=np.float64)
        cov = np.asarray(cov, dtype=np.float64)
        pm, S = self.project(mean, cov, confidence)
        K = np.linalg.solve(S, (cov @ self.H.T).T).T
        innov = np.asarray(z, dtype=np.float64) - pm
        if self.kind == "naive":  # kalman_filter.py:151-153: P - K (H P)
            return mean + K @ innov, cov - K @ (self.H @ cov)
        return mean + innov @ K.T, cov - K @ S @ K.T

    def gating_distance(self, mean, cov, zs, only_position=False):
        pm, S = self.project(mean, cov)
        zs = np.asarray(zs, dtype=np.float64)
        if only_position:
            pm, S, zs = pm[:2], S[:2, :2], zs[:, :2]
        d = zs - pm
        L = np.linalg.cholesky(S)
        y = np.linalg.solve(L, d.T)
        return np.sum(y * y, axis=0)
