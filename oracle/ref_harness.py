"""oracle/ref_harness.py -- TEST INFRASTRUCTURE ONLY (build-container only).

Imports the REFERENCE's own Python sources from /root/reference so that the oracle
restatements and the committed golden vectors are pinned by the reference itself.
The reference cannot be imported as-is here:
torchvision, cv2, lap, cython_bbox are absent and numpy 2 removed `np.float`.  This
harness therefore

  * shims `np.float = float` (7 uses in tracker/matching.py),
  * registers permissive stub modules for torchvision(.ops/.utils/.transforms), cv2,
    seaborn, reid_models.deepsort_reid.Extractor (constructed, never used, by
    tracker/bytetrack.py:12),
  * injects oracle/cnative.py's restatements as the modules `lap` and `cython_bbox`
    (the only two pieces of third-party ARITHMETIC on the tracker path),

and then imports the reference modules under their own names.  Nothing here is
available on the GPU box (/root/reference does not exist there); GPU-side tests use
the committed fixtures in tests/golden/ and the restatements in oracle/.
"""
import contextlib
import importlib
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("Y7T_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "tracker", "basetrack.py"))


class _Permissive(types.ModuleType):
    """A module whose every attribute is a do-nothing callable/class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        class _Anything:
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                return None

            def __getattr__(self, n):
                return _Anything()

        _Anything.__name__ = name
        setattr(self, name, _Anything)
        return _Anything


def _stub(name, **attrs):
    m = _Permissive(name)
    m.__path__ = []  # behave like a package so submodule imports resolve
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


_TRACKER_MODS = ["kalman_filter", "matching", "basetrack", "bytetrack", "botsort", "deepsort"]
_STUB_NAMES = ["torchvision", "torchvision.ops", "torchvision.utils", "torchvision.transforms", "cv2", "seaborn",
               "reid_models", "reid_models.deepsort_reid", "lap", "cython_bbox", "thop"]


@contextlib.contextmanager
def _patched_modules(extra_paths, cwd=None):
    from . import cnative

    saved = {k: sys.modules.get(k) for k in _STUB_NAMES + _TRACKER_MODS}
    saved_path = list(sys.path)
    saved_cwd = os.getcwd()
    had_float = hasattr(np, "float")
    try:
        if not had_float:
            np.float = float  # noqa: NPY001 - the reference targets numpy<1.24

        class _Extractor:
            def __init__(self, *a, **k):
                pass

        sys.modules["torchvision"] = _stub("torchvision")
        sys.modules["torchvision.ops"] = _stub("torchvision.ops")
        sys.modules["torchvision.utils"] = _stub("torchvision.utils")
        sys.modules["torchvision.transforms"] = _stub("torchvision.transforms")
        sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]
        sys.modules["cv2"] = _stub("cv2")
        sys.modules["seaborn"] = _stub("seaborn")
        sys.modules["reid_models"] = _stub("reid_models")
        sys.modules["reid_models.deepsort_reid"] = _stub("reid_models.deepsort_reid", Extractor=_Extractor)
        lap = types.ModuleType("lap")
        lap.lapjv = cnative.lapjv
        sys.modules["lap"] = lap
        cb = types.ModuleType("cython_bbox")
        cb.bbox_overlaps = cnative.bbox_overlaps
        sys.modules["cython_bbox"] = cb
        for p in reversed(extra_paths):
            sys.path.insert(0, p)
        if cwd:
            os.chdir(cwd)
        sys.dont_write_bytecode = True
        yield
    finally:
        os.chdir(saved_cwd)
        sys.path[:] = saved_path
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        if not had_float and hasattr(np, "float"):
            del np.float


_tracker_ns = None


def load_tracker():
    """-> namespace with the reference's kalman_filter, matching, basetrack, bytetrack modules."""
    global _tracker_ns
    if _tracker_ns is not None:
        return _tracker_ns
    if not available():
        raise RuntimeError("reference sources not present at %s" % REF_ROOT)
    ns = types.SimpleNamespace()
    with _patched_modules([os.path.join(REF_ROOT, "tracker")]):
        for name in _TRACKER_MODS:
            sys.modules.pop(name, None)
        for name in _TRACKER_MODS:
            setattr(ns, name, importlib.import_module(name))
    # The modules stay alive through `ns`; they captured `lap`, `bbox_ious`, `np` at import.
    # matching.py looks up `np.float` at CALL time, so keep a private shim on its numpy handle.
    ns.matching.np = _NumpyWithFloat()
    ns.basetrack.matching = ns.matching
    ns.bytetrack.matching = ns.matching
    ns.botsort.matching = ns.matching
    ns.deepsort.matching = ns.matching
    _tracker_ns = ns
    return ns


class _NumpyWithFloat:
    """Proxy for numpy that also answers `.float` (removed in numpy 1.24)."""
    float = float

    def __getattr__(self, name):
        return getattr(np, name)


def make_opts(**kw):
    o = types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280,
                              iou_thresh=0.5, reid_model_path="", gamma=0.1, min_area=150)
    o.__dict__.update(kw)
    return o


def run_reference_tracker(name, dets_per_frame, opts=None, reset_ids=True, collect_all=False, warps=None, feature_fn=None):
    """Run the reference SORT/ByteTrack over a list of (N,6) float32 arrays.

    Returns per-frame lists of (track_id, tlwh[4] float64, cls, score) for the tracks the
    reference's `update` returns (tracker/track.py:151).
    """
    ns = load_tracker()
    opts = opts or make_opts()
    if reset_ids:
        ns.basetrack.BaseTrack._count = 0
    cls = {"sort": ns.basetrack.BaseTracker, "bytetrack": ns.bytetrack.ByteTrack, "botsort": ns.botsort.BoTSORT,
           "deepsort": ns.deepsort.DeepSORT}[name]
    trk = cls(opts, frame_rate=30, gamma=opts.gamma)
    if name == "deepsort":
        # the ReID network (reid_models/deepsort_reid.py, weights/ckpt.t7 -- not shipped) is replaced at its call site,
        # DeepSORT.get_feature (deepsort.py:19-41), by the deterministic embedding both sides of the parity test use
        trk.get_feature = lambda tlbrs, ori_img, _fn=feature_fn: _fn(tlbrs)
    out = []
    for fi, det in enumerate(dets_per_frame):
        if name == "botsort":
            # GMC.apply (botsort.py:13-248) is OpenCV ORB/RANSAC -- out of scope; feed the frame's synthetic 2x3 warp instead
            w = np.eye(2, 3) if warps is None else np.asarray(warps[fi], dtype=np.float64).reshape(2, 3)
            trk.gmc.apply = (lambda raw_frame, detections=None, _w=w: _w)
        if det is None:
            cur = trk.update_without_detection(None, np.zeros((1, 1, 3), np.uint8))
        else:
            cur = trk.update(np.asarray(det, dtype=np.float32), np.zeros((1, 1, 3), np.uint8))
        rows = [(int(t.track_id), np.asarray(t.tlwh, dtype=np.float64).copy(), float(t.cls), float(t.score))
                for t in cur]
        if collect_all:
            extra = dict(
                tracked=[int(t.track_id) for t in trk.tracked_stracks],
                lost=[int(t.track_id) for t in trk.lost_stracks],
                n_removed=len(trk.removed_stracks),
            )
            out.append((rows, extra))
        else:
            out.append(rows)
    return out


_det_ns = None


def load_detector():
    """-> namespace with the reference's models.yolo (Model), utils.general, utils.torch_utils."""
    global _det_ns
    if _det_ns is not None:
        return _det_ns
    if not available():
        raise RuntimeError("reference sources not present at %s" % REF_ROOT)
    os.environ.setdefault("MPLBACKEND", "Agg")
    ns = types.SimpleNamespace()
    purge = [k for k in list(sys.modules) if k == "models" or k.startswith("models.") or k == "utils" or
             k.startswith("utils.")]
    saved = {k: sys.modules.pop(k) for k in purge}
    try:
        with _patched_modules([REF_ROOT], cwd=REF_ROOT):
            import logging
            logging.disable(logging.INFO)
            ns.yolo = importlib.import_module("models.yolo")
            ns.common = importlib.import_module("models.common")
            ns.general = importlib.import_module("utils.general")
            ns.torch_utils = importlib.import_module("utils.torch_utils")
            logging.disable(logging.NOTSET)
    finally:
        for k in [k for k in list(sys.modules) if k == "models" or k.startswith("models.") or k == "utils" or
                  k.startswith("utils.")]:
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    ns.root = REF_ROOT
    _det_ns = ns
    return ns


def build_reference_model(cfg_rel, nc, ch=3):
    """Instantiate the reference Model from one of ITS yaml files (e.g. 'cfg/deploy/yolov7-w6.yaml')."""
    ns = load_detector()
    cwd = os.getcwd()
    try:
        os.chdir(REF_ROOT)
        import logging
        logging.disable(logging.INFO)
        with _patched_modules([REF_ROOT]):
            m = ns.yolo.Model(cfg_rel, ch=ch, nc=nc)
        logging.disable(logging.NOTSET)
    finally:
        os.chdir(cwd)
    return m.eval()


@contextlib.contextmanager
def reference_pickle_context():
    """While active, the reference's `models.*` / `utils.*` modules are importable by name, so that a reference Model can be pickled with
    torch.save and unpickled with torch.load exactly like the reference's own checkpoints (models/experimental.py:88-89)."""
    ns = load_detector()
    mods = {"models.yolo": ns.yolo, "models.common": ns.common, "utils.general": ns.general, "utils.torch_utils": ns.torch_utils}
    saved = {k: sys.modules.get(k) for k in list(mods) + ["models", "utils"]}
    with _patched_modules([REF_ROOT]):
        try:
            pkg_m, pkg_u = types.ModuleType("models"), types.ModuleType("utils")
            pkg_m.__path__, pkg_u.__path__ = [os.path.join(REF_ROOT, "models")], [os.path.join(REF_ROOT, "utils")]
            pkg_m.yolo, pkg_m.common = ns.yolo, ns.common
            sys.modules["models"], sys.modules["utils"] = pkg_m, pkg_u
            sys.modules.update(mods)
            yield
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
