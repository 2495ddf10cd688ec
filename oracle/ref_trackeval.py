"""oracle/ref_trackeval.py -- TEST INFRASTRUCTURE ONLY.

Imports the metric and dataset classes of the reference's vendored TrackEval (/root/reference/tracker/trackeval/) in THIS
container, so that yolov7_tracker_amd.tracker.trackeval can be pinned against them (tests/test_trackeval.py) and golden vectors
generated (tests/golden/make_golden.py).  The package's own `__init__` files import every dataset / metric (pycocotools, PIL,
...), so the packages are registered as bare namespaces and only the needed modules are executed; `np.float` / `np.int` /
`np.bool` (removed in numpy 1.24; 85 uses) are answered by a per-module numpy proxy.  Nothing here exists on the GPU box."""
import importlib
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("Y7T_REFERENCE_ROOT", "/root/reference")
_PKG = os.path.join(REF_ROOT, "tracker", "trackeval")


def available():
    return os.path.isfile(os.path.join(_PKG, "metrics", "hota.py"))


class _OldNumpy:
    float, int, bool = float, int, bool

    def __getattr__(self, name):
        return getattr(np, name)


_ns = None


def load():
    """-> namespace(HOTA, CLEAR, Identity, MotChallenge2DBox, VisDrone2DBox) -- the reference's classes"""
    global _ns
    if _ns is not None:
        return _ns
    if not available():
        raise RuntimeError("reference TrackEval not present at %s" % _PKG)
    names = ["trackeval", "trackeval.metrics", "trackeval.datasets"]
    saved = {k: sys.modules.get(k) for k in names}
    try:
        for name, sub in zip(names, ["", "metrics", "datasets"]):
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(_PKG, sub)]
            sys.modules[name] = m
        sys.dont_write_bytecode = True
        mods = {}
        for name in ["trackeval._timing", "trackeval.utils", "trackeval.metrics._base_metric", "trackeval.metrics.hota",
                     "trackeval.metrics.clear", "trackeval.metrics.identity", "trackeval.datasets._base_dataset",
                     "trackeval.datasets.mot_challenge_2d_box", "trackeval.datasets.visdrone"]:
            sys.modules.pop(name, None)
            mods[name] = importlib.import_module(name)
            if hasattr(mods[name], "np"):
                mods[name].np = _OldNumpy()
        sys.modules["trackeval"]._timing = mods["trackeval._timing"]
        sys.modules["trackeval"].utils = mods["trackeval.utils"]
        mods["trackeval._timing"].DO_TIMING = False
        _ns = types.SimpleNamespace(HOTA=mods["trackeval.metrics.hota"].HOTA, CLEAR=mods["trackeval.metrics.clear"].CLEAR,
                                    Identity=mods["trackeval.metrics.identity"].Identity,
                                    MotChallenge2DBox=mods["trackeval.datasets.mot_challenge_2d_box"].MotChallenge2DBox,
                                    VisDrone2DBox=mods["trackeval.datasets.visdrone"].VisDrone2DBox)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return _ns
