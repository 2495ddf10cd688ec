"""CPU: liby7t.so builds for gfx950, loads, and exports every symbol include/y7t.h declares (no compute calls --
there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from yolov7_tracker_amd import build, _lib
    build.build()
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "y7t.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(y7t_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from yolov7_tracker_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "liby7t.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    for s in _lib.SIGNATURES:
        assert s in syms, "%s bound in _lib.py but not declared in include/y7t.h" % s


def test_version_and_error_string(lib):
    assert lib.y7t_version() >= 100
    assert isinstance(lib.y7t_last_error(), bytes)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import types
    from yolov7_tracker_amd import _lib
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    with pytest.raises(_lib.Y7TError):
        ByteTrack(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5))


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under yolov7-tracker_amd/ may reference it."""
    pkg = os.path.join(ROOT, "yolov7-tracker_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)
                assert "_hostsim" not in txt or f.endswith((".h",)), os.path.join(dp, f)


def test_integration_doc_shows_every_entry_point():
    """INTEGRATION.md is where a maintainer of the reference finds the binding for each C entry point: none may be missing"""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    syms = sorted(set(re.findall(r"\b(y7t_[a-z0-9_]+)\s*\(", open(os.path.join(root, "include", "y7t.h")).read())))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert len(syms) >= 40 and not [s for s in syms if s not in doc]
