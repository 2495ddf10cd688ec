"""CPU: tracker/timer.py against the reference's Timer (tracker/timer.py:4-37) -- the clock the published fps is read from (track.py:140,174,181) --
driven by the same fake clock: every attribute after every call must be equal."""
import importlib.util
import os

import pytest

from yolov7_tracker_amd.tracker import timer as ours

REF = "/root/reference/tracker/timer.py"
FIELDS = ("total_time", "calls", "start_time", "diff", "average_time", "duration")


class _Clock:
    def __init__(self, ticks):
        self.ticks, self.i = list(ticks), 0

    def time(self):
        t = self.ticks[self.i]
        self.i += 1
        return t


def _drive(mod, monkeypatch, script, ticks):
    clock = _Clock(ticks)
    monkeypatch.setattr(mod.time, "time", clock.time)
    t = mod.Timer()
    trace = [tuple(getattr(t, f) for f in FIELDS)]
    for op in script:
        ret = {"tic": t.tic, "toc": t.toc, "toc_last": lambda: t.toc(average=False), "clear": t.clear}[op]()
        trace.append((ret,) + tuple(getattr(t, f) for f in FIELDS))
    return trace


SCRIPT = ["tic", "toc", "tic", "toc", "tic", "toc_last", "clear", "tic", "toc", "toc"]
TICKS = [10.0, 10.5, 11.0, 11.25, 20.0, 20.125, 30.0, 30.75, 31.0]


def test_timer_semantics(monkeypatch):
    tr = _drive(ours, monkeypatch, SCRIPT, TICKS)
    assert tr[2][0] == 0.5 and tr[4][0] == 0.375 and tr[6][0] == 0.125            # running mean, running mean, last interval
    assert tr[7][1:] == (0.0, 0, 0.0, 0.0, 0.0, 0.0)                              # clear
    assert tr[-1][2] == 2 and tr[-1][1] == 0.75 + 1.0                             # toc twice after one tic: both intervals from the same start


def test_timer_equals_reference(have_reference, monkeypatch):
    if not have_reference or not os.path.isfile(REF):
        pytest.skip("/root/reference not present")
    spec = importlib.util.spec_from_file_location("ref_timer", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    assert _drive(ours, monkeypatch, SCRIPT, TICKS) == _drive(ref, monkeypatch, SCRIPT, TICKS)
