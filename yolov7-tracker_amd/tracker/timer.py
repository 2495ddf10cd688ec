"""tic/toc wall-clock timer with the reference's attribute surface (/root/reference/tracker/timer.py:4-37): this is what the
published "fps" is measured with (tracker/track.py:140,174,181 read `total_time` and call `tic` / `toc` / `clear`)."""
import time

_FIELDS = ("total_time", "diff", "average_time", "duration", "start_time")


class Timer(object):
    """`toc()` returns the running mean of the tic->toc intervals (or the last interval with average=False)"""

    def __init__(self):
        self.clear()

    def clear(self):
        for name in _FIELDS:
            setattr(self, name, 0.0)
        self.calls = 0

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        now = time.time()
        self.diff = now - self.start_time
        self.calls += 1
        self.total_time += self.diff
        self.average_time = self.total_time / self.calls
        self.duration = self.average_time if average else self.diff
        return self.duration
