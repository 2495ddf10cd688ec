"""tic/toc wall-clock timer with the reference's semantics (/root/reference/tracker/timer.py:4-37): this is what the
published "fps" is measured with (tracker/track.py:140,174,181)."""
import time


class Timer(object):
    def __init__(self):
        self.clear()

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        self.duration = self.average_time if average else self.diff
        return self.duration

    def clear(self):
        self.total_time = self.diff = self.average_time = self.duration = self.start_time = 0.
        self.calls = 0
