"""Evaluation harness behind `track.py --track_eval` (HOTA / CLEAR / Identity on MOTChallenge-format files): the subset of the
reference's vendored TrackEval that its CLI drives (/root/reference/tracker/track.py:196-227), restated; pinned in
tests/test_trackeval.py against the reference's own classes."""
from . import datasets, metrics
from .eval import Evaluator

__all__ = ["Evaluator", "datasets", "metrics"]
