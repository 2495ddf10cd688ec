"""Drop-in entry point: `python tracker/track.py ...` from the repo root, like the reference's tracker/track.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd.tracker.track import cli  # noqa: E402

if __name__ == '__main__':
    cli()
