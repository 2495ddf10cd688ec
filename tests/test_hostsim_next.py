"""CPU: placement branches of the tracker programs -- the host build with a fast scratch of the device's size against the oracle on crowded random scenes
(candidate lists on a run-time row stride so that they fit in LDS at 500 objects, csrc/y7t_track_step.h::y7t_assoc_sparse_fn), and with no fast scratch at all."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_candidate_stride_keeps_parity_on_crowded_scenes():
    env = dict(os.environ, Y7T_HOSTSIM_FAST_BYTES="131072")
    env.pop("Y7T_HOSTSIM_DEFS", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_sweep.py"), "bytetrack", "default", "0", "8", "--big"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "seeds 0..8: 0 mismatches" in out.stdout, out.stdout[-2000:]
    short = int(out.stdout.split("short candidate stride ")[1].split(",")[0])
    assert short > 0          # the crowded frames did take the LDS-sized lists


def test_programs_without_fast_scratch_keep_parity():
    """tests/test_hostsim.py runs the programs with a fast scratch of the device's size; the branches that keep the work arrays, cost matrix and candidate
    lists in the state blob instead (what a larger problem falls back to) are run here: every tracker kind on crowded and ordinary scenes"""
    env = dict(os.environ, Y7T_HOSTSIM_FAST_BYTES="0")
    env.pop("Y7T_HOSTSIM_DEFS", None)
    for args in (["bytetrack", "default", "0", "6", "--big"], ["botsort", "botsort", "0", "25"], ["sort", "default", "5", "30"], ["deepsort", "default", "20", "23"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_sweep.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        assert ": 0 mismatches" in out.stdout, out.stdout[-2000:]
