"""Randomised sweep of the convolution kernels' REAL source on the host simulator (tests/_convsim) against a plain convolution -- more shapes than the test
suite runs, for the kernels that have not been on a GPU yet (the stride-2 LDS-patch kernel in all its forms, the 8-wave instances of the generic kernel,
the DMA-late order of the stride-1 patch kernel).

    python scripts/convsim_sweep.py s2 0 40          # kind (s2 | nw8 | late), first seed, last seed

End of round 2: s2 seeds 0..60 (all four forms x both panel widths, ragged sizes, slices), nw8 seeds 0..40 (four tile / ring shapes, 1x1 and 3x3, stride 1 and 2,
both K orders), late seeds 0..30 (16x16 tiles and the 40-wide strip, three weight orders): 0 mismatches.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _convsim as cs  # noqa: E402
from tests.test_convsim import run_case  # noqa: E402

kind, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L = cs.lib()
bad = []
for seed in range(lo, hi):
    rng = np.random.default_rng(seed * 101 + 7)
    B = int(rng.integers(1, 3))
    Cin = int(rng.choice([64, 128, 192, 256]))
    act = int(rng.integers(0, 3))
    kw = {}
    if rng.random() < 0.4:
        kw.update(in_ld=Cin + 64, in_coff=int(rng.choice([0, 64])))
    try:
        if kind == "s2":
            H, W = int(rng.integers(5, 50)), int(rng.integers(5, 70))
            Cout = int(rng.integers(1, 65)) * 8
            if rng.random() < 0.4:
                kw.update(out_ld=Cout + 64, out_coff=int(rng.choice([0, 64])))
            force = int(rng.choice([0, 8, 16, 24]))
            name = run_case(L, B, H, W, Cin, Cout, 3, 2, act, 0, korder=4, force_patch=force, seed=seed, **kw)
            assert name.startswith("patch_s2<") and ((",8>" in name) == bool(force & 8)) and (name.endswith("dma-late") == bool(force & 16)), name
        elif kind == "nw8":
            H, W = int(rng.integers(5, 30)), int(rng.integers(5, 30))
            Cout = int(rng.integers(1, 5)) * 128
            k = int(rng.choice([1, 3]))
            s = int(rng.choice([1, 2])) if k == 3 else 1
            tile = int(rng.choice([256256564, 256128564, 128128564, 256256532])) if Cout % 256 == 0 else int(rng.choice([256128564, 128128564]))
            name = run_case(L, B, H, W, Cin, Cout, k, s, act, tile, korder=int(k == 3 and rng.random() < 0.5), seed=seed, **kw)
            assert name.endswith("8-wave"), name
        else:
            H, W = int(rng.integers(1, 4)) * 16, int(rng.integers(1, 4)) * 16
            if rng.random() < 0.3:
                H, W = int(rng.integers(10, 44)), 40
            Cout = int(rng.choice([64, 128, 192, 256]))
            name = run_case(L, B, H, W, Cin, Cout, 3, 1, act, 0, korder=int(rng.choice([0, 1, 2])), force_patch=65, seed=seed, **kw)
            assert name.startswith("patch") and (name.endswith("dma-late") or name.startswith("patch<32,8")), name
        print("seed", seed, "ok", name, (B, H, W, Cin, Cout), kw, flush=True)
    except AssertionError as e:
        bad.append(seed)
        print("seed", seed, "MISMATCH", (B, H, W, Cin, Cout), kw, str(e)[:300].replace("\n", " "), flush=True)
print(kind, "seeds %d..%d: %d mismatches %s" % (lo, hi, len(bad), bad))
