#!/usr/bin/env python3
"""Distance of ffmlp_forward on the GPU to the oracle's reference-literal model (running sums rounded to half every 16 k: acc_mode 1), per shape of
tests/test_gpu_ffmlp.py::test_forward_random, relative to the largest output. Run on the GPU box; prints one line per shape."""
import math
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests")]
import oracle  # noqa: E402
import test_gpu_ffmlp as t  # noqa: E402

worst = 0.0
for I, Hd, nl in t.SHAPES:
    rng = np.random.default_rng(7 + I + Hd + nl)
    W = (rng.uniform(-1, 1, t._n_params(I, Hd, nl)) * math.sqrt(3 / Hd)).astype(np.float16)
    x = rng.standard_normal((512, I)).astype(np.float16)
    out, _ = t._run_forward(x, W, I, Hd, nl, 0, True)
    ref = oracle.ffmlp_forward(x, W, I, Hd, nl, 0, training=False, acc_mode=1).astype(np.float32)
    d = np.abs(out.astype(np.float32) - ref).max() / max(1.0, np.abs(ref).max())
    worst = max(worst, d)
    print(f"{I:4d} -> {Hd:3d} x {nl}: max |diff| / max(1, |out|max) = {d:.3e}   (|out|max {np.abs(ref).max():.2f})")
print(f"worst {worst:.3e}")
