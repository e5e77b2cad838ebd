"""Micro-benchmark of north_star's MFMA question: kernel time of the 32x32 forward and inverse DCT-2 for N blocks x
REPS repetitions, v_dot2 / v_mad_i24 code (what the search kernel ran) against the i8-MFMA versions it runs now
(wrenc_amd/csrc/dev_transform.h: fwd_dct32_mfma, inv_dct32_mfma).  One wave per block, both bit-exact against each other.
    python tools/dct_bench.py [N=16384] [REPS=64]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wrenc_amd import gpu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(1)
blocks = rng.integers(-255, 256, (n, 32, 32)).astype(np.int16)
enc = gpu.Encoder(64, 64, qp=32, max_split_depth=0)
doc = {"blocks": n, "reps": reps}
for direction, fn, data in (("forward", enc.fwd_dct32, blocks),
                            ("inverse", enc.inv_dct32, rng.integers(-4000, 4001, (n, 32, 32)).astype(np.int16))):
    res = {}
    for name, flag in (("vdot2", 0), ("mfma_i8", 1)):
        out, _ = fn(data, flag, 1)          # warm-up, and the output for the equality check
        best = min(fn(data, flag, reps)[1] for _ in range(3))
        res[name] = {"kernel_ms": best, "ns_per_transform": best * 1e6 / (n * reps), "out": out}
    same = bool(np.array_equal(res["vdot2"].pop("out"), res["mfma_i8"].pop("out")))
    doc[direction] = {"outputs_equal": same, **res, "speedup_mfma": res["vdot2"]["kernel_ms"] / res["mfma_i8"]["kernel_ms"]}
print(json.dumps(doc))
enc.close()
