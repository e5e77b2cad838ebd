"""Regenerates tests/golden/*.npz.

PROVENANCE: these vectors are produced by THIS repository's CPU restatement
(oracle/wrenc_oracle.cpp) of the reference algorithm, not by the reference
binary -- the Rust reference cannot be built in this image and ships no golden
vectors for the path.  They pin the oracle against accidental
change and give the GPU path a fixed target; they do not pin parity with wrenc.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import pyoracle as po  # noqa: E402
from wrenc_amd import synth  # noqa: E402

CASES = [
    # name, width, height, qp, depth, textured, frame
    ("tex64_qp32_d3", 64, 64, 32, 3, 1, 1),
    ("tex96x64_qp27_d2", 96, 64, 27, 2, 1, 2),
    ("smooth64_qp37_d1", 64, 64, 37, 1, 0, 0),
    ("tex64_qp22_d0", 64, 64, 22, 0, 1, 5),
    # the reference's own test geometry (scripts/intergration_test.sh:6: CIF, QP 20, default depth 3), synthetic content
    ("cif_qp20_d3", 352, 288, 20, 3, 1, 7),
]

if __name__ == "__main__":
    for name, w, h, qp, depth, tex, frame in CASES:
        y, cb, cr = (synth.synth_textured_frame if tex else synth.synth_frame)(w, h, frame)
        out = po.encode_picture(y, cb, cr, qp, depth)
        assert out.pop("final_pass_mismatches") == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), y=y, cb=cb, cr=cr,
                            qp=np.int32(qp), depth=np.int32(depth), **out)
        print(name, "written")
