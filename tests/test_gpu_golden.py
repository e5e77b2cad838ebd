"""The committed fixtures tests/golden/*.npz fed to the HIP path DIRECTLY: inputs (y, cb, cr, qp, depth) through the C
ABI, every array of the stored record compared bit for bit (the GPU box then checks the
fixtures themselves, not only the live oracle), in both schedules.  The fixtures are produced by this repository's CPU
restatement (tests/golden/make_golden.py) -- PARITY UNPINNED against the Rust reference, as their generator says."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def test_there_are_fixtures():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("schedule", [1, 2])
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_path_reproduces_the_stored_record(built, path, schedule):
    from wrenc_amd import bitstream as bs, gpu
    g = np.load(path)
    y, cb, cr, qp, depth = g["y"], g["cb"], g["cr"], int(g["qp"]), int(g["depth"])
    h, w = y.shape
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, schedule=schedule)
    got = enc.encode_picture(y, cb, cr)
    assert enc.final_pass_mismatches() == 0
    enc.close()
    for k in KEYS:
        assert np.array_equal(got[k], g[k]), (os.path.basename(path), k)
    # ... and the stream written from the device's record is the stream written from the stored one
    stored = {k: g[k] for k in KEYS}
    assert bs.write_picture(w, h, qp, 0, got) == bs.write_picture(w, h, qp, 0, stored)
