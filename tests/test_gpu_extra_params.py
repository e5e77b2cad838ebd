"""--extra-params (the reference's RD-model tuning knobs, main.rs:202-217): with the same string given to the
device path and to the oracle, the whole picture result stays bit-identical, and it differs from the result
with the defaults (the knobs really reach the search, the trellis and the chroma cost)."""
import numpy as np
import pytest

from content import content
from test_abi import EXTRA

pytestmark = pytest.mark.gpu

KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")


@pytest.mark.parametrize("kind,w,h,qp,depth,extra", [
    ("cclm", 96, 64, 32, 2, EXTRA), ("noise", 64, 64, 27, 3, EXTRA), ("stripes70", 64, 96, 37, 2, EXTRA),
    ("stripes45", 64, 64, 22, 3, "a=0.05"), ("cclm", 64, 64, 22, 3, "a=3.0"), ("cclm", 64, 64, 32, 2, "quant_lambda_mul_trellis=3.0,quant_lambda_offset_trellis=40"),
    # the quantiser's lambda far from its default: the head proof's ranges (DevConst::head_rng) are derived from it
    ("noise", 64, 64, 32, 3, "quant_lambda_mul_trellis=0.02"), ("noise", 64, 64, 32, 3, "quant_lambda_mul_trellis=60"),
    ("cclm", 96, 64, 22, 2, "quant_lv_pow=0.8,quant_lambda_offset_trellis=9"), ("stripes70", 64, 64, 37, 3, "quant_qp_div_trellis=3.2"),
    # ... and at the edge of what wrenc_gpu_create accepts: lambda_q x dq_table[1023] = 24.9 M and 24.75 M of 25.17 M
    ("noise", 64, 64, 32, 3, "quant_lambda_mul_trellis=86"), ("stripes70", 64, 64, 37, 3, "quant_lambda_mul_trellis=44"),
])
def test_extra_params_parity(built, kind, w, h, qp, depth, extra):
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    y, cb, cr = content(kind, w, h, 17)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, extra_params=extra)
    got = enc.encode_picture(y, cb, cr)
    assert enc.final_pass_mismatches() == 0
    enc.close()
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    plain = enc.encode_picture(y, cb, cr)
    enc.close()
    try:
        po.set_extra_params(extra)
        ref = po.encode_picture(y, cb, cr, qp, depth)
    finally:
        po.set_extra_params(None)
    for k in KEYS:
        assert np.array_equal(got[k], ref[k]), k
    assert any(not np.array_equal(got[k], plain[k]) for k in KEYS)
