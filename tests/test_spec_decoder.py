"""The independent pixel-domain decoder (oracle/spec_decoder.cpp, written from H.266 clauses 6.4.4, 8.4.5.2.8-15,
8.7.3-8.7.5 and sharing no code with the oracle) against the encoder's reconstruction.

The reference's only end-to-end test is "VTM's decode of the stream == --reconst"
(scripts/intergration_test.sh:1-15 of the reference); VTM is not available here.  The chain below is the
strongest statement available in this image: stream --(vvc_parse.cpp: CABAC decoding written from the
decoding direction)--> record --(spec_decoder.cpp)--> samples == the encoder's reconstruction.  The
perturbation tests show it is not vacuous: an oracle that misreads one rounding offset stays
self-consistent (its own decoder-side reconstruction still agrees) and is caught by the spec decoder."""
import numpy as np
import pytest

from content import content
from test_bitstream import _random_record, _roundtrip

PLANES = (("rec_y", 0), ("rec_cb", 1), ("rec_cr", 2))


def _assert_decodes(rec, qp, record=None):
    from oracle import pyoracle as po
    out = po.spec_decode_record(record if record is not None else rec, qp)
    for (k, i) in PLANES:
        assert np.array_equal(out[i], rec[k]), k


def test_transform_matrix_is_the_standards(built):
    """The decoder's transMatrix (built from the even/odd families of 8.7.4.5) equals the oracle's table, which
    tests/test_oracle.py pins to the reference's transformer.rs:934-1191."""
    from oracle import pyoracle as po
    assert np.array_equal(po.spec_trans_matrix(), po.dct64())


CASES = [
    ("flat", 64, 64, 32, 2), ("ramp", 96, 64, 27, 3), ("stripes0", 64, 64, 32, 2), ("stripes20", 96, 64, 22, 3),
    ("stripes45", 64, 64, 32, 3), ("stripes65", 64, 64, 37, 2), ("stripes90", 64, 64, 32, 2), ("stripes110", 64, 96, 27, 3),
    ("stripes135", 64, 64, 17, 3), ("stripes160", 64, 64, 42, 1), ("checker", 96, 96, 37, 2), ("noise", 64, 64, 27, 3),
    ("noise", 64, 64, 51, 3), ("noise", 32, 32, 4, 3), ("noise", 64, 32, 63, 1), ("cclm", 128, 64, 32, 2),
    ("cclm", 64, 64, 22, 3), ("cclm", 64, 64, 42, 1), ("extremes", 64, 64, 32, 2), ("extremes", 64, 64, 18, 3),
    ("cclm", 32, 256, 32, 2), ("stripes45", 256, 32, 32, 2), ("noise", 32, 128, 40, 3), ("ramp", 160, 32, 32, 0),
]


@pytest.mark.parametrize("kind,w,h,qp,depth", CASES)
def test_spec_decoder_reproduces_the_encoders_reconstruction(built, kind, w, h, qp, depth):
    """Through the stream: the oracle's record -> host writer -> parser -> spec decoder == the oracle's rec planes."""
    from oracle import pyoracle as po
    y, cb, cr = content(kind, w, h, 11)
    rec = po.encode_picture(y, cb, cr, qp, depth)
    _, back = _roundtrip(rec, w, h, qp, poc=3)
    _assert_decodes(rec, qp, record=back)


@pytest.mark.parametrize("frame,w,h,qp,depth", [(0, 128, 96, 32, 2), (5, 96, 96, 22, 3), (9, 160, 64, 37, 3)])
def test_spec_decoder_on_textured_pictures(built, frame, w, h, qp, depth):
    from wrenc_amd import synth
    from oracle import pyoracle as po
    y, cb, cr = synth.synth_textured_frame(w, h, frame)
    rec = po.encode_picture(y, cb, cr, qp, depth)
    assert len(np.unique(rec["luma_mode"])) > 8 and np.count_nonzero(rec["chroma_mode"] >= 81) > 0
    _assert_decodes(rec, qp)


@pytest.mark.parametrize("seed,qp,amp", [(1, 32, 60), (2, 22, 400), (3, 37, 4000), (4, 12, 30000), (5, 51, 30000),
                                        (6, 27, 8), (7, 0, 2000), (8, 63, 32000), (9, 32, 200), (10, 40, 1000)])
def test_two_decoders_agree_on_records_no_search_would_emit(built, seed, qp, amp):
    """Every luma mode 0..66 at every size, DM / CCLM / explicit chroma modes, levels up to the i16 range
    (clips of 8.7.3 and of the first transform stage): the oracle's decoder-side reconstruction (the reference's
    predictor / dequantiser / inverse transform restated) and the spec decoder produce the same samples."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(seed)
    w, h = 96, 96
    rec = _random_record(rng, w, h, qp, amp)
    a = po.reconstruct_from_record(rec, qp)
    b = po.spec_decode_record(rec, qp)
    for i in range(3):
        assert np.array_equal(a[i], b[i]), i
    assert len(np.unique(rec["luma_mode"])) > 30


@pytest.mark.parametrize("which,kind", [(1, "stripes20"), (2, "cclm"), (3, "noise"), (4, "noise")])
def test_a_misread_constant_in_the_oracle_is_caught(built, which, kind):
    """wro_debug_perturb makes the oracle misread one constant (PDPC rounding, CCLM down-sampling rounding,
    a level-scale entry, inverse-transform first-stage offset).  The perturbed oracle still agrees with
    itself -- exactly the blind spot of tests that compare the GPU with the oracle -- and the spec decoder
    reports the difference."""
    from oracle import pyoracle as po
    w, h, qp, depth = 96, 64, 27, 2
    y, cb, cr = content(kind, w, h, 5)
    po.debug_perturb(which)
    try:
        rec = po.encode_picture(y, cb, cr, qp, depth)
        own = po.reconstruct_from_record(rec, qp, depth)
    finally:
        po.debug_perturb(0)
    assert all(np.array_equal(own[i], rec[k]) for k, i in PLANES)          # self-consistent
    spec = po.spec_decode_record(rec, qp)
    assert not all(np.array_equal(spec[i], rec[k]) for k, i in PLANES)     # ... and wrong
    clean = po.encode_picture(y, cb, cr, qp, depth)                        # the unperturbed oracle is fine
    _assert_decodes(clean, qp)


def test_rejects_records_that_are_not_quadtrees(built):
    from oracle import pyoracle as po
    y, cb, cr = content("noise", 64, 64, 1)
    rec = po.encode_picture(y, cb, cr, 32, 2)
    bad = {k: v.copy() for k, v in rec.items() if isinstance(v, np.ndarray)}
    bad["cu_log2_size"][0, 0] = 6
    with pytest.raises(ValueError):
        po.spec_decode_record(bad, 32)
    bad["cu_log2_size"][0, 0] = rec["cu_log2_size"][0, 0]
    bad["luma_mode"][0, 0] = 70
    with pytest.raises(ValueError):
        po.spec_decode_record(bad, 32)
