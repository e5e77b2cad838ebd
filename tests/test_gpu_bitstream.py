"""The device's search record through the host bitstream writer and back through the test-side parser:
the stream must decode to exactly the record the GPU produced, and reconstructing from the decoded
record must give the GPU's reconstruction (the reference's integration test, scripts/intergration_test.sh,
with oracle/vvc_parse.cpp + oracle/spec_decoder.cpp -- a decoder written from H.266 that shares no code with
the oracle -- standing in for VTM; the oracle's own wro_reconstruct_from_record is checked alongside)."""
import numpy as np
import pytest

from content import content

pytestmark = pytest.mark.gpu

REC_KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr")


def _decode_and_compare(got, w, h, qp, poc):
    from wrenc_amd import bitstream as bs
    from oracle import pyoracle as po
    stream = bs.write_parameter_sets(w, h, qp) + bs.write_picture(w, h, qp, poc, got)
    bits = bs.last_slice_data_bits()
    back = po.parse_picture(stream, 0)
    assert back["poc_lsb"] == poc & 15 and back["slice_qp"] == qp
    for k in REC_KEYS:
        assert np.array_equal(back[k], got[k]), k
    ry, rcb, rcr = po.reconstruct_from_record(back, qp)
    assert np.array_equal(ry, got["rec_y"]) and np.array_equal(rcb, got["rec_cb"]) and np.array_equal(rcr, got["rec_cr"])
    # and through the decoder written from H.266 alone (oracle/spec_decoder.cpp shares no code with the oracle)
    sy, scb, scr = po.spec_decode_record(back, qp)
    assert np.array_equal(sy, got["rec_y"]) and np.array_equal(scb, got["rec_cb"]) and np.array_equal(scr, got["rec_cr"])
    return stream, bits


@pytest.mark.parametrize("kind,w,h,qp,depth", [
    ("cclm", 128, 64, 32, 2), ("noise", 64, 64, 27, 3), ("stripes20", 96, 64, 22, 3), ("checker", 96, 96, 37, 2),
    ("extremes", 64, 64, 32, 1), ("ramp", 160, 32, 45, 0), ("stripes135", 64, 64, 17, 3),
])
def test_device_record_round_trips_through_the_stream(built, kind, w, h, qp, depth):
    from wrenc_amd import gpu
    y, cb, cr = content(kind, w, h, 21)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    got = enc.encode_picture(y, cb, cr)
    enc.close()
    _decode_and_compare(got, w, h, qp, poc=9)


def test_full_size_picture_round_trips_and_rate_tracks_the_estimate(built):
    """BASELINE.json configs[1] size (1920x1088, QP32, depth 2): one whole picture through the writer and
    the parser.  Also a sanity bound tying the search's rate model to the real CABAC rate: the bits the
    entropy coder produces stay within a factor of two of the level-cost bits the search charged."""
    from wrenc_amd import gpu, synth
    w, h, qp = 1920, 1088, 32
    y, cb, cr = synth.synth_textured_frame(w, h, 1)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=2)
    got = enc.encode_picture(y, cb, cr)
    enc.close()
    stream, bits = _decode_and_compare(got, w, h, qp, poc=0)
    assert 0 < bits <= 8 * len(stream)
    nz = sum(int(np.count_nonzero(got[k])) for k in ("lev_y", "lev_cb", "lev_cr"))
    assert nz > 0 and 0.5 < bits / nz < 40      # a few bits per significant level


@pytest.mark.parametrize("kind,w,h,qp,depth", [("cclm", 128, 64, 32, 2), ("noise", 64, 64, 27, 3), ("stripes20", 96, 64, 22, 1)])
def test_stream_bytes_equal_the_cpu_paths(built, kind, w, h, qp, depth):
    """BASELINE.json's metric: the .vvc written from the device's record is byte-identical to the one written from
    the CPU oracle's record of the same input."""
    from wrenc_amd import bitstream as bs, gpu
    from oracle import pyoracle as po
    y, cb, cr = content(kind, w, h, 33)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    got = enc.encode_picture(y, cb, cr)
    enc.close()
    ref = po.encode_picture(y, cb, cr, qp, depth)
    assert bs.write_picture(w, h, qp, 4, got) == bs.write_picture(w, h, qp, 4, ref)


def test_config1_1080p_stream_byte_exact_vs_cpu(built):
    """BASELINE.json configs[1]: one whole 1920x1088 picture at QP32, max-split-depth 2: the device's record equals the
    CPU oracle's (every plane, every CTU cost) and the two .vvc streams are the same bytes."""
    from wrenc_amd import bitstream as bs, gpu, synth
    from oracle import pyoracle as po
    w, h, qp, depth = 1920, 1088, 32, 2
    y, cb, cr = synth.synth_frame(w, h, 3)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    got = enc.encode_picture(y, cb, cr)
    assert enc.final_pass_mismatches() == 0
    enc.close()
    ref = po.encode_picture(y, cb, cr, qp, depth)
    for k in ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost"):
        assert np.array_equal(got[k], ref[k]), k
    a = bs.write_parameter_sets(w, h, qp) + bs.write_picture(w, h, qp, 0, got)
    b = bs.write_parameter_sets(w, h, qp) + bs.write_picture(w, h, qp, 0, ref)
    assert a == b and len(a) > 10_000
