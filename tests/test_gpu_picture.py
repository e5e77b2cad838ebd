"""Picture-level parity: HIP search + final pass == CPU oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr",
        "ctu_cost")


def _compare(got, ref, tag):
    for k in KEYS:
        if not np.array_equal(got[k], ref[k]):
            bad = np.argwhere(got[k] != ref[k])
            raise AssertionError("%s: %s differs at %d positions, first %s (got %s want %s)" % (
                tag, k, len(bad), bad[0], got[k][tuple(bad[0])], ref[k][tuple(bad[0])]))


@pytest.mark.parametrize("w,h,qp,depth,tex", [
    (32, 32, 32, 0, 0),
    (64, 64, 32, 0, 1),
    (64, 64, 32, 1, 1),
    (64, 64, 27, 2, 1),
    (96, 64, 32, 3, 1),
    (128, 96, 22, 2, 0),
    (128, 96, 37, 3, 1),
])
@pytest.mark.parametrize("schedule", [1, 2])    # one wave per CTU / a team of four waves per CTU
def test_picture_matches_oracle(built, w, h, qp, depth, tex, schedule):
    from wrenc_amd import gpu, synth
    from oracle import pyoracle as po
    y, cb, cr = (synth.synth_textured_frame if tex else synth.synth_frame)(w, h, 3)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, schedule=schedule)
    got = enc.encode_picture(y, cb, cr)
    assert enc.final_pass_mismatches() == 0 and enc.last_schedule() == schedule
    enc.close()
    ref = po.encode_picture(y, cb, cr, qp, depth)
    _compare(got, ref, "%dx%d qp%d d%d" % (w, h, qp, depth))


@pytest.mark.parametrize("kind,w,h,qp,depth", [("noise", 64, 64, 63, 3), ("cclm", 96, 64, 63, 2), ("extremes", 64, 64, 63, 3),
                                               ("stripes70", 64, 64, 60, 3), ("noise", 64, 64, 57, 3), ("cclm", 96, 64, 4, 2),
                                               ("noise", 64, 64, 12, 3)])
@pytest.mark.parametrize("schedule", [1, 2])
def test_both_ends_of_the_qp_range(built, kind, w, h, qp, depth, schedule):
    """QP 57 .. 63 -- where lambda_q x dq_table comes closest to what the trellis' 32-bit path costs cover (22.6 M of 25.2 M at
    QP 63) -- and QP 4 / 12, where levels are large (noise at QP 0 reaches level 1024: WRENC_GPU_ELEVEL, as in the oracle)."""
    from content import content
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    y, cb, cr = content(kind, w, h, 17)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, schedule=schedule)
    got = enc.encode_picture(y, cb, cr)
    assert enc.final_pass_mismatches() == 0
    enc.close()
    _compare(got, po.encode_picture(y, cb, cr, qp, depth), "%s qp%d d%d" % (kind, qp, depth))


@pytest.mark.parametrize("w,h,qp,depth", [(512, 32, 32, 3), (32, 512, 32, 3), (2048, 32, 27, 1), (32, 1024, 37, 2)])
def test_one_row_and_one_column_pictures(built, w, h, qp, depth):
    """A single row of CTUs (every anti-diagonal holds one CTU, nothing above it) and a single column (every CTU is a
    picture's left AND right edge, two diagonals apart): two pictures per call in all three schedules, the record against
    the oracle and the device's token stream against the stream from the planes."""
    from wrenc_amd import bitstream as bs, gpu, synth
    from oracle import pyoracle as po
    frames = [synth.synth_textured_frame(w, h, 11), synth.synth_frame(w, h, 2)]
    refs = [po.encode_picture(*f, qp, depth) for f in frames]
    for schedule in (0, 1, 2):
        enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=2, schedule=schedule)
        for s, f in enumerate(frames):
            enc.upload(s, *f)
        enc.encode(0, 2)
        enc.sync()
        assert enc.final_pass_mismatches() == 0
        pool, pics = enc.download_tokens(0, 2)
        for s in range(2):
            got = enc.download(s)
            _compare(got, refs[s], "%dx%d schedule %d slot %d" % (w, h, schedule, s))
            assert bs.write_picture(w, h, qp, s, got) == bs.write_picture_tokens(w, h, qp, s, pool, pics[s])
        enc.close()


@pytest.mark.parametrize("schedule", [0, 1, 2])    # AUTO / wave / team
def test_the_reference_own_test_geometry(built, schedule):
    """wrenc's only end-to-end test encodes CIF 352x288 at QP 20 with the default max-split-depth 3
    (scripts/intergration_test.sh:6 of the reference): two 11 x 9-CTU pictures of that geometry (one textured, one
    smooth; synthetic: the bus clip is not in the image) in one encode call, every array of the record against the
    oracle, and the stream written from it decodes -- parser + the independent spec decoder -- to the reconstruction."""
    from wrenc_amd import bitstream as bs, gpu, synth
    from oracle import pyoracle as po
    w, h, qp, depth = 352, 288, 20, 3
    frames = [synth.synth_textured_frame(w, h, 7), synth.synth_frame(w, h, 2)]
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=2, schedule=schedule)
    for s, (y, cb, cr) in enumerate(frames):
        enc.upload(s, y, cb, cr)
    enc.encode(0, 2)
    enc.sync()
    assert enc.final_pass_mismatches() == 0
    for s, (y, cb, cr) in enumerate(frames):
        got = enc.download(s)
        _compare(got, po.encode_picture(y, cb, cr, qp, depth), "CIF picture %d" % s)
        stream = bs.write_parameter_sets(w, h, qp) + bs.write_picture(w, h, qp, s, got)
        back = po.parse_picture(stream, 0)
        for a, k in zip(po.spec_decode_record(back, qp), ("rec_y", "rec_cb", "rec_cr")):
            assert np.array_equal(a, got[k]), (s, k)
    enc.close()


@pytest.mark.parametrize("schedule", [1, 2])
def test_batch_of_pictures(built, schedule):
    """Several pictures in flight in one encode call give the same result as one by one."""
    from wrenc_amd import gpu, synth
    from oracle import pyoracle as po
    w, h, qp, depth = 96, 64, 32, 2
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=3, schedule=schedule)
    frames = [synth.synth_textured_frame(w, h, f) for f in range(3)]
    for s, (y, cb, cr) in enumerate(frames):
        enc.upload(s, y, cb, cr)
    enc.encode(0, 3)
    enc.sync()
    for s, (y, cb, cr) in enumerate(frames):
        _compare(enc.download(s), po.encode_picture(y, cb, cr, qp, depth), "slot %d" % s)
    enc.close()
