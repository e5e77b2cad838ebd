"""Parity at BASELINE.json's full sizes through size-independent properties:
decoder-side reconstruction of the GPU's record == the GPU's reconstruction (the in-repo form of
the reference's integration test), the final pass reproduces the search's reconstruction, the
top CTU rows equal the oracle's encode of the same rows, identical inputs give identical
outputs, and out-of-range levels are reported (the reference panics there)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr")


def test_1080p_depth2_properties(built):
    from wrenc_amd import gpu, synth
    from oracle import pyoracle as po
    w, h, qp, depth = 1920, 1088, 32, 2
    y, cb, cr = synth.synth_textured_frame(w, h, 7)
    y2, cb2, cr2 = synth.synth_frame(w, h, 2)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=3)
    enc.upload(0, y, cb, cr)
    enc.upload(1, y2, cb2, cr2)
    enc.upload(2, y, cb, cr)          # same input as slot 0, different wave of the workgroup
    enc.encode(0, 3)
    enc.sync()
    assert enc.final_pass_mismatches() == 0
    a, b, c = enc.download(0), enc.download(1), enc.download(2)
    enc.close()
    for k in KEYS + ("ctu_cost",):
        assert np.array_equal(a[k], c[k]), k                       # determinism across slots / waves
    for rec in (a, b):
        ry, rcb, rcr = po.reconstruct_from_record(rec, qp, depth)  # decoder-side reconstruction
        assert np.array_equal(ry, rec["rec_y"])
        assert np.array_equal(rcb, rec["rec_cb"])
        assert np.array_equal(rcr, rec["rec_cr"])
        sy, scb, scr = po.spec_decode_record(rec, qp)             # ... and by the independent spec decoder
        assert np.array_equal(sy, rec["rec_y"]) and np.array_equal(scb, rec["rec_cb"]) and np.array_equal(scr, rec["rec_cr"])
    # the first two CTU rows depend on nothing below them: they must equal the oracle on the crop
    ref = po.encode_picture(y[:64], cb[:32], cr[:32], qp, depth)
    for k in KEYS:
        scale = {"cu_log2_size": 4, "luma_mode": 4, "chroma_mode": 8, "rec_cb": 2, "rec_cr": 2, "lev_cb": 2,
                 "lev_cr": 2}.get(k, 1)
        assert np.array_equal(a[k][:64 // scale], ref[k]), k
    assert np.array_equal(a["ctu_cost"][:2 * (w // 32)], ref["ctu_cost"])
    assert len(np.unique(a["cu_log2_size"])) >= 2 and np.count_nonzero(a["chroma_mode"] >= 81) > 0


def _top_rows_equal_oracle(rec, y, cb, cr, qp, depth, rows=32):
    """CTU rows depend on nothing below them: the top `rows` luma rows must equal the oracle's encode of the crop."""
    from oracle import pyoracle as po
    w = y.shape[1]
    ref = po.encode_picture(y[:rows], cb[:rows // 2], cr[:rows // 2], qp, depth)
    for k, scale in (("cu_log2_size", 4), ("luma_mode", 4), ("chroma_mode", 8), ("lev_y", 1), ("lev_cb", 2), ("lev_cr", 2),
                     ("rec_y", 1), ("rec_cb", 2), ("rec_cr", 2)):
        assert np.array_equal(rec[k][:rows // scale], ref[k]), k
    assert np.array_equal(rec["ctu_cost"][:(rows // 32) * (w // 32)], ref["ctu_cost"])


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_2160p_depth3_decoder_check(built, qp):
    """BASELINE.json configs[2]: 3840x2176, the four QPs of the RD sweep, full search (max-split-depth 3)."""
    from wrenc_amd import gpu, synth
    from oracle import pyoracle as po
    w, h, depth = 3840, 2176, 3
    y, cb, cr = synth.synth_textured_frame(w, h, 11)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    rec = enc.encode_picture(y, cb, cr)
    assert enc.final_pass_mismatches() == 0
    enc.close()
    ry, rcb, rcr = po.reconstruct_from_record(rec, qp, depth)
    assert np.array_equal(ry, rec["rec_y"]) and np.array_equal(rcb, rec["rec_cb"]) and np.array_equal(rcr, rec["rec_cr"])
    sy, scb, scr = po.spec_decode_record(rec, qp)             # the decoder that shares no code with the oracle
    assert np.array_equal(sy, rec["rec_y"]) and np.array_equal(scb, rec["rec_cb"]) and np.array_equal(scr, rec["rec_cr"])
    _top_rows_equal_oracle(rec, y, cb, cr, qp, depth, rows=64)
    assert set(np.unique(rec["cu_log2_size"])) >= {2, 3, 4}      # the 8x8 -> 4x4 local dual tree is exercised
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((rec["rec_y"].astype(np.float64) - y) ** 2))
    assert psnr > {22: 38.0, 27: 35.0, 32: 32.0, 37: 29.0}[qp]


def test_level_overflow_is_reported(built):
    """A level that would index the reference's 1024-entry tables out of range is an error
    (WRENC_GPU_ELEVEL), exactly where the oracle reports the reference's panic."""
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    yy, xx = np.indices((32, 32))
    y = (((xx + yy) & 1) * 255).astype(np.uint8)
    c = (((xx[:16, :16] + yy[:16, :16]) & 1) * 255).astype(np.uint8)
    with pytest.raises(ValueError):
        po.encode_picture(y, c, c, 0, 2)
    enc = gpu.Encoder(32, 32, qp=0, max_split_depth=2)
    with pytest.raises(gpu.WrencGpuError) as ei:
        enc.encode_picture(y, c, c)
    assert ei.value.code == -6
    enc.close()


@pytest.mark.parametrize("depth,tex", [(3, 1), (2, 0)])
def test_4320p_decoder_check(built, depth, tex):
    """BASELINE.json configs[4] (7680x4320 QP32 max-split-depth 3; 32400 CTUs) and the depth-2 variant:
    decoder-side reconstruction of the record equals the encoder's, the final pass reproduces the search, and
    the picture's first CTU row equals the oracle's encode of that row alone (a CTU row depends on nothing
    below it)."""
    from wrenc_amd import gpu, synth
    from oracle import pyoracle as po
    w, h, qp = 7680, 4320, 32
    y, cb, cr = (synth.synth_textured_frame if tex else synth.synth_frame)(w, h, 5)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    rec = enc.encode_picture(y, cb, cr)
    assert enc.final_pass_mismatches() == 0
    enc.close()
    ry, rcb, rcr = po.reconstruct_from_record(rec, qp, depth)
    assert np.array_equal(ry, rec["rec_y"]) and np.array_equal(rcb, rec["rec_cb"]) and np.array_equal(rcr, rec["rec_cr"])
    sy, scb, scr = po.spec_decode_record(rec, qp)
    assert np.array_equal(sy, rec["rec_y"]) and np.array_equal(scb, rec["rec_cb"]) and np.array_equal(scr, rec["rec_cr"])
    _top_rows_equal_oracle(rec, y, cb, cr, qp, depth, rows=32)
    if depth == 3:
        assert set(np.unique(rec["cu_log2_size"])) >= {2, 3, 4}


@pytest.mark.parametrize("w,h,depth", [(3840, 2176, 3), (1920, 1088, 2)])
def test_schedules_agree_at_full_size(built, w, h, depth):
    """One wavefront per CTU (ctu_search_kernel) and the level schedule (four wavefronts per CTU, one tree level each,
    ctu_search_team_kernel) produce the same record for whole pictures of BASELINE's sizes: every plane, mode map and f32
    CTU cost -- with real anti-diagonals (up to 60 CTUs), pictures of different content in one workgroup, a padding
    wave; and the stream written from either record is the same."""
    from wrenc_amd import bitstream as bs, gpu, synth
    qp = 32
    frames = [synth.synth_textured_frame(w, h, 3), synth.synth_frame(w, h, 4), synth.synth_textured_frame(w, h, 5)]
    recs = {}
    for schedule in (1, 2):
        enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=3, schedule=schedule)
        for s, f in enumerate(frames):
            enc.upload(s, *f)
        enc.encode(0, 3)
        enc.sync()
        assert enc.final_pass_mismatches() == 0
        assert enc.last_schedule() == schedule
        recs[schedule] = [enc.download(s) for s in range(3)]
        enc.close()
    for s in range(3):
        for k in KEYS + ("ctu_cost",):
            assert np.array_equal(recs[1][s][k], recs[2][s][k]), (s, k)
    assert bs.write_picture(w, h, qp, 0, recs[1][0]) == bs.write_picture(w, h, qp, 0, recs[2][0])


@pytest.mark.parametrize("schedule", [1, 2])
def test_a_full_gpu_keeps_its_scratch_behind_the_xcds_own_l2(built, schedule):
    """Enough pictures in flight to have every CU hold its five workgroups: each workgroup takes its scratch region
    (the saved reconstructions of the search) from the partition of the XCD it runs on, none has to fall back to the
    shared overflow partition (include/wrenc_gpu.h, wrenc_gpu_test_scratch_overflows), the final pass reproduces the
    search everywhere, and the same input gives the same record in the first and in the last slot."""
    from wrenc_amd import gpu, synth
    w, h, qp, depth, n = 1920, 1088, 32, 2, 96
    frames = [synth.synth_textured_frame(w, h, 7), synth.synth_frame(w, h, 2)]
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=n, schedule=schedule)
    for s in range(n):
        enc.upload(s, *frames[0 if s in (0, n - 1) else 1 - (s & 1)])
    for _ in range(2):
        enc.encode(0, n)
    enc.sync()
    assert enc.final_pass_mismatches() == 0
    assert enc.test_scratch_overflows() == 0
    a, b = enc.download(0), enc.download(n - 1)
    enc.close()
    for k in KEYS + ("ctu_cost",):
        assert np.array_equal(a[k], b[k]), k
