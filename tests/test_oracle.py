"""CPU tests of the oracle (test infrastructure): self-consistency, spec
invariants, and the committed golden fixtures.  PARITY UNPINNED (see
oracle/wrenc_oracle.h): the fixtures come from this restatement, not from wrenc."""
import glob
import os
import re

import numpy as np
import pytest

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))


def test_dct_matrix_structure():
    m = po.dct64().astype(np.int64)
    assert np.all(m[0] == 64)
    # rows are near-orthogonal with norm ~ 64*64*... (H.266 8.7.4.5 integer DCT-2)
    g = m @ m.T
    off = g - np.diag(np.diag(g))
    assert np.abs(off).max() <= 0.02 * np.diag(g).min()
    # symmetry B[k][63-n] = (-1)^k B[k][n]
    for k in range(64):
        assert np.array_equal(m[k, ::-1], m[k] * (1 if k % 2 == 0 else -1))


@pytest.mark.skipif(not os.path.exists("/root/reference/src/transformer.rs"), reason="reference not mounted")
def test_dct_matrix_equals_reference_rows_in_use():
    """Rows k*(64/N), N<=32 (the only rows the live config uses) equal the reference's table.
    Row 11 of the reference (64-point only, never used) carries 4 entries copied from row 10."""
    txt = open("/root/reference/src/transformer.rs").read()
    i = txt.index("const TRANS_MATRIX_0_")
    body = txt[i:txt.index("lazy_static!", i)]
    body = body[body.index("= [") + 2:]
    nums = [int(x) for x in re.findall(r"-?\d+", body)][:64 * 32]
    ref = np.array(nums).reshape(64, 32)
    m = po.dct64()
    assert np.array_equal(ref[0::2], m[0::2, :32])
    bad = np.argwhere(ref != m[:, :32])
    assert all(k == 11 for k, _ in bad) and len(bad) == 4


def _ints(text):
    return [int(x) for x in re.findall(r"-?\d+", text)]


def _c_table(path, name):
    txt = open(path).read()
    i = txt.index(name)
    i = txt.index("=", i)
    return _ints(txt[i:txt.index("};", i)])


@pytest.mark.skipif(not os.path.exists("/root/reference/src/common.rs"), reason="reference not mounted")
def test_prediction_tables_equal_the_reference_source():
    """intraPredAngle (common.rs:145), the cubic filter fC (:153) and the Gaussian filter fG (:188), as the
    reference's source text holds them, against the oracle's and the device code's copies."""
    root = os.path.dirname(HERE)
    txt = open("/root/reference/src/common.rs").read()

    def ref_table(name):
        i = txt.index("pub const %s" % name)
        i = txt.index("= [", i)
        return _ints(txt[i:txt.index("];", i)])

    angle, fc, fg = ref_table("INTRA_ANGLE_TABLE"), ref_table("F_C"), ref_table("F_G")
    assert len(angle) == 95 and len(fc) == 128 and len(fg) == 128
    assert fg == [v for p in range(32) for v in (16 - (p >> 1), 32 - (p >> 1), 16 + (p >> 1), p >> 1)]   # both copies compute it
    for path in (os.path.join(root, "oracle", "wrenc_oracle.cpp"), os.path.join(root, "wrenc_amd", "csrc", "wrenc_gpu.hip")):
        assert _c_table(path, "kIntraAngle[95]") == angle, path
        assert _c_table(path, "kFC[32][4]") == fc, path
    ls = open("/root/reference/src/quantizer.rs").read()
    i = ls.index("=", ls.index("const LEVEL_SCALE"))
    assert _ints(ls[i:ls.index("];", i)])[:6] == [40, 45, 51, 57, 64, 72]
    assert _c_table(os.path.join(root, "wrenc_amd", "csrc", "wrenc_gpu.hip"), "level_scale[6]") == [40, 45, 51, 57, 64, 72]
    # dependent-quantisation state machine (encoder_context.rs:339): oracle, host writer, stream parser (the device walk
    # has it folded into its lane permutations, dev_quant.h:295-296, and is covered by the parity tests)
    ec = open("/root/reference/src/encoder_context.rs").read()
    i = ec.index("q_state_trans_table: [[0")
    trans = _ints(ec[i + len("q_state_trans_table:"):ec.index("]],", i)])
    assert trans == [0, 2, 2, 0, 1, 3, 3, 1]
    for rel, name in (("oracle/wrenc_oracle.cpp", "kQStateTrans[4][2]"), ("wrenc_amd/csrc/host/slice_data.cpp", "kQStateTrans[4][2]"),
                      ("oracle/vvc_parse.cpp", "kTrans[4][2]")):
        assert _c_table(os.path.join(root, rel), name) == trans, rel


def test_tables_sanity_values():
    """Values derived from the formulas in SURVEY.md 8c."""
    expect = {22: (34, 9216), 27: (56, 16384), 32: (100, 29184), 37: (184, 52224)}
    for qp, (lq, _ls) in expect.items():
        lv, dq, lam_q, _ = po.tables(qp)
        assert lam_q == lq
        assert list(dq[:6]) == [0, 128, 181, 222, 257, 287]
        assert list(lv[:5]) == [6548, 17546, 23774, 28619, 32720]


def test_trellis_dfs_equals_viterbi():
    rng = np.random.default_rng(1)
    for it in range(400):
        n = [4, 8, 16, 32][it % 4]
        qp = [22, 27, 32, 37, 45, 12][it % 6]
        scale = [3, 30, 200, 1500][(it // 4) % 4]
        decay = np.exp(-np.add.outer(np.arange(n), np.arange(n)) / (n / 3.0))
        c = (rng.standard_normal((n, n)) * scale * decay).astype(np.int16)
        if it % 7 == 0:
            c[:] = 0
        if it % 11 == 0:
            c = rng.integers(-3, 4, (n, n)).astype(np.int16)
        assert np.array_equal(po.quantize(c, qp), po.quantize(c, qp, viterbi=True)), it


def _trellis_block(rng, it):
    n = [4, 8, 16, 32][it % 4]
    scale = [3, 30, 200, 1500][(it // 4) % 4]
    decay = np.exp(-np.add.outer(np.arange(n), np.arange(n)) / (n / [3.0, 1.5, 6.0][(it // 16) % 3]))
    c = (rng.standard_normal((n, n)) * scale * decay).astype(np.int16)
    if it % 7 == 0:
        c[:] = 0
    if it % 11 == 0:
        c = rng.integers(-3, 4, (n, n)).astype(np.int16)
    if it % 13 == 0:  # one stray coefficient at a high frequency: the head ends early
        c[rng.integers(n // 2, n), rng.integers(n // 2, n)] = rng.integers(-400, 400)
    if it % 17 == 0:  # only DC and its neighbours
        c[2:, :] = 0
        c[:, 2:] = 0
    return c


@pytest.mark.parametrize("qp", [18, 22, 27, 30, 32, 33, 34, 35, 37, 41, 45, 51])
def test_trellis_shortcuts_equal_the_literal_dfs(qp):
    """The device quantiser's exits (round 4) -- the head proven zero without walking it, an all-quotient-zero sub-block
    in closed form -- modelled on the CPU (oracle: quantize_viterbi_sc) give the literal memoised DFS's levels
    (quantizer.rs:338-517) at every QP class ((qp + 1) % 6 = 0..5), each exit alone and both together; so does the
    walk of the long linear part of a chain as four segments from (min, +) basis vectors (proven here for a later round)."""
    rng = np.random.default_rng(100 + qp)
    for it in range(240):
        c = _trellis_block(rng, it)
        want = po.quantize(c, qp)
        for head, z in ((True, False), (False, True), (True, True)):
            got = po.quantize_sc(c, qp, head, z)
            assert np.array_equal(got, want), (qp, it, head, z)
        if c.shape[0] >= 16:    # the long linear part of a chain as four segments side by side (not in the kernel yet)
            for head in (False, True):
                assert np.array_equal(po.quantize_sc(c, qp, head, False, True), want), (qp, it, head, "segments")


@pytest.mark.parametrize("extra", ["quant_lambda_mul_trellis=0.02", "quant_lambda_mul_trellis=60", "quant_lambda_mul_trellis=0",
                                   "quant_qp_div_trellis=3.2", "quant_lv_pow=0.8,quant_lambda_offset_trellis=9", "quant_lv_pow=0.3"])
def test_trellis_shortcuts_under_other_rate_models(extra):
    """The same comparison with the quantiser's lambda and level-cost table moved far from their defaults (--extra-params):
    the proofs make no assumption about their size, only about their sign -- whole-block proof on and off."""
    try:
        po.set_extra_params(extra)
        for qp in (22, 32, 37):
            rng = np.random.default_rng(700 + qp)
            for it in range(120):
                c = _trellis_block(rng, it)
                want = po.quantize(c, qp)
                assert np.array_equal(po.quantize_sc(c, qp, True, True), want), (extra, qp, it)
                assert np.array_equal(po.quantize_sc(c, qp, True, False, False, False), want), (extra, qp, it, "no whole-block proof")
    finally:
        po.set_extra_params(None)


def test_trellis_shortcuts_on_the_search_own_blocks():
    """... and on every block the search itself quantises (smooth and textured content, two QPs); the exits do fire."""
    from wrenc_amd import synth
    for qp, fn in ((32, synth.synth_frame), (37, synth.synth_textured_frame), (24, synth.synth_textured_frame)):
        y, cb, cr = fn(1920, 1088, 3)
        y, cb, cr = y[64:128, 128:256].copy(), cb[32:64, 64:128].copy(), cr[32:64, 64:128].copy()
        po.dq_sc_stats_enable(True)
        po.encode_picture(y, cb, cr, qp, 2)
        mism, st = po.dq_sc_stats_read()
        po.dq_sc_stats_enable(False)
        assert mism == 0
        assert st[4]["blocks"] > 0 and st[5]["blocks"] > 0
        if qp == 32:
            assert st[5]["head_sb_skipped"] > st[5]["sub_blocks"] // 2


def _diag(n):
    out, x, y = [], 0, 0
    while len(out) < n * n:
        while y >= 0:
            if x < n and y < n:
                out.append((x, y))
            y -= 1
            x += 1
        y, x = x, 0
    return out


def test_dep_quant_parity_matches_state():
    """The reference's release-mode assert (ctu_encoder.rs:1975-1978): walking the levels in
    reverse scan order, every non-zero TransCoeffLevel has parity == (state > 1)."""
    rng = np.random.default_rng(5)
    trans = [[0, 2], [2, 0], [1, 3], [3, 1]]
    checked = 0
    for it in range(60):
        n = [4, 8, 16, 32][it % 4]
        c = (rng.standard_normal((n, n)) * [40, 400, 3000][it % 3]).clip(-32768, 32767).astype(np.int16)
        lev = po.quantize(c, [22, 32, 37][it % 3])
        sb = _diag(n // 4)
        co = _diag(4)
        state = 0
        for xs, ys in reversed(sb):
            for xc, yc in reversed(co):
                q = abs(int(lev[ys * 4 + yc, xs * 4 + xc]))
                if q:
                    assert (q & 1) == int(state > 1)
                    checked += 1
                a = (q + int(state > 1)) // 2
                state = trans[state][a & 1]
    assert checked > 1000


def test_dct_round_trip_small_error():
    rng = np.random.default_rng(7)
    for n in (4, 8, 16, 32):
        r = rng.integers(-255, 256, (n, n)).astype(np.int16)
        c = po.fwd_dct(r)
        # forward output is scaled by 1/ (n/ ... ) such that dequantised coefficients invert;
        # without quantisation inverse(coef << 0) is not the identity, so only check linearity:
        c2 = po.fwd_dct((-r).astype(np.int16))
        assert np.abs(c.astype(int) + c2.astype(int)).max() <= 1


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))))
def test_golden_fixture(path):
    g = np.load(path)
    out = po.encode_picture(g["y"], g["cb"], g["cr"], int(g["qp"]), int(g["depth"]))
    assert out["final_pass_mismatches"] == 0
    for k in ("rec_y", "rec_cb", "rec_cr", "lev_y", "lev_cb", "lev_cr", "cu_log2_size", "luma_mode",
              "chroma_mode", "ctu_cost"):
        assert np.array_equal(out[k], g[k]), k
    # decoder-side reconstruction of the record equals the encoder's reconstruction
    ry, rcb, rcr = po.reconstruct_from_record(out, int(g["qp"]), int(g["depth"]))
    assert np.array_equal(ry, out["rec_y"]) and np.array_equal(rcb, out["rec_cb"]) and np.array_equal(rcr, out["rec_cr"])
    sy, scb, scr = po.spec_decode_record(out, int(g["qp"]))
    assert np.array_equal(sy, out["rec_y"]) and np.array_equal(scb, out["rec_cb"]) and np.array_equal(scr, out["rec_cr"])


def test_edge_cases_single_ctu_and_flat():
    flat = np.full((32, 32), 128, np.uint8)
    c = np.full((16, 16), 128, np.uint8)
    out = po.encode_picture(flat, c, c, 32, 3)
    assert out["final_pass_mismatches"] == 0
    assert np.array_equal(out["rec_y"], flat)          # first block: all refs 128 -> exact
    assert not out["lev_y"].any()
    assert np.all(out["cu_log2_size"] == 5)            # split never wins on a flat CTU
    with pytest.raises(ValueError):
        po.encode_picture(np.zeros((30, 32), np.uint8), c, c, 32, 0)   # not a multiple of 32
    with pytest.raises(ValueError):
        po.encode_picture(flat, c, c, 32, 4)                            # depth out of range


def test_extreme_content():
    """Max-contrast checkerboard: large levels at QP12 still index the 1024-entry tables; at
    QP0 a table index reaches 1024, where the reference panics (block_splitter.rs:453) and the
    oracle reports failure."""
    yy, xx = np.indices((32, 32))
    y = (((xx // 2 + yy // 2) & 1) * 255).astype(np.uint8)
    cb = (((xx[:16, :16] + yy[:16, :16]) & 1) * 255).astype(np.uint8)
    out = po.encode_picture(y, cb, cb, 12, 2)
    assert out["final_pass_mismatches"] == 0
    assert 1024 < np.abs(out["lev_y"]).max() < 2046
    with pytest.raises(ValueError):
        po.encode_picture(y, cb, cb, 0, 2)


def test_reciprocal_quotient_is_exact_at_every_qp():
    """The device divides by the level scale with one 32-bit multiply-high (wrenc_gpu.hip fill_dev_const: m = floor(2^k /
    lsc) + 1, k = 26 + ceil(log2 lsc), q = mulhi(n, m) >> (k - 32)).  n = |(tc << sh) - off| <= (32768 << 9) + 256 < 2^26 at
    every block size whatever the QP; the quotient is exact there for every level scale QP 0..63 gives: checked at every
    multiple of lsc +- 1 below 2^26 (where a reciprocal first fails) and at random n."""
    scale = [40, 45, 51, 57, 64, 72]
    rng = np.random.default_rng(5)
    for qp in range(64):
        lsc = (16 * scale[(qp + 1) % 6]) << ((qp + 1) // 6)
        lg = 0
        while (1 << lg) < lsc:
            lg += 1
        k = 26 + lg
        m = (1 << k) // lsc + 1
        assert m < 1 << 32 and k >= 32
        q = np.arange(1, (1 << 26) // lsc + 1, dtype=np.uint64) * np.uint64(lsc)
        n = np.concatenate([q - np.uint64(1), q, q + np.uint64(1), rng.integers(0, 1 << 26, 20000).astype(np.uint64),
                            np.array([0, 1, (1 << 26) - 1, (32768 << 9) + 256], np.uint64)])
        n = n[n < (1 << 26)]
        got = ((n * np.uint64(m)) >> np.uint64(32)) >> np.uint64(k - 32)
        assert np.array_equal(got, n // np.uint64(lsc)), qp
