"""Kernel-level parity (bit-exact) of the HIP building blocks against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc(built):
    from wrenc_amd import gpu
    e = gpu.Encoder(64, 64, qp=32, max_split_depth=0)
    yield e
    e.close()


# the quantiser's constants depend on the QP (level scale 16 * LEVEL_SCALE[(qp + 1) % 6] << ((qp + 1) / 6), its 32-bit
# reciprocal, lambda_q): one context per QP, every class of (qp + 1) % 6 and both ends of the range the tables allow
QPS = [18, 22, 27, 30, 31, 33, 34, 35, 37, 45, 51]


@pytest.fixture(scope="module")
def enc_at(built):
    from wrenc_amd import gpu
    made = {}

    def get(qp):
        if qp not in made:
            made[qp] = gpu.Encoder(64, 64, qp=qp, max_split_depth=0)
        return made[qp]
    yield get
    for e in made.values():
        e.close()


def _trellis_blocks(rng, n, count, big):
    """Decaying spectra at four scales, zero blocks, +-3 noise, a lone DC, a lone last coefficient; `big`: coefficients up
    to the i16 range (levels beyond the 1024-entry tables are the caller's to avoid: QP >= 30 keeps them inside)."""
    decay = np.exp(-np.add.outer(np.arange(n), np.arange(n)) / (n / 3.0))
    blocks = []
    for it in range(count):
        scale = [3, 30, 200, 1500][it % 4]
        b = (rng.standard_normal((n, n)) * scale * decay).clip(-32768, 32767).astype(np.int16)
        if it % 7 == 0:
            b[:] = 0
        if it % 11 == 0:
            b = rng.integers(-3, 4, (n, n)).astype(np.int16)
        if it % 13 == 0:
            b[0, 0] = [1, -1, 40, -40][it % 4]
        if it % 17 == 5:
            b[:] = 0
            b[n - 1, n - 1] = [2, -700][it % 2]
        if big and it % 19 == 3:
            b[0, 0] = [32767, -32768][it % 2]
            b[n - 1, 0] = [-32768, 32767][it % 2]
        blocks.append(b)
    return np.stack(blocks)


def _rand_blocks(rng, count, n, scale):
    return (rng.standard_normal((count, n, n)) * scale).clip(-32768, 32767).astype(np.int16)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_fwd_dct(enc, n):
    from oracle import pyoracle as po
    rng = np.random.default_rng(10 + n)
    blocks = rng.integers(-255, 256, (24, n, n)).astype(np.int16)
    blocks[0] = 255
    blocks[1] = -255
    blocks[2] = ((np.indices((n, n)).sum(0) & 1) * 510 - 255).astype(np.int16)
    got = enc.fwd_dct(blocks)
    for i in range(blocks.shape[0]):
        assert np.array_equal(got[i], po.fwd_dct(blocks[i])), (n, i)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_inv_dct(enc, n):
    from oracle import pyoracle as po
    rng = np.random.default_rng(20 + n)
    blocks = _rand_blocks(rng, 24, n, 400)
    blocks[0] = 32767
    blocks[1] = -32768
    got = enc.inv_dct(blocks)
    for i in range(blocks.shape[0]):
        assert np.array_equal(got[i], po.inv_dct(blocks[i])), (n, i)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_dequantize(enc, n):
    from oracle import pyoracle as po
    rng = np.random.default_rng(30 + n)
    blocks = _rand_blocks(rng, 8, n, 60)
    got = enc.dequantize(blocks)
    for i in range(blocks.shape[0]):
        assert np.array_equal(got[i], po.dequantize(blocks[i], 32)), (n, i)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_quantize_trellis(enc, n):
    """Viterbi-on-lanes == the reference's memoised DFS, incl. the level cost walk."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(40 + n)
    decay = np.exp(-np.add.outer(np.arange(n), np.arange(n)) / (n / 3.0))
    blocks = []
    for it in range(40):
        scale = [3, 30, 200, 1500][it % 4]
        b = (rng.standard_normal((n, n)) * scale * decay).clip(-32768, 32767).astype(np.int16)
        if it % 7 == 0:
            b[:] = 0
        if it % 11 == 0:
            b = rng.integers(-3, 4, (n, n)).astype(np.int16)
        if it % 13 == 0:
            b[0, 0] = [1, -1, 40, -40][it % 4]
        blocks.append(b)
    blocks = np.stack(blocks)
    got, cost = enc.quantize(blocks)
    for i in range(blocks.shape[0]):
        ref = po.quantize(blocks[i], 32)
        assert np.array_equal(got[i], ref), (n, i)
        assert int(cost[i]) == po.level_cost(ref), (n, i)


@pytest.mark.parametrize("qp", QPS)
def test_quantize_trellis_at_every_qp_class(enc_at, qp):
    """The lane Viterbi with its head exit, the 32-bit reciprocal quotient (exact for n < 2^26: |tc << sh| <= 2^24 + 256 at
    every size) and the dequantiser at QPs other than the bench's: levels and level cost equal the literal DFS's
    (quantizer.rs:338-517, block_splitter.rs:436-458), coefficients of +-32767 included where the level tables allow."""
    from oracle import pyoracle as po
    e = enc_at(qp)
    rng = np.random.default_rng(4000 + qp)
    for n in (4, 8, 16, 32):
        blocks = _trellis_blocks(rng, n, 24, big=False)
        if qp >= 37:  # 32767 / step stays below 1024 levels from here on (step = lsc / 2^sh grows with the QP)
            blocks[3, 0, 0] = 32767
            blocks[4, n - 1, n - 1] = -32768
        got, cost = e.quantize(blocks)
        deq = e.dequantize(got)
        for i in range(blocks.shape[0]):
            ref = po.quantize(blocks[i], qp)
            assert np.array_equal(got[i], ref), (qp, n, i)
            assert int(cost[i]) == po.level_cost(ref), (qp, n, i)
            assert np.array_equal(deq[i], po.dequantize(ref, qp)), (qp, n, i)


@pytest.mark.parametrize("w,h", [(32, 32), (64, 32), (32, 64), (64, 64), (96, 96), (160, 128), (96, 32), (32, 96)])
def test_segment_availability_table_equals_the_rules(built, w, h):
    """build_refs reads a block's five segment availabilities from a table of its place in the CTU, cut by the picture's
    edges: the same as the reference's rules evaluated directly, for every block of every CTU of pictures one to five CTUs
    wide and high (every combination of left / right / top / bottom edge), luma and chroma."""
    from wrenc_amd import gpu
    e = gpu.Encoder(w, h, qp=32, max_split_depth=0)
    try:
        assert e.test_avail_tab() == 0
    finally:
        e.close()


def test_head_proof_ranges_equal_the_formulas_at_every_qp(built):
    """The head proof reads "ends the region" and "quotient >= 2" off range tests whose bounds the host derives per QP and
    block size (DevConst::head_rng): at every QP 0..63 they say what the device's formulas (head_alpha, quotient) say for
    every 16-bit coefficient, 4 block sizes, the DC position and the others; so does the quotient-free alpha."""
    from wrenc_amd import gpu
    for qp in range(64):
        e = gpu.Encoder(64, 64, qp=qp, max_split_depth=0)
        try:
            counts, ranges = e.test_head_ranges()
        finally:
            e.close()
        assert counts == [0, 0, 0, 0], (qp, counts, ranges.tolist())
        assert (ranges[:, 5] > 0).all(), (qp, ranges.tolist())  # a zero coefficient has quotient 0


@pytest.mark.parametrize("extra", ["quant_lambda_mul_trellis=0.02", "quant_lambda_mul_trellis=60", "quant_qp_div_trellis=3.2",
                                   "quant_lv_pow=0.8,quant_lambda_offset_trellis=9", "quant_lambda_mul_trellis=0"])
def test_head_proof_ranges_never_admit_too_much_under_other_rate_models(built, extra):
    """The CLI's --extra-params change lambda_q and the level-cost table the quantiser works from: whatever they are, a
    range may end a region early (counts[1], costing a longer walk) but never admits a coefficient the formulas reject,
    and the quotient and alpha tests stay exact -- at every QP of five at which the library takes the model at all (it
    refuses one whose step costs do not fit the trellis' 32 bits: tests/test_abi.py)."""
    from wrenc_amd import gpu
    ran = 0
    for qp in (22, 27, 32, 37, 45):
        try:
            e = gpu.Encoder(64, 64, qp=qp, max_split_depth=0, extra_params=extra)
        except gpu.WrencGpuError as err:
            assert "rate model" in str(err), err
            continue
        try:
            counts, ranges = e.test_head_ranges()
        finally:
            e.close()
        assert counts[0] == 0 and counts[2] == 0 and counts[3] == 0, (extra, qp, counts, ranges.tolist())
        ran += 1
    assert ran >= 1, extra


@pytest.mark.parametrize("qp", [18, 27, 34, 45, 51])
def test_packed_quantisers_at_other_qps(enc_at, qp):
    """quantize_p16 and quantize_pk<3 / 4> at other QPs: every block of every pack equals the literal DFS."""
    from oracle import pyoracle as po
    e = enc_at(qp)
    rng = np.random.default_rng(5000 + qp)
    b4 = _trellis_blocks(rng, 4, 41, big=False)
    got, cost = e.quantize_p16(b4)
    for i in range(b4.shape[0]):
        ref = po.quantize(b4[i], qp)
        assert np.array_equal(got[i], ref), (qp, i)
        assert int(cost[i]) == po.level_cost(ref), (qp, i)
    for log2n, nc in ((3, 3), (3, 2), (4, 2), (4, 1)):
        n, nch = 1 << log2n, 1 << (log2n - 1)
        n_packs = 12
        luma = _trellis_blocks(rng, n, n_packs * nc, big=False)
        chroma = _trellis_blocks(rng, nch, n_packs * 2 * nc, big=False)
        packs = [np.concatenate([b.ravel() for b in list(luma[p * nc:(p + 1) * nc]) + list(chroma[p * 2 * nc:(p + 1) * 2 * nc])])
                 for p in range(n_packs)]
        levels, cost = e.quantize_pk(np.stack(packs), log2n, nc)
        for p in range(n_packs):
            at = 0
            for c in range(nc):
                ref = po.quantize(luma[p * nc + c], qp)
                assert np.array_equal(levels[p, at:at + n * n].reshape(n, n), ref), (qp, log2n, nc, p, c)
                assert int(cost[p, c, 0]) == po.level_cost(ref), (qp, log2n, nc, p, c)
                at += n * n
            for c in range(nc):
                want = 0
                for pl in range(2):
                    ref = po.quantize(chroma[p * 2 * nc + 2 * c + pl], qp)
                    assert np.array_equal(levels[p, at:at + nch * nch].reshape(nch, nch), ref), (qp, log2n, nc, p, c, pl)
                    want += po.level_cost(ref)
                    at += nch * nch
                assert int(cost[p, c, 1]) == want, (qp, log2n, nc, p, c)


@pytest.mark.parametrize("count", [1, 2, 3, 4, 5, 203])
def test_quantize_packed_4x4(enc, count):
    """quantize_p16, the quantiser of the packed 4x4 leaf search (up to four blocks per wavefront, one 16-lane row
    each): every block equals the reference's memoised DFS and its level cost, whatever its neighbours in the pack
    are -- zero blocks next to saturated ones, partial packs (count % 4 != 0)."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(900 + count)
    decay = np.exp(-np.add.outer(np.arange(4), np.arange(4)) / 1.5)
    blocks = []
    for it in range(count):
        scale = [3, 30, 200, 1500, 9000][it % 5]
        b = (rng.standard_normal((4, 4)) * scale * decay).clip(-32768, 32767).astype(np.int16)
        if it % 7 == 3:
            b[:] = 0
        if it % 11 == 5:
            b = rng.integers(-3, 4, (4, 4)).astype(np.int16)
        if it % 13 == 6:
            b[:] = 0
            b[0, 0] = [1, -1, 40, -40][it % 4]
        if it % 17 == 9:
            b[:] = 0
            b[3, 3] = [2, -700][it % 2]
        blocks.append(b)
    blocks = np.stack(blocks)
    got, cost = enc.quantize_p16(blocks)
    solo, solo_cost = enc.quantize(blocks)
    for i in range(count):
        ref = po.quantize(blocks[i], 32)
        assert np.array_equal(got[i], ref), (count, i)
        assert int(cost[i]) == po.level_cost(ref), (count, i)
    assert np.array_equal(got, solo) and np.array_equal(cost, solo_cost)


@pytest.mark.parametrize("use_mfma", [0, 1])
def test_fwd_dct32_mfma_experiment(enc, use_mfma):
    """north_star's MFMA question: the 32x32 forward transform as i8 MFMAs over balanced base-256 digits is exact
    (equal to the oracle's transformer.rs:2040-2378 restatement), as is the v_dot2 version the search uses."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(77)
    blocks = rng.integers(-255, 256, (40, 32, 32)).astype(np.int16)
    blocks[0] = 255
    blocks[1] = -255
    blocks[2] = ((np.indices((32, 32)).sum(0) & 1) * 510 - 255).astype(np.int16)
    blocks[3] = 0
    blocks[4] = rng.integers(-3, 4, (32, 32))
    blocks[5] = np.where(np.indices((32, 32))[1] < 16, 255, -255)
    got, ms = enc.fwd_dct32(blocks, use_mfma)
    assert ms > 0
    for i in range(blocks.shape[0]):
        assert np.array_equal(got[i], po.fwd_dct(blocks[i])), (use_mfma, i)


@pytest.mark.parametrize("use_mfma", [0, 1])
def test_inv_dct32_mfma(enc, use_mfma):
    """The inverse 32x32 transform as i8 MFMAs (16-bit operands as a signed high byte and a low byte minus 128, the
    128 * column-sum constant folded into the rounding offset) equals the oracle's transformer.rs:2380-2737
    restatement on the whole i16 input range -- the first stage's clip to 16 bits included -- as does the v_dot2
    version it replaced."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(78)
    blocks = rng.integers(-2000, 2001, (48, 32, 32)).astype(np.int16)
    blocks[0] = 32767
    blocks[1] = -32768
    blocks[2] = ((np.indices((32, 32)).sum(0) & 1) * 65535 - 32768).astype(np.int16)
    blocks[3] = 0
    blocks[4] = rng.integers(-3, 4, (32, 32))
    blocks[5] = np.where(np.indices((32, 32))[1] < 16, 32767, -32768)
    blocks[6] = rng.integers(-32768, 32768, (32, 32))
    blocks[7, 1:, :] = 0          # a single row of large coefficients: the clip of the first stage
    blocks[7, 0, :] = 32767
    blocks[8] = 0
    blocks[8, 0, 0] = -32768
    for i in range(9, 16):
        blocks[i] = rng.integers(-32768, 32768, (32, 32)) * (rng.random((32, 32)) < 0.1)
    got, ms = enc.inv_dct32(blocks, use_mfma)
    assert ms > 0
    for i in range(blocks.shape[0]):
        assert np.array_equal(got[i], po.inv_dct(blocks[i])), (use_mfma, i)


def test_fwd_dct32_refuses_residuals_outside_9_bits(enc):
    """The 32x32 forward transform runs as i8 MFMAs over two base-256 digits of the residual: exact for |r| <= 255,
    which is every residual the search can form.  The public test entries refuse anything else instead of returning
    wrong coefficients (include/wrenc_gpu.h); smaller sizes take the whole i16 range."""
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    blocks = np.zeros((2, 32, 32), np.int16)
    blocks[1, 3, 4] = 256
    for call in (lambda: enc.fwd_dct(blocks), lambda: enc.fwd_dct32(blocks, 1)):
        with pytest.raises(gpu.WrencGpuError) as e:
            call()
        assert e.value.code == -1          # WRENC_GPU_EINVAL
    got, _ = enc.fwd_dct32(blocks, 0)          # the v_dot2 code has no such limit
    assert np.array_equal(got[1], po.fwd_dct(blocks[1]))
    small = np.full((1, 16, 16), -3000, np.int16)
    assert np.array_equal(enc.fwd_dct(small)[0], po.fwd_dct(small[0]))


@pytest.mark.parametrize("log2n,nc", [(3, 1), (3, 2), (3, 3), (4, 1), (4, 2)])
def test_quantize_packs_of_candidates(enc, log2n, nc):
    """quantize_pk, the quantiser of the packed 8x8 / 16x16 leaf searches: the transform blocks of nc candidates (a luma
    block and a Cb + Cr pair each) walked side by side, 16 positions per 16-lane row and round.  Every block equals the
    reference's memoised DFS and its level cost (block_splitter.rs:436-458), whatever rides along in the pack: zero
    blocks next to saturated ones, a single coefficient at either end of the scan, candidates that are all zero."""
    from oracle import pyoracle as po
    n, nch = 1 << log2n, 1 << (log2n - 1)
    rng = np.random.default_rng(1200 + 10 * log2n + nc)
    n_packs = 37

    def block(side, it):
        decay = np.exp(-np.add.outer(np.arange(side), np.arange(side)) / (side / 3.0))
        scale = [3, 30, 200, 1500, 9000][it % 5]
        b = (rng.standard_normal((side, side)) * scale * decay).clip(-32768, 32767).astype(np.int16)
        if it % 7 == 3:
            b[:] = 0
        if it % 11 == 5:
            b = rng.integers(-3, 4, (side, side)).astype(np.int16)
        if it % 13 == 6:
            b[:] = 0
            b[0, 0] = [1, -1, 40, -40][it % 4]
        if it % 17 == 9:
            b[:] = 0
            b[side - 1, side - 1] = [2, -700][it % 2]
        return b

    packs, blocks = [], []
    it = 0
    for p in range(n_packs):
        luma = [block(n, it + c) for c in range(nc)]
        chroma = [block(nch, it + 3 + k) for k in range(2 * nc)]
        it += 9
        if p == 5:                      # a pack with nothing to code at all
            luma = [np.zeros_like(b) for b in luma]
            chroma = [np.zeros_like(b) for b in chroma]
        packs.append(np.concatenate([b.ravel() for b in luma + chroma]))
        blocks.append((luma, chroma))
    levels, cost = enc.quantize_pk(np.stack(packs), log2n, nc)
    for p, (luma, chroma) in enumerate(blocks):
        at = 0
        for c in range(nc):
            ref = po.quantize(luma[c], 32)
            assert np.array_equal(levels[p, at:at + n * n].reshape(n, n), ref), (p, c, "luma")
            assert int(cost[p, c, 0]) == po.level_cost(ref), (p, c, "luma cost")
            at += n * n
        for c in range(nc):
            want = 0
            for pl in range(2):
                ref = po.quantize(chroma[2 * c + pl], 32)
                assert np.array_equal(levels[p, at:at + nch * nch].reshape(nch, nch), ref), (p, c, pl)
                want += po.level_cost(ref)
                at += nch * nch
            assert int(cost[p, c, 1]) == want, (p, c, "chroma cost")
