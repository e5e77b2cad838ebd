"""Kernel-level parity of intra prediction (intra_predictor.rs:56-144, 146-353, 355-757, 759-1602, 1604-2055):
every mode x block size x position of a 3x3-CTU picture (so every availability pattern: picture corner, top
row, left column, right column, bottom row, interior; every position inside a CTU) against the oracle's
Predictor, on planes with content and on planes that make the clips and the i16 wraps bite."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W = H = 96


def _planes(kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return [rng.integers(0, 256, s, dtype=np.uint8) for s in ((H, W), (H // 2, W // 2), (H // 2, W // 2))]
    if kind == "extreme":   # 0 / 255 patches: PDPC and CCLM clips, planar i16 range
        return [(rng.integers(0, 2, (s[0] // 2, s[1] // 2)).repeat(2, 0).repeat(2, 1) * 255).astype(np.uint8)
                for s in ((H, W), (H // 2, W // 2), (H // 2, W // 2))]
    yy, xx = np.indices((H, W))
    y = (128 + 60 * np.sin(xx / 7.0) + 50 * np.cos(yy / 5.0) + rng.normal(0, 6, (H, W))).clip(0, 255).astype(np.uint8)
    cb = (y[::2, ::2].astype(np.int32) // 2 + 40 + rng.integers(-4, 5, (H // 2, W // 2))).clip(0, 255).astype(np.uint8)
    cr = (200 - y[::2, ::2].astype(np.int32) // 3 + rng.integers(-4, 5, (H // 2, W // 2))).clip(0, 255).astype(np.uint8)
    return [y, cb, cr]


def _items(lg, comp):
    """GPU items (x, y, lg, comp, mode) and the oracle's items for the same blocks."""
    n = 1 << lg
    modes = list(range(67)) + ([81, 82, 83] if comp == 1 else [])
    gpu_items, ora_items = [], []
    for y in range(0, H, n):
        for x in range(0, W, n):
            for m in modes:
                gpu_items.append((x, y, lg, comp, m))
                if comp != 1:   # comp 2: the same 4x4 luma block through the packed predictor (predict4_lane)
                    ora_items.append([(x, y, lg, 1 if lg == 2 else 0, 0, m)])
                else:   # the pair: Cb then Cr; an 8x8's 4x4 chroma exists in both tree types
                    ora_items.append([(x, y, lg, 0, 1, m), (x, y, lg, 0, 2, m)])
    return gpu_items, ora_items


@pytest.mark.parametrize("kind", ["smooth", "noise", "extreme"])
@pytest.mark.parametrize("lg,comp", [(5, 0), (4, 0), (3, 0), (2, 0), (2, 2), (5, 1), (4, 1), (3, 1)])
def test_predict_all_modes(built, kind, lg, comp):
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    planes = _planes(kind, 100 * lg + comp)
    gi, oi = _items(lg, comp)
    enc = gpu.Encoder(W, H, qp=32, max_split_depth=3)
    got = enc.predict_blocks(*planes, np.array(gi, np.int32))
    enc.close()
    flat = [q for grp in oi for q in grp]
    ref = po.predict_blocks(*planes, np.array(flat, np.int32))
    at = 0
    for item, g, grp in zip(gi, got, oi):
        r = ref[at:at + len(grp)]
        at += len(grp)
        want = r[0] if comp != 1 else np.stack(r)
        assert np.array_equal(g, want), (kind, item)


def test_predict_dual_tree_chroma_matches_single(built):
    """The chroma block of an 8x8 whose luma was split into 4x4 (DUAL_TREE_CHROMA, ctu.rs:1990-2063) sees the same
    neighbourhood as the single-tree 8x8's chroma: the oracle's two tree chains agree, and the GPU (which has one
    code path for both) equals them."""
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    planes = _planes("smooth", 7)
    gi, single, dual = [], [], []
    for y in range(0, H, 8):
        for x in range(0, W, 8):
            for m in (0, 1, 2, 18, 34, 50, 66, 81, 82, 83):
                gi.append((x, y, 3, 1, m))
                single += [(x, y, 3, 0, 1, m), (x, y, 3, 0, 2, m)]
                dual += [(x, y, 3, 2, 1, m), (x, y, 3, 2, 2, m)]
    a = po.predict_blocks(*planes, np.array(single, np.int32))
    b = po.predict_blocks(*planes, np.array(dual, np.int32))
    enc = gpu.Encoder(W, H, qp=32, max_split_depth=3)
    got = enc.predict_blocks(*planes, np.array(gi, np.int32))
    enc.close()
    for i, g in enumerate(got):
        assert np.array_equal(a[2 * i], b[2 * i]) and np.array_equal(a[2 * i + 1], b[2 * i + 1]), gi[i]
        assert np.array_equal(g, np.stack(a[2 * i:2 * i + 2])), gi[i]


@pytest.mark.parametrize("kind", ["smooth", "noise", "extreme"])
@pytest.mark.parametrize("lg", [5, 4, 3, 2])
def test_sad_lists_of_every_mode_against_the_oracles_predictions(built, kind, lg):
    """The SAD lists of the search (sad_list_angular: one sample per lane, or a 4x4 block of samples per lane with the
    taps of a row shared and packed v_sad_u8) for EVERY angular mode 2..66, in lists of 13 entries (the block-per-lane
    code from 8x8 up), of 2 entries (a step-search round: block-per-lane only for 32x32 luma and 16x16 chroma) and of 6
    (both rounds of a small block at once), luma block and chroma pair, at picture corners, edges and inside: each
    entry's SAD against the block's own samples equals the one computed from the ORACLE's prediction of that mode."""
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    planes = _planes(kind, 300 + lg)
    n = 1 << lg
    comps = 3 if lg >= 3 else 1
    rng = np.random.default_rng(17 * lg)
    blocks = [(x, y) for y in range(0, H, n) for x in range(0, W, n)]
    if len(blocks) > 36:
        corners = [(0, 0), (W - n, 0), (0, H - n), (W - n, H - n), (32, 0), (0, 32), (32, 32), (64, 32 + n)]
        pick = rng.choice(len(blocks), 28, replace=False)
        blocks = corners + [blocks[i] for i in pick]
    lists = [(m0, 13, 5) for m0 in range(2, 7)] + [(m0, 2, 2) for m0 in (2, 17, 33, 34, 49, 64, 65)] + [(9, 6, 1), (30, 6, 1), (61, 6, 1)]
    items = [(x, y, lg, comps, m0, nm, st) for (x, y) in blocks for (m0, nm, st) in lists]
    enc = gpu.Encoder(W, H, qp=32, max_split_depth=3)
    got = enc.sad_lists(*planes, np.array(items, np.int32))
    enc.close()
    # the oracle's predictions of every mode of every block
    ora = []
    for (x, y) in blocks:
        for m in range(2, 67):
            ora.append((x, y, lg, 1 if lg == 2 else 0, 0, m))
            if comps & 2:
                ora.append((x, y, lg, 0, 1, m))
                ora.append((x, y, lg, 0, 2, m))
    preds = po.predict_blocks(*planes, np.array(ora, np.int32))
    per = 3 if comps & 2 else 1
    sad = {}
    for bi, (x, y) in enumerate(blocks):
        oy = planes[0][y:y + n, x:x + n].astype(np.int64)
        ocb = planes[1][y // 2:(y + n) // 2, x // 2:(x + n) // 2].astype(np.int64)
        ocr = planes[2][y // 2:(y + n) // 2, x // 2:(x + n) // 2].astype(np.int64)
        for mi, m in enumerate(range(2, 67)):
            at = (bi * 65 + mi) * per
            v = int(np.abs(oy - preds[at].astype(np.int64)).sum())
            if comps & 2:
                v += int(np.abs(ocb - preds[at + 1].astype(np.int64)).sum()) + int(np.abs(ocr - preds[at + 2].astype(np.int64)).sum())
            sad[(x, y, m)] = v
    for item, row in zip(items, got):
        x, y, _, _, m0, nm, st = item
        for j in range(nm):
            m = m0 + j * st
            want = sad[(x, y, m)] if m <= 66 else 0
            assert int(row[j]) == want, (kind, item, j, m, int(row[j]), want)


@pytest.mark.parametrize("kind", ["smooth", "noise", "extreme"])
@pytest.mark.parametrize("lg", [5, 4, 3])
def test_cclm_sad_lists_against_the_oracles_predictions(built, kind, lg):
    """sad_list_cclm (the three CCLM models' parameters derived side by side, one neighbour position per lane) at every
    block position of a 3x3-CTU picture: the SADs of LT_CCLM / T_CCLM / L_CCLM against the block's own chroma samples
    equal the ones computed from the oracle's CCLM predictions."""
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    planes = _planes(kind, 400 + lg)
    n = 1 << lg
    blocks = [(x, y) for y in range(0, H, n) for x in range(0, W, n)]
    items = [(x, y, lg, 4, 0, 0, 0) for (x, y) in blocks]
    enc = gpu.Encoder(W, H, qp=32, max_split_depth=3)
    got = enc.sad_lists(*planes, np.array(items, np.int32))
    enc.close()
    ora = [(x, y, lg, 0, pc, m) for (x, y) in blocks for m in (81, 83, 82) for pc in (1, 2)]   # LT, T, L
    preds = po.predict_blocks(*planes, np.array(ora, np.int32))
    for bi, ((x, y), row) in enumerate(zip(blocks, got)):
        ocb = planes[1][y // 2:(y + n) // 2, x // 2:(x + n) // 2].astype(np.int64)
        ocr = planes[2][y // 2:(y + n) // 2, x // 2:(x + n) // 2].astype(np.int64)
        for mi in range(3):
            at = (bi * 3 + mi) * 2
            want = int(np.abs(ocb - preds[at].astype(np.int64)).sum()) + int(np.abs(ocr - preds[at + 1].astype(np.int64)).sum())
            assert int(row[mi]) == want, (kind, (x, y, lg), mi, int(row[mi]), want)
