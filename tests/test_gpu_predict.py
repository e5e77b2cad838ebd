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
