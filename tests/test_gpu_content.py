"""Picture-level parity over a sweep of content types, shapes and QPs (bit-exact against the CPU oracle).

The content is chosen to drive every mode family through the search and the final pass: flat areas
(DC / planar, all-zero levels), ramps (planar), stripes at many angles (angular 2..66 incl. the
filtered / PDPC variants), checkerboards and noise (deep splits, many levels), saturated chroma with
luma-correlated chroma (CCLM), and picture shapes that are one CTU wide or one CTU tall (all
availability patterns at the picture edges)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr",
        "ctu_cost")


def _content(kind, w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "flat":
        y = np.full((h, w), 93)
        cb, cr = np.full((h // 2, w // 2), 140), np.full((h // 2, w // 2), 77)
    elif kind == "ramp":
        y = (xx * 2 + yy * 3) % 256
        cb, cr = (xx[::2, ::2] + 60) % 256, (yy[::2, ::2] * 2 + 30) % 256
    elif kind.startswith("stripes"):
        ang = float(kind[7:]) * np.pi / 180.0
        ph = xx * np.cos(ang) + yy * np.sin(ang)
        y = 128 + 90 * np.sign(np.sin(ph * 0.55))
        cb = 128 + 40 * np.sign(np.sin(ph[::2, ::2] * 0.55 + 1.0))
        cr = 128 - 50 * np.sign(np.sin(ph[::2, ::2] * 0.35))
    elif kind == "checker":
        y = 40 + 170 * (((xx // 3) + (yy // 5)) & 1)
        cb = 100 + 60 * (((xx[::2, ::2] // 4) + (yy[::2, ::2] // 2)) & 1)
        cr = 200 - cb // 2
    elif kind == "noise":
        y = rng.integers(0, 256, (h, w))
        cb, cr = rng.integers(0, 256, (h // 2, w // 2)), rng.integers(0, 256, (h // 2, w // 2))
    elif kind == "cclm":     # chroma is an affine function of the (sub-sampled) luma plus a little noise
        y = (128 + 70 * np.sin(xx * 0.21) * np.cos(yy * 0.13) + rng.integers(-6, 7, (h, w))).clip(0, 255)
        ys = y.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
        cb = (0.6 * ys + 40 + rng.integers(-2, 3, ys.shape)).clip(0, 255)
        cr = (220 - 0.7 * ys + rng.integers(-2, 3, ys.shape)).clip(0, 255)
    elif kind == "extremes":  # black / white blocks: clamps in prediction and reconstruction
        y = 255 * (((xx // 16) + (yy // 8)) & 1)
        cb, cr = 255 * ((xx[::2, ::2] // 8) & 1), 255 * ((yy[::2, ::2] // 4) & 1)
    else:
        raise ValueError(kind)
    return (np.ascontiguousarray(y, dtype=np.uint8), np.ascontiguousarray(cb, dtype=np.uint8),
            np.ascontiguousarray(cr, dtype=np.uint8))


CASES = [
    # kind, w, h, qp, depth
    ("flat", 64, 64, 32, 2), ("flat", 64, 32, 22, 3),
    ("ramp", 96, 64, 27, 2), ("ramp", 64, 64, 37, 3),
    ("stripes0", 64, 64, 32, 2), ("stripes90", 64, 64, 32, 2), ("stripes45", 64, 64, 27, 2),
    ("stripes135", 64, 64, 27, 3), ("stripes20", 96, 64, 32, 2), ("stripes70", 64, 96, 32, 2),
    ("stripes110", 64, 64, 22, 2), ("stripes160", 64, 64, 42, 3),
    ("checker", 64, 64, 32, 3), ("checker", 96, 96, 45, 2),
    ("noise", 64, 64, 37, 2), ("noise", 64, 64, 51, 3), ("noise", 32, 32, 27, 3),
    ("cclm", 128, 64, 32, 2), ("cclm", 64, 64, 22, 3), ("cclm", 64, 64, 42, 1),
    ("extremes", 64, 64, 32, 2), ("extremes", 64, 64, 27, 3),
    ("cclm", 32, 256, 32, 2), ("stripes45", 256, 32, 32, 2), ("noise", 32, 128, 40, 3), ("ramp", 160, 32, 32, 0),
]


@pytest.mark.parametrize("kind,w,h,qp,depth", CASES)
def test_content_matches_oracle(built, kind, w, h, qp, depth):
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    y, cb, cr = _content(kind, w, h, 1234 + w + h + qp)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    got = enc.encode_picture(y, cb, cr)
    assert enc.final_pass_mismatches() == 0
    enc.close()
    ref = po.encode_picture(y, cb, cr, qp, depth)
    for k in KEYS:
        if not np.array_equal(got[k], ref[k]):
            bad = np.argwhere(got[k] != ref[k])
            raise AssertionError("%s %dx%d qp%d d%d: %s differs at %d positions, first %s (got %s want %s)" % (
                kind, w, h, qp, depth, k, len(bad), bad[0], got[k][tuple(bad[0])], ref[k][tuple(bad[0])]))
    # the sweep must really exercise the tools it is meant for
    if kind == "cclm":
        assert np.count_nonzero(got["chroma_mode"] >= 81) > 0
    if kind.startswith("stripes"):
        assert np.count_nonzero(got["luma_mode"] >= 2) > 0
