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


from content import content as _content  # noqa: E402


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


@pytest.mark.parametrize("schedule", [1, 2])    # one wave per CTU / a team of four waves per CTU
@pytest.mark.parametrize("kind,w,h,qp,depth", CASES)
def test_content_matches_oracle(built, kind, w, h, qp, depth, schedule):
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    y, cb, cr = _content(kind, w, h, 1234 + w + h + qp)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, schedule=schedule)
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
