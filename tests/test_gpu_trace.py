"""Candidate-level parity: every evaluation the GPU search makes -- the SAD of every listed mode, the
RD cost of every full candidate, every chroma cost -- equals what the oracle's search computed for the
same block, mode and tree type, bit for bit.  Picture-level parity only sees the winners; this also
pins the candidates that lose.

Uses the diagnostic build libwrenc_gpu_trace.so (same sources, -DWRENC_TRACE: the kernel appends a
record per evaluation); the oracle records its own evaluations in the same layout."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def trace_gpu(built):
    from wrenc_amd import gpu
    path = os.path.join(os.path.dirname(gpu.LIB_PATH), "libwrenc_gpu_trace.so")
    assert os.path.exists(path), "run __graft_entry__.build() first"
    saved = (gpu._lib, gpu.LIB_PATH)
    gpu._lib, gpu.LIB_PATH = None, path
    try:
        yield gpu
    finally:
        gpu._lib, gpu.LIB_PATH = saved


def _gpu_trace(gpu, y, cb, cr, qp, depth, schedule):
    h, w = y.shape
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, schedule=schedule)
    fn = enc.lib.wrenc_gpu_trace_read
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    fn(enc.ctx, None, 0)                      # reset
    out = enc.encode_picture(y, cb, cr)
    buf = np.zeros((1 << 19, 8), np.int32)
    n = fn(enc.ctx, buf.ctypes.data_as(C.c_void_p), buf.shape[0])
    enc.close()
    assert 0 < n <= buf.shape[0]
    trace = {}
    for rec in buf[:n].tolist():
        trace.setdefault(tuple(rec[:7]), set()).add(rec[7] & 0xFFFFFFFF)
    return out, trace


@pytest.mark.parametrize("kind,w,h,qp,depth", [
    ("tex", 64, 64, 32, 2), ("tex", 96, 64, 27, 3), ("cclm", 128, 64, 32, 2), ("stripes45", 64, 64, 27, 2),
    ("noise", 64, 64, 37, 3), ("tex", 64, 32, 22, 1), ("tex3", 96, 64, 32, 3), ("tex3", 128, 96, 37, 3),
])
@pytest.mark.parametrize("schedule", [1, 2])    # one wave per CTU / a team of four waves per CTU
def test_every_candidate_cost_matches_oracle(trace_gpu, kind, w, h, qp, depth, schedule):
    from oracle import pyoracle as po
    from wrenc_amd import synth
    if kind.startswith("tex"):
        y, cb, cr = synth.synth_textured_frame(w, h, 3 if kind == "tex3" else 9)
    else:
        from test_gpu_content import _content
        y, cb, cr = _content(kind, w, h, 77)
    got, gtrace = _gpu_trace(trace_gpu, y, cb, cr, qp, depth, schedule)
    ref, otrace = po.encode_picture_traced(y, cb, cr, qp, depth)
    assert np.array_equal(got["ctu_cost"], ref["ctu_cost"])
    kinds = set()
    for key, vals in gtrace.items():
        assert len(vals) == 1, ("GPU evaluated %s twice with different results" % (key,))
        assert key in otrace, ("GPU evaluated a candidate the reference never evaluates: %s" % (key,))
        assert vals == otrace[key], ("candidate %s: GPU %s oracle %s" % (key, vals, otrace[key]))
        kinds.add(key[4])
    assert kinds == {0, 1, 2, 3}
    # the GPU skips only re-evaluations; it must still have seen most distinct candidates
    assert len(gtrace) >= 0.9 * len(otrace), (len(gtrace), len(otrace))
