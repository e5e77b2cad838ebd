"""BASELINE.json configs[2] / SURVEY.md 8f rank 4: the RD-curve sweep the reference makes with
tools/evaluation/evaluate_mp.py:37-120 -- encode the same pictures at several QPs with the full search, take the
bytes of the stream, DECODE the stream and measure PSNR / SSIM of the decoded pictures per QP.  Here at
3840x2176, QP 22 / 27 / 32 / 37, max-split-depth 3; the decoder is the test-side parser + the spec-derived decoder
(oracle/vvc_parse.cpp, oracle/spec_decoder.cpp) in place of VTM, the metrics are computed from ITS output."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("rd_sweep", os.path.join(ROOT, "tools", "rd_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rd_sweep_2160p(built):
    from oracle import pyoracle as po
    from wrenc_amd import metrics
    rd = _tool()
    w, h, n, qps = 3840, 2176, 2, (22, 27, 32, 37)
    doc = rd.run_sweep(w, h, frames=n, depth=3, qps=qps, threads=8, keep_streams=True, verbose=False)
    res = doc["results"]
    assert [r["qp"] for r in res] == list(qps)
    for r in res:      # the result shape of evaluate_mp.py:78-120
        assert set(("title", "qp", "bytes", "duration", "metrics")) <= set(r)
        for m in ("PSNR", "SSIM"):      # metrics.json: per-frame metrics with the attributes Avg, Y, U, V
            assert set(r["metrics"][m]["summary"]) == {"Avg", "Y", "U", "V"}
            assert len(r["metrics"][m]["per_frame"]) == n
        assert r["final_pass_mismatches"] == 0
        assert r["bytes"] == len(r["_stream"])
        info = po.parse_stream_info(r["_stream"])
        assert (info["width"], info["height"], info["n_pictures"]) == (w, h, n)
        # decode every picture; the metrics of the DECODED pictures are the reported ones (decoded == reconstruction)
        for f in range(n):
            back = po.parse_picture(r["_stream"], f)
            dy, dcb, dcr = po.spec_decode_record(back, r["qp"])
            rec = r["_recs"][f]
            assert np.array_equal(dy, rec["rec_y"]) and np.array_equal(dcb, rec["rec_cb"]) and np.array_equal(dcr, rec["rec_cr"])
            got = metrics.frame_metrics(doc["_frames"][f], (dy, dcb, dcr))
            for m in ("PSNR", "SSIM"):
                pf = r["metrics"][m]["per_frame"][f]
                for k in ("Avg", "Y", "U", "V"):
                    assert abs(got[m][k] - pf[k]) < 1e-9
            p = got["PSNR"]     # Avg is ffmpeg's psnr_avg: from the plane-weighted MSE, not a mean of dB values
            assert abs(metrics.psnr_avg_from_planes(p["Y"], p["U"], p["V"]) - p["Avg"]) < 1e-6
    # an RD curve: rate and quality both fall as QP rises
    b = [r["bytes"] for r in res]
    p = [r["metrics"]["PSNR"]["summary"]["Y"] for r in res]
    s = [r["metrics"]["SSIM"]["summary"]["Y"] for r in res]
    assert b[0] > b[1] > b[2] > b[3] and p[0] > p[1] > p[2] > p[3] and s[0] > s[1] > s[2] > s[3]
    assert 1.2 < b[0] / b[1] < 3.0 and p[0] - p[3] > 5.0 and p[3] > 29.0
