"""The evaluation harness's metrics (wrenc_amd/metrics.py) against the reference's own data: tests/golden/ref_metrics.json
holds what ffmpeg printed per frame for the reference's streams (tools/evaluation/summary.json: psnr.sh / ssim.sh through
evaluate_mp.py:37-120).  The pictures behind them cannot be decoded here, but the RELATIONS between the printed numbers
decide the definitions: psnr_avg comes from the plane-weighted MSE (a mean of the three dB values is off by a dB), SSIM
"All" is the plane-weighted mean, and a QP's summary is the mean over its frames."""
import json
import os

import numpy as np
import pytest

from wrenc_amd import metrics

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref():
    return json.load(open(os.path.join(HERE, "golden", "ref_metrics.json")))["results"]


def test_psnr_avg_is_the_plane_weighted_mse_form(ref):
    n = 0
    worst_db_mean = 0.0
    for r in ref:
        for p in r["metrics"]["PSNR"]["per_frame"]:
            got = metrics.psnr_avg_from_planes(p["Y"], p["U"], p["V"])
            assert abs(got - p["Avg"]) < 0.012, (r["title"], p)     # ffmpeg prints two decimals
            worst_db_mean = max(worst_db_mean, abs((4 * p["Y"] + p["U"] + p["V"]) / 6 - p["Avg"]))
            n += 1
    assert n == 16 * 30
    assert worst_db_mean > 0.5          # ... and the mean of dB values is NOT what the reference reports


def test_bus_qp32_block_of_the_reference_summary(ref):
    r = [x for x in ref if x["title"].startswith("bus_") and x["qp"] == 32][0]
    s = r["metrics"]["PSNR"]["summary"]
    assert (round(s["Y"], 2), round(s["U"], 2), round(s["V"], 2), round(s["Avg"], 2)) == (33.16, 40.27, 40.75, 34.54)
    assert r["bytes"] == 301521
    # the summary is the mean over the frames (evaluate_mp.py:95-110) of per-frame values in the MSE form
    mine = metrics.summarise([{"Avg": metrics.psnr_avg_from_planes(p["Y"], p["U"], p["V"]), "Y": p["Y"], "U": p["U"], "V": p["V"]}
                              for p in r["metrics"]["PSNR"]["per_frame"]])
    assert abs(mine["Avg"] - s["Avg"]) < 0.01 and abs(mine["Y"] - s["Y"]) < 1e-9


def test_ssim_all_is_the_plane_weighted_mean_and_summaries_are_frame_means(ref):
    for r in ref:
        for m in ("PSNR", "SSIM"):
            pf = r["metrics"][m]["per_frame"]
            s = metrics.summarise(pf)
            for k in ("Avg", "Y", "U", "V"):
                assert abs(s[k] - r["metrics"][m]["summary"][k]) < 1e-9
        for p in r["metrics"]["SSIM"]["per_frame"]:
            assert abs((4 * p["Y"] + p["U"] + p["V"]) / 6 - p["Avg"]) < 2e-6   # six printed decimals


def _ssim_plane_loops(a, b):
    """vf_ssim.c, ssim_4x4xn_8bit / ssim_end1 / ssim_plane as plain loops (small planes only)."""
    h, w = a.shape
    bw, bh = w >> 2, h >> 2
    c1, c2 = int(.01 * .01 * 255 * 255 * 64 + .5), int(.03 * .03 * 255 * 255 * 64 * 63 + .5)
    sums = np.zeros((bh, bw, 4), dtype=np.int64)
    for by in range(bh):
        for bx in range(bw):
            s1 = s2 = ss = s12 = 0
            for y in range(4):
                for x in range(4):
                    p, q = int(a[4 * by + y, 4 * bx + x]), int(b[4 * by + y, 4 * bx + x])
                    s1 += p; s2 += q; ss += p * p + q * q; s12 += p * q
            sums[by, bx] = (s1, s2, ss, s12)
    total = 0.0
    for by in range(bh - 1):
        for bx in range(bw - 1):
            s1, s2, ss, s12 = (int(v) for v in (sums[by, bx] + sums[by, bx + 1] + sums[by + 1, bx] + sums[by + 1, bx + 1]))
            var, cov = ss * 64 - s1 * s1 - s2 * s2, s12 * 64 - s1 * s2
            total += float(np.float32(2 * s1 * s2 + c1) * np.float32(2 * cov + c2) /
                           (np.float32(s1 * s1 + s2 * s2 + c1) * np.float32(var + c2)))
    return total / ((bh - 1) * (bw - 1))


def test_ssim_windows_against_the_loop_form():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (24, 40), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-9, 10, a.shape), 0, 255).astype(np.uint8)
    assert abs(metrics.ssim_plane(a, b) - _ssim_plane_loops(a, b)) < 1e-6
    assert abs(metrics.ssim_plane(a, a) - 1.0) < 1e-6
    flat = np.full((16, 16), 100, np.uint8)
    assert abs(metrics.ssim_plane(flat, flat) - 1.0) < 1e-6
    assert metrics.ssim_plane(a, 255 - a) < 0.2


def test_frame_metrics_shape_and_identity():
    rng = np.random.default_rng(6)
    y = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    cb = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    cr = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    m = metrics.frame_metrics((y, cb, cr), (y, cb, cr))
    assert set(m) == {"PSNR", "SSIM"} and set(m["PSNR"]) == {"Avg", "Y", "U", "V"} == set(m["SSIM"])
    assert m["PSNR"]["Y"] == float("inf") and metrics.summarise([m["PSNR"]])["Y"] == 100
    y2 = y.copy(); y2[0, 0] ^= 8
    m2 = metrics.psnr_frame((y, cb, cr), (y2, cb, cr))
    assert abs(m2["Avg"] - 10 * np.log10(255 ** 2 / (64 / 1536))) < 1e-9 and abs(m2["Y"] - 10 * np.log10(255 ** 2 / (64 / 1024))) < 1e-9
