"""PSNR and SSIM as the reference's evaluation harness defines them.

The reference measures with ffmpeg (`tools/evaluation/psnr.sh`: `-lavfi psnr`, `ssim.sh`: `-lavfi ssim`, summarised by
`evaluate_mp.py:37-120`); ffmpeg is not in this image, so the two filters' arithmetic is restated here for 8-bit
4:2:0 planes:

* `psnr`: per plane `10 log10(255^2 / mse)`; `psnr_avg` ("Avg") is taken from the MSE over ALL samples of the frame,
  i.e. `(4 mse_y + mse_u + mse_v) / 6` for 4:2:0 -- not from an average of the three dB values.  The reference's own
  numbers decide between the two: `tests/golden/ref_metrics.json` (from `tools/evaluation/summary.json`).
* `ssim`: 4x4 block sums (s1, s2, ss, s12), a window = 2x2 such blocks (8x8 samples), windows at a stride of 4 samples,
  integer sums with c1 = int(.01^2 255^2 64 + .5), c2 = int(.03^2 255^2 64 63 + .5), the per-window ratio in f32; a
  plane's SSIM is the mean over its (W/4 - 1)(H/4 - 1) windows and "All" = (4 Y + U + V) / 6 (plane areas).

`frame_metrics` returns one per-frame entry in the harness's shape ({"Avg", "Y", "U", "V"} for each metric);
`summarise` averages per-frame entries as `evaluate_mp.py:95-110` does (infinite PSNR becomes 100).
"""
import numpy as np


def plane_mse(a, b):
    d = a.astype(np.int64) - b.astype(np.int64)
    return float(np.sum(d * d)) / d.size


def psnr_from_mse(mse):
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)


def psnr_avg_from_planes(psnr_y, psnr_u, psnr_v):
    """ffmpeg's psnr_avg of a 4:2:0 frame from its three plane PSNRs (the relation the reference's data follow)."""
    mse = [0.0 if p == float("inf") else 255.0 * 255.0 / 10.0 ** (p / 10.0) for p in (psnr_y, psnr_u, psnr_v)]
    return psnr_from_mse((4.0 * mse[0] + mse[1] + mse[2]) / 6.0)


def psnr_frame(org, rec):
    """org, rec: (y, cb, cr) u8 planes of a 4:2:0 frame -> {"Avg", "Y", "U", "V"}."""
    mse = [plane_mse(o, r) for o, r in zip(org, rec)]
    sizes = [o.size for o in org]
    avg = sum(m * n for m, n in zip(mse, sizes)) / float(sum(sizes))
    return {"Avg": psnr_from_mse(avg), "Y": psnr_from_mse(mse[0]), "U": psnr_from_mse(mse[1]), "V": psnr_from_mse(mse[2])}


_C1 = int(.01 * .01 * 255 * 255 * 64 + .5)
_C2 = int(.03 * .03 * 255 * 255 * 64 * 63 + .5)


def ssim_plane(a, b):
    """ffmpeg vf_ssim.c ssim_plane for 8-bit samples."""
    h, w = a.shape
    bw, bh = w >> 2, h >> 2
    if bw < 2 or bh < 2:
        raise ValueError("plane too small for an 8x8 window")
    a = a[:bh * 4, :bw * 4].astype(np.int64)
    b = b[:bh * 4, :bw * 4].astype(np.int64)

    def blocks(x):  # 4x4 block sums
        return x.reshape(bh, 4, bw, 4).sum(axis=(1, 3))

    def windows(x):  # 2x2 blocks = 8x8 samples, stride one block
        return x[:-1, :-1] + x[:-1, 1:] + x[1:, :-1] + x[1:, 1:]

    s1, s2 = windows(blocks(a)), windows(blocks(b))
    ss, s12 = windows(blocks(a * a + b * b)), windows(blocks(a * b))
    var = ss * 64 - s1 * s1 - s2 * s2
    cov = s12 * 64 - s1 * s2
    f = np.float32
    val = (f(1) * (2 * s1 * s2 + _C1).astype(f) * (2 * cov + _C2).astype(f)) / ((s1 * s1 + s2 * s2 + _C1).astype(f) * (var + _C2).astype(f))
    return float(np.sum(val.astype(np.float64)) / ((bh - 1) * (bw - 1)))


def ssim_frame(org, rec):
    v = [ssim_plane(o, r) for o, r in zip(org, rec)]
    sizes = [o.size for o in org]
    return {"Avg": sum(x * n for x, n in zip(v, sizes)) / float(sum(sizes)), "Y": v[0], "U": v[1], "V": v[2]}


def frame_metrics(org, rec):
    return {"PSNR": psnr_frame(org, rec), "SSIM": ssim_frame(org, rec)}


def summarise(per_frame):
    """evaluate_mp.py:95-110: mean of each attribute over the frames; an infinite mean is reported as 100."""
    out = {}
    for attr in ("Avg", "Y", "U", "V"):
        m = sum(p[attr] for p in per_frame) / len(per_frame)
        out[attr] = 100 if m == float("inf") else m
    return out
