"""Synthetic picture content shared by the parity tests: chosen to drive every mode family (see
test_gpu_content.py)."""
import numpy as np


def content(kind, w, h, seed):
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
