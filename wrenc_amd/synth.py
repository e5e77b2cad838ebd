"""Deterministic synthetic YUV420 8-bit pictures.

Integer-only (no libm) so that the same frame index gives the same bytes on every
machine: Y = 128 + 64*sine(3x/W + 2y/H + f/30) + 32*value-noise(8x8 grid) + noise,
Cb/Cr = 128 +/- 32*gradient + noise.  Residuals stay small enough that quantised
levels remain far below the reference's 1024-entry tables (block_splitter.rs:453).
"""
import numpy as np

SEED_BASE = 0x5EED0000
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """splitmix64 finaliser, vectorised over uint64 arrays."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _hash2d(seed, plane, h, w, salt):
    ys, xs = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64), indexing="ij")
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0x100000001B3)
               + np.uint64(plane) * np.uint64(0x9E3779B1)
               + np.uint64(salt) * np.uint64(0x85EBCA77)
               + ys * np.uint64(0x1F123BB5) * np.uint64(65537) + xs * np.uint64(0xC2B2AE3D))
    return _splitmix64(key)


def _isin(phase):
    """Parabolic integer sine: phase in [0, 65536) -> [-32768, 32768]."""
    half = phase & 32767
    v = (half * (32768 - half)) >> 13
    return np.where(phase & 32768, -v, v)


def _noise_sum(seed, plane, h, w, amp):
    """Sum of four uniform integers in [-amp, amp] (Irwin-Hall ~ gaussian)."""
    r = _hash2d(seed, plane, h, w, 7)
    span = np.uint64(2 * amp + 1)
    total = np.zeros((h, w), dtype=np.int64)
    for k in range(4):
        part = (r >> np.uint64(16 * k)) & np.uint64(0xFFFF)
        total += (part % span).astype(np.int64) - amp
    return total


def synth_frame(width, height, frame):
    """Return (y, cb, cr) uint8 arrays of shapes (H, W), (H/2, W/2), (H/2, W/2)."""
    assert width % 32 == 0 and height % 32 == 0
    seed = SEED_BASE + frame
    ys, xs = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    phase = ((3 * xs * 65536) // width + (2 * ys * 65536) // height + (frame * 65536) // 30) & 65535
    base = 128 + ((64 * _isin(phase)) >> 15)
    # value noise on an 8x8 lattice, bilinear
    gh, gw = height // 8 + 2, width // 8 + 2
    lattice = (_hash2d(seed, 0, gh, gw, 3) % np.uint64(65)).astype(np.int64) - 32
    gx, gy = xs >> 3, ys >> 3
    fx, fy = xs & 7, ys & 7
    v00 = lattice[gy, gx]
    v01 = lattice[gy, gx + 1]
    v10 = lattice[gy + 1, gx]
    v11 = lattice[gy + 1, gx + 1]
    smooth = ((v00 * (8 - fx) + v01 * fx) * (8 - fy) + (v10 * (8 - fx) + v11 * fx) * fy) >> 6
    y = np.clip(base + smooth + _noise_sum(seed, 0, height, width, 3), 0, 255).astype(np.uint8)
    hc, wc = height // 2, width // 2
    cys, cxs = np.meshgrid(np.arange(hc, dtype=np.int64), np.arange(wc, dtype=np.int64), indexing="ij")
    gradx = (32 * (2 * cxs - wc)) // wc
    grady = (32 * (2 * cys - hc)) // hc
    cb = np.clip(128 + gradx + _noise_sum(seed, 1, hc, wc, 1), 0, 255).astype(np.uint8)
    cr = np.clip(128 - grady + _noise_sum(seed, 2, hc, wc, 1), 0, 255).astype(np.uint8)
    return y, cb, cr


def synth_textured_frame(width, height, frame):
    """Harder content for parity tests: edges, flat areas and strong texture.

    Exercises splits, angular modes and CCLM much more than synth_frame.
    """
    seed = SEED_BASE + 0x1000 + frame
    y0, cb0, cr0 = synth_frame(width, height, frame)
    ys, xs = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    blk = (_hash2d(seed, 5, height // 16 + 1, width // 16 + 1, 11) % np.uint64(6)).astype(np.int64)
    kind = blk[ys >> 4, xs >> 4]
    stripes_d = ((xs + ys + frame) >> 2 & 1) * 90 + 60
    stripes_h = ((ys >> 1) & 1) * 120 + 40
    stripes_a = (((3 * xs - 2 * ys) >> 3) & 1) * 70 + 80
    flat = np.full_like(xs, 100) + (blk[ys >> 4, xs >> 4] * 17) % 64
    y = y0.astype(np.int64)
    y = np.where(kind == 1, stripes_d, y)
    y = np.where(kind == 2, stripes_h, y)
    y = np.where(kind == 3, stripes_a, y)
    y = np.where(kind == 4, flat, y)
    y = np.clip(y + _noise_sum(seed, 6, height, width, 1), 0, 255).astype(np.uint8)
    hc, wc = height // 2, width // 2
    ysub = y[::2, ::2].astype(np.int64)
    cb = np.clip(cb0.astype(np.int64) // 2 + ysub // 3 + 20, 0, 255).astype(np.uint8)
    cr = np.clip(220 - ysub // 2 + (cr0.astype(np.int64) - 128), 0, 255).astype(np.uint8)
    return y, cb, cr
