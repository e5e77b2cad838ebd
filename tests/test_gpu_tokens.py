"""residual_coding on the device (wrenc_amd/csrc/dev_bins.h): the token record through the host's arithmetic coder writes
the bytes the host-only writer writes from the level planes of the same search result -- on searched pictures at several
QPs and depths, and on random records no search would emit (every luma mode and block size, levels up to the i16 range:
escape codes, transform blocks that exhaust the context-coded bin budget)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(enc, w, h, qp, slots):
    from wrenc_amd import bitstream as bs
    pool, pics = enc.download_tokens(0, slots)
    out = []
    for s in range(slots):
        rec = enc.download(s)
        a = bs.write_picture(w, h, qp, s, rec)
        b = bs.write_picture_tokens(w, h, qp, s, pool, pics[s])
        out.append((a, b))
    return out, pool


@pytest.mark.parametrize("w,h,qp,depth", [(64, 64, 32, 2), (96, 64, 22, 3), (128, 96, 37, 3), (64, 96, 27, 1), (352, 288, 20, 3),
                                          (64, 64, 45, 0)])
def test_token_stream_of_searched_pictures(built, w, h, qp, depth):
    from wrenc_amd import gpu, synth
    frames = [synth.synth_textured_frame(w, h, 5), synth.synth_frame(w, h, 1), synth.synth_textured_frame(w, h, 9)]
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=3)
    for s, f in enumerate(frames):
        enc.upload(s, *f)
    enc.encode(0, 3)
    enc.sync()
    pairs, pool = _both(enc, w, h, qp, 3)
    enc.close()
    for s, (a, b) in enumerate(pairs):
        assert a == b, "picture %d: %d bytes from the planes, %d from the tokens" % (s, len(a), len(b))
    assert pool.size % gpu.TOKEN_PAGE == 0 and pool.size > 0


def _random_record(rng, w, h, kind):
    """A record with a random quadtree (sizes 32..4), random modes, and levels whose parity follows the dependent-
    quantisation state of their transform block -- the one thing the syntax requires of them."""
    cu = np.zeros((h // 4, w // 4), np.uint8)
    for y0 in range(0, h, 32):
        for x0 in range(0, w, 32):
            def fill(x, y, lg):
                if lg > 2 and rng.random() < (0.55 if lg > 3 else 0.4):
                    if lg == 3:
                        cu[y // 4:y // 4 + 2, x // 4:x // 4 + 2] = 2
                        return
                    for i in range(4):
                        fill(x + ((i & 1) << (lg - 1)), y + ((i >> 1) << (lg - 1)), lg - 1)
                else:
                    cu[y // 4:(y + (1 << lg)) // 4, x // 4:(x + (1 << lg)) // 4] = lg
            fill(x0, y0, 5)
    luma = rng.integers(0, 67, (h // 4, w // 4)).astype(np.uint8)
    # one mode per CU (the top-left unit's)
    for y in range(h // 4):
        for x in range(w // 4):
            lg = cu[y, x]
            n4 = 1 << (lg - 2)
            luma[y, x] = luma[(y // n4) * n4, (x // n4) * n4]
    chroma = np.zeros((h // 8, w // 8), np.uint8)
    for y in range(h // 8):
        for x in range(w // 8):
            lg = max(cu[2 * y, 2 * x], 3)
            n8 = 1 << (lg - 3)
            by, bx = (y // n8) * n8, (x // n8) * n8
            if (y, x) == (by, bx):
                r = rng.random()
                if r < 0.3:
                    chroma[y, x] = 81 + rng.integers(0, 3)
                else:
                    ref = luma[2 * y, 2 * x] if cu[2 * y, 2 * x] >= 3 else luma[2 * y + 1, 2 * x + 1]
                    chroma[y, x] = ref
            else:
                chroma[y, x] = chroma[by, bx]
    lev = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]

    def diag(n):
        out, x, y = [], 0, 0
        while len(out) < n * n:
            while y >= 0:
                if x < n and y < n:
                    out.append((x, y))
                y -= 1
                x += 1
            y, x = x, 0
        return out
    scans = {n: diag(n) for n in (1, 2, 4, 8)}
    trans = [[0, 2], [2, 0], [1, 3], [3, 1]]

    def fill_tb(plane, x0, y0, lg):
        n = 1 << lg
        mode = rng.integers(0, 6)
        if mode == 0:
            return  # an uncoded block
        dens = [0.0, 0.02, 0.15, 0.5, 0.9, 1.0][mode]
        big = kind == "big" and rng.random() < 0.5
        amp = [1, 2, 3, 8, 40, 300][rng.integers(0, 6)] * (60 if big else 1)
        state = 0
        sbs, cs = scans[n // 4], scans[4]
        for i in range(len(sbs) - 1, -1, -1):
            if rng.random() < 0.25 and mode < 4:
                continue      # an empty sub-block keeps the state
            for k in range(15, -1, -1):
                x, y = (sbs[i][0] << 2) + cs[k][0], (sbs[i][1] << 2) + cs[k][1]
                a = 0
                if rng.random() < dens:
                    a = int(min(abs(rng.normal(0, amp)) + 1, 16000))
                d = 1 if state > 1 else 0
                q = max(2 * a - d, 0) if a else 0
                if a and q == 0:
                    a, q = 1, 2 - d
                plane[y0 + y, x0 + x] = q if rng.random() < 0.5 else -q
                state = trans[state][a & 1]
    for y in range(0, h, 4):
        for x in range(0, w, 4):
            lg = cu[y // 4, x // 4]
            n = 1 << lg
            if y % n or x % n:
                continue
            fill_tb(lev[0], x, y, lg)
            if lg >= 3:
                fill_tb(lev[1], x // 2, y // 2, lg - 1)
                fill_tb(lev[2], x // 2, y // 2, lg - 1)
    for y in range(0, h, 8):
        for x in range(0, w, 8):
            if cu[y // 4, x // 4] == 2:
                fill_tb(lev[1], x // 2, y // 2, 2)
                fill_tb(lev[2], x // 2, y // 2, 2)
    return {"cu_log2_size": cu, "luma_mode": luma, "chroma_mode": chroma, "lev_y": lev[0], "lev_cb": lev[1], "lev_cr": lev[2]}


@pytest.mark.parametrize("kind", ["small", "big"])
def test_token_stream_of_random_records(built, kind):
    from wrenc_amd import bitstream as bs, gpu
    w, h, qp = 128, 96, 30
    rng = np.random.default_rng(77 if kind == "small" else 78)
    n = 6
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=3, n_slots=n)
    recs = [_random_record(rng, w, h, kind) for _ in range(n)]
    for s, r in enumerate(recs):
        enc.test_load_record(s, r)
    pool, pics = enc.download_tokens(0, n, pool_words=n * w * h * 8)
    enc.close()
    for s, r in enumerate(recs):
        a = bs.write_picture(w, h, qp, s, r)
        b = bs.write_picture_tokens(w, h, qp, s, pool, pics[s])
        assert a == b, "record %d (%s): %d bytes from the planes, %d from the tokens" % (s, kind, len(a), len(b))


def test_a_pool_that_is_too_small_is_reported(built):
    from wrenc_amd import gpu, synth
    w, h = 64, 64
    enc = gpu.Encoder(w, h, qp=22, max_split_depth=2, n_slots=1)
    enc.upload(0, *synth.synth_textured_frame(w, h, 2))
    enc.encode(0, 1)
    enc.sync()
    with pytest.raises(gpu.WrencGpuError) as e:
        enc.download_tokens(0, 1, pool_words=gpu.TOKEN_PAGE * 2)
    assert e.value.code == -3      # WRENC_GPU_ENOMEM
    pool, pics = enc.download_tokens(0, 1)      # and the context is fine afterwards
    assert enc.last_token_words > gpu.TOKEN_PAGE * 2 and pool.size >= enc.last_token_words
    enc.close()
