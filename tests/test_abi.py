"""CPU tests of the product's host side: the C-ABI library loads, exports every
symbol the header declares, resolves the same constant tables as the oracle, and
fails loudly (never falls back) when no gfx950 device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "wrenc_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wrenc_gpu_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    from wrenc_amd import gpu
    lib = C.CDLL(gpu.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(gpu.EXPORTED_SYMBOLS) == declared


def test_default_config_matches_oracle_tables(built):
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    for qp in (0, 12, 22, 26, 27, 32, 37, 45, 51, 63):
        cfg = gpu.default_config(64, 64, qp, 2)
        lv, dq, lq, lr = po.tables(qp)
        assert np.array_equal(np.array(cfg.lv_table), lv)
        assert np.array_equal(np.array(cfg.dq_table), dq)
        assert cfg.lambda_q == lq
        assert cfg.lambda_rd == lr and cfg.lambda_rd_chroma == lr
    cfg = gpu.default_config(64, 64, 32, 2)
    hb = np.array(cfg.header_bits_luma).reshape(2, 4, 67)
    for tree in range(2):
        for cc in range(4):
            if tree == 1 and cc > 0:
                continue
            for cls in range(67):
                want = po.header_bits(tree, int(cls > 0), int(cls <= 5), cls - 1 if 1 <= cls <= 5 else 0,
                                      cls - 6 if cls > 5 else 0, int(cc > 0), max(cc - 1, 0))
                assert hb[tree, cc, cls] == want, (tree, cc, cls)
    hc = np.array(cfg.header_bits_chroma)
    for cc in range(4):
        assert hc[cc] == po.chroma_header_bits(int(cc > 0), max(cc - 1, 0))


EXTRA = ("lv_pow_dq_trellis=0.52,lv_offset_dq_trellis=0.2,quant_lv_pow=0.49,quant_qp_div_trellis=5.0,"
         "quant_lambda_mul_trellis=1.4,quant_lambda_offset_trellis=7,qp_div_dq_trellis=4.2,lambda_mul_dq_trellis=1.3,"
         "non_planar_offset_dq_trellis=2.0,mpm_idx_offset_dq_trellis=1.5,mpm_remainder_mult_dq_trellis=0.6,"
         "mpm_remainder_offset_dq_trellis=2.1,planer_offset_dq_trellis=1.1,header_bits_dq_trellis=1.0,"
         "chroma_header_bits_dq_trellis=1.5,cclm_pow=0.5,mpm_idx_pow=0.45,mpm_remainder_pow=0.3,"
         "cclm_mode_idx_offset_dq_trellis=2.4,non_cclm_offset_dq_trellis=0.7,cclm_offset_dq_trellis=0.6,a=0.9,"
         "lv_pow_dq=0.1,header_bits=9.0,some_future_knob=1")


def test_extra_params_resolve_like_the_oracle(built):
    """--extra-params (main.rs:202-217): every live key moves the product's tables exactly as it moves the
    oracle's; dead and unknown keys change nothing; malformed items are rejected."""
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    try:
        for qp in (22, 32, 37):
            po.set_extra_params(EXTRA)
            cfg = gpu.default_config(64, 64, qp, 2, extra_params=EXTRA)
            lv, dq, lq, lr = po.tables(qp)
            assert np.array_equal(np.array(cfg.lv_table), lv) and np.array_equal(np.array(cfg.dq_table), dq)
            assert cfg.lambda_q == lq and cfg.lambda_rd == lr
            assert cfg.lambda_rd_chroma == po.lambda_rd_chroma(qp) != lr
            hb = np.array(cfg.header_bits_luma).reshape(2, 4, 67)
            for tree, cc, cls in [(0, 0, 0), (0, 0, 3), (0, 2, 40), (1, 0, 5), (1, 0, 66), (0, 3, 6), (0, 1, 1)]:
                want = po.header_bits(tree, int(cls > 0), int(cls <= 5), cls - 1 if 1 <= cls <= 5 else 0,
                                      cls - 6 if cls > 5 else 0, int(cc > 0), max(cc - 1, 0))
                assert hb[tree, cc, cls] == want, (tree, cc, cls)
            for cc in range(4):
                assert cfg.header_bits_chroma[cc] == po.chroma_header_bits(int(cc > 0), max(cc - 1, 0))
            default = gpu.default_config(64, 64, qp, 2)
            assert cfg.lambda_q != default.lambda_q and cfg.lambda_rd != default.lambda_rd
            assert not np.array_equal(np.array(cfg.lv_table), np.array(default.lv_table))
            only_dead = gpu.default_config(64, 64, qp, 2, extra_params="lv_pow_dq=0.1,header_bits=9.0,x=y")
            assert bytes(only_dead) == bytes(default)
    finally:
        po.set_extra_params(None)
    for bad in ("a", "a=1=2", "qp_div_dq_trellis=abc", "a=1,,b=2", "quant_lambda_offset_trellis=1.5"):
        with pytest.raises(gpu.WrencGpuError):
            gpu.default_config(64, 64, 32, 2, extra_params=bad)
    with pytest.raises(ValueError):
        po.set_extra_params("novalue")
    po.set_extra_params(None)


def test_create_rejects_bad_arguments_and_never_falls_back(built):
    from wrenc_amd import gpu
    lib = gpu.load_library()
    for (w, h, qp, d) in [(100, 64, 32, 2), (64, 60, 32, 2), (64, 64, 64, 2), (64, 64, 32, 4), (0, 0, 32, 0)]:
        cfg = gpu.default_config(w, h, qp, d)
        ctx = C.c_void_p()
        rc = lib.wrenc_gpu_create(C.byref(cfg), C.byref(ctx))
        assert rc == -1 and not ctx.value           # WRENC_GPU_EINVAL
        assert lib.wrenc_gpu_last_error(None)
    # a quantiser rate model whose step costs do not fit the device's 32-bit trellis arithmetic is refused, not searched
    # with other results than the reference's (include/wrenc_gpu.h); the defaults fit at every QP, with room
    for extra in ("quant_lv_pow=2.5", "quant_qp_div_trellis=1.5", "quant_lambda_mul_trellis=-3"):
        cfg = gpu.default_config(64, 64, 37, 2, extra_params=extra)
        ctx = C.c_void_p()
        rc = lib.wrenc_gpu_create(C.byref(cfg), C.byref(ctx))
        assert rc == -1 and not ctx.value, extra
        assert b"rate model" in lib.wrenc_gpu_last_error(None)
    for qp in (0, 32, 63):
        cfg = gpu.default_config(64, 64, qp, 2)
        assert 0 <= cfg.lambda_q * max(cfg.dq_table) < (1 << 25) - 128 * 65535
    import torch
    if not torch.cuda.is_available():
        cfg = gpu.default_config(64, 64, 32, 2)
        ctx = C.c_void_p()
        rc = lib.wrenc_gpu_create(C.byref(cfg), C.byref(ctx))
        assert rc == -2 and not ctx.value           # WRENC_GPU_ENODEV: loud failure, no CPU path
        with pytest.raises(gpu.WrencGpuError):
            gpu.Encoder(64, 64, qp=32, max_split_depth=2)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "wrenc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "wrenc_oracle" not in txt and "libwrenc_oracle" not in txt, f


def test_synth_is_deterministic():
    import hashlib
    from wrenc_amd import synth
    y, cb, cr = synth.synth_frame(64, 64, 3)
    h = hashlib.sha256(y.tobytes() + cb.tobytes() + cr.tobytes()).hexdigest()
    y2, cb2, cr2 = synth.synth_frame(64, 64, 3)
    assert np.array_equal(y, y2) and np.array_equal(cb, cb2) and np.array_equal(cr, cr2)
    g = np.load(os.path.join(ROOT, "tests", "golden", "smooth64_qp37_d1.npz"))
    y0, cb0, cr0 = synth.synth_frame(64, 64, 0)
    assert np.array_equal(g["y"], y0) and np.array_equal(g["cb"], cb0) and np.array_equal(g["cr"], cr0)
    assert len(h) == 64


def test_expand_levels_is_the_inverse_of_the_compact_layout(built):
    """wrenc_gpu_expand_levels (host only, no device): the compact read-back's layout -- one mask bit per picture-aligned
    4x4 block of levels (luma plane, then Cb, then Cr, raster order of the blocks), the coded blocks' 16 levels row-major in
    mask order -- rebuilt here from random sparse level planes and expanded back by the library."""
    import ctypes as C
    from wrenc_amd import gpu
    lib = C.CDLL(gpu.LIB_PATH)
    lib.wrenc_gpu_compact_mask_words.restype = C.c_size_t
    lib.wrenc_gpu_compact_mask_words.argtypes = [C.c_int, C.c_int]
    lib.wrenc_gpu_expand_levels.restype = None
    lib.wrenc_gpu_expand_levels.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5
    rng = np.random.default_rng(31)
    for w, h in ((64, 32), (96, 64), (32, 32)):
        planes = []
        for pw, ph in ((w, h), (w // 2, h // 2), (w // 2, h // 2)):
            p = np.zeros((ph, pw), np.int16)
            coded = rng.random((ph // 4, pw // 4)) < 0.3
            for by, bx in zip(*np.nonzero(coded)):
                blk = rng.integers(-300, 300, (4, 4)).astype(np.int16) * (rng.random((4, 4)) < 0.4)
                blk[rng.integers(4), rng.integers(4)] = rng.integers(1, 900) * (1 if rng.random() < 0.5 else -1)   # really coded
                p[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] = blk
            planes.append(p)
        bits, payload = [], []
        for p in planes:
            ph, pw = p.shape
            for by in range(ph // 4):
                for bx in range(pw // 4):
                    blk = p[4 * by:4 * by + 4, 4 * bx:4 * bx + 4]
                    bits.append(bool(np.any(blk)))
                    if bits[-1]:
                        payload.append(blk.reshape(16))
        words = lib.wrenc_gpu_compact_mask_words(w, h)
        assert words == (len(bits) + 31) // 32 == ((w // 4) * (h // 4) * 3 // 2 + 31) // 32
        mask = np.zeros(words, np.uint32)
        for i, b in enumerate(bits):
            if b:
                mask[i >> 5] |= np.uint32(1) << np.uint32(i & 31)
        pay = np.ascontiguousarray(np.stack(payload) if payload else np.zeros((1, 16), np.int16), np.int16)
        out = [np.full_like(p, 7) for p in planes]     # (the library must clear what it does not code)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        lib.wrenc_gpu_expand_levels(w, h, ptr(mask), ptr(pay), ptr(out[0]), ptr(out[1]), ptr(out[2]))
        for got, want in zip(out, planes):
            assert np.array_equal(got, want)


def test_device_token_contexts_match_the_host_context_array():
    """The residual tokens the device emits (wrenc_amd/csrc/dev_bins.h) name contexts by their index in the host coder's flat
    array (wrenc_amd/csrc/host/cabac.h, CtxBase): the six bases the device uses are the host's, and the page size of the
    token pool is one constant in the three places that know it."""
    import re
    host = open(os.path.join(ROOT, "wrenc_amd", "csrc", "host", "cabac.h")).read()
    vals = {}
    for name, expr in re.findall(r"\b(CTX_[A-Z_]+) = ([A-Z_0-9 +]+?)\s*(?:,|//|\n)", host):
        expr = expr.strip()
        m = re.match(r"(CTX_[A-Z_]+) \+ (\d+)$", expr)
        vals[name] = int(expr) if expr.isdigit() else vals[m.group(1)] + int(m.group(2))
    dev = open(os.path.join(ROOT, "wrenc_amd", "csrc", "dev_bins.h")).read()
    m = re.search(r"CTXD_LAST_X = (\d+), CTXD_LAST_Y = CTXD_LAST_X \+ (\d+), CTXD_SB_CODED = CTXD_LAST_Y \+ (\d+), CTXD_SIG = CTXD_SB_CODED \+ (\d+),\s*"
                  r"CTXD_PAR = CTXD_SIG \+ (\d+), CTXD_GTX = CTXD_PAR \+ (\d+);", dev)
    a = [int(g) for g in m.groups()]
    got = {"CTX_LAST_X": a[0], "CTX_LAST_Y": a[0] + a[1], "CTX_SB_CODED": a[0] + a[1] + a[2], "CTX_SIG": a[0] + a[1] + a[2] + a[3],
           "CTX_PAR": a[0] + a[1] + a[2] + a[3] + a[4], "CTX_GTX": a[0] + a[1] + a[2] + a[3] + a[4] + a[5]}
    for k, v in got.items():
        assert vals[k] == v, (k, vals[k], v)
    assert vals["CTX_COUNT"] == 253
    page = int(re.search(r"constexpr int kTokPage = (\d+);", dev).group(1))
    gh = open(os.path.join(ROOT, "include", "wrenc_gpu.h")).read()
    bh = open(os.path.join(ROOT, "include", "wrenc_bitstream.h")).read()
    assert page == int(re.search(r"#define WRENC_GPU_TOKEN_PAGE (\d+)", gh).group(1)) == int(re.search(r"#define WRENC_BS_TOKEN_PAGE (\d+)", bh).group(1))
