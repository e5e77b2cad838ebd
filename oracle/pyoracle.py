"""ctypes binding of the CPU oracle (oracle/wrenc_oracle.h). TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package (wrenc_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwrenc_oracle.so")
_SPEC_SO = os.path.join(_HERE, "libwrenc_specdec.so")


class _Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("qp", C.c_int), ("max_split_depth", C.c_int)]


class _PicOut(C.Structure):
    _fields_ = [
        ("rec_y", C.c_void_p), ("rec_cb", C.c_void_p), ("rec_cr", C.c_void_p),
        ("lev_y", C.c_void_p), ("lev_cb", C.c_void_p), ("lev_cr", C.c_void_p),
        ("cu_log2_size", C.c_void_p), ("luma_mode", C.c_void_p), ("chroma_mode", C.c_void_p),
        ("ctu_cost", C.c_void_p),
    ]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("wrenc_oracle.cpp", "wrenc_oracle.h", "vvc_parse.cpp", "vvc_parse.h",
                                            "vvc_ctx_init.inc")]
    stale = force
    for so, deps in ((_SO, srcs), (_SPEC_SO, [os.path.join(_HERE, "spec_decoder.cpp")])):
        stale = stale or not os.path.exists(so) or any(
            os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(so) for f in deps)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.wro_encode_picture.restype = C.c_int
        _lib.wro_last_final_pass_mismatches.restype = C.c_long
        _lib.wro_level_cost.restype = C.c_int64
        _lib.wro_header_bits.restype = C.c_int64
        _lib.wro_chroma_header_bits.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_extra_params(text):
    """The reference's --extra-params string for every later call in this process (None / "" = defaults)."""
    rc = lib().wro_set_extra_params((text or "").encode())
    if rc != 0:
        raise ValueError("Invalid extra-params: %s" % text)


def lambda_rd_chroma(qp):
    lib().wro_lambda_rd_chroma.restype = C.c_float
    return float(lib().wro_lambda_rd_chroma(int(qp)))


def encode_picture(y, cb, cr, qp, max_split_depth):
    """Run search + final pass of one picture. Returns a dict of numpy arrays."""
    h, w = y.shape
    y = np.ascontiguousarray(y, dtype=np.uint8)
    cb = np.ascontiguousarray(cb, dtype=np.uint8)
    cr = np.ascontiguousarray(cr, dtype=np.uint8)
    out = {
        "rec_y": np.zeros((h, w), np.uint8),
        "rec_cb": np.zeros((h // 2, w // 2), np.uint8),
        "rec_cr": np.zeros((h // 2, w // 2), np.uint8),
        "lev_y": np.zeros((h, w), np.int16),
        "lev_cb": np.zeros((h // 2, w // 2), np.int16),
        "lev_cr": np.zeros((h // 2, w // 2), np.int16),
        "cu_log2_size": np.zeros((h // 4, w // 4), np.uint8),
        "luma_mode": np.zeros((h // 4, w // 4), np.uint8),
        "chroma_mode": np.zeros((h // 8, w // 8), np.uint8),
        "ctu_cost": np.zeros(((h // 32) * (w // 32),), np.float32),
    }
    po = _PicOut(*[_p(out[k]) for k in ("rec_y", "rec_cb", "rec_cr", "lev_y", "lev_cb", "lev_cr",
                                         "cu_log2_size", "luma_mode", "chroma_mode", "ctu_cost")])
    prm = _Params(w, h, qp, max_split_depth)
    rc = lib().wro_encode_picture(C.byref(prm), _p(y), _p(cb), _p(cr), C.byref(po))
    if rc != 0:
        raise ValueError("wro_encode_picture failed: %d" % rc)
    out["final_pass_mismatches"] = int(lib().wro_last_final_pass_mismatches())
    return out


def reconstruct_from_record(rec, qp, max_split_depth=3):
    """Decoder-side reconstruction from (cu_log2_size, luma_mode, chroma_mode, lev_*)."""
    h4, w4 = rec["cu_log2_size"].shape
    h, w = h4 * 4, w4 * 4
    arrs = {k: np.ascontiguousarray(rec[k]) for k in ("lev_y", "lev_cb", "lev_cr", "cu_log2_size", "luma_mode",
                                                      "chroma_mode")}
    po = _PicOut(None, None, None, _p(arrs["lev_y"]), _p(arrs["lev_cb"]), _p(arrs["lev_cr"]),
                 _p(arrs["cu_log2_size"]), _p(arrs["luma_mode"]), _p(arrs["chroma_mode"]), None)
    y = np.zeros((h, w), np.uint8)
    cb = np.zeros((h // 2, w // 2), np.uint8)
    cr = np.zeros((h // 2, w // 2), np.uint8)
    prm = _Params(w, h, qp, max_split_depth)
    rc = lib().wro_reconstruct_from_record(C.byref(prm), C.byref(po), _p(y), _p(cb), _p(cr))
    if rc != 0:
        raise ValueError("wro_reconstruct_from_record failed: %d" % rc)
    return y, cb, cr


def predict_blocks(rec_y, rec_cb, rec_cr, items, qp=32):
    """items: (n, 6) int32 {x, y, log2 luma size, tree type, component, mode}; returns the list of predicted
    blocks of that component (wro_predict_blocks)."""
    items = np.ascontiguousarray(items, np.int32).reshape(-1, 6)
    h, w = rec_y.shape
    sizes = [((1 << int(q[2])) >> (1 if q[4] else 0)) for q in items]
    out = np.zeros(int(sum(s * s for s in sizes)), np.uint8)
    planes = [np.ascontiguousarray(a, np.uint8) for a in (rec_y, rec_cb, rec_cr)]
    prm = _Params(w, h, qp, 3)
    rc = lib().wro_predict_blocks(C.byref(prm), _p(planes[0]), _p(planes[1]), _p(planes[2]), len(items), _p(items), _p(out))
    if rc != 0:
        raise ValueError("wro_predict_blocks failed: %d" % rc)
    res, at = [], 0
    for s in sizes:
        res.append(out[at:at + s * s].reshape(s, s))
        at += s * s
    return res


_spec_lib = None


def spec_lib():
    """libwrenc_specdec.so: the decoder written from H.266 alone (spec_decoder.cpp), nothing shared with the oracle."""
    global _spec_lib
    if _spec_lib is None:
        build()
        _spec_lib = C.CDLL(_SPEC_SO)
        _spec_lib.wsd_decode_record.restype = C.c_int
    return _spec_lib


def spec_decode_record(rec, qp):
    """Reconstruction of a picture from its record (cu_log2_size, luma_mode, chroma_mode, lev_*) by the
    independent spec decoder; returns (y, cb, cr)."""
    h4, w4 = rec["cu_log2_size"].shape
    h, w = h4 * 4, w4 * 4
    a = {k: np.ascontiguousarray(rec[k]) for k in ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr")}
    assert a["lev_y"].dtype == np.int16 and a["cu_log2_size"].dtype == np.uint8
    y = np.zeros((h, w), np.uint8)
    cb = np.zeros((h // 2, w // 2), np.uint8)
    cr = np.zeros((h // 2, w // 2), np.uint8)
    rc = spec_lib().wsd_decode_record(w, h, int(qp), _p(a["cu_log2_size"]), _p(a["luma_mode"]), _p(a["chroma_mode"]),
                                      _p(a["lev_y"]), _p(a["lev_cb"]), _p(a["lev_cr"]), _p(y), _p(cb), _p(cr))
    if rc != 0:
        raise ValueError("wsd_decode_record failed: %d" % rc)
    return y, cb, cr


def spec_trans_matrix():
    m = np.zeros((64, 64), np.int16)
    spec_lib().wsd_trans_matrix(_p(m))
    return m


def debug_perturb(which):
    """Deliberate misreading of one constant inside the oracle (wro_debug_perturb); 0 switches it off."""
    lib().wro_debug_perturb(int(which))


def fwd_dct(res):
    n = res.shape[0]
    res = np.ascontiguousarray(res, np.int16)
    out = np.zeros_like(res)
    lib().wro_fwd_dct(_p(res), int(n).bit_length() - 1, _p(out))
    return out


def inv_dct(deq):
    n = deq.shape[0]
    deq = np.ascontiguousarray(deq, np.int16)
    out = np.zeros_like(deq)
    lib().wro_inv_dct(_p(deq), int(n).bit_length() - 1, _p(out))
    return out


def quantize(coef, qp, viterbi=False):
    n = coef.shape[0]
    coef = np.ascontiguousarray(coef, np.int16)
    out = np.zeros_like(coef)
    fn = lib().wro_quantize_viterbi if viterbi else lib().wro_quantize
    fn(_p(coef), int(n).bit_length() - 1, int(qp), _p(out))
    return out


def quantize_sc(coef, qp, use_head=True, use_z=True, use_seg=False, use_whole=True):
    """Model of the device quantiser's shortcuts (wro_quantize_viterbi_sc); must equal quantize()."""
    n = coef.shape[0]
    coef = np.ascontiguousarray(coef, np.int16)
    out = np.zeros_like(coef)
    lib().wro_quantize_viterbi_sc(_p(coef), int(n).bit_length() - 1, int(qp), _p(out), int(use_head), int(use_z) | (2 if use_seg else 0) | (0 if use_whole else 4))
    return out


def dq_sc_stats_enable(on):
    lib().wro_dq_sc_stats_enable(int(on))


def dq_sc_stats_read():
    """(mismatching blocks, {log2n: dict of counters}) gathered since dq_sc_stats_enable(True)."""
    buf = (C.c_longlong * 72)()
    lib().wro_dq_sc_stats_read.restype = C.c_longlong
    mism = lib().wro_dq_sc_stats_read(buf)
    names = ("blocks", "nz_blocks", "sub_blocks", "head_sb_skipped", "head_tests", "head_fail", "z_eligible", "z_pass", "walked", "seg_sb",
             "seg_kept", "whole_zero")
    return int(mism), {l: dict(zip(names, [int(buf[12 * l + i]) for i in range(12)])) for l in range(2, 6)}


def dequantize(levels, qp):
    n = levels.shape[0]
    levels = np.ascontiguousarray(levels, np.int16)
    out = np.zeros_like(levels)
    lib().wro_dequantize(_p(levels), int(n).bit_length() - 1, int(qp), _p(out))
    return out


def level_cost(levels):
    n = levels.shape[0]
    levels = np.ascontiguousarray(levels, np.int16)
    return int(lib().wro_level_cost(_p(levels), int(n).bit_length() - 1))


def tables(qp):
    lv = np.zeros(1024, np.int64)
    dq = np.zeros(1024, np.int64)
    lq = C.c_int64()
    lr = C.c_float()
    lib().wro_tables(int(qp), _p(lv), _p(dq), C.byref(lq), C.byref(lr))
    return lv, dq, int(lq.value), float(lr.value)


def header_bits(tree, non_planar, mpm_flag, mpm_idx, mpm_rem, cclm_flag, cclm_idx):
    return int(lib().wro_header_bits(tree, non_planar, mpm_flag, mpm_idx, mpm_rem, cclm_flag, cclm_idx))


def chroma_header_bits(cclm_flag, cclm_idx):
    return int(lib().wro_chroma_header_bits(cclm_flag, cclm_idx))


def encode_picture_traced(y, cb, cr, qp, max_split_depth):
    """encode_picture plus the trace of every candidate evaluation: dict (x, y, log2n, tree, kind, ml, mc) ->
    set of f32 bit patterns returned for that key (re-evaluations give the same value)."""
    lib().wro_trace_enable(1)
    try:
        out = encode_picture(y, cb, cr, qp, max_split_depth)
        lib().wro_trace_read.restype = C.c_long
        n = lib().wro_trace_read(None, C.c_long(0))
        buf = np.zeros((n, 8), np.int32)
        lib().wro_trace_read(_p(buf), C.c_long(n))
    finally:
        lib().wro_trace_enable(0)
    trace = {}
    for rec in buf.tolist():
        trace.setdefault(tuple(rec[:7]), set()).add(rec[7] & 0xFFFFFFFF)
    return out, trace


def dct64():
    m = np.zeros((64, 64), np.int16)
    lib().wro_dct64(_p(m))
    return m


class _StreamInfo(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("init_qp", C.c_int), ("n_pictures", C.c_int)]


def parse_stream_info(stream):
    """Parameter sets of a byte stream written by wrenc_amd.bitstream (vvc_parse.h)."""
    buf = np.frombuffer(bytes(stream), np.uint8)
    info = _StreamInfo()
    rc = lib().wro_parse_stream_info(_p(buf), C.c_size_t(buf.size), C.byref(info))
    if rc != 0:
        raise ValueError("wro_parse_stream_info failed: %d" % rc)
    return {"width": info.width, "height": info.height, "init_qp": info.init_qp, "n_pictures": info.n_pictures}


def parse_picture(stream, index=0):
    """CABAC-decode picture `index` of the stream back into a record (no reconstruction)."""
    buf = np.frombuffer(bytes(stream), np.uint8)
    info = parse_stream_info(stream)
    w, h = info["width"], info["height"]
    out = {
        "lev_y": np.zeros((h, w), np.int16),
        "lev_cb": np.zeros((h // 2, w // 2), np.int16),
        "lev_cr": np.zeros((h // 2, w // 2), np.int16),
        "cu_log2_size": np.zeros((h // 4, w // 4), np.uint8),
        "luma_mode": np.zeros((h // 4, w // 4), np.uint8),
        "chroma_mode": np.zeros((h // 8, w // 8), np.uint8),
    }
    po = _PicOut(None, None, None, _p(out["lev_y"]), _p(out["lev_cb"]), _p(out["lev_cr"]), _p(out["cu_log2_size"]),
                 _p(out["luma_mode"]), _p(out["chroma_mode"]), None)
    poc, qp = C.c_int(), C.c_int()
    rc = lib().wro_parse_picture(_p(buf), C.c_size_t(buf.size), int(index), C.byref(poc), C.byref(qp), C.byref(po))
    if rc != 0:
        raise ValueError("wro_parse_picture failed: %d" % rc)
    out["poc_lsb"] = poc.value
    out["slice_qp"] = qp.value
    return out
