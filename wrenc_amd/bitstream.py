"""ctypes binding of the host bitstream writer (include/wrenc_bitstream.h, wrenc_amd/csrc/host).

Turns the record the device search returns (gpu.Encoder.download) into the reference's byte
stream: parameter sets once (main.rs:223-260), then per picture a picture-header NAL and one
IDR slice NAL (main.rs:294-385).  CPU code, as in the reference: CABAC is serial per picture.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "host", "libwrenc_host.so")
EXPORTED_SYMBOLS = ("wrenc_bs_picture_bound", "wrenc_bs_write_parameter_sets", "wrenc_bs_write_picture",
                    "wrenc_bs_write_picture_tokens", "wrenc_bs_last_slice_data_bits")

OK, EINVAL, ENOSPC, EDATA = 0, -1, -2, -3


class BitstreamError(RuntimeError):
    def __init__(self, code, what):
        RuntimeError.__init__(self, "%s failed with %d (%s)" % (
            what, code, {EINVAL: "bad argument", ENOSPC: "buffer too small", EDATA: "inconsistent record"}.get(code, "?")))
        self.code = code


class _Record(C.Structure):
    _fields_ = [("cu_log2_size", C.c_void_p), ("luma_mode", C.c_void_p), ("chroma_mode", C.c_void_p),
                ("lev_y", C.c_void_p), ("lev_cb", C.c_void_p), ("lev_cr", C.c_void_p)]


class _Tokens(C.Structure):
    _fields_ = [("cu_log2_size", C.c_void_p), ("luma_mode", C.c_void_p), ("chroma_mode", C.c_void_p),
                ("pool", C.c_void_p), ("pool_words", C.c_size_t), ("first_page", C.c_void_p)]


_lib = None


def load_library():
    """The writer has no Python or CPU-oracle fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.wrenc_bs_picture_bound.restype = C.c_size_t
        lib.wrenc_bs_picture_bound.argtypes = [C.c_int, C.c_int]
        lib.wrenc_bs_write_parameter_sets.restype = C.c_int
        lib.wrenc_bs_write_parameter_sets.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                      C.POINTER(C.c_size_t)]
        lib.wrenc_bs_write_picture.restype = C.c_int
        lib.wrenc_bs_write_picture.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Record), C.c_void_p,
                                               C.c_size_t, C.POINTER(C.c_size_t)]
        lib.wrenc_bs_last_slice_data_bits.restype = C.c_longlong
        _lib = lib
    return _lib


def write_parameter_sets(width, height, qp):
    """VPS + SPS + PPS NAL units (bytes)."""
    lib = load_library()
    buf = np.zeros(4096, np.uint8)
    n = C.c_size_t()
    rc = lib.wrenc_bs_write_parameter_sets(width, height, qp, buf.ctypes.data, buf.size, C.byref(n))
    if rc != OK:
        raise BitstreamError(rc, "wrenc_bs_write_parameter_sets")
    return buf[:n.value].tobytes()


def _plane(rec, key, shape, dtype):
    a = np.ascontiguousarray(rec[key], dtype=dtype)
    if a.shape != shape:
        raise ValueError("%s has shape %r, expected %r" % (key, a.shape, shape))
    return a


def write_picture(width, height, qp, poc, rec):
    """Picture header NAL + IDR slice NAL (bytes) of one picture from its search record: a dict with
    cu_log2_size, luma_mode, chroma_mode, lev_y, lev_cb, lev_cr as gpu.Encoder.download returns them."""
    lib = load_library()
    w, h = int(width), int(height)
    arrs = [_plane(rec, "cu_log2_size", (h // 4, w // 4), np.uint8), _plane(rec, "luma_mode", (h // 4, w // 4), np.uint8),
            _plane(rec, "chroma_mode", (h // 8, w // 8), np.uint8), _plane(rec, "lev_y", (h, w), np.int16),
            _plane(rec, "lev_cb", (h // 2, w // 2), np.int16), _plane(rec, "lev_cr", (h // 2, w // 2), np.int16)]
    r = _Record(*[a.ctypes.data for a in arrs])
    # wrenc_bs_picture_bound is the proven worst case (12 bytes per luma sample); real pictures need a small
    # fraction, and the writer reports the size it needs when the buffer is too small
    cap = min(lib.wrenc_bs_picture_bound(w, h), w * h // 2 + 65536)
    buf = np.empty(cap, np.uint8)
    n = C.c_size_t()
    rc = lib.wrenc_bs_write_picture(w, h, int(qp), int(poc), C.byref(r), buf.ctypes.data, cap, C.byref(n))
    if rc == ENOSPC:
        cap = n.value
        buf = np.empty(cap, np.uint8)
        rc = lib.wrenc_bs_write_picture(w, h, int(qp), int(poc), C.byref(r), buf.ctypes.data, cap, C.byref(n))
    if rc != OK:
        raise BitstreamError(rc, "wrenc_bs_write_picture")
    return buf[:n.value].tobytes()


def write_picture_tokens(width, height, qp, poc, pool, pic):
    """The same NAL units from the device's token record (gpu.Encoder.download_tokens: `pool` and one of its per-picture
    dicts): the host runs the CU-level syntax and the arithmetic coder only."""
    lib = load_library()
    w, h = int(width), int(height)
    pool = np.ascontiguousarray(pool, np.uint32)
    arrs = [_plane(pic, "cu_log2_size", (h // 4, w // 4), np.uint8), _plane(pic, "luma_mode", (h // 4, w // 4), np.uint8),
            _plane(pic, "chroma_mode", (h // 8, w // 8), np.uint8)]
    first = np.ascontiguousarray(pic["first_page"], np.uint32)
    t = _Tokens(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, pool.ctypes.data, pool.size, first.ctypes.data)
    lib.wrenc_bs_write_picture_tokens.restype = C.c_int
    lib.wrenc_bs_write_picture_tokens.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Tokens), C.c_void_p, C.c_size_t,
                                                  C.POINTER(C.c_size_t)]
    cap = min(lib.wrenc_bs_picture_bound(w, h), w * h // 2 + 65536)
    buf = np.empty(cap, np.uint8)
    n = C.c_size_t()
    rc = lib.wrenc_bs_write_picture_tokens(w, h, int(qp), int(poc), C.byref(t), buf.ctypes.data, cap, C.byref(n))
    if rc == ENOSPC:
        cap = n.value
        buf = np.empty(cap, np.uint8)
        rc = lib.wrenc_bs_write_picture_tokens(w, h, int(qp), int(poc), C.byref(t), buf.ctypes.data, cap, C.byref(n))
    if rc != OK:
        raise BitstreamError(rc, "wrenc_bs_write_picture_tokens")
    return buf[:n.value].tobytes()


def last_slice_data_bits():
    """CABAC bits of the CTU data of this thread's last write_picture call."""
    return int(load_library().wrenc_bs_last_slice_data_bits())
