"""ctypes binding of the C ABI in include/wrenc_gpu.h (libwrenc_gpu.so).

This is the Python-side mirror used by tests and bench.py; a C++/Rust host binds
the same symbols.  There is no CPU fallback: if the HIP
library is missing or no gfx950 device is present, creation fails loudly.
"""
import ctypes as C
import os

import numpy as np

# The HIP runtime gives a process 4 hardware queues per device by default; the context uses 4 encode lanes
# plus one copy stream, and the copy stream only overlaps the search if it has a queue of its own.  Must be
# in the environment before the runtime initialises (first HIP call of the process).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WRENC_GPU_LIB", os.path.join(_HERE, "csrc", "libwrenc_gpu.so"))

EXPORTED_SYMBOLS = [
    "wrenc_gpu_default_config", "wrenc_gpu_config_extra_params", "wrenc_gpu_create", "wrenc_gpu_destroy", "wrenc_gpu_last_error",
    "wrenc_gpu_upload", "wrenc_gpu_encode", "wrenc_gpu_sync", "wrenc_gpu_download", "wrenc_gpu_download_compact", "wrenc_gpu_compact_mask_words", "wrenc_gpu_expand_levels", "wrenc_gpu_download_tokens", "wrenc_gpu_test_load_record", "wrenc_gpu_device_info",
    "wrenc_gpu_alloc_host", "wrenc_gpu_free_host", "wrenc_gpu_encode_picture", "wrenc_gpu_set_schedule", "wrenc_gpu_last_schedule", "wrenc_gpu_stats_enable", "wrenc_gpu_last_encode_stats", "wrenc_gpu_last_encode_kernel_stats", "wrenc_gpu_final_pass_mismatches",
    "wrenc_gpu_test_fwd_dct", "wrenc_gpu_test_inv_dct", "wrenc_gpu_test_quantize",
    "wrenc_gpu_test_dequantize", "wrenc_gpu_test_predict", "wrenc_gpu_test_fwd_dct32", "wrenc_gpu_test_inv_dct32", "wrenc_gpu_test_quantize_p16", "wrenc_gpu_test_quantize_pk", "wrenc_gpu_test_set_wave_slots", "wrenc_gpu_test_scratch_overflows", "wrenc_gpu_test_head_ranges", "wrenc_gpu_test_avail_tab",
]


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("qp", C.c_int32), ("max_split_depth", C.c_int32),
        ("device", C.c_int32), ("n_slots", C.c_int32),
        ("lv_table", C.c_int64 * 1024), ("dq_table", C.c_int64 * 1024),
        ("lambda_q", C.c_int64), ("lambda_rd", C.c_float), ("lambda_rd_chroma", C.c_float),
        ("header_bits_luma", C.c_int64 * 67 * 4 * 2),
        ("header_bits_chroma", C.c_int64 * 4),
    ]


class Picture(C.Structure):
    _fields_ = [
        ("rec_y", C.c_void_p), ("rec_cb", C.c_void_p), ("rec_cr", C.c_void_p),
        ("lev_y", C.c_void_p), ("lev_cb", C.c_void_p), ("lev_cr", C.c_void_p),
        ("cu_log2_size", C.c_void_p), ("luma_mode", C.c_void_p), ("chroma_mode", C.c_void_p),
        ("ctu_cost", C.c_void_p),
    ]


class Compact(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("payload", C.c_void_p), ("payload_cap", C.c_size_t), ("n_blocks", C.c_size_t),
                ("cu_log2_size", C.c_void_p), ("luma_mode", C.c_void_p), ("chroma_mode", C.c_void_p),
                ("rec_y", C.c_void_p), ("rec_cb", C.c_void_p), ("rec_cr", C.c_void_p)]


class Tokens(C.Structure):
    _fields_ = [("first_page", C.c_void_p), ("cu_log2_size", C.c_void_p), ("luma_mode", C.c_void_p), ("chroma_mode", C.c_void_p),
                ("rec_y", C.c_void_p), ("rec_cb", C.c_void_p), ("rec_cr", C.c_void_p)]


TOKEN_PAGE = 64  # WRENC_GPU_TOKEN_PAGE


class WrencGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("wrenc_gpu error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """Load libwrenc_gpu.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.wrenc_gpu_last_error.restype = C.c_char_p
        lib.wrenc_gpu_last_error.argtypes = [C.c_void_p]
        lib.wrenc_gpu_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        lib.wrenc_gpu_destroy.argtypes = [C.c_void_p]
        lib.wrenc_gpu_destroy.restype = None
        lib.wrenc_gpu_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_size_t]
        lib.wrenc_gpu_encode.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.wrenc_gpu_sync.argtypes = [C.c_void_p]
        lib.wrenc_gpu_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(Picture)]
        lib.wrenc_gpu_encode_picture.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(Picture)]
        lib.wrenc_gpu_last_encode_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                    C.POINTER(C.c_int)]
        lib.wrenc_gpu_final_pass_mismatches.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        for name in ("fwd_dct", "inv_dct", "dequantize"):
            getattr(lib, "wrenc_gpu_test_" + name).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                               C.c_void_p]
        lib.wrenc_gpu_test_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p]
        _lib = lib
    return _lib


def default_config(width, height, qp, max_split_depth, device=0, n_slots=1, extra_params=None):
    cfg = Config()
    lib = load_library()
    rc = lib.wrenc_gpu_default_config(C.byref(cfg), width, height, qp, max_split_depth)
    if rc:
        raise WrencGpuError(rc, "default_config")
    if extra_params:
        rc = lib.wrenc_gpu_config_extra_params(C.byref(cfg), extra_params.encode())
        if rc:
            raise WrencGpuError(rc, lib.wrenc_gpu_last_error(None).decode())
    cfg.device = device
    cfg.n_slots = n_slots
    return cfg


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def alloc_picture(width, height):
    return {
        "rec_y": np.zeros((height, width), np.uint8),
        "rec_cb": np.zeros((height // 2, width // 2), np.uint8),
        "rec_cr": np.zeros((height // 2, width // 2), np.uint8),
        "lev_y": np.zeros((height, width), np.int16),
        "lev_cb": np.zeros((height // 2, width // 2), np.int16),
        "lev_cr": np.zeros((height // 2, width // 2), np.int16),
        "cu_log2_size": np.zeros((height // 4, width // 4), np.uint8),
        "luma_mode": np.zeros((height // 4, width // 4), np.uint8),
        "chroma_mode": np.zeros((height // 8, width // 8), np.uint8),
        "ctu_cost": np.zeros(((height // 32) * (width // 32),), np.float32),
    }


_PIC_KEYS = ("rec_y", "rec_cb", "rec_cr", "lev_y", "lev_cb", "lev_cr", "cu_log2_size", "luma_mode",
             "chroma_mode", "ctu_cost")


class Encoder:
    """One context = one GPU.  Mirrors the picture-granular call surface a C++
    SliceEncoder::encode uses in place of the per-CTU split_ct loop."""

    def __init__(self, width, height, qp=26, max_split_depth=3, device=0, n_slots=1, config=None, extra_params=None,
                 schedule=None):
        self.lib = load_library()
        self.cfg = config if config is not None else default_config(width, height, qp, max_split_depth,
                                                                     device, n_slots, extra_params)
        self.width, self.height = self.cfg.width, self.cfg.height
        self.ctx = C.c_void_p()
        rc = self.lib.wrenc_gpu_create(C.byref(self.cfg), C.byref(self.ctx))
        if rc:
            raise WrencGpuError(rc, self.lib.wrenc_gpu_last_error(None).decode())
        self._keep = {}
        self._pinned = []
        if schedule is not None:
            self.set_schedule(schedule)

    def _check(self, rc):
        if rc:
            raise WrencGpuError(rc, self.lib.wrenc_gpu_last_error(self.ctx).decode())

    def close(self):
        if self.ctx:
            self.lib.wrenc_gpu_free_host.restype = None
            self.lib.wrenc_gpu_free_host.argtypes = [C.c_void_p, C.c_void_p]
            for ptr in self._pinned:
                self.lib.wrenc_gpu_free_host(self.ctx, ptr)
            self._pinned = []
            self.lib.wrenc_gpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, slot, y, cb, cr):
        y = np.ascontiguousarray(y, np.uint8)
        cb = np.ascontiguousarray(cb, np.uint8)
        cr = np.ascontiguousarray(cr, np.uint8)
        assert y.shape == (self.height, self.width)
        self._keep[slot] = (y, cb, cr)  # host buffers must outlive the async copy
        self._check(self.lib.wrenc_gpu_upload(self.ctx, slot, _p(y), _p(cb), _p(cr), self.width, self.width // 2))

    def encode(self, first_slot=0, n_pictures=1):
        self._check(self.lib.wrenc_gpu_encode(self.ctx, first_slot, n_pictures))

    def sync(self):
        self._check(self.lib.wrenc_gpu_sync(self.ctx))

    def download_compact(self, first_slot, n, payload_cap=None):
        """The compact read-back of n slots (include/wrenc_gpu.h): per picture (mask, payload[:n_blocks], maps dict)."""
        w, h = self.width, self.height
        self.lib.wrenc_gpu_compact_mask_words.restype = C.c_size_t
        self.lib.wrenc_gpu_compact_mask_words.argtypes = [C.c_int, C.c_int]
        words = self.lib.wrenc_gpu_compact_mask_words(w, h)
        blocks = (w // 4) * (h // 4) * 3 // 2
        cap = blocks if payload_cap is None else payload_cap
        outs = (Compact * n)()
        keep = []
        for k in range(n):
            mask = np.zeros(words, np.uint32)
            pay = np.zeros((max(cap, 1), 16), np.int16)
            maps = {"cu_log2_size": np.zeros((h // 4, w // 4), np.uint8), "luma_mode": np.zeros((h // 4, w // 4), np.uint8),
                    "chroma_mode": np.zeros((h // 8, w // 8), np.uint8)}
            outs[k].mask, outs[k].payload, outs[k].payload_cap = _p(mask).value, _p(pay).value, cap
            for name, arr in maps.items():
                setattr(outs[k], name, _p(arr).value)
            keep.append((mask, pay, maps))
        self.lib.wrenc_gpu_download_compact.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self._check(self.lib.wrenc_gpu_download_compact(self.ctx, first_slot, n, outs))
        return [(m, p[:outs[k].n_blocks], maps) for k, (m, p, maps) in enumerate(keep)]

    def download_tokens(self, first_slot, n, pool_words=None):
        """The token read-back of n slots (include/wrenc_gpu.h: residual_coding done on the device): (pool, [per picture a
        dict with first_page and the maps]); feed both to bitstream.write_picture_tokens."""
        w, h = self.width, self.height
        grow = pool_words is None
        if pool_words is None:
            pool_words = max(n * w * h * 3 // 2, 1 << 16)   # 6 bytes per luma sample: plenty at QP 27 and above
        pool = np.zeros(pool_words, np.uint32)
        outs = (Tokens * n)()
        pics = []
        for k in range(n):
            d = {"first_page": np.zeros((h // 32) * (w // 32), np.uint32), "cu_log2_size": np.zeros((h // 4, w // 4), np.uint8),
                 "luma_mode": np.zeros((h // 4, w // 4), np.uint8), "chroma_mode": np.zeros((h // 8, w // 8), np.uint8)}
            for name, arr in d.items():
                setattr(outs[k], name, _p(arr).value)
            pics.append(d)
        used = C.c_size_t()
        self.lib.wrenc_gpu_download_tokens.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        rc = self.lib.wrenc_gpu_download_tokens(self.ctx, first_slot, n, outs, _p(pool), pool_words, C.byref(used))
        while rc == -3 and grow and pool_words < (1 << 31):     # WRENC_GPU_ENOMEM: a bigger pool (the pass is cheap)
            pool_words *= 4
            pool = np.zeros(pool_words, np.uint32)
            rc = self.lib.wrenc_gpu_download_tokens(self.ctx, first_slot, n, outs, _p(pool), pool_words, C.byref(used))
        self._check(rc)
        self.last_token_words = int(used.value)   # words of the pages in use (they are spread over the whole pool)
        return pool, pics

    def device_info(self):
        """(wavefronts of the search kernel the device holds at once, HIP streams an encode call uses)."""
        slots, lanes = C.c_longlong(), C.c_int()
        self.lib.wrenc_gpu_device_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(self.lib.wrenc_gpu_device_info(self.ctx, C.byref(slots), C.byref(lanes)))
        return int(slots.value), int(lanes.value)

    def test_load_record(self, slot, rec):
        """Test entry: put a record (maps + level planes) into a slot as if a search had produced it."""
        keep = {k: np.ascontiguousarray(rec[k]) for k in ("lev_y", "lev_cb", "lev_cr", "cu_log2_size", "luma_mode", "chroma_mode")}
        pic = Picture(*[_p(keep[k]) if k in keep else None for k in _PIC_KEYS])
        self.lib.wrenc_gpu_test_load_record.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._check(self.lib.wrenc_gpu_test_load_record(self.ctx, slot, C.byref(pic)))

    def expand_levels(self, mask, payload):
        """Dense level planes from a compact record (host only)."""
        w, h = self.width, self.height
        ly = np.empty((h, w), np.int16)
        lcb = np.empty((h // 2, w // 2), np.int16)
        lcr = np.empty((h // 2, w // 2), np.int16)
        pay = np.ascontiguousarray(payload, np.int16)
        self.lib.wrenc_gpu_expand_levels.restype = None
        self.lib.wrenc_gpu_expand_levels.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.wrenc_gpu_expand_levels(w, h, _p(mask), _p(pay), _p(ly), _p(lcb), _p(lcr))
        return ly, lcb, lcr

    def alloc_host(self, nbytes):
        """A page-locked uint8 array (wrenc_gpu_alloc_host): transfers from / to it run at PCIe rate.  Freed
        by close(); do not use it afterwards."""
        self.lib.wrenc_gpu_alloc_host.restype = C.c_void_p
        self.lib.wrenc_gpu_alloc_host.argtypes = [C.c_void_p, C.c_size_t]
        ptr = self.lib.wrenc_gpu_alloc_host(self.ctx, nbytes)
        if not ptr:
            raise WrencGpuError(-3, self.lib.wrenc_gpu_last_error(self.ctx).decode())
        self._pinned.append(ptr)
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))

    def alloc_picture_host(self, keys=None):
        """alloc_picture() in page-locked memory (optionally only `keys`)."""
        shapes = alloc_picture(8, 8)
        w, h = self.width, self.height
        dims = {"rec_y": (h, w), "rec_cb": (h // 2, w // 2), "rec_cr": (h // 2, w // 2), "lev_y": (h, w),
                "lev_cb": (h // 2, w // 2), "lev_cr": (h // 2, w // 2), "cu_log2_size": (h // 4, w // 4),
                "luma_mode": (h // 4, w // 4), "chroma_mode": (h // 8, w // 8), "ctu_cost": ((h // 32) * (w // 32),)}
        out = {}
        for k in _PIC_KEYS:
            if keys is not None and k not in keys:
                continue
            dt = shapes[k].dtype
            n = int(np.prod(dims[k])) * dt.itemsize
            out[k] = self.alloc_host(n).view(dt).reshape(dims[k])
        return out

    def download(self, slot, keys=None, out=None):
        """All planes of the record, or only `keys` (the C ABI skips NULL pointers); `out` = arrays to fill
        (e.g. from alloc_picture_host) instead of fresh ones."""
        if out is None:
            out = alloc_picture(self.width, self.height)
        if keys is not None:
            out = {k: v for k, v in out.items() if k in keys}
        pic = Picture(*[_p(out[k]) if k in out else None for k in _PIC_KEYS])
        self._check(self.lib.wrenc_gpu_download(self.ctx, slot, C.byref(pic)))
        return out

    def encode_picture(self, y, cb, cr):
        self.upload(0, y, cb, cr)
        self.encode(0, 1)
        return self.download(0)

    SCHEDULE_AUTO, SCHEDULE_WAVE, SCHEDULE_TEAM = 0, 1, 2

    def set_schedule(self, schedule):
        """How CTUs map to wavefronts (include/wrenc_gpu.h: wrenc_gpu_schedule); results are the same either way."""
        self.lib.wrenc_gpu_set_schedule.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.wrenc_gpu_set_schedule(self.ctx, int(schedule)))

    def last_schedule(self):
        self.lib.wrenc_gpu_last_schedule.argtypes = [C.c_void_p]
        return self.lib.wrenc_gpu_last_schedule(self.ctx)

    def test_set_wave_slots(self, slots):
        """Test entry: the wave-slot count AUTO compares a diagonal with (<= 0: the device's)."""
        self.lib.wrenc_gpu_test_set_wave_slots.argtypes = [C.c_void_p, C.c_longlong]
        self._check(self.lib.wrenc_gpu_test_set_wave_slots(self.ctx, int(slots)))

    def test_scratch_overflows(self):
        """Test entry: workgroups that took a scratch region outside their XCD's partition (0 on MI355X)."""
        n = C.c_longlong(0)
        self.lib.wrenc_gpu_test_scratch_overflows.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        self._check(self.lib.wrenc_gpu_test_scratch_overflows(self.ctx, C.byref(n)))
        return int(n.value)

    def test_head_ranges(self):
        """Test entry: (counts[4], ranges[4][6]) of wrenc_gpu_test_head_ranges (include/wrenc_gpu.h)."""
        counts = (C.c_int * 4)()
        ranges = (C.c_int * 24)()
        self.lib.wrenc_gpu_test_head_ranges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(self.lib.wrenc_gpu_test_head_ranges(self.ctx, counts, ranges))
        return list(counts), np.array(list(ranges), dtype=np.int64).reshape(4, 6)

    def test_avail_tab(self):
        """Test entry: blocks whose table-derived segment availability differs from the reference's rules (must be 0)."""
        n = C.c_int(0)
        self.lib.wrenc_gpu_test_avail_tab.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._check(self.lib.wrenc_gpu_test_avail_tab(self.ctx, C.byref(n)))
        return int(n.value)

    def stats_enable(self, on=True):
        """Per-launch timing events (measurement mode, see include/wrenc_gpu.h)."""
        self.lib.wrenc_gpu_stats_enable.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.wrenc_gpu_stats_enable(self.ctx, 1 if on else 0))

    def last_encode_stats(self):
        t, k, n = C.c_float(), C.c_float(), C.c_int()
        self._check(self.lib.wrenc_gpu_last_encode_stats(self.ctx, C.byref(t), C.byref(k), C.byref(n)))
        return {"total_ms": t.value, "kernel_ms_sum": k.value, "n_launches": n.value}

    def last_encode_kernel_stats(self):
        """Per kernel of the last encode call: {"wave": {...}, "team": {...}} with ms_sum, launches, ctu_pictures."""
        class KS(C.Structure):
            _fields_ = [("ms_sum", C.c_float), ("launches", C.c_int32), ("ctu_pictures", C.c_int64)]
        out = (KS * 2)()
        self.lib.wrenc_gpu_last_encode_kernel_stats.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.lib.wrenc_gpu_last_encode_kernel_stats(self.ctx, out))
        return {n: {"ms_sum": out[i].ms_sum, "launches": out[i].launches, "ctu_pictures": out[i].ctu_pictures}
                for i, n in enumerate(("wave", "team"))}

    def final_pass_mismatches(self):
        v = C.c_longlong()
        self._check(self.lib.wrenc_gpu_final_pass_mismatches(self.ctx, C.byref(v)))
        return v.value

    # ---- building blocks ----
    def _blocks(self, fn, arr):
        arr = np.ascontiguousarray(arr, np.int16)
        count, n, _ = arr.shape
        out = np.zeros_like(arr)
        self._check(fn(self.ctx, _p(arr), int(n).bit_length() - 1, count, _p(out)))
        return out

    def fwd_dct(self, blocks):
        return self._blocks(self.lib.wrenc_gpu_test_fwd_dct, blocks)

    def fwd_dct32(self, blocks, use_mfma, reps=1):
        """(coefficients, kernel ms) of the 32x32 forward transform: v_dot2 code or the i8-MFMA experiment."""
        arr = np.ascontiguousarray(blocks, np.int16)
        assert arr.shape[1:] == (32, 32)
        out = np.zeros_like(arr)
        ms = C.c_float()
        self.lib.wrenc_gpu_test_fwd_dct32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                      C.POINTER(C.c_float)]
        self._check(self.lib.wrenc_gpu_test_fwd_dct32(self.ctx, _p(arr), arr.shape[0], _p(out), int(use_mfma), int(reps),
                                                      C.byref(ms)))
        return out, ms.value

    def inv_dct32(self, blocks, use_mfma, reps=1):
        """(residuals, kernel ms) of the 32x32 inverse transform: v_dot2 code or the i8-MFMA version."""
        arr = np.ascontiguousarray(blocks, np.int16)
        assert arr.shape[1:] == (32, 32)
        out = np.zeros_like(arr)
        ms = C.c_float()
        self.lib.wrenc_gpu_test_inv_dct32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                      C.POINTER(C.c_float)]
        self._check(self.lib.wrenc_gpu_test_inv_dct32(self.ctx, _p(arr), arr.shape[0], _p(out), int(use_mfma), int(reps),
                                                      C.byref(ms)))
        return out, ms.value

    def inv_dct(self, blocks):
        return self._blocks(self.lib.wrenc_gpu_test_inv_dct, blocks)

    def dequantize(self, blocks):
        return self._blocks(self.lib.wrenc_gpu_test_dequantize, blocks)

    def predict_blocks(self, rec_y, rec_cb, rec_cr, items):
        """items: (n, 5) int32 {x, y, log2 luma size, comp (0 luma, 1 Cb+Cr pair, 2 luma 4x4 through the packed predictor
        of the 4x4 leaf search), mode}; returns the list of predicted blocks (luma: (n, n); pair: (2, n/2, n/2))."""
        items = np.ascontiguousarray(items, np.int32).reshape(-1, 5)
        sizes = [((1 << int(q[2])) ** 2) // (2 if q[3] == 1 else 1) for q in items]
        out = np.zeros(int(sum(sizes)), np.uint8)
        planes = [np.ascontiguousarray(a, np.uint8) for a in (rec_y, rec_cb, rec_cr)]
        assert planes[0].shape == (self.height, self.width)
        self.lib.wrenc_gpu_test_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                    C.c_void_p, C.c_size_t]
        self._check(self.lib.wrenc_gpu_test_predict(self.ctx, _p(planes[0]), _p(planes[1]), _p(planes[2]), len(items),
                                                    _p(items), _p(out), out.size))
        res, at = [], 0
        for q, sz in zip(items, sizes):
            n = 1 << int(q[2])
            blk = out[at:at + sz]
            res.append(blk.reshape(2, n // 2, n // 2) if q[3] == 1 else blk.reshape(n, n))
            at += sz
        return res

    def sad_lists(self, rec_y, rec_cb, rec_cr, items):
        """SAD lists (the search's sad_list_angular) of blocks against their own samples in the planes as originals.
        items: (n, 7) {x, y, log2 luma size, comps (1 luma, 2 chroma pair, 3 both), first mode, entries (<= 13), stride};
        entry j = first mode + j * stride (not evaluated beyond 66: its SAD stays 0).  Returns (n, 16) uint32."""
        it = np.ascontiguousarray(items, np.int32).reshape(-1, 7)
        dev = np.zeros((len(it), 5), np.int32)
        dev[:, :3] = it[:, :3]
        dev[:, 3] = np.where(it[:, 3] == 4, 7, 3 + it[:, 3])   # comps 4: the CCLM list of the chroma pair (lanes 0..2: LT, T, L)
        dev[:, 4] = np.where(it[:, 3] == 4, 0, it[:, 4] | (it[:, 5] << 8) | (it[:, 6] << 16))
        out = np.zeros(64 * len(it), np.uint8)
        planes = [np.ascontiguousarray(a, np.uint8) for a in (rec_y, rec_cb, rec_cr)]
        assert planes[0].shape == (self.height, self.width)
        self.lib.wrenc_gpu_test_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                    C.c_void_p, C.c_size_t]
        self._check(self.lib.wrenc_gpu_test_predict(self.ctx, _p(planes[0]), _p(planes[1]), _p(planes[2]), len(dev),
                                                    _p(dev), _p(out), out.size))
        return out.view(np.uint32).reshape(len(it), 16)

    def quantize_p16(self, blocks):
        """4x4 blocks through the packed quantiser of the 4x4 leaf search (four blocks per wavefront)."""
        arr = np.ascontiguousarray(blocks, np.int16)
        count, n, _ = arr.shape
        assert n == 4
        out = np.zeros_like(arr)
        cost = np.zeros(count, np.int64)
        self.lib.wrenc_gpu_test_quantize_p16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._check(self.lib.wrenc_gpu_test_quantize_p16(self.ctx, _p(arr), count, _p(out), _p(cost)))
        return out, cost

    def quantize_pk(self, packs, log2n, nc):
        """Packs of nc candidates (each: luma block, then Cb and Cr of half the side) through the packed quantiser of
        the 8x8 / 16x16 leaf searches.  packs: (n_packs, nc * 1.5 * 4^log2n) int16; returns (levels, cost[n_packs, nc, 2])."""
        arr = np.ascontiguousarray(packs, np.int16)
        n_packs = arr.shape[0]
        assert arr.shape[1] == nc * 3 * (1 << (2 * log2n)) // 2
        out = np.zeros_like(arr)
        cost = np.zeros((n_packs, nc, 2), np.int64)
        self.lib.wrenc_gpu_test_quantize_pk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self._check(self.lib.wrenc_gpu_test_quantize_pk(self.ctx, _p(arr), int(log2n), int(nc), n_packs, _p(out), _p(cost)))
        return out, cost

    def quantize(self, blocks):
        arr = np.ascontiguousarray(blocks, np.int16)
        count, n, _ = arr.shape
        out = np.zeros_like(arr)
        cost = np.zeros(count, np.int64)
        self._check(self.lib.wrenc_gpu_test_quantize(self.ctx, _p(arr), int(n).bit_length() - 1, count,
                                                     _p(out), _p(cost)))
        return out, cost
