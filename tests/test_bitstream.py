"""Host bitstream writer (wrenc_amd/csrc/host, include/wrenc_bitstream.h) on the CPU.

The reference's only end-to-end check is "what a decoder reconstructs from the stream equals the
encoder's reconstruction" (scripts/intergration_test.sh, needs VTM).  Here the test-side parser of
oracle/vvc_parse.cpp decodes the stream back into the record (size map, modes, levels) and
wro_reconstruct_from_record rebuilds the picture; the record comes from the CPU oracle in this file and
from the device in test_gpu_bitstream.py."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from content import content

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC_KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr")


def _roundtrip(rec, w, h, qp, poc=0):
    from wrenc_amd import bitstream as bs
    from oracle import pyoracle as po
    stream = bs.write_parameter_sets(w, h, qp) + bs.write_picture(w, h, qp, poc, rec)
    info = po.parse_stream_info(stream)
    assert info == {"width": w, "height": h, "init_qp": max(qp, 26), "n_pictures": 1}
    back = po.parse_picture(stream, 0)
    assert back["poc_lsb"] == poc & 15 and back["slice_qp"] == qp
    for k in REC_KEYS:
        assert np.array_equal(back[k], rec[k]), k
    return stream, back


def test_library_exports_every_declared_symbol(built):
    from wrenc_amd import bitstream as bs
    txt = open(os.path.join(ROOT, "include", "wrenc_bitstream.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = sorted(set(re.findall(r"\b(wrenc_bs_[a-z0-9_]+)\s*\(", txt)))
    lib = C.CDLL(bs.LIB_PATH)
    assert len(declared) == 5
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(bs.EXPORTED_SYMBOLS) == declared


def test_context_tables_equal_the_reference_fixture(built):
    """initValue / shiftIdx of every live context: the writer's table (cabac.h) and the parser's
    (oracle/vvc_ctx_init.inc) against tests/golden/cabac_ctx_init.json, which make_cabac_ctx.py
    extracted from the reference's cabac_contexts.rs."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "cabac_ctx_init.json")))
    want = []
    for name in g["order"]:
        c = g["contexts"][name]
        assert len(c["init_value"]) == len(c["shift_idx"])
        want += list(zip(c["init_value"], c["shift_idx"]))
    assert len(want) == 253
    txt = open(os.path.join(ROOT, "wrenc_amd", "csrc", "host", "cabac.h")).read()
    body = txt[txt.index("kCtxInit[CTX_COUNT] = {"):]
    body = body[:body.index("};")]
    assert [(int(a), int(b)) for a, b in re.findall(r"\{(\d+), (\d+)\}", body)] == want
    inc = open(os.path.join(ROOT, "oracle", "vvc_ctx_init.inc")).read()
    assert [(int(a), int(b)) for a, b in re.findall(r"\{(\d+), (\d+)\}", inc)] == want
    # Rice parameter table (cabac_contexts.rs:919)
    for path in ("wrenc_amd/csrc/host/slice_data.cpp", "oracle/vvc_parse.cpp"):
        t = open(os.path.join(ROOT, path)).read()
        m = re.search(r"k(?:RiceParams|Rice)\[32\] = \{([^}]*)\}", t)
        assert [int(x) for x in m.group(1).split(",")] == g["c_rice_params"], path


def _diag_scan(lg):
    """6.5.2 / ctu.rs:52-78, restated."""
    n, out, x, y = 1 << lg, [], 0, 0
    while len(out) < n * n:
        while y >= 0:
            if x < n and y < n:
                out.append((x, y))
            y -= 1
            x += 1
        y, x = x, 0
    return out


def test_parser_scan_order(built):
    from oracle import pyoracle as po
    lib = po.lib()
    assert _diag_scan(2) == [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0), (0, 3), (1, 2), (2, 1), (3, 0), (1, 3),
                             (2, 2), (3, 1), (2, 3), (3, 2), (3, 3)]
    for lg in range(4):
        buf = np.zeros(2 << (2 * lg), np.uint8)
        lib.wro_parse_debug_scan(lg, buf.ctypes.data_as(C.c_void_p))
        assert [tuple(p) for p in buf.reshape(-1, 2).tolist()] == _diag_scan(lg)


CASES = [
    ("flat", 64, 64, 32, 2), ("flat", 64, 32, 22, 3), ("ramp", 96, 64, 27, 2), ("ramp", 64, 64, 37, 3),
    ("stripes0", 64, 64, 32, 2), ("stripes90", 64, 64, 32, 2), ("stripes45", 64, 64, 27, 2),
    ("stripes135", 64, 64, 27, 3), ("stripes20", 96, 64, 32, 2), ("stripes70", 64, 96, 12, 3),
    ("stripes110", 64, 64, 22, 2), ("stripes160", 64, 64, 42, 3), ("checker", 64, 64, 32, 3),
    ("checker", 96, 96, 45, 2), ("noise", 64, 64, 37, 2), ("noise", 64, 64, 51, 3), ("noise", 32, 32, 27, 3),
    ("noise", 32, 32, 4, 3), ("noise", 64, 32, 63, 1), ("cclm", 128, 64, 32, 2), ("cclm", 64, 64, 22, 3),
    ("cclm", 64, 64, 42, 1), ("extremes", 64, 64, 32, 2), ("extremes", 64, 64, 18, 3), ("cclm", 32, 256, 32, 2),
    ("stripes45", 256, 32, 32, 2), ("noise", 32, 128, 40, 3), ("ramp", 160, 32, 32, 0),
]


@pytest.mark.parametrize("kind,w,h,qp,depth", CASES)
def test_stream_decodes_to_the_encoder_reconstruction(built, kind, w, h, qp, depth):
    from oracle import pyoracle as po
    y, cb, cr = content(kind, w, h, 11)
    rec = po.encode_picture(y, cb, cr, qp, depth)
    _, back = _roundtrip(rec, w, h, qp, poc=5)
    ry, rcb, rcr = po.reconstruct_from_record(back, qp)
    assert np.array_equal(ry, rec["rec_y"]) and np.array_equal(rcb, rec["rec_cb"]) and np.array_equal(rcr, rec["rec_cr"])
    sy, scb, scr = po.spec_decode_record(back, qp)          # the decoder that shares no code with the oracle
    assert np.array_equal(sy, rec["rec_y"]) and np.array_equal(scb, rec["rec_cb"]) and np.array_equal(scr, rec["rec_cr"])


def _random_record(rng, w, h, qp, amp):
    """A record no search would produce: random quadtree, every luma mode, every chroma signalling, levels
    from the oracle's dependent quantiser on random coefficients (valid parities, state 0 at each TB)."""
    from oracle import pyoracle as po
    size = np.zeros((h // 4, w // 4), np.uint8)
    lm = np.zeros((h // 4, w // 4), np.uint8)
    cm = np.zeros((h // 8, w // 8), np.uint8)
    lev = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]

    def levels(n):
        kind = rng.integers(0, 5)
        if kind == 0:
            return np.zeros((n, n), np.int16)
        coef = np.zeros((n, n), np.int64)
        if kind == 1:      # a single coefficient somewhere
            coef[rng.integers(0, n), rng.integers(0, n)] = rng.integers(-amp, amp + 1)
        elif kind == 2:    # low-frequency corner
            k = max(1, n // 4)
            coef[:k, :k] = rng.integers(-amp, amp + 1, (k, k))
        elif kind == 3:    # sparse everywhere
            coef = rng.integers(-amp, amp + 1, (n, n)) * (rng.random((n, n)) < 0.08)
        else:              # dense: exhausts the context-coded bin budget
            coef = rng.integers(-amp, amp + 1, (n, n))
        return po.quantize(coef.astype(np.int16), qp, viterbi=True)

    def leaf(x, y, lg, tree):
        n = 1 << lg
        if tree != 2:
            size[y // 4:(y + n) // 4, x // 4:(x + n) // 4] = lg
            lm[y // 4:(y + n) // 4, x // 4:(x + n) // 4] = rng.integers(0, 67)
            lev[0][y:y + n, x:x + n] = levels(n)
        if tree != 1:
            luma_ref = lm[(y + n // 2) // 4, (x + n // 2) // 4]
            pick = rng.integers(0, 4)
            mode = luma_ref if pick < 2 else (81 + rng.integers(0, 3) if pick == 2 else
                                              [0, 50, 18, 1, 66][rng.integers(0, 5)])
            if mode == 66 and luma_ref not in (0, 50, 18, 1):
                mode = luma_ref      # 66 is only expressible as the replacement of a clash
            cm[y // 8:(y + n) // 8, x // 8:(x + n) // 8] = mode
            for c in (1, 2):
                lev[c][y // 2:(y + n) // 2, x // 2:(x + n) // 2] = levels(n // 2)

    def tree(x, y, lg):
        if lg > 3 and rng.random() < 0.6:
            for i in range(4):
                tree(x + (i & 1) * (1 << (lg - 1)), y + (i >> 1) * (1 << (lg - 1)), lg - 1)
        elif lg == 3 and rng.random() < 0.4:
            for i in range(4):
                leaf(x + (i & 1) * 4, y + (i >> 1) * 4, 2, 1)
            leaf(x, y, 3, 2)
        else:
            leaf(x, y, lg, 0)

    for y in range(0, h, 32):
        for x in range(0, w, 32):
            tree(x, y, 5)
    return {"cu_log2_size": size, "luma_mode": lm, "chroma_mode": cm, "lev_y": lev[0], "lev_cb": lev[1], "lev_cr": lev[2]}


@pytest.mark.parametrize("seed,qp,amp", [(1, 32, 60), (2, 22, 400), (3, 37, 4000), (4, 12, 30000), (5, 51, 30000),
                                        (6, 27, 8), (7, 0, 2000), (8, 63, 32000)])
def test_random_records_round_trip(built, seed, qp, amp):
    """Syntax coverage beyond what the search picks: every mode, clashing chroma modes, escape-coded
    levels, TBs that run out of context-coded bins."""
    rng = np.random.default_rng(seed)
    w, h = 96, 64
    rec = _random_record(rng, w, h, qp, amp)
    stream, _ = _roundtrip(rec, w, h, qp, poc=seed)
    # the reference's emulation prevention (nal.rs:156-182): outside the last three payload bytes no
    # 00 00 0x (x <= 3) survives; the decoder removes every inserted 03 again (checked by the round trip)
    data = np.frombuffer(stream, np.uint8)
    starts = [i + 3 for i in range(len(data) - 2) if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 1]
    assert len(starts) == 5
    for k, b in enumerate(starts):
        e = starts[k + 1] - 6 if k + 1 < len(starts) else len(data)
        body = data[b + 2:e]
        for i in range(len(body) - 5):
            assert not (body[i] == 0 and body[i + 1] == 0 and body[i + 2] <= 2), (k, i)


def test_production_arithmetic_coder_writes_the_bit_serial_coders_bytes(built):
    """cabac.h holds two encoders: the bit-at-a-time form of bool_coder.rs:136-296 and the register /
    byte-output form the product runs.  Same bins in, same bytes out, over records that exercise long
    carry chains (dense noise) and almost-empty pictures."""
    from wrenc_amd import bitstream as bs
    from oracle import pyoracle as po
    spec = C.CDLL(os.path.join(os.path.dirname(bs.LIB_PATH), "libwrenc_host_spec.so"))
    spec.wrenc_bs_picture_bound.restype = C.c_size_t
    cases = [("noise", 64, 64, 22, 3, 1), ("noise", 64, 64, 4, 2, 2), ("flat", 64, 64, 40, 1, 3), ("cclm", 96, 64, 30, 3, 4),
             ("checker", 64, 96, 37, 2, 5)]
    records = []
    for kind, w, h, qp, depth, seed in cases:
        y, cb, cr = content(kind, w, h, seed)
        records.append((w, h, qp, po.encode_picture(y, cb, cr, qp, depth)))
    rng = np.random.default_rng(99)
    for qp, amp in ((30, 500), (10, 30000), (50, 20000)):
        records.append((96, 64, qp, _random_record(rng, 96, 64, qp, amp)))
    for w, h, qp, rec in records:
        want = bs.write_picture(w, h, qp, 2, rec)
        arrs = [np.ascontiguousarray(rec[k]) for k in REC_KEYS]
        r = bs._Record(*[a.ctypes.data for a in arrs])
        cap = spec.wrenc_bs_picture_bound(w, h)
        buf = np.zeros(cap, np.uint8)
        n = C.c_size_t()
        rc = spec.wrenc_bs_write_picture(w, h, qp, 2, C.byref(r), C.c_void_p(buf.ctypes.data), C.c_size_t(cap), C.byref(n))
        assert rc == 0
        assert buf[:n.value].tobytes() == want


def test_multi_picture_stream_and_qp_below_26(built):
    from wrenc_amd import bitstream as bs
    from oracle import pyoracle as po
    w, h, qp = 64, 32, 19
    recs = []
    stream = bs.write_parameter_sets(w, h, qp)
    for poc in (0, 1, 17):
        y, cb, cr = content("cclm" if poc else "stripes20", w, h, poc)
        rec = po.encode_picture(y, cb, cr, qp, 2)
        recs.append(rec)
        stream += bs.write_picture(w, h, qp, poc, rec)
        assert bs.last_slice_data_bits() > 0
    info = po.parse_stream_info(stream)
    assert info == {"width": w, "height": h, "init_qp": 26, "n_pictures": 3}
    for i, poc in enumerate((0, 1, 17)):
        back = po.parse_picture(stream, i)
        assert back["poc_lsb"] == poc & 15 and back["slice_qp"] == qp
        for k in REC_KEYS:
            assert np.array_equal(back[k], recs[i][k]), (i, k)
    with pytest.raises(ValueError):
        po.parse_picture(stream, 3)


def test_parameter_sets_known_bytes(built):
    """The parameter sets are a pure function of (width, height, QP): byte-for-byte what the field
    sequence of vps/sps/pps_encoder.rs gives for the reference's defaults (worked out by hand in
    DESIGN.md 8), for CIF at QP 32."""
    from wrenc_amd import bitstream as bs
    s = bs.write_parameter_sets(352, 288, 32)
    vps = bytes([0, 0, 0, 0, 0, 1, 0x01, 0x71,            # layer 1, VPS_NUT 14 << 3 | tid+1
                 0x80, 0x01, 0x20,                        # id 8, 1 layer, 1 sublayer, layer id 9, alignment
                 0x00, 0x00, 0x03, 0x00, 0x00,             # PTL: 4 zero bytes, one 03 inserted
                 0x89, 0x2a, 0x20])                       # ue(0) ue(8) ue(4) ue(1) 0 0 stop
    assert s[:len(vps)] == vps
    assert s.count(bytes([0, 0, 0, 0, 0, 1])) == 3
    assert s == bs.write_parameter_sets(352, 288, 32)
    assert s != bs.write_parameter_sets(352, 288, 33) and s[:60] == bs.write_parameter_sets(352, 288, 33)[:60]
    # QPs up to 26 share init_qp 26 (pps.rs:183-187): identical parameter sets
    assert bs.write_parameter_sets(64, 64, 3) == bs.write_parameter_sets(64, 64, 26)


def test_error_paths(built):
    from wrenc_amd import bitstream as bs
    from oracle import pyoracle as po
    lib = bs.load_library()
    n = C.c_size_t()
    assert lib.wrenc_bs_write_parameter_sets(100, 64, 32, None, 0, C.byref(n)) == bs.EINVAL
    assert lib.wrenc_bs_write_parameter_sets(64, 64, 64, None, 0, C.byref(n)) == bs.EINVAL
    assert lib.wrenc_bs_write_parameter_sets(64, 64, 32, None, 0, C.byref(n)) == bs.ENOSPC and n.value > 50
    buf = np.zeros(n.value, np.uint8)
    assert lib.wrenc_bs_write_parameter_sets(64, 64, 32, buf.ctypes.data, n.value - 1, C.byref(n)) == bs.ENOSPC
    assert lib.wrenc_bs_write_parameter_sets(64, 64, 32, buf.ctypes.data, n.value, C.byref(n)) == bs.OK
    with pytest.raises(bs.BitstreamError):
        bs.write_parameter_sets(64, 60, 32)
    y, cb, cr = content("noise", 32, 32, 3)
    rec = po.encode_picture(y, cb, cr, 30, 3)
    whole = bs.write_picture(32, 32, 30, 0, rec)
    assert len(whole) > 20
    # too small a buffer: the size that is needed comes back, and a buffer of that size is enough
    arrs = [np.ascontiguousarray(rec[k]) for k in REC_KEYS]
    r = bs._Record(*[a.ctypes.data for a in arrs])
    small = np.zeros(len(whole), np.uint8)
    assert lib.wrenc_bs_write_picture(32, 32, 30, 0, C.byref(r), small.ctypes.data, 10, C.byref(n)) == bs.ENOSPC
    assert n.value == len(whole)
    assert lib.wrenc_bs_write_picture(32, 32, 30, 0, C.byref(r), small.ctypes.data, n.value, C.byref(n)) == bs.OK
    assert small.tobytes() == whole
    with pytest.raises(ValueError):
        bs.write_picture(64, 32, 30, 0, rec)             # planes of the wrong shape
    # a level whose parity contradicts the quantiser state (the reference asserts, ctu_encoder.rs:1975)
    bad = {k: np.array(rec[k]) for k in REC_KEYS}
    ys, xs = np.nonzero(bad["lev_y"])
    assert len(ys) > 0
    bad["lev_y"][ys[0], xs[0]] += 1
    with pytest.raises(bs.BitstreamError) as e:
        bs.write_picture(32, 32, 30, 0, bad)
    assert e.value.code == bs.EDATA
    # a size map that is not a quadtree
    bad = {k: np.array(rec[k]) for k in REC_KEYS}
    bad["cu_log2_size"][0, 0] = 6
    with pytest.raises(bs.BitstreamError) as e:
        bs.write_picture(32, 32, 30, 0, bad)
    assert e.value.code == bs.EDATA
    bad["cu_log2_size"][:] = 5
    bad["cu_log2_size"][1, 1] = 4
    with pytest.raises(bs.BitstreamError) as e:
        bs.write_picture(32, 32, 30, 0, bad)
    assert e.value.code == bs.EDATA
    # a chroma mode the chroma syntax cannot express
    bad = {k: np.array(rec[k]) for k in REC_KEYS}
    bad["cu_log2_size"][:] = 5
    bad["luma_mode"][:] = 30
    bad["chroma_mode"][:] = 31
    with pytest.raises(bs.BitstreamError) as e:
        bs.write_picture(32, 32, 30, 0, bad)
    assert e.value.code == bs.EDATA


def test_parser_rejects_damaged_streams(built):
    from wrenc_amd import bitstream as bs
    from oracle import pyoracle as po
    y, cb, cr = content("checker", 64, 32, 3)
    rec = po.encode_picture(y, cb, cr, 32, 2)
    stream = bytearray(bs.write_parameter_sets(64, 32, 32) + bs.write_picture(64, 32, 32, 0, rec))
    assert po.parse_stream_info(bytes(stream))["n_pictures"] == 1
    with pytest.raises(ValueError):
        po.parse_stream_info(bytes(stream[:40]))          # SPS cut short
    damaged = bytearray(stream)
    damaged[12] ^= 0x10                                   # a PTL byte of the VPS
    with pytest.raises(ValueError):
        po.parse_stream_info(bytes(damaged))
    damaged = bytearray(stream)
    damaged[-9] ^= 0x5a                                   # CABAC payload: decodes to something else or fails
    try:
        back = po.parse_picture(bytes(damaged), 0)
        assert any(not np.array_equal(back[k], rec[k]) for k in REC_KEYS)
    except ValueError:
        pass


def test_streams_match_the_recorded_hashes(built):
    """Regression pin (tests/golden/make_stream_hashes.py): the bytes written for eight oracle records have
    not changed since they were recorded."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_stream_hashes", os.path.join(ROOT, "tests", "golden", "make_stream_hashes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "stream_hashes.json")))
    assert len(want) == len(mod.CASES)
    for case in mod.CASES:
        sha, n = mod.stream_hash(*case)
        assert want["%s_%dx%d_qp%d_d%d" % case] == {"sha256": sha, "bytes": n}, case


@pytest.mark.skipif(not os.path.exists("/root/reference/src/sps.rs"), reason="reference not mounted")
def test_parameter_set_defaults_still_read_as_assumed():
    """headers.cpp writes the parameter sets from the defaults of the reference's constructors.  Every default that
    decides a written value or a taken branch is listed here next to where the reference sets it; the test fails if
    the reference's source no longer says so."""
    src = "/root/reference/src/"
    assumed = {
        "main.rs": ["VideoParameterSet::new(8,", "SequenceParameterSet::new(1, 8,", "PictureParameterSet::new(1, &sps",
                    "write_byte_stream_nal_unit_bits(1, NALUnitType::VPS_NUT, 0", "write_byte_stream_nal_unit_bits(9, NALUnitType::SPS_NUT, 0",
                    "write_byte_stream_nal_unit_bits(9, NALUnitType::PPS_NUT, 0", "let nuh_layer_id = 9;", "NALUnitType::IDR_W_RADL"],
        "vps.rs": ["max_layers: 1,", "max_sublayers: 1,", "VpsLayer::new(9)", "each_layer_is_an_ols: false,", "num_ptls: 1,",
                   "default_ptl_dpb_hrd_max_tid_flag: true,", "general_timing_hrd_parameters: None,", "ptl_max_tids: vec![1, 2],",
                   "extension_data: vec![],"],
        "ptl.rs": ["general_level_idc: 0,", "general_profile_idc: 0,", "general_constraints_info: None,", "ptl_num_sub_profiles: 0,"],
        "dpb.rs": ["max_tid: 1,", "max_dec_pic_buffering: 8,", "max_num_reorder_pics: 4,", "max_latency_increase: 1,"],
        "reference_picture.rs": ["num_ref_entries: 3,", "abs_delta_poc_st: vec![0, 2, 3],", "strp_entry_sign_flag: vec![lx == 0; 3],",
                                 "num_ref_pic_list: 1,", "st_ref_pic_flag: vec![true; 3],"],
        "sps.rs": ["max_sublayers: 1,", "chroma_format: ChromaFormat::YCbCr420,", "log2_ctu_size: 5,", "ptl_dpb_hrd_params_present_flag: true,",
                   "conformance_window: None,", "subpic_info: None,", "bitdepth: 8,", "log2_max_pic_order_cnt_lsb: 4,",
                   "log2_min_luma_coding_block_size: 2,", "transform_skip_enabled_flag: true,", "log2_transform_skip_max_size: 5,",
                   "bdpcm_enabled_flag: false,", "mts_enabled_flag: true,", "explicit_mts_intra_enabled_flag: true,",
                   "explicit_mts_inter_enabled_flag: true,", "lfnst_enabled_flag: false,", "joint_cbcr_enabled_flag: false,",
                   "same_qp_table_for_chroma_flag: true,", "QpTable::new(bit_depth, 63, 0)", "sao_enabled_flag: false,",
                   "alf_enabled_flag: false,", "idr_rpl_present_flag: false,", "rpl1_same_as_rpl0_flag: false,",
                   "six_minus_max_num_merge_cand: 0,", "log2_parallel_merge_level: 2,", "cclm_enabled_flag: true,",
                   "palette_enabled_flag: false,", "min_qp_prime_ts: 0,", "ibc_enabled_flag: false,", "ladf_parameters: None,",
                   "dep_quant_enabled_flag: true,", "sign_data_hiding_enabled_flag: false,", "vui_parameters_present_flag: false,"],
        "partition.rs": ["log2_diff_min_qt_min_cb_intra_slice_luma: 0,", "max_mtt_hierarchy_depth_intra_slice_luma: 0,",
                         "qtbtt_dual_tree_intra_flag: false,", "max_mtt_hierarchy_depth_inter_slice: 0,"],
        "pps.rs": ["no_pic_partition_flag: true,", "cabac_init_present_flag: false,", "num_ref_idx_default_active: [3, 3],",
                   "qp.max(26)", "cu_qp_delta_enabled_flag: true,", "chroma_tool_offsets_present_flag: false,",
                   "deblocking_filter_control_present_flag: true,", "deblocking_filter_override_enabled_flag: false,",
                   "deblocking_filter_disabled_flag: true,", "picture_header_extension_present_flag: false,"],
        "picture_header.rs": ["gdr_or_irap_pic_flag: true,", "non_ref_pic_flag: false,", "gdr_pic_flag: false,", "inter_slice_allowed_flag: !intra,",
                              "pic_order_cnt_lsb: poc & 0b1111,", "cu_qp_delta_subdiv_intra_slice: 0,"],
        "slice_header.rs": ["ph_in_sh: None,", "slice_type: SliceType::I,", "no_output_of_prior_pics_flag: false,", "qp - ectx.slice_qp_y",
                            "dep_quant_used_flag: true,"],
        "nal.rs": ["let header_bytes: [u8; 3] = [0, 0, 0];", "let start_code_prefix_one_3bytes: [u8; 3] = [0, 0, 1];",
                   "while idx + 3 < bytes.len()", "bytes[idx + 2] <= 3"],
    }
    for name, items in assumed.items():
        txt = open(src + name).read()
        for item in items:
            assert item in txt, (name, item)
