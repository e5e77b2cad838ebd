"""Command lines (the native program wrenc_amd/csrc/host/wrenc and its Python twin wrenc_amd/cli.py): the
reference's options (main.rs:85-115) and its error behaviour (message on stderr, exit status 0:
main.rs:127-133)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from content import content

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


NATIVE = os.path.join(ROOT, "wrenc_amd", "csrc", "host", "wrenc")
FRONT_ENDS = ["native", "python"]


def _run(front, args, stdin=None):
    cmd = [NATIVE] if front == "native" else [sys.executable, "-m", "wrenc_amd.cli"]
    return subprocess.run(cmd + args, cwd=ROOT, input=stdin, capture_output=True, timeout=600)


@pytest.mark.parametrize("front", FRONT_ENDS)
def test_argument_errors_print_and_exit_zero(built, tmp_path, front):
    out = str(tmp_path / "o.vvc")
    base = ["-i", str(tmp_path / "missing.yuv"), "-o", out, "--num-pictures", "1", "--qp", "32"]
    r = _run(front, base + ["--input-size", "64x64", "--output-size", "64by64"])
    assert r.returncode == 0 and b"error: Invalid output-size: 64by64" in r.stderr
    r = _run(front, base + ["--input-size", "x", "--output-size", "64x64"])
    assert r.returncode == 0 and b"error: Invalid input-size: x" in r.stderr
    r = _run(front, base + ["--input-size", "64x64", "--output-size", "64x64", "--extra-params", "a"])
    assert r.returncode == 0 and b"error: Invalid extra-params: a" in r.stderr
    r = _run(front, base + ["--input-size", "64x64", "--output-size", "64x60"])
    assert r.returncode == 0 and b"multiple of the 32x32 CTU" in r.stderr
    r = _run(front, base + ["--input-size", "64x64", "--output-size", "64x64"])
    assert r.returncode == 0 and b"error: failed to open input file" in r.stderr
    assert not os.path.exists(out) or os.path.getsize(out) == 0


@pytest.mark.parametrize("front", FRONT_ENDS)
def test_without_a_gpu_the_command_fails_loudly(built, tmp_path, front):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    src = tmp_path / "in.yuv"
    src.write_bytes(bytes(64 * 64 * 3 // 2))
    r = _run(front, ["-i", str(src), "-o", str(tmp_path / "o.vvc"), "--input-size", "64x64", "--output-size", "64x64",
              "--num-pictures", "1", "--qp", "32"])
    assert r.returncode == 101 and r.stderr.startswith(b"error: ")    # no CPU fallback, and not a success status


@pytest.mark.gpu
@pytest.mark.parametrize("front", FRONT_ENDS)
def test_encodes_a_sequence_like_the_reference_binary(built, tmp_path, front):
    """3 pictures through files, then 2 through stdin/stdout: the stream parses, every picture decodes to the
    record of a direct encode, and --reconst holds the decoder-side reconstruction (the reference's
    integration test compares exactly these two files)."""
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    w, h, qp, depth = 96, 64, 30, 2
    frames = [content(k, w, h, i) for i, k in enumerate(("cclm", "stripes20", "noise"))]
    raw = b"".join(p.tobytes() for f in frames for p in f)
    src, out, rec = tmp_path / "in.yuv", tmp_path / "out.vvc", tmp_path / "rec.yuv"
    src.write_bytes(raw)
    r = _run(front, ["-i", str(src), "-o", str(out), "-r", str(rec), "--input-size", "%dx%d" % (w, h), "--output-size",
              "%dx%d" % (w, h), "--num-pictures", "3", "--qp", str(qp), "--max-split-depth", str(depth), "--batch", "2"])
    assert r.returncode == 0 and r.stderr == b"", r.stderr
    stream = out.read_bytes()
    assert po.parse_stream_info(stream) == {"width": w, "height": h, "init_qp": qp, "n_pictures": 3}
    recon = np.frombuffer(rec.read_bytes(), np.uint8)
    per = w * h * 3 // 2
    assert recon.size == 3 * per
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    for i, f in enumerate(frames):
        want = enc.encode_picture(*f)
        back = po.parse_picture(stream, i)
        assert back["poc_lsb"] == i and back["slice_qp"] == qp
        for k in ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr"):
            assert np.array_equal(back[k], want[k]), (i, k)
        ry, rcb, rcr = po.spec_decode_record(back, qp)      # the decoder written from H.266 alone
        oy, ocb, ocr = po.reconstruct_from_record(back, qp)
        assert np.array_equal(ry, oy) and np.array_equal(rcb, ocb) and np.array_equal(rcr, ocr)
        got = recon[i * per:(i + 1) * per]
        assert np.array_equal(got[:w * h], ry.ravel())
        assert np.array_equal(got[w * h:w * h + w * h // 4], rcb.ravel())
        assert np.array_equal(got[w * h + w * h // 4:], rcr.ravel())
    enc.close()
    r = _run(front, ["-i", "-", "-o", "-", "--input-size", "%dx%d" % (w, h), "--output-size", "%dx%d" % (w, h),
              "--num-pictures", "2", "--qp", str(qp), "--max-split-depth", str(depth)], stdin=raw)
    assert r.returncode == 0 and po.parse_stream_info(r.stdout)["n_pictures"] == 2
    two = po.parse_picture(r.stdout, 1)
    one = po.parse_picture(stream, 1)
    assert all(np.array_equal(two[k], one[k]) for k in ("lev_y", "luma_mode"))
    # input shorter than --num-pictures asks for
    r = _run(front, ["-i", str(src), "-o", str(out), "--input-size", "%dx%d" % (w, h), "--output-size", "%dx%d" % (w, h),
              "--num-pictures", "4", "--qp", str(qp)])
    assert r.returncode == 0 and b"error: input ended after 3 of 4 pictures" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("front", FRONT_ENDS)
def test_extra_params_reach_the_encoder(built, tmp_path, front):
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    w, h, qp = 64, 64, 30
    f = content("noise", w, h, 2)
    src = tmp_path / "in.yuv"
    src.write_bytes(b"".join(p.tobytes() for p in f))
    extra = "a=0.05,lambda_mul_dq_trellis=3.0,quant_lambda_mul_trellis=3.0,unknown_knob=3"
    base = ["-i", str(src), "--input-size", "64x64", "--output-size", "64x64", "--num-pictures", "1", "--qp", str(qp)]
    r = _run(front, base + ["-o", str(tmp_path / "x.vvc"), "--extra-params", extra])
    assert r.returncode == 0 and r.stderr == b"", r.stderr
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=3, extra_params=extra)
    want = enc.encode_picture(*f)
    enc.close()
    back = po.parse_picture((tmp_path / "x.vvc").read_bytes(), 0)
    for k in ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr"):
        assert np.array_equal(back[k], want[k]), k
    r = _run(front, base + ["-o", str(tmp_path / "y.vvc")])
    assert (tmp_path / "y.vvc").read_bytes() != (tmp_path / "x.vvc").read_bytes()
    r = _run(front, base + ["-o", str(tmp_path / "z.vvc"), "--extra-params", "qp_div_dq_trellis=fast"])
    assert r.returncode == 101 and r.stderr.startswith(b"error: ")     # the reference's parse().unwrap() panics


@pytest.mark.gpu
def test_native_and_python_front_ends_write_the_same_bytes(built, tmp_path):
    """7 pictures in batches of 2 (both front ends alternate two sets of slots, the last batch is short) and
    in one batch of 7: four identical streams and reconstructions."""
    w, h = 64, 96
    frames = [content(k, w, h, i) for i, k in enumerate(("cclm", "noise", "ramp", "checker", "stripes70", "flat", "extremes"))]
    src = tmp_path / "in.yuv"
    src.write_bytes(b"".join(p.tobytes() for f in frames for p in f))
    outs = []
    for front in FRONT_ENDS:
        for batch in ("2", "7"):
            out, rec = tmp_path / ("%s%s.vvc" % (front, batch)), tmp_path / ("%s%s.yuv" % (front, batch))
            r = _run(front, ["-i", str(src), "-o", str(out), "-r", str(rec), "--input-size", "64x96", "--output-size", "64x96",
                             "--num-pictures", "7", "--qp", "27", "--max-split-depth", "3", "--batch", batch, "--threads", "3"])
            assert r.returncode == 0 and r.stderr == b"", r.stderr
            outs.append((out.read_bytes(), rec.read_bytes()))
    # the native program over two contexts (here both on device 0: batches alternate between them)
    for batch in ("1", "2", "3"):
        out, rec = tmp_path / ("multi%s.vvc" % batch), tmp_path / ("multi%s.yuv" % batch)
        r = _run("native", ["-i", str(src), "-o", str(out), "-r", str(rec), "--input-size", "64x96", "--output-size", "64x96",
                            "--num-pictures", "7", "--qp", "27", "--max-split-depth", "3", "--batch", batch, "--devices", "0,0",
                            "--verbose"])
        assert r.returncode == 0 and b"2 GPU context(s)" in r.stderr, r.stderr
        outs.append((out.read_bytes(), rec.read_bytes()))
    assert len(outs[0][0]) > 1000 and len(outs[0][1]) == 7 * w * h * 3 // 2
    for o in outs[1:]:
        assert o == outs[0]
    # no pictures: the parameter sets alone
    for front in FRONT_ENDS:
        r = _run(front, ["-i", str(src), "-o", str(tmp_path / "e.vvc"), "--input-size", "64x96", "--output-size", "64x96",
                         "--num-pictures", "0", "--qp", "27"])
        assert r.returncode == 0 and r.stderr == b"", r.stderr
        assert (tmp_path / "e.vvc").read_bytes() == outs[0][0][:len((tmp_path / "e.vvc").read_bytes())]
        assert 60 < len((tmp_path / "e.vvc").read_bytes()) < 200
    r = _run("native", ["-i", str(src), "-o", str(tmp_path / "x.vvc"), "--input-size", "64x96", "--output-size", "64x96",
                        "--num-pictures", "7", "--devices", "0,x"])
    assert r.returncode == 0 and b"error: Invalid devices: 0,x" in r.stderr


@pytest.mark.gpu
def test_a_run_that_ends_on_smaller_batches_writes_the_same_bytes(built, tmp_path):
    """The native program ends a run whose host tail is heavy on smaller batches (a half, a quarter, a quarter of --batch);
    forced here (--ramp-down always) on 21 pictures in batches of 8 -- 8, 8 (the run is not over), then 4, 2, 2 ... down to the
    last picture -- with tokens and with the compact record: the streams and reconstructions are those of plain batches."""
    w, h = 64, 64
    kinds = ("cclm", "noise", "ramp", "checker", "stripes70", "flat", "extremes")
    frames = [content(kinds[i % 7], w, h, i) for i in range(21)]
    src = tmp_path / "in.yuv"
    src.write_bytes(b"".join(p.tobytes() for f in frames for p in f))
    outs = []
    for extra in (["--ramp-down", "never"], ["--ramp-down", "always"], ["--ramp-down", "always", "--tokens", "off"],
                  ["--ramp-down", "always", "--batch", "5"]):
        out, rec = tmp_path / ("o%d.vvc" % len(outs)), tmp_path / ("o%d.yuv" % len(outs))
        r = _run("native", ["-i", str(src), "-o", str(out), "-r", str(rec), "--input-size", "64x64", "--output-size", "64x64",
                            "--num-pictures", "21", "--qp", "30", "--max-split-depth", "2", "--batch", "8", "--threads", "3", "--verbose"] + extra)
        assert r.returncode == 0 and b"21 pictures" in r.stderr, r.stderr
        outs.append((out.read_bytes(), rec.read_bytes(), r.stderr.count(b"read back (")))
    assert outs[0][2] == 3 and outs[1][2] > 3 and outs[2][2] > 3      # 8 + 8 + 5 pictures against more, smaller batches
    for o in outs[1:]:
        assert o[:2] == outs[0][:2]
    r = _run("native", ["-i", str(src), "-o", str(tmp_path / "x.vvc"), "--input-size", "64x64", "--output-size", "64x64",
                        "--num-pictures", "21", "--ramp-down", "sometimes"])
    assert r.returncode == 0 and b"error: Invalid ramp-down: sometimes" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("front", FRONT_ENDS)
def test_a_failure_inside_the_search_is_not_status_zero(built, tmp_path, front):
    """argument / I/O errors exit 0 like the reference (main.rs:127-133), but where the reference
    panics -- here a quantised level that overflows the 1024-entry rate tables (block_splitter.rs:453), reported by
    the library as WRENC_GPU_ELEVEL -- the status is 101 (Rust's panic status), never 0 with a truncated stream."""
    yy, xx = np.indices((32, 32))
    y = (((xx + yy) & 1) * 255).astype(np.uint8)
    c = (((xx[:16, :16] + yy[:16, :16]) & 1) * 255).astype(np.uint8)
    src = tmp_path / "in.yuv"
    src.write_bytes(y.tobytes() + c.tobytes() + c.tobytes())
    r = _run(front, ["-i", str(src), "-o", str(tmp_path / "o.vvc"), "--input-size", "32x32", "--output-size", "32x32",
              "--num-pictures", "1", "--qp", "0", "--max-split-depth", "2"])
    assert r.returncode == 101, (r.returncode, r.stderr)
    assert b"error: " in r.stderr and b"1024" in r.stderr
