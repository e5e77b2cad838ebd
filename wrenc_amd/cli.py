"""Command line of the encoder: the reference's `wrenc` options (main.rs:85-115) over the MI355X search
and the host bitstream writer.

    python -m wrenc_amd.cli -i in.yuv -o out.vvc --input-size 1920x1088 --output-size 1920x1088 \
        --num-pictures 30 --qp 32 --max-split-depth 2 [--reconst rec.yuv]

Flow of main.rs:223-402: VPS, SPS, PPS once; then per picture read Y, Cb, Cr (8-bit 4:2:0 at the output
size), search + final pass (on the GPU, `--batch` pictures at a time: they are independent IDR pictures),
picture header NAL + slice NAL, optionally the reconstruction.  `-` means stdin / stdout.  Like the
reference, argument and I/O errors print `error: ...` on stderr and end the process with status 0
(main.rs:127-133,171-191); a failure inside the search or the stream writer (HIP error, a level that
overflows the rate tables: the reference panics there, block_splitter.rs:453) ends it with status 101,
Rust's panic status.  There is no CPU path: without an MI355X the command fails.
"""
import argparse
import sys
import time
from concurrent.futures import ThreadPoolExecutor


def _die(msg):
    sys.stderr.write("error: %s\n" % msg)
    sys.exit(0)     # main.rs:132: process::exit(0) on argument and I/O errors


def _fatal(msg):
    sys.stderr.write("error: %s\n" % msg)
    sys.exit(101)   # where the reference panics (a truncated stream must not come with status 0)


def _size(text, what):
    parts = text.split("x")
    try:
        w, h = [int(p) for p in parts]
    except ValueError:
        w = h = -1
    if len(parts) != 2 or w <= 0 or h <= 0:
        _die("Invalid %s: %s" % (what, text))
    return w, h


def _read_into(f, arr):
    """Fill the uint8 array from the file; False at end of input."""
    view = memoryview(arr)
    got = 0
    while got < len(view):
        n = f.readinto(view[got:])
        if not n:
            return False
        got += n
    return True


def main(argv=None):
    ap = argparse.ArgumentParser(prog="wrenc_amd", description="VVC all-intra encoder (MI355X search, host CABAC)")
    ap.add_argument("-i", "--input", required=True, help="Path to input raw video")
    ap.add_argument("-o", "--output", required=True, help="Path to output bitstream")
    ap.add_argument("-r", "--reconst", help="Path to reconstructed frames")
    ap.add_argument("--input-size", required=True, help="Input video resolution (WIDTHxHEIGHT)")
    ap.add_argument("--output-size", required=True, help="Output video resolution (WIDTHxHEIGHT)")
    ap.add_argument("--num-pictures", required=True, type=int, help="Number of pictures to encode")
    ap.add_argument("--qp", type=int, help="Fixed quantization parameter for entire video stream")
    ap.add_argument("--max-split-depth", type=int, default=3, help="Max split depth of coding trees to search")
    ap.add_argument("--extra-params", help="Extra parameters (PARAM1=VAL1[,PARAM2=VAL2,...])")
    ap.add_argument("--batch", type=int, default=64, help="pictures searched per GPU call (not in the reference)")
    ap.add_argument("--device", type=int, default=0, help="HIP device ordinal (not in the reference)")
    ap.add_argument("--threads", type=int, default=8, help="host threads writing slices (not in the reference)")
    ap.add_argument("--verbose", action="store_true", help="print the end-to-end rate on stderr (not in the reference)")
    a = ap.parse_args(argv)

    _size(a.input_size, "input-size")           # parsed and otherwise unused, as in main.rs:164-174
    w, h = _size(a.output_size, "output-size")
    qp = 26 if a.qp is None else a.qp           # ctu.rs:382 default when --qp is absent
    if a.extra_params:
        for item in a.extra_params.split(","):
            if len(item.split("=")) != 2:
                _die("Invalid extra-params: %s" % a.extra_params)     # main.rs:205-215
    if w % 32 or h % 32:
        _die("output-size must be a multiple of the 32x32 CTU (picture.rs:178-181): %dx%d" % (w, h))
    if not 0 <= qp <= 63 or not 0 <= a.max_split_depth <= 3 or a.num_pictures < 0:
        _die("qp must be 0..63, max-split-depth 0..3")

    from . import bitstream, gpu
    try:
        fin = sys.stdin.buffer if a.input == "-" else open(a.input, "rb")
    except OSError as e:
        _die("failed to open input file: %s" % e)
    try:
        fout = sys.stdout.buffer if a.output == "-" else open(a.output, "wb")
    except OSError as e:
        _die("failed to open output file: %s" % e)
    frec = None
    if a.reconst:
        try:
            frec = open(a.reconst, "wb")
        except OSError as e:
            _die("failed to open reconst file: %s" % e)

    batch = max(1, min(a.batch, max(a.num_pictures, 1)))
    halves = 2 if a.num_pictures > batch else 1     # two sets of slots: one is searched while the other is read back
    try:
        enc = gpu.Encoder(w, h, qp=qp, max_split_depth=a.max_split_depth, device=a.device, n_slots=halves * batch,
                          extra_params=a.extra_params)
    except (gpu.WrencGpuError, ImportError, OSError) as e:
        _fatal(str(e))    # no device / a non-numeric extra-params value (parse().unwrap() panics in the reference)

    fout.write(bitstream.write_parameter_sets(w, h, qp))
    ysz, csz = w * h, (w // 2) * (h // 2)
    # page-locked staging: one input buffer and one record per slot (transfers at PCIe rate, truly asynchronous)
    keys_wanted = None if frec is not None else ("lev_y", "lev_cb", "lev_cr", "cu_log2_size", "luma_mode", "chroma_mode")
    try:
        stage_in = [enc.alloc_host(ysz + 2 * csz) for _ in range(halves * batch)]
        stage_out = [enc.alloc_picture_host(keys_wanted) for _ in range(halves * batch)]
    except gpu.WrencGpuError as e:
        _fatal(str(e))
    pool = ThreadPoolExecutor(max_workers=max(1, a.threads))
    t_start = time.perf_counter()
    stats = {"pictures": 0, "bytes": 0, "read_upload": 0.0, "download": 0.0, "write": 0.0}
    want = None if frec is not None else ("lev_y", "lev_cb", "lev_cr", "cu_log2_size", "luma_mode", "chroma_mode")

    def submit(first_poc, base):
        """Read and upload the next batch into slots base.. and start its search (asynchronous); returns its size."""
        n = 0
        t0 = time.perf_counter()
        for s in range(min(batch, a.num_pictures - first_poc)):
            buf = stage_in[base + s]
            if not _read_into(fin, buf):
                _die("input ended after %d of %d pictures" % (first_poc + n, a.num_pictures))
            enc.upload(base + s, buf[:ysz].reshape(h, w), buf[ysz:ysz + csz].reshape(h // 2, w // 2),
                       buf[ysz + csz:].reshape(h // 2, w // 2))
            n += 1
        if n:
            enc.encode(base, n)
        stats["read_upload"] += time.perf_counter() - t0
        return n

    def flush(batch_out):
        if batch_out is None:
            return
        futures, recs = batch_out
        t0 = time.perf_counter()
        for t, fut in enumerate(futures):
            nal = fut.result()
            fout.write(nal)
            stats["bytes"] += len(nal)
            if frec is not None:
                for k in ("rec_y", "rec_cb", "rec_cr"):     # main.rs:387-399
                    frec.write(recs[t][k].tobytes())
        stats["write"] += time.perf_counter() - t0
        stats["pictures"] += len(futures)

    pending = None
    try:
        poc, base = 0, 0
        n = submit(0, 0) if a.num_pictures > 0 else 0
        while n:
            first, done, dbase = poc, n, base
            poc += n
            base = (batch - base) if halves == 2 else 0
            # the next batch is queued behind the current one: the GPU searches it while this batch is read
            # back (the library's copy stream waits for this batch's search only) and entropy coded
            t0 = time.perf_counter()
            if halves == 1:
                enc.sync()
            n = submit(poc, base) if poc < a.num_pictures else 0
            t1 = time.perf_counter()
            recs = [enc.download(dbase + s, want, out=stage_out[dbase + s]) for s in range(done)]
            stats["download"] += time.perf_counter() - t1
            # pictures are independent: their slices are written in parallel (the C call drops the GIL), and
            # collected one batch later so that the writing overlaps the next read-back
            futures = [pool.submit(bitstream.write_picture, w, h, qp, first + t, recs[t]) for t in range(done)]
            flush(pending)
            pending = (futures, recs)
        flush(pending)
        fout.flush()
        stats["seconds"] = time.perf_counter() - t_start     # before the page-locked buffers are released
    except (gpu.WrencGpuError, bitstream.BitstreamError) as e:
        _fatal(str(e))
    finally:
        pool.shutdown()
        enc.close()
        fout.flush()
        if frec is not None:
            frec.close()
        if fout is not sys.stdout.buffer:
            fout.close()
    if a.verbose:
        dt = stats.get("seconds", time.perf_counter() - t_start)
        sys.stderr.write("%d pictures, %d bytes, %.2f s, %.1f pictures/s (file to stream, %d host threads)\n" % (
            stats["pictures"], stats["bytes"], dt, stats["pictures"] / max(dt, 1e-9), a.threads))
        sys.stderr.write("host time: read+upload %.2f s, waiting for the GPU + download %.2f s, waiting for slices %.2f s\n" % (
            stats["read_upload"], stats["download"], stats["write"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
