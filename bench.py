#!/usr/bin/env python3
"""bench.py -- all-intra encode throughput of the MI355X RD-search path.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 it is
launched by torch.distributed.run with one rank per GPU.  One "step" is one pass of
the hot path (CTU search + final pass of every CTU) over one batch of pictures that
is already resident in HBM.

`value` (every N): BASELINE.json configs[1] -- 1920x1088 (1080p padded to a multiple
of 32, README.md:37 of the reference) synthetic YUV420, QP32, --max-split-depth 2;
pictures are independent I-slices, so with N GPUs every rank runs its own batch
(weak scaling, no data-path collective).

Objects next to it on the same JSON line (rank 0 prints ONE line):
  roofline      the search kernel against HBM with the algorithmic 6 bytes per luma pixel
                (SURVEY.md 8d), from the kernel's own HIP-event durations
  cpu_baseline  the CPU oracle (a port of the reference algorithm: the Rust reference cannot be
                built here) on one host core, bounded sample; N = 1 only
  parity        the record of picture 0 of the timed run compared with the record the
                cpu_baseline leg computed for the same input (bit-exact or not, per plane)
  config3       BASELINE.json configs[3]: 3840x2176, 240 pictures in total, QP32,
                max-split-depth 3, picture p on rank p mod N (strong scaling); frames/s of the
                whole job at this N
  textured      the same workload on textured content (`value` is on smooth content: N = 1)
  fill_curve    frames/s against pictures in flight (N = 1)
  e2e           file to stream with the native program: upload + search + read-back + host CABAC
                (N = 1)
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT, QP, DEPTH = 1920, 1088, 32, 2
ALGO_BYTES_PER_PIXEL = 6.0      # 1.5 B read + 1.5 B recon + 3.0 B levels
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")
REC_KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")


def measured_traffic(batch, width, height, qp, depth):
    """HBM bytes per launch of the search kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs of this same command, see profiles/).  Per the
    guide FETCH_SIZE counts half of the bytes read on gfx950, so it is doubled; both are in KiB.
    None when the passes were made for another workload."""
    try:
        t = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None
    if [t.get("batch"), t.get("width"), t.get("height"), t.get("qp"), t.get("depth")] != [batch, width, height, qp, depth]:
        return None
    return (2.0 * t["fetch_size_kb_per_launch"] + t["write_size_kb_per_launch"]) * 1024.0


def measured_issue_bound():
    """What actually bounds the kernel (it is nowhere near HBM): instruction counts per CTU and the issue
    model derived from them (profiles/), reported next to the roofline."""
    try:
        t = json.load(open(TRAFFIC_FILE))
        return {k: t[k] for k in t if k.startswith(("valu_", "salu_", "sq_", "issue_"))}
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(width, height, qp, depth, rows=None):
    """Oracle on one host core over a bounded sample of the same workload; also returns the record."""
    import numpy as np
    from oracle import pyoracle as po
    from wrenc_amd import synth
    y, cb, cr = synth.synth_frame(width, height, 0)
    if rows is not None:
        y, cb, cr = y[:rows], cb[:rows // 2], cr[:rows // 2]
    t0 = time.perf_counter()
    rec = po.encode_picture(np.ascontiguousarray(y), np.ascontiguousarray(cb), np.ascontiguousarray(cr), qp, depth)
    dt = time.perf_counter() - t0
    frac = y.shape[0] / float(height)
    return {"value": frac / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%dx%d rows of one %dx%d frame, QP%d depth %d, %.1f s on 1 core"
                      % (width, y.shape[0], width, height, qp, depth, dt),
            "mpix_per_s": width * y.shape[0] / dt / 1e6}, rec


def cpu_all_cores(width, height, qp, depth, limit_s=120):
    """The same oracle on every host core this process may use, one whole frame per child process
    (pictures are independent, the reference itself is single-threaded): reported next to the 1-core
    figure.  Plain subprocesses with a time limit: nothing here may hang the bench."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 32))
    code = ("import sys; sys.path.insert(0, %r); from oracle import pyoracle as po; from wrenc_amd import synth; "
            "y, cb, cr = synth.synth_frame(%d, %d, int(sys.argv[1])); po.encode_picture(y, cb, cr, %d, %d)"
            % (ROOT, width, height, qp, depth))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(f)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
             for f in range(cores)]
    ok = 0
    for p in procs:
        try:
            ok += p.wait(timeout=max(1.0, limit_s - (time.perf_counter() - t0))) == 0
        except subprocess.TimeoutExpired:
            p.kill()
    dt = time.perf_counter() - t0
    return {"value": ok / dt, "unit": "frames/s", "cores": cores,
            "sample": "%d frames of %dx%d, one per process, %.1f s" % (ok, width, height, dt)}


def compare_records(got, ref):
    """Bit-exact comparison of two records of the same picture; {"bit_exact": bool, "planes": {...}}."""
    import numpy as np
    planes = {k: bool(np.array_equal(got[k], ref[k])) for k in REC_KEYS}
    return {"bit_exact": all(planes.values()), "planes_equal": planes}


def run_resident(enc, grp, first, count, steps, warmup, timed_stats=False):
    """`steps` encode calls over resident slots [first, first + count), bracketed as the contract says;
    returns (seconds (max over ranks), kernel ms sum, launches)."""
    import torch
    for _ in range(warmup):
        enc.encode(first, count)
        enc.sync()
    kernel_ms, launches = 0.0, 0
    grp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if count:
            enc.encode(first, count)
            enc.sync()
            if timed_stats:
                st = enc.last_encode_stats()
                kernel_ms += st["kernel_ms_sum"]
                launches += st["n_launches"]
    torch.cuda.synchronize()
    grp.barrier()
    dt = grp.max(time.perf_counter() - t0)
    return dt, kernel_ms, launches


def config3(grp, rank, world, local_rank, total=240, steps=1, warmup=1):
    """BASELINE.json configs[3]: 3840x2176, `total` pictures in all, QP32, max-split-depth 3; picture p is
    encoded by rank p mod world (wrenc_amd/sharding.py), no data-path collective: strong scaling."""
    from wrenc_amd import gpu, sharding, synth
    w, h, qp, depth = 3840, 2176, 32, 3
    mine = list(sharding.picture_shard(total, rank, world))
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, device=local_rank, n_slots=max(len(mine), 1))
    frames = {}
    for s, poc in enumerate(mine):
        f = poc % 8
        if f not in frames:
            frames[f] = synth.synth_frame(w, h, f)
        enc.upload(s, *frames[f])
    enc.sync()
    dt, _, _ = run_resident(enc, grp, 0, len(mine), steps, warmup)
    mism = enc.final_pass_mismatches()
    enc.close()
    fps = total * steps / dt
    return {"workload": "3840x2176 synthetic YUV420 QP32 max-split-depth 3, %d pictures in total, picture p on rank p mod %d"
                        % (total, world),
            "value": fps, "unit": "frames/s", "mpix_per_s": fps * w * h / 1e6, "scaling": "strong", "n_gpus": world,
            "pictures_per_gpu": len(mine), "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3 / steps,
            "final_pass_mismatches": mism}


def fill_curve(grp, local_rank, quick=False):
    """frames/s against pictures in flight (one encode call of B resident pictures, best of 2), with the schedule the
    library picked for that call (include/wrenc_gpu.h: wave = one wavefront per CTU, team = four per CTU)."""
    from wrenc_amd import gpu, synth
    out = {}
    for name, w, h, qp, depth, points in (("1920x1088_d2", 1920, 1088, 32, 2, (8, 32, 128, 512, 1024)),
                                          ("3840x2176_d3", 3840, 2176, 32, 3, (8, 30, 128))):
        if quick:
            points = points[:2]
        enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, device=local_rank, n_slots=max(points))
        frames = [synth.synth_frame(w, h, f) for f in range(4)]
        for s in range(max(points)):
            enc.upload(s, *frames[s % 4])
        enc.sync()
        curve, sched = {}, {}
        for b in points:
            best = None
            for _ in range(2):
                dt, _, _ = run_resident(enc, grp, 0, b, 1, 0)
                best = dt if best is None else min(best, dt)
            curve[str(b)] = b / best
            sched[str(b)] = {0: "team on thin diagonals, wave on wide ones", 1: "wave", 2: "team"}.get(enc.last_schedule(), "?")
        enc.close()
        out[name] = curve
        out[name + "_schedule"] = sched
    return out


def textured_rate(grp, local_rank, w, h, qp, depth, batch):
    """The timed workload again on synth_textured_frame content (206 KB instead of 31 KB of stream per picture: few
    all-zero transform blocks, which the quantiser's zero-block exits favour): one warm-up call, one timed call."""
    from wrenc_amd import gpu, synth
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, device=local_rank, n_slots=batch)
    frames = [synth.synth_textured_frame(w, h, f) for f in range(8)]
    for s in range(batch):
        enc.upload(s, *frames[s % 8])
    enc.sync()
    dt, _, _ = run_resident(enc, grp, 0, batch, 1, 1)
    mism = enc.final_pass_mismatches()
    enc.close()
    return {"value": batch / dt, "unit": "frames/s", "content": "synth_textured_frame", "pictures": batch,
            "final_pass_mismatches": mism}


def e2e_native(w, h, qp, depth, n_pictures, batch, threads, textured):
    """File to stream with the native program (wrenc_amd/csrc/host/wrenc): raw YUV file in (tmpfs), .vvc out;
    read + upload + search + read-back + host CABAC on `threads` threads.  The program's own clock, which starts
    after the device context and the page-locked buffers exist."""
    from wrenc_amd import synth
    exe = os.path.join(ROOT, "wrenc_amd", "csrc", "host", "wrenc")
    if not os.path.exists(exe):
        return {"error": "native program not built"}
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    src = os.path.join(tmp, "wrenc_bench_in_%d.yuv" % os.getpid())
    dst = os.path.join(tmp, "wrenc_bench_out_%d.vvc" % os.getpid())
    make = synth.synth_textured_frame if textured else synth.synth_frame
    try:
        frames = [b"".join(p.tobytes() for p in make(w, h, f)) for f in range(8)]
        with open(src, "wb") as f:
            for i in range(n_pictures):
                f.write(frames[i % 8])
        r = subprocess.run([exe, "-i", src, "-o", dst, "--input-size", "%dx%d" % (w, h), "--output-size", "%dx%d" % (w, h),
                            "--num-pictures", str(n_pictures), "--qp", str(qp), "--max-split-depth", str(depth),
                            "--batch", str(batch), "--threads", str(threads), "--verbose"],
                           capture_output=True, timeout=900)
        m = re.search(rb"(\d+) pictures, (\d+) bytes, ([\d.]+) s, ([\d.]+) pictures/s", r.stderr)
        if r.returncode != 0 or not m:
            return {"error": "status %d: %s" % (r.returncode, r.stderr[-300:].decode(errors="replace"))}
        return {"value": float(m.group(4)), "unit": "frames/s", "pictures": int(m.group(1)), "stream_bytes": int(m.group(2)),
                "seconds": float(m.group(3)), "batch": batch, "host_threads": threads,
                "content": "synth_textured_frame" if textured else "synth_frame",
                "what": "raw YUV file -> .vvc file: read, upload, search + final pass, read-back, host CABAC, write"}
    except (OSError, subprocess.TimeoutExpired) as e:
        return {"error": repr(e)}
    finally:
        for p in (src, dst):
            try:
                os.remove(p)
            except OSError:
                pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("WRENC_BENCH_BATCH", "1024")),
                    help="pictures resident per GPU and encoded per step")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--qp", type=int, default=QP)
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the contract's line: no config3 / fill_curve / e2e")
    ap.add_argument("--config3-pictures", type=int, default=240)
    args = ap.parse_args()

    import torch
    from wrenc_amd import gpu, sharding, synth

    rank, local_rank, world = sharding.world_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with python -m torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    grp = sharding.Group(backend="nccl", device="cuda:%d" % local_rank)

    w, h, B = args.width, args.height, args.batch
    enc = gpu.Encoder(w, h, qp=args.qp, max_split_depth=args.depth, device=local_rank, n_slots=B)
    enc.stats_enable(True)      # per-launch HIP events: measurement mode (off in the product path)
    # synthetic pictures, resident in HBM before the timed region; each rank owns the POCs
    # p with p mod world == rank of a (world * B)-picture sequence
    distinct = min(B, 8)
    frames = {}
    for s, poc in enumerate(sharding.picture_shard(world * B, rank, world)):
        f = poc % (distinct * world)
        if f not in frames:
            frames[f] = synth.synth_frame(w, h, f)
        enc.upload(s, *frames[f])
    enc.sync()

    dt, kernel_ms, launches = run_resident(enc, grp, 0, B, args.steps, args.warmup, timed_stats=True)
    mism = enc.final_pass_mismatches()
    host_bs, rec0 = None, None
    if rank == 0:
        # what follows the hot path on the host: CABAC of one of the pictures just
        # searched, on one core; outside the timed region and not part of `value`
        from wrenc_amd import bitstream
        rec0 = enc.download(0)
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            nal = bitstream.write_picture(w, h, args.qp, 0, rec0)
            t2 = time.perf_counter()
            best = t2 - t1 if best is None else min(best, t2 - t1)
        host_bs = {"ms_per_picture_one_core": best * 1e3, "bytes_per_picture": len(nal),
                   "slice_data_bits": bitstream.last_slice_data_bits(),
                   "note": "host CABAC + syntax of one searched picture; pictures are independent, one host thread each"}
    enc.close()

    total_frames = world * B * args.steps
    fps = total_frames / dt
    result = None
    if rank == 0:
        pix = float(w) * h
        per_launch_bytes = ALGO_BYTES_PER_PIXEL * pix * B * args.steps / max(launches, 1)
        avg_launch_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = per_launch_bytes / avg_launch_s / 1e9
        device_gbs = ALGO_BYTES_PER_PIXEL * pix * B * args.steps / dt / 1e9     # this GPU: all lanes together
        result = {
            "metric": "all-intra encode fps at fixed QP (CTU RD search + final pass); bit-exactness vs the CPU oracle is checked in `parity`",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i16/i32 (+i64 trellis costs, f32 RD cost)", "data": "synthetic",
            "mpix_per_s": fps * pix / 1e6,
            "config": {"workload": "%dx%d synthetic YUV420 QP%d max-split-depth %d" % (w, h, args.qp, args.depth),
                       "pictures_per_step_per_gpu": B, "parallelism": "picture-sharded x%d, no collective" % world,
                       "final_pass_mismatches": mism},
            # `achieved` is per launch as the contract defines it (algorithmic bytes of a launch over its own HIP-event
            # duration); an encode call keeps 4 HIP streams of launches co-resident, so the GPU as a whole moves
            # `achieved_device` = the step's algorithmic bytes over its wall time.  The kernel is nowhere near HBM:
            # what bounds it is instruction issue (`issue_bound`, from the committed SQ counter passes).
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "achieved_device": device_gbs, "frac_device": device_gbs / HBM_PEAK_GBS,
                         "traffic": measured_traffic(B, w, h, args.qp, args.depth),
                         "kernel": "ctu_search_kernel", "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": per_launch_bytes,
                         "concurrent_streams": 4,
                         "limiter": "valu/issue (not hbm)", "issue_bound": measured_issue_bound()},
        }
        result["host_bitstream"] = host_bs
        if world == 1 and not args.no_cpu_baseline:
            full = (w, h, args.qp, args.depth) == (WIDTH, HEIGHT, QP, DEPTH) or w * h <= WIDTH * HEIGHT
            base, ref0 = cpu_baseline(w, h, args.qp, args.depth, rows=None if full else 128)   # one full frame, ~7 s
            result["cpu_baseline"] = base
            if full:    # slot 0 of the timed run holds synth_frame(0): the very picture the CPU leg just encoded
                par = compare_records(rec0, ref0)
                par["what"] = "record of picture 0 of the timed run (all planes, CTU costs) == the CPU oracle's record of the same input"
                par["oracle"] = "oracle/wrenc_oracle.cpp (PARITY UNPINNED against the Rust reference, see DESIGN.md)"
                result["parity"] = par
                result["parity_checked"] = par["bit_exact"]
            try:
                result["cpu_baseline"]["all_cores"] = cpu_all_cores(w, h, args.qp, args.depth)   # ~10 s
            except Exception as e:  # the 1-core figure is the contract; this one is extra
                result["cpu_baseline"]["all_cores"] = {"error": repr(e)}
    if not args.no_extras:
        c3 = config3(grp, rank, world, local_rank, total=args.config3_pictures)
        if rank == 0:
            result["config3"] = c3
        if world == 1:
            result["textured"] = textured_rate(grp, local_rank, w, h, args.qp, args.depth, B)
            result["fill_curve"] = fill_curve(grp, local_rank)
            threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8))
            # 4 batches of 512: the first batch's search and the last batch's entropy coding have nothing to overlap with
            result["e2e"] = {"smooth": e2e_native(w, h, args.qp, args.depth, 2048, 512, threads, False),
                             "textured": e2e_native(w, h, args.qp, args.depth, 2048, 512, threads, True)}
    grp.close()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
