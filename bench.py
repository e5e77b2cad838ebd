#!/usr/bin/env python3
"""bench.py -- all-intra encode throughput of the MI355X RD-search path.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  One "step" is one pass of the hot path (CTU search + final pass of
every CTU) over one batch of pictures that is already resident in HBM.

`value` (every N): the LARGEST single-GPU configuration of BASELINE.json -- configs[2] / [3]'s workload, 3840x2176
(2160 padded to a multiple of the 32-sample CTU, README.md:37 of the reference) synthetic YUV420, QP32,
max-split-depth 3 (the full CT-partition search), 240 pictures per step and GPU.  Pictures are independent I-slices, so
with N GPUs every rank runs its own 240 (weak scaling, no data-path collective).

Objects next to it on the same JSON line (rank 0 prints ONE line), each measured IN this run unless labelled:
  roofline      the dominant kernel (ctu_search_kernel) against HBM with the algorithmic 6 bytes per luma pixel
                (SURVEY.md 8d): per launch from the kernel's own HIP events (its average named separately from the
                team kernel's), and for the whole device; `traffic` / `issue_bound` come from the committed rocprofv3
                counter passes of the same workload and say which commit they were taken on
  cpu_baseline  the CPU oracle (a port of the reference algorithm: the Rust reference cannot be built here) on one
                host core over a bounded crop of one picture of the same workload; N = 1 only
  parity        the record of picture 0 of the timed run compared, bit for bit, with the record the cpu_baseline leg
                computed (CTU rows depend on nothing below them, so the crop's rows are the whole picture's)
  config1       BASELINE.json configs[1], 1920x1088 QP32 max-split-depth 2, 1024 pictures: the same objects again
                (value, roofline, cpu_baseline on one whole picture + all host cores, parity); N = 1
  config3       BASELINE.json configs[3] as written: 240 pictures of 3840x2176 IN TOTAL, picture p on rank p mod N
                (strong scaling); at N = 1 it is the headline itself
  textured      the headline workload on textured content; fill_curve: frames/s against pictures in flight;
                e2e / e2e_4k: file to stream with the native program (upload + search + read-back + host CABAC), 1080p
                depth 2 and the headline workload itself; N = 1
The process exits with status 1 (after printing the line) if any parity check or final-pass check fails.
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
# 4 encode lanes + a copy stream need more than the HIP runtime's default of 4 hardware queues per process; the variable
# must be in the environment before ANYTHING initialises the runtime (torch.cuda does), so it is set here and not
# only in wrenc_amd/gpu.py (with 4 queues two lanes share one and run one after the other: -30 % at 128 pictures)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HEADLINE = {"name": "3840x2176_qp32_d3_b240", "w": 3840, "h": 2176, "qp": 32, "depth": 3, "batch": 240, "cpu_rows": 576}
CONFIG1 = {"name": "1920x1088_qp32_d2_b1024", "w": 1920, "h": 1088, "qp": 32, "depth": 2, "batch": 1024, "cpu_rows": None}
ALGO_BYTES_PER_PIXEL = 6.0      # 1.5 B read + 1.5 B recon + 3.0 B levels
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")
REC_KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")


def committed_profile(name):
    """The committed rocprofv3 counter summary of this workload (profiles/traffic.json, written by
    tools/profile_kernel.sh + tools/profile_summary.py), or None.  NOT measured in this run: every user labels it."""
    try:
        return json.load(open(TRAFFIC_FILE)).get(name)
    except (OSError, ValueError):
        return None


def cpu_baseline(wl, frame=0):
    """Oracle on one host core over a bounded sample of the workload (the top `cpu_rows` rows of one picture, or the
    whole picture); also returns the record."""
    import numpy as np
    from oracle import pyoracle as po
    from wrenc_amd import synth
    w, h = wl["w"], wl["h"]
    y, cb, cr = synth.synth_frame(w, h, frame)
    rows = wl["cpu_rows"] or h
    y, cb, cr = y[:rows], cb[:rows // 2], cr[:rows // 2]
    t0 = time.perf_counter()
    rec = po.encode_picture(np.ascontiguousarray(y), np.ascontiguousarray(cb), np.ascontiguousarray(cr), wl["qp"], wl["depth"])
    dt = time.perf_counter() - t0
    frac = rows / float(h)
    return {"value": frac / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "the top %d of %d rows of one %dx%d picture, QP%d max-split-depth %d, %.1f s on 1 core"
                      % (rows, h, w, h, wl["qp"], wl["depth"], dt),
            "mpix_per_s": w * rows / dt / 1e6}, rec


def cpu_all_cores(wl, limit_s=120):
    """The same oracle on every host core this process may use, one whole picture per child process (pictures are
    independent, the reference itself is single-threaded).  Plain subprocesses with a time limit."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 32))
    code = ("import sys; sys.path.insert(0, %r); from oracle import pyoracle as po; from wrenc_amd import synth; "
            "y, cb, cr = synth.synth_frame(%d, %d, int(sys.argv[1])); po.encode_picture(y, cb, cr, %d, %d)"
            % (ROOT, wl["w"], wl["h"], wl["qp"], wl["depth"]))
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
            "sample": "%d pictures of %dx%d, one per process, %.1f s" % (ok, wl["w"], wl["h"], dt)}


def compare_records(got, ref, ctu_rows, width):
    """Bit-exact comparison of the top `ctu_rows` CTU rows of two records of the same picture."""
    import numpy as np
    planes = {}
    for k in REC_KEYS:
        if k == "ctu_cost":
            n = ctu_rows * (width // 32)
            planes[k] = bool(np.array_equal(got[k][:n], ref[k][:n]))
        else:
            rows = ctu_rows * 32 * got[k].shape[0] // got["rec_y"].shape[0]
            planes[k] = bool(np.array_equal(got[k][:rows], ref[k][:rows]))
    return {"bit_exact": all(planes.values()), "planes_equal": planes}


def run_resident(enc, grp, first, count, steps, warmup, timed_stats=False):
    """`steps` encode calls over resident slots [first, first + count), bracketed as the contract says; returns
    (seconds (max over ranks), per-kernel stats summed over the timed steps)."""
    import torch
    for _ in range(warmup):
        enc.encode(first, count)
        enc.sync()
    ks = {"wave": {"ms_sum": 0.0, "launches": 0, "ctu_pictures": 0}, "team": {"ms_sum": 0.0, "launches": 0, "ctu_pictures": 0}}
    grp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if count:
            enc.encode(first, count)
            enc.sync()
            if timed_stats:
                st = enc.last_encode_kernel_stats()
                for k in ks:
                    for f in ks[k]:
                        ks[k][f] += st[k][f]
    torch.cuda.synchronize()
    grp.barrier()
    dt = grp.max(time.perf_counter() - t0)
    return dt, ks


def kernel_roofline(ks, dt, total_ctu_pictures, name, lanes=4):
    """The roofline object of one workload from the per-launch HIP events of the timed steps.  A CTU-picture is 1024
    luma pixels = 6144 algorithmic bytes."""
    bytes_per_ctu = ALGO_BYTES_PER_PIXEL * 1024.0

    def one(k):
        s = ks[k]
        if not s["launches"]:
            return None
        avg_s = s["ms_sum"] / 1e3 / s["launches"]
        per_launch = bytes_per_ctu * s["ctu_pictures"] / s["launches"]
        return {"launches": s["launches"], "avg_launch_ms": avg_s * 1e3, "ctu_pictures_per_launch": s["ctu_pictures"] / s["launches"],
                "algorithmic_bytes_per_launch": per_launch, "achieved": per_launch / avg_s / 1e9}

    wave, team = one("wave"), one("team")
    dom = wave or team
    device_gbs = bytes_per_ctu * total_ctu_pictures / dt / 1e9
    prof = committed_profile(name)
    out = {"bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["achieved"] / HBM_PEAK_GBS,
           "kernel": "ctu_search_kernel" if wave else "ctu_search_team_kernel",
           "avg_launch_ms": dom["avg_launch_ms"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
           "ctu_search_kernel": wave, "ctu_search_team_kernel": team,
           # an encode call keeps 4 HIP streams of launches co-resident: the GPU as a whole moves the step's algorithmic
           # bytes over its wall time
           "achieved_device": device_gbs, "frac_device": device_gbs / HBM_PEAK_GBS, "concurrent_streams": lanes,
           "traffic": None, "limiter": "per-wave latency at 5 waves per SIMD, both issue pipes ~60 % busy (not hbm): DESIGN.md section 4, profiles/r04_issue_model.md"}
    if prof and "traffic_bytes_per_launch" in prof:
        out["traffic"] = prof["traffic_bytes_per_launch"]
        out["traffic_source"] = "committed profile of %s (profiles/traffic.json; FETCH_SIZE x 2 + WRITE_SIZE per launch), NOT measured in this run" % prof.get("commit")
        out["traffic_over_algorithmic"] = prof.get("traffic_over_algorithmic")
    if prof:
        out["issue_bound"] = {k: prof[k] for k in prof if k.startswith(("valu_", "salu_", "branch_", "lds_", "sq_", "issue_"))}
        out["issue_bound"]["source"] = "committed profile of %s, NOT measured in this run" % prof.get("commit")
    return out


def measure(grp, rank, world, local_rank, wl, steps, warmup, with_cpu):
    """One workload on this rank's GPU: `batch` resident pictures, `steps` timed encode calls; on rank 0 the result
    objects (value, roofline, host_bitstream; with_cpu: cpu_baseline + parity)."""
    from wrenc_amd import gpu, sharding, synth
    w, h, B = wl["w"], wl["h"], wl["batch"]
    enc = gpu.Encoder(w, h, qp=wl["qp"], max_split_depth=wl["depth"], device=local_rank, n_slots=B)
    enc.stats_enable(True)      # per-launch HIP events: measurement mode (off in the product path)
    # synthetic pictures, resident in HBM before the timed region; each rank owns the POCs p with p mod world == rank
    distinct = min(B, 8)
    frames = {}
    for s, poc in enumerate(sharding.picture_shard(world * B, rank, world)):
        f = poc % (distinct * world)
        if f not in frames:
            frames[f] = synth.synth_frame(w, h, f)
        enc.upload(s, *frames[f])
    enc.sync()
    frames = None
    dt, ks = run_resident(enc, grp, 0, B, steps, warmup, timed_stats=True)
    mism = enc.final_pass_mismatches()
    out = None
    if rank == 0:
        from wrenc_amd import bitstream
        rec0 = enc.download(0)
        pool0, pics0 = enc.download_tokens(0, 1)     # the same picture as residual tokens made on the device (dev_bins.h)
        best = best_tok = None
        for _ in range(2):      # what follows the hot path on the host, one core; not part of `value`
            t1 = time.perf_counter()
            nal = bitstream.write_picture(w, h, wl["qp"], 0, rec0)
            t2 = time.perf_counter()
            nal_tok = bitstream.write_picture_tokens(w, h, wl["qp"], 0, pool0, pics0[0])
            t3 = time.perf_counter()
            best = t2 - t1 if best is None else min(best, t2 - t1)
            best_tok = t3 - t2 if best_tok is None else min(best_tok, t3 - t2)
        fps = world * B * steps / dt
        ctus = B * steps * (w // 32) * (h // 32)
        out = {"value": fps, "unit": "frames/s", "mpix_per_s": fps * w * h / 1e6, "ms_per_step": dt * 1e3 / steps,
               "workload": "%dx%d synthetic YUV420 QP%d max-split-depth %d" % (w, h, wl["qp"], wl["depth"]),
               "pictures_per_step_per_gpu": B, "final_pass_mismatches": mism,
               "roofline": kernel_roofline(ks, dt, ctus, wl["name"], lanes=enc.device_info()[1]),
               "host_bitstream": {"ms_per_picture_one_core": best * 1e3, "bytes_per_picture": len(nal),
                                  "ms_per_picture_one_core_from_device_tokens": best_tok * 1e3, "token_bytes_per_picture": enc.last_token_words * 4,
                                  "same_bytes_from_tokens": nal == nal_tok,
                                  "note": "host CABAC + syntax of one searched picture from its level planes, and from the residual "
                                          "tokens the device makes of it (the host then runs the CU-level syntax and the arithmetic "
                                          "coder only); pictures are independent, one host thread each"}}
        if nal != nal_tok:
            out["host_bitstream"]["error"] = "the token path wrote other bytes"
        if with_cpu:
            base, ref0 = cpu_baseline(wl)
            out["cpu_baseline"] = base
            rows = wl["cpu_rows"] or h
            ctu_rows = rows // 32      # a CTU row depends on nothing below it: the crop's rows are the picture's
            par = compare_records(rec0, ref0, ctu_rows, w)
            par["what"] = ("record of picture 0 of the timed run (all planes, CTU costs), CTU rows 0..%d, == the CPU oracle's "
                           "record of the same input" % (ctu_rows - 1))
            par["oracle"] = "oracle/wrenc_oracle.cpp (PARITY UNPINNED against the Rust reference, see DESIGN.md)"
            out["parity"] = par
    enc.close()
    return out


def config3(grp, rank, world, local_rank, total=240, steps=1, warmup=1):
    """BASELINE.json configs[3]: 3840x2176, `total` pictures in all, QP32, max-split-depth 3; picture p is encoded by
    rank p mod world (wrenc_amd/sharding.py), no data-path collective: strong scaling."""
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
    dt, _ = run_resident(enc, grp, 0, len(mine), steps, warmup)
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
    for name, w, h, qp, depth, points in (("3840x2176_d3", 3840, 2176, 32, 3, (8, 30, 128)),
                                          ("1920x1088_d2", 1920, 1088, 32, 2, (8, 32, 128, 512))):
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
                dt, _ = run_resident(enc, grp, 0, b, 1, 0)
                best = dt if best is None else min(best, dt)
            curve[str(b)] = b / best
            sched[str(b)] = {0: "team on thin diagonals, wave on wide ones", 1: "wave", 2: "team"}.get(enc.last_schedule(), "?")
        enc.close()
        out[name] = curve
        out[name + "_schedule"] = sched
    return out


def textured_rate(grp, local_rank, wl):
    """The headline workload again on synth_textured_frame content (several times the stream bytes per picture: few
    all-zero transform blocks, which the quantiser's zero-block exits favour): one warm-up call, one timed call."""
    from wrenc_amd import gpu, synth
    w, h, batch = wl["w"], wl["h"], wl["batch"]
    enc = gpu.Encoder(w, h, qp=wl["qp"], max_split_depth=wl["depth"], device=local_rank, n_slots=batch)
    frames = [synth.synth_textured_frame(w, h, f) for f in range(4)]
    for s in range(batch):
        enc.upload(s, *frames[s % 4])
    enc.sync()
    dt, _ = run_resident(enc, grp, 0, batch, 1, 1)
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
                            "--batch", str(batch), "--threads", str(threads), "--verbose"]
                           + (["--no-tokens"] if os.environ.get("WRENC_E2E_NO_TOKENS") else []),   # (A/B: residual syntax on the host)
                           capture_output=True, timeout=900)
        m = re.search(rb"(\d+) pictures, (\d+) bytes, ([\d.]+) s, ([\d.]+) pictures/s", r.stderr)
        if r.returncode != 0 or not m:
            return {"error": "status %d: %s" % (r.returncode, r.stderr[-300:].decode(errors="replace"))}
        out = {"value": float(m.group(4)), "unit": "frames/s", "pictures": int(m.group(1)), "stream_bytes": int(m.group(2)),
               "seconds": float(m.group(3)), "batch": batch, "host_threads": threads,
               "content": "synth_textured_frame" if textured else "synth_frame",
               "what": "raw YUV file -> .vvc file: read, upload, search + final pass, read-back, host CABAC, write"}
        # the program's timeline (--verbose): when each batch came back from the device.  Between the first and the last
        # read-back the pipeline is full: pictures that came back in that interval / its length = the rate a long run has
        tl = [(float(t), int(p)) for t, p in re.findall(rb"([\d.]+) s: batch at picture (\d+) read back", r.stderr)]
        if len(tl) >= 3 and tl[-1][0] > tl[0][0]:
            last_count = int(m.group(1)) - tl[-1][1]
            out["batches"] = len(tl)
            out["steady_state"] = {"value": round((tl[-1][1] + last_count - tl[1][1]) / (tl[-1][0] - tl[0][0]), 1), "unit": "frames/s",
                                   "what": "pictures read back after the first batch / time from the first to the last read-back"}
        return out
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
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("WRENC_BENCH_BATCH", str(HEADLINE["batch"]))),
                    help="pictures resident per GPU and encoded per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the contract's line: no config1 / config3 / fill_curve / e2e")
    ap.add_argument("--config3-pictures", type=int, default=240)
    args = ap.parse_args()

    import torch
    from wrenc_amd import gpu, sharding  # noqa: F401  (gpu: see GPU_MAX_HW_QUEUES above)

    rank, local_rank, world = sharding.world_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with python -m torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    grp = sharding.Group(backend="nccl", device="cuda:%d" % local_rank)

    wl = dict(HEADLINE, batch=args.batch)
    if args.batch != HEADLINE["batch"]:
        wl["name"] = "3840x2176_qp32_d3_b%d" % args.batch
    with_cpu = world == 1 and not args.no_cpu_baseline
    head = measure(grp, rank, world, local_rank, wl, args.steps, args.warmup, with_cpu)
    result, failed = None, []
    if rank == 0:
        result = {
            "metric": "all-intra encode fps at fixed QP (CTU RD search + final pass); bit-exactness vs the CPU oracle is checked in `parity`",
            "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i16/i32 (+i64 trellis costs, f32 RD cost)", "data": "synthetic",
            "mpix_per_s": head["mpix_per_s"],
            "config": {"workload": head["workload"] + " (BASELINE.json configs[2]/[3]: the largest single-GPU configuration)",
                       "pictures_per_step_per_gpu": wl["batch"], "parallelism": "picture-sharded x%d, no collective" % world,
                       "final_pass_mismatches": head["final_pass_mismatches"]},
            "roofline": head["roofline"], "host_bitstream": head["host_bitstream"]}
        if head["final_pass_mismatches"]:
            failed.append("final pass mismatches (headline)")
        if not head["host_bitstream"]["same_bytes_from_tokens"]:
            failed.append("token path bytes (headline)")
        if "cpu_baseline" in head:
            result["cpu_baseline"] = head["cpu_baseline"]
            result["parity"] = head["parity"]
            result["parity_checked"] = head["parity"]["bit_exact"]
            if not head["parity"]["bit_exact"]:
                failed.append("parity (headline)")
    if not args.no_extras:
        if world == 1:
            c1 = measure(grp, rank, world, local_rank, CONFIG1, 2, 1, with_cpu)
            if with_cpu:
                try:
                    c1["cpu_baseline"]["all_cores"] = cpu_all_cores(CONFIG1)   # ~10 s
                except Exception as e:  # the 1-core figure is the contract; this one is extra
                    c1["cpu_baseline"]["all_cores"] = {"error": repr(e)}
                if not c1["parity"]["bit_exact"]:
                    failed.append("parity (config1)")
            if c1["final_pass_mismatches"]:
                failed.append("final pass mismatches (config1)")
            c1["steps"], c1["warmup"] = 2, 1
            result["config1"] = c1
            result["config3"] = {"note": "at one GPU BASELINE.json configs[3] (240 pictures of 3840x2176 in total) IS the headline: see `value`",
                                 "value": head["value"], "unit": "frames/s", "scaling": "strong", "n_gpus": 1, "pictures_per_gpu": wl["batch"]}
            result["textured"] = textured_rate(grp, local_rank, wl)
            result["fill_curve"] = fill_curve(grp, local_rank)
            threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8))
            # 4 batches of 512: the first batch's search and the last batch's entropy coding have nothing to overlap with
            result["e2e"] = {"smooth": e2e_native(1920, 1088, 32, 2, 2048, 512, threads, False),
                             "textured": e2e_native(1920, 1088, 32, 2, 2048, 512, threads, True),
                             "workload": "1920x1088 QP32 max-split-depth 2 (configs[1]), 2048 pictures in 4 batches of 512"}
            # ... and the headline workload itself, file to stream (missing 1): BASELINE's metric is encoded
            # frames/s; `value` is the search alone.  480 pictures in 2 batches of 240
            result["e2e_4k"] = {"smooth": e2e_native(3840, 2176, 32, 3, 960, 240, threads, False),
                                "textured": e2e_native(3840, 2176, 32, 3, 960, 240, threads, True),
                                "workload": "3840x2176 QP32 max-split-depth 3 (configs[2]/[3]), 960 pictures in 4 batches of 240",
                                "search_rate_smooth": head["value"], "search_rate_textured": result["textured"]["value"]}
            for k in ("textured",):
                if result[k].get("final_pass_mismatches"):
                    failed.append("final pass mismatches (%s)" % k)
        else:
            c3 = config3(grp, rank, world, local_rank, total=args.config3_pictures)
            if rank == 0:
                result["config3"] = c3
                if c3["final_pass_mismatches"]:
                    failed.append("final pass mismatches (config3)")
    grp.close()
    if rank == 0:
        if failed:
            result["error"] = "FAILED: " + ", ".join(failed)
        print(json.dumps(result))
        if failed:      # a throughput figure of a wrong encoder must not pass for a result
            sys.exit(1)


if __name__ == "__main__":
    main()
