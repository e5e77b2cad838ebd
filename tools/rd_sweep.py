#!/usr/bin/env python3
"""RD-curve sweep (BASELINE.json configs[2]): encode the same pictures at several QPs
with the full CT-partition search, write the real bitstream, and report bytes, PSNR / SSIM of the
reconstruction and the rates of the two stages.  The result list follows the shape of the reference's
tools/evaluation/evaluate_mp.py:78-120 (title, qp, bytes, duration, metrics.{PSNR,SSIM}.{summary,per_frame} with the
attributes Avg / Y / U / V of metrics.json), without ffmpeg / VTM: PSNR and SSIM are ffmpeg's definitions restated in
wrenc_amd/metrics.py (psnr_avg from the plane-weighted MSE, SSIM over 8x8 windows at stride 4; pinned against the
reference's own summary.json by tests/test_metrics.py), computed from the encoder's reconstruction, which the stream
parser tests show is what a decoder rebuilds.

    python tools/rd_sweep.py [--width 3840 --height 2176 --frames 8 --depth 3 --qps 22,27,32,37] [--out file.json]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_sweep(width=3840, height=2176, frames=8, depth=3, qps=(22, 27, 32, 37), threads=8, extra_params=None,
              keep_streams=False, verbose=True):
    """The sweep as a function (tests/test_gpu_rd_sweep.py runs it too).  keep_streams: each result also carries
    "_stream" (parameter sets + pictures), "_recs" and the doc "_frames", for a decoder-side check by the caller."""
    from wrenc_amd import bitstream, gpu, metrics, synth
    w, h, n = width, height, frames
    pics = [synth.synth_textured_frame(w, h, f) for f in range(n)]
    results = []
    pool = ThreadPoolExecutor(max_workers=threads)
    for qp in [int(q) for q in qps]:
        enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=n, extra_params=extra_params)
        for s in range(n):
            enc.upload(s, *pics[s])
        enc.sync()
        t0 = time.perf_counter()
        enc.encode(0, n)
        enc.sync()
        t_search = time.perf_counter() - t0
        recs = [enc.download(s) for s in range(n)]
        mism = enc.final_pass_mismatches()
        enc.close()
        t0 = time.perf_counter()
        nals = list(pool.map(lambda t: bitstream.write_picture(w, h, qp, t[0], t[1]), enumerate(recs)))
        t_write = time.perf_counter() - t0
        head = bitstream.write_parameter_sets(w, h, qp)
        total = len(head) + sum(len(x) for x in nals)
        per_frame = []
        for f in range(n):
            m = metrics.frame_metrics(pics[f], (recs[f]["rec_y"], recs[f]["rec_cb"], recs[f]["rec_cr"]))
            m["n"] = f + 1
            m["bytes"] = len(nals[f])
            per_frame.append(m)
        summ = {k: metrics.summarise([p[k] for p in per_frame]) for k in ("PSNR", "SSIM")}
        results.append({
            "title": "synth_textured_%dx%d[wrenc_amd@max_split_depth=%d,qp=%d]" % (w, h, depth, qp), "qp": qp,
            "bytes": total, "duration": t_search + t_write, "frames": n,
            "bits_per_pixel": 8.0 * total / (n * w * h),
            "search_fps": n / t_search, "bitstream_fps_%d_threads" % threads: n / t_write,
            "final_pass_mismatches": mism,
            "metrics": {k: {"summary": summ[k], "per_frame": [dict(p[k], n=p["n"]) for p in per_frame]} for k in ("PSNR", "SSIM")},
            "frame_bytes": [p["bytes"] for p in per_frame]})
        if keep_streams:
            results[-1]["_stream"] = head + b"".join(nals)
            results[-1]["_recs"] = recs
        if verbose:
            print("qp %2d  %9d bytes  %.4f bpp  PSNR Avg %.2f Y %.2f U %.2f V %.2f dB  SSIM All %.4f Y %.4f  search %.1f fps  writer %.1f fps" % (
                qp, total, results[-1]["bits_per_pixel"], summ["PSNR"]["Avg"], summ["PSNR"]["Y"], summ["PSNR"]["U"], summ["PSNR"]["V"],
                summ["SSIM"]["Avg"], summ["SSIM"]["Y"], n / t_search, n / t_write), flush=True)
    pool.shutdown()
    doc = {"config": {"width": w, "height": h, "frames": n, "max_split_depth": depth, "content": "synth_textured_frame",
                      "extra_params": extra_params},
           "results": results}
    if keep_streams:
        doc["_frames"] = pics
    return doc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2176)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--qps", default="22,27,32,37")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--extra-params", help="the reference's RD-model knobs, K1=V1,K2=V2")
    ap.add_argument("--out")
    a = ap.parse_args()
    doc = run_sweep(a.width, a.height, a.frames, a.depth, [int(q) for q in a.qps.split(",")], a.threads, a.extra_params)
    if a.out:
        json.dump(doc, open(a.out, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
