"""Extracts the reference's own evaluation results for wrenc (tools/evaluation/summary.json of hjmkt/wrenc: stream bytes,
PSNR and SSIM per QP for bus / mobile CIF, 30 frames, commit 1d5b5ec) into tests/golden/ref_metrics.json.  Data only: the
per-frame {"Avg", "Y", "U", "V"} entries ffmpeg printed and the per-QP summaries evaluate_mp.py made of them.  They pin
the DEFINITIONS of the harness's metrics (wrenc_amd/metrics.py, tests/test_metrics.py).

    python tests/golden/make_ref_metrics.py [/root/reference]
"""
import json
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
doc = json.load(open(os.path.join(ref, "tools", "evaluation", "summary.json")))
out = {"source": "tools/evaluation/summary.json", "commit_id": doc["commit_id"], "results": []}


def walk(x):
    if isinstance(x, dict):
        if "title" in x and "metrics" in x:
            if "[wrenc#" in x["title"]:
                out["results"].append({"title": x["title"], "qp": x["qp"], "bytes": x["bytes"],
                                       "metrics": {m: {"summary": x["metrics"][m]["summary"], "per_frame": x["metrics"][m]["per_frame"]}
                                                   for m in ("PSNR", "SSIM")}})
            return
        for v in x.values():
            walk(v)
    elif isinstance(x, list):
        for v in x:
            walk(v)


walk(doc)
here = os.path.dirname(os.path.abspath(__file__))
json.dump(out, open(os.path.join(here, "ref_metrics.json"), "w"), separators=(",", ":"))
print(len(out["results"]), "results")
