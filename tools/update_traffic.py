"""Merges tools/profile_kernel.sh summaries into profiles/traffic.json (one entry per workload; what bench.py's
`roofline.traffic` / `issue_bound` quote, labelled with the commit they were taken on) and copies the per-workload
files into profiles/:  python tools/update_traffic.py gpurun_out/prof/r04_WORKLOAD_COMMIT_summary.json ..."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"4kd3": "3840x2176_qp32_d3_b240", "1080d2": "1920x1088_qp32_d2_b1024", "8kd3": "7680x4320_qp32_d3_b32",
        "4kd3b30": "3840x2176_qp32_d3_b30"}
path = os.path.join(ROOT, "profiles", "traffic.json")
try:
    doc = json.load(open(path))
    if "kernel" in doc:         # the round-2 single-workload layout
        doc = {}
except (OSError, ValueError):
    doc = {}
for f in sys.argv[1:]:
    s = json.load(open(f))
    base = f[:-len("_summary.json")]
    s["files"] = []
    for suffix in ("_summary.json", "_pmc.txt", "_kernel_stats.csv"):
        if os.path.exists(base + suffix):
            shutil.copy(base + suffix, os.path.join(ROOT, "profiles", os.path.basename(base) + suffix))
            s["files"].append("profiles/" + os.path.basename(base) + suffix)
    doc[KEYS[s["workload"]]] = s
json.dump(doc, open(path, "w"), indent=1)
print("traffic.json:", ", ".join("%s @ %s" % (k, v["commit"]) for k, v in doc.items()))
