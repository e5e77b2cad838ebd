"""Sums the counters of the search kernels over the dispatches in rocprofv3's *counter_collection.csv files of
tools/profile_kernel.sh's passes, per kernel and per CTU-picture, and writes summary.json (what profiles/traffic.json
holds per workload): python tools/profile_summary.py WORKDIR WORKLOAD COMMIT UNITS "ARGS"."""
import collections
import csv
import glob
import json
import os
import sys

work, wl, commit, units, args = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
tot = collections.defaultdict(collections.Counter)
nd = collections.defaultdict(collections.Counter)
for f in glob.glob(work + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "ctu_search" not in name:
            continue
        k = "team" if "team" in name else "wave"
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        nd[k][row["Counter_Name"]] += 1
both = collections.Counter()
for k in tot:
    both.update(tot[k])
print("command: python3 tools/encode_workload.py %s   (commit %s; per unit = per CTU-picture, %d of them)" % (args, commit, units))
for c in sorted(both):
    print("%-28s %14.4e  per CTU %12.1f  (wave kernel %.4e in %d dispatches, team kernel %.4e in %d)" % (
        c, both[c], both[c] / units, tot["wave"][c], nd["wave"][c], tot["team"][c], nd["team"][c]))
stats = {}
for f in glob.glob(work + "/stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ctu_search" in row["Name"]:
            stats["team" if "team" in row["Name"] else "wave"] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                                                                  "total_ms": float(row["TotalDurationNs"]) / 1e6}
launches = sum(v["calls"] for v in stats.values()) or 1
wc = both.get("SQ_WAVE_CYCLES", 0.0)
summ = {"workload": wl, "command": "tools/encode_workload.py " + args, "commit": commit, "ctu_pictures": units,
        "kernel_stats": stats, "launches": launches}
if "FETCH_SIZE" in both and "WRITE_SIZE" in both:
    # both counters are in KiB; FETCH_SIZE counts half of the bytes read on gfx950 (MI355X_MICROARCH.md): doubled
    summ["fetch_size_kb_per_launch"] = both["FETCH_SIZE"] / launches
    summ["write_size_kb_per_launch"] = both["WRITE_SIZE"] / launches
    summ["fetch_kib_per_ctu_corrected"] = 2.0 * both["FETCH_SIZE"] / units
    summ["write_kib_per_ctu"] = both["WRITE_SIZE"] / units
    summ["traffic_bytes_per_launch"] = (2.0 * both["FETCH_SIZE"] + both["WRITE_SIZE"]) * 1024.0 / launches
    summ["traffic_over_algorithmic"] = (2.0 * both["FETCH_SIZE"] + both["WRITE_SIZE"]) * 1024.0 / (units * 1024 * 6.0)
for name, key in (("valu_insts_per_ctu", "SQ_INSTS_VALU"), ("salu_insts_per_ctu", "SQ_INSTS_SALU"), ("branch_insts_per_ctu", "SQ_INSTS_BRANCH"),
                  ("lds_insts_per_ctu", "SQ_INSTS_LDS"), ("sq_wave_quad_cycles_per_ctu", "SQ_WAVE_CYCLES")):
    if key in both:
        summ[name] = both[key] / units
if wc:
    summ["issue_wave_time_shares"] = {
        "issuing_instructions": both.get("SQ_ACTIVE_INST_ANY", 0) / wc, "valu": both.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        "scalar": both.get("SQ_ACTIVE_INST_SCA", 0) / wc, "waiting_s_waitcnt_or_barrier": both.get("SQ_WAIT_ANY", 0) / wc,
        "waiting_for_an_issue_slot": both.get("SQ_WAIT_INST_ANY", 0) / wc}
if "SQ_THREAD_CYCLES_VALU" in both and both.get("SQ_ACTIVE_INST_VALU"):
    summ["valu_lane_utilisation"] = both["SQ_THREAD_CYCLES_VALU"] / (64.0 * both["SQ_ACTIVE_INST_VALU"])
json.dump(summ, open(os.path.join(work, "summary.json"), "w"), indent=1)
