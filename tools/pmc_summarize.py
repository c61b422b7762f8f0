#!/usr/bin/env python3
"""tools/pmc_summarize.py <dir> — fold the rocprofv3 CSVs written by tools/profile_round.sh into one JSON: per vsyn_* kernel the
mean duration (kernel trace) and the mean of every counter per launch (counters are summed over the XCD/SE instances rocprofv3
reports per dispatch)."""
import collections
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
root = sys.argv[1]


def kname(raw):
    """'void vsyn_residue_vq_kernel<true>(...)' -> 'vsyn_residue_vq_kernel<true>'; plain kernels unchanged"""
    n = raw.split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n


out = {"kernels": {}}
for path in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if kname(row["Name"]).startswith("vsyn_"):
            k = out["kernels"].setdefault(kname(row["Name"]), {})
            k["calls"] = int(row["Calls"])
            k["avg_ns"] = float(row["AverageNs"])
            k["min_ns"] = float(row["MinNs"])
            k["max_ns"] = float(row["MaxNs"])
for path in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))  # (kernel, counter) -> dispatch -> value
    for row in csv.DictReader(open(path)):
        name = kname(row["Kernel_Name"])
        if name.startswith("vsyn_"):
            per[(name, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (name, ctr), d in per.items():
        vals = sorted(d.values())
        out["kernels"].setdefault(name, {}).setdefault("pmc_mean_per_launch", {})[ctr] = sum(vals) / len(vals)
lk = out["kernels"].get("vsyn_fused_kernel", {}).get("pmc_mean_per_launch", {})
if "FETCH_SIZE" in lk and "WRITE_SIZE" in lk:
    # KiB units; FETCH_SIZE counts half of the streamed bytes on gfx950 (MI355X_MICROARCH.md HBM section; tools/fetch_calib.hip)
    out["long_kernel_hbm_bytes_per_launch"] = {"read_corrected": lk["FETCH_SIZE"] * 1024 * 2, "write": lk["WRITE_SIZE"] * 1024}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:  # the sources these counters were collected on: bench.py quotes the summary only while they are unchanged
    from bench import csrc_fingerprint
    out["csrc_fingerprint"] = csrc_fingerprint()
except Exception as exc:
    out["csrc_fingerprint_error"] = str(exc)[:200]
print(json.dumps(out, indent=1, sort_keys=True))
