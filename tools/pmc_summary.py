#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile_round.sh: per kernel (grouped by the name before the
template arguments) the HBM bytes per launch from the TCC counters, corrected as MI355X_MICROARCH.md prescribes for
gfx950 (FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B... here: KB = 1000 B as rocprofv3 reports them;
FETCH_SIZE counts 1/2 of the bytes of wide 16 B/lane streaming reads -> x2 for kernels whose reads are such), the
average duration from the kernel trace of the same pass, and the MFMA instruction count.

usage: pmc_summary.py <dir with the csv files> <tag>
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void\s+", "", name)
    return name.strip()


def counters(prefix):
    """{kernel: {counter: [per-dispatch values]}} from <prefix>*counter_collection.csv"""
    res = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for f in glob.glob(os.path.join(out_dir, "**", prefix + "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            res[k][row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
    return res


def durations(prefix):
    res = defaultdict(list)
    for f in glob.glob(os.path.join(out_dir, "**", prefix + "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            res[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    return res


fetch, write, sq = counters("tcc2"), counters("tcc3"), counters("sq1")
dur = durations("stats")
summary = {"tag": tag, "source": "rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0` (tools/profile_round.sh); "
           "durations from the --kernel-trace --stats pass of `bench.py --steps 4 --warmup 1`",
           "fetch_correction": "FETCH_SIZE x 2 (gfx950: wide streaming reads are reported at half their bytes, MI355X_MICROARCH.md)",
           "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not any(s in k for s in ("legendre", "ringfft", "draw", "clarray", "cl_", "chol", "factor", "jacobi", "zig_")):
        continue
    nf = max(1, len(fetch.get(k, {})))
    nw = max(1, len(write.get(k, {})))
    f_kb = sum(d.get("FETCH_SIZE", 0.0) for d in fetch.get(k, {}).values()) / nf
    w_kb = sum(d.get("WRITE_SIZE", 0.0) for d in write.get(k, {}).values()) / nw
    ms = sum(dur.get(k, [0.0])) / max(1, len(dur.get(k, [])))
    e = {"launches_in_pmc_pass": nf, "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
         "hbm_bytes_per_launch": 2.0 * f_kb * 1024.0 + w_kb * 1024.0, "avg_ms": ms}
    if ms > 0:
        e["hbm_GBs"] = e["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9
    if k in sq:
        n = len(sq[k])
        for c in ("SQ_INSTS_VALU_MFMA_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
            e[c + "_per_launch"] = sum(d.get(c, 0.0) for d in sq[k].values()) / n
    summary["kernels"][k] = e
# cfg-5 rank share: K4 from the SQ pass + the kernel trace of the cfg-5 runs
sq5, dur5 = counters("cfg5sq1"), durations("cfg5stats")
for k in sorted(sq5):
    if "legendre" not in k:
        continue
    n = len(sq5[k])
    ms = sum(dur5.get(k, [0.0])) / max(1, len(dur5.get(k, [])))
    e = {"avg_ms": ms, "launches_in_pmc_pass": n}
    for c in ("SQ_INSTS_VALU_MFMA_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
        e[c + "_per_launch"] = sum(d.get(c, 0.0) for d in sq5[k].values()) / n
    if ms > 0:
        e["executed_TFLOPs"] = 2048.0 * e["SQ_INSTS_VALU_MFMA_F64_per_launch"] / (ms * 1e-3) / 1e12
        e["frac_of_78.6_TF"] = e["executed_TFLOPs"] / 78.6
    summary.setdefault("cfg5_rank_share", {})[k] = e
print(json.dumps(summary, indent=1))
