#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>_lds*/ (profiles/tools/lds_counters.sh) -> gpurun_out/<tag>_pmc_lds.json: per kernel symbol and launch the
LDS-side and wait-side SQ counters, per transformed row where the kernel has rows, stamped with the kernel-source hash."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
from pinocchio_amd import _lib  # noqa: E402
from summarise import short  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
GO = os.path.join(ROOT, "gpurun_out")
N = int(os.environ.get("PF_SUMMARY_N", "1024"))
FB = int(os.environ.get("PF_SUMMARY_FB", "8"))
out = {"_method": "rocprofv3 --pmc <group> --kernel-trace, one group per run, one bench step (12 radii + 3LPT); counters summed over the dispatches of a "
                  "kernel symbol and divided by their number.  SQ_* cycle counters are in quad-cycles summed over waves (MI355X_MICROARCH.md); "
                  "per_row = per launch / (n^2 rows of six components) for the invariant z-pass",
       "kernel_source_sha": _lib.source_sha(), "config": {"grid": N, "field_bytes": FB}, "kernels": collections.defaultdict(dict)}
for d in sorted(glob.glob(os.path.join(GO, f"pmc_{tag}_lds*"))):
    if not os.path.isdir(d):
        continue
    paths = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not paths:
        continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for row in csv.DictReader(open(paths[-1])):
        k = short(row["Kernel_Name"])
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
    dur = collections.defaultdict(float)
    tp = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if tp:
        for row in csv.DictReader(open(tp[-1])):
            dur[short(row["Kernel_Name"])] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    for k, v in tot.items():
        nd = len(disp[k])
        if dur.get(k, 0) / max(nd, 1) < 0.5e6:      # kernels under half a millisecond per launch are not listed
            continue
        ent = out["kernels"][k]
        ent["dispatches"] = nd
        ent.setdefault("ms_per_launch_in_counter_runs", {})[os.path.basename(d)] = dur[k] / nd / 1e6
        for c, x in v.items():
            ent[c] = x / nd
for k, ent in out["kernels"].items():
    if "k_c2r_invariants" in k:
        rows = float(N) * N
        ent["per_row"] = {c: x / rows for c, x in ent.items() if c.startswith("SQ_")}
    if ent.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in ent:
                ent[c + "_over_WAVE_CYCLES"] = ent[c] / ent["SQ_WAVE_CYCLES"]
    if ent.get("SQ_LDS_IDX_ACTIVE"):
        ent["bank_conflict_share_of_lds_cycles"] = ent.get("SQ_LDS_BANK_CONFLICT", 0.0) / ent["SQ_LDS_IDX_ACTIVE"]
out["kernels"] = dict(out["kernels"])
json.dump(out, open(os.path.join(GO, f"{tag}_pmc_lds.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:5000])
