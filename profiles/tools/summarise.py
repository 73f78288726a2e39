#!/usr/bin/env python3
"""Turns the rocprofv3 output of profiles/tools/collect.sh (gpurun_out/prof_<tag>, gpurun_out/pmc_<tag>_*) into the small
files kept under profiles/:
  <tag>_kernel_stats.csv    kernel statistics of the profiled bench command (name, calls, total / average ns, share)
  <tag>_pmc_traffic.json    HBM bytes per launch and kernel symbol: FETCH_SIZE (doubled, the gfx950 correction of
                            MI355X_MICROARCH.md for 16-byte-per-lane streaming reads) and WRITE_SIZE, separate passes
  <tag>_pmc_valu.json       vector instructions per cell, VALU utilisation, engine clock under the kernel
All are stamped with the hash of the kernel sources (pinocchio_amd/_lib.source_sha); bench.py ignores them when it runs other sources.
Written to gpurun_out/ (merged back by gpurun); copy into profiles/ to commit."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pinocchio_amd import _lib  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
GO = os.path.join(ROOT, "gpurun_out")
N, FB = int(os.environ.get("PF_SUMMARY_N", "1024")), int(os.environ.get("PF_SUMMARY_FB", "8"))
SLAB_OF = int(os.environ.get("PF_SUMMARY_SLAB_OF", "1"))   # bench.py --slab-of P: one rank's slab of the n^3 box
CELLS = float(N) ** 3 / SLAB_OF


def short(name):
    """'void k_strided<double, 1024, 4, 1>(PfStridedParams, long long, int)' -> 'k_strided<double, 1024, 4, 1>'"""
    name = re.sub(r"^void\s+", "", name.strip().strip('"'))
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name.replace(" [clone .kd]", "").replace(".kd", "")


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(GO, pattern), recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def counters(name, wanted):
    """per kernel symbol: {counter: sum over dispatches}, number of dispatches; plus per-dispatch durations if traced"""
    path = newest(f"pmc_{tag}_{name}/**/*counter_collection.csv")
    if not path:
        return {}, {}
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] not in wanted:
            continue
        k = short(row["Kernel_Name"])
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
    dur = collections.defaultdict(float)
    tpath = newest(f"pmc_{tag}_{name}/**/*kernel_trace.csv")
    if tpath:
        for row in csv.DictReader(open(tpath)):
            dur[short(row["Kernel_Name"])] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return {k: (dict(v), len(disp[k])) for k, v in tot.items()}, dur


def main():
    sha = _lib.source_sha()
    # ---- kernel statistics of the profiled bench command
    for sub, suffix, how in (("", "", ""), ("_inline", "_inline", "PF_SOLVE_BESIDE_Z=0 (every kernel in line: no two durations overlap) ")):
      spath = newest(f"prof_{tag}{sub}/**/*kernel_stats.csv")
      if spath:
        rows = list(csv.DictReader(open(spath)))
        with open(os.path.join(GO, f"{tag}_kernel_stats{suffix}.csv"), "w") as out:
            out.write(f"# {how}rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 {os.environ.get('BENCH_ARGS', '')}; kernel sources {sha}\n")
            if not suffix:
                out.write("# (the durations of k_c2r_invariants<..., 0> and k_collapse_inv overlap here: the solve of sweep radius i runs beside the z-pass of radius i + 1; their own times are in the _inline file)\n")
            out.write("kernel,calls,total_ns,average_ns,percent\n")
            for r in rows:
                if float(r["Percentage"]) < 0.05:
                    continue
                out.write('"%s",%s,%s,%.1f,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
        print("kernel stats:", spath)
    # ---- HBM traffic
    fe, _ = counters("fetch", ("FETCH_SIZE",))
    wr, _ = counters("write", ("WRITE_SIZE",))
    traffic = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, each with --kernel-trace only, over ONE step of the "
                          "bench command (12 radii + 3LPT, pruning on: the same launch mix as the timed region); counter unit KB -> bytes; FETCH_SIZE "
                          "doubled (MI355X_MICROARCH.md: on gfx950 it reports half the bytes of 16-byte-per-lane streaming reads); summed over the "
                          "dispatches of a kernel symbol and divided by their number",
               "kernel_source_sha": sha, "config": dict({"grid": N, "field_bytes": FB}, **({"slab_of": SLAB_OF} if SLAB_OF > 1 else {})), "kernels": {}}
    for k in sorted(set(fe) & set(wr)):
        (f, nf), (w, nw) = fe[k], wr[k]
        if nf != nw or f.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0) < 1e6:
            continue
        traffic["kernels"][k] = {"fetch_bytes_per_launch": 2.0 * 1024.0 * f["FETCH_SIZE"] / nf, "write_bytes_per_launch": 1024.0 * w["WRITE_SIZE"] / nw,
                                 "dispatches": nf}
    json.dump(traffic, open(os.path.join(GO, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    # ---- issue side of the per-cell kernels
    va, vdur = counters("valu", ("SQ_INSTS_VALU", "SQ_WAVES"))
    bu, bdur = counters("busy", ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"))
    ck, cdur = counters("clk", ("GRBM_GUI_ACTIVE",))
    valu = {"_method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES | SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE, separate passes with --kernel-trace only, "
                       "one bench step; per launch.  insts_per_cell = SQ_INSTS_VALU * 64 lanes / cells; issue_ms = wave-instructions * 4 cycles / (1024 SIMDs * clock); "
                       "engine clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same run",
            "kernel_source_sha": sha, "config": dict({"grid": N, "field_bytes": FB}, **({"slab_of": SLAB_OF} if SLAB_OF > 1 else {})), "kernels": {}}
    for k, (v, nv) in va.items():
        if not any(w in k for w in ("collapse", "invariants", "strided", "mixed", "c2r", "r2c")):
            continue
        insts = v.get("SQ_INSTS_VALU", 0.0) / nv
        if insts < 1e6:
            continue
        ent = {"dispatches": nv, "valu_wave_insts_per_launch": insts, "valu_insts_per_cell": insts * 64.0 / CELLS, "ms_per_launch_in_counter_run": vdur.get(k, 0.0) / nv / 1e6}
        if k in ck and cdur.get(k):
            ghz = ck[k][0]["GRBM_GUI_ACTIVE"] / 8.0 / cdur[k]
            ent["engine_clock_GHz_measured"] = ghz
            ent["issue_ms_at_measured_clock"] = insts * 4.0 / (1024.0 * ghz * 1e9) * 1e3
            if ent["ms_per_launch_in_counter_run"]:
                ent["valu_utilisation_at_measured_clock"] = ent["issue_ms_at_measured_clock"] / ent["ms_per_launch_in_counter_run"]
        ent["issue_ms_at_2.4GHz"] = insts * 4.0 / (1024.0 * 2.4e9) * 1e3
        if k in bu and bu[k][0].get("SQ_BUSY_CYCLES"):
            ent["active_valu_over_busy_cycles"] = bu[k][0]["SQ_ACTIVE_INST_VALU"] / bu[k][0]["SQ_BUSY_CYCLES"]
        valu["kernels"][k] = ent
    # (optional pass, PF_COLLECT_LDS=1 in collect.sh: LDS instruction cycles and bank-conflict cycles against the busy cycles of the same run)
    ld, _ = counters("lds", ("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"))
    for k, (v, nv) in ld.items():
        if k in valu["kernels"] and v.get("SQ_BUSY_CYCLES"):
            valu["kernels"][k]["active_lds_over_busy_cycles"] = v.get("SQ_ACTIVE_INST_LDS", 0.0) / v["SQ_BUSY_CYCLES"]
            valu["kernels"][k]["lds_bank_conflict_over_busy_cycles"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_BUSY_CYCLES"]
    json.dump(valu, open(os.path.join(GO, f"{tag}_pmc_valu.json"), "w"), indent=1)
    print(json.dumps({"traffic": traffic["kernels"], "valu": valu["kernels"]}, indent=1)[:6000])


if __name__ == "__main__":
    main()
