#!/bin/bash
# Memory-path stall counters of one bench step (run from the repo root ON THE GPU BOX):  profiles/tools/stall_counters.sh r04
# Each group is its own rocprofv3 run with --kernel-trace only; only counters the installed rocprofv3 lists are asked for.
# Output: gpurun_out/pmc_<tag>_stall*/ and gpurun_out/<tag>_pmc_stall.json (copy into profiles/ to commit).
tag=${1:-r04}
R=$PWD
export TMPDIR=/tmp
export PF_SOLVE_BESIDE_Z=0
mkdir -p gpurun_out
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "Counter_Name *:[[:space:]]*[A-Za-z0-9_]*" | awk '{print $NF}' | sort -u > $R/gpurun_out/${tag}_counter_names.txt
have() { out=""; for c in "$@"; do grep -qx "$c" $R/gpurun_out/${tag}_counter_names.txt && out="$out $c"; done; echo $out; }
pmc() {  # pmc <name> <counters...>
  name=$1; shift
  [ $# -gt 0 ] || return
  rm -rf $R/gpurun_out/pmc_${tag}_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-n 0 --exact-steps 0 --table-steps 0 --check 0 $BENCH_ARGS \
    > $R/gpurun_out/pmc_${tag}_$name.out 2> $R/gpurun_out/pmc_${tag}_$name.err || tail -3 $R/gpurun_out/pmc_${tag}_$name.err
}
pmc stall1 $(have TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum)
pmc stall2 $(have TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum)
# (the TA_* group -- TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum -- hung the profiled run on this pool in round 4: left out)
pmc stall3 $(have TCC_BUSY_sum)
pmc stall4 $(have TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum)
pmc stall5 $(have GRBM_GUI_ACTIVE TCC_CYCLE_sum TCC_REQ_sum TCC_TAG_STALL_sum)
cd $R
python3 - "$tag" <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
out = {}
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_stall*")):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name") or row.get("kernel_name") or ""
            c = row.get("Counter_Name") or row.get("counter_name")
            v = float(row.get("Counter_Value") or row.get("counter_value") or 0.0)
            k = k.split("(")[0].strip()
            e = out.setdefault(k, {}).setdefault(c, [0.0, 0])
            e[0] += v; e[1] += 1
res = {"note": "per kernel symbol: average per launch of each counter over one bench step (every kernel in line); separate rocprofv3 passes per group",
       "kernels": {k: {c: s / n for c, (s, n) in sorted(cs.items())} | {"launches": max(n for _, n in cs.values())} for k, cs in sorted(out.items())}}
json.dump(res, open(f"gpurun_out/{tag}_pmc_stall.json", "w"), indent=1)
for k, cs in res["kernels"].items():
    if any(t in k for t in ("k_strided<double, 1024, 8, 1", "k_c2r_invariants", "k_collapse_inv")):
        print(k[:60], {c: (round(v) if v > 100 else v) for c, v in cs.items()})
PY
