#!/bin/bash
# registers, LDS, scratch and occupancy of every kernel of one source file of the library (the compiler's own report):
#   profiles/tools/resources.sh pf_mixed_kernels [extra flags]  ->  one line per kernel
f=$1; shift
cd "$(dirname "$0")/../../pinocchio_amd/csrc" || exit 1
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DPF_FP_CONTRACT_ON"
[ "$f" = pf_cell_kernels ] && flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
/opt/rocm/bin/hipcc $flags "$@" -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/res_$$.o 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: .*?: (.*)$", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    else:
        k, _, v = t.partition(":")
        if k.strip() in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs"): cur[k.strip().split(" ")[0]] = v.strip()
if cur: print(cur)
' | while read -r l; do echo "$l"; done
rm -f /tmp/res_$$.o
