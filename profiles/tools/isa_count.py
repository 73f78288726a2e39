"""Static instruction mix of the kernels of one source file of the library whose demangled name contains a pattern:
    python3 profiles/tools/isa_count.py pf_fft_kernels 'k_c2r_invariants_spec<float' [extra hipcc flags]
compiles the device side to assembly (hipcc --cuda-device-only -S) and counts, per kernel, vector / packed-fp32 / scalar-fp32 / fp64 /
v_mov / LDS / global instructions -- the static companion of the SQ_INSTS_VALU counters (round 6: the fp32 z-passes in packed algebra)."""
import os
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pinocchio_amd", "csrc")
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-DPF_FP_CONTRACT_ON"]
if src == "pf_cell_kernels":
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off"]
out = f"/tmp/isa_{os.getpid()}.s"
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + extra + ["--cuda-device-only", "-S", os.path.join(csrc, src + ".hip"), "-o", out])
text = open(out).read()
os.remove(out)
names = re.findall(r"^(_Z\w+):\s*; @", text, flags=re.M)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for mangled, d in zip(names, dem):
    if pat not in d:
        continue
    body = text[text.index(mangled + ":"):]
    body = body[:body.index("s_endpgm")]
    ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = lambda f: sum(1 for i in ins if f(i))   # noqa: E731
    row = {"valu": c(lambda i: i.startswith("v_")), "pk_f32": c(lambda i: i.startswith("v_pk_") and i.endswith("f32")),
           "scalar_f32": c(lambda i: i.startswith("v_") and not i.startswith("v_pk") and "f32" in i and "cvt" not in i),
           "f64": c(lambda i: i.startswith("v_") and "f64" in i and "cvt" not in i), "cvt": c(lambda i: "cvt" in i),
           "v_mov": c(lambda i: i.startswith("v_mov") or i.startswith("v_accvgpr")), "int_valu": c(lambda i: re.match(r"v_(add|sub|mul|mad|lshl|lshr|and|or|xor|bfe|cndmask|cmp|ashr|add3|lshl_add|mad_u)\w*(_u32|_i32|_b32|_u64|_i64|_co_u32|_u16|_u24|_i24)", i) is not None),
           "ds": c(lambda i: i.startswith("ds_")), "global": c(lambda i: i.startswith(("global_", "buffer_", "flat_", "scratch_"))),
           "salu": c(lambda i: i.startswith("s_") and not i.startswith(("s_waitcnt", "s_nop", "s_barrier"))), "waitcnt": c(lambda i: i.startswith("s_waitcnt")),
           "barrier": c(lambda i: i.startswith("s_barrier")), "scratch": c(lambda i: i.startswith("scratch_"))}
    print(d[:120])
    print("   ", row)
