#!/usr/bin/env python3
"""bench.py -- grid-cells/s through the full multi-radius Fmax sweep + 3LPT
displacement build (BASELINE.json metric), on N MI355X of one node.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 3 --warmup 1        (starts its own eight ranks: launch_ranks below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1

One step = one pass of the hot path over one synthetic density field already
resident in HBM: pf_sweep (Ns = 12 radii: six second derivatives + collapse
times each) then pf_displacements (2LPT/3LPT sources + 12 displacement fields).
Workload at every N: the 1024^3 fp64 box the metric is quoted on (fits one
GPU: ~226 GB), sharded in x-slabs over the ranks -> strong scaling.  Rank 0
prints ONE JSON line.

What the line carries beyond the contract fields:
  roofline           the dominant kernel BY SYMBOL (what `rocprofv3 --stats` lists first): all launch classes that run the
                     same kernel function are summed; achieved = algorithmic bytes / HIP-event time of those launches
  roofline_by_class  the same for the single most expensive launch class (the fp64-VALU-bound collapse solve, whose byte
                     rate is a consequence, with its issue-side counters)
  hbm_streaming      GB/s of kernels that only read / write / copy one field, measured in this process after the timed region;
                     roofline.streaming_ceiling prices the dominant kernel's counted reads and writes at those rates
  path_roofline      whole step: the bytes this design really moves (the survey's contract figure is quoted beside it, no fraction)
  kernels            per launch class: launches, ms per step, algorithmic GB/s, symbol.  In the timed configuration the collapse
                     solve of sweep radius i runs on its own stream beside the z-pass of radius i + 1 and the
                     HIP-event spans of those two classes overlap; the table -- and the time of a class that overlaps -- therefore
                     comes from a second, short pass of the same step with every kernel in line (kernel_table says so and gives that
                     pass's step time); the dominant kernel's roofline is still formed from the events of the timed region itself
  exchange           (N > 1) kind negotiated, bytes per step per rank, time on the communication stream
  exact_libm         (N = 1) step time with PF_EXACT_LIBM=1 (the reference's own libm calls in the solve), informational
  cpu_baseline       (N = 1) the CPU oracle on all host threads, on a bounded sample of the workload
Counter-derived fields (`traffic`, `valu`) come from committed rocprofv3 passes (profiles/<round>_pmc_*.json) and are emitted only
when those passes were made on the very kernel sources this run loads (hash of pinocchio_amd/csrc); otherwise null.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# pinocchio_amd is imported inside the functions that run on a rank: the parent of a self-launched multi-GPU run
# (launch_ranks) must neither load the HIP library nor touch the GPU.

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_ROUND = os.environ.get("PROFILE_ROUND", "r06")  # which profiles/<round>_pmc_*.json the counters come from


def alg_bytes_per_cell(ns: int, w: int, lpt: bool) -> float:
    """SURVEY.md section 8d contract figure: Ns*(49W+16) + 207W + 48"""
    return ns * (49 * w + 16) + ((207 * w + 48) if lpt else 0)


# grid sizes whose stage plans csrc/pf_mixed_kernels.hip compiles in (PF_MIXED_CT_SIZES there; tests/test_bench_report.py holds the two
# lists against each other), and the rule that makes a plan (pf_radices): first radix 8 (or 4 for the half-length lines of the z-pass
# where 8 does not divide), then 8s, a 4 or a 2, 5s and 3s
MIXED_CT_SIZES = (200, 384, 400, 640, 768, 800, 1000, 1280, 1536, 1600, 2000,
                  96, 120, 144, 160, 192, 216, 240, 288, 320, 360, 432, 480, 576,
                  600, 648, 720, 864, 960, 1080, 1152, 1200, 1296, 1440, 1728, 1800, 1920, 1944)   # (the three lists of the three translation units)


def mixed_radices(n: int, allow4: bool):
    r0 = 8 if n % 8 == 0 else (4 if allow4 and n % 4 == 0 else 0)
    if not r0:
        return None
    rest, r = n // r0, [r0]
    while rest % 8 == 0:
        r.append(8); rest //= 8
    if rest % 4 == 0:
        r.append(4); rest //= 4
    if rest % 2 == 0:
        r.append(2); rest //= 2
    while rest % 5 == 0:
        r.append(5); rest //= 5
    while rest % 3 == 0:
        r.append(3); rest //= 3
    return r if rest == 1 else None


def mixed_plan_names(n: int):
    """(strided plan, z-pass plan) as the template argument rocprofv3 prints"""
    if n not in MIXED_CT_SIZES:
        return "PfPlanRT", "PfPlanRT"
    return tuple("PfPlanCT<%s>" % ", ".join(str(x) for x in mixed_radices(m, a4)) for m, a4 in ((n, False), (n // 2, True)))


GENERAL_PATH = False   # set by run_config when the context says pf_transform_path() == 2: chirp-z transforms, one 3-D transform per component


def symbol_of(cls: str, n: int, fb: int, fast: bool = True) -> str:
    """kernel function behind a launch class, spelled as rocprofv3 prints it (csrc/pf_fft_kernels.hip dispatch tables)"""
    F = "double" if fb == 8 else "float"
    if GENERAL_PATH and cls in ("zpass_c2r_plain", "zpass_r2c", "zpass_c2r_hess_6", "zpass_c2r_disp_3"):
        # csrc/pf_gfft.hip: a 3-D transform is three passes of k_blue<M, MODE> (M = 2^p >= 2 n - 1; MODE 0 complex lines, 1 / 2 the
        # Hermitian <-> real lines along z), timed as one launch of this class: the family is named
        m = 1 << max(4, (2 * n - 2).bit_length())
        return f"k_blue<{m}, mode> x 3"
    t = max(1, min(128 // (2 * fb), (128 * 1024) // (n * 2 * fb), 8192 // n))     # PfTileCols (128 KB of LDS per tile)
    FS = F
    if fb == 4 and n >= 1024:      # fp32 lines of 1024 points and more: two columns per thread, 16-byte elements (launch_strided_f32)
        FS = "float __vector(2)"
        t = max(1, min(8, (128 * 1024) // (n * 16), 8192 // n))
    nt = n // 16
    tl = 1 if nt >= 256 else 256 // nt
    b = "true" if fast else "false"
    if n & (n - 1):   # not a power of two: the stage plans of csrc/pf_mixed_kernels.hip (compile-time for the sizes it names, else run-time)
        r0 = 8 if (n // 2) % 8 == 0 else 4
        ps, pz = mixed_plan_names(n)
        e = " " if n in MIXED_CT_SIZES else ""   # (a nested template argument list closes with a blank in rocprofv3's spelling)
        if cls in ("xpass_hess_1to3", "ypass_hess_3to6", "xpass_disp_1to2", "ypass_disp_2to3", "xpass_plain", "ypass_plain"):
            return f"k_mixed_strided<{F}, 1, {ps}{e}>"
        if cls in ("xpass_fwd", "ypass_fwd"):
            return f"k_mixed_strided<{F}, -1, {ps}{e}>"
        if cls in ("zpass_c2r_hess_6", "zpass_c2r_disp_3", "zpass_c2r_plain"):
            return f"k_mixed_c2r<{F}, {r0}, {pz}{e}>"
        if cls == "zpass_r2c":
            return f"k_mixed_r2c<{F}, {r0}, {pz}{e}>"
        if cls in ("zpass_c2r_hess_6to3inv", "zpass_c2r_hess_6_lpt3b"):
            return f"k_mixed_c2r_invariants<{F}, {r0}, {pz}, {0 if cls.endswith('inv') else 1}>"
    # fp32 lines of 1024 / 2048 points: the kernels in packed (re, im) arithmetic (csrc/pf_fft16_kernels.hip), one instantiation per
    # (direction, first-pass filter, band limit) -- a launch class runs several of them (pruned and unpruned radii): the family is named
    if fb == 4 and n in (1024, 2048) and cls in ("xpass_hess_1to3", "ypass_hess_3to6", "xpass_disp_1to2", "ypass_disp_2to3", "xpass_plain", "ypass_plain", "xpass_fwd", "ypass_fwd"):
        return f"{'k_strided16' if n == 2048 else 'k_strided_pk8'}<{-1 if cls.endswith('_fwd') else 1}, pre, band>"
    # last template argument of k_strided: addresses split into a scalar and a 32-bit lane part (one rank and up to eight: true)
    if cls in ("xpass_hess_1to3", "ypass_hess_3to6", "xpass_disp_1to2", "ypass_disp_2to3", "xpass_plain", "ypass_plain"):
        return f"k_strided<{FS}, {n}, {t}, 1, true>"
    if cls in ("xpass_fwd", "ypass_fwd"):
        return f"k_strided<{FS}, {n}, {t}, -1, true>"
    if cls in ("zpass_c2r_hess_6", "zpass_c2r_disp_3", "zpass_c2r_plain"):
        return f"k_c2r_persistent<{F}, {n}, {tl}>"
    # the invariant z-pass of rows of 512 and 1024 points (and of 2048 with fp32 fields) runs with two (four) more waves that only reduce: PfZiPlan::spec
    zinv = "k_c2r_invariants_spec" if (n in (512, 1024) or (n == 2048 and fb == 4)) else "k_c2r_invariants"
    zinv_sym = f"{zinv}<{F}, {n}, 0>"
    if fb == 4 and n in (512, 1024):   # round 6: fp32 rows of 512 and 1024 points, two rows per thread (second argument: the waves that only reduce)
        zinv_sym = f"k_c2r_invariants_pk2<{n}, {2 if n == 1024 else 1}>"
    return {"zpass_c2r_hess_6to3inv": zinv_sym, "zpass_c2r_hess_6_lpt3b": f"k_c2r_invariants<{F}, {n}, 1>",
            "collapse": f"k_collapse<{F}, {b}, float>", "collapse_inv": f"k_collapse_inv<{b}, float>", "lpt_sources": f"k_lpt_sources<{F}>",
            "collapse_lpt_sources": f"k_collapse_src<{F}, {b}, float>",      # last argument: PRODFLOAT of the build
            "lpt_accum": f"k_lpt_accum<{F}>", "zpass_r2c": f"k_r2c<{F}, {n}, {tl}>"}.get(cls, cls)


def counter_symbol(kernels: dict, symbol: str):
    """the key of `kernels` (a counter summary of profiles/) that holds `symbol`: itself, or -- k_strided only -- the instantiation with
    whole 64-bit addresses per lane (last template argument false), which the launcher takes where a lane's byte offset does not fit 32
    bits (slabs thinner than n / 8, the replicated spectrum of a 2048 box with fp32 fields: launch_strided_n)"""
    if symbol in kernels:
        return symbol
    if symbol.startswith("k_strided<") and symbol.endswith(", true>"):
        alt = symbol[:-len("true>")] + "false>"
        if alt in kernels:
            return alt
    return None


def committed_counters(kind: str, n: int, fb: int):
    """profiles/<round>[_fp32]_pmc_<kind>.json if it was measured on the kernel sources loaded now, else None"""
    from pinocchio_amd import _lib
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}{'_fp32' if fb == 4 else ''}_pmc_{kind}.json")) as fh:
            d = json.load(fh)
        if d.get("kernel_source_sha") == _lib.source_sha() and d.get("config") == {"grid": n, "field_bytes": fb}:
            return d
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(n: int, ns: int, lpt: bool, budget_s: float = 60.0) -> dict:
    """The CPU oracle (a port of the reference's structure: one k-loop + one c2r per derivative, per-cell ell_classic, AoS fp32
    products; OpenMP over x-planes) timed on this box's host cores.  The host is shared (several GPU slots per node) and the
    oracle's strided transforms stop scaling long before 256 threads, so the thread count is CALIBRATED first: a 256^3 box,
    two radii + the 3LPT part, on 16 / 32 / 64 / 128 threads; the fastest count then runs the n^3 box with ALL `ns` radii and
    the displacement part if the calibration predicts that fits `budget_s`, otherwise as many mid-ladder radii as fit
    (the oracle does the same work for every radius) with the sweep part scaled -- the sample string says which."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    # one thread per core, spread over both sockets (the first touch in orc_create places the pages accordingly).  The wait
    # policy stays the default: the oracle has many short parallel loops, and passive waiting costs 7x at 64^3 (measured)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    import oracle_lib
    from pinocchio_amd import synth
    cores = os.cpu_count() or 1
    # what the job may actually use: the CPU-time quota of its control group (cpu.max = "<quota> <period>": on the boxes of this
    # pool 16 of the host's 256 hardware threads -- more threads than that only get throttled, which is why the oracle "stopped
    # scaling" at 16 in rounds 2 and 3) and the CPU set it may run on
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    usable = min(cores, quota) if quota else cores
    x, y = synth.invgrow_table("lcdm")
    full = synth.radii_ladder(ns)

    def run(nn, threads, radii):
        dk = synth.philox_density(nn, synth.SEED, 2.5, -2.0)
        o = oracle_lib.Oracle(nn, threads)
        o.set_density(dk)
        o.set_invgrow(x, y)
        o.set_growth(synth.growth_multipliers())
        t0 = time.perf_counter()
        o.compute_fmax(radii, do_lpt=lpt)
        dt = time.perf_counter() - t0
        tm = o.timers()
        o.close()
        return dt, tm

    two = np.array([full[len(full) // 2], full[-1]])
    ncal = min(256, n)
    cal = {}
    for t in sorted({usable, max(1, usable // 2)} if quota else {16, 32, 64, 128}):
        if t <= cores:
            cal[t] = run(ncal, t, two)
    threads = min(cal, key=lambda t: cal[t][0]) if cal else max(1, cores)
    dtc, tmc = cal[threads] if cal else run(ncal, threads, two)
    t_lpt_c = tmc["lpt"] if lpt else 0.0
    scale = (n / ncal) ** 3
    per_radius = (dtc - t_lpt_c) / len(two) * scale
    fit = int((budget_s - t_lpt_c * scale) / max(per_radius, 1e-9))
    if fit >= ns:
        radii, scaled = full, False
    else:
        k = max(1, min(ns - 1, fit - 1))
        radii = np.concatenate([full[len(full) // 2:len(full) // 2 + k][:k], full[-1:]])   # mid-ladder radii + R = 0
        scaled = True
    dt, tm = run(n, threads, radii)
    t_lpt = tm["lpt"] if lpt else 0.0
    t_full = (dt - t_lpt) * ns / len(radii) + t_lpt if scaled else dt
    return {"value": n ** 3 / t_full, "unit": "grid-cells/s", "cores": threads, "kind": "port",
            "host": {"hardware_threads": os.cpu_count(), "cgroup_cpu_quota_cores": quota, "usable": usable,
                     "note": "the job's control group caps its CPU time (cpu.max); `cores` = the threads used, all the job may have"},
            "calibration": {str(t): round(v[0], 3) for t, v in cal.items()},
            "sample": f"{n}^3 box, " + (f"{len(radii)} of the {ns} radii" if scaled else f"all {ns} radii") +
                      f"{' + the whole 3LPT part' if lpt else ''} timed in {dt:.2f} s on {threads} OpenMP threads "
                      f"(fft {tm['fft']:.2f} s, collapse {tm['coll']:.2f} s, lpt {t_lpt:.2f} s)" +
                      (f", sweep time scaled by {ns}/{len(radii)} -> {t_full:.1f} s for the full job" if scaled else ", nothing scaled") +
                      f"; same synthetic spectrum; thread count picked by a {ncal}^3 calibration over {sorted(cal)} threads "
                      f"(seconds: {', '.join(f'{t}: {v[0]:.2f}' for t, v in sorted(cal.items()))}); host has {os.cpu_count()} hardware threads, of which the job's "
                      f"control group allows {quota if quota else 'all'}"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1024, help="grid side (default: the 1024^3 box of the metric)")
    ap.add_argument("--ns", type=int, default=12, help="number of smoothing radii")
    ap.add_argument("--field-bytes", type=int, default=8, choices=(4, 8))
    ap.add_argument("--no-lpt", action="store_true", help="Fmax-only (BASELINE config 2)")
    ap.add_argument("--cpu-n", type=int, default=512, help="grid side of the CPU-baseline sample (0: skip)")
    ap.add_argument("--exact-steps", type=int, default=1, help="steps of the PF_EXACT_LIBM=1 informational run (0: skip)")
    ap.add_argument("--table-steps", type=int, default=2,
                    help="steps of the in-line pass the per-kernel table comes from when the timed configuration overlaps kernels (0: skip)")
    ap.add_argument("--exchange", default="rccl", choices=("rccl", "torch"))
    ap.add_argument("--replicate", default="both", choices=("auto", "0", "1", "both"),
                    help="(N > 1) delta(k) kept whole on every rank (1), exchanged for every transform (0), the library's choice by rank "
                         "count (auto), or -- at 2 and 4 ranks -- the library's choice timed as the line plus the other mode timed "
                         "beside it in exchange.alternative (both, the default: one run decides DESIGN.md section 5's model)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo-host"),
                    help="nccl (= RCCL): one rank per GPU, the measurement.  gloo-host: a bring-up path for boxes with ONE GPU -- gloo process "
                         "group, every rank on device 0, exchanges staged through host memory (pinocchio_amd/dist.py HostStagedKind): it "
                         "exercises this script's multi-rank code end to end (tests/test_gpu_gloo_ranks.py); its numbers mean nothing")
    ap.add_argument("--check", type=int, default=1,
                    help="1: after the timed region, fingerprint Fmax / Rmax / the displacements and compare with the single-GPU golden of the configuration (tests/golden/bench_fingerprints.json)")
    ap.add_argument("--boundary", type=int, default=1,
                    help="1 (one GPU): after the timed region, time the hand-off itself -- pf_get_products, pf_update_products, pf_set_density -- and print it as `boundary`")
    ap.add_argument("--slab-of", type=int, default=0,
                    help="P > 1: time the kernels of ONE rank of a P-rank run of the n^3 box on this GPU, exchange short-circuited (see run_slab)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check: every rank prints its RANK / LOCAL_RANK / WORLD_SIZE as a JSON line and exits; nothing touches a GPU")
    return ap.parse_args(argv)


def gpu_untouched() -> bool:
    """True while this process has neither loaded the HIP library nor imported torch"""
    lib_mod = sys.modules.get("pinocchio_amd._lib")
    return "torch" not in sys.modules and (lib_mod is None or lib_mod._lib is None)


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` invoked bare: start the N ranks as FRESH child processes through torch.distributed.run and
    relay rank 0's line.  This parent never initialises the GPU (no torch import, no library load) and nothing is exec'ed
    from a process that has: the children are ordinary subprocesses, the parent returns their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PF_BENCH_LAUNCHED_BY"] = str(os.getpid())
    # the arguments travel in the environment: torch.distributed.run's own parser would try to read script options that are a
    # prefix of one of its own (`--n 64` is "ambiguous" to it: --nnodes, --nproc-per-node, ...) even behind the script name
    env["PF_BENCH_ARGV"] = json.dumps(list(argv))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    lines = 0
    for line in proc.stdout:          # JSON lines of the ranks go to stdout as they come; the ranks' stderr is inherited
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if args.dry_launch:
        print(json.dumps({"dry_launch": True, "role": "parent", "children_rc": rc, "json_lines_relayed": lines,
                          "parent_gpu_untouched": gpu_untouched(), "command": cmd}), flush=True)
    if rc == 0 and lines == 0:
        print("[bench] the ranks exited 0 without printing a line", file=sys.stderr, flush=True)
        return 4
    return rc


FINGERPRINT_STRIDE = 7      # every 7th cell of the box (a stride coprime to the grid: every plane, row and column is sampled)
FINGERPRINT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "bench_fingerprints.json")


def fingerprint_of_column(values: np.ndarray, first_global: int) -> int:
    """Order-independent, exact fingerprint of one column of products: the sum over the sampled cells of (bits of the value) x
    (a 64-bit weight of the cell's GLOBAL index), modulo 2^64.  Integer arithmetic: the sums of the ranks add up to the
    single-GPU number exactly, whatever the decomposition, if and only if every sampled value is bit for bit the same.
    values: this rank's slab, one 32-bit word per entry ((cells,) or (cells, 3)), first_global: global index of its first entry"""
    flat = np.ascontiguousarray(values).reshape(-1).view(np.uint32)
    total = 0
    chunk = 1 << 24
    start = (-first_global) % FINGERPRINT_STRIDE
    with np.errstate(over="ignore"):
        for a in range(start, flat.size, chunk * FINGERPRINT_STRIDE):
            v = flat[a:a + chunk * FINGERPRINT_STRIDE:FINGERPRINT_STRIDE].astype(np.uint64)
            g = np.uint64(first_global + a) + np.arange(v.size, dtype=np.uint64) * np.uint64(FINGERPRINT_STRIDE)
            w = (g * np.uint64(0x9E3779B97F4A7C15)) ^ (g >> np.uint64(7)) | np.uint64(1)
            total = (total + int(np.sum(v * w, dtype=np.uint64))) & 0xFFFFFFFFFFFFFFFF
    return total


def result_fingerprint(f, rank: int, lpt: bool) -> dict:
    """this rank's share of the fingerprints of Fmax, Rmax and (with the LPT part) the last-formed displacement, 3LPT(b)"""
    cells = f.nxl * f.n * f.n
    out = {"FMAX": fingerprint_of_column(f.block("FMAX"), rank * cells), "RMAX": fingerprint_of_column(f.block("RMAX"), rank * cells)}
    if lpt:
        for name in ("ZEL ", "2LPT", "31PT"):
            out[name] = fingerprint_of_column(f.block(name), 3 * rank * cells)
        # the 3LPT(b) displacement carries the all-reduced mean of the 2LPT source, whose summation order is the decomposition's:
        # equal to a few ulp, not bit for bit (tests/test_gpu_multirank.py) -- its sampled sum of squares instead, in float
        v = f.block("32PT").reshape(-1)[(-3 * rank * cells) % FINGERPRINT_STRIDE::FINGERPRINT_STRIDE].astype(np.float64)
        out["32PT_sumsq"] = float(np.dot(v, v))
    return out


def fingerprints_agree(a: dict, b: dict) -> bool:
    """hex entries (exact sums) equal; float entries (sums of squares of the one column that is not bit-reproducible over
    decompositions) to 1e-6"""
    if set(a) != set(b):
        return False
    for k in a:
        if isinstance(a[k], str) or isinstance(b[k], str):
            if a[k] != b[k]:
                return False
        elif abs(a[k] - b[k]) > 1e-6 * abs(b[k]):
            return False
    return True


def fingerprint_key(n: int, ns: int, lpt: bool, fb: int) -> str:
    return f"n{n}_ns{ns}_{'lpt' if lpt else 'fmax'}_fb{fb}"


def run_config(args, rank, world, device, dist, torch, replicate_env, votes_out, solve_inline=False):
    """One timed configuration on this rank: context, exchange, warm-up, exactly `steps` timed steps between fences.
    replicate_env: None (the library's choice) or "0" / "1" for PF_REPLICATE_DK, read by pf_create.
    solve_inline: PF_SOLVE_BESIDE_Z=0 for this context -- every kernel in line on one stream, so that the HIP-event spans of
    the kernels do not overlap (the pass the per-kernel table comes from; never the timed configuration).
    -> dict of what rank 0 prints (every rank returns its own; only rank 0's is used)"""
    from pinocchio_amd import api, synth
    n, ns, lpt = args.n, args.ns, not args.no_lpt
    if replicate_env is None:
        os.environ.pop("PF_REPLICATE_DK", None)
    else:
        os.environ["PF_REPLICATE_DK"] = replicate_env
    user_beside = os.environ.get("PF_SOLVE_BESIDE_Z")
    if solve_inline:
        os.environ["PF_SOLVE_BESIDE_Z"] = "0"
    try:
        f = api.Fmax(n, rank=rank, nranks=world, device=device, field_bytes=args.field_bytes, timing=True)
    finally:
        if solve_inline:
            if user_beside is None:
                del os.environ["PF_SOLVE_BESIDE_Z"]
            else:
                os.environ["PF_SOLVE_BESIDE_Z"] = user_beside
    keep = None
    res = {"exchange_kind": None}
    try:
        if world > 1:
            from pinocchio_amd import dist as pfdist
            # built-in RCCL exchange first; if some rank cannot bind it, set it up or pass its self-test, the same collectives go
            # through torch.distributed (also RCCL) on tensors aliasing the library's buffers.  Collective and vote-guarded: no
            # rank is left inside a communicator set-up (pinocchio_amd/dist.py).  A failure here exits non-zero; nothing re-execs.
            votes = []
            if args.backend == "gloo-host":
                res["exchange_kind"], keep = pfdist.negotiate_exchange(f, dist, torch, preferred="host", device="cpu", votes=votes,
                                                                       kinds={"host": pfdist.HostStagedKind})
            else:
                res["exchange_kind"], keep = pfdist.negotiate_exchange(f, dist, torch, preferred=args.exchange, device="cuda", votes=votes)
            votes_out[:] = votes
            # how many ranks the communicator that moves the data really has: ncclCommCount for the built-in kind, the
            # process group's size for the torch kind
            rc_count = int(f.L.pf_rccl_comm_count(f.h))
            res["ranks_in_communicator"] = rc_count if res["exchange_kind"] == "rccl" else int(dist.get_world_size())
            rep = int(f.L.pf_replicated_spectrum(f.h))
            if rep < 0:
                raise RuntimeError("pf_replicated_spectrum failed")
            res["replicated_spectrum"] = rep == 1
            if rank == 0:
                print(f"[bench] {world} ranks ({res['ranks_in_communicator']} in the communicator), all-to-all via '{res['exchange_kind']}', "
                      f"delta(k) {'replicated' if rep else 'distributed'}", file=sys.stderr, flush=True)

        f.synth_density(synth.SEED, 2.5, -2.0)
        x, y = synth.invgrow_table("lcdm")
        f.set_invgrow(x, y)
        f.set_growth(synth.growth_multipliers())
        radii = synth.radii_ladder(ns)

        def step(ctx):
            return ctx.compute_fmax(radii, do_lpt=lpt)   # the sweep, then compute_displacements(1, 0) as src/fmax.c:36-190

        def fence(ctx):
            ctx.synchronize()
            if world > 1:
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()

        tv = None
        for _ in range(args.warmup):
            tv = step(f)
        f.reset_kernel_stats()
        f.reset_cputime()
        fence(f)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tv = step(f)
        fence(f)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.backend == "gloo-host" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # what the timed steps left in `products`, after the timed region: exact fingerprints (and the Fmax histogram), to be held
        # against those of a single-GPU run of the same configuration -- a decomposition that moves a block to the wrong place
        # still runs at full speed
        check = None
        if args.check and not solve_inline:
            try:
                pdf = f.Fmax_PDF()   # (collective: all-reduced over the ranks)
                mine = result_fingerprint(f, rank, lpt)
                parts = [mine]
                if world > 1:
                    parts = [None] * world
                    dist.all_gather_object(parts, mine)
                fp = {k: ("%016x" % (sum(p[k] for p in parts) & 0xFFFFFFFFFFFFFFFF) if isinstance(mine[k], int) else float(sum(p[k] for p in parts)))
                      for k in mine}
                fp["PDF"] = "%016x" % (int(np.sum(pdf * (np.arange(1, pdf.size + 1, dtype=np.uint64) * np.uint64(0x100000001B3)), dtype=np.uint64)) & 0xFFFFFFFFFFFFFFFF)
                check = {"fingerprint": fp, "cells_in_fmax_pdf": int(pdf.sum())}
            except (RuntimeError, api.PinfmaxError) as e:
                check = {"error": str(e)}
        res["check"] = check
        # streaming yardsticks of this box and process, after the timed region: a kernel that only reads / writes / copies a field
        stream = None
        if world == 1 and not (n & (n - 1)):
            stream = {}
            v = C.c_double()
            for kind, name in ((0, "read_GBps"), (1, "write_GBps"), (2, "copy_GBps")):
                if f.L.pf_debug_stream_rate(f.h, kind, 5, C.byref(v)) == 0:
                    stream[name] = v.value
        boundary = None
        if world == 1 and args.boundary:
            try:
                boundary = boundary_report(f, api, n, dt / args.steps)
            except (RuntimeError, MemoryError, api.PinfmaxError) as e:
                boundary = {"error": str(e)}
        res["boundary"] = boundary
        global GENERAL_PATH
        GENERAL_PATH = int(f.L.pf_transform_path(f.h)) == 2
        res.update(dt=dt, stats=f.kernel_stats(), cput=f.cputime(), device_gb=f.device_bytes / 1e9, stream=stream,
                   reruns=int(f.L.pf_debug_invariant_reruns(f.h)), sigma_R0=float(np.sqrt(tv[-1])), step=step,
                   solve_beside=int(f.L.pf_solve_ran_beside_zpass(f.h)) == 1)
    finally:
        if keep is not None:
            try:
                keep.release()
            except Exception:  # noqa: BLE001
                pass
        f.close()
        del keep
    return res


def broadcast_verdict(dist, torch, invalid: bool, backend: str) -> bool:
    """every rank leaves with rank 0's verdict (the launcher reports the first non-zero exit).  The flag lives where the process group
    has a backend: an NCCL-only group (the measurement) has none for CPU tensors -- round 5 built it on the CPU and every rank of a real
    multi-GPU run would have died here after printing its line (tests: gloo world 2 on the CPU, nccl world 1 on the GPU)"""
    flag = torch.tensor([1 if invalid else 0], device="cpu" if backend == "gloo-host" else "cuda")
    dist.broadcast(flag, src=0)
    return bool(flag.item())


def boundary_report(f, api, n, step_s):
    """What the boundary itself costs at the bench size ("report separately with the D2H of products"), after the
    timed region and NEVER part of `value`: compute_fmax() of the reference leaves `products` in host memory (src/fmax-pfft.c:563-631
    writes the AoS) and takes kdensity from host memory; the re-entrant compute_displacements of src/fragment.c:398-410 rewrites the
    Vel* fields of records the caller holds.  Every call is timed twice: the first one pays the first touch of the caller's pages."""
    from pinocchio_amd import _lib
    nc = n ** 3
    out = {"host_threads_env": os.environ.get("PF_HANDOFF_THREADS"), "registered": os.environ.get("PF_HOST_REGISTER", "0") == "1"}
    # host memory this process may still take: the smaller of what the machine has available and what its control group leaves (the
    # caller's arrays here are 60 and 112 GB at 1024^3: a measurement that the kernel's out-of-memory killer ends would take the line with it)
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
        lim, cur = open("/sys/fs/cgroup/memory.max").read().strip(), open("/sys/fs/cgroup/memory.current").read().strip()
        if lim != "max":
            avail = min(avail, int(lim) - int(cur)) if avail is not None else int(lim) - int(cur)
    except (OSError, ValueError):
        pass
    out["host_bytes_available"] = avail

    def fits(nbytes):
        return avail is None or nbytes * 1.25 + (8 << 30) < avail

    def timed(fn):
        f.synchronize()
        t0 = time.perf_counter()
        fn()
        f.synchronize()
        return 1e3 * (time.perf_counter() - t0)

    lay = _lib.ProductLayout()
    f.L.pf_layout_3lpt(C.byref(lay))
    if not fits(nc * lay.stride):
        out["skipped"] = "the host does not leave room for the caller's product array"
        return out
    rec = np.empty(nc * lay.stride, dtype=np.uint8)          # untouched pages: the first call faults them in
    get = lambda: f._chk(f.L.pf_get_products(f.h, rec.ctypes.data_as(C.c_void_p), C.byref(lay)))   # noqa: E731
    out["records_GB"] = rec.nbytes / 1e9
    out["d2h_ms_first_call"] = timed(get)
    out["d2h_ms"] = timed(get)
    out["d2h_GBps"] = rec.nbytes / 1e6 / out["d2h_ms"]
    del rec
    # the 104-byte record of the default build with RECOMPUTE_DISPLACEMENTS (src/pinocchio.h:233-259: the 56 bytes above + four *_prev vectors)
    lay2 = _lib.ProductLayout(stride=104, off_Rmax=-1, off_Fmax=-1, off_Vel=8, off_Vel_2LPT=20, off_Vel_3LPT_1=32, off_Vel_3LPT_2=44)
    if fits(nc * 104):
        rec2 = np.empty(nc * 104, dtype=np.uint8)
        upd = lambda: f._chk(f.L.pf_update_products(f.h, rec2.ctypes.data_as(C.c_void_p), C.byref(lay2)))   # noqa: E731
        out["update_records_GB"] = rec2.nbytes / 1e9
        out["update_link_GB"] = nc * 48 / 1e9
        out["update_ms_first_call"] = timed(upd)
        out["update_ms"] = timed(upd)
        out["update_link_GBps"] = nc * 48 / 1e6 / out["update_ms"]
        del rec2
    else:
        out["update_skipped"] = "the host does not leave room for 104-byte records"
    dk = np.empty((n, n, n // 2 + 1), dtype=np.complex128)
    dpp = dk.view(np.float64).ctypes.data_as(C.POINTER(C.c_double))
    out["density_GB"] = dk.nbytes / 1e9
    out["density_d2h_ms_first_call"] = timed(lambda: f._chk(f.L.pf_get_density(f.h, dpp)))
    out["h2d_ms"] = timed(lambda: f._chk(f.L.pf_set_density(f.h, dpp)))
    out["h2d_GBps"] = dk.nbytes / 1e6 / out["h2d_ms"]
    del dk
    out["cells_per_s_including_handoff"] = nc / (step_s + 1e-3 * (out["d2h_ms"] + out["h2d_ms"]))
    out["note"] = ("after the timed region; kdensity in (pf_set_density) + one step + products out (pf_get_products), pageable host arrays, second calls "
                   "(pages touched); never part of `value`")
    return out


def kernel_report(stats, steps, n, w, inline=None, solve_beside=False, stream=None):
    """Everything the line says about single kernels, from the HIP-event statistics of the timed region (`stats`, `steps`) and -- when
    the timed configuration runs two kernel classes beside each other -- of the in-line pass (`inline`: {"stats", "steps", "dt"}).
    -> (kern, table, table_steps, overlapped, roofline, roofline_cls)   (pure: tests/test_bench_report.py feeds it synthetic statistics)"""
    kern = [s for s in stats if s["name"] != "exchange"]          # the timed region's own HIP events
    for s in kern:
        s["symbol"] = symbol_of(s["name"], n, w)
    # the table: kernels in line (their spans add up to the step of that pass); without a second pass, the timed region's
    table, table_steps = kern, steps
    overlapped = set()
    if inline:
        table, table_steps = [s for s in inline["stats"] if s["name"] != "exchange"], inline["steps"]
        for s in table:
            s["symbol"] = symbol_of(s["name"], n, w)
        # the solve of radius i starts beside the z-pass of radius i + 1 and, where it does not fit a CU beside that pass's workgroups,
        # trails into the strided passes that follow: NO span of the timed region is a kernel time then (round 5: the x- and y-passes
        # of a 1024^3 fp32 run showed 282 ms per step in the timed region and 125 in line) -- every per-kernel number comes from the in-line pass
        overlapped = {s["name"] for s in table}
    elif solve_beside:
        overlapped = {s["name"] for s in kern}
    table_total = sum(s["total_ms"] for s in table)

    def roof(group, label, steps):
        tms = sum(s["total_ms"] for s in group)
        byt = sum(s["alg_bytes"] for s in group)
        nl = sum(s["launches"] for s in group)
        ach = byt / (tms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": label, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "launches": nl, "avg_ms": tms / nl, "alg_bytes_per_launch": byt / nl,
                "ms_per_step": tms / steps, "classes": [s["name"] for s in group]}

    def roof_of(names, label):
        """roofline of the launch classes `names`: from the timed region's own events, unless one of them runs beside another
        kernel there (its span is then no kernel time): those come from the in-line pass and say so"""
        live = not (set(names) & overlapped) or not inline
        src, nsteps = (kern, steps) if live else (table, table_steps)
        r = roof([s for s in src if s["name"] in names], label, nsteps)
        tgroup = [s for s in table if s["name"] in names]
        r["share_of_gpu_time"] = sum(s["total_ms"] for s in tgroup) / table_total
        r["measured"] = ("HIP events of the timed region" if live else
                         f"HIP events of the in-line pass ({table_steps} steps, PF_SOLVE_BESIDE_Z=0): in the timed region the collapse solve runs on "
                         "its own stream beside the passes of the next radius, and a span there is not a kernel's time")
        if live and inline:
            r["avg_ms_in_line_pass"] = sum(s["total_ms"] for s in tgroup) / max(1, sum(s["launches"] for s in tgroup))
        return r

    # ranking: by kernel time.  Without the in-line pass (--table-steps 0) the spans of the two classes that share the chip are
    # no kernel times and cannot be ranked: the dominant kernel is then taken among the others, and the line says so
    rank_pool = [s for s in table if s["name"] not in ("zpass_c2r_hess_6to3inv", "collapse_inv")] if (overlapped and not inline) else table
    by_symbol = {}
    for s in rank_pool:
        by_symbol.setdefault(s["symbol"], []).append(s)
    dom_sym = max(by_symbol, key=lambda k: sum(s["total_ms"] for s in by_symbol[k]))
    roofline = roof_of([s["name"] for s in by_symbol[dom_sym]], dom_sym)
    dom_cls = max(rank_pool, key=lambda s: s["total_ms"])
    roofline_cls = roof_of([dom_cls["name"]], dom_cls["symbol"])
    roofline_cls["class"] = dom_cls["name"]
    if overlapped and not inline:
        for r in (roofline, roofline_cls):
            r["ranking"] = ("by HIP-event spans of a timed region in which the collapse solve runs on its own stream beside the passes of the next "
                            "radius: spans, not kernel times -- no in-line pass was made (--table-steps 0); the z-pass and the solve of the "
                            "invariant radii, whose spans overlap most, are left out of the ranking")
            r["share_of_gpu_time"] = None
    # HBM bytes per launch from the PMC counters of the same command (profiles/tools/collect.sh), only if measured on these sources
    pmc = committed_counters("traffic", n, w)
    for r in (roofline, roofline_cls):
        sym = counter_symbol(pmc["kernels"], r["kernel"]) if pmc else None
        if sym:
            k = pmc["kernels"][sym]
            if sym != r["kernel"]:
                r["traffic_kernel"] = sym
            r["traffic"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
            r["traffic_source"] = f"profiles/{PROFILE_ROUND}_pmc_traffic.json (counter passes of this command on these kernel sources, not of this process)"
    # the same kernels against what THIS memory system gives plain streaming kernels (measured above, in this process): reads and
    # writes share the bus, t = reads / read rate + writes / write rate; the split of a kernel's bytes comes from the counters
    if stream and stream.get("read_GBps") and stream.get("write_GBps"):
        for r in (roofline, roofline_cls):
            if pmc and r["kernel"] in pmc["kernels"]:
                k = pmc["kernels"][r["kernel"]]
                t_ceiling = k["fetch_bytes_per_launch"] / (stream["read_GBps"] * 1e9) + k["write_bytes_per_launch"] / (stream["write_GBps"] * 1e9)
                r["streaming_ceiling"] = {"ms_per_launch": 1e3 * t_ceiling, "frac": 1e3 * t_ceiling / r["avg_ms"],
                                          "note": "time a plain streaming kernel of this box needs for the kernel's counted reads and writes (hbm_streaming), over its measured time"}
    pv = committed_counters("valu", n, w)
    if pv and roofline_cls["kernel"] in pv["kernels"]:
        roofline_cls["valu"] = dict(pv["kernels"][roofline_cls["kernel"]], source=f"profiles/{PROFILE_ROUND}_pmc_valu.json")
    if dom_cls["name"].startswith("collapse"):
        roofline_cls["note"] = "the collapse solve is fp64-VALU bound; its HBM stream is a consequence, see DESIGN.md section 6"
    return kern, table, table_steps, overlapped, roofline, roofline_cls


def run_slab(args):
    """--slab-of P: the compute side of ONE rank of a P-rank run of the n^3 box, on one GPU, with the exchange short-circuited
    (pf_set_loopback_exchange: the rank's own blocks come back to it -- copied during the warm-up step so that every buffer holds
    finite, field-like numbers, not moved at all in the timed steps -- and the reductions keep the rank's own contribution).  The
    kernels run with the launch geometry of the real run (line lengths, pitches, tile counts, slab thickness); the numbers in
    the fields are NOT those of the box, so nothing is checked and nothing here is a scaling measurement: it is the time the
    kernels of one rank need per step, e.g. for BASELINE config 5 (2048^3, fp32 fields, eight GPUs), whose box fits no single GPU."""
    from pinocchio_amd import _lib, api, synth
    n, ns, lpt, P = args.n, args.ns, not args.no_lpt, args.slab_of
    f = api.Fmax(n, rank=0, nranks=P, device=0, field_bytes=args.field_bytes, timing=True)
    try:
        f._chk(f.L.pf_set_loopback_exchange(f.h, 1 << 20))          # warm-up: every hand-back copies
        physical = int(f.L.pf_replicated_spectrum(f.h)) == 1
        if physical:
            # the rank keeps the whole delta(k) (PF_REPLICATE_DK=1): a GenIC density is generated whole by the rank itself (pf_genic_density
            # behind the loopback exchange), so the SWEEP of this run is the box's own -- physical fields in every per-cell kernel
            # (round 6; tests/test_gpu_config5.py checks exactly this configuration against the plane oracle)
            f.genic_density(seed=5 * n + P, box_true_mpc=float(n) / 0.7, omega0=0.25, omega_baryon=0.044, hubble100=0.7, primordial_index=0.96, sigma8=0.8)
        else:
            f.synth_density(synth.SEED, 2.5, -2.0)
        x, y = synth.invgrow_table("lcdm")
        f.set_invgrow(x, y)
        f.set_growth(synth.growth_multipliers())
        radii = synth.radii_ladder(ns)
        for _ in range(max(1, args.warmup)):
            f.compute_fmax(radii, do_lpt=lpt)
        f.synchronize()
        # timed: nothing moves (PF_BENCH_LOOPBACK_COPIES=n, an experiment: the first n hand-backs of the timed region copy as in the
        # warm-up, so that the passes behind the exchange see this step's data instead of the warm-up's)
        f._chk(f.L.pf_set_loopback_exchange(f.h, int(os.environ.get("PF_BENCH_LOOPBACK_COPIES", "0"))))
        f.reset_kernel_stats()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            f.compute_fmax(radii, do_lpt=lpt)
        f.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        stats = [s for s in f.kernel_stats() if s["launches"] and s["name"] != "exchange"]
        w = args.field_bytes
        slab_cells = float(n) ** 3 / P
        beside = int(f.L.pf_solve_ran_beside_zpass(f.h)) == 1
        out = {"metric": f"compute side of one rank's slab, {n}^3 box on {P} ranks, loopback exchange (not a scaling measurement)",
               "value": slab_cells / dt, "unit": "grid-cells/s per rank (kernels only)", "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup),
               "ms_per_step": 1e3 * dt, "higher_is_better": True, "data": "synthetic, exchanged blocks replaced by the rank's own",
               "dtype": "f64" if w == 8 else "f32 fields / f64 collapse",
               "sweep_on_the_box_s_own_fields": physical,
               "config": {"workload": f"rank 0 of {P}: slab of {n // P} planes of the {n}^3 box, {ns} radii{' + 3LPT' if lpt else ''}", "grid": n, "slab_of": P,
                          "device_GB": f.device_bytes / 1e9, "kernel_source_sha": _lib.source_sha(),
                          "replicated_spectrum": int(f.L.pf_replicated_spectrum(f.h)) == 1},
               "box_cells_per_s_if_the_exchanges_hide": float(n) ** 3 / dt,
               "kernels": [{"name": s["name"], "symbol": symbol_of(s["name"], n, w), "launches": s["launches"], "ms_per_step": s["total_ms"] / args.steps,
                            "GBps": s["alg_bytes"] / max(s["total_ms"], 1e-9) / 1e6} for s in stats],
               "note": ("kernel spans: zpass_c2r_hess_6to3inv and collapse_inv run beside each other (solve stream) and overlap; " if beside else "") +
                       "an upper bound on what P GPUs can reach on this box: the all-to-alls (P - 1 of P of every field per transform) are not in it.  "
                       "The rank gets its own blocks back as every peer's: the field that makes is not a physical one (its variance is 10^5 times the "
                       "box's, nearly every cell collapses), which the transform passes do not notice and the per-cell classes do -- collapse_inv takes "
                       "about 12 % longer on it than on as many cells of a physical field (profiles/r05_notes.md section 6)"}
        print(json.dumps(out), flush=True)
    finally:
        f.close()


XGMI_GBS_PER_LINK = 153.0   # one xGMI link between two GPUs of a node, per direction


def slab_model(n: int, P: int, fb: int, replicated: bool):
    """the compute side of one rank of a P-rank run of the n^3 box as measured on ONE GPU with the exchange short-circuited
    (`bench.py --slab-of P`, committed as profiles/<round>_slab_<n>_p<P>[_rep<0|1>][_fp32].json): ms per step and rank, or None"""
    names = [f"{PROFILE_ROUND}_slab_{n}_p{P}{'_fp32' if fb == 4 else ''}.json", f"{PROFILE_ROUND}_slab_{n}_p{P}_rep{1 if replicated else 0}{'_fp32' if fb == 4 else ''}.json"]
    for nm in names:
        try:
            with open(os.path.join(ROOT, "profiles", nm)) as fh:
                d = json.loads(fh.read().strip().splitlines()[-1])
            if bool(d["config"].get("replicated_spectrum")) == bool(replicated):
                return {"ms_per_step_per_rank": d["ms_per_step"], "file": "profiles/" + nm, "kernel_source_sha": d["config"].get("kernel_source_sha")}
        except (OSError, ValueError, KeyError, IndexError):
            continue
    return None


def exchange_report(res, args, world=None):
    """what rank 0 saw of the all-to-alls, and what it means: per transposed field the time on the communication stream against the
    compute time beside it, the rate per xGMI link (every peer pair has its own link: a rank's field goes out in P - 1 pieces of
    1 / P each), and the prediction of the one-GPU slab measurements + the survey's link rate for this rank count -- so that ONE
    run on real peers says whether the exchanges hide, whether the link rate is what was assumed, and which way of feeding the
    x-pass (exchange.alternative) is the right default"""
    ex = [s for s in res["stats"] if s["name"] == "exchange"]
    kern = [s for s in res["stats"] if s["name"] != "exchange"]
    e = ex[0] if ex else {"launches": 0, "total_ms": 0.0, "alg_bytes": 0.0}
    P = int(world) if world else int(res.get("ranks_in_communicator") or 0)
    calls = e["launches"] / args.steps
    comm_ms = e["total_ms"] / args.steps
    compute_ms = sum(s["total_ms"] for s in kern) / args.steps
    gb = e["alg_bytes"] / args.steps / 1e9          # bytes of the exchanged fields of this rank, its own block included
    step_ms = 1e3 * res["dt"] / args.steps
    out = {"kind": res["exchange_kind"], "replicated_spectrum": res["replicated_spectrum"],
           "ranks_in_communicator": res["ranks_in_communicator"],
           "ms_per_step": step_ms,
           "calls_per_step": calls, "GB_per_step_per_rank": gb,
           "ms_per_step_on_comm_stream": comm_ms,
           "GBps_per_rank": e["alg_bytes"] / max(e["total_ms"], 1e-9) / 1e6,
           "compute_ms_per_step": compute_ms,
           # (the solve of sweep radius i runs beside the z-pass of radius i + 1: those two classes' spans overlap and the sum above counts the shared time twice)
           "compute_spans_overlap_on_solve_stream": bool(res.get("solve_beside"))}
    if calls > 0 and P > 1:
        wire_gb_per_link = gb / P                    # one of the P - 1 equal pieces that leave over P - 1 links at once
        out["per_transposed_field"] = {"ms_on_comm_stream": comm_ms / calls, "compute_ms_beside_it": compute_ms / calls,
                                       "MB_per_link": 1e3 * wire_gb_per_link / calls}
        out["GBps_per_link"] = wire_gb_per_link / max(comm_ms, 1e-9) * 1e3
        out["link_rate_assumed_GBps"] = XGMI_GBS_PER_LINK
        out["exchange_hidden_fraction"] = max(0.0, min(1.0, (compute_ms + comm_ms - step_ms) / max(comm_ms, 1e-9)))
        m = slab_model(args.n, P, args.field_bytes, bool(res["replicated_spectrum"]))
        wire_ms = 1e3 * wire_gb_per_link / XGMI_GBS_PER_LINK
        pred = {"wire_ms_per_step_at_assumed_link_rate": wire_ms}
        if m:
            pred.update({"compute_ms_per_step_per_rank_on_one_gpu": m["ms_per_step_per_rank"], "from": m["file"],
                         "step_ms_if_exchanges_hide": max(m["ms_per_step_per_rank"], wire_ms), "step_ms_if_nothing_hides": m["ms_per_step_per_rank"] + wire_ms,
                         "measured_step_over_hidden_prediction": step_ms / max(m["ms_per_step_per_rank"], wire_ms)})
        out["model"] = pred
    return out


def main():
    argv = sys.argv[1:]
    if not argv and os.environ.get("PF_BENCH_ARGV") and os.environ.get("PF_BENCH_LAUNCHED_BY"):
        argv = json.loads(os.environ["PF_BENCH_ARGV"])   # a rank started by launch_ranks (see there)
    args = parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    world_env = os.environ.get("WORLD_SIZE")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world_env is None:
        raise SystemExit(launch_ranks(args, argv))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or unset WORLD_SIZE")
    if args.dry_launch:
        os.write(1, (json.dumps({"dry_launch": True, "role": "rank", "rank": rank, "local_rank": local_rank, "world": world,
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                          "launched_by_bench": os.environ.get("PF_BENCH_LAUNCHED_BY") is not None, "args": vars(args),
                          "gpu_untouched": gpu_untouched()}) + "\n").encode())   # one write: the ranks share the pipe
        return
    from pinocchio_amd import _lib, api, synth
    lpt = not args.no_lpt
    n, ns = args.n, args.ns
    if args.slab_of > 1:
        if world != 1:
            raise SystemExit("--slab-of runs on one GPU")
        return run_slab(args)

    dist = torch = None
    devices = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")   # RCCL's own account of a failure goes to stderr (a first multi-GPU lease must not be lost to a silent one)
        # one process per GPU: LOCAL_RANK names the device, unless the launcher already narrowed the visible devices to one
        ndev = torch.cuda.device_count()
        if ndev < 1:
            raise SystemExit("no GPU visible")
        device = local_rank % ndev
        if args.backend == "gloo-host":
            device = 0
            torch.cuda.set_device(device)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        # which physical device each rank drives (N distinct ones, or the line says so)
        props = torch.cuda.get_device_properties(device)
        ident = str(getattr(props, "uuid", "")) or f"{getattr(props, 'pci_bus_id', device)}"
        devices = [None] * world
        dist.all_gather_object(devices, f"{os.uname().nodename}:{ident}")
    else:
        device = 0

    votes = []
    modes = [None if args.replicate in ("auto", "both") else args.replicate]
    try:
        res = run_config(args, rank, world, device, dist, torch, modes[0], votes)
        alt = None
        if world in (2, 4) and args.replicate == "both":
            # the other way of feeding the x-pass, same steps, same fences: the line is the library's
            # default, the alternative is reported beside it
            other = "0" if res["replicated_spectrum"] else "1"
            alt = run_config(args, rank, world, device, dist, torch, other, [])
            os.environ.pop("PF_REPLICATE_DK", None)
    except (RuntimeError, api.PinfmaxError) as e:
        print(f"[rank {rank}] {e}", file=sys.stderr, flush=True)
        print(f"[rank {rank}] exchange negotiation so far: {json.dumps(votes)}", file=sys.stderr, flush=True)
        try:
            b = C.c_int(0)
            print(f"[rank {rank}] RCCL bound at run time: {_lib.load().pf_rccl_version(C.byref(b))}, built against {b.value}; "
                  f"library error: {_lib.load().pf_last_error().decode(errors='replace')}", file=sys.stderr, flush=True)
        except Exception:
            pass
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)
    dt, stats, cput, device_gb, reruns, step = res["dt"], res["stats"], res["cput"], res["device_gb"], res["reruns"], res["step"]
    exchange_kind = res["exchange_kind"]

    # The timed configuration runs the solve of a sweep radius beside the z-pass of the next one (two streams): the HIP-event
    # spans of those two kernel classes overlap and are no shares of the step.  The per-kernel table therefore comes from a
    # second, short pass of the same step with every kernel in line (PF_SOLVE_BESIDE_Z=0); `value` never does.
    inline = None
    if world == 1 and res.get("solve_beside") and args.table_steps > 0:
        import copy
        a2 = copy.copy(args)
        a2.steps, a2.warmup = args.table_steps, 1
        try:
            inline = run_config(a2, rank, world, device, dist, torch, modes[0], [], solve_inline=True)
            inline["steps"] = a2.steps
        except (RuntimeError, api.PinfmaxError) as e:
            print(f"[bench] in-line pass for the kernel table failed: {e}", file=sys.stderr, flush=True)

    exact = None
    if world == 1 and args.exact_steps > 0:
        # the same step with the reference's own libm calls in the solve (bit-comparable with the CPU): what the default
        # arithmetic (series forms, hardware-seeded division: each within ~1 ulp of those calls) buys
        os.environ["PF_EXACT_LIBM"] = "1"
        try:
            with api.Fmax(n, field_bytes=args.field_bytes) as fx:
                fx.synth_density(synth.SEED, 2.5, -2.0)
                x, y = synth.invgrow_table("lcdm")
                fx.set_invgrow(x, y)
                fx.set_growth(synth.growth_multipliers())
                step(fx)
                fx.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.exact_steps):
                    step(fx)
                fx.synchronize()
                exact = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / args.exact_steps, "steps": args.exact_steps,
                         "note": "PF_EXACT_LIBM=1: cos x3, pow, pow, log10, acos, exp and IEEE division / sqrt as the reference calls them"}
        finally:
            del os.environ["PF_EXACT_LIBM"]

    invalid = False   # set by rank 0 when the result check says the timed steps computed something else (see result_check below)
    if rank == 0:
        ms = 1e3 * dt / args.steps
        cells = float(n) ** 3
        value = cells * args.steps / dt
        w = args.field_bytes
        stream = res.get("stream")
        kern, table, table_steps, overlapped, roofline, roofline_cls = kernel_report(stats, args.steps, n, w, inline=inline,
                                                                                     solve_beside=bool(res.get("solve_beside")), stream=stream)
        design_bytes = sum(s["alg_bytes"] for s in kern) / args.steps
        n_gpus = world
        out = {
            "metric": "grid-cells/sec for full Fmax sweep (all smoothing radii) + 3LPT, 1024^3 box"
                      if (n == 1024 and lpt) else f"grid-cells/sec, Fmax sweep{' + 3LPT' if lpt else ''}, {n}^3 box",
            "value": value, "unit": "grid-cells/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if w == 8 else "f32 fields / f64 collapse", "data": "synthetic",
            "config": {"workload": f"{n}^3 box, {ns} smoothing radii, Fmax sweep{' + 2LPT/3LPT displacements' if lpt else ' only'}, "
                                   f"{'fp64' if w == 8 else 'fp32-field'} path, Philox white noise with P(k)~k^-2, sigma(R=0)=2.5",
                       "grid": n, "nsmooth": ns, "lpt": lpt, "parallelism": f"x-slabs over {world} GPU(s)" + (f", all-to-all via {exchange_kind}" if exchange_kind else ""),
                       "device_GB": device_gb, "sigma_R0": res["sigma_R0"], "kernel_source_sha": _lib.source_sha(),
                       "invariant_reruns": reruns},
            "roofline": roofline,
            "roofline_by_class": roofline_cls,
            "hbm_streaming": dict(stream, note="GB/s of kernels that only read, write or copy one field of this context (16 bytes per lane, 5 launches "
                                               "between events), measured after the timed region: the practical ceiling of this memory system beside the "
                                               "8 TB/s of the specification that `peak` quotes") if stream else None,
            "path_roofline": {"design_bytes_per_step_per_gpu": design_bytes,
                              "design_bytes_per_cell": design_bytes * world / cells,
                              "frac_of_hbm_peak_design": design_bytes / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                              "contract_bytes_per_cell": alg_bytes_per_cell(ns, w, lpt),
                              "note": "design: the bytes the shared-pass kernels really move (sum of the per-launch algorithmic bytes) over the whole "
                                      "step -- the whole-path roofline fraction; contract_bytes_per_cell is the survey's figure for the reference's "
                                      "unshared structure, quoted for comparison only (no fraction is formed from it: this design never moves those bytes)"},
            "kernels": [dict({"name": s["name"], "symbol": s["symbol"], "launches": s["launches"], "ms_per_step": s["total_ms"] / table_steps,
                              "GBps": s["alg_bytes"] / max(s["total_ms"], 1e-9) / 1e6},
                             **({"span_ms_per_step_in_timed_region": next((t["total_ms"] for t in kern if t["name"] == s["name"]), 0.0) / args.steps}
                                if s["name"] in overlapped and inline else {})) for s in table],
            "kernel_table": ({"order": "in line", "steps": table_steps, "ms_per_step": 1e3 * inline["dt"] / inline["steps"],
                              "note": "per-kernel times of a second pass of the same step with every kernel in line (PF_SOLVE_BESIDE_Z=0): in the "
                                      "timed region the solve of sweep radius i runs on its own stream beside the z-pass of radius i + 1 "
                                      "and what follows it (DESIGN.md section 3): the HIP-event spans there overlap "
                                      "(span_ms_per_step_in_timed_region) and add up to more than the step"} if inline else
                             {"order": "timed region", "steps": args.steps,
                              "note": ("the collapse solve ran on its own stream beside the passes of the next radius: spans overlap, they are not shares of the step"
                                       if res.get("solve_beside") else "every kernel in line")}),
            "phases_s_per_step": {k: v / args.steps for k, v in cput.items()},
        }
        if world > 1:
            ex = exchange_report(res, args, world)
            ex["process_group_size"] = int(dist.get_world_size())
            ex["kind_votes"] = votes
            ex["devices"] = devices
            ex["distinct_devices"] = len(set(devices))
            # n_gpus is what the communicator and the device census say, not what the command line asked for
            out["n_gpus"] = min(ex["ranks_in_communicator"], ex["distinct_devices"])
            if out["n_gpus"] != world:
                ex["warning"] = f"--gpus {world} but {ex['ranks_in_communicator']} ranks in the communicator on {ex['distinct_devices']} distinct devices"
            ex["note"] = ("rank 0's HIP events around its all-to-alls on the communication stream; they run beside the compute stream (DESIGN.md "
                          "section 5): step time well below compute + exchange means the overlap works.  replicated_spectrum (2-4 ranks by "
                          "default): every rank keeps the whole delta(k), the sweep exchanges nothing and only the LPT sources transpose")
            if alt is not None:
                ex["alternative"] = exchange_report(alt, args, world)
                ex["alternative"]["value"] = cells * args.steps / alt["dt"]
                print("[bench] alternative: " + json.dumps(ex["alternative"]), file=sys.stderr, flush=True)
            out["exchange"] = ex
        chk = res.get("check")
        if chk is not None:
            golden = None
            try:
                golden = json.load(open(FINGERPRINT_FILE)).get(fingerprint_key(n, ns, lpt, w))
            except (OSError, ValueError):
                pass
            if "fingerprint" in chk:
                chk["golden"] = golden["fingerprint"] if golden else None
                chk["matches_single_gpu_golden"] = fingerprints_agree(chk["fingerprint"], golden["fingerprint"]) if golden else None
                # (a golden made from other kernel sources may legitimately differ in last bits: say which it is)
                chk["golden_from_these_sources"] = (golden.get("kernel_source_sha") == _lib.source_sha()) if golden else None
                chk["note"] = ("hex entries: sums over every 7th cell of (bits of the value) x (weight of the global cell index) mod 2^64, added over the ranks -- equal to "
                               "the golden (a single-GPU run of this configuration, tests/golden/make_bench_fingerprints.py) iff those cells are bit for "
                               "bit the same; 32PT_sumsq: the one column whose last bits depend on the summation order of an all-reduce, compared to 1e-6; "
                               "PDF: the 210-bin Fmax histogram; computed after the timed region")
                if golden and not chk["matches_single_gpu_golden"]:
                    print(f"[bench] RESULT CHECK FAILED: fingerprints {chk['fingerprint']} differ from the single-GPU golden {golden['fingerprint']}", file=sys.stderr, flush=True)
            # the other way of feeding the x-pass that was timed beside the line (exchange.alternative): its results are held against the same golden
            ac = alt.get("check") if (world > 1 and alt is not None) else None
            if ac is not None and "exchange" in out and "alternative" in out["exchange"]:
                a = out["exchange"]["alternative"]
                a["result_check_error"] = ac.get("error")
                a["matches_single_gpu_golden"] = (fingerprints_agree(ac["fingerprint"], golden["fingerprint"]) if (golden and "fingerprint" in ac) else None)
            # A wrong result must not pass as a measurement.  With a golden made from THESE kernel sources a mismatch means a block went to
            # the wrong place (or a rank computed something else), and a check that could not run says nothing good either: the line is
            # kept for diagnosis, marked invalid, its metric nulled, and the process exits non-zero.  (A golden from other sources may differ
            # in last bits legitimately: reported, not fatal.)
            same_src = bool(golden) and golden.get("kernel_source_sha") == _lib.source_sha()
            bad = []
            if golden and "error" in chk:
                bad.append("the result check raised: " + str(chk["error"]))
            if same_src and chk.get("matches_single_gpu_golden") is False:
                bad.append("fingerprints differ from the single-GPU golden of these kernel sources")
            if same_src and ac is not None and (("error" in ac) or out["exchange"]["alternative"].get("matches_single_gpu_golden") is False):
                bad.append("the alternative exchange mode's results differ from the golden (or its check raised)")
            out["result_check"] = chk
            if bad:
                out["valid"] = False
                out["invalid_because"] = bad
                out["value_as_measured"] = out["value"]
                out["value"] = None
                invalid = True
        if exact:
            out["exact_libm"] = exact
        if res.get("boundary") is not None:
            out["boundary"] = res["boundary"]
        if world == 1 and args.cpu_n:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n, ns, lpt)
        print(json.dumps(out), flush=True)
    if world > 1:
        invalid = broadcast_verdict(dist, torch, invalid, args.backend)
        dist.destroy_process_group()
    if invalid:
        sys.exit(5)


if __name__ == "__main__":
    main()
