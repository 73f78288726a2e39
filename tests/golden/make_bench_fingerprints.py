#!/usr/bin/env python3
"""tests/golden/bench_fingerprints.json: the result fingerprints (bench.py: result_fingerprint) of ONE-GPU runs of the bench's
configurations -- what `bench.py --gpus N` holds its own results against at any N (`result_check`), and what the multi-process tests of
tests/test_gpu_gloo_ranks.py assert.  Run on a GPU box from the repo root; each configuration is one `python3 bench.py` of one step:
    python3 tests/golden/make_bench_fingerprints.py            (all configurations)
    python3 tests/golden/make_bench_fingerprints.py 64 256     (only these grid sizes)
The single-GPU path itself is pinned on the oracle by tests/test_gpu_parity.py and, at 1024^3, by the sampled planes of
tests/test_lpt_analytic.py; this file only carries that result to the decompositions."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONFIGS = [(64, 3, True, 8), (64, 4, True, 8), (256, 12, True, 8), (512, 12, True, 8), (1024, 12, True, 8), (1024, 12, True, 4)]


def main():
    only = {int(a) for a in sys.argv[1:]}
    try:
        out = json.load(open(bench.FINGERPRINT_FILE))
    except (OSError, ValueError):
        out = {}
    for n, ns, lpt, fb in CONFIGS:
        if only and n not in only:
            continue
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--n", str(n), "--ns", str(ns), "--steps", "1", "--warmup", "0", "--cpu-n", "0",
               "--exact-steps", "0", "--table-steps", "0", "--field-bytes", str(fb)] + ([] if lpt else ["--no-lpt"])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr[-2000:])
        d = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        chk = d["result_check"]
        out[bench.fingerprint_key(n, ns, lpt, fb)] = {"fingerprint": chk["fingerprint"], "cells_in_fmax_pdf": chk["cells_in_fmax_pdf"],
                                                     "kernel_source_sha": d["config"]["kernel_source_sha"]}
        print(bench.fingerprint_key(n, ns, lpt, fb), chk["fingerprint"], flush=True)
    json.dump(out, open(bench.FINGERPRINT_FILE, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
