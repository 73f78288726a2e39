"""What the replicated delta(k) costs in kernels (one GPU, P virtual ranks of the in-process fabric, 512^3 fp64, the bench's
spectrum and ladder): the P ranks share one device, so the wall time of a step is the SUM of their kernels -- divided by P it
estimates one rank's compute time on P real GPUs (no link time in it: the fabric's exchange is a device copy).
    python3 profiles/tools/replication_model.py [n]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pinocchio_amd import _lib, api, synth  # noqa: E402


def step_time(n, P, replicate, steps=2):
    os.environ["PF_REPLICATE_DK"] = str(replicate)
    L = _lib.load()
    radii = synth.radii_ladder(12)
    x, y = synth.invgrow_table("lcdm")
    g = synth.growth_multipliers()
    fab = L.pf_fabric_create(P) if P > 1 else None
    ctxs = [api.Fmax(n, rank=r, nranks=P, timing=True) for r in range(P)]
    for c in ctxs:
        if fab:
            assert L.pf_fabric_attach(fab, c.h) == 0
    bar = threading.Barrier(P)
    out = [None] * P

    def work(r):
        f = ctxs[r]
        f.synth_density(synth.SEED, 2.5, -2.0)
        f.set_invgrow(x, y); f.set_growth(g)
        f.compute_fmax(radii, do_lpt=True)       # warm-up, includes the gather
        f.synchronize(); bar.wait()
        f.reset_kernel_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            f.compute_fmax(radii, do_lpt=True)
        f.synchronize(); bar.wait()
        ks = {k["name"]: k["total_ms"] / steps for k in f.kernel_stats()}
        out[r] = ((time.perf_counter() - t0) / steps, ks)

    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in ctxs:
        c.close()
    if fab:
        L.pf_fabric_destroy(fab)
    wall = max(o[0] for o in out)
    return wall, out[0][1]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    t1, _ = step_time(n, 1, 0)
    print(f"n = {n}: one rank {1e3 * t1:.1f} ms per step")
    for P in (2, 4, 8):
        for rep in (0, 1):
            w, ks = step_time(n, P, rep)
            print(f"P = {P} replicate = {rep}: all ranks on one device {1e3 * w:.1f} ms per step -> {1e3 * w / P:.1f} ms per rank "
                  f"(x {t1 / (w / P):.2f} of one rank's step); exchange class {ks.get('exchange', 0.0):.1f} ms, x-pass classes "
                  f"{ks.get('xpass_hess_1to3', 0.0) + ks.get('xpass_disp_1to2', 0.0):.1f} ms (as timed under sharing)")
