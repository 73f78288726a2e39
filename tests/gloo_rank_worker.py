"""One rank of a world-size-2 run of libpinfmax_hip.so with REAL processes (tests/test_gpu_gloo_ranks.py starts two of these, each
a fresh interpreter, both on the one GPU of the box): torch.distributed gloo group for the control plane and the host-staged
exchange of pinocchio_amd/dist.py for the data plane.  Writes its slab of the products to <outdir>/rank<r>.npz."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    rank, world, port, n, outdir, scenario = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pinocchio_amd import api, synth
    from pinocchio_amd import dist as pfdist

    class Flaky(pfdist.HostStagedKind):
        """a kind that breaks on ONE rank only, at the step the scenario names: the ranks must still decide together"""
        name = "flaky"

        def can_bind(self):
            return not (scenario == "bind" and rank == 1)

        def setup(self):
            ok = super().setup()
            return ok and not (scenario == "setup" and rank == 1)

        def release(self):
            # what a real kind does when it is dropped: nothing of it stays installed
            L = self.f.L
            L.pf_set_exchange(self.f.h, pfdist._lib.ALLTOALL_FN(), None)
            L.pf_set_exchange_rows(self.f.h, pfdist._lib.ALLTOALLV_FN(), None)
            L.pf_set_allreduce(self.f.h, pfdist._lib.ALLREDUCE_FN(), None)
            super().release()

    f = api.Fmax(n, rank=rank, nranks=world, device=0, timing=True)
    votes = []
    kinds = {"flaky": Flaky, "host": pfdist.HostStagedKind}
    name, keep = pfdist.negotiate_exchange(f, dist, torch, preferred="flaky" if scenario != "none" else "host", device="cpu",
                                           kinds=kinds, votes=votes, log=lambda m: print(m, flush=True))
    dk = synth.make_density(n, seed=23)
    dk[0, 0, 0] = 0.17 * n ** 3
    nxl = n // world
    x, y = synth.invgrow_table("lcdm")
    f.set_density(dk[rank * nxl:(rank + 1) * nxl])
    f.set_invgrow(x, y)
    f.set_growth(synth.growth_multipliers())
    radii = np.array([8.0, 2.0, 1.0, 0.0])      # the first one is band-limited at n = 64: the row-range form of the exchange
    tv = f.compute_fmax(radii, do_lpt=True)
    pdf = f.Fmax_PDF()
    p = f.products()
    stats = {s["name"]: s for s in f.kernel_stats()}
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), tv=tv, pdf=pdf, **{k: p[k] for k in p.dtype.names})
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as fh:
        json.dump({"kind": name, "votes": votes, "replicated": int(f.L.pf_replicated_spectrum(f.h)),
                   "exchange_calls": stats.get("exchange", {}).get("launches", 0),
                   "exchange_bytes": stats.get("exchange", {}).get("alg_bytes", 0.0)}, fh)
    keep.release()
    f.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
