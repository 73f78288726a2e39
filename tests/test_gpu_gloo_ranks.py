"""World size 2 with REAL processes on the one GPU of the box: two fresh interpreters (started before either touches the GPU),
a torch.distributed gloo group between them, libpinfmax_hip.so contexts with rank 0 / 1 of 2 on device 0, the exchange
negotiated by pinocchio_amd/dist.py with a failure injected on one rank only, the data moved by the host-staged kind through
the callback ABI (pf_set_exchange, pf_set_exchange_rows, pf_set_allreduce).  Results: bitwise those of one rank.

(The reference: one MPI rank per x-slab, src/initialization.c:1317-1325; the transposes inside pfft_execute,
src/fmax-pfft.c:197, 211; the reductions at src/collapse_times.c:656-667 and src/fmax.c:527.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from pinocchio_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("replicate,scenario", [(1, "setup"), (0, "bind"), (0, "none")])
def test_two_processes_on_one_gpu_match_one_rank(tmp_path, replicate, scenario):
    from pinocchio_amd import api
    n, world = 64, 2
    port = _free_port()
    env = dict(os.environ, PF_REPLICATE_DK=str(replicate), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "gloo_rank_worker.py"), str(r), str(world), str(port), str(n), str(tmp_path), scenario],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][-3000:])
    meta = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    # both ranks took the same decision through the same votes: the flaky kind lost its vote on the step where ONE rank broke
    assert meta[0]["kind"] == meta[1]["kind"] == "host"
    assert [(v["kind"], v["step"], v["all"]) for v in meta[0]["votes"]] == [(v["kind"], v["step"], v["all"]) for v in meta[1]["votes"]]
    if scenario != "none":
        lost = [v for v in meta[0]["votes"] if not v["all"]]
        assert len(lost) == 1 and lost[0]["kind"] == "flaky" and lost[0]["step"] == scenario
        assert [v["here"] for v in meta[0]["votes"] if v["kind"] == "flaky" and v["step"] == scenario] == [True]     # rank 0 was fine
        assert [v["here"] for v in meta[1]["votes"] if v["kind"] == "flaky" and v["step"] == scenario] == [False]    # rank 1 broke
    assert meta[0]["replicated"] == meta[1]["replicated"] == replicate
    assert meta[0]["exchange_calls"] > 0
    if replicate:
        # the sweep exchanges nothing: only the LPT transposes travel
        assert meta[0]["exchange_calls"] < 20

    # one rank, same inputs
    dk = synth.make_density(n, seed=23)
    dk[0, 0, 0] = 0.17 * n ** 3
    x, y = synth.invgrow_table("lcdm")
    with api.Fmax(n) as f:
        f.set_density(dk)
        f.set_invgrow(x, y)
        f.set_growth(synth.growth_multipliers())
        tv = f.compute_fmax(np.array([8.0, 2.0, 1.0, 0.0]), do_lpt=True)
        pdf = f.Fmax_PDF()
        p = f.products()
    nxl = n // world
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert d["tv"] == pytest.approx(tv, rel=1e-13)
        assert np.array_equal(d["pdf"], pdf)
        for k in p.dtype.names:
            assert np.array_equal(d[k], p[k][r * nxl:(r + 1) * nxl]), (r, k)


def test_bench_multi_rank_line_end_to_end_on_one_gpu():
    """`python bench.py --gpus 2` invoked bare (no WORLD_SIZE): the launcher starts two ranks, which -- on this one-GPU box, through
    the bring-up backend (gloo group, both on device 0, host-staged exchange) -- run the script's whole multi-rank path: negotiation,
    timed steps between fences, the alternative delta(k) mode, and the assembly of the line.  The numbers mean nothing here; the
    fields must be there and must tell the truth (two ranks in the communicator, ONE distinct device, hence n_gpus 1 and a warning)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PF_REPLICATE_DK")}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--n", "64", "--ns", "4", "--steps", "1",
                        "--warmup", "1", "--backend", "gloo-host", "--cpu-n", "0", "--exact-steps", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    ex = d["exchange"]
    assert d["steps"] == 1 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
    assert ex["kind"] == "host" and ex["process_group_size"] == 2 and ex["ranks_in_communicator"] == 2
    assert ex["distinct_devices"] == 1 and d["n_gpus"] == 1 and "warning" in ex
    assert ex["kind_votes"] and all(v["all"] and v["here"] for v in ex["kind_votes"])
    assert [v["step"] for v in ex["kind_votes"]] == ["bind", "setup", "selftest"]
    assert ex["replicated_spectrum"] is True and ex["alternative"]["replicated_spectrum"] is False     # the default at two ranks, and the other mode
    assert ex["alternative"]["calls_per_step"] > ex["calls_per_step"] > 0                                 # the sweep's transposes come on top of the LPT ones
    assert ex["GB_per_step_per_rank"] > 0 and d["config"]["grid"] == 64
    # per transposed field: time on the communication stream, compute beside it, bytes and rate per link, the model's wire time
    for e in (ex, ex["alternative"]):
        f = e["per_transposed_field"]
        assert f["ms_on_comm_stream"] > 0 and f["compute_ms_beside_it"] > 0 and f["MB_per_link"] > 0
        assert e["GBps_per_link"] > 0 and 0.0 <= e["exchange_hidden_fraction"] <= 1.0 and e["model"]["wire_ms_per_step_at_assumed_link_rate"] > 0
    assert ex["alternative"]["matches_single_gpu_golden"] is True and d.get("valid", True) is True
    # what the two ranks left in `products` is, bit for bit on the sampled cells, what one GPU leaves (tests/golden/make_bench_fingerprints.py)
    assert d["result_check"]["matches_single_gpu_golden"] is True, d["result_check"]
    names = {k["name"] for k in d["kernels"]}
    assert {"xpass_hess_1to3", "ypass_hess_3to6", "collapse_inv", "zpass_c2r_hess_6to3inv"} <= names


def test_bench_eight_ranks_exchange_every_transform_on_one_gpu():
    """the same through eight real processes sharing the GPU: at eight ranks the library's default is the exchanging path (every
    transform goes through the pipelined all-to-all, delta(k) stays distributed) and `--replicate both` has nothing to add"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PF_REPLICATE_DK")}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "8", "--n", "64", "--ns", "3", "--steps", "1",
                        "--warmup", "1", "--backend", "gloo-host", "--cpu-n", "0", "--exact-steps", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    ex = d["exchange"]
    assert ex["process_group_size"] == 8 and ex["ranks_in_communicator"] == 8 and ex["distinct_devices"] == 1 and d["n_gpus"] == 1
    assert ex["replicated_spectrum"] is False and "alternative" not in ex
    assert ex["calls_per_step"] >= 3 * 3 + 12          # three fields per radius in the sweep, twelve in the LPT part
    assert ex["per_transposed_field"]["MB_per_link"] > 0 and ex["GBps_per_link"] > 0 and "model" in ex
    assert d["value"] > 0 and np.isfinite(d["config"]["sigma_R0"]) and abs(d["config"]["sigma_R0"] - 2.5) < 1e-9
    assert d["result_check"]["matches_single_gpu_golden"] is True, d["result_check"]       # eight slabs, every transform exchanged: one GPU's bits
