"""The MPI side of the drop-in boundary, run once under a real MPI.

pinocchio_amd/host/pf_compat.c built the way it is built inside the reference tree (-DPF_IN_PINOCCHIO_TREE: MPI_Bcast of the
RCCL id, pf_init_rccl on every task, communicator count against NTasks in compute_fft_plans; MPI_Barrier in dump_products;
MPI_Bcast of TrueVariance in read_dumps; MPI_Reduce in the host form of Fmax_PDF -- callers src/pinocchio.c:146-150, 229,
src/initialization.c:139, 512, what they replace: src/fmax.c:372-506, 527 and the plans of src/fmax-pfft.c:139-188), linked
against tests/mpi_boundary/mock_pinfmax.c -- a recording mock of include/pinfmax.h, no device -- and driven by
tests/mpi_boundary/driver.c under MPICH's mpiexec with 2 and 4 tasks.  CPU only; skipped where no MPI is installed (the GPU box).
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "mpi_boundary")
ADAPTER = os.path.join(ROOT, "pinocchio_amd", "host", "pf_compat.c")


def _tool(name):
    for cand in (shutil.which(name), os.path.join("/opt/conda/bin", name)):
        if cand and os.path.exists(cand):
            return cand
    return None


MPICC, MPIEXEC = _tool("mpicc"), _tool("mpiexec")
pytestmark = pytest.mark.skipif(not (MPICC and MPIEXEC), reason="no MPI compiler / launcher in this image")


def build(tmp_path, adapter=ADAPTER, name="mpib"):
    exe = str(tmp_path / name)
    cmd = [MPICC, "-cc=gcc", "-std=gnu99", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-Wno-unused-function",
           "-DPF_IN_PINOCCHIO_TREE", "-DPF_TEST_REAL_MPI", "-DTWO_LPT", "-DTHREE_LPT", "-I" + os.path.join(ROOT, "tests", "intree_decls"),
           "-I" + os.path.join(ROOT, "pinocchio_amd", "host"), adapter, os.path.join(HERE, "driver.c"), os.path.join(HERE, "mock_pinfmax.c"), "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def launch(exe, ntasks, cwd, n=16, env=None, timeout=60):
    e = dict(os.environ)
    e.update(env or {})
    os.makedirs(cwd, exist_ok=True)
    return subprocess.run(["timeout", "-k", "2", str(timeout), MPIEXEC, "-n", str(ntasks), exe, str(n)], cwd=cwd, env=e, capture_output=True, text=True)


def facts(stdout, tag):
    """{task: {key: value}} of the driver's lines `TAG task=.. key=value ...`"""
    out = {}
    for line in stdout.splitlines():
        if line.startswith(tag + " "):
            kv = dict(w.split("=", 1) for w in line.split()[1:])
            out[int(kv["task"])] = kv
    return out


def calls(cwd, task):
    with open(os.path.join(cwd, f"calls.{task}.log")) as fh:
        return [ln.strip() for ln in fh if ln.strip()]


def mock_fmax(n, ntasks, task):
    """what the mock's pf_get_products writes into the records of `task`"""
    k = np.arange(n * n * (n // ntasks), dtype=np.int64)
    return (np.float32(0.05) + np.float32(0.1) * ((k + task) % 250).astype(np.float32)).astype(np.float32)


@pytest.mark.parametrize("ntasks", [2, 4])
def test_adapter_under_mpi_makes_its_collectives_in_order(tmp_path, ntasks):
    n = 16
    exe = build(tmp_path)
    cwd = str(tmp_path / f"run{ntasks}")
    r = launch(exe, ntasks, cwd, n)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    grid, plans, fmax, dump, read, pdf = (facts(r.stdout, t) for t in ("GRID", "PLANS", "FMAX", "DUMP", "READ", "PDF"))
    assert sorted(grid) == sorted(plans) == sorted(fmax) == sorted(dump) == sorted(read) == sorted(pdf) == list(range(ntasks))
    for t in range(ntasks):
        # set_one_grid: the x-slab of this task (src/fmax-pfft.c:95-111 for one-dimensional decompositions)
        assert grid[t]["rc"] == "0" and int(grid[t]["xl"]) == n // ntasks and int(grid[t]["x0"]) == t * (n // ntasks)
        assert int(grid[t]["cells"]) == n * n * (n // ntasks)
        assert plans[t]["rc"] == fmax[t]["rc"] == dump[t]["rc"] == read[t]["rc"] == pdf[t]["rc"] == "0"
        log = calls(cwd, t)
        names = [c.split()[0] for c in log]
        # compute_fft_plans: context -> [task 0: unique id] -> (broadcast) -> pf_init_rccl with the broadcast id on EVERY task -> count == NTasks
        assert log[0] == f"pf_create n={n} rank={t} nranks={ntasks} device=0 field_bytes=8"
        assert ("pf_rccl_unique_id" in names) == (t == 0)
        i_init = names.index("pf_init_rccl")
        assert log[i_init] == "pf_init_rccl id_ok=1"                       # the id task 0 made has arrived: the broadcast came first
        if t == 0:
            assert names.index("pf_rccl_unique_id") < i_init
        assert log[i_init + 1] == f"pf_rccl_comm_count -> {ntasks}"
        # compute_fmax: inputs, sweep, displacements, products, histogram from the device, context released
        order = [names.index(x) for x in ("pf_set_density", "pf_set_invgrow", "pf_sweep", "pf_set_growth", "pf_displacements", "pf_get_products", "pf_fmax_pdf", "pf_destroy")]
        assert order == sorted(order) and order[0] > i_init
        assert fmax[t]["tv"] == "1000,1001,1002"
        # read_dumps: TrueVariance is read by task 0 only and reaches the others through the broadcast; the records come back
        assert read[t]["tv"] == "1000,1001,1002"
        F = mock_fmax(n, ntasks, t)
        assert int(read[t]["sum"]) == int(np.sum((F * np.float32(10.) + np.float32(0.5)).astype(np.int64)) + 1000 * t * F.size)
    # dump_products: the reference's files (src/fmax.c:372-426)
    with open(os.path.join(cwd, "Dumps", "summary")) as fh:
        assert fh.read().splitlines() == [f"{ntasks}   # NTasks", "486604   # random seed", f"{n}   # grid size", "56   # length of product_data"]
    assert np.array_equal(np.fromfile(os.path.join(cwd, "Dumps", "TrueVariance")), [1000., 1001., 1002.])
    for t in range(ntasks):
        assert os.path.getsize(os.path.join(cwd, "Dumps", f"Task.{t}")) == 56 * n * n * (n // ntasks)
    # Fmax_PDF without a device context (a run restarted from the dumps): every task bins its own records, MPI_Reduce to task 0
    want = np.zeros(210, dtype=np.int64)
    for t in range(ntasks):
        b = (mock_fmax(n, ntasks, t).astype(np.float64) * 10.).astype(np.int64)      # (int)(F * 10.): the float promoted, src/fmax.c:517-525
        want += np.bincount(np.clip(b, 0, 209), minlength=210)
    got = np.loadtxt(os.path.join(cwd, "pinocchio.mpitest_host.FmaxPDF.out"), usecols=2).astype(np.int64)
    assert np.array_equal(got, want) and got.sum() == n ** 3
    # ... and with one: the library's histogram (already summed over the ranks) written once, by task 0
    dev = np.loadtxt(os.path.join(cwd, "pinocchio.mpitest.FmaxPDF.out"), usecols=2).astype(np.int64)
    assert np.array_equal(dev, np.arange(210) * ntasks)


def test_a_task_outside_the_communicator_is_reported_by_every_task(tmp_path):
    """RCCL came up with fewer ranks than MPI started: compute_fft_plans fails on every task, in the reference's format"""
    exe = build(tmp_path)
    r = launch(exe, 2, str(tmp_path / "outsider"), env={"PF_MOCK_OUTSIDER": "1"})
    plans = facts(r.stdout, "PLANS")
    assert r.returncode != 0 and plans[0]["rc"] == plans[1]["rc"] == "1"
    for t in (0, 1):
        assert f"ERROR on task {t}: the RCCL communicator has 1 ranks, MPI has 2 tasks" in r.stdout


def faulty_adapter(tmp_path, old, new):
    src = open(ADAPTER).read()
    assert src.count(old) == 1
    path = tmp_path / "pf_compat_faulty.c"
    path.write_text(src.replace(old, new).replace('#include "../../include/pinfmax.h"', f'#include "{ROOT}/include/pinfmax.h"'))
    return str(path)


def test_the_test_sees_a_missing_broadcast_and_a_task_that_skips_the_rccl_setup(tmp_path):
    """the checks above can fail: (1) without the broadcast of the id the other tasks hand pf_init_rccl something else;
    (2) a task that never enters pf_init_rccl leaves the others waiting in it (the run does not end)"""
    no_bcast = build(tmp_path, faulty_adapter(tmp_path, "MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);", "memset(id, ThisTask ? 0 : id[0], ThisTask ? 128 : 0);"), "nobcast")
    cwd = str(tmp_path / "nobcast_run")
    r = launch(no_bcast, 2, cwd)
    assert r.returncode != 0
    assert calls(cwd, 0)[-1] == "pf_init_rccl id_ok=1" and calls(cwd, 1)[-1] == "pf_init_rccl id_ok=0"
    skipper = build(tmp_path, faulty_adapter(tmp_path, "if (pf_init_rccl(pf_context, id)) return 1;", "if (ThisTask != 1 && pf_init_rccl(pf_context, id)) return 1;"), "skipper")
    cwd = str(tmp_path / "skipper_run")
    r = launch(skipper, 2, cwd, timeout=10)
    assert r.returncode != 0                                     # killed by the timeout: task 0 still waits for task 1 inside the set-up
    assert "pf_init_rccl" not in [c.split()[0] for c in calls(cwd, 1)]
    assert not facts(r.stdout, "FMAX")
