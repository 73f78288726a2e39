"""The C example (examples/hmf_validation.c): a plain C host over the C ABI reproduces the reference's validation log."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_compiles_against_the_header():
    """gcc only needs include/pinfmax.h and the shared object (no HIP, no C++ in the host program)"""
    if not os.path.exists(os.path.join(ROOT, "pinocchio_amd", "libpinfmax_hip.so")):
        import __graft_entry__ as g
        g.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s", "-B"])
    assert os.path.exists(os.path.join(ROOT, "examples", "hmf_validation"))


@pytest.mark.gpu
def test_c_host_reproduces_the_validation_log():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    out = subprocess.run([os.path.join(ROOT, "examples", "hmf_validation")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    with open(os.path.join(ROOT, "tests", "golden", "hmf_validation_kat.json")) as fh:
        kat = json.load(fh)
    sig = [float(m.group(1)) for m in re.finditer(r"computed sigma:\s*([0-9.]+)", out.stdout)]
    assert np.allclose(sig, kat["computed_sigma"], atol=1.01e-4)          # four printed decimals on both sides
    coll = int(re.search(r"collapsed particles to z=0: (\d+)", out.stdout).group(1))
    assert abs(coll - kat["collapsed"]) <= 5
