import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from pinocchio_amd import api, synth
x, y = synth.invgrow_table("lcdm")
with api.Fmax(1024, timing=True) as f:
    f.synth_density(synth.SEED, 2.5, -2.0); f.set_invgrow(x, y)
    for rs in ([16.0, 0.0], [1.0, 0.0]):
        f.sweep(np.array(rs)); f.reset_kernel_stats()
        for _ in range(3): f.sweep(np.array(rs))
        s = {k["name"]: k["total_ms"] / 3 for k in f.kernel_stats()}
        print(os.environ.get("PINFMAX_LIB", "default")[-14:], rs[0], "zinv %.2f" % s["zpass_c2r_hess_6to3inv"])
