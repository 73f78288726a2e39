"""host-boundary transfer times at 1024^3: pf_set_density (H2D) and pf_get_products (D2H, 56-byte AoS), and the blocks"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinocchio_amd import api, synth
n = int(os.environ.get("ZN", "1024"))
f = api.Fmax(n, timing=True)
f.synth_density(synth.SEED, 2.5, -2.0)
t0 = time.perf_counter(); dk = f.density(); t1 = time.perf_counter()
print("pf_get_density  %.2f s  %.1f GB/s" % (t1 - t0, dk.nbytes / (t1 - t0) / 1e9))
t0 = time.perf_counter(); f.set_density(dk); t1 = time.perf_counter()
print("pf_set_density  %.2f s  %.1f GB/s" % (t1 - t0, dk.nbytes / (t1 - t0) / 1e9))
t_in = t1 - t0
x, y = synth.invgrow_table("lcdm")
f.set_invgrow(x, y); f.set_growth(synth.growth_multipliers())
r = synth.radii_ladder(12)
f.compute_fmax(r, do_lpt=True); f.synchronize()
t0 = time.perf_counter(); f.compute_fmax(r, do_lpt=True); f.synchronize(); t1 = time.perf_counter()
t_c = t1 - t0
print("sweep + displacements %.3f s" % t_c)
t0 = time.perf_counter(); p = f.products(); t1 = time.perf_counter()
t_out = t1 - t0
print("pf_get_products %.2f s  %.1f GB/s" % (t_out, p.nbytes / t_out / 1e9))
t0 = time.perf_counter(); b = f.block("FMAX"); t1 = time.perf_counter()
print("pf_get_block FMAX %.2f s  %.1f GB/s" % (t1 - t0, b.nbytes / (t1 - t0) / 1e9))
print("cells/s device-resident %.3e, with H2D of delta(k) and D2H of products %.3e" % (n ** 3 / t_c, n ** 3 / (t_c + t_in + t_out)))
