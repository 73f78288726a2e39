"""profiles/r01_pmc_traffic.json from the newest FETCH_SIZE / WRITE_SIZE passes (scratch/pmc.sh fetch|write)"""
import csv, glob, json, os, collections
def newest(tag):
    fs = sorted(glob.glob(f'gpurun_out/pmc_{tag}/*/*counter_collection.csv'), key=os.path.getmtime)
    return fs[-1]
def per_kernel(path, counter):
    tot = collections.defaultdict(float); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter: continue
        k = row['Kernel_Name'].split('(')[0]
        tot[k] += float(row['Counter_Value']); disp[k].add(row['Dispatch_Id'])
    return {k: (tot[k], len(disp[k])) for k in tot}
fe = per_kernel(newest('fetch'), 'FETCH_SIZE'); wr = per_kernel(newest('write'), 'WRITE_SIZE')
names = {'collapse': 'k_collapse<double, true>', 'collapse_inv': 'k_collapse_inv<true>', 'zpass_c2r_hess_6': 'k_c2r_persistent<double, 1024, 4>',
         'zpass_c2r_hess_6to3inv': 'k_c2r_invariants<1024>', 'strided_inverse': 'k_strided<double, 1024, 4, 1>'}
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, each with --kernel-trace only (scratch/pmc.sh: bench.py --n 1024 --ns 2 --no-lpt --steps 1 --warmup 0), KB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane streaming reads on gfx950; averaged over the dispatches of the run (PF_PRUNE_EPS=0: both radii of the run transform every mode; the first goes through the invariant kernels, the last through the six-component ones); strided_inverse = x- and y-pass launches together",
       "config": {"grid": 1024, "field_bytes": 8}, "kernels": {}}
for key, kn in names.items():
    if not [v for k, v in fe.items() if kn in k]: continue
    f = [v for k, v in fe.items() if kn in k][0]; w = [v for k, v in wr.items() if kn in k][0]
    out["kernels"][key] = {"fetch_bytes_per_launch": 2.0 * 1024.0 * f[0] / f[1], "write_bytes_per_launch": 1024.0 * w[0] / w[1], "dispatches": f[1]}
json.dump(out, open('profiles/r01_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(out["kernels"], indent=1))
