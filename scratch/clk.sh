#!/bin/bash
# engine clock during the kernels: GRBM_GUI_ACTIVE cycles over the kernel durations of the same run's trace
bash scratch/pmc.sh clk GRBM_GUI_ACTIVE GRBM_COUNT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_clk/*/*counter_collection.csv')[0]
t = glob.glob('gpurun_out/pmc_clk/*/*kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(t)):
    dur[r['Dispatch_Id']] = (r['Kernel_Name'].split('(')[0][:48], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    k, ns = dur[r['Dispatch_Id']]
    a = acc[k]; a[0] += float(r['Counter_Value']); a[1] += ns; a[2] += 1
for k, (cyc, ns, n) in acc.items():
    if ns > 1e6: print('%-50s launches %3d  %.3e cycles  %8.2f ms  -> %.3f cycles/ns' % (k, n, cyc, ns / 1e6, cyc / ns))
PY
