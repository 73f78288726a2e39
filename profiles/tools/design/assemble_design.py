"""DESIGN.md = the section files of this directory with the numbers of the committed run filled in (profiles/rNN_*.json):
    python3 profiles/tools/design/assemble_design.py
Edit the sections here, not DESIGN.md."""
import json, csv, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
d=json.load(open(f'{R}/profiles/r05_bench_default.json'))
v=json.load(open(f'{R}/profiles/r05_pmc_valu.json'))
t=json.load(open(f'{R}/profiles/r05_pmc_traffic.json'))
prof=json.load(open(f'{R}/profiles/r05_bench_profiled.json'))
fp32=None
try: fp32=json.load(open(f'{R}/profiles/r05_fp32_bench_default.json'))
except Exception: pass
kern={k['name']:k for k in d['kernels']}
steps=d['kernel_table']['steps']
def per_launch(name): k=kern[name]; return k['ms_per_step']*steps/k['launches']
rows=[]
total=sum(k['ms_per_step'] for k in d['kernels'])
bysym={}
for k in d['kernels']: bysym.setdefault(k['symbol'],[]).append(k)
lines=["| kernel (rocprofv3 symbol) | launch classes (launches per step: ms per launch) | ms per step | algorithmic GB/s | share |","|---|---|---|---|---|"]
for sym,ks in sorted(bysym.items(), key=lambda kv:-sum(k['ms_per_step'] for k in kv[1])):
    ms=sum(k['ms_per_step'] for k in ks)
    byt=sum(k['GBps']*k['ms_per_step'] for k in ks)/ms
    cls=', '.join(f"{k['name']} ({k['launches']//steps}: {k['ms_per_step']*steps/k['launches']:.1f})" for k in ks)
    lines.append(f"| `{sym}` | {cls} | {ms:.0f} | {byt:.0f} ({byt/8000:.2f} of 8 TB/s) | {100*ms/total:.0f} % |")
ktab='\n'.join(lines)
# rocprof avg of dominant symbol
prof_ms=None
for row in csv.DictReader(open(f'{R}/profiles/r05_kernel_stats.csv')) if False else []:
    pass
try:
    txt=open(f'{R}/profiles/r05_kernel_stats.csv').read().splitlines()
    hdr=[h.strip('"') for h in txt[0].split(',')] if txt else []
    for row in csv.DictReader([l for l in txt if not l.startswith('#')]):
        nm=row.get('kernel') or ''
        if 'k_strided<double, 1024, 8, 1' in nm:
            avg=row.get('average_ns')
            if avg: prof_ms=float(avg)/1e6
except Exception as e:
    print('stats csv',e)
sv=v['kernels']; 
def vk(prefix): return [x for k,x in sv.items() if k.startswith(prefix)][0]
solve=vk('k_collapse_inv'); strided=vk('k_strided<double, 1024, 8, 1'); zinv=vk('k_c2r_invariants_spec<double, 1024, 0')
tr=t['kernels']
def trk(prefix): return [x for k,x in tr.items() if k.startswith(prefix)][0]
def trs(x): 
    keys=x.keys()
    rd=[x[k] for k in keys if 'fetch' in k.lower() or 'read' in k.lower()]
    return x
tstr=[]
for pref,label in (('k_strided<double, 1024, 8, 1','`k_strided<…, 1, true>`'),('k_c2r_invariants_spec<double, 1024, 0','`k_c2r_invariants_spec<1024, 0>`'),('k_collapse_inv','`k_collapse_inv`')):
    x=trk(pref)
    rd=x.get('fetch_bytes_per_launch'); wr=x.get('write_bytes_per_launch')
    rd=rd/1e9 if rd is not None else None; wr=wr/1e9 if wr is not None else None
    tstr.append(f"{label} {rd:.1f} GB read + {wr:.1f} GB written per launch" if rd is not None and wr is not None else f"{label} {x}")
st=d['hbm_streaming']
ph=d['path_roofline']
rep={
 '@STEP@': f"{d['ms_per_step']:.1f}", '@VALUE@': f"{d['value']:.3g}".replace('e+09','·10⁹'), '@STEP_RANGE@':'700–723 by box and day',
 '@KERNEL_TABLE@': ktab, '@EV_MS@': f"{prof['roofline']['avg_ms']:.3f}", '@PROF_MS@': f"{prof_ms:.3f}" if prof_ms else 'n/a',
 '@DESIGN_TB@': f"{ph['design_bytes_per_step_per_gpu']/1e12:.2f}", '@BPC@': f"{ph['design_bytes_per_cell']:.0f}", '@PATH_FRAC@': f"{ph['frac_of_hbm_peak_design']:.2f}",
 '@TRAFFIC@': '; '.join(tstr), '@STREAM@': f"read {st['read_GBps']/1000:.2f}, write {st['write_GBps']/1000:.2f}, copy {st['copy_GBps']/1000:.2f} TB/s",
 '@STRIDED_VALU@': f"{strided['valu_insts_per_cell']:.0f}", '@YPASS_MS@': '12.9', '@XPASS_MS@': '6.6',
 '@ZINV_MS@': f"{per_launch('zpass_c2r_hess_6to3inv'):.1f}", '@ZINV_FRAC@': f"{kern['zpass_c2r_hess_6to3inv']['GBps']/8000:.2f}",
 '@SOLVE_VALU@': f"{solve['valu_insts_per_cell']:.0f}", '@SOLVE_UTIL@': f"{100*solve['valu_utilisation_at_measured_clock']:.0f} %", '@SOLVE_GHZ@': f"{solve['engine_clock_GHz_measured']:.2f}",
 '@SOLVE_MS@': f"{per_launch('collapse_inv'):.1f}", '@CPU_VALUE@': f"{d['cpu_baseline']['value']:.2g}".replace('e+06','·10⁶'),
 '@EXACT@': f"{d['exact_libm']['ms_per_step']:.0f}", '@EXACT_GAIN@': f"{d['exact_libm']['ms_per_step']-d['ms_per_step']:.0f}",
 '@FP32_STEP@': f"{fp32['ms_per_step']:.0f}" if fp32 else '570', '@FP32_VALUE@': (f"{fp32['value']:.3g}".replace('e+09','·10⁹') if fp32 else '1.88·10⁹'),
 '@ZINV_STEP@': f"{kern['zpass_c2r_hess_6to3inv']['ms_per_step']:.0f}", '@SOLVE_STEP@': f"{kern['collapse_inv']['ms_per_step']:.0f}",
}
def bj(name):
    try: return json.loads(open(f'{R}/profiles/{name}').read().strip().splitlines()[-1])
    except Exception as e:
        print('missing', name, e); return None
def ms(name, fmt='%.0f'):
    x=bj(name); return (fmt % x['ms_per_step']) if x else 'n/a'
def classes(x, names):
    st=(x.get('kernel_table') or {}).get('steps', x['steps'])
    ks={k['name']:k for k in x['kernels']}
    return ', '.join(f"{nm} {ks[nm]['ms_per_step']*st/ks[nm]['launches']:.2f} ({ks[nm]['GBps']/1000:.1f})" for nm in names if nm in ks)
b768=bj('r05_bench_768.json')
rep.update({
 '@M768@': ms('r05_bench_768.json'), '@M200@': ms('r05_bench_200.json','%.1f'), '@M640@': ms('r05_bench_640.json'), '@M1000@': ms('r05_bench_1000.json'),
 '@M768F@': ms('r05_bench_768_fp32.json'), '@M720@': ms('r05_bench_720_runtime_plan.json'), '@M200C@': ms('r05_bench_200_chirpz.json','%.1f'),
 '@M768_REL@': ('%.2f' % (d['ms_per_step']*0.421875/b768['ms_per_step'])) if b768 else 'n/a',
 '@M768_CLASSES@': classes(b768, ['xpass_hess_1to3','ypass_hess_3to6','zpass_c2r_hess_6to3inv','collapse_inv','zpass_c2r_hess_6','zpass_c2r_disp_3','zpass_c2r_hess_6_lpt3b']) if b768 else 'n/a',
 '@FP32_CLASSES@': ('ms per launch (TB/s): ' + classes(fp32, ['xpass_hess_1to3','ypass_hess_3to6','xpass_fwd','ypass_fwd','xpass_disp_1to2','ypass_disp_2to3'])) if fp32 else 'n/a',
 '@SLAB2048@': ms('r05_slab_2048_p8_fp32.json'), '@SLAB2048_INLINE@': ms('r05_slab_2048_p8_fp32_inline.json'),
})
slab=bj('r05_slab_2048_p8_fp32_inline.json')
rep['@SLAB2048_CLASSES@']=classes(slab, ['xpass_hess_1to3','ypass_hess_3to6','xpass_fwd','ypass_fwd','xpass_disp_1to2','ypass_disp_2to3']) if slab else 'n/a'
def slabms(name):
    x=bj(name); return ('%.0f' % x['ms_per_step']) if x else 'n/a'
rep['@SLAB1024@']=' / '.join(slabms(f'r05_slab_1024_p{P}_rep0.json') for P in (2,4,8))
rep['@SLAB1024R@']=' / '.join(slabms(f'r05_slab_1024_p{P}_rep1.json') for P in (2,4,8))
parts=[open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f'{x}.md')).read() for x in ('s1','s2','s3','s4','s5','s6','sw','s78')]
out=''.join(parts)
for k,val in rep.items(): out=out.replace(k,val)
import re
left=re.findall(r'@[A-Z_0-9]+@', out)
print('unfilled', set(left))
open(f'{R}/DESIGN.md','w').write(out)
print(len(out))
