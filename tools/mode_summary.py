"""Per-kernel summary of a tools/profile_mode.sh output directory: average time (rocprofv3 --stats) and the SQ / memory counters per launch.
Usage: python tools/mode_summary.py gpurun_out/<tag> [max kernels]"""
import csv, glob, collections, sys
d = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 99
stats = {r['Name']: r for r in csv.DictReader(open(glob.glob(f'{d}/stats/*/*_kernel_stats.csv')[0]))}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('pmc_sq', 'pmc_fetch', 'pmc_write'):
    for f in glob.glob(f'{d}/{sub}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
print(open(f'{d}/bench.json').read()[:140])
for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]['TotalDurationNs'])):
    if int(r['Calls']) < 10: continue
    a = {c: sum(v) / len(v) for c, v in acc.get(name, {}).items()}
    if not a: continue
    limit -= 1
    if limit < 0: break
    wc = a.get('SQ_WAVE_CYCLES', 1) or 1
    print(f"{name.split('(')[0][-44:]:44s} {float(r['AverageNs'])/1e3:7.1f} us  VALU {a.get('SQ_INSTS_VALU',0)/1e6:6.1f}M  SALU {a.get('SQ_INSTS_SALU',0)/1e6:5.1f}M  VMEM {a.get('SQ_INSTS_VMEM_RD',0)/1e6:5.2f}M  "
          f"waves {a.get('SQ_WAVES',0)/1e3:6.1f}k  wait {a.get('SQ_WAIT_ANY',0)/wc:.2f}  fetch {a.get('FETCH_SIZE',0)*1024/1e6:6.0f}MB  write {a.get('WRITE_SIZE',0)*1024/1e6:5.0f}MB")
