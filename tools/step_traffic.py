#!/usr/bin/env python3
"""HBM-side bytes per encode step from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs of the same bench
command).  FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section); both counters are
in KiB.  usage: step_traffic.py <name> <steps_in_run> <fetch_dir> [<write_dir>]  -> one line per kernel class + the total."""
import collections, csv, glob, json, re, sys

def load(d, counter):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    agg, cnt = collections.defaultdict(float), collections.Counter()
    if not fs:
        return agg, cnt
    occ = collections.Counter()
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r['Dispatch_Id']))
    for r in rows:
        if r['Counter_Name'] != counter:
            continue
        k = re.sub(r'^void hg::', '', r['Kernel_Name'])
        k = re.sub(r'\(.*', '', k)[:48]
        if 'gemm_ring2<10>' in k or 'gemm_ring<4, 10' in k:      # out_proj / c_proj alternate (12 + 11 per all-rows step)
            n = occ[k]; occ[k] += 1
            k += ' out_proj' if (n % 23) % 2 == 0 else ' c_proj'
        agg[k] += float(r['Counter_Value'])
        cnt[k] += 1
    return agg, cnt

name, steps = sys.argv[1], int(sys.argv[2])
f, fc = load(sys.argv[3], 'FETCH_SIZE')
w, wc = load(sys.argv[4], 'WRITE_SIZE') if len(sys.argv) > 4 else ({}, {})
mb = lambda kib: kib * 1024 / 1e6
keys = sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0)))
tot_f = sum(2 * mb(v) for v in f.values()) / steps
tot_w = sum(mb(v) for v in w.values()) / steps
print(f"[traffic] {name}: per step fetched {tot_f:.0f} MB (x2 corrected) + written {tot_w:.0f} MB = {tot_f + tot_w:.0f} MB")
for k in keys[:12]:
    n = max(fc.get(k, 0), wc.get(k, 0), 1)
    print(f"[traffic] {name}:   {k:58s} n/step {n / steps:5.1f}  fetch {2 * mb(f.get(k, 0)) / n:7.1f} MB  write {mb(w.get(k, 0)) / n:7.1f} MB per launch")
print(json.dumps({"name": name, "fetch_mb_per_step": round(tot_f, 1), "write_mb_per_step": round(tot_w, 1)}))
