import json,sys
for f in sys.argv[1:]:
    l=[x for x in open(f) if x.startswith('{')]
    d=json.loads(l[-1])
    print(f, d['value'], d['ms_per_step'], ' | '.join(f"{k['name'][:22]} {k['avg_ms']*1e3:.0f}x{k['launches_per_step']:.0f}" for k in d['kernels']))
