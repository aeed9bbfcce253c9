"""Per-kernel averages of the rocprofv3 --pmc csv passes written by tools/pmc_run.sh."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '?')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    if 'pack' in k or 'fill' in k.lower() or 'Cijk' in k or 'elementwise' in k or 'distribution' in k:
        continue
    print('== %s' % k[:110])
    for c, v in sorted(cs.items()):
        v = v[1:] if len(v) > 2 else v          # drop the first (cold) launch
        print('   %-28s n=%d avg %.4g' % (c, len(v), sum(v) / len(v)))
    if 'FETCH_SIZE' in cs or 'WRITE_SIZE' in cs:
        f = cs.get('FETCH_SIZE', [0]); w = cs.get('WRITE_SIZE', [0])
        f = f[1:] if len(f) > 2 else f; w = w[1:] if len(w) > 2 else w
        fb, wb = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
        print('   HBM traffic/launch: read %.2f MB (FETCH_SIZE KB x1024, x2 gfx950 wide-read correction = %.2f MB), write %.2f MB'
              % (fb / 1e6, 2 * fb / 1e6, wb / 1e6))
