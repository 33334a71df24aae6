"""Summarise a rocprofv3 --kernel-trace CSV: per-step kernel time, launch count and gaps."""
import collections, csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'state_pre_kernel' in r['Kernel_Name']]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[skip], starts[-1]
seg = rows[a:b]
ns = len(starts) - 1 - skip
wall = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3 / ns
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3 / ns
print('per step: wall %.1f us, busy %.1f us, launches %.1f' % (wall, busy, len(seg) / ns))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    k = r['Kernel_Name'].split('(')[0][-44:]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print(f"{k:46s} {c/ns:5.1f} calls/step {t/ns:8.1f} us/step")
gaps = [(int(seg[i + 1]['Start_Timestamp']) - int(seg[i]['End_Timestamp'])) / 1e3 for i in range(len(seg) - 1)]
print('gap mean %.2f us, total per step %.1f us' % (np.mean(gaps), np.sum(gaps) / ns))
