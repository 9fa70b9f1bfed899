#!/usr/bin/env python
"""Per-step summary of a rocprofv3 kernel_stats.csv: kstats.py FILE STEPS [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
groups = {}
for r in rows:
    n = r['Name']
    key = ('wgrad' if 'wgrad' in n else 'bn' if n.startswith('void bn_') or n.startswith('bn_') else 'conv' if 'conv_' in n or 'stem' in n
           else 'other')
    groups[key] = groups.get(key, 0.0) + float(r['TotalDurationNs']) / 1e6 / steps
for r in rows[:top]:
    print('%7.3f ms %6.1f calls %8.1f us  %s' % (float(r['TotalDurationNs']) / 1e6 / steps, int(r['Calls']) / steps,
                                                 float(r['AverageNs']) / 1e3, r['Name'][:120]))
print({k: round(v, 2) for k, v in groups.items()}, 'total %.2f ms/step' % sum(groups.values()))
