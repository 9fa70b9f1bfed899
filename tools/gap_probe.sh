#!/bin/bash
# Idle time between consecutive kernels of the forward pass (rocprofv3 kernel trace): tools/gap_probe.sh [bench.py flags]
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
rm -rf /tmp/gp; rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o p -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-northstar --no-train-key --no-f32-key --no-repeats --no-roofline "$@" > /dev/null 2>&1
python - "$(find /tmp/gp -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last 20 steps: find step boundaries by the first kernel name of a step (stem_down / stem)
idx = [i for i, r in enumerate(rows) if 'stem_down' in r[2] or 'stem_mfma' in r[2]]
idx = idx[-21:]
tot = busy = 0; gaps = []; per = collections.defaultdict(list)
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    tot += seg[-1][1] - seg[0][0]
    for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
        g = s1 - e0
        gaps.append(g); per[n1[:70]].append(g)
    busy += sum(e - s for s, e, _ in seg)
n = len(idx) - 1
print('steps %d  launches/step %.1f  span/step %.1f us  kernel time/step %.1f us  gaps/step %.1f us  mean gap %.2f us  median %.2f' % (
    n, (idx[-1] - idx[0]) / n, tot / n / 1e3, busy / n / 1e3, sum(gaps) / n / 1e3, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print('  before %-70s n/step %4.1f  mean gap %.2f us' % (k, len(v) / n, sum(v) / len(v) / 1e3))
PY
