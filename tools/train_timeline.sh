#!/bin/bash
# kernel trace of the training step with the side stream ON: per-queue busy time and overlap of the last step
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=${1:-r2}
TC=gpurun_out/${tag}_tune_train.json
python bench.py --mode train --steps 6 --warmup 2 --tune-cache $TC > /dev/null 2>&1
rm -rf /tmp/trtl
rocprofv3 --kernel-trace --output-format csv -d /tmp/trtl -- python bench.py --mode train --steps 6 --warmup 2 --tune-cache $TC > /dev/null 2>&1
f=$(find /tmp/trtl -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name'])
prev = max(i for i, r in enumerate(rows[:last]) if 'nchw_to_nhwc' in r['Kernel_Name'])
step = rows[prev:last]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
qs = {}
for r in step:
    qs.setdefault(r['Queue_Id'], []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
print('step %.2f ms, queues: %s' % ((t1 - t0) / 1e6, {q: len(v) for q, v in qs.items()}))
def busy(iv):
    iv = sorted(iv); tot = 0; cur_s, cur_e = iv[0][0], iv[0][1]
    for s, e, *_ in iv[1:]:
        if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    return tot + cur_e - cur_s
for q, v in qs.items():
    print('queue %s: busy %.2f ms, first %.2f last %.2f' % (q, busy(v) / 1e6, (v[0][0] - t0) / 1e6, (max(e for _, e, _ in v) - t0) / 1e6))
allb = busy([x for v in qs.values() for x in v])
print('any queue busy %.2f ms -> idle %.2f ms' % (allb / 1e6, (t1 - t0 - allb) / 1e6))
# gaps on the main queue larger than 20 us
main = max(qs.values(), key=len)
main.sort()
gaps = [(main[i + 1][0] - main[i][1], main[i][2][:50], main[i + 1][2][:50]) for i in range(len(main) - 1)]
big = sorted(gaps, reverse=True)[:8]
print('sum of main-queue gaps %.2f ms; largest:' % (sum(g for g, _, _ in gaps) / 1e6))
for g, a, b in big: print('  %.1f us  after %s  before %s' % (g / 1e3, a, b))
PY
