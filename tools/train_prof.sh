#!/bin/bash
# kernel trace of the training step with the weight gradients on the main stream (un-overlapped per-kernel times);
# kernel choices pinned by a tune cache written by an un-profiled run first
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=${1:-r2}
TC=gpurun_out/${tag}_tune_train.json
python bench.py --mode train --steps 10 --warmup 2 --tune-cache $TC > gpurun_out/${tag}_train.json 2>/dev/null
rm -rf /tmp/trprof
YOLO_LAB=1 YOLO_TRAIN_SERIAL_WGRAD=${SERIAL-1} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trprof -- python bench.py --mode train --steps 10 --warmup 2 --tune-cache $TC > gpurun_out/${tag}_train_serial.json 2>/dev/null
f=$(find /tmp/trprof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_train_serial_kernel_stats.csv
f=$(find /tmp/trprof -name "*kernel_trace.csv" | head -1); python - "$f" > gpurun_out/${tag}_train_last_step.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last training step: from the last nchw_to_nhwc launch on
last = max(i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name'])
t0 = int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    print('%9.1f %8.1f us  grid %-8s wg %-5s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                              r['Grid_Size_X'], r['Workgroup_Size_X'], r['Kernel_Name'][:90]))
PY
