#!/bin/bash
# row-walk weight gradient: ablations + one kernel trace (run on the GPU box)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for dbg in 0 1 2 3; do for tg in 256 512; do
  echo "== DBG $dbg target $tg"
  YOLO_WW_DBG=$dbg YOLO_WW_TARGET=$tg timeout -k 5 120 python tools/train_ops_bench.py --batch 64 --what wgrad --algos 2,3 --k3s1 2>&1 | grep wgrad
done; done
rm -rf /tmp/wwprof; YOLO_WW_TARGET=512 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wwprof -- python tools/train_ops_bench.py --batch 64 --what wgrad --algos 1,2,3 --k3s1 > /dev/null 2>&1
f=$(find /tmp/wwprof -name "*kernel_stats.csv" | head -1); cut -c1-200 "$f"
