#!/bin/bash
# Samples the shader clock and socket power (rocm-smi) while the 608x608 bs 64 forward pass runs back to back:
# evidence for the sustained clock the MFMA-dense kernels actually get.
#   bash tools/clock_probe.sh [out.txt]      (run through gpurun from the repo root)
set -u
OUT=${1:-gpurun_out/clock_probe.txt}
cd "$GRAFT_REPO_ROOT"
python bench.py --steps 10 --warmup 3 --size 608 --batch 64 --no-cpu-baseline --no-roofline --tune-cache gpurun_out/tc_clk.json > /dev/null 2>&1
echo "idle:" > $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> $OUT
python bench.py --steps 1200 --warmup 3 --size 608 --batch 64 --no-cpu-baseline --no-roofline --tune-cache gpurun_out/tc_clk.json > gpurun_out/clock_probe_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8; do
  echo "sample $i:" >> $OUT
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> $OUT
  sleep 1
done
wait $BP
cut -c1-200 gpurun_out/clock_probe_bench.json >> $OUT
cat $OUT
