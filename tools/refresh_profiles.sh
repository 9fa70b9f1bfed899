#!/bin/bash
# Regenerates the measurement artefacts behind profiles/ on a GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh <tag>      -> gpurun_out/<tag>_*  (copy the ones to keep into profiles/)
# 1. un-profiled bench lines (608 bs64, 608 bs64 + NMS, training bs64; the 416 bs32 headline line after step 3)
# 2. rocprofv3 --kernel-trace --stats of the headline bench with the kernel choices pinned by a tune cache
# 3. two PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass) summarised by tools/pmc_traffic.py
# 4. rocprofv3 kernel stats of the training step
set -u
TAG=${1:-r02}
PT=${2:-r02}            # prefix of the copies under profiles/
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TC=$OUT/${TAG}_tune_416.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune-cache $TC > $OUT/${TAG}_bench_first.json 2> $OUT/${TAG}_bench.err     # (measures the kernel choices; the headline line is printed at the end)
python bench.py --steps 10 --warmup 3 --size 608 --batch 64 --no-cpu-baseline --tune-cache $OUT/${TAG}_tune_608.json > $OUT/${TAG}_bench_608.json 2>> $OUT/${TAG}_bench.err
python bench.py --steps 10 --warmup 3 --size 608 --batch 64 --post nms --no-cpu-baseline --no-roofline --tune-cache $OUT/${TAG}_tune_608.json > $OUT/${TAG}_bench_608_nms.json 2>> $OUT/${TAG}_bench.err
python bench.py --mode train --steps 10 --warmup 2 --tune-cache $OUT/${TAG}_tune_train.json > $OUT/${TAG}_train_bench.json 2>> $OUT/${TAG}_bench.err
# profiled passes (kernel choices come from the caches written above: no autotune launches in the trace)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-northstar --no-train-key --no-f32-key --no-repeats --tune-cache $TC > $OUT/${TAG}_bench_profiled.json 2>> $OUT/${TAG}_bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_f -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-northstar --no-train-key --no-f32-key --no-repeats --tune-cache $TC > /dev/null 2>> $OUT/${TAG}_bench.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_w -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-northstar --no-train-key --no-f32-key --no-repeats --tune-cache $TC > /dev/null 2>> $OUT/${TAG}_bench.err
python tools/pmc_traffic.py $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w 32 416 $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_summary.txt 2>&1
# the headline line last, so that its roofline.traffic comes from THIS box's PMC passes (bench.py reads profiles/*_pmc_traffic.json)
cp $OUT/${TAG}_pmc_traffic.json profiles/${PT}_pmc_traffic.json
python bench.py --steps 20 --warmup 5 --tune-cache $TC > $OUT/${TAG}_bench.json 2>> $OUT/${TAG}_bench.err
# the same two passes for the 608x608 bs 64 shape, then its bench line again with roofline.traffic filled in
T6=$OUT/${TAG}_tune_608.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc6_f -o p -- python bench.py --steps 3 --warmup 2 --size 608 --batch 64 --no-cpu-baseline --no-roofline --no-northstar --no-train-key --no-f32-key --no-repeats --tune-cache $T6 > /dev/null 2>> $OUT/${TAG}_bench.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc6_w -o p -- python bench.py --steps 3 --warmup 2 --size 608 --batch 64 --no-cpu-baseline --no-roofline --no-northstar --no-train-key --no-f32-key --no-repeats --tune-cache $T6 > /dev/null 2>> $OUT/${TAG}_bench.err
python tools/pmc_traffic.py $OUT/${TAG}_pmc6_f $OUT/${TAG}_pmc6_w 64 608 $OUT/${TAG}_608_pmc_traffic.json >> $OUT/${TAG}_pmc_summary.txt 2>&1
cp $OUT/${TAG}_608_pmc_traffic.json profiles/${PT}_608_pmc_traffic.json
python bench.py --steps 10 --warmup 3 --size 608 --batch 64 --no-cpu-baseline --tune-cache $T6 > $OUT/${TAG}_bench_608.json 2>> $OUT/${TAG}_bench.err
rm -rf $OUT/${TAG}_pmc6_f $OUT/${TAG}_pmc6_w
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_m -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-northstar --no-train-key --no-f32-key --no-repeats --tune-cache $TC > /dev/null 2>> $OUT/${TAG}_bench.err
python tools/pmc_mfma.py $OUT/${TAG}_pmc_m $OUT/${TAG}_pmc_mfma.json >> $OUT/${TAG}_pmc_summary.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train -o p -- python bench.py --mode train --steps 10 --warmup 2 --tune-cache $OUT/${TAG}_tune_train.json > $OUT/${TAG}_train_bench_profiled.json 2>> $OUT/${TAG}_bench.err
cp $OUT/${TAG}_prof/p_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
cp $OUT/${TAG}_prof_train/p_kernel_stats.csv $OUT/${TAG}_train_kernel_stats.csv 2>/dev/null
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_prof_train $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w $OUT/${TAG}_pmc_m
ls -la $OUT | grep ${TAG}_
tail -c 600 $OUT/${TAG}_bench.json
