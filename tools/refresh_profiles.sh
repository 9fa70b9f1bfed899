#!/bin/bash
# Regenerates the measurement artefacts behind profiles/ on a GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh <tag> <prefix>     -> gpurun_out/<prefix>_*   (gpurun merges gpurun_out/ back; then: cp gpurun_out/<prefix>_* profiles/)
# EVERY command launches the COMMITTED plan (profiles/plan.json, bench.py --tune plan = the default): the bench lines, the
# rocprofv3 kernel traces and the PMC passes describe the same launches (plan_md5 is carried into the PMC summaries and checked
# by bench.py before it quotes roofline.traffic).  Per workload W in {416 bs 32 (headline), 608 bs 64 (north star)}:
#   kernel trace + stats -> <prefix>[_608]_kernel_stats.csv      two PMC passes (FETCH_SIZE | WRITE_SIZE) -> _pmc_traffic.json
#   one PMC pass (MFMA busy, GUI active, wave cycles)            -> _pmc_mfma.json          bench line last (traffic filled in)
# then the training step: bench line + kernel stats.
set -u
TAG=${1:-r06}
PT=${2:-r06}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ERR=$OUT/${TAG}_bench.err
LEAN="--no-cpu-baseline --no-northstar --no-train-key --no-f32-key --no-repeats"
md5_of() { python -c "import sys,json; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['plan_md5'])" $1; }
for W in 416 608; do
  if [ $W = 416 ]; then SH=""; SUF=""; B=32; else SH="--size 608 --batch 64"; SUF="_608"; B=64; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof$SUF -o p -- python bench.py --steps 50 --warmup 5 $SH $LEAN > $OUT/${TAG}_bench${SUF}_profiled.json 2>> $ERR
  cp $OUT/${TAG}_prof$SUF/p_kernel_stats.csv $OUT/${PT}${SUF}_kernel_stats.csv 2>/dev/null
  MD5=$(md5_of $OUT/${TAG}_bench${SUF}_profiled.json)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_f -o p -- python bench.py --steps 5 --warmup 2 $SH $LEAN --no-roofline > /dev/null 2>> $ERR
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_w -o p -- python bench.py --steps 5 --warmup 2 $SH $LEAN --no-roofline > /dev/null 2>> $ERR
  python tools/pmc_traffic.py $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w $B $W $OUT/${PT}${SUF}_pmc_traffic.json $MD5 7 >> $OUT/${TAG}_pmc_summary.txt 2>&1
  cp $OUT/${PT}${SUF}_pmc_traffic.json profiles/${PT}${SUF}_pmc_traffic.json      # (bench.py reads profiles/*_pmc_traffic.json: the lines below quote THIS box's passes)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_m -o p -- python bench.py --steps 5 --warmup 2 $SH $LEAN --no-roofline > /dev/null 2>> $ERR
  python tools/pmc_mfma.py $OUT/${TAG}_pmc_m $OUT/${PT}${SUF}_pmc_mfma.json >> $OUT/${TAG}_pmc_summary.txt 2>&1
  rm -rf $OUT/${TAG}_prof$SUF $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w $OUT/${TAG}_pmc_m
done
cp $OUT/${TAG}_pmc_summary.txt $OUT/${PT}_pmc_summary.txt
cp $OUT/${TAG}_bench_profiled.json $OUT/${PT}_bench_profiled.json
# bench lines (un-profiled): 608 bs 64 alone and with NMS, the training step, then the full headline line (the driver's command)
python bench.py --steps 10 --warmup 3 --size 608 --batch 64 --no-cpu-baseline > $OUT/${PT}_bench_608.json 2>> $ERR
python bench.py --steps 10 --warmup 3 --size 608 --batch 64 --post nms --no-cpu-baseline --no-roofline > $OUT/${PT}_bench_608_nms.json 2>> $ERR
python bench.py --mode train --steps 20 --warmup 3 > $OUT/${PT}_train_bench.json 2>> $ERR
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train -o p -- python bench.py --mode train --steps 10 --warmup 2 > $OUT/${PT}_train_bench_profiled.json 2>> $ERR
cp $OUT/${TAG}_prof_train/p_kernel_stats.csv $OUT/${PT}_train_kernel_stats.csv 2>/dev/null
rm -rf $OUT/${TAG}_prof_train
# (round 6) the training step's counters: HBM traffic per kernel (two passes) and MFMA-busy, 1 warm-up + 3 steps each; bench.py's
# train_416_bs64.roofline.traffic sums the BatchNorm-backward kernels of profiles/*_train_pmc_traffic.json
MD5T=$(python -c "import sys,json; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['plan']['state_md5'])" $OUT/${PT}_train_bench_profiled.json)
TR="--mode train --steps 3 --warmup 1 --no-roofline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_tf -o p -- python bench.py $TR > /dev/null 2>> $ERR
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_tw -o p -- python bench.py $TR > /dev/null 2>> $ERR
python tools/pmc_traffic.py $OUT/${TAG}_pmc_tf $OUT/${TAG}_pmc_tw 64 416 $OUT/${PT}_train_pmc_traffic.json $MD5T 4 >> $OUT/${TAG}_pmc_summary.txt 2>&1
cp $OUT/${PT}_train_pmc_traffic.json profiles/${PT}_train_pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_tm -o p -- python bench.py $TR > /dev/null 2>> $ERR
python tools/pmc_mfma.py $OUT/${TAG}_pmc_tm $OUT/${PT}_train_pmc_mfma.json >> $OUT/${TAG}_pmc_summary.txt 2>&1
rm -rf $OUT/${TAG}_pmc_tf $OUT/${TAG}_pmc_tw $OUT/${TAG}_pmc_tm
cp $OUT/${TAG}_pmc_summary.txt $OUT/${PT}_pmc_summary.txt
python bench.py --mode train --steps 20 --warmup 3 > $OUT/${PT}_train_bench.json 2>> $ERR        # (again: now with roofline.traffic)
# (round 6) the split bf16 parity path (dtype bf16x3) on the headline workload: kernel trace + bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_x3 -o p -- python bench.py --dtype bf16x3 --steps 20 --warmup 3 $LEAN > $OUT/${PT}_bench_x3_profiled.json 2>> $ERR
cp $OUT/${TAG}_prof_x3/p_kernel_stats.csv $OUT/${PT}_x3_kernel_stats.csv 2>/dev/null
rm -rf $OUT/${TAG}_prof_x3
python bench.py --dtype bf16x3 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${PT}_bench_x3.json 2>> $ERR
python bench.py --steps 20 --warmup 5 > $OUT/${PT}_bench.json 2>> $ERR
ls -la $OUT | grep ${PT}_
tail -c 1500 $OUT/${PT}_bench.json
