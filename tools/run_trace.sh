set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
python bench.py --mode train --steps 4 --warmup 2 --tune-cache $OUT/tc_t.json > $OUT/tt0.json 2> $OUT/tt.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/tt_prof -o p -- python bench.py --mode train --steps 4 --warmup 2 --tune-cache $OUT/tc_t.json > $OUT/tt1.json 2>> $OUT/tt.err
python tools/train_trace.py $OUT/tt_prof $OUT/train_trace.txt
rm -rf $OUT/tt_prof
tail -3 $OUT/train_trace.txt; cat $OUT/tt0.json | cut -c1-300
