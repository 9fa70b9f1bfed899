cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
rm -rf /tmp/pw; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pw -- python tools/train_ops_bench.py --batch 64 --what wgrad --algos 3 --k3s1 --iters 5 > /dev/null 2>&1
f=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad_walk' not in r['Kernel_Name']: continue
    key = r['Grid_Size']
    acc[key][r['Counter_Name']] += float(r['Counter_Value']); 
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[key] += 1
for k, v in acc.items():
    m = n[k]
    wc = v['SQ_WAVE_CYCLES'] / m
    print('grid', k, 'launches', m, 'mfma_busy/wave_cycles*4waves?', round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / v['SQ_WAVE_CYCLES'], 3),
          'wait_any', round(v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], 3), 'wait_inst', round(v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'], 3),
          'active', round(v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES'], 3))
PY
