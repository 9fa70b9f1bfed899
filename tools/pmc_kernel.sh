#!/bin/bash
# Wave-cycle split of one kernel family: tools/pmc_kernel.sh <kernel-name regex> <command ...>   (four rocprofv3 --pmc passes)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
pat="$1"; shift
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
rm -rf /tmp/pw; rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pw -- "$@" > /dev/null 2>&1
f=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
python - "$f" "$pat" <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if not re.search(sys.argv[2], r['Kernel_Name']): continue
    key = r['Kernel_Name'][:48]
    acc[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[key] += 1
for k, v in acc.items():
    wc = v['SQ_WAVE_CYCLES']
    print(k, 'launches', n[k], {c: round(x / wc, 4) for c, x in v.items() if c != 'SQ_WAVE_CYCLES'}, 'wave_cycles/launch %.4g' % (wc / max(n[k], 1)))
PY
done
