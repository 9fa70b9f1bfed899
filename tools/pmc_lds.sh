#!/bin/bash
# LDS bank-conflict share per kernel of a command: tools/pmc_lds.sh <command ...>   (one rocprofv3 --pmc pass)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
rm -rf /tmp/pl; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pl -- "$@" > /dev/null 2>&1
python - "$(find /tmp/pl -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:100]][r['Counter_Name']] += float(r['Counter_Value'])
rows = sorted(acc.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])
tot = sum(v['SQ_WAVE_CYCLES'] for _, v in rows)
print('%-100s %7s %9s %9s %9s' % ('kernel', 'share', 'lds_act', 'conflict', 'confl/act'))
for k, v in rows[:30]:
    wc = v['SQ_WAVE_CYCLES']
    a, c = v['SQ_LDS_IDX_ACTIVE'] / wc, v['SQ_LDS_BANK_CONFLICT'] / wc
    print('%-100s %6.1f%% %9.3f %9.3f %9.2f' % (k, 100 * wc / tot, a, c, c / a if a else 0))
PY
