#!/bin/bash
# Per-kernel average durations of one command on the GPU box:  tools/kstats.sh <name-filter-regex> -- <command...>
# (rocprofv3 --kernel-trace --stats; prints name, calls, average ns for the kernels whose name matches the filter)
FILTER="$1"; shift; [ "$1" = "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$(mktemp -d /tmp/kstats.XXXX)
export TMPDIR=/tmp
(cd $ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- "$@" > $D/out.txt 2> $D/err.txt)
python - "$D" "$FILTER" <<'PY'
import csv, glob, re, sys
fs = glob.glob(sys.argv[1] + '/**/p_kernel_stats.csv', recursive=True)
if not fs:
    print(open(sys.argv[1] + '/err.txt').read()[-2000:]); sys.exit(1)
tot = 0.0
for r in csv.DictReader(open(fs[0])):
    if re.search(sys.argv[2], r['Name']):
        print('%-60s %6s %12.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])))
        tot += float(r['Calls']) * float(r['AverageNs'])
print('total matched ns', tot)
PY
