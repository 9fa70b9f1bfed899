q() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['value'], d['northstar_608']['value'], d['northstar_608']['frac_of_peak'])"; }
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | q base
python bench.py --no-cpu-baseline --no-roofline --no-fuse-concat 2>/dev/null | q nocat
done
