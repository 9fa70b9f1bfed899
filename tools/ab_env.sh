#!/bin/bash
# same-box A/B of the training step under environment knobs: ab_env.sh "VAR=1" "VAR2=x" ...   ('' = defaults)
cd "$(dirname "$0")/.."
for rep in 1 2; do for cfg in "$@"; do
  ms=$(env $cfg python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "[$cfg] $ms ms"
done; done
