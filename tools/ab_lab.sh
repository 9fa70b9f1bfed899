#!/bin/bash
# Same-box A/B of library builds / lab knobs:
#   tools/ab_lab.sh "<bench.py arguments>" "" "YOLO_EPI_X=1" "YOLO_AMD_LIB=yolo_amd/csrc/_ab/libyolo_base.so" ...
# '' = the shipped library, no knobs; a configuration that names no YOLO_AMD_LIB runs the lab build (`make -C yolo_amd/csrc lab`)
# with the knobs given.  Each configuration runs `bench.py <arguments>` REPS times (default 2), alternating; prints value /
# ms_per_step per run.
cd "$(dirname "$0")/.."
args="$1"; shift
for rep in $(seq 1 ${REPS:-2}); do for cfg in "$@"; do
  case "$cfg" in
    "") pre="";;
    *YOLO_AMD_LIB=*) pre="env $cfg";;
    *) pre="env YOLO_AMD_LIB=$PWD/yolo_amd/csrc/_lab/libyolo_amd_lab.so $cfg";;
  esac
  line=$($pre python bench.py $args 2>/dev/null | tail -1)
  echo "[${cfg:-shipped}] $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d.get('value_repeats'))")"
done; done
