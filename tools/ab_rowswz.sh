#!/bin/bash
# Same-box A/B of the 3x3 halo layout (lab build): YOLO_NO_ROW_SWZ=1 = pitch TWt + 2 with the slot swizzle (round 4), 0 = pitch
# TWt + 4 with the row-relative swizzle (conflict-free input fragments).  Whole passes on the committed plan, then per layer.
export YOLO_AMD_LIB=$PWD/yolo_amd/csrc/_lab/libyolo_amd_lab.so
LEAN="--no-cpu-baseline --no-northstar --no-train-key --no-f32-key --no-repeats --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for off in 1 0; do
  echo -n "416 bs32 NO_ROW_SWZ=$off: "; YOLO_NO_ROW_SWZ=$off python bench.py --steps 100 --warmup 10 $LEAN 2>/dev/null | val
  echo -n "608 bs64 NO_ROW_SWZ=$off: "; YOLO_NO_ROW_SWZ=$off python bench.py --steps 50 --warmup 5 --size 608 --batch 64 $LEAN 2>/dev/null | val
done; done
run() { for off in 1 0; do echo -n "NO_ROW_SWZ=$off $* : "; YOLO_NO_ROW_SWZ=$off python tools/algo_times.py "$@" --iters 100 2>/dev/null | grep -E "algo +(2|6|8|27|28|4|3) " | tr '\n' ';'; echo; done; }
run --n 32 --hw 52 --cin 128 --cout 256 --k 3 --res 1
run --n 32 --hw 26 --cin 256 --cout 512 --k 3 --res 1
run --n 32 --hw 13 --cin 512 --cout 1024 --k 3 --res 1
run --n 32 --hw 13 --cin 1024 --cout 2048 --k 3 --res 0
run --n 64 --hw 76 --cin 128 --cout 256 --k 3 --res 1
run --n 64 --hw 38 --cin 256 --cout 512 --k 3 --res 1
run --n 64 --hw 19 --cin 512 --cout 1024 --k 3 --res 1
run --n 64 --hw 76 --cin 256 --cout 512 --k 3 --res 0
