#!/bin/bash
# Same-box A/B (lab build, YOLO_NO_ROW_SWZ=1 = the round-4 halo layout) of the stride-2 / 2x2-window kernels and the training step.
export YOLO_AMD_LIB=$PWD/yolo_amd/csrc/_lab/libyolo_amd_lab.so
run() { for off in 1 0 1 0; do echo -n "NO_ROW_SWZ=$off $* : "; YOLO_NO_ROW_SWZ=$off python tools/algo_times.py "$@" --iters 100 2>/dev/null | head -3 | tr '\n' ';'; echo; done; }
run --n 32 --hw 26 --cin 512 --cout 1024 --k 3 --s 2
run --n 32 --hw 52 --cin 256 --cout 512 --k 3 --s 2
run --n 32 --hw 104 --cin 128 --cout 256 --k 3 --s 2
run --n 64 --hw 38 --cin 512 --cout 1024 --k 3 --s 2
run --n 64 --hw 76 --cin 256 --cout 512 --k 3 --s 2
for off in 1 0 1 0; do echo -n "train NO_ROW_SWZ=$off: "; YOLO_NO_ROW_SWZ=$off python bench.py --mode train --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
