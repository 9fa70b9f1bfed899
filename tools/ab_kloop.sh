#!/bin/bash
# K-loop probes of the lab build (YOLO_EPI_AB bits, conv_args.h; WRONG results, timing only): what a part of the loop costs inside the
# kernel.  bits: 32 every second barrier dropped, 64 no weight DMAs after the first ring, 128 no input DMAs after the first two chunks.
export YOLO_AMD_LIB=$PWD/yolo_amd/csrc/_lab/libyolo_amd_lab.so
run() { for ab in 0 64 128 192; do echo -n "AB=$ab $* : "; YOLO_EPI_AB=$ab python tools/algo_times.py "$@" --iters 100 2>/dev/null | grep -E "algo +($ALGOS) " | tr '\n' ';'; echo; done; }
ALGOS="6|2|8" run --n 32 --hw 26 --cin 256 --cout 512 --k 3 --res 1
ALGOS="6|2|8" run --n 32 --hw 26 --cin 512 --cout 1024 --k 3 --res 0
ALGOS="2|6|26" run --n 64 --hw 38 --cin 512 --cout 1024 --k 3 --res 0
ALGOS="2|6|8" run --n 64 --hw 76 --cin 128 --cout 256 --k 3 --res 1
