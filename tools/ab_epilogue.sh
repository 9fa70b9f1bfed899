#!/bin/bash
# Epilogue ablations of the lab build (YOLO_EPI_AB bits, conv_args.h) on representative layers: what each part of the conv epilogue
# costs inside the kernel.  bits: 1 no stores, 2 no residual loads, 4 no epilogue, 8 no scale/bias loads, 16 no LDS transpose.
export YOLO_AMD_LIB=$PWD/yolo_amd/csrc/_lab/libyolo_amd_lab.so
run() { for ab in ${ABS:-0 1 2 8 16 24 27 4}; do echo -n "AB=$ab $* : "; YOLO_EPI_AB=$ab python tools/algo_times.py "$@" --iters 100 2>/dev/null | grep -E "algo +($ALGOS) " | tr '\n' ';'; echo; done; }
ALGOS="6|8" run --n 32 --hw 26 --cin 256 --cout 512 --k 3 --res 1
ALGOS="6|8" run --n 32 --hw 52 --cin 128 --cout 256 --k 3 --res 1
ALGOS="8|4" run --n 32 --hw 104 --cin 64 --cout 128 --k 3 --res 1
ALGOS="2|6" run --n 64 --hw 38 --cin 256 --cout 512 --k 3 --res 1
ALGOS="2|8" run --n 64 --hw 76 --cin 128 --cout 256 --k 3 --res 1
ALGOS="8|4" run --n 32 --hw 26 --cin 512 --cout 256 --k 1 --res 0
