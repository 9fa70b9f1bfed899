#!/bin/bash
# Lab probe YOLO_EPI_AB=256 (WRONG results, timing only): the input fragments of the 3x3 K loop read from consecutive halo slots =
# no LDS bank conflicts.  What the conflicts of the row-crossing fragments cost inside the kernel.
export YOLO_AMD_LIB=$PWD/yolo_amd/csrc/_lab/libyolo_amd_lab.so
run() { for ab in 0 256 0 256; do echo -n "AB=$ab $* : "; YOLO_EPI_AB=$ab python tools/algo_times.py "$@" --iters 100 2>/dev/null | grep -E "algo +($ALGOS) " | tr '\n' ';'; echo; done; }
ALGOS="6|2|27|28" run --n 32 --hw 13 --cin 1024 --cout 2048 --k 3 --res 0
ALGOS="6|2|27|28" run --n 32 --hw 26 --cin 256 --cout 512 --k 3 --res 1
ALGOS="6|2|27|28" run --n 32 --hw 26 --cin 512 --cout 1024 --k 3 --res 0
ALGOS="6|2|27|28" run --n 32 --hw 52 --cin 256 --cout 512 --k 3 --res 0
ALGOS="6|2|27|28" run --n 64 --hw 38 --cin 512 --cout 1024 --k 3 --res 0
ALGOS="6|2|8|27" run --n 64 --hw 76 --cin 128 --cout 256 --k 3 --res 1
