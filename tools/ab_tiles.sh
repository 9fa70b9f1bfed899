#!/bin/bash
# Per-layer times of the 3x3 tile variants incl. the pixel-heavy ones (algo 27: 384 x 128, 28: 512 x 128) on the D53 shapes.
run() { echo -n "$* : "; python tools/algo_times.py "$@" --iters 100 2>/dev/null | grep -E "algo +(2|6|8|26|27|28|4) " | tr '\n' ';'; echo; }
run --n 32 --hw 26 --cin 256 --cout 512 --k 3 --res 1
run --n 32 --hw 26 --cin 512 --cout 1024 --k 3 --res 0
run --n 32 --hw 52 --cin 128 --cout 256 --k 3 --res 1
run --n 32 --hw 52 --cin 256 --cout 512 --k 3 --res 0
run --n 32 --hw 13 --cin 512 --cout 1024 --k 3 --res 1
run --n 32 --hw 13 --cin 1024 --cout 2048 --k 3 --res 0
run --n 64 --hw 38 --cin 256 --cout 512 --k 3 --res 1
run --n 64 --hw 38 --cin 512 --cout 1024 --k 3 --res 0
run --n 64 --hw 76 --cin 128 --cout 256 --k 3 --res 1
run --n 64 --hw 76 --cin 256 --cout 512 --k 3 --res 0
run --n 64 --hw 19 --cin 512 --cout 1024 --k 3 --res 1
run --n 64 --hw 19 --cin 1024 --cout 2048 --k 3 --res 0
