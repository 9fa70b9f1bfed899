#!/bin/bash
# Builds the stand-alone probes of the packed-fp32 finding into tools/_build/ (hipcc cross-compiles; nothing here is part of the
# product build -- __graft_entry__.build() compiles the library only).   bash tools/erratum/build.sh
set -e
cd "$(dirname "$0")/../.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p tools/_build
$HIPCC --offload-arch=gfx950 -O3 -ffp-contract=off tools/erratum/pk_repro.hip -o tools/_build/pk_repro
$HIPCC --offload-arch=gfx950 -O3 -shared -fPIC tools/erratum/pk_spin.hip -o tools/_build/libpk_spin.so
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DYOLO_BN_PAIRED_FACTORS -Wno-unused-function \
    -I yolo_amd/csrc -I include tools/erratum/pk_repro2.hip -o tools/_build/pk_repro2
$HIPCC --offload-arch=gfx950 -O3 tools/erratum/pk_min.hip -o tools/_build/pk_min
