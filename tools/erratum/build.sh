#!/bin/bash
# Builds the stand-alone reproducer of the packed-fp32 finding into tools/_build/ (hipcc cross-compiles; not part of the product
# build -- __graft_entry__.build() compiles the library only).   bash tools/erratum/build.sh
set -e
cd "$(dirname "$0")/../.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p tools/_build
$HIPCC --offload-arch=gfx950 -O3 tools/erratum/pk_min.hip -o tools/_build/pk_min
