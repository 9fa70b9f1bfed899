#!/bin/bash
# Build (anywhere: hipcc cross-compiles) and, on a GPU box, run the stand-alone packed-fp32 probe (tools/erratum/pk_repro.hip).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
[ -x tools/_build/pk_repro ] && [ tools/_build/pk_repro -nt tools/erratum/pk_repro.hip ] || \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/erratum/pk_repro.hip -o tools/_build/pk_repro
if [ "$1" != "--build-only" ]; then tools/_build/pk_repro "${1:-200}"; fi
