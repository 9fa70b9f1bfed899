#!/bin/bash
# Trigger variants for tools/erratum/pk_bisect.py: the generic convolution kernel -- the simplest co-runner beside which the
# packed BatchNorm backward is corrupted -- built with ONE part cut out (-DYOLO_TRIG=n, the cut points in conv_igemm.hip), linked
# with the shipped objects into yolo_amd/csrc/_ab/libyolo_trig_<n>.so.
#   bash tools/erratum/pk_trigger.sh   then   TRIG_LIB=.../libyolo_trig_1.so PK_ONLY=generic python tools/erratum/pk_bisect.py 20
set -e
cd "$(dirname "$0")/../yolo_amd/csrc"
make -s pk >/dev/null 2>&1
OBJS="conv_pipe.o conv_pipe_b.o conv_sk.o conv_stream.o stem.o stem_down.o res_block.o elementwise.o detect.o wgrad_walk.o loss.o train.o"
for n in ${TRIG_VARIANTS:-1 2 3 4 5 6}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops \
        -Wno-unused-command-line-argument -DYOLO_TRIG=$n -c conv_igemm.hip -o _ab/conv_igemm_trig_$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libyolo_trig_$n.so $OBJS _ab/conv_igemm_trig_$n.o
done
ls -la _ab/libyolo_trig_*.so
