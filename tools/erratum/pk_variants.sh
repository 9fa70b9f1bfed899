#!/bin/bash
# Victim variants for tools/erratum/pk_bisect.py: train.hip built WITH the packed fp32 operations and ONE change to the
# set-up of bn_apply_kernel's per-channel factors k1 / k2 = (float)sum * inv_n, linked with the shipped objects into
# yolo_amd/csrc/_ab/libyolo_pk_<v>.so.   bash tools/erratum/pk_variants.sh   then   PK_LIB=.../libyolo_pk_v3.so python tools/erratum/pk_bisect.py
#   v3  the two multiplies kept scalar (values pinned in single registers)          -> measured CLEAN beside every co-runner
#   v4  conversions, 32 wait states, then the (packed) multiplies                    -> still corrupted
#   v5  the factor in a VGPR instead of the SGPR pair                                -> still corrupted
#   v6  k1 and k2 in loops of their own: the 8 packed multiplies stay, but hipcc no longer needs its 8 re-pairing moves
#       v_pk_mov_b32 v[n:n+1], v[n:n+1] op_sel:[1,0] behind them                         -> measured CLEAN
set -e
cd "$(dirname "$0")/../yolo_amd/csrc"
make -s pk >/dev/null 2>&1
python - <<'PY'
s = open('train.hip').read()
a = "                if (FUSED) { k1[e] = (float)f.sums[c] * inv_n; k2[e] = (float)f.sums[C + c] * inv_n; }"
assert s.count(a) == 1                                        # (the YOLO_BN_PAIRED_FACTORS form: the variants build with that macro)
v3 = s.replace(a, '''                if (FUSED) {
                    float t1 = (float)f.sums[c], t2 = (float)f.sums[C + c];
                    asm volatile("" : "+v"(t1)); asm volatile("" : "+v"(t2));
                    t1 = t1 * inv_n; t2 = t2 * inv_n;
                    asm volatile("" : "+v"(t1)); asm volatile("" : "+v"(t2));
                    k1[e] = t1; k2[e] = t2;
                }''')
b = "        auto apply = [&](const float (&v)[8], const float (&o)[8], float (&r)[8]) {"
assert s.count(b) == 1
v4 = s.replace(a, "                if (FUSED) { k1[e] = (float)f.sums[c]; k2[e] = (float)f.sums[C + c]; }").replace(b, '''        if (MODE == 1 && FUSED) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\\n\\ts_nop 15" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) { k1[e] *= inv_n; k2[e] *= inv_n; }
        }
''' + b)
v5 = s.replace(a, '                if (FUSED) { float iv = inv_n; asm volatile("" : "+v"(iv)); k1[e] = (float)f.sums[c] * iv; k2[e] = (float)f.sums[C + c] * iv; }')
v6 = s.replace(a, "                if (FUSED) { k1[e] = 0.f; k2[e] = 0.f; }").replace(b, '''        if (MODE == 1 && FUSED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) k1[e] = (float)f.sums[oct * 8 + e] * inv_n;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) k2[e] = (float)f.sums[C + oct * 8 + e] * inv_n;
        }
''' + b)
for n, t in (('v3', v3), ('v4', v4), ('v5', v5), ('v6', v6)):
    open('_ab_train_%s.hip' % n, 'w').write(t)
PY
OBJS="conv_igemm.o conv_pipe.o conv_pipe_b.o conv_sk.o conv_stream.o stem.o stem_down.o res_block.o elementwise.o detect.o wgrad_walk.o loss.o"
for v in v3 v4 v5 v6; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -DYOLO_BN_PAIRED_FACTORS -c _ab_train_$v.hip -o _ab/train_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libyolo_pk_$v.so $OBJS _ab/train_$v.o
    rm -f _ab_train_$v.hip
done
ls -la _ab/libyolo_pk_v*.so
