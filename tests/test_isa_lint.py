"""The shipped library must not contain the instruction forms DESIGN 4.2 found unsafe on MI355X: VOP3P packed-fp32 arithmetic
(v_pk_mul/add/fma_f32, v_pk_mov_b32) and, more generally, any instruction with op_sel / op_sel_hi operand selection -- the
cross-half packed forms read +0 in lanes 48-63 beside a wave that interleaves VALU work with its MFMAs (tools/erratum/pk_min.hip).
csrc/Makefile's -packed-fp32-ops keeps hipcc from emitting them; this test disassembles what was actually built."""
import os
import re
import shutil
import struct
import subprocess

import pytest

from yolo_amd import lib as L

LLVM = '/opt/rocm/lib/llvm/bin'


def _device_code_objects(so_path, tmp):
    fat = os.path.join(tmp, 'fat.bin')
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', so_path, fat])
    d = open(fat, 'rb').read()
    out = []
    for k, m in enumerate(re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', d)):
        o = m.start()
        nb = struct.unpack_from('<Q', d, o + 24)[0]
        p = o + 32
        for _ in range(nb):
            off, size, ts = struct.unpack_from('<QQQ', d, p)
            p += 24
            triple = d[p:p + ts].decode()
            p += ts
            if 'gfx950' in triple and size:
                path = os.path.join(tmp, 'dev_%d.co' % k)
                open(path, 'wb').write(d[o + off:o + off + size])
                out.append(path)
    return out


def test_no_packed_fp32_and_no_op_sel_in_the_shipped_library(tmp_path):
    if not os.path.exists(os.path.join(LLVM, 'llvm-objdump')) or not shutil.which('make'):
        pytest.skip('no ROCm LLVM tools here')
    L.build()
    so = os.path.join(L.CSRC, 'libyolo_amd.so')
    cos = _device_code_objects(so, str(tmp_path))
    assert len(cos) >= 12, 'expected one gfx950 code object per translation unit, found %d' % len(cos)
    bad, kernels = [], 0
    for co in cos:
        dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], capture_output=True, text=True, check=True).stdout
        kernels += len(re.findall(r'^[0-9a-f]+ <[^>]+>:', dis, re.M))
        for line in dis.split('\n'):
            if re.search(r'\bv_pk_(mul|add|fma)_f32\b|\bv_pk_mov_b32\b|\bop_sel', line):
                bad.append(line.strip())
    assert kernels > 100, 'the disassembly looks empty (%d symbols)' % kernels
    assert not bad, 'unsafe instruction forms in the shipped library (DESIGN 4.2): %s' % bad[:5]


# The kernels the committed launch plan spends its time in: no scratch.  Several of them keep indexed register arrays (acc[][],
# xo[], yb[][], the weight-gradient walk's carried fragments) out of scratch only because every loop around them is FULLY unrolled;
# csrc/Makefile silences the compiler's "loop not unrolled" diagnostic (-Wno-pass-failed), so a future unroll failure would show as
# a silent slowdown -- here it shows as a failed test.
HOT_KERNELS = [
    r'conv_pipe_kernel<bf16_t, 3, 4, 2, 2, 3, 768, 1, 0, 1, 0, 0>',       # 384 x 128 tile: the 416x416 headline's dominant kernel
    r'conv_pipe_kernel<bf16_t, 3, 2, 4, 2, 4, 512, 1, 0, 1, 0, 0>',       # 256 x 256 tile: the 608x608 pass's dominant kernel
    r'conv_pipe_kernel<bf16_t, 3, 2, 2, 2, 3, 384, 1, 0, 1, 0, 0>',
    r'conv_pipe_kernel<bf16_t, 3, 2, 4, 1, 3, 384, 1, 0, 1, 0, 0>',
    r'conv_pipe_kernel<bf16_t, 3, 2, 4, 2, 2, 768, 2, 0, 1, 0, 0>',
    r'conv_pipe_kernel<bf16_t, 1, 2, 2, 2, 3, 192, 1, 3, 1, 1, 0>',
    r'conv_pipe_kernel<bf16_t, 1, 2, 2, 2, 3, 192, 1, 0, 1, 1, 0>',
    r'conv_pipe_kernel<bf16x3_t, 3, 4, 2, 2, 3, 768, 1, 0, 1, 0, 0>',     # the split path's dominant kernels
    r'conv_pipe_kernel<bf16x3_t, 3, 2, 2, 2, 3, 384, 1, 0, 1, 0, 0>',
    r'conv_pipe_kernel<bf16x3_t, 1, 2, 2, 2, 3, 192, 1, 0, 1, 1, 0>',
    r'stem_split_kernel<bf16x3_t>',
    r'wgrad_walk_kernel<16, 4, 1>', r'wgrad_walk_kernel<4, 4, 2>', r'wgrad_walk_kernel<4, 4, 1>', r'wgrad_gemm_kernel<2, 2, 3>',
    r'res_block_kernel<64', r'res_block2_kernel<128', r'stem_down_kernel<bf16_t>', r'bn_reduce_kernel<bf16_t, 1>', r'bn_apply_kernel<bf16_t, 1, 1>',
]


def test_hot_kernels_have_no_scratch(tmp_path):
    if not os.path.exists(os.path.join(LLVM, 'llvm-readelf')) or not shutil.which('make') or not shutil.which('c++filt'):
        pytest.skip('no ROCm LLVM tools here')
    L.build()
    so = os.path.join(L.CSRC, 'libyolo_amd.so')
    found = {}
    for co in _device_code_objects(so, str(tmp_path)):
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=True).stdout
        blocks = notes.split('- .agpr_count:')[1:]
        names = [dict(re.findall(r'\.(\w+):\s+(\S+)', '.agpr_count:' + b.split('\n    - .a')[0])) for b in blocks]
        dem = subprocess.run(['c++filt'], input='\n'.join(f.get('name', '?') for f in names), capture_output=True, text=True).stdout.split('\n')
        for f, d in zip(names, dem):
            for pat in HOT_KERNELS:
                if pat in d:
                    found.setdefault(pat, []).append((d, int(f.get('private_segment_fixed_size', -1)), int(f.get('vgpr_count', -1))))
    missing = [p for p in HOT_KERNELS if p not in found]
    assert not missing, 'kernels the plan launches are not in the library (renamed?): %s' % missing
    spilled = [(d, s) for v in found.values() for d, s, _ in v if s != 0]
    assert not spilled, 'hot kernels with scratch (an unroll failure or a register cliff): %s' % spilled
