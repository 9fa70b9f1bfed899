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
