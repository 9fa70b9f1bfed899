"""Register / scratch / LDS figures of the kernels in the built library (from the code objects' metadata notes).
    python tools/kernel_regs.py [name regex] [library]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_isa_lint import _device_code_objects, LLVM

pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else '.')
so = sys.argv[2] if len(sys.argv) > 2 else 'yolo_amd/csrc/libyolo_amd.so'
with tempfile.TemporaryDirectory() as tmp:
    for co in _device_code_objects(so, tmp):
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True).stdout
        for blk in notes.split('- .agpr_count:')[1:]:
            f = dict(re.findall(r'\.(\w+):\s+(\S+)', '.agpr_count:' + blk.split('\n    - .a')[0]))
            name = f.get('name', '?')
            dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            if pat.search(dem):
                print('%-110s vgpr %3s agpr %3s scratch %4s lds %6s' % (dem[:110], f.get('vgpr_count'), f.get('agpr_count'), f.get('private_segment_fixed_size'), f.get('group_segment_fixed_size')))
