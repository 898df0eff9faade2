"""CPU tool: register / scratch / spill figures of every kernel in the gfx950 code objects of gen_adversarial_amd/csrc/*.o (the
AMDGPU metadata note the compiler writes).  python tools/code_object_notes.py [--all]   (default: only kernels that spill or
use scratch).  Run after `make -C gen_adversarial_amd/csrc`."""
import glob
import os
import subprocess
import sys
import tempfile

import yaml

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
show_all = '--all' in sys.argv
listed = total = 0
for obj in sorted(glob.glob(os.path.join(ROOT, 'gen_adversarial_amd', 'csrc', '*.o'))):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, 'fat.bin'), os.path.join(td, 'co.o')
        if subprocess.run([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', obj, os.path.join(td, 'x.o')],
                          capture_output=True).returncode != 0:
            continue                                             # host-only translation unit (plan.o)
        subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}',
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], check=True)
        note = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True, check=True).stdout
    body = note.split('---', 1)[1].rsplit('...', 1)[0]
    for k in yaml.safe_load(body)['amdhsa.kernels']:
        total += 1
        bad = k.get('.vgpr_spill_count', 0) or k.get('.private_segment_fixed_size', 0)    # SGPR spills go to VGPR lanes: no memory
        if show_all or bad:
            nm = subprocess.run(['c++filt', k['.name']], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(obj):16s} {nm[:120]:120s} vgpr {k['.vgpr_count']:3d} agpr {k.get('.agpr_count', 0):3d} "
                  f"spill v{k.get('.vgpr_spill_count', 0)} s{k.get('.sgpr_spill_count', 0)} scratch {k.get('.private_segment_fixed_size', 0)} B "
                  f"lds {k.get('.group_segment_fixed_size', 0)}")
            listed += 1
print(f'{listed} kernels listed of {total}')
