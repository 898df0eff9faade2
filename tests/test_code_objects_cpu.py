"""The built gfx950 code objects (gen_adversarial_amd/csrc/*.o, written by `make` / __graft_entry__.build()) hold no kernel that
spills registers or uses scratch memory: a dispatch that needs scratch costs far more than its own time (round 3: three kernels
spilling 2 - 16 registers cost 8 % of the headline step).  Reads the AMDGPU metadata notes with tools/code_object_notes.py."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_of_the_library_spills():
    if not glob.glob(os.path.join(ROOT, 'gen_adversarial_amd', 'csrc', '*.o')):
        pytest.skip('library objects not built here (run __graft_entry__.build())')
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-readelf'):
        pytest.skip('ROCm LLVM tools not installed')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'code_object_notes.py')], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith('0 kernels listed of '), r.stdout[-2000:]
    assert int(last.split()[-1]) >= 150, last          # every translation unit was read
