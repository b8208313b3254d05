"""Build-time checks of the generated gfx950 code (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_no_store_data_overwrite_hazard_in_the_gemm_kernels(tmp_path):
    """A MUBUF store of more than 8 bytes whose write-data VGPRs the next VALU instruction overwrites sends the NEW values on
    gfx950 (round 5: 1.6 % of pw_gemm_cuh_kernel<_Float16>'s 16-byte stores carried four bytes of the next channel row; hipcc
    assumes the hazard away when the store's soffset is an SGPR).  The GEMM kernels store through store_b128_guarded
    (csrc/sfa_mfma.h); this compiles the one translation unit that uses buffer stores and scans its assembly for the pattern."""
    asm = tmp_path / 'sfa_stage.s'
    csrc = os.path.join(ROOT, 'dhd_amd', 'csrc')
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
           '-munsafe-fp-atomics', '-Wno-unused-function', '--cuda-device-only', '-S', os.path.join(csrc, 'sfa_stage.hip'), '-o', str(asm)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = asm.read_text()
    assert text.count('buffer_store_dwordx4') > 100          # the scan has something to look at
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'experiments', 'lint_store_hazard.py'), str(asm)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith('hazards: 0'), out.stdout[-2000:]
