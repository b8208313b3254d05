"""Build-time checks of the generated gfx950 code (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _device_asm(src, out):
    csrc = os.path.join(ROOT, 'dhd_amd', 'csrc')
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
           '-munsafe-fp-atomics', '-Wno-unused-function', '--cuda-device-only', '-S', os.path.join(csrc, src), '-o', str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return str(out)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_no_store_data_overwrite_hazard_in_any_translation_unit(tmp_path):
    """A MUBUF store of more than 8 bytes whose write-data VGPRs the next VALU instruction overwrites sends the NEW values on
    gfx950 (round 5: 1.6 % of pw_gemm_cuh_kernel<_Float16>'s 16-byte stores carried four bytes of the next channel row; hipcc
    assumes the hazard away when the store's soffset is an SGPR).  The GEMM kernels store through store_b128_guarded
    (csrc/sfa_mfma.h).  EVERY .hip file the Makefile builds is compiled to gfx950 assembly with the Makefile's flags and scanned
    for the pattern (round 5 scanned sfa_stage.hip only: a buffer store added to any other file would have gone unnoticed)."""
    from concurrent.futures import ThreadPoolExecutor
    mk = open(os.path.join(ROOT, 'dhd_amd', 'csrc', 'Makefile')).read()
    srcs = next(ln for ln in mk.splitlines() if ln.startswith('SRCS')).split(':=')[1].split()
    on_disk = sorted(f for f in os.listdir(os.path.join(ROOT, 'dhd_amd', 'csrc')) if f.endswith('.hip'))
    assert sorted(srcs) == on_disk               # the Makefile builds every .hip file that is there
    with ThreadPoolExecutor(max_workers=4) as pool:
        asms = list(pool.map(lambda f: _device_asm(f, tmp_path / (f[:-4] + '.s')), srcs))
    text = open(os.path.join(tmp_path, 'sfa_stage.s')).read()
    assert text.count('buffer_store_dwordx4') > 100          # the scan has something to look at
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'experiments', 'lint_store_hazard.py')] + asms, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith('hazards: 0'), out.stdout[-2000:]
    n_stores = {os.path.basename(a): open(a).read().count('buffer_store_dwordx') for a in asms}
    print('buffer_store_dwordx[2-4] per translation unit:', {k: v for k, v in n_stores.items() if v})
