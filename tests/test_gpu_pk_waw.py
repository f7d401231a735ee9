"""DESIGN.md section 6b as a regression test: the packed-FP32 forms the kernels rely on are re-validated on whatever board
and ROCm the suite runs on.  tools/ubench/pk_waw.hip (built here with hipcc) runs the compiled stretch of the template
sampler next to other wavefronts' MFMAs and counts wrong results per quarter of the wavefront:

* the forms the library uses - no packed instruction, plain v_pk_fma_f32, the op_sel_hi:[0,1,1] broadcast form, two scalar
  FMAs - must be clean (the 64 hand-written v_pk_fma_f32 of the scoring depend on it);
* the half-swapping op_sel form that the ISA guard (tests/test_isa_guard.py) keeps out of the build is reported: if a
  future board no longer shows the hazard the guard could be relaxed, if a "clean" form starts failing the test says so.
"""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_fp32_forms_used_by_the_kernels_are_clean(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available on this box')
    exe = str(tmp_path / 'pk_waw')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-o', exe, os.path.join(ROOT, 'tools', 'ubench', 'pk_waw.hip')],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = subprocess.run([exe, '4000'], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                         universal_newlines=True).stdout
    blocks = re.findall(r'\[(.*?) \| (.*?)\] wrong results.*?\n\s+copy v82 = v8: ([\d ]+?)\s+low half of the packed FMAs: ([\d ]+?)\s+high half: ([\d ]+?)\s+\(',
                        out)
    assert len(blocks) >= 8, out[-2000:]
    seen_clean = 0
    for company, variant, cp, lo, hi in blocks:
        wrong = sum(int(x) for x in (cp + ' ' + lo + ' ' + hi).split())
        risky = variant.startswith('as emitted') or 'idle cycles' in variant            # the half-swapping op_sel form
        if not risky:
            assert wrong == 0, 'a packed-FP32 form the kernels rely on failed: [%s | %s] %s / %s / %s' % (company, variant, cp, lo, hi)
            seen_clean += 1
        elif wrong:
            print('hazard reproduced on this board: [%s | %s] low half %s' % (company, variant, lo))
    assert seen_clean >= 4
