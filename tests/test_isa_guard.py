"""The device code must not contain packed-FP32 instructions that read across halves through `op_sel:`.

On gfx950 `v_pk_fma_f32 ... op_sel:[0,1,0]` - the form hipcc's SLP vectoriser produced in the template sampler - loses the
product term in lanes 48..63 while another wavefront of the SIMD issues MFMAs (tools/ubench/pk_waw.hip, measured in
profiles/r02_packed_fp32_op_sel_hazard.txt).  The build therefore switches the SLP vectoriser off and this test greps the
ISA of every BUILT object (the gfx950 code objects are disassembled: what ships; `make isa-check-src` recompiles every source to
assembly instead).  Builds on the CPU box (no GPU needed), seconds once the library is built."""
import os
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'sea_ice_drift_amd', 'csrc')


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='hipcc not installed')
def test_no_half_swapping_packed_fp32_in_the_isa():
    res = subprocess.run(['make', '-s', '-C', CSRC, 'isa-check'], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr


def test_the_guard_pattern_catches_the_bad_form():
    import re
    pat = re.compile(r'v_pk_[a-z0-9_]*_f32 .*op_sel:\[[01,]*1')
    assert pat.search('\tv_pk_fma_f32 v[6:7], v[18:19], v[8:9], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]')
    assert not pat.search('\tv_pk_fma_f32 v[6:7], v[18:19], v[8:9], v[6:7] op_sel_hi:[0,1,1]')
    assert not pat.search('\tv_pk_fma_f32 v[6:7], v[18:19], v[8:9], v[6:7]')
    assert not pat.search('\tv_pk_add_f32 v[82:83], v[8:9], 0 neg_lo:[1,1] neg_hi:[1,1]')
