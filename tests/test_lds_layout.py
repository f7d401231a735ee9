"""Host-side invariants of the row-pair kernel's LDS layout (csrc/pm_kernel.h): the header is plain C++ on the host, so
the regions, their order and the residency classes DESIGN.md quotes are checked without a GPU (g++ only)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lds_layout_invariants(tmp_path):
    exe = str(tmp_path / 'lds_layout_check')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-o', exe, os.path.join(ROOT, 'tests', 'cpp', 'lds_layout_check.cpp')])
    p = subprocess.run([exe], stdout=subprocess.PIPE, universal_newlines=True)
    assert p.returncode == 0 and p.stdout.strip().endswith('0 violations'), p.stdout[-3000:]
