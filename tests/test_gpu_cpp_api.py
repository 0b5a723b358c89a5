"""GPU tests of the drop-in C++ surface (gl-radix-sort_amd/glu/*.hpp over the C ABI): runs the ported reference
test programs in tests/cpp (built by __graft_entry__.build())."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("prog", ["test_radix_sort_api", "test_scan_api", "test_reduce_api", "test_dist_api"])
def test_cpp_program(built, prog):
    exe = os.path.join(ROOT, "tests", "cpp", "bin", prog)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(p.stdout[-3000:])
    print(p.stderr[-2000:])
    assert p.returncode == 0
    assert "0 failure(s)" in p.stdout


def test_cpp_benchmark_program_runs(built):
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "bench_ladder")
    p = subprocess.run([exe, "all", "1048576"], capture_output=True, text=True, timeout=600)
    print(p.stdout)
    assert p.returncode == 0
    assert "Radix sort; Num elements: 1048576, Elapsed:" in p.stdout  # radix_sort_tests.cpp:192 line format
    assert "BlellochScan; Num elements: 1048576" in p.stdout and "Reduce; Num elements: 1048576" in p.stdout
