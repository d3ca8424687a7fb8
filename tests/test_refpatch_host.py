"""The record form of the reference-patch cache (align_kernels.hip, PLSVO_BYTE_CACHE A/B build) rebuilds, bit for bit, the
interpolated intensity and gradient rows the precompute writes: pl-svo_amd/csrc/align_refpatch.hpp -- the very source the kernel
includes -- compiled for the host (tests/host/refpatch_host_test.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_record_form_rebuilds_the_cached_rows_bitwise(tmp_path):
    exe = str(tmp_path / "refpatch_host_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "pl-svo_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "refpatch_host_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe, "30000"], capture_output=True, text=True)
    cases, bad = (int(t) for t in out.stdout.split())
    assert out.returncode == 0 and cases == 30000 and bad == 0, out.stdout
