"""The near-tie resolver of align_kernels.hip (`exact_chi2_pair`, `exact_chi2_pair_lds`: the reference's sequential float chi2 sums,
re-added on the device when two chi2 values come within their rounding noise) run on the CPU: the functions are cut out of the source
file, compiled with g++ against the lock-step wave emulator (tests/host/emu/) and compared BITWISE with plain sequential loops on
chi2-like and adversarial term vectors (tests/host/exact_sum_host_test.cpp).  The tree holds the slot-parallel form with DPP scans
(round 4: bit-exact on the MI355X in the parity suite, -2 ... -3 % per small-batch step against the one-lane chain it replaced)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN = "__device__ __forceinline__ float chain4"
END = "// ------------------------------------------------------------------------------------------------\n// SparseImgAlign::run for every job"


def test_exact_chi2_sums_are_the_sequential_float_sums(tmp_path):
    src = os.path.join(ROOT, "pl-svo_amd", "csrc", "align_kernels.hip")
    text = open(src).read()
    snippet = tmp_path / "snippet.inc"
    snippet.write_text(text[text.index(BEGIN):text.index(END)])
    assert "ds_bpermute" not in snippet.read_text() and "__shfl" not in snippet.read_text()      # scans and hand-overs are DPP / readlane
    exe = str(tmp_path / "exact_sum_host_test")
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", '-DSNIPPET="%s"' % snippet,
                    "-I", os.path.join(ROOT, "tests", "host", "emu"), "-I", os.path.join(ROOT, "pl-svo_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "exact_sum_host_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe, "240"], capture_output=True, text=True)
    cases, bad = (int(t) for t in out.stdout.split())
    assert out.returncode == 0 and cases == 240 and bad == 0, out.stdout + out.stderr
