"""No single-precision arithmetic of the hot kernels may be fused behind the source's back.

The float image arithmetic of the alignment (bilinear samples, residuals, chi2 terms) and the float error norms / Tukey weights of the
pose optimiser are bit-for-bit the reference's only if every product and sum is rounded on its own; hipcc contracts a * b + c into an fma
by default, and HIP's __fmul_rn / __fadd_rn do not prevent it (they are plain `*` / `+`).  Twice a contracted expression slipped through
review and was only found by a seed that happened to sit on the rounding boundary (the bilinear sample in round 2, the line-error norm
of the pose optimiser's scale pass in round 3).  This test makes the compiler say it: the gfx950 code of each translation unit must
contain exactly as many single-precision fused instructions when built normally as when built with -ffp-contract=off -- the ones that
remain are the fmas the source asks for by name (robust_weight's error-free sequence) and the expansions of correctly rounded float
division / sqrt.  Double-precision contraction is allowed: the reference's double arithmetic is matched to tolerance, not bit for bit.
hipcc cross-compiles without a GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pl-svo_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FUSED_F32 = re.compile(r"\b(v_fma_f32|v_fmac_f32|v_mad_f32|v_mac_f32|v_pk_fma_f32|v_fma_mix_f32|v_fma_legacy_f32)")


def _fused_f32(unit, tmp_path, extra):
    out = tmp_path / (unit + ("_off" if extra else "_on") + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", *extra, os.path.join(CSRC, unit + ".hip"), "-o", str(out)],
                   check=True, capture_output=True)
    return sum(1 for line in open(out) if FUSED_F32.search(line))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("unit", ["align_kernels", "poseopt_kernels", "chain_kernels"])
def test_no_float_contraction_in_the_hot_kernels(unit, tmp_path):
    default = _fused_f32(unit, tmp_path, [])
    uncontracted = _fused_f32(unit, tmp_path, ["-ffp-contract=off"])
    assert default == uncontracted, (f"{unit}.hip: {default} single-precision fused instructions in the default build, {uncontracted} with "
                                     "-ffp-contract=off: some float expression is being contracted (wrap it in `#pragma clang fp contract(off)`)")
