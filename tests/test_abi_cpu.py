"""C-ABI checks that need no GPU: the library loads, exports every symbol include/plsvo_hip.h declares,
the ctypes mirrors have the C struct layouts, and without a device the product path fails loudly."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "plsvo_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(plsvo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(P):
    lib = P.capi.lib()
    declared = _declared_functions()
    assert len(declared) >= 30
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, f"libplsvo_hip.so lacks {missing}"
    assert sorted(P.capi.SYMBOLS) == declared, "capi.SYMBOLS out of sync with include/plsvo_hip.h"
    assert b"gfx950" in lib.plsvo_hip_version()


def test_shared_object_contains_gfx950_code(P):
    blob = open(P.capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"align_fused_kernel" in blob and b"pose_opt_kernel" in blob


def test_ctypes_structs_match_the_c_header(P, tmp_path):
    structs = {"plsvo_pinhole": P.abi.Pinhole, "plsvo_align_in": P.abi.AlignIn, "plsvo_align_out": P.abi.AlignOut,
               "plsvo_align_iterlog": P.abi.AlignIterLog, "plsvo_poseopt_in": P.abi.PoseOptIn,
               "plsvo_poseopt_out": P.abi.PoseOptOut, "plsvo_poseopt_iterlog": P.abi.PoseOptIterLog,
               "plsvo_structopt_in": P.abi.StructOptIn, "plsvo_structopt_out": P.abi.StructOptOut,
               "plsvo_match_in": P.abi.MatchIn, "plsvo_match_out": P.abi.MatchOut, "plsvo_reproject_in": P.abi.ReprojectIn,
               "plsvo_reproject_out": P.abi.ReprojectOut, "plsvo_seeds_in": P.abi.SeedsIn, "plsvo_seeds_out": P.abi.SeedsOut,
               "plsvo_chain_in": P.abi.ChainIn, "plsvo_chain_params": P.abi.ChainParams, "plsvo_chain_out": P.abi.ChainOut,
               "plsvo_pose_record": P.abi.PoseRecord}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, ct in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(l.split() for l in out.strip().splitlines())
    assert int(got["plsvo_pose_record"]) == 96 == P.abi.POSE_RECORD_DTYPE.itemsize   # the wire record of the pose all-gather
    for cname, ct in structs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"


def test_header_compiles_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(f'#include "{HEADER}"\nint main(void){{return PLSVO_MAX_LEVELS == 8 ? 0 : 1;}}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)
    src2 = tmp_path / "t.cpp"
    src2.write_text(f'#include "{HEADER}"\nint main(){{return sizeof(plsvo_align_in) > 0 ? 0 : 1;}}\n')
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-c", str(src2), "-o", str(tmp_path / "t2.o")], check=True)


def test_no_device_means_loud_failure_not_fallback(P):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on the CPU-only builder")
    with pytest.raises(P.capi.PlsvoError) as e:
        P.capi.Context(0)
    assert e.value.code == P.abi.E_NODEVICE and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    """the product package must not reach into oracle/ (a product path through the oracle voids parity)"""
    pkg = os.path.join(ROOT, "pl-svo_amd")
    offenders = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"plsvo_oracle|from oracle|import oracle|oracle/", txt):
                    offenders.append(os.path.join(dp, f))
    assert not offenders, offenders
    out = subprocess.run(["ldd", os.path.join(pkg, "libplsvo_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
