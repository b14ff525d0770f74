"""The C-ABI library builds for gfx950 here (no GPU), loads, and exports exactly what include/cseg_hip.h declares;
the ctypes binding lists every symbol; the product refuses CPU tensors instead of falling back. No compute calls."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from contrastiveseg_amd import _hip
    return _hip


def _declared():
    header = open(os.path.join(ROOT, "include", "cseg_hip.h")).read()
    return sorted(set(re.findall(r"\b(cseg_[a-z0-9_]+)\s*\(", header)))


def test_library_exports_every_declared_symbol(built):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH]).decode()
    exported = set(re.findall(r" T (cseg_[a-z0-9_]+)", out))
    assert set(_declared()) <= exported
    assert exported <= set(_declared()), "exported but undeclared: %s" % (exported - set(_declared()))


def test_binding_covers_every_symbol_and_loads(built):
    assert sorted(built.SIGNATURES) == _declared()
    lib = built.lib()
    assert lib.cseg_abi_version() == 6
    assert lib.cseg_contrast_ws_bytes(1024, 4096) == 1024 * 4096 * 4
    assert lib.cseg_contrast_ws_bytes(33, 33) == 64 * 64 * 4
    assert lib.cseg_upsample_ce_blocks(8, 512, 1024) == 8 * 64 * 32
    assert 1 <= lib.cseg_contrast_bwd_parts(1024, 1024, 256) <= 64


def test_code_object_is_gfx950_only(built):
    blob = open(built.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_cpu_fallback(built):
    from contrastiveseg_amd import kernels as K
    with pytest.raises(RuntimeError, match="GPU"):
        K.upsample_ce(torch.zeros(1, 3, 4, 4), torch.zeros(1, 8, 8, dtype=torch.long))
    with pytest.raises(RuntimeError, match="GPU"):
        K.upsample_concat([torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 2, 2)])
    with pytest.raises(RuntimeError, match="GPU"):
        K.classify_partition(torch.zeros(1, 8, 8, dtype=torch.long), -1, seg=torch.zeros(1, 3, 4, 4))


def test_product_does_not_import_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "contrastiveseg_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_the_torch_free_probe_compiles_against_the_header():
    """tools/probes/conv_probe.cpp drives the C-ABI directly (dlopen + HIP runtime, no Python): it must keep compiling against
    include/cseg_hip.h -- the calls are bound by decltype of the declarations, so a changed signature is a compile error here."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs g++ and the HIP runtime headers")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "probes", "conv_probe.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
