"""Builds the host-side native code with plain g++ against the installed libtorch: contrastiveseg_amd/libcseg_host.so (random draws,
ctypes) and contrastiveseg_amd/_cseg_native.so (block_exec.cpp: the residual-block executor, a pybind11 extension module)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libcseg_host.so")
SRC = os.path.join(HERE, "rng_draws.cpp")


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    import torch
    from torch.utils.cpp_extension import include_paths
    lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", OUT,
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for inc in include_paths():
        cmd += ["-I", inc]
    cmd += ["-L", lib_dir, "-Wl,-rpath," + lib_dir, "-ltorch_cpu", "-lc10"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


NATIVE_SRC = os.path.join(HERE, "block_exec.cpp")
NATIVE_OUT = os.path.join(PKG, "_cseg_native.so")


def build_native(force=False, verbose=False):
    if not force and os.path.exists(NATIVE_OUT) and os.path.getmtime(NATIVE_OUT) >= os.path.getmtime(NATIVE_SRC):
        return NATIVE_OUT
    import sysconfig
    import torch
    from torch.utils.cpp_extension import include_paths
    lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", NATIVE_SRC, "-o", NATIVE_OUT,
           "-DTORCH_EXTENSION_NAME=_cseg_native", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-I", sysconfig.get_paths()["include"],
           "-Wno-attributes"]                       # (pybind11: the copy bundled with torch, whose tensor casters these are)
    for inc in include_paths():
        cmd += ["-I", inc]
    cmd += ["-L", lib_dir, "-Wl,-rpath," + lib_dir, "-ltorch_python", "-ltorch", "-ltorch_cpu", "-lc10"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return NATIVE_OUT


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    print(build(force="--force" in sys.argv, verbose=True))
