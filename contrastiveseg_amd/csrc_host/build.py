"""Builds the host-side native code with plain g++ against the installed libtorch: contrastiveseg_amd/libcseg_host.so (random draws,
ctypes). (Round 4-5 also built _cseg_native.so, a pybind11 executor for one residual block; the grouped launches of round 6 run a
whole depth of blocks from one node and superseded it -- measured no gain at batch 8, git history has it.)"""
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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
