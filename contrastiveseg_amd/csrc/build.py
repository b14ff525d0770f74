"""Builds contrastiveseg_amd/libcseg_hip.so (gfx950) from the .hip sources in this directory with hipcc.
In-tree on purpose: the built .so travels to the GPU box with the repo snapshot."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libcseg_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(HERE, "_obj", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(ROOT, "include"),
               "-I", HERE, "-Wall", "-Wno-unused-function", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
