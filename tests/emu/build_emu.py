"""TEST INFRASTRUCTURE. Compiles kernel sources of contrastiveseg_amd/csrc for the HOST against the CPU emulation of the
execution model in tests/emu/hip/hip_runtime.h + emu_runtime.cpp -> tests/emu/_build/libcseg_emu.so, which exports the
same C-ABI entry points as libcseg_hip.so for those files, taking host pointers. The only edit made to a source is the
declaration of its dynamic LDS (`extern __shared__ T name[];` has no host equivalent): it becomes a pointer to the
emulator's 160 KB buffer. Used by tests/test_emu_*.py; never by the product."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "contrastiveseg_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libcseg_emu.so")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
CLANG = os.environ.get("CSEG_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
_DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(16\)\)\)\s+)?([\w ]+?)\s+(\w+)\[\];")


class EmuBuildError(RuntimeError):
    """The host toolchain could not build the emulated library (tests that need it skip with this message)."""


def _rewrite(text):
    return _DYN.sub(lambda m: "%s* %s = reinterpret_cast<%s*>(emu::dyn_lds());" % (m.group(1), m.group(2), m.group(1)), text)


def _deps():
    return ([os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + [
            os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "emu_runtime.cpp"), os.path.abspath(__file__),
            os.path.join(ROOT, "include", "cseg_hip.h")])


def _fresh():
    return os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in _deps())


def build(force=False):
    """Builds libcseg_emu.so when a source is newer than it. Serialised across processes with a file lock: the ranks of a
    world-size-2 test are spawned together and would otherwise both rebuild a stale library into the same files."""
    if not force and _fresh():
        return OUT
    if not os.path.exists(CLANG):
        raise EmuBuildError("host clang++ of the ROCm toolchain not found at %s" % CLANG)
    os.makedirs(OUT_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and _fresh():                   # another process built it while this one waited
            return OUT
        return _build_locked()


def _build_locked():
    for h in os.listdir(CSRC):                       # headers that declare dynamic LDS get the same rewrite
        if h.endswith(".h"):
            with open(os.path.join(OUT_DIR, h), "w") as f:
                f.write('#line 1 "%s"\n' % os.path.join(CSRC, h) + _rewrite(open(os.path.join(CSRC, h)).read()))
    flags = ["-std=c++20", "-O1", "-g", "-fPIC", "-pthread", "-I", HERE, "-I", OUT_DIR, "-I", CSRC, "-I", os.path.join(ROOT, "include"),
             "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unknown-pragmas", "-Wno-pass-failed"]
    objs, procs = [], []
    for name in SOURCES:
        text = _rewrite(open(os.path.join(CSRC, name)).read())
        gen = os.path.join(OUT_DIR, name.replace(".hip", "_emu.cpp"))
        with open(gen, "w") as f:
            f.write('#line 1 "%s"\n' % os.path.join(CSRC, name) + text)
        objs.append(gen[:-4] + ".o")
        procs.append((name, subprocess.Popen([CLANG] + flags + ["-c", gen, "-o", objs[-1]])))
    rt = os.path.join(OUT_DIR, "emu_runtime.o")
    procs.append(("emu_runtime.cpp", subprocess.Popen([CLANG] + flags + ["-c", os.path.join(HERE, "emu_runtime.cpp"), "-o", rt])))
    for name, p in procs:
        if p.wait() != 0:
            raise EmuBuildError("host compilation of %s for the emulator failed" % name)
    if subprocess.call([CLANG, "-shared", "-pthread", "-o", OUT] + objs + [rt]) != 0:
        raise EmuBuildError("linking libcseg_emu.so failed")
    return OUT


if __name__ == "__main__":
    print(build(force=True))
