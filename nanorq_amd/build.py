"""Builds the in-tree native libraries.

  libnanorq_hip.so   product: gfx950 kernels + C ABI (include/nanorq_hip.h) + drop-in nanorq.h/io.h
                     layer, compiled with hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
  tests/emu/libsolve_emu.so   test support: CPU emulation of the solve workgroup (g++)

The .so files are git-ignored but travel to the GPU box with the working tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnanorq_hip.so")
EMU = os.path.join(ROOT, "tests", "emu", "libsolve_emu.so")
PEMU = os.path.join(ROOT, "tests", "emu", "libplanner_emu.so")

HIP_SOURCES = ["nrq_device.hip"]
CXX_SOURCES = ["planner_host.cpp"]
C_SOURCES = ["nanorq_api.c", "io.c"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (no CPU fallback exists)")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps():
    out = []
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            out.append(os.path.join(d, f))
    return out


def build_lib(force=False, verbose=False, out=None, extra=(), host_extra=(), reuse_hip_obj=False):
    """Build libnanorq_hip.so.  `out`/`extra` build a tuning variant (other file, extra compiler flags) that
    NANORQ_HIP_LIB=<path> makes the binding load instead -- for A/B runs on the GPU box.  `host_extra`: flags for the C / C++
    host sources only (gcc / g++: the sanitizer builds of tools/sanitize.sh); `reuse_hip_obj`: take the kernels' object of the
    regular build instead of compiling nrq_device.hip again."""
    if out is None and not force and not _newer(LIB, _deps()):
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build" if out is None else "build_" + os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    objs = []
    common = ["-O3", "-fPIC", "-Wno-pass-failed", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    common += list(extra)
    # an object is rebuilt when its source or any header is newer than it (the .hip takes ~100 s, the rest seconds)
    headers = [f for f in _deps() if f.endswith(".h")]

    def stale(o, src):
        return force or out is not None or _newer(o, [src] + headers)

    for s in HIP_SOURCES:
        o = os.path.join(objdir, s + ".o")
        if reuse_hip_obj and os.path.exists(os.path.join(HERE, "build", s + ".o")):
            objs.append(os.path.join(HERE, "build", s + ".o"))
            continue
        if stale(o, os.path.join(CSRC, s)):
            subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", *common, "-c", os.path.join(CSRC, s), "-o", o],
                           check=True)
        objs.append(o)
    for s in CXX_SOURCES:
        o = os.path.join(objdir, s + ".o")
        if stale(o, os.path.join(CSRC, s)):
            subprocess.run(["g++", "-std=c++17", "-Wall", *common, *host_extra, "-c", os.path.join(CSRC, s), "-o", o], check=True)
        objs.append(o)
    for s in C_SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        o = os.path.join(objdir, s + ".o")
        if stale(o, src):
            subprocess.run(["gcc", "-std=gnu11", "-Wall", "-D_FILE_OFFSET_BITS=64", *common, *host_extra, "-c", src, "-o", o], check=True)
        objs.append(o)
    target = out or LIB
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target, *objs, "-lpthread"], check=True)
    if verbose:
        print("built", target)
    return target


def build_tools(force=False):
    """tools/rqfile: the file encoder / decoder on the object API (C, linked against the library in place)."""
    src = os.path.join(ROOT, "tools", "rqfile.c")
    exe = os.path.join(ROOT, "tools", "rqfile")
    deps = [src, LIB] + [os.path.join(ROOT, "include", h) for h in ("nanorq.h", "nanorq_batch.h", "io.h")]
    if force or _newer(exe, deps):
        subprocess.run(["gcc", "-std=c99", "-D_DEFAULT_SOURCE", "-D_FILE_OFFSET_BITS=64", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                        src, "-o", exe, "-L" + os.path.dirname(LIB), "-lnanorq_hip", "-Wl,-rpath,$ORIGIN/../nanorq_amd", "-lm"], check=True)
    return exe


def build_emu(force=False):
    src = os.path.join(ROOT, "tests", "emu", "solve_emu.cpp")
    deps = [src, os.path.join(CSRC, "solve_body.h"), os.path.join(CSRC, "plan.h")]
    if force or _newer(EMU, deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", EMU, src], check=True)
    return EMU


def build_planner_emu(force=False):
    src = os.path.join(ROOT, "tests", "emu", "planner_emu.cpp")
    deps = [src] + [os.path.join(CSRC, f) for f in ("planner_body.h", "planner_seq.h", "solve_body.h", "plan.h", "rq_math.h")]
    if force or _newer(PEMU, deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", PEMU, src], check=True)
    return PEMU


if __name__ == "__main__":
    build_lib(force="-f" in sys.argv, verbose=True)
    build_emu(force="-f" in sys.argv)
    build_planner_emu(force="-f" in sys.argv)
    build_tools(force="-f" in sys.argv)
