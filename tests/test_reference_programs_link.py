"""Drop-in proof at the source level: the reference's own programs (encode.c, decode.c, benchmark.c) compile
against THIS repository's include/nanorq.h + include/io.h and link against libnanorq_hip.so, unmodified.
Needs /root/reference (build container only: the sources are read in place, never copied); skipped elsewhere.
Running them needs a GPU, which this tier does not have -- the same call sequences are exercised on the GPU
by tests/test_gpu_api.py."""
import os
import subprocess

import pytest

import nanorq_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "benchmark.c")), reason="reference tree not mounted")


@pytest.mark.parametrize("prog", ["benchmark.c", "encode.c", "decode.c"])
def test_reference_program_builds_against_this_library(tmp_path, prog):
    lib = nanorq_amd.lib_path()
    nanorq_amd.lib()
    out = str(tmp_path / prog.replace(".c", ""))
    # our headers first (nanorq.h, io.h); the reference include dir only supplies its kvec.h utility macros
    cmd = ["gcc", "-std=c99", "-D_DEFAULT_SOURCE", "-D_FILE_OFFSET_BITS=64", "-O1", "-I" + os.path.join(ROOT, "include"),
           "-idirafter", os.path.join(REF, "include"), os.path.join(REF, prog), "-o", out,
           "-L" + os.path.dirname(lib), "-lnanorq_hip", "-Wl,-rpath," + os.path.dirname(lib), "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    # every nanorq_/ioctx_ symbol the program imports is provided by our library
    und = subprocess.run(["nm", "-u", out], stdout=subprocess.PIPE).stdout.decode()
    wanted = [l.split()[-1] for l in und.splitlines() if "nanorq_" in l or "ioctx_" in l]
    assert wanted, "program does not use the API?"
    defined = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE).stdout.decode()
    for sym in wanted:
        assert (" T " + sym.split("@")[0]) in defined, sym
