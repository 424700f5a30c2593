"""-m gpu tier: tools/rqfile, the file encoder / decoder on the object API (counterpart of the reference's encode.c /
decode.c, same data.rq container): a file goes through encode -> data.rq with dropped packets -> decode and must come
back byte for byte; the container header must carry the OTI the API reports."""
import os
import struct
import subprocess

import numpy as np
import pytest

from capi import api
from util import payload

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe():
    import nanorq_amd.build as nbuild
    return nbuild.build_tools()


@pytest.mark.parametrize("size,T,loss,extra", [(3 * 1024 * 1024 + 17, 1280, 10.0, 4), (40_000, 64, 25.0, 3),
                                               (5 * 1024 * 1024, 72, 6.0, 5)])
def test_file_round_trip(tmp_path, size, T, loss, extra):
    src, rq, out = tmp_path / "in.bin", tmp_path / "data.rq", tmp_path / "out.bin"
    data = payload(size, seed=size % 1000)
    data.tofile(src)
    r = subprocess.run([_exe(), "encode", str(src), str(T), "-o", str(rq), "-l", str(loss), "-x", str(extra), "-s", "7"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    # the header is the OTI of an encoder with these parameters
    L = api()
    h = L.nanorq_encoder_new(size, T, 8)
    with open(rq, "rb") as f:
        common, scheme = struct.unpack("=QI", f.read(12))
    assert (common, scheme) == (L.nanorq_oti_common(h), L.nanorq_oti_scheme_specific(h))
    nblocks, Tn = L.nanorq_blocks(h), L.nanorq_symbol_size(h)
    L.nanorq_free(h)
    assert r.stdout.decode().count("\n") == nblocks
    assert (os.path.getsize(rq) - 12) % (4 + Tn) == 0
    r = subprocess.run([_exe(), "decode", str(out), "-i", str(rq)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    got = np.fromfile(out, dtype=np.uint8)
    assert got.size == size and np.array_equal(got, data)


def test_too_few_packets_is_reported(tmp_path):
    src, rq, out = tmp_path / "in.bin", tmp_path / "data.rq", tmp_path / "out.bin"
    payload(200_000, seed=3).tofile(src)
    r = subprocess.run([_exe(), "encode", str(src), "256", "-o", str(rq), "-l", "20", "-x", "0", "-s", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0
    # cut the container: the last packets (repair symbols) never arrive
    with open(rq, "r+b") as f:
        f.truncate(12 + (4 + 256) * 700)
    r = subprocess.run([_exe(), "decode", str(out), "-i", str(rq)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 2


def _refprog(name):
    p = os.path.join(ROOT, "oracle", "_refprog", name)
    if not os.path.isfile(p):
        pytest.skip("oracle/_refprog/%s not built (needs the reference tree at build time)" % name)
    return p


def test_container_is_the_reference_programs_container(tmp_path):
    """The reference's own encode / decode programs (compiled in the build container from its sources, against this
    library) and tools/rqfile read each other's data.rq."""
    size, T = 700_000, 1280
    data = payload(size, seed=11)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    src = tmp_path / "in.bin"
    data.tofile(src)
    # reference encode (6 % loss from the clock, +5) -> rqfile decode
    r = subprocess.run([_refprog("encode_hip"), str(src), str(T)], cwd=tmp_path / "a", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-300:]
    r = subprocess.run([_exe(), "decode", str(tmp_path / "a" / "out.bin"), "-i", str(tmp_path / "a" / "data.rq")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-300:]
    assert np.array_equal(np.fromfile(tmp_path / "a" / "out.bin", dtype=np.uint8), data)
    # rqfile encode -> reference decode
    r = subprocess.run([_exe(), "encode", str(src), str(T), "-o", str(tmp_path / "b" / "data.rq"), "-l", "8", "-x", "6", "-s", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-300:]
    r = subprocess.run([_refprog("decode_hip"), str(tmp_path / "b" / "out.bin")], cwd=tmp_path / "b", stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"failed" not in r.stdout, r.stdout.decode()[-300:]
    assert np.array_equal(np.fromfile(tmp_path / "b" / "out.bin", dtype=np.uint8), data)
