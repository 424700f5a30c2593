"""RFC 6330 constants: the product's generated header (tools/gen_tables.py), the oracle's own generated header
(tools/gen_oracle_tables.py) and both libraries' params() against the committed fixture, row by row.  The oracle does
not include the product's table file, so a wrong entry cannot be wrong on both sides of an oracle comparison unnoticed."""
import hashlib
import json
import os
import re

import pytest

import nanorq_amd
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fix():
    with open(os.path.join(ROOT, "tests", "golden", "rfc6330_tables.json")) as f:
        d = json.load(f)
    rows = d["table2_rows"]
    assert len(rows) == 477 and len(d["v_words"]) == 1024
    assert hashlib.sha256(("\n".join(",".join(map(str, r)) for r in rows)).encode()).hexdigest() == d["table2_sha256"]
    assert hashlib.sha256((",".join(map(str, d["v_words"]))).encode()).hexdigest() == d["v_sha256"]
    return d


def _macro_body(text, name):
    m = re.search(r"#define %s \\\n((?:.*\\\n)+)" % name, text)
    assert m, name
    return m.group(1)


def test_product_header_rows(fix):
    text = open(os.path.join(ROOT, "nanorq_amd", "csrc", "rfc6330_tables.h")).read()
    rows = [[int(x) for x in g] for g in re.findall(r"\{(\d+),(\d+),(\d+),(\d+),(\d+)\}", _macro_body(text, "RQ_TABLE2_ROWS"))]
    assert len(rows) == 477
    for got, want in zip(rows, fix["table2_rows"]):
        assert got == want
    v = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{8})u", _macro_body(text, "RQ_V_WORDS"))]
    assert v == fix["v_words"]


def test_oracle_header_rows(fix):
    text = open(os.path.join(ROOT, "oracle", "orc_tables.h")).read()
    cols = []
    for name in ("ORC_KP", "ORC_J", "ORC_S", "ORC_H", "ORC_W"):
        m = re.search(r"%s\[ORC_TABLE2_COUNT\] = \{(.*?)\};" % name, text, re.S)
        cols.append([int(x) for x in re.findall(r"\d+", m.group(1))])
        assert len(cols[-1]) == 477
    for i, want in enumerate(fix["table2_rows"]):
        assert [c[i] for c in cols] == want
    m = re.search(r"ORC_V\[4\]\[256\] = \{(.*)\};", text, re.S)
    assert [int(x) for x in re.findall(r"(\d+)u", m.group(1))] == fix["v_words"]


def test_params_of_both_libraries_row_by_row(fix):
    """K' -> (J, S, H, W) through the compiled tables of the product library and of the oracle; also K = K' - 1
    (the row a block of that size is padded to)"""
    prev = 0
    for kp, j, s, h, w in fix["table2_rows"]:
        for K in {kp, max(prev + 1, kp - 1)}:
            for p in (nanorq_amd.params(K), oracle.params(K)):
                assert (p["Kp"], p["J"], p["S"], p["H"], p["W"]) == (kp, j, s, h, w), (K, p)
                assert p["L"] == kp + s + h and p["P"] == p["L"] - w
        prev = kp


def test_lt_rows_of_the_product_base_matrix_equal_the_oracle_tuples():
    """the V tables IN USE: every source LT row of the product's base constraint matrix (built from the product's copy of
    V0..V3 and of the degree table) against the column list the oracle's rnd / tuple code gives for that ISI"""
    import numpy as np
    for K in (10, 101, 1032, 8194):
        kc = nanorq_amd.host_kconst(K)
        h = np.frombuffer(kc, dtype=np.uint32, count=16)
        Kp, S, H, L, nnz, off_rptr, off_cidx = int(h[0]), int(h[1]), int(h[2]), int(h[7]), int(h[10]), int(h[11]), int(h[12])
        assert Kp == K
        rptr = np.frombuffer(kc, dtype=np.uint32, count=L + 1, offset=off_rptr)
        cidx = np.frombuffer(kc, dtype=np.uint16, count=nnz, offset=off_cidx)
        step = 1 if K <= 1032 else 37
        for isi in range(0, Kp, step):
            r = S + H + isi
            got = sorted(int(x) for x in cidx[rptr[r]:rptr[r + 1]])
            want = sorted(int(x) for x in oracle.lt_columns(K, isi))
            assert got == want, (K, isi)
