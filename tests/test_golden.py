"""CPU tier: the oracle against the committed fixtures (tests/golden/, provenance inside each file) -- the SURVEY
section 8(c) known answers of the reference and the oracle-generated vectors that pin later edits of the oracle."""
import numpy as np
import pytest

import golden_support as GS
from util import kat_payload

SURVEY = GS.load("survey_kat.json")
VEC = GS.load("oracle_vectors.json")


def test_fixtures_are_present_and_carry_provenance():
    assert "SURVEY.md" in SURVEY["provenance"] and "rq_oracle.c" in VEC["provenance"]
    assert len(VEC["encode"]) >= 12 and len(VEC["decode"]) >= 12 and len(VEC["failure_sweep"]["cases"]) >= 100


def test_survey_small_symbols(orc):
    s = SURVEY["small"]
    rep, _, _ = orc.encode_block(kat_payload(s["K"] * s["T"]), s["K"], s["T"], [int(e) for e in s["symbols"]])
    assert {e: rep[k].tobytes().hex() for k, e in enumerate(s["symbols"])} == s["symbols"]


@pytest.mark.parametrize("c", SURVEY["sha256_of_repair"], ids=lambda c: "K%d" % c["K"])
def test_survey_sha(orc, c):
    rep, _, _ = orc.encode_block(kat_payload(c["K"] * c["T"]), c["K"], c["T"], list(range(c["esi_lo"], c["esi_hi"])))
    assert GS.sha(rep) == c["sha256"]


def test_survey_params_and_schedule(orc):
    for K, row in SURVEY["params"].items():
        q = orc.params(int(K))
        assert [q[n] for n in SURVEY["params_order"]] == row
    for s in SURVEY["schedule_encode"]:
        p = orc.params(s["K"])
        r, st = orc.plan_probe(s["K"], np.arange(p["Kp"], dtype=np.uint32))
        assert r == 1 and (st["i"], st["u"], st["recorded_ops"], st["n1"], st["nB"], st["n0"]) == \
            (s["i"], s["u"], s["recorded"], s["n1"], s["nB"], s["n0"])


@pytest.mark.parametrize("c", VEC["encode"], ids=lambda c: "K%d_Kp%d_T%d" % (c["K"], c["Kp"], c["T"]))
def test_encode_vectors(orc, c):
    from util import payload
    src = payload(c["K"] * c["T"], seed=c["payload_seed"]).reshape(c["K"], c["T"])
    rep, inter, st = orc.encode_block(src, c["K"], c["T"], c["esis"], want_inter=True, Kp=c["Kp"])
    assert GS.sha(rep) == c["sha256_repair"] and GS.sha(inter) == c["sha256_intermediate"]
    assert rep[0][:16].tobytes().hex() == c["first_repair_hex"] and (st["i"], st["u"]) == (c["i"], c["u"])
    # systematic property of the intermediate symbols, padding symbols of a larger table row included (zero)
    Kp = c["Kp"] or orc.params(c["K"])["Kp"]
    for esi in {0, c["K"] // 2, c["K"] - 1, Kp - 1}:
        acc = np.zeros(c["T"], np.uint8)
        for col in orc.lt_columns(Kp, esi):
            acc ^= inter[col]
        assert np.array_equal(acc, src[esi] if esi < c["K"] else np.zeros(c["T"], np.uint8))


@pytest.mark.parametrize("c", VEC["decode"], ids=lambda c: "K%d_Kp%d_oh%d_%s" % (c["K"], c["Kp"], c["overhead"], c["order"]))
def test_decode_vectors(orc, c):
    src, lost, esis = GS.decode_inputs(c)
    K, T = c["K"], c["T"]
    rep, _, _ = orc.encode_block(src, K, T, np.arange(K, K + len(lost) + c["overhead"], dtype=np.uint32), Kp=c["Kp"])
    syms = np.stack([src[e] if e < K else rep[e - K] for e in esis])
    ok, out, st = orc.decode_block(esis, syms, K, T, Kp=c["Kp"])
    assert ok == c["decodable"] and (st["i"], st["u"]) == (c["i"], c["u"])
    if ok:
        assert GS.sha(out) == c["sha256_recovered"] and np.array_equal(out, src) == c["equals_source"]


def test_failure_sweep_vectors(orc):
    fs = VEC["failure_sweep"]
    K = fs["K"]
    p = orc.params(K)
    for c in fs["cases"]:
        isis = list(range(p["Kp"]))
        for g, e in enumerate(c["lost"]):
            isis[e] = c["repair_esis"][g] + p["Kp"] - K
        assert (orc.plan_probe(K, np.array(isis, np.uint32))[0] == 1) == c["decodable"]


def test_every_table2_row_is_nonsingular_and_matches_its_fixture(orc):
    """All 477 K' rows of RFC 6330 Table 2 (reference include/table2.h:6-211, lib/params.c:21-45): with the oracle's Rand / Deg /
    Tuple / LDPC / HDPC restatement and the table's J(K') the constraint matrix of K' source symbols must be nonsingular -- that
    is what J(K') was chosen for -- the encoding systematic (checked when the fixture was written: tools/gen_kprime_sweep.py),
    and the first four repair symbols and the intermediate symbols the committed ones.  A mis-restated generator survives one
    K' with p ~ 0.99, all 477 with p < 1 %."""
    import hashlib
    doc = GS.load("kprime_sweep.json")
    rows = doc["rows"]
    assert len(rows) == 477 and "oracle/rq_oracle.c" in doc["provenance"]
    T = doc["T"]
    from util import payload
    for r in rows:
        Kp = r["Kp"]
        assert orc.params(Kp)["Kp"] == Kp and orc.params(Kp)["J"] == r["J"]
        src = payload(Kp * T, seed=doc["payload_seed"], block=Kp).reshape(Kp, T)
        rep, inter, st = orc.encode_block(src, Kp, T, np.arange(Kp, Kp + doc["repair_per_row"], dtype=np.uint32), want_inter=True)
        assert rep is not None, "K'=%d: the oracle finds the systematic constraint matrix singular" % Kp
        assert (int(st["i"]), int(st["u"])) == (r["i"], r["u"]), Kp
        assert hashlib.sha256(rep.tobytes()).hexdigest() == r["sha256_repair"], Kp
        assert hashlib.sha256(inter.tobytes()).hexdigest() == r["sha256_intermediate"], Kp


def test_host_planner_solves_every_table2_row():
    """The product's host planner (planner_host.cpp: the encoder's plan below L = 12000, the capacity fallback of every decode)
    on the same 477 matrices: every one must come out solvable (status 0) with L = K' + S + H pivots + inactive columns."""
    import nanorq_amd
    from nanorq_amd import binding as b
    for r in GS.load("kprime_sweep.json")["rows"]:
        Kp = r["Kp"]
        kc = b.host_kconst(Kp)
        h = b.plan_header(b.host_plan(Kp, np.arange(Kp, dtype=np.uint32), kc))
        assert h["status"] == 0 and h["npiv"] + h["u"] == h["L"] == nanorq_amd.params(Kp)["L"], (Kp, h["status"])
