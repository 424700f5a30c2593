"""-m gpu tier: the HIP path (C ABI) against the committed fixtures of tests/golden/ -- no oracle in the loop
except to produce the repair symbols a decode fixture receives."""
import numpy as np
import pytest

import golden_support as GS
from util import kat_payload, payload

pytestmark = pytest.mark.gpu

SURVEY = GS.load("survey_kat.json")
VEC = GS.load("oracle_vectors.json")


@pytest.fixture(scope="module")
def G():
    import gpu_support
    gpu_support.ctx()
    return gpu_support


def test_survey_small_symbols(G):
    s = SURVEY["small"]
    rep, _ = G.gpu_encode(kat_payload(s["K"] * s["T"]).reshape(1, s["K"], s["T"]), s["K"], s["T"], [int(e) for e in s["symbols"]])
    assert {e: rep[0, k].tobytes().hex() for k, e in enumerate(s["symbols"])} == s["symbols"]


@pytest.mark.parametrize("c", SURVEY["sha256_of_repair"], ids=lambda c: "K%d" % c["K"])
def test_survey_sha(G, c):
    K, T = c["K"], c["T"]
    rep, _ = G.gpu_encode(kat_payload(K * T).reshape(1, K, T), K, T, list(range(c["esi_lo"], c["esi_hi"])))
    assert GS.sha(rep[0]) == c["sha256"]


@pytest.mark.parametrize("c", VEC["encode"], ids=lambda c: "K%d_Kp%d_T%d" % (c["K"], c["Kp"], c["T"]))
def test_encode_vectors(G, c):
    K, T = c["K"], c["T"]
    src = payload(K * T, seed=c["payload_seed"]).reshape(1, K, T)
    rep, inter = G.gpu_encode(src, K, T, c["esis"], want_inter=True, Kp=c["Kp"])
    assert GS.sha(rep[0]) == c["sha256_repair"], "repair symbols"
    assert GS.sha(inter[0]) == c["sha256_intermediate"], "intermediate symbols"


@pytest.mark.parametrize("c", VEC["decode"], ids=lambda c: "K%d_Kp%d_oh%d_%s" % (c["K"], c["Kp"], c["overhead"], c["order"]))
def test_decode_vectors(G, c):
    """Repair symbols come from the HIP encoder (already pinned by test_encode_vectors); the i-th missing ESI takes
    the i-th repair symbol IN ARRIVAL ORDER, surplus symbols become extra rows (nanorq.c:527-565)."""
    src, lost, esis = GS.decode_inputs(c)
    K, T = c["K"], c["T"]
    rep_in_order = np.array([e for e in esis if e >= K], np.uint32)
    rep, _ = G.gpu_encode(src.reshape(1, K, T), K, T, rep_in_order, Kp=c["Kp"])
    work = src.copy()
    work[lost] = 0xA5
    st, out, _ = G.gpu_decode(work.reshape(1, K, T), K, T, [lost], [rep_in_order], [rep[0]], Kp=c["Kp"])
    assert bool(st[0]) == c["decodable"]
    if c["decodable"]:
        assert GS.sha(out[0]) == c["sha256_recovered"]


def test_failure_sweep_vectors(G):
    fs = VEC["failure_sweep"]
    K, T, n = fs["K"], fs["T"], len(fs["cases"])
    src = payload(K * T, seed=4).reshape(K, T)
    all_rep, _ = G.gpu_encode(src.reshape(1, K, T), K, T, np.arange(K, K + 60, dtype=np.uint32))
    lost = [np.array(c["lost"], np.uint32) for c in fs["cases"]]
    resi = [np.array(c["repair_esis"], np.uint32) for c in fs["cases"]]
    work = np.repeat(src.reshape(1, K, T), n, axis=0).copy()
    for b in range(n):
        work[b][lost[b]] = 0xFF
    st, out, _ = G.gpu_decode(work, K, T, lost, resi, [all_rep[0][r - K] for r in resi])
    assert [bool(x) for x in st] == [c["decodable"] for c in fs["cases"]]
    for b in range(n):
        assert np.array_equal(out[b], src if st[b] else work[b])


def test_every_table2_row_on_the_hip_encoder(G):
    """All 477 K' rows of RFC 6330 Table 2 through the HIP encoder (K = K', T = 8): every systematic constraint matrix must be
    found nonsingular by the product's planners (host planner below L = 12000, nrq_plan_kernel above), the intermediate
    symbols and the repair symbols ESI K'..K'+3 must be the committed ones (tests/golden/kprime_sweep.json, written by
    tools/gen_kprime_sweep.py), and the source symbols must come back out of the intermediate symbols (systematic property,
    RFC 6330 5.3.3.4.2; on the device: nrq_gen_symbols).  Reference: include/table2.h:6-211, lib/params.c:21-45,
    lib/tuple.c:21-43, lib/precode.c:90-97."""
    import nanorq_amd
    doc = GS.load("kprime_sweep.json")
    T, nrep = doc["T"], doc["repair_per_row"]
    assert len(doc["rows"]) == 477
    c = G.ctx()
    for r in doc["rows"]:
        Kp = r["Kp"]
        L = nanorq_amd.params(Kp)["L"]
        src = payload(Kp * T, seed=doc["payload_seed"], block=Kp).reshape(Kp, T)
        d_src, d_rep, d_int, d_out = c.alloc(Kp * T), c.alloc(nrep * T), c.alloc(L * T), c.alloc(Kp * T)
        try:
            c.upload(d_src, src)
            c.memset(d_rep, 0xCD, nrep * T); c.memset(d_int, 0xCD, L * T); c.memset(d_out, 0xCD, Kp * T)
            c.encode_blocks(Kp, T, 1, d_src, Kp * T, d_rep, nrep * T, np.arange(Kp, Kp + nrep, dtype=np.uint32), d_int, L * T)
            c.gen_symbols(Kp, T, 1, d_int, L * T, np.arange(Kp, dtype=np.uint32), d_out, Kp * T)
            c.sync()
            assert GS.sha(c.download(d_rep, nrep * T)) == r["sha256_repair"], Kp
            assert GS.sha(c.download(d_int, L * T)) == r["sha256_intermediate"], Kp
            assert np.array_equal(c.download(d_out, Kp * T).reshape(Kp, T), src), "K'=%d: LT(intermediate) is not the source block" % Kp
        finally:
            for d in (d_src, d_rep, d_int, d_out):
                c.free(d)
        if Kp % 7 == 0:
            c.clear_plan_cache()
