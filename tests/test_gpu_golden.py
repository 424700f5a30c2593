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
