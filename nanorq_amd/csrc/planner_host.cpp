/*
 * planner_host.cpp -- symbolic stage on the host: constraint structure -> device plan (plan.h).
 *
 * Stands where the reference's precode_matrix_gen / patch_precode_matrix /
 * precode_matrix_invert stand (precode.c:90-97, :347-377; nanorq.c:527-547), but it is a
 * different algorithm with the same mathematical result (the solution C of A*C = D is
 * unique when rank(A) = L, SURVEY.md headline fact 5):
 *   1. breadth-first peeling: every row with exactly one unresolved column in V claims it in
 *      the same round (rounds = dependency levels, ~400 at K=8192 instead of the ~1900 of the
 *      reference's LIFO order); when no such row exists the sparsest remaining row is taken and all
 *      but one of its columns are inactivated (RFC 6330 section 5.4.2.2 phase 1; the reference only
 *      ever takes rows of weight <= 2, precode.c:115-126);
 *   2. W = X^-1 * A_top,U as a bit matrix (replaces the reference's undo/redo passes over X,
 *      precode.c:23-32 passes B and D);
 *   3. the u inactive columns: GF(2) Gauss-Jordan on the leftover binary rows first, the H HDPC
 *      rows (GF(256)) only for the columns the binary rows cannot resolve (reference:
 *      solve_gf2 / fill_HDPC / solve_gf256, precode.c:232-315).
 * The GPU planner (planner_body.h, instantiated by nrq_plan_kernel in nrq_device.hip) produces the same format; this one is the portable twin used
 * for plans that are built once per K' (encode) and as its cross-check.
 */
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "plan.h"
#include "planner_host.h"
#include "rq_math.h"

namespace {

/* ---- GF(256), RFC 6330 section 5.7 ---- */
struct GF {
  uint8_t exp[510], log[256], inv[256];
  GF() {
    uint32_t x = 1;
    for (int e = 0; e < 255; e++) {
      exp[e] = (uint8_t)x;
      log[x] = (uint8_t)e;
      x <<= 1;
      if (x & 0x100) x ^= 0x11D;
    }
    for (int e = 255; e < 510; e++) exp[e] = exp[e - 255];
    log[0] = 0; inv[0] = 0;
    for (int v = 1; v < 256; v++) inv[v] = exp[255 - log[v]];
  }
  uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
};
const GF gf;

inline uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

struct Arena {
  std::vector<uint8_t> buf;
  uint32_t reserve(uint32_t bytes) {
    uint32_t off = align16((uint32_t)buf.size());
    buf.resize((size_t)off + bytes, 0);
    return off;
  }
  template <class T> T *at(uint32_t off) { return reinterpret_cast<T *>(buf.data() + off); }
};

inline bool bit(const uint32_t *row, uint32_t x) { return (row[x >> 5] >> (x & 31)) & 1u; }
inline void flip(uint32_t *row, uint32_t x) { row[x >> 5] ^= 1u << (x & 31); }

} // namespace

/* ------------------------------------------------------------------------------------------ */
extern "C" int nrq_host_kconst_build(uint32_t K, uint8_t **out, uint32_t *out_bytes) {
  rq_params p;
  if (!rq_params_init(K, &p)) return -1;
  const uint32_t n = p.Kp + p.S, H = p.H, S = p.S, L = p.L, W = p.W;
  /* base structure: LDPC rows, empty HDPC rows, LT rows of ISI 0..K'-1 */
  std::vector<uint32_t> rptr(L + 1, 0);
  std::vector<uint16_t> cidx;
  {
    std::vector<std::vector<uint16_t>> ldpc(S);
    for (uint32_t c = 0; c < p.B; c++) {
      uint32_t blk = c / S;
      ldpc[c % S].push_back((uint16_t)c);
      ldpc[(c + blk + 1) % S].push_back((uint16_t)c);
      ldpc[(c + 2 * (blk + 1)) % S].push_back((uint16_t)c);
    }
    for (uint32_t r = 0; r < S; r++) {
      ldpc[r].push_back((uint16_t)(p.B + r));
      ldpc[r].push_back((uint16_t)(W + r % p.P));
      ldpc[r].push_back((uint16_t)(W + (r + 1) % p.P));
      rptr[r] = (uint32_t)cidx.size();
      cidx.insert(cidx.end(), ldpc[r].begin(), ldpc[r].end());
    }
    for (uint32_t r = S; r < S + H; r++) rptr[r] = (uint32_t)cidx.size();
    uint32_t tmp[RQ_MAX_LT_COLS];
    for (uint32_t j = 0; j < p.Kp; j++) {
      rptr[S + H + j] = (uint32_t)cidx.size();
      uint32_t m = rq_lt_columns(&p, j, tmp);
      for (uint32_t q = 0; q < m; q++) cidx.push_back((uint16_t)tmp[q]);
    }
    rptr[L] = (uint32_t)cidx.size();
  }
  const uint32_t nnz = (uint32_t)cidx.size();
  std::vector<uint32_t> cptr(L + 1, 0), state(L, 0);
  std::vector<uint16_t> ridx(nnz);
  for (uint16_t c : cidx) cptr[c + 1]++;
  for (uint32_t c = 0; c < L; c++) cptr[c + 1] += cptr[c];
  {
    std::vector<uint32_t> fill(cptr.begin(), cptr.end() - 1);
    for (uint32_t r = 0; r < L; r++) {
      uint32_t cnt = 0, sum = 0;
      for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++) {
        ridx[fill[cidx[e]]++] = (uint16_t)r;
        if (cidx[e] < W) { cnt++; sum += cidx[e]; }
      }
      if (cnt > 255 || sum >= (1u << 24)) return -6; /* does not happen for RFC 6330 parameters */
      state[r] = (cnt << 24) | sum;
    }
  }
  nrq_kconst_hdr h;
  memset(&h, 0, sizeof(h));
  h.Kp = p.Kp; h.S = S; h.H = H; h.n = n; h.L = L; h.W = W; h.P = p.P; h.nnz = nnz;
  uint32_t off = align16((uint32_t)sizeof(h));
  h.off_g = off; off = align16(off + H * n);
  h.off_b12 = off; off = align16(off + n);
  h.off_rptr = off; off = align16(off + (L + 1) * 4);
  h.off_cidx = off; off = align16(off + nnz * 2);
  h.off_cptr = off; off = align16(off + (L + 1) * 4);
  h.off_ridx = off; off = align16(off + nnz * 2);
  h.off_state = off; off = align16(off + L * 4);
  h.off_gt = off; off = align16(off + n * 16);
  h.off_erow = off; off = align16(off + nnz * 2);
  h.off_chead = off; off = align16(off + L * NRQ_CHEAD * 2);
  h.total_bytes = off;
  uint8_t *buf = (uint8_t *)calloc(off, 1);
  if (!buf) return -2;
  uint8_t *G = buf + h.off_g, *b12 = buf + h.off_b12;
  /* HDPC = MT * GAMMA evaluated right to left: column c = alpha * column c+1, plus the two unit
   * entries of MT's column c; the last column is alpha^row (RFC 6330 section 5.3.3.3). */
  for (uint32_t r = 0; r < H; r++) G[(size_t)r * n + n - 1] = gf.exp[r];
  for (int64_t c = (int64_t)n - 2; c >= 0; c--) {
    for (uint32_t r = 0; r < H; r++) {
      uint8_t right = G[(size_t)r * n + c + 1];
      G[(size_t)r * n + c] = right ? gf.exp[gf.log[right] + 1] : 0;
    }
    uint32_t b1 = rq_rand((uint32_t)c + 1, 6, H);
    uint32_t b2 = (b1 + rq_rand((uint32_t)c + 1, 7, H - 1) + 1) % H;
    G[(size_t)b1 * n + c] ^= 1;
    G[(size_t)b2 * n + c] ^= 1;
    b12[c] = (uint8_t)(b1 | (b2 << 4));
  }
  memcpy(buf + h.off_rptr, rptr.data(), (size_t)(L + 1) * 4);
  memcpy(buf + h.off_cidx, cidx.data(), (size_t)nnz * 2);
  memcpy(buf + h.off_cptr, cptr.data(), (size_t)(L + 1) * 4);
  memcpy(buf + h.off_ridx, ridx.data(), (size_t)nnz * 2);
  memcpy(buf + h.off_state, state.data(), (size_t)L * 4);
  for (uint32_t c = 0; c < L; c++)
    for (uint32_t q = 0; q < NRQ_CHEAD; q++)
      reinterpret_cast<uint16_t *>(buf + h.off_chead)[(size_t)c * NRQ_CHEAD + q] = cptr[c] + q < cptr[c + 1] ? ridx[cptr[c] + q] : (uint16_t)0xFFFFu;
  for (uint32_t r = 0; r < L; r++)
    for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++) reinterpret_cast<uint16_t *>(buf + h.off_erow)[e] = (uint16_t)r;
  for (uint32_t c = 0; c < n; c++)
    for (uint32_t r = 0; r < H; r++) buf[h.off_gt + (size_t)c * 16 + r] = G[(size_t)r * n + c];
  memcpy(buf, &h, sizeof(h));
  *out = buf;
  *out_bytes = off;
  return 0;
}

extern "C" void nrq_host_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------ */
/* comp_rows: 0 = an inactivation event takes the first open rows with two columns in V it finds, one after the other (what the
 * device planner does); n > 0 = RFC 6330 section 5.4.2.2's choice -- a row from the LARGEST component of the graph whose edges
 * are those rows -- and, in the same event, one row from each of the next n - 1 components (rows of one component must not
 * share an event: the first inactivation lets the whole component peel, a second one in it is a column wasted). */
static int host_plan_build_with(uint32_t K, uint32_t nrows, const uint32_t *isis, const uint8_t *kconst, uint8_t **out, uint32_t *out_bytes,
                                uint32_t comp_rows) {
  rq_params p;
  if (!rq_params_init(K, &p)) return -1;
  if (nrows < p.Kp) return -1;
  const nrq_kconst_hdr *kh = reinterpret_cast<const nrq_kconst_hdr *>(kconst);
  if (!kh || kh->Kp != p.Kp) return -1;
  const uint8_t *G = kconst + kh->off_g;
  const uint32_t S = p.S, H = p.H, W = p.W, L = p.L, n_hd = p.Kp + p.S;
  const uint32_t M = S + H + nrows;
  if (M > 65535u) return -3; /* slots are 16-bit */

  /* ---- A: binary constraint rows as CSR (HDPC rows stay empty) ---- */
  /* (the LDPC rows and the LT rows of ISI < K' are the per-K' constants' base rows -- nrq_host_kconst_build made them with the
   * same generator, in the same order: copied, not generated again; only the repair symbols' rows are generated here.  A tenth of
   * a decode plan's build time at K=1000.) */
  std::vector<uint32_t> rptr(M + 1, 0);
  std::vector<uint16_t> cidx;
  {
    const uint32_t *brp = reinterpret_cast<const uint32_t *>(kconst + kh->off_rptr);
    const uint16_t *bci = reinterpret_cast<const uint16_t *>(kconst + kh->off_cidx);
    cidx.reserve((size_t)kh->nnz + (size_t)(nrows - p.Kp + 8u) * RQ_MAX_LT_COLS);
    for (uint32_t r = 0; r < S; r++) {
      rptr[r] = (uint32_t)cidx.size();
      cidx.insert(cidx.end(), bci + brp[r], bci + brp[r + 1]);
    }
    for (uint32_t r = S; r < S + H; r++) rptr[r] = (uint32_t)cidx.size();
    uint32_t tmp[RQ_MAX_LT_COLS];
    for (uint32_t k = 0; k < nrows; k++) {
      rptr[S + H + k] = (uint32_t)cidx.size();
      if (isis[k] < p.Kp) {
        const uint32_t br = S + H + isis[k];
        cidx.insert(cidx.end(), bci + brp[br], bci + brp[br + 1]);
      } else {
        uint32_t n = rq_lt_columns(&p, isis[k], tmp);
        for (uint32_t q = 0; q < n; q++) cidx.push_back((uint16_t)tmp[q]);
      }
    }
    rptr[M] = (uint32_t)cidx.size();
  }
  const uint32_t nnz = (uint32_t)cidx.size();
  /* column lists are sets by construction (W, P1 prime; S > B/S + 1), as in the reference */
  (void)nnz;
  /* CSC */
  std::vector<uint32_t> cptr(L + 1, 0);
  std::vector<uint16_t> ridx(cidx.size());
  for (uint16_t c : cidx) cptr[c + 1]++;
  for (uint32_t c = 0; c < L; c++) cptr[c + 1] += cptr[c];
  {
    std::vector<uint32_t> fill(cptr.begin(), cptr.end() - 1);
    for (uint32_t r = 0; r < M; r++)
      for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++) ridx[fill[cidx[e]]++] = (uint16_t)r;
  }

  /* ---- phase 1: breadth-first peeling with inactivation ---- */
  enum : uint8_t { IN_V = 0, PIVOT = 1, INACTIVE = 2 };
  std::vector<uint8_t> cstate(L, IN_V);
  for (uint32_t c = W; c < L; c++) cstate[c] = INACTIVE; /* PI columns start inactive */
  std::vector<uint32_t> cnt(M, 0), xs(M, 0);
  for (uint32_t r = 0; r < M; r++)
    for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++)
      if (cidx[e] < W) { cnt[r]++; xs[r] ^= cidx[e]; }
  std::vector<uint8_t> assigned(M, 0);
  std::vector<uint16_t> owner(L, NRQ_NOSLOT); /* pivot row of a column */
  std::vector<uint32_t> kof(L, 0);            /* pivot index of a column */
  std::vector<uint32_t> level(M, 0);
  std::vector<uint16_t> pivslot, pivcol;
  std::vector<uint32_t> inact_order; /* V columns in the order they were inactivated */
  pivslot.reserve(L); pivcol.reserve(L);
  uint32_t nV = W, nlev = 0;
  std::vector<uint32_t> frontier, next;
  for (uint32_t r = 0; r < M; r++)
    if (cnt[r] == 1) frontier.push_back(r);

  std::vector<uint32_t> edge_a, edge_b; /* component rule: the two V columns of a row that has two left */
  std::vector<uint32_t> par, sz, two_rows, touched, comp_of; /* ... union-find over the columns, the rows with two columns, what an event touched */
  std::vector<uint8_t> seen;
  bool track_two = false;
  auto drop_column = [&](uint32_t c) { /* c leaves V */
    for (uint32_t e = cptr[c]; e < cptr[c + 1]; e++) {
      uint32_t r = ridx[e];
      cnt[r]--; xs[r] ^= c;
      if (cnt[r] == 1 && !assigned[r]) next.push_back(r);
      else if (cnt[r] == 2 && track_two && !assigned[r]) two_rows.push_back(r);
    }
  };

  std::vector<uint32_t> claimed;
  while (nV > 0) {
    if (!frontier.empty()) {
      claimed.clear();
      for (uint32_t r : frontier) {
        if (assigned[r] || cnt[r] != 1) continue;
        uint32_t c = xs[r];
        if (owner[c] != NRQ_NOSLOT) continue; /* another row of this round got it first */
        owner[c] = (uint16_t)r;
        assigned[r] = 1;
        claimed.push_back(c);
      }
      next.clear();
      for (uint32_t c : claimed) {
        uint32_t r = owner[c], lv = 0;
        for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++) {
          uint32_t c2 = cidx[e];
          if (c2 != c && cstate[c2] == PIVOT) lv = std::max(lv, level[owner[c2]] + 1);
        }
        level[r] = lv;
        nlev = std::max(nlev, lv + 1);
        kof[c] = (uint32_t)pivslot.size();
        pivslot.push_back((uint16_t)r);
        pivcol.push_back((uint16_t)c);
      }
      for (uint32_t c : claimed) { cstate[c] = PIVOT; nV--; }
      for (uint32_t c : claimed) drop_column(c);
      frontier.swap(next);
      continue;
    }
    /* no weight-1 row: take the sparsest open rows (up to NRQ_MULTI_INACT of them, one after the other) and
     * inactivate all but one column of each */
    next.clear();
    bool none_left = false;
    std::vector<uint32_t> ev_rows;
    const uint32_t reps_ = comp_rows ? comp_rows : NRQ_MULTI_INACT;
    for (uint32_t rep = 0; rep < reps_; rep++) {
      uint32_t best = M, bestc = 0xFFFFFFFFu;
      for (uint32_t r = 0; r < M; r++)
        if (!assigned[r] && cnt[r] >= 2 && cnt[r] < bestc) {
          best = r; bestc = cnt[r];
          if (bestc == 2) break;
        }
      if (best == M) { none_left = (rep == 0); break; }
      if (comp_rows && bestc == 2) {
        if (rep == 0) { /* the components of the two-column rows as the event finds them: union-find over the columns */
          ev_rows.clear();
          if (par.empty()) { /* first event: the rows that have two columns now; the others join as they get there (drop_column) */
            par.resize(W); sz.assign(W, 1); seen.assign(W, 0); comp_of.assign(W, 0); edge_a.assign(M, 0xFFFFFFFFu); edge_b.assign(M, 0xFFFFFFFFu);
            for (uint32_t c = 0; c < W; c++) par[c] = c;
            for (uint32_t r = 0; r < M; r++) if (cnt[r] == 2) two_rows.push_back(r);
            track_two = true;
          }
          auto find = [&](uint32_t x) { while (par[x] != x) { par[x] = par[par[x]]; x = par[x]; } return x; };
          size_t keepn = 0;
          touched.clear();
          for (size_t i = 0; i < two_rows.size(); i++) {
            const uint32_t r = two_rows[i];
            if (assigned[r] || cnt[r] != 2) continue; /* (left the list for good: a row only loses columns) */
            two_rows[keepn++] = r;
            /* (a row keeps its last two V columns for as long as it has two: found once, by a walk of the row) */
            if (edge_a[r] == 0xFFFFFFFFu) {
              for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++)
                if (cstate[cidx[e]] == IN_V) { if (edge_a[r] == 0xFFFFFFFFu) edge_a[r] = cidx[e]; else edge_b[r] = cidx[e]; }
            }
            const uint32_t a = edge_a[r], b = edge_b[r];
            if (b == 0xFFFFFFFFu) continue;
            touched.push_back(a); touched.push_back(b);
            uint32_t ra = find(a), rb = find(b);
            if (ra != rb) { if (sz[ra] < sz[rb]) std::swap(ra, rb); par[rb] = ra; sz[ra] += sz[rb]; }
          }
          two_rows.resize(keepn);
          std::vector<std::pair<uint32_t, uint32_t>> comps; /* (columns, a row) per component */
          for (uint32_t r : two_rows) { /* (a component's row: the one with the lowest number, whatever order the list is in) */
            if (edge_b[r] == 0xFFFFFFFFu) continue;
            const uint32_t rt = find(edge_a[r]);
            if (!seen[rt]) { seen[rt] = 1; comp_of[rt] = (uint32_t)comps.size(); comps.push_back({sz[rt], r}); }
            else if (r < comps[comp_of[rt]].second) comps[comp_of[rt]].second = r;
          }
          const size_t top = std::min<size_t>(reps_, comps.size());
          std::partial_sort(comps.begin(), comps.begin() + top, comps.end(),
                            [](const std::pair<uint32_t, uint32_t> &x, const std::pair<uint32_t, uint32_t> &y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
          for (size_t i = 0; i < top; i++) ev_rows.push_back(comps[i].second);
          for (uint32_t c : touched) { par[c] = c; sz[c] = 1; seen[c] = 0; } /* (for the next event) */
        }
        if (rep >= ev_rows.size()) break;
        best = ev_rows[rep];
        if (assigned[best] || cnt[best] != 2) continue;
      }
      uint32_t keep = 0xFFFFFFFFu, keepdeg = 0xFFFFFFFFu;
      for (uint32_t e = rptr[best]; e < rptr[best + 1]; e++) {
        uint32_t c = cidx[e];
        if (cstate[c] != IN_V) continue;
        uint32_t dg = cptr[c + 1] - cptr[c];
        if (dg < keepdeg) { keepdeg = dg; keep = c; }
      }
      for (uint32_t e = rptr[best]; e < rptr[best + 1]; e++) {
        uint32_t c = cidx[e];
        if (cstate[c] != IN_V || c == keep) continue;
        cstate[c] = INACTIVE; inact_order.push_back(c); nV--;
        drop_column(c);
      }
    }
    if (none_left) { /* nothing left that touches V: every remaining V column is inactivated */
      for (uint32_t c = 0; c < W; c++)
        if (cstate[c] == IN_V) { cstate[c] = INACTIVE; inact_order.push_back(c); nV--; }
      break;
    }
    frontier.swap(next);
  }
  const uint32_t npiv = (uint32_t)pivslot.size();
  const uint32_t u = L - npiv;

  /* inactive-column numbering: the P permanently inactive columns first, then by inactivation time */
  std::vector<uint32_t> xof(L, 0xFFFFFFFFu);
  std::vector<uint16_t> ucol;
  ucol.reserve(u);
  for (uint32_t c = W; c < L; c++) { xof[c] = (uint32_t)ucol.size(); ucol.push_back((uint16_t)c); }
  for (uint32_t c : inact_order) { xof[c] = (uint32_t)ucol.size(); ucol.push_back((uint16_t)c); }
  if (ucol.size() != u) return -4;

  /* ---- W = X^-1 * A_top,U (bit rows, pivot order is a topological order) ---- */
  const uint32_t wpr = std::max(1u, (u + 31u) / 32u);
  std::vector<uint32_t> Wm((size_t)npiv * wpr, 0);
  for (uint32_t k = 0; k < npiv; k++) {
    uint32_t r = pivslot[k];
    uint32_t *wk = &Wm[(size_t)k * wpr];
    for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++) {
      uint32_t c = cidx[e];
      if (cstate[c] == INACTIVE) flip(wk, xof[c]);
      else if (c != pivcol[k]) {
        const uint32_t *wj = &Wm[(size_t)kof[c] * wpr];
        for (uint32_t w = 0; w < wpr; w++) wk[w] ^= wj[w];
      }
    }
  }

  /* ---- leftover binary rows and their reduced coefficient rows over the inactive columns ---- */
  std::vector<uint16_t> lowslot;
  for (uint32_t r = 0; r < M; r++)
    if (!assigned[r] && !(r >= S && r < S + H)) lowslot.push_back((uint16_t)r);
  const uint32_t nlow = (uint32_t)lowslot.size();
  const uint32_t lpr = std::max(1u, (nlow + 31u) / 32u);
  const uint32_t rowlen = wpr + lpr;
  std::vector<uint32_t> Mb((size_t)nlow * rowlen, 0);
  for (uint32_t j = 0; j < nlow; j++) {
    uint32_t r = lowslot[j];
    uint32_t *mj = &Mb[(size_t)j * rowlen];
    for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++) {
      uint32_t c = cidx[e];
      if (cstate[c] == INACTIVE) flip(mj, xof[c]);
      else {
        const uint32_t *wj = &Wm[(size_t)kof[c] * wpr];
        for (uint32_t w = 0; w < wpr; w++) mj[w] ^= wj[w];
      }
    }
    flip(mj + wpr, j); /* augmented identity: which original low rows are summed into this one */
  }

  /* ---- XOR op stream: pivot pass by level, then the low-row pass ---- */
  std::vector<uint32_t> ops;
  ops.reserve(((size_t)cidx.size() / NRQ_ROW + (size_t)(NRQ_PIPE + 1u) * (nlev + 2u) + NRQ_PAD_ROWS + 16u) * NRQ_ROW + (size_t)u * nlow / 2u); /* (no regrowth: ops, level spacers, padding) */
  uint32_t n_real_ops = 0;
  /* (whole rows of padding ops; ops.size() is a multiple of NRQ_ROW wherever this is called, so lane l of a row is NRQ_NOP_AT(l).
   * One resize and a fill: the 246 rows behind the stream were 15.7 k push_backs, a third of a K=100 plan's build time) */
  auto pad_rows = [&](uint32_t nrows_) {
    static const struct NopRow { uint32_t w[NRQ_ROW]; NopRow() { for (uint32_t l = 0; l < NRQ_ROW; l++) w[l] = NRQ_NOP_AT(l); } } nop_row;
    const size_t n0 = ops.size();
    ops.resize(n0 + (size_t)nrows_ * NRQ_ROW);
    for (size_t k = n0; k < ops.size(); k += NRQ_ROW) memcpy(&ops[k], nop_row.w, sizeof(nop_row.w));
  };
  /* place one group (plan.h): `fin` ops that complete rows other groups may read next, then `early` ops whose
   * targets are read later; the last NRQ_PIPE-1 rows of a group hold no finishing op */
  /* Lanes inside a group are chosen for the LDS banks (plan.h "lane placement"): of every 16 consecutive lanes, two per
   * class of the target slot (slot mod 8).  An op of class d with rank r among the group's ops of that class -- finishing
   * ops first -- takes lane 2d + (r & 1) of 16-lane block r >> 1; what a class has beyond two per block fills the lanes
   * other classes leave empty, class by class. */
  auto place_group = [&](std::vector<uint32_t> &fin, std::vector<uint32_t> &early) {
    /* (an empty group still takes its NRQ_PIPE-1 rows: early ops in the group before it may complete rows that the
     * group after it reads) */
    n_real_ops += (uint32_t)(fin.size() + early.size());
    uint32_t cf[NRQ_LANE_CLASSES] = {0}, ct[NRQ_LANE_CLASSES] = {0};
    for (uint32_t w : fin) { cf[nrq_op_class(w)]++; ct[nrq_op_class(w)]++; }
    for (uint32_t w : early) ct[nrq_op_class(w)]++;
    const uint32_t span = nrq_group_span((uint32_t)(fin.size() + early.size()), cf);
    const size_t at = ops.size();
    pad_rows(span);
    uint32_t rank[NRQ_LANE_CLASSES] = {0};
    for (int part = 0; part < 2; part++)
      for (uint32_t w : part ? early : fin) {
        const uint32_t d = nrq_op_class(w);
        ops[at + nrq_lane_place(span, ct, d, rank[d]++)] = w;
      }
  };
  {
    /* group of an op dst <- src: the dst's level t if src sits on the level right below (a "finishing" op); else any group of
     * its window [level(src)+1, t-1] ("early" op: fills lanes that thin levels leave empty).  An early op with a narrow
     * window takes a group of it by hash.  The others -- most: the median window is ~200 levels -- are dealt out in the order
     * of their release (plan.h "early ops by release"): a group takes as many of the waiting ones as its rows have lanes left,
     * and whole extra rows only when the groups behind it could not hold the rest; picked by hash alone the groups' loads
     * scatter around their row boundaries and a quarter of the stream was padding (1086 rows for 55.6 k ops at K=8192,
     * 870 are needed, this gives ~900). */
    std::vector<std::vector<uint32_t>> fin(nlev + 1), early(nlev + 1);
    struct WideOp { uint32_t lo, t, word; };
    std::vector<WideOp> wide;
    auto add_row = [&](uint32_t r, uint32_t own, uint32_t t) {
      for (uint32_t e = rptr[r]; e < rptr[r + 1]; e++) {
        const uint32_t col = cidx[e];
        if (cstate[col] != PIVOT || col == own) continue;
        const uint32_t src = owner[col], lo = level[src] + 1u;
        if (lo >= t) { fin[t].push_back(NRQ_OP(r, src)); continue; }
        if (t - lo > NRQ_EARLY_NARROW) { wide.push_back(WideOp{lo, t, NRQ_OP(r, src)}); continue; }
        uint32_t h = r * 0x9E3779B1u ^ col * 0x85EBCA6Bu;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        early[lo + h % (t - lo)].push_back(NRQ_OP(r, src));
      }
    };
    for (uint32_t k = 0; k < npiv; k++)
      if (level[pivslot[k]] > 0) add_row(pivslot[k], pivcol[k], level[pivslot[k]]); /* level 0 rows have nothing to gather */
    for (uint32_t j = 0; j < nlow; j++) add_row(lowslot[j], 0xFFFFFFFFu, nlev);
    if (!wide.empty()) {
      std::stable_sort(wide.begin(), wide.end(), [](const WideOp &a, const WideOp &b) { return a.lo < b.lo; });
      /* lanes a group has left at its minimal span, and what the groups behind it have */
      std::vector<uint32_t> cap(nlev + 2, 0), rel(nlev + 2, 0);
      std::vector<uint64_t> behind(nlev + 3, 0);
      for (uint32_t l = 1; l <= nlev; l++) {
        uint32_t cf[NRQ_LANE_CLASSES] = {0};
        for (uint32_t w : fin[l]) cf[nrq_op_class(w)]++;
        const uint32_t fixed = (uint32_t)(fin[l].size() + early[l].size());
        cap[l] = nrq_group_span(fixed, cf) * NRQ_ROW - fixed;
      }
      for (uint32_t l = nlev; l >= 1; l--) behind[l] = behind[l + 1] + cap[l];
      for (const WideOp &o : wide) rel[o.lo]++;
      uint64_t backlog = 0, unreleased = wide.size();
      size_t next = 0;
      for (uint32_t g = 1; g <= nlev; g++) {
        backlog += rel[g]; unreleased -= rel[g];
        uint64_t c = cap[g];
        while (backlog >= c + NRQ_ROW && backlog + unreleased - c > behind[g + 1]) c += NRQ_ROW; /* whole extra rows */
        uint64_t take = backlog < c ? backlog : c;
        backlog -= take;
        for (; take; take--, next++) {
          const WideOp &o = wide[next];
          early[g < o.t ? g : o.t - 1u].push_back(o.word); /* (served after its window: its last group takes it) */
        }
      }
      for (; next < wide.size(); next++) early[wide[next].t - 1u].push_back(wide[next].word);
    }
    for (uint32_t l = 1; l <= nlev; l++) place_group(fin[l], early[l]);
  }

  /* ---- HDPC rows over the inactive columns: Mh = G_U ^ G_left * W ---- */
  std::vector<uint8_t> Mh((size_t)H * u, 0);
  for (uint32_t x = 0; x < u; x++) {
    uint32_t c = ucol[x];
    for (uint32_t h = 0; h < H; h++)
      Mh[(size_t)h * u + x] = (c < n_hd) ? G[(size_t)h * n_hd + c] : (uint8_t)(c - n_hd == h);
  }
  { /* G_left * W with the HDPC block by COLUMN (kconst off_gt: 16 bytes per column, rows >= H zero): one 16-byte XOR per set
     * bit of W into MhT[x], transposed into Mh at the end (H single bytes at stride u per set bit were 3.5 of the 11.7 ms of a
     * K'=8192 build) */
    const uint8_t *GT = kconst + kh->off_gt;
    std::vector<uint64_t> MhT((size_t)u * 2u, 0);
    for (uint32_t k = 0; k < npiv; k++) {
      const uint32_t *wk = &Wm[(size_t)k * wpr];
      uint64_t g0, g1;
      memcpy(&g0, GT + (size_t)pivcol[k] * 16u, 8); memcpy(&g1, GT + (size_t)pivcol[k] * 16u + 8u, 8);
      for (uint32_t w = 0; w < wpr; w++) {
        uint32_t bits = wk[w];
        while (bits) {
          const uint32_t x = w * 32 + (uint32_t)__builtin_ctz(bits);
          bits &= bits - 1;
          MhT[(size_t)x * 2u] ^= g0; MhT[(size_t)x * 2u + 1u] ^= g1;
        }
      }
    }
    const uint8_t *mt = reinterpret_cast<const uint8_t *>(MhT.data());
    for (uint32_t x = 0; x < u; x++)
      for (uint32_t h = 0; h < H; h++) Mh[(size_t)h * u + x] ^= mt[(size_t)x * 16u + h];
  }

  /* ---- GF(2) Gauss-Jordan on the leftover binary rows ---- */
  std::vector<uint8_t> used(nlow, 0);
  std::vector<uint32_t> red_row, red_x, freex;
  for (uint32_t x = 0; x < u; x++) {
    uint32_t pr = nlow;
    for (uint32_t j = 0; j < nlow; j++)
      if (!used[j] && bit(&Mb[(size_t)j * rowlen], x)) { pr = j; break; }
    if (pr == nlow) { freex.push_back(x); continue; }
    used[pr] = 1;
    const uint32_t *src = &Mb[(size_t)pr * rowlen];
    for (uint32_t j = 0; j < nlow; j++) {
      if (j == pr) continue;
      uint32_t *dst = &Mb[(size_t)j * rowlen];
      if (!bit(dst, x)) continue;
      for (uint32_t w = 0; w < rowlen; w++) dst[w] ^= src[w];
    }
    red_row.push_back(pr);
    red_x.push_back(x);
  }
  const uint32_t r2 = (uint32_t)red_row.size(), nfree = (uint32_t)freex.size();
  /* GF(2) combinations E_p (slot M+p) = XOR of the leftover rows named by the augmented part of reduced row p: ops of the
   * stream for small blocks, a bit matrix for big ones (plan.h NRQ_AUG_MATRIX_MIN_L) */
  if (M + r2 + NRQ_SCRATCH > 65535u) return -3;
  const bool aug_in_stream = L < NRQ_AUG_MATRIX_MIN_L;
  const uint32_t aug_stride = std::max(4u, (r2 + 3u) & ~3u);
  std::vector<uint32_t> augt(aug_in_stream ? 0 : (size_t)lpr * aug_stride, 0);
  if (aug_in_stream) {
    /* round-robin over p so that neighbouring ops hit different targets */
    std::vector<uint32_t> g, curj(r2, 0);
    bool any = r2 > 0;
    while (any) {
      any = false;
      for (uint32_t q = 0; q < r2; q++) {
        const uint32_t *aug = &Mb[(size_t)red_row[q] * rowlen + wpr];
        uint32_t &j = curj[q];
        while (j < nlow) { /* next set bit from j on, a word at a time */
          const uint32_t w = aug[j >> 5] >> (j & 31u);
          if (w) { j += (uint32_t)__builtin_ctz(w); break; }
          j = (j | 31u) + 1u;
        }
        if (j < nlow) {
          g.push_back(NRQ_OP(M + q, lowslot[j]));
          j++; any = true;
        }
      }
    }
    std::vector<uint32_t> none;
    place_group(g, none);
  } else {
    for (uint32_t q = 0; q < r2; q++) {
      const uint32_t *aug = &Mb[(size_t)red_row[q] * rowlen + wpr];
      for (uint32_t w = 0; w < lpr; w++) {
        augt[(size_t)w * aug_stride + q] = aug[w];
        n_real_ops += (uint32_t)__builtin_popcount(aug[w]);
      }
    }
  }
  const uint32_t op_rows = (uint32_t)(ops.size() / NRQ_ROW);
  pad_rows(NRQ_PAD_ROWS);
  uint32_t status = 0;
  if (nfree > H || nfree > NRQ_MAX_FREE) status = 1;

  /* ---- the free columns: H x nfree system over GF(256) from the HDPC rows ---- */
  std::vector<uint8_t> mh((size_t)H * std::max(1u, r2), 0), hinv((size_t)std::max(1u, nfree) * H, 0);
  std::vector<uint32_t> fbits(std::max(1u, r2), 0);
  if (!status) {
    for (uint32_t q = 0; q < r2; q++) {
      const uint32_t *row = &Mb[(size_t)red_row[q] * rowlen];
      uint32_t fb = 0;
      for (uint32_t f = 0; f < nfree; f++)
        if (bit(row, freex[f])) fb |= 1u << f;
      fbits[q] = fb;
      for (uint32_t h = 0; h < H; h++) mh[(size_t)h * r2 + q] = Mh[(size_t)h * u + red_x[q]];
    }
    /* aug = [ Mh restricted to free columns, with the pivot columns folded in | I_H ] */
    const uint32_t aw = nfree + H;
    std::vector<uint8_t> aug((size_t)H * aw, 0);
    for (uint32_t h = 0; h < H; h++) {
      for (uint32_t f = 0; f < nfree; f++) {
        uint8_t v = Mh[(size_t)h * u + freex[f]];
        for (uint32_t q = 0; q < r2; q++)
          if ((fbits[q] >> f) & 1u) v ^= mh[(size_t)h * r2 + q];
        aug[(size_t)h * aw + f] = v;
      }
      aug[(size_t)h * aw + nfree + h] = 1;
    }
    std::vector<uint8_t> taken(H, 0), solver(std::max(1u, nfree), 0);
    for (uint32_t f = 0; f < nfree && !status; f++) {
      uint32_t pr = H;
      for (uint32_t h = 0; h < H; h++)
        if (!taken[h] && aug[(size_t)h * aw + f]) { pr = h; break; }
      if (pr == H) { status = 1; break; }
      taken[pr] = 1;
      solver[f] = (uint8_t)pr;
      uint8_t iv = gf.inv[aug[(size_t)pr * aw + f]];
      for (uint32_t w = 0; w < aw; w++) aug[(size_t)pr * aw + w] = gf.mul(aug[(size_t)pr * aw + w], iv);
      for (uint32_t h = 0; h < H; h++) {
        if (h == pr) continue;
        uint8_t b = aug[(size_t)h * aw + f];
        if (!b) continue;
        for (uint32_t w = 0; w < aw; w++) aug[(size_t)h * aw + w] ^= gf.mul(b, aug[(size_t)pr * aw + w]);
      }
    }
    if (!status)
      for (uint32_t f = 0; f < nfree; f++)
        for (uint32_t h = 0; h < H; h++) hinv[(size_t)f * H + h] = aug[(size_t)solver[f] * aw + nfree + h];
  }

  /* ---- homes of the intermediate symbols ---- */
  std::vector<uint16_t> colslot(L, NRQ_NOSLOT), pivof(n_hd, NRQ_NOSLOT), uslot(u, NRQ_NOSLOT);
  for (uint32_t k = 0; k < npiv; k++) {
    colslot[pivcol[k]] = pivslot[k];
    pivof[pivcol[k]] = pivslot[k];
  }
  {
    uint32_t x = 0;
    for (uint32_t r = 0; r < M && x < u; r++)
      if (!assigned[r]) { uslot[x] = (uint16_t)r; colslot[ucol[x]] = (uint16_t)r; x++; }
    if (x != u) return -5;
  }

  /* ---- serialise ---- */
  Arena A;
  nrq_plan_hdr hd;
  memset(&hd, 0, sizeof(hd));
  A.buf.reserve(ops.size() * 4u + (size_t)wpr * ((npiv + 63u) & ~63u) * 4u + (size_t)L * 16u + (size_t)augt.size() * 4u + 8192u); /* (no regrowth on the way) */
  A.reserve((uint32_t)sizeof(hd));
  hd.magic = NRQ_PLAN_MAGIC; hd.status = status;
  hd.K = K; hd.Kp = p.Kp; hd.J = p.J; hd.S = S; hd.H = H; hd.W = W; hd.L = L; hd.P = p.P; hd.P1 = p.P1; hd.B = p.B;
  hd.M = M; hd.npiv = npiv; hd.u = u; hd.nlow = nlow; hd.r2 = r2; hd.nfree = nfree; hd.nlev = nlev;
  hd.nrows = op_rows; hd.pipe = NRQ_PIPE; hd.wpr = wpr;
  hd.npiv_pad = (npiv + 63u) & ~63u;
  hd.n_xor_ops = n_real_ops;
  { /* the stream goes out quad-interleaved (plan.h: NRQ_OP_INDEX), whole quads of rows */
    while ((ops.size() / NRQ_ROW) % 4u) pad_rows(1);
    hd.off_ops = A.reserve((uint32_t)(ops.size() * 4));
    uint32_t *out = A.at<uint32_t>(hd.off_ops);
    const size_t nquads = ops.size() / (4u * NRQ_ROW);
    for (size_t g = 0; g < nquads; g++) { /* out[NRQ_OP_INDEX(row, lane)] = ops[row * NRQ_ROW + lane], a quad of rows at a time */
      const uint32_t *r0 = &ops[g * 4u * NRQ_ROW];
      uint32_t *o = out + g * 4u * NRQ_ROW;
      for (uint32_t l = 0; l < NRQ_ROW; l++) {
        o[4u * l] = r0[l]; o[4u * l + 1u] = r0[NRQ_ROW + l]; o[4u * l + 2u] = r0[2u * NRQ_ROW + l]; o[4u * l + 3u] = r0[3u * NRQ_ROW + l];
      }
    }
  }
  hd.off_pivslot = A.reserve(npiv * 2);
  memcpy(A.at<uint8_t>(hd.off_pivslot), pivslot.data(), (size_t)npiv * 2);
  hd.off_pivcol = A.reserve(npiv * 2);
  memcpy(A.at<uint8_t>(hd.off_pivcol), pivcol.data(), (size_t)npiv * 2);
  hd.off_wt = A.reserve(wpr * hd.npiv_pad * 4);
  {
    uint32_t *wt = A.at<uint32_t>(hd.off_wt);
    for (uint32_t k = 0; k < npiv; k++)
      for (uint32_t w = 0; w < wpr; w++) wt[(size_t)w * hd.npiv_pad + k] = Wm[(size_t)k * wpr + w];
  }
  hd.off_lowslot = A.reserve(std::max(1u, nlow) * 2);
  if (nlow) memcpy(A.at<uint8_t>(hd.off_lowslot), lowslot.data(), (size_t)nlow * 2);
  hd.off_pivx = A.reserve(std::max(1u, r2) * 2);
  for (uint32_t q = 0; q < r2; q++) A.at<uint16_t>(hd.off_pivx)[q] = (uint16_t)red_x[q];
  hd.off_fbits = A.reserve(std::max(1u, r2) * 4);
  memcpy(A.at<uint8_t>(hd.off_fbits), fbits.data(), (size_t)std::max(1u, r2) * 4);
  hd.off_mh = A.reserve(H * std::max(1u, r2));
  memcpy(A.at<uint8_t>(hd.off_mh), mh.data(), (size_t)H * std::max(1u, r2));
  hd.off_freex = A.reserve(std::max(1u, nfree) * 2);
  for (uint32_t f = 0; f < nfree; f++) A.at<uint16_t>(hd.off_freex)[f] = (uint16_t)freex[f];
  hd.off_hinv = A.reserve(std::max(1u, nfree) * H);
  memcpy(A.at<uint8_t>(hd.off_hinv), hinv.data(), (size_t)std::max(1u, nfree) * H);
  hd.off_colslot = A.reserve(L * 2);
  memcpy(A.at<uint8_t>(hd.off_colslot), colslot.data(), (size_t)L * 2);
  hd.off_pivof = A.reserve(n_hd * 2);
  memcpy(A.at<uint8_t>(hd.off_pivof), pivof.data(), (size_t)n_hd * 2);
  hd.off_uslot = A.reserve(std::max(1u, u) * 2);
  if (u) memcpy(A.at<uint8_t>(hd.off_uslot), uslot.data(), (size_t)u * 2);
  hd.lpr = aug_in_stream ? 0u : lpr; hd.aug_stride = aug_stride;
  hd.off_augt = A.reserve((uint32_t)(augt.size() * 4 + 16));
  if (!augt.empty()) memcpy(A.at<uint8_t>(hd.off_augt), augt.data(), augt.size() * 4);
  hd.total_bytes = align16((uint32_t)A.buf.size());
  A.buf.resize(hd.total_bytes, 0);
  memcpy(A.buf.data(), &hd, sizeof(hd));

  uint8_t *res = (uint8_t *)malloc(hd.total_bytes);
  if (!res) return -2;
  memcpy(res, A.buf.data(), hd.total_bytes);
  *out = res;
  *out_bytes = hd.total_bytes;
  return 0;
}

/* The plan of an ENCODER (no symbol missing: rows are ISI 0 .. K'-1) is walked by every strip of every block of its object, and
 * it is built on the host beside the GPU's work: it can afford RFC 6330's component rule.  One row an event ends at the fewest
 * inactive columns (K'=8192: 184 against 209 first-found, 197 with six rows an event) but at more levels and stream rows, six
 * rows an event at the shortest stream; measured on the encode solve per 256 blocks (first-found / 6 / 3 / 1 rows an event):
 * K=500 10.99 / 9.77 / 10.15 / 10.30 ms, K=1000 8.53 / 8.52 / 8.36 / 8.50, K=2000 7.18 / 7.16 / 7.15 / 6.98, K=5000 4.55 / 4.25 /
 * 4.37 / 4.37, K=8192 6.48 / 6.42 / 6.42 / 6.25, K=10000 12.31 / 12.09 / 11.80 / 11.61 -- so: one row from K' = 1500 on, six
 * below.  (Building all four and keeping the cheapest by a fitted cost was tried: four builds a step no longer hide behind the
 * GPU.)  The build takes ~1.5 x the first-found one.  A decoder's plan is used once: it keeps the first-found rule, which is
 * also what the device planner implements.  NRQ_HOST_WAY=n forces n rows an event (0: first-found) for measurements. */
extern "C" int nrq_host_plan_build(uint32_t K, uint32_t nrows, const uint32_t *isis, const uint8_t *kconst, uint8_t **out, uint32_t *out_bytes) {
  rq_params p;
  if (!rq_params_init(K, &p)) return -1;
  bool encoder = nrows == p.Kp;
  for (uint32_t k = 0; encoder && k < nrows; k++) encoder = isis[k] == k;
  uint32_t way = !encoder ? 0u : p.Kp >= 1500u ? 1u : 6u;
  if (const char *e = getenv("NRQ_HOST_WAY")) { if (encoder) way = (uint32_t)atoi(e); }
  return host_plan_build_with(K, nrows, isis, kconst, out, out_bytes, way);
}
