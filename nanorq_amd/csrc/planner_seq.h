/*
 * planner_seq.h -- the phase sequence of the GPU planner, shared verbatim by the HIP kernel and the
 * CPU emulator.  The includer defines
 *     PL_PHASE(fn)        run phase fn(c, tid, PL_NT) on every thread of the workgroup, then barrier
 *     PL_PHASE1(fn, a)    same with one extra leading argument
 *     PL_WFAST_RUN(wb)    run the op-stream rows [0, pl_wfast_rows(c)) on the wb-byte slot image at the start of
 *                         the dynamic LDS region (one wave), then barrier
 *     PL_SEG              which part to run: 0 = everything (small and medium blocks, the emulator);
 *                         1 = up to the op stream, 2 = from the leftover rows' coefficients on; 3 + 4 = part 1 cut once more, in
 *                         front of the entry pass over the constraint matrix (pl_w_init), which nrq_wentry_kernel then runs on
 *                         several workgroups: 3 | nrq_wentry_kernel | 4 | nrq_wpass_kernel | nrq_mh_kernel | 2 | nrq_wt_kernel.  Big blocks (peeling state
 *                         in HBM) run as 1 | nrq_wpass_kernel | nrq_mh_kernel | 2 | nrq_wt_kernel: the W pass (the op stream
 *                         on 2-byte strips of the bit rows, a workgroup per strip), the HDPC fold over the pivots and the
 *                         transposition of W are parallel work and take 40 % of a one-workgroup planner at K'=56403, so many
 *                         workgroups do them between the two parts (pl_shared travels through the block's workspace)
 *     PL_PHASE1_CLAIM(fn, a)  PL_PHASE1 for pl_round_claim: what that phase writes to HBM is read after peeling only (when the
 *                         peeling decisions are taken in LDS), so the barrier behind it need not wait for those stores
 *     PL_STEER_SYNC       a workgroup barrier (nothing in the emulator)
 *     PL_NT_              threads of the workgroup
 *     PL_Z                the instance of the phase functions the includer runs (kernel: its PK template argument; emulator: 1)
 * and provides `PlanCtx c`.  Every value that steers control flow is read from workgroup-shared state right after a
 * barrier -- and where the phase that follows may CHANGE that value (peeling counts, `best`, `status`), every thread
 * reads it into a local first and PL_STEER_SYNC separates the reads from that phase: without it a wave that is late
 * to the read sees what an early wave has already written, takes the other branch, and the workgroup's barriers pair
 * up across different phases from there on (seen as rare corrupt plans with four waves, every fifth plan with sixteen).
 */
{
  pl_shared *sh_ = c.sh;
  if (PL_SEG != 2 && PL_SEG != 4) {
  PL_PHASE(pl_init_a);
  PL_PHASE(pl_init_b);
  PL_PHASE(pl_scan_a);
  PL_PHASE(pl_scan_b);
  PL_PHASE(pl_scan_c);
  PL_PHASE(pl_pcsc_fill);
  /* ---- peeling ---- */
  {
    const uint32_t guard_max = 4u * c.p.L + 64u;
    uint32_t rd_ = 0;
    for (;;) {
      /* (one LDS round trip for the three words, not one per test: a round is a chain of such trips) */
      PL_ST(c, 0);
      const uint32_t st_ = sh_->status, nv_ = sh_->nV, nq_ = sh_->nq[rd_ & 1u];
      if (st_ + nv_ + nq_ != 0x12345u) PL_ST(c, 1);
      PL_STEER_SYNC; /* (pl_round_claim lowers nV and may raise status) */
      PL_ST(c, 2);
      if (st_ != 0 || nv_ == 0 || rd_ >= guard_max) break;
#ifdef PL_TRACE_ROUND
      PL_TRACE_ROUND(nq_, nv_); /* (the emulator's trace of the peel: frontier width and open columns, round by round) */
#endif
      if (PL_LIKELY(nq_ > 0)) {
        PL_PHASE1_CLAIM(pl_round_claim, rd_);
        PL_ST(c, 8);
        PL_PHASE1(pl_round_drop, rd_);
        if (pl_chained(PL_Z)) PL_PHASE1(pl_round_chain_end, rd_); /* (the chained form leaves the books to one thread behind its barrier) */
        PL_ST(c, 15);
      } else if (pl_peel_in_lds(c)) {
        /* one inactivation event, peeling state in LDS: up to NRQ_MULTI_INACT open rows with two V columns, all at once */
        PL_PHASE1(pl_event_scan, rd_);
        PL_PHASE1(pl_event_pick, rd_);
        PL_PHASE1(pl_event_drop, rd_);
      } else {
        /* one inactivation event: up to NRQ_MULTI_INACT open rows with two V columns from the top of the stack, all at once
         * (pl_inact_find lists them, pl_inact_apply_a gives a thread to each); the sparsest open row if there is none */
        for (uint32_t rep_ = 0; rep_ < 1u; rep_++) {
          const uint32_t a_ = rd_ | (rep_ << 24);
          if (rep_) PL_PHASE1(pl_inact_next, a_);
          /* peeling state in LDS: all open rows are scanned (cheap there, and ties go to the lowest row); in HBM: the
           * top of the stack of two-column rows, a chunk at a time, all rows only if it runs empty */
          while (!pl_peel_in_lds(c)) {
            PL_PHASE1(pl_inact_find, a_);
            PL_PHASE1(pl_inact_find_c, a_);
            const bool found_ = sh_->best != PL_NONE || sh_->ncand[0] == 0u;
            PL_STEER_SYNC; /* (the next search writes best) */
            if (found_) break;
          }
          const bool scan_ = sh_->best == PL_NONE;
          PL_STEER_SYNC; /* (pl_inact_find_b writes best) */
          if (scan_) PL_PHASE1(pl_inact_find_b, a_);
          PL_PHASE1(pl_inact_apply_a, a_);
          PL_PHASE1(pl_inact_apply_b, a_);
          const bool last_ = sh_->tmp1 || sh_->status != 0 || sh_->nV == 0;
          PL_STEER_SYNC;
          if (last_) break;
        }
      }
      rd_++;
    }
  }
  PL_PHASE(pl_lev_0);
  if (pl_chained(PL_Z)) PL_PHASE(pl_pivot_sort); /* (the chain numbers the pivots as it reaches them: back to level order) */
  PL_PHASE(pl_lev_a);
  const bool peeled_ = sh_->status == 0 && sh_->nV == 0;
  PL_STEER_SYNC; /* (later phases raise status when a capacity is exceeded) */
  if (peeled_) {
#ifdef PL_SABOTAGE_HOOK
    PL_SABOTAGE_HOOK; /* (the emulator's tests damage the peel's books here, to see the check find it) */
#endif
    PL_PHASE(pl_check_a); /* the peel's books describe a permutation, or the block goes to the host planner (planner_body.h) */
    PL_PHASE(pl_check_b);
    PL_PHASE(pl_check_c);
    PL_PHASE(pl_lev_b);
    PL_PHASE(pl_low_a);
    PL_PHASE(pl_low_b);
  }
  } /* PL_SEG != 2, 4 */
  const uint32_t st2_ = sh_->status, nv2_ = sh_->nV;
  PL_STEER_SYNC;
  if (st2_ == 0 && nv2_ == 0) {
    if (PL_SEG != 2 && PL_SEG != 3) {
    if (PL_SEG == 4) PL_PHASE(pl_cls_fetch); /* (nrq_wentry_kernel ran the entry pass) */
    else PL_PHASE(pl_w_init);
    PL_PHASE(pl_w_init_b);
    PL_PHASE(pl_ops_layout_a);
    PL_PHASE(pl_ops_layout);
    PL_PHASE(pl_ops_clear);
    PL_PHASE(pl_ops_emit);
#ifdef NRQ_PLAN_SELFCHECK
    PL_PHASE(pl_ops_check_a);
    PL_PHASE(pl_ops_check_b);
    PL_PHASE(pl_ops_check_c);
#endif
    /* W = X^-1 * A_U and the leftover rows' reduced coefficients: the op stream run on bit rows -- on strips of
     * them in LDS by the solve kernel's row pipeline (PL_WFAST_RUN: wave 0 only; defined by the includer),
     * or, when no strip width fits, level by level on the HBM rows */
    if (PL_SEG == 0) { /* (a segmented run leaves the W pass to nrq_wpass_kernel: one workgroup per 2-byte strip of the W rows) */
      const uint32_t wb_ = sh_->status == 0 ? pl_wfast_wb(c) : 0u;
      PL_STEER_SYNC;
      if (wb_) {
        PL_PHASE(pl_wfast_spill);
        const uint32_t ns_ = (sh_->wpr * 4u + wb_ - 1u) / wb_;
        for (uint32_t s_ = 0; s_ < ns_; s_++) {
          PL_PHASE1(pl_wfast_load, s_);
          PL_WFAST_RUN(wb_);
          PL_PHASE1(pl_wfast_store, s_);
        }
        if (pl_mhrev_ok(c)) /* the HDPC fold's z = g^T X^-1 while the image is here: the stream once more, transposed and in reverse */
          for (uint32_t s_ = 0; s_ * wb_ < 16u; s_++) {
            PL_PHASE1(pl_mhrev_load, s_);
            PL_PHASE1(pl_mhrev_load_b, s_);
            PL_MHREV_RUN(wb_);
            PL_PHASE1(pl_mhrev_store, s_);
          }
        PL_PHASE(pl_wfast_restore);
      } else {
        PL_PHASE(pl_w_stage);
        for (uint32_t lv_ = 1; lv_ <= sh_->nlev; lv_++) PL_PHASE1(pl_w_group, lv_);
      }
    }
    } /* PL_SEG != 2, 3 */
    if (PL_SEG != 1 && PL_SEG != 3 && PL_SEG != 4) {
    if (PL_SEG == 0) {
      PL_PHASE(pl_mh_init);
      const bool rev_ = sh_->mhrev != 0u;
      PL_STEER_SYNC;
      if (rev_) PL_PHASE(pl_mhrev_scatter); /* (z is in the workspace: one pass over the inactive columns' row lists) */
      else
      for (uint32_t tl_ = 0; tl_ * PL_MH_TILE < sh_->npiv; tl_++) {
        PL_PHASE1(pl_mh_load, tl_);
        PL_PHASE1(pl_mh_acc, tl_);
      }
    } else {
      PL_PHASE(pl_mh_fetch); /* nrq_mh_kernel left MhT in the block's workspace */
    }
    PL_PHASE(pl_low_c); /* (after the fold: Mb takes the place of the fold's tiles) */
    {
      const uint32_t u_ = c.p.L - sh_->npiv;
      const bool wave_ = PL_GJ_WAVE && pl_gjw_ok(c); /* the 32 columns of a panel by one wave in one phase */
      if (wave_ || pl_gj_blocked(c, PL_NT_)) { /* a panel of 32 columns at a time: one pass over the matrix per panel, not per column */
        for (uint32_t w_ = 0; w_ * 32u < u_; w_++) {
          if (wave_) PL_PHASE1(pl_gjp_wave, w_); /* (sets up its own books, leaves the rows' masks in their final form) */
          else {
          PL_PHASE1(pl_gjp_init, w_);
          for (uint32_t x_ = w_ * 32u; x_ < u_ && x_ < w_ * 32u + 32u; x_++) {
            PL_PHASE1(pl_gjp_bid, x_);
            PL_PHASE1(pl_gjp_step, x_);
          }
          PL_PHASE1(pl_gjp_comb, w_);
          }
          PL_PHASE1(pl_gjp_stage, w_);
          PL_PHASE1(pl_gjp_apply, w_);
        }
      } else {
      if (u_) PL_PHASE1(pl_gj_a, 0u);
      for (uint32_t x_ = 0; x_ < u_; x_++) {
        const uint32_t a_ = x_ | (x_ + 1u < u_ ? 0x80000000u : 0u); /* step B also prepares the next column */
        PL_PHASE1(pl_gj_b, a_);
      }
      }
    }
    /* the free columns over GF(256); while that fails and the caller holds further symbols, add one row */
    for (uint32_t try_ = 0; try_ <= PL_EXTRA_ROWS + 1u; try_++) {
      PL_PHASE(pl_dense_a);
      PL_PHASE(pl_dense_b);
      const bool solve_ = sh_->status == 0 && sh_->dense_ok;
      const uint32_t nf_ = sh_->nfree;
      PL_STEER_SYNC;
      if (solve_) {
        for (uint32_t f_ = 0; f_ < nf_; f_++) {
          PL_PHASE1(pl_dense_step_a, f_);
          PL_PHASE1(pl_dense_step_b, f_);
        }
      }
      const bool done_ = sh_->status != 0 || sh_->dense_ok;
      PL_STEER_SYNC; /* (pl_extra_a may raise status) */
      if (done_) break;
      PL_PHASE(pl_extra_a);
      const bool stop_ = sh_->status != 0;
      PL_STEER_SYNC;
      if (stop_) break;
      PL_PHASE(pl_extra_b);
      PL_PHASE(pl_extra_c);
      PL_PHASE(pl_extra_d);
      const uint32_t xc_ = sh_->xcol;
      PL_STEER_SYNC;
      if (xc_ != PL_NONE) {
        PL_PHASE1(pl_gj_a, xc_);
        PL_PHASE1(pl_gj_b, xc_);
      }
    }
    PL_PHASE(pl_dense_c);
    PL_PHASE(pl_bin_a);
    PL_PHASE(pl_bin_b);
    PL_PHASE(pl_bin_c);
    PL_PHASE(pl_final_a);
    PL_PHASE(pl_final_b);
    PL_PHASE(pl_final_c);
    PL_PHASE(pl_final_c2);
    PL_PHASE(pl_final_c3);
    } /* PL_SEG != 1 */
  } else if (st2_ == 0 && PL_SEG != 1 && PL_SEG != 3 && PL_SEG != 4) {
    PL_PHASE(pl_mark_failed); /* peeling did not terminate: report the block as undecodable */
  }
  if (PL_SEG != 1 && PL_SEG != 3 && PL_SEG != 4) {
    PL_PHASE(pl_final_d);
    PL_PHASE(pl_final_e);
  }
}
