// Constraint solve stage: mj_fwdConstraint with the PGS (dual) solver, one wavefront per env.
//
// The PGS sweep is inherently sequential over constraint rows (Gauss-Seidel); what the wavefront
// parallelises is each row's residual b_i + AR_i . f.  The reference sums that dot product with
// four interleaved partial sums (mju_dot, engine_util_blas.c:493-527); lanes 0..3 each carry one of
// them and the wave combines them with three adds, so the sweep reproduces the CPU result bit for
// bit -- including the iteration at which `improvement < tolerance` fires -- while the residual
// costs ceil(nefc/4) dependent multiply-adds instead of nefc.
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)


// wave-cooperative dot product in mju_dot's association; result is wave-uniform.
// a and b may be written by other lanes before the call (caller syncs).
template <class P0, class P1>
MJH_DEV real wave_dot_ref(P0 a, P1 b, int n) {
#if MJH_LANE_MODE
  return dot_ref(a, b, n);
#else
  const int lane = wv_lane();
  const int n4 = n & ~3;
  real r = 0;
  if (lane < 4) {
    for (int i = lane; i < n4; i += 4) r += a[i]*b[i];
  }
  real r0 = wv_bcast(r, 0), r1 = wv_bcast(r, 1), r2 = wv_bcast(r, 2), r3 = wv_bcast(r, 3);
  real res = (r0 + r2) + (r1 + r3);
  int rem = n - n4;
  if (rem == 3) res += a[n4]*b[n4] + a[n4+1]*b[n4+1] + a[n4+2]*b[n4+2];
  else if (rem == 2) res += a[n4]*b[n4] + a[n4+1]*b[n4+1];
  else if (rem == 1) res += a[n4]*b[n4];
  return res;
#endif
}

// mju_dotSparse(row of efc_AR, b) when efc_AR is kept dense here: the row's structural pattern (nw 64-bit words, two
// ints each) names the entries the reference stores, in ascending order; four accumulators by POSITION in that
// compressed row, (r0 + r2) + (r1 + r3), the remainder one by one.  Lanes 0..3 take an accumulator each.
template <class P0, class P1>
MJH_DEV real wave_dot_masked(P0 a, P1 b, ciptr mw, int nw) {
#if MJH_LANE_MODE
  (void)a; (void)b; (void)mw; (void)nw;
  return 0;
#else
  const int lane = wv_lane();
  int nnz = 0;
  for (int w = 0; w < nw; w++) nnz += __builtin_popcount((unsigned)mw[2*w]) + __builtin_popcount((unsigned)mw[2*w + 1]);
  const int n4 = nnz & ~3;
  real r = 0;
  int tail[3] = {0, 0, 0};
  int k = 0;
  for (int w = 0; w < nw; w++) {
    unsigned long long m = ((unsigned long long)(unsigned)mw[2*w + 1] << 32) | (unsigned)mw[2*w];
    while (m) {
      const int j = 64*w + __builtin_ctzll(m);
      m &= m - 1;
      if (k < n4) { if ((k & 3) == lane) r += a[j]*b[j]; }
      else tail[k - n4] = j;
      k++;
    }
  }
  const real r0 = wv_bcast(r, 0), r1 = wv_bcast(r, 1), r2 = wv_bcast(r, 2), r3 = wv_bcast(r, 3);
  real res = (r0 + r2) + (r1 + r3);
  for (int t = 0; t < nnz - n4; t++) res += a[tail[t]]*b[tail[t]];
  return res;
#endif
}

// PCG32, engine_solver.c:241-254
struct Pcg32 { uint64_t state, inc; };
MJH_DEV uint32_t pcg32_next(Pcg32* rng) {
  uint64_t old = rng->state;
  rng->state = old * 6364136223846793005ULL + (rng->inc | 1);
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}

#if !MJH_LANE_MODE && MJH_W == 64
// ------------------------------------------------------------------------------------------------
// solPGS for nefc <= 64 with the iterate in registers           (engine_solver.c:457-741)
//
// Lane layout.  mju_dot sums a row product in four interleaved chains (j = c, c+4, c+8, ...), each
// strictly left to right, then (r0+r2)+(r1+r3), then the 1..3 tail products as one expression.
// Chain c is given DPP row c: constraint j < n4 lives in lane 16*(j&3) + (j>>2), tail constraint
// n4+t in lane 16*t + 15 (free because n4 <= 60 whenever a tail exists).  A chain sum is then
// L = n4/4 dependent adds, each fed by a row broadcast (v_mov_dpp row_newbcast:k) -- no LDS, no
// barrier -- and reproduces the reference's rounding exactly.
// Every lane keeps force, b, 1/AR_jj, frictionloss and the Nesterov history of its own constraint;
// the owner lane of the row being visited updates itself.  The visitation order of iteration k
// comes from the precomputed table M.pgs_order (it depends on (nefc, k) only).
// ------------------------------------------------------------------------------------------------
#define MJH_PGS_CHAIN_STEP(k) if (L > k) { acc = acc + wv_row_bcast<k>(p);
#define MJH_PGS_CHAIN_END }}}}}}}}}}}}}}}

// ARL = 1: AR is resident in LDS and is read with ds_read through a local-address-space pointer
// (in-order returns let the next row's prefetch stay in flight; a flat load would have to drain)
// LT >= 0: the chain length L = nefc/4 is a compile-time constant (the caller dispatches on it): the
// chain of a row update is then straight-line code -- no scalar branch per chain step, the DPP moves
// of a row scheduled ahead of the dependent adds.
// MJH_PGS_PREFETCH2 = 1: the exact sweeps request AR rows TWO visits ahead when AR sits in global memory (solve_pgs_fast<0>,
// solve_pgs_wide).  Measured and NOT the default (profiles/r06/negative_results.txt): three alternating pairs on one box, the
// humanoid's testspeed regime 2.83 M with it against 2.88 M without (-2 %), the driver configuration unchanged within noise --
// a visit is already longer than an L2 round trip at four wavefronts per SIMD, and the extra live row costs registers.
#ifndef MJH_PGS_PREFETCH2
#define MJH_PGS_PREFETCH2 0
#endif
template <int ARL, int LT = -1, int NT = -1>
MJH_DEVN_HOT void solve_pgs_fast(MREF M_, BREF B_, int e_) {
  const auto& M = wv_uniform_ref(M_);
  BREF B = B_;
  const int e = wv_uniform_i(e_);
  iptr counts = MJH_F(B, counts, e);
  const int n = wv_uniform_i(counts[MJH_C_NEFC]), ne = wv_uniform_i(counts[MJH_C_NE]), nf = wv_uniform_i(counts[MJH_C_NF]);
  Efc P;
  efc_layout(M, B, e, n, P);
  const int lane = wv_lane();
  const int n4 = n & ~3, L = LT >= 0 ? LT : (n4 >> 2), ntail = NT >= 0 ? NT : (n - n4);
  const int row = lane >> 4, col = lane & 15;
  // constraint owned by this lane (-1: none)
  int j = -1;
  if (col < L) j = 4*col + row;
  else if (col == 15 && row < ntail) j = n4 + row;
  const int own = (j >= 0);
  const int jj = own ? j : 0;
  const int kind = (jj < ne) ? 0 : (jj < ne + nf ? 1 : 2);   // equality / friction / inequality
  const bool isfric = (kind == 1), isineq = (kind == 2);
  real f = own ? P.force[jj] : 0;
  const real bj = own ? P.b[jj] : 0;
  const real fl = own ? P.floss[jj] : 0;
#if defined(MJH_HOSTSIM)
  const real* ARl = nullptr;
#else
  // byte offset of AR inside the workgroup's LDS block (the block starts at LDS address 0)
  const __attribute__((address_space(3))) real* ARl =
      (const __attribute__((address_space(3))) real*)(unsigned)(size_t)((const char*)P.AR.p - mjh_lds());
#endif
  (void)ARl;
  auto ar_load = [&](int r) -> real {       // AR[r][jj]
    if (ARL) return ARl[r*n + jj];
    return P.AR[(size_t)r*n + jj];
  };
  const real arjj = own ? ar_load(jj) : 1;
  const real ainv = 1 / arjj;
  const real A = 1/ainv;                  // costChange's A (:216-237), the same bits every visit
  // projection bounds of my constraint: equality (-inf, inf), friction [-fl, fl], inequality [0, inf)
  const real pinf = __builtin_huge_val();
  const real blo = isfric ? -fl : (isineq ? 0.0 : -pinf);
  const real bhi = isfric ? fl : pinf;
  const int maxiter = M.o.iterations;
  const real scale = 1 / (M.o.meaninertia * (real)(M.s.nv > 1 ? M.s.nv : 1));
  const auto* otab_all = wv_uniform_ptr(M.pgs_order);
  const auto* otab_adr = wv_uniform_ptr(M.pgs_order_adr);

  // Constraint islands (engine_forward.c:1187-1222): with islands on, the reference runs this same
  // loop once per island over that island's rows (their original relative order, its own PCG
  // stream, momentum and termination), residuals still taken over the full row.  nisland == 0
  // (islands disabled) is one pass over everything.
  const int nisl_raw = wv_uniform_i(counts[MJH_C_NISLAND]);
  const int nisl = nisl_raw > 1 ? nisl_raw : 1;
  const int myisl = (own && nisl > 1) ? P.island[jj] : 0;
  int niter0 = 0;

  for (int isl = 0; isl < nisl; isl++) {
    const int member = own && myisl == isl;
    int nk = n;                 // rows of this island
    int crank = jj;             // island-local index of my constraint
    if (nisl > 1) {
      nk = wv_uniform_i(wv_sum_i(member));
      crank = 0;
      for (int q = 0; q < n; q++) {
        const int inq = (P.island[q] == isl);
        if (inq && q < jj) crank++;
      }
      // efclist: island-local index -> global row (kept in the order scratch array)
      if (member) P.order[crank] = jj;
      wv_sync();
    }
    if (nk == 0) continue;
    // global row / owner lane of island-local index `lane`
    int grow = lane;
    if (nisl > 1) grow = (lane < nk) ? P.order[lane] : 0;
    const int growlane = (grow < n4) ? 16*(grow & 3) + (grow >> 2) : 16*(grow - n4) + 15;
    wv_sync();
    const auto* otab = otab_all + otab_adr[nk];
    real fprev = f, fmom = f;

  int iter = 0, nesterov_k = 0;
  // visitation order of the coming iteration (lane b holds order[b]); fetched one iteration ahead
  int ord_next = (lane < nk) ? otab[lane] : 0;
  while (iter < maxiter) {
    // island-local index visited at position b -> its global row and owner lane (per lane b)
    const int ordc = ord_next;
    if (iter + 1 < maxiter) ord_next = (lane < nk) ? otab[(iter + 1)*nk + lane] : 0;
    int ord = ordc, ordlane;
    if (nisl > 1) {
      ord = wv_shfl_i(grow, ordc);
      ordlane = wv_shfl_i(growlane, ordc);
    } else {
      ordlane = (ord < n4) ? 16*(ord & 3) + (ord >> 2) : 16*(ord - n4) + 15;
    }
    // ---- Nesterov extrapolation (:508-554)
    real beta = 0;
    if (iter > 0) beta = (real)(nesterov_k - 1) / (real)(nesterov_k + 2);
    if (member) {
      if (beta > 0) {
        real f_save = f;
        real fx = f_save + beta*(f_save - fprev);
        fprev = f_save;
        if (kind == 1) fx = r_clip(fx, -fl, fl);
        else if (kind == 2 && fx < 0) fx = 0;
        f = fx;
        fmom = fx;
      } else {
        fprev = f;
        fmom = f;
      }
    }

    // ---- one sweep
    real improvement = 0;
    int i = wv_bcast_i(ord, 0);
    real a = own ? ar_load(i) : 0;                   // row of the first visited constraint
#if MJH_PGS_PREFETCH2
    // (AR in global memory: TWO rows ahead -- a visit of a short row is shorter than an L2 round trip, and a lone
    //  straggler has no other wavefront to hide it behind)
    int i1 = wv_bcast_i(ord, 1 < nk ? 1 : 0);
    real a1 = (!ARL && own) ? ar_load(i1) : 0;
#endif
    for (int bi = 0; bi < nk; bi++) {
      const real p = a*f;
      // prefetch the next row while this one is reduced (issued after the product so that the
      // wait for the current row does not also drain this load)
#if MJH_PGS_PREFETCH2
      int inext = ARL ? wv_bcast_i(ord, bi + 1 < nk ? bi + 1 : bi) : wv_bcast_i(ord, bi + 2 < nk ? bi + 2 : nk - 1);
#else
      int inext = wv_bcast_i(ord, bi + 1 < nk ? bi + 1 : bi);
#endif
#if !defined(MJH_HOSTSIM)
      asm volatile("" : "+s"(inext) : "v"(p));      // orders the prefetch after the product
#endif
      const real anext = own ? ar_load(inext) : 0;
      // chain sums: acc of DPP row c = r_c (mju_dot's res_c), valid in every lane of the row
      real acc = 0;
      if (L > 0) { acc = acc + wv_row_bcast<0>(p);
      MJH_PGS_CHAIN_STEP(1) MJH_PGS_CHAIN_STEP(2) MJH_PGS_CHAIN_STEP(3) MJH_PGS_CHAIN_STEP(4)
      MJH_PGS_CHAIN_STEP(5) MJH_PGS_CHAIN_STEP(6) MJH_PGS_CHAIN_STEP(7) MJH_PGS_CHAIN_STEP(8)
      MJH_PGS_CHAIN_STEP(9) MJH_PGS_CHAIN_STEP(10) MJH_PGS_CHAIN_STEP(11) MJH_PGS_CHAIN_STEP(12)
      MJH_PGS_CHAIN_STEP(13) MJH_PGS_CHAIN_STEP(14) MJH_PGS_CHAIN_STEP(15)
      MJH_PGS_CHAIN_END }
      // (r0 + r2) + (r1 + r3) through v_readlane + scalar operands: measured 3 % faster than the
      // all-VALU v_permlane32_swap / v_permlane16_swap combine (wv_rows_sum, -DMJH_PERMLANE_COMBINE) --
      // the kernel is VALU-issue bound and the readlane path runs beside it
#ifdef MJH_PERMLANE_COMBINE
      real dot = wv_rows_sum(acc);
#else
      real dot = (wv_bcast(acc, 0) + wv_bcast(acc, 32)) + (wv_bcast(acc, 16) + wv_bcast(acc, 48));
#endif
      if (ntail == 3) dot += wv_bcast(p, 15) + wv_bcast(p, 31) + wv_bcast(p, 47);
      else if (ntail == 2) dot += wv_bcast(p, 15) + wv_bcast(p, 31);
      else if (ntail == 1) dot += wv_bcast(p, 15);
      // every lane evaluates the update for its own constraint; only the owner of row i keeps it
      const real res = bj + dot;
      const real oldf = f;
      real fn = oldf - res*ainv;
      // projection onto [blo, bhi] (compare-and-select keeps the reference's `if (f < lo) f = lo`
      // semantics for -0.0 and NaN)
      fn = (fn < blo) ? blo : ((fn > bhi) ? bhi : fn);
      // costChange (:216-237) with A = 1/ARinv
      const real delta = fn - oldf;
      real change = 0.5*delta*delta*A + delta*res;
      if (change > 1e-10) { fn = oldf; change = 0; }
      const int li = wv_bcast_i(ordlane, bi);          // owner lane of constraint i
      if (lane == li) f = fn;
      improvement -= wv_bcast(change, li);
#if MJH_PGS_PREFETCH2
      if (ARL) { i = inext; a = anext; }
      else { i = i1; a = a1; i1 = inext; a1 = anext; }
#else
      i = inext;
      a = anext;
#endif
    }
    improvement *= scale;

    // ---- gradient restart (:694-713): sum over the island's constraints in index order
    int restart = 0;
    if (iter > 0) {
      const real ce = (f - fmom) * (fmom - fprev);
      real dotce = 0;
      for (int q = 0; q < nk; q++) dotce += wv_bcast(ce, wv_bcast_i(growlane, q));
      restart = (dotce < 0);
    }
    if (restart) nesterov_k = 0; else nesterov_k++;
    iter++;
#if MJH_STEP_PRIO && !MJH_LANE_MODE
    // (a solve that does not converge is what makes an environment the launch's straggler: raise the wavefront's issue
    // priority while it lasts -- a quarter / half of the iteration budget spent; the step loop sets the level again after
    // the step (rollout_env, mjh_step.h); a forward-kernel dispatch ends with the wavefront)
    // One-wavefront workgroups only, like the step loop's levels: in a multi-wavefront environment only the waves that run
    // the solver would be raised -- the skew rollout_env avoids.
    if ((int)blockDim.x == MJH_WAVE) {
      if (iter == (maxiter >> 2)) __builtin_amdgcn_s_setprio(2);
      else if (iter == (maxiter >> 1)) __builtin_amdgcn_s_setprio(3);
    }
#endif
    if (improvement < M.o.tolerance) break;
  }
    if (isl == 0) niter0 = iter;
  }

  // final dual state (dualState, :270-345), forces back to memory, iteration count
  if (own) {
    int st;
    if (kind == 0) st = MJH_STATE_QUADRATIC;
    else if (kind == 1) {
      if (f <= -fl) st = MJH_STATE_LINEARPOS;
      else if (f >= fl) st = MJH_STATE_LINEARNEG;
      else st = MJH_STATE_QUADRATIC;
    } else st = (f <= 0) ? MJH_STATE_SATISFIED : MJH_STATE_QUADRATIC;
    P.state[jj] = st;
    P.force[jj] = f;
  }
  if (lane == 0) counts[MJH_C_NITER] = niter0;
  wv_sync();
}
// ------------------------------------------------------------------------------------------------
// solPGS in RESIDUAL-UPDATE form, nefc <= 64 -- OPT-IN, tolerance parity (mjhip_batch_set_pgs_mode(1), $MJHIP_PGS=residual).
//
// Lane j owns constraint j and carries, next to force / bounds / momentum, the row's residual r_j = b_j + AR_j . f.
// Visiting row i changes ONE force by delta; every lane folds that into its own residual with one multiply-add,
// r_j += AR[i][j] * delta (row i of AR is contiguous across the lanes).  A row visit is then
//   owner: f_i - r_i/AR_ii, projection, cost guard  ->  v_readlane(delta)  ->  one v_fma per lane
// instead of a fresh mju_dot per visit (nefc/4 dependent adds + the combine): ~10 dependent instructions instead of
// ~20 + nefc/4, and a third of the vector instructions.  The sweep, the projections, the cost guard, Nesterov momentum,
// the gradient restart and the termination test are the reference's (engine_solver.c:457-741); what differs is the
// ROUNDING of a residual (an accumulation of updates since the start of the iteration, where every lane's residual is
// computed afresh, instead of a four-chain dot product per visit) and of the restart's sum (a butterfly instead of a
// serial loop).  Forces therefore agree with the reference to rounding, not bit for bit, and the iteration at which
// `improvement < tolerance` fires may move: next states within 1e-6 (measured: bench.py `pgs_residual`,
// tests/test_gpu_parity.py), contact and constraint counts exact.  The default (mode 0) stays the bit-exact sweep above.
// ------------------------------------------------------------------------------------------------
template <int ARL>
MJH_DEVN_HOT void solve_pgs_resid(MREF M_, BREF B_, int e_) {
  const auto& M = wv_uniform_ref(M_);
  BREF B = B_;
  const int e = wv_uniform_i(e_);
  iptr counts = MJH_F(B, counts, e);
  const int n = wv_uniform_i(counts[MJH_C_NEFC]), ne = wv_uniform_i(counts[MJH_C_NE]), nf = wv_uniform_i(counts[MJH_C_NF]);
  Efc P;
  efc_layout(M, B, e, n, P);
  const int lane = wv_lane();
  const int own = lane < n;
  const int jj = own ? lane : 0;
  const int kind = (jj < ne) ? 0 : (jj < ne + nf ? 1 : 2);   // equality / friction / inequality
  const bool isfric = (kind == 1), isineq = (kind == 2);
  real f = own ? P.force[jj] : 0;
  const real bj = own ? P.b[jj] : 0;
  const real fl = own ? P.floss[jj] : 0;
#if defined(MJH_HOSTSIM)
  const real* ARl = nullptr;
#else
  const __attribute__((address_space(3))) real* ARl =
      (const __attribute__((address_space(3))) real*)(unsigned)(size_t)((const char*)P.AR.p - mjh_lds());
#endif
  (void)ARl;
  auto ar_load = [&](int r) -> real {       // AR[r][jj]
    if (ARL) return ARl[r*n + jj];
    return P.AR[(size_t)r*n + jj];
  };
  const real arjj = own ? ar_load(jj) : 1;
  const real ainv = 1 / arjj;
  const real A = 1/ainv;
  const real pinf = __builtin_huge_val();
  const real blo = isfric ? -fl : (isineq ? 0.0 : -pinf);
  const real bhi = isfric ? fl : pinf;
  const int maxiter = M.o.iterations;
  const real scale = 1 / (M.o.meaninertia * (real)(M.s.nv > 1 ? M.s.nv : 1));
  const auto* otab_all = wv_uniform_ptr(M.pgs_order);
  const auto* otab_adr = wv_uniform_ptr(M.pgs_order_adr);
  const int nisl_raw = wv_uniform_i(counts[MJH_C_NISLAND]);
  const int nisl = nisl_raw > 1 ? nisl_raw : 1;
  const int myisl = (own && nisl > 1) ? P.island[jj] : 0;
  int niter0 = 0;

  for (int isl = 0; isl < nisl; isl++) {
    const int member = own && myisl == isl;
    int nk = n;
    int crank = jj;
    if (nisl > 1) {
      nk = wv_uniform_i(wv_sum_i(member));
      crank = 0;
      for (int q = 0; q < n; q++) {
        const int inq = (P.island[q] == isl);
        if (inq && q < jj) crank++;
      }
      if (member) P.order[crank] = jj;
      wv_sync();
    }
    if (nk == 0) continue;
    int grow = lane;                 // global row (= owner lane) of island-local index `lane`
    if (nisl > 1) grow = (lane < nk) ? P.order[lane] : 0;
    wv_sync();
    const auto* otab = otab_all + otab_adr[nk];
    real fprev = f, fmom = f;
    int iter = 0, nesterov_k = 0;
    int ord_next = (lane < nk) ? otab[lane] : 0;
    while (iter < maxiter) {
      const int ordc = ord_next;
      if (iter + 1 < maxiter) ord_next = (lane < nk) ? otab[(iter + 1)*nk + lane] : 0;
      const int ord = (nisl > 1) ? wv_shfl_i(grow, ordc) : ordc;      // lane b: the row visited at position b
      // ---- Nesterov extrapolation (:508-554)
      real beta = 0;
      if (iter > 0) beta = (real)(nesterov_k - 1) / (real)(nesterov_k + 2);
      if (member) {
        if (beta > 0) {
          const real f_save = f;
          real fx = f_save + beta*(f_save - fprev);
          fprev = f_save;
          if (kind == 1) fx = r_clip(fx, -fl, fl);
          else if (kind == 2 && fx < 0) fx = 0;
          f = fx;
          fmom = fx;
        } else {
          fprev = f;
          fmom = f;
        }
      }
      // ---- every row's residual afresh, once per iteration: r_j = b_j + sum_k AR[k][j] f_k (AR is symmetric; row k of it
      // is contiguous across the lanes), two interleaved partial sums
      real r = bj;
      {
        // (eight rows of AR in flight at a time: out of LDS the loads return in order at ds latency, out of global memory
        //  -- AR beyond the LDS plan's room, nefc above ~27 at 4096 environments -- one load per multiply-add would wait a
        //  full L2 round trip each)
        real r1 = 0;
        int k = 0;
        for (; k + 8 <= n; k += 8) {
          real av[8];
#pragma unroll
          for (int q = 0; q < 8; q++) av[q] = ar_load(k + q);
#pragma unroll
          for (int q = 0; q < 8; q += 2) {
            r = __builtin_fma(av[q], wv_bcast(f, k + q), r);
            r1 = __builtin_fma(av[q + 1], wv_bcast(f, k + q + 1), r1);
          }
        }
        for (; k + 1 < n; k += 2) {
          const real a0 = ar_load(k), a1 = ar_load(k + 1);
          r = __builtin_fma(a0, wv_bcast(f, k), r);
          r1 = __builtin_fma(a1, wv_bcast(f, k + 1), r1);
        }
        if (k < n) r = __builtin_fma(ar_load(k), wv_bcast(f, k), r);
        r += r1;
      }

      // ---- one sweep
      real improvement = 0;
      // a visit: every lane evaluates the update of its own constraint from its residual; the visited row's owner's
      // counts.  The residuals move on with the owner's delta at once; the cost guard (costChange, :216-237) is evaluated
      // beside that and, where it rejects the update (change > 1e-10: rare), the step is taken back.
      auto visit = [&](int i, real a) {
        const real res = r;
        real fn = f - res*ainv;
        fn = (fn < blo) ? blo : ((fn > bhi) ? bhi : fn);
        const real delta = fn - f;
        const real d = wv_bcast(delta, i);
        const real rn = __builtin_fma(a, d, r);
        const real change = 0.5*delta*delta*A + delta*res;
        const real ch = wv_bcast(change, i);
        if (!(ch > 1e-10)) {
          r = rn;
          if (lane == i) f = fn;
          improvement -= ch;
        }
      };
      if (ARL) {
        // AR in LDS: the coming row is read while this one is applied
        int i = wv_bcast_i(ord, 0);
        real a = ar_load(i);
        for (int bi = 0; bi < nk; bi++) {
          const int inext = wv_bcast_i(ord, bi + 1 < nk ? bi + 1 : bi);
          const real anext = ar_load(inext);
          visit(i, a);
          i = inext;
          a = anext;
        }
      } else {
        // AR in global memory: the visiting order is known, so the rows of the NEXT four visits are requested before the
        // current four are applied (a visit is ~170 cycles, an L2 round trip several times that)
        int i0 = wv_bcast_i(ord, 0), i1 = wv_bcast_i(ord, 1 < nk ? 1 : 0), i2 = wv_bcast_i(ord, 2 < nk ? 2 : 0), i3 = wv_bcast_i(ord, 3 < nk ? 3 : 0);
        real a0 = ar_load(i0), a1 = ar_load(i1), a2 = ar_load(i2), a3 = ar_load(i3);
        for (int bi = 0; bi < nk; bi += 4) {
          const int last = nk - 1;
          const int j0 = wv_bcast_i(ord, bi + 4 < nk ? bi + 4 : last), j1 = wv_bcast_i(ord, bi + 5 < nk ? bi + 5 : last),
                    j2 = wv_bcast_i(ord, bi + 6 < nk ? bi + 6 : last), j3 = wv_bcast_i(ord, bi + 7 < nk ? bi + 7 : last);
          const real n0 = ar_load(j0), n1 = ar_load(j1), n2 = ar_load(j2), n3 = ar_load(j3);
          visit(i0, a0);
          if (bi + 1 < nk) visit(i1, a1);
          if (bi + 2 < nk) visit(i2, a2);
          if (bi + 3 < nk) visit(i3, a3);
          i0 = j0; i1 = j1; i2 = j2; i3 = j3;
          a0 = n0; a1 = n1; a2 = n2; a3 = n3;
        }
      }
      improvement *= scale;

      // ---- gradient restart (:694-713)
      int restart = 0;
      if (iter > 0) {
        real ce = member ? (f - fmom) * (fmom - fprev) : 0;
        for (int m = 1; m < MJH_W; m <<= 1) ce += wv_shfl_xor(ce, m);
        restart = (wv_bcast(ce, 0) < 0);
      }
      if (restart) nesterov_k = 0; else nesterov_k++;
      iter++;
      if (improvement < M.o.tolerance) break;
    }
    if (isl == 0) niter0 = iter;
  }

  // final dual state (dualState, :270-345), forces back to memory, iteration count
  if (own) {
    int st;
    if (kind == 0) st = MJH_STATE_QUADRATIC;
    else if (kind == 1) {
      if (f <= -fl) st = MJH_STATE_LINEARPOS;
      else if (f >= fl) st = MJH_STATE_LINEARNEG;
      else st = MJH_STATE_QUADRATIC;
    } else st = (f <= 0) ? MJH_STATE_SATISFIED : MJH_STATE_QUADRATIC;
    P.state[jj] = st;
    P.force[jj] = f;
  }
  if (lane == 0) counts[MJH_C_NITER] = niter0;
  wv_sync();
}

// The residual-update sweep for 64 < nefc <= 128: two constraints per lane (row lane in slot 0, row lane + 64 in slot 1),
// AR in global memory (> 32 KB), the rows of the next four visits requested ahead.  One island (the caller falls back
// otherwise, like solve_pgs_wide).  Opt-in with the routine above: in the settled humanoid regime the handful of
// environments beyond 64 rows end every launch (tools/regime_stats.py: 126 ms against a mean of 99 ms per 100 steps).
MJH_DEVN_HOT void solve_pgs_resid_wide(MREF M_, BREF B_, int e_) {
  const auto& M = wv_uniform_ref(M_);
  BREF B = B_;
  const int e = wv_uniform_i(e_);
  iptr counts = MJH_F(B, counts, e);
  const int n = wv_uniform_i(counts[MJH_C_NEFC]), ne = wv_uniform_i(counts[MJH_C_NE]), nf = wv_uniform_i(counts[MJH_C_NF]);
  Efc P;
  efc_layout(M, B, e, n, P);
  const int lane = wv_lane();
  const int j0 = lane, own1 = lane + MJH_W < n, j1 = own1 ? lane + MJH_W : 0;
  auto kind_of = [&](int j) -> int { return (j < ne) ? 0 : (j < ne + nf ? 1 : 2); };
  const int kind0 = kind_of(j0), kind1 = kind_of(j1);
  real f0 = P.force[j0], f1 = own1 ? (real)P.force[j1] : 0;
  const real b0 = P.b[j0], b1 = own1 ? (real)P.b[j1] : 0;
  const real fl0 = P.floss[j0], fl1 = own1 ? (real)P.floss[j1] : 0;
  auto ar0 = [&](int r) -> real { return P.AR[(size_t)r*n + j0]; };
  auto ar1 = [&](int r) -> real { return P.AR[(size_t)r*n + j1]; };
  const real ainv0 = 1 / ar0(j0), ainv1 = 1 / (own1 ? ar1(j1) : (real)1);
  const real A0 = 1/ainv0, A1 = 1/ainv1;
  const real pinf = __builtin_huge_val();
  const real blo0 = kind0 == 1 ? -fl0 : (kind0 == 2 ? 0.0 : -pinf), bhi0 = kind0 == 1 ? fl0 : pinf;
  const real blo1 = kind1 == 1 ? -fl1 : (kind1 == 2 ? 0.0 : -pinf), bhi1 = kind1 == 1 ? fl1 : pinf;
  const int maxiter = M.o.iterations;
  const real scale = 1 / (M.o.meaninertia * (real)(M.s.nv > 1 ? M.s.nv : 1));
  const auto* otab = wv_uniform_ptr(M.pgs_order) + wv_uniform_ptr(M.pgs_order_adr)[n];
  real fprev0 = f0, fmom0 = f0, fprev1 = f1, fmom1 = f1;
  int iter = 0, nesterov_k = 0;
  int ordn0 = otab[lane], ordn1 = own1 ? otab[MJH_W + lane] : 0;
  auto order_at = [&](int o0, int o1, int b) -> int { return b < MJH_W ? wv_bcast_i(o0, b) : wv_bcast_i(o1, b - MJH_W); };
  while (iter < maxiter) {
    const int ord0 = ordn0, ord1 = ordn1;
    if (iter + 1 < maxiter) {
      ordn0 = otab[(iter + 1)*n + lane];
      ordn1 = own1 ? otab[(iter + 1)*n + MJH_W + lane] : 0;
    }
    // ---- Nesterov extrapolation (:508-554)
    real beta = 0;
    if (iter > 0) beta = (real)(nesterov_k - 1) / (real)(nesterov_k + 2);
    if (beta > 0) {
      {
        const real fs = f0;
        real fx = fs + beta*(fs - fprev0);
        fprev0 = fs;
        if (kind0 == 1) fx = r_clip(fx, -fl0, fl0);
        else if (kind0 == 2 && fx < 0) fx = 0;
        f0 = fx; fmom0 = fx;
      }
      if (own1) {
        const real fs = f1;
        real fx = fs + beta*(fs - fprev1);
        fprev1 = fs;
        if (kind1 == 1) fx = r_clip(fx, -fl1, fl1);
        else if (kind1 == 2 && fx < 0) fx = 0;
        f1 = fx; fmom1 = fx;
      }
    } else {
      fprev0 = f0; fmom0 = f0;
      fprev1 = f1; fmom1 = f1;
    }
    // ---- residuals afresh: r_j = b_j + sum_k AR[k][j] f_k, four rows of AR in flight
    real r0 = b0, r1 = b1;
    for (int k = 0; k < n; k += 4) {
      real a0v[4], a1v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { const int kk = k + q < n ? k + q : n - 1; a0v[q] = ar0(kk); a1v[q] = ar1(kk); }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (k + q < n) {
          const int kk = k + q;
          const real fk = kk < MJH_W ? wv_bcast(f0, kk) : wv_bcast(f1, kk - MJH_W);
          r0 = __builtin_fma(a0v[q], fk, r0);
          r1 = __builtin_fma(a1v[q], fk, r1);
        }
      }
    }
    // ---- one sweep
    real improvement = 0;
    auto visit = [&](int i, real a0, real a1) {
      const int s = i >= MJH_W, src = i & (MJH_W - 1);
      const real res = s ? r1 : r0, fc = s ? f1 : f0, ainv = s ? ainv1 : ainv0, A = s ? A1 : A0;
      const real blo = s ? blo1 : blo0, bhi = s ? bhi1 : bhi0;
      real fn = fc - res*ainv;
      fn = (fn < blo) ? blo : ((fn > bhi) ? bhi : fn);
      const real delta = fn - fc;
      const real d = wv_bcast(delta, src);
      const real rn0 = __builtin_fma(a0, d, r0), rn1 = __builtin_fma(a1, d, r1);
      const real change = 0.5*delta*delta*A + delta*res;
      const real ch = wv_bcast(change, src);
      if (!(ch > 1e-10)) {
        r0 = rn0; r1 = rn1;
        if (lane == src) { if (s) f1 = fn; else f0 = fn; }
        improvement -= ch;
      }
    };
    {
      const int last = n - 1;
      int i0 = order_at(ord0, ord1, 0), i1 = order_at(ord0, ord1, 1), i2 = order_at(ord0, ord1, 2), i3 = order_at(ord0, ord1, 3);
      real p0 = ar0(i0), p1 = ar0(i1), p2 = ar0(i2), p3 = ar0(i3);
      real q0 = ar1(i0), q1 = ar1(i1), q2 = ar1(i2), q3 = ar1(i3);
      for (int bi = 0; bi < n; bi += 4) {
        const int g0 = order_at(ord0, ord1, bi + 4 < n ? bi + 4 : last), g1 = order_at(ord0, ord1, bi + 5 < n ? bi + 5 : last),
                  g2 = order_at(ord0, ord1, bi + 6 < n ? bi + 6 : last), g3 = order_at(ord0, ord1, bi + 7 < n ? bi + 7 : last);
        const real np0 = ar0(g0), np1 = ar0(g1), np2 = ar0(g2), np3 = ar0(g3);
        const real nq0 = ar1(g0), nq1 = ar1(g1), nq2 = ar1(g2), nq3 = ar1(g3);
        visit(i0, p0, q0);
        if (bi + 1 < n) visit(i1, p1, q1);
        if (bi + 2 < n) visit(i2, p2, q2);
        if (bi + 3 < n) visit(i3, p3, q3);
        i0 = g0; i1 = g1; i2 = g2; i3 = g3;
        p0 = np0; p1 = np1; p2 = np2; p3 = np3;
        q0 = nq0; q1 = nq1; q2 = nq2; q3 = nq3;
      }
    }
    improvement *= scale;
    // ---- gradient restart (:694-713)
    int restart = 0;
    if (iter > 0) {
      real ce = (f0 - fmom0) * (fmom0 - fprev0) + (own1 ? (f1 - fmom1) * (fmom1 - fprev1) : (real)0);
      for (int m = 1; m < MJH_W; m <<= 1) ce += wv_shfl_xor(ce, m);
      restart = (wv_bcast(ce, 0) < 0);
    }
    if (restart) nesterov_k = 0; else nesterov_k++;
    iter++;
    if (improvement < M.o.tolerance) break;
  }
  auto final_state = [&](int kind, real f, real fl) -> int {
    if (kind == 0) return MJH_STATE_QUADRATIC;
    if (kind == 1) return f <= -fl ? MJH_STATE_LINEARPOS : (f >= fl ? MJH_STATE_LINEARNEG : MJH_STATE_QUADRATIC);
    return f <= 0 ? MJH_STATE_SATISFIED : MJH_STATE_QUADRATIC;
  };
  P.state[j0] = final_state(kind0, f0, fl0);
  P.force[j0] = f0;
  if (own1) { P.state[j1] = final_state(kind1, f1, fl1); P.force[j1] = f1; }
  if (lane == 0) counts[MJH_C_NITER] = iter;
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// solPGS for 64 < nefc <= 128, iterate in registers, TWO constraints per lane.
//
// Same layout idea with chains of up to 32 positions: position k of chain c is lane 16*c + (k & 15),
// slot k >> 4.  Slot 0 holds constraints 4*col + row (all 64 exist because nefc > 64), slot 1 holds
// 4*(col + 16) + row while col + 16 < L, and the 1..3 tail constraints in column 15 of slot 1 (free
// because L <= 31 whenever a tail exists).  A chain sum is the 16 row broadcasts of the slot-0
// products followed by L - 16 of the slot-1 products: mju_dot's order.  AR (> 32 KB here) is read from
// global memory, the coming row prefetched while the current one is reduced.  A settled humanoid
// spends a fraction of a percent of its steps with more than 64 rows, but a launch ends with its
// slowest environment: on the memory-based sweep those few steps set the kernel time of the whole batch.
// No islands (the caller falls back to the generic sweep when island discovery is on).
// ------------------------------------------------------------------------------------------------
#define MJH_PGSW_STEP0(k) acc = acc + wv_row_bcast<k>(p0);
#define MJH_PGSW_STEP1(k) if (L > 16 + k) { acc = acc + wv_row_bcast<k>(p1);
#define MJH_PGSW_END }}}}}}}}}}}}}}}}

MJH_DEVN_HOT void solve_pgs_wide(MREF M_, BREF B_, int e_) {
  const auto& M = wv_uniform_ref(M_);
  BREF B = B_;
  const int e = wv_uniform_i(e_);
  iptr counts = MJH_F(B, counts, e);
  const int n = wv_uniform_i(counts[MJH_C_NEFC]), ne = wv_uniform_i(counts[MJH_C_NE]), nf = wv_uniform_i(counts[MJH_C_NF]);
  Efc P;
  efc_layout(M, B, e, n, P);
  const int lane = wv_lane();
  const int n4 = n & ~3, L = n4 >> 2, ntail = n - n4;      // 16 <= L <= 32
  const int row = lane >> 4, col = lane & 15;
  // my two constraints (slot 0 always exists)
  const int j0 = 4*col + row;
  int j1 = -1;
  if (col + 16 < L) j1 = 4*(col + 16) + row;
  else if (col == 15 && row < ntail) j1 = n4 + row;
  const int own1 = (j1 >= 0);
  const int jj1 = own1 ? j1 : 0;
  auto kind_of = [&](int j) -> int { return (j < ne) ? 0 : (j < ne + nf ? 1 : 2); };   // equality / friction / inequality
  const int kind0 = kind_of(j0), kind1 = kind_of(jj1);
  real f0 = P.force[j0], f1 = own1 ? (real)P.force[jj1] : 0;
  const real b0 = P.b[j0], b1 = own1 ? (real)P.b[jj1] : 0;
  const real fl0 = P.floss[j0], fl1 = own1 ? (real)P.floss[jj1] : 0;
  auto ar = [&](int r, int j) -> real { return P.AR[(size_t)r*n + j]; };
  const real ainv0 = 1 / ar(j0, j0), ainv1 = 1 / (own1 ? ar(jj1, jj1) : (real)1);
  const real A0 = 1/ainv0, A1 = 1/ainv1;
  const real pinf = __builtin_huge_val();
  const real blo0 = kind0 == 1 ? -fl0 : (kind0 == 2 ? 0.0 : -pinf), bhi0 = kind0 == 1 ? fl0 : pinf;
  const real blo1 = kind1 == 1 ? -fl1 : (kind1 == 2 ? 0.0 : -pinf), bhi1 = kind1 == 1 ? fl1 : pinf;
  const int maxiter = M.o.iterations;
  const real scale = 1 / (M.o.meaninertia * (real)(M.s.nv > 1 ? M.s.nv : 1));
  const auto* otab = wv_uniform_ptr(M.pgs_order) + wv_uniform_ptr(M.pgs_order_adr)[n];
  // owner (lane, slot) of constraint q
  auto lane_of = [&](int q) -> int { return (q < n4) ? 16*(q & 3) + ((q >> 2) & 15) : 16*(q - n4) + 15; };
  auto slot_of = [&](int q) -> int { return (q < n4) ? (q >> 6) : 1; };

  real fprev0 = f0, fmom0 = f0, fprev1 = f1, fmom1 = f1;
  int iter = 0, nesterov_k = 0;
  // visitation order of the coming iteration: lane b holds order[b] and order[64 + b]
  int ordn0 = otab[lane], ordn1 = (64 + lane < n) ? otab[64 + lane] : 0;
  while (iter < maxiter) {
    const int ord0 = ordn0, ord1 = ordn1;
    if (iter + 1 < maxiter) {
      ordn0 = otab[(iter + 1)*n + lane];
      ordn1 = (64 + lane < n) ? otab[(iter + 1)*n + 64 + lane] : 0;
    }
    // ---- Nesterov extrapolation (:508-554)
    real beta = 0;
    if (iter > 0) beta = (real)(nesterov_k - 1) / (real)(nesterov_k + 2);
    if (beta > 0) {
      {
        const real f_save = f0;
        real fx = f_save + beta*(f_save - fprev0);
        fprev0 = f_save;
        if (kind0 == 1) fx = r_clip(fx, -fl0, fl0);
        else if (kind0 == 2 && fx < 0) fx = 0;
        f0 = fx; fmom0 = fx;
      }
      if (own1) {
        const real f_save = f1;
        real fx = f_save + beta*(f_save - fprev1);
        fprev1 = f_save;
        if (kind1 == 1) fx = r_clip(fx, -fl1, fl1);
        else if (kind1 == 2 && fx < 0) fx = 0;
        f1 = fx; fmom1 = fx;
      }
    } else {
      fprev0 = f0; fmom0 = f0;
      fprev1 = f1; fmom1 = f1;
    }

    // ---- one sweep
    real improvement = 0;
    int i = wv_bcast_i(ord0, 0);
    real a0 = ar(i, j0), a1 = own1 ? ar(i, jj1) : 0;
#if MJH_PGS_PREFETCH2
    // (two rows ahead, as in solve_pgs_fast's global-memory case: these environments are the ones a launch waits for)
    int ix = wv_bcast_i(ord0, 1);
    real ax0 = ar(ix, j0), ax1 = own1 ? ar(ix, jj1) : 0;
#endif
    for (int bi = 0; bi < n; bi++) {
      const real p0 = a0*f0, p1 = a1*f1;
#if MJH_PGS_PREFETCH2
      const int bn = bi + 2 < n ? bi + 2 : n - 1;
#else
      const int bn = bi + 1 < n ? bi + 1 : bi;
#endif
      int inext = bn < 64 ? wv_bcast_i(ord0, bn) : wv_bcast_i(ord1, bn - 64);
#if !defined(MJH_HOSTSIM)
      asm volatile("" : "+s"(inext) : "v"(p0), "v"(p1));      // orders the prefetch after the products
#endif
      const real an0 = ar(inext, j0), an1 = own1 ? ar(inext, jj1) : 0;
      real acc = 0;
      MJH_PGSW_STEP0(0) MJH_PGSW_STEP0(1) MJH_PGSW_STEP0(2) MJH_PGSW_STEP0(3) MJH_PGSW_STEP0(4) MJH_PGSW_STEP0(5)
      MJH_PGSW_STEP0(6) MJH_PGSW_STEP0(7) MJH_PGSW_STEP0(8) MJH_PGSW_STEP0(9) MJH_PGSW_STEP0(10) MJH_PGSW_STEP0(11)
      MJH_PGSW_STEP0(12) MJH_PGSW_STEP0(13) MJH_PGSW_STEP0(14) MJH_PGSW_STEP0(15)
      MJH_PGSW_STEP1(0) MJH_PGSW_STEP1(1) MJH_PGSW_STEP1(2) MJH_PGSW_STEP1(3) MJH_PGSW_STEP1(4) MJH_PGSW_STEP1(5)
      MJH_PGSW_STEP1(6) MJH_PGSW_STEP1(7) MJH_PGSW_STEP1(8) MJH_PGSW_STEP1(9) MJH_PGSW_STEP1(10) MJH_PGSW_STEP1(11)
      MJH_PGSW_STEP1(12) MJH_PGSW_STEP1(13) MJH_PGSW_STEP1(14) MJH_PGSW_STEP1(15)
      MJH_PGSW_END
      real dot = (wv_bcast(acc, 0) + wv_bcast(acc, 32)) + (wv_bcast(acc, 16) + wv_bcast(acc, 48));
      if (ntail == 3) dot += wv_bcast(p1, 15) + wv_bcast(p1, 31) + wv_bcast(p1, 47);
      else if (ntail == 2) dot += wv_bcast(p1, 15) + wv_bcast(p1, 31);
      else if (ntail == 1) dot += wv_bcast(p1, 15);
      // every lane evaluates the update for its constraint in the visited row's slot; the owner keeps it
      const int si = slot_of(i), li = lane_of(i);
      const real bj = si ? b1 : b0, oldf = si ? f1 : f0, ainv = si ? ainv1 : ainv0, A = si ? A1 : A0;
      const real blo = si ? blo1 : blo0, bhi = si ? bhi1 : bhi0;
      const real res = bj + dot;
      real fn = oldf - res*ainv;
      fn = (fn < blo) ? blo : ((fn > bhi) ? bhi : fn);
      const real delta = fn - oldf;
      real change = 0.5*delta*delta*A + delta*res;
      if (change > 1e-10) { fn = oldf; change = 0; }
      if (lane == li) { if (si) f1 = fn; else f0 = fn; }
      improvement -= wv_bcast(change, li);
#if MJH_PGS_PREFETCH2
      i = ix; a0 = ax0; a1 = ax1;
      ix = inext; ax0 = an0; ax1 = an1;
#else
      i = inext;
      a0 = an0; a1 = an1;
#endif
    }
    improvement *= scale;

    // ---- gradient restart (:694-713): sum over the constraints in index order
    int restart = 0;
    if (iter > 0) {
      const real ce0 = (f0 - fmom0) * (fmom0 - fprev0), ce1 = (f1 - fmom1) * (fmom1 - fprev1);
      real dotce = 0;
      for (int q = 0; q < n; q++) dotce += slot_of(q) ? wv_bcast(ce1, lane_of(q)) : wv_bcast(ce0, lane_of(q));
      restart = (dotce < 0);
    }
    if (restart) nesterov_k = 0; else nesterov_k++;
    iter++;
    if (improvement < M.o.tolerance) break;
  }

  // final dual state (dualState, :270-345), forces back to memory, iteration count
  auto finish = [&](int j, int kind, real f, real fl) {
    int st;
    if (kind == 0) st = MJH_STATE_QUADRATIC;
    else if (kind == 1) {
      if (f <= -fl) st = MJH_STATE_LINEARPOS;
      else if (f >= fl) st = MJH_STATE_LINEARNEG;
      else st = MJH_STATE_QUADRATIC;
    } else st = (f <= 0) ? MJH_STATE_SATISFIED : MJH_STATE_QUADRATIC;
    P.state[j] = st;
    P.force[j] = f;
  };
  finish(j0, kind0, f0, fl0);
  if (own1) finish(jj1, kind1, f1, fl1);
  if (lane == 0) counts[MJH_C_NITER] = iter;
  wv_sync();
}
#endif  // 64-lane layout

#if !MJH_LANE_MODE && MJH_W == 32
// ------------------------------------------------------------------------------------------------
// solPGS for nefc <= 32, iterate in registers, one environment per 32-lane group (two DPP rows)
//
// Same idea as the 64-lane layout above with two of mju_dot's four chains per DPP row: chain
// c = 2*row + slot occupies lanes [8*slot, 8*slot + L) of row `row` (L = n4/4 <= 8), constraint
// j < n4 lives in lane 16*((j&3)>>1) + 8*(j&1) + (j>>2), tail constraint n4+t in position 7 of
// chain slot t (free because L <= 7 whenever a tail exists).  Every lane carries two chain sums
// (its row's slots), fed by v_mov_b64_dpp row_newbcast; the rows are combined with one
// v_permlane16_swap exchange per sum: (r0 + r2) + (r1 + r3), mju_dot's association.
// The two groups of a wavefront run this loop independently (their nefc, visitation tables and
// iteration counts differ): control flow diverges per group, every cross-lane read stays inside
// the reader's group.
// ------------------------------------------------------------------------------------------------
#define MJH_PGS2_STEP(k) if (L > k) { accA = accA + wv_row_bcast<k>(p); accB = accB + wv_row_bcast<8 + k>(p);
#define MJH_PGS2_END }}}}}}}

template <int ARL>
MJH_DEVN_HOT void solve_pgs_fast(MREF M_, BREF B_, int e_) {
  const auto& M = wv_uniform_ref(M_);
  BREF B = B_;
  const int e = e_;
  iptr counts = MJH_F(B, counts, e);
  const int n = counts[MJH_C_NEFC], ne = counts[MJH_C_NE], nf = counts[MJH_C_NF];
  Efc P;
  efc_layout(M, B, e, n, P);
  const int lane = wv_lane();
  const int n4 = n & ~3, L = n4 >> 2, ntail = n - n4;
  const int chain = 2*(lane >> 4) + ((lane >> 3) & 1), pos = lane & 7;
  // lane (inside the group) that owns constraint q
  auto lane_of = [&](int q) -> int {
    const int c = (q < n4) ? (q & 3) : (q - n4), k = (q < n4) ? (q >> 2) : 7;
    return 16*(c >> 1) + 8*(c & 1) + k;
  };
  // constraint owned by this lane (-1: none)
  int j = -1;
  if (pos < L) j = 4*pos + chain;
  else if (pos == 7 && chain < ntail) j = n4 + chain;
  const int own = (j >= 0);
  const int jj = own ? j : 0;
  const int kind = (jj < ne) ? 0 : (jj < ne + nf ? 1 : 2);   // equality / friction / inequality
  const bool isfric = (kind == 1), isineq = (kind == 2);
  real f = own ? P.force[jj] : 0;
  const real bj = own ? P.b[jj] : 0;
  const real fl = own ? P.floss[jj] : 0;
#if defined(MJH_HOSTSIM)
  const real* ARl = nullptr;
#else
  // byte offset of AR inside the workgroup's LDS allocation (which starts at LDS address 0)
  const __attribute__((address_space(3))) real* ARl =
      (const __attribute__((address_space(3))) real*)(unsigned)(size_t)((const char*)P.AR.p - mjh_lds());
#endif
  (void)ARl;
  auto ar_load = [&](int r) -> real {       // AR[r][jj]
    if (ARL) return ARl[r*n + jj];
    return P.AR[(size_t)r*n + jj];
  };
  const real arjj = own ? ar_load(jj) : 1;
  const real ainv = 1 / arjj;
  const real A = 1/ainv;                  // costChange's A (:216-237), the same bits every visit
  const real pinf = __builtin_huge_val();
  const real blo = isfric ? -fl : (isineq ? 0.0 : -pinf);
  const real bhi = isfric ? fl : pinf;
  const int maxiter = M.o.iterations;
  const real scale = 1 / (M.o.meaninertia * (real)(M.s.nv > 1 ? M.s.nv : 1));
  const auto* otab_all = wv_uniform_ptr(M.pgs_order);
  const auto* otab_adr = wv_uniform_ptr(M.pgs_order_adr);

  // constraint islands as in the 64-lane version (one pass of the loop per island)
  const int nisl_raw = MJH_HAS(MJH_FT_ISLANDS) ? (int)counts[MJH_C_NISLAND] : 1;
  const int nisl = nisl_raw > 1 ? nisl_raw : 1;
  const int myisl = (own && nisl > 1) ? (int)P.island[jj] : 0;
  int niter0 = 0;

  for (int isl = 0; isl < nisl; isl++) {
    const int member = own && myisl == isl;
    int nk = n;                 // rows of this island
    int crank = jj;             // island-local index of my constraint
    if (nisl > 1) {
      nk = wv_sum_i(member);
      crank = 0;
      for (int q = 0; q < n; q++) {
        const int inq = (P.island[q] == isl);
        if (inq && q < jj) crank++;
      }
      if (member) P.order[crank] = jj;
      wv_sync();
    }
    if (nk == 0) continue;
    // global row / owner lane of island-local index `lane`
    int grow = lane;
    if (nisl > 1) grow = (lane < nk) ? (int)P.order[lane] : 0;
    const int growlane = lane_of(grow);
    wv_sync();
    const auto* otab = otab_all + otab_adr[nk];
    real fprev = f, fmom = f;

    int iter = 0, nesterov_k = 0;
    int ord_next = (lane < nk) ? (int)otab[lane] : 0;
    while (iter < maxiter) {
      const int ordc = ord_next;
      if (iter + 1 < maxiter) ord_next = (lane < nk) ? (int)otab[(iter + 1)*nk + lane] : 0;
      int ord = ordc, ordlane;
      if (nisl > 1) {
        ord = wv_shfl_i(grow, ordc);
        ordlane = wv_shfl_i(growlane, ordc);
      } else {
        ordlane = lane_of(ord);
      }
      // ---- Nesterov extrapolation (:508-554)
      real beta = 0;
      if (iter > 0) beta = (real)(nesterov_k - 1) / (real)(nesterov_k + 2);
      if (member) {
        if (beta > 0) {
          real f_save = f;
          real fx = f_save + beta*(f_save - fprev);
          fprev = f_save;
          if (kind == 1) fx = r_clip(fx, -fl, fl);
          else if (kind == 2 && fx < 0) fx = 0;
          f = fx;
          fmom = fx;
        } else {
          fprev = f;
          fmom = f;
        }
      }

      // ---- one sweep
      real improvement = 0;
      int i = wv_bcast_i(ord, 0);
      real a = own ? ar_load(i) : 0;                   // row of the first visited constraint
      for (int bi = 0; bi < nk; bi++) {
        const real p = a*f;
        const int inext = wv_bcast_i(ord, bi + 1 < nk ? bi + 1 : bi);
        const real anext = own ? ar_load(inext) : 0;
        // chain sums of my DPP row: accA = slot 0 (lanes 0..L-1), accB = slot 1 (lanes 8..8+L-1)
        real accA = 0, accB = 0;
        if (L > 0) { accA = accA + wv_row_bcast<0>(p); accB = accB + wv_row_bcast<8>(p);
        MJH_PGS2_STEP(1) MJH_PGS2_STEP(2) MJH_PGS2_STEP(3) MJH_PGS2_STEP(4)
        MJH_PGS2_STEP(5) MJH_PGS2_STEP(6) MJH_PGS2_STEP(7)
        MJH_PGS2_END }
        // row 0 holds (r0, r1), row 1 holds (r2, r3): (r0 + r2) + (r1 + r3)
        real dot = (accA + sw_row_swap(accA)) + (accB + sw_row_swap(accB));
        if (ntail == 3) dot += wv_bcast(p, 7) + wv_bcast(p, 15) + wv_bcast(p, 23);
        else if (ntail == 2) dot += wv_bcast(p, 7) + wv_bcast(p, 15);
        else if (ntail == 1) dot += wv_bcast(p, 7);
        // every lane evaluates the update for its own constraint; only the owner of row i keeps it
        const real res = bj + dot;
        const real oldf = f;
        real fn = oldf - res*ainv;
        fn = (fn < blo) ? blo : ((fn > bhi) ? bhi : fn);
        const real delta = fn - oldf;
        real change = 0.5*delta*delta*A + delta*res;
        if (change > 1e-10) { fn = oldf; change = 0; }
        const int li = wv_bcast_i(ordlane, bi);          // owner lane of constraint i
        if (lane == li) f = fn;
        improvement -= wv_bcast(change, li);
        i = inext;
        a = anext;
      }
      improvement *= scale;

      // ---- gradient restart (:694-713): sum over the island's constraints in index order
      int restart = 0;
      if (iter > 0) {
        const real ce = (f - fmom) * (fmom - fprev);
        real dotce = 0;
        for (int q = 0; q < nk; q++) dotce += wv_bcast(ce, nisl > 1 ? wv_bcast_i(growlane, q) : lane_of(q));
        restart = (dotce < 0);
      }
      if (restart) nesterov_k = 0; else nesterov_k++;
      iter++;
      if (improvement < M.o.tolerance) break;
    }
    if (isl == 0) niter0 = iter;
  }

  // final dual state (dualState, :270-345), forces back to memory, iteration count
  if (own) {
    int st;
    if (kind == 0) st = MJH_STATE_QUADRATIC;
    else if (kind == 1) {
      if (f <= -fl) st = MJH_STATE_LINEARPOS;
      else if (f >= fl) st = MJH_STATE_LINEARNEG;
      else st = MJH_STATE_QUADRATIC;
    } else st = (f <= 0) ? MJH_STATE_SATISFIED : MJH_STATE_QUADRATIC;
    P.state[jj] = st;
    P.force[jj] = f;
  }
  if (lane == 0) counts[MJH_C_NITER] = niter0;
  wv_sync();
}
#endif  // 32-lane layout

// ------------------------------------------------------------------------------------------------
// solPGS, generic form                                            (engine_solver.c:457-741)
// Any nefc, constraint islands (one pass of the reference's loop per island, residuals over the
// full row) and elliptic cone blocks: normal or ray update followed by the friction QCQP
// (:598-672, mju_QCQP2/3/N engine_util_solve.c:1197-1443).  The residual dot products are wave
// cooperative in mju_dot's association; the small block algebra is evaluated redundantly by every
// lane and lane 0 stores the result, so the sweep stays bit-compatible with the CPU.
// ------------------------------------------------------------------------------------------------
// The friction QCQP of an elliptic block (mju_QCQP2 / QCQP3 / QCQP, engine_util_solve.c:1197-1443):
//     minimise  x'Ax/2 + x'b   subject to   sum (x_k / d_k)^2 <= r^2
// In the scaled variable y = x ./ d the constraint is a ball; the KKT system (S + lambda I) y = -c with
// S = D A D, c = D b is solved for the multiplier lambda >= 0 by Newton's method on
// phi(lambda) = |y(lambda)|^2 - r^2 (at most 20 steps, stops at 1e-10 like the reference).  One routine
// for every size: the shifted matrix is inverted through its cofactors for n = 2, 3 and through a
// Cholesky factor for n = 4, 5; the operation order of each formula is the reference's, which is what
// keeps the sweep bit-compatible.  Returns 1 when the constraint is active (lambda != 0).
struct SmallSym {          // symmetric n x n, n <= 5, full storage (row-major) so that rows are contiguous
  real a[25];
  int n;
  MJH_MEM real& at(int i, int j) { return a[i*n + j]; }
  MJH_MEM real at(int i, int j) const { return a[i*n + j]; }
};

// lower Cholesky factor in place; a pivot below `floor_` is clamped and its column zeroed; returns the
// number of pivots that passed (engine_util_solve.c:33-80)
MJH_DEV int small_chol_factor(SmallSym& m, real floor_) {
  const int n = m.n;
  int passed = 0;
  for (int c = 0; c < n; c++) {
    real piv = m.at(c, c);
    if (c) piv -= dot_ref(m.a + c*n, m.a + c*n, c);
    const int ok = !(piv < floor_);
    passed += ok;
    const real root = sqrt(ok ? piv : floor_);
    m.at(c, c) = root;
    const real inv = ok ? 1/root : (real)0;
    for (int r = c + 1; r < n; r++)
      m.at(r, c) = ok ? (m.at(r, c) - dot_ref(m.a + r*n, m.a + c*n, c)) * inv : (real)0;
  }
  return passed;
}
// out = (L L')^-1 rhs                                             (engine_util_solve.c:84-101)
MJH_DEV void small_chol_solve(real* out, const SmallSym& m, const real* rhs) {
  const int n = m.n;
  for (int i = 0; i < n; i++) {
    real v = rhs[i];
    if (i) { out[i] = v; v = out[i] - dot_ref(m.a + i*n, out, i); }
    out[i] = v / m.at(i, i);
  }
  for (int i = n - 1; i >= 0; i--) {
    real v = out[i];
    for (int j = i + 1; j < n; j++) v -= m.at(j, i) * out[j];
    out[i] = v / m.at(i, i);
  }
}

MJH_DEV int qcqp_solve(real* res, const real* Ain, const real* bin, const real* d, real r, int n) {
  // scaled problem: c = D b, S = D A D (upper triangle s[i][j], i <= j)
  real c[5], y[5];
  SmallSym S;
  S.n = n;
  for (int i = 0; i < n; i++) {
    c[i] = bin[i]*d[i];
    y[i] = 0;
    for (int j = 0; j < n; j++) S.at(i, j) = Ain[i*n + j]*d[i]*d[j];
  }
  real lambda = 0;
  for (int step = 0; step < 20; step++) {
    real slope;           // phi'(lambda) = -2 y' (S + lambda I)^-1 y
    if (n == 2) {
      const real g0 = S.at(0, 0) + lambda, g1 = S.at(1, 1) + lambda, o = S.at(0, 1);
      const real det = g0*g1 - o*o;
      if (det < 1e-10) { res[0] = 0; res[1] = 0; return 0; }
      const real inv = 1/det;
      const real i00 = g1*inv, i11 = g0*inv, i01 = -o*inv;            // inverse of the shifted matrix
      y[0] = -i00*c[0] - i01*c[1];
      y[1] = -i01*c[0] - i11*c[1];
      slope = -2.0*(i00*y[0]*y[0] + 2.0*i01*y[0]*y[1] + i11*y[1]*y[1]);
    } else if (n == 3) {
      const real g0 = S.at(0, 0) + lambda, g1 = S.at(1, 1) + lambda, g2 = S.at(2, 2) + lambda;
      const real o01 = S.at(0, 1), o02 = S.at(0, 2), o12 = S.at(1, 2);
      // cofactors (the matrix is symmetric, so is its adjugate)
      real k00 = g1*g2 - o12*o12, k11 = g0*g2 - o02*o02, k22 = g0*g1 - o01*o01;
      real k01 = o02*o12 - o01*g2, k02 = o01*o12 - o02*g1, k12 = o01*o02 - o12*g0;
      const real det = g0*k00 + o01*k01 + o02*k02;
      if (det < 1e-10) { res[0] = 0; res[1] = 0; res[2] = 0; return 0; }
      const real inv = 1/det;
      k00 *= inv; k11 *= inv; k22 *= inv; k01 *= inv; k02 *= inv; k12 *= inv;
      y[0] = -k00*c[0] - k01*c[1] - k02*c[2];
      y[1] = -k01*c[0] - k11*c[1] - k12*c[2];
      y[2] = -k02*c[0] - k12*c[1] - k22*c[2];
      slope = -2.0*(k00*y[0]*y[0] + k11*y[1]*y[1] + k22*y[2]*y[2]) - 4.0*(k01*y[0]*y[1] + k02*y[0]*y[2] + k12*y[1]*y[2]);
    } else {
      SmallSym F = S;
      for (int i = 0; i < n; i++) F.at(i, i) += lambda;
      if (small_chol_factor(F, 1e-10) < n) { for (int i = 0; i < n; i++) res[i] = 0; return 0; }
      small_chol_solve(y, F, c);
      for (int i = 0; i < n; i++) y[i] = y[i]*-1;
      real z[5];
      small_chol_solve(z, F, y);
      slope = -2.0 * dot_ref(y, z, n);
    }
    const real gap = (n == 2 ? y[0]*y[0] + y[1]*y[1] : (n == 3 ? y[0]*y[0] + y[1]*y[1] + y[2]*y[2] : dot_ref(y, y, n))) - r*r;
    if (gap < 1e-10) break;
    const real advance = -gap/slope;
    if (advance < 1e-10) break;
    lambda += advance;
  }
  for (int i = 0; i < n; i++) res[i] = y[i]*d[i];
  return lambda != 0;
}

// scale the friction part of a contact force onto / into the friction ellipsoid  (:366-381)
MJH_DEV void project_ellipsoid(real* fr, real normal, const real* mu, int dim, int feasible) {
  real ss = 0;
  for (int j = 0; j < dim - 1; j++) ss += fr[j]*fr[j] / (mu[j]*mu[j]);
  const real normal2 = normal*normal;
  if (!feasible || ss > normal2) {
    const real scl = sqrt(normal2 / r_max(MJH_MINVAL, ss));
    for (int j = 0; j < dim - 1; j++) fr[j] *= scl;
  }
}

MJH_DEVN void solve_pgs(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ne_all = counts[MJH_C_NE], nf_all = counts[MJH_C_NF];
  Efc P;
  efc_layout(M, B, e, nefc, P);
  crptr AR = P.AR;
  crptr b = P.b;
  crptr floss = P.floss;
  rptr force = P.force;
  rptr ARinv = P.ARinv;                 // [nefc] by row
  rptr force_prev = P.fprev;            // [nk] by island position
  rptr force_mom = P.fmom;              // [nk]
  iptr blockstart = P.order;            // [nblocks] island positions, shuffled in place
  iptr efclist = MJH_G(B, iscratch, e) + M.s.nefcmax;     // [nk] island position -> row
  iptr state = P.state;
  const int lane = wv_lane();
  const int maxiter = M.o.iterations;
  const real scale = 1 / (M.o.meaninertia * (real)(M.s.nv > 1 ? M.s.nv : 1));
  const int elliptic = MJH_HAS(MJH_FT_ELLIPTIC) && (M.o.cone != 0);
  // sparse path: residuals in mju_dotSparse's order over the row's stored entries (wave_dot_masked)
  const int sparse = MJH_HAS(MJH_FT_PRIMAL) && M.s.sparse;
  const int nARw = M.s.nARw;
  ciptr armask = MJH_G(B, sp_ARmask, e);

  MJH_FOR_LANES(i, nefc) ARinv[i] = 1 / AR[(size_t)i*nefc + i];
  wv_sync();

  const int nisl_raw = counts[MJH_C_NISLAND];
  const int nisl = nisl_raw > 1 ? nisl_raw : 1;
  int niter0 = 0;
  for (int isl = 0; isl < nisl; isl++) {
    // ---- efclist, per-island ne/nf, blocks (lane 0; short serial scans)
    int nk = 0, ne = 0, nf = 0, nblocks = 0;
    if (lane == 0) {
      for (int i = 0; i < nefc; i++) {
        if (nisl > 1 && P.island[i] != isl) continue;
        efclist[nk++] = i;
        if (i < ne_all) ne++; else if (i < ne_all + nf_all) nf++;
      }
      for (int c = 0; c < nk; ) {
        blockstart[nblocks++] = c;
        const int i = efclist[c];
        c += (elliptic && P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC) ? cone_dim(P, i, nefc) : 1;
      }
    }
    nk = wv_bcast_i(nk, 0); ne = wv_bcast_i(ne, 0); nf = wv_bcast_i(nf, 0); nblocks = wv_bcast_i(nblocks, 0);
    wv_sync();
    if (nk == 0) continue;
    MJH_FOR_LANES(c, nk) force_prev[c] = force[efclist[c]];
    wv_sync();

    Pcg32 rng;
    rng.state = 0; rng.inc = 1;
    pcg32_next(&rng);

    int iter = 0, nesterov_k = 0;
    while (iter < maxiter) {
      // ---- Nesterov extrapolation (:508-554)
      real beta = 0;
      if (iter > 0) beta = (real)(nesterov_k - 1) / (real)(nesterov_k + 2);
      if (beta > 0) {
        MJH_FOR_LANES(c, nk) {
          const int i = efclist[c];
          real f_save = force[i];
          real f = f_save + beta*(f_save - force_prev[c]);
          force_prev[c] = f_save;
          if (c >= ne && c < ne + nf) f = r_clip(f, -floss[i], floss[i]);
          else if (c >= ne + nf && !(elliptic && P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC) && f < 0) f = 0;
          force[i] = f;
        }
        wv_sync();
        if (elliptic) {
          // projectCone (:421-437) on every elliptic block
          MJH_FOR_LANES(c, nk) {
            const int i = efclist[c];
            if (c < ne + nf || !cone_leader(P, i)) continue;
            const int dim = cone_dim(P, i, nefc);
            if (force[i] < 0) {
              for (int j = 0; j < dim; j++) force[i+j] = 0;
            } else {
              real fr[5], mu[5];
              for (int j = 0; j < dim - 1; j++) { fr[j] = force[i+1+j]; mu[j] = P.cone[i+1+j]; }
              project_ellipsoid(fr, force[i], mu, dim, 1);
              for (int j = 0; j < dim - 1; j++) force[i+1+j] = fr[j];
            }
          }
          wv_sync();
        }
        MJH_FOR_LANES(c, nk) force_mom[c] = force[efclist[c]];
      } else {
        MJH_FOR_LANES(c, nk) {
          real f = force[efclist[c]];
          force_prev[c] = f;
          force_mom[c] = f;
        }
      }
      // ---- shuffle block order (Fisher-Yates with the shared PCG32 stream, :256-265)
      for (int i = nblocks - 1; i > 0; i--) {
        uint32_t j = pcg32_next(&rng) % (uint32_t)(i + 1);
        if (lane == 0) {
          int t = blockstart[i]; blockstart[i] = blockstart[j]; blockstart[j] = t;
        }
      }
      wv_sync();

      // ---- one sweep
      real improvement = 0;
      for (int bi = 0; bi < nblocks; bi++) {
        const int c = blockstart[bi];
        const int i = efclist[c];
        if (!(elliptic && P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC)) {
          real res = b[i] + (sparse ? wave_dot_masked(AR + (size_t)i*nefc, force, armask + 2*i*nARw, nARw) : wave_dot_ref(AR + (size_t)i*nefc, force, nefc));
          real oldf = force[i];
          real f = oldf - res*ARinv[i];
          if (c >= ne && c < ne + nf) {
            if (f < -floss[i]) f = -floss[i];
            else if (f > floss[i]) f = floss[i];
          } else if (c >= ne + nf) {
            if (f < 0) f = 0;
          }
          // costChange (:216-237) with A = 1/ARinv
          real A = 1/ARinv[i];
          real delta = f - oldf;
          real change = 0.5*delta*delta*A + delta*res;
          if (change > 1e-10) { f = oldf; change = 0; }
          improvement -= change;
          wv_sync();                  // all lanes have consumed force[] for this row
          if (lane == 0) force[i] = f;
          wv_sync();
          continue;
        }
        // ---- elliptic block (:598-672)
        const int dim = cone_dim(P, i, nefc);
        real res[6], oldforce[6], fl[6], Athis[36], mu[5];
        for (int j = 0; j < dim; j++) {
          res[j] = b[i+j] + (sparse ? wave_dot_masked(AR + (size_t)(i+j)*nefc, force, armask + 2*(i+j)*nARw, nARw) : wave_dot_ref(AR + (size_t)(i+j)*nefc, force, nefc));
          oldforce[j] = force[i+j];
          fl[j] = oldforce[j];
          for (int k = 0; k < dim; k++) Athis[j*dim + k] = AR[(size_t)(i+j)*nefc + i + k];
        }
        for (int j = 0; j < dim - 1; j++) mu[j] = P.cone[i+1+j];
        if (fl[0] < MJH_MINVAL) {
          // normal update
          fl[0] -= res[0]*ARinv[i];
          if (fl[0] < 0) fl[0] = 0;
          for (int j = 1; j < dim; j++) fl[j] = 0;
        } else {
          // ray update
          real v[6], v1[6];
          for (int j = 0; j < dim; j++) v[j] = fl[j];
          for (int j = 0; j < dim; j++) v1[j] = dot_ref(Athis + j*dim, v, dim);
          const real denom = dot_ref(v, v1, dim);
          if (denom >= MJH_MINVAL) {
            real x = -dot_ref(v, res, dim) / denom;
            if (fl[0] + x*v[0] < 0) x = -v[0]/fl[0];
            for (int j = 0; j < dim; j++) fl[j] += x*v[j];
          }
        }
        // friction update with the normal force fixed
        real bc[5], Ac[25];
        for (int j = 0; j < dim - 1; j++) bc[j] = res[j+1];
        for (int j = 0; j < dim - 1; j++) {
          for (int k = 0; k < dim - 1; k++) Ac[j*(dim-1) + k] = Athis[(j+1)*dim + 1 + k];
          bc[j] -= dot_ref(Ac + j*(dim-1), oldforce + 1, dim - 1);
          bc[j] += Athis[(j+1)*dim]*(fl[0] - oldforce[0]);
        }
        if (fl[0] < MJH_MINVAL) {
          for (int j = 1; j < dim; j++) fl[j] = 0;
        } else {
          real v[5];
          const int active = qcqp_solve(v, Ac, bc, mu, fl[0], dim - 1);
          if (active) project_ellipsoid(v, fl[0], mu, dim, 0);
          for (int j = 0; j < dim - 1; j++) fl[1+j] = v[j];
        }
        // costChange, block form
        real delta[6];
        for (int j = 0; j < dim; j++) delta[j] = fl[j] - oldforce[j];
        real quadf = 0;
        for (int j = 0; j < dim; j++) quadf += delta[j] * dot_ref(Athis + j*dim, delta, dim);
        real change = 0.5*quadf + dot_ref(delta, res, dim);
        if (change > 1e-10) {
          for (int j = 0; j < dim; j++) fl[j] = oldforce[j];
          change = 0;
        }
        improvement -= change;
        wv_sync();
        if (lane == 0) for (int j = 0; j < dim; j++) force[i+j] = fl[j];
        wv_sync();
      }
      improvement *= scale;

      // ---- gradient restart (:694-713)
      int restart = 0;
      if (iter > 0) {
        real dotce = 0;
        for (int c = 0; c < nk; c++) {
          const int i = efclist[c];
          real correction = force[i] - force_mom[c];
          real extrapolation = force_mom[c] - force_prev[c];
          dotce += correction * extrapolation;
        }
        restart = (dotce < 0);
      }
      if (restart) nesterov_k = 0; else nesterov_k++;
      iter++;
      if (improvement < M.o.tolerance) break;
      wv_sync();
    }
    wv_sync();
    if (isl == 0) niter0 = iter;

    // final dual state of this island (dualState, :270-345)
    MJH_FOR_LANES(c, nk) {
      const int i = efclist[c];
      int st;
      if (c < ne) st = MJH_STATE_QUADRATIC;
      else if (c < ne + nf) {
        if (force[i] <= -floss[i]) st = MJH_STATE_LINEARPOS;
        else if (force[i] >= floss[i]) st = MJH_STATE_LINEARNEG;
        else st = MJH_STATE_QUADRATIC;
      } else if (elliptic && P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC) {
        if (!cone_leader(P, i)) continue;
        const int dim = cone_dim(P, i, nefc);
        const real mu = P.cone[i];
        real f[6];
        f[0] = force[i]/mu;
        for (int j = 1; j < dim; j++) f[j] = force[i+j]/P.cone[i+j];
        const real N = f[0];
        const real T = sqrt(dot_ref(f + 1, f + 1, dim - 1));
        if (mu*N >= T) st = MJH_STATE_SATISFIED;
        else if (N + mu*T <= 0) st = MJH_STATE_QUADRATIC;
        else st = MJH_STATE_CONE;
        for (int j = 1; j < dim; j++) state[i+j] = st;
      } else st = (force[i] <= 0) ? MJH_STATE_SATISFIED : MJH_STATE_QUADRATIC;
      state[i] = st;
    }
    wv_sync();
  }
  if (lane == 0) counts[MJH_C_NITER] = niter0;
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// mj_solNoSlip / mj_solNoSlip_island                 (engine_solver.c:764-972)
// A Gauss-Seidel pass over the friction rows only (dry friction, then the friction dimensions of every contact) with the
// regulariser R taken out of efc_AR's diagonal; runs after the main solver -- PGS, CG or Newton -- on the forces it
// left, per island when islands are in use.  Dense constraint path (residuals are mju_dot over a dense efc_AR row).
// ------------------------------------------------------------------------------------------------
MJH_DEVN void solve_noslip(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ne_all = counts[MJH_C_NE], nf_all = counts[MJH_C_NF];
  Efc P;
  efc_layout(M, B, e, nefc, P);
  crptr AR = P.AR;
  crptr b = P.b;
  crptr R = P.R;
  crptr floss = P.floss;
  rptr force = P.force;
  rptr ARinv = P.ARinv;                                   // [nk] by island position
  iptr efclist = MJH_G(B, iscratch, e) + M.s.nefcmax;     // [nk] island position -> row
  iptr state = P.state;
  const int lane = wv_lane();
  const int maxiter = M.o.noslip_iterations;
  const real scale = 1 / (M.o.meaninertia * (real)(M.s.nv > 1 ? M.s.nv : 1));
  const int elliptic = MJH_HAS(MJH_FT_ELLIPTIC) && (M.o.cone != 0);
  // residual(..., flg_subR = 1): b + AR row . force - R force
  auto resid = [&](int i) -> real { return (b[i] + wave_dot_ref(AR + (size_t)i*nefc, force, nefc)) - R[i]*force[i]; };
  // extractBlock(..., flg_subR = 1): the diagonal block with R taken out, its diagonal clamped to 1e-10 from below
  auto block = [&](real* Ac, int start, int n) {
    for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) Ac[j*n + k] = AR[(size_t)(start + j)*nefc + start + k];
    for (int j = 0; j < n; j++) { real dd = Ac[j*(n + 1)] - R[start + j]; Ac[j*(n + 1)] = dd > 1e-10 ? dd : (real)1e-10; }
  };
  // costChange with dim >= 2 (:216-237); restores the old forces on a positive change
  auto cost_change = [&](const real* A, real* f, const real* oldf, const real* res, int dim) -> real {
    real delta[6];
    for (int j = 0; j < dim; j++) delta[j] = f[j] - oldf[j];
    real quadf = 0;
    for (int j = 0; j < dim; j++) quadf += delta[j] * dot_ref(A + j*dim, delta, dim);
    real change = 0.5*quadf + dot_ref(delta, res, dim);
    if (change > 1e-10) { for (int j = 0; j < dim; j++) f[j] = oldf[j]; change = 0; }
    return change;
  };

  const int nisl_raw = counts[MJH_C_NISLAND];
  const int nisl = nisl_raw > 1 ? nisl_raw : 1;
  for (int isl = 0; isl < nisl; isl++) {
    int nk = 0, ne = 0, nf = 0;
    if (lane == 0) {
      for (int i = 0; i < nefc; i++) {
        if (nisl > 1 && P.island[i] != isl) continue;
        efclist[nk++] = i;
        if (i < ne_all) ne++; else if (i < ne_all + nf_all) nf++;
      }
    }
    nk = wv_bcast_i(nk, 0); ne = wv_bcast_i(ne, 0); nf = wv_bcast_i(nf, 0);
    wv_sync();
    if (nk == 0) continue;
    // ARdiaginv(..., flg_subR = 1)
    MJH_FOR_LANES(c, nk) { const int i = efclist[c]; real dd = AR[(size_t)i*nefc + i] - R[i]; if (dd < MJH_MINVAL) dd = MJH_MINVAL; ARinv[c] = 1/dd; }
    // dualState (:270-345)
    auto dual_state = [&]() {
      MJH_FOR_LANES(c, nk) {
        const int i = efclist[c];
        int st;
        if (c < ne) st = MJH_STATE_QUADRATIC;
        else if (c < ne + nf) {
          if (force[i] <= -floss[i]) st = MJH_STATE_LINEARPOS;
          else if (force[i] >= floss[i]) st = MJH_STATE_LINEARNEG;
          else st = MJH_STATE_QUADRATIC;
        } else if (elliptic && P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC) {
          if (!cone_leader(P, i)) continue;
          const int dim = cone_dim(P, i, nefc);
          const real mu = P.cone[i];
          real f[6];
          f[0] = force[i]/mu;
          for (int j = 1; j < dim; j++) f[j] = force[i+j]/P.cone[i+j];
          const real N = f[0];
          const real T = sqrt(dot_ref(f + 1, f + 1, dim - 1));
          if (mu*N >= T) st = MJH_STATE_SATISFIED;
          else if (N + mu*T <= 0) st = MJH_STATE_QUADRATIC;
          else st = MJH_STATE_CONE;
          for (int j = 1; j < dim; j++) state[i+j] = st;
        } else st = (force[i] <= 0) ? MJH_STATE_SATISFIED : MJH_STATE_QUADRATIC;
        state[i] = st;
      }
      wv_sync();
    };
    dual_state();

    int iter = 0;
    while (iter < maxiter) {
      real improvement = 0;
      // correct for the cost change at iteration 0 (the regulariser's share of the cost the main solver minimised)
      if (iter == 0) for (int c = 0; c < nk; c++) { const int i = efclist[c]; improvement += 0.5*force[i]*force[i]*R[i]; }
      // dry friction
      for (int c = ne; c < ne + nf; c++) {
        const int i = efclist[c];
        const real res = resid(i);
        const real oldf = force[i];
        real f = oldf - res*ARinv[c];
        if (f < -floss[i]) f = -floss[i];
        else if (f > floss[i]) f = floss[i];
        const real delta = f - oldf;
        improvement -= 0.5*delta*delta/ARinv[c] + delta*res;
        wv_sync();
        if (lane == 0) force[i] = f;
        wv_sync();
      }
      // contact friction
      for (int c = ne + nf; c < nk; c++) {
        const int i = efclist[c];
        if (P.type[i] == MJH_CNSTR_CONTACT_PYRAMIDAL) {
          // (contact.dim from the rows: 2 (dim - 1) pyramid edges share the contact's id; the contact records' LDS
          // slots are no longer live at this stage)
          int nrow = 1;
          while (i + nrow < nefc && P.type[i + nrow] == MJH_CNSTR_CONTACT_PYRAMIDAL && P.id[i + nrow] == P.id[i]) nrow++;
          const int dim = nrow/2 + 1;
          // pairs of opposing pyramid edges
          for (int j = i; j < i + 2*(dim - 1); j += 2) {
            real res[2], oldforce[2], fl[2], Ac[4], bc[2];
            res[0] = resid(j); res[1] = resid(j + 1);
            oldforce[0] = force[j]; oldforce[1] = force[j + 1];
            block(Ac, j, 2);
            for (int k = 0; k < 2; k++) bc[k] = res[k] - dot_ref(Ac + k*2, oldforce, 2);
            const real mid = 0.5*(oldforce[0] + oldforce[1]);
            real y = 0.5*(oldforce[0] - oldforce[1]);
            const real K1 = Ac[0] + Ac[3] - Ac[1] - Ac[2];
            const real K0 = mid*(Ac[0] - Ac[3]) + bc[0] - bc[1];
            if (K1 < MJH_MINVAL) { fl[0] = mid; fl[1] = mid; }
            else {
              y = -K0/K1;
              if (y < -mid) { fl[0] = 0; fl[1] = 2*mid; }
              else if (y > mid) { fl[0] = 2*mid; fl[1] = 0; }
              else { fl[0] = mid + y; fl[1] = mid - y; }
            }
            improvement -= cost_change(Ac, fl, oldforce, res, 2);
            wv_sync();
            if (lane == 0) { force[j] = fl[0]; force[j + 1] = fl[1]; }
            wv_sync();
          }
          c += 2*(dim - 1) - 1;
        } else if (elliptic && P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC) {
          const int dim = cone_dim(P, i, nefc);
          real res[5], oldforce[5], fl[5], Ac[25], bc[5], mu[5];
          for (int j = 0; j < dim - 1; j++) { res[j] = resid(i + 1 + j); oldforce[j] = force[i + 1 + j]; mu[j] = P.cone[i + 1 + j]; }
          block(Ac, i + 1, dim - 1);
          for (int j = 0; j < dim - 1; j++) bc[j] = res[j] - dot_ref(Ac + j*(dim - 1), oldforce, dim - 1);
          if (force[i] < MJH_MINVAL) { for (int j = 0; j < dim - 1; j++) fl[j] = 0; }
          else {
            // solveQCQP (:388-410)
            const int active = qcqp_solve(fl, Ac, bc, mu, force[i], dim - 1);
            if (active) project_ellipsoid(fl, force[i], mu, dim, 0);
          }
          if (dim == 2) {
            // costChange, dim == 1 (one friction dimension cannot occur with condim 3 / 4 / 6; kept for completeness)
            const real delta = fl[0] - oldforce[0];
            real change = 0.5*delta*delta*Ac[0] + delta*res[0];
            if (change > 1e-10) { fl[0] = oldforce[0]; change = 0; }
            improvement -= change;
          } else improvement -= cost_change(Ac, fl, oldforce, res, dim - 1);
          wv_sync();
          if (lane == 0) for (int j = 0; j < dim - 1; j++) force[i + 1 + j] = fl[j];
          wv_sync();
          c += dim - 1;
        }
      }
      // dualStateChange: the states follow the forces (counts of active / changed rows are statistics only)
      dual_state();
      improvement *= scale;
      iter++;
      if (improvement < M.o.noslip_tolerance) break;
    }
    // solver_niter of island 0 accumulates the noslip iterations after the main solver's
    if (isl == 0 && lane == 0) counts[MJH_C_NITER] += iter;
    wv_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// mj_fwdConstraint (PGS path)                      (engine_forward.c:1148-1252, warmstart :1056-1132)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void solve_newton(MREF M_, BREF B_, int e_);
MJH_DEVN void solve_cg(MREF M_, BREF B_, int e_);
MJH_DEVN void stage_fwd_constraint(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC];
  rptr qfc = MJH_F(B, qfrc_constraint, e);
  crptr qas = MJH_F(B, qacc_smooth, e);

  if (!nefc) {
    MJH_FOR_LANES(i, nv) qfc[i] = 0;
    if (wv_lane() == 0) counts[MJH_C_NITER] = 0;
    wv_sync();
    return;
  }
  Efc P;
  efc_layout(M, B, e, nefc, P);
  crptr J = P.J;
  crptr aref = P.aref;
  crptr AR = P.AR;
  rptr eb = P.b;
  rptr force = P.force;
  rptr jar = P.jar;                     // [nefc]
  rptr ARf = P.ARf;                     // [nefc]
  crptr qws = MJH_F(B, qacc_warmstart, e);

#ifdef MJH_PROFILE
  long long pc0 = wv_clock();
#define MJH_SUBPROF(slot) do { long long c_ = wv_clock(); if (wv_lane() == 0) MJH_G(B, prof, e)[slot] += (real)(c_ - pc0)*0.01; pc0 = c_; } while (0)
#else
#define MJH_SUBPROF(slot) do {} while (0)
#endif
  // efc_b = J*qacc_smooth - aref ; jar = J*qacc_warmstart - aref
  if (MJH_HAS(MJH_FT_PRIMAL) && s.csr) {
    MJH_FOR_LANES(r, nefc) {
      eb[r] = csr_row_dot(P, r, qas) - aref[r];
      jar[r] = csr_row_dot(P, r, qws) - aref[r];
    }
  } else if (MJH_HAS(MJH_FT_PRIMAL) && s.sparse) {
    MJH_FOR_LANES(r, nefc) {
      eb[r] = sp_row_dot(P, r, qas) - aref[r];
      jar[r] = sp_row_dot(P, r, qws) - aref[r];
    }
  } else
  MJH_FOR_LANES(r, nefc) {
    crptr Jr = J + (size_t)r*nv;
    real t = dot_ref(Jr, qas, nv);
    eb[r] = t - aref[r];
    real u = dot_ref(Jr, qws, nv);
    jar[r] = u - aref[r];
  }
  wv_sync();

  if (MJH_HAS(MJH_FT_PRIMAL) && M.o.solver != MJH_SOL_PGS) {
    // primal solvers (mjh_newton.h) leave qacc, qfrc_constraint, efc_force/state
    wv_sync();
    if (M.o.solver == MJH_SOL_NEWTON) solve_newton(M, B, e); else solve_cg(M, B, e);
    if (M.o.noslip_iterations > 0) {
      // mj_solNoSlip on the forces the primal solver left, then mj_dualFinish's first half; stage_finish solves for qacc
      solve_noslip(M, B, e);
      MJH_FOR_LANES(j, nv) {
        real acc = 0;
        for (int r = 0; r < nefc; r++) { const real f = force[r]; if (f != 0) acc += J[(size_t)r*nv + j]*f; }
        qfc[j] = acc;
      }
      wv_sync();
    }
    return;
  }
  if (!(M.o.disableflags & (1<<9))) {
    constraint_update(B, e, P, jar, 0, MJH_HAS(MJH_FT_ELLIPTIC) && M.o.cone != 0);        // efc_force(qacc_warmstart), syncs internally
    // PGS_warmstart = f.b + 0.5 f.AR.f ; keep the warmstart forces only if that is <= 0
    if (MJH_HAS(MJH_FT_PRIMAL) && s.sparse) {
      // mju_mulMatVecSparse(ARf, efc_AR, force): one mju_dotSparse per row over its stored entries
      ciptr armask = MJH_G(B, sp_ARmask, e);
      const int nARw = s.nARw;
      MJH_FOR_LANES(r, nefc) {
        ciptr mw = armask + 2*r*nARw;
        int nnz = 0;
        for (int w = 0; w < nARw; w++) nnz += __builtin_popcount((unsigned)mw[2*w]) + __builtin_popcount((unsigned)mw[2*w + 1]);
        const int n4 = nnz & ~3;
        crptr a = AR + (size_t)r*nefc;
        real acc[4] = {0, 0, 0, 0}, tl = 0;
        int k = 0;
        real res = 0;
        for (int w = 0; w < nARw; w++) {
          unsigned long long m = ((unsigned long long)(unsigned)mw[2*w + 1] << 32) | (unsigned)mw[2*w];
          while (m) {
            const int j = 64*w + __builtin_ctzll(m);
            m &= m - 1;
            if (k < n4) acc[k & 3] += a[j]*force[j];
            else { if (k == n4) res = (acc[0] + acc[2]) + (acc[1] + acc[3]); res += a[j]*force[j]; }
            k++;
          }
        }
        if (nnz == n4) res = (acc[0] + acc[2]) + (acc[1] + acc[3]);
        (void)tl;
        ARf[r] = res;
      }
    } else
    MJH_FOR_LANES(r, nefc) ARf[r] = dot_ref(AR + (size_t)r*nefc, force, nefc);
    wv_sync();
    real pgs_ws = wave_dot_ref(force, eb, nefc);
    pgs_ws += 0.5*wave_dot_ref(force, ARf, nefc);
    wv_sync();
    if (pgs_ws > 0) {
      MJH_FOR_LANES(r, nefc) force[r] = 0;
    }
  } else {
    MJH_FOR_LANES(r, nefc) force[r] = 0;
  }
  wv_sync();
  MJH_SUBPROF(22);     // efc_b, jar, warm start

#if !MJH_LANE_MODE && MJH_W >= 32
  if (MJH_HAS(MJH_FT_PRIMAL) && s.sparse) {
    // (the register-resident sweeps group a row's products by dense index; the reference's sparse sweep groups them
    // by position among the row's stored entries: the generic sweep takes those through wave_dot_masked)
    solve_pgs(M, B, e);
  } else
#endif
#if !MJH_LANE_MODE && MJH_W == 64
  if (nefc > 64 && nefc <= 128 && nefc <= M.s.pgs_nmax && M.o.iterations <= M.s.pgs_iters && counts[MJH_C_NISLAND] <= 1 &&
      (!MJH_HAS(MJH_FT_ELLIPTIC) || M.o.cone == 0)) {
    if (B.pgs_mode == 1) solve_pgs_resid_wide(M, B, e);      // (opt-in: the residual-update sweep, tolerance parity)
    else solve_pgs_wide(M, B, e);
  } else
#endif
#if !MJH_LANE_MODE && MJH_W == 64
  if (B.pgs_mode == 1 && nefc <= MJH_W && M.o.iterations <= M.s.pgs_iters && (!MJH_HAS(MJH_FT_ELLIPTIC) || M.o.cone == 0)) {
    // (opt-in: the residual-update sweep, tolerance parity)
#if defined(MJH_HOSTSIM)
    solve_pgs_resid<0>(M, B, e);
#else
    if (mjh_in_lds(P.AR)) solve_pgs_resid<1>(M, B, e);
    else solve_pgs_resid<0>(M, B, e);
#endif
  } else
#endif
#if !MJH_LANE_MODE && MJH_W >= 32
  if (nefc <= MJH_W && M.o.iterations <= M.s.pgs_iters && (!MJH_HAS(MJH_FT_ELLIPTIC) || M.o.cone == 0)) {
#if defined(MJH_HOSTSIM)
    solve_pgs_fast<0>(M, B, e);
#else
#if MJH_W == 64 && !defined(MJH_PGS_NO_LSPEC)
    if (mjh_in_lds(P.AR)) {
      // (chain length nefc/4 and tail length nefc%4 as compile-time constants: 65 instances)
      switch (nefc) {
#define MJH_PGS_CASE(n_) case n_: solve_pgs_fast<1, (n_) / 4, (n_) % 4>(M, B, e); break;
#define MJH_PGS_CASE8(b_) MJH_PGS_CASE(b_) MJH_PGS_CASE(b_ + 1) MJH_PGS_CASE(b_ + 2) MJH_PGS_CASE(b_ + 3) \
                          MJH_PGS_CASE(b_ + 4) MJH_PGS_CASE(b_ + 5) MJH_PGS_CASE(b_ + 6) MJH_PGS_CASE(b_ + 7)
        MJH_PGS_CASE8(0) MJH_PGS_CASE8(8) MJH_PGS_CASE8(16) MJH_PGS_CASE8(24)
        MJH_PGS_CASE8(32) MJH_PGS_CASE8(40) MJH_PGS_CASE8(48) MJH_PGS_CASE8(56)
#undef MJH_PGS_CASE8
#undef MJH_PGS_CASE
        default: solve_pgs_fast<1, 16, 0>(M, B, e); break;
      }
    }
#else
    if (mjh_in_lds(P.AR)) solve_pgs_fast<1>(M, B, e);
#endif
    else solve_pgs_fast<0>(M, B, e);
#endif
  } else
#endif
  {
    solve_pgs(M, B, e);
  }
  MJH_SUBPROF(23);     // PGS
  if (MJH_HAS(MJH_FT_PRIMAL) && M.o.noslip_iterations > 0) solve_noslip(M, B, e);

  // mj_dualFinish, first half (engine_solver.c:72-85): qfrc_constraint = J' f
  MJH_FOR_LANES(j, nv) {
    real acc = 0;
    for (int r = 0; r < nefc; r++) {
      real f = force[r];
      if (f != 0) acc += J[(size_t)r*nv + j]*f;
    }
    qfc[j] = acc;
  }
  wv_sync();
  MJH_SUBPROF(24);     // J' f
}

// mj_dualFinish, second half: qacc = M \ qfrc_constraint + qacc_smooth   (engine_solver.c:80-84)
MJH_DEVN void stage_finish(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const int nv = M.s.nv;
  const int nefc = MJH_F(B, counts, e)[MJH_C_NEFC];
  crptr qfc = MJH_F(B, qfrc_constraint, e);
  crptr qas = MJH_F(B, qacc_smooth, e);
  rptr qacc = MJH_F(B, qacc, e);
  iptr counts = MJH_F(B, counts, e);
  if (wv_lane() == 0) counts[MJH_C_PAIRED] = 0;
  if (!nefc) {
    MJH_FOR_LANES(i, nv) qacc[i] = qas[i];
    wv_sync();
    return;
  }
  // the primal solvers work on qacc itself -- unless the noslip pass changed the forces: mj_dualFinish then
  if (MJH_HAS(MJH_FT_PRIMAL) && M.o.solver != MJH_SOL_PGS && !(M.o.noslip_iterations > 0)) return;
  MJH_FOR_LANES(j, nv) qacc[j] = qfc[j];
#if !MJH_LANE_MODE && MJH_W == 64
  // mj_Euler with joint damping solves (M + h*diag(B)) qe = qfrc_smooth + qfrc_constraint right after
  // this stage: an independent system of the same sparsity, so it shares the pass (solve_ld_pair).
  // Its factor goes to a slot of its own (the constraint arrays are dead by now).
  if (pairs_euler_solve(M, B)) {
    const MJH_CONST_AS DSizes& s = M.s;
    rptr qH = MJH_F(B, qH2, e);
    rptr qHDiagInv = MJH_F(B, qH2DiagInv, e);
    crptr fs = MJH_F(B, qfrc_smooth, e);
    rptr qe = MJH_F(B, qe, e);
    MJH_FOR_LANES(i, nv) qe[i] = fs[i] + qfc[i];
    crptr qHDg = MJH_G(B, qH2DiagInv, e);
    if (pairs_euler_factor(M, B, e)) {
      // the factor stage_factor_m parked in the global home (nothing to fetch if that is where qH2 lives)
      crptr qHg = MJH_G(B, qH2, e);
      if (qH.p != qHg.p) MJH_FOR_LANES(k, s.nC) qH[k] = qHg[k];
      if (qHDiagInv.p != qHDg.p) MJH_FOR_LANES(i, nv) qHDiagInv[i] = qHDg[i];
      wv_sync();
    } else {
      const real h = M.o.timestep;
      crptr Mq = MJH_G(B, M, e);
      crptr qvel = MJH_F(B, qvel, e);
      MJH_FOR_LANES(k, s.nC) qH[k] = Mq[k];
      wv_sync();
      MJH_FOR_LANES(i, nv) {
        real dd = poly_force_deriv(M.dof_damping_eff[i], M.dof_dampingpoly_eff + 2*i, qvel[i], 1);
        qH[M.M_rowadr[i] + M.M_rownnz[i] - 1] += h * dd;
      }
      wv_sync();
      factor_ld(M, qH, qHDiagInv);
    }
    solve_ld_two(M, qacc, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e), qe, qH, qHDiagInv);
    if (wv_lane() == 0) counts[MJH_C_PAIRED] = 1;
    MJH_FOR_LANES(j, nv) qacc[j] += qas[j];
    wv_sync();
    return;
  }
#endif
  wv_sync();
  solve_ld(M, qacc, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e));
  MJH_FOR_LANES(j, nv) qacc[j] += qas[j];
  wv_sync();
}
