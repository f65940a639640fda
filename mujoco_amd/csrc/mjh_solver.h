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

// PCG32, engine_solver.c:241-254
struct Pcg32 { uint64_t state, inc; };
MJH_DEV uint32_t pcg32_next(Pcg32* rng) {
  uint64_t old = rng->state;
  rng->state = old * 6364136223846793005ULL + (rng->inc | 1);
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}

#if !MJH_LANE_MODE
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
template <int ARL>
MJH_DEVN void solve_pgs_fast(MREF M_, BREF B_, int e_) {
  const auto& M = wv_uniform_ref(M_);
  BREF B = B_;
  const int e = wv_uniform_i(e_);
  iptr counts = MJH_F(B, counts, e);
  const int n = wv_uniform_i(counts[MJH_C_NEFC]), ne = wv_uniform_i(counts[MJH_C_NE]), nf = wv_uniform_i(counts[MJH_C_NF]);
  Efc P;
  efc_layout(M, B, e, n, P);
  const int lane = wv_lane();
  const int n4 = n & ~3, L = n4 >> 2, ntail = n - n4;
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
    for (int bi = 0; bi < nk; bi++) {
      const real p = a*f;
      // prefetch the next row while this one is reduced (issued after the product so that the
      // wait for the current row does not also drain this load)
      int inext = wv_bcast_i(ord, bi + 1 < nk ? bi + 1 : bi);
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
      real dot = (wv_bcast(acc, 0) + wv_bcast(acc, 32)) + (wv_bcast(acc, 16) + wv_bcast(acc, 48));
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
      i = inext;
      a = anext;
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
#endif  // !MJH_LANE_MODE

// ------------------------------------------------------------------------------------------------
// solPGS, scalar blocks (pyramidal / frictionless / limits / friction loss)   (engine_solver.c:457-741)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void solve_pgs(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ne = counts[MJH_C_NE], nf = counts[MJH_C_NF];
  Efc P;
  efc_layout(M, B, e, nefc, P);
  crptr AR = P.AR;
  crptr b = P.b;
  crptr floss = P.floss;
  rptr force = P.force;
  rptr ARinv = P.ARinv;                 // [nefc]
  rptr force_prev = P.fprev;            // [nefc]
  rptr force_mom = P.fmom;              // [nefc]
  iptr order = P.order;                  // [nefc] block visitation order (persists across iterations)
  const int lane = wv_lane();
  const int maxiter = M.o.iterations;
  const real scale = 1 / (M.o.meaninertia * (real)(M.s.nv > 1 ? M.s.nv : 1));

  MJH_FOR_LANES(i, nefc) {
    ARinv[i] = 1 / AR[(size_t)i*nefc + i];
    force_prev[i] = force[i];
    order[i] = i;
  }
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
      MJH_FOR_LANES(i, nefc) {
        real f_save = force[i];
        real f = f_save + beta*(f_save - force_prev[i]);
        force_prev[i] = f_save;
        if (i >= ne && i < ne + nf) f = r_clip(f, -floss[i], floss[i]);
        else if (i >= ne + nf && f < 0) f = 0;
        force[i] = f;
        force_mom[i] = f;
      }
    } else {
      MJH_FOR_LANES(i, nefc) {
        real f = force[i];
        force_prev[i] = f;
        force_mom[i] = f;
      }
    }
    // ---- shuffle block order (Fisher-Yates with the shared PCG32 stream, :256-265)
    // every lane advances its own copy of the generator identically; lane 0 owns the array
    for (int i = nefc - 1; i > 0; i--) {
      uint32_t j = pcg32_next(&rng) % (uint32_t)(i + 1);
      if (lane == 0) {
        int t = order[i]; order[i] = order[j]; order[j] = t;
      }
    }
    wv_sync();

    // ---- one sweep
    real improvement = 0;
    for (int bi = 0; bi < nefc; bi++) {
      const int i = order[bi];
      real res = b[i] + wave_dot_ref(AR + (size_t)i*nefc, force, nefc);
      real oldf = force[i];
      real f = oldf - res*ARinv[i];
      if (i >= ne && i < ne + nf) {
        if (f < -floss[i]) f = -floss[i];
        else if (f > floss[i]) f = floss[i];
      } else if (i >= ne + nf) {
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
    }
    improvement *= scale;

    // ---- gradient restart (:694-713)
    int restart = 0;
    if (iter > 0) {
      real dotce = 0;
      for (int i = 0; i < nefc; i++) {
        real correction = force[i] - force_mom[i];
        real extrapolation = force_mom[i] - force_prev[i];
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

  // final dual state (dualState, :270-345) and iteration count
  iptr state = P.state;
  MJH_FOR_LANES(i, nefc) {
    int st;
    if (i < ne) st = MJH_STATE_QUADRATIC;
    else if (i < ne + nf) {
      if (force[i] <= -floss[i]) st = MJH_STATE_LINEARPOS;
      else if (force[i] >= floss[i]) st = MJH_STATE_LINEARNEG;
      else st = MJH_STATE_QUADRATIC;
    } else st = (force[i] <= 0) ? MJH_STATE_SATISFIED : MJH_STATE_QUADRATIC;
    state[i] = st;
  }
  if (lane == 0) counts[MJH_C_NITER] = iter;
  wv_sync();
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
  MJH_FOR_LANES(r, nefc) {
    crptr Jr = J + (size_t)r*nv;
    real t = dot_ref(Jr, qas, nv);
    eb[r] = t - aref[r];
    real u = dot_ref(Jr, qws, nv);
    jar[r] = u - aref[r];
  }
  wv_sync();

  if (M.o.solver != MJH_SOL_PGS) {
    // primal solvers (mjh_newton.h) leave qacc, qfrc_constraint, efc_force/state
    wv_sync();
    if (M.o.solver == MJH_SOL_NEWTON) solve_newton(M, B, e); else solve_cg(M, B, e);
    return;
  }
  if (!(M.o.disableflags & (1<<9))) {
    constraint_update(B, e, P, jar, 0);        // efc_force(qacc_warmstart), syncs internally
    // PGS_warmstart = f.b + 0.5 f.AR.f ; keep the warmstart forces only if that is <= 0
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

#if !MJH_LANE_MODE
  if (nefc <= 64 && M.o.iterations <= M.s.pgs_iters) {
#if defined(MJH_HOSTSIM)
    solve_pgs_fast<0>(M, B, e);
#else
    if (mjh_in_lds(P.AR)) solve_pgs_fast<1>(M, B, e);
    else solve_pgs_fast<0>(M, B, e);
#endif
  } else
#endif
  {
    // the generic sweep is monolithic: several islands need the per-island loop of the fast path
    if (counts[MJH_C_NISLAND] > 1 && wv_lane() == 0) MJH_F(B, warning, e)[MJH_WARN_UNSUPPORTED]++;
    solve_pgs(M, B, e);
  }
  MJH_SUBPROF(23);     // PGS

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
  if (!nefc) {
    MJH_FOR_LANES(i, nv) qacc[i] = qas[i];
    wv_sync();
    return;
  }
  if (M.o.solver != MJH_SOL_PGS) return;         // the primal solvers work on qacc itself
  MJH_FOR_LANES(j, nv) qacc[j] = qfc[j];
  wv_sync();
  solve_ld(M, qacc, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e));
  MJH_FOR_LANES(j, nv) qacc[j] += qas[j];
  wv_sync();
}
