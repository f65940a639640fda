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

// ------------------------------------------------------------------------------------------------
// solPGS, scalar blocks (pyramidal / frictionless / limits / friction loss)   (engine_solver.c:457-741)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void solve_pgs(const DModel& M, const DBatch& B, int e) {
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
MJH_DEVN void stage_fwd_constraint(const DModel& M, const DBatch& B, int e) {
  const DSizes& s = M.s;
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

  // efc_b = J*qacc_smooth - aref ; jar = J*qacc_warmstart - aref
  MJH_FOR_LANES(r, nefc) {
    crptr Jr = J + (size_t)r*nv;
    real t = dot_ref(Jr, qas, nv);
    eb[r] = t - aref[r];
    real u = dot_ref(Jr, qws, nv);
    jar[r] = u - aref[r];
  }
  wv_sync();

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

  solve_pgs(M, B, e);

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
}

// mj_dualFinish, second half: qacc = M \ qfrc_constraint + qacc_smooth   (engine_solver.c:80-84)
MJH_DEVN void stage_finish(const DModel& M, const DBatch& B, int e) {
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
  MJH_FOR_LANES(j, nv) qacc[j] = qfc[j];
  wv_sync();
  solve_ld(M, qacc, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e));
  MJH_FOR_LANES(j, nv) qacc[j] += qas[j];
  wv_sync();
}
