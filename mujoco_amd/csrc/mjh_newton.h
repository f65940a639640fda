// Primal Newton solver (mj_solNewton / mj_solPrimal with flg_Newton, engine_solver.c:2344-2587):
// dense Jacobian, scalar constraint rows (friction loss, limits, frictionless and pyramidal
// contacts), exact line search (PrimalSearch :1856-2054), dense Hessian H = M + J' D_active J
// recomputed and Cholesky-factorised every iteration.
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)
//
// Parity note.  Unlike the PGS path this is NOT an operation-for-operation restatement: sums are
// wave reductions, the Hessian is rebuilt instead of rank-1 updated, and several islands are solved
// as one problem (the cost is separable across islands, so the minimiser is the same; only the
// moment of termination can differ).  It converges to the reference's solution within the solver
// tolerance, which keeps qpos/qvel inside the 1e-6 bar; solver_niter is not guaranteed to match.

// wave-uniform sum of one double per lane
MJH_DEV real wv_sum_d(real v) {
#if MJH_LANE_MODE
  return v;
#else
  for (int m = MJH_W/2; m >= 1; m >>= 1) v += wv_shfl_xor(v, m);
  return v;
#endif
}

// cost of one scalar row at residual x, with first/second derivative  (mj_constraintUpdate_impl,
// engine_core_constraint.c:3275-3420).  kind: 0 equality, 1 friction loss, 2 inequality
MJH_DEV real nt_row_cost(int kind, real x, real D, real R, real f, real* d1, real* d2) {
  if (kind == 1) {
    const real Rf = R*f;
    if (x <= -Rf) { *d1 = -f; *d2 = 0; return f*(-0.5*Rf - x); }
    if (x >= Rf) { *d1 = f; *d2 = 0; return f*(-0.5*Rf + x); }
  } else if (kind == 2) {
    if (x >= 0) { *d1 = 0; *d2 = 0; return 0; }
  }
  *d1 = D*x; *d2 = D;
  return 0.5*D*x*x;
}

// cost(x1) - cost(x0) of one row, cancellation-free where both ends are in the quadratic zone
MJH_DEV real nt_row_costdif(int kind, real x0, real x1, real D, real R, real f) {
  real a, b;
  int q0 = 1, q1 = 1;
  if (kind == 1) { const real Rf = R*f; q0 = (x0 > -Rf && x0 < Rf); q1 = (x1 > -Rf && x1 < Rf); }
  else if (kind == 2) { q0 = (x0 < 0); q1 = (x1 < 0); }
  if (q0 && q1) { const real dx = x1 - x0; return D*(x0*dx + 0.5*dx*dx); }
  return nt_row_cost(kind, x1, D, R, f, &a, &b) - nt_row_cost(kind, x0, D, R, f, &a, &b);
}

struct NtPoint { real alpha, cost, d1, d2; };

// line-search data of one elliptic block (PrimalPrepare, engine_solver.c:1476-1516): the bottom-zone
// quadratic q[0..2], and the regular-cone quantities U0, V0, UU, UV, VV, Dm
struct NtCone { real q0, q1, q2, U0, V0, UU, UV, VV, Dm, mu; };
template <class P0, class P1>
MJH_DEV void nt_cone_prepare(const Efc& P, int i, int dim, P0 jar, P1 Jv, NtCone& c) {
  const real mu = P.cone[i];
  real q0 = 0, q1 = 0, q2 = 0, UU = 0, UV = 0, VV = 0;
  for (int j = 0; j < dim; j++) {
    const real DJ = P.D[i+j]*jar[i+j];
    q0 += jar[i+j]*DJ;
    q1 += Jv[i+j]*DJ;
    q2 += Jv[i+j]*P.D[i+j]*Jv[i+j];
    if (j) {
      const real U = jar[i+j]*P.cone[i+j], V = Jv[i+j]*P.cone[i+j];
      UU += U*U; UV += U*V; VV += V*V;
    }
  }
  c.q0 = 0.5*q0; c.q1 = q1; c.q2 = 0.5*q2;
  c.U0 = jar[i]*mu; c.V0 = Jv[i]*mu; c.UU = UU; c.UV = UV; c.VV = VV;
  c.Dm = P.D[i] / ((mu*mu) * (1 + (mu*mu)));
  c.mu = mu;
}
// zone of a (N, T^2) pair: 1 top, 2 bottom, 3 middle; T returned
MJH_DEV int nt_cone_zone(real N, real Tsqr, real mu, real* T) {
  *T = 0;
  if (Tsqr <= 0) return (N < 0) ? 2 : 1;
  *T = sqrt(Tsqr);
  if (N >= mu*(*T)) return 1;
  if (mu*N + (*T) <= 0) return 2;
  return 3;
}
// cost(alpha) - cost(0) of the block, cancellation-free per zone pair, plus the first and second
// derivative along the line                   (ellipticCostDif / PrimalEval, engine_solver.c:1573-1790)
MJH_DEV real nt_cone_eval(const NtCone& c, real alpha, real* d1, real* d2) {
  const real mu = c.mu, Dm = c.Dm;
  real T0, T;
  const int z0 = nt_cone_zone(c.U0, c.UU, mu, &T0);
  const real N = c.U0 + alpha*c.V0;
  const real Tsqr = c.UU + alpha*(2*c.UV + alpha*c.VV);
  const int za = nt_cone_zone(N, Tsqr, mu, &T);
  *d1 = 0; *d2 = 0;
  if (za == 2) { *d1 = 2*alpha*c.q2 + c.q1; *d2 = 2*c.q2; }
  else if (za == 3) {
    const real N1 = c.V0;
    const real T1 = (c.UV + alpha*c.VV)/T;
    const real T2 = c.VV/T - (c.UV + alpha*c.VV)*T1/(T*T);
    *d1 = Dm*(N - mu*T)*(N1 - mu*T1);
    *d2 = Dm*((N1 - mu*T1)*(N1 - mu*T1) + (N - mu*T)*(-mu*T2));
  }
  const real quad = alpha*alpha*c.q2 + alpha*c.q1;
  if (z0 == 1 && za == 1) return 0;
  if (z0 == 2 && za == 2) return quad;
  if (z0 == 3 && za == 3) {
    const real Tsqr_delta = alpha*(2*c.UV + alpha*c.VV);
    const real T_delta = Tsqr_delta / (T + T0);
    const real r_delta = alpha*c.V0 - mu*T_delta;
    const real r0 = c.U0 - mu*T0;
    return 0.5*Dm*r_delta*(2*r0 + r_delta);
  }
  if (z0 == 3 && za == 2) { const real b0 = mu*c.U0 + T0; return alpha*(alpha*c.q2 + c.q1) + 0.5*Dm*b0*b0; }
  if (z0 == 2 && za == 3) { const real bb = mu*N + T; return alpha*(alpha*c.q2 + c.q1) - 0.5*Dm*bb*bb; }
  if (z0 == 1 && za == 2) return quad + c.q0;
  if (z0 == 1 && za == 3) { const real r = N - mu*T; return 0.5*Dm*r*r; }
  if (z0 == 3 && za == 1) { const real r0 = c.U0 - mu*T0; return -0.5*Dm*r0*r0; }
  if (z0 == 2 && za == 1) return -c.q0;
  return 0;
}

// cone-block helpers of solve_primal<1>
MJH_DEV real nt_cone_line(const Efc& P, int r, int nefc, real alpha, real* d1, real* d2) {
  NtCone cb;
  nt_cone_prepare(P, r, cone_dim(P, r, nefc), P.jar, P.ARf, cb);
  return nt_cone_eval(cb, alpha, d1, d2);
}
MJH_DEV real nt_cone_cost(const Efc& P, int r, int nefc, crptr x) { return cone_cost(P, r, cone_dim(P, r, nefc), x); }
MJH_DEV void nt_cone_update(const Efc& P, int r, int nefc, rptr Hc, int want_hessian) {
  cone_update(P, r, cone_dim(P, r, nefc), P.jar, Hc, want_hessian);
}
// (J_blk' Hc J_blk)(i, j) of the middle-zone cone block starting at row r; returns the block's dim
MJH_DEV int nt_cone_hessian_term(const Efc& P, int r, int nefc, int nv, crptr Hc, int i, int j, real* acc) {
  const int dim = cone_dim(P, r, nefc);
  crptr J = P.J;
  real sum = 0;
  for (int a = 0; a < dim; a++) {
    const real Ja = J[(size_t)(r + a)*nv + i];
    if (Ja == 0) continue;
    real t = 0;
    for (int b2 = 0; b2 < dim; b2++) t += Hc[a*dim + b2]*J[(size_t)(r + b2)*nv + j];
    sum += Ja*t;
  }
  *acc += sum;
  return dim;
}

// flg_newton = 0: the conjugate-gradient variant (mj_solCG): same cost, line search and warm start;
// the search direction is the M^-1-preconditioned gradient with Hager-Zhang conjugation
// (engine_solver.c:2506-2536) and no Hessian is built.
// ELL = 0: instantiation without any elliptic-cone code (the common pyramidal case keeps its
// register budget); ELL = 1: cone blocks enabled
template <int ELL>
MJH_DEVN void solve_primal(MREF M_, BREF B_, int e_, int flg_newton) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ne = counts[MJH_C_NE], nf = counts[MJH_C_NF];
  Efc P;
  efc_layout(M, B, e, nefc, P);
  crptr J = P.J;
  crptr aref = P.aref;
  crptr qas = MJH_F(B, qacc_smooth, e);
  crptr qfs = MJH_F(B, qfrc_smooth, e);
  crptr qws = MJH_F(B, qacc_warmstart, e);
  rptr qacc = MJH_F(B, qacc, e);
  rptr qfc = MJH_F(B, qfrc_constraint, e);
  rptr Md = MJH_G(B, nt_M, e);          // dense M
  rptr H = MJH_G(B, nt_H, e);           // Hessian, then its Cholesky factor (lower)
  // the work vectors live in the unused tail of the LDS plan when it is large enough (the dual-only arrays
  // AR / Y take no LDS under the primal solvers), else in their global home
  rptr vec = MJH_G(B, nt_vec, e);
  if (P.free_bytes >= (int)(8*nv*sizeof(real))) vec = SP<real>{(real*)P.free_p, 1};
  rptr Ma = vec, grad = vec + nv, Mgrad = vec + 2*nv, search = vec + 3*nv, Mv = vec + 4*nv;
  rptr gradold = vec + 6*nv, Mgradold = vec + 7*nv;
  rptr jar = P.jar, Jv = P.ARf;
  const int lane = wv_lane();
  const real tol = M.o.tolerance;

  // ---- dense M from the parked sparse copy (CSR lower triangle, diagonal last in each row)
  crptr Ms = MJH_G(B, M, e);
  MJH_FOR_LANES(k, nv*nv) Md[k] = 0;
  wv_sync();
  MJH_FOR_LANES(i, nv) {
    const int adr = M.M_rowadr[i], nnz = M.M_rownnz[i];
    for (int a = 0; a < nnz; a++) {
      const int j = M.M_colind[adr + a];
      Md[i*nv + j] = Ms[adr + a];
      Md[j*nv + i] = Ms[adr + a];
    }
  }
  wv_sync();

  auto mul_M = [&](rptr out, crptr v) {            // out = M v
    MJH_FOR_LANES(i, nv) {
      real acc = 0;
      for (int j = 0; j < nv; j++) acc += Md[i*nv + j]*v[j];
      out[i] = acc;
    }
    wv_sync();
  };
  auto mul_J = [&](rptr out, crptr v, int sub_aref) {   // out = J v (- aref)
    MJH_FOR_LANES(r, nefc) {
      real acc = 0;
      for (int j = 0; j < nv; j++) acc += J[(size_t)r*nv + j]*v[j];
      out[r] = sub_aref ? acc - aref[r] : acc;
    }
    wv_sync();
  };
  // 0 equality, 1 friction loss, 2 inequality, 3 first row of an elliptic block, 4 its other rows
  const int elliptic = ELL ? (M.o.cone != 0) : 0;
  auto row_kind = [&](int r) {
    if (r < ne) return 0;
    if (r < ne + nf) return 1;
    if (elliptic && P.type[r] == MJH_CNSTR_CONTACT_ELLIPTIC) return cone_leader(P, r) ? 3 : 4;
    return 2;
  };
  rptr conH = MJH_G(B, con_H, e);
  auto constraint_cost = [&](crptr x) {             // sum of row costs at residual x
    real c = 0;
    MJH_FOR_LANES(r, nefc) {
      const int kind = row_kind(r);
      if (kind == 3) c += nt_cone_cost(P, r, nefc, x);
      else if (kind < 3) { real a, b; c += nt_row_cost(kind, x[r], P.D[r], P.R[r], P.floss[r], &a, &b); }
    }
    return wv_sum_d(c);
  };
  auto dot_nv = [&](crptr a, crptr b) {
    real c = 0;
    MJH_FOR_LANES(i, nv) c += a[i]*b[i];
    return wv_sum_d(c);
  };

  // ---- warm start: best of (qacc_warmstart, qacc_smooth)        (engine_forward.c:1056-1132)
  if (!(M.o.disableflags & (1<<9))) {
    mul_J(jar, qws, 1);
    mul_M(Ma, qws);
    real g = 0;
    MJH_FOR_LANES(i, nv) g += 0.5*(Ma[i] - qfs[i])*(qws[i] - qas[i]);
    const real cost_ws = constraint_cost(jar) + wv_sum_d(g);
    const real cost_smooth = constraint_cost(P.b);
    const int use_smooth = cost_ws > cost_smooth;
#ifdef MJH_DEBUG_NT
    if (lane == 0) printf("warmstart: cost_ws %.15g cost_smooth %.15g use_smooth %d\n", cost_ws, cost_smooth, use_smooth);
#endif
    MJH_FOR_LANES(i, nv) qacc[i] = use_smooth ? qas[i] : qws[i];
  } else {
    MJH_FOR_LANES(i, nv) qacc[i] = qas[i];
  }
  wv_sync();

  // ---- constraint islands (engine_forward.c:1187-1212): one solve per island, each with its own
  // scale, line searches and termination.  M and the Hessian are block diagonal across islands, so
  // masking the gradient to the island's dofs confines the whole iteration to it.
  const int nisl_raw = counts[MJH_C_NISLAND];
  const int nisl = nisl_raw > 1 ? nisl_raw : 1;
  const int multi_tree = (s.ntree > 1) && (nisl_raw > 0);
  ciptr tree_island = MJH_G(B, island_work, e) + s.nefcmax + s.ntree;   // left by stage_island
  int isl = 0;
  auto in_dof = [&](int i) { return !multi_tree || tree_island[M.dof_treeid[i]] == isl; };
  auto in_row = [&](int r) { return nisl_raw <= 1 || P.island[r] == isl; };
  if (multi_tree && !(M.o.disableflags & (1<<9))) {
    // dofs of unconstrained trees start (and stay) at qacc_smooth   (warmstart, :1117-1124)
    MJH_FOR_LANES(i, nv) if (tree_island[M.dof_treeid[i]] < 0) qacc[i] = qas[i];
    wv_sync();
  }

  // ---- initial Ma, jar
  mul_M(Ma, qacc);
  mul_J(jar, qacc, 1);
  int niter0 = 0;
  auto update_constraint = [&]() {                  // efc_force, efc_state, qfrc_constraint, grad
    MJH_FOR_LANES(r, nefc) {
      if (!in_row(r)) continue;
      const int kind = row_kind(r);
      if (kind >= 3) {
        // elliptic block: forces, state and (Newton) the cone Hessian of the middle zone
        if (kind == 3) nt_cone_update(P, r, nefc, conH + 36*P.id[r], flg_newton);
        continue;
      }
      real d1, d2;
      nt_row_cost(kind, jar[r], P.D[r], P.R[r], P.floss[r], &d1, &d2);
      P.force[r] = -d1;
      int st = MJH_STATE_QUADRATIC;
      if (d2 == 0) st = (kind == 2) ? MJH_STATE_SATISFIED : (d1 < 0 ? MJH_STATE_LINEARNEG : MJH_STATE_LINEARPOS);
      P.state[r] = st;
    }
    wv_sync();
    MJH_FOR_LANES(j, nv) {
      real acc = 0;
      // rows of other islands have zero Jacobian entries on this island's dofs, but their forces may
      // not have been written yet (uninitialised LDS): they must not enter the sum
      for (int r = 0; r < nefc; r++) if (in_row(r)) acc += J[(size_t)r*nv + j]*P.force[r];
      if (in_dof(j)) { qfc[j] = acc; grad[j] = Ma[j] - qfs[j] - acc; }
      else grad[j] = 0;
    }
    wv_sync();
  };
  if (multi_tree) { MJH_FOR_LANES(j, nv) qfc[j] = 0; wv_sync(); }
  for (isl = 0; isl < nisl; isl++) {
  update_constraint();

  // termination scale: 1/trace(M) over the dofs of constrained trees when islands are on
  // (engine_solver.c:2383-2390), 1/(meaninertia*nv) otherwise
  real scale;
  if (!(M.o.disableflags & (1<<18))) {
    real tr = 0;
    MJH_FOR_LANES(i, nv) {
      // inertia of this island's dofs (engine_solver.c:2383-2390)
      tr += Md[i*nv + i] * (real)(in_dof(i) ? 1 : 0);
    }
    tr = wv_sum_d(tr);
    scale = 1 / (tr > 0 ? tr : 1);
  } else {
    scale = 1 / (M.o.meaninertia * (real)(nv > 1 ? nv : 1));
  }

  // H = M + J' D_active J, Cholesky, Mgrad = H \ grad.
  // Lanes split the LOWER triangle of H (its upper half is never read).  The factorisation is
  // right-looking: at step k the finished column k sits in registers (lane i - k holds L[i][k]) and the
  // trailing update fetches its two factors with cross-lane reads instead of going back to memory
  // for values other lanes have just written; the two triangular solves keep the right-hand side in
  // registers (lane = row) and sweep column by column, so each of their nv steps is one broadcast and
  // one multiply-subtract instead of a serial dot product on one lane.
  auto factor_and_solve = [&]() {
    const int ntri = nv*(nv + 1)/2;
    MJH_FOR_LANES(w, ntri) {
      int i = (int)((sqrt(8.0*w + 1.0) - 1.0)*0.5);
      while (i*(i+1)/2 > w) i--;
      while ((i+1)*(i+2)/2 <= w) i++;
      const int j = w - i*(i+1)/2;
      real acc = Md[i*nv + j];
      for (int r = 0; r < nefc; r++) {
        if (!in_row(r)) continue;       // the blocks of other islands stay M (never used: grad is 0 there)
        const int st = P.state[r];
        if (st == MJH_STATE_QUADRATIC) acc += P.D[r]*J[(size_t)r*nv + i]*J[(size_t)r*nv + j];
        else if (ELL && st == MJH_STATE_CONE) {
          // J_blk' Hc J_blk of a middle-zone cone block (HessianCone, engine_solver.c:2219-2281)
          r += nt_cone_hessian_term(P, r, nefc, nv, conH + 36*P.id[r], i, j, &acc) - 1;
        }
      }
      H[i*nv + j] = acc;
    }
    wv_sync();
#if MJH_LANE_MODE
    for (int k = 0; k < nv; k++) {                  // right-looking Cholesky, lower triangle
      const real dkk = sqrt(r_max(H[k*nv + k], MJH_MINVAL));
      for (int i = k; i < nv; i++) H[i*nv + k] = (i == k) ? dkk : H[i*nv + k] / dkk;
      for (int i = k + 1; i < nv; i++) for (int j = k + 1; j <= i; j++) H[i*nv + j] -= H[i*nv + k]*H[j*nv + k];
    }
    for (int i = 0; i < nv; i++) Mgrad[i] = grad[i];
    for (int i = 0; i < nv; i++) {
      real acc = Mgrad[i];
      for (int j = 0; j < i; j++) acc -= H[i*nv + j]*Mgrad[j];
      Mgrad[i] = acc / H[i*nv + i];
    }
    for (int i = nv - 1; i >= 0; i--) {
      real acc = Mgrad[i];
      for (int j = nv - 1; j > i; j--) acc -= H[j*nv + i]*Mgrad[j];
      Mgrad[i] = acc / H[i*nv + i];
    }
#else
    if (nv <= MJH_W) {
      for (int k = 0; k < nv; k++) {
        // column k: lane q holds L[k + q][k]
        const int myrow = k + lane;
        real col = (myrow < nv) ? (real)H[myrow*nv + k] : (real)0;
        const real dkk = sqrt(r_max(wv_bcast(col, 0), MJH_MINVAL));
        col = (lane == 0) ? dkk : col / dkk;
        if (myrow < nv) H[myrow*nv + k] = col;
        // trailing update: entry (i, j), k < j <= i, takes L[i][k] and L[j][k] from the column's lanes
        const int m = nv - k - 1;
        const int mtri = m*(m + 1)/2;
        for (int w0 = 0; w0 < mtri; w0 += MJH_W) {
          const int w = w0 + lane;
          int a = 0, c = 0;
          if (w < mtri) {
            a = (int)((sqrt(8.0*w + 1.0) - 1.0)*0.5);
            while (a*(a+1)/2 > w) a--;
            while ((a+1)*(a+2)/2 <= w) a++;
            c = w - a*(a+1)/2;
          }
          const real lik = wv_shfl(col, a + 1), ljk = wv_shfl(col, c + 1);
          if (w < mtri) H[(k + 1 + a)*nv + (k + 1 + c)] -= lik*ljk;
        }
        wv_sync();
      }
      // L y = grad, column sweep: y in registers (lane = row)
      real y = (lane < nv) ? (real)grad[lane] : (real)0;
      for (int j = 0; j < nv; j++) {
        const real lij = (lane >= j && lane < nv) ? (real)H[lane*nv + j] : (real)1;
        if (lane == j) y = y / lij;
        const real yj = wv_bcast(y, j);
        if (lane > j && lane < nv) y -= lij*yj;
      }
      // L' x = y
      for (int j = nv - 1; j >= 0; j--) {
        const real lji = (lane <= j) ? (real)H[j*nv + lane] : (real)1;
        if (lane == j) y = y / lji;
        const real xj = wv_bcast(y, j);
        if (lane < j) y -= lji*xj;
      }
      if (lane < nv) Mgrad[lane] = y;
      wv_sync();
    } else {
      for (int k = 0; k < nv; k++) {                  // right-looking Cholesky, lower triangle
        const real dkk = sqrt(r_max(H[k*nv + k], MJH_MINVAL));
        wv_sync();
        MJH_FOR_LANES(i, nv) if (i >= k) H[i*nv + k] = (i == k) ? dkk : H[i*nv + k] / dkk;
        wv_sync();
        const int m = nv - k - 1;
        MJH_FOR_LANES(w, m*m) {
          const int i = k + 1 + w / m, j = k + 1 + w % m;
          if (j <= i) H[i*nv + j] -= H[i*nv + k]*H[j*nv + k];
        }
        wv_sync();
      }
      MJH_FOR_LANES(i, nv) Mgrad[i] = grad[i];
      wv_sync();
      for (int i = 0; i < nv; i++) {                  // L y = grad
        if (lane == 0) {
          real acc = Mgrad[i];
          for (int j = 0; j < i; j++) acc -= H[i*nv + j]*Mgrad[j];
          Mgrad[i] = acc / H[i*nv + i];
        }
        wv_sync();
      }
      for (int i = nv - 1; i >= 0; i--) {             // L' x = y
        if (lane == 0) {
          real acc = Mgrad[i];
          for (int j = nv - 1; j > i; j--) acc -= H[j*nv + i]*Mgrad[j];
          Mgrad[i] = acc / H[i*nv + i];
        }
        wv_sync();
      }
    }
#endif
  };

  auto precondition = [&]() {                      // Mgrad = M \ grad (CG preconditioner, certificate)
    MJH_FOR_LANES(i, nv) Mgrad[i] = grad[i];
    wv_sync();
    solve_ld(M, Mgrad, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e));
  };
  int iter = 0;
  int done;
  {
    // convergence certificate with M^-1 (engine_solver.c:2393-2409)
    precondition();
    const real gap = r_max(0, 0.5*scale*dot_nv(grad, Mgrad));
    const real gnorm = scale*sqrt(dot_nv(grad, grad));
    done = (gap < tol) && (!flg_newton || gnorm < tol);
    if (!done && flg_newton) {
      factor_and_solve();
      done = (gnorm < tol) && (r_max(0, 0.5*scale*dot_nv(grad, Mgrad)) < tol);
    }
  }
  if (!done) { MJH_FOR_LANES(i, nv) search[i] = -Mgrad[i]; wv_sync(); }

  const int maxiter = M.o.iterations;
  while (!done && iter < maxiter) {
    // ---- exact line search along `search`                      (PrimalSearch :1856-2054)
    const real snorm = sqrt(dot_nv(search, search));
    if (snorm < MJH_MINVAL) break;
    const real gtol = tol*M.o.ls_tolerance*snorm/scale;
    mul_M(Mv, search);
    mul_J(Jv, search, 0);
    real g1 = 0, g2 = 0;
    MJH_FOR_LANES(i, nv) { g1 += search[i]*(Ma[i] - qfs[i]); g2 += 0.5*search[i]*Mv[i]; }
    g1 = wv_sum_d(g1); g2 = wv_sum_d(g2);
    int lsiter = 0;
    auto eval = [&](NtPoint& p) {
      const real al = p.alpha;
      real c = 0, d1 = 0, d2 = 0;
      MJH_FOR_LANES(r, nefc) {
        if (!in_row(r)) continue;
        const int kind = row_kind(r);
        if (kind >= 3) {
          if (kind == 3) {
            real a1, a2;
            c += nt_cone_line(P, r, nefc, al, &a1, &a2);      // reads P.jar / P.ARf (= jar, Jv)
            d1 += a1; d2 += a2;
          }
          continue;
        }
        const real x0 = jar[r], dx = Jv[r], x1 = x0 + al*dx;
        real a1, a2;
        nt_row_cost(kind, x1, P.D[r], P.R[r], P.floss[r], &a1, &a2);
        c += nt_row_costdif(kind, x0, x1, P.D[r], P.R[r], P.floss[r]);
        d1 += a1*dx;
        d2 += a2*dx*dx;
      }
      p.cost = wv_sum_d(c) + al*g1 + al*al*g2;
      p.d1 = wv_sum_d(d1) + g1 + 2*al*g2;
      p.d2 = r_max(wv_sum_d(d2) + 2*g2, MJH_MINVAL);
      lsiter++;
    };
    const int lsmax = M.o.ls_iterations;
    NtPoint p0, p1, p2, pmid, p1next, p2next;
    real alpha = 0, improvement = 0;
    p0.alpha = 0; eval(p0);
#ifdef MJH_DEBUG_NT
    if (lane == 0) printf("iter %d  p0: cost %g d1 %.12g d2 %.12g  -d1/d2 %.12g\n", iter, p0.cost, p0.d1, p0.d2, -p0.d1/p0.d2);
#endif
    p1.alpha = p0.alpha - p0.d1/p0.d2; eval(p1);
    int found = 0;
    if (fabs(p1.d1) < gtol && (p1.alpha == 0 || p1.cost < 0)) {
      alpha = p1.alpha; improvement = -p1.cost; found = 1;
    }
    if (!found) {
      const int dir = (p1.d1 < 0) ? 1 : -1;
      p2 = p0;
      while (p1.d1*dir <= -gtol && lsiter < lsmax) {          // one-sided search
        p2 = p1;
        p1.alpha -= p1.d1/p1.d2; eval(p1);
        if (fabs(p1.d1) < gtol && p1.cost < 0) { alpha = p1.alpha; improvement = -p1.cost; found = 1; break; }
      }
      if (!found && lsiter >= lsmax) { alpha = p1.alpha; improvement = -p1.cost; found = 1; }
    }
    if (!found) {
      p2next = p1;
      p1next.alpha = p1.alpha - p1.d1/p1.d2; eval(p1next);
      auto update_bracket = [&](NtPoint& p, const NtPoint* cand, NtPoint& pnext) {
        int flag = 0;
        for (int i = 0; i < 3; i++) {
          if (p.d1 < 0 && cand[i].d1 < 0 && p.d1 < cand[i].d1) { p = cand[i]; flag = 1; }
          else if (p.d1 > 0 && cand[i].d1 > 0 && p.d1 > cand[i].d1) { p = cand[i]; flag = 2; }
        }
        if (flag) { pnext.alpha = p.alpha - p.d1/p.d2; eval(pnext); }
        return flag;
      };
      while (lsiter < lsmax) {                                  // bracketed search
        pmid.alpha = 0.5*(p1.alpha + p2.alpha); eval(pmid);
        const NtPoint cand[3] = {p1next, p2next, pmid};
        int best = -1; real bestcost = 0;
        for (int i = 0; i < 3; i++)
          if (fabs(cand[i].d1) < gtol && (best == -1 || cand[i].cost < bestcost)) { bestcost = cand[i].cost; best = i; }
        if (best >= 0) { alpha = cand[best].alpha; improvement = -cand[best].cost; found = 1; break; }
        const int b1 = update_bracket(p1, cand, p1next);
        const int b2 = update_bracket(p2, cand, p2next);
        if (!b1 && !b2) { alpha = pmid.alpha; improvement = -pmid.cost; found = 1; break; }
      }
      if (!found) {
        if (p1.cost <= p2.cost && p1.cost < 0) { alpha = p1.alpha; improvement = -p1.cost; }
        else if (p2.cost <= p1.cost && p2.cost < 0) { alpha = p2.alpha; improvement = -p2.cost; }
        else alpha = 0;
      }
    }
#ifdef MJH_DEBUG_NT
    if (lane == 0) printf("iter %d alpha %.15g improvement %g lsiter %d gtol %g p1.d1 %g p1.cost %g\n", iter, alpha, improvement, lsiter, gtol, p1.d1, p1.cost);
#endif
    if (alpha == 0) break;

    // ---- move, update constraints / gradient / Hessian
    MJH_FOR_LANES(i, nv) { qacc[i] += search[i]*alpha; Ma[i] += Mv[i]*alpha; }
    MJH_FOR_LANES(r, nefc) jar[r] += Jv[r]*alpha;
    if (!flg_newton) MJH_FOR_LANES(i, nv) { gradold[i] = grad[i]; Mgradold[i] = Mgrad[i]; }
    wv_sync();
    update_constraint();
    if (flg_newton) factor_and_solve(); else precondition();
    const real imp = scale*improvement;
    const real gradient = scale*sqrt(dot_nv(grad, grad));
    const real decrement = flg_newton ? r_max(0, 0.5*scale*dot_nv(grad, Mgrad)) : 0;
    iter++;
    if ((imp > 0 && imp < tol) || gradient < tol || (flg_newton && decrement < tol)) break;
    if (flg_newton) {
      MJH_FOR_LANES(i, nv) search[i] = -Mgrad[i];
    } else {
      // Hager-Zhang conjugate direction (engine_solver.c:2506-2536)
      real dy = 0, yMy = 0, yMg = 0, dg = 0, dd = 0, gg = 0;
      MJH_FOR_LANES(i, nv) {
        const real y = grad[i] - gradold[i], My = Mgrad[i] - Mgradold[i];
        dy += search[i]*y; yMy += y*My; yMg += y*Mgrad[i]; dg += search[i]*grad[i];
        dd += search[i]*search[i]; gg += grad[i]*grad[i];
      }
      dy = wv_sum_d(dy); yMy = wv_sum_d(yMy); yMg = wv_sum_d(yMg); dg = wv_sum_d(dg);
      dd = wv_sum_d(dd); gg = wv_sum_d(gg);
      real beta = 0;
      if (!(dy < MJH_MINVAL)) {
        const real beta_hz = (yMg - 2*(yMy/dy)*dg) / dy;
        const real eta_k = -1.0 / r_max(MJH_MINVAL, sqrt(dd) * r_min(0.01, sqrt(gg)));
        beta = r_max(eta_k, beta_hz);
      }
      wv_sync();
      MJH_FOR_LANES(i, nv) search[i] = -Mgrad[i] + beta*search[i];
    }
    wv_sync();
  }
  if (isl == 0) niter0 = iter;
  }   // islands
  if (lane == 0) counts[MJH_C_NITER] = niter0;
  wv_sync();
}

MJH_DEVN void solve_newton(MREF M_, BREF B_, int e_) {
  if (M_.o.cone != 0) solve_primal<1>(M_, B_, e_, 1); else solve_primal<0>(M_, B_, e_, 1);
}
MJH_DEVN void solve_cg(MREF M_, BREF B_, int e_) {
  if (M_.o.cone != 0) solve_primal<1>(M_, B_, e_, 0); else solve_primal<0>(M_, B_, e_, 0);
}
