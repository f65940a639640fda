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
  for (int m = 32; m >= 1; m >>= 1) v += wv_shfl_xor(v, m);
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

// flg_newton = 0: the conjugate-gradient variant (mj_solCG): same cost, line search and warm start;
// the search direction is the M^-1-preconditioned gradient with Hager-Zhang conjugation
// (engine_solver.c:2506-2536) and no Hessian is built.
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
  rptr vec = MJH_G(B, nt_vec, e);
  rptr Ma = vec, grad = vec + nv, Mgrad = vec + 2*nv, search = vec + 3*nv, Mv = vec + 4*nv;
  rptr gradold = vec + 6*nv, Mgradold = vec + 7*nv;
  rptr jar = P.jar, Jv = P.ARf;
  const int lane = wv_lane();
  const real tol = M.o.tolerance;

  // ---- dense M from the parked sparse copy (CSR lower triangle, diagonal last in each row)
  crptr Ms = MJH_G(B, qH, e);
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
  auto row_kind = [&](int r) { return r < ne ? 0 : (r < ne + nf ? 1 : 2); };
  auto constraint_cost = [&](crptr x) {             // sum of row costs at residual x
    real c = 0;
    MJH_FOR_LANES(r, nefc) { real a, b; c += nt_row_cost(row_kind(r), x[r], P.D[r], P.R[r], P.floss[r], &a, &b); }
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
    MJH_FOR_LANES(i, nv) qacc[i] = use_smooth ? qas[i] : qws[i];
  } else {
    MJH_FOR_LANES(i, nv) qacc[i] = qas[i];
  }
  wv_sync();

  // ---- initial Ma, jar, forces, gradient
  mul_M(Ma, qacc);
  mul_J(jar, qacc, 1);
  auto update_constraint = [&]() {                  // efc_force, efc_state, qfrc_constraint, grad
    MJH_FOR_LANES(r, nefc) {
      real d1, d2;
      nt_row_cost(row_kind(r), jar[r], P.D[r], P.R[r], P.floss[r], &d1, &d2);
      P.force[r] = -d1;
      int st = MJH_STATE_QUADRATIC;
      if (d2 == 0) st = (row_kind(r) == 2) ? MJH_STATE_SATISFIED : (d1 < 0 ? MJH_STATE_LINEARNEG : MJH_STATE_LINEARPOS);
      P.state[r] = st;
    }
    wv_sync();
    MJH_FOR_LANES(j, nv) {
      real acc = 0;
      for (int r = 0; r < nefc; r++) acc += J[(size_t)r*nv + j]*P.force[r];
      qfc[j] = acc;
      grad[j] = Ma[j] - qfs[j] - acc;
    }
    wv_sync();
  };
  update_constraint();

  // termination scale: 1/trace(M) over the dofs of constrained trees when islands are on
  // (engine_solver.c:2383-2390), 1/(meaninertia*nv) otherwise
  real scale;
  if (!(M.o.disableflags & (1<<18))) {
    real tr = 0;
    MJH_FOR_LANES(i, nv) {
      int touched = 0;
      for (int r = 0; r < nefc && !touched; r++) if (J[(size_t)r*nv + i] != 0) touched = 1;
      // a tree is in an island as soon as one of its dofs carries a constraint
      tr += Md[i*nv + i] * (real)touched;
    }
    tr = wv_sum_d(tr);
    scale = 1 / (tr > 0 ? tr : 1);
  } else {
    scale = 1 / (M.o.meaninertia * (real)(nv > 1 ? nv : 1));
  }

  // H = M + J' D_active J, Cholesky, Mgrad = H \ grad
  auto factor_and_solve = [&]() {
    MJH_FOR_LANES(k, nv*nv) {
      const int i = k / nv, j = k - i*nv;
      real acc = Md[k];
      for (int r = 0; r < nefc; r++)
        if (P.state[r] == MJH_STATE_QUADRATIC) acc += P.D[r]*J[(size_t)r*nv + i]*J[(size_t)r*nv + j];
      H[k] = acc;
    }
    wv_sync();
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
        for (int j = i + 1; j < nv; j++) acc -= H[j*nv + i]*Mgrad[j];
        Mgrad[i] = acc / H[i*nv + i];
      }
      wv_sync();
    }
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
        const int kind = row_kind(r);
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
  if (lane == 0) counts[MJH_C_NITER] = iter;
  wv_sync();
}

MJH_DEVN void solve_newton(MREF M_, BREF B_, int e_) { solve_primal(M_, B_, e_, 1); }
MJH_DEVN void solve_cg(MREF M_, BREF B_, int e_) { solve_primal(M_, B_, e_, 0); }
