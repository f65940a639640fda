// Passes over the dofs of the explicit-index CG solver (mjh_newton.h, SPA = 2) as functions of an ARGUMENT BLOCK, so that a
// multi-wavefront workgroup can run them on its helper wavefronts (mjh_modes.h: namespace wq, MJH_HELPERS_ARGS) -- and the
// value pass of the compressed rows (csr_row_values) on all of them (MJH_WIDE_ARGS).
//
// A CG iteration of mj_solPrimal (engine_solver.c:2344-2563) touches every dof a dozen times -- M search, the move along
// the search direction, qfrc_constraint = J' force, the gradient, the preconditioner, the Hager-Zhang differences, the new
// direction -- and with nv = 1536 (jelly.xml) one wavefront spends most of the solve waiting for those element-wise passes'
// memory round trips, 24 elements per lane at a time.  Nothing in them crosses dofs: here they are FUSED into two passes per
// iteration (CSR_OP_STEP after the line search, CSR_OP_DIR after the termination tests) and two for the set-up
// (CSR_OP_WARM, CSR_OP_START), each a loop over "lane = dof" with the width the argument block names (lane0, width) -- the 64
// lanes of the wavefront in the one-wavefront mappings, the 64 (MJH_MW - 1) lanes of the helper wavefronts in a
// multi-wavefront workgroup (wave 0 keeps the solver's registers and only posts the block).  Every element is computed by the expressions of the unfused code in
// mjh_newton.h (same operands, same order), so results are bit-identical.  The ORDERED sums of the iteration (mju_dot's four
// accumulator chains) stay on one wavefront, but their ADDENDS -- the element-wise products -- are formed here too and left
// in six LDS vectors: a link of a chain is then one LDS read and one dependent addition (csr_sums, mjh_newton.h) instead of
// two reads, a multiplication and the addition.  The vectors nothing but these passes touch (Ma, Mv, Mgrad) live in
// global memory, which is what makes room for the products.
//
// Preconditions (solve_primal checks them, else it takes the unfused path): CG, environment-major batch, the island spans
// every dof (flex stiffness unions the trees of a flex: engine_island.c:409-440), diagonal mass matrix (every dof a slider of
// its own body: nC == nv), so mul_M is Ms[i]*v[i] and mj_solveLD is x[i]*qLDiagInv[i].
// (included once per SPMD mode by mjh_stages.inc and by namespace wq: no include guard; CsrPass itself lives in mjh_types.h)

#if !MJH_LANE_MODE

#define MJH_CSR_U 4           // elements per lane whose loads are issued together

// qfrc_constraint[i] = mju_dotSparse over the rows that contain dof i (mju_mulMatVecSparse on J', engine_util_sparse.c):
// four accumulators over the stored entries in groups of four, then the rest one by one
MJH_DEV real csr_pass_jtf(const CsrPass& A, int a0, int n) {
  const real* v = A.spJT + a0;
  const int* ri = A.JTrow + a0;
  real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  int k = 0;
  for (; k <= n - 4; k += 4) {
    r0 += v[k]*A.force[ri[k]]; r1 += v[k + 1]*A.force[ri[k + 1]];
    r2 += v[k + 2]*A.force[ri[k + 2]]; r3 += v[k + 3]*A.force[ri[k + 3]];
  }
  real res = (r0 + r2) + (r1 + r3);
  for (; k < n; k++) res += v[k]*A.force[ri[k]];
  return res;
}

// PV: how the solver's block (grad | search | six product vectors, contiguous) is addressed -- local (ds_read / ds_write) when
// the plan placed it in LDS, else flat
MJH_DEV LP<real> csr_at(LP<real> b, int k) { return LP<real>{b.p + k}; }
MJH_DEV SP<real> csr_at(SP<real> b, int k) { return SP<real>{b.p + k, 1}; }
template <class PV>
MJH_DEV void csr_pass_body(MREF M, const CsrPass& A, PV blk) {
  const int lane = wv_lane() - A.lane0, nv = A.nv, W = A.width;
  if (lane >= 0) {
  real* const Ma = A.Ma; real* const Mgrad = A.Mgrad; real* const Mv = A.Mv;
  const PV grad = blk, search = csr_at(blk, nv);
  const PV p0 = csr_at(blk, 2*nv), p1 = csr_at(blk, 3*nv), p2 = csr_at(blk, 4*nv), p3 = csr_at(blk, 5*nv), p4 = csr_at(blk, 6*nv), p5 = csr_at(blk, 7*nv);
  const PV stage = p5;
  if (A.op == CSR_OP_WARM) {
    // warm start (engine_forward.c:1056-1132): Ma = M qacc_warmstart and the addends of
    // 0.5 (Ma - qfrc_smooth)' (qacc_warmstart - qacc_smooth), which the caller sums in the reference's order
    for (int i0 = lane; i0 < nv; i0 += MJH_CSR_U*W) {
      real ms[MJH_CSR_U], w[MJH_CSR_U], f[MJH_CSR_U], s[MJH_CSR_U];
#pragma unroll
      for (int u = 0; u < MJH_CSR_U; u++) { const int i = i0 + u*W < nv ? i0 + u*W : 0; ms[u] = A.Ms[i]; w[u] = A.qws[i]; f[u] = A.qfs[i]; s[u] = A.qas[i]; }
#pragma unroll
      for (int u = 0; u < MJH_CSR_U; u++) {
        const int i = i0 + u*W;
        if (i < nv) { const real ma = ms[u]*w[u]; Ma[i] = ma; stage[i] = 0.5*(ma - f[u])*(w[u] - s[u]); }
      }
    }
  } else if (A.op == CSR_OP_START) {
    // the starting point: qacc_warmstart or qacc_smooth (flag bit 0), qacc_smooth on the dofs of unconstrained trees
    // (bit 1: islands), Ma = M qacc; bit 2: the diagonal of M into the staging vector (the island's trace, summed by the
    // caller)
    const int use_smooth = A.flag & 1, trees = A.flag & 2, trace = A.flag & 4;
    for (int i0 = lane; i0 < nv; i0 += MJH_CSR_U*W) {
      real ms[MJH_CSR_U], q[MJH_CSR_U], s[MJH_CSR_U]; int out[MJH_CSR_U];
#pragma unroll
      for (int u = 0; u < MJH_CSR_U; u++) {
        const int i = i0 + u*W < nv ? i0 + u*W : 0;
        ms[u] = A.Ms[i]; s[u] = A.qas[i]; q[u] = use_smooth ? s[u] : (real)A.qws[i];
        out[u] = trees ? (int)(A.tree_island[M.dof_treeid[i]] < 0) : 0;
      }
#pragma unroll
      for (int u = 0; u < MJH_CSR_U; u++) {
        const int i = i0 + u*W;
        if (i < nv) {
          const real qq = out[u] ? s[u] : q[u];
          A.qacc[i] = qq; Ma[i] = ms[u]*qq;
          if (trace) stage[i] = ms[u];
        }
      }
    }
  } else if (A.op == CSR_OP_GRAD || A.op == CSR_OP_STEP) {
    // PrimalUpdateConstraint's qfrc_constraint = J' force, PrimalUpdateGrad, the preconditioner Mgrad = M \ grad.
    // CSR_OP_GRAD leaves the addends of grad.Mgrad and grad.grad (the convergence certificate).  CSR_OP_STEP first moves qacc
    // and Ma along the search direction, forms the Hager-Zhang differences graddif = grad - grad_old, Mgraddif = Mgrad -
    // Mgrad_old (engine_solver.c:2489-2496) and leaves the addends of the six sums of the direction update:
    // search.graddif, graddif.Mgraddif, graddif.Mgrad, search.grad, search.search, grad.grad
    const int step = A.op == CSR_OP_STEP;
    const real alpha = A.alpha;
    for (int i0 = lane; i0 < nv; i0 += MJH_CSR_U*W) {
      int b0[MJH_CSR_U], b1[MJH_CSR_U];
      real ma[MJH_CSR_U], f[MJH_CSR_U], di[MJH_CSR_U], q[MJH_CSR_U], sv[MJH_CSR_U], mv[MJH_CSR_U], g0[MJH_CSR_U], mg0[MJH_CSR_U];
#pragma unroll
      for (int u = 0; u < MJH_CSR_U; u++) {
        const int i = i0 + u*W < nv ? i0 + u*W : 0;
        b0[u] = A.JTadr[i]; b1[u] = A.JTadr[i + 1];
        ma[u] = Ma[i]; f[u] = A.qfs[i]; di[u] = A.dinv[i];
        if (step) { q[u] = A.qacc[i]; sv[u] = search[i]; mv[u] = Mv[i]; g0[u] = grad[i]; mg0[u] = Mgrad[i]; }
      }
#pragma unroll
      for (int u = 0; u < MJH_CSR_U; u++) {
        const int i = i0 + u*W;
        if (i >= nv) continue;
        if (step) { A.qacc[i] = q[u] + sv[u]*alpha; ma[u] = ma[u] + mv[u]*alpha; Ma[i] = ma[u]; }
        const real res = csr_pass_jtf(A, b0[u], b1[u] - b0[u]);
        A.qfc[i] = res;
        const real g = ma[u] - f[u] - res;
        const real mg = g*di[u];
        grad[i] = g; Mgrad[i] = mg;
        if (step) {
          const real gd = g - g0[u], mgd = mg - mg0[u];
          p0[i] = sv[u]*gd; p1[i] = gd*mgd; p2[i] = gd*mg; p3[i] = sv[u]*g; p4[i] = sv[u]*sv[u]; p5[i] = g*g;
        } else {
          p0[i] = g*mg; p1[i] = g*g;
        }
      }
    }
  } else if (A.op == CSR_OP_DIR) {
    // the search direction -- -Mgrad at the start, Hager-Zhang's -Mgrad + beta search afterwards --, Mv = M search, and the
    // addends of |search|^2 and PrimalPrepare's three sums: search.search, search.Ma, qfrc_smooth.search, search.Mv
    const int first = A.flag & 1;
    const real beta = A.alpha;
    for (int i0 = lane; i0 < nv; i0 += MJH_CSR_U*W) {
      real mg[MJH_CSR_U], sv[MJH_CSR_U], ms[MJH_CSR_U], ma[MJH_CSR_U], f[MJH_CSR_U];
#pragma unroll
      for (int u = 0; u < MJH_CSR_U; u++) {
        const int i = i0 + u*W < nv ? i0 + u*W : 0;
        mg[u] = Mgrad[i]; sv[u] = first ? (real)0 : search[i]; ms[u] = A.Ms[i]; ma[u] = Ma[i]; f[u] = A.qfs[i];
      }
#pragma unroll
      for (int u = 0; u < MJH_CSR_U; u++) {
        const int i = i0 + u*W;
        if (i < nv) {
          const real s = first ? -1*mg[u] : -mg[u] + beta*sv[u];
          const real mv = ms[u]*s;
          search[i] = s; Mv[i] = mv;
          p0[i] = s*s; p1[i] = s*ma[u]; p2[i] = f[u]*s; p3[i] = s*mv;
        }
      }
    }
  }
  }
  wv_sync();
}
// the solver's block is grad (it starts there) .. the last product vector
MJH_DEV void csr_pass(MREF M, const CsrPass& A) {
  const long long off = mjh_lds_offset((const void*)A.grad);
  if (A.search == A.grad + A.nv && A.prod == A.grad + 2*A.nv && off >= 0 && off < 160*1024) csr_pass_body(M, A, mjh_local(A.grad));
  else csr_pass_body(M, A, SP<real>{A.grad, 1});
}

// The values of the contact rows of the explicit-index Jacobian (mjh_csr.h, pass 2b): one lane per (contact, stored dof)
// item computes that dof's column of the contact's rows -- the expressions of stage_make_constraint's dense contact rows.
MJH_DEV void csr_row_values(MREF M, BREF B, int e, const CsrRowArgs& A) {
  const MJH_CONST_AS DSizes& s = M.s;
  const int ispyramid = A.ispyramid;
  const ciptr items = A.items, rowadr = A.rowadr; const iptr colind = A.colind; const rptr val = A.val;
  crptr cdof = MJH_F(B, cdof, e);
  crptr subtree_com = MJH_F(B, subtree_com, e);
  MJH_FOR_LANES(w, A.nitems) {
    const int k = items[w]/MJH_CSR_CHAIN_MAX, c = items[w]%MJH_CSR_CHAIN_MAX;
    const int r0 = MJH_CON(B, con_efcadr, e, 1, k)[0];
    const int dim = MJH_CON(B, con_dim, e, 1, k)[0];
    ConSides S;
    contact_sides(M, B, e, k, S);
    crptr point = MJH_CON(B, con_pos, e, 3, k);
    crptr fr = MJH_CON(B, con_frame, e, 9, k);
    auto fri = M.pair_friction + 5*MJH_CON(B, con_pair, e, 1, k)[0];
    const int a0 = rowadr[r0];
    const int stride = rowadr[r0 + 1] - a0;             // every row of the contact has the same pattern
    const int j = colind[a0 + c];
    // (the expressions of stage_make_constraint's dense contact rows)
    // (condim 4 / 6: rows 3..5 are the rotational difference against frame rows 0..2 -- torsion about the normal, rolling
    // about the tangents; mj_instantiateContact, engine_core_constraint.c:1613-1700)
    real jd[3], rd[3];
    contact_jac_col(M, S, cdof, subtree_com, point, j, jd, dim > 3 ? rd : (real*)nullptr);
    const int nr = dim > 1 ? (dim < 3 ? dim : 3) : 1;
    real jr[6] = {0, 0, 0, 0, 0, 0};
    for (int a = 0; a < nr; a++) {
      real acc = 0;
      for (int q = 0; q < 3; q++) { const real t = fr[3*a + q]; if (t != 0) acc += jd[q]*t; }
      jr[a] = acc;
    }
    for (int a = 3; a < dim; a++) {
      real acc = 0;
      for (int q = 0; q < 3; q++) { const real t = fr[3*(a - 3) + q]; if (t != 0) acc += rd[q]*t; }
      jr[a] = acc;
    }
    if (dim == 1) { val[a0 + c] = jr[0]; }
    else if (ispyramid) {
      for (int a = 1; a < dim; a++) {
        const int ra = a0 + (2*(a - 1))*stride + c, rb = a0 + (2*(a - 1) + 1)*stride + c;
        colind[ra] = j; val[ra] = jr[0] + jr[a]*fri[a - 1];
        colind[rb] = j; val[rb] = jr[0] + jr[a]*(-fri[a - 1]);
      }
    } else {
      for (int a = 0; a < dim; a++) { colind[a0 + a*stride + c] = j; val[a0 + a*stride + c] = jr[a]; }
    }
  }
  wv_sync();
}

#endif   // !MJH_LANE_MODE
