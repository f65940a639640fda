// Sparse constraint path: what the reference does when mj_isSparse(m) holds (jacobian = sparse, or auto and
// nv >= 60; engine_core_util.c:32) and the solver is a primal one (Newton / CG).
//
//   efc_J in compressed rows       mj_addConstraint, engine_core_constraint.c:449-490 (row pattern = the dof
//                                  chain(s) of the constraint: mj_mergeChain / mj_mergeChainSimple / mj_bodyChain,
//                                  engine_core_util.c:55-160; explicit zeros included)
//   J v, J' f                      mju_mulMatVecSparse / mju_dotSparse (engine_util_sparse.c:180, engine_util_sparse.h:197)
//   transpose                      mju_transposeSparse (engine_util_sparse.c:529), PrimalAllocate engine_solver.c:1376
//   H = J' D J + M                 mju_sqrMatTDSparseSymbolic / Numeric (engine_util_sparse.c:747, :910), mju_addToMatSparse (:219)
//   reverse Cholesky H = L' L      mju_cholFactorSymbolic / Numeric (engine_util_solve.c:198, :312)
//   solve, rank-one update         mju_cholSolveSparse (:387), mju_cholUpdateSparse (:431)
//
// Every floating-point result is the reference's, bit for bit: a sparse routine visits the structural
// non-zeros in a fixed order, and that order is what is reproduced here -- not the data structure.
//
// Data model.  A dof set is a 128-bit mask (nv <= 128).  Row r of J is (pattern mask, address, values in
// ascending dof order): the k-th stored entry belongs to the k-th member of the mask, so no column-index
// array exists.  The transpose is compressed by dof (constraint index + value per entry): its rows are what
// J' f and J' D J walk.  The Hessian and its factor live in ONE packed lower-triangular array (row r at
// r(r+1)/2): entries outside the symbolic pattern hold exact zeros, which turns the reference's scattered
// "dense[colind[i]] -= ..." loops into coalesced lane = column sweeps whose extra terms are +-0.  The
// symbolic factorisation (elimination tree, the order in which column r's rows are visited) is fused into
// the numeric one: per row, a scalar walk over bit masks.
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)

#if !MJH_LANE_MODE

MJH_DEV M128 sp_body_chain(MREF M, int b) { return m128_ldw(M.body_dofanc + (size_t)b*M.s.nvw, M.s.nvw); }
MJH_DEV M128 sp_tendon_pattern(MREF M, int t) {
  M128 pm = m128_zero();
  const int adr = M.ten_J_rowadr[t], n = M.ten_J_rownnz[t];
  for (int k = 0; k < n; k++) pm = m128_or(pm, m128_bit(M.ten_J_colind[adr + k]));
  return pm;
}
// dofs of constraint row (type, id) in the reference's sparse efc_J -- the `chain` argument of mj_addConstraint:
//   friction loss / slide, hinge limit: the dof; ball limit: its three dofs (engine_core_constraint.c:1455)
//   tendon rows: the tendon's ten_J pattern (:1318, :1505)
//   contacts: mj_jacDifPair with flg_skipcommon = 1 (:1551) -- the two body chains without their common part
//   connect / weld: both body chains, common dofs included (:655, :676); joint / tendon couplings: union of the two objects
MJH_DEV M128 sp_row_pattern(MREF M, BREF B, int e, int type, int id, int r) {
  if (type == MJH_CNSTR_FRICTION_DOF) return m128_bit(id);
  if (type == MJH_CNSTR_LIMIT_JOINT) {
    const int adr = M.jnt_dofadr[id];
    if (M.jnt_type[id] == MJH_JNT_BALL) return m128_or(m128_bit(adr), m128_or(m128_bit(adr + 1), m128_bit(adr + 2)));
    return m128_bit(adr);
  }
  if (type == MJH_CNSTR_FRICTION_TENDON || type == MJH_CNSTR_LIMIT_TENDON) return sp_tendon_pattern(M, id);
  if (type >= MJH_CNSTR_CONTACT_FRICTIONLESS) {
    if (MJH_HAS(MJH_FT_FLEX) && M.s.nconflex) {
      // flex contacts: one body on each side as below; a flex element on a side: the union of the chains of every body
      // involved (mj_jacSum; contact_sides, mjh_flex.h)
      ConSides S;
      contact_sides(M, B, e, id, S);
      if (!S.simple) {
        M128 pm = m128_zero();
        if (S.ext) { for (int q = 0; q < S.n; q++) pm = m128_or(pm, sp_body_chain(M, S.xb[q])); return pm; }
#pragma unroll
        for (int q = 0; q < 8; q++) if (q < S.n) pm = m128_or(pm, sp_body_chain(M, S.body[q]));
        return pm;
      }
      return m128_xor(sp_body_chain(M, S.body[0]), sp_body_chain(M, S.body[1]));
    }
    ciptr cg = MJH_CON(B, con_geom, e, 2, id);
    return m128_xor(sp_body_chain(M, M.geom_bodyid[cg[0]]), sp_body_chain(M, M.geom_bodyid[cg[1]]));
  }
  const int et = M.eq_type[id];
  if (MJH_HAS(MJH_FT_FLEX) && et == MJH_EQ_FLEX) {
    // a flex edge constraint's row is the edge's flexedge_J row (mj_instantiateEquality :982-1010)
    const int ed = M.eqrow_edge[M.eq_rowadr[id] + (r - MJH_G(B, eq_efcadr, e)[id])];
    M128 pm = m128_zero();
    const int adr = M.flexedge_J_rowadr[ed], n = M.flexedge_J_rownnz[ed];
    for (int k = 0; k < n; k++) pm = m128_or(pm, m128_bit(M.flexedge_J_colind[adr + k]));
    return pm;
  }
  int o1 = M.eq_obj1id[id], o2 = M.eq_obj2id[id];
  if (et == MJH_EQ_CONNECT || et == MJH_EQ_WELD) {
    if (M.eq_objsite[id]) { o1 = M.site_bodyid[o1]; o2 = M.site_bodyid[o2]; }
    return m128_or(sp_body_chain(M, o1), sp_body_chain(M, o2));
  }
  if (et == MJH_EQ_JOINT) {
    M128 pm = m128_bit(M.jnt_dofadr[o1]);
    if (o2 >= 0) pm = m128_or(pm, m128_bit(M.jnt_dofadr[o2]));
    return pm;
  }
  M128 pm = sp_tendon_pattern(M, o1);
  if (o2 >= 0) pm = m128_or(pm, sp_tendon_pattern(M, o2));
  return pm;
}

// The dense rows stage_make_constraint wrote (values at the pattern's dofs, exact zeros elsewhere) -> compressed
// rows + transpose.  Runs right after constraint assembly; from here on the primal path reads only the sparse form.
MJH_DEVN void stage_sparsify(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], nv = s.nv;
  const int lane = wv_lane();
  if (!nefc) {
    if (lane == 0) counts[MJH_C_NJ] = 0;
    wv_sync();
    return;
  }
  Efc P;
  efc_layout(M, B, e, nefc, P);
  // row patterns and addresses
  int base = 0;
  for (int r0 = 0; r0 < nefc; r0 += MJH_W) {
    const int r = r0 + lane;
    int nnz = 0;
    if (r < nefc) {
      const M128 pm = sp_row_pattern(M, B, e, P.type[r], P.id[r], r);
      nnz = m128_count(pm);
      m128_st(P.rowmask + 4*r, pm);
    }
    const int off = wv_exscan_i(nnz);
    if (r < nefc) P.rowadr[r] = base + off;
    base += wv_sum_i(nnz);
  }
  if (lane == 0) { P.rowadr[nefc] = base; counts[MJH_C_NJ] = base; }
  wv_sync();
  efc_layout(M, B, e, nefc, P);               // (the transposed arrays are sized by nJ)
  crptr J = P.J;
  crptr cdof = MJH_F(B, cdof, e);
  crptr subtree_com = MJH_F(B, subtree_com, e);
  const int ispyramid = M.o.cone == 0;
  const int dual = M.o.solver == MJH_SOL_PGS;       // (the dual solver keeps the dense rows: everything is cut from them)
  MJH_FOR_LANES(r, nefc) {
    M128 pm = m128_ld(P.rowmask + 4*r);
    int a = P.rowadr[r];
    const int type = P.type[r];
    if (dual || type < MJH_CNSTR_CONTACT_FRICTIONLESS) {
      // non-contact rows: cut from the dense row stage_make_constraint wrote
      if (MJH_HAS(MJH_FT_FLEX) && type == MJH_CNSTR_EQUALITY && M.eq_type[P.id[r]] == MJH_EQ_FLEX) {
        // (no dense row is written for a flex edge constraint: the edge's row, whose columns ascend like the pattern's)
        const int ed = M.eqrow_edge[M.eq_rowadr[P.id[r]] + (r - MJH_G(B, eq_efcadr, e)[P.id[r]])];
        crptr fJ = MJH_F(B, flexedge_J, e);
        const int f0 = M.flexedge_J_rowadr[ed], fn = M.flexedge_J_rownnz[ed];
        for (int q = 0; q < fn; q++) P.spJ[a++] = fJ[f0 + q];
        continue;
      }
      crptr Jr = J + (size_t)r*nv;
      while (m128_any(pm)) { const int j = m128_lowest(pm); pm = m128_drop_lowest(pm); P.spJ[a++] = Jr[j]; }
      continue;
    }
    // contact row, computed here for the dofs of its pattern (mj_contactJacobian / mj_instantiateContact,
    // engine_core_constraint.c:1535-1700: point Jacobians of the two bodies, their difference rotated into the
    // contact frame with mju_mulMatMat's zero-skip; the same expressions, dof by dof, as the dense assembly)
    const int k = P.id[r];
    const int sub = r - MJH_CON(B, con_efcadr, e, 1, k)[0];        // row of the contact's block
    const int dim = MJH_CON(B, con_dim, e, 1, k)[0];
    ciptr cg = MJH_CON(B, con_geom, e, 2, k);
    ConSides S;
    int general = 0;
    if (MJH_HAS(MJH_FT_FLEX) && s.nconflex) { contact_sides(M, B, e, k, S); general = 1; }
    const int b1 = general ? S.body[0] : (int)M.geom_bodyid[cg[0]], b2 = general ? S.body[1] : (int)M.geom_bodyid[cg[1]];
    const int w1 = M.body_weldid[b1], w2 = M.body_weldid[b2];
    crptr point = MJH_CON(B, con_pos, e, 3, k);
    crptr fr = MJH_CON(B, con_frame, e, 9, k);
    auto fri = M.pair_friction + 5*MJH_CON(B, con_pair, e, 1, k)[0];
    real off1[3], off2[3];
    v3_sub(off1, point, subtree_com + 3*M.body_rootid[b1]);
    v3_sub(off2, point, subtree_com + 3*M.body_rootid[b2]);
    // which frame rows this constraint row combines: a0 (+ a1 * scale)
    int a0 = 0, a1 = -1;
    real scl = 0;
    if (dim == 1) a0 = 0;
    else if (ispyramid) { a0 = 0; a1 = 1 + (sub >> 1); scl = (sub & 1) ? -fri[sub >> 1] : fri[sub >> 1]; }
    else a0 = sub;
    while (m128_any(pm)) {
      const int j = m128_lowest(pm);
      pm = m128_drop_lowest(pm);
      const int in1 = (M.body_dofanc[w1*s.nvw + (j >> 5)] >> (j & 31)) & 1;
      const int in2 = (M.body_dofanc[w2*s.nvw + (j >> 5)] >> (j & 31)) & 1;
      real j1[3] = {0, 0, 0}, j2[3] = {0, 0, 0};
      crptr cd = cdof + 6*j;
      if (in1) { real t[3]; v3_cross(t, cd, off1); j1[0] = cd[3] + t[0]; j1[1] = cd[4] + t[1]; j1[2] = cd[5] + t[2]; }
      if (in2) { real t[3]; v3_cross(t, cd, off2); j2[0] = cd[3] + t[0]; j2[1] = cd[4] + t[1]; j2[2] = cd[5] + t[2]; }
      real jd[3] = {j2[0] - j1[0], j2[1] - j1[1], j2[2] - j1[2]};
      real rd[3] = {(in2 ? cd[0] : (real)0) - (in1 ? cd[0] : (real)0), (in2 ? cd[1] : (real)0) - (in1 ? cd[1] : (real)0),
                    (in2 ? cd[2] : (real)0) - (in1 ? cd[2] : (real)0)};
      if (MJH_HAS(MJH_FT_FLEX) && general && !S.simple) contact_jac_col(M, S, cdof, subtree_com, point, j, jd, rd);
      auto frame_row = [&](int arow) -> real {
        // rows 0..2: translational difference, rows 3..5: rotational difference, each against frame row (arow mod 3)
        const real* v = arow < 3 ? jd : rd;
        const int f = arow < 3 ? arow : arow - 3;
        real acc = 0;
        for (int q = 0; q < 3; q++) { const real t = fr[3*f + q]; if (t != 0) acc += v[q]*t; }
        return acc;
      };
      real val = frame_row(a0);
      if (a1 >= 0) val = val + frame_row(a1)*scl;
      P.spJ[a++] = val;
    }
  }
  if (dual) { wv_sync(); return; }
  // transpose (mju_transposeSparse: row j of J' lists the constraints that contain dof j, ascending): lane = dof
  int cnt0 = 0, cnt1 = 0;
  for (int r = 0; r < nefc; r++) {
    const M128 pm = m128_ld(P.rowmask + 4*r);
    cnt0 += (int)((pm.lo >> lane) & 1);
    cnt1 += (int)((pm.hi >> lane) & 1);
  }
  const int n0 = wv_sum_i(cnt0);
  const int a0 = wv_exscan_i(cnt0), a1 = n0 + wv_exscan_i(cnt1);
  if (lane < nv) P.JTadr[lane] = a0;
  if (lane + MJH_W < nv) P.JTadr[lane + MJH_W] = a1;
  if (lane == 0) P.JTadr[nv] = base;
  wv_sync();
  int w0 = a0, w1 = a1;
  for (int r = 0; r < nefc; r++) {
    const M128 pm = m128_ld(P.rowmask + 4*r);
    const int adr = P.rowadr[r];
    if ((pm.lo >> lane) & 1) { P.JTrow[w0] = r; P.spJT[w0] = P.spJ[adr + m128_rank(pm, lane)]; w0++; }
    if ((pm.hi >> lane) & 1) { P.JTrow[w1] = r; P.spJT[w1] = P.spJ[adr + m128_rank(pm, lane + MJH_W)]; w1++; }
  }
  wv_sync();
}


// ------------------------------------------------------------------------------------------------------------
// The factor.  Row r of L holds its off-diagonal entries in ascending column order, then the diagonal, at
// Ladr[r]; Lmask[r] is the set of off-diagonal columns, so entry (r, j) sits at Ladr[r] + rank(Lmask[r], j).
// The routines below are out of line (their own register scope) and keep, per lane, the pattern / address /
// diagonal of the rows the lane is named after (rows lane and lane + 64): the serial sweeps fetch them with
// v_readlane instead of chasing them through memory.
// ------------------------------------------------------------------------------------------------------------

// H row pattern strictly below the diagonal for rows lane (hm0) and lane + 64 (hm1): the union of the patterns of
// the rows of J that contain the dof (mju_sqrMatTDSparseSymbolic), plus M's row (mju_addToMatSparse)
MJH_DEV void sp_hmask(MREF M, const Efc& P, M128 isl, int nv, M128& hm0, M128& hm1) {
  const int lane = wv_lane();
  hm0 = m128_zero(); hm1 = m128_zero();
  for (int slot = 0; slot < 2; slot++) {
    const int r = lane + MJH_W*slot;
    if (r < nv && m128_test(isl, r)) {
      M128 pat = m128_zero();
      const int t1 = P.JTadr[r + 1];
      for (int t = P.JTadr[r]; t < t1; t++) pat = m128_or(pat, m128_ld(P.rowmask + 4*P.JTrow[t]));
      const int ma = M.M_rowadr[r], mn = M.M_rownnz[r];
      for (int q = 0; q < mn; q++) pat = m128_or(pat, m128_bit(M.M_colind[ma + q]));
      pat = m128_and(pat, m128_below(r));
      if (slot) hm1 = pat; else hm0 = pat;
    }
  }
}

// pattern and address of row i of the factor from the lane registers that hold them (rows lane: slot 0, lane + 64:
// slot 1).  A row below 64 has no column above 63: its high word is known to be zero and is not fetched.
MJH_DEV void sp_row_bcast(M128 lm0, M128 lm1, int adr0, int adr1, int i, M128& lm, int& adr) {
  if (i < MJH_W) {
    lm.lo = wv_bcast_u64(lm0.lo, i);
    lm.hi = 0;
    adr = wv_bcast_i(adr0, i);
  } else {
    lm = wv_bcast_m128(lm1, i - MJH_W);
    adr = wv_bcast_i(adr1, i - MJH_W);
  }
}

// mju_cholFactorSymbolic for the island's rows: elimination-tree parents (spar), row patterns (Lmask), row addresses
// (Ladr); returns the number of stored entries.  Row r (descending): seeds = the rows i > r with H[i][r] != 0 in
// ascending order; from each seed the walk climbs the tree until it meets a row already visited for r; every row c
// it passes has L[c][r] != 0 and contributes its pattern below r to row r's.
MJH_DEVN_HOT int sp_symbolic(MREF M_, BREF B_, int e_, const Efc& P, M128 isl) {
  MJH_ENTER(M_, B_, e_);
  (void)B; (void)e;
  const int nv = M.s.nv, lane = wv_lane();
  isl = wv_uniform_m128(isl);
  M128 hm0, hm1;
  sp_hmask(M, P, isl, nv, hm0, hm1);
  M128 lm0 = m128_zero(), lm1 = m128_zero();
  int par0 = -1, par1 = -1;
  for (M128 rows = isl; m128_any(rows); ) {
    const int r = m128_highest(rows);
    rows = m128_xor(rows, m128_bit(r));
    M128 colm;
    colm.lo = wv_ballot(m128_test(hm0, r));
    colm.hi = wv_ballot(m128_test(hm1, r));
    M128 pat = wv_bcast_m128(r < MJH_W ? hm0 : hm1, r & (MJH_W - 1));
    M128 visited = m128_bit(r);
    const M128 below_r = m128_below(r);
    for (; m128_any(colm); colm = m128_drop_lowest(colm)) {
      int c = m128_lowest(colm);
      while (!m128_test(visited, c)) {
        int p = wv_bcast_i(c < MJH_W ? par0 : par1, c & (MJH_W - 1));
        if (p == -1) {
          p = r;
          if (lane == (c & (MJH_W - 1))) { if (c < MJH_W) par0 = r; else par1 = r; }
        }
        visited = m128_or(visited, m128_bit(c));
        pat = m128_or(pat, m128_and(wv_bcast_m128(c < MJH_W ? lm0 : lm1, c & (MJH_W - 1)), below_r));
        c = p;
      }
    }
    if (lane == (r & (MJH_W - 1))) { if (r < MJH_W) lm0 = pat; else lm1 = pat; }
  }
  const int c0 = (lane < nv && m128_test(isl, lane)) ? m128_count(lm0) + 1 : 0;
  const int c1 = (lane + MJH_W < nv && m128_test(isl, lane + MJH_W)) ? m128_count(lm1) + 1 : 0;
  const int n0 = wv_sum_i(c0), n1 = wv_sum_i(c1);
  const int a0 = wv_exscan_i(c0), a1 = n0 + wv_exscan_i(c1);
  if (lane < nv) { m128_st(P.Lmask + 4*lane, lm0); P.spar[lane] = par0; P.Ladr[lane] = a0; }
  if (lane + MJH_W < nv) { m128_st(P.Lmask + 4*(lane + MJH_W), lm1); P.spar[lane + MJH_W] = par1; P.Ladr[lane + MJH_W] = a1; }
  if (lane == 0) P.Ladr[nv] = n0 + n1;
  wv_sync();
  return n0 + n1;
}

// H = J' D J + M written into the factor's storage (mju_sqrMatTDSparseNumeric: row r accumulates, over the rows k of
// J that contain dof r in ascending order, (D[k] J[k][r]) * J[k][c]; mju_addToMatSparse), then mju_cholFactorNumeric:
// rows descending, dense[j] -= L[c][r] * L[c][j] over the rows c in the symbolic visiting order, scale by 1/L[r][r].
// Lane = column; the visiting order of a row is first walked into the lanes (list), its L[c][r] gathered in one go,
// then applied one c at a time.
template <class PL>
MJH_DEVN_HOT void sp_numeric(MREF M_, BREF B_, int e_, const Efc& P, PL L, M128 isl, crptr Dact, crptr Ms) {
  MJH_ENTER(M_, B_, e_);
  (void)B; (void)e;
  const int nv = M.s.nv, lane = wv_lane();
  isl = wv_uniform_m128(isl);
  const int in0 = lane < nv && m128_test(isl, lane), in1 = lane + MJH_W < nv && m128_test(isl, lane + MJH_W);
  M128 lm0 = m128_zero(), lm1 = m128_zero(), hm0, hm1;
  int par0 = -1, par1 = -1, adr0 = 0, adr1 = 0;
  if (in0) { lm0 = m128_ld(P.Lmask + 4*lane); par0 = P.spar[lane]; adr0 = P.Ladr[lane]; }
  if (in1) { lm1 = m128_ld(P.Lmask + 4*(lane + MJH_W)); par1 = P.spar[lane + MJH_W]; adr1 = P.Ladr[lane + MJH_W]; }
  sp_hmask(M, P, isl, nv, hm0, hm1);
  // ---- H rows, ascending
  for (M128 rows = isl; m128_any(rows); rows = m128_drop_lowest(rows)) {
    const int r = m128_lowest(rows);
    const M128 lmr = wv_bcast_m128(r < MJH_W ? lm0 : lm1, r & (MJH_W - 1));
    const int adr_r = wv_bcast_i(r < MJH_W ? adr0 : adr1, r & (MJH_W - 1));
    real acc0 = 0, acc1 = 0;
    const int t0 = P.JTadr[r], n = P.JTadr[r + 1] - t0;
    for (int tb = 0; tb < n; tb += MJH_W) {
      // the row's J' entries, one per lane: scale, pattern and address of the J row each one points to
      real sc = 0;
      M128 pm = m128_zero();
      int adr = 0;
      if (tb + lane < n) {
        const int t = t0 + tb + lane, k = P.JTrow[t];
        sc = Dact[k]*P.spJT[t];
        pm = m128_ld(P.rowmask + 4*k);
        adr = P.rowadr[k];
      }
      const int m = n - tb < MJH_W ? n - tb : MJH_W;
#ifdef MJH_SP_PIPELINE
      // (measurement build, -DMJH_SP_PIPELINE: 1 % SLOWER on the cube, profiles/r05/negative_results.txt -- not the default)
      // (four entries at a time: their loads are issued together, the additions stay in entry order.  An entry with a zero
      // scale, or a lane outside the entry's pattern, adds +-0 -- no accumulator here can be -0, it starts at +0 -- where the
      // reference skips the entry: the sum is the same)
      for (int u = 0; u < m; u += 4) {
        real scv[4], v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int uu = u + q < m ? u + q : m - 1;
          scv[q] = u + q < m ? wv_bcast(sc, uu) : (real)0;
          const M128 pmu = wv_bcast_m128(pm, uu);
          const int adru = wv_bcast_i(adr, uu);
          v0[q] = (lane <= r && m128_test(pmu, lane)) ? (real)P.spJ[adru + m128_rank_lane0(pmu)] : (real)0;
          v1[q] = 0;
          if (r >= MJH_W) v1[q] = (lane + MJH_W <= r && m128_test(pmu, lane + MJH_W)) ? (real)P.spJ[adru + m128_rank_lane1(pmu)] : (real)0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { acc0 += scv[q]*v0[q]; if (r >= MJH_W) acc1 += scv[q]*v1[q]; }
      }
#else
      for (int u = 0; u < m; u++) {
        const real scu = wv_bcast(sc, u);
        if (scu == 0) continue;
        const M128 pmu = wv_bcast_m128(pm, u);
        const int adru = wv_bcast_i(adr, u);
        if (lane <= r && m128_test(pmu, lane)) acc0 += scu*P.spJ[adru + m128_rank_lane0(pmu)];
        if (r >= MJH_W) { if (lane + MJH_W <= r && m128_test(pmu, lane + MJH_W)) acc1 += scu*P.spJ[adru + m128_rank_lane1(pmu)]; }
      }
#endif
    }
    const int ma = M.M_rowadr[r], mn = M.M_rownnz[r];
    for (int q = 0; q < mn; q++) {
      const int c = M.M_colind[ma + q];
      if (c == lane) acc0 += Ms[ma + q];
      if (c == lane + MJH_W) acc1 += Ms[ma + q];
    }
    const int cnt = m128_count(lmr);
    if (lane < r && m128_test(lmr, lane)) L[adr_r + m128_rank_lane0(lmr)] = acc0;
    if (lane + MJH_W < r && m128_test(lmr, lane + MJH_W)) L[adr_r + m128_rank_lane1(lmr)] = acc1;
    if (lane == r) L[adr_r + cnt] = acc0;
    if (lane + MJH_W == r) L[adr_r + cnt] = acc1;
  }
  wv_sync();
  // ---- reverse Cholesky, rows descending
  for (M128 rows = isl; m128_any(rows); ) {
    const int r = m128_highest(rows);
    rows = m128_xor(rows, m128_bit(r));
    const M128 lmr = wv_bcast_m128(r < MJH_W ? lm0 : lm1, r & (MJH_W - 1));
    const int adr_r = wv_bcast_i(r < MJH_W ? adr0 : adr1, r & (MJH_W - 1));
    const int cnt = m128_count(lmr);
    real d0 = 0, d1 = 0;
    if (lane < r && m128_test(lmr, lane)) d0 = L[adr_r + m128_rank_lane0(lmr)];
    if (lane + MJH_W < r && m128_test(lmr, lane + MJH_W)) d1 = L[adr_r + m128_rank_lane1(lmr)];
    if (lane == r) d0 = L[adr_r + cnt];
    if (lane + MJH_W == r) d1 = L[adr_r + cnt];
    // the visiting order of column r, walked into the lanes (list entry u in lane u % 64, slot u / 64)
    M128 colm;
    colm.lo = wv_ballot(m128_test(hm0, r));
    colm.hi = wv_ballot(m128_test(hm1, r));
    int nlist = 0, myc0 = -1, myc1 = -1;
    M128 visited = m128_bit(r);
    for (; m128_any(colm); colm = m128_drop_lowest(colm)) {
      int c = m128_lowest(colm);
      while (!m128_test(visited, c)) {
        if (lane == (nlist & (MJH_W - 1))) { if (nlist < MJH_W) myc0 = c; else myc1 = c; }
        nlist++;
        visited = m128_or(visited, m128_bit(c));
        c = wv_bcast_i(c < MJH_W ? par0 : par1, c & (MJH_W - 1));
      }
    }
    // every list entry's pattern, address and L[c][r], gathered by the lane that holds the entry
    M128 lc0 = m128_zero(), lc1 = m128_zero();
    int ac0 = 0, ac1 = 0;
    real v0 = 0, v1 = 0;
    if (myc0 >= 0) { lc0 = m128_ld(P.Lmask + 4*myc0); ac0 = P.Ladr[myc0]; v0 = L[ac0 + m128_rank(lc0, r)]; }
    if (myc1 >= 0) { lc1 = m128_ld(P.Lmask + 4*myc1); ac1 = P.Ladr[myc1]; v1 = L[ac1 + m128_rank(lc1, r)]; }
#ifdef MJH_SP_PIPELINE
    // (four rows of the visiting list at a time, loads together, subtractions in list order; a lane outside a row's pattern
    // subtracts L[c][r] * 0 = +-0)
    for (int u = 0; u < nlist; u += 4) {
      real Lcr[4], w0[4], w1[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int uu = u + q < nlist ? u + q : nlist - 1;
        const int src = uu & (MJH_W - 1);
        Lcr[q] = u + q < nlist ? wv_bcast(uu < MJH_W ? v0 : v1, src) : (real)0;
        const M128 lmu = wv_bcast_m128(uu < MJH_W ? lc0 : lc1, src);
        const int au = wv_bcast_i(uu < MJH_W ? ac0 : ac1, src);
        w0[q] = (lane <= r && m128_test(lmu, lane)) ? (real)L[au + m128_rank_lane0(lmu)] : (real)0;
        w1[q] = 0;
        if (r >= MJH_W) w1[q] = (lane + MJH_W <= r && m128_test(lmu, lane + MJH_W)) ? (real)L[au + m128_rank_lane1(lmu)] : (real)0;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) { d0 -= Lcr[q]*w0[q]; if (r >= MJH_W) d1 -= Lcr[q]*w1[q]; }
    }
#else
    for (int u = 0; u < nlist; u++) {
      const int src = u & (MJH_W - 1);
      const real Lcr = wv_bcast(u < MJH_W ? v0 : v1, src);
      const M128 lmu = wv_bcast_m128(u < MJH_W ? lc0 : lc1, src);
      const int au = wv_bcast_i(u < MJH_W ? ac0 : ac1, src);
      if (lane <= r && m128_test(lmu, lane)) d0 -= Lcr*L[au + m128_rank_lane0(lmu)];
      if (r >= MJH_W) { if (lane + MJH_W <= r && m128_test(lmu, lane + MJH_W)) d1 -= Lcr*L[au + m128_rank_lane1(lmu)]; }
    }
#endif
    real diag = wv_bcast(r < MJH_W ? d0 : d1, r & (MJH_W - 1));
    if (diag < MJH_MINVAL) diag = MJH_MINVAL;
    const real Lrr = sqrt(diag);
    const real inv = 1.0/Lrr;
    if (lane < r && m128_test(lmr, lane)) L[adr_r + m128_rank_lane0(lmr)] = d0*inv;
    if (lane + MJH_W < r && m128_test(lmr, lane + MJH_W)) L[adr_r + m128_rank_lane1(lmr)] = d1*inv;
    if (lane == (r & (MJH_W - 1))) L[adr_r + cnt] = Lrr;
    wv_sync();
  }
}

// mju_cholSolveSparse: x <- L^-T x over the rows descending (x[j] -= L[i][j] x[i]), then x <- L^-1 x ascending
// (x[i] -= mju_dotSparse(row i, x)); x in registers (y0: dof lane, y1: dof lane + 64)
// (the vector goes in and comes back BY VALUE: a reference parameter of an out-of-line routine lives in scratch memory,
// and the sweeps would round-trip it through there at every row)
struct SpVec2 { real y0, y1; };
template <class PL>
MJH_DEVN_HOT SpVec2 sp_solve(MREF M_, const Efc& P, PL L, M128 isl, real y0, real y1) {
  const MJH_CONST_AS DModel& M = wv_uniform_ref(M_);
  const int nv = M.s.nv, lane = wv_lane();
  isl = wv_uniform_m128(isl);
  const int in0 = lane < nv && m128_test(isl, lane), in1 = lane + MJH_W < nv && m128_test(isl, lane + MJH_W);
  M128 lm0 = m128_zero(), lm1 = m128_zero();
  int adr0 = 0, adr1 = 0;
  real dg0 = 1, dg1 = 1;
  if (in0) { lm0 = m128_ld(P.Lmask + 4*lane); adr0 = P.Ladr[lane]; dg0 = L[adr0 + m128_count(lm0)]; }
  if (in1) { lm1 = m128_ld(P.Lmask + 4*(lane + MJH_W)); adr1 = P.Ladr[lane + MJH_W]; dg1 = L[adr1 + m128_count(lm1)]; }
  for (M128 rows = isl; m128_any(rows); ) {
    const int i = m128_highest(rows), src = i & (MJH_W - 1);
    rows = m128_xor(rows, m128_bit(i));
    M128 lm;
    int adr;
    sp_row_bcast(lm0, lm1, adr0, adr1, i, lm, adr);
    const int hi = i > MJH_W;           // (uniform) the row has columns in the second slot
    const int t0 = lane < i && m128_test(lm, lane);
    const real l0 = t0 ? (real)L[adr + m128_rank_lane0(lm)] : (real)0;
    int t1 = 0;
    real l1 = 0;
    if (hi) { t1 = lane + MJH_W < i && m128_test(lm, lane + MJH_W); l1 = t1 ? (real)L[adr + m128_rank_lane1(lm)] : (real)0; }
    real xi = wv_bcast(i < MJH_W ? y0 : y1, src);
    if (xi == 0) continue;
    xi /= wv_bcast(i < MJH_W ? dg0 : dg1, src);
    if (i < MJH_W) { if (lane == i) y0 = xi; } else { if (lane == i - MJH_W) y1 = xi; }
    if (t0) y0 -= l0*xi;
    if (hi && t1) y1 -= l1*xi;
  }
  // (the line search's LDS block, idle during a solve: staging space for the row dot products; 128 entries suffice)
  const int ev = P.ev != nullptr;
  const auto evp = mjh_local(P.ev);
  for (M128 rows = isl; m128_any(rows); rows = m128_drop_lowest(rows)) {
    const int i = m128_lowest(rows), src = i & (MJH_W - 1);
    M128 lm;
    int adr;
    sp_row_bcast(lm0, lm1, adr0, adr1, i, lm, adr);
    real xi = wv_bcast(i < MJH_W ? y0 : y1, src);
    if (m128_any(lm)) {
      const real p0 = (lane < i && m128_test(lm, lane)) ? (real)(L[adr + m128_rank_lane0(lm)]*y0) : (real)0;
      real p1 = 0;
      if (i > MJH_W) p1 = (lane + MJH_W < i && m128_test(lm, lane + MJH_W)) ? (real)(L[adr + m128_rank_lane1(lm)]*y1) : (real)0;
      if (ev) {
        // (mju_dotSparse over the row's stored entries: the products go to the LDS block in storage order -- a lane's
        // entry sits at its rank in the row's pattern -- and the four accumulators walk them with one read and one
        // addition per entry, instead of a v_readlane pair per entry; then (r0 + r2) + (r1 + r3) and the tail in order)
        const int cnt = m128_count(lm);
        if (lane < i && m128_test(lm, lane)) evp[m128_rank_lane0(lm)] = p0;
        if (i > MJH_W && lane + MJH_W < i && m128_test(lm, lane + MJH_W)) evp[m128_rank_lane1(lm)] = p1;
        wv_sync();
        const int G = cnt >> 2, a = lane & 3;
        real r = 0;
        for (int g = 0; g < G; g++) r += evp[4*g + a];
        real res = (wv_bcast(r, 0) + wv_bcast(r, 2)) + (wv_bcast(r, 1) + wv_bcast(r, 3));
        for (int k = 4*G; k < cnt; k++) res += evp[k];
        wv_sync();
        xi -= res;
      } else
      xi -= wv_dot4m(p0, p1, lm.lo, lm.hi, 0);
    }
    xi /= wv_bcast(i < MJH_W ? dg0 : dg1, src);
    if (i < MJH_W) { if (lane == i) y0 = xi; } else { if (lane == i - MJH_W) y1 = xi; }
  }
  return SpVec2{y0, y1};
}

// mju_cholUpdateSparse(L, x, flg_plus): rows from the last non-zero of x downwards, Givens rotation of (row, x) for
// every row whose x entry is non-zero; x in registers, xm its pattern.  Returns the number of clamped pivots.
// Only the rows that can hold a non-zero are visited: the pattern of x grown by the pattern of every rotated row.
template <class PL>
MJH_DEVN_HOT int sp_update(MREF M_, const Efc& P, PL L, real x0, real x1, M128 xm, int flg_plus) {
  const MJH_CONST_AS DModel& M = wv_uniform_ref(M_);
  const int nv = M.s.nv, lane = wv_lane();
  xm = wv_uniform_m128(xm);
  M128 lm0 = m128_zero(), lm1 = m128_zero();
  int adr0 = 0, adr1 = 0;
  real dg0 = 1, dg1 = 1;
  if (lane < nv) { lm0 = m128_ld(P.Lmask + 4*lane); adr0 = P.Ladr[lane]; dg0 = L[adr0 + m128_count(lm0)]; }
  if (lane + MJH_W < nv) { lm1 = m128_ld(P.Lmask + 4*(lane + MJH_W)); adr1 = P.Ladr[lane + MJH_W]; dg1 = L[adr1 + m128_count(lm1)]; }
  int clamped = 0;
  M128 nz = xm;
  while (m128_any(nz)) {
    const int row = m128_highest(nz), src = row & (MJH_W - 1);
    nz = m128_xor(nz, m128_bit(row));
    const real xr = wv_bcast(row < MJH_W ? x0 : x1, src);
    if (xr == 0) continue;
    M128 lm;
    int adr;
    sp_row_bcast(lm0, lm1, adr0, adr1, row, lm, adr);
    // (the pivot travels in its owner's register: no lane reads it from memory while the owner rewrites it)
    const real diag = wv_bcast(row < MJH_W ? dg0 : dg1, src);
    nz = m128_or(nz, lm);
    const int hi = row > MJH_W;         // (uniform) the row has columns in the second slot
    const int t0 = lane < row && m128_test(lm, lane);
    const int k0 = adr + m128_rank_lane0(lm), kd = adr + m128_count(lm);
    const real m0 = t0 ? (real)L[k0] : (real)0;
    int t1 = 0, k1 = 0;
    real m1 = 0;
    if (hi) { t1 = lane + MJH_W < row && m128_test(lm, lane + MJH_W); k1 = adr + m128_rank_lane1(lm); m1 = t1 ? (real)L[k1] : (real)0; }
    real tmp = diag*diag + (flg_plus ? xr*xr : -xr*xr);
    if (tmp < MJH_MINVAL) { tmp = MJH_MINVAL; clamped++; }
    const real rr = sqrt(tmp);
    const real c = diag/rr;
    const real sn = -xr/rr;
    const real ss = flg_plus ? -sn : sn;
    if (lane == src) { if (row < MJH_W) dg0 = rr; else dg1 = rr; L[kd] = rr; }
    if (t0) { L[k0] = c*m0 + ss*x0; x0 = sn*m0 + c*x0; }
    if (hi && t1) { L[k1] = c*m1 + ss*x1; x1 = sn*m1 + c*x1; }
  }
  wv_sync();
  return clamped;
}

// Several rank-one updates in ONE sweep over the rows.  The reference applies the updates of a Newton iteration one
// after the other, each as a sweep from its last non-zero row down to row 0 (HessianIncremental -> mju_cholUpdateSparse).
// Update u rotates row r using only row r of the factor and its own vector x_u; so update u+1 may rotate row r as soon
// as update u has -- before update u has gone on to row r-1.  Visiting the rows once, top down, and applying at each row
// the pending rotations in update order performs exactly the reference's operations on exactly the same operands; the
// row's pattern / address / values are fetched and stored once for all of them.
#define MJH_SP_KB 8
// the caller names the constraint rows (two 16-bit indices per word); their scaled Jacobian rows J[i]*sqrt(D[i]) are
// spread over the dof lanes here, so that the register-hungry part stays out of the solver's own frame
struct SpBatch {
  int rows[MJH_SP_KB/2];
  int plus;                               // bit u: update (1) or downdate (0)
  int n;
};
struct SpBatchRegs { real x0[MJH_SP_KB], x1[MJH_SP_KB]; M128 nz[MJH_SP_KB]; int n, plus; };
template <class PL>
MJH_DEVN_HOT int sp_update_batch(MREF M_, const Efc& P, PL L, SpBatch bi) {
  const MJH_CONST_AS DModel& M = wv_uniform_ref(M_);
  const int nv = M.s.nv, lane = wv_lane();
  SpBatchRegs b;
  b.n = wv_uniform_i(bi.n); b.plus = wv_uniform_i(bi.plus);
#pragma unroll
  for (int u = 0; u < MJH_SP_KB; u++) {
    b.x0[u] = 0; b.x1[u] = 0; b.nz[u] = m128_zero();
    if (u < b.n) {
      const int i = wv_uniform_i((bi.rows[u >> 1] >> (16*(u & 1))) & 0xffff);
      const M128 pm = m128_ld(P.rowmask + 4*i);
      const int adr = P.rowadr[i];
      const real scl = sqrt(P.D[i]);
      b.nz[u] = pm;
      if (lane < nv && m128_test(pm, lane)) b.x0[u] = P.spJ[adr + m128_rank(pm, lane)]*scl;
      if (lane + MJH_W < nv && m128_test(pm, lane + MJH_W)) b.x1[u] = P.spJ[adr + m128_rank(pm, lane + MJH_W)]*scl;
    }
  }
  M128 lm0 = m128_zero(), lm1 = m128_zero();
  int adr0 = 0, adr1 = 0;
  real dg0 = 1, dg1 = 1;
  if (lane < nv) { lm0 = m128_ld(P.Lmask + 4*lane); adr0 = P.Ladr[lane]; dg0 = L[adr0 + m128_count(lm0)]; }
  if (lane + MJH_W < nv) { lm1 = m128_ld(P.Lmask + 4*(lane + MJH_W)); adr1 = P.Ladr[lane + MJH_W]; dg1 = L[adr1 + m128_count(lm1)]; }
  const int n = b.n, plus = b.plus;
  M128 all = m128_zero();
#pragma unroll
  for (int u = 0; u < MJH_SP_KB; u++) { b.nz[u] = wv_uniform_m128(b.nz[u]); if (u < n) all = m128_or(all, b.nz[u]); }
  int clamped = 0;
  while (m128_any(all)) {
    const int row = m128_highest(all), src = row & (MJH_W - 1);
    all = m128_xor(all, m128_bit(row));
    M128 lm;
    int adr;
    sp_row_bcast(lm0, lm1, adr0, adr1, row, lm, adr);
    const int hi = row > MJH_W;
    const int t0 = lane < row && m128_test(lm, lane);
    const int k0 = adr + m128_rank_lane0(lm), kd = adr + m128_count(lm);
    real m0 = t0 ? (real)L[k0] : (real)0;
    int t1 = 0, k1 = 0;
    real m1 = 0;
    if (hi) { t1 = lane + MJH_W < row && m128_test(lm, lane + MJH_W); k1 = adr + m128_rank_lane1(lm); m1 = t1 ? (real)L[k1] : (real)0; }
    real d = wv_bcast(row < MJH_W ? dg0 : dg1, src);
    int touched = 0;
#pragma unroll
    for (int u = 0; u < MJH_SP_KB; u++) {
      if (u >= n || !m128_test(b.nz[u], row)) continue;
      const real xr = wv_bcast(row < MJH_W ? b.x0[u] : b.x1[u], src);
      if (xr == 0) continue;
      b.nz[u] = m128_or(b.nz[u], lm);
      all = m128_or(all, lm);
      const int up = (plus >> u) & 1;
      real tmp = d*d + (up ? xr*xr : -xr*xr);
      if (tmp < MJH_MINVAL) { tmp = MJH_MINVAL; clamped++; }
      const real rr = sqrt(tmp);
      const real c = d/rr;
      const real sn = -xr/rr;
      const real ss = up ? -sn : sn;
      if (t0) { const real nm = c*m0 + ss*b.x0[u]; b.x0[u] = sn*m0 + c*b.x0[u]; m0 = nm; }
      if (hi && t1) { const real nm = c*m1 + ss*b.x1[u]; b.x1[u] = sn*m1 + c*b.x1[u]; m1 = nm; }
      d = rr;
      touched = 1;
    }
    if (touched) {
      if (lane == src) { if (row < MJH_W) dg0 = d; else dg1 = d; L[kd] = d; }
      if (t0) L[k0] = m0;
      if (hi && t1) L[k1] = m1;
    }
  }
  wv_sync();
  return clamped;
}

#endif  // !MJH_LANE_MODE
