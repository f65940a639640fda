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
MJH_DEV M128 sp_row_pattern(MREF M, BREF B, int e, int type, int id) {
  if (type == MJH_CNSTR_FRICTION_DOF) return m128_bit(id);
  if (type == MJH_CNSTR_LIMIT_JOINT) {
    const int adr = M.jnt_dofadr[id];
    if (M.jnt_type[id] == MJH_JNT_BALL) return m128_or(m128_bit(adr), m128_or(m128_bit(adr + 1), m128_bit(adr + 2)));
    return m128_bit(adr);
  }
  if (type == MJH_CNSTR_FRICTION_TENDON || type == MJH_CNSTR_LIMIT_TENDON) return sp_tendon_pattern(M, id);
  if (type >= MJH_CNSTR_CONTACT_FRICTIONLESS) {
    ciptr cg = MJH_CON(B, con_geom, e, 2, id);
    return m128_xor(sp_body_chain(M, M.geom_bodyid[cg[0]]), sp_body_chain(M, M.geom_bodyid[cg[1]]));
  }
  const int et = M.eq_type[id];
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
      const M128 pm = sp_row_pattern(M, B, e, P.type[r], P.id[r]);
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
  MJH_FOR_LANES(r, nefc) {
    M128 pm = m128_ld(P.rowmask + 4*r);
    int a = P.rowadr[r];
    crptr Jr = J + (size_t)r*nv;
    while (m128_any(pm)) { const int j = m128_lowest(pm); pm = m128_drop_lowest(pm); P.spJ[a++] = Jr[j]; }
  }
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

#endif  // !MJH_LANE_MODE
