// Compressed constraint Jacobian with explicit column indices, for models beyond the 128 dofs the mask form of
// mjh_sparse.h covers (flexes: jelly.xml has 1536).  The reference keeps efc_J compressed whenever mj_isSparse holds
// (engine_core_util.c:32); a row stores the dofs of the bodies it touches in ascending order
// (mj_jacDifPair / mj_jacSum, engine_core_util.c:437-600, common dofs of the two chains left out for contacts), products
// J v and J' f are mju_dotSparse sums over the stored entries (four accumulators by position, engine_util_sparse.c).
// This header builds the same rows -- values by the expressions of the dense path (mjh_constraint.h), kept next to
// their column indices -- and their transpose; the CG solver (mjh_newton.h, SPA = 2) and the row products of
// mj_referenceConstraint / mj_fwdConstraint read them instead of streaming nv-wide dense rows.
//
// Scope (mjh_model_build.h: s.csr): CG, no tendon rows and no equality rows other than flex edge constraints (dense
// rows would have to be cut by a scan; an edge constraint's row is the model's flexedge_J row),
// contacts up to condim 3, islands enabled.  (included once per SPMD mode by mjh_stages.inc: no include guard)

#if !MJH_LANE_MODE

// dofs of body b's chain into out[] (descending), returns the count: the walk from the last dof of the weld body through
// dof_parentid, tabulated at upload (M.body_chain) so that the loads do not depend on each other
template <class IP>
MJH_DEV int csr_body_chain(MREF M, int b, IP out) {
  const int a0 = M.body_chainadr[b], n = M.body_chainadr[b + 1] - a0;
  for (int i = 0; i < n; i++) out[i] = M.body_chain[a0 + i];
  return n;
}

// the stored dofs of contact k's rows, ascending; returns their number.  One body on each side: the merged chains with the
// dofs common to both removed (flg_skipcommon of mj_jacDifPair); a flex element on a side: the union of the chains of all
// bodies involved (mj_jacSum / mju_addToSparseMat).  (contact_sides, mjh_flex.h)
template <class IP>
MJH_DEV int csr_contact_cols(MREF M, BREF B, int e, int k, IP cols) {
  ConSides S;
  contact_sides(M, B, e, k, S);
  int n = 0;
  for (int q = 0; q < S.n; q++) n += csr_body_chain(M, cs_body(S, q), cols + n);
  for (int a = 1; a < n; a++) { const int c = cols[a]; int b = a - 1; while (b >= 0 && cols[b] > c) { cols[b + 1] = cols[b]; b--; } cols[b + 1] = c; }
  int m = 0;
  for (int a = 0; a < n; a++) {
    if (S.simple) { if (a + 1 < n && cols[a] == cols[a + 1]) { a++; continue; } }
    else if (m > 0 && cols[m - 1] == cols[a]) continue;
    cols[m++] = cols[a];
  }
  return m;
}

// The same for merged chains of at most 16 dofs (csr_rowmax <= 16: a flex element against a geom is 12 + the geom's
// chain), entirely in registers: the private array of the general form lives in scratch memory, and its insertion sort is
// a chain of dependent memory round trips per contact.  Slots are filled from the chain table by position, sorted by a
// 63-exchange network (Batcher's odd-even merge sort); in the two-body form equal entries cancel in pairs (a dof common to
// two chains is not stored), in the weighted-sum form one of each run stays.  v: the sorted entries, keep: bit i set if v[i]
// is stored, m: their number.
struct CsrCols16 { int v[16]; unsigned keep; int m; };
MJH_DEV void csr_contact_cols16(MREF M, BREF B, int e, int k, CsrCols16& R) {
  ConSides S;
  contact_sides(M, B, e, k, S);
  int adr[8], start[9];
  start[0] = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    int len = 0;
    adr[j] = 0;
    if (j < S.n) { adr[j] = M.body_chainadr[S.body[j]]; len = M.body_chainadr[S.body[j] + 1] - adr[j]; }
    start[j + 1] = start[j] + len;
  }
#pragma unroll
  for (int q = 0; q < 16; q++) {
    int src = -1;
#pragma unroll
    for (int j = 0; j < 8; j++) if (q >= start[j] && q < start[j + 1]) src = adr[j] + (q - start[j]);
    R.v[q] = src >= 0 ? (int)M.body_chain[src] : 0x7fffffff;
  }
  constexpr unsigned char NA[63] = {0,2,4,6,8,10,12,14,0,1,4,5,8,9,12,13,1,5,9,13,0,1,2,3,8,9,10,11,2,3,10,11,1,3,5,9,11,13,0,1,2,3,4,5,6,7,4,5,6,7,2,3,6,7,10,11,1,3,5,7,9,11,13};
  constexpr unsigned char NB[63] = {1,3,5,7,9,11,13,15,2,3,6,7,10,11,14,15,2,6,10,14,4,5,6,7,12,13,14,15,4,5,12,13,2,4,6,10,12,14,8,9,10,11,12,13,14,15,8,9,10,11,4,5,8,9,12,13,2,4,6,8,10,12,14};
#pragma unroll
  for (int t = 0; t < 63; t++) {
    const int a = R.v[NA[t]], b = R.v[NB[t]];
    R.v[NA[t]] = a < b ? a : b;
    R.v[NB[t]] = a < b ? b : a;
  }
  // (two-body form: equal entries cancel in pairs from the left, as in the general form -- the last of a run of odd length
  // stays; weighted-sum form: the last of every run stays)
  unsigned keep = 0;
  int m = 0, run = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    run = (i > 0 && R.v[i] == R.v[i - 1]) ? run + 1 : 1;
    const int last = i == 15 || R.v[i] != R.v[i + 1];
    if (last && ((run & 1) || !S.simple) && R.v[i] != 0x7fffffff) { keep |= 1u << i; m++; }
  }
  R.keep = keep; R.m = m;
}

// stored dofs of the rows of equality constraint id (connect / weld: both body chains, common dofs kept -- mj_jacDifPair
// without flg_skipcommon, engine_core_constraint.c:655, :676; joint couplings: the joints' dofs), ascending
template <class IP>
MJH_DEV int csr_equality_cols(MREF M, int id, IP cols) {
  const int et = M.eq_type[id];
  int o1 = M.eq_obj1id[id], o2 = M.eq_obj2id[id];
  int n = 0;
  if (et == MJH_EQ_CONNECT || et == MJH_EQ_WELD) {
    if (M.eq_objsite[id]) { o1 = M.site_bodyid[o1]; o2 = M.site_bodyid[o2]; }
    n = csr_body_chain(M, o1, cols);
    n += csr_body_chain(M, o2, cols + n);
  } else {
    cols[n++] = M.jnt_dofadr[o1];
    if (o2 >= 0) cols[n++] = M.jnt_dofadr[o2];
  }
  for (int a = 1; a < n; a++) { const int c = cols[a]; int b = a - 1; while (b >= 0 && cols[b] > c) { cols[b + 1] = cols[b]; b--; } cols[b + 1] = c; }
  int m = 0;
  for (int a = 0; a < n; a++) { if (m > 0 && cols[m - 1] == cols[a]) continue; cols[m++] = cols[a]; }
  return m;
}

MJH_DEVN void stage_csr_rows(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ncon = counts[MJH_C_NCON];
  if (!nefc) { if (wv_lane() == 0) counts[MJH_C_NJ] = 0; wv_sync(); return; }
  const int nv = s.nv;
  const int ispyramid = M.o.cone == 0;
  // (the arrays sized by nJ take their global homes while nJ is unknown: same convention as stage_sparsify)
  Efc P;
  efc_layout(M, B, e, nefc, P);
  iptr rowadr = P.rowadr;
  crptr Jd = MJH_G(B, efc_J, e);
  crptr cdof = MJH_F(B, cdof, e);
  crptr subtree_com = MJH_F(B, subtree_com, e);

  const int small = s.csr_rowmax <= 16;       // merged chains fit the register form (csr_contact_cols16)
#ifdef MJH_PROFILE
  // (profile builds, slots 57..59: row lengths + addresses | columns and values | transpose)
  long long ptick = wv_clock();
  auto tick = [&](int slot) { const long long c_ = wv_clock(); if (wv_lane() == 0) MJH_G(B, prof, e)[slot] += (real)(c_ - ptick)*0.01; ptick = c_; };
#else
  auto tick = [](int) {};
#endif

  // ---- pass 1: stored entries per row (rowadr[r + 1] <- nnz of row r)
  MJH_FOR_LANES(r, nefc) {
    const int type = P.type[r], id = P.id[r];
    int nnz = 0;
    if (type == MJH_CNSTR_EQUALITY) {
      const int et = M.eq_type[id];
      if (et == MJH_EQ_FLEX) {
        // (flex edge constraints: the edge's flexedge_J row)
        const int ed = M.eqrow_edge[M.eq_rowadr[id] + (r - MJH_G(B, eq_efcadr, e)[id])];
        nnz = M.flexedge_J_rownnz[ed];
      } else if (et == MJH_EQ_FLEXVERT) {
        nnz = M.fv_rownnz[M.eqrow_edge[M.eq_rowadr[id] + (r - MJH_G(B, eq_efcadr, e)[id])]];
      } else {
        int cols[MJH_CSR_CHAIN_MAX];
        nnz = csr_equality_cols(M, id, cols);
      }
    }
    else if (type == MJH_CNSTR_FRICTION_DOF) nnz = 1;
    else if (type == MJH_CNSTR_LIMIT_JOINT) nnz = M.jnt_type[id] == MJH_JNT_BALL ? 3 : 1;
    else if (type >= MJH_CNSTR_CONTACT_FRICTIONLESS) nnz = -1;          // set by its contact below
    if (nnz >= 0) rowadr[r + 1] = nnz;
  }
  wv_sync();
  MJH_FOR_LANES(k, ncon) {
    const int r0 = MJH_CON(B, con_efcadr, e, 1, k)[0];
    if (r0 < 0) continue;
    const int dim = MJH_CON(B, con_dim, e, 1, k)[0];
    const int nrow = dim == 1 ? 1 : (ispyramid ? 2*(dim - 1) : dim);
    int nnz;
    if (small) { CsrCols16 R; csr_contact_cols16(M, B, e, k, R); nnz = R.m; }
    else {
      int cols[MJH_CSR_CHAIN_MAX];
      nnz = csr_contact_cols(M, B, e, k, cols);
    }
    for (int a = 0; a < nrow; a++) rowadr[r0 + a + 1] = nnz;
  }
  wv_sync();
  // exclusive scan of the counts -> row addresses
  int total = 0;
  for (int r0 = 0; r0 < nefc; r0 += MJH_W) {
    const int r = r0 + wv_lane();
    const int c = r < nefc ? rowadr[r + 1] : 0;
    const int before = wv_exscan_i(c);
    const int sum = wv_sum_i(c);
    wv_sync();
    if (r < nefc) rowadr[r + 1] = total + before + c;
    total += sum;
  }
  if (wv_lane() == 0) { rowadr[0] = 0; counts[MJH_C_NJ] = total; }
  wv_sync();
  // (nJ is known now: with the CU's whole LDS block to itself -- launches of one workgroup per CU -- the layout gives
  // the compressed rows and their transpose LDS slots, and they are written there directly)
  efc_layout(M, B, e, nefc, P);
  iptr colind = P.colind;
  rptr val = P.spJ;
  iptr JTadr = P.JTadr;
  iptr JTrow = P.JTrow;
  rptr JTval = P.spJT;
  tick(57);
  // ---- pass 2: columns and values
  MJH_FOR_LANES(r, nefc) {
    const int type = P.type[r], id = P.id[r];
    const int a0 = rowadr[r];
    if (type == MJH_CNSTR_EQUALITY) {
      if (M.eq_type[id] == MJH_EQ_FLEX) {
        const int ed = M.eqrow_edge[M.eq_rowadr[id] + (r - MJH_G(B, eq_efcadr, e)[id])];
        const int f0 = M.flexedge_J_rowadr[ed], fn = M.flexedge_J_rownnz[ed];
        crptr fJ = MJH_F(B, flexedge_J, e);
        for (int q = 0; q < fn; q++) { colind[a0 + q] = M.flexedge_J_colind[f0 + q]; val[a0 + q] = fJ[f0 + q]; }
      } else if (M.eq_type[id] == MJH_EQ_FLEXVERT) {
        // (vertex constraints: the vertex's flexvert_J row)
        const int vr = M.eqrow_edge[M.eq_rowadr[id] + (r - MJH_G(B, eq_efcadr, e)[id])];
        const int f0 = M.fv_rowadr[vr], fn = M.fv_rownnz[vr];
        crptr fJ = MJH_G(B, flexvert_J, e);
        for (int q = 0; q < fn; q++) { colind[a0 + q] = M.fv_colind[f0 + q]; val[a0 + q] = fJ[f0 + q]; }
      } else {
        // connect / weld / joint couplings: cut from the dense row stage_equality_rows wrote
        int cols[MJH_CSR_CHAIN_MAX];
        const int n = csr_equality_cols(M, id, cols);
        for (int q = 0; q < n; q++) { colind[a0 + q] = cols[q]; val[a0 + q] = Jd[(size_t)r*nv + cols[q]]; }
      }
    }
    else if (type == MJH_CNSTR_FRICTION_DOF) { colind[a0] = id; val[a0] = Jd[(size_t)r*nv + id]; }
    else if (type == MJH_CNSTR_LIMIT_JOINT) {
      const int d0 = M.jnt_dofadr[id], nd = M.jnt_type[id] == MJH_JNT_BALL ? 3 : 1;
      for (int q = 0; q < nd; q++) { colind[a0 + q] = d0 + q; val[a0 + q] = Jd[(size_t)r*nv + d0 + q]; }
    }
  }
  // contacts, (a): one lane per contact lists the stored dofs in the first row's column slots and deals the (contact,
  // column) items of (b) -- into the transpose's row array, which is not written before the transpose below
  iptr items = JTrow;
  int nitems = 0;
  for (int k0 = 0; k0 < ncon; k0 += MJH_W) {
    const int k = k0 + wv_lane();
    int m = 0, a0 = 0;
    int cols[MJH_CSR_CHAIN_MAX];
    CsrCols16 R;
    R.keep = 0;
    if (k < ncon) {
      const int r0 = MJH_CON(B, con_efcadr, e, 1, k)[0];
      if (r0 >= 0) {
        if (small) { csr_contact_cols16(M, B, e, k, R); m = R.m; }
        else m = csr_contact_cols(M, B, e, k, cols);
        a0 = rowadr[r0];
      }
    }
    const int before = wv_exscan_i(m);
    const int sum = wv_sum_i(m);
    if (small) {
      int c = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) if ((R.keep >> i) & 1) { colind[a0 + c] = R.v[i]; items[nitems + before + c] = k*MJH_CSR_CHAIN_MAX + c; c++; }
    } else {
      for (int c = 0; c < m; c++) { colind[a0 + c] = cols[c]; items[nitems + before + c] = k*MJH_CSR_CHAIN_MAX + c; }
    }
    nitems += sum;
  }
  wv_sync();
  // (b): one lane per (contact, stored dof): the column of the contact's rows -- on every wavefront of a multi-wavefront
  // workgroup (csr_row_values, mjh_csrpass.h)
  {
    CsrRowArgs ra;
    ra.nitems = nitems; ra.ispyramid = ispyramid; ra.colind = colind; ra.val = val; ra.rowadr = rowadr; ra.items = items;
    MJH_WIDE_ARGS(MJH_MWS_CSRVALS, ra, csr_row_values(M, B, e, ra));
  }

  tick(58);
  // ---- transpose: entries of every dof in ascending row order (mju_transposeSparse): counting pass with integer atomics,
  //      one lane per stored entry scatters it (its row by bisection of the row addresses), then each dof sorts its
  //      (short) list by row.  The cursors sit in the unused tail of the LDS regions when it has room.
  iptr cursor = MJH_G(B, csr_idof, e);
  if (P.free_bytes >= nv*(int)sizeof(int)) cursor = SP<int>{(int*)P.free_p, 1};
  MJH_FOR_LANES(j, nv + 1) JTadr[j] = 0;
  MJH_FOR_LANES(j, nv) cursor[j] = 0;
  wv_sync();
  MJH_FOR_LANES(q, total) wv_atomic_add_i(&JTadr[colind[q] + 1], 1);
  wv_sync();
  {
    int run = 0;
    for (int j0 = 0; j0 <= nv; j0 += MJH_W) {
      const int j = j0 + wv_lane();
      const int c = j <= nv ? JTadr[j] : 0;
      const int before = wv_exscan_i(c);
      const int sum = wv_sum_i(c);
      wv_sync();
      if (j <= nv) JTadr[j] = run + before + c;
      run += sum;
    }
  }
  wv_sync();
  MJH_FOR_LANES(q, total) {
    int lo = 0, hi = nefc;                                 // the row r with rowadr[r] <= q < rowadr[r + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rowadr[mid] <= q) lo = mid; else hi = mid; }
    const int j = colind[q];
    const int pos = JTadr[j] + wv_atomic_add_i(&cursor[j], 1);
    JTrow[pos] = lo; JTval[pos] = val[q];
  }
  wv_sync();
  MJH_FOR_LANES(j, nv) {
    const int a0 = JTadr[j], a1 = JTadr[j + 1];
    for (int a = a0 + 1; a < a1; a++) {
      const int rr = JTrow[a]; const real vv = JTval[a];
      int b = a - 1;
      while (b >= a0 && JTrow[b] > rr) { JTrow[b + 1] = JTrow[b]; JTval[b + 1] = JTval[b]; b--; }
      JTrow[b + 1] = rr; JTval[b + 1] = vv;
    }
  }
  wv_sync();
  tick(59);
}

#endif   // !MJH_LANE_MODE
