// Constraint assembly stages: mj_makeConstraint, mj_projectConstraint (Y, AR), mj_referenceConstraint,
// mj_constraintUpdate -- dense Jacobian, pyramidal/frictionless contacts, joint/tendon limits,
// dof/tendon friction loss.  One wavefront per environment.  (reference: engine_core_constraint.c)
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)


// ------------------------------------------------------------------------------------------------
// constraint arrays of one environment
// Their sizes are known only once nefc is: with an LDS plan they are packed, smallest and most
// latency-critical first, into the dynamic LDS region [dyn_off, lds_bytes) that the plan keeps
// free from MJH_T_MAKE to MJH_T_CONSTRAINT; whatever does not fit stays in its global home.  The
// packing is a pure function of (nefc, plan), so every stage recomputes the same pointers.
// ------------------------------------------------------------------------------------------------
struct Efc {
  rptr force, b, ARinv, fprev, fmom, R, D, floss, aref, jar, ARf, pos, margin, KBIP,
       diagA, vel, sqrtInvD, AR, J, Y, cone;
  iptr order, state, type, id, island;
  // sparse constraint path (mjh_sparse.h): compressed J and its transpose, packed factor(s), row patterns
  // (the factor is stored compressed by its symbolic pattern; spL is an LDS slot of spL_cap entries when the plan has
  // room -- a factorisation with more fill than that works in spL_home -- else spL == spL_home)
  rptr spJ, spJT, spL, spLc, spL_home, vec;
  int spL_cap;
  iptr rowmask, rowadr, JTadr, JTrow, Lmask, Ladr, spar;
  iptr colind;       // explicit column indices of the compressed rows (mjh_csr.h), else unused
  // the larger of the two regions' unused tails (staging space for stage_project)
  char* free_p;
  int free_bytes;
  // primal solvers' line search: LDS block of one 64-row group's addends (6 reals per row), or null
  real* ev;
};

// Two regions of the workgroup's LDS block take these arrays:
//   region 1 [dyn_off, lds_bytes)  free from MJH_T_MAKE on: arrays written by constraint assembly
//   region 2 [dyn2_off, dyn_off)   holds the fields that die with MJH_T_MAKE (cdof, subtree_com, the
//                                  contact slots, tendon rows); free from MJH_T_PROJECT on, it takes the
//                                  arrays born there (AR first -- the PGS sweep reads it every row)
// Packing order = priority; an array either fits entirely or stays in its global home.
// lists: (member, global home expression, element count, region it is born in)
#define MJH_EFC_REAL_ARRAYS(X)                                       \
  X(force, MJH_G(B, efc_force, e), nefc, 1)                          \
  X(b, MJH_G(B, efc_b, e), nefc, 1)                                  \
  X(floss, MJH_G(B, efc_frictionloss, e), nefc, 1)                   \
  X(AR, MJH_G(B, efc_AR, e), dual_*nefc*nefc, 2)                     \
  X(R, MJH_G(B, efc_R, e), nefc, 1)                                  \
  X(D, MJH_G(B, efc_D, e), nefc, 1)                                  \
  X(aref, MJH_G(B, efc_aref, e), nefc, 1)                            \
  X(jar, MJH_G(B, scratch, e) + 3*nmax, nefc, 1)                     \
  X(ARf, MJH_G(B, scratch, e) + 4*nmax, nefc, 1)                     \
  X(pos, MJH_G(B, efc_pos, e), nefc, 1)                              \
  X(margin, MJH_G(B, efc_margin, e), nefc, 1)                        \
  X(KBIP, MJH_G(B, efc_KBIP, e), 4*nefc, 1)                          \
  X(diagA, MJH_G(B, efc_diagA, e), nefc, 1)                          \
  X(vel, MJH_G(B, efc_vel, e), nefc, 1)                              \
  X(sqrtInvD, MJH_G(B, scratch, e) + 5*nmax, dual_*nv, 2)            \
  X(Y, MJH_G(B, efc_Y, e), dual_*nefc*nv, 2)                         \
  X(J, MJH_G(B, efc_J, e), (1 - sp_*primal_)*nefc*nv, 1)             \
  X(ARinv, MJH_G(B, scratch, e), nefc, 1)                            \
  X(fprev, MJH_G(B, scratch, e) + nmax, nefc, 1)                     \
  X(fmom, MJH_G(B, scratch, e) + 2*nmax, nefc, 1)                    \
  X(cone, MJH_G(B, efc_cone, e), nefc, 1)                            \
  X(spJ, MJH_G(B, sp_J, e), (1 - csr_ + csrl_)*nJ_, 1)                    \
  X(spJT, MJH_G(B, sp_JT, e), (1 - csr_ + csrl_)*primal_*nJ_, 1)
#define MJH_EFC_INT_ARRAYS(X)                                        \
  X(order, MJH_G(B, iscratch, e), nefc, 1)                           \
  X(state, MJH_G(B, efc_state, e), nefc, 1)                          \
  X(type, MJH_G(B, efc_type, e), nefc, 1)                            \
  X(id, MJH_G(B, efc_id, e), nefc, 1)                                \
  X(island, MJH_G(B, efc_island, e), nefc, 1)                        \
  X(rowadr, MJH_G(B, sp_rowadr, e), sp_*(nefc + 1), 1)               \
  X(JTadr, MJH_G(B, sp_JTadr, e), sp_*primal_*(nv + 1), 1)           \
  X(rowmask, MJH_G(B, sp_rowmask, e), 4*spm_*nefc, 1)
// int arrays packed after the real ones (sized by nJ, see efc_layout)
#define MJH_EFC_LATE_INT_ARRAYS(X)                                   \
  X(JTrow, MJH_G(B, sp_JTrow, e), (1 - csr_ + csrl_)*primal_*nJ_, 1)      \
  X(colind, MJH_G(B, sp_colind, e), csrl_*nJ_, 1)

// (stage_project) does this array already live in the LDS plan?  (pointer inside the workgroup's block)
template <class T> MJH_DEV int mjh_staged_home_impl(const SP<T>& v, const char* lds, int bytes) {
  const char* p = (const char*)v.p;
  return bytes > 0 && p >= lds && p < lds + bytes;
}
#define mjh_staged_home(v) mjh_staged_home_impl((v), MJH_LDS(B), B.lds_bytes)

// returns a bit mask of the arrays that were placed in LDS (bit = position in the lists above,
// ints first)
MJH_DEV unsigned long long efc_layout(MREF M, BREF B, int e, int nefc, Efc& P) {
  const int nv = M.s.nv, nmax = M.s.nefcmax;
  // AR, Y and sqrtInvD belong to the dual (PGS) solver: the primal ones leave their bytes to J
  const int dual_ = (!MJH_HAS(MJH_FT_PRIMAL) || M.o.solver == MJH_SOL_PGS) ? 1 : 0;
  // sparse primal path: the dense J stays in its global home (it is only the staging copy the compressed
  // rows are cut from); the arrays sized by nJ come last in the packing order, so the layout of everything
  // else is already final while nJ is still being counted
  // (sp_: a compressed Jacobian exists -- the 128-bit-mask form of mjh_sparse.h (spm_) or the explicit-index form of mjh_csr.h (csr_))
  const int spm_ = (MJH_HAS(MJH_FT_PRIMAL) && M.s.sparse) ? 1 : 0;
  const int csr_ = (MJH_HAS(MJH_FT_PRIMAL) && M.s.csr) ? 1 : 0;
  const int sp_ = spm_ | csr_;
  // (explicit-index rows take LDS only in launches whose workgroups own most of a CU's block: with less, the solver's
  // ordered sums need the room as staging space)
  const int csrl_ = (csr_ && B.lds_bytes >= 128*1024) ? 1 : 0;
  const int primal_ = (MJH_HAS(MJH_FT_PRIMAL) && M.o.solver != MJH_SOL_PGS) ? 1 : 0;
  const int nJ_ = sp_ ? MJH_F(B, counts, e)[MJH_C_NJ] : 0;
  int off1 = B.dyn_off, off2 = B.dyn2_off;
  const int end1 = B.lds_bytes, end2 = B.dyn_off;
  unsigned long long mask = 0, bit = 1;
  char* const lds_ = MJH_LDS(B);
  // Primal solvers: what the solver's serial chains touch is placed first -- row addresses / patterns of the sparse
  // factor, the dof vectors, the factor itself (sparse: compressed by pattern, takes all that is left up to its full
  // size; dense: packed lower triangle).  These arrays are born in the solver, after the fields of region 2 have
  // died, and the dual-only arrays that would use region 2 do not exist here: the block grows upward from the bottom
  // of region 2 across the boundary into region 1, and the arrays written by constraint assembly get what is left
  // above it.  They are streamed lane-parallel and can live in global memory.
  P.spL_home = spm_ ? MJH_G(B, sp_L, e) : MJH_G(B, nt_H, e);
  P.spL = P.spL_home; P.spL_cap = spm_ ? M.s.nLp : nv*nv;
  P.spLc = spm_ ? MJH_G(B, sp_Lc, e) : MJH_G(B, nt_M, e);
  P.Ladr = MJH_G(B, sp_Ladr, e); P.Lmask = MJH_G(B, sp_Lmask, e);
  P.vec = MJH_G(B, nt_vec, e);
  P.spar = MJH_G(B, iscratch, e) + nmax;
  P.ev = nullptr;
  if (primal_ && B.lds_bytes) {
    int lo = off2;
    const int newton = M.o.solver == MJH_SOL_NEWTON;
    // (Newton on the explicit-index rows -- mjh_newtonx.h -- shares the CG layout of eight dof vectors; its factor is a
    // packed triangle in global memory)
    const int nvec = ((newton && !csr_) ? 5 : 8)*nv*(int)sizeof(real);
    if (spm_) {
      const int b_adr = (((nv + 1)*(int)sizeof(int)) + 7) & ~7, b_mask = 4*nv*(int)sizeof(int);
      if (lo + b_adr <= end1) { P.Ladr = SP<int>{(int*)(lds_ + lo), 1}; lo += b_adr; }
      if (lo + b_mask <= end1) { P.Lmask = SP<int>{(int*)(lds_ + lo), 1}; lo += b_mask; }
      if (lo + b_adr <= end1) { P.spar = SP<int>{(int*)(lds_ + lo), 1}; lo += b_adr; }
    }
    if (lo + nvec <= end1) { P.vec = SP<real>{(real*)(lds_ + lo), 1}; lo += nvec; }
    if (newton && !csr_) {
      // (sparse Newton: 3 KB ahead of the factor for the line search's row-ordered sums, when the factor keeps a useful share)
      if (spm_ && end1 - lo >= 6*MJH_WAVE*(int)sizeof(real) + 512*(int)sizeof(real)) { P.ev = (real*)(lds_ + lo); lo += 6*MJH_WAVE*(int)sizeof(real); }
      const int full = (spm_ ? M.s.nLp : nv*(nv + 1)/2)*(int)sizeof(real);
      int lb = (end1 - lo) & ~7;
      if (lb > full) lb = full;
      // (the dense factor is not addressed through a capacity: all or nothing)
      if (spm_ ? lb >= 64*(int)sizeof(real) : lb == full) { P.spL = SP<real>{(real*)(lds_ + lo), 1}; P.spL_cap = lb/(int)sizeof(real); lo += lb; }
      if (!spm_ && M.o.cone != 0 && lo + full <= end1) { P.spLc = SP<real>{(real*)(lds_ + lo), 1}; lo += full; }
    }
    off2 = end2;
    if (lo > off1) off1 = lo;
  }
  // a region-2 array that does not fit its region falls through to what is left of region 1
#define MJH_EFC_PLACE(T, m, home, bytes, region)                                              \
    if ((bytes) <= 0) P.m = (home);                                                           \
    else if ((region) == 2 && off2 + (bytes) <= end2) { P.m = SP<T>{(T*)(lds_ + off2), 1}; off2 += ((bytes) + 7) & ~7; mask |= bit; } \
    else if (off1 + (bytes) <= end1) { P.m = SP<T>{(T*)(lds_ + off1), 1}; off1 += ((bytes) + 7) & ~7; mask |= bit; }            \
    else P.m = (home);                                                                        \
    bit <<= 1;
#define X(m, home, cnt, region) { const int bytes_ = (int)sizeof(int)*(cnt); MJH_EFC_PLACE(int, m, home, bytes_, region) }
  MJH_EFC_INT_ARRAYS(X)
#undef X
#define X(m, home, cnt, region) { const int bytes_ = (int)sizeof(real)*(cnt); MJH_EFC_PLACE(real, m, home, bytes_, region) }
  MJH_EFC_REAL_ARRAYS(X)
#undef X
#define X(m, home, cnt, region) { const int bytes_ = (int)sizeof(int)*(cnt); MJH_EFC_PLACE(int, m, home, bytes_, region) }
  MJH_EFC_LATE_INT_ARRAYS(X)
#undef X
#undef MJH_EFC_PLACE
  {
    const int f1 = end1 - off1, f2 = end2 - off2;
    P.free_p = lds_ + (f2 >= f1 ? off2 : off1);
    P.free_bytes = B.lds_bytes ? (f2 >= f1 ? f2 : f1) : 0;
  }
  return mask;
}

// ---- sparse constraint path (mjh_sparse.h)
// mju_dotSparse(row r of J, v): four accumulators over the stored entries in groups of four, then the rest one by one
template <class P0>
MJH_DEV real sp_row_dot(const Efc& P, int r, P0 v) {
  M128 pm = m128_ld(P.rowmask + 4*r);
  crptr a = P.spJ + P.rowadr[r];
  const int nnz = m128_count(pm);
  real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  int k = 0;
  for (; k <= nnz - 4; k += 4) {
    const int j0 = m128_lowest(pm); pm = m128_drop_lowest(pm);
    const int j1 = m128_lowest(pm); pm = m128_drop_lowest(pm);
    const int j2 = m128_lowest(pm); pm = m128_drop_lowest(pm);
    const int j3 = m128_lowest(pm); pm = m128_drop_lowest(pm);
    r0 += a[k]*v[j0]; r1 += a[k + 1]*v[j1]; r2 += a[k + 2]*v[j2]; r3 += a[k + 3]*v[j3];
  }
  real res = (r0 + r2) + (r1 + r3);
  for (; k < nnz; k++) { const int j = m128_lowest(pm); pm = m128_drop_lowest(pm); res += a[k]*v[j]; }
  return res;
}

// the same for the explicit-index form (mjh_csr.h): row r = spJ / colind [rowadr[r], rowadr[r+1])
template <class P0>
MJH_DEV real csr_row_dot(const Efc& P, int r, P0 v) {
  const int a0 = P.rowadr[r], nnz = P.rowadr[r + 1] - a0;
  crptr a = P.spJ + a0;
  ciptr ci = P.colind + a0;
  real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  int k = 0;
  for (; k <= nnz - 4; k += 4) { r0 += a[k]*v[ci[k]]; r1 += a[k + 1]*v[ci[k + 1]]; r2 += a[k + 2]*v[ci[k + 2]]; r3 += a[k + 3]*v[ci[k + 3]]; }
  real res = (r0 + r2) + (r1 + r3);
  for (; k < nnz; k++) res += a[k]*v[ci[k]];
  return res;
}

// debug write-back of the LDS-resident constraint arrays to their global homes
MJH_DEVN void efc_writeback(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const int nefc = MJH_F(B, counts, e)[MJH_C_NEFC];
  const int nv = M.s.nv, nmax = M.s.nefcmax;
  const int dual_ = (!MJH_HAS(MJH_FT_PRIMAL) || M.o.solver == MJH_SOL_PGS) ? 1 : 0;
  const int spm_ = (MJH_HAS(MJH_FT_PRIMAL) && M.s.sparse) ? 1 : 0;
  const int csr_ = (MJH_HAS(MJH_FT_PRIMAL) && M.s.csr) ? 1 : 0;
  const int sp_ = spm_ | csr_;
  const int csrl_ = (csr_ && B.lds_bytes >= 128*1024) ? 1 : 0;
  const int primal_ = (MJH_HAS(MJH_FT_PRIMAL) && M.o.solver != MJH_SOL_PGS) ? 1 : 0;
  const int nJ_ = sp_ ? MJH_F(B, counts, e)[MJH_C_NJ] : 0;
  (void)csrl_; (void)nv; (void)nmax; (void)dual_; (void)sp_; (void)nJ_; (void)primal_;
  if (!nefc) return;
  Efc P;
  const unsigned long long mask = efc_layout(M, B, e, nefc, P);
  unsigned long long bit = 1;
#define X(m, home, cnt, region) { iptr g = (home); if (mask & bit) MJH_FOR_LANES(i, (cnt)) g[i] = P.m[i]; bit <<= 1; }
  MJH_EFC_INT_ARRAYS(X)
#undef X
#define X(m, home, cnt, region) { rptr g = (home); if (mask & bit) MJH_FOR_LANES(i, (cnt)) g[i] = P.m[i]; bit <<= 1; }
  MJH_EFC_REAL_ARRAYS(X)
#undef X
#define X(m, home, cnt, region) { iptr g = (home); if (mask & bit) MJH_FOR_LANES(i, (cnt)) g[i] = P.m[i]; bit <<= 1; }
  MJH_EFC_LATE_INT_ARRAYS(X)
#undef X
}

// impedance curve                                  (getimpedance, engine_core_constraint.c:2099)
MJH_DEV real imp_power(real a, real b) {
  if (b == 1) return a;
  if (b == 2) return a*a;
  return pow(a, b);
}
template <class P0, class P1, class P2>
MJH_DEV void get_impedance(P0 solimp, real pos, real margin, P1 imp, P2 impP) {
  if (solimp[0] == solimp[1] || solimp[2] <= MJH_MINVAL) {
    *imp = 0.5*(solimp[0] + solimp[1]);
    *impP = 0;
    return;
  }
  real x = (pos - margin) / solimp[2];
  real sgn = 1;
  if (x < 0) { x = -x; sgn = -1; }
  if (x >= 1 || x <= 0) {
    *imp = (x >= 1 ? solimp[1] : solimp[0]);
    *impP = 0;
    return;
  }
  real y, yP;
  if (solimp[4] == 1) {
    y = x; yP = 1;
  } else if (x <= solimp[3]) {
    real a = 1/imp_power(solimp[3], solimp[4]-1);
    y = a*imp_power(x, solimp[4]);
    yP = solimp[4] * a*imp_power(x, solimp[4]-1);
  } else {
    real b = 1/imp_power(1-solimp[3], solimp[4]-1);
    y = 1-b*imp_power(1-x, solimp[4]);
    yP = solimp[4] * b*imp_power(1-x, solimp[4]-1);
  }
  *imp = solimp[0] + y*(solimp[1]-solimp[0]);
  *impP = yP * sgn * (solimp[1]-solimp[0]) / solimp[2];
}

// sanitise solref/solimp                           (getsolparam tail, engine_core_constraint.c:2019-2047)
// (mixed-sign solref is rejected at model upload, so only the clamps remain)
template <class P0, class P1>
MJH_DEV void fix_solparam(MREF M, P0 solref, P1 solimp) {
  if (!(M.o.disableflags & (1<<12)) && solref[0] > 0) solref[0] = r_max(solref[0], 2*M.o.timestep);
  solimp[0] = r_min(0.9999, r_max(0.0001, solimp[0]));
  solimp[1] = r_min(0.9999, r_max(0.0001, solimp[1]));
  solimp[2] = r_max(0, solimp[2]);
  solimp[3] = r_min(0.9999, r_max(0.0001, solimp[3]));
  solimp[4] = r_max(1, solimp[4]);
}

// K, B, I, P of one row                            (mj_makeImpedance body, engine_core_constraint.c:2170-2207)
template <class P0, class P1, class P2>
MJH_DEV void set_kbip(P0 KBIP, P1 ref, P2 solimp, real imp, real impP, int friction_row) {
  if (friction_row) {
    KBIP[0] = 0;
  } else if (ref[0] > 0) {
    KBIP[0] = 1 / r_max(MJH_MINVAL, solimp[1]*solimp[1] * ref[0]*ref[0] * ref[1]*ref[1]);
  } else {
    KBIP[0] = -ref[0] / r_max(MJH_MINVAL, solimp[1]*solimp[1]);
  }
  if (ref[1] > 0) {
    KBIP[1] = 2 / r_max(MJH_MINVAL, solimp[1]*ref[0]);
  } else {
    KBIP[1] = -ref[1] / r_max(MJH_MINVAL, solimp[1]);
  }
  KBIP[2] = imp;
  KBIP[3] = impP;
}

// ------------------------------------------------------------------------------------------------
// mj_makeConstraint                                (engine_core_constraint.c:2824-2914)
// row order: friction(dof, tendon) | limits(joint lower/upper, tendon lower/upper) | contacts
// Phase 1: every candidate (one per lane) decides how many rows it emits; an ordered prefix sum
//          gives each its first row.  Phase 2: owners write the scalar row data and the sparse
//          non-contact Jacobian rows.  Phase 3: contact Jacobians, lanes over dof columns.
// Phase 4: diagApprox + impedance (R, D, KBIP), lanes over constraint blocks.
// ------------------------------------------------------------------------------------------------
// quaternion helpers of the weld constraint (engine_util_spatial.c:81-92, :96-101, :225-230)
template <class P0, class P1, class P2>
MJH_DEV void q_mulaxis(P0 res, P1 quat, P2 axis) {
  real t0 = -quat[1]*axis[0] - quat[2]*axis[1] - quat[3]*axis[2];
  real t1 = quat[0]*axis[0] + quat[2]*axis[2] - quat[3]*axis[1];
  real t2 = quat[0]*axis[1] + quat[3]*axis[0] - quat[1]*axis[2];
  real t3 = quat[0]*axis[2] + quat[1]*axis[1] - quat[2]*axis[0];
  res[0] = t0; res[1] = t1; res[2] = t2; res[3] = t3;
}
template <class P0, class P1>
MJH_DEV void q_neg(P0 res, P1 q) { res[0] = q[0]; res[1] = -q[1]; res[2] = -q[2]; res[3] = -q[3]; }
template <class P0, class P1, class P2>
MJH_DEV void q_deriv(P0 res, P1 quat, P2 vel) {
  res[0] = 0.5*(-vel[0]*quat[1] - vel[1]*quat[2] - vel[2]*quat[3]);
  res[1] = 0.5*( vel[0]*quat[0] + vel[1]*quat[3] - vel[2]*quat[2]);
  res[2] = 0.5*(-vel[0]*quat[3] + vel[1]*quat[0] + vel[2]*quat[1]);
  res[3] = 0.5*( vel[0]*quat[2] - vel[1]*quat[1] + vel[2]*quat[0]);
}

// anchors of a connect/weld equality in global coordinates   (mj_equalityAnchors, :561-590)
MJH_DEV void equality_anchors(MREF M, BREF B, int e, int q, real* pos0, real* pos1, int* body0, int* body1) {
  const int o1 = M.eq_obj1id[q], o2 = M.eq_obj2id[q];
  if (!M.eq_objsite[q]) {
    crptr xpos = MJH_F(B, xpos, e);
    crptr xmat = MJH_F(B, xmat, e);
    auto data = M.eq_data + 11*q;
    // connect: anchors data[0:3], data[3:6]; weld: data[3:6] on body 1, data[0:3] on body 2
    const int a0 = (M.eq_type[q] == MJH_EQ_CONNECT) ? 0 : 3, a1 = 3 - a0;
    m3_mulvec(pos0, xmat + 9*o1, data + a0);
    v3_addto(pos0, xpos + 3*o1);
    m3_mulvec(pos1, xmat + 9*o2, data + a1);
    v3_addto(pos1, xpos + 3*o2);
    *body0 = o1; *body1 = o2;
  } else {
    crptr site_xpos = MJH_F(B, site_xpos, e);
    v3_copy(pos0, site_xpos + 3*o1);
    v3_copy(pos1, site_xpos + 3*o2);
    *body0 = M.site_bodyid[o1]; *body1 = M.site_bodyid[o2];
  }
}

// the two orientation quaternions of a weld: quat = q0*relpose, quat1 = neg(q1)   (:668-686)
MJH_DEV void weld_quats(MREF M, BREF B, int e, int q, int body0, int body1, real* quat, real* quat1) {
  crptr xquat = MJH_F(B, xquat, e);
  if (!M.eq_objsite[q]) {
    q_mul(quat, xquat + 4*M.eq_obj1id[q], M.eq_data + 11*q + 6);
    q_neg(quat1, xquat + 4*M.eq_obj2id[q]);
  } else {
    real qs1[4];
    q_mul(quat, xquat + 4*body0, M.site_quat + 4*M.eq_obj1id[q]);
    q_mul(qs1, xquat + 4*body1, M.site_quat + 4*M.eq_obj2id[q]);
    q_neg(quat1, qs1);
  }
}

// rows of the active equality constraints: efc_pos and the dense Jacobian   (mj_instantiateEquality)
MJH_DEV void stage_equality_rows(MREF M, BREF B, int e, const Efc& P) {
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  crptr qpos = MJH_F(B, qpos, e);
  crptr cdof = MJH_F(B, cdof, e);
  crptr subtree_com = MJH_F(B, subtree_com, e);
  ciptr efcadr = MJH_G(B, eq_efcadr, e);
  rptr J = P.J;
  // (do the dense rows exist?  everywhere but under a primal solver on one of the compressed-row paths)
  const int densej = !(s.sparse || s.csr) || M.o.solver == MJH_SOL_PGS;
  for (int q = 0; q < s.neq; q++) {
    const int r0 = efcadr[q];
    if (r0 < 0) continue;
    const int et = M.eq_type[q];
    auto data = M.eq_data + 11*q;
    if (MJH_HAS(MJH_FT_FLEX) && et == MJH_EQ_FLEX) {
      // edge constraints (mj_instantiateEquality, engine_core_constraint.c:982-1010): one row per non-rigid edge,
      // position error = length - rest length; the row itself is the edge's flexedge_J row (cut by stage_csr_rows)
      crptr len = MJH_F(B, flexedge_length, e);
      const int k0 = M.eq_rowadr[q], nk = M.eq_rowadr[q + 1] - k0;
      MJH_FOR_LANES(k, nk) { const int ed = M.eqrow_edge[k0 + k]; P.pos[r0 + k] = len[ed] - M.flexedge_length0[ed]; }
      if (densej) {
        // (dense rows -- the dual solver, or a model below the reference's sparse threshold: the edge's row scattered into a
        // cleared row, mj_instantiateEquality :998-1006)
        crptr fJ = MJH_F(B, flexedge_J, e);
        for (int k = 0; k < nk; k++) MJH_FOR_LANES(j, nv) J[(size_t)(r0 + k)*nv + j] = 0;
        wv_sync();
        MJH_FOR_LANES(k, nk) {
          const int ed = M.eqrow_edge[k0 + k];
          const int a = M.flexedge_J_rowadr[ed], n = M.flexedge_J_rownnz[ed];
          for (int c = 0; c < n; c++) J[(size_t)(r0 + k)*nv + M.flexedge_J_colind[a + c]] = fJ[a + c];
        }
      }
      continue;
    }
    if (MJH_HAS(MJH_FT_FLEX) && et == MJH_EQ_FLEXVERT) {
      // vertex constraints (:1013-1038): two rows per vertex, position error = the strain invariants mj_flex left in
      // flexvert_length; the rows are the vertex's flexvert_J rows (cut by stage_csr_rows)
      crptr vl = MJH_G(B, flexvert_length, e);
      const int k0 = M.eq_rowadr[q], nk = M.eq_rowadr[q + 1] - k0;
      MJH_FOR_LANES(k, nk) P.pos[r0 + k] = vl[M.eqrow_edge[k0 + k]];
      continue;
    }
    if (et == MJH_EQ_CONNECT || et == MJH_EQ_WELD) {
      real pos0[3], pos1[3], cpos[6] = {0, 0, 0, 0, 0, 0};
      int b0, b1;
      equality_anchors(M, B, e, q, pos0, pos1, &b0, &b1);
      v3_sub(cpos, pos0, pos1);
      real quat[4] = {1, 0, 0, 0}, quat1[4] = {1, 0, 0, 0};
      const real torquescale = data[10];
      if (et == MJH_EQ_WELD) {
        real quat2[4];
        weld_quats(M, B, e, q, b0, b1, quat, quat1);
        q_mul(quat2, quat1, quat);
        v3_scl(cpos + 3, quat2 + 1, torquescale);
      }
      const int w0 = M.body_weldid[b0], w1 = M.body_weldid[b1];
      real off0[3], off1[3];
      v3_sub(off0, pos0, subtree_com + 3*M.body_rootid[b0]);
      v3_sub(off1, pos1, subtree_com + 3*M.body_rootid[b1]);
      MJH_FOR_LANES(j, nv) {
        const int in0 = (M.body_dofanc[w0*s.nvw + (j >> 5)] >> (j & 31)) & 1;
        const int in1 = (M.body_dofanc[w1*s.nvw + (j >> 5)] >> (j & 31)) & 1;
        crptr cd = cdof + 6*j;
        real p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0}, r0v[3] = {0, 0, 0}, r1v[3] = {0, 0, 0};
        if (in0) {
          real t[3];
          v3_cross(t, cd, off0);
          p0[0] = cd[3] + t[0]; p0[1] = cd[4] + t[1]; p0[2] = cd[5] + t[2];
          r0v[0] = cd[0]; r0v[1] = cd[1]; r0v[2] = cd[2];
        }
        if (in1) {
          real t[3];
          v3_cross(t, cd, off1);
          p1[0] = cd[3] + t[0]; p1[1] = cd[4] + t[1]; p1[2] = cd[5] + t[2];
          r1v[0] = cd[0]; r1v[1] = cd[1]; r1v[2] = cd[2];
        }
        // difference (opposite of contact: 0 - 1)
        for (int k = 0; k < 3; k++) J[(size_t)(r0 + k)*nv + j] = p0[k] - p1[k];
        if (et == MJH_EQ_WELD) {
          // 0.5 * neg(q1) * (jac0-jac1) * q0 * relpose, then torquescale
          real axis[3] = {r0v[0] - r1v[0], r0v[1] - r1v[1], r0v[2] - r1v[2]};
          real quat2[4], quat3[4];
          q_mulaxis(quat2, quat1, axis);
          q_mul(quat3, quat2, quat);
          for (int k = 0; k < 3; k++) J[(size_t)(r0 + 3 + k)*nv + j] = (0.5*quat3[1 + k]) * torquescale;
        }
      }
      const int size = (et == MJH_EQ_WELD) ? 6 : 3;
      if (wv_lane() == 0) for (int k = 0; k < size; k++) P.pos[r0 + k] = cpos[k];
    } else {
      // joint / tendon coupling: pos0 - ref0 - data0 - poly(pos1 - ref1)        (:726-806)
      const int o1 = M.eq_obj1id[q], o2 = M.eq_obj2id[q];
      real p[2] = {0, 0}, ref[2] = {0, 0};
      crptr ten_length = MJH_F(B, ten_length, e);
      crptr ten_J = MJH_F(B, ten_J, e);
      for (int k = 0; k < 1 + (o2 >= 0); k++) {
        const int id = k ? o2 : o1;
        if (et == MJH_EQ_JOINT) { p[k] = qpos[M.jnt_qposadr[id]]; ref[k] = M.qpos0[M.jnt_qposadr[id]]; }
        else { p[k] = ten_length[id]; ref[k] = M.tendon_length0[id]; }
      }
      real cp, deriv = 0;
      if (o2 >= 0) {
        const real dif = p[1] - ref[1];
        cp = p[0] - ref[0] - data[0] - (data[1]*dif + data[2]*dif*dif + data[3]*dif*dif*dif + data[4]*dif*dif*dif*dif);
        deriv = data[1] + 2*data[2]*dif + 3*data[3]*dif*dif + 4*data[4]*dif*dif*dif;
      } else {
        cp = p[0] - ref[0] - data[0];
      }
      MJH_FOR_LANES(j, nv) {
        real j0 = 0, j1 = 0;
        if (et == MJH_EQ_JOINT) {
          j0 = (j == M.jnt_dofadr[o1]) ? 1 : 0;
          if (o2 >= 0) j1 = (j == M.jnt_dofadr[o2]) ? 1 : 0;
        } else {
          for (int a = 0; a < M.ten_J_rownnz[o1]; a++)
            if (M.ten_J_colind[M.ten_J_rowadr[o1] + a] == j) j0 = ten_J[M.ten_J_rowadr[o1] + a];
          if (o2 >= 0)
            for (int a = 0; a < M.ten_J_rownnz[o2]; a++)
              if (M.ten_J_colind[M.ten_J_rowadr[o2] + a] == j) j1 = ten_J[M.ten_J_rowadr[o2] + a];
        }
        // dense: jac0 += jac1 * (-deriv)
        J[(size_t)r0*nv + j] = (o2 >= 0) ? j0 + j1*(-deriv) : j0;
      }
      if (wv_lane() == 0) P.pos[r0] = cp;
    }
  }
  wv_sync();
}

// one candidate's classification: how many rows it emits and their scalar data
struct Cand { int nrow, type, id, side; real dist, margin, floss; };

MJH_DEVN void stage_make_constraint(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  iptr counts = MJH_F(B, counts, e);
  iptr warn = MJH_F(B, warning, e);
  const int dsbl = M.o.disableflags;
  crptr qpos = MJH_F(B, qpos, e);
  crptr ten_length = MJH_F(B, ten_length, e);
  crptr ten_J = MJH_F(B, ten_J, e);
  const int ncon = counts[MJH_C_NCON];

  if (dsbl & (1<<0)) {
    if (wv_lane() == 0) { counts[MJH_C_NEFC] = 0; counts[MJH_C_NE] = 0; counts[MJH_C_NF] = 0; counts[MJH_C_NL] = 0; }
    wv_sync();
    return;
  }

  // candidate index space: [equalities | dof friction | tendon friction | joints (2 sides) | tendons (2 sides) | contacts]
  const int c_df = (!MJH_HAS(MJH_FT_EQUALITY) || (dsbl & (1<<1))) ? 0 : s.neq;          // mjDSBL_EQUALITY
  // (a model without friction-loss dofs / limited joints -- a flex has thousands of dofs and joints and neither -- leaves
  // those ranges out: the scans below cost a round of table reads per 64 candidates)
  const int c_tf = c_df + (s.ndoffric ? nv : 0);
  const int c_jl = c_tf + s.ntendon;
  const int c_tl = c_jl + (s.njntlim ? 2*s.njnt : 0);
  const int c_con = c_tl + 2*s.ntendon;
  const int ncand = c_con + ncon;
  const int ispyramid = !MJH_HAS(MJH_FT_ELLIPTIC) || (M.o.cone == 0);

  auto classify = [&](int c, Cand& k) {
    k.nrow = 0; k.type = 0; k.id = 0; k.side = 0; k.dist = 0; k.margin = 0; k.floss = 0;
    if (c >= ncand) return;
    if (c < c_df) {                       // equality (mj_instantiateEquality :596)
      if (MJH_G(B, eq_active, e)[c]) {
        k.nrow = M.eq_rowadr[c + 1] - M.eq_rowadr[c]; k.type = MJH_CNSTR_EQUALITY; k.id = c;
      }
    } else if (c < c_tf) {                // dof friction loss (mj_instantiateFriction :1270)
      const int dof = c - c_df;
      if (!(dsbl & (1<<2)) && M.dof_frictionloss[dof] != 0) {
        k.nrow = 1; k.type = MJH_CNSTR_FRICTION_DOF; k.id = dof; k.floss = M.dof_frictionloss[dof];
      }
    } else if (c < c_jl) {                // tendon friction loss
      int t = c - c_tf;
      if (!(dsbl & (1<<2)) && M.tendon_frictionloss[t] > 0) {
        // empty-row guard of mj_addConstraint (:431-447): skip all-zero Jacobian rows
        int nz = 0;
        for (int q = 0; q < M.ten_J_rownnz[t]; q++) if (ten_J[M.ten_J_rowadr[t] + q] != 0) nz = 1;
        if (nz) { k.nrow = 1; k.type = MJH_CNSTR_FRICTION_TENDON; k.id = t; k.floss = M.tendon_frictionloss[t]; }
      }
    } else if (c < c_tl) {                // joint limits (mj_instantiateLimit :1360)
      int j = (c - c_jl) >> 1;
      k.side = ((c - c_jl) & 1) ? 1 : -1;
      if (!(dsbl & (1<<3)) && M.jnt_limited[j]) {
        int jt = M.jnt_type[j];
        k.margin = M.jnt_margin[j];
        if (jt == MJH_JNT_SLIDE || jt == MJH_JNT_HINGE) {
          real value = qpos[M.jnt_qposadr[j]];
          k.dist = k.side * (M.jnt_range[2*j + (k.side+1)/2] - value);
          if (k.dist < k.margin) { k.nrow = 1; k.type = MJH_CNSTR_LIMIT_JOINT; k.id = j; }
        } else if (jt == MJH_JNT_BALL && k.side == -1) {
          int adr = M.jnt_qposadr[j];
          real quat[4] = {qpos[adr], qpos[adr+1], qpos[adr+2], qpos[adr+3]};
          real aa[3];
          q_normalize(quat);
          q_tovel(aa, quat, 1);
          real value = v3_normalize(aa);
          k.dist = r_max(M.jnt_range[2*j], M.jnt_range[2*j+1]) - value;
          if (k.dist < k.margin) { k.nrow = 1; k.type = MJH_CNSTR_LIMIT_JOINT; k.id = j; k.side = 0; }
        }
      }
    } else if (c < c_con) {               // tendon limits
      int t = (c - c_tl) >> 1;
      k.side = ((c - c_tl) & 1) ? 1 : -1;
      if (!(dsbl & (1<<3)) && M.tendon_limited[t]) {
        k.margin = M.tendon_margin[t];
        k.dist = k.side * (M.tendon_range[2*t + (k.side+1)/2] - ten_length[t]);
        if (k.dist < k.margin) {
          int nz = 0;
          for (int q = 0; q < M.ten_J_rownnz[t]; q++) if (ten_J[M.ten_J_rowadr[t] + q] != 0) nz = 1;
          if (nz) { k.nrow = 1; k.type = MJH_CNSTR_LIMIT_TENDON; k.id = t; }
        }
      }
    } else {                              // contacts (mj_instantiateContact :1617)
      int kc = c - c_con;
      if (!(dsbl & (1<<4)) && !MJH_CON(B, con_exclude, e, 1, kc)[0]) {
        int dim = MJH_CON(B, con_dim, e, 1, kc)[0];
        k.id = kc;
        k.dist = MJH_CON(B, con_dist, e, 1, kc)[0];
        k.margin = M.pair_includemargin[MJH_CON(B, con_pair, e, 1, kc)[0]];
        if (dim == 1) { k.nrow = 1; k.type = MJH_CNSTR_CONTACT_FRICTIONLESS; }
        else if (ispyramid) { k.nrow = 2*(dim - 1); k.type = MJH_CNSTR_CONTACT_PYRAMIDAL; }
        else { k.nrow = dim; k.type = MJH_CNSTR_CONTACT_ELLIPTIC; }
      }
    }
  };

  // ---- pass 1: count rows (the reference's count_only pass, :2833-2860) ---------------------------
  int nefc = 0, ne_total = 0, nf_total = 0, nl_total = 0;
  for (int c0 = 0; c0 < ncand; c0 += MJH_W) {
    Cand k;
    classify(c0 + wv_lane(), k);
    nefc += wv_sum_i(k.nrow);
    ne_total += wv_sum_i(k.type == MJH_CNSTR_EQUALITY ? k.nrow : 0);
    nf_total += wv_sum_i((k.type == MJH_CNSTR_FRICTION_DOF || k.type == MJH_CNSTR_FRICTION_TENDON) ? k.nrow : 0);
    nl_total += wv_sum_i((k.type == MJH_CNSTR_LIMIT_JOINT || k.type == MJH_CNSTR_LIMIT_TENDON) ? k.nrow : 0);
  }
  const int overflow = nefc > s.nefcmax;
  if (overflow) {
    // arena-full semantics of the reference (arenaAllocEfc :145-152): no constraints this step
    nefc = 0; ne_total = 0; nf_total = 0; nl_total = 0;
    for (int k = wv_lane(); k < ncon; k += MJH_W) MJH_CON(B, con_efcadr, e, 1, k)[0] = -1;
  }
  if (wv_lane() == 0) {
    if (overflow) warn[MJH_WARN_CNSTRFULL]++;
    counts[MJH_C_NEFC] = nefc;
    counts[MJH_C_NE] = ne_total;
    counts[MJH_C_NF] = nf_total;
    counts[MJH_C_NL] = nl_total;
  }
  wv_sync();
  if (nefc == 0) return;

  Efc P;
  efc_layout(M, B, e, nefc, P);
  rptr J = P.J;

  // ---- pass 2: scalar row data + sparse (non-contact) Jacobian rows -------------------------------
  int row_base = 0;     // rows emitted by earlier chunks (wave-uniform)
  for (int c0 = 0; c0 < ncand; c0 += MJH_W) {
    const int c = c0 + wv_lane();
    Cand k;
    classify(c, k);
    const int r0 = row_base + wv_exscan_i(k.nrow);
    row_base += wv_sum_i(k.nrow);
    for (int a = 0; a < k.nrow; a++) {
      int r = r0 + a;
      P.type[r] = k.type;
      P.id[r] = k.id;
      // elliptic cone: only the normal row carries the distance (:1696-1700)
      const int tangent = (k.type == MJH_CNSTR_CONTACT_ELLIPTIC && a > 0);
      P.pos[r] = tangent ? (real)0 : k.dist;
      P.margin[r] = tangent ? (real)0 : k.margin;
      P.floss[r] = k.floss;
      if (!ispyramid) P.cone[r] = 0;      // set for elliptic blocks with the impedance below
      if (k.type != MJH_CNSTR_EQUALITY && k.type < MJH_CNSTR_CONTACT_FRICTIONLESS) {
        rptr Jr = J + (size_t)r*nv;
        for (int q = 0; q < nv; q++) Jr[q] = 0;
        if (k.type == MJH_CNSTR_FRICTION_DOF) {
          Jr[k.id] = 1;
        } else if (k.type == MJH_CNSTR_LIMIT_JOINT) {
          if (k.side == 0) {
            // ball joint: J = -axis
            int adr = M.jnt_qposadr[k.id];
            real quat[4] = {qpos[adr], qpos[adr+1], qpos[adr+2], qpos[adr+3]};
            real aa[3];
            q_normalize(quat);
            q_tovel(aa, quat, 1);
            v3_normalize(aa);
            int d = M.jnt_dofadr[k.id];
            Jr[d] = aa[0]*-1; Jr[d+1] = aa[1]*-1; Jr[d+2] = aa[2]*-1;
          } else {
            Jr[M.jnt_dofadr[k.id]] = -(real)k.side;
          }
        } else {
          // tendon friction / limit: +-ten_J scattered to dense
          int a0 = M.ten_J_rowadr[k.id];
          for (int q = 0; q < M.ten_J_rownnz[k.id]; q++) {
            real v = ten_J[a0 + q];
            if (k.type == MJH_CNSTR_LIMIT_TENDON) v = v * (real)(-k.side);
            Jr[M.ten_J_colind[a0 + q]] = v;
          }
        }
      }
    }
    if (k.type >= MJH_CNSTR_CONTACT_FRICTIONLESS) MJH_CON(B, con_efcadr, e, 1, k.id)[0] = k.nrow ? r0 : -1;
    if (c < c_df) MJH_G(B, eq_efcadr, e)[c] = k.nrow ? r0 : -1;
  }
  wv_sync();

  // ---- equality rows: residual and Jacobian per active equality, lanes over dof columns ------------
  if (MJH_HAS(MJH_FT_EQUALITY) && ne_total) stage_equality_rows(M, B, e, P);

  // ---- contact Jacobians: per contact, lanes over dof columns ------------------------------------
  crptr cdof = MJH_F(B, cdof, e);
  crptr subtree_com = MJH_F(B, subtree_com, e);
  // (nv <= 32: the two halves of the wavefront take a contact each)
#if !MJH_LANE_MODE && MJH_W == 64
  const int cpair = nv <= 32;
#else
  const int cpair = 0;
#endif
  // (sparse path under a primal solver: the contact rows are written straight into the compressed Jacobian by
  // stage_sparsify, one row per lane -- this loop takes the contacts one at a time and would leave most lanes idle behind
  // its loads; the dual solver still needs the dense rows for Y = J L^-T D^-1/2)
  const int contact_rows_here = !(MJH_HAS(MJH_FT_PRIMAL) && (s.sparse || s.csr) && M.o.solver != MJH_SOL_PGS);
  for (int k0 = 0; contact_rows_here && k0 < ncon; k0 += 1 + cpair) {
    const int k = k0 + (cpair ? (wv_lane() >> 5) : 0);
    if (k >= ncon) continue;
    int r0 = MJH_CON(B, con_efcadr, e, 1, k)[0];
    if (r0 < 0) continue;
    int dim = MJH_CON(B, con_dim, e, 1, k)[0];
    ciptr cg = MJH_CON(B, con_geom, e, 2, k);
    // flex contacts: a side is a vertex body, or the weighted corner bodies of an element (mj_contactJacobian
    // :1559-1611: sum of the bodies' point Jacobians times their weights, -1 for a geom's body; contact_sides, mjh_flex.h)
    ConSides S;
    if (MJH_HAS(MJH_FT_FLEX)) contact_sides(M, B, e, k, S);
    const int general = MJH_HAS(MJH_FT_FLEX) && !S.simple;
    int b1 = MJH_HAS(MJH_FT_FLEX) ? S.body[0] : (int)M.geom_bodyid[cg[0]], b2 = MJH_HAS(MJH_FT_FLEX) ? S.body[1] : (int)M.geom_bodyid[cg[1]];
    int w1 = M.body_weldid[b1], w2 = M.body_weldid[b2];
    crptr point = MJH_CON(B, con_pos, e, 3, k);
    crptr fr = MJH_CON(B, con_frame, e, 9, k);
    auto fri = M.pair_friction + 5*MJH_CON(B, con_pair, e, 1, k)[0];
    real off1[3], off2[3];
    v3_sub(off1, point, subtree_com + 3*M.body_rootid[b1]);
    v3_sub(off2, point, subtree_com + 3*M.body_rootid[b2]);
    for (int j = cpair ? (wv_lane() & 31) : wv_lane(); j < nv; j += cpair ? 32 : MJH_W) {
      // translational point Jacobians of both bodies (mj_jac, engine_core_util.c:176)
      int in1 = (M.body_dofanc[w1*s.nvw + (j >> 5)] >> (j & 31)) & 1;
      int in2 = (M.body_dofanc[w2*s.nvw + (j >> 5)] >> (j & 31)) & 1;
      real j1[3] = {0, 0, 0}, j2[3] = {0, 0, 0};
      crptr cd = cdof + 6*j;
      if (in1) {
        real t[3];
        v3_cross(t, cd, off1);
        j1[0] = cd[3] + t[0]; j1[1] = cd[4] + t[1]; j1[2] = cd[5] + t[2];
      }
      if (in2) {
        real t[3];
        v3_cross(t, cd, off2);
        j2[0] = cd[3] + t[0]; j2[1] = cd[4] + t[1]; j2[2] = cd[5] + t[2];
      }
      real jd[3] = {j2[0] - j1[0], j2[1] - j1[1], j2[2] - j1[2]};
      real rdg[3] = {0, 0, 0};
      if (general) contact_jac_col(M, S, cdof, subtree_com, point, j, jd, (MJH_HAS(MJH_FT_CONDIM46) && dim > 3) ? rdg : (real*)nullptr);
      // rotate into the contact frame (mju_mulMatMat with zero-skip, engine_util_blas.c:619)
      int nr = dim > 1 ? 3 : 1;
      real jr[6] = {0, 0, 0, 0, 0, 0};
      for (int a = 0; a < nr; a++) {
        real acc = 0;
        for (int q = 0; q < 3; q++) {
          real t = fr[3*a + q];
          if (t != 0) acc += jd[q]*t;
        }
        jr[a] = acc;
      }
      if (MJH_HAS(MJH_FT_CONDIM46) && dim > 3) {
        // torsional / rolling rows: rotational Jacobian difference in the contact frame (:1663-1665)
        real rd[3] = {(in2 ? cd[0] : (real)0) - (in1 ? cd[0] : (real)0),
                      (in2 ? cd[1] : (real)0) - (in1 ? cd[1] : (real)0),
                      (in2 ? cd[2] : (real)0) - (in1 ? cd[2] : (real)0)};
        if (general) { rd[0] = rdg[0]; rd[1] = rdg[1]; rd[2] = rdg[2]; }
        for (int a = 0; a < dim - 3; a++) {
          real acc = 0;
          for (int q = 0; q < 3; q++) {
            real t = fr[3*a + q];
            if (t != 0) acc += rd[q]*t;
          }
          jr[3 + a] = acc;
        }
      }
      if (dim == 1) {
        J[(size_t)r0*nv + j] = jr[0];
      } else if (ispyramid) {
        for (int a = 1; a < dim; a++) {
          J[(size_t)(r0 + 2*(a-1))*nv + j] = jr[0] + jr[a]*fri[a-1];
          J[(size_t)(r0 + 2*(a-1) + 1)*nv + j] = jr[0] + jr[a]*(-fri[a-1]);
        }
      } else {
        for (int a = 0; a < dim; a++) J[(size_t)(r0 + a)*nv + j] = jr[a];
      }
    }
  }
  wv_sync();

  // ---- diagApprox + impedance: lanes over blocks -------------------------------------------------
  // non-contact rows: one block per row
  const int nnc = ne_total + nf_total + nl_total;
  MJH_FOR_LANES(r, nnc) {
    int type = P.type[r], id = P.id[r];
    real solref[2], solimp[5], dA;
    real imp_pos = P.pos[r];
    if (MJH_HAS(MJH_FT_EQUALITY) && type == MJH_CNSTR_EQUALITY) {
      // mj_diagApprox :1733-1780, getsolparam :1988, getposdim :2070-2078
      const int et = M.eq_type[id];
      const int r0 = MJH_G(B, eq_efcadr, e)[id];
      if (MJH_HAS(MJH_FT_FLEX) && et == MJH_EQ_FLEX) {
        dA = M.flexedge_invweight0[M.eqrow_edge[M.eq_rowadr[id] + (r - r0)]];      // (mj_diagApprox :1779-1790)
      } else if (MJH_HAS(MJH_FT_FLEX) && et == MJH_EQ_FLEXVERT) {
        dA = M.body_invweight0[2*M.flexvert_bodyid[M.eqrow_edge[M.eq_rowadr[id] + (r - r0)] >> 1]];      // (:1794-1806)
      } else if (et == MJH_EQ_CONNECT || et == MJH_EQ_WELD) {
        int b1 = M.eq_obj1id[id], b2 = M.eq_obj2id[id];
        if (M.eq_objsite[id]) { b1 = M.site_bodyid[b1]; b2 = M.site_bodyid[b2]; }
        const int rot = (et == MJH_EQ_WELD && r - r0 > 2) ? 1 : 0;
        dA = M.body_invweight0[2*b1 + rot] + M.body_invweight0[2*b2 + rot];
        imp_pos = sqrt(dot_ref(P.pos + r0, P.pos + r0, et == MJH_EQ_WELD ? 6 : 3));
      } else {
        const int o1 = M.eq_obj1id[id], o2 = M.eq_obj2id[id];
        dA = (et == MJH_EQ_JOINT) ? M.dof_invweight0[M.jnt_dofadr[o1]] : M.tendon_invweight0[o1];
        if (o2 >= 0) dA += (et == MJH_EQ_JOINT) ? M.dof_invweight0[M.jnt_dofadr[o2]] : M.tendon_invweight0[o2];
      }
      solref[0] = M.eq_solref[2*id]; solref[1] = M.eq_solref[2*id+1];
      for (int q = 0; q < 5; q++) solimp[q] = M.eq_solimp[5*id + q];
    } else if (type == MJH_CNSTR_FRICTION_DOF) {
      dA = M.dof_invweight0[id];
      solref[0] = M.dof_solref[2*id]; solref[1] = M.dof_solref[2*id+1];
      for (int q = 0; q < 5; q++) solimp[q] = M.dof_solimp[5*id + q];
    } else if (type == MJH_CNSTR_LIMIT_JOINT) {
      dA = M.dof_invweight0[M.jnt_dofadr[id]];
      solref[0] = M.jnt_solref[2*id]; solref[1] = M.jnt_solref[2*id+1];
      for (int q = 0; q < 5; q++) solimp[q] = M.jnt_solimp[5*id + q];
    } else {
      // tendon limit / tendon friction loss (getsolparam :2002-2010)
      dA = M.tendon_invweight0[id];
      if (type == MJH_CNSTR_FRICTION_TENDON) {
        solref[0] = M.tendon_solref_fri[2*id]; solref[1] = M.tendon_solref_fri[2*id+1];
        for (int q = 0; q < 5; q++) solimp[q] = M.tendon_solimp_fri[5*id + q];
      } else {
        solref[0] = M.tendon_solref_lim[2*id]; solref[1] = M.tendon_solref_lim[2*id+1];
        for (int q = 0; q < 5; q++) solimp[q] = M.tendon_solimp_lim[5*id + q];
      }
    }
    fix_solparam(M, solref, solimp);
    real imp, impP;
    get_impedance(solimp, imp_pos, P.margin[r], &imp, &impP);
    real R = r_max(MJH_MINVAL, (1-imp)*dA/imp);
    int fr_row = (type == MJH_CNSTR_FRICTION_DOF || type == MJH_CNSTR_FRICTION_TENDON);
    set_kbip(P.KBIP + 4*r, solref, solimp, imp, impP, fr_row);
    P.R[r] = R;
    P.D[r] = 1 / R;
    P.diagA[r] = R * imp / (1 - imp);
  }
  // contact blocks
  MJH_FOR_LANES(k, ncon) {
    int r0 = MJH_CON(B, con_efcadr, e, 1, k)[0];
    if (r0 < 0) continue;
    int p = MJH_CON(B, con_pair, e, 1, k)[0];
    int dim = MJH_CON(B, con_dim, e, 1, k)[0];
    int type = P.type[r0];
    ciptr cg = MJH_CON(B, con_geom, e, 2, k);
    // mj_diagApprox, contact case (:1895-1970): the bodies of side 0, then of side 1, with their (unsigned) weights
    real tran = 0, rot = 0;
    if (MJH_HAS(MJH_FT_FLEX)) {
      ConSides S;
      contact_sides(M, B, e, k, S);
      if (S.ext) {
        // (a list of node bodies: mj_diagApprox passes the side's unsigned vertex weights, so the basis values enter as they are)
        for (int q = 0; q < S.n; q++) {
          const real wq = fabs(S.xw[q]);
          tran += M.body_invweight0[2*S.xb[q]] * wq;  rot += M.body_invweight0[2*S.xb[q]+1] * wq;
        }
      } else
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (q >= S.n) continue;
        const real wq = fabs(S.w[q]);
        tran += M.body_invweight0[2*S.body[q]] * wq;  rot += M.body_invweight0[2*S.body[q]+1] * wq;
      }
    } else {
      int b1 = M.geom_bodyid[cg[0]];
      tran += M.body_invweight0[2*b1] * 1;  rot += M.body_invweight0[2*b1+1] * 1;
      int b2 = M.geom_bodyid[cg[1]];
      tran += M.body_invweight0[2*b2] * 1;  rot += M.body_invweight0[2*b2+1] * 1;
    }
    auto fri = M.pair_friction + 5*p;
    real solref[2] = {M.pair_solref[2*p], M.pair_solref[2*p+1]};
    real solimp[5];
    for (int q = 0; q < 5; q++) solimp[q] = M.pair_solimp[5*p + q];
    fix_solparam(M, solref, solimp);
    // solreffriction of a predefined pair (getsolparam :2031-2040: REFSAFE clamp on the time constant)
    real srf[2] = {M.pair_solreffriction[2*p], M.pair_solreffriction[2*p+1]};
    if (!(M.o.disableflags & (1<<12)) && srf[0] > 0) srf[0] = r_max(srf[0], 2*M.o.timestep);
    real imp, impP;
    get_impedance(solimp, P.pos[r0], P.margin[r0], &imp, &impP);
    int nrow = (type == MJH_CNSTR_CONTACT_FRICTIONLESS) ? 1 : (type == MJH_CNSTR_CONTACT_PYRAMIDAL ? 2*(dim-1) : dim);
    for (int a = 0; a < nrow; a++) {
      real dA;
      if (type == MJH_CNSTR_CONTACT_FRICTIONLESS) dA = tran;
      else if (type == MJH_CNSTR_CONTACT_PYRAMIDAL) {
        int jj = a >> 1;
        dA = tran + fri[jj]*fri[jj]*(jj < 2 ? tran : rot);
      } else dA = (a < 3 ? tran : rot);
      P.R[r0 + a] = r_max(MJH_MINVAL, (1-imp)*dA/imp);
      // elliptic friction rows: K = 0, B from solreffriction when a predefined pair sets it (:2181-2189)
      const int fric_row = (type == MJH_CNSTR_CONTACT_ELLIPTIC && a > 0);
      if (fric_row && (srf[0] != 0 || srf[1] != 0)) set_kbip(P.KBIP + 4*(r0 + a), srf, solimp, imp, impP, 1);
      else set_kbip(P.KBIP + 4*(r0 + a), solref, solimp, imp, impP, fric_row);
    }
    if (type == MJH_CNSTR_CONTACT_ELLIPTIC) {
      // (:2213-2237) R[1] = R[0]/impratio, mu = friction[0]*sqrt(R[1]/R[0]), R[j]*mu[j]^2 = R[1]*mu[1]^2
      P.R[r0 + 1] = P.R[r0] / r_max(MJH_MINVAL, M.o.impratio);
      real mu = fri[0] * sqrt(P.R[r0 + 1]/P.R[r0]);
      MJH_CON(B, con_mu, e, 1, k)[0] = mu;
      for (int a = 1; a < dim - 1; a++) P.R[r0 + a + 1] = P.R[r0 + 1]*fri[0]*fri[0]/(fri[a]*fri[a]);
      P.cone[r0] = mu;
      for (int a = 1; a < dim; a++) P.cone[r0 + a] = fri[a - 1];
    }
    if (type == MJH_CNSTR_CONTACT_PYRAMIDAL) {
      // (:2213-2253) R[1] = R[0]/impratio; mu = friction[0]*sqrt(R[1]/R[0]); all rows Rpy = 2 mu^2 R[0]
      real R0 = P.R[r0];
      real R1 = R0 / r_max(MJH_MINVAL, M.o.impratio);
      real mu = fri[0] * sqrt(R1/R0);
      MJH_CON(B, con_mu, e, 1, k)[0] = mu;
      real Rpy = 2*mu*mu*R0;
      for (int a = 0; a < nrow; a++) P.R[r0 + a] = Rpy;
    }
    for (int a = 0; a < nrow; a++) {
      real R = P.R[r0 + a];
      P.D[r0 + a] = 1 / R;
      P.diagA[r0 + a] = R * imp / (1 - imp);
    }
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// mj_island, the part the PGS solver needs          (engine_island.c:363-483, :129-150)
// island of every constraint row = connected component (union-find over the kinematic trees a
// constraint touches), numbered in ascending order of the component's smallest tree.  One tree
// (or islands disabled): every row is in island 0.  The union-find itself is a short serial scan.
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_island(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC];
  if (!nefc) {
    if (wv_lane() == 0) counts[MJH_C_NISLAND] = 0;
    wv_sync();
    return;
  }
  Efc P;
  efc_layout(M, B, e, nefc, P);
  const int nv = s.nv, ntree = s.ntree;
  if (!MJH_HAS(MJH_FT_ISLANDS) || ntree <= 1 || (M.o.disableflags & (1<<18))) {
    MJH_FOR_LANES(i, nefc) P.island[i] = 0;
    if (wv_lane() == 0) counts[MJH_C_NISLAND] = (M.o.disableflags & (1<<18)) ? 0 : 1;
    wv_sync();
    return;
  }
  // The components are what mj_dsuMerge / mj_dsuAssign produce (engine_island.c: union by smaller root, islands
  // numbered by ascending smallest tree); they do not depend on the order of the unions, so the rows are taken
  // lane-parallel: every row names the trees it touches, labels (smallest tree of the component so far) are
  // lowered along those edges with atomic minima and compressed by pointer jumping until nothing moves.
  {
    iptr work = MJH_G(B, island_work, e);
    iptr efc_tree = work;                       // [nefc] first tree of the row
    iptr label = work + s.nefcmax;              // [ntree] -1: no constraint touches the tree
    iptr tree_island = label + ntree;           // [ntree]
    iptr second = MJH_G(B, iscratch, e) + 2*s.nefcmax;     // [nefc] second tree of the row, -2: none, -3: scan the row
    // (every pass below is a handful of dependent reads of these arrays: they work in the unused tail of the LDS regions
    // when it has room; tree_island, which the primal solvers read, is copied to its global home at the end)
    const iptr tree_island_home = tree_island;
    // (only a tail of region 1: region 2 still holds the contact slots this stage reads)
    // (sparse Newton: the factor takes the whole tail, but the line search's block ahead of it is idle until the solve)
    const int need = (2*nefc + 2*ntree)*(int)sizeof(int);
    int* w = nullptr;
    if (P.free_bytes >= need && P.free_p >= MJH_LDS(B) + B.dyn_off) w = (int*)P.free_p;
    else if (P.ev && need <= 6*MJH_WAVE*(int)sizeof(real) && (char*)P.ev >= MJH_LDS(B) + B.dyn_off) w = (int*)P.ev;
    const int in_lds = w != nullptr;
    if (in_lds) {
      efc_tree = SP<int>{w, 1}; second = SP<int>{w + nefc, 1}; label = SP<int>{w + 2*nefc, 1}; tree_island = SP<int>{w + 2*nefc + ntree, 1};
    }
    MJH_FOR_LANES(t, ntree) label[t] = -1;
    wv_sync();
    auto row_trees = [&](int i, int* ta, int* tb) {
      const int type = P.type[i], id = P.id[i];
      int t1 = -2, t2 = -2;
      if (type == MJH_CNSTR_FRICTION_DOF) t1 = M.dof_treeid[id];
      else if (type == MJH_CNSTR_LIMIT_JOINT) t1 = M.dof_treeid[M.jnt_dofadr[id]];
      else if (type >= MJH_CNSTR_CONTACT_FRICTIONLESS) {
        ciptr cg = MJH_CON(B, con_geom, e, 2, id);
        if (cg[1] >= 0) {
          t1 = M.body_treeid[M.geom_bodyid[cg[0]]];
          t2 = M.body_treeid[M.geom_bodyid[cg[1]]];
        } else t2 = -3;          // flex contacts: the generic scan of the row (treeIterInit, engine_island.c:315-318)
      } else if (type == MJH_CNSTR_EQUALITY && (M.eq_type[id] == MJH_EQ_CONNECT || M.eq_type[id] == MJH_EQ_WELD)) {
        int b1 = M.eq_obj1id[id], b2 = M.eq_obj2id[id];
        if (M.eq_objsite[id]) { b1 = M.site_bodyid[b1]; b2 = M.site_bodyid[b2]; }
        t1 = M.body_treeid[b1];
        t2 = M.body_treeid[b2];
      } else t2 = -3;
      *ta = t1; *tb = t2;
    };
    // trees of a row that has to be scanned (tendon rows, joint / tendon couplings): in order of the first non-zero
    // dof of each tree; f(prev, next) is called for consecutive distinct trees, returns the first
    auto scan_row = [&](int i, auto&& f) -> int {
      crptr Jr = P.J + (size_t)i*nv;
      int first = -2, prev = -2;
      if (MJH_HAS(MJH_FT_PRIMAL) && s.csr) {
        // compressed rows (mjh_csr.h): the trees of the stored dofs, in order
        for (int q = P.rowadr[i]; q < P.rowadr[i + 1]; q++) {
          const int tj = M.dof_treeid[P.colind[q]];
          if (tj != prev) { if (first == -2) first = tj; else f(prev, tj); prev = tj; }
        }
        return first;
      }
      for (int j = 0; j < nv; j++) {
        if (Jr[j] != 0) {
          const int tj = M.dof_treeid[j];
          if (tj != prev) { if (first == -2) first = tj; else f(prev, tj); prev = tj; }
          j = M.tree_dofadr[tj] + M.tree_dofnum[tj] - 1;
        }
      }
      return first;
    };
    // compressed rows with explicit column indices (mjh_csr.h): a scanned row's trees are those of its stored dofs, and
    // its edges join the trees of consecutive entries -- taken one lane per stored ENTRY instead of one lane walking a
    // row (a flex contact's row holds a dozen dofs; the labels live in global memory)
    const int csr = MJH_HAS(MJH_FT_PRIMAL) && s.csr;
    const int nJ = csr ? counts[MJH_C_NJ] : 0;
    auto entry_row = [&](int q) -> int {
      int lo = 0, hi = nefc;                               // the row r with rowadr[r] <= q < rowadr[r + 1]
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P.rowadr[mid] <= q) lo = mid; else hi = mid; }
      return lo;
    };
    // (these passes walk index tables -- column -> tree -> label: four items per lane and trip, their loads issued together)
    if (csr) {
      for (int q0 = wv_lane(); q0 < nJ; q0 += 4*MJH_W) {
        int t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int q = q0 + u*MJH_W; t[u] = q < nJ ? (int)M.dof_treeid[P.colind[q]] : -1; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (t[u] >= 0) label[t[u]] = t[u];
      }
    }
    MJH_FOR_LANES(i, nefc) {
      int t1, t2;
      row_trees(i, &t1, &t2);
      if (t2 == -3 && csr) {
        const int a = P.rowadr[i];
        efc_tree[i] = a < P.rowadr[i + 1] ? (int)M.dof_treeid[P.colind[a]] : 0;
      } else if (t2 == -3) {
        t1 = scan_row(i, [&](int a, int b) { label[a] = a; label[b] = b; });
        if (t1 >= 0) label[t1] = t1;
        efc_tree[i] = t1 >= 0 ? t1 : 0;
      } else {
        if (t1 >= 0) label[t1] = t1;
        if (t2 >= 0) label[t2] = t2;
        efc_tree[i] = t1 >= 0 ? t1 : t2;
      }
      second[i] = t2;
    }
    wv_sync();
    // the trees of every stiffness-active flex form one component, with or without rows of their own (unionConstraintTrees,
    // engine_island.c:409-447): start them at the flex's smallest tree
    if (MJH_HAS(MJH_FT_FLEX) && s.nflex) {
      for (int v0 = wv_lane(); v0 < s.nflexvert; v0 += 4*MJH_W) {
        int mt[4], tv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int v = v0 + u*MJH_W;
          mt[u] = -1; tv[u] = -1;
          if (v < s.nflexvert && M.flexvert_bodyid[v] >= 0) { mt[u] = M.flex_mintree[M.flexvert_flex[v]]; tv[u] = M.body_treeid[M.flexvert_bodyid[v]]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (mt[u] >= 0 && tv[u] >= 0) { const int l = label[tv[u]]; if (l < 0 || l > mt[u]) label[tv[u]] = mt[u]; }
      }
      // (interpolated flexes: the node bodies)
      MJH_FOR_LANES(i, s.nflexnode) {
        const int mtn = M.flex_mintree[M.flexnode_flex[i]], tn = M.body_treeid[M.flexnode_bodyid[i]];
        if (mtn >= 0 && tn >= 0) { const int l = label[tn]; if (l < 0 || l > mtn) label[tn] = mtn; }
      }
      wv_sync();
    }
    for (int round = 0; round < 4*ntree + 8; round++) {
      int moved = 0;
      MJH_FOR_LANES(i, nefc) {
        const int t2 = second[i];
        auto join = [&](int a, int b) {
          if (a < 0 || b < 0) return;
          const int la = label[a], lb = label[b];
          if (la == lb) return;
          const int m = la < lb ? la : lb;
          wv_atomic_min_i(&label[a], m);
          wv_atomic_min_i(&label[b], m);
          // (the component's current representatives as well, so that a lowered label reaches every member)
          wv_atomic_min_i(&label[la], m);
          wv_atomic_min_i(&label[lb], m);
          moved = 1;
        };
        if (t2 == -3) { if (!csr) scan_row(i, join); }
        else if (t2 >= 0) join(efc_tree[i], t2);
      }
      if (csr) {
        for (int q0 = wv_lane(); q0 < nJ - 1; q0 += 4*MJH_W) {
          int a[4], b[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int q = q0 + u*MJH_W;
            a[u] = -1; b[u] = -1;
            if (q < nJ - 1) {
              const int r = entry_row(q);
              if (second[r] == -3 && q + 1 < P.rowadr[r + 1]) { a[u] = M.dof_treeid[P.colind[q]]; b[u] = M.dof_treeid[P.colind[q + 1]]; }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (a[u] < 0 || a[u] == b[u]) continue;
            const int la = label[a[u]], lb = label[b[u]];
            if (la == lb) continue;
            const int m = la < lb ? la : lb;
            wv_atomic_min_i(&label[a[u]], m);
            wv_atomic_min_i(&label[b[u]], m);
            wv_atomic_min_i(&label[la], m);
            wv_atomic_min_i(&label[lb], m);
            moved = 1;
          }
        }
      }
      wv_sync();
      MJH_FOR_LANES(t, ntree) {
        int l = label[t];
        if (l >= 0) { while (label[l] != l) l = label[l]; if (label[t] != l) { label[t] = l; moved = 1; } }
      }
      wv_sync();
      if (!wv_any(moved)) break;
    }
    // mj_dsuAssign: islands in ascending order of their smallest tree
    int nisland = 0;
    for (int t0 = 0; t0 < ntree; t0 += MJH_W) {
      const int t = t0 + wv_lane();
      const int root = t < ntree && label[t] == t;
      const int before = wv_exscan_i(root);
      if (root) tree_island[t] = nisland + before;
      nisland += wv_sum_i(root);
    }
    wv_sync();
    MJH_FOR_LANES(t, ntree) { const int l = label[t]; if (l < 0) tree_island[t] = -1; else if (l != t) tree_island[t] = tree_island[l]; }
    wv_sync();
    MJH_FOR_LANES(i, nefc) P.island[i] = tree_island[efc_tree[i]];
    if (in_lds) MJH_FOR_LANES(t, ntree) tree_island_home[t] = tree_island[t];
    if (wv_lane() == 0) counts[MJH_C_NISLAND] = nisland;
  }
  wv_sync();
}

// Sparse path under the dual solver: the reference keeps efc_Y and efc_AR compressed (computeY_precount / _fill /
// _backsub, mju_sqrMatTDSparse; engine_core_constraint.c:2698-3090).  Their VALUES are those of the dense arrays built
// above -- a sparse routine only skips structural zeros -- but the PGS sweep and the warm start then take
// mju_dotSparse over a row's stored entries, four accumulators by position in the COMPRESSED row.  What is needed
// from the sparse representation is therefore each row's structural pattern: row i of Y holds the dofs of J's row
// closed under "ancestor of" (row j of M for every dof j of the row), and AR[i][j] is stored iff the Y patterns of
// rows i and j share a dof.  One bit per (i, j), nARw 64-bit words per row.
MJH_DEV void project_sparse_patterns(MREF M, BREF B, int e, const Efc& P, int nefc) {
#if !MJH_LANE_MODE
  const int nw = M.s.nARw;
  iptr ymask = MJH_G(B, iscratch, e);                 // [4*nefc] closed row patterns (scratch: free between make and the solve)
  iptr arm = MJH_G(B, sp_ARmask, e);
  MJH_FOR_LANES(r, nefc) {
    M128 pm = m128_ld(P.rowmask + 4*r), ym = m128_zero();
    while (m128_any(pm)) {
      const int j = m128_lowest(pm);
      pm = m128_drop_lowest(pm);
      const int ma = M.M_rowadr[j], mn = M.M_rownnz[j];
      for (int q = 0; q < mn; q++) ym = m128_or(ym, m128_bit(M.M_colind[ma + q]));
    }
    m128_st(ymask + 4*r, ym);
  }
  wv_sync();
  for (int i = 0; i < nefc; i++) {
    const M128 yi = m128_ld(ymask + 4*i);
    for (int w = 0; w < nw; w++) {
      const int j = 64*w + wv_lane();
      int bit = 0;
      if (j < nefc) { const M128 yj = m128_ld(ymask + 4*j); bit = m128_any(m128_and(yi, yj)); }
      const unsigned long long m = wv_ballot(bit);
      if (wv_lane() == 0) { arm[2*(i*nw + w)] = (int)(unsigned)m; arm[2*(i*nw + w) + 1] = (int)(unsigned)(m >> 32); }
    }
  }
  wv_sync();
#else
  (void)M; (void)B; (void)e; (void)P; (void)nefc;
#endif
}

// ------------------------------------------------------------------------------------------------
// mj_projectConstraint for dual solvers: Y = J L^-T D^-1/2, AR = Y Y' + diag(R)
//                                                  (engine_core_constraint.c:2918-3137)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_project(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  const int nefc = MJH_F(B, counts, e)[MJH_C_NEFC];
  if (!nefc) return;
  crptr qLD = MJH_F(B, qLD, e);
  Efc P;
  efc_layout(M, B, e, nefc, P);
  crptr J = P.J;
  rptr Y = P.Y;
  rptr AR = P.AR;
  crptr R = P.R;
  rptr sqrtInvD = P.sqrtInvD;

  MJH_FOR_LANES(i, nv) sqrtInvD[i] = 1 / sqrt(qLD[M.M_rowadr[i] + M.M_rownnz[i] - 1]);
  wv_sync();
  // one lane per row: half back-substitution (mj_solveM2, engine_core_smooth.c:2130).
  // The sweep is a chain of ~100 dependent read-modify-writes on the row; when the row's home is global
  // memory (Y rarely fits the plan) it is staged through the unused tail of the LDS plan, a batch of
  // rows at a time, element-major ([dof][row of the batch]: the dof index is the same in every lane, so
  // the lanes of a batch touch consecutive words), and copied out coalesced.
  int rows_done = 0;
#if !MJH_LANE_MODE && MJH_W == 64
  if (M.s.ld_fast && nv <= 32) {
    rows_done = 1;
    // Register form: lane = dof, a constraint row's vector lives in one register per lane, the two
    // halves of the wavefront take a row each and every pass carries PR rows per half (independent
    // chains for the scheduler).  Step i of the sweep broadcasts x_i and subtracts q_i * x_i in the
    // lanes that are ancestors of dof i (bit mask) -- the same products in the same order as the row
    // loop below; the coefficient q_i is the same for every row and is read once per pass.
    const auto* ancmask = wv_uniform_ptr(M.dof_ancmask);
    const int lane = wv_lane();
    const int half = lane >> 5, hl = lane & 31;
    const int li = hl < nv ? hl : 0;
    const int myadr = wv_uniform_ptr(M.M_rowadr)[li];
    const int mynnz = wv_uniform_ptr(M.M_rownnz)[li];
    const int mydepth = mynnz - 1;
    const int anc_lo = ancmask[2*li];
    const real sq = hl < nv ? (real)sqrtInvD[li] : (real)0;
    int ilast = nv - 1;
    while (ilast > 0 && wv_bcast_i(mynnz, ilast) == 1) ilast--;
    // PR rows per half and pass: every pass pays the sweep's bookkeeping once, so few passes with many rows
    // each -- but a slot without a row still costs its updates, hence the choice by nefc
    auto sweep = [&](auto pr_) {
      constexpr int PR = decltype(pr_)::value;
      for (int r0 = 0; r0 < nefc; r0 += 2*PR) {
        real x[PR];
        int rr[PR];
        for (int g = 0; g < PR; g++) {
          rr[g] = r0 + 2*g + half;
          x[g] = (hl < nv && rr[g] < nefc) ? (real)J[(size_t)rr[g]*nv + li] : (real)0;
        }
        real q = 0;
        {
          const int lo = wv_bcast_i(anc_lo, ilast), adr = wv_bcast_i(myadr, ilast);
          if ((lo >> hl) & 1) q = qLD[adr + mydepth];
        }
        for (int i = ilast; i > 0; ) {
          int inext = i - 1;
          while (inext > 0 && wv_bcast_i(mynnz, inext) == 1) inext--;
          real qnext = 0;
          if (inext > 0) {
            const int lon = wv_bcast_i(anc_lo, inext), adrn = wv_bcast_i(myadr, inext);
            if ((lon >> hl) & 1) qnext = qLD[adrn + mydepth];
          }
          const int isanc = (wv_bcast_i(anc_lo, i) >> hl) & 1;
          for (int g = 0; g < PR; g++) {
            const real a = wv_bcast(x[g], i), b = wv_bcast(x[g], 32 + i);
            const real xi = half ? b : a;
            if (xi != 0 && isanc) x[g] -= q * xi;
          }
          q = qnext;
          i = inext;
        }
        for (int g = 0; g < PR; g++)
          if (hl < nv && rr[g] < nefc) Y[(size_t)rr[g]*nv + li] = x[g] * sq;
      }
    };
    struct PR2 { enum { value = 2 }; }; struct PR3 { enum { value = 3 }; }; struct PR5 { enum { value = 5 }; };
    if (nefc <= 4) sweep(PR2{});
    else if (nefc <= 6 || (nefc > 10 && nefc <= 12)) sweep(PR3{});
    else sweep(PR5{});
  }
#endif
  const int stage_rows = (int)((unsigned)P.free_bytes / ((unsigned)nv*sizeof(real)));
  if (rows_done) {
    // (register form above)
  } else if (!MJH_LANE_MODE && stage_rows >= 4 && !mjh_staged_home(Y)) {
    const int RB = stage_rows < MJH_W ? stage_rows : MJH_W;
    const auto xs = mjh_local((real*)P.free_p);        // ds_read / ds_write
    const int lane = wv_lane();
    for (int r0 = 0; r0 < nefc; r0 += RB) {
      const int nb = (nefc - r0) < RB ? (nefc - r0) : RB;
      // J rows of the batch in, transposed
      MJH_FOR_LANES(w, nb*nv) { const int r = w / nv, i = w - r*nv; xs[i*RB + r] = J[(size_t)(r0 + r)*nv + i]; }
      wv_sync();
      if (lane < nb) {
        const auto x = mjh_local((real*)P.free_p + lane);
        for (int i = nv - 1; i > 0; i--) {
          if (M.dof_simplenum[i]) continue;
          real xi = x[i*RB];
          if (xi != 0) {
            int start = M.M_rowadr[i], end = start + M.M_rownnz[i] - 1;
            for (int adr = start; adr < end; adr++) x[M.M_colind[adr]*RB] -= qLD[adr] * xi;
          }
        }
        for (int i = 0; i < nv; i++) x[i*RB] *= sqrtInvD[i];
      }
      wv_sync();
      MJH_FOR_LANES(w, nb*nv) { const int r = w / nv, i = w - r*nv; Y[(size_t)(r0 + r)*nv + i] = xs[i*RB + r]; }
      wv_sync();
    }
  } else
  MJH_FOR_LANES(r, nefc) {
    rptr x = Y + (size_t)r*nv;
    crptr y = J + (size_t)r*nv;
    for (int i = 0; i < nv; i++) x[i] = y[i];
    for (int i = nv - 1; i > 0; i--) {
      if (M.dof_simplenum[i]) continue;
      real xi = x[i];
      if (xi != 0) {
        int start = M.M_rowadr[i], end = start + M.M_rownnz[i] - 1;
        for (int adr = start; adr < end; adr++) x[M.M_colind[adr]] -= qLD[adr] * xi;
      }
    }
    for (int i = 0; i < nv; i++) x[i] *= sqrtInvD[i];
  }
  wv_sync();
#if !MJH_LANE_MODE && MJH_W == 64 && !defined(MJH_HOSTSIM)
  // The one dense contraction of the step on the matrix cores (opt-in, mjhip_batch_set_mfma):
  // AR is tiled 16 x 16, a tile is ceil(nv/4) v_mfma_f64_16x16x4_f64 over the dof dimension with
  // A = rows of Y of the tile's row block, B = rows of Y of its column block (lane l feeds
  // element [l & 15][4c + (l >> 4)] of both), accumulators in 4 VGPR pairs per lane:
  // D[(l >> 4) + 4 r][l & 15].  The hardware's summation order (and its fused multiply-adds)
  // differ from mju_sqrMatTD's, so this path agrees with the reference to rounding (1e-15 relative
  // on AR), not bit for bit: solver iteration counts may move by one.
  if (B.mfma) {
    typedef double d4_ __attribute__((ext_vector_type(4)));
    const int lane = wv_lane(), r16 = lane & 15, kq = lane >> 4;
    const int nt = (nefc + 15) >> 4;
    for (int I = 0; I < nt; I++) {
      for (int K = 0; K <= I; K++) {
        d4_ acc = {0, 0, 0, 0};
        const int ri = 16*I + r16, rk = 16*K + r16;
        for (int c = 0; c < nv; c += 4) {
          const int j = c + kq;
          const real a = (ri < nefc && j < nv) ? (real)Y[(size_t)ri*nv + j] : (real)0;
          const real b = (rk < nefc && j < nv) ? (real)Y[(size_t)rk*nv + j] : (real)0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; r++) {
          const int row = 16*I + kq + 4*r, col = 16*K + r16;
          if (row < nefc && col <= row) {
            real v = acc[r];
            if (row == col) v += R[row];
            AR[(size_t)row*nefc + col] = v;
            AR[(size_t)col*nefc + row] = v;
          }
        }
      }
    }
    wv_sync();
    if (MJH_HAS(MJH_FT_PRIMAL) && s.sparse) project_sparse_patterns(M, B, e, P, nefc);
    return;
  }
#endif
  // AR lower triangle: AR[i][k] = sum_j Y[k][j]*Y[i][j] in j order (mju_sqrMatTD on Y', engine_util_blas.c:664)
  const int npairs = nefc*(nefc + 1)/2;
  MJH_FOR_LANES(w, npairs) {
    int i = (int)((sqrt(8.0*w + 1.0) - 1.0)*0.5);
    while (i*(i+1)/2 > w) i--;
    while ((i+1)*(i+2)/2 <= w) i++;
    int k = w - i*(i+1)/2;
    // (lane mode runs this loop serially; it is only reached there on the rare reset-and-redo path)
    crptr Yi = Y + (size_t)i*nv;
    crptr Yk = Y + (size_t)k*nv;
    // (the reference skips zero entries of row i; selecting on the sum instead of branching around the
    // second load keeps the 2*nv loads independent of each other -- they are in flight together)
    real acc = 0;
    int j = 0;
    for (; j + 4 <= nv; j += 4) {
      const real t0 = Yi[j], t1 = Yi[j+1], t2 = Yi[j+2], t3 = Yi[j+3];
      const real u0 = Yk[j], u1 = Yk[j+1], u2 = Yk[j+2], u3 = Yk[j+3];
      real c;
      c = acc + u0*t0; acc = (t0 != 0) ? c : acc;
      c = acc + u1*t1; acc = (t1 != 0) ? c : acc;
      c = acc + u2*t2; acc = (t2 != 0) ? c : acc;
      c = acc + u3*t3; acc = (t3 != 0) ? c : acc;
    }
    for (; j < nv; j++) {
      const real t = Yi[j], u = Yk[j];
      const real c = acc + u*t;
      acc = (t != 0) ? c : acc;
    }
    if (i == k) acc += R[i];
    AR[(size_t)i*nefc + k] = acc;
    AR[(size_t)k*nefc + i] = acc;
  }
  wv_sync();
  if (MJH_HAS(MJH_FT_PRIMAL) && s.sparse) project_sparse_patterns(M, B, e, P, nefc);
}

// ------------------------------------------------------------------------------------------------
// mj_referenceConstraint: efc_vel = J qvel, aref   (engine_core_constraint.c:3245-3270)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_reference(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const int nv = M.s.nv;
  const int nefc = MJH_F(B, counts, e)[MJH_C_NEFC];
  if (!nefc) return;
  Efc P;
  efc_layout(M, B, e, nefc, P);
  crptr J = P.J;
  crptr qvel = MJH_F(B, qvel, e);
  crptr KBIP = P.KBIP;
  crptr pos = P.pos;
  crptr margin = P.margin;
  rptr vel = P.vel;
  rptr aref = P.aref;
  const int sparse = MJH_HAS(MJH_FT_PRIMAL) && M.s.sparse;     // mj_mulJacVec takes mju_mulMatVecSparse
  const int csr = MJH_HAS(MJH_FT_PRIMAL) && M.s.csr;
  MJH_FOR_LANES(r, nefc) {
    real v = csr ? csr_row_dot(P, r, qvel) : sparse ? sp_row_dot(P, r, qvel) : dot_ref(J + (size_t)r*nv, qvel, nv);
    vel[r] = v;
    aref[r] = -KBIP[4*r+1]*v - KBIP[4*r]*KBIP[4*r+2]*(pos[r] - margin[r]);
  }
  wv_sync();
  // relative surface velocity of the contacting geoms (conveyor belts) enters efc_vel of the
  // tangential / torsional rows before aref is formed  (mj_addSurfaceVel, :3141-3204;
  // mj_geomSurfaceVelocity, engine_core_util.c:892-905)
  if (MJH_HAS(MJH_FT_PASSIVEMISC) && M.o.has_surfacevel) {
    const MJH_CONST_AS DSizes& s = M.s;
    const int ncon = MJH_F(B, counts, e)[MJH_C_NCON];
    crptr gx = MJH_F(B, geom_xpos, e);
    crptr gm = MJH_F(B, geom_xmat, e);
    MJH_FOR_LANES(c, ncon) {
      const int adr = MJH_CON(B, con_efcadr, e, 1, c)[0];
      if (adr < 0) continue;
      ciptr cg = MJH_CON(B, con_geom, e, 2, c);
      crptr cp = MJH_CON(B, con_pos, e, 3, c);
      crptr fr = MJH_CON(B, con_frame, e, 9, c);
      real svel[3] = {0, 0, 0}, sang[3] = {0, 0, 0};
      int active = 0;
      for (int side = 0; side < 2; side++) {
        const int g = cg[side];
        if (g < 0) continue;
        auto sv = M.geom_surfacevel + 6*g;
        if (!sv[0] && !sv[1] && !sv[2] && !sv[3] && !sv[4] && !sv[5]) continue;
        active = 1;
        const real sgn = side ? 1 : -1;
        real lin[3], ang[3], arm[3], wxr[3], svl[3] = {sv[0], sv[1], sv[2]}, sva[3] = {sv[3], sv[4], sv[5]};
        m3_mulvec(lin, gm + 9*g, svl);
        m3_mulvec(ang, gm + 9*g, sva);
        v3_sub(arm, cp, gx + 3*g);
        v3_cross(wxr, ang, arm);
        v3_addto(lin, wxr);
        v3_addtoscl(svel, lin, sgn);
        v3_addtoscl(sang, ang, sgn);
      }
      if (!active) continue;
      real cs[6];
      m3_mulvec(cs, fr, svel);
      m3_mulvec(cs + 3, fr, sang);
      cs[0] = 0; cs[4] = 0; cs[5] = 0;
      const int dim = MJH_CON(B, con_dim, e, 1, c)[0];
      auto mu = M.pair_friction + 5*MJH_CON(B, con_pair, e, 1, c)[0];
      const int nrow = (dim == 1) ? 1 : (M.o.cone == 0 ? 2*(dim - 1) : dim);
      if (dim == 1 || M.o.cone != 0) {
        for (int j = 0; j < dim; j++) vel[adr + j] += cs[j];
      } else {
        for (int k = 1; k < dim; k++) {
          vel[adr + 2*(k-1)] += cs[0] + mu[k-1]*cs[k];
          vel[adr + 2*(k-1) + 1] += cs[0] - mu[k-1]*cs[k];
        }
      }
      for (int j = 0; j < nrow; j++) {
        const int r = adr + j;
        aref[r] = -KBIP[4*r+1]*vel[r] - KBIP[4*r]*KBIP[4*r+2]*(pos[r] - margin[r]);
      }
    }
    wv_sync();
  }
  // adhesive contact rows: aref += R * adhesion on the normal row, or the pyramid's edges in equal shares
  // (mj_adhesionRef, :3214-3241)
  if (MJH_HAS(MJH_FT_PASSIVEMISC) && M.o.has_adhesion) {
    const int ncon = MJH_F(B, counts, e)[MJH_C_NCON];
    crptr R = P.R;
    MJH_FOR_LANES(c, ncon) {
      const int adr = MJH_CON(B, con_efcadr, e, 1, c)[0];
      const real adh = M.pair_adhesion[MJH_CON(B, con_pair, e, 1, c)[0]];
      if (adh == 0 || adr < 0) continue;
      const int dim = MJH_CON(B, con_dim, e, 1, c)[0];
      if (dim == 1 || M.o.cone != 0) {
        aref[adr] += R[adr]*adh;
      } else {
        const real edge = adh / (2*(dim - 1));
        for (int j = 0; j < 2*(dim - 1); j++) aref[adr + j] += R[adr + j]*edge;
      }
    }
    wv_sync();
  }
  // subtract Jdot*v for connect / weld equalities               (mj_Jdotv, :1056-1250)
  const int ne = MJH_F(B, counts, e)[MJH_C_NE];
  if (MJH_HAS(MJH_FT_EQUALITY) && ne) {
    const MJH_CONST_AS DSizes& s = M.s;
    crptr cdof = MJH_F(B, cdof, e);
    crptr cdof_dot = MJH_F(B, cdof_dot, e);
    crptr cvel = MJH_F(B, cvel, e);
    crptr subtree_com = MJH_F(B, subtree_com, e);
    ciptr efcadr = MJH_G(B, eq_efcadr, e);
    rptr tmp = MJH_G(B, scratch, e) + 6*M.s.nefcmax;      // 4 x 3 partial sums per lane pass
    for (int q = 0; q < s.neq; q++) {
      const int r0 = efcadr[q];
      const int et = M.eq_type[q];
      if (r0 < 0 || (et != MJH_EQ_CONNECT && et != MJH_EQ_WELD)) continue;
      real pos0[3], pos1[3];
      int b0, b1;
      equality_anchors(M, B, e, q, pos0, pos1, &b0, &b1);
      // mj_jacDot(point, body) * qvel, dense: every lane accumulates the columns it owns in dof order
      // (mju_mulMatVec = row-wise mju_dot: the four-accumulator order is reproduced below)
      real jdv[2][3], jrdv[2][3];
      for (int side = 0; side < 2; side++) {
        const int body = side ? b1 : b0;
        auto point = side ? pos1 : pos0;
        const int wb = M.body_weldid[body];
        real offset[3], pvel3[3];
        auto com = subtree_com + 3*M.body_rootid[body];
        v3_sub(offset, point, com);
        {
          // mju_transformSpatial(pvel, cvel[body], 0, point, com, 0): linear part
          real dif[3], cros[3];
          v3_sub(dif, point, com);
          v3_cross(cros, dif, cvel + 6*body);
          v3_sub(pvel3, cvel + 6*body + 3, cros);
        }
        // columns
        rptr colp = MJH_G(B, nt_vec, e);         // [3*nv] translational, then [3*nv] rotational (needs 6*nv <= 8*nv)
        MJH_FOR_LANES(j, nv) {
          const int in = (M.body_dofanc[wb*s.nvw + (j >> 5)] >> (j & 31)) & 1;
          real jp[3] = {0, 0, 0}, jr[3] = {0, 0, 0};
          if (in) {
            real cd_dot[6];
            for (int k = 0; k < 6; k++) cd_dot[k] = cdof_dot[6*j + k];
            const int jt = M.dof_jnttype[j];
            const int dadr = M.jnt_dofadr[M.dof_jntid[j]];
            const int is_quat = jt == MJH_JNT_BALL || (jt == MJH_JNT_FREE && j >= dadr + 3);
            if (is_quat) sp_cross_motion(cd_dot, cvel + 6*M.dof_bodyid[j], cdof + 6*j);
            real t1[3], t2[3];
            v3_cross(t1, cd_dot, offset);
            v3_cross(t2, cdof + 6*j, pvel3);
            // zero-initialised, then +=
            jr[0] += cd_dot[0]; jr[1] += cd_dot[1]; jr[2] += cd_dot[2];
            jp[0] += cd_dot[3] + t1[0] + t2[0]; jp[1] += cd_dot[4] + t1[1] + t2[1]; jp[2] += cd_dot[5] + t1[2] + t2[2];
          }
          for (int k = 0; k < 3; k++) { colp[k*nv + j] = jp[k]; colp[(3 + k)*nv + j] = jr[k]; }
        }
        wv_sync();
        if (sparse) {
          // sparse: mju_dotSparseX3 over the merged chain of BOTH bodies (mj_mergeChain, common dofs kept), one
          // running sum per component in ascending dof order (engine_util_sparse.c:30-55)
          const M128 both = m128_or(m128_ldw(M.body_dofanc + (size_t)b0*s.nvw, s.nvw), m128_ldw(M.body_dofanc + (size_t)b1*s.nvw, s.nvw));
          for (int k = 0; k < 3; k++) { jdv[side][k] = 0; jrdv[side][k] = 0; }
          for (M128 um = both; m128_any(um); um = m128_drop_lowest(um)) {
            const int j = m128_lowest(um);
            const real v2 = qvel[j];
            for (int k = 0; k < 3; k++) { jdv[side][k] += colp[k*nv + j]*v2; jrdv[side][k] += colp[(3 + k)*nv + j]*v2; }
          }
        } else
        for (int k = 0; k < 3; k++) {
          jdv[side][k] = dot_ref(colp + k*nv, qvel, nv);
          jrdv[side][k] = dot_ref(colp + (3 + k)*nv, qvel, nv);
        }
        wv_sync();
      }
      if (wv_lane() == 0) {
        for (int k = 0; k < 3; k++) aref[r0 + k] -= jdv[0][k] - jdv[1][k];
        if (et == MJH_EQ_WELD) {
          auto data = M.eq_data + 11*q;
          const real torquescale = data[10];
          crptr xquat = MJH_F(B, xquat, e);
          real q0r[4], negq1[4];
          weld_quats(M, B, e, q, b0, b1, q0r, negq1);
          auto omega1 = cvel + 6*b0;
          auto omega2 = cvel + 6*b1;
          real domega[3];
          v3_sub(domega, omega1, omega2);
          real qdot0[4], qdot0r[4], negqdot1[4];
          if (!M.eq_objsite[q]) {
            q_deriv(qdot0, xquat + 4*b0, omega1);
            q_mul(qdot0r, qdot0, data + 6);
            real qdot1[4];
            q_deriv(qdot1, xquat + 4*b1, omega2);
            q_neg(negqdot1, qdot1);
          } else {
            real qfull0[4], qfull1[4], qdot1[4];
            q_mul(qfull0, xquat + 4*b0, M.site_quat + 4*M.eq_obj1id[q]);
            q_deriv(qdot0, qfull0, omega1);
            q_copy(qdot0r, qdot0);
            q_mul(qfull1, xquat + 4*b1, M.site_quat + 4*M.eq_obj2id[q]);
            q_deriv(qdot1, qfull1, omega2);
            q_neg(negqdot1, qdot1);
          }
          real djrdv[3] = {jrdv[0][0] - jrdv[1][0], jrdv[0][1] - jrdv[1][1], jrdv[0][2] - jrdv[1][2]};
          real t1a[4], t1[4], t2a[4], t2[4], t3a[4], t3[4];
          q_mulaxis(t1a, negqdot1, domega); q_mul(t1, t1a, q0r);
          q_mulaxis(t2a, negq1, djrdv);     q_mul(t2, t2a, q0r);
          q_mulaxis(t3a, negq1, domega);    q_mul(t3, t3a, qdot0r);
          for (int k = 0; k < 3; k++) aref[r0 + 3 + k] -= 0.5 * (t1[1 + k] + t2[1 + k] + t3[1 + k]) * torquescale;
        }
      }
      wv_sync();
    }
    (void)tmp;
  }
}

// ------------------------------------------------------------------------------------------------
// elliptic cone blocks: rows [i, i+dim) of one contact; P.cone holds mu on the normal row and
// friction[j-1] on the others
// ------------------------------------------------------------------------------------------------
MJH_DEV int cone_leader(const Efc& P, int i) {
  return P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC &&
         (i == 0 || P.type[i-1] != MJH_CNSTR_CONTACT_ELLIPTIC || P.id[i-1] != P.id[i]);
}
MJH_DEV int cone_dim(const Efc& P, int i, int nefc) {
  int dim = 1;
  while (i + dim < nefc && P.type[i + dim] == MJH_CNSTR_CONTACT_ELLIPTIC && P.id[i + dim] == P.id[i]) dim++;
  return dim;
}

// zone of the block at residual jar: 0 top (satisfied), 1 bottom (quadratic), 2 middle (cone);
// fills U (regular-cone coordinates), N, T                 (engine_core_constraint.c:3357-3372)
template <class P0>
MJH_DEV int cone_zone(const Efc& P, int i, int dim, P0 jar, real* U, real* N_, real* T_) {
  const real mu = P.cone[i];
  U[0] = jar[i]*mu;
  for (int j = 1; j < dim; j++) U[j] = jar[i+j]*P.cone[i+j];
  const real N = U[0];
  const real T = sqrt(dot_ref(U + 1, U + 1, dim - 1));
  *N_ = N; *T_ = T;
  if (N >= mu*T || (T <= 0 && N >= 0)) return 0;
  if (mu*N + T <= 0 || (T <= 0 && N < 0)) return 1;
  return 2;
}

// force/state of one block; optional cone Hessian (dim x dim, row-major) for the middle zone
//                                                           (engine_core_constraint.c:3352-3452)
template <class P0, class P1>
MJH_DEV void cone_update(const Efc& P, int i, int dim, P0 jar, P1 Hc, int want_hessian) {
  real U[6], N, T;
  const int zone = cone_zone(P, i, dim, jar, U, &N, &T);
  const real mu = P.cone[i];
  int st;
  if (zone == 0) {
    for (int j = 0; j < dim; j++) P.force[i+j] = 0;
    st = MJH_STATE_SATISFIED;
  } else if (zone == 1) {
    for (int j = 0; j < dim; j++) P.force[i+j] = -P.D[i+j]*jar[i+j];
    st = MJH_STATE_QUADRATIC;
  } else {
    const real Dm = P.D[i]/(mu*mu*(1+mu*mu));
    const real NmT = N - mu*T;
    const real f0 = -Dm*NmT*mu;
    P.force[i] = f0;
    for (int j = 1; j < dim; j++) P.force[i+j] = -f0/T*U[j]*P.cone[i+j];
    st = MJH_STATE_CONE;
    if (want_hessian) {
      real scl = -mu/T;
      Hc[0] = 1;
      for (int j = 1; j < dim; j++) Hc[j] = scl*U[j];
      scl = mu*N/(T*T*T);
      for (int k = 1; k < dim; k++)
        for (int j = k; j < dim; j++) Hc[k*dim+j] = scl*U[j]*U[k];
      scl = mu*mu - mu*N/T;
      for (int j = 1; j < dim; j++) Hc[j*(dim+1)] += scl;
      for (int k = 0; k < dim; k++) {
        scl = Dm * (k == 0 ? mu : (real)P.cone[i+k]);
        for (int j = k; j < dim; j++) Hc[k*dim+j] *= scl * (j == 0 ? mu : (real)P.cone[i+j]);
      }
      for (int k = 0; k < dim; k++)
        for (int j = k + 1; j < dim; j++) Hc[j*dim+k] = Hc[k*dim+j];
    }
  }
  for (int j = 0; j < dim; j++) P.state[i+j] = st;
}

// cost of one block at residual jar                        (engine_core_constraint.c:3374-3392)
template <class P0>
MJH_DEV real cone_cost(const Efc& P, int i, int dim, P0 jar) {
  real U[6], N, T;
  const int zone = cone_zone(P, i, dim, jar, U, &N, &T);
  real c = 0;
  if (zone == 1) {
    for (int j = 0; j < dim; j++) c += 0.5*P.D[i+j]*jar[i+j]*jar[i+j];
  } else if (zone == 2) {
    const real mu = P.cone[i];
    const real Dm = P.D[i]/(mu*mu*(1+mu*mu));
    const real NmT = N - mu*T;
    c = 0.5*Dm*NmT*NmT;
  }
  return c;
}

// ------------------------------------------------------------------------------------------------
// mj_constraintUpdate_impl without cone Hessians    (engine_core_constraint.c:3275-3468)
// writes force/state; returns the cost in lane-uniform form (summed in row order by every lane)
// ------------------------------------------------------------------------------------------------
template <class P0>
MJH_DEV real constraint_update(BREF B, int e, const Efc& P, P0 jar, int want_cost, int elliptic) {
  ciptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ne = counts[MJH_C_NE], nf = counts[MJH_C_NF];
  crptr D = P.D;
  crptr R = P.R;
  crptr floss = P.floss;
  rptr force = P.force;
  iptr state = P.state;
  MJH_FOR_LANES(i, nefc) {
    if (elliptic && i >= ne + nf && P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC) {
      if (cone_leader(P, i)) cone_update(P, i, cone_dim(P, i, nefc), jar, (real*)nullptr, 0);
      continue;
    }
    real f = -D[i]*jar[i];
    int st;
    if (i < ne) {
      st = MJH_STATE_QUADRATIC;
    } else if (i < ne + nf) {
      if (jar[i] <= -R[i]*floss[i]) { f = floss[i]; st = MJH_STATE_LINEARNEG; }
      else if (jar[i] >= R[i]*floss[i]) { f = -floss[i]; st = MJH_STATE_LINEARPOS; }
      else st = MJH_STATE_QUADRATIC;
    } else {
      if (jar[i] >= 0) { f = 0; st = MJH_STATE_SATISFIED; }
      else st = MJH_STATE_QUADRATIC;
    }
    force[i] = f;
    state[i] = st;
  }
  wv_sync();
  real cost = 0;
  if (want_cost) {
    for (int i = 0; i < nefc; i++) {
      if (i < ne) {
        cost += 0.5*D[i]*jar[i]*jar[i];
      } else if (i < ne + nf) {
        if (jar[i] <= -R[i]*floss[i]) cost += -0.5*R[i]*floss[i]*floss[i] - floss[i]*jar[i];
        else if (jar[i] >= R[i]*floss[i]) cost += -0.5*R[i]*floss[i]*floss[i] + floss[i]*jar[i];
        else cost += 0.5*D[i]*jar[i]*jar[i];
      } else if (elliptic && P.type[i] == MJH_CNSTR_CONTACT_ELLIPTIC) {
        const int dim = cone_dim(P, i, nefc);
        cost += cone_cost(P, i, dim, jar);
        i += dim - 1;
      } else if (jar[i] < 0) {
        cost += 0.5*D[i]*jar[i]*jar[i];
      }
    }
  }
  return cost;
}
