// The implicit effective metric Mtilde = M + K of flex models under CG with an implicit integrator
// (mj_flexCG, engine_forward.c:1640; mjd_effBuild / mjd_effShift / mjd_effMulAdd / mjd_effPrec / mjd_effSolve,
// engine_derivative.c:3168-3467; mjd_flexStiff_assemble :1810, mjd_flexBend_mul :1358, mjd_flexStretch_mul :1443).
//
//   stage_eff_build   K = (h^2 + h damping)(K_bend + K_stretch) into the static CSR structure of the model build, the factored
//                     3 x 3 diagonal blocks of the flex part of M + K, and the shift c = -h (K_bend + K_stretch) qvel taken
//                     through the reference's stencil operators (different sums than K qvel)
//   eff_mul_add       res += K vec                      (mjd_effMulAdd: one mju_dotSparse per covered row)
//   eff_block_apply   the block preconditioner         (effBlockApply: M^-1 off the covered dofs, block solves on them)
//   eff_solve         (M + K) x = b by PCG              (mjd_effSolve; qacc_smooth)
//
// Mapping.  The reference assembles K element by element into shared CSR slots; here an element computes its (dim+1)^2
// blocks once (lane per element), and a lane per CSR block then sums the block's contributions in the reference's order
// (bending stencils in edge order, then stretch elements in element order: the list of the model build).  The stencil
// operators of the shift likewise: per-edge / per-element terms first, then a lane per vertex sums them in the order the
// reference adds them to that vertex's dofs.  Scope (model build): standard flexes whose vertices are bodies of three
// sliders or pinned; no passive contacts.  With bending stiffness ONLY (s.efm == 2) the reference assembles no CSR: K vec
// is the bending stencil operator (mjd_flexBend_mul :1358) and the preconditioner solves the covered dofs with the CONSTANT
// sparse factor of M + (h^2 + h damping) K_bend that mj_setConst left in the model (effBlockApply's flg_bend branch :3288;
// mju_cholSolveSparse, engine_util_solve.c:387 -- its two sweeps run level by level, a lane per row whose operands are final,
// every row's subtractions in the reference's order).
// (included once per SPMD mode by mjh_stages.inc: no include guard)

#if !MJH_LANE_MODE

template <class PA, class PB>
MJH_DEV void em_mulmat3(real* r, PA a, PB b) {          // mji_mulMatMat3
  r[0] = a[0]*b[0] + a[1]*b[3] + a[2]*b[6]; r[1] = a[0]*b[1] + a[1]*b[4] + a[2]*b[7]; r[2] = a[0]*b[2] + a[1]*b[5] + a[2]*b[8];
  r[3] = a[3]*b[0] + a[4]*b[3] + a[5]*b[6]; r[4] = a[3]*b[1] + a[4]*b[4] + a[5]*b[7]; r[5] = a[3]*b[2] + a[4]*b[5] + a[5]*b[8];
  r[6] = a[6]*b[0] + a[7]*b[3] + a[8]*b[6]; r[7] = a[6]*b[1] + a[7]*b[4] + a[8]*b[7]; r[8] = a[6]*b[2] + a[7]*b[5] + a[8]*b[8];
}
template <class PA, class PB>
MJH_DEV void em_multmat3(real* r, PA a, PB b) {         // mji_mulMatTMat3
  r[0] = a[0]*b[0] + a[3]*b[3] + a[6]*b[6]; r[1] = a[0]*b[1] + a[3]*b[4] + a[6]*b[7]; r[2] = a[0]*b[2] + a[3]*b[5] + a[6]*b[8];
  r[3] = a[1]*b[0] + a[4]*b[3] + a[7]*b[6]; r[4] = a[1]*b[1] + a[4]*b[4] + a[7]*b[7]; r[5] = a[1]*b[2] + a[4]*b[5] + a[7]*b[8];
  r[6] = a[2]*b[0] + a[5]*b[3] + a[8]*b[6]; r[7] = a[2]*b[1] + a[5]*b[4] + a[8]*b[7]; r[8] = a[2]*b[2] + a[5]*b[5] + a[8]*b[8];
}
// local edge -> corner tables of mjd_flexStretch_mul (:1431)
MJH_DEV int em_edge_corner(int dim, int ed, int which) { return flex_edge_corner(dim, ed, which); }

// unpacked metric, edge vectors and clamped edge tensions of element t (global id) of flex f
struct EmElem { real metric[36]; real dvec[6][3]; real Me[6]; int nedge, nvrt; };
template <class VX, class LEN>
MJH_DEV void em_element(MREF M, int f, int t, VX vx, LEN len, EmElem& E) {
  const int dim = M.flex_dim[f];
  const int nedge = dim == 2 ? 3 : 6;
  E.nedge = nedge; E.nvrt = dim + 1;
  for (int ed = 0; ed < nedge; ed++) {
    const int v0 = M.flexelem_vert[4*t + em_edge_corner(dim, ed, 0)], v1 = M.flexelem_vert[4*t + em_edge_corner(dim, ed, 1)];
    for (int x = 0; x < 3; x++) E.dvec[ed][x] = vx[3*v0 + x] - vx[3*v1 + x];
  }
  auto k = M.flex_stiffness + M.flex_stiffnessadr[f] + 21*(t - M.flex_elemadr[f]);
  int id = 0;
  for (int e1 = 0; e1 < nedge; e1++)
    for (int e2 = e1; e2 < nedge; e2++) { E.metric[nedge*e1 + e2] = k[id]; E.metric[nedge*e2 + e1] = k[id]; id++; }
  for (int e1 = 0; e1 < nedge; e1++) {
    real me = 0;
    for (int e2 = 0; e2 < nedge; e2++) {
      const int idx = M.flexelem_edge[6*t + e2];
      me += E.metric[nedge*e1 + e2]*(len[idx]*len[idx] - M.flexedge_length0[idx]*M.flexedge_length0[idx]);
    }
    E.Me[e1] = me >= 0 ? me : (real)0;                  // mju_max(Me, 0)
  }
}
MJH_DEV int em_flex_stretch(MREF M, int f) {
  const int sadr = M.flex_stiffnessadr[f];
  return !M.flex_rigid[f] && M.flex_dim[f] >= 2 && sadr >= 0 && M.flex_stiffness[sadr] != 0;
}
MJH_DEV int em_flex_bend(MREF M, int f) { return !M.flex_rigid[f] && M.flex_dim[f] == 2 && M.flex_bendingadr[f] >= 0; }


// per-edge terms of (s1 + s2 damping) K_bend vec (mjd_flexBend_mul :1358): the four flap vertices' 3-vectors of every
// interior edge, into flexbend_frc[24 ed + 3 i + x]; a lane per vertex then adds them in edge order (eff_bend_gather)
template <class VP>
MJH_DEV void eff_bend_terms(MREF M, BREF B, int e, VP vec, real s1, real s2) {
  const MJH_CONST_AS DSizes& s = M.s;
  crptr xmat = MJH_F(B, xmat, e);
  rptr bterm = MJH_G(B, flexbend_frc, e);
  if (!s.nflexbend) return;
  MJH_FOR_LANES(ed, s.nflexedge) {
    const int f = M.flexedge_flex[ed];
    const int v3i = M.flexedge_flap[2*ed + 1];
    if (!em_flex_bend(M, f) || v3i < 0) continue;
    const real scale = s1 + s2*M.flex_damping[f];
    const int v[4] = {(int)M.flexedge_vert[2*ed], (int)M.flexedge_vert[2*ed + 1], (int)M.flexedge_flap[2*ed], v3i};
    auto b = M.flex_bending + M.flex_bendingadr[f] + 17*(ed - M.flex_edgeadr[f]);
    real w[4][3]; int has[4];
    for (int j = 0; j < 4; j++) {
      const int bj = M.flexvert_bodyid[v[j]];
      has[j] = M.body_dofnum[bj] != 0;
      if (has[j]) { real qv[3] = {vec[M.body_dofadr[bj]], vec[M.body_dofadr[bj] + 1], vec[M.body_dofadr[bj] + 2]}; m3_mulvec(w[j], xmat + 9*bj, qv); }
      else { w[j][0] = 0; w[j][1] = 0; w[j][2] = 0; }
    }
    for (int i = 0; i < 4; i++) {
      real vl[3] = {0, 0, 0};
      if (has[i]) {
        real vw[3] = {0, 0, 0};
        for (int j = 0; j < 4; j++) {
          if (!has[j]) continue;
          const real q = b[4*i + j];
          for (int x = 0; x < 3; x++) vw[x] += q*w[j][x];
        }
        m3_multvec(vl, xmat + 9*M.flexvert_bodyid[v[i]], vw);
      }
      for (int x = 0; x < 3; x++) bterm[24*ed + 3*i + x] = scale*vl[x];
    }
  }
}
// res += K_bend-terms of eff_bend_terms, vertex by vertex in the order the reference's edge loop reaches the vertex
template <class RP>
MJH_DEV void eff_bend_gather(MREF M, BREF B, int e, RP res) {
  const MJH_CONST_AS DSizes& s = M.s;
  crptr bterm = MJH_G(B, flexbend_frc, e);
  if (!s.nflexbend) return;
  MJH_FOR_LANES(v, s.nflexvert) {
    const int bid = M.flexvert_bodyid[v];
    if (!M.body_dofnum[bid] || !em_flex_bend(M, M.flexvert_flex[v])) continue;
    const int d0 = M.body_dofadr[bid];
    real acc[3] = {res[d0], res[d0 + 1], res[d0 + 2]};
    for (int a = M.flexvert_bendadr[v]; a < M.flexvert_bendadr[v + 1]; a++) {
      const int it = M.flexvert_bend[a];
      const int ed = it >> 2, i = it & 3;
      for (int x = 0; x < 3; x++) acc[x] += bterm[24*ed + 3*i + x];
    }
    for (int x = 0; x < 3; x++) res[d0 + x] = acc[x];
  }
}

MJH_DEVN void stage_eff_build(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const real h = M.o.timestep;
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr xmat = MJH_F(B, xmat, e);
  crptr len = MJH_F(B, flexedge_length, e);
  crptr qvel = MJH_F(B, qvel, e);
  crptr Ms = MJH_G(B, M, e);
  rptr eblk = MJH_G(B, efm_eblk, e);
  rptr Kv = MJH_G(B, efm_K_val, e);
  rptr Lb = MJH_G(B, efm_L, e);
  rptr cs = MJH_G(B, efm_c, e);

  if (s.efm == 1) {
  // ---- (a) stretch blocks of every element, in dof space: blkd(i, j) = R_bi' [2 scale sum_ab M_ab s_a s_b d_a d_b' + scale (sum_a Me_a s_a s_b) I] R_bj
  MJH_FOR_LANES(t, s.nflexelem) {
    const int f = M.flexelem_flex[t];
    if (!em_flex_stretch(M, f)) continue;
    const real scale = h*h + h*M.flex_damping[f];
    if (!scale) continue;
    EmElem E;
    em_element(M, f, t, vx, len, E);
    const int dim = M.flex_dim[f], nvrt = dim + 1, nedge = E.nedge;
    for (int i = 0; i < nvrt; i++) {
      const int vi = M.flexelem_vert[4*t + i];
      if (M.efm_vertslot[vi] < 0) continue;
      for (int j = 0; j < nvrt; j++) {
        const int vj = M.flexelem_vert[4*t + j];
        if (M.efm_vertslot[vj] < 0) continue;
        real blk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < nedge; a++) {
          const real sa = (i == em_edge_corner(dim, a, 0)) ? 1 : ((i == em_edge_corner(dim, a, 1)) ? -1 : 0);
          if (!sa) continue;
          for (int bb = 0; bb < nedge; bb++) {
            const real sb = (j == em_edge_corner(dim, bb, 0)) ? 1 : ((j == em_edge_corner(dim, bb, 1)) ? -1 : 0);
            if (!sb) continue;
            const real w = 2*scale*E.metric[nedge*a + bb]*sa*sb;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) blk[3*r + c] += w*E.dvec[a][r]*E.dvec[bb][c];
          }
        }
        real geo = 0;
        for (int a = 0; a < nedge; a++) {
          const real sa = (i == em_edge_corner(dim, a, 0)) ? 1 : ((i == em_edge_corner(dim, a, 1)) ? -1 : 0);
          const real sb = (j == em_edge_corner(dim, a, 0)) ? 1 : ((j == em_edge_corner(dim, a, 1)) ? -1 : 0);
          if (!sa || !sb) continue;
          geo += E.Me[a]*sa*sb;
        }
        geo *= scale;
        blk[0] += geo; blk[4] += geo; blk[8] += geo;
        real tmp[9], blkd[9];
        em_mulmat3(tmp, blk, xmat + 9*M.flexvert_bodyid[vj]);
        em_multmat3(blkd, xmat + 9*M.flexvert_bodyid[vi], tmp);
        for (int q = 0; q < 9; q++) eblk[144*t + 9*(4*i + j) + q] = blkd[q];
      }
    }
  }
  wv_sync();
  // ---- (b) every block of K: its contributions in the reference's order (bending: q I on the diagonal of the block)
  MJH_FOR_LANES(bk, s.nefmblk) {
    const int sl = M.efmblk_slot[bk], pos = M.efmblk_pos[bk];
    const int vi = M.efm_slotvert[sl];
    const int f = M.flexvert_flex[vi];
    const real scale = h*h + h*M.flex_damping[f];
    real acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (scale) {
      for (int a = M.efmblk_cadr[bk]; a < M.efmblk_cadr[bk + 1]; a++) {
        const int code = M.efmblk_c[a], ij = M.efmblk_cij[a];
        const int i = ij >> 2, j = ij & 3;
        if (code & 1) {
          const int t = code >> 1;
          for (int q = 0; q < 9; q++) acc[q] += eblk[144*t + 9*(4*i + j) + q];
        } else {
          const int ed = code >> 1;
          const real q = scale*M.flex_bending[M.flex_bendingadr[f] + 17*(ed - M.flex_edgeadr[f]) + 4*i + j];
          if (!q) continue;
          acc[0] += q; acc[4] += q; acc[8] += q;
        }
      }
    }
    const int d0 = M.body_dofadr[M.flexvert_bodyid[vi]];
    for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) Kv[M.efm_rowadr[d0 + k] + 3*pos + c] = acc[3*k + c];
  }
  wv_sync();
  // ---- (c) factored diagonal blocks of the flex part of M + K (effBlocks :3216; slider vertices: M's rows are diagonal)
  MJH_FOR_LANES(sl, s.nefmslot) {
    const int vi = M.efm_slotvert[sl];
    const int d0 = M.body_dofadr[M.flexvert_bodyid[vi]];
    const int pos = M.efm_slotdiag[sl];
    real Bk[9];
    for (int r = 0; r < 3; r++) {
      const int row = d0 + r;
      for (int c = 0; c < 3; c++) Bk[3*r + c] = 0;
      for (int a = M.M_rowadr[row]; a < M.M_rowadr[row] + M.M_rownnz[row]; a++) {
        const int c = M.M_colind[a];
        if (c >= d0 && c < d0 + 3) Bk[3*r + (c - d0)] += Ms[a];
      }
      // (K's row: the entries whose column falls into the triple are those of the diagonal block)
      for (int c = 0; c < 3; c++) Bk[3*r + c] += Kv[M.efm_rowadr[row] + 3*pos + c];
    }
    // mju_cholFactor(Bk, 3, mjMINVAL) (engine_util_solve.c:33)
    for (int j = 0; j < 3; j++) {
      real tmp = Bk[4*j];
      if (j) tmp -= dot_ref(Bk + 3*j, Bk + 3*j, j);
      const int deficient = tmp < MJH_MINVAL;
      if (deficient) tmp = MJH_MINVAL;
      Bk[4*j] = sqrt(tmp);
      if (deficient) { for (int i = j + 1; i < 3; i++) Bk[3*i + j] = 0; }
      else {
        tmp = 1/Bk[4*j];
        for (int i = j + 1; i < 3; i++) Bk[3*i + j] = (Bk[3*i + j] - dot_ref(Bk + 3*i, Bk + 3*j, j))*tmp;
      }
    }
    for (int q = 0; q < 9; q++) Lb[9*sl + q] = Bk[q];
  }
  }
  // ---- (d) the shift c = -h (K_bend + K_stretch) qvel through the stencil operators (mjd_effShift :3398)
  // per-edge bending terms (scale vl of the four flap vertices) and per-element stretch terms (rl of both ends of every
  // local edge); eblk is free again: the blocks above have been consumed
  wv_sync();
  crptr bterm = MJH_G(B, flexbend_frc, e);
  eff_bend_terms(M, B, e, qvel, -h, (real)0);
  MJH_FOR_LANES(t, s.nflexelem) {
    const int f = M.flexelem_flex[t];
    if (!em_flex_stretch(M, f)) continue;
    const real scale = -h + 0*M.flex_damping[f];
    EmElem E;
    em_element(M, f, t, vx, len, E);
    const int dim = M.flex_dim[f], nedge = E.nedge;
    real dw[6][3], g[6];
    for (int ed = 0; ed < nedge; ed++) {
      const int v0 = M.flexelem_vert[4*t + em_edge_corner(dim, ed, 0)], v1 = M.flexelem_vert[4*t + em_edge_corner(dim, ed, 1)];
      const int b0 = M.flexvert_bodyid[v0], b1 = M.flexvert_bodyid[v1];
      real w0[3] = {0, 0, 0}, w1[3] = {0, 0, 0};
      if (M.body_dofnum[b0]) { real qv[3] = {qvel[M.body_dofadr[b0]], qvel[M.body_dofadr[b0] + 1], qvel[M.body_dofadr[b0] + 2]}; m3_mulvec(w0, xmat + 9*b0, qv); }
      if (M.body_dofnum[b1]) { real qv[3] = {qvel[M.body_dofadr[b1]], qvel[M.body_dofadr[b1] + 1], qvel[M.body_dofadr[b1] + 2]}; m3_mulvec(w1, xmat + 9*b1, qv); }
      g[ed] = 0;
      for (int x = 0; x < 3; x++) { dw[ed][x] = w0[x] - w1[x]; g[ed] += E.dvec[ed][x]*dw[ed][x]; }
    }
    for (int ed = 0; ed < nedge; ed++) {
      real coef = 0;
      for (int a = 0; a < nedge; a++) coef += E.metric[nedge*ed + a]*g[a];
      coef *= 2*scale;
      real rw[3];
      for (int x = 0; x < 3; x++) rw[x] = coef*E.dvec[ed][x] + scale*E.Me[ed]*dw[ed][x];
      const int b0 = M.flexvert_bodyid[M.flexelem_vert[4*t + em_edge_corner(dim, ed, 0)]];
      const int b1 = M.flexvert_bodyid[M.flexelem_vert[4*t + em_edge_corner(dim, ed, 1)]];
      real r0[3], r1[3];
      m3_multvec(r0, xmat + 9*b0, rw);
      m3_multvec(r1, xmat + 9*b1, rw);
      for (int x = 0; x < 3; x++) { eblk[144*t + 6*ed + x] = r0[x]; eblk[144*t + 6*ed + 3 + x] = r1[x]; }
    }
  }
  wv_sync();
  MJH_FOR_LANES(j, s.nv) cs[j] = 0;
  wv_sync();
  MJH_FOR_LANES(v, s.nflexvert) {
    const int bid = M.flexvert_bodyid[v];
    if (!M.body_dofnum[bid]) continue;
    const int f = M.flexvert_flex[v];
    const int d0 = M.body_dofadr[bid];
    real acc[3] = {0, 0, 0};
    if (s.nflexbend && em_flex_bend(M, f)) {
      for (int a = M.flexvert_bendadr[v]; a < M.flexvert_bendadr[v + 1]; a++) {
        const int it = M.flexvert_bend[a];
        const int ed = it >> 2, i = it & 3;
        for (int x = 0; x < 3; x++) acc[x] += bterm[24*ed + 3*i + x];
      }
    }
    if (em_flex_stretch(M, f)) {
      const int dim = M.flex_dim[f], nedge = dim == 2 ? 3 : 6;
      for (int a = M.flexvert_elemadr[v]; a < M.flexvert_elemadr[v + 1]; a++) {
        const int it = M.flexvert_elem[a];
        const int t = it >> 2, i = it & 3;
        for (int ed = 0; ed < nedge; ed++) {
          if (em_edge_corner(dim, ed, 0) == i) { for (int x = 0; x < 3; x++) acc[x] += eblk[144*t + 6*ed + x]; }
          else if (em_edge_corner(dim, ed, 1) == i) { for (int x = 0; x < 3; x++) acc[x] -= eblk[144*t + 6*ed + 3 + x]; }
        }
      }
    }
    for (int x = 0; x < 3; x++) cs[d0 + x] = acc[x];
  }
  wv_sync();
}

// res += K vec (mjd_effMulAdd :3185: rows with stored entries only)
template <class RP, class VP>
MJH_DEV void eff_mul_add(MREF M, BREF B, int e, RP res, VP vec) {
  if (M.s.efm == 2) {
    // no assembled CSR: the bending stencil operator with scale h^2 + h damping (mjd_effMulAdd :3203)
    const real h = M.o.timestep;
    eff_bend_terms(M, B, e, vec, h*h, h);
    wv_sync();
    eff_bend_gather(M, B, e, res);
    wv_sync();
    return;
  }
  crptr Kv = MJH_G(B, efm_K_val, e);
  MJH_FOR_LANES(i, M.s.nv) {
    const int nnz = M.efm_rownnz[i];
    if (!nnz) continue;
    const int adr = M.efm_rowadr[i];
    res[i] += dot_sparse_ref(Kv + adr, vec, nnz, M.efm_colind + adr);
  }
  wv_sync();
}

// out = M v (mju_mulSymVecSparse: diagonal, own row right to left, then the column's entries by ascending row)
template <class RP, class VP>
MJH_DEV void eff_mul_M(MREF M, BREF B, int e, RP out, VP v) {
  crptr Ms = MJH_G(B, M, e);
  MJH_FOR_LANES(t, M.s.nv) {
    const int adr = M.M_rowadr[t], diag = M.M_rownnz[t] - 1;
    real acc = Ms[adr + diag]*v[t];
    for (int k = diag - 1; k >= 0; k--) acc += Ms[adr + k]*v[M.M_colind[adr + k]];
    for (int q = M.M_cscadr[t]; q < M.M_cscadr[t + 1]; q++) { const int a = M.M_cscind[q]; acc += Ms[a]*v[M.M_rowid[a]]; }
    out[t] = acc;
  }
  wv_sync();
}

// x = P b with the block preconditioner (effBlockApply :3262): the covered entries of the right-hand side are zeroed for
// the M^-1 sweep (covered and uncovered dofs must not see each other), then every covered triple is solved with its factor.
// rhs: nv reals of work space (b may alias x)
template <class XP, class BP, class WP>
MJH_DEV void eff_block_apply(MREF M, BREF B, int e, XP x, BP b, WP rhs) {
  const MJH_CONST_AS DSizes& s = M.s;
  crptr Lb = MJH_G(B, efm_L, e);
  MJH_FOR_LANES(i, s.nv) rhs[i] = b[i];
  wv_sync();
  if (s.efm == 2) {
    // bending only: M^-1 off the dofs the constant factor covers, the exact (M + K_bend)^-1 on them
    MJH_FOR_LANES(i, s.nv) x[i] = M.e0_cov[i] ? (real)0 : (real)rhs[i];
    wv_sync();
    solve_ld(M, x, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e));
    rptr z = MJH_G(B, efm_bz, e);
    // x <- L^-T x: row c is final once every row i > c that holds an entry in column c is; its subtractions in descending i
    for (int l = 0; l < s.ne0lev1; l++) {
      const int k0 = M.e0_l1adr[l], kn = M.e0_l1adr[l + 1] - k0;
      MJH_FOR_LANES(k, kn) {
        const int c = M.e0_l1row[k0 + k];
        real acc = rhs[M.e0_dof[c]];
        for (int q = M.e0_cscadr[c]; q < M.e0_cscadr[c + 1]; q++) {
          const real t = z[M.e0_cscrow[q]];
          if (t != 0) acc -= M.e0_L[M.e0_cscind[q]]*t;
        }
        if (acc != 0) acc /= M.e0_L[M.e0_rowadr[c] + M.e0_rownnz[c] - 1];
        z[c] = acc;
      }
      wv_sync();
    }
    // x <- L^-1 x: row i is final once the rows of its columns are
    for (int l = 0; l < s.ne0lev2; l++) {
      const int k0 = M.e0_l2adr[l], kn = M.e0_l2adr[l + 1] - k0;
      MJH_FOR_LANES(k, kn) {
        const int i = M.e0_l2row[k0 + k];
        const int adr = M.e0_rowadr[i], nnz = M.e0_rownnz[i];
        real v = z[i];
        if (nnz > 1) v -= dot_sparse_ref(M.e0_L + adr, z, nnz - 1, M.e0_colind + adr);
        v /= M.e0_L[adr + nnz - 1];
        z[i] = v;
      }
      wv_sync();
    }
    MJH_FOR_LANES(i, s.ne0) x[M.e0_dof[i]] = z[i];
    wv_sync();
    return;
  }
  MJH_FOR_LANES(i, s.nv) x[i] = M.efm_rownnz[i] ? (real)0 : (real)rhs[i];
  wv_sync();
  solve_ld(M, x, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e));
  MJH_FOR_LANES(sl, s.nefmslot) {
    const int d0 = M.body_dofadr[M.flexvert_bodyid[M.efm_slotvert[sl]]];
    auto L = Lb + 9*sl;
    // mju_cholSolve(x + d0, L, rhs + d0, 3) (engine_util_solve.c:84)
    real r[3] = {rhs[d0], rhs[d0 + 1], rhs[d0 + 2]};
    for (int i = 0; i < 3; i++) {
      if (i) { real dsum = 0; if (i == 1) dsum = (real)0 + L[3]*r[0]; else dsum = (real)0 + (L[6]*r[0] + L[7]*r[1]); r[i] -= dsum; }
      r[i] /= L[4*i];
    }
    for (int i = 2; i >= 0; i--) {
      for (int j = i + 1; j < 3; j++) r[i] -= L[3*j + i]*r[j];
      r[i] /= L[4*i];
    }
    x[d0] = r[0]; x[d0 + 1] = r[1]; x[d0 + 2] = r[2];
  }
  wv_sync();
}

// (M + K) x = b by PCG with the block preconditioner, relative residual to opt.tolerance (mjd_effSolve :3309).
// work: 5 nv reals.  Returns 1 when the iteration cap was reached with the residual above tolerance (mjWARN_INERTIA).
template <class XP, class BP>
MJH_DEV int eff_solve(MREF M, BREF B, int e, XP x, BP b) {
  const int nv = M.s.nv;
  rptr wk = MJH_G(B, efm_work, e);
  rptr r = wk, z = wk + nv, p = wk + 2*nv, Ap = wk + 3*nv, tmp = wk + 4*nv;
  MJH_FOR_LANES(i, nv) r[i] = b[i];
  wv_sync();
  MJH_FOR_LANES(i, nv) x[i] = 0;
  wv_sync();
  const real bn = wave_dot_ref(r, r, nv);
  int capped_out = 0;
  if (bn > MJH_MINVAL) {
    const real tolerance = M.o.tolerance;
    const real tol = tolerance*tolerance*bn;
    int capped = 1;
    eff_block_apply(M, B, e, z, r, tmp);
    MJH_FOR_LANES(i, nv) p[i] = z[i];
    wv_sync();
    real rz = wave_dot_ref(r, z, nv);
    for (int it = 0; it < M.o.iterations; it++) {
      eff_mul_M(M, B, e, Ap, p);
      eff_mul_add(M, B, e, Ap, p);
      const real pAp = wave_dot_ref(p, Ap, nv);
      if (pAp <= 0) { capped = 0; break; }
      const real alpha = rz/pAp;
      MJH_FOR_LANES(i, nv) { x[i] += p[i]*alpha; r[i] += Ap[i]*(-alpha); }
      wv_sync();
      if (wave_dot_ref(r, r, nv) < tol) { capped = 0; break; }
      eff_block_apply(M, B, e, z, r, tmp);
      const real rznew = wave_dot_ref(r, z, nv);
      const real beta = rznew/rz;
      MJH_FOR_LANES(i, nv) p[i] = z[i] + p[i]*beta;
      wv_sync();
      rz = rznew;
    }
    if (capped && wave_dot_ref(r, r, nv) >= tol) capped_out = 1;
  }
  return capped_out;
}

// qacc_smooth of the linearly-implicit dynamics (mj_fwdAcceleration, engine_forward.c:1033-1040); the iteration cap with
// the residual above tolerance raises mjWARN_INERTIA like the reference
MJH_DEVN void stage_eff_accel(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const int nv = M.s.nv;
  crptr qfs = MJH_F(B, qfrc_smooth, e);
  crptr cs = MJH_G(B, efm_c, e);
  rptr qeff = MJH_G(B, efm_work, e) + 5*nv;
  MJH_FOR_LANES(i, nv) qeff[i] = qfs[i] + cs[i];
  wv_sync();
  const int capped = eff_solve(M, B, e, MJH_F(B, qacc_smooth, e), qeff);
  if (capped && wv_lane() == 0) MJH_F(B, warning, e)[MJH_WARN_INERTIA]++;
  wv_sync();
}

#endif   // !MJH_LANE_MODE
