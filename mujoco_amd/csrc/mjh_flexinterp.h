// Interpolated flexes (flex_interp 1: trilinear, 2: triquadratic): the flex's NODES are bodies -- a grid of (order cx + 1) x
// (order cy + 1) x (order cz + 1) bodies with three sliders each -- and its vertices, which carry the collision elements, are
// interpolated from the (order + 1)^3 nodes of the cell that holds their parametric coordinate flex_vert0.
//   flex_interp_pos      node and vertex positions                 mj_flex, engine_core_smooth.c:580-626
//   flex_passive_interp  corotational cell elasticity              mj_flexPassiveInterp, engine_passive.c:62-213
//   flex_contact_nodes   the node bodies and weights of a contact  mj_vertBodyWeight, engine_core_constraint.c:265-384
// Shape functions, the cell lookup and the rotation extraction restate engine_util_misc.c:541-760 and mju_mat2Rot
// (engine_util_spatial.c:286) operation by operation: the sums of products below are in the reference's order.
// (included once per SPMD mode by mjh_stages.inc: no include guard)

// 1D shape function and its derivative (mju_flexPhi / mju_flexDphi, engine_util_misc.h:130-149)
MJH_DEV real interp_phi(real x, int i, int order) {
  if (order == 1) return i == 0 ? 1 - x : x;
  if (i == 0) return 2*x*x - 3*x + 1;
  if (i == 1) return 4*(x - x*x);
  return 2*x*x - x;
}
MJH_DEV real interp_dphi(real x, int i, int order) {
  if (order == 1) return i == 0 ? (real)-1 : (real)1;
  if (i == 0) return 4*x - 3;
  if (i == 1) return 4*(1 - 2*x);
  return 4*x - 1;
}

// the (order + 1)^3 basis values at local (mju_evalBasisArray :578)
MJH_DEV void interp_basis(real* basis, const real* x, int order) {
  real p[3][3];
  for (int d = 0; d < 3; d++)
    for (int i = 0; i <= order; i++) p[d][i] = order == 1 ? (i == 0 ? 1 - x[d] : x[d]) : interp_phi(x[d], i, 2);
  int j = 0;
  for (int i0 = 0; i0 <= order; i0++) {
    const real w0 = p[0][i0];
    for (int i1 = 0; i1 <= order; i1++) {
      const real w01 = w0 * p[1][i1];
      for (int i2 = 0; i2 <= order; i2++) basis[j++] = w01 * p[2][i2];
    }
  }
}

// the cell of parametric coordinate coord: local coordinates in it and the (flex-local) indices of its nodes (mju_cellLookup :627)
MJH_DEV void interp_cell_lookup(MREF M, int f, const real* coord, real* local, int* nodeidx) {
  const int order = M.flex_interp[f];
  const int cx = M.flex_cellnum[3*f], cy = M.flex_cellnum[3*f + 1], cz = M.flex_cellnum[3*f + 2];
  int ci = (int)floor(coord[0]*cx), cj = (int)floor(coord[1]*cy), ck = (int)floor(coord[2]*cz);
  ci = ci < cx - 1 ? ci : cx - 1; ci = ci > 0 ? ci : 0;
  cj = cj < cy - 1 ? cj : cy - 1; cj = cj > 0 ? cj : 0;
  ck = ck < cz - 1 ? ck : cz - 1; ck = ck > 0 ? ck : 0;
  local[0] = r_clip(coord[0]*cx - ci, 0, 1);
  local[1] = r_clip(coord[1]*cy - cj, 0, 1);
  local[2] = r_clip(coord[2]*cz - ck, 0, 1);
  const int ny_g = cy*order + 1, nz_g = cz*order + 1;
  int ni = 0;
  for (int li = 0; li <= order; li++)
    for (int lj = 0; lj <= order; lj++)
      for (int lk = 0; lk <= order; lk++)
        nodeidx[ni++] = (ci*order + li)*ny_g*nz_g + (cj*order + lj)*nz_g + (ck*order + lk);
}

// node positions (body frame offset rotated into the world, or the body origin), then every vertex as the basis-weighted sum
// of its cell's nodes in node order, starting from zero (mju_interpolate3D :672)
// (out of line, like flex_passive_interp and flex_contact_nodes: their 27-entry basis / index arrays otherwise sit in the
//  frame of every kernel that inlines them -- round 6's bisect put 3 % of the jelly benchmark, a model without
//  interpolated flexes, on the commit that added these routines: profiles/r06/flex_bisect.txt)
MJH_DEVN_HOT void flex_interp_pos(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr xpos = MJH_F(B, xpos, e);
  crptr xmat = MJH_F(B, xmat, e);
  rptr nx = MJH_F(B, flexnode_xpos, e);
  rptr vx = MJH_F(B, flexvert_xpos, e);
  MJH_FOR_LANES(i, s.nflexnode) {
    const int b = M.flexnode_bodyid[i];
    auto ln = M.flex_node + 3*i;
    if (ln[0] == 0 && ln[1] == 0 && ln[2] == 0) {
      nx[3*i] = xpos[3*b]; nx[3*i + 1] = xpos[3*b + 1]; nx[3*i + 2] = xpos[3*b + 2];
    } else {
      real l[3] = {ln[0], ln[1], ln[2]}, r[3];
      m3_mulvec(r, xmat + 9*b, l);
      nx[3*i] = r[0] + xpos[3*b]; nx[3*i + 1] = r[1] + xpos[3*b + 1]; nx[3*i + 2] = r[2] + xpos[3*b + 2];
    }
  }
  wv_sync();
  MJH_FOR_LANES(v, s.nflexvert) {
    const int f = M.flexvert_flex[v];
    const int order = M.flex_interp[f];
    if (!order) continue;
    const int npc = (order + 1)*(order + 1)*(order + 1), na = M.flex_nodeadr[f];
    real coord[3] = {M.flex_vert0[3*v], M.flex_vert0[3*v + 1], M.flex_vert0[3*v + 2]}, local[3], basis[27];
    int idx[27];
    interp_cell_lookup(M, f, coord, local, idx);
    interp_basis(basis, local, order);
    real r[3] = {0, 0, 0};
    for (int j = 0; j < npc; j++) {
      const int n = na + idx[j];
      r[0] += nx[3*n]*basis[j]; r[1] += nx[3*n + 1]*basis[j]; r[2] += nx[3*n + 2]*basis[j];
    }
    vx[3*v] = r[0]; vx[3*v + 1] = r[1]; vx[3*v + 2] = r[2];
  }
}

// rotation of a cell: the deformation gradient at the cell centre (mju_defGradient :541), its rotational part by the
// iteration of Mueller et al. from the identity (mju_mat2Rot: at most 500 steps, stop below 1e-9), conjugated
// (flexInterpRotation :693)
MJH_DEV void interp_cell_rotation(const real* xc, int order, real* quat) {
  real mat[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const real p = 0.5;
  int idx = 0;
  for (int i = 0; i <= order; i++)
    for (int j = 0; j <= order; j++)
      for (int k = 0; k <= order; k++) {
        const real g0 = interp_dphi(p, i, order) * interp_phi(p, j, order) * interp_phi(p, k, order);
        const real g1 = interp_phi(p, i, order) * interp_dphi(p, j, order) * interp_phi(p, k, order);
        const real g2 = interp_phi(p, i, order) * interp_phi(p, j, order) * interp_dphi(p, k, order);
        mat[0] += xc[3*idx]*g0;     mat[1] += xc[3*idx]*g1;     mat[2] += xc[3*idx]*g2;
        mat[3] += xc[3*idx + 1]*g0; mat[4] += xc[3*idx + 1]*g1; mat[5] += xc[3*idx + 1]*g2;
        mat[6] += xc[3*idx + 2]*g0; mat[7] += xc[3*idx + 2]*g1; mat[8] += xc[3*idx + 2]*g2;
        idx++;
      }
  quat[0] = 1; quat[1] = 0; quat[2] = 0; quat[3] = 0;
  const real c1[3] = {mat[0], mat[3], mat[6]}, c2[3] = {mat[1], mat[4], mat[7]}, c3[3] = {mat[2], mat[5], mat[8]};
  for (int iter = 0; iter < 500; iter++) {
    real rot[9];
    q_tomat(rot, quat);
    const real r1[3] = {rot[0], rot[3], rot[6]}, r2[3] = {rot[1], rot[4], rot[7]}, r3[3] = {rot[2], rot[5], rot[8]};
    real omega[3], v1[3], v2[3], v3[3];
    v3_cross(v1, r1, c1);
    v3_cross(v2, r2, c2);
    v3_cross(v3, r3, c3);
    v3_add(omega, v1, v2);
    v3_addto(omega, v3);
    const real sc = 1.0 / (fabs(v3_dot(r1, c1) + v3_dot(r2, c2) + v3_dot(r3, c3)) + MJH_MINVAL);
    omega[0] *= sc; omega[1] *= sc; omega[2] *= sc;
    const real w = v3_normalize(omega);
    if (w < 1e-9) break;
    real qrot[4];
    q_axisangle(qrot, omega, w);
    q_mul(quat, qrot, quat);
    q_normalize(quat);
  }
  quat[1] = -quat[1]; quat[2] = -quat[2]; quat[3] = -quat[3];
}

// mju_rotVecQuat (engine_util_spatial.c:28: a zero vector stays zero, res may alias vec)
MJH_DEV void interp_rotvec(real* r, const real* v, const real* q) {
  if (v[0] == 0 && v[1] == 0 && v[2] == 0) { r[0] = 0; r[1] = 0; r[2] = 0; return; }
  real t[3];
  q_rotvec(t, v, q);
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
}

// Passive forces of the cells.  Node velocities (mju_flexGatherState, engine_core_util.c:1022: the body's velocity at its
// centre of mass -- mj_objectVelocity(mjOBJ_BODY) -- carried to the node); per cell one lane gathers the nodes, finds the
// rotation, rotates positions and velocities into the cell frame and forms the displacement from the rest positions; one lane
// per ROW of the cell's stiffness matrix takes the two products K displ and K vel (mju_mulMatVec: a 4-accumulator dot per row);
// per node one lane rotates the cells' forces back and adds them in the order the reference's cell loop scatters them, scales
// the damper force and writes the node body's three dofs (xmat' f).
MJH_DEVN_HOT void flex_passive_interp(MREF M_, BREF B_, int e_, int enbl_spring, int enbl_damper) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  if (!s.nflexcell) return;
  crptr xmat = MJH_F(B, xmat, e);
  crptr xipos = MJH_F(B, xipos, e);
  crptr scom = MJH_F(B, subtree_com, e);
  crptr cvel = MJH_F(B, cvel, e);
  crptr nx = MJH_F(B, flexnode_xpos, e);
  rptr nvl = MJH_F(B, flexnode_vel, e);
  rptr cin = MJH_F(B, flexcell_in, e);
  rptr cfrc = MJH_F(B, flexcell_frc, e);
  rptr fs = MJH_F(B, qfrc_spring, e);
  rptr fd = MJH_F(B, qfrc_damper, e);
  MJH_FOR_LANES(i, s.nflexnode) {
    const int b = M.flexnode_bodyid[i];
    // (mju_transformSpatial of cvel from the subtree centre of mass to the body's centre of mass, then omega x (node - com))
    real dif[3], cros[3], lin[3], r[3], cr2[3];
    v3_sub(dif, xipos + 3*b, scom + 3*M.body_rootid[b]);
    const real ang[3] = {cvel[6*b], cvel[6*b + 1], cvel[6*b + 2]};
    v3_cross(cros, dif, ang);
    lin[0] = cvel[6*b + 3] - cros[0]; lin[1] = cvel[6*b + 4] - cros[1]; lin[2] = cvel[6*b + 5] - cros[2];
    v3_sub(r, nx + 3*i, xipos + 3*b);
    v3_cross(cr2, ang, r);
    nvl[3*i] = lin[0] + cr2[0]; nvl[3*i + 1] = lin[1] + cr2[1]; nvl[3*i + 2] = lin[2] + cr2[2];
  }
  wv_sync();
  MJH_FOR_LANES(c, s.nflexcell) {
    if (M.flexcell_kadr[c] < 0) continue;
    const int f = M.flexcell_flex[c];
    const int order = M.flex_interp[f];
    const int npe = (order + 1)*(order + 1)*(order + 1);
    real xc[81], quat[4];
    for (int n = 0; n < npe; n++) {
      const int g = M.flexcell_node[27*c + n];
      xc[3*n] = nx[3*g]; xc[3*n + 1] = nx[3*g + 1]; xc[3*n + 2] = nx[3*g + 2];
    }
    interp_cell_rotation(xc, order, quat);
    for (int n = 0; n < npe; n++) {
      const int g = M.flexcell_node[27*c + n];
      real xr[3], vr[3];
      const real vg[3] = {nvl[3*g], nvl[3*g + 1], nvl[3*g + 2]};
      interp_rotvec(xr, xc + 3*n, quat);
      interp_rotvec(vr, vg, quat);
      for (int x = 0; x < 3; x++) {
        cin[166*c + 3*n + x] = xr[x] + M.flex_node0[3*g + x]*(real)-1;
        cin[166*c + 81 + 3*n + x] = vr[x];
      }
    }
    cin[166*c + 162] = quat[0]; cin[166*c + 163] = -quat[1]; cin[166*c + 164] = -quat[2]; cin[166*c + 165] = -quat[3];
  }
  wv_sync();
  MJH_FOR_LANES(it, 162*s.nflexcell) {
    const int c = it/162, rr = it % 162, which = rr >= 81, row = which ? rr - 81 : rr;
    const int kadr = M.flexcell_kadr[c];
    if (kadr < 0) continue;
    const int order = M.flex_interp[M.flexcell_flex[c]];
    const int n3 = 3*(order + 1)*(order + 1)*(order + 1);
    if (row >= n3 || (which ? !enbl_damper : !enbl_spring)) continue;
    cfrc[162*c + rr] = dot_ref(M.flex_stiffness + kadr + row*n3, cin + 166*c + 81*which, n3);
  }
  wv_sync();
  MJH_FOR_LANES(i, s.nflexnode) {
    const int f = M.flexnode_flex[i];
    if (!M.flex_interp[f]) continue;
    real fg[3] = {0, 0, 0}, dg[3] = {0, 0, 0};
    for (int a = M.flexnode_celladr[i]; a < M.flexnode_celladr[i + 1]; a++) {
      const int c = M.flexnode_cell[a] >> 5, n = M.flexnode_cell[a] & 31;
      const real q[4] = {cin[166*c + 162], cin[166*c + 163], cin[166*c + 164], cin[166*c + 165]};
      if (enbl_spring) {
        const real fe[3] = {cfrc[162*c + 3*n], cfrc[162*c + 3*n + 1], cfrc[162*c + 3*n + 2]};
        real t[3];
        q_rotvec(t, fe, q);
        fg[0] += t[0]; fg[1] += t[1]; fg[2] += t[2];
      }
      if (enbl_damper) {
        const real de[3] = {cfrc[162*c + 81 + 3*n], cfrc[162*c + 81 + 3*n + 1], cfrc[162*c + 81 + 3*n + 2]};
        real t[3];
        q_rotvec(t, de, q);
        dg[0] += t[0]; dg[1] += t[1]; dg[2] += t[2];
      }
    }
    const real damp = M.flex_damping[f];
    dg[0] *= damp; dg[1] *= damp; dg[2] *= damp;
    const int b = M.flexnode_bodyid[i];
    if (M.body_dofnum[b] != 3) continue;        // (a node fixed to the world: mj_applyFT on the world body)
    const int dadr = M.body_dofadr[b];
    real ql[3];
    if (enbl_spring) { m3_multvec(ql, xmat + 9*b, fg); for (int x = 0; x < 3; x++) fs[dadr + x] += ql[x]; }
    if (enbl_damper) { m3_multvec(ql, xmat + 9*b, dg); for (int x = 0; x < 3; x++) fd[dadr + x] += ql[x]; }
  }
  wv_sync();
}
