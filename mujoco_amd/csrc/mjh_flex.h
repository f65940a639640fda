// Flexes: vertex-based deformable objects (lines, triangle shells, tetrahedral solids).
//
//   stage_flex_pos     vertex positions                                  mj_flex, engine_core_smooth.c:558-575
//   stage_flex_edges   edge lengths and the sparse edge Jacobian         mj_flex, engine_core_smooth.c:700-745
//   flex_edge_velocity flexedge_velocity = flexedge_J qvel               mj_fwdVelocity, engine_forward.c:188-197
//   flex_passive       bending, stretch (finite-element) and edge spring-damper forces
//                                                                        engine_passive.c:459-649, :739-787
//
// Mapping.  One wavefront owns the environment; its lanes take vertices, edges or elements in turn.  The
// reference accumulates the forces of all elements / edges into shared dof slots in element / edge order;
// here every element (edge) writes its own force block, and every vertex (dof) then sums the blocks that
// touch it in that same order through the gather tables of the model (mjh_model_build.h), so the sums carry
// the reference's rounding without a serial pass.
//
// Scope (checked by the model build): vertex flexes (flex_interp 0) whose vertices are bodies with three
// axis-aligned sliders (body_simple 2) or are pinned to the static world; no flex equality constraints.
// (included once per SPMD mode by mjh_stages.inc: no include guard)

// ------------------------------------------------------------------------------------------------
// vertex positions                                     (mj_flex, engine_core_smooth.c:558-575)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_flex_pos(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr xpos = MJH_F(B, xpos, e);
  crptr xmat = MJH_F(B, xmat, e);
  rptr vx = MJH_F(B, flexvert_xpos, e);
  if (s.nflexnode) flex_interp_pos(M, B, e);
  MJH_FOR_LANES(v, s.nflexvert) {
    const int b = M.flexvert_bodyid[v];
    if (b < 0) continue;                 // (a vertex of an interpolated flex: placed by flex_interp_pos)
    auto lv = M.flex_vert + 3*v;
    if (lv[0] == 0 && lv[1] == 0 && lv[2] == 0) {
      vx[3*v] = xpos[3*b]; vx[3*v + 1] = xpos[3*b + 1]; vx[3*v + 2] = xpos[3*b + 2];
    } else {
      real l[3] = {lv[0], lv[1], lv[2]}, r[3];
      m3_mulvec(r, xmat + 9*b, l);
      vx[3*v] = r[0] + xpos[3*b]; vx[3*v + 1] = r[1] + xpos[3*b + 1]; vx[3*v + 2] = r[2] + xpos[3*b + 2];
    }
  }
  wv_sync();
  // element bounding boxes: centre and half sizes, inflated by the flex radius (mj_flex :630-660)
  if (s.nflexpair) {
    rptr aabb = MJH_F(B, flexelem_aabb, e);
    MJH_FOR_LANES(t, s.nflexelem) {
      const int f = M.flexelem_flex[t];
      const int dim = M.flex_dim[f];
      const real radius = M.flex_radius[f];
      const int v0 = M.flexelem_vert[4*t];
      real lo[3] = {vx[3*v0], vx[3*v0 + 1], vx[3*v0 + 2]}, hi[3] = {lo[0], lo[1], lo[2]};
      for (int i = 1; i <= dim; i++) {
        const int v = M.flexelem_vert[4*t + i];
        for (int j = 0; j < 3; j++) { const real x = vx[3*v + j]; lo[j] = r_min(lo[j], x); hi[j] = r_max(hi[j], x); }
      }
      for (int j = 0; j < 3; j++) {
        aabb[6*t + j] = 0.5*(hi[j] + lo[j]);
        aabb[6*t + 3 + j] = 0.5*(hi[j] - lo[j]) + radius;
      }
    }
    wv_sync();
  }
  // the flexes' bounding volume hierarchies, recomputed bottom-up from the element boxes exactly as mj_updateDynamicBVH
  // does (engine_core_smooth.c:490-533: an inner box is the union of its children's boxes, converted back to centre /
  // half size at every level): the sweep axis of a self-collision is the longest side of the ROOT box
  if (s.nflexbvh) {
    crptr aabb = MJH_F(B, flexelem_aabb, e);
    rptr bb = MJH_G(B, flexbvh_aabb, e);
    MJH_FOR_LANES(i, s.nflexbvh) {
      const int el = M.flexbvh_elem[i];
      if (el >= 0) for (int q = 0; q < 6; q++) bb[6*i + q] = aabb[6*el + q];
    }
    wv_sync();
    for (int h = 1; h < s.nflexbvhh; h++) {
      for (int t = M.flexbvh_hadr[h] + wv_lane(); t < M.flexbvh_hadr[h + 1]; t += MJH_W) {
        const int i = M.flexbvh_order[t];
        const int c1 = M.flexbvh_child[2*i], c2 = M.flexbvh_child[2*i + 1];
        for (int k = 0; k < 3; k++) {
          const real lo = r_min(bb[6*c1 + k] - bb[6*c1 + k + 3], bb[6*c2 + k] - bb[6*c2 + k + 3]);
          const real hi = r_max(bb[6*c1 + k] + bb[6*c1 + k + 3], bb[6*c2 + k] + bb[6*c2 + k + 3]);
          bb[6*i + k] = 0.5*(hi + lo);
          bb[6*i + k + 3] = 0.5*(hi - lo);
        }
      }
      wv_sync();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// vertex constraints of shell flexes with edge equality "vert" (mj_flex, engine_core_smooth.c:745-918; Chen, Kry, Vouga:
// "Locking-free Simulation of Isometric Thin Plates"): per vertex the mass-weighted deformation gradient over its edge
// fan, the two invariants of its Cauchy strain as constraint residuals (flexvert_length) and their Jacobian rows
// (flexvert_J).  One lane per vertex.  The reference scatters every edge's end-point Jacobians into two dense rows
// (J0_dense / J1_dense: first body's chain, then the second's, edge after edge) and compresses them at the end; here an
// entry of the row accumulates the same terms in the same order in place -- for every edge of the fan, every entry
// whose dof lies on the first / second end body's chain (body_dofanc) takes that body's point-Jacobian column
// (cdof_lin + cdof_ang x (pos - subtree_com[root]), mj_jacSparse) against the edge's derivative vectors
// (mju_mulMatTVec: three rows in order, zero components skipped) -- and is scaled by sqrt(mass) at the end.
// ------------------------------------------------------------------------------------------------
MJH_DEV void flex_vert_rows(MREF M, BREF B, int e) {
  const MJH_CONST_AS DSizes& s = M.s;
  if (!s.nfv) return;
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr cdof = MJH_F(B, cdof, e);
  crptr scom = MJH_F(B, subtree_com, e);
  rptr vlen = MJH_G(B, flexvert_length, e);
  rptr Jv = MJH_G(B, flexvert_J, e);
  MJH_FOR_LANES(v, s.nflexvert) {
    const int f = M.flexvert_flex[v];
    const int a0 = M.fv_rowadr[2*v], a1 = M.fv_rowadr[2*v + 1], nnz = M.fv_rownnz[2*v];
    for (int j = 0; j < nnz; j++) { Jv[a0 + j] = 0; Jv[a1 + j] = 0; }
    if (M.flex_edgeequality[f] != 2) { vlen[2*v] = 0; vlen[2*v + 1] = 0; continue; }
    const int ebase = M.flex_edgeadr[f];
    const int ne = M.fv_edgenum[v], ea = M.fv_edgeadr[v];
    auto metric = M.fv_metric + 4*v;
    auto mul322 = [](real* C, const real* A, auto Bm) {
      C[0] = A[0]*Bm[0] + A[1]*Bm[2]; C[1] = A[0]*Bm[1] + A[1]*Bm[3];
      C[2] = A[2]*Bm[0] + A[3]*Bm[2]; C[3] = A[2]*Bm[1] + A[3]*Bm[3];
      C[4] = A[4]*Bm[0] + A[5]*Bm[2]; C[5] = A[4]*Bm[1] + A[5]*Bm[3];
    };
    auto edge_weight = [&](int v1, int v2) -> real {
      const int bn = M.flexvert_bodyid[v == v1 ? v2 : v1];
      real w = 1;
      if (bn >= 0) { w = M.body_mass[bn]; if (w < MJH_MINVAL) w = MJH_MINVAL; }
      return w;
    };
    real A[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < ne; k++) {
      const int ge = ebase + M.fv_edge[ea + k];
      const int v1 = M.flexedge_vert[2*ge], v2 = M.flexedge_vert[2*ge + 1];
      auto dx = M.fv_dx + 3*ge;
      const real dy[3] = {vx[3*v2] - vx[3*v1], vx[3*v2 + 1] - vx[3*v1 + 1], vx[3*v2 + 2] - vx[3*v1 + 2]};
      const real w = edge_weight(v1, v2);
      A[0] += w*dy[0]*dx[0]; A[1] += w*dy[0]*dx[1];
      A[2] += w*dy[1]*dx[0]; A[3] += w*dy[1]*dx[1];
      A[4] += w*dy[2]*dx[0]; A[5] += w*dy[2]*dx[1];
    }
    real F[6], cauchy[4];
    mul322(F, A, metric);
    cauchy[0] = F[0]*F[0] + F[2]*F[2] + F[4]*F[4];
    cauchy[1] = F[0]*F[1] + F[2]*F[3] + F[4]*F[5];
    cauchy[2] = F[1]*F[0] + F[3]*F[2] + F[5]*F[4];
    cauchy[3] = F[1]*F[1] + F[3]*F[3] + F[5]*F[5];
    real scale = 1;
    {
      const int b = M.flexvert_bodyid[v];
      if (b >= 0) { const real mass = M.body_mass[b]; if (mass > MJH_MINVAL) scale = sqrt(mass); }
    }
    vlen[2*v] = (cauchy[0] + cauchy[3] - 2)*scale;
    vlen[2*v + 1] = (cauchy[0]*cauchy[3] - cauchy[1]*cauchy[2] - 1)*scale;
    real FB[6], Fadj[6], FadjBinv[6];
    const real adj[4] = {cauchy[3], -cauchy[1], -cauchy[2], cauchy[0]};
    mul322(FB, F, metric);
    mul322(Fadj, F, adj);
    mul322(FadjBinv, Fadj, metric);
    for (int k = 0; k < ne; k++) {
      const int ge = ebase + M.fv_edge[ea + k];
      const int v1 = M.flexedge_vert[2*ge], v2 = M.flexedge_vert[2*ge + 1];
      auto dx = M.fv_dx + 3*ge;
      const real w = edge_weight(v1, v2);
      real d1a[3], d1b[3], d2a[3], d2b[3];        // dI1/dy1, dI1/dy2, dI2/dy1, dI2/dy2
      for (int r = 0; r < 3; r++) {
        const real t1 = dot_ref(FB + 2*r, dx, 2), t2 = dot_ref(FadjBinv + 2*r, dx, 2);
        d1a[r] = t1*(-2*w); d1b[r] = d1a[r]*(real)(-1);
        d2a[r] = t2*(-2*w); d2b[r] = d2a[r]*(real)(-1);
      }
      const int b1 = M.flexvert_bodyid[v1], b2 = M.flexvert_bodyid[v2];
      const int w1 = M.body_weldid[b1], w2 = M.body_weldid[b2];
      real off1[3], off2[3];
      v3_sub(off1, vx + 3*v1, scom + 3*M.body_rootid[b1]);
      v3_sub(off2, vx + 3*v2, scom + 3*M.body_rootid[b2]);
      for (int j = 0; j < nnz; j++) {
        const int col = M.fv_colind[a0 + j];
        const int in1 = (M.body_dofanc[w1*s.nvw + (col >> 5)] >> (col & 31)) & 1;
        const int in2 = (M.body_dofanc[w2*s.nvw + (col >> 5)] >> (col & 31)) & 1;
        if (!in1 && !in2) continue;
        crptr cd = cdof + 6*col;
        real r0 = Jv[a0 + j], r1 = Jv[a1 + j];
        auto add = [&](const real* off, const real* da, const real* db) {
          real t[3];
          v3_cross(t, cd, off);
          const real jc[3] = {cd[3] + t[0], cd[4] + t[1], cd[5] + t[2]};
          real q0 = 0, q1 = 0;
          for (int r = 0; r < 3; r++) { if (da[r] != 0) q0 += jc[r]*da[r]; }
          for (int r = 0; r < 3; r++) { if (db[r] != 0) q1 += jc[r]*db[r]; }
          r0 += q0; r1 += q1;
        };
        if (in1) add(off1, d1a, d2a);
        if (in2) add(off2, d1b, d2b);
        Jv[a0 + j] = r0; Jv[a1 + j] = r1;
      }
    }
    for (int j = 0; j < nnz; j++) { Jv[a0 + j] = (real)0 + Jv[a0 + j]*scale; Jv[a1 + j] = (real)0 + Jv[a1 + j]*scale; }
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// edge lengths and Jacobians                           (mj_flex, engine_core_smooth.c:700-745)
// One lane per edge.  Row e of flexedge_J holds, for every dof of the two end bodies (the model's
// flexedge_J_colind), vec' (jac2 - jac1) with vec the unit vector from vertex 1 to vertex 2: the point
// Jacobian column of a dof is cdof_lin + cdof_ang x (pos - subtree_com[root]) (mj_jacSparseSimple,
// engine_core_util.c:375-433), entered with a minus sign for the first body; the product with vec adds the
// three rows in order and skips zero components of vec (mju_mulMatTVec, engine_util_blas.c:554-563).
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_flex_edges(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr cdof = MJH_F(B, cdof, e);
  crptr scom = MJH_F(B, subtree_com, e);
  rptr len = MJH_F(B, flexedge_length, e);
  rptr J = MJH_F(B, flexedge_J, e);
  // (a model of one deformable flex whose edges all carry six entries -- both ends slider bodies, jelly.xml: 2863 edges, 45
  // per lane -- takes its edges two at a time per lane in a branch-free body, for the reason given at the stretch pass: the
  // second edge's chain of dependent loads hides the first one's round trips)
  int paired = 0;
  if (s.nflex == 1 && s.flex_sliders && !M.flex_rigid[0] && !M.flex_interp[0] && s.nJfe == 6*s.nflexedge &&
      !(M.flex_edgeequality[0] != 1 && M.flex_edgedamping[0] == 0 && M.flex_edgestiffness[0] == 0 && M.flex_damping[0] == 0)) paired = 1;
  if (paired) {
    const int ne = s.nflexedge;
    auto edge = [&](int ed, real& length, real* Jr) {
      const int v1 = M.flexedge_vert[2*ed], v2 = M.flexedge_vert[2*ed + 1];
      const int b1 = M.flexvert_bodyid[v1], b2 = M.flexvert_bodyid[v2];
      real vec[3] = {vx[3*v2] - vx[3*v1], vx[3*v2 + 1] - vx[3*v1 + 1], vx[3*v2 + 2] - vx[3*v1 + 2]};
      length = v3_normalize(vec);
      real off1[3], off2[3];
      v3_sub(off1, vx + 3*v1, scom + 3*M.body_rootid[b1]);
      v3_sub(off2, vx + 3*v2, scom + 3*M.body_rootid[b2]);
      const int adr = M.flexedge_J_rowadr[ed];
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int col = M.flexedge_J_colind[adr + q];
        const int second = M.dof_bodyid[col] == b2;
        crptr cd = cdof + 6*col;
        real t[3];
        v3_cross(t, cd, second ? off2 : off1);
        real jd[3] = {cd[3] + t[0], cd[4] + t[1], cd[5] + t[2]};
        if (!second) { jd[0] = -jd[0]; jd[1] = -jd[1]; jd[2] = -jd[2]; }
        real acc = 0;
        for (int r = 0; r < 3; r++)
          if (vec[r] != 0) acc += jd[r]*vec[r];
        Jr[q] = acc;
      }
    };
    for (int e0 = wv_lane(); e0 < ne; e0 += 2*MJH_W) {
      const int e1 = e0 + MJH_W < ne ? e0 + MJH_W : e0;
      real la, lb, Ja[6], Jb[6];
      edge(e0, la, Ja);
      edge(e1, lb, Jb);
      const int a0 = M.flexedge_J_rowadr[e0], a1 = M.flexedge_J_rowadr[e1];
      len[e0] = la;
#pragma unroll
      for (int q = 0; q < 6; q++) J[a0 + q] = Ja[q];
      if (e1 != e0) {
        len[e1] = lb;
#pragma unroll
        for (int q = 0; q < 6; q++) J[a1 + q] = Jb[q];
      }
    }
    wv_sync();
    flex_vert_rows(M, B, e);
    return;
  }
  MJH_FOR_LANES(ed, s.nflexedge) {
    const int f = M.flexedge_flex[ed];
    const int adr = M.flexedge_J_rowadr[ed], nnz = M.flexedge_J_rownnz[ed];
    if (M.flex_rigid[f] || M.flex_interp[f]) {   // (rigid and interpolated flexes: no edge forces; lengths stay zero, Jacobian cleared)
      len[ed] = 0;
      for (int j = adr; j < adr + nnz; j++) J[j] = 0;
      continue;
    }
    const int v1 = M.flexedge_vert[2*ed], v2 = M.flexedge_vert[2*ed + 1];
    const int b1 = M.flexvert_bodyid[v1], b2 = M.flexvert_bodyid[v2];
    real vec[3] = {vx[3*v2] - vx[3*v1], vx[3*v2 + 1] - vx[3*v1 + 1], vx[3*v2 + 2] - vx[3*v1 + 2]};
    len[ed] = v3_normalize(vec);
    const int skipjac = M.flex_edgeequality[f] != 1 && M.flex_edgedamping[f] == 0 && M.flex_edgestiffness[f] == 0 && M.flex_damping[f] == 0;
    if (skipjac) {
      for (int j = adr; j < adr + nnz; j++) J[j] = 0;
      continue;
    }
    real off1[3], off2[3];
    v3_sub(off1, vx + 3*v1, scom + 3*M.body_rootid[b1]);
    v3_sub(off2, vx + 3*v2, scom + 3*M.body_rootid[b2]);
    const int w1 = M.body_weldid[b1], w2 = M.body_weldid[b2];
    const int both_simple = M.body_simple[b1] && M.body_simple[b2];
    for (int j = adr; j < adr + nnz; j++) {
      const int col = M.flexedge_J_colind[j];
      crptr cd = cdof + 6*col;
      real jd[3];
      if (both_simple) {
        // (mj_jacSparseSimple: a dof belongs to exactly one of the two bodies; the first body's block is written negated)
        const int second = M.dof_bodyid[col] == b2;
        real t[3];
        v3_cross(t, cd, second ? off2 : off1);
        jd[0] = cd[3] + t[0]; jd[1] = cd[4] + t[1]; jd[2] = cd[5] + t[2];
        if (!second) { jd[0] = -jd[0]; jd[1] = -jd[1]; jd[2] = -jd[2]; }
      } else {
        // (mj_jacDifPair over the merged chain, common dofs kept: jac2 - jac1, each zero off its body's chain)
        const int in1 = (M.body_dofanc[w1*s.nvw + (col >> 5)] >> (col & 31)) & 1;
        const int in2 = (M.body_dofanc[w2*s.nvw + (col >> 5)] >> (col & 31)) & 1;
        real j1[3] = {0, 0, 0}, j2[3] = {0, 0, 0}, t[3];
        if (in1) { v3_cross(t, cd, off1); j1[0] = cd[3] + t[0]; j1[1] = cd[4] + t[1]; j1[2] = cd[5] + t[2]; }
        if (in2) { v3_cross(t, cd, off2); j2[0] = cd[3] + t[0]; j2[1] = cd[4] + t[1]; j2[2] = cd[5] + t[2]; }
        jd[0] = j2[0] - j1[0]; jd[1] = j2[1] - j1[1]; jd[2] = j2[2] - j1[2];
      }
      real acc = 0;
      for (int r = 0; r < 3; r++)
        if (vec[r] != 0) acc += jd[r]*vec[r];
      J[j] = acc;
    }
  }
  wv_sync();
  flex_vert_rows(M, B, e);
}

// flexedge_velocity = flexedge_J qvel                  (mj_fwdVelocity, engine_forward.c:188-197)
MJH_DEV void flex_edge_velocity(MREF M, BREF B, int e) {
  const MJH_CONST_AS DSizes& s = M.s;
  crptr qvel = MJH_F(B, qvel, e);
  crptr J = MJH_F(B, flexedge_J, e);
  rptr vel = MJH_F(B, flexedge_velocity, e);
  if (s.nflex == 1 && s.flex_sliders && !M.flex_rigid[0] && s.nJfe == 6*s.nflexedge) {
    // (rows of six entries, two edges per lane at a time: see stage_flex_edges)
    const int ne = s.nflexedge;
    for (int e0 = wv_lane(); e0 < ne; e0 += 2*MJH_W) {
      const int e1 = e0 + MJH_W < ne ? e0 + MJH_W : e0;
      const int a0 = M.flexedge_J_rowadr[e0], a1 = M.flexedge_J_rowadr[e1];
      const real r0 = dot_sparse_ref(J + a0, qvel, 6, M.flexedge_J_colind + a0);
      const real r1 = dot_sparse_ref(J + a1, qvel, 6, M.flexedge_J_colind + a1);
      vel[e0] = r0;
      if (e1 != e0) vel[e1] = r1;
    }
    wv_sync();
    return;
  }
  MJH_FOR_LANES(ed, s.nflexedge) {
    const int adr = M.flexedge_J_rowadr[ed];
    vel[ed] = M.flex_rigid[M.flexedge_flex[ed]] ? (real)0
                                                 : dot_sparse_ref(J + adr, qvel, M.flexedge_J_rownnz[ed], M.flexedge_J_colind + adr);
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// passive flex forces, added to qfrc_spring / qfrc_damper after the joint springs and dampers and
// before the tendons (mj_springdamper, engine_passive.c:739-787)
// ------------------------------------------------------------------------------------------------
// local edge -> corner indexing of an element (engine_passive.c:43-44)
MJH_DEV int flex_edge_corner(int dim, int ed, int i) {
  // dim 2: {1,2},{2,0},{0,1}      dim 3: {0,1},{1,2},{2,0},{2,3},{0,3},{1,3}
  const unsigned t2 = 0x21u | (0x02u << 8) | (0x10u << 16);            // nibbles: corner 0 | corner 1 << 4
  const unsigned long long t3 = 0x10ull | (0x21ull << 8) | (0x02ull << 16) | (0x32ull << 24) | (0x30ull << 32) | (0x31ull << 40);
  const unsigned byte = dim == 2 ? (t2 >> (8*ed)) & 0xffu : (unsigned)((t3 >> (8*ed)) & 0xffu);
  return (byte >> (4*i)) & 0xf;
}

// force block of one element: force[3*corner + x] -= elongation[ed1] * gradient[ed2][i][x] * metric[ed1][ed2] over ed1, ed2, the
// two ends i of edge ed2, x -- in that loop order (:606-617).  DIM is a template argument so that the edge -> corner table
// and every index into the force block are compile-time constants: the block stays in registers.
template <int DIM, class KP, class VX, class LP, class EP>
MJH_DEV void flex_stretch_force(MREF M, int t, KP k, real kD, real h, VX vx, LP len, EP evel, real* force) {
  constexpr int NE = DIM == 2 ? 3 : 6;
  constexpr int E0[6] = {DIM == 2 ? 1 : 0, DIM == 2 ? 2 : 1, DIM == 2 ? 0 : 2, 2, 0, 1};     // first corner of local edge q
  constexpr int E1[6] = {DIM == 2 ? 2 : 1, DIM == 2 ? 0 : 2, DIM == 2 ? 1 : 0, 3, 3, 3};     // second corner
  real p[DIM + 1][3];
#pragma unroll
  for (int i = 0; i <= DIM; i++) {
    const int v = M.flexelem_vert[4*t + i];
    p[i][0] = vx[3*v]; p[i][1] = vx[3*v + 1]; p[i][2] = vx[3*v + 2];
  }
  real elong[NE];
#pragma unroll
  for (int q = 0; q < NE; q++) {
    const int idx = M.flexelem_edge[6*t + q];
    const real def = len[idx], ref = M.flexedge_length0[idx];
    const real prev = def - evel[idx] * h;
    elong[q] = def*def - ref*ref + (def*def - prev*prev) * kD;
  }
#pragma unroll
  for (int i = 0; i < 3*(DIM + 1); i++) force[i] = 0;
#pragma unroll
  for (int ed1 = 0; ed1 < NE; ed1++) {
#pragma unroll
    for (int ed2 = 0; ed2 < NE; ed2++) {
      // (metric: the packed upper triangle, row-major)
      const int a = ed1 < ed2 ? ed1 : ed2, c = ed1 < ed2 ? ed2 : ed1;
      const real metric = k[a*NE - a*(a - 1)/2 + (c - a)];
      const int c0 = E0[ed2], c1 = E1[ed2];
#pragma unroll
      for (int x = 0; x < 3; x++) force[3*c0 + x] -= elong[ed1] * (p[c0][x] - p[c1][x]) * metric;
#pragma unroll
      for (int x = 0; x < 3; x++) force[3*c1 + x] -= elong[ed1] * (p[c1][x] - p[c0][x]) * metric;
    }
  }
}
template <int DIM, class KP, class VX, class LP, class EP, class FP>
MJH_DEV void flex_stretch_element(MREF M, int t, KP k, real kD, real h, VX vx, LP len, EP evel, FP efrc) {
  real force[3*(DIM + 1)];
  flex_stretch_force<DIM>(M, t, k, kD, h, vx, len, evel, force);
#pragma unroll
  for (int i = 0; i < 3*(DIM + 1); i++) efrc[12*t + i] = force[i];
}

MJH_DEV void flex_passive(MREF M, BREF B, int e, int enbl_spring, int enbl_damper) {
  const MJH_CONST_AS DSizes& s = M.s;
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr xmat = MJH_F(B, xmat, e);
  crptr qvel = MJH_F(B, qvel, e);
  crptr len = MJH_F(B, flexedge_length, e);
  crptr evel = MJH_F(B, flexedge_velocity, e);
  crptr J = MJH_F(B, flexedge_J, e);
  rptr fs = MJH_F(B, qfrc_spring, e);
  rptr fd = MJH_F(B, qfrc_damper, e);
  rptr efrc = MJH_F(B, flexelem_frc, e);
  const real h = M.o.timestep;

  // ---- bending: one lane per interior edge of a shell (mj_flexPassiveBend, :459-547); block = spring[12] | damper[12],
  //      already rotated into the vertex bodies' frames
  if (s.nflexbend) {
    rptr bfrc = MJH_F(B, flexbend_frc, e);
    MJH_FOR_LANES(ed, s.nflexedge) {
      const int f = M.flexedge_flex[ed];
      const int v3i = M.flexedge_flap[2*ed + 1];
      if (M.flex_dim[f] != 2 || M.flex_bendingadr[f] < 0 || M.flex_rigid[f] || v3i < 0) continue;
      const int v[4] = {M.flexedge_vert[2*ed], M.flexedge_vert[2*ed + 1], M.flexedge_flap[2*ed], v3i};
      auto b = M.flex_bending + M.flex_bendingadr[f] + 17*(ed - M.flex_edgeadr[f]);
      real ed3[3][3];
      for (int k = 0; k < 3; k++) v3_sub(ed3[k], vx + 3*v[k + 1], vx + 3*v[0]);
      real frc[4][3];
      v3_cross(frc[1], ed3[1], ed3[2]);
      v3_cross(frc[2], ed3[2], ed3[0]);
      v3_cross(frc[3], ed3[0], ed3[1]);
      for (int x = 0; x < 3; x++) frc[0][x] = -(frc[1][x] + frc[2][x] + frc[3][x]);
      int isfree[4], dadr[4];
      for (int i = 0; i < 4; i++) {
        const int bid = M.flexvert_bodyid[v[i]];
        isfree[i] = M.body_dofnum[bid] == 3;
        dadr[i] = M.body_dofadr[bid];
      }
      for (int i = 0; i < 4; i++) {
        real spring[3] = {0, 0, 0}, damper[3] = {0, 0, 0};
        for (int x = 0; x < 3; x++) {
          for (int j = 0; j < 4; j++) {
            if (enbl_spring) spring[x] += b[4*i + j] * vx[3*v[j] + x];
            if (enbl_damper) damper[x] += b[4*i + j] * (isfree[j] ? qvel[dadr[j] + x] : (real)0);
          }
          if (enbl_spring) spring[x] += b[16] * frc[i][x];
        }
        real sl[3], dl[3];
        m3_multvec(sl, xmat + 9*M.flexvert_bodyid[v[i]], spring);
        m3_multvec(dl, xmat + 9*M.flexvert_bodyid[v[i]], damper);
        for (int x = 0; x < 3; x++) {
          bfrc[24*ed + 3*i + x] = sl[x];
          bfrc[24*ed + 12 + 3*i + x] = dl[x] * M.flex_damping[f];
        }
      }
    }
  }

  // ---- stretch: one lane per element (mj_flexPassiveStretch, :551-630); block = force on the element's corners
  // (a model of one solid flex -- jelly.xml: 2058 tetrahedra, 33 per lane -- takes its elements two at a time per lane, in a
  // branch-free body: an element is a chain of dependent loads (corner ids -> positions, edge ids -> lengths), and with one
  // or two wavefronts on a SIMD only a second, independent chain hides its round trips)
  const int sadr0 = s.nflex == 1 ? (int)M.flex_stiffnessadr[0] : -1;
  if (s.nflex == 1 && M.flex_dim[0] == 3 && !M.flex_rigid[0] && !M.flex_interp[0] && sadr0 >= 0 && M.flex_stiffness[sadr0] != 0 && s.nflexelem > 0) {
    const real kD = h > 0 ? M.flex_damping[0] / h : 0;
    const int e0 = M.flex_elemadr[0], nel = s.nflexelem;
    for (int t0 = wv_lane(); t0 < nel; t0 += 2*MJH_W) {
      const int t1 = t0 + MJH_W < nel ? t0 + MJH_W : t0;
      real fa[12], fb[12];
      flex_stretch_force<3>(M, t0, M.flex_stiffness + sadr0 + 21*(t0 - e0), kD, h, vx, len, evel, fa);
      flex_stretch_force<3>(M, t1, M.flex_stiffness + sadr0 + 21*(t1 - e0), kD, h, vx, len, evel, fb);
#pragma unroll
      for (int i = 0; i < 12; i++) efrc[12*t0 + i] = fa[i];
      if (t1 != t0) {
#pragma unroll
        for (int i = 0; i < 12; i++) efrc[12*t1 + i] = fb[i];
      }
    }
  } else
  MJH_FOR_LANES(t, s.nflexelem) {
    const int f = M.flexelem_flex[t];
    const int dim = M.flex_dim[f];
    const int sadr = M.flex_stiffnessadr[f];
    if (dim < 2 || M.flex_rigid[f] || M.flex_interp[f] || sadr < 0 || M.flex_stiffness[sadr] == 0) continue;
    auto k = M.flex_stiffness + sadr + 21*(t - M.flex_elemadr[f]);
    const real kD = h > 0 ? M.flex_damping[f] / h : 0;
    if (dim == 3) flex_stretch_element<3>(M, t, k, kD, h, vx, len, evel, efrc);
    else flex_stretch_element<2>(M, t, k, kD, h, vx, len, evel, efrc);
  }
  wv_sync();

  // ---- per vertex: bending blocks in edge order, then the stretch sum (elements in order, rotated into the body frame),
  //      for vertices that are slider bodies (pinned vertices have no dofs to receive force)
  MJH_FOR_LANES(v, s.nflexvert) {
    const int bid = M.flexvert_bodyid[v];
    if (bid < 0) continue;               // (interpolated flexes: flex_passive_interp)
    const int nd = M.body_dofnum[bid];
    if (nd == 0 && (!s.nflexdofv || M.body_dofnum[M.body_weldid[bid]] == 0)) continue;
    const int dadr = M.body_dofadr[bid];
    const int f = M.flexvert_flex[v];
    if (M.flex_dim[f] == 1 || M.flex_rigid[f]) continue;
    if (s.nflexbend && nd == 3) {
      crptr bfrc = MJH_F(B, flexbend_frc, e);
      for (int a = M.flexvert_bendadr[v]; a < M.flexvert_bendadr[v + 1]; a++) {
        const int it = M.flexvert_bend[a];
        const int ed = it >> 2, i = it & 3;
        for (int x = 0; x < 3; x++) {
          if (enbl_spring) fs[dadr + x] -= bfrc[24*ed + 3*i + x];
          if (enbl_damper) fd[dadr + x] -= bfrc[24*ed + 12 + 3*i + x];
        }
      }
    }
    const int sadr = M.flex_stiffnessadr[f];
    if (M.flex_dim[f] >= 2 && sadr >= 0 && M.flex_stiffness[sadr] != 0) {
      real q[3] = {0, 0, 0};
      // (four blocks' loads in flight at a time; the additions stay in element order)
      const int ea0 = M.flexvert_elemadr[v], ea1 = M.flexvert_elemadr[v + 1];
      for (int a = ea0; a < ea1; a += 4) {
        int it[4]; real fx[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) it[u] = M.flexvert_elem[a + u < ea1 ? a + u : ea1 - 1];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int o = 12*(it[u] >> 2) + 3*(it[u] & 3); fx[u][0] = efrc[o]; fx[u][1] = efrc[o + 1]; fx[u][2] = efrc[o + 2]; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (a + u < ea1) { q[0] += fx[u][0]; q[1] += fx[u][1]; q[2] += fx[u][2]; }
      }
      if (M.body_simple[bid] == 2) {
        real ql[3];
        m3_multvec(ql, xmat + 9*bid, q);
        for (int x = 0; x < nd; x++) fs[dadr + x] += ql[x];
      } else {
        // (a vertex riding on an articulated body: its force is applied below, dof by dof)
        rptr vf = MJH_G(B, flexvert_frc, e);
        vf[3*v] = q[0]; vf[3*v + 1] = q[1]; vf[3*v + 2] = q[2];
      }
    }
  }
  wv_sync();
  // ---- stretch forces of vertices on articulated bodies: mj_applyFT (engine_support.c) adds J' f to every dof of the
  //      body's chain; a dof sums its vertices in vertex order (flexdof_vert)
  if (s.nflexdofv) {
    crptr vf = MJH_G(B, flexvert_frc, e);
    crptr cdof = MJH_F(B, cdof, e);
    crptr scom = MJH_F(B, subtree_com, e);
    MJH_FOR_LANES(j, s.nv) {
      const int a0 = M.flexdof_vadr[j], a1 = M.flexdof_vadr[j + 1];
      if (a0 == a1) continue;
      crptr cd = cdof + 6*j;
      real acc = fs[j];
      for (int a = a0; a < a1; a++) {
        const int v = M.flexdof_vert[a];
        const int f = M.flexvert_flex[v];
        const int sadr = M.flex_stiffnessadr[f];
        if (M.flex_dim[f] < 2 || M.flex_rigid[f] || sadr < 0 || M.flex_stiffness[sadr] == 0) continue;
        const int bid = M.flexvert_bodyid[v];
        real off[3], t[3];
        v3_sub(off, vx + 3*v, scom + 3*M.body_rootid[bid]);
        v3_cross(t, cd, off);
        const real jp[3] = {cd[3] + t[0], cd[4] + t[1], cd[5] + t[2]};
        real qf = 0;
        for (int r = 0; r < 3; r++) { const real fr = vf[3*v + r]; if (fr != 0) qf += jp[r]*fr; }
        acc += qf;
      }
      fs[j] = acc;
    }
    wv_sync();
  }

  // ---- interpolated flexes: cell elasticity on the node bodies (mj_flexPassiveInterp)
  if (s.nflexnode) flex_passive_interp(M, B, e, enbl_spring, enbl_damper);

  // ---- edge spring-dampers: every dof sums its edges in edge order (:757-787).  (flexedge_k / flexedge_d: the edge's flex
  //      coefficients, zero for rigid edges -- an edge whose two coefficients are off adds nothing and is skipped as in
  //      the reference; the entry's edge id sits next to its index, so the loads of an entry do not wait on one another)
  MJH_FOR_LANES(i, s.nv) {
    const int a0 = M.flexJ_cscadr[i], a1 = M.flexJ_cscadr[i + 1];
    if (a0 == a1) continue;
    real as = fs[i], ad = fd[i];
    int any = 0;
    for (int a = a0; a < a1; a += 4) {
      // (four entries' loads in flight at a time; the sums stay in edge order)
      int jj[4], ee[4]; real kk[4], dd[4], l0[4], ll[4], vv[4], Jv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int idx = a + u < a1 ? a + u : a1 - 1; jj[u] = M.flexJ_cscind[idx]; ee[u] = M.flexJ_cscedge[idx]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        kk[u] = enbl_spring ? (real)M.flexedge_k[ee[u]] : (real)0;
        dd[u] = enbl_damper ? (real)M.flexedge_d[ee[u]] : (real)0;
        l0[u] = M.flexedge_length0[ee[u]]; ll[u] = len[ee[u]]; vv[u] = evel[ee[u]]; Jv[u] = J[jj[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (a + u >= a1) continue;
        const real stiffness = kk[u], damping = dd[u];
        if (stiffness == 0 && damping == 0) continue;
        const real frc_spring = stiffness * (l0[u] - ll[u]);
        const real frc_damper = -damping * vv[u];
        as += Jv[u] * frc_spring;
        ad += Jv[u] * frc_damper;
        any = 1;
      }
    }
    if (any) { fs[i] = as; fd[i] = ad; }
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// bodies and weights of a contact (mj_contactJacobian / mj_diagApprox, engine_core_constraint.c:1535-1613, :1895-1944).
// `simple`: one body on each side (geom : geom, geom : flex vertex) -- the Jacobian is the difference of the two bodies'
// point Jacobians, dofs common to both chains left out of a compressed row (mj_jacDifPair, flg_skipcommon).  Otherwise a
// flex element is involved and the Jacobian is the weighted sum, in list order, of the listed bodies' point Jacobians
// (mj_jacSum; a compressed row stores the union of the chains): side 0 first with negative weights (-1 for a geom's
// body), then side 1; the corners of an element are weighted by inverse distance to the contact point, normalised
// (mj_elemBodyWeight :223-261).
// ------------------------------------------------------------------------------------------------
// A contact with an interpolated flex on a side lists NODE bodies: up to 27 per side, kept per contact in con_nodeb /
// con_nodew by flex_contact_nodes below (xb / xw point at the contact's lists; null otherwise).
struct ConSides { int n, simple, ext; int body[8]; real w[8]; ciptr xb; crptr xw; };
MJH_DEV int cs_body(const ConSides& S, int q) { return S.ext ? (int)S.xb[q] : S.body[q]; }
MJH_DEV real cs_w(const ConSides& S, int q) { return S.ext ? (real)S.xw[q] : S.w[q]; }

// corners of element el (global id) of flex f: bodies and normalised inverse-distance weights times sign, into S from slot at
MJH_DEV void flex_elem_weights(MREF M, BREF B, int e, int k, int f, int el, real sign, ConSides& S, int at) {
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr point = MJH_CON(B, con_pos, e, 3, k);
  const int n = M.flex_dim[f] + 1;
  real w[4] = {0, 0, 0, 0};
  int body[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (i >= n) continue;
    const int v = M.flexelem_vert[4*el + i];
    const real dx = point[0] - vx[3*v], dy = point[1] - vx[3*v + 1], dz = point[2] - vx[3*v + 2];
    const real dist = sqrt(dx*dx + dy*dy + dz*dz);
    w[i] = 1.0/r_max(MJH_MINVAL, dist);
    body[i] = M.flexvert_bodyid[v];
  }
  real sum = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) if (i < n) sum += w[i];
  const real inv = 1.0/sum;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int i = q - at;
    if (i < 0 || i >= n) continue;
    const real wi = i == 0 ? w[0] : (i == 1 ? w[1] : (i == 2 ? w[2] : w[3]));
    S.body[q] = i == 0 ? body[0] : (i == 1 ? body[1] : (i == 2 ? body[2] : body[3]));
    S.w[q] = (wi*inv)*sign;
  }
}

MJH_DEV void contact_sides(MREF M, BREF B, int e, int k, ConSides& S) {
  ciptr cg = MJH_CON(B, con_geom, e, 2, k);
#pragma unroll
  for (int q = 0; q < 8; q++) { S.body[q] = 0; S.w[q] = 0; }
  S.n = 2; S.simple = 1; S.ext = 0;
  int f1 = -1, e1 = -1, v1 = -1, f0 = -1, e0 = -1;
  if (MJH_HAS(MJH_FT_FLEX) && M.s.nconflex) {
    ciptr cf = MJH_G(B, con_flex, e) + MJH_CONFLEX*k;
    f1 = cf[0]; e1 = cf[1]; v1 = cf[2]; f0 = cf[3]; e0 = cf[4];
    if (M.s.nconside && f1 >= 0 && (M.flex_interp[f1] || (f0 >= 0 && M.flex_interp[f0]))) {
      ciptr nb = MJH_G(B, con_nodeb, e) + (M.s.nconside + 1)*k;
      S.simple = 0; S.ext = 1; S.n = nb[0]; S.xb = nb + 1; S.xw = MJH_G(B, con_nodew, e) + M.s.nconside*k;
      return;
    }
  }
  if (f1 < 0) {
    S.body[0] = M.geom_bodyid[cg[0]]; S.w[0] = -1;
    S.body[1] = M.geom_bodyid[cg[1]]; S.w[1] = 1;
    return;
  }
  if (f0 < 0 && v1 >= 0) {
    S.body[0] = M.geom_bodyid[cg[0]]; S.w[0] = -1;
    S.body[1] = M.flexvert_bodyid[M.flex_vertadr[f1] + v1]; S.w[1] = 1;
    return;
  }
  S.simple = 0;
  int at = 0;
  if (f0 < 0) { S.body[0] = M.geom_bodyid[cg[0]]; S.w[0] = -1; at = 1; }
  else { flex_elem_weights(M, B, e, k, f0, M.flex_elemadr[f0] + e0, -1, S, 0); at = M.flex_dim[f0] + 1; }
  flex_elem_weights(M, B, e, k, f1, M.flex_elemadr[f1] + e1, 1, S, at);
  S.n = at + M.flex_dim[f1] + 1;
}

// The node lists of the contacts that touch an interpolated flex, one lane per contact, once per step after the collision
// pass.  A side that is a vertex or an element of an interpolated flex: its parametric coordinate (the vertices' flex_vert0
// weighted by the absolute vertex weights), the cell that holds it, the cell's nodes with basis value at least 1e-5, weights
// signed like the side (mj_vertBodyWeight :265-384; nodes that share a body are merged within the side).  The other side
// as in contact_sides.  Layout: con_nodeb[(nconside + 1) k] = number of entries, then the bodies; con_nodew: the weights.
MJH_DEVN_HOT void flex_contact_nodes(MREF M_, BREF B_, int e_, int ncon) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int cap = s.nconside;
  MJH_FOR_LANES(k, ncon) {
    ciptr cf = MJH_G(B, con_flex, e) + MJH_CONFLEX*k;
    const int ff[2] = {cf[3], cf[0]}, ee[2] = {cf[4], cf[1]}, vv[2] = {cf[5], cf[2]};
    if (ff[1] < 0 || !(M.flex_interp[ff[1]] || (ff[0] >= 0 && M.flex_interp[ff[0]]))) continue;
    ciptr cg = MJH_CON(B, con_geom, e, 2, k);
    iptr ob = MJH_G(B, con_nodeb, e) + (cap + 1)*k + 1;
    rptr ow = MJH_G(B, con_nodew, e) + cap*k;
    int nb = 0;
    for (int side = 0; side < 2; side++) {
      const int f = ff[side];
      if (f < 0) { ob[nb] = M.geom_bodyid[cg[side]]; ow[nb] = side ? 1 : -1; nb++; continue; }
      int vid[4] = {0, 0, 0, 0}, nw = 0;
      real vw[4] = {0, 0, 0, 0};
      if (vv[side] >= 0) { vid[0] = M.flex_vertadr[f] + vv[side]; vw[0] = side ? 1 : -1; nw = 1; }
      else {
        ConSides T;
        const int el = M.flex_elemadr[f] + ee[side];
        flex_elem_weights(M, B, e, k, f, el, side ? 1 : -1, T, 0);
        nw = M.flex_dim[f] + 1;
        for (int i = 0; i < nw; i++) { vid[i] = M.flexelem_vert[4*el + i]; vw[i] = T.w[i]; }
      }
      if (!M.flex_interp[f]) {
        for (int i = 0; i < nw; i++) { ob[nb] = M.flexvert_bodyid[vid[i]]; ow[nb] = vw[i]; nb++; }
        continue;
      }
      const real sign = vw[0] < 0 ? -1 : 1;
      real coord[3] = {0, 0, 0}, local[3], basis[27];
      int idx[27];
      for (int i = 0; i < nw; i++) {
        const real a = fabs(vw[i]);
        coord[0] += M.flex_vert0[3*vid[i]]*a; coord[1] += M.flex_vert0[3*vid[i] + 1]*a; coord[2] += M.flex_vert0[3*vid[i] + 2]*a;
      }
      const int order = M.flex_interp[f], npc = (order + 1)*(order + 1)*(order + 1), na = M.flex_nodeadr[f];
      interp_cell_lookup(M, f, coord, local, idx);
      interp_basis(basis, local, order);
      const int start = nb;
      for (int j = 0; j < npc; j++) {
        const real w = basis[j];
        if (w < 1e-5) continue;
        const int b = M.flexnode_bodyid[na + idx[j]];
        int found = 0;
        for (int q = start; q < nb; q++) if (ob[q] == b) { ow[q] = ow[q] + sign*w; found = 1; break; }
        if (!found && nb < cap) { ob[nb] = b; ow[nb] = sign*w; nb++; }
      }
    }
    MJH_G(B, con_nodeb, e)[(cap + 1)*k] = nb;
  }
  wv_sync();
}

// column j of the contact's translational point Jacobian (world frame, before the rotation into the contact frame):
// jd = sum over the bodies whose chain holds dof j of weight x (cdof_lin + cdof_ang x (point - subtree_com[root]))
// (mj_jac, engine_core_util.c:176).  simple: jac2 - jac1.  rd (may be null): the rotational rows' column.
template <class CD, class SC, class PT>
MJH_DEV void contact_jac_col(MREF M, const ConSides& S, CD cdof, SC subtree_com, PT point, int j, real* jd, real* rd) {
  const MJH_CONST_AS DSizes& s = M.s;
  auto cd = cdof + 6*j;
  if (S.simple) {
    const int b1 = S.body[0], b2 = S.body[1];
    const int w1 = M.body_weldid[b1], w2 = M.body_weldid[b2];
    const int in1 = (M.body_dofanc[w1*s.nvw + (j >> 5)] >> (j & 31)) & 1;
    const int in2 = (M.body_dofanc[w2*s.nvw + (j >> 5)] >> (j & 31)) & 1;
    real j1[3] = {0, 0, 0}, j2[3] = {0, 0, 0};
    if (in1) {
      real off[3], t[3];
      v3_sub(off, point, subtree_com + 3*M.body_rootid[b1]);
      v3_cross(t, cd, off);
      j1[0] = cd[3] + t[0]; j1[1] = cd[4] + t[1]; j1[2] = cd[5] + t[2];
    }
    if (in2) {
      real off[3], t[3];
      v3_sub(off, point, subtree_com + 3*M.body_rootid[b2]);
      v3_cross(t, cd, off);
      j2[0] = cd[3] + t[0]; j2[1] = cd[4] + t[1]; j2[2] = cd[5] + t[2];
    }
    jd[0] = j2[0] - j1[0]; jd[1] = j2[1] - j1[1]; jd[2] = j2[2] - j1[2];
    if (rd) {
      rd[0] = (in2 ? cd[0] : (real)0) - (in1 ? cd[0] : (real)0);
      rd[1] = (in2 ? cd[1] : (real)0) - (in1 ? cd[1] : (real)0);
      rd[2] = (in2 ? cd[2] : (real)0) - (in1 ? cd[2] : (real)0);
    }
    return;
  }
  jd[0] = 0; jd[1] = 0; jd[2] = 0;
  if (rd) { rd[0] = 0; rd[1] = 0; rd[2] = 0; }
  int have = 0;
  if (S.ext) {
    for (int q = 0; q < S.n; q++) {
      const int b = S.xb[q];
      const int wq = M.body_weldid[b];
      if (!((M.body_dofanc[wq*s.nvw + (j >> 5)] >> (j & 31)) & 1)) continue;
      real off[3], t[3];
      v3_sub(off, point, subtree_com + 3*M.body_rootid[b]);
      v3_cross(t, cd, off);
      const real wt = S.xw[q];
      const real x0 = (cd[3] + t[0])*wt, x1 = (cd[4] + t[1])*wt, x2 = (cd[5] + t[2])*wt;
      jd[0] = have ? jd[0] + x0 : x0; jd[1] = have ? jd[1] + x1 : x1; jd[2] = have ? jd[2] + x2 : x2;
      if (rd) {
        const real r0 = cd[0]*wt, r1 = cd[1]*wt, r2 = cd[2]*wt;
        rd[0] = have ? rd[0] + r0 : r0; rd[1] = have ? rd[1] + r1 : r1; rd[2] = have ? rd[2] + r2 : r2;
      }
      have = 1;
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 8; q++) {
    if (q >= S.n) continue;
    const int b = S.body[q];
    const int wq = M.body_weldid[b];
    if (!((M.body_dofanc[wq*s.nvw + (j >> 5)] >> (j & 31)) & 1)) continue;
    real off[3], t[3];
    v3_sub(off, point, subtree_com + 3*M.body_rootid[b]);
    v3_cross(t, cd, off);
    const real wt = S.w[q];
    // (a geom's body enters negated, a weighted corner multiplied: mju_scl by -1 / mju_addToScl)
    const real x0 = (cd[3] + t[0])*wt, x1 = (cd[4] + t[1])*wt, x2 = (cd[5] + t[2])*wt;
    jd[0] = have ? jd[0] + x0 : x0; jd[1] = have ? jd[1] + x1 : x1; jd[2] = have ? jd[2] + x2 : x2;
    if (rd) {
      const real r0 = cd[0]*wt, r1 = cd[1]*wt, r2 = cd[2]*wt;
      rd[0] = have ? rd[0] + r0 : r0; rd[1] = have ? rd[1] + r1 : r1; rd[2] = have ? rd[2] + r2 : r2;
    }
    have = 1;
  }
}
