// Smooth (constraint-free) dynamics stages of the batched step, one wavefront per environment.
//
// Every function here is called by all 64 lanes of the wavefront that owns environment `e`.
// Lanes split the independent work items of a phase (bodies of one tree level, joints, dofs,
// matrix rows); phases are separated by wv_sync().  Per-item arithmetic follows the reference
// engine's operation order (files below are relative to /root/reference/src/engine) so results
// agree with the no-FMA CPU build to the last bit wherever the item decomposition allows it.
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)


// ------------------------------------------------------------------------------------------------
// frame of a geom/site/inertial frame attached to a body      (mj_local2Global, engine_core_util.c:975)
// ------------------------------------------------------------------------------------------------
template <class P0, class P1, class P2, class P3, class P4, class P5, class P6, class P7, class P8>
MJH_DEV void local2global(P0 opos, P1 omat, P2 pos, P3 quat,
                          P4 xpos, P5 xquat, P6 xmat,
                          P7 xipos, P8 ximat, int sameframe) {
  // position
  if (sameframe == MJH_SAMEFRAME_BODY) {
    v3_copy(opos, xpos);
  } else if (sameframe == MJH_SAMEFRAME_INERTIA) {
    v3_copy(opos, xipos);
  } else {
    real t[3];
    m3_mulvec(t, xmat, pos);
    v3_addto(t, xpos);
    v3_copy(opos, t);
  }
  // orientation
  if (sameframe == MJH_SAMEFRAME_NONE) {
    real q[4];
    q_mul(q, xquat, quat);
    q_tomat(omat, q);
  } else if (sameframe == MJH_SAMEFRAME_BODY || sameframe == MJH_SAMEFRAME_BODYROT) {
    for (int k = 0; k < 9; k++) omat[k] = xmat[k];
  } else {
    for (int k = 0; k < 9; k++) omat[k] = ximat[k];
  }
}

// ------------------------------------------------------------------------------------------------
// mj_kinematics                                  (engine_core_smooth.c:40-242)
// level-synchronous: all bodies of one depth level are independent given their parents
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_kinematics(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr qpos = MJH_F(B, qpos, e);
  rptr xpos = MJH_F(B, xpos, e);
  rptr xquat = MJH_F(B, xquat, e);
  rptr xmat = MJH_F(B, xmat, e);
  rptr xipos = MJH_F(B, xipos, e);
  rptr ximat = MJH_F(B, ximat, e);
  rptr xanchor = MJH_F(B, xanchor, e);
  rptr xaxis = MJH_F(B, xaxis, e);

  // world body
  if (wv_lane() == 0) {
    v3_zero(xpos); v3_zero(xipos);
    xquat[0] = 1; xquat[1] = 0; xquat[2] = 0; xquat[3] = 0;
    for (int k = 0; k < 9; k++) { xmat[k] = (k % 4 == 0) ? 1 : 0; ximat[k] = (k % 4 == 0) ? 1 : 0; }
  }
  wv_sync();

  // hinge angles -> (sin, cos) of the half angle for ALL joints at once, before the level loop: the
  // trigonometric evaluations are the most expensive part of a level and do not depend on the tree.
  // Parked in the joint's xanchor slot (read back by the lane that then overwrites it with the
  // anchor); cos = 2 marks a zero angle, whose quaternion is exactly the identity (mji_axisAngle2Quat).
  MJH_FOR_LANES(j, s.njnt) {
    if (M.jnt_type[j] == MJH_JNT_HINGE) {
      const int qadr = M.jnt_qposadr[j];
      const real angle = qpos[qadr] - M.qpos0[qadr];
      real sn = 0, cs = 2;
      if (angle != 0) r_sincos(angle*0.5, &sn, &cs);
      xanchor[3*j] = sn;
      xanchor[3*j + 1] = cs;
    }
  }
  wv_sync();

  for (int L = 1; L < s.nlevel; L++) {
    int a0 = M.body_level_adr[L], a1 = M.body_level_adr[L+1];
    MJH_FOR_LANES(k, a1 - a0) {
      int i = M.body_level_ids[a0 + k];
      real pos[3], quat[4];
      int jntadr = M.body_jntadr[i], jntnum = M.body_jntnum[i];

      if (jntnum == 1 && M.jnt_type[jntadr] == MJH_JNT_FREE) {
        int qadr = M.jnt_qposadr[jntadr];
        v3_copy(pos, qpos + qadr);
        q_copy(quat, qpos + qadr + 3);
        q_normalize(quat);
        v3_copy(xanchor + 3*jntadr, pos);
        v3_copy(xaxis + 3*jntadr, M.jnt_axis + 3*jntadr);
      } else {
        int pid = M.body_parentid[i];
        // body pose in the parent: from the model, or the user-driven mocap arrays (:84-93)
        real bpos[3], bquat[4];
        const int mid = (MJH_HAS(MJH_FT_MOCAP) && s.nmocap) ? (int)M.body_mocapid[i] : -1;
        if (mid >= 0) {
          v3_copy(bpos, MJH_G(B, mocap_pos, e) + 3*mid);
          q_copy(bquat, MJH_G(B, mocap_quat, e) + 4*mid);
          q_normalize(bquat);
        } else {
          v3_copy(bpos, M.body_pos + 3*i);
          q_copy(bquat, M.body_quat + 4*i);
        }
        if (pid) {
          m3_mulvec(pos, xmat + 9*pid, bpos);
          v3_addto(pos, xpos + 3*pid);
          q_mul(quat, xquat + 4*pid, bquat);
        } else {
          v3_copy(pos, bpos);
          q_copy(quat, bquat);
        }
        for (int j = 0; j < jntnum; j++) {
          int jid = jntadr + j;
          int qadr = M.jnt_qposadr[jid];
          int jt = M.jnt_type[jid];
          real anchor[3], axis[3];
          // (a hinge's half-angle sine / cosine wait in its xanchor slot, see above)
          const real hsn = xanchor[3*jid], hcs = xanchor[3*jid + 1];
          q_rotvec(axis, M.jnt_axis + 3*jid, quat);
          q_rotvec(anchor, M.jnt_pos + 3*jid, quat);
          v3_addto(anchor, pos);
          if (jt == MJH_JNT_SLIDE) {
            v3_addtoscl(pos, axis, qpos[qadr] - M.qpos0[qadr]);
          } else {
            real qloc[4];
            if (jt == MJH_JNT_BALL) {
              q_copy(qloc, qpos + qadr);
              q_normalize(qloc);
            } else if (hcs == 2) {
              qloc[0] = 1; qloc[1] = 0; qloc[2] = 0; qloc[3] = 0;
            } else {
              qloc[0] = hcs;
              qloc[1] = M.jnt_axis[3*jid]*hsn;
              qloc[2] = M.jnt_axis[3*jid + 1]*hsn;
              qloc[3] = M.jnt_axis[3*jid + 2]*hsn;
            }
            q_mul(quat, quat, qloc);
            real vec[3];
            q_rotvec(vec, M.jnt_pos + 3*jid, quat);
            v3_sub(pos, anchor, vec);
          }
          v3_copy(xanchor + 3*jid, anchor);
          v3_copy(xaxis + 3*jid, axis);
        }
      }
      q_normalize(quat);
      q_copy(xquat + 4*i, quat);
      v3_copy(xpos + 3*i, pos);
      q_tomat(xmat + 9*i, quat);
    }
    wv_sync();
  }

  // inertial frames
  MJH_FOR_LANES(k, s.nbody - 1) {
    int i = k + 1;
    local2global(xipos + 3*i, ximat + 9*i, M.body_ipos + 3*i, M.body_iquat + 4*i,
                 xpos + 3*i, xquat + 4*i, xmat + 9*i, xipos + 3*i, ximat + 9*i,
                 M.body_sameframe[i]);
  }
  wv_sync();

  // geoms and sites
  rptr geom_xpos = MJH_F(B, geom_xpos, e);
  rptr geom_xmat = MJH_F(B, geom_xmat, e);
  MJH_FOR_LANES(g, s.ngeom) {
    int b = M.geom_bodyid[g];
    local2global(geom_xpos + 3*g, geom_xmat + 9*g, M.geom_pos + 3*g, M.geom_quat + 4*g,
                 xpos + 3*b, xquat + 4*b, xmat + 9*b, xipos + 3*b, ximat + 9*b,
                 M.geom_sameframe[g]);
  }
  rptr site_xpos = MJH_F(B, site_xpos, e);
  rptr site_xmat = MJH_F(B, site_xmat, e);
  MJH_FOR_LANES(g, s.nsite) {
    int b = M.site_bodyid[g];
    local2global(site_xpos + 3*g, site_xmat + 9*g, M.site_pos + 3*g, M.site_quat + 4*g,
                 xpos + 3*b, xquat + 4*b, xmat + 9*b, xipos + 3*b, ximat + 9*b,
                 M.site_sameframe[g]);
  }
  wv_sync();
}

// accumulate per-body n-vectors into parents, deepest level first; children of one parent are
// added in decreasing body id, which reproduces the reference's `for b = nbody-1 .. 1` order.
template <class P0>
MJH_DEV void tree_accumulate_to_parent(MREF M, P0 x, int n, int include_world) {
  const MJH_CONST_AS DSizes& s = M.s;
  for (int L = s.nlevel - 2; L >= (include_world ? 0 : 1); L--) {
    int a0 = M.body_level_adr[L], a1 = M.body_level_adr[L+1];
    // a lane per (parent of the level, component): the children are added in their order, per component
    MJH_FOR_LANES(w, (a1 - a0)*n) {
      const int k = w / n, q = w - k*n;
      int p = M.body_level_ids[a0 + k];
      int c0 = M.body_child_adr[p], c1 = M.body_child_adr[p+1];
      real acc = x[n*p + q];
      // (a flex body hangs hundreds of vertex bodies on one parent: the children's values are fetched sixteen at a
      // time -- independent loads in flight together -- and added in order)
      int c = c0;
      for (; c + 16 <= c1; c += 16) {
        int id[16];
        real v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) id[u] = M.body_child_ids[c + u];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = x[n*id[u] + q];
#pragma unroll
        for (int u = 0; u < 16; u++) acc += v[u];
      }
      for (; c < c1; c++) acc += x[n*M.body_child_ids[c] + q];
      x[n*p + q] = acc;
    }
    wv_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// mj_comPos                                      (engine_core_smooth.c:246-350)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_compos(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr xipos = MJH_F(B, xipos, e);
  crptr ximat = MJH_F(B, ximat, e);
  crptr xmat = MJH_F(B, xmat, e);
  crptr xanchor = MJH_F(B, xanchor, e);
  crptr xaxis = MJH_F(B, xaxis, e);
  rptr subtree_com = MJH_F(B, subtree_com, e);
  rptr cinert = MJH_F(B, cinert, e);
  rptr cdof = MJH_F(B, cdof, e);

  MJH_FOR_LANES(i, s.nbody) v3_scl(subtree_com + 3*i, xipos + 3*i, M.body_mass[i]);
  wv_sync();
  tree_accumulate_to_parent(M, subtree_com, 3, 1);
  MJH_FOR_LANES(i, s.nbody) {
    if (M.body_subtreemass[i] < MJH_MINVAL) {
      v3_copy(subtree_com + 3*i, xipos + 3*i);
    } else {
      real inv = 1.0 / M.body_subtreemass[i];
      v3_scl(subtree_com + 3*i, subtree_com + 3*i, inv);
    }
  }
  wv_sync();

  MJH_FOR_LANES(i, s.nbody) {
    if (i == 0) {
      for (int k = 0; k < 10; k++) cinert[k] = 0;
    } else {
      real off[3];
      v3_sub(off, xipos + 3*i, subtree_com + 3*M.body_rootid[i]);
      sp_inert_com(cinert + 10*i, M.body_inertia + 3*i, ximat + 9*i, off, M.body_mass[i]);
    }
  }
  MJH_FOR_LANES(j, s.njnt) {
    int i = M.jnt_bodyid[j];
    int da = 6*M.jnt_dofadr[j];
    real off[3];
    v3_sub(off, subtree_com + 3*M.body_rootid[i], xanchor + 3*j);
    int jt = M.jnt_type[j];
    int skip = 0;
    if (jt == MJH_JNT_FREE) {
      for (int k = 0; k < 18; k++) cdof[da + k] = 0;
      cdof[da + 3 + 0] = 1;
      cdof[da + 3 + 7] = 1;
      cdof[da + 3 + 14] = 1;
      skip = 18;
    }
    if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
      for (int k = 0; k < 3; k++) {
        real axis[3] = {xmat[9*i + k], xmat[9*i + k + 3], xmat[9*i + k + 6]};
        rptr r = cdof + da + skip + 6*k;
        v3_copy(r, axis);
        v3_cross(r + 3, axis, off);
      }
    } else if (jt == MJH_JNT_SLIDE) {
      v3_zero(cdof + da);
      v3_copy(cdof + da + 3, xaxis + 3*j);
    } else {
      v3_copy(cdof + da, xaxis + 3*j);
      v3_cross(cdof + da + 3, xaxis + 3*j, off);
    }
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// mj_tendon, fixed tendons only                  (engine_core_smooth.c:927-986)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_tendon(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  if (!s.ntendon) return;
  crptr qpos = MJH_F(B, qpos, e);
  rptr L = MJH_F(B, ten_length, e);
  rptr J = MJH_F(B, ten_J, e);
  MJH_FOR_LANES(i, s.ntendon) {
    int adr = M.tendon_adr[i], num = M.tendon_num[i];
    int radr = M.ten_J_rowadr[i], rnnz = M.ten_J_rownnz[i];
    real len = 0;
    for (int k = 0; k < rnnz; k++) J[radr + k] = 0;
    if (MJH_HAS(MJH_FT_TENDONSPATIAL) && M.wrap_type[adr] != 1) {
      // spatial tendon through sites, with pulleys and wrapping geoms (mj_tendon, engine_core_smooth.c:988-1105): straight
      // segments between consecutive path points, moments from the difference of the end-point Jacobians along the
      // segment direction.  A site - geom - site triple (round 6) asks mju_wrap for the two points where the path meets
      // the sphere / cylinder: with a wrap the path is site -> wpnt1 -> (arc) -> wpnt2 -> site, the two inner points on
      // the geom's body (the arc itself has no moment: both ends on one body)
      crptr sx = MJH_F(B, site_xpos, e);
      crptr cdof = MJH_F(B, cdof, e);
      crptr com = MJH_F(B, subtree_com, e);
      real divisor = 1;
      // moment of the straight segment pa -> pb between bodies ba and bb (skipped when they coincide)
      auto segment = [&](const real* pa, const real* pb, int ba, int bb) {
        if (ba == bb) return;
        real dif[3];
        v3_sub(dif, pb, pa);
        v3_normalize(dif);
        real off0[3], off1[3];
        v3_sub(off0, pa, com + 3*M.body_rootid[ba]);
        v3_sub(off1, pb, com + 3*M.body_rootid[bb]);
        const real binv = 1/divisor;
        for (int k = 0; k < rnnz; k++) {
          const int c = M.ten_J_colind[radr + k];
          const int in0 = (M.body_dofanc[ba*s.nvw + (c >> 5)] >> (c & 31)) & 1;
          const int in1 = (M.body_dofanc[bb*s.nvw + (c >> 5)] >> (c & 31)) & 1;
          if (!in0 && !in1) continue;
          crptr cd = cdof + 6*c;
          real j0[3] = {0, 0, 0}, j1[3] = {0, 0, 0}, t[3];
          if (in0) { v3_cross(t, cd, off0); j0[0] = cd[3] + t[0]; j0[1] = cd[4] + t[1]; j0[2] = cd[5] + t[2]; }
          if (in1) { v3_cross(t, cd, off1); j1[0] = cd[3] + t[0]; j1[1] = cd[4] + t[1]; j1[2] = cd[5] + t[2]; }
          real tmp = 0;
          for (int r = 0; r < 3; r++) if (dif[r] != 0) tmp += (j1[r] - j0[r])*dif[r];
          J[radr + k] += binv*tmp;
        }
      };
      auto dist3 = [](const real* a, const real* b) -> real {
        const real dd[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
        return sqrt(dd[0]*dd[0] + dd[1]*dd[1] + dd[2]*dd[2]);
      };
      int j = 0;
      while (j < num - 1) {
        const int type0 = M.wrap_type[adr + j];
        int type1 = M.wrap_type[adr + j + 1];
        if (type0 == 2 || type1 == 2) {            // mjWRAP_PULLEY
          if (type0 == 2) divisor = M.wrap_prm[adr + j];
          j++;
          continue;
        }
        const int id0 = M.wrap_objid[adr + j];
        int id1 = M.wrap_objid[adr + j + 1];
        real wp[12];
        v3_copy(wp, sx + 3*id0);
        const int b0 = M.site_bodyid[id0];
        real wlen = -1;
        int wrapid = -1, wrapped = 0;
        if (type1 == 4 || type1 == 5) {            // mjWRAP_SPHERE / mjWRAP_CYLINDER: site - geom - site
          wrapped = 1;
          wrapid = id1;
          const int wtype = type1;
          type1 = M.wrap_type[adr + j + 2];
          id1 = M.wrap_objid[adr + j + 2];
          const int sideid = (int)round(M.wrap_prm[adr + j + 1]);      // (mju_round; -1: no side site)
          real x0[3], x1[3], gp[3], gm[9], sd[3];
          v3_copy(x0, sx + 3*id0);
          v3_copy(x1, sx + 3*id1);
          crptr gxp = MJH_F(B, geom_xpos, e);
          crptr gxm = MJH_F(B, geom_xmat, e);
          v3_copy(gp, gxp + 3*wrapid);
          for (int k = 0; k < 9; k++) gm[k] = gxm[9*wrapid + k];
          if (sideid >= 0) v3_copy(sd, sx + 3*sideid);
          wlen = mjh_wrap(wp + 3, x0, x1, gp, gm, M.geom_size[3*wrapid], wtype, sideid >= 0 ? (const real*)sd : (const real*)nullptr);
        }
        const int b1 = M.site_bodyid[id1];
        if (wlen < 0) {
          v3_copy(wp + 3, sx + 3*id1);
          len += dist3(wp, wp + 3) / divisor;
          segment(wp, wp + 3, b0, b1);
        } else {
          v3_copy(wp + 9, sx + 3*id1);
          const int gb = M.geom_bodyid[wrapid];
          len += (dist3(wp, wp + 3) + wlen + dist3(wp + 6, wp + 9)) / divisor;
          segment(wp, wp + 3, b0, gb);
          segment(wp + 6, wp + 9, gb, b1);
        }
        j += wrapped ? 2 : 1;
      }
      L[i] = len;
      continue;
    }
    for (int j = 0; j < num; j++) {
      int jid = M.wrap_objid[adr + j];
      real coef = M.wrap_prm[adr + j];
      len += coef * qpos[M.jnt_qposadr[jid]];
      int dof = M.jnt_dofadr[jid];
      // dst = 1*dst + coef*1 at the matching column (mju_combineSparseInc)
      for (int k = 0; k < rnnz; k++) {
        if (M.ten_J_colind[radr + k] == dof) J[radr + k] = 1*J[radr + k] + coef*1;
      }
    }
    L[i] = len;
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// mj_transmission: joint (slide/hinge) transmissions   (engine_core_smooth.c:1265-1329)
// moment is kept sparse with a static per-actuator row capacity (actuator_momentadr)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_transmission(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  if (!s.nu) return;
  crptr qpos = MJH_F(B, qpos, e);
  rptr length = MJH_F(B, actuator_length, e);
  rptr moment = MJH_F(B, actuator_moment, e);
  iptr rownnz = MJH_F(B, moment_rownnz, e);
  iptr colind = MJH_F(B, moment_colind, e);
  MJH_FOR_LANES(i, s.nu) {
    int id = M.actuator_trnid[2*i];
    auto gear = M.actuator_gear + 6*i;
    int adr = M.actuator_momentadr[i];
    if (MJH_HAS(MJH_FT_TRNMISC) && M.actuator_trntype[i] == MJH_TRN_SLIDERCRANK) {
      // slider-crank (engine_core_smooth.c:1396-1465): length = a.v - sqrt((a.v)^2 + r^2 - v.v)
      crptr site_xpos = MJH_F(B, site_xpos, e);
      crptr site_xmat = MJH_F(B, site_xmat, e);
      crptr cdof = MJH_F(B, cdof, e);
      crptr subtree_com = MJH_F(B, subtree_com, e);
      const int idslider = M.actuator_trnid[2*i + 1];
      const real rod = M.actuator_cranklength[i];
      real axis[3] = {site_xmat[9*idslider + 2], site_xmat[9*idslider + 5], site_xmat[9*idslider + 8]};
      real vec[3];
      v3_sub(vec, site_xpos + 3*id, site_xpos + 3*idslider);
      real av = v3_dot(vec, axis);
      real sdet, det = av*av + rod*rod - v3_dot(vec, vec);
      int ok = 1;
      real len;
      if (det <= 0) { ok = 0; sdet = 0; len = av; }
      else { sdet = sqrt(det); len = av - sdet; }
      real dlda[3], dldv[3];
      if (ok) {
        v3_scl(dldv, axis, 1 - av/sdet);
        v3_scl(dlda, vec, 1/sdet);
        v3_addto(dldv, dlda);
        v3_scl(dlda, vec, 1 - av/sdet);
      } else {
        v3_copy(dlda, vec);
        v3_copy(dldv, axis);
      }
      // Jacobians of the slider point / axis and of the crank site (mj_jacPointAxis, mj_jacSite),
      // chain rule, compression of the non-zero entries
      const int bs = M.site_bodyid[idslider], bc = M.site_bodyid[id];
      const int ws = M.body_weldid[bs], wc = M.body_weldid[bc];
      real offs[3], offc[3];
      v3_sub(offs, site_xpos + 3*idslider, subtree_com + 3*M.body_rootid[bs]);
      v3_sub(offc, site_xpos + 3*id, subtree_com + 3*M.body_rootid[bc]);
      int nnz = 0;
      for (int j = 0; j < s.nv; j++) {
        int ins = (M.body_dofanc[ws*s.nvw + (j >> 5)] >> (j & 31)) & 1;
        int inc = (M.body_dofanc[wc*s.nvw + (j >> 5)] >> (j & 31)) & 1;
        crptr cd = cdof + 6*j;
        real jS[3] = {0, 0, 0}, jr[3] = {0, 0, 0}, jC[3] = {0, 0, 0};
        if (ins) {
          real t[3];
          v3_cross(t, cd, offs);
          jS[0] = cd[3] + t[0]; jS[1] = cd[4] + t[1]; jS[2] = cd[5] + t[2];
          jr[0] = cd[0]; jr[1] = cd[1]; jr[2] = cd[2];
        }
        if (inc) {
          real t[3];
          v3_cross(t, cd, offc);
          jC[0] = cd[3] + t[0]; jC[1] = cd[4] + t[1]; jC[2] = cd[5] + t[2];
        }
        real jA[3] = {jr[1]*axis[2] - jr[2]*axis[1], jr[2]*axis[0] - jr[0]*axis[2], jr[0]*axis[1] - jr[1]*axis[0]};
        real jac[3] = {jC[0] - jS[0], jC[1] - jS[1], jC[2] - jS[2]};
        real mrow = 0;
        for (int k = 0; k < 3; k++) mrow += dlda[k]*jA[k] + dldv[k]*jac[k];
        if (mrow != 0) {
          moment[adr + nnz] = mrow * gear[0];
          colind[adr + nnz] = j;
          nnz++;
        }
      }
      length[i] = len * gear[0];
      rownnz[i] = nnz;
    } else if (MJH_HAS(MJH_FT_TRNMISC) && M.actuator_trntype[i] == MJH_TRN_SITE) {
      // site, no reference site (engine_core_smooth.c:1573-1593, :1705-1715): the gear is a wrench in
      // the site frame; moment = site Jacobians projected on it, length 0
      crptr site_xpos = MJH_F(B, site_xpos, e);
      crptr site_xmat = MJH_F(B, site_xmat, e);
      crptr cdof = MJH_F(B, cdof, e);
      crptr subtree_com = MJH_F(B, subtree_com, e);
      real g[6] = {gear[0], gear[1], gear[2], gear[3], gear[4], gear[5]};
      const int refid = M.actuator_trnid[2*i + 1];
      if (refid >= 0) {
        // reference site defined (:1596-1702): the site's pose RELATIVE to the reference site.  Length = position in the
        // reference frame . gear + orientation difference (expmap) . gear; moment = (J_site - J_ref), the columns of the dofs
        // both chains share cleared, projected on the gear expressed in the reference site's frame
        crptr xquat = MJH_F(B, xquat, e);
        const int b0 = M.site_bodyid[id], b1 = M.site_bodyid[refid];
        const int w0 = M.body_weldid[b0], w1 = M.body_weldid[b1];
        const int tr = !(g[0] == 0 && g[1] == 0 && g[2] == 0), ro = !(g[3] == 0 && g[4] == 0 && g[5] == 0);
        real len = 0, wt[3] = {0, 0, 0}, wr[3] = {0, 0, 0};
        if (tr) {
          real d3[3], v[3];
          v3_sub(d3, site_xpos + 3*id, site_xpos + 3*refid);
          m3_multvec(v, site_xmat + 9*refid, d3);
          len += v3_dot(v, g);
          m3_mulvec(wt, site_xmat + 9*refid, g);
        }
        if (ro) {
          real qa[4], qb[4], v[3];
          q_mul(qa, M.site_quat + 4*id, xquat + 4*b0);
          q_mul(qb, M.site_quat + 4*refid, xquat + 4*b1);
          q_sub(v, qa, qb);
          len += v3_dot(v, g + 3);
          m3_mulvec(wr, site_xmat + 9*refid, g + 3);
        }
        real off0[3], off1[3];
        v3_sub(off0, site_xpos + 3*id, subtree_com + 3*M.body_rootid[b0]);
        v3_sub(off1, site_xpos + 3*refid, subtree_com + 3*M.body_rootid[b1]);
        int nnz = 0;
        for (int j = 0; j < s.nv; j++) {
          const int in0 = (M.body_dofanc[w0*s.nvw + (j >> 5)] >> (j & 31)) & 1;
          const int in1 = (M.body_dofanc[w1*s.nvw + (j >> 5)] >> (j & 31)) & 1;
          real jp[3] = {0, 0, 0}, jr[3] = {0, 0, 0};
          if (in0 != in1) {
            crptr cd = cdof + 6*j;
            real cr[3];
            v3_cross(cr, cd, in0 ? off0 : off1);
            for (int r = 0; r < 3; r++) { jp[r] = cd[3 + r] + cr[r]; jr[r] = cd[r]; }
            if (in1) for (int r = 0; r < 3; r++) { jp[r] = 0 - jp[r]; jr[r] = 0 - jr[r]; }
          }
          real t1 = 0, t2 = 0;
          if (tr) for (int r = 0; r < 3; r++) if (wt[r] != 0) t1 += jp[r]*wt[r];
          if (ro) for (int r = 0; r < 3; r++) if (wr[r] != 0) t2 += jr[r]*wr[r];
          const real mrow = ro ? t1 + t2 : t1;
          if (mrow != 0) { moment[adr + nnz] = mrow; colind[adr + nnz] = j; nnz++; }
        }
        length[i] = len;
        rownnz[i] = nnz;
        continue;
      }
      real wrench[6];
      m3_mulvec(wrench, site_xmat + 9*id, g);
      m3_mulvec(wrench + 3, site_xmat + 9*id, g + 3);
      const int bs = M.site_bodyid[id];
      real off[3];
      v3_sub(off, site_xpos + 3*id, subtree_com + 3*M.body_rootid[bs]);
      int nnz = 0;
      for (int j = 0; j < s.nv; j++) {
        real t1 = 0, t2 = 0;
        if ((M.body_dofanc[bs*s.nvw + (j >> 5)] >> (j & 31)) & 1) {
          crptr cd = cdof + 6*j;
          real cr[3];
          v3_cross(cr, cd, off);
          for (int r = 0; r < 3; r++) if (wrench[r] != 0) t1 += (cd[3 + r] + cr[r])*wrench[r];
          for (int r = 0; r < 3; r++) if (wrench[3 + r] != 0) t2 += cd[r]*wrench[3 + r];
        }
        const real mrow = t1 + t2;
        if (mrow != 0) { moment[adr + nnz] = mrow; colind[adr + nnz] = j; nnz++; }
      }
      length[i] = 0;
      rownnz[i] = nnz;
    } else if (MJH_HAS(MJH_FT_TRNMISC) && M.actuator_trntype[i] == MJH_TRN_BODY) {
      // body (adhesion actuators; engine_core_smooth.c:1719-1830): moment = minus the average over the body's contacts of
      // the normal Jacobian.  Active contacts enter through mj_mulJacTVec(efc_J, w) with w = 1 on the normal row
      // (condim 1, elliptic cones) or 0.5 / (dim - 1) on every pyramid edge -- a sum over the constraint rows in order, i.e.
      // over the contacts in order and their rows; efc_J's contact rows are a function of the contact and cdof alone
      // (stage_make_constraint builds them later with the arithmetic restated here), and a contact has rows exactly when it
      // is not excluded.  Contacts in the gap add their normal Jacobian directly; then the sum of both is scaled by
      // -1 / count.  No meaningful length.
      crptr cdof = MJH_F(B, cdof, e);
      crptr com = MJH_F(B, subtree_com, e);
      const int ncon = (M.o.disableflags & (1<<4)) ? 0 : MJH_F(B, counts, e)[MJH_C_NCON];
      const int ispyramid = M.o.cone == 0;
      int counter = 0;
      for (int c = 0; c < ncon; c++) {
        ciptr cg = MJH_CON(B, con_geom, e, 2, c);
        if (cg[0] < 0 || cg[1] < 0) continue;
        if (M.geom_bodyid[cg[0]] != id && M.geom_bodyid[cg[1]] != id) continue;
        if (MJH_CON(B, con_exclude, e, 1, c)[0] <= 1) counter++;
      }
      int nnz = 0;
      for (int j = 0; counter && j < s.nv; j++) {
        crptr cd = cdof + 6*j;
        real act = 0, exc = 0;
        for (int c = 0; c < ncon; c++) {
          ciptr cg = MJH_CON(B, con_geom, e, 2, c);
          if (cg[0] < 0 || cg[1] < 0) continue;
          const int b1 = M.geom_bodyid[cg[0]], b2 = M.geom_bodyid[cg[1]];
          if (b1 != id && b2 != id) continue;
          const int excl = MJH_CON(B, con_exclude, e, 1, c)[0];
          if (excl > 1) continue;
          crptr point = MJH_CON(B, con_pos, e, 3, c);
          crptr fr = MJH_CON(B, con_frame, e, 9, c);
          const int w1 = M.body_weldid[b1], w2 = M.body_weldid[b2];
          const int in1 = (M.body_dofanc[w1*s.nvw + (j >> 5)] >> (j & 31)) & 1;
          const int in2 = (M.body_dofanc[w2*s.nvw + (j >> 5)] >> (j & 31)) & 1;
          if (!in1 && !in2) continue;        // (a zero column: nothing to add)
          real j1[3] = {0, 0, 0}, j2[3] = {0, 0, 0};
          if (in1) {
            real off[3], t[3];
            v3_sub(off, point, com + 3*M.body_rootid[b1]);
            v3_cross(t, cd, off);
            j1[0] = cd[3] + t[0]; j1[1] = cd[4] + t[1]; j1[2] = cd[5] + t[2];
          }
          if (in2) {
            real off[3], t[3];
            v3_sub(off, point, com + 3*M.body_rootid[b2]);
            v3_cross(t, cd, off);
            j2[0] = cd[3] + t[0]; j2[1] = cd[4] + t[1]; j2[2] = cd[5] + t[2];
          }
          const real jd[3] = {j2[0] - j1[0], j2[1] - j1[1], j2[2] - j1[2]};
          const real rd[3] = {(in2 ? cd[0] : (real)0) - (in1 ? cd[0] : (real)0),
                              (in2 ? cd[1] : (real)0) - (in1 ? cd[1] : (real)0),
                              (in2 ? cd[2] : (real)0) - (in1 ? cd[2] : (real)0)};
          // one row of the contact-frame Jacobian: translational rows 0..2, rotational rows 3..5 (mju_mulMatMat with zero-skip)
          auto frame_row = [&](int a) -> real {
            const real* v = a < 3 ? jd : rd;
            const int r = a < 3 ? a : a - 3;
            real acc = 0;
            for (int q = 0; q < 3; q++) {
              const real t = fr[3*r + q];
              if (t != 0) acc += v[q]*t;
            }
            return acc;
          };
          const real jn = frame_row(0);
          if (excl == 1) { exc += jn; continue; }
          const int dim = MJH_CON(B, con_dim, e, 1, c)[0];
          if (dim == 1 || !ispyramid) {
            act += jn*(real)1;
          } else {
            auto fri = M.pair_friction + 5*MJH_CON(B, con_pair, e, 1, c)[0];
            const real w = 0.5/(dim - 1);
            for (int a = 1; a < dim; a++) {
              const real ja = frame_row(a);
              act += (jn + ja*fri[a-1])*w;
              act += (jn + ja*(-fri[a-1]))*w;
            }
          }
        }
        const real mrow = (act + exc)*(-1.0/counter);
        if (mrow != 0) { moment[adr + nnz] = mrow; colind[adr + nnz] = j; nnz++; }
      }
      length[i] = 0;
      rownnz[i] = nnz;
    } else if (MJH_HAS(MJH_FT_TRNMISC) && M.actuator_trntype[i] == MJH_TRN_TENDON) {
      // tendon (engine_core_smooth.c:1468-1480): the tendon's length and moment row, scaled by the gear
      crptr tl = MJH_F(B, ten_length, e);
      crptr tJ = MJH_F(B, ten_J, e);
      const int ra = M.ten_J_rowadr[id], rn = M.ten_J_rownnz[id];
      length[i] = tl[id]*gear[0];
      rownnz[i] = rn;
      for (int k = 0; k < rn; k++) {
        colind[adr + k] = M.ten_J_colind[ra + k];
        moment[adr + k] = tJ[ra + k]*gear[0];
      }
    } else if (MJH_HAS(MJH_FT_TRNMISC) && M.jnt_type[id] == MJH_JNT_BALL) {
      // ball joint: 3D gear; length = expmap(quat) . gear (engine_core_smooth.c:1331-1362)
      const int qa = M.jnt_qposadr[id], da = M.jnt_dofadr[id];
      real quat[4] = {qpos[qa], qpos[qa+1], qpos[qa+2], qpos[qa+3]};
      real axis[3], ga[3], g[3] = {gear[0], gear[1], gear[2]};
      q_normalize(quat);
      q_tovel(axis, quat, 1);
      if (M.actuator_trntype[i] == MJH_TRN_JOINT) { ga[0] = g[0]; ga[1] = g[1]; ga[2] = g[2]; }
      else { real nq[4] = {quat[0], -quat[1], -quat[2], -quat[3]}; q_rotvec(ga, g, nq); }
      length[i] = v3_dot(axis, ga);
      rownnz[i] = 3;
      for (int k = 0; k < 3; k++) { colind[adr + k] = da + k; moment[adr + k] = ga[k]; }
    } else if (MJH_HAS(MJH_FT_TRNMISC) && M.jnt_type[id] == MJH_JNT_FREE) {
      // free joint: 6D gear, no meaningful length (:1364-1392)
      const int qa = M.jnt_qposadr[id], da = M.jnt_dofadr[id];
      real ga[3], g[3] = {gear[3], gear[4], gear[5]};
      if (M.actuator_trntype[i] == MJH_TRN_JOINT) { ga[0] = g[0]; ga[1] = g[1]; ga[2] = g[2]; }
      else {
        real quat[4] = {qpos[qa+3], qpos[qa+4], qpos[qa+5], qpos[qa+6]};
        q_normalize(quat);
        real nq[4] = {quat[0], -quat[1], -quat[2], -quat[3]};
        q_rotvec(ga, g, nq);
      }
      length[i] = 0;
      rownnz[i] = 6;
      for (int k = 0; k < 3; k++) { colind[adr + k] = da + k; moment[adr + k] = gear[k]; }
      for (int k = 0; k < 3; k++) { colind[adr + 3 + k] = da + 3 + k; moment[adr + 3 + k] = ga[k]; }
    } else {
      // slide / hinge joint: scalar gear
      rownnz[i] = 1;
      colind[adr] = M.jnt_dofadr[id];
      length[i] = qpos[M.jnt_qposadr[id]]*gear[0];
      moment[adr] = gear[0];
    }
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// mj_crb (+ mj_makeM)                            (engine_core_smooth.c:1890-1971)
// ------------------------------------------------------------------------------------------------
MJH_DEV int pairs_euler_factor(MREF M, BREF B, int e);
MJH_DEVN void stage_crb(MREF M_, BREF B_, int e_, int nopark) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr cinert = MJH_F(B, cinert, e);
  crptr cdof = MJH_F(B, cdof, e);
  rptr crb = MJH_F(B, crb, e);
  rptr Mq = MJH_F(B, qLD, e);       // M is assembled where it will be factorised

  MJH_FOR_LANES(k, 10*s.nbody) crb[k] = cinert[k];
  wv_sync();
  tree_accumulate_to_parent(M, crb, 10, 0);

  MJH_FOR_LANES(i, s.nv) {
    int adr = M.M_rowadr[i];
    if (M.dof_simplenum[i]) {
      Mq[adr] = M.dof_M0[i];
    } else {
      int a = adr + M.M_rownnz[i] - 1;
      real buf[6];
      sp_mul_inert(buf, crb + 10*M.dof_bodyid[i], cdof + 6*i);
      // diagonal starts from the (effective) armature, off-diagonals from zero
      real init = M.dof_armature_eff[i];
      for (int j = i; j >= 0; j = M.dof_parentid[j]) {
        Mq[a--] = init + sp_dot6(cdof + 6*j, buf);
        init = 0;
      }
    }
  }
  wv_sync();
  // mj_tendonArmature (engine_core_smooth.c:1845-1886): M += armature * ten_J' ten_J, restricted to M's own pattern -- per
  // non-zero ten_J[i], row i of M gains (armature * ten_J[i]) * ten_J where the column lists meet
  // (mju_addToSclSparseInc).  Tendons in order (their rows may coincide), a tendon's rows lane-parallel.
  if (MJH_HAS(MJH_FT_PASSIVEMISC) && M.o.has_ten_armature) {
    crptr tJ = MJH_F(B, ten_J, e);
    for (int k = 0; k < s.ntendon; k++) {
      const real arm = M.tendon_armature_eff[k];
      if (!arm) continue;
      const int radr = M.ten_J_rowadr[k], rnnz = M.ten_J_rownnz[k];
      MJH_FOR_LANES(j, rnnz) {
        const real Ji = tJ[radr + j];
        if (Ji != 0) {
          const int i = M.ten_J_colind[radr + j];
          const int madr = M.M_rowadr[i], mn = M.M_rownnz[i];
          const real scl = arm * Ji;
          int as = 0, ad = 0;
          while (as < rnnz && ad < mn) {
            const int is = M.ten_J_colind[radr + as], id = M.M_colind[madr + ad];
            if (is == id) { Mq[madr + ad] += scl * tJ[radr + as]; as++; ad++; }
            else if (is < id) as++;
            else ad++;
          }
        }
      }
      wv_sync();
    }
  }
  // the global copy of M: what tests inspect and what mj_Euler's fallback, implicitfast and the primal
  // solvers read once qLD has been factorised in place.  A step of the PGS + Euler path whose qH factor
  // is produced next to M's never reads it.
  const int unread = nopark && pairs_euler_factor(M, B, e) && (!MJH_HAS(MJH_FT_PRIMAL) || M.o.solver == MJH_SOL_PGS) &&
                     !(MJH_HAS(MJH_FT_SENSOR) && (s.sens_energy & 2));      // (a kinetic-energy sensor reads M)
  if (!unread) {
    rptr Mhome = MJH_G(B, M, e);
    MJH_FOR_LANES(k, s.nC) Mhome[k] = Mq[k];
    wv_sync();
  }
}

#if !MJH_LANE_MODE
// ------------------------------------------------------------------------------------------------
// Wavefront L'DL routines for nv <= MJH_W (M.s.ld_fast): lane i of the environment's lane group owns dof i.
// Row addresses, row lengths and the strict-ancestor masks sit in registers and are handed around
// with v_readlane, so the serial sweeps contain no dependent model-memory loads; qLD itself is
// read where it lives (LDS by plan).  Same arithmetic, same order as the generic versions below.
// ------------------------------------------------------------------------------------------------
template <class P0, class P1>
MJH_DEVN_HOT void factor_ld_fast(MREF M_, P0 mat, P1 diaginv) {
  MREF M = wv_uniform_ref(M_);
  const auto* ld_prog = wv_uniform_ptr(M.ld_prog);
  const int nv = M.s.nv;
  const int lane = wv_lane();
  const int li = lane < nv ? lane : 0;
  const int myadr = wv_uniform_ptr(M.M_rowadr)[li];
  const int mynnz = wv_uniform_ptr(M.M_rownnz)[li];
  const int myprog = wv_uniform_ptr(M.ld_prog_adr)[li];
  // first 64 update items of the pivot about to be processed (prefetched one pivot ahead)
  const int first = wv_bcast_i(myprog, nv - 1) + lane;
  int item = ld_prog[first < M.s.nldprog ? first : 0];
  for (int k = nv - 1; k >= 0; k--) {
    const int start = wv_bcast_i(myadr, k);
    const int diag = wv_bcast_i(mynnz, k) - 1;
    const int pbase = wv_bcast_i(myprog, k);
    const int total = diag*(diag + 1)/2;
    int item_next = 0;
    if (k > 0) {
      int nb = wv_bcast_i(myprog, k - 1) + lane;
      item_next = ld_prog[nb < M.s.nldprog ? nb : 0];
    }
    const real invD = 1 / mat[start + diag];
    for (int w = lane; w < total; w += MJH_W) {
      const int it = (w < MJH_W) ? item : (int)ld_prog[pbase + w];
      const int dst = it & 1023, src = (it >> 10) & 1023, sc = (it >> 20) & 1023;
      const real scl = -mat[sc] * invD;
      mat[dst] += mat[src] * scl;
    }
    wv_sync();
    if (lane < diag) mat[start + lane] = mat[start + lane] * invD;     // diag <= 16
    if (lane == 0) diaginv[k] = invD;
    wv_sync();
    item = item_next;
  }
}

#if MJH_W == 64
// Two matrices of M's sparsity factorised in one pass: lanes 0..31 update matrix A, lanes 32..63
// matrix B, pivot by pivot with the same work list.  The per-pivot chain (pivot read, reciprocal,
// two barriers) is paid once for both.
template <class P0, class P1>
MJH_DEVN_HOT void factor_ld_pair(MREF M_, P0 matA, P1 diaginvA, P0 matB, P1 diaginvB) {
  MREF M = wv_uniform_ref(M_);
  const auto* ld_prog = wv_uniform_ptr(M.ld_prog);
  const int nv = M.s.nv;
  const int lane = wv_lane();
  const int half = lane >> 5, hl = lane & 31;
  const int li = hl < nv ? hl : 0;
  const int myadr = wv_uniform_ptr(M.M_rowadr)[li];
  const int mynnz = wv_uniform_ptr(M.M_rownnz)[li];
  const int myprog = wv_uniform_ptr(M.ld_prog_adr)[li];
  auto at = [&](int k) -> real { return half ? (real)matB[k] : (real)matA[k]; };
  auto put = [&](int k, real v) { if (half) matB[k] = v; else matA[k] = v; };
  const int first = wv_bcast_i(myprog, nv - 1) + hl;
  int item = ld_prog[first < M.s.nldprog ? first : 0];
  for (int k = nv - 1; k >= 0; k--) {
    const int start = wv_bcast_i(myadr, k);
    const int diag = wv_bcast_i(mynnz, k) - 1;
    const int pbase = wv_bcast_i(myprog, k);
    const int total = diag*(diag + 1)/2;
    int item_next = 0;
    if (k > 0) {
      int nb = wv_bcast_i(myprog, k - 1) + hl;
      item_next = ld_prog[nb < M.s.nldprog ? nb : 0];
    }
    const real invD = 1 / at(start + diag);
    for (int w = hl; w < total; w += 32) {
      const int it = (w < 32) ? item : (int)ld_prog[pbase + w];
      const int dst = it & 1023, src = (it >> 10) & 1023, sc = (it >> 20) & 1023;
      const real scl = -at(sc) * invD;
      put(dst, at(dst) + at(src) * scl);
    }
    wv_sync();
    if (hl < diag) put(start + hl, at(start + hl) * invD);     // diag <= 16
    if (hl == 0) { if (half) diaginvB[k] = invD; else diaginvA[k] = invD; }
    wv_sync();
    item = item_next;
  }
}
#endif

template <class P0, class P1, class P2>
MJH_DEVN_HOT void solve_ld_fast(MREF M_, P0 xmem, P1 qLD, P2 diaginv) {
  MREF M = wv_uniform_ref(M_);
  const auto* colind = wv_uniform_ptr(M.M_colind);
  const auto* ancmask = wv_uniform_ptr(M.dof_ancmask);
  const int nv = M.s.nv;
  const int lane = wv_lane();
  const int li = lane < nv ? lane : 0;
  const int myadr = wv_uniform_ptr(M.M_rowadr)[li];
  const int mynnz = wv_uniform_ptr(M.M_rownnz)[li];
  const int mydepth = mynnz - 1;                       // my position in every descendant's row
  const int anc_lo = ancmask[2*li], anc_hi = ancmask[2*li + 1];
  real x = lane < nv ? xmem[li] : 0;
  const real dinv = lane < nv ? diaginv[li] : 0;

  // x <- L^-T x : row i scatters into its ancestors; lane a is an ancestor of i iff bit a of i's mask
  int ilast = nv - 1;
  while (ilast > 0 && wv_bcast_i(mynnz, ilast) == 1) ilast--;
  real q = 0;
  {
    const int lo = wv_bcast_i(anc_lo, ilast), hi = wv_bcast_i(anc_hi, ilast);
    const int isanc = lane < 32 ? (lo >> lane) & 1 : (hi >> (lane - 32)) & 1;
    const int adr = wv_bcast_i(myadr, ilast);
    if (isanc) q = qLD[adr + mydepth];
  }
  for (int i = ilast; i > 0; ) {
    // next row with off-diagonals, and its coefficient for this lane (prefetch)
    int inext = i - 1;
    while (inext > 0 && wv_bcast_i(mynnz, inext) == 1) inext--;
    real qnext = 0;
    if (inext > 0) {
      const int lo = wv_bcast_i(anc_lo, inext), hi = wv_bcast_i(anc_hi, inext);
      const int isanc = lane < 32 ? (lo >> lane) & 1 : (hi >> (lane - 32)) & 1;
      const int adr = wv_bcast_i(myadr, inext);
      if (isanc) qnext = qLD[adr + mydepth];
    }
    const real xi = wv_bcast(x, i);
    if (xi != 0) {
      const int lo = wv_bcast_i(anc_lo, i), hi = wv_bcast_i(anc_hi, i);
      const int isanc = lane < 32 ? (lo >> lane) & 1 : (hi >> (lane - 32)) & 1;
      if (isanc) x -= q * xi;
    }
    q = qnext;
    i = inext;
  }
  // x <- D^-1 x
  x *= dinv;
  // x <- L^-1 x : for row i, lane k < nnz-1 takes position k of the row (ancestor colind[adr+k]);
  // mju_dotSparse sums positions in four interleaved chains, (r0+r2)+(r1+r3), then the tail one by one
  // (lane k's ancestor index and coefficient of the next row are fetched while this row is reduced)
  int anc_n = 0;
  real qk_n = 0;
  if (nv > 1) {
    const int nnz1 = wv_bcast_i(mynnz, 1) - 1, adr = wv_bcast_i(myadr, 1);
    if (lane < nnz1) { anc_n = colind[adr + lane]; qk_n = qLD[adr + lane]; }
  }
  for (int i = 1; i < nv; i++) {
    const int nnz1 = wv_bcast_i(mynnz, i) - 1;
    const int anc = anc_n;
    const real qk = qk_n;
    anc_n = 0; qk_n = 0;
    if (i + 1 < nv) {
      const int nnz1n = wv_bcast_i(mynnz, i + 1) - 1, adrn = wv_bcast_i(myadr, i + 1);
      if (lane < nnz1n) { anc_n = colind[adrn + lane]; qk_n = qLD[adrn + lane]; }
    }
    if (nnz1 == 0) continue;
    const real xa = wv_shfl(x, anc);
    const real p = qk * xa;
    const int n4 = nnz1 & ~3, L = n4 >> 2;
    real acc = 0;
    if (L > 0) {
      acc = acc + p;
      if (L > 1) {
        acc = acc + wv_row_shl<4>(p);
        if (L > 2) {
          acc = acc + wv_row_shl<8>(p);
          if (L > 3) acc = acc + wv_row_shl<12>(p);
        }
      }
    }
    real res = (wv_bcast(acc, 0) + wv_bcast(acc, 2)) + (wv_bcast(acc, 1) + wv_bcast(acc, 3));
    for (int t = n4; t < nnz1; t++) res += wv_bcast(p, t);
    if (lane == i) x -= res;
  }
  if (lane < nv) xmem[li] = x;
  wv_sync();
}
#if MJH_W == 64
// Two independent solves with the same sparsity in one pass, nv <= 32: lanes 0..31 carry system A
// (xa, qLDa, dinva), lanes 32..63 system B.  The sweeps of solve_ld_fast are chains of dependent
// cross-lane steps that keep fewer than half of the wavefront busy; running a second system in the
// idle half costs no extra steps.  Per system the arithmetic and its order are those of
// solve_ld_fast (and of mj_solveLD).
template <class XA, class XB, class Q, class DI>
MJH_DEVN_HOT void solve_ld_pair(MREF M_, XA xa_mem, Q qLDa, DI dinva, XB xb_mem, Q qLDb, DI dinvb) {
  MREF M = wv_uniform_ref(M_);
  const auto* colind = wv_uniform_ptr(M.M_colind);
  const auto* ancmask = wv_uniform_ptr(M.dof_ancmask);
  const int nv = M.s.nv;
  const int lane = wv_lane();
  const int half = lane >> 5, hl = lane & 31, base = lane & 32;
  const int li = hl < nv ? hl : 0;
  const int myadr = wv_uniform_ptr(M.M_rowadr)[li];
  const int mynnz = wv_uniform_ptr(M.M_rownnz)[li];
  const int mydepth = mynnz - 1;
  const int anc_lo = ancmask[2*li];
  // value of v in lane i of my half
  auto hb = [&](real v, int i) -> real { const real a = wv_bcast(v, i), b = wv_bcast(v, 32 + i); return half ? b : a; };
  auto qld = [&](int k) -> real { return half ? (real)qLDb[k] : (real)qLDa[k]; };
  real x = 0, dinv = 0;
  if (hl < nv) { x = half ? (real)xb_mem[li] : (real)xa_mem[li]; dinv = half ? (real)dinvb[li] : (real)dinva[li]; }

  // x <- L^-T x
  int ilast = nv - 1;
  while (ilast > 0 && wv_bcast_i(mynnz, ilast) == 1) ilast--;
  real q = 0;
  {
    const int lo = wv_bcast_i(anc_lo, ilast);
    const int adr = wv_bcast_i(myadr, ilast);
    if ((lo >> hl) & 1) q = qld(adr + mydepth);
  }
  for (int i = ilast; i > 0; ) {
    int inext = i - 1;
    while (inext > 0 && wv_bcast_i(mynnz, inext) == 1) inext--;
    real qnext = 0;
    if (inext > 0) {
      const int lo = wv_bcast_i(anc_lo, inext);
      const int adr = wv_bcast_i(myadr, inext);
      if ((lo >> hl) & 1) qnext = qld(adr + mydepth);
    }
    const real xi = hb(x, i);
    const int lo = wv_bcast_i(anc_lo, i);
    if (xi != 0 && ((lo >> hl) & 1)) x -= q * xi;
    q = qnext;
    i = inext;
  }
  // x <- D^-1 x
  x *= dinv;
  // x <- L^-1 x
  int anc_n = 0;
  real qk_n = 0;
  if (nv > 1) {
    const int nnz1 = wv_bcast_i(mynnz, 1) - 1, adr = wv_bcast_i(myadr, 1);
    if (hl < nnz1) { anc_n = colind[adr + hl]; qk_n = qld(adr + hl); }
  }
  for (int i = 1; i < nv; i++) {
    const int nnz1 = wv_bcast_i(mynnz, i) - 1;
    const int anc = anc_n;
    const real qk = qk_n;
    anc_n = 0; qk_n = 0;
    if (i + 1 < nv) {
      const int nnz1n = wv_bcast_i(mynnz, i + 1) - 1, adrn = wv_bcast_i(myadr, i + 1);
      if (hl < nnz1n) { anc_n = colind[adrn + hl]; qk_n = qld(adrn + hl); }
    }
    if (nnz1 == 0) continue;
    const real xa = wv_shfl(x, base + anc);
    const real p = qk * xa;
    const int n4 = nnz1 & ~3, L = n4 >> 2;
    real acc = 0;
    if (L > 0) {
      acc = acc + p;
      if (L > 1) {
        acc = acc + wv_row_shl<4>(p);
        if (L > 2) {
          acc = acc + wv_row_shl<8>(p);
          if (L > 3) acc = acc + wv_row_shl<12>(p);
        }
      }
    }
    real res = (hb(acc, 0) + hb(acc, 2)) + (hb(acc, 1) + hb(acc, 3));
    for (int t = n4; t < nnz1; t++) res += hb(p, t);
    if (hl == i) x -= res;
  }
  if (hl < nv) { if (half) xb_mem[li] = x; else xa_mem[li] = x; }
  wv_sync();
}
#endif
#endif  // !MJH_LANE_MODE

// ------------------------------------------------------------------------------------------------
// sparse L'DL factorisation in place              (mj_factorI, engine_core_smooth.c:2005-2029)
// rows nv-1 .. 0 in order; for one row k the updates of its ancestor rows are independent
// ------------------------------------------------------------------------------------------------
template <class P0, class P1>
MJH_DEVN void factor_ld(MREF M, P0 mat, P1 diaginv) {
  const int nv = M.s.nv;
#if !MJH_LANE_MODE
  if (M.s.ld_fast && (MJH_W == 64 || M.s.nv <= MJH_W)) {
    if (mjh_in_lds(mat)) factor_ld_fast(M, mjh_local(mat.p), diaginv);
    else factor_ld_fast(M, mat, diaginv);
    return;
  }
#endif
  // (rows without off-diagonals update nobody: only the rows that have some -- M.ld_rows, ascending -- are taken in
  // sequence, the others all at once afterwards, when their descendants have finished with their diagonals)
  for (int q = M.s.nldrows - 1; q >= 0; q--) {
    const int k = M.ld_rows[q];
    int start = M.M_rowadr[k];
    int diag = M.M_rownnz[k] - 1;
    int end = start + diag;
    real invD = 1 / mat[end];
    // flattened (ancestor entry, element) work list: entry a (0..diag-1) updates row i=colind[start+a],
    // elements 0..a  (row i of the CSR has exactly a+1 entries: ancestors of i, then i)
#if MJH_LANE_MODE
    for (int a = 0; a < diag; a++) {
      int adr_i = M.M_rowadr[M.M_colind[start + a]];
      real scl = -mat[start + a] * invD;
      for (int el = 0; el <= a; el++) mat[adr_i + el] += mat[start + el] * scl;
    }
#else
    int total = diag*(diag + 1)/2;
    MJH_FOR_LANES(w, total) {
      // invert w = a*(a+1)/2 + el, 0 <= el <= a
      int a = (int)((sqrt(8.0*w + 1.0) - 1.0)*0.5);
      while (a*(a+1)/2 > w) a--;
      while ((a+1)*(a+2)/2 <= w) a++;
      int el = w - a*(a+1)/2;
      int i = M.M_colind[start + a];
      real scl = -mat[start + a] * invD;
      mat[M.M_rowadr[i] + el] += mat[start + el] * scl;
    }
#endif
    wv_sync();
    MJH_FOR_LANES(a, diag) mat[start + a] = mat[start + a] * invD;
    if (wv_lane() == 0) diaginv[k] = invD;
    wv_sync();
  }
  MJH_FOR_LANES(k, nv) if (M.M_rownnz[k] == 1) diaginv[k] = 1 / mat[M.M_rowadr[k]];
  wv_sync();
}

// does stage_finish also produce mj_Euler's damped acceleration, and stage_factor_m the factor it needs?
// (Euler integrator with joint damping, register-resident L'DL routines, nv <= 32, environment-major batch)
MJH_DEV int pairs_euler_solve(MREF M, BREF B) {
#if !MJH_LANE_MODE && MJH_W == 64
  return M.o.euler_damp && M.o.integrator == MJH_INT_EULER && M.s.ld_fast && M.s.nv <= 32 && B.soa == 0;
#else
  return 0;
#endif
}

template <class P0> MJH_DEV real poly_force_deriv(real linear, P0 poly, real x, int odd);
// did stage_factor_m leave the factor of qH in qH2's global home?  (the condition it evaluates)
MJH_DEV int pairs_euler_factor(MREF M, BREF B, int e) {
#if !MJH_LANE_MODE && MJH_W == 64
  return pairs_euler_solve(M, B) && mjh_in_lds(MJH_F(B, qLD, e)) == mjh_in_lds(MJH_F(B, qHtmp, e));
#else
  return 0;
#endif
}

MJH_DEVN void stage_factor_m(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  rptr qLD = MJH_F(B, qLD, e);      // holds M (stage_crb)
#if !MJH_LANE_MODE && MJH_W == 64
  if (pairs_euler_factor(M, B, e)) {
    // qH = M + h*diag(B) is factorised in the same pass as M (factor_ld_pair) while the LDS that will hold
    // the constraint arrays is still free, parked in its global home, and picked up again by stage_finish
    rptr qHt = MJH_F(B, qHtmp, e);
    rptr qHtD = MJH_F(B, qHtmpDiagInv, e);
    rptr qLDD = MJH_F(B, qLDiagInv, e);
    const real h = M.o.timestep;
    crptr qvel = MJH_F(B, qvel, e);
    MJH_FOR_LANES(k, M.s.nC) qHt[k] = qLD[k];
    wv_sync();
    MJH_FOR_LANES(i, M.s.nv) {
      real dd = poly_force_deriv(M.dof_damping_eff[i], M.dof_dampingpoly_eff + 2*i, qvel[i], 1);
      qHt[M.M_rowadr[i] + M.M_rownnz[i] - 1] += h * dd;
    }
    wv_sync();
    if (mjh_in_lds(qLD)) factor_ld_pair(M, mjh_local(qLD.p), qLDD, mjh_local(qHt.p), qHtD);
    else factor_ld_pair(M, qLD, qLDD, qHt, qHtD);
    rptr qHg = MJH_G(B, qH2, e);
    rptr qHgD = MJH_G(B, qH2DiagInv, e);
    MJH_FOR_LANES(k, M.s.nC) qHg[k] = qHt[k];
    MJH_FOR_LANES(i, M.s.nv) qHgD[i] = qHtD[i];
    wv_sync();
    return;
  }
#endif
  factor_ld(M, qLD, MJH_F(B, qLDiagInv, e));
}

// ------------------------------------------------------------------------------------------------
// x <- inv(L'DL) x, one vector                    (mj_solveLD, engine_core_smooth.c:2033-2109)
// ------------------------------------------------------------------------------------------------
template <class P0, class P1, class P2>
MJH_DEVN void solve_ld(MREF M, P0 x, P1 qLD, P2 diaginv) {
  const int nv = M.s.nv;
#if !MJH_LANE_MODE
  if (M.s.ld_fast && (MJH_W == 64 || M.s.nv <= MJH_W)) {
    if (mjh_in_lds(qLD)) solve_ld_fast(M, x, mjh_local(qLD.p), diaginv);
    else solve_ld_fast(M, x, qLD, diaginv);
    return;
  }
#endif
  // x <- L^-T x : row i scatters into its ancestors (independent targets)
  for (int q = M.s.nldrows - 1; q >= 0; q--) {
    const int i = M.ld_rows[q];
    int nnz = M.M_rownnz[i];
    real xi = x[i];
    if (xi != 0) {
      int start = M.M_rowadr[i];
      MJH_FOR_LANES(a, nnz - 1) x[M.M_colind[start + a]] -= qLD[start + a] * xi;
    }
    wv_sync();
  }
  // x <- D^-1 x
  MJH_FOR_LANES(i, nv) x[i] *= diaginv[i];
  wv_sync();
  // x <- L^-1 x : row i gathers from its ancestors (mju_dotSparse association)
  for (int q = 0; q < M.s.nldrows; q++) {
    const int i = M.ld_rows[q];
    int nnz = M.M_rownnz[i];
    if (wv_lane() == 0) {
      int adr = M.M_rowadr[i];
      x[i] -= dot_sparse_ref(qLD + adr, x, nnz - 1, M.M_colind + adr);
    }
    wv_sync();
  }
}

// two solves at once where the paired routine applies (both factors in the same address space),
// else one after the other
template <class XA, class XB, class QA, class QB, class DA, class DB>
MJH_DEV void solve_ld_two(MREF M, XA xa, QA qLDa, DA dinva, XB xb, QB qLDb, DB dinvb) {
#if !MJH_LANE_MODE && MJH_W == 64
  if (M.s.ld_fast && M.s.nv <= 32) {
    const int la = mjh_in_lds(qLDa), lb = mjh_in_lds(qLDb);
    if (la && lb) { solve_ld_pair(M, xa, mjh_local(qLDa.p), dinva, xb, mjh_local(qLDb.p), dinvb); return; }
    if (!la && !lb) { solve_ld_pair(M, xa, qLDa, dinva, xb, qLDb, dinvb); return; }
  }
#endif
  solve_ld(M, xa, qLDa, dinva);
  solve_ld(M, xb, qLDb, dinvb);
}

// ------------------------------------------------------------------------------------------------
// mj_comVel                                      (engine_core_smooth.c:2179-2239)
// ------------------------------------------------------------------------------------------------
// res(6) = sum_r dof[r](6) * vec[r], r < n          (mju_mulDofVec, engine_util_spatial.c:466)
template <class P0, class P1, class P2>
MJH_DEV void mul_dof_vec(P0 res, P1 dof, P2 vec, int n) {
  if (n == 1) {
    for (int k = 0; k < 6; k++) res[k] = dof[k]*vec[0];
  } else {
    for (int k = 0; k < 6; k++) res[k] = 0;
    for (int r = 0; r < n; r++) {
      real t = vec[r];
      if (t != 0) for (int k = 0; k < 6; k++) res[k] += dof[6*r + k]*t;
    }
  }
}

MJH_DEVN void stage_comvel(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr qvel = MJH_F(B, qvel, e);
  crptr cdof = MJH_F(B, cdof, e);
  rptr cvel = MJH_F(B, cvel, e);
  rptr cdof_dot = MJH_F(B, cdof_dot, e);
  // A body's dofs come in groups that enter its velocity one after the other: a hinge / slide dof, the three
  // rotational dofs of a ball or free joint, the three translational dofs of a free joint.  Only the
  // running sum v (parent velocity, then group after group) follows the tree.  So:
  //  (1) every group's term cdof * qvel at once, parked in the cdof_dot slot of the group's first dof;
  //  (2) the level loop adds the terms, a lane per (body of the level, component), and leaves the velocity
  //      BEFORE each group in that slot;
  //  (3) every dof's cdof_dot = v_before x cdof at once (zero for free translations).
  // Operands and operation order per component are those of the body-by-body recursion.
  // group start of dof d: itself for hinge / slide, the joint's first (or fourth) dof otherwise
  auto group_of = [&](int d, int& first, int& n, int& spins) {
    const int jt = M.dof_jnttype[d];
    if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
      const int off = d - M.jnt_dofadr[M.dof_jntid[d]];
      first = d - off % 3;
      n = 3;
      spins = !(jt == MJH_JNT_FREE && off < 3);
    } else { first = d; n = 1; spins = 1; }
  };
  MJH_FOR_LANES(d, s.nv) {
    int first, n, spins;
    group_of(d, first, n, spins);
    if (first == d) {
      real tmp[6];
      mul_dof_vec(tmp, cdof + 6*d, qvel + d, n);
      for (int q = 0; q < 6; q++) cdof_dot[6*d + q] = tmp[q];
    }
  }
  if (wv_lane() == 0) for (int k = 0; k < 6; k++) cvel[k] = 0;
  wv_sync();
  for (int L = 1; L < s.nlevel; L++) {
    int a0 = M.body_level_adr[L], a1 = M.body_level_adr[L+1];
    MJH_FOR_LANES(w, (a1 - a0)*6) {
      const int k = w / 6, q = w - 6*k;
      const int i = M.body_level_ids[a0 + k];
      real v = cvel[6*M.body_parentid[i] + q];
      const int dofnum = M.body_dofnum[i], bda = M.body_dofadr[i];
      for (int j = 0; j < dofnum; ) {
        const int jt = M.dof_jnttype[bda + j];
        const int n = (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) ? 3 : 1;
        const real t = cdof_dot[6*(bda + j) + q];
        cdof_dot[6*(bda + j) + q] = v;
        v += t;
        j += n;
      }
      cvel[6*i + q] = v;
    }
    wv_sync();
  }
  // (rounds from the top: the dofs of a group that sit in a later round than its first dof read the slot
  // before the first dof's round overwrites it; inside a round the barrier separates reads from writes)
  for (int d0 = ((s.nv - 1) / MJH_W) * MJH_W; d0 >= 0; d0 -= MJH_W) {
    const int d = d0 + wv_lane();
    real out[6] = {0, 0, 0, 0, 0, 0};
    if (d < s.nv) {
      int first, n, spins;
      group_of(d, first, n, spins);
      if (spins) {
        real vb[6];
        for (int q = 0; q < 6; q++) vb[q] = cdof_dot[6*first + q];
        sp_cross_motion(out, vb, cdof + 6*d);
      }
    }
    wv_sync();          // the group's first slot is read by its other dofs before it is overwritten
    if (d < s.nv) for (int q = 0; q < 6; q++) cdof_dot[6*d + q] = out[q];
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// mj_passive: joint springs, dof dampers, tendon spring-dampers   (engine_passive.c:655-842,1047)
// ------------------------------------------------------------------------------------------------
// mju_polyForce, engine_util_misc.c:2314 (mjNPOLY = 2)
template <class P0>
MJH_DEV real poly_force(real linear, P0 poly, real x, int odd) {
  x = odd ? fabs(x) : x;
  real res = linear;
  real xpow = 1;
  for (int i = 0; i < 2; i++) {
    xpow *= x;
    res += poly[i] * xpow;
  }
  return res;
}
// mjd_xPolyForce, engine_util_misc.c:2329
template <class P0>
MJH_DEV real poly_force_deriv(real linear, P0 poly, real x, int odd) {
  x = odd ? fabs(x) : x;
  real res = linear;
  real xpow = 1;
  for (int i = 0; i < 2; i++) {
    xpow *= x;
    res += (i+2) * poly[i] * xpow;
  }
  return res;
}

// ------------------------------------------------------------------------------------------------
// Ellipsoid fluid model, per geom (round 6).  mj_ellipsoidFluidModel / mj_addedMassForces / mj_viscousForces
// (engine_passive.c:1213-1410) for the wrench, mjd_ellipsoidFluid and its components (engine_derivative.c:2531-2880)
// for the 6 x 6 derivative B of the local wrench (torque; force) by the local velocity (angular; linear) that the
// implicit integrators put into qDeriv as J' B J.  One lane per geom; results in fluid_geom: the world-frame wrench
// (torque, force), then B row-major.  Every expression keeps the reference's order of operations.
// ------------------------------------------------------------------------------------------------
MJH_DEV real el_pow2(real v) { return v*v; }
MJH_DEV real el_pow4(real v) { return (v*v)*(v*v); }
MJH_DEV real el_max_moment(const real* sz, int dir) {
  const real d0 = sz[dir], d1 = sz[(dir + 1) % 3], d2 = sz[(dir + 2) % 3];
  return 8.0/15.0 * MJH_PI * d0 * el_pow4(r_max(d1, d2));
}
// Da = d(a x b)/da, Db = d(a x b)/db (row-major 3 x 3; mjd_cross)
MJH_DEV void el_dcross(const real* a, const real* b, real* Da, real* Db) {
  for (int k = 0; k < 9; k++) { Da[k] = 0; Db[k] = 0; }
  Da[1] = b[2]; Da[2] = -b[1]; Da[3] = -b[2]; Da[5] = b[0]; Da[6] = b[1]; Da[7] = -b[0];
  Db[1] = -a[2]; Db[2] = a[1]; Db[3] = a[2]; Db[5] = -a[0]; Db[6] = -a[1]; Db[7] = a[0];
}
// 3 x 3 block D (as the reference fills it) into quadrant (column block cq, row block rq) of the 6 x 6 matrix:
// element D[3 i + j] lands at Bm[6 (3 cq + i) + 3 rq + j]   (addToQuadrant)
MJH_DEV void el_add_quadrant(real* Bm, const real* D, int cq, int rq) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Bm[6*(3*cq + i) + 3*rq + j] += D[3*i + j];
}
MJH_DEVN void ellipsoid_fluid_geoms(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  crptr cvel = MJH_F(B, cvel, e);
  crptr com = MJH_F(B, subtree_com, e);
  rptr out = MJH_G(B, fluid_geom, e);
  const real dens = M.o.density, visc = M.o.viscosity;
  const int want_B = MJH_HAS(MJH_FT_IMPLICIT) && M.o.integrator >= MJH_INT_IMPLICIT;
  MJH_FOR_LANES(g, s.ngeom) {
    const int b = M.geom_bodyid[g];
    for (int k = 0; k < 42; k++) out[42*g + k] = 0;
    if (!M.body_ellipsoid[b] || M.body_mass[b] < MJH_MINVAL) continue;
    auto fc = M.geom_fluid + 12*g;
    const real icoef = fc[0], blunt = fc[1], slender = fc[2], angc = fc[3], kutta = fc[4], magnus = fc[5];
    if (icoef == 0) continue;
    const real vm[3] = {fc[6], fc[7], fc[8]}, vi[3] = {fc[9], fc[10], fc[11]};
    const real sz[3] = {M.geom_semiaxes[3*g], M.geom_semiaxes[3*g + 1], M.geom_semiaxes[3*g + 2]};
    // local 6D velocity of the geom frame (mj_objectVelocity, flg_local) minus the wind
    real lvel[6] = {0, 0, 0, 0, 0, 0}, lwind[6];
    crptr cref = com + 3*M.body_rootid[b];
    if (M.body_dofnum[M.body_weldid[b]] != 0) {
      real dif[3], cr[3], tran[6];
      v3_sub(dif, gx + 3*g, cref);
      for (int k = 0; k < 6; k++) tran[k] = cvel[6*b + k];
      v3_cross(cr, dif, cvel + 6*b);
      tran[3] = cvel[6*b + 3] - cr[0]; tran[4] = cvel[6*b + 4] - cr[1]; tran[5] = cvel[6*b + 5] - cr[2];
      m3_multvec(lvel, gm + 9*g, tran);
      m3_multvec(lvel + 3, gm + 9*g, tran + 3);
    }
    {
      real wind[6] = {0, 0, 0, M.o.wind[0], M.o.wind[1], M.o.wind[2]};
      real dif[3], cr[3], tran[6];
      v3_sub(dif, gx + 3*g, cref);
      for (int k = 0; k < 6; k++) tran[k] = wind[k];
      v3_cross(cr, dif, wind);
      tran[3] = wind[3] - cr[0]; tran[4] = wind[4] - cr[1]; tran[5] = wind[5] - cr[2];
      m3_multvec(lwind, gm + 9*g, tran);
      m3_multvec(lwind + 3, gm + 9*g, tran + 3);
    }
    lvel[3] -= lwind[3]; lvel[4] -= lwind[4]; lvel[5] -= lwind[5];
    const real lin[3] = {lvel[3], lvel[4], lvel[5]}, ang[3] = {lvel[0], lvel[1], lvel[2]};
    real lfrc[6] = {0, 0, 0, 0, 0, 0};
    // ---- added mass (mj_addedMassForces)
    const real plin[3] = {dens*vm[0]*lin[0], dens*vm[1]*lin[1], dens*vm[2]*lin[2]};
    const real pang[3] = {dens*vi[0]*ang[0], dens*vi[1]*ang[1], dens*vi[2]*ang[2]};
    {
      real f[3], t1[3], t2[3];
      v3_cross(f, plin, ang);
      v3_cross(t1, plin, lin);
      v3_cross(t2, pang, ang);
      for (int k = 0; k < 3; k++) { lfrc[k] += t1[k]; }
      for (int k = 0; k < 3; k++) { lfrc[k] += t2[k]; }
      for (int k = 0; k < 3; k++) { lfrc[3 + k] += f[k]; }
    }
    // ---- lift and drag (mj_viscousForces)
    const real volume = 4.0/3.0 * MJH_PI * sz[0] * sz[1] * sz[2];
    const real d_max = r_max(r_max(sz[0], sz[1]), sz[2]);
    const real d_min = r_min(r_min(sz[0], sz[1]), sz[2]);
    const real d_mid = sz[0] + sz[1] + sz[2] - d_max - d_min;
    const real A_max = MJH_PI * d_max * d_mid;
    real magf[3];
    v3_cross(magf, ang, lin);
    for (int k = 0; k < 3; k++) magf[k] *= magnus * dens * volume;
    const real proj_denom = el_pow4(sz[1]*sz[2])*el_pow2(lin[0]) + el_pow4(sz[2]*sz[0])*el_pow2(lin[1]) + el_pow4(sz[0]*sz[1])*el_pow2(lin[2]);
    const real proj_num = el_pow2(sz[1]*sz[2]*lin[0]) + el_pow2(sz[2]*sz[0]*lin[1]) + el_pow2(sz[0]*sz[1]*lin[2]);
    const real A_proj = MJH_PI * sqrt(proj_denom/r_max(MJH_MINVAL, proj_num));
    const real nrm[3] = {el_pow2(sz[1]*sz[2])*lin[0], el_pow2(sz[2]*sz[0])*lin[1], el_pow2(sz[0]*sz[1])*lin[2]};
    const real linnorm = sqrt(lin[0]*lin[0] + lin[1]*lin[1] + lin[2]*lin[2]);
    const real cos_alpha = proj_num / r_max(MJH_MINVAL, linnorm * proj_denom);
    real circ[3], kutf[3];
    v3_cross(circ, nrm, lin);
    for (int k = 0; k < 3; k++) circ[k] *= kutta * dens * cos_alpha * A_proj;
    v3_cross(kutf, circ, lin);
    const real eqD = 2.0/3.0 * (sz[0] + sz[1] + sz[2]);
    const real lin_force_coef = 3.0 * MJH_PI * eqD;
    const real lin_torq_coef = MJH_PI * eqD*eqD*eqD;
    const real I_max = 8.0/15.0 * MJH_PI * d_mid * el_pow4(d_max);
    const real II[3] = {el_max_moment(sz, 0), el_max_moment(sz, 1), el_max_moment(sz, 2)};
    const real momv[3] = {ang[0]*(angc*II[0] + slender*(I_max - II[0])), ang[1]*(angc*II[1] + slender*(I_max - II[1])),
                          ang[2]*(angc*II[2] + slender*(I_max - II[2]))};
    const real drag_lin = visc*lin_force_coef + dens*linnorm*(A_proj*blunt + slender*(A_max - A_proj));
    const real drag_ang = visc*lin_torq_coef + dens*sqrt(momv[0]*momv[0] + momv[1]*momv[1] + momv[2]*momv[2]);
    for (int k = 0; k < 3; k++) lfrc[k] -= drag_ang*ang[k];
    for (int k = 0; k < 3; k++) lfrc[3 + k] += magf[k] + kutf[k] - drag_lin*lin[k];
    for (int k = 0; k < 6; k++) lfrc[k] = lfrc[k]*icoef;
    real w6[6];
    m3_mulvec(w6, gm + 9*g, lfrc);
    m3_mulvec(w6 + 3, gm + 9*g, lfrc + 3);
    for (int k = 0; k < 6; k++) out[42*g + k] = w6[k];
    if (!want_B) continue;

    // ---- the derivative (mjd_ellipsoidFluid :2832-2850: magnus, kutta, viscous drag, viscous torque, added mass)
    real Bm[36], D[9], Da[9], Db[9];
    for (int k = 0; k < 36; k++) Bm[k] = 0;
    {  // mjd_magnus_force
      const real mc = magnus * dens * volume;
      const real l3[3] = {mc*lvel[3], mc*lvel[4], mc*lvel[5]}, a3[3] = {mc*lvel[0], mc*lvel[1], mc*lvel[2]};
      el_dcross(a3, l3, Da, Db);
      el_add_quadrant(Bm, Da, 1, 0);
      el_add_quadrant(Bm, Db, 1, 1);
    }
    const real qa = el_pow2(sz[1]*sz[2]), qb = el_pow2(sz[2]*sz[0]), qc = el_pow2(sz[0]*sz[1]);
    const real aa = qa*qa, bb = qb*qb, cc = qc*qc;
    {  // mjd_kutta_lift
      const real x = lvel[3], y = lvel[4], z = lvel[5];
      const real xx = x*x, yy = y*y, zz = z*z, xy = x*y, yz = y*z, xz = x*z;
      const real pden = aa*xx + bb*yy + cc*zz;
      const real pnum = qa*xx + qb*yy + qc*zz;
      const real norm2 = xx + yy + zz;
      const real df_denom = MJH_PI * kutta * dens / r_max(MJH_MINVAL, sqrt(pden * pnum * norm2));
      const real dfx = yy*(qa - qb) + zz*(qa - qc);
      const real dfy = xx*(qb - qa) + zz*(qb - qc);
      const real dfz = xx*(qc - qa) + yy*(qc - qb);
      const real proj_term = pnum / r_max(MJH_MINVAL, pden);
      const real cos_term = pnum / r_max(MJH_MINVAL, norm2);
      D[0] = qa - qa; D[1] = qb - qa; D[2] = qc - qa;
      D[3] = qa - qb; D[4] = qb - qb; D[5] = qc - qb;
      D[6] = qa - qc; D[7] = qb - qc; D[8] = qc - qc;
      for (int k = 0; k < 9; k++) D[k] = D[k]*(2*pnum);
      const real inner[3] = {aa*proj_term - qa + cos_term, bb*proj_term - qb + cos_term, cc*proj_term - qc + cos_term};
      for (int k = 0; k < 3; k++) { D[k] += inner[k]*dfx; D[3 + k] += inner[k]*dfy; D[6 + k] += inner[k]*dfz; }
      D[0] *= xx; D[1] *= xy; D[2] *= xz;
      D[3] *= xy; D[4] *= yy; D[5] *= yz;
      D[6] *= xz; D[7] *= yz; D[8] *= zz;
      D[0] -= dfx*pnum; D[4] -= dfy*pnum; D[8] -= dfz*pnum;
      for (int k = 0; k < 9; k++) D[k] = D[k]*df_denom;
      el_add_quadrant(Bm, D, 1, 1);
    }
    {  // mjd_viscous_drag
      const real x = lvel[3], y = lvel[4], z = lvel[5];
      const real xx = x*x, yy = y*y, zz = z*z, xy = x*y, yz = y*z, xz = x*z;
      const real pden = aa*xx + bb*yy + cc*zz;
      const real pnum = qa*xx + qb*yy + qc*zz;
      const real dA_coef = MJH_PI / r_max(MJH_MINVAL, sqrt(pnum*pnum*pnum * pden));
      const real Ap = MJH_PI * sqrt(pden/r_max(MJH_MINVAL, pnum));
      const real norm = sqrt(xx + yy + zz);
      const real inv_norm = 1.0 / r_max(MJH_MINVAL, norm);
      const real lin_coef = visc * 3.0 * MJH_PI * eqD;
      const real quad_coef = dens * (Ap*blunt + slender*(A_max - Ap));
      const real Aproj_coef = dens * norm * (blunt - slender);
      const real dAp[3] = {Aproj_coef * dA_coef * qa * x * (qb * yy * (qa - qb) + qc * zz * (qa - qc)),
                           Aproj_coef * dA_coef * qb * y * (qa * xx * (qb - qa) + qc * zz * (qb - qc)),
                           Aproj_coef * dA_coef * qc * z * (qa * xx * (qc - qa) + qb * yy * (qc - qb))};
      D[0] = xx; D[1] = xy; D[2] = xz; D[3] = xy; D[4] = yy; D[5] = yz; D[6] = xz; D[7] = yz; D[8] = zz;
      const real inner = xx + yy + zz;
      D[0] += inner; D[4] += inner; D[8] += inner;
      const real sc = -quad_coef*inv_norm;
      for (int k = 0; k < 9; k++) D[k] = D[k]*sc;
      for (int k = 0; k < 3; k++) { D[k] += dAp[k]*(-x); D[3 + k] += dAp[k]*(-y); D[6 + k] += dAp[k]*(-z); }
      D[0] -= lin_coef; D[4] -= lin_coef; D[8] -= lin_coef;
      el_add_quadrant(Bm, D, 1, 1);
    }
    {  // mjd_viscous_torque
      const real x = lvel[0], y = lvel[1], z = lvel[2];
      const real mcf[3] = {angc*II[0] + slender*(I_max - II[0]), angc*II[1] + slender*(I_max - II[1]), angc*II[2] + slender*(I_max - II[2])};
      const real mv[3] = {x*mcf[0], y*mcf[1], z*mcf[2]};
      const real density = dens / r_max(MJH_MINVAL, sqrt(mv[0]*mv[0] + mv[1]*mv[1] + mv[2]*mv[2]));
      const real msq[3] = {-density * x * mcf[0] * mcf[0], -density * y * mcf[1] * mcf[1], -density * z * mcf[2] * mcf[2]};
      const real lin_coef = visc * lin_torq_coef;
      for (int k = 0; k < 9; k++) D[k] = 0;
      const real dg = x*msq[0] + y*msq[1] + z*msq[2] - lin_coef;
      D[0] = dg; D[4] = dg; D[8] = dg;
      for (int k = 0; k < 3; k++) { D[k] += msq[k]*x; D[3 + k] += msq[k]*y; D[6 + k] += msq[k]*z; }
      el_add_quadrant(Bm, D, 0, 0);
    }
    {  // mjd_addedMassForces
      el_dcross(pang, ang, Da, Db);
      el_add_quadrant(Bm, Db, 0, 0);
      for (int k = 0; k < 9; k++) Da[k] *= dens * vi[k % 3];
      el_add_quadrant(Bm, Da, 0, 0);
      el_dcross(plin, lin, Da, Db);
      el_add_quadrant(Bm, Db, 0, 1);
      for (int k = 0; k < 9; k++) Da[k] *= dens * vm[k % 3];
      el_add_quadrant(Bm, Da, 0, 1);
      el_dcross(plin, ang, Da, Db);
      el_add_quadrant(Bm, Db, 1, 0);
      for (int k = 0; k < 9; k++) Da[k] *= dens * vm[k % 3];
      el_add_quadrant(Bm, Da, 1, 1);
    }
    // (implicitfast symmetrises B except on standalone free bodies: mju_symmetrize, 0.5 (B + B'))
    if (M.o.integrator == MJH_INT_IMPLICITFAST) {
      int freebody = 0;
      if (M.body_jntnum[b] == 1 && M.jnt_freebody[M.body_jntadr[b]]) freebody = 1;
      if (!freebody)
        for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++) { const real v = 0.5*(Bm[6*i + j] + Bm[6*j + i]); Bm[6*i + j] = v; Bm[6*j + i] = v; }
    }
    for (int k = 0; k < 36; k++) out[42*g + 6 + k] = Bm[k];
  }
  wv_sync();
}

MJH_DEVN void stage_passive(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr qpos = MJH_F(B, qpos, e);
  crptr qvel = MJH_F(B, qvel, e);
  rptr fs = MJH_F(B, qfrc_spring, e);
  rptr fd = MJH_F(B, qfrc_damper, e);
  rptr fp = MJH_F(B, qfrc_passive, e);
  const int dsbl = M.o.disableflags;
  const int enbl_spring = !(dsbl & (1<<5)), enbl_damper = !(dsbl & (1<<6));

  MJH_FOR_LANES(i, s.nv) { fs[i] = 0; fd[i] = 0; fp[i] = 0; }
  wv_sync();
  if (!enbl_spring && !enbl_damper) return;

  if (enbl_spring) {
    MJH_FOR_LANES(j, s.njnt) {
      real k0 = M.jnt_stiffness[j];
      auto sp = M.jnt_stiffnesspoly + 2*j;
      if (k0 == 0 && sp[0] == 0 && sp[1] == 0) continue;
      int padr = M.jnt_qposadr[j], dadr = M.jnt_dofadr[j];
      int jt = M.jnt_type[j];
      if (jt == MJH_JNT_FREE) {
        real dif[3];
        v3_sub(dif, qpos + padr, M.qpos_spring + padr);
        real r = v3_norm(dif);
        real k = poly_force(k0, sp, r, 0);
        v3_addtoscl(fs + dadr, dif, -k);
        dadr += 3; padr += 3;
      }
      if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
        real dif[3], quat[4];
        q_copy(quat, qpos + padr);
        q_normalize(quat);
        q_sub(dif, quat, M.qpos_spring + padr);
        real r = v3_norm(dif);
        real k = poly_force(k0, sp, r, 0);
        v3_addtoscl(fs + dadr, dif, -k);
      } else {
        real x = qpos[padr] - M.qpos_spring[padr];
        fs[dadr] = -x * poly_force(k0, sp, x, 0);
      }
    }
  }
  if (enbl_damper) {
    MJH_FOR_LANES(i, s.nv) {
      real damping = M.dof_damping_eff[i];
      auto dp = M.dof_dampingpoly_eff + 2*i;
      if (damping != 0 || dp[0] != 0 || dp[1] != 0) {
        real v = qvel[i];
        fd[i] = -v * poly_force(damping, dp, v, 1);
      }
    }
  }
  wv_sync();
  if (MJH_HAS(MJH_FT_FLEX) && s.nflex) flex_passive(M, B, e, enbl_spring, enbl_damper);

  // tendon spring-dampers accumulate into shared dofs: keep the reference's tendon order
  if (s.ntendon && wv_lane() == 0) {
    crptr tl = MJH_F(B, ten_length, e);
    crptr tv = MJH_F(B, ten_velocity, e);
    crptr tJ = MJH_F(B, ten_J, e);
    for (int i = 0; i < s.ntendon; i++) {
      real stiffness = enbl_spring ? M.tendon_stiffness[i] : 0;
      auto sp = M.tendon_stiffnesspoly + 2*i;
      real damping = enbl_damper ? M.tendon_damping_eff[i] : 0;
      real dp[2] = {0, 0};
      if (enbl_damper) { dp[0] = M.tendon_dampingpoly_eff[2*i]; dp[1] = M.tendon_dampingpoly_eff[2*i+1]; }
      if (stiffness == 0 && (!enbl_spring || (sp[0] == 0 && sp[1] == 0)) &&
          damping == 0 && dp[0] == 0 && dp[1] == 0) continue;
      real length = tl[i];
      real lower = M.tendon_lengthspring[2*i], upper = M.tendon_lengthspring[2*i+1];
      real x = (length > upper) ? length - upper : (length < lower) ? length - lower : 0;
      real frc_spring = enbl_spring ? -x * poly_force(stiffness, sp, x, 0) : 0;
      real v = tv[i];
      real frc_damper = enbl_damper ? -v * poly_force(damping, dp, v, 1) : 0;
      if (frc_spring != 0 || frc_damper != 0) {
        int a0 = M.ten_J_rowadr[i], a1 = a0 + M.ten_J_rownnz[i];
        for (int j = a0; j < a1; j++) {
          int k = M.ten_J_colind[j];
          fs[k] += tJ[j] * frc_spring;
          fd[k] += tJ[j] * frc_damper;
        }
      }
    }
  }
  wv_sync();
  MJH_FOR_LANES(i, s.nv) fp[i] = fs[i] + fd[i];
  wv_sync();
  // fluid forces, inertia-box model (mj_fluid / mj_inertiaBoxFluidModel, engine_passive.c:871-903,
  // :1154-1210): viscous and quadratic drag on the equivalent inertia box of every body, in the
  // body's inertial frame, applied at its COM through mj_applyFT
  if (MJH_HAS(MJH_FT_PASSIVEMISC) && M.o.has_fluid && (M.o.viscosity != 0 || M.o.density != 0)) {
    crptr xipos = MJH_F(B, xipos, e);
    crptr ximat = MJH_F(B, ximat, e);
    crptr cvel = MJH_F(B, cvel, e);
    crptr cdof = MJH_F(B, cdof, e);
    crptr com = MJH_F(B, subtree_com, e);
    rptr bf = MJH_G(B, fluid_frc, e);
    MJH_FOR_LANES(i, s.nbody) {
      real out[6] = {0, 0, 0, 0, 0, 0};
      real coef[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const real mass = M.body_mass[i];
      if (!(mass < MJH_MINVAL) && !(s.ngeom_fluid && M.body_ellipsoid[i])) {
        auto inertia = M.body_inertia + 3*i;
        real box[3];
        box[0] = sqrt(r_max(MJH_MINVAL, (inertia[1] + inertia[2] - inertia[0])) / mass * 6.0);
        box[1] = sqrt(r_max(MJH_MINVAL, (inertia[0] + inertia[2] - inertia[1])) / mass * 6.0);
        box[2] = sqrt(r_max(MJH_MINVAL, (inertia[0] + inertia[1] - inertia[2])) / mass * 6.0);
        // local 6D velocity of the inertial frame (mj_objectVelocity, flg_local) minus the wind
        real lvel[6] = {0, 0, 0, 0, 0, 0}, lwind[6];
        crptr cref = com + 3*M.body_rootid[i];
        if (M.body_dofnum[M.body_weldid[i]] != 0) {
          real dif[3], cr[3], tran[6];
          v3_sub(dif, xipos + 3*i, cref);
          for (int k = 0; k < 6; k++) tran[k] = cvel[6*i + k];
          v3_cross(cr, dif, cvel + 6*i);
          tran[3] = cvel[6*i + 3] - cr[0]; tran[4] = cvel[6*i + 4] - cr[1]; tran[5] = cvel[6*i + 5] - cr[2];
          m3_multvec(lvel, ximat + 9*i, tran);
          m3_multvec(lvel + 3, ximat + 9*i, tran + 3);
        }
        {
          real wind[6] = {0, 0, 0, M.o.wind[0], M.o.wind[1], M.o.wind[2]};
          real dif[3], cr[3], tran[6];
          v3_sub(dif, xipos + 3*i, cref);
          for (int k = 0; k < 6; k++) tran[k] = wind[k];
          v3_cross(cr, dif, wind);
          tran[3] = wind[3] - cr[0]; tran[4] = wind[4] - cr[1]; tran[5] = wind[5] - cr[2];
          m3_multvec(lwind, ximat + 9*i, tran);
          m3_multvec(lwind + 3, ximat + 9*i, tran + 3);
        }
        lvel[3] -= lwind[3]; lvel[4] -= lwind[4]; lvel[5] -= lwind[5];
        real lfrc[6] = {0, 0, 0, 0, 0, 0};
        const real visc = M.o.viscosity, dens = M.o.density;
        if (visc > 0) {
          const real diam = (box[0] + box[1] + box[2])/3.0;
          const real ka = -MJH_PI*diam*diam*diam*visc, kl = -3.0*MJH_PI*diam*visc;
          for (int k = 0; k < 3; k++) { lfrc[k] = lvel[k]*ka; lfrc[3 + k] = lvel[3 + k]*kl; }
        }
        if (dens > 0) {
          lfrc[3] -= 0.5*dens*box[1]*box[2]*fabs(lvel[3])*lvel[3];
          lfrc[4] -= 0.5*dens*box[0]*box[2]*fabs(lvel[4])*lvel[4];
          lfrc[5] -= 0.5*dens*box[0]*box[1]*fabs(lvel[5])*lvel[5];
          lfrc[0] -= dens*box[0]*(box[1]*box[1]*box[1]*box[1]+box[2]*box[2]*box[2]*box[2])*fabs(lvel[0])*lvel[0]/64.0;
          lfrc[1] -= dens*box[1]*(box[0]*box[0]*box[0]*box[0]+box[2]*box[2]*box[2]*box[2])*fabs(lvel[1])*lvel[1]/64.0;
          lfrc[2] -= dens*box[2]*(box[0]*box[0]*box[0]*box[0]+box[1]*box[1]*box[1]*box[1])*fabs(lvel[2])*lvel[2]/64.0;
        }
        m3_mulvec(out, ximat + 9*i, lfrc);
        m3_mulvec(out + 3, ximat + 9*i, lfrc + 3);
        // d(lfrc) / d(lvel), one scalar per local axis (mjd_inertiaBoxFluid, engine_derivative.c:2953-3035): read by the
        // implicit integrators at the end of the step (same qvel)
        if (MJH_HAS(MJH_FT_IMPLICIT)) {
          if (visc > 0) {
            const real diam = (box[0] + box[1] + box[2])/3.0;
            coef[0] = -MJH_PI*diam*diam*diam*visc;
            coef[1] = -3.0*MJH_PI*diam*visc;
          }
          if (dens > 0) {
            coef[2] = -dens*box[0]*(box[1]*box[1]*box[1]*box[1]+box[2]*box[2]*box[2]*box[2])*2*fabs(lvel[0])/64.0;
            coef[3] = -dens*box[1]*(box[0]*box[0]*box[0]*box[0]+box[2]*box[2]*box[2]*box[2])*2*fabs(lvel[1])/64.0;
            coef[4] = -dens*box[2]*(box[0]*box[0]*box[0]*box[0]+box[1]*box[1]*box[1]*box[1])*2*fabs(lvel[2])/64.0;
            coef[5] = -0.5*dens*box[1]*box[2]*2*fabs(lvel[3]);
            coef[6] = -0.5*dens*box[0]*box[2]*2*fabs(lvel[4]);
            coef[7] = -0.5*dens*box[0]*box[1]*2*fabs(lvel[5]);
          }
        }
      }
      for (int k = 0; k < 6; k++) bf[6*i + k] = out[k];
      if (MJH_HAS(MJH_FT_IMPLICIT)) for (int k = 0; k < 8; k++) bf[6*s.nbody + 8*i + k] = coef[k];
    }
    // ellipsoid model (mj_ellipsoidFluidModel, engine_passive.c:1213-1270), one lane per geom of a body that uses it
    if (s.ngeom_fluid) ellipsoid_fluid_geoms(M, B, e);
    wv_sync();
    MJH_FOR_LANES(j, s.nv) {
      real acc = 0;
      crptr cd = cdof + 6*j;
      for (int b = 0; b < s.nbody; b++) {
        if (M.body_mass[b] < MJH_MINVAL) continue;
        if (s.ngeom_fluid && M.body_ellipsoid[b]) {
          // (mj_applyFT per geom, at the geom's position: force part, then torque part)
          crptr gx = MJH_F(B, geom_xpos, e);
          crptr gw = MJH_G(B, fluid_geom, e);
          const int inch = (M.body_dofanc[b*s.nvw + (j >> 5)] >> (j & 31)) & 1;
          for (int g = M.body_geomadr[b]; g < M.body_geomadr[b] + M.body_geomnum[b]; g++) {
            if (M.geom_fluid[12*g] == 0) continue;
            real tf = 0, tt = 0;
            if (inch) {
              real off[3], cr[3];
              v3_sub(off, gx + 3*g, com + 3*M.body_rootid[b]);
              v3_cross(cr, cd, off);
              for (int r = 0; r < 3; r++) { const real f = gw[42*g + 3 + r]; if (f != 0) tf += (cd[3 + r] + cr[r])*f; }
              for (int r = 0; r < 3; r++) { const real t = gw[42*g + r]; if (t != 0) tt += cd[r]*t; }
            }
            acc += tf;
            acc += tt;
          }
          continue;
        }
        real tf = 0, tt = 0;
        if ((M.body_dofanc[b*s.nvw + (j >> 5)] >> (j & 31)) & 1) {
          real off[3], cr[3];
          v3_sub(off, xipos + 3*b, com + 3*M.body_rootid[b]);
          v3_cross(cr, cd, off);
          for (int r = 0; r < 3; r++) { const real f = bf[6*b + 3 + r]; if (f != 0) tf += (cd[3 + r] + cr[r])*f; }
          for (int r = 0; r < 3; r++) { const real t = bf[6*b + r]; if (t != 0) tt += cd[r]*t; }
        }
        acc += tf;
        acc += tt;
      }
      fp[j] += acc;
    }
    wv_sync();
  }
  // adhesion (mj_adhesion, engine_passive.c:982-1050 + :1104-1109): a constant attraction -adhesion along the normal of
  // every adhesive contact (active or in the gap), through the normal row of the contact Jacobian; the contacts in
  // order per dof, summed from zero (qfrc_adhesion) and then added to qfrc_passive -- after the fluid forces, before
  // gravity compensation
  if (MJH_HAS(MJH_FT_PASSIVEMISC) && M.o.has_adhesion && !(dsbl & (1<<4))) {
    const int ncon = MJH_F(B, counts, e)[MJH_C_NCON];
    crptr cdof = MJH_F(B, cdof, e);
    crptr com = MJH_F(B, subtree_com, e);
    MJH_FOR_LANES(j, s.nv) {
      real acc = 0;
      int any = 0;
      crptr cd = cdof + 6*j;
      for (int c = 0; c < ncon; c++) {
        const real adh = M.pair_adhesion[MJH_CON(B, con_pair, e, 1, c)[0]];
        if (adh == 0 || MJH_CON(B, con_exclude, e, 1, c)[0] > 1) continue;
        any = 1;
        ciptr cg = MJH_CON(B, con_geom, e, 2, c);
        crptr point = MJH_CON(B, con_pos, e, 3, c);
        crptr fr = MJH_CON(B, con_frame, e, 9, c);
        const int b1 = M.geom_bodyid[cg[0]], b2 = M.geom_bodyid[cg[1]];
        const int w1 = M.body_weldid[b1], w2 = M.body_weldid[b2];
        const int in1 = (M.body_dofanc[w1*s.nvw + (j >> 5)] >> (j & 31)) & 1;
        const int in2 = (M.body_dofanc[w2*s.nvw + (j >> 5)] >> (j & 31)) & 1;
        // translational point Jacobians of both bodies (mj_jac, engine_core_util.c:176), their difference,
        // its projection on the normal (mju_mulMatMat with zero-skip)
        real j1[3] = {0, 0, 0}, j2[3] = {0, 0, 0};
        if (in1) {
          real off[3], t[3];
          v3_sub(off, point, com + 3*M.body_rootid[b1]);
          v3_cross(t, cd, off);
          j1[0] = cd[3] + t[0]; j1[1] = cd[4] + t[1]; j1[2] = cd[5] + t[2];
        }
        if (in2) {
          real off[3], t[3];
          v3_sub(off, point, com + 3*M.body_rootid[b2]);
          v3_cross(t, cd, off);
          j2[0] = cd[3] + t[0]; j2[1] = cd[4] + t[1]; j2[2] = cd[5] + t[2];
        }
        real jn = 0;
        for (int q = 0; q < 3; q++) {
          const real t = fr[q];
          if (t != 0) jn += (j2[q] - j1[q])*t;
        }
        acc += jn*(-adh);
      }
      if (any) fp[j] += acc;
    }
    wv_sync();
  }
  // gravity compensation (mj_gravcomp, engine_passive.c:846-867 + :1112-1122): per compensated
  // body a force -gravity*mass*gravcomp at its COM, mapped through the point Jacobian (mj_applyFT)
  if (MJH_HAS(MJH_FT_PASSIVEMISC) && M.o.has_gravcomp && !(dsbl & (1<<7)) &&
      sqrt(M.o.gravity[0]*M.o.gravity[0] + M.o.gravity[1]*M.o.gravity[1] + M.o.gravity[2]*M.o.gravity[2]) != 0) {
    crptr xipos = MJH_F(B, xipos, e);
    crptr cdof = MJH_F(B, cdof, e);
    crptr com = MJH_F(B, subtree_com, e);
    MJH_FOR_LANES(j, s.nv) {
      real acc = 0;
      crptr cd = cdof + 6*j;
      for (int b = 1; b < s.nbody; b++) {
        const real gc = M.body_gravcomp[b];
        if (gc == 0) continue;
        real t = 0;
        if ((M.body_dofanc[b*s.nvw + (j >> 5)] >> (j & 31)) & 1) {
          const real scl = -(M.body_mass[b]*gc);
          real off[3], cr[3];
          v3_sub(off, xipos + 3*b, com + 3*M.body_rootid[b]);
          v3_cross(cr, cd, off);
          for (int r = 0; r < 3; r++) {
            const real f = M.o.gravity[r]*scl;
            if (f != 0) t += (cd[3 + r] + cr[r])*f;
          }
        }
        acc += t;
      }
      // (a joint with actuator-level compensation: the force waits for stage_actuation)
      if (s.nv_actgc && M.dof_actgravcomp[j]) MJH_G(B, qfrc_gravcomp, e)[j] = acc;
      else fp[j] += acc;
    }
    wv_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// mj_tendonBias (engine_core_smooth.c:2606-2641): qfrc += ten_J * armature * (d/dt(ten_J) . qvel), with
// mj_tendonDot (:1115-1260) for spatial tendons through sites (a fixed tendon's Jacobian is constant).  Per straight
// segment between sites on different bodies: dpnt = the unit direction, dvel = its time derivative;
//   d/dt(ten_J) = (JacDot(p1) - JacDot(p0))' dpnt + (Jac(p1) - Jac(p0))' dvel      (both over the divisor of the pulley branch)
// Every lane builds the two addends of the dofs it owns (mj_jacDot's / mj_jac's column, mju_mulMatTVec's row order with
// its skip of zero components); the contraction with qvel follows the reference: one running sum over the merged dof
// chain of the two bodies in ascending order, first term then second (sparse Jacobians), or mju_dot over all dofs (dense).
// ------------------------------------------------------------------------------------------------
template <class PB>
MJH_DEV void tendon_bias(MREF M, BREF B, int e, PB bias) {
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  crptr qvel = MJH_F(B, qvel, e);
  crptr cdof = MJH_F(B, cdof, e);
  crptr cdof_dot = MJH_F(B, cdof_dot, e);
  crptr cvel = MJH_F(B, cvel, e);
  crptr sx = MJH_F(B, site_xpos, e);
  crptr com = MJH_F(B, subtree_com, e);
  crptr tJ = MJH_F(B, ten_J, e);
  rptr col = MJH_G(B, scratch, e) + 8*s.nefcmax;        // [2*nv]: the two addends per dof (scratch holds 8*nv there)
  for (int t = 0; t < s.ntendon; t++) {
    const real arm = M.tendon_armature_eff[t];
    const int adr = M.tendon_adr[t], num = M.tendon_num[t];
    if (!arm || M.wrap_type[adr] == 1) continue;          // (mjWRAP_JOINT: a fixed tendon has zero Jdot)
    real res = 0, divisor = 1;
    int j = 0;
    while (j < num - 1) {
      const int type0 = M.wrap_type[adr + j], type1 = M.wrap_type[adr + j + 1];
      if (type0 == 2 || type1 == 2) {                     // mjWRAP_PULLEY
        if (type0 == 2) divisor = M.wrap_prm[adr + j];
        j++;
        continue;
      }
      const int id0 = M.wrap_objid[adr + j], id1 = M.wrap_objid[adr + j + 1];
      const int b0 = M.site_bodyid[id0], b1 = M.site_bodyid[id1];
      if (b0 != b1) {
        real p0[3], p1[3], v0[3], v1[3], off0[3], off1[3];
        v3_copy(p0, sx + 3*id0);
        v3_copy(p1, sx + 3*id1);
        v3_sub(off0, p0, com + 3*M.body_rootid[b0]);
        v3_sub(off1, p1, com + 3*M.body_rootid[b1]);
        // mj_objectVelocity(site, flg_local = 0): linear part of mju_transformSpatial(cvel[body], site, subtree_com[root])
        { real cr[3]; v3_cross(cr, off0, cvel + 6*b0); v3_sub(v0, cvel + 6*b0 + 3, cr); }
        { real cr[3]; v3_cross(cr, off1, cvel + 6*b1); v3_sub(v1, cvel + 6*b1 + 3, cr); }
        real dpnt[3], dvel[3];
        v3_sub(dpnt, p1, p0);
        const real norm = v3_normalize(dpnt);
        v3_sub(dvel, v1, v0);
        const real dt = dpnt[0]*dvel[0] + dpnt[1]*dvel[1] + dpnt[2]*dvel[2];
        for (int r = 0; r < 3; r++) dvel[r] += dpnt[r]*(-dt);
        const real sc = norm > MJH_MINVAL ? 1/norm : 0;
        for (int r = 0; r < 3; r++) dvel[r] = dvel[r]*sc;
        MJH_FOR_LANES(c, nv) {
          const int in0 = (M.body_dofanc[b0*s.nvw + (c >> 5)] >> (c & 31)) & 1;
          const int in1 = (M.body_dofanc[b1*s.nvw + (c >> 5)] >> (c & 31)) & 1;
          real a = 0, b = 0;
          if (in0 || in1) {
            crptr cd = cdof + 6*c;
            real cdd[6];
            for (int q = 0; q < 6; q++) cdd[q] = cdof_dot[6*c + q];
            const int jt = M.dof_jnttype[c];
            const int dadr = M.jnt_dofadr[M.dof_jntid[c]];
            if (jt == MJH_JNT_BALL || (jt == MJH_JNT_FREE && c >= dadr + 3)) sp_cross_motion(cdd, cvel + 6*M.dof_bodyid[c], cd);
            real jd0[3] = {0, 0, 0}, jd1[3] = {0, 0, 0}, j0[3] = {0, 0, 0}, j1[3] = {0, 0, 0}, t1[3], t2[3];
            if (in0) {
              v3_cross(t1, cdd, off0); v3_cross(t2, cd, v0);
              for (int r = 0; r < 3; r++) jd0[r] += cdd[3 + r] + t1[r] + t2[r];
              v3_cross(t1, cd, off0);
              for (int r = 0; r < 3; r++) j0[r] = cd[3 + r] + t1[r];
            }
            if (in1) {
              v3_cross(t1, cdd, off1); v3_cross(t2, cd, v1);
              for (int r = 0; r < 3; r++) jd1[r] += cdd[3 + r] + t1[r] + t2[r];
              v3_cross(t1, cd, off1);
              for (int r = 0; r < 3; r++) j1[r] = cd[3 + r] + t1[r];
            }
            for (int r = 0; r < 3; r++) if (dpnt[r] != 0) a += (jd1[r] - jd0[r])*dpnt[r];
            for (int r = 0; r < 3; r++) if (dvel[r] != 0) b += (j1[r] - j0[r])*dvel[r];
          }
          col[c] = a; col[nv + c] = b;
        }
        wv_sync();
        if (s.sparse) {
          const M128 both = m128_or(m128_ldw(M.body_dofanc + (size_t)b0*s.nvw, s.nvw), m128_ldw(M.body_dofanc + (size_t)b1*s.nvw, s.nvw));
          for (int term = 0; term < 2; term++)
            for (M128 um = both; m128_any(um); um = m128_drop_lowest(um)) {
              const int c = m128_lowest(um);
              res += (col[term*nv + c] / divisor) * qvel[c];
            }
        } else {
          res += dot_ref(col, qvel, nv) / divisor;
          res += dot_ref(col + nv, qvel, nv) / divisor;
        }
        wv_sync();
      }
      j++;
    }
    const real coef = arm * res;
    if (coef != 0) {
      const int radr = M.ten_J_rowadr[t], rnnz = M.ten_J_rownnz[t];
      MJH_FOR_LANES(k, rnnz) bias[M.ten_J_colind[radr + k]] += coef * tJ[radr + k];
    }
    wv_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// mj_rne(flg_acc=0) -> qfrc_bias                  (engine_core_smooth.c:2328-2389)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_rne(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr qvel = MJH_F(B, qvel, e);
  crptr cdof = MJH_F(B, cdof, e);
  crptr cdof_dot = MJH_F(B, cdof_dot, e);
  crptr cvel = MJH_F(B, cvel, e);
  crptr cinert = MJH_F(B, cinert, e);
  rptr cacc = MJH_F(B, cacc, e);
  rptr cfrc = MJH_F(B, cfrc, e);
  rptr bias = MJH_F(B, qfrc_bias, e);

  if (wv_lane() == 0) {
    for (int k = 0; k < 6; k++) cacc[k] = 0;
    if (!(M.o.disableflags & (1<<7))) {
      cacc[3] = M.o.gravity[0]*-1; cacc[4] = M.o.gravity[1]*-1; cacc[5] = M.o.gravity[2]*-1;
    }
  }
  wv_sync();
  // Only the sum cacc[i] = cacc[parent] + cdof_dot_i * qvel_i follows the tree; the joint term and the body
  // force are per-body work.  So: (1) every body's joint term at once (parked in its cacc slot), (2) the level
  // loop adds the parent's acceleration, a lane per (body of the level, component), (3) every body's
  // force at once.  Same operands, same operations as the body-by-body recursion.
  MJH_FOR_LANES(k, s.nbody - 1) {
    const int i = k + 1;
    const int bda = M.body_dofadr[i];
    real tmp[6];
    mul_dof_vec(tmp, cdof_dot + 6*bda, qvel + bda, M.body_dofnum[i]);
    for (int q = 0; q < 6; q++) cacc[6*i + q] = tmp[q];
  }
  wv_sync();
  for (int L = 1; L < s.nlevel; L++) {
    int a0 = M.body_level_adr[L], a1 = M.body_level_adr[L+1];
    MJH_FOR_LANES(w, (a1 - a0)*6) {
      const int k = w / 6, q = w - 6*k;
      const int i = M.body_level_ids[a0 + k];
      cacc[6*i + q] = cacc[6*M.body_parentid[i] + q] + cacc[6*i + q];
    }
    wv_sync();
  }
  MJH_FOR_LANES(k, s.nbody) {
    const int i = k;
    if (i == 0) { for (int q = 0; q < 6; q++) cfrc[q] = 0; continue; }
    real f[6], tmp[6], tmp1[6];
    sp_mul_inert(f, cinert + 10*i, cacc + 6*i);
    sp_mul_inert(tmp, cinert + 10*i, cvel + 6*i);
    sp_cross_force(tmp1, cvel + 6*i, tmp);
    for (int q = 0; q < 6; q++) cfrc[6*i + q] = f[q] + tmp1[q];
  }
  wv_sync();
  tree_accumulate_to_parent(M, cfrc, 6, 0);
  MJH_FOR_LANES(i, s.nv) bias[i] = sp_dot6(cdof + 6*i, cfrc + 6*M.dof_bodyid[i]);
  wv_sync();
  if (MJH_HAS(MJH_FT_TENDONSPATIAL) && MJH_HAS(MJH_FT_PASSIVEMISC) && M.o.has_ten_armature) tendon_bias(M, B, e, bias);
}

// ------------------------------------------------------------------------------------------------
// tendon / actuator velocities                    (mj_fwdVelocity head, engine_forward.c:197-208)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_ten_act_velocity(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr qvel = MJH_F(B, qvel, e);
  if (s.ntendon) {
    crptr tJ = MJH_F(B, ten_J, e);
    rptr tv = MJH_F(B, ten_velocity, e);
    MJH_FOR_LANES(i, s.ntendon) {
      int adr = M.ten_J_rowadr[i];
      tv[i] = dot_sparse_ref(tJ + adr, qvel, M.ten_J_rownnz[i], M.ten_J_colind + adr);
    }
  }
  if (s.nu) {
    crptr mom = MJH_F(B, actuator_moment, e);
    ciptr rownnz = MJH_F(B, moment_rownnz, e);
    ciptr colind = MJH_F(B, moment_colind, e);
    rptr av = MJH_F(B, actuator_velocity, e);
    const int dsbl_act = M.o.disableflags & (1<<11);
    MJH_FOR_LANES(i, s.nu) {
      int adr = M.actuator_momentadr[i];
      av[i] = dsbl_act ? 0 : dot_sparse_ref(mom + adr, qvel, rownnz[i], colind + adr);
    }
  }
  wv_sync();
  if (MJH_HAS(MJH_FT_FLEX) && s.nflexedge) flex_edge_velocity(M, B, e);
}

// ------------------------------------------------------------------------------------------------
// mj_fwdActuation: stateless actuators (dyntype none), fixed/affine gain, none/affine bias
//                                                 (engine_forward.c:353-1003)
// ------------------------------------------------------------------------------------------------
// mj_nextActivation                               (engine_support.c:706-775)
// integrator / filter: Euler; filterexact: closed form; then the actrange clamp
// muscle model (mju_muscleGainLength / mju_muscleGain / mju_muscleBias / mju_muscleDynamics, engine_util_misc.c:1049-1195;
// mjd_muscleGain_vel, engine_derivative.c:969-1014): prm = (range[2], force, scale, lmin, lmax, vmax, fpmax, fvmax)
MJH_DEV real muscle_gain_length(real length, real lmin, real lmax) {
  if (lmin <= length && length <= lmax) {
    const real a = 0.5*(lmin + 1), b = 0.5*(1 + lmax);
    if (length <= a) { const real x = (length - lmin) / r_max(MJH_MINVAL, a - lmin); return 0.5*x*x; }
    else if (length <= 1) { const real x = (1 - length) / r_max(MJH_MINVAL, 1 - a); return 1 - 0.5*x*x; }
    else if (length <= b) { const real x = (length - 1) / r_max(MJH_MINVAL, b - 1); return 1 - 0.5*x*x; }
    else { const real x = (lmax - length) / r_max(MJH_MINVAL, lmax - b); return 0.5*x*x; }
  }
  return 0.0;
}
// deriv = 0: the active force -force FL FV; 1: its derivative with respect to the actuator velocity
template <class PP, class PR>
MJH_DEV real muscle_gain(real len, real vel, PR lengthrange, real acc0, PP prm, int deriv) {
  const real range0 = prm[0], range1 = prm[1];
  real force = prm[2];
  const real scale = prm[3], lmin = prm[4], lmax = prm[5], vmax = prm[6], fvmax = prm[8];
  if (force < 0) force = scale / r_max(MJH_MINVAL, acc0);
  const real L0 = (lengthrange[1] - lengthrange[0]) / r_max(MJH_MINVAL, range1 - range0);
  const real L = range0 + (len - lengthrange[0]) / r_max(MJH_MINVAL, L0);
  const real V = vel / r_max(MJH_MINVAL, L0*vmax);
  const real FL = muscle_gain_length(L, lmin, lmax);
  const real y = fvmax - 1;
  if (deriv) {
    real dFV;
    if (V <= -1) dFV = 0;
    else if (V <= 0) dFV = 2*V + 2;
    else if (V <= y) dFV = (-2*V + 2*y) / r_max(MJH_MINVAL, y);
    else dFV = 0;
    return -force*FL*dFV/r_max(MJH_MINVAL, L0*vmax);
  }
  real FV;
  if (V <= -1) FV = 0;
  else if (V <= 0) FV = (V + 1)*(V + 1);
  else if (V <= y) FV = fvmax - (y - V)*(y - V) / r_max(MJH_MINVAL, y);
  else FV = fvmax;
  return -force*FL*FV;
}
template <class PP, class PR>
MJH_DEV real muscle_bias(real len, PR lengthrange, real acc0, PP prm) {
  const real range0 = prm[0], range1 = prm[1];
  real force = prm[2];
  const real scale = prm[3], lmax = prm[5], fpmax = prm[7];
  if (force < 0) force = scale / r_max(MJH_MINVAL, acc0);
  const real L0 = (lengthrange[1] - lengthrange[0]) / r_max(MJH_MINVAL, range1 - range0);
  const real L = range0 + (len - lengthrange[0]) / r_max(MJH_MINVAL, L0);
  const real b = 0.5*(1 + lmax);
  if (L <= 1) return 0;
  else if (L <= b) { const real x = (L - 1) / r_max(MJH_MINVAL, b - 1); return -force*fpmax*0.5*x*x; }
  else { const real x = (L - b) / r_max(MJH_MINVAL, b - 1); return -force*fpmax*(0.5 + x); }
}
template <class PP>
MJH_DEV real muscle_dynamics(real ctrl, real act, PP prm) {
  const real ctrlclamp = r_clip(ctrl, 0, 1), actclamp = r_clip(act, 0, 1);
  const real tau_act = prm[0] * (0.5 + 1.5*actclamp);
  const real tau_deact = prm[1] / (0.5 + 1.5*actclamp);
  const real width = prm[2];
  const real dctrl = ctrlclamp - act;
  real tau;
  if (width < MJH_MINVAL) tau = dctrl > 0 ? tau_act : tau_deact;
  else {
    // mju_sigmoid: 0 below 0, 1 above 1, 6x^5 - 15x^4 + 10x^3 between
    const real x = dctrl/width + 0.5;
    const real sg = x <= 0 ? (real)0 : (x >= 1 ? (real)1 : x*x*x * (3*x * (2*x - 5) + 10));
    tau = tau_deact + (tau_act - tau_deact)*sg;
  }
  return dctrl / r_max(MJH_MINVAL, tau);
}

MJH_DEV real next_activation(MREF M, int a, real act, real act_dot) {
  if (M.actuator_dyntype[a] == MJH_DYN_FILTEREXACT) {
    const real tau = r_max(MJH_MINVAL, M.actuator_dyntau[a]);
    act = act + act_dot * tau * (1 - r_exp(-M.o.timestep / tau));
  } else {
    act = act + act_dot * M.o.timestep;
  }
  if (M.actuator_actlimited[a]) act = r_clip(act, M.actuator_actrange[2*a], M.actuator_actrange[2*a+1]);
  return act;
}

MJH_DEVN void stage_actuation(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  rptr force = MJH_F(B, actuator_force, e);
  rptr qfa = MJH_F(B, qfrc_actuator, e);
  const int dsbl = M.o.disableflags;
  if (s.nu == 0 || (dsbl & (1<<11))) {
    MJH_FOR_LANES(i, s.nu) force[i] = 0;
    MJH_FOR_LANES(i, s.nv) qfa[i] = 0;
    wv_sync();
    return;
  }
  crptr ctrl_in = MJH_F(B, ctrl, e);
  crptr len = MJH_F(B, actuator_length, e);
  crptr vel = MJH_F(B, actuator_velocity, e);
  iptr warn = MJH_F(B, warning, e);

  // any bad control (after clamping) zeroes ALL controls
  int bad = 0;
  MJH_FOR_LANES(i, s.nu) {
    real c = ctrl_in[i];
    if (!(dsbl & (1<<8)) && M.actuator_ctrllimited[i])
      c = r_clip(c, M.actuator_ctrlrange[2*i], M.actuator_ctrlrange[2*i+1]);
    if (r_isbad(c)) bad = 1;
  }
  bad = wv_any(bad);
  if (bad && wv_lane() == 0) warn[MJH_WARN_BADCTRL]++;

  rptr act = MJH_F(B, act, e);
  rptr act_dot = MJH_F(B, act_dot, e);
  MJH_FOR_LANES(i, s.nu) {
    real c = ctrl_in[i];
    if (!(dsbl & (1<<8)) && M.actuator_ctrllimited[i])
      c = r_clip(c, M.actuator_ctrlrange[2*i], M.actuator_ctrlrange[2*i+1]);
    if (bad) c = 0;
    // stateful actuators: act_dot from the control, the force sees the activation (:403-447, :800-817)
    const int dyn = (MJH_HAS(MJH_FT_ACT) && s.na) ? (int)M.actuator_dyntype[i] : (int)MJH_DYN_NONE;
    if (dyn != MJH_DYN_NONE) {
      const int aa = M.actuator_actadr[i];
      real ad;
      if (dyn == MJH_DYN_INTEGRATOR) ad = c;
      else if (dyn == MJH_DYN_MUSCLE) ad = muscle_dynamics(c, act[aa], M.actuator_dynprm + 3*i);
      else ad = (c - act[aa]) / r_max(MJH_MINVAL, M.actuator_dyntau[i]);
      act_dot[aa] = ad;
      c = M.actuator_actearly[i] ? next_activation(M, i, act[aa], ad) : (real)act[aa];
    }
    auto gp = M.actuator_gainprm + 10*i;
    auto bp = M.actuator_biasprm + 10*i;
    real gain;
    if (!MJH_HAS(MJH_FT_GAINBIAS) || M.actuator_gaintype[i] == MJH_GAIN_FIXED) gain = gp[0];
    else if (M.actuator_gaintype[i] == MJH_GAIN_MUSCLE) gain = muscle_gain(len[i], vel[i], M.actuator_lengthrange + 2*i, M.actuator_acc0[i], gp, 0);
    else gain = gp[0] + gp[1]*len[i] + gp[2]*vel[i];
    real f = gain * c;
    real bias = 0.0;
    if (MJH_HAS(MJH_FT_GAINBIAS) && M.actuator_biastype[i] == MJH_BIAS_AFFINE) bias = bp[0] + bp[1]*len[i] + bp[2]*vel[i];
    else if (MJH_HAS(MJH_FT_GAINBIAS) && M.actuator_biastype[i] == MJH_BIAS_MUSCLE) bias = muscle_bias(len[i], M.actuator_lengthrange + 2*i, M.actuator_acc0[i], bp);
    f += bias;
    // (a disabled actuator keeps its act_dot -- mj_advance ignores it -- and produces no force: engine_forward.c:617)
    if (MJH_HAS(MJH_FT_GAINBIAS) && M.o.has_act_disabled && M.actuator_disabled[i]) f = 0;     // (its force stays 0; the clamp below still sees it)
    if (MJH_HAS(MJH_FT_GAINBIAS) && M.o.has_ten_actfrc) { force[i] = f; continue; }      // (clamped after the tendon totals, below)
    if (MJH_HAS(MJH_FT_GAINBIAS) && M.actuator_forcelimited[i])
      f = r_clip(f, M.actuator_forcerange[2*i], M.actuator_forcerange[2*i+1]);
    force[i] = f;
  }
  wv_sync();
  if (MJH_HAS(MJH_FT_GAINBIAS) && M.o.has_ten_actfrc) {
    // total actuator force per limited tendon (in actuator order), then every actuator on a tendon whose total leaves the
    // range is scaled by range / total (engine_forward.c:880-915); the per-actuator force clamp comes after that
    MJH_FOR_LANES(i, s.nu) {
      real f = force[i];
      if (M.actuator_trntype[i] == MJH_TRN_TENDON) {
        const int t = M.actuator_trnid[2*i];
        if (M.tendon_actfrclimited[t]) {
          real total = 0;
          for (int k = 0; k < s.nu; k++)
            if (M.actuator_trntype[k] == MJH_TRN_TENDON && M.actuator_trnid[2*k] == t) total += force[k];
          if (total) {
            if (total < M.tendon_actfrcrange[2*t]) f *= M.tendon_actfrcrange[2*t] / total;
            else if (total > M.tendon_actfrcrange[2*t + 1]) f *= M.tendon_actfrcrange[2*t + 1] / total;
          }
        }
      }
      if (M.actuator_forcelimited[i])
        f = r_clip(f, M.actuator_forcerange[2*i], M.actuator_forcerange[2*i+1]);
      MJH_G(B, scratch, e)[i] = f;
    }
    wv_sync();
    MJH_FOR_LANES(i, s.nu) force[i] = MJH_G(B, scratch, e)[i];
    wv_sync();
  }

  // qfrc_actuator = moment' * force, rows added in actuator order (mju_mulMatTVecSparse)
  crptr mom = MJH_F(B, actuator_moment, e);
  ciptr rownnz = MJH_F(B, moment_rownnz, e);
  ciptr colind = MJH_F(B, moment_colind, e);
  if (!MJH_HAS(MJH_FT_TRNMISC)) {
    // scalar joint transmissions only: the rows that touch dof j are known at upload (same order, same skips)
    MJH_FOR_LANES(j, s.nv) {
      real r = 0;
      for (int a = M.dof_act_adr[j]; a < M.dof_act_adr[j + 1]; a++) {
        const int i = M.dof_act_ids[a];
        const real scl = force[i];
        if (scl == 0) continue;
        r += mom[M.actuator_momentadr[i]]*scl;
      }
      qfa[j] = r;
    }
  } else
  MJH_FOR_LANES(j, s.nv) {
    real r = 0;
    for (int i = 0; i < s.nu; i++) {
      real scl = force[i];
      if (scl == 0) continue;
      int adr = M.actuator_momentadr[i];
      for (int k = 0; k < rownnz[i]; k++) if (colind[adr + k] == j) r += mom[adr + k]*scl;
    }
    qfa[j] = r;
  }
  wv_sync();
  // actuator-level gravity compensation (engine_forward.c:981-996; the condition of the passive stage's gravcomp block)
  if (MJH_HAS(MJH_FT_PASSIVEMISC) && s.nv_actgc && M.o.has_gravcomp && !(dsbl & (1<<7)) &&
      sqrt(M.o.gravity[0]*M.o.gravity[0] + M.o.gravity[1]*M.o.gravity[1] + M.o.gravity[2]*M.o.gravity[2]) != 0) {
    crptr gc = MJH_G(B, qfrc_gravcomp, e);
    MJH_FOR_LANES(j, s.nv) if (M.dof_actgravcomp[j]) qfa[j] += gc[j];
    wv_sync();
  }
  // joint-level actuator force limits (clampVec with jnt_dofadr index)
  MJH_FOR_LANES(j, s.njnt) {
    if (MJH_HAS(MJH_FT_GAINBIAS) && M.jnt_actfrclimited[j]) {
      int d = M.jnt_dofadr[j];
      qfa[d] = r_clip(qfa[d], M.jnt_actfrcrange[2*j], M.jnt_actfrcrange[2*j+1]);
    }
  }
  wv_sync();
}

// ------------------------------------------------------------------------------------------------
// mj_fwdAcceleration                              (engine_forward.c:1007-1052)
// (xfrc_applied is handled by the host API: non-zero Cartesian forces are rejected for now)
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_acceleration(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  crptr fp = MJH_F(B, qfrc_passive, e);
  crptr fb = MJH_F(B, qfrc_bias, e);
  crptr fa = MJH_F(B, qfrc_applied, e);
  crptr fu = MJH_F(B, qfrc_actuator, e);
  rptr fsm = MJH_F(B, qfrc_smooth, e);
  rptr qas = MJH_F(B, qacc_smooth, e);
  // Cartesian forces on bodies, projected body by body (mj_xfrcAccumulate -> mj_applyFT at the body's
  // COM, engine_support.c:438-512): force then torque of each body with a non-zero wrench
  const int xfrc_on = B.xfrc_on;
  crptr xf = MJH_G(B, xfrc_applied, e);
  crptr xipos = MJH_F(B, xipos, e);        // (kept readable by the plan when xfrc_on)
  crptr cdof = MJH_F(B, cdof, e);
  crptr com = MJH_F(B, subtree_com, e);
  MJH_FOR_LANES(i, s.nv) {
    real f = fp[i] - fb[i];
    f += fa[i];
    f += fu[i];
    if (xfrc_on) {
      crptr cd = cdof + 6*i;
      for (int b = 1; b < s.nbody; b++) {
        crptr w = xf + 6*b;
        if (w[0] == 0 && w[1] == 0 && w[2] == 0 && w[3] == 0 && w[4] == 0 && w[5] == 0) continue;
        real tf = 0, tt = 0;
        const int wb = M.body_weldid[b];
        if ((M.body_dofanc[wb*s.nvw + (i >> 5)] >> (i & 31)) & 1) {
          real off[3], cr[3];
          v3_sub(off, xipos + 3*b, com + 3*M.body_rootid[b]);
          v3_cross(cr, cd, off);
          for (int r = 0; r < 3; r++) { const real fr = w[r]; if (fr != 0) tf += (cd[3 + r] + cr[r])*fr; }
          for (int r = 0; r < 3; r++) { const real tr = w[3 + r]; if (tr != 0) tt += cd[r]*tr; }
        }
        f += tf;
        f += tt;
      }
    }
    fsm[i] = f;
    qas[i] = f;
  }
  wv_sync();
  solve_ld(M, qas, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e));
}

// ------------------------------------------------------------------------------------------------
// integration (kept with the smooth stages so that a multi-wavefront workgroup can run it on all of its wavefronts)
// ------------------------------------------------------------------------------------------------
// mj_integratePos: qpos <- qpos (+) qvel*h, joint by joint            (engine_support.c:639-690)
template <class P0, class P1>
MJH_DEV void integrate_pos(MREF M, P0 qpos, P1 qvel, real h) {
  MJH_FOR_LANES(j, M.s.njnt) {
    int padr = M.jnt_qposadr[j], vadr = M.jnt_dofadr[j];
    int jt = M.jnt_type[j];
    if (jt == MJH_JNT_FREE) {
      for (int i = 0; i < 3; i++) qpos[padr + i] += h * qvel[vadr + i];
      padr += 3; vadr += 3;
    }
    if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
      q_integrate(qpos + padr, qvel + vadr, h);
    } else {
      qpos[padr] += h * qvel[vadr];
    }
  }
}

// mj_advance, activation part                      (engine_forward.c:1314-1323)
template <class P0>
MJH_DEV void advance_act(MREF M, BREF B, int e, P0 act_dot) {
  const MJH_CONST_AS DSizes& s = M.s;
  if (!MJH_HAS(MJH_FT_ACT) || !s.na || (M.o.disableflags & (1<<11))) return;
  rptr act = MJH_F(B, act, e);
  MJH_FOR_LANES(i, s.nu) {
    if (M.actuator_dyntype[i] == MJH_DYN_NONE) continue;
    const int aa = M.actuator_actadr[i];
    // (a disabled actuator's activation advances with act_dot = 0: mj_advance, engine_forward.c:1321)
    const real ad = (MJH_HAS(MJH_FT_GAINBIAS) && M.o.has_act_disabled && M.actuator_disabled[i]) ? (real)0 : (real)act_dot[aa];
    act[aa] = next_activation(M, i, act[aa], ad);
  }
}

// mj_EulerSkip + mj_advance                        (engine_forward.c:1398-1476, :1261-1395)
// The implicit-damping matrix qH = M + h*diag(B): normally its factor was produced next to M's
// (stage_factor_m, two matrices per pass) and its solve shared stage_finish's pass -- then the damped
// acceleration already waits in qe (counts[MJH_C_PAIRED]).  Otherwise the parked factor is picked up
// here, or (models outside the paired routines' range) qH is rebuilt from the copy of M that stage_crb
// left in M's global home and factorised in the slots of qLD/qLDiagInv, dead once qacc is known.
MJH_DEVN void euler_advance(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  const real h = M.o.timestep;
  rptr qvel = MJH_F(B, qvel, e);
  rptr qpos = MJH_F(B, qpos, e);
  crptr qacc = MJH_F(B, qacc, e);
  rptr qe = MJH_F(B, qe, e);               // integrated acceleration [nv]

  iptr counts = MJH_F(B, counts, e);
  const int paired = counts[MJH_C_PAIRED];      // stage_finish already solved the damped system into qe
  wv_sync();
  if (wv_lane() == 0) counts[MJH_C_PAIRED] = 0;
  if (paired) {
    // nothing to do
  } else if (M.o.euler_damp) {
    crptr Mq = MJH_G(B, M, e);
    rptr qH = MJH_F(B, qLD, e);
    rptr qHDiagInv = MJH_F(B, qLDiagInv, e);
    if (pairs_euler_factor(M, B, e)) {
      // stage_factor_m factorised qH next to M and parked it (primal solvers, steps without constraints:
      // stage_finish had no solve to share, so the factor is picked up here)
      crptr qHg = MJH_G(B, qH2, e);
      crptr qHDg = MJH_G(B, qH2DiagInv, e);
      MJH_FOR_LANES(k, s.nC) qH[k] = qHg[k];
      MJH_FOR_LANES(i, nv) qHDiagInv[i] = qHDg[i];
      wv_sync();
    } else {
      MJH_FOR_LANES(k, s.nC) qH[k] = Mq[k];
      wv_sync();
      MJH_FOR_LANES(i, nv) {
        real dd = poly_force_deriv(M.dof_damping_eff[i], M.dof_dampingpoly_eff + 2*i, qvel[i], 1);
        qH[M.M_rowadr[i] + M.M_rownnz[i] - 1] += h * dd;
      }
      wv_sync();
      factor_ld(M, qH, qHDiagInv);
    }
    crptr fs = MJH_F(B, qfrc_smooth, e);
    crptr fc = MJH_F(B, qfrc_constraint, e);
    MJH_FOR_LANES(i, nv) qe[i] = fs[i] + fc[i];
    wv_sync();
    solve_ld(M, qe, qH, qHDiagInv);
  } else {
    MJH_FOR_LANES(i, nv) qe[i] = qacc[i];
    wv_sync();
  }

  // mj_advance: activations ; qvel += h*qacc ; qpos integrates the NEW qvel ; time ; warmstart
  advance_act(M, B, e, MJH_F(B, act_dot, e));
  MJH_FOR_LANES(i, nv) qvel[i] += qe[i]*h;
  wv_sync();
  integrate_pos(M, qpos, qvel, h);
  rptr ws = MJH_F(B, qacc_warmstart, e);
  MJH_FOR_LANES(i, nv) ws[i] = qacc[i];
  if (wv_lane() == 0) MJH_F(B, time, e)[0] += h;
  wv_sync();
}

