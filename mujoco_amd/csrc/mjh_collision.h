// Collision stage of the batched step: candidate geom pairs -> contacts, one wavefront per env.
//
// The reference runs sweep-and-prune over bodies, then a BVH midphase, then the narrowphase
// (engine_collision_driver.c:595-886).  Broad- and midphase only CULL pairs conservatively, so the
// contact list equals "narrowphase over every geom pair that survives the model-constant filters"
// in the reference's order: ascending body-pair signature, then (g1,g2) within a body pair, then
// the collider's own emission order.  The host precomputes that ordered static
// pair list with its mixed contact parameters (mjh_host.cpp: build_pairs); here each lane takes
// pairs, applies the reference's bounding-sphere filter and the analytic collider, and the wave
// compacts the survivors in order with a prefix sum.
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)


struct PreContact {        // mjPreContact, include/mujoco/mjdata.h:29
  real dist, pos[3], normal[3], tangent[3];
};

// mjraw_PlaneSphere, engine_collision_primitive.c:28
template <class P0, class P1, class P2>
MJH_DEV int col_plane_sphere(PreContact* c, real margin, P0 pos1, P1 mat1,
                             P2 pos2, real radius) {
  c->normal[0] = mat1[2]; c->normal[1] = mat1[5]; c->normal[2] = mat1[8];
  real tmp[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
  real cdist = v3_dot(tmp, c->normal);
  if (cdist > margin + radius) return 0;
  c->dist = cdist - radius;
  v3_scl(tmp, c->normal, -c->dist / 2 - radius);
  v3_add(c->pos, pos2, tmp);
  v3_zero(c->tangent);
  return 1;
}

// mjc_PlaneCapsule, engine_collision_primitive.c:66
template <class P0, class P1, class P2, class P3, class P4>
MJH_DEV int col_plane_capsule(PreContact* c, real margin, P0 pos1, P1 mat1,
                              P2 pos2, P3 mat2, P4 size2) {
  real axis[3] = {mat2[2], mat2[5], mat2[8]};
  real seg[3] = {size2[1]*axis[0], size2[1]*axis[1], size2[1]*axis[2]};
  real end[3];
  v3_add(end, pos2, seg);
  int n1 = col_plane_sphere(c, margin, pos1, mat1, end, size2[0]);
  v3_sub(end, pos2, seg);
  int n2 = col_plane_sphere(c + n1, margin, pos1, mat1, end, size2[0]);
  if (n1) v3_copy(c[0].tangent, axis);
  if (n2) v3_copy(c[n1].tangent, axis);
  return n1 + n2;
}

// mjraw_SphereSphere, engine_collision_primitive.c:262
template <class P0, class P1, class P2, class P3>
MJH_DEV int col_sphere_sphere(PreContact* c, real margin, P0 pos1, P1 mat1, real r1,
                              P2 pos2, P3 mat2, real r2) {
  real dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  real cdist_sqr = v3_dot(dif, dif);
  real min_dist = margin + r1 + r2;
  if (cdist_sqr > min_dist*min_dist) return 0;
  c->dist = sqrt(cdist_sqr) - r1 - r2;
  v3_sub(c->normal, pos2, pos1);
  real len = v3_normalize(c->normal);
  if (len < MJH_MINVAL) {
    real a1[3] = {mat1[2], mat1[5], mat1[8]};
    real a2[3] = {mat2[2], mat2[5], mat2[8]};
    v3_cross(c->normal, a1, a2);
    v3_normalize(c->normal);
  }
  v3_scl(c->pos, c->normal, r1 + c->dist / 2);
  v3_addto(c->pos, pos1);
  v3_zero(c->tangent);
  return 1;
}

// mjraw_SphereCapsule, engine_collision_primitive.c:313
template <class P0, class P1, class P2, class P3, class P4>
MJH_DEV int col_sphere_capsule(PreContact* c, real margin, P0 pos1, P1 mat1, real r1,
                               P2 pos2, P3 mat2, P4 size2) {
  real len = size2[1];
  real axis[3] = {mat2[2], mat2[5], mat2[8]};
  real vec[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  real x = r_clip(v3_dot(axis, vec), -len, len);
  v3_scl(vec, axis, x);
  v3_addto(vec, pos2);
  return col_sphere_sphere(c, margin, pos1, mat1, r1, vec, mat2, size2[0]);
}

// mjraw_CapsuleCapsule, engine_collision_primitive.c:425
template <class P0, class P1, class P2, class P3, class P4, class P5>
MJH_DEV int col_capsule_capsule(PreContact* c, real margin, P0 pos1, P1 mat1,
                                P2 size1, P3 pos2, P4 mat2,
                                P5 size2) {
  real axis1[3] = {mat1[2]*size1[1], mat1[5]*size1[1], mat1[8]*size1[1]};
  real axis2[3] = {mat2[2]*size2[1], mat2[5]*size2[1], mat2[8]*size2[1]};
  real dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  real ma =  v3_dot(axis1, axis1);
  real mb = -v3_dot(axis1, axis2);
  real mc =  v3_dot(axis2, axis2);
  real u  = -v3_dot(axis1, dif);
  real v  =  v3_dot(axis2, dif);
  real det = ma*mc - mb*mb;
  real r1 = size1[0], r2 = size2[0];
  real vec1[3], vec2[3];

  if (fabs(det) >= MJH_MINVAL) {
    real x1 = (mc*u - mb*v) / det;
    real x2 = (ma*v - mb*u) / det;
    if (x1 > 1) {
      x1 = 1;
      x2 = (v - mb) / mc;
    } else if (x1 < -1) {
      x1 = -1;
      x2 = (v + mb) / mc;
    }
    if (x2 > 1) {
      x2 = 1;
      x1 = r_clip((u - mb) / ma, -1, 1);
    } else if (x2 < -1) {
      x2 = -1;
      x1 = r_clip((u + mb) / ma, -1, 1);
    }
    v3_scl(vec1, axis1, x1);
    v3_addto(vec1, pos1);
    v3_scl(vec2, axis2, x2);
    v3_addto(vec2, pos2);
    return col_sphere_sphere(c, margin, vec1, mat1, r1, vec2, mat2, r2);
  }

  // parallel axes: up to two contacts from the four end-point tests
  v3_add(vec1, pos1, axis1);
  real x2 = r_clip((v - mb) / mc, -1, 1);
  v3_scl(vec2, axis2, x2);
  v3_addto(vec2, pos2);
  int n1 = col_sphere_sphere(c, margin, vec1, mat1, r1, vec2, mat2, r2);

  v3_sub(vec1, pos1, axis1);
  x2 = r_clip((v + mb) / mc, -1, 1);
  v3_scl(vec2, axis2, x2);
  v3_addto(vec2, pos2);
  int n2 = col_sphere_sphere(c + n1, margin, vec1, mat1, r1, vec2, mat2, r2);
  if (n1 + n2 >= 2) return n1 + n2;

  v3_add(vec2, pos2, axis2);
  real x1 = r_clip((u - mb) / ma, -1, 1);
  v3_scl(vec1, axis1, x1);
  v3_addto(vec1, pos1);
  int n3 = col_sphere_sphere(c + n1 + n2, margin, vec1, mat1, r1, vec2, mat2, r2);
  if (n1 + n2 + n3 >= 2) return n1 + n2 + n3;

  v3_sub(vec2, pos2, axis2);
  x1 = r_clip((u + mb) / ma, -1, 1);
  v3_scl(vec1, axis1, x1);
  v3_addto(vec1, pos1);
  int n4 = col_sphere_sphere(c + n1 + n2 + n3, margin, vec1, mat1, r1, vec2, mat2, r2);
  return n1 + n2 + n3 + n4;
}

// ---- box : box (mjc_BoxBox, engine_collision_box.c:697-1066) -------------------------------------
// separating-axis search over 6 face and 9 edge-cross axes (with the reference's rounding slack,
// edge bias and face substitution), then either one edge-edge contact at the midpoint of the
// closest segment points, or the incident face clipped against the reference face (<= 8 contacts)
#define MJH_BB_SEPEPS 1e-13
#define MJH_BB_PAREPS 1e-16
#define MJH_BB_SGNEPS 1e-9
#define MJH_BB_DUPEPS 1e-14
#define MJH_BB_EDGEBIAS 1e-6
#define MJH_BB_MAXVERT 12

// clip the polygon in buffer `cur` of P against sign*v[coord] <= limit; returns the new count and
// flips *cur when something was clipped (clipHalfPlane, :651-692)
MJH_DEV int bb_clip(real (*P)[MJH_BB_MAXVERT][3], int* cur, int nin, int coord, real sign, real limit) {
  real (*in)[3] = P[*cur];
  real d[MJH_BB_MAXVERT];
  int all_inside = 1;
  for (int k = 0; k < nin; k++) {
    d[k] = sign*in[k][coord] - limit;
    all_inside &= d[k] <= 0;
  }
  if (all_inside) return nin;
  real (*out)[3] = P[1 - *cur];
  int nout = 0;
  for (int k = 0; k < nin; k++) {
    const int k1 = (k + 1 == nin) ? 0 : k + 1;
    const real dp = d[k], dq = d[k1];
    if (dp <= 0 && nout < MJH_BB_MAXVERT) {
      out[nout][0] = in[k][0]; out[nout][1] = in[k][1]; out[nout][2] = in[k][2];
      nout++;
    }
    if (((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) && nout < MJH_BB_MAXVERT) {
      const real t = dp / (dp - dq);
      out[nout][0] = in[k][0] + t*(in[k1][0] - in[k][0]);
      out[nout][1] = in[k][1] + t*(in[k1][1] - in[k][1]);
      out[nout][2] = in[k][2] + t*(in[k1][2] - in[k][2]);
      nout++;
    }
  }
  *cur = 1 - *cur;
  return nout;
}

// out of line: its polygon buffers stay out of stage_collision's frame; g = {pos1[3], mat1[9],
// size1[3], pos2[3], mat2[9], size2[3]} copied by the caller
MJH_DEVN int col_box_box_impl(PreContact* con, real margin, const real* g) {
  const real *pos1 = g, *mat1 = g + 3, *size1 = g + 12, *pos2 = g + 15, *mat2 = g + 18, *size2 = g + 27;
  real rot[9], rotabs[9], pos21[3], pos12[3], tmp[3];
  v3_sub(tmp, pos2, pos1);
  m3_multvec(pos21, mat1, tmp);
  v3_sub(tmp, pos1, pos2);
  m3_multvec(pos12, mat2, tmp);
  // rot = mat1' * mat2 (mju_mulMatTMat3 = mji_mulMatTMat3 order)
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
    rot[3*r + c] = mat1[r]*mat2[c] + mat1[3 + r]*mat2[3 + c] + mat1[6 + r]*mat2[6 + c];
  for (int k = 0; k < 9; k++) rotabs[k] = fabs(rot[k]);

  const real septol = margin + MJH_BB_SEPEPS*(size1[0] + size1[1] + size1[2] + size2[0] + size2[1] + size2[2]);
  real sep_best = -MJH_MAXVAL, sep_face = -MJH_MAXVAL;
  int code = -1;
  for (int i = 0; i < 3; i++) {
    const real radius2 = rotabs[3*i+0]*size2[0] + rotabs[3*i+1]*size2[1] + rotabs[3*i+2]*size2[2];
    const real sep = fabs(pos21[i]) - size1[i] - radius2;
    if (sep > septol) return 0;
    if (sep > sep_best) { sep_best = sep; code = i; }
  }
  for (int j = 0; j < 3; j++) {
    const real radius1 = rotabs[0+j]*size1[0] + rotabs[3+j]*size1[1] + rotabs[6+j]*size1[2];
    const real sep = fabs(pos12[j]) - size2[j] - radius1;
    if (sep > septol) return 0;
    if (sep > sep_best) { sep_best = sep; code = 3 + j; }
  }
  sep_face = sep_best;
  const int code_face = code;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
      real ax1 = -rot[3*i2+j];
      real ax2 = rot[3*i1+j];
      const real norm2 = ax1*ax1 + ax2*ax2;
      if (norm2 < MJH_BB_PAREPS) continue;
      const real inv = 1/sqrt(norm2);
      ax1 *= inv;
      ax2 *= inv;
      const real radius1 = size1[i1]*fabs(ax1) + size1[i2]*fabs(ax2);
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const real a2_1 = ax1*rot[3*i1+j1] + ax2*rot[3*i2+j1];
      const real a2_2 = ax1*rot[3*i1+j2] + ax2*rot[3*i2+j2];
      const real radius2 = size2[j1]*fabs(a2_1) + size2[j2]*fabs(a2_2);
      const real sep = fabs(ax1*pos21[i1] + ax2*pos21[i2]) - radius1 - radius2;
      if (sep > septol) return 0;
      if (sep - MJH_BB_EDGEBIAS*fabs(sep) > sep_best && sep > sep_face) { sep_best = sep; code = 6 + 3*i + j; }
    }
  }
  if (code < 0) return 0;

  // an edge axis nearly parallel to the best face axis gives way to the face (:806-826)
  if (code >= 6) {
    const int i = (code - 6) / 3, j = (code - 6) % 3;
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    real axis[3];
    axis[i] = 0; axis[i1] = -rot[3*i2+j]; axis[i2] = rot[3*i1+j];
    v3_normalize(axis);
    real face_dot;
    if (code_face < 3) face_dot = fabs(axis[code_face]);
    else { const int f = code_face - 3; face_dot = fabs(axis[0]*rot[0+f] + axis[1]*rot[3+f] + axis[2]*rot[6+f]); }
    if (face_dot > 0.99 && sep_best < sep_face + 0.05*fabs(sep_face) + MJH_MINVAL) { code = code_face; sep_best = sep_face; }
  }

  // ---- edge-edge contact (:830-945)
  if (code >= 6) {
    const int i = (code - 6) / 3, j = (code - 6) % 3;
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    real axis[3];
    axis[i] = 0; axis[i1] = -rot[3*i2+j]; axis[i2] = rot[3*i1+j];
    v3_normalize(axis);
    if (v3_dot(axis, pos21) < 0) { axis[0] = -axis[0]; axis[1] = -axis[1]; axis[2] = -axis[2]; }
    real a2[3] = {axis[0]*rot[0+0] + axis[1]*rot[3+0] + axis[2]*rot[6+0],
                  axis[0]*rot[0+1] + axis[1]*rot[3+1] + axis[2]*rot[6+1],
                  axis[0]*rot[0+2] + axis[1]*rot[3+2] + axis[2]*rot[6+2]};
    int amb1 = -1, amb2 = -1;
    if (fabs(axis[i1]) < MJH_BB_SGNEPS) amb1 = i1; else if (fabs(axis[i2]) < MJH_BB_SGNEPS) amb1 = i2;
    if (fabs(a2[j1]) < MJH_BB_SGNEPS) amb2 = j1; else if (fabs(a2[j2]) < MJH_BB_SGNEPS) amb2 = j2;
    real d2[3] = {rot[0+j], rot[3+j], rot[6+j]};
    const real b = d2[i];
    const real denom = 1 - b*b;
    real w1[3] = {0, 0, 0}, w2[3] = {0, 0, 0};
    real best_d2 = MJH_MAXVAL;
    for (int v1 = 0; v1 < (amb1 >= 0 ? 2 : 1); v1++) {
      for (int v2 = 0; v2 < (amb2 >= 0 ? 2 : 1); v2++) {
        real c1[3], cc[3], c2[3], ev[3];
        c1[i] = 0;
        c1[i1] = axis[i1] >= 0 ? size1[i1] : -size1[i1];
        c1[i2] = axis[i2] >= 0 ? size1[i2] : -size1[i2];
        if (amb1 >= 0 && v1) c1[amb1] = -c1[amb1];
        cc[j] = 0;
        cc[j1] = a2[j1] >= 0 ? -size2[j1] : size2[j1];
        cc[j2] = a2[j2] >= 0 ? -size2[j2] : size2[j2];
        if (amb2 >= 0 && v2) cc[amb2] = -cc[amb2];
        m3_mulvec(c2, rot, cc);
        v3_addto(c2, pos21);
        v3_sub(ev, c2, c1);
        const real d1e = ev[i];
        const real d2e = v3_dot(d2, ev);
        real sp = denom < MJH_MINVAL ? 0 : (d1e - b*d2e) / denom;
        sp = r_clip(sp, -size1[i], size1[i]);
        const real tp = r_clip(b*sp - d2e, -size2[j], size2[j]);
        sp = r_clip(d1e + b*tp, -size1[i], size1[i]);
        real p1[3] = {c1[0], c1[1], c1[2]}, p2[3] = {c2[0], c2[1], c2[2]}, gap[3];
        p1[i] += sp;
        v3_addtoscl(p2, d2, tp);
        v3_sub(gap, p2, p1);
        const real gap2 = v3_dot(gap, gap);
        if (gap2 < best_d2) { best_d2 = gap2; v3_copy(w1, p1); v3_copy(w2, p2); }
      }
    }
    real gap[3];
    v3_sub(gap, w2, w1);
    const real dist = v3_dot(gap, axis);
    if (dist > septol) return 0;
    real mid[3] = {0.5*(w1[0] + w2[0]), 0.5*(w1[1] + w2[1]), 0.5*(w1[2] + w2[2])};
    con[0].dist = dist;
    m3_mulvec(tmp, mat1, mid);
    v3_add(con[0].pos, tmp, pos1);
    m3_mulvec(con[0].normal, mat1, axis);
    v3_zero(con[0].tangent);
    return 1;
  }

  // ---- face contact: clip the incident face against the reference face (:949-1066)
  const int ref1 = code < 3;
  const int a = ref1 ? code : code - 3;
  const real* sizeref = ref1 ? size1 : size2;
  const real* sizeinc = ref1 ? size2 : size1;
  const real* posref = ref1 ? pos1 : pos2;
  const real* matref = ref1 ? mat1 : mat2;
  const real* posoi = ref1 ? pos21 : pos12;
  real rinc[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rinc[3*r + c] = ref1 ? rot[3*r + c] : rot[3*c + r];
  const real sgn = posoi[a] >= 0 ? 1 : -1;
  int binc = 0;
  for (int k = 1; k < 3; k++) if (fabs(rinc[3*a+k]) > fabs(rinc[3*a+binc])) binc = k;
  const real tinc = sgn*rinc[3*a+binc] > 0 ? -1 : 1;
  const int ax = (a + 1) % 3, ay = (a + 2) % 3, bu = (binc + 1) % 3, bv = (binc + 2) % 3;
  real poly[2][MJH_BB_MAXVERT][3];
  real cx[3], du[3], dv[3];
  for (int r = 0; r < 3; r++) {
    const int c = r == 0 ? ax : (r == 1 ? ay : a);
    cx[r] = posoi[c] + tinc*sizeinc[binc]*rinc[3*c+binc];
    du[r] = sizeinc[bu]*rinc[3*c+bu];
    dv[r] = sizeinc[bv]*rinc[3*c+bv];
  }
  cx[2] = sgn*cx[2] - sizeref[a];
  du[2] *= sgn;
  dv[2] *= sgn;
  for (int k = 0; k < 4; k++) {
    const real su = (k == 0 || k == 3) ? 1 : -1, sv = (k < 2) ? 1 : -1;
    poly[0][k][0] = cx[0] + su*du[0] + sv*dv[0];
    poly[0][k][1] = cx[1] + su*du[1] + sv*dv[1];
    poly[0][k][2] = cx[2] + su*du[2] + sv*dv[2];
  }
  int nvert = 4, cur = 0;
  nvert = bb_clip(poly, &cur, nvert, 0, 1, sizeref[ax]);
  nvert = bb_clip(poly, &cur, nvert, 0, -1, sizeref[ax]);
  nvert = bb_clip(poly, &cur, nvert, 1, 1, sizeref[ay]);
  nvert = bb_clip(poly, &cur, nvert, 1, -1, sizeref[ay]);
  real accepted[MJH_BB_MAXVERT][3];
  int naccept = 0;
  const real dupe2 = MJH_BB_DUPEPS*(sizeref[ax]*sizeref[ax] + sizeref[ay]*sizeref[ay]);
  for (int k = 0; k < nvert; k++) {
    if (poly[cur][k][2] > margin) continue;
    int dupe = 0;
    for (int q = 0; q < naccept; q++) {
      const real dx = accepted[q][0] - poly[cur][k][0];
      const real dy = accepted[q][1] - poly[cur][k][1];
      if (dx*dx + dy*dy < dupe2) { dupe = 1; break; }
    }
    if (!dupe) { accepted[naccept][0] = poly[cur][k][0]; accepted[naccept][1] = poly[cur][k][1]; accepted[naccept][2] = poly[cur][k][2]; naccept++; }
  }
  if (naccept == 0) return 0;
  const real nsign = ref1 ? sgn : -sgn;
  real normal[3] = {nsign*matref[3*0+a], nsign*matref[3*1+a], nsign*matref[3*2+a]};
  for (int k = 0; k < naccept; k++) {
    real posc[3];
    posc[ax] = accepted[k][0];
    posc[ay] = accepted[k][1];
    posc[a] = sgn*(sizeref[a] + 0.5*accepted[k][2]);
    con[k].dist = accepted[k][2];
    m3_mulvec(tmp, matref, posc);
    v3_add(con[k].pos, tmp, posref);
    v3_copy(con[k].normal, normal);
    v3_zero(con[k].tangent);
  }
  return naccept;
}

template <class P0, class P1, class P2, class P3, class P4, class P5>
MJH_DEV int col_box_box(PreContact* con, real margin, P0 pos1, P1 mat1, P2 size1, P3 pos2, P4 mat2, P5 size2) {
  real g[30];
  for (int k = 0; k < 3; k++) { g[k] = pos1[k]; g[12 + k] = size1[k]; g[15 + k] = pos2[k]; g[27 + k] = size2[k]; }
  for (int k = 0; k < 9; k++) { g[3 + k] = mat1[k]; g[18 + k] = mat2[k]; }
  return col_box_box_impl(con, margin, g);
}

// mjc_PlaneBox (engine_collision_primitive.c:210-256): corners below the plane, at most 4
template <class P0, class P1, class P2, class P3, class P4>
MJH_DEV int col_plane_box(PreContact* con, real margin, P0 pos1, P1 mat1, P2 pos2, P3 mat2, P4 size2) {
  real norm[3] = {mat1[2], mat1[5], mat1[8]};
  real dif[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
  const real dist = v3_dot(dif, norm);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    real vec[3], corner[3];
    vec[0] = (i & 1) ? (real)size2[0] : -(real)size2[0];
    vec[1] = (i & 2) ? (real)size2[1] : -(real)size2[1];
    vec[2] = (i & 4) ? (real)size2[2] : -(real)size2[2];
    m3_mulvec(corner, mat2, vec);
    const real ldist = v3_dot(norm, corner);
    if (dist + ldist > margin || ldist > 0) continue;
    con[cnt].dist = dist + ldist;
    v3_copy(con[cnt].normal, norm);
    v3_addto(corner, pos2);
    v3_scl(vec, norm, -con[cnt].dist / 2);
    v3_add(con[cnt].pos, corner, vec);
    v3_zero(con[cnt].tangent);
    if (++cnt >= 4) return 4;
  }
  return cnt;
}

// mjraw_SphereBox (engine_collision_box.c:35-88)
template <class P0, class P1, class P2, class P3>
MJH_DEV int col_sphere_box(PreContact* c, real margin, P0 pos1, real r1, P1 pos2, P2 mat2, P3 size2) {
  real tmp[3], center[3], clamped[3], deepest[3], pos[3];
  v3_sub(tmp, pos1, pos2);
  m3_multvec(center, mat2, tmp);
  for (int i = 0; i < 3; i++) {
    clamped[i] = center[i];
    if (clamped[i] < -size2[i]) clamped[i] = -size2[i];
    else if (clamped[i] > size2[i]) clamped[i] = size2[i];
    deepest[i] = center[i];
  }
  v3_sub(tmp, clamped, center);
  real dist = v3_normalize(tmp);
  if (dist - r1 > margin) return 0;
  if (dist <= MJH_MINVAL) {
    // sphere centre inside the box: push out through the nearest face
    real closest = (size2[0] + size2[1] + size2[2]) * 2;
    int k = 0;
    for (int i = 0; i < 6; i++) {
      const real d = fabs(((i % 2) ? 1 : -1)*size2[i / 2] - center[i / 2]);
      if (closest > d) { closest = d; k = i; }
    }
    real nearest[3] = {0, 0, 0};
    nearest[k / 2] = (k % 2) ? -1 : 1;
    v3_copy(pos, center);
    v3_addtoscl(pos, nearest, (r1 - closest) / 2);
    m3_mulvec(c->normal, mat2, nearest);
    dist = -closest;
  } else {
    v3_addtoscl(deepest, tmp, r1);
    v3_zero(pos);
    v3_addtoscl(pos, clamped, 0.5);
    v3_addtoscl(pos, deepest, 0.5);
    m3_mulvec(c->normal, mat2, tmp);
  }
  m3_mulvec(tmp, mat2, pos);
  v3_add(c->pos, tmp, pos2);
  c->dist = dist - r1;
  v3_zero(c->tangent);
  return 1;
}

// mjc_SphereCylinder (engine_collision_primitive.c:345-421): side / cap / rim cases
template <class P0, class P1, class P2, class P3, class P4>
MJH_DEV int col_sphere_cylinder(PreContact* c, real margin, P0 pos1, P1 mat1, real r1, P2 pos2, P3 mat2, P4 size2) {
  const real radius = size2[0], height = size2[1];
  real axis[3] = {mat2[2], mat2[5], mat2[8]};
  real vec[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  const real x = v3_dot(axis, vec);
  real a_proj[3], p_proj[3];
  v3_scl(a_proj, axis, x);
  v3_sub(p_proj, vec, a_proj);
  const real p_proj_sqr = v3_dot(p_proj, p_proj);
  int collide_side = fabs(x) < height;
  int collide_cap = p_proj_sqr < radius*radius;
  if (collide_side && collide_cap) {
    const real dist_cap = height - fabs(x);
    const real dist_radius = radius - sqrt(p_proj_sqr);
    if (dist_cap < dist_radius) collide_side = 0; else collide_cap = 0;
  }
  if (collide_side) {
    v3_addto(a_proj, pos2);
    return col_sphere_sphere(c, margin, pos1, mat1, r1, a_proj, mat2, radius);
  }
  if (collide_cap) {
    real flipmat[9] = {-mat2[0], mat2[1], -mat2[2], -mat2[3], mat2[4], -mat2[5], -mat2[6], mat2[7], -mat2[8]};
    real capmat[9], pos_cap[3];
    const real hs = (x > 0) ? height : -height;
    for (int k = 0; k < 3; k++) pos_cap[k] = pos2[k] + axis[k]*hs;
    for (int k = 0; k < 9; k++) capmat[k] = (x > 0) ? (real)mat2[k] : flipmat[k];
    int n = col_plane_sphere(c, margin, pos_cap, capmat, pos1, r1);
    if (n) { c->normal[0] *= -1; c->normal[1] *= -1; c->normal[2] *= -1; }
    return n;
  }
  // rim: point sphere at the nearest point of the cap's edge
  v3_scl(p_proj, p_proj, radius / sqrt(p_proj_sqr));
  v3_scl(vec, axis, x > 0 ? height : -height);
  v3_addto(vec, p_proj);
  v3_addto(vec, pos2);
  return col_sphere_sphere(c, margin, pos1, mat1, r1, vec, mat2, (real)0);
}

// mjc_PlaneCylinder, engine_collision_primitive.c:101-208 (up to 4 contacts)
template <class P0, class P1, class P2, class P3, class P4>
MJH_DEV int col_plane_cylinder(PreContact* con, real margin, P0 pos1, P1 mat1, P2 pos2, P3 mat2, P4 size2) {
  real normal[3] = {mat1[2], mat1[5], mat1[8]};
  real axis[3] = {mat2[2], mat2[5], mat2[8]};
  real prjaxis = v3_dot(normal, axis);
  if (prjaxis > 0) {
    v3_scl(axis, axis, -1);
    prjaxis = -prjaxis;
  }
  real vec[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
  real dist0 = v3_dot(vec, normal);
  v3_scl(vec, axis, prjaxis);
  v3_subfrom(vec, normal);
  real len_sqr = v3_dot(vec, vec);
  if (len_sqr >= MJH_MINVAL*MJH_MINVAL) {
    real scl = size2[0]/sqrt(len_sqr);
    vec[0] *= scl; vec[1] *= scl; vec[2] *= scl;
  } else {
    vec[0] = mat2[0]*size2[0];
    vec[1] = mat2[3]*size2[0];
    vec[2] = mat2[6]*size2[0];
  }
  real prjvec = v3_dot(vec, normal);
  v3_scl(axis, axis, size2[1]);
  prjaxis *= size2[1];
  int cnt = 0;
  if (dist0 + prjaxis + prjvec <= margin) {
    con[cnt].dist = dist0 + prjaxis + prjvec;
    v3_add(con[cnt].pos, pos2, vec);
    v3_addto(con[cnt].pos, axis);
    v3_addtoscl(con[cnt].pos, normal, -con[cnt].dist * 0.5);
    v3_copy(con[cnt].normal, normal);
    v3_zero(con[cnt].tangent);
    cnt++;
  } else {
    return 0;
  }
  if (dist0 - prjaxis + prjvec <= margin) {
    con[cnt].dist = dist0 - prjaxis + prjvec;
    v3_add(con[cnt].pos, pos2, vec);
    v3_subfrom(con[cnt].pos, axis);
    v3_addtoscl(con[cnt].pos, normal, -con[cnt].dist * 0.5);
    v3_copy(con[cnt].normal, normal);
    v3_zero(con[cnt].tangent);
    cnt++;
  }
  real prjvec1 = -prjvec*0.5;
  if (dist0 + prjaxis + prjvec1 <= margin) {
    real vec1[3];
    v3_cross(vec1, vec, axis);
    v3_normalize(vec1);
    v3_scl(vec1, vec1, size2[0] * sqrt(3.0) / 2);
    con[cnt].dist = dist0 + prjaxis + prjvec1;
    v3_add(con[cnt].pos, pos2, vec1);
    v3_addto(con[cnt].pos, axis);
    v3_addtoscl(con[cnt].pos, vec, -0.5);
    v3_addtoscl(con[cnt].pos, normal, -con[cnt].dist * 0.5);
    v3_copy(con[cnt].normal, normal);
    v3_zero(con[cnt].tangent);
    cnt++;
    con[cnt].dist = dist0 + prjaxis + prjvec1;
    v3_sub(con[cnt].pos, pos2, vec1);
    v3_addto(con[cnt].pos, axis);
    v3_addtoscl(con[cnt].pos, vec, -0.5);
    v3_addtoscl(con[cnt].pos, normal, -con[cnt].dist * 0.5);
    v3_copy(con[cnt].normal, normal);
    v3_zero(con[cnt].tangent);
    cnt++;
  }
  return cnt;
}

// complete a contact frame from its normal (+ optional tangent)   (mju_makeFrame, engine_util_spatial.c:512)
template <class P0>
MJH_DEV void make_frame(P0 frame) {
  v3_normalize(frame);
  if (v3_dot(frame + 3, frame + 3) < 0.25) {
    v3_zero(frame + 3);
    if (frame[1] < 0.5 && frame[1] > -0.5) frame[4] = 1;
    else frame[5] = 1;
  }
  real tmp[3];
  v3_scl(tmp, frame, v3_dot(frame, frame + 3));
  v3_subfrom(frame + 3, tmp);
  v3_normalize(frame + 3);
  v3_cross(frame + 6, frame, frame + 3);
}

// mj_filterSphere, engine_collision_driver.c:267: 1 = cull
template <class P0, class P1>
MJH_DEV int filter_sphere(MREF M, P0 gx, P1 gm, int g1, int g2, real margin) {
  real rb1 = M.geom_rbound[g1], rb2 = M.geom_rbound[g2];
  if (rb1 > 0 && rb2 > 0) {
    crptr p1 = gx + 3*g1; crptr p2 = gx + 3*g2;
    real bound = rb1 + rb2 + margin;
    real dif[3] = {p1[0]-p2[0], p1[1]-p2[1], p1[2]-p2[2]};
    real d2 = dif[0]*dif[0] + dif[1]*dif[1] + dif[2]*dif[2];
    return d2 > bound*bound;
  }
  if (M.geom_type[g1] == MJH_GEOM_PLANE && rb2 > 0) {
    crptr m1 = gm + 9*g1;
    real nrm[3] = {m1[2], m1[5], m1[8]};
    real dif[3];
    v3_sub(dif, gx + 3*g2, gx + 3*g1);
    if (v3_dot(dif, nrm) > margin + rb2) return 1;
  }
  if (M.geom_type[g2] == MJH_GEOM_PLANE && rb1 > 0) {
    crptr m2 = gm + 9*g2;
    real nrm[3] = {m2[2], m2[5], m2[8]};
    real dif[3];
    v3_sub(dif, gx + 3*g1, gx + 3*g2);
    if (v3_dot(dif, nrm) > margin + rb1) return 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// mj_collision over the static pair list
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_collision(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  iptr counts = MJH_F(B, counts, e);
  const int dsbl = M.o.disableflags;
  if ((dsbl & (1<<0)) || (dsbl & (1<<4)) || s.npair == 0) {
    if (wv_lane() == 0) counts[MJH_C_NCON] = 0;
    wv_sync();
    return;
  }
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  iptr warn = MJH_F(B, warning, e);

  int base = 0;        // contacts emitted by earlier chunks (wave-uniform)
  int overflow = 0;
  for (int p0 = 0; p0 < s.npair; p0 += MJH_W) {
    int p = p0 + wv_lane();
    PreContact pc[8];
    int n = 0;
    int unsupported = 0;
    if (p < s.npair) {
      int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
      real margin = M.pair_margin[p];       // margin + gap: collider threshold
      if (!filter_sphere(M, gx, gm, g1, g2, margin)) {
        crptr pos1 = gx + 3*g1; crptr mat1 = gm + 9*g1; auto size1 = M.geom_size + 3*g1;
        crptr pos2 = gx + 3*g2; crptr mat2 = gm + 9*g2; auto size2 = M.geom_size + 3*g2;
        switch (M.pair_func[p]) {
          case MJH_COL_PLANE_SPHERE:
            n = col_plane_sphere(pc, margin, pos1, mat1, pos2, size2[0]); break;
          case MJH_COL_PLANE_CAPSULE:
            n = col_plane_capsule(pc, margin, pos1, mat1, pos2, mat2, size2); break;
          case MJH_COL_SPHERE_SPHERE:
            n = col_sphere_sphere(pc, margin, pos1, mat1, size1[0], pos2, mat2, size2[0]); break;
          case MJH_COL_SPHERE_CAPSULE:
            n = col_sphere_capsule(pc, margin, pos1, mat1, size1[0], pos2, mat2, size2); break;
          case MJH_COL_CAPSULE_CAPSULE:
            n = col_capsule_capsule(pc, margin, pos1, mat1, size1, pos2, mat2, size2); break;
          case MJH_COL_PLANE_CYLINDER:
            n = col_plane_cylinder(pc, margin, pos1, mat1, pos2, mat2, size2); break;
          case MJH_COL_PLANE_BOX:
            n = col_plane_box(pc, margin, pos1, mat1, pos2, mat2, size2); break;
          case MJH_COL_SPHERE_BOX:
            n = col_sphere_box(pc, margin, pos1, size1[0], pos2, mat2, size2); break;
          case MJH_COL_BOX_BOX:
            n = col_box_box(pc, margin, pos1, mat1, size1, pos2, mat2, size2); break;
          case MJH_COL_SPHERE_CYLINDER:
            n = col_sphere_cylinder(pc, margin, pos1, mat1, size1[0], pos2, mat2, size2); break;
          case MJH_COL_UNSUPPORTED:
            unsupported = 1; break;
          default: break;
        }
      }
    }
    // a pair whose collider mjhip does not have reached the narrowphase: the result could differ
    // from the reference's, so the environment is flagged (and frozen by the rollout loop)
    if (wv_any(unsupported) && wv_lane() == 0) warn[MJH_WARN_UNSUPPORTED]++;
    int off = base + wv_exscan_i(n);
    int total = wv_sum_i(n);
    for (int k = 0; k < n; k++) {
      int c = off + k;
      if (c >= s.nconmax) { overflow = 1; continue; }
      // mj_narrowphase fill + mj_setContact, engine_collision_driver.c:2050-2075, :1839-1875
      MJH_CON(B, con_dist, e, 1, c)[0] = pc[k].dist;
      v3_copy(MJH_CON(B, con_pos, e, 3, c), pc[k].pos);
      real fr[9];
      v3_copy(fr, pc[k].normal);
      v3_copy(fr + 3, pc[k].tangent);
      v3_zero(fr + 6);
      make_frame(fr);
      rptr cframe = MJH_CON(B, con_frame, e, 9, c);
      for (int q = 0; q < 9; q++) cframe[q] = fr[q];
      MJH_CON(B, con_pair, e, 1, c)[0] = p;
      iptr cgeom = MJH_CON(B, con_geom, e, 2, c);
      cgeom[0] = M.pair_geom1[p];
      cgeom[1] = M.pair_geom2[p];
      MJH_CON(B, con_dim, e, 1, c)[0] = M.pair_dim[p];
      MJH_CON(B, con_exclude, e, 1, c)[0] = (pc[k].dist >= M.pair_includemargin[p]) ? 1 : 0;
      MJH_CON(B, con_efcadr, e, 1, c)[0] = -1;
      MJH_CON(B, con_mu, e, 1, c)[0] = 0;
    }
    base += total;
  }
  overflow = wv_any(overflow);
  if (wv_lane() == 0) {
    if (overflow) {
      // the reference drops the whole narrowphase batch when the arena is full
      // (engine_collision_driver.c:2028-2031); we keep the first nconmax and raise the same warning
      warn[MJH_WARN_CONTACTFULL]++;
      counts[MJH_C_NCON] = s.nconmax;
    } else {
      counts[MJH_C_NCON] = base;
    }
  }
  wv_sync();
}
