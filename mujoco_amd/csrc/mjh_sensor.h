// Sensors: mj_sensorPos / mj_sensorVel / mj_sensorAcc (engine_sensor.c:1498-1660) evaluated once
// per mj_forward after the constraint solve, with the support computations they pull in:
// mj_subtreeVel (engine_core_smooth.c:2249), mj_rnePostConstraint (:2394), mj_objectVelocity /
// mj_objectAcceleration (engine_core_util.c:835, :909), mj_contactForce (:1075).
// Models with sensors run without an LDS residency plan (every field in its global home), so all
// the kinematic / velocity / constraint quantities read here are still valid.
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)


// mju_transformSpatial (engine_util_spatial.c:482): motion (flg_force 0) or force vector moved
// from oldpos to newpos, optionally rotated into the frame rot (new -> old)
template <class P0, class P1, class P2, class P3>
MJH_DEV void sp_transform(real* res, P0 vec, int flg_force, P1 newpos, P2 oldpos, P3 rot, int has_rot) {
  real dif[3], cros[3], tran[6];
  for (int k = 0; k < 6; k++) tran[k] = vec[k];
  v3_sub(dif, newpos, oldpos);
  if (flg_force) {
    real f[3] = {vec[3], vec[4], vec[5]};
    v3_cross(cros, dif, f);
    tran[0] = vec[0] - cros[0]; tran[1] = vec[1] - cros[1]; tran[2] = vec[2] - cros[2];
  } else {
    real w[3] = {vec[0], vec[1], vec[2]};
    v3_cross(cros, dif, w);
    tran[3] = vec[3] - cros[0]; tran[4] = vec[4] - cros[1]; tran[5] = vec[5] - cros[2];
  }
  if (has_rot) {
    m3_multvec(res, rot, tran);
    m3_multvec(res + 3, rot, tran + 3);
  } else {
    for (int k = 0; k < 6; k++) res[k] = tran[k];
  }
}

// frame of a sensor object: position, orientation matrix, carrying body
struct SensFrame { crptr pos, mat; int body; };
MJH_DEV SensFrame sens_frame(MREF M, BREF B, int e, int objtype, int id) {
  SensFrame f;
  if (objtype == MJH_OBJ_BODY) { f.pos = MJH_F(B, xipos, e) + 3*id; f.mat = MJH_F(B, ximat, e) + 9*id; f.body = id; }
  else if (objtype == MJH_OBJ_XBODY) { f.pos = MJH_F(B, xpos, e) + 3*id; f.mat = MJH_F(B, xmat, e) + 9*id; f.body = id; }
  else if (objtype == MJH_OBJ_GEOM) { f.pos = MJH_F(B, geom_xpos, e) + 3*id; f.mat = MJH_F(B, geom_xmat, e) + 9*id; f.body = M.geom_bodyid[id]; }
  else if (objtype == MJH_OBJ_CAMERA) { f.pos = MJH_G(B, cam_xpos, e) + 3*id; f.mat = MJH_G(B, cam_xmat, e) + 9*id; f.body = M.cam_bodyid[id]; }
  else { f.pos = MJH_F(B, site_xpos, e) + 3*id; f.mat = MJH_F(B, site_xmat, e) + 9*id; f.body = M.site_bodyid[id]; }
  return f;
}
MJH_DEV void sens_quat(MREF M, BREF B, int e, int objtype, int id, real* quat) {
  crptr xquat = MJH_F(B, xquat, e);
  if (objtype == MJH_OBJ_XBODY) q_copy(quat, xquat + 4*id);
  else if (objtype == MJH_OBJ_BODY) q_mul(quat, xquat + 4*id, M.body_iquat + 4*id);
  else if (objtype == MJH_OBJ_GEOM) q_mul(quat, xquat + 4*M.geom_bodyid[id], M.geom_quat + 4*id);
  else if (objtype == MJH_OBJ_CAMERA) q_mul(quat, xquat + 4*M.cam_bodyid[id], M.cam_quat + 4*id);
  else q_mul(quat, xquat + 4*M.site_bodyid[id], M.site_quat + 4*id);
}

// mj_objectVelocity: 6D velocity [rot; lin] of a frame object, world or local orientation
MJH_DEV void object_velocity(MREF M, BREF B, int e, int objtype, int id, real* res, int flg_local) {
  const SensFrame f = sens_frame(M, B, e, objtype, id);
  if (M.body_dofnum[M.body_weldid[f.body]] == 0) { for (int k = 0; k < 6; k++) res[k] = 0; return; }
  sp_transform(res, MJH_F(B, cvel, e) + 6*f.body, 0, f.pos,
               MJH_F(B, subtree_com, e) + 3*M.body_rootid[f.body], f.mat, flg_local);
}
// mj_objectAcceleration: needs cacc of mj_rnePostConstraint; adds the rotating-frame correction
MJH_DEV void object_acceleration(MREF M, BREF B, int e, int objtype, int id, real* res, int flg_local) {
  const SensFrame f = sens_frame(M, B, e, objtype, id);
  if (M.body_dofnum[M.body_weldid[f.body]] == 0) { for (int k = 0; k < 6; k++) res[k] = 0; return; }
  crptr com = MJH_F(B, subtree_com, e) + 3*M.body_rootid[f.body];
  sp_transform(res, MJH_G(B, cacc_post, e) + 6*f.body, 0, f.pos, com, f.mat, flg_local);
  real vel[6], corr[3];
  sp_transform(vel, MJH_F(B, cvel, e) + 6*f.body, 0, f.pos, com, f.mat, flg_local);
  v3_cross(corr, vel, vel + 3);
  res[3] += corr[0]; res[4] += corr[1]; res[5] += corr[2];
}

// mj_contactForce: contact-frame [force; torque] from the constraint forces
MJH_DEV void contact_force(MREF M, BREF B, int e, const Efc& P, int k, real* result) {
  for (int q = 0; q < 6; q++) result[q] = 0;
  const int r0 = MJH_CON(B, con_efcadr, e, 1, k)[0];
  if (r0 < 0) return;
  const int dim = MJH_CON(B, con_dim, e, 1, k)[0];
  if (M.o.cone == 0) {
    // mju_decodePyramid (engine_util_misc.c:1584)
    if (dim == 1) { result[0] = P.force[r0]; return; }
    auto mu = M.pair_friction + 5*MJH_CON(B, con_pair, e, 1, k)[0];
    real fn = 0;
    for (int i = 0; i < 2*(dim - 1); i++) fn += P.force[r0 + i];
    result[0] = fn;
    for (int i = 0; i < dim - 1; i++) result[i + 1] = (P.force[r0 + 2*i] - P.force[r0 + 2*i + 1]) * mu[i];
  } else {
    for (int i = 0; i < dim; i++) result[i] = P.force[r0 + i];
  }
}

// mju_insideGeom (engine_util_misc.c:452-496): is the world point inside the site's volume?
template <class PT>
MJH_DEV int site_holds_point(MREF M, BREF B, int e, int site, PT pt) {
  crptr sp = MJH_F(B, site_xpos, e) + 3*site;
  crptr sm = MJH_F(B, site_xmat, e) + 9*site;
  auto sz = M.site_size + 3*site;
  const int st = M.site_type[site];
  real vec[3], pl[3];
  v3_sub(vec, pt, sp);
  if (st == 2) return v3_dot(vec, vec) < sz[0]*sz[0];
  m3_multvec(pl, sm, vec);
  if (st == 3) {
    const real z = pl[2], zc = r_clip(z, -sz[1], sz[1]);
    const real zd = (z - zc)*(z - zc);
    return pl[0]*pl[0] + pl[1]*pl[1] + zd < sz[0]*sz[0];
  }
  if (st == 4) return pl[0]*pl[0]/(sz[0]*sz[0]) + pl[1]*pl[1]/(sz[1]*sz[1]) + pl[2]*pl[2]/(sz[2]*sz[2]) < 1;
  if (st == 5) return fabs(pl[2]) < sz[1] && pl[0]*pl[0] + pl[1]*pl[1] < sz[0]*sz[0];
  if (st == 6) return fabs(pl[0]) < sz[0] && fabs(pl[1]) < sz[1] && fabs(pl[2]) < sz[2];
  return 0;
}

// matchContact (engine_sensor.c:339-392): 0 no match, 1 match, -1 match with the normal flipped
MJH_DEV int contact_matches(MREF M, BREF B, int e, int k, int type1, int id1, int type2, int id2) {
  if (type1 == MJH_OBJ_NONE && type2 == MJH_OBJ_NONE) return 1;
  if (type1 == MJH_OBJ_SITE && !site_holds_point(M, B, e, id1, MJH_CON(B, con_pos, e, 3, k))) return 0;
  ciptr cg = MJH_CON(B, con_geom, e, 2, k);
  const int g1 = cg[0], g2 = cg[1];
  const int b1 = M.geom_bodyid[g1], b2 = M.geom_bodyid[g2];
  auto check = [&](int body, int geom, int type, int id) -> int {
    if (type == MJH_OBJ_NONE || type == MJH_OBJ_SITE) return 1;
    if (type == MJH_OBJ_GEOM) return id == geom;
    if (type == MJH_OBJ_BODY) return id == body;
    while (body > id) body = M.body_parentid[body];        // (subtree: up the tree until at or above id)
    return body == id;
  };
  const int m11 = check(b1, g1, type1, id1), m12 = check(b2, g2, type1, id1);
  const int m21 = check(b1, g1, type2, id2), m22 = check(b2, g2, type2, id2);
  if (!m11 && !m12) return 0;
  if (!m21 && !m22) return 0;
  if (type1 != MJH_OBJ_NONE && type2 != MJH_OBJ_NONE) {
    const int regular = m11 && m22, reverse = m12 && m21;
    if (regular && !reverse) return 1;
    if (reverse && !regular) return -1;
    if (regular && reverse) return 1;
  } else if (type1 != MJH_OBJ_NONE) return m11 ? 1 : -1;
  else if (type2 != MJH_OBJ_NONE) return m22 ? 1 : -1;
  return 0;
}

// ---- rays against primitive shapes (mju_rayGeom, engine_ray.c:103-560, distances only) ----------
// A ray is carried in the shape's own frame: origin o, direction d (not normalised: distances are in
// units of |d|, like the reference).  Curved surfaces reduce to the roots of a x^2 + 2 b x + c = 0 with
// a = <d,d>_W, b = <d,o>_W, c = <o,o>_W - 1 for a diagonal metric W; flat faces to one division per
// slab.  The nearest admissible root wins; -1 = no hit.
struct RayRoots { real near_, far_; };
MJH_DEV RayRoots ray_roots(real a, real b, real c) {
  real disc = b*b - a*c;
  if (disc < 0 || a < MJH_MINVAL) return RayRoots{-1, -1};
  disc = sqrt(disc);
  return RayRoots{(-b - disc)/a, (-b + disc)/a};
}
MJH_DEV real ray_first(RayRoots r) { return r.near_ >= 0 ? r.near_ : (r.far_ >= 0 ? r.far_ : (real)-1); }
MJH_DEV void ray_keep_nearest(real& best, real x) { if (best < 0 || x < best) best = x; }
// (the same with the part of the surface that was hit: capsule / cylinder 0 = barrel, +-1 = cap; box 4 axis + 2 + side)
MJH_DEV void ray_keep_nearest(real& best, real x, int& part, int p) { if (best < 0 || x < best) { best = x; part = p; } }
// quadric <p,p>_W = radius2 about `centre`, metric W = diag(w)
MJH_DEV RayRoots ray_quadric(V3 o, V3 d, V3 centre, V3 w, real radius2) {
  const V3 p = o - centre;
  const real a = w.x*d.x*d.x + w.y*d.y*d.y + w.z*d.z*d.z;
  const real b = w.x*d.x*p.x + w.y*d.y*p.y + w.z*d.z*p.z;
  const real c = w.x*p.x*p.x + w.y*p.y*p.y + w.z*p.z*p.z - radius2;
  return ray_roots(a, b, c);
}
// the two faces perpendicular to axis k at +-half: keep the nearest crossing that `inside` accepts
template <class F>
MJH_DEV void ray_slab(real& best, V3 o, V3 d, int k, real half, F inside, int* part = nullptr, int base = 0) {
  const real dk = comp(d, k);
  if (!(fabs(dk) > MJH_MINVAL)) return;
  for (int side = -1; side <= 1; side += 2) {
    const real x = (side*half - comp(o, k))/dk;
    if (x >= 0 && inside(V3{o.x + x*d.x, o.y + x*d.y, o.z + x*d.z})) {
      if (part) ray_keep_nearest(best, x, *part, base + side); else ray_keep_nearest(best, x);
    }
  }
}

template <class P0, class P1, class P2>
MJH_DEV real ray_geom_dist(int type, P0 pos, P1 mat, P2 size, const real* pnt, const real* vec, int* part = nullptr) {
  const V3 wo{pnt[0] - pos[0], pnt[1] - pos[1], pnt[2] - pos[2]}, wd{vec[0], vec[1], vec[2]};
  const V3 one{1, 1, 1}, origin{0, 0, 0};
  // bounding sphere in world coordinates (the rotation does not change it)
  auto misses_ball = [&](real radius2) -> int { return ray_first(ray_quadric(wo, wd, origin, one, radius2)) < 0; };
  if (type == 2) return ray_first(ray_quadric(wo, wd, origin, one, size[0]*size[0]));          // sphere
  const V3 o{mtrow(mat, 0, wo), mtrow(mat, 1, wo), mtrow(mat, 2, wo)};
  const V3 d{mtrow(mat, 0, wd), mtrow(mat, 1, wd), mtrow(mat, 2, wd)};
  const V3 xy{1, 1, 0};                                                                       // metric of the z axis' cylinder
  if (type == 0) {                    // plane z = 0, seen from above; size[0], size[1] > 0 bound it
    if (d.z > -MJH_MINVAL) return -1;
    const real x = -o.z/d.z;
    if (x < 0) return -1;
    const real px = o.x + x*d.x, py = o.y + x*d.y;
    return ((size[0] <= 0 || fabs(px) <= size[0]) && (size[1] <= 0 || fabs(py) <= size[1])) ? x : (real)-1;
  }
  if (type == 3) {                    // capsule: barrel between the cap centres, a half sphere beyond each
    const real reach = size[0] + size[1];
    if (misses_ball(reach*reach)) return -1;
    real best = -1;
    const real r2 = size[0]*size[0];
    int pt = 0;
    const real barrel = ray_first(ray_quadric(o, d, origin, xy, r2));
    if (barrel >= 0 && fabs(o.z + barrel*d.z) <= size[1]) ray_keep_nearest(best, barrel, pt, 0);
    for (int cap = 1; cap >= -1; cap -= 2) {
      const RayRoots rr = ray_quadric(o, d, V3{0, 0, cap*size[1]}, one, r2);
      for (int k = 0; k < 2; k++) {
        const real x = k ? rr.far_ : rr.near_;
        if (x >= 0 && cap*(o.z + x*d.z) >= size[1]) ray_keep_nearest(best, x, pt, cap);
      }
    }
    if (part) *part = pt;
    return best;
  }
  if (type == 4)                      // ellipsoid
    return ray_first(ray_quadric(o, d, origin, V3{1/(size[0]*size[0]), 1/(size[1]*size[1]), 1/(size[2]*size[2])}, 1));
  if (type == 5) {                    // cylinder: the two caps, then the barrel
    if (misses_ball(size[0]*size[0] + size[1]*size[1])) return -1;
    real best = -1;
    const real r2 = size[0]*size[0];
    int pt = 0;
    ray_slab(best, o, d, 2, size[1], [&](V3 p) -> int { return p.x*p.x + p.y*p.y <= r2; }, &pt, 0);
    const real barrel = ray_first(ray_quadric(o, d, origin, xy, r2));
    if (barrel >= 0 && fabs(o.z + barrel*d.z) <= size[1]) ray_keep_nearest(best, barrel, pt, 0);
    if (part) *part = pt;
    return best;
  }
  // box: three slabs
  if (misses_ball(size[0]*size[0] + size[1]*size[1] + size[2]*size[2])) return -1;
  real best = -1;
  int pt = 0;
  for (int k = 0; k < 3; k++) {
    const int u = (k == 0) ? 1 : 0, v = (k == 2) ? 1 : 2;
    ray_slab(best, o, d, k, size[k], [&](V3 p) -> int { return fabs(comp(p, u)) <= size[u] && fabs(comp(p, v)) <= size[v]; }, &pt, 4*k + 2);
  }
  if (part) *part = pt;
  return best;
}

// surface normal at the intersection x of the ray with a primitive (mju_rayGeom's `normal` output, engine_ray.c:204-560:
// the gradient of the surface that was hit -- `part` from ray_geom_dist -- in the shape's frame, normalised, rotated into
// the world frame)
template <class P0, class P1, class P2>
MJH_DEV void ray_geom_normal(int type, P0 pos, P1 mat, P2 size, const real* pnt, const real* vec, real x, int part, real* out) {
  auto rotate = [&](const real* n) {            // mju_mulMatVec3
    const real r0 = mat[0]*n[0] + mat[1]*n[1] + mat[2]*n[2];
    const real r1 = mat[3]*n[0] + mat[4]*n[1] + mat[5]*n[2];
    const real r2 = mat[6]*n[0] + mat[7]*n[1] + mat[8]*n[2];
    out[0] = r0; out[1] = r1; out[2] = r2;
  };
  if (type == 0) { out[0] = mat[2]; out[1] = mat[5]; out[2] = mat[8]; return; }
  if (type == 2) {
    real n[3] = {(pnt[0] + vec[0]*x) - pos[0], (pnt[1] + vec[1]*x) - pos[1], (pnt[2] + vec[2]*x) - pos[2]};
    v3_normalize(n);
    out[0] = n[0]; out[1] = n[1]; out[2] = n[2];
    return;
  }
  const real dif[3] = {pnt[0] - pos[0], pnt[1] - pos[1], pnt[2] - pos[2]};
  const real lp[3] = {mat[0]*dif[0] + mat[3]*dif[1] + mat[6]*dif[2], mat[1]*dif[0] + mat[4]*dif[1] + mat[7]*dif[2], mat[2]*dif[0] + mat[5]*dif[1] + mat[8]*dif[2]};
  const real lv[3] = {mat[0]*vec[0] + mat[3]*vec[1] + mat[6]*vec[2], mat[1]*vec[0] + mat[4]*vec[1] + mat[7]*vec[2], mat[2]*vec[0] + mat[5]*vec[1] + mat[8]*vec[2]};
  real n[3] = {0, 0, 0};
  if (type == 3) {
    n[0] = lp[0] + lv[0]*x; n[1] = lp[1] + lv[1]*x;
    n[2] = part == 0 ? (real)0 : (real)(lp[2] + lv[2]*x - size[1]*part);
    v3_normalize(n);
  } else if (type == 4) {
    const real sc[3] = {1/(size[0]*size[0]), 1/(size[1]*size[1]), 1/(size[2]*size[2])};
    for (int k = 0; k < 3; k++) n[k] = sc[k]*(lp[k] + lv[k]*x);
    v3_normalize(n);
  } else if (type == 5) {
    if (part == 0) { n[0] = lp[0] + lv[0]*x; n[1] = lp[1] + lv[1]*x; n[2] = 0; v3_normalize(n); }
    else n[2] = part;
  } else {
    n[part >> 2] = (part & 3) == 3 ? (real)1 : (real)-1;        // (part = 4 axis + 2 + side)
  }
  rotate(n);
}
// does the ray meet a site's zone at all (touch sensors: sphere, ellipsoid, box, capsule and cylinder zones)?
template <class P0, class P1, class P2>
MJH_DEV int ray_hits_zone(int type, P0 pos, P1 mat, P2 size, const real* pnt, const real* vec) {
  return ray_geom_dist(type, pos, mat, size, pnt, vec) >= 0;
}

// mj_subtreeVel (lane 0)
MJH_DEV void sens_subtree_vel(MREF M, BREF B, int e) {
  const MJH_CONST_AS DSizes& s = M.s;
  rptr bvel = MJH_G(B, sens_bvel, e);
  rptr linvel = MJH_G(B, subtree_linvel, e);
  rptr angmom = MJH_G(B, subtree_angmom, e);
  crptr ximat = MJH_F(B, ximat, e);
  crptr xipos = MJH_F(B, xipos, e);
  crptr com = MJH_F(B, subtree_com, e);
  for (int i = 0; i < s.nbody; i++) {
    real v[6];
    object_velocity(M, B, e, MJH_OBJ_BODY, i, v, 0);
    for (int k = 0; k < 6; k++) bvel[6*i + k] = v[k];
    for (int k = 0; k < 3; k++) linvel[3*i + k] = v[3 + k]*M.body_mass[i];
    real dv[3], out[3];
    m3_multvec(dv, ximat + 9*i, v);
    dv[0] *= M.body_inertia[3*i]; dv[1] *= M.body_inertia[3*i+1]; dv[2] *= M.body_inertia[3*i+2];
    m3_mulvec(out, ximat + 9*i, dv);
    for (int k = 0; k < 3; k++) angmom[3*i + k] = out[k];
  }
  for (int i = s.nbody - 1; i >= 0; i--) {
    if (i) for (int k = 0; k < 3; k++) linvel[3*M.body_parentid[i] + k] += linvel[3*i + k];
    const real inv = 1/r_max(MJH_MINVAL, M.body_subtreemass[i]);
    for (int k = 0; k < 3; k++) linvel[3*i + k] = linvel[3*i + k]*inv;
  }
  for (int i = s.nbody - 1; i > 0; i--) {
    const int parent = M.body_parentid[i];
    real dx[3], dv[3], dp[3], dL[3];
    v3_sub(dx, xipos + 3*i, com + 3*i);
    for (int k = 0; k < 3; k++) dv[k] = bvel[6*i + 3 + k] - linvel[3*i + k];
    v3_scl(dp, dv, M.body_mass[i]);
    v3_cross(dL, dx, dp);
    for (int k = 0; k < 3; k++) angmom[3*i + k] += dL[k];
    for (int k = 0; k < 3; k++) angmom[3*parent + k] += angmom[3*i + k];
    v3_sub(dx, com + 3*i, com + 3*parent);
    for (int k = 0; k < 3; k++) dv[k] = linvel[3*i + k] - linvel[3*parent + k];
    v3_scl(dv, dv, M.body_subtreemass[i]);
    v3_cross(dL, dx, dv);
    for (int k = 0; k < 3; k++) angmom[3*parent + k] += dL[k];
  }
}

// mj_rnePostConstraint (lane 0): cacc, cfrc_ext (contacts, connect/weld), cfrc_int
MJH_DEV void sens_rne_post(MREF M, BREF B, int e) {
  const MJH_CONST_AS DSizes& s = M.s;
  rptr cacc = MJH_G(B, cacc_post, e);
  rptr cint = MJH_G(B, cfrc_int, e);
  rptr cext = MJH_G(B, cfrc_ext, e);
  crptr com = MJH_F(B, subtree_com, e);
  ciptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ncon = counts[MJH_C_NCON], ne = counts[MJH_C_NE];
  for (int k = 0; k < 6; k++) cacc[k] = 0;
  if (!(M.o.disableflags & (1<<7))) for (int k = 0; k < 3; k++) cacc[3 + k] = M.o.gravity[k]*-1;
  for (int k = 0; k < 6*s.nbody; k++) cext[k] = 0;
  Efc P;
  if (nefc) efc_layout(M, B, e, nefc, P);
  // contacts
  for (int c = 0; nefc && c < ncon; c++) {
    if (MJH_CON(B, con_efcadr, e, 1, c)[0] < 0) continue;
    real lfrc[6], cfrc[6], cc[6];
    contact_force(M, B, e, P, c, lfrc);
    crptr fr = MJH_CON(B, con_frame, e, 9, c);
    crptr cpos = MJH_CON(B, con_pos, e, 3, c);
    m3_multvec(cfrc, fr, lfrc + 3);
    m3_multvec(cfrc + 3, fr, lfrc);
    ciptr cg = MJH_CON(B, con_geom, e, 2, c);
    int k = M.geom_bodyid[cg[0]];
    if (k) {
      sp_transform(cc, cfrc, 1, com + 3*M.body_rootid[k], cpos, cpos, 0);
      for (int q = 0; q < 6; q++) cext[6*k + q] -= cc[q];
    }
    k = M.geom_bodyid[cg[1]];
    if (k) {
      sp_transform(cc, cfrc, 1, com + 3*M.body_rootid[k], cpos, cpos, 0);
      for (int q = 0; q < 6; q++) cext[6*k + q] += cc[q];
    }
  }
  // connect / weld equalities
  for (int i = 0; i < ne; ) {
    const int id = P.id[i];
    const int et = M.eq_type[id];
    if (et != MJH_EQ_CONNECT && et != MJH_EQ_WELD) { i++; continue; }
    real cfrc[6], cc[6], pos0[3], pos1[3];
    for (int q = 0; q < 3; q++) { cfrc[3 + q] = P.force[i + q]; cfrc[q] = (et == MJH_EQ_WELD) ? (real)P.force[i + 3 + q] : (real)0; }
    int b0, b1;
    equality_anchors(M, B, e, id, pos0, pos1, &b0, &b1);
    if (b0) {
      sp_transform(cc, cfrc, 1, com + 3*M.body_rootid[b0], pos0, pos0, 0);
      for (int q = 0; q < 6; q++) cext[6*b0 + q] += cc[q];
    }
    if (b1) {
      sp_transform(cc, cfrc, 1, com + 3*M.body_rootid[b1], pos1, pos1, 0);
      for (int q = 0; q < 6; q++) cext[6*b1 + q] -= cc[q];
    }
    i += (et == MJH_EQ_WELD) ? 6 : 3;
  }
  // forward pass: cacc, cfrc_int = cfrc_body - cfrc_ext ; backward pass: accumulate to parents
  crptr cdof = MJH_F(B, cdof, e);
  crptr cdof_dot = MJH_F(B, cdof_dot, e);
  crptr cvel = MJH_F(B, cvel, e);
  crptr cinert = MJH_F(B, cinert, e);
  crptr qvel = MJH_F(B, qvel, e);
  crptr qacc = MJH_F(B, qacc, e);
  for (int k = 0; k < 6; k++) cint[k] = 0;
  for (int j = 1; j < s.nbody; j++) {
    const int bda = M.body_dofadr[j];
    real tmp[6], fb[6], fc[6], fx[6];
    mul_dof_vec(tmp, cdof_dot + 6*bda, qvel + bda, M.body_dofnum[j]);
    for (int q = 0; q < 6; q++) cacc[6*j + q] = cacc[6*M.body_parentid[j] + q] + tmp[q];
    mul_dof_vec(tmp, cdof + 6*bda, qacc + bda, M.body_dofnum[j]);
    for (int q = 0; q < 6; q++) cacc[6*j + q] += tmp[q];
    sp_mul_inert(fb, cinert + 10*j, cacc + 6*j);
    sp_mul_inert(fc, cinert + 10*j, cvel + 6*j);
    sp_cross_force(fx, cvel + 6*j, fc);
    for (int q = 0; q < 6; q++) fb[q] += fx[q];
    for (int q = 0; q < 6; q++) cint[6*j + q] = fb[q] - cext[6*j + q];
  }
  for (int j = s.nbody - 1; j > 0; j--)
    for (int q = 0; q < 6; q++) cint[6*M.body_parentid[j] + q] += cint[6*j + q];
}

// mj_geomDistance (engine_support.c:553-604) for pairs with a closed-form collider: the collider runs with the cutoff as its
// margin, the contact of smallest distance wins (the first one on ties), and the segment runs from the surface of geom1
// to the surface of geom2 through the contact point.  Returns the distance (distmax when nothing is in reach).
template <class GX, class GM>
MJH_DEV real sens_geom_distance(MREF M, GX gx, GM gm, int geom1, int geom2, real distmax, real* fromto) {
  for (int k = 0; k < 6; k++) fromto[k] = 0;
  const int flip = M.geom_type[geom1] > M.geom_type[geom2];
  const int g1 = flip ? geom2 : geom1, g2 = flip ? geom1 : geom2;
  const int t1 = M.geom_type[g1], t2 = M.geom_type[g2];
  auto mat1 = gm + 9*g1; auto size1 = M.geom_size + 3*g1;
  auto mat2 = gm + 9*g2; auto size2 = M.geom_size + 3*g2;
  const V3 c1 = ld3(gx + 3*g1), c2 = ld3(gx + 3*g2);
  Hit ha, hb;
  int n = 0;
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_SPHERE) n = hit_plane_sphere(ha, distmax, c1, mcol(mat1, 2), c2, size2[0]);
  else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CAPSULE) n = hit_plane_capsule(ha, hb, distmax, c1, mcol(mat1, 2), c2, mat2, size2);
  else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_SPHERE) n = hit_sphere_sphere(ha, distmax, c1, mat1, size1[0], c2, mat2, size2[0]);
  else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_CAPSULE) n = hit_sphere_capsule(ha, distmax, c1, mat1, size1[0], c2, mat2, size2);
  else if (t1 == MJH_GEOM_CAPSULE && t2 == MJH_GEOM_CAPSULE) n = hit_capsule_capsule(ha, hb, distmax, c1, mat1, size1, c2, mat2, size2);
  else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_BOX) n = hit_sphere_box(ha, distmax, c1, size1[0], c2, mat2, size2);
  else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_CYLINDER) n = hit_sphere_cylinder(ha, distmax, c1, mat1, size1[0], c2, mat2, size2);
  real dist = distmax;
  int smallest = -1;
  if (n > 0 && ha.dist < dist) { dist = ha.dist; smallest = 0; }
  if (n > 1 && hb.dist < dist) { dist = hb.dist; smallest = 1; }
  if (smallest >= 0) {
    const V3 pos = smallest ? hb.pos : ha.pos, nrm = smallest ? hb.nrm : ha.nrm;
    const real sign = flip ? -1 : 1;
    const real s0 = -0.5*sign*dist, s1 = 0.5*sign*dist;
    fromto[0] = pos.x + nrm.x*s0; fromto[1] = pos.y + nrm.y*s0; fromto[2] = pos.z + nrm.z*s0;
    fromto[3] = pos.x + nrm.x*s1; fromto[4] = pos.y + nrm.y*s1; fromto[5] = pos.z + nrm.z*s1;
  }
  return dist;
}

// mj_energyPos (engine_sensor.c:1659-1762): gravitational potential of the body inertial frames, joint springs (with
// their polynomial terms: mju_polyPotential, engine_util_misc.c:2344), tendon springs outside their dead band.  One lane,
// sums in the reference's order.  (Flex edge springs: sensors in models with flexes are rejected.)
MJH_DEV void sens_energy_pos(MREF M, BREF B, int e) {
  const MJH_CONST_AS DSizes& s = M.s;
  crptr xipos = MJH_F(B, xipos, e);
  crptr qpos = MJH_F(B, qpos, e);
  auto pot = [&](real linear, auto poly, real x) -> real {
    real res = 0.5*linear*(x*x);
    real xpow = x;
    for (int i = 0; i < 2; i++) { xpow *= x; res += poly[i]/(i + 3)*(xpow*x); }
    return res;
  };
  real en = 0;
  if (!(M.o.disableflags & (1<<7)))
    for (int i = 1; i < s.nbody; i++) en -= M.body_mass[i]*(M.o.gravity[0]*xipos[3*i] + M.o.gravity[1]*xipos[3*i + 1] + M.o.gravity[2]*xipos[3*i + 2]);
  if (!(M.o.disableflags & (1<<5))) {
    for (int j = 0; j < s.njnt; j++) {          // (joints in body order are joints in index order)
      const real stiffness = M.jnt_stiffness[j];
      auto poly = M.jnt_stiffnesspoly + 2*j;
      if (stiffness == 0 && poly[0] == 0 && poly[1] == 0) continue;
      int padr = M.jnt_qposadr[j];
      const int jt = M.jnt_type[j];
      if (jt == MJH_JNT_FREE) {
        const real d0 = qpos[padr] - M.qpos_spring[padr], d1 = qpos[padr + 1] - M.qpos_spring[padr + 1], d2 = qpos[padr + 2] - M.qpos_spring[padr + 2];
        en += pot(stiffness, poly, sqrt(d0*d0 + d1*d1 + d2*d2));
        padr += 3;
      }
      if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
        real dif[3];
        q_sub(dif, qpos + padr, M.qpos_spring + padr);
        en += pot(stiffness, poly, sqrt(dif[0]*dif[0] + dif[1]*dif[1] + dif[2]*dif[2]));
      } else {
        en += pot(stiffness, poly, qpos[padr] - M.qpos_spring[padr]);
      }
    }
    crptr len = MJH_F(B, ten_length, e);
    for (int i = 0; i < s.ntendon; i++) {
      const real length = len[i], lower = M.tendon_lengthspring[2*i], upper = M.tendon_lengthspring[2*i + 1];
      const real x = (length > upper) ? length - upper : (length < lower) ? length - lower : (real)0;
      en += pot(M.tendon_stiffness[i], M.tendon_stiffnesspoly + 2*i, x);
    }
  }
  MJH_G(B, energy, e)[0] = en;
}

// mj_energyVel (:1766-1779): 0.5 qvel' M qvel with M v in mju_mulSymVecSparse's order (engine_util_sparse.c:254) and mju_dot
MJH_DEV void sens_energy_vel(MREF M, BREF B, int e) {
  const int nv = M.s.nv;
  crptr Mq = MJH_G(B, M, e);
  crptr qvel = MJH_F(B, qvel, e);
  rptr vec = MJH_G(B, scratch, e) + 5*M.s.nefcmax;          // (sqrtInvD's slot: dead after the projection stage)
  for (int i = 0; i < nv; i++) vec[i] = 0;
  for (int i = 0; i < nv; i++) {
    const int adr = M.M_rowadr[i], diag = M.M_rownnz[i] - 1;
    vec[i] = Mq[adr + diag]*qvel[i];
    for (int k = diag - 1; k >= 0; k--) {
      const int j = M.M_colind[adr + k];
      const real val = Mq[adr + k];
      vec[i] += val*qvel[j];
      vec[j] += val*qvel[i];
    }
  }
  MJH_G(B, energy, e)[1] = 0.5*dot_ref(vec, qvel, nv);
}

// which: bit 0 = position-stage sensors (mj_sensorPos, engine_sensor.c:369), bit 1 = velocity stage
// (mj_sensorVel, :640), bit 2 = acceleration stage (mj_sensorAcc, :868); a rollout step evaluates all
// three at once, mj_step1 / mj_step2 split them around the controller call
MJH_DEVN void stage_sensors(MREF M_, BREF B_, int e_, int which) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  if (!s.nsensor || (M.o.disableflags & (1<<13))) return;
  // camera frames (mj_camlight, engine_core_smooth.c:354-432), one lane per camera, before the position-stage sensors
  if (s.ncam_s && (which & 1)) {
    crptr xpos = MJH_F(B, xpos, e);
    crptr xquat = MJH_F(B, xquat, e);
    crptr xmat = MJH_F(B, xmat, e);
    crptr com = MJH_F(B, subtree_com, e);
    rptr cpos = MJH_G(B, cam_xpos, e);
    rptr cmat = MJH_G(B, cam_xmat, e);
    MJH_FOR_LANES(c, s.ncam_s) {
      const int id = M.cam_bodyid[c], id1 = M.cam_targetbodyid[c], mode = M.cam_mode[c];
      real p[3], R[9];
      local2global(p, R, M.cam_pos + 3*c, M.cam_quat + 4*c, xpos + 3*id, xquat + 4*id, xmat + 9*id, xpos, xmat, MJH_SAMEFRAME_NONE);
      if (mode == 1 || mode == 2) {            // mjCAMLIGHT_TRACK / TRACKCOM: fixed global orientation
        for (int k = 0; k < 9; k++) R[k] = M.cam_mat0[9*c + k];
        if (mode == 1) for (int k = 0; k < 3; k++) p[k] = xpos[3*id + k] + M.cam_pos0[3*c + k];
        else for (int k = 0; k < 3; k++) p[k] = com[3*id + k] + M.cam_poscom0[3*c + k];
      } else if ((mode == 3 || mode == 4) && id1 >= 0) {     // TARGETBODY / TARGETBODYCOM: look at the target
        real t[3], T[9];
        for (int k = 0; k < 3; k++) t[k] = mode == 3 ? (real)xpos[3*id1 + k] : (real)com[3*id1 + k];
        v3_sub(T + 6, p, t);
        v3_normalize(T + 6);
        T[3] = 0; T[4] = 0; T[5] = 1;
        v3_cross(T, T + 3, T + 6);
        v3_normalize(T);
        v3_cross(T + 3, T + 6, T);
        v3_normalize(T + 3);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3*j + i] = T[3*i + j];
      }
      for (int k = 0; k < 3; k++) cpos[3*c + k] = p[k];
      for (int k = 0; k < 9; k++) cmat[9*c + k] = R[k];
    }
    wv_sync();
  }
  if (wv_lane() == 0) {
    if (s.sens_subtreevel && (which & 2)) sens_subtree_vel(M, B, e);
    if (s.sens_rnepost && (which & 4)) sens_rne_post(M, B, e);
    if ((s.sens_energy & 1) && (which & 1)) sens_energy_pos(M, B, e);
    if ((s.sens_energy & 2) && (which & 2)) sens_energy_vel(M, B, e);
  }
  wv_sync();
  ciptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ne = counts[MJH_C_NE], nf = counts[MJH_C_NF];
  Efc P;
  if (nefc) efc_layout(M, B, e, nefc, P);
  rptr out = MJH_G(B, sensordata, e);
  MJH_FOR_LANES(i, s.nsensor) {
    // mjSTAGE_POS = 1, mjSTAGE_VEL = 2, mjSTAGE_ACC = 3 (mjtStage, include/mujoco/mjmodel.h)
    if (!((which >> (M.sensor_needstage[i] - 1)) & 1)) continue;
    const int type = M.sensor_type[i], objtype = M.sensor_objtype[i], objid = M.sensor_objid[i];
    const int reftype = M.sensor_reftype[i], refid = M.sensor_refid[i];
    const int dim = M.sensor_dim[i];
    real v[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (a site rangefinder with every field: 14 values)
    // limit sensors: the first matching constraint row
    int lrow = -1;
    if (type >= MJH_SENS_JOINTLIMITPOS && type <= MJH_SENS_TENDONLIMITFRC) {
      const int want = (type <= MJH_SENS_JOINTLIMITFRC) ? MJH_CNSTR_LIMIT_JOINT : MJH_CNSTR_LIMIT_TENDON;
      for (int j = ne + nf; j < nefc; j++) if (P.type[j] == want && P.id[j] == objid) { lrow = j; break; }
    }
    switch (type) {
    case MJH_SENS_JOINTPOS: v[0] = MJH_F(B, qpos, e)[M.jnt_qposadr[objid]]; break;
    case MJH_SENS_JOINTVEL: v[0] = MJH_F(B, qvel, e)[M.jnt_dofadr[objid]]; break;
    case MJH_SENS_TENDONPOS: v[0] = MJH_F(B, ten_length, e)[objid]; break;
    case MJH_SENS_TENDONVEL: v[0] = MJH_F(B, ten_velocity, e)[objid]; break;
    case MJH_SENS_ACTUATORPOS: v[0] = MJH_F(B, actuator_length, e)[objid]; break;
    case MJH_SENS_ACTUATORVEL: v[0] = MJH_F(B, actuator_velocity, e)[objid]; break;
    case MJH_SENS_ACTUATORFRC: v[0] = MJH_F(B, actuator_force, e)[objid]; break;
    case MJH_SENS_JOINTACTFRC: v[0] = MJH_F(B, qfrc_actuator, e)[M.jnt_dofadr[objid]]; break;
    case MJH_SENS_BALLQUAT: {
      crptr q = MJH_F(B, qpos, e) + M.jnt_qposadr[objid];
      real qq[4] = {q[0], q[1], q[2], q[3]};
      q_normalize(qq);
      for (int k = 0; k < 4; k++) v[k] = qq[k];
    } break;
    case MJH_SENS_BALLANGVEL: { crptr w = MJH_F(B, qvel, e) + M.jnt_dofadr[objid]; v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; } break;
    case MJH_SENS_JOINTLIMITPOS: case MJH_SENS_TENDONLIMITPOS: if (lrow >= 0) v[0] = P.pos[lrow] - P.margin[lrow]; break;
    case MJH_SENS_JOINTLIMITVEL: case MJH_SENS_TENDONLIMITVEL: if (lrow >= 0) v[0] = P.vel[lrow]; break;
    case MJH_SENS_JOINTLIMITFRC: case MJH_SENS_TENDONLIMITFRC: if (lrow >= 0) v[0] = P.force[lrow]; break;
    case MJH_SENS_FRAMEPOS: case MJH_SENS_FRAMEXAXIS: case MJH_SENS_FRAMEYAXIS: case MJH_SENS_FRAMEZAXIS: {
      const SensFrame f = sens_frame(M, B, e, objtype, objid);
      real a[3];
      if (type == MJH_SENS_FRAMEPOS) { a[0] = f.pos[0]; a[1] = f.pos[1]; a[2] = f.pos[2]; }
      else { const int off = type - MJH_SENS_FRAMEXAXIS; a[0] = f.mat[off]; a[1] = f.mat[off + 3]; a[2] = f.mat[off + 6]; }
      if (refid == -1) { v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; }
      else {
        const SensFrame r = sens_frame(M, B, e, reftype, refid);
        if (type == MJH_SENS_FRAMEPOS) { real rv[3]; v3_sub(rv, a, r.pos); m3_multvec(v, r.mat, rv); }
        else m3_multvec(v, r.mat, a);
      }
    } break;
    case MJH_SENS_CAMPROJECTION: {
      // cam_project (engine_sensor.c:281-316): the site in the camera frame, pinhole projection to pixels
      crptr tp = MJH_F(B, site_xpos, e) + 3*objid;
      crptr cp = MJH_G(B, cam_xpos, e) + 3*refid;
      crptr cm = MJH_G(B, cam_xmat, e) + 9*refid;
      const real fx = M.cam_proj[4*refid], fy = M.cam_proj[4*refid + 1], rx = M.cam_proj[4*refid + 2], ry = M.cam_proj[4*refid + 3];
      real rel[3], cpv[3];
      v3_sub(rel, tp, cp);
      // (mju_mulMatTVec: rows of the matrix in order, zero components of the vector skipped)
      for (int k = 0; k < 3; k++) cpv[k] = 0;
      for (int r = 0; r < 3; r++) { const real t = rel[r]; if (t) for (int k = 0; k < 3; k++) cpv[k] += cm[3*r + k]*t; }
      real denom = cpv[2];
      if (fabs(denom) < MJH_MINVAL) denom = denom < 0 ? r_min(denom, -MJH_MINVAL) : r_max(denom, MJH_MINVAL);
      v[0] = -fx * (cpv[0] / denom) + 0.5 * rx;
      v[1] = fy * (cpv[1] / denom) + 0.5 * ry;
    } break;
    case MJH_SENS_FRAMEQUAT: {
      real oq[4];
      sens_quat(M, B, e, objtype, objid, oq);
      if (refid == -1) { for (int k = 0; k < 4; k++) v[k] = oq[k]; }
      else {
        real rq[4], nq[4], res[4];
        sens_quat(M, B, e, reftype, refid, rq);
        q_neg(nq, rq);
        q_mul(res, nq, oq);
        for (int k = 0; k < 4; k++) v[k] = res[k];
      }
    } break;
    case MJH_SENS_FRAMELINVEL: case MJH_SENS_FRAMEANGVEL: {
      real xvel[6];
      object_velocity(M, B, e, objtype, objid, xvel, 0);
      if (refid > -1) {
        const SensFrame f = sens_frame(M, B, e, objtype, objid);
        const SensFrame r = sens_frame(M, B, e, reftype, refid);
        real xref[6], rel[6], rv[3], cr[3];
        object_velocity(M, B, e, reftype, refid, xref, 0);
        for (int k = 0; k < 6; k++) rel[k] = xvel[k] - xref[k];
        v3_sub(rv, f.pos, r.pos);
        v3_cross(cr, rv, xref);
        rel[3] += cr[0]; rel[4] += cr[1]; rel[5] += cr[2];
        m3_multvec(xvel, r.mat, rel);
        m3_multvec(xvel + 3, r.mat, rel + 3);
      }
      const int o = (type == MJH_SENS_FRAMELINVEL) ? 3 : 0;
      v[0] = xvel[o]; v[1] = xvel[o + 1]; v[2] = xvel[o + 2];
    } break;
    case MJH_SENS_FRAMELINACC: case MJH_SENS_FRAMEANGACC: {
      real acc[6];
      object_acceleration(M, B, e, objtype, objid, acc, 0);
      const int o = (type == MJH_SENS_FRAMELINACC) ? 3 : 0;
      v[0] = acc[o]; v[1] = acc[o + 1]; v[2] = acc[o + 2];
    } break;
    case MJH_SENS_SUBTREECOM: { crptr c = MJH_F(B, subtree_com, e) + 3*objid; v[0] = c[0]; v[1] = c[1]; v[2] = c[2]; } break;
    case MJH_SENS_SUBTREELINVEL: { crptr c = MJH_G(B, subtree_linvel, e) + 3*objid; v[0] = c[0]; v[1] = c[1]; v[2] = c[2]; } break;
    case MJH_SENS_SUBTREEANGMOM: { crptr c = MJH_G(B, subtree_angmom, e) + 3*objid; v[0] = c[0]; v[1] = c[1]; v[2] = c[2]; } break;
    case MJH_SENS_CLOCK: v[0] = MJH_F(B, time, e)[0]; break;
    case MJH_SENS_GEOMDIST: case MJH_SENS_GEOMNORMAL: case MJH_SENS_GEOMFROMTO: {
      // (engine_sensor.c:759-812: every geom of the one side against every geom of the other, the closest pair wins)
      const real cut = M.sensor_cutoff[i];
      crptr gx = MJH_F(B, geom_xpos, e);
      crptr gm = MJH_F(B, geom_xmat, e);
      real dist = cut, ft[6] = {0, 0, 0, 0, 0, 0};
      const int n1 = objtype == MJH_OBJ_BODY ? (int)M.body_geomnum[objid] : 1, id1 = objtype == MJH_OBJ_BODY ? (int)M.body_geomadr[objid] : objid;
      const int n2 = reftype == MJH_OBJ_BODY ? (int)M.body_geomnum[refid] : 1, id2 = reftype == MJH_OBJ_BODY ? (int)M.body_geomadr[refid] : refid;
      for (int g1 = id1; g1 < id1 + n1; g1++)
        for (int g2 = id2; g2 < id2 + n2; g2++) {
          real fn[6];
          const real dn = sens_geom_distance(M, gx, gm, g1, g2, cut, fn);
          if (dn < dist) { dist = dn; for (int k = 0; k < 6; k++) ft[k] = fn[k]; }
        }
      if (type == MJH_SENS_GEOMDIST) v[0] = dist;
      else if (type == MJH_SENS_GEOMNORMAL) {
        real nn[3] = {ft[3] - ft[0], ft[4] - ft[1], ft[5] - ft[2]};
        if (nn[0] != 0 || nn[1] != 0 || nn[2] != 0) v3_normalize(nn);
        v[0] = nn[0]; v[1] = nn[1]; v[2] = nn[2];
      } else {
        // (the cutoff of a fromto sensor is its search radius, not a clamp: apply_cutoff returns early, :204-208)
        const int adr = M.sensor_adr[i];
        for (int k = 0; k < 6; k++) out[adr + k] = ft[k];
        continue;
      }
    } break;
    case MJH_SENS_E_POTENTIAL: v[0] = MJH_G(B, energy, e)[0]; break;
    case MJH_SENS_E_KINETIC: v[0] = MJH_G(B, energy, e)[1]; break;
    case MJH_SENS_VELOCIMETER: case MJH_SENS_GYRO: {
      real xvel[6];
      object_velocity(M, B, e, MJH_OBJ_SITE, objid, xvel, 1);
      const int o = (type == MJH_SENS_VELOCIMETER) ? 3 : 0;
      v[0] = xvel[o]; v[1] = xvel[o + 1]; v[2] = xvel[o + 2];
    } break;
    case MJH_SENS_ACCELEROMETER: {
      real acc[6];
      object_acceleration(M, B, e, MJH_OBJ_SITE, objid, acc, 1);
      v[0] = acc[3]; v[1] = acc[4]; v[2] = acc[5];
    } break;
    case MJH_SENS_FORCE: case MJH_SENS_TORQUE: {
      const int body = M.site_bodyid[objid];
      real t[6];
      sp_transform(t, MJH_G(B, cfrc_int, e) + 6*body, 1, MJH_F(B, site_xpos, e) + 3*objid,
                   MJH_F(B, subtree_com, e) + 3*M.body_rootid[body], MJH_F(B, site_xmat, e) + 9*objid, 1);
      const int o = (type == MJH_SENS_FORCE) ? 3 : 0;
      v[0] = t[o]; v[1] = t[o + 1]; v[2] = t[o + 2];
    } break;
    case MJH_SENS_TOUCH: {
      // sum of the normal forces of the contacts of the site's body whose force ray meets the
      // site's zone (engine_sensor.c:980-1025)
      const int body = M.site_bodyid[objid];
      const int ncon = counts[MJH_C_NCON];
      real total = 0;
      for (int c = 0; nefc && c < ncon; c++) {
        if (MJH_CON(B, con_efcadr, e, 1, c)[0] < 0) continue;
        ciptr cg = MJH_CON(B, con_geom, e, 2, c);
        const int cb0 = M.geom_bodyid[cg[0]], cb1 = M.geom_bodyid[cg[1]];
        if (body != cb0 && body != cb1) continue;
        real cf[6];
        contact_force(M, B, e, P, c, cf);
        if (cf[0] <= 0) continue;
        crptr fr = MJH_CON(B, con_frame, e, 9, c);
        real ray[3] = {fr[0]*cf[0], fr[1]*cf[0], fr[2]*cf[0]};
        v3_normalize(ray);
        if (body == cb1) { ray[0] = ray[0]*-1; ray[1] = ray[1]*-1; ray[2] = ray[2]*-1; }
        crptr cp = MJH_CON(B, con_pos, e, 3, c);
        real pnt[3] = {cp[0], cp[1], cp[2]};
        if (ray_hits_zone(M.site_type[objid], MJH_F(B, site_xpos, e) + 3*objid, MJH_F(B, site_xmat, e) + 9*objid,
                          M.site_size + 3*objid, pnt, ray)) total += cf[0];
      }
      v[0] = total;
    } break;
    case MJH_SENS_RANGEFINDER: {
      // site-attached rangefinder: one ray along the site's z axis against every visible geom
      // outside the site's body (mj_ray engine_ray.c:1308-1351, fill_raydata engine_sensor.c:470-520)
      crptr sp = MJH_F(B, site_xpos, e) + 3*objid;
      crptr sm = MJH_F(B, site_xmat, e) + 9*objid;
      real origin[3] = {sp[0], sp[1], sp[2]};
      real rvec[3] = {sm[2], sm[5], sm[8]};
      const int exclude = M.site_bodyid[objid];
      crptr gx = MJH_F(B, geom_xpos, e);
      crptr gm = MJH_F(B, geom_xmat, e);
      real dist = -1;
      int gbest = -1, pbest = 0;
      for (int g = 0; g < s.ngeom; g++) {
        if (M.geom_bodyid[g] == exclude || M.geom_rayskip[g]) continue;
        int pt = 0;
        const real nd = ray_geom_dist(M.geom_type[g], gx + 3*g, gm + 9*g, M.geom_size + 3*g, origin, rvec, &pt);
        if (nd >= 0 && (nd < dist || dist < 0)) { dist = nd; gbest = g; pbest = pt; }
      }
      const int spec = M.sensor_intprm0[i];
      const int hit = dist >= 0;
      int o = 0;
      if (spec & 1) v[o++] = dist;                                             // mjRAYDATA_DIST
      if (spec & 2) { for (int k = 0; k < 3; k++) v[o + k] = hit ? rvec[k] : (real)0; o += 3; }   // DIR
      if (spec & 4) { for (int k = 0; k < 3; k++) v[o + k] = origin[k]; o += 3; }          // ORIGIN
      if (spec & 8) { for (int k = 0; k < 3; k++) v[o + k] = hit ? origin[k] + rvec[k]*dist : (real)0; o += 3; }   // POINT
      if (spec & 16) {                                                         // NORMAL
        real nrm[3] = {0, 0, 0};
        if (hit) ray_geom_normal(M.geom_type[gbest], gx + 3*gbest, gm + 9*gbest, M.geom_size + 3*gbest, origin, rvec, dist, pbest, nrm);
        for (int k = 0; k < 3; k++) v[o + k] = nrm[k];
        o += 3;
      }
      if (spec & 32) v[o++] = hit ? dist : (real)-1;                            // DEPTH
    } break;
    case MJH_SENS_TENDONACTFRC: {
      // sum of the forces of the actuators acting on this tendon (engine_sensor.c:1313-1321)
      crptr af = MJH_F(B, actuator_force, e);
      real frc = 0.0;
      for (int a = 0; a < s.nu; a++)
        if (M.actuator_trntype[a] == MJH_TRN_TENDON && M.actuator_trnid[2*a] == objid) frc += af[a];
      v[0] = frc;
    } break;
    case MJH_SENS_INSIDESITE: {
      // 1 if the object's frame origin lies inside the reference site (mju_insideGeom,
      // engine_util_misc.c:452-496)
      const SensFrame f = sens_frame(M, B, e, objtype, objid);
      crptr pt = f.pos;
      if (objtype == MJH_OBJ_BODY && objid > 0 && M.body_mass[objid] < MJH_MINVAL && M.body_subtreemass[objid] >= MJH_MINVAL)
        pt = MJH_F(B, subtree_com, e) + 3*objid;
      crptr sp = MJH_F(B, site_xpos, e) + 3*refid;
      crptr sm = MJH_F(B, site_xmat, e) + 9*refid;
      auto sz = M.site_size + 3*refid;
      const int st = M.site_type[refid];
      real vec[3], pl[3];
      v3_sub(vec, pt, sp);
      int inside = 0;
      if (st == 2) inside = v3_dot(vec, vec) < sz[0]*sz[0];
      else {
        m3_multvec(pl, sm, vec);
        if (st == 3) {
          const real z = pl[2], zc = r_clip(z, -sz[1], sz[1]);
          const real zd = (z - zc)*(z - zc);
          inside = (pl[0]*pl[0] + pl[1]*pl[1] + zd < sz[0]*sz[0]);
        } else if (st == 4) inside = (pl[0]*pl[0]/(sz[0]*sz[0]) + pl[1]*pl[1]/(sz[1]*sz[1]) + pl[2]*pl[2]/(sz[2]*sz[2]) < 1);
        else if (st == 5) inside = (fabs(pl[2]) < sz[1] && pl[0]*pl[0] + pl[1]*pl[1] < sz[0]*sz[0]);
        else if (st == 6) inside = (fabs(pl[0]) < sz[0] && fabs(pl[1]) < sz[1] && fabs(pl[2]) < sz[2]);
      }
      v[0] = inside ? 1 : 0;
    } break;
    case MJH_SENS_CONTACT: {
      // contact sensor (engine_sensor.c:1027-1150): the contacts that match (object, reference) fill `num` slots of the
      // fields the data specification names -- in contact order, or the nearest / strongest first, or reduced to one net
      // wrench.  The reference collects the matches in an array and partially sorts it; here a lane has no such array: the
      // order (criterion, contact id) is a strict total order, so slot j takes the smallest key above slot j - 1's by a scan
      const int spec = M.sensor_intprm0[i], reduce = M.sensor_intprm1[i];
      const int ncon = counts[MJH_C_NCON];
      const int fsize[7] = {1, 3, 3, 1, 3, 3, 3};          // found, force, torque, dist, pos, normal, tangent
      int size = 0, foff[7];
      for (int q = 0; q < 7; q++) { foff[q] = -1; if (spec & (1 << q)) { foff[q] = size; size += fsize[q]; } }
      const int num = size ? dim/size : 0;
      const int adr0 = M.sensor_adr[i];
      for (int k = 0; k < dim; k++) out[adr0 + k] = 0;
      int nmatch = 0;
      for (int k = 0; k < ncon; k++) if (contact_matches(M, B, e, k, objtype, objid, reftype, refid)) nmatch++;
      const int nslot = num < nmatch ? num : nmatch;
      auto criterion = [&](int k) -> real {
        if (reduce == 1) return MJH_CON(B, con_dist, e, 1, k)[0];
        real ft[6];
        contact_force(M, B, e, P, k, ft);
        return -(ft[0]*ft[0] + ft[1]*ft[1] + ft[2]*ft[2]);
      };
      auto fill = [&](int slot, int k, int flip) {           // copySensorData
        const int a = adr0 + slot*size;
        if (foff[0] >= 0) out[a + foff[0]] = nmatch;
        if (foff[1] >= 0 || foff[2] >= 0) {
          real ft[6];
          contact_force(M, B, e, P, k, ft);
          if (foff[1] >= 0) { out[a + foff[1]] = ft[0]; out[a + foff[1] + 1] = ft[1]; out[a + foff[1] + 2] = flip ? (real)(ft[2]*-1) : ft[2]; }
          if (foff[2] >= 0) { out[a + foff[2]] = ft[3]; out[a + foff[2] + 1] = ft[4]; out[a + foff[2] + 2] = flip ? (real)(ft[5]*-1) : ft[5]; }
        }
        if (foff[3] >= 0) out[a + foff[3]] = MJH_CON(B, con_dist, e, 1, k)[0];
        crptr cp = MJH_CON(B, con_pos, e, 3, k);
        crptr fr = MJH_CON(B, con_frame, e, 9, k);
        if (foff[4] >= 0) for (int x = 0; x < 3; x++) out[a + foff[4] + x] = cp[x];
        if (foff[5] >= 0) for (int x = 0; x < 3; x++) out[a + foff[5] + x] = flip ? (real)(fr[x]*-1) : (real)fr[x];
        if (foff[6] >= 0) for (int x = 0; x < 3; x++) out[a + foff[6] + x] = flip ? (real)(fr[3 + x]*-1) : (real)fr[3 + x];
      };
      if (reduce == 0) {
        int slot = 0;
        for (int k = 0; k < ncon && slot < nslot; k++) {
          const int mt = contact_matches(M, B, e, k, objtype, objid, reftype, refid);
          if (mt) fill(slot++, k, mt < 0);
        }
      } else if (reduce == 1 || reduce == 2) {
        real lastc = 0; int lastk = -1;
        for (int slot = 0; slot < nslot; slot++) {
          int best = -1, bflip = 0; real bc = 0;
          for (int k = 0; k < ncon; k++) {
            const int mt = contact_matches(M, B, e, k, objtype, objid, reftype, refid);
            if (!mt) continue;
            const real c = criterion(k);
            if (lastk >= 0 && (c < lastc || (c == lastc && k <= lastk))) continue;       // already taken
            if (best < 0 || c < bc) { best = k; bc = c; bflip = mt < 0; }
          }
          fill(slot, best, bflip);
          lastc = bc; lastk = best;
        }
      } else if (nmatch > 0 && num > 0) {
        // net force: force-weighted centroid of the contact positions, then the total wrench about it in the world frame
        real point[3] = {0, 0, 0}, total = 0;
        for (int k = 0; k < ncon; k++) {
          const int mt = contact_matches(M, B, e, k, objtype, objid, reftype, refid);
          if (!mt) continue;
          real ft[6];
          contact_force(M, B, e, P, k, ft);
          if (mt < 0) for (int q = 0; q < 6; q++) ft[q] = ft[q]*-1;
          const real w = sqrt(ft[0]*ft[0] + ft[1]*ft[1] + ft[2]*ft[2]);
          crptr cp = MJH_CON(B, con_pos, e, 3, k);
          for (int x = 0; x < 3; x++) point[x] += cp[x]*w;
          total += w;
        }
        const real inv = 1.0/r_max(total, MJH_MINVAL);
        for (int x = 0; x < 3; x++) point[x] = point[x]*inv;
        real force[3] = {0, 0, 0}, torque[3] = {0, 0, 0};
        for (int k = 0; k < ncon; k++) {
          const int mt = contact_matches(M, B, e, k, objtype, objid, reftype, refid);
          if (!mt) continue;
          real ft[6];
          contact_force(M, B, e, P, k, ft);
          if (mt < 0) for (int q = 0; q < 6; q++) ft[q] = ft[q]*-1;
          crptr fr = MJH_CON(B, con_frame, e, 9, k);
          crptr cp = MJH_CON(B, con_pos, e, 3, k);
          real fj[3], tj[3], diff[3], ind[3];
          m3_multvec(fj, fr, ft);
          m3_multvec(tj, fr, ft + 3);
          for (int x = 0; x < 3; x++) { force[x] += fj[x]; torque[x] += tj[x]; }
          v3_sub(diff, cp, point);
          v3_cross(ind, diff, fj);
          for (int x = 0; x < 3; x++) torque[x] += ind[x];
        }
        const int a = adr0;
        if (foff[0] >= 0) out[a + foff[0]] = nmatch;
        if (foff[1] >= 0) for (int x = 0; x < 3; x++) out[a + foff[1] + x] = force[x];
        if (foff[2] >= 0) for (int x = 0; x < 3; x++) out[a + foff[2] + x] = torque[x];
        if (foff[3] >= 0) out[a + foff[3]] = 0;
        if (foff[4] >= 0) for (int x = 0; x < 3; x++) out[a + foff[4] + x] = point[x];
        if (foff[5] >= 0) out[a + foff[5]] = 1;
        if (foff[6] >= 0) out[a + foff[6] + 1] = 1;
      }
      // (no cutoff: apply_cutoff returns early for contact and fromto sensors, engine_sensor.c:204-208)
      continue;
    }
    case MJH_SENS_MAGNETOMETER: {
      real mg[3] = {M.o.magnetic[0], M.o.magnetic[1], M.o.magnetic[2]};
      m3_multvec(v, MJH_F(B, site_xmat, e) + 9*objid, mg);
    } break;
    default: break;
    }
    // apply_cutoff (engine_sensor.c:198-223): datatype 0 real -> both sides, 1 positive -> upper only
    const real cutoff = M.sensor_cutoff[i];
    const int adr = M.sensor_adr[i];
    for (int k = 0; k < dim; k++) {
      real x = v[k];
      if (cutoff > 0) {
        if (M.sensor_datatype[i] == 0) x = r_clip(x, -cutoff, cutoff);
        else if (M.sensor_datatype[i] == 1) x = r_min(cutoff, x);
      }
      out[adr + k] = x;
    }
  }
  wv_sync();
}
