// Flex collisions: one body against one flex (a "job"), in the place the reference's bodyflex order gives the pair
// (mj_collision / mj_collideTree, engine_collision_driver.c:637-730, :996-1150).
//
//   planes of a static body   every vertex sphere against the plane          mj_collidePlaneFlex :2086-2140
//   other geoms               every BVH leaf (active element) whose box the geom's bounds reach: GJK / EPA of the geom
//                             against the element                           mj_collideGeomElem :2372-2515, mjc_ConvexElem
//   filterFlexContacts        more than mjMAXCONPAIR candidates: deepest first, then farthest-point sampling :447-515
//   contactSort               by (geom, vertex / element)                   :410-443
//
// Mapping.  Lanes take vertices / BVH leaves; hits are compacted in order into the job's candidate table (global memory),
// the leaves that pass the box tests are compacted first so that a narrowphase round runs with every lane busy.  The
// midphase walk is replaced by the leaf tests it ends in (the inner boxes contain their leaves); the candidates of a
// geom are generated in the order the walk visits the leaves, which only matters when filterFlexContacts has to break
// a tie.  (included once per SPMD mode by mjh_stages.inc, after mjh_collision.h: no include guard)

#if !MJH_LANE_MODE

#define MJH_FLEX_MAXCON 50          // mjMAXCONPAIR
enum { FC_DIST = 0, FC_POS = 1, FC_NRM = 4, FC_MIND = 7, FC_NREAL = 8 };
enum { FI_GEOM = 0, FI_OBJ = 1, FI_PAIR = 2, FI_KIND = 3, FI_SEL = 4, FI_NINT = 5 };

// (value, index) maximum over the wavefront, smallest index among equals; lanes without a candidate pass index < 0
MJH_DEV int flex_argmax(real v, int idx) {
  real bv = idx >= 0 ? v : (real)-MJH_MAXVAL*MJH_MAXVAL;
  int bi = idx >= 0 ? idx : 0x7fffffff;
  for (int mask = MJH_W/2; mask >= 1; mask >>= 1) {
    const real ov = wv_shfl_xor(bv, mask);
    const int oi = wv_shfl_i(bi, wv_lane() ^ mask);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
  }
  return bi == 0x7fffffff ? -1 : bi;
}


// ------------------------------------------------------------------------------------------------
// triangle elements of a shell flex against spheres, capsules and boxes: closed forms
// (mjraw_SphereTriangle / mjraw_CapsuleTriangle / mjraw_BoxTriangle, engine_collision_primitive.c:566-809).
// A (geom, triangle) pair yields up to 1 / 5 / 11 contacts, each from an independent test; the tests are enumerated
// as "sub-items" in the reference's emission order, one per lane.
// ------------------------------------------------------------------------------------------------
// sphere (centre s, radius rs) against triangle (t1, t2, t3) of radius rt
MJH_DEV int tri_sphere(Hit& h, real margin, V3 s, real rs, V3 t1, V3 t2, V3 t3, real rt) {
  const real rbound = margin + rs + rt;
  const V3 S = s - t1, A = t2 - t1, Bv = t3 - t1;
  V3 N = cross(A, Bv);
  unitize(N);
  const real dstS = dot(N, S);
  if (fabs(dstS) > rbound) return 0;
  const V3 P{S.x + N.x*(-dstS), S.y + N.y*(-dstS), S.z + N.z*(-dstS)};
  V3 V1 = A;
  const real lenA = unitize(V1);
  V3 V2 = cross(N, A);
  unitize(V2);
  const real o[2] = {0, 0}, a[2] = {lenA, 0}, b[2] = {dot(V1, Bv), dot(V2, Bv)}, p[2] = {dot(V1, P), dot(V2, P)};
  auto area_sign = [](const real* p1, const real* p2, const real* p3) {
    const real x = (p1[0] - p3[0])*(p2[1] - p3[1]) - (p2[0] - p3[0])*(p1[1] - p3[1]);
    return (real)((x > 0) - (x < 0));
  };
  const real sign1 = area_sign(p, o, a), sign2 = area_sign(p, a, b), sign3 = area_sign(p, b, o);
  V3 X;
  if (sign1 == sign2 && sign2 == sign3) X = P;
  else {
    // nearest point of the three edges (pointSegment :540), first smallest distance in the order (o,a) (a,b) (b,o)
    auto point_segment = [](real* res, const real* pp, const real* u, const real* v) {
      const real uv[2] = {v[0] - u[0], v[1] - u[1]}, up[2] = {pp[0] - u[0], pp[1] - u[1]};
      const real num = (real)0 + (uv[0]*up[0] + uv[1]*up[1]), den = (real)0 + (uv[0]*uv[0] + uv[1]*uv[1]);
      const real t = num / r_max(MJH_MINVAL, den);
      if (t <= 0) { res[0] = u[0]; res[1] = u[1]; }
      else if (t >= 1) { res[0] = v[0]; res[1] = v[1]; }
      else { res[0] = u[0] + uv[0]*t; res[1] = u[1] + uv[1]*t; }
      return sqrt((res[0] - pp[0])*(res[0] - pp[0]) + (res[1] - pp[1])*(res[1] - pp[1]));
    };
    real x0[2], x1[2], x2[2];
    const real d0 = point_segment(x0, p, o, a), d1 = point_segment(x1, p, a, b), d2 = point_segment(x2, p, b, o);
    const int best = (d0 < d1 && d0 < d2) ? 0 : (d1 < d2 ? 1 : 2);
    const real bx = best == 0 ? x0[0] : (best == 1 ? x1[0] : x2[0]), by = best == 0 ? x0[1] : (best == 1 ? x1[1] : x2[1]);
    X = V3{V1.x*bx, V1.y*bx, V1.z*bx};
    X = V3{X.x + V2.x*by, X.y + V2.y*by, X.z + V2.z*by};
  }
  V3 nrm = X - S;
  const real dst = unitize(nrm);
  if (dst > rbound) return 0;
  h.dist = dst - rs - rt;
  const real scl = rs + h.dist/2;
  h.pos = V3{s.x + nrm.x*scl, s.y + nrm.y*scl, s.z + nrm.z*scl};
  h.nrm = nrm;
  h.tan = V3{0, 0, 0};
  return 1;
}

// sub-item `sub` of capsule (pos, mat, size) against the triangle: 0, 1 the end spheres, 2..4 the triangle's vertices
// against the interior of the axis segment
template <class PM, class PS>
MJH_DEV int tri_capsule_item(Hit& h, int sub, real margin, V3 pos, PM mat, PS size, V3 t1, V3 t2, V3 t3, real rt) {
  const real radius = size[0], len = size[1];
  const V3 axis = mcol(mat, 2);
  const V3 p1{pos.x + axis.x*(-len), pos.y + axis.y*(-len), pos.z + axis.z*(-len)};
  const V3 p2{pos.x + axis.x*len, pos.y + axis.y*len, pos.z + axis.z*len};
  if (sub < 2) return tri_sphere(h, margin, sub == 0 ? p1 : p2, radius, t1, t2, t3, rt);
  const V3 vert = sub == 2 ? t1 : (sub == 3 ? t2 : t3);
  V3 vec = vert - p1;
  const V3 ab = p2 - p1;
  const real t = dot(vec, ab) / (4*len*len);
  if (t <= MJH_MINVAL || t >= 1 - MJH_MINVAL) return 0;
  const V3 closest{p1.x + ab.x*t, p1.y + ab.y*t, p1.z + ab.z*t};
  vec = vert - closest;
  const real dist = unitize(vec);
  if (dist > radius + rt + margin) return 0;
  h.dist = dist - radius - rt;
  h.nrm = vec;
  h.tan = V3{0, 0, 0};
  V3 q = closest + vert;
  const real k = radius - rt;
  q = V3{q.x + vec.x*k, q.y + vec.y*k, q.z + vec.z*k};
  h.pos = V3{q.x*0.5, q.y*0.5, q.z*0.5};
  return 1;
}

// sub-item `sub` of box (pos, mat, size) against the triangle: 0..2 the triangle's vertices against the box faces,
// 3..10 the box corners (radius 0) against the triangle
template <class PM, class PS>
MJH_DEV int tri_box_item(Hit& h, int sub, real margin, V3 pos, PM mat, PS size, V3 t1, V3 t2, V3 t3, real rt) {
  if (sub >= 3) {
    const int i = sub - 3;
    const V3 vec{(i & 1) ? (real)size[0] : -(real)size[0], (i & 2) ? (real)size[1] : -(real)size[1], (i & 4) ? (real)size[2] : -(real)size[2]};
    V3 corner = mmul(mat, vec);
    corner = corner + pos;
    return tri_sphere(h, margin, corner, 0, t1, t2, t3, rt);
  }
  const V3 vert = sub == 0 ? t1 : (sub == 1 ? t2 : t3);
  const V3 diff = vert - pos;
  const real local[3] = {mtrow(mat, 0, diff), mtrow(mat, 1, diff), mtrow(mat, 2, diff)};
  int maxaxis = 0;
  real maxval = fabs(local[0]) - size[0];
  for (int j = 1; j < 3; j++) {
    const real val = fabs(local[j]) - size[j];
    if (val > maxval) { maxval = val; maxaxis = j; }
  }
  if (maxval - rt > margin) return 0;
  for (int j = 0; j < 3; j++) if (fabs(local[j]) > size[j] + margin + rt) return 0;
  const real sgn = (maxaxis == 0 ? local[0] : (maxaxis == 1 ? local[1] : local[2])) > 0 ? 1 : -1;
  const V3 nl{maxaxis == 0 ? sgn : (real)0, maxaxis == 1 ? sgn : (real)0, maxaxis == 2 ? sgn : (real)0};
  h.nrm = mmul(mat, nl);
  h.dist = maxval - rt;
  const real offset = rt + h.dist*0.5;
  h.pos = V3{vert.x + h.nrm.x*(-offset), vert.y + h.nrm.y*(-offset), vert.z + h.nrm.z*(-offset)};
  h.tan = V3{0, 0, 0};
  return 1;
}

// mjc_fixNormal (engine_collision_convex.c:1410) for a contact between a CYLINDER geom (first side) and a triangle of a
// shell flex: on the round wall the normal found by GJK / EPA is replaced by the radial direction
template <class PM, class PS>
MJH_DEV V3 tri_fix_normal_cylinder(V3 normal, V3 cpos, V3 gpos, PM mat, PS size) {
  const V3 dif = cpos - gpos;
  const real pos1[3] = {mtrow(mat, 0, dif), mtrow(mat, 1, dif), mtrow(mat, 2, dif)};
  if (fabs(pos1[2]) > 0.95*size[1]) return normal;
  const real dst1 = fabs(size[1] - fabs(pos1[2]));
  const real dst2 = fabs(size[0] - sqrt((real)0 + (pos1[0]*pos1[0] + pos1[1]*pos1[1])));
  if (dst1 < 0.25*dst2) return normal;
  V3 nrm{pos1[0], pos1[1], 0};
  unitize(nrm);
  return mmul(mat, nrm);
}

// the same for an ELLIPSOID geom: the normal of the ellipsoid's surface at the point closest to the contact position --
// from inside by ray projection along the current normal (mjc_ellipsoidInside, :1306), from outside by Newton's method on
// the multiplier of the diagonal QCQP (mjc_ellipsoidOutside, :1361)
template <class PM, class PS>
MJH_DEV V3 tri_fix_normal_ellipsoid(V3 normal, V3 cpos, V3 gpos, PM mat, PS size) {
  if (size[0] < MJH_MINVAL || size[1] < MJH_MINVAL || size[2] < MJH_MINVAL) return normal;
  const V3 dif = cpos - gpos;
  const real pos[3] = {mtrow(mat, 0, dif), mtrow(mat, 1, dif), mtrow(mat, 2, dif)};
  real nrm[3] = {mtrow(mat, 0, normal), mtrow(mat, 1, normal), mtrow(mat, 2, normal)};
  const real dst1 = pos[0]*pos[0]/(size[0]*size[0]) + pos[1]*pos[1]/(size[1]*size[1]) + pos[2]*pos[2]/(size[2]*size[2]);
  int processed;
  if (dst1 <= 1) {
    const real S2inv[3] = {1/(size[0]*size[0]), 1/(size[1]*size[1]), 1/(size[2]*size[2])};
    const real C = pos[0]*pos[0]*S2inv[0] + pos[1]*pos[1]*S2inv[1] + pos[2]*pos[2]*S2inv[2] - 1;
    if (C > 0) return normal;
    { V3 n{nrm[0], nrm[1], nrm[2]}; unitize(n); nrm[0] = n.x; nrm[1] = n.y; nrm[2] = n.z; }
    processed = 1;
    for (int iter = 0; iter < 30; iter++) {
      const real A = nrm[0]*nrm[0]*S2inv[0] + nrm[1]*nrm[1]*S2inv[1] + nrm[2]*nrm[2]*S2inv[2];
      const real Bq = pos[0]*nrm[0]*S2inv[0] + pos[1]*nrm[1]*S2inv[1] + pos[2]*nrm[2]*S2inv[2];
      const real det = Bq*Bq - A*C;
      if (det < MJH_MINVAL || A < MJH_MINVAL) { processed = iter > 0; break; }
      const real x = (-Bq + sqrt(det))/A;
      if (x < 0) { processed = iter > 0; break; }
      const real pnt[3] = {pos[0] + nrm[0]*x, pos[1] + nrm[1]*x, pos[2] + nrm[2]*x};
      V3 nn{pnt[0]*S2inv[0], pnt[1]*S2inv[1], pnt[2]*S2inv[2]};
      unitize(nn);
      const V3 dd{nrm[0] - nn.x, nrm[1] - nn.y, nrm[2] - nn.z};
      const real change = sqrt(dd.x*dd.x + dd.y*dd.y + dd.z*dd.z);
      nrm[0] = nn.x; nrm[1] = nn.y; nrm[2] = nn.z;
      if (change < 1e-6) break;
    }
  } else {
    const real S2[3] = {size[0]*size[0], size[1]*size[1], size[2]*size[2]};
    const real PS2[3] = {pos[0]*pos[0]*S2[0], pos[1]*pos[1]*S2[1], pos[2]*pos[2]*S2[2]};
    real la = 0;
    for (int iter = 0; iter < 30; iter++) {
      const real R[3] = {1/(S2[0] + la), 1/(S2[1] + la), 1/(S2[2] + la)};
      const real val = PS2[0]*R[0]*R[0] + PS2[1]*R[1]*R[1] + PS2[2]*R[2]*R[2] - 1;
      if (val < 1e-6) break;
      const real deriv = -2*(PS2[0]*R[0]*R[0]*R[0] + PS2[1]*R[1]*R[1]*R[1] + PS2[2]*R[2]*R[2]*R[2]);
      if (deriv > -MJH_MINVAL) break;
      const real delta = -val/deriv;
      if (delta < 1e-6) break;
      la += delta;
    }
    nrm[0] = pos[0]/(S2[0] + la); nrm[1] = pos[1]/(S2[1] + la); nrm[2] = pos[2]/(S2[2] + la);
    processed = 1;
  }
  if (!processed) return normal;
  V3 n{nrm[0], nrm[1], nrm[2]};
  unitize(n);
  return mmul(mat, n);
}

// capsule of a line element: the two vertices' segment with the flex radius (mj_makeCapsule, engine_collision_driver.c:1879;
// the frame through mju_quatZ2Vec / mju_quat2Mat, engine_util_spatial.c)
MJH_DEV void flex_make_capsule(V3 v1, V3 v2, real radius, V3& pos, real* mat, real* size) {
  V3 dif = v1 - v2;
  size[0] = radius;
  size[1] = 0.5*unitize(dif);
  const V3 sum = v1 + v2;
  pos = V3{sum.x*0.5, sum.y*0.5, sum.z*0.5};
  real quat[4] = {1, 0, 0, 0};
  V3 vn = dif;
  if (!(unitize(vn) < MJH_MINVAL)) {
    const V3 z{0, 0, 1};
    V3 axis = cross(z, vn);
    real a = unitize(axis);
    if (fabs(a) < MJH_MINVAL) {
      if (dot(vn, z) < 0) { quat[0] = 0; quat[1] = 1; }
    } else {
      a = r_atan2(a, dot(vn, z));
      const real ax[3] = {axis.x, axis.y, axis.z};
      q_axisangle(quat, ax, a);
    }
  }
  q_tomat(mat, quat);
}

// Position of a leaf pair in the order of mj_collideTree's walk over two bounding volume hierarchies (:1053-1240).  From
// (root, root) a node pair splits the tree that is not at a leaf, or, both inner, the one whose box has the larger
// "surface" (as the reference computes it: entries 3..5 minus entries 0..2 of the node's box), pushes child 0 then child 1
// and pops child 1 first.  Two leaf pairs share their path up to the node pair where they part, and the one in child 1
// comes first: the path's (1 - child) bits, most significant first, are a sort key (at most 52 steps: exact in a double).
// Tree 1: a flex's hierarchy (body_tree 0: flexbvh_* tables, dynamic boxes bb) or a body's (jobbvh_* tables, static
// surfaces); tree 2: a flex's.
// valid (may be null; a flex's hierarchy against ITSELF): 0 when some node pair on the path has node1 > node2 -- mj_collideTree
// drops such a pair when it pops it ("self-collision: avoid repeated pairs", :1065), so the walk never reaches the leaf pair in
// this orientation
template <class BB>
MJH_DEV real flex_walk_key(MREF M, BB bb, int leaf1, int body_tree, int leaf2, int* valid = nullptr) {
  int c1[32], c2[32], d1 = 0, d2 = 0;
  for (int nd = leaf1; nd >= 0 && d1 < 32; nd = body_tree ? (int)M.jobbvh_parent[nd] : (int)M.flexbvh_parent[nd]) c1[d1++] = nd;
  for (int nd = leaf2; nd >= 0 && d2 < 32; nd = M.flexbvh_parent[nd]) c2[d2++] = nd;
  // (c[d - 1] is the root, c[0] the leaf)
  int i1 = d1 - 1, i2 = d2 - 1, len = 0;
  unsigned long long key = 0;
  int ok = 1;
  while ((i1 > 0 || i2 > 0) && len < 52) {
    const int n1 = c1[i1], n2 = c2[i2];
    if (n1 > n2) ok = 0;
    int split1;
    if (i1 == 0) split1 = 0;
    else if (i2 == 0) split1 = 1;
    else {
      real surface1;
      if (body_tree) surface1 = M.jobbvh_surface[n1];
      else {
        const real x1 = bb[6*n1 + 3] - bb[6*n1], y1 = bb[6*n1 + 4] - bb[6*n1 + 1], z1 = bb[6*n1 + 5] - bb[6*n1 + 2];
        surface1 = x1*y1 + y1*z1 + z1*x1;
      }
      const real x2 = bb[6*n2 + 3] - bb[6*n2], y2 = bb[6*n2 + 4] - bb[6*n2 + 1], z2 = bb[6*n2 + 5] - bb[6*n2 + 2];
      const real surface2 = x2*y2 + y2*z2 + z2*x2;
      split1 = surface1 > surface2;
    }
    int child;
    if (split1) { i1--; child = c1[i1] == (body_tree ? (int)M.jobbvh_child[2*n1 + 1] : (int)M.flexbvh_child[2*n1 + 1]); }
    else { i2--; child = c2[i2] == M.flexbvh_child[2*n2 + 1]; }
    key = (key << 1) | (unsigned long long)(1 - child);
    len++;
  }
  if (c1[0] > c2[0] || i1 > 0 || i2 > 0) ok = 0;      // (the leaf pair itself; a path longer than the key holds)
  if (valid) *valid = ok;
  key <<= (52 - len);                            // (paths part before the shorter one ends)
  return (real)(long long)key;
}
// candidates [0, n) re-ordered by the key in their FC_MIND column (ties: candidate order) into the n slots behind them
template <class CP, class IP>
MJH_DEV void flex_order_by_key(CP cand, IP ci, int n) {
  MJH_FOR_LANES(i, n) {
    const real ki = cand[FC_NREAL*i + FC_MIND];
    int rank = 0;
    for (int j = 0; j < n; j++) {
      const real kj = cand[FC_NREAL*j + FC_MIND];
      rank += kj < ki || (kj == ki && j < i);
    }
    for (int q = 0; q < 7; q++) cand[FC_NREAL*(n + rank) + q] = cand[FC_NREAL*i + q];
    for (int q = 0; q < 4; q++) ci[FI_NINT*(n + rank) + q] = ci[FI_NINT*i + q];
  }
  wv_sync();
}

// filterFlexContacts + (optionally) contactSort + emission of the n candidates in (cand, ci); returns the number of
// contacts written from slot `base` on | overflow << 16.  sorted: stable sort by (geom, vertex / element) (body : flex
// jobs, mj_collision :717-724); self-collisions are emitted in candidate order.
template <class CP, class IP>
MJH_DEV int flex_filter_emit(MREF M, BREF B, int e, int f, CP cand, IP ci, int n, int base, int sorted, int f0 = -1) {
  const MJH_CONST_AS DSizes& s = M.s;
  // ---- filterFlexContacts: more candidates than a pair may keep.  The reference works on array positions: it swaps the
  //      chosen contact forward but leaves the `selected` / `min_dist` entries where they are -- reproduced as is.
  int nsel = n;
  if (n > MJH_FLEX_MAXCON) {
    MJH_FOR_LANES(i, n) { cand[FC_NREAL*i + FC_MIND] = MJH_MAXVAL; ci[FI_NINT*i + FI_SEL] = 0; }
    wv_sync();
    int best;
    {
      real bv = 0; int bi = -1;
      MJH_FOR_LANES(i, n) { const real v = -cand[FC_NREAL*i + FC_DIST]; if (bi < 0 || v > bv) { bv = v; bi = i; } }
      best = flex_argmax(bv, bi);
    }
    nsel = 0;
    while (nsel < MJH_FLEX_MAXCON && best >= 0) {
      if (wv_lane() == 0) ci[FI_NINT*best + FI_SEL] = 1;
      wv_sync();
      const V3 bp = ld3(cand + FC_NREAL*best + FC_POS);
      real bv = 0; int bi = -1;
      MJH_FOR_LANES(i, n) {
        if (ci[FI_NINT*i + FI_SEL]) continue;
        const real dx = cand[FC_NREAL*i + FC_POS] - bp.x, dy = cand[FC_NREAL*i + FC_POS + 1] - bp.y, dz = cand[FC_NREAL*i + FC_POS + 2] - bp.z;
        const real d2 = dx*dx + dy*dy + dz*dz;
        real md = cand[FC_NREAL*i + FC_MIND];
        if (d2 < md) { md = d2; cand[FC_NREAL*i + FC_MIND] = md; }
        if (bi < 0 || md > bv) { bv = md; bi = i; }
      }
      int next = flex_argmax(bv, bi);
      wv_sync();
      if (nsel < MJH_FLEX_MAXCON - 1) {
        if (wv_lane() == 0 && best != nsel) {
          for (int q = 0; q < 7; q++) { const real t = cand[FC_NREAL*nsel + q]; cand[FC_NREAL*nsel + q] = cand[FC_NREAL*best + q]; cand[FC_NREAL*best + q] = t; }
          for (int q = 0; q < 4; q++) { const int t = ci[FI_NINT*nsel + q]; ci[FI_NINT*nsel + q] = ci[FI_NINT*best + q]; ci[FI_NINT*best + q] = t; }
        }
        if (next == nsel) next = best;
        wv_sync();
      }
      nsel++;
      best = next;
    }
  }

  // ---- contactSort (stable, by geom then vertex / element) and emission
  int overflow = 0;
  for (int i0 = 0; i0 < nsel; i0 += MJH_W) {
    const int i = i0 + wv_lane();
    if (i >= nsel) continue;
    int rank = i;
    if (sorted) {
      const long long key = ((long long)ci[FI_NINT*i + FI_GEOM] << 32) | (unsigned)ci[FI_NINT*i + FI_OBJ];
      rank = 0;
      for (int j = 0; j < nsel; j++) {
        const long long kj = ((long long)ci[FI_NINT*j + FI_GEOM] << 32) | (unsigned)ci[FI_NINT*j + FI_OBJ];
        if (kj < key || (kj == key && j < i)) rank++;
      }
    }
    const int c = base + rank;
    if (c >= s.nconmax) { overflow = 1; continue; }
    Hit h{cand[FC_NREAL*i + FC_DIST], ld3(cand + FC_NREAL*i + FC_POS), ld3(cand + FC_NREAL*i + FC_NRM), V3{0, 0, 0}};
    store_contact(M, B, e, c, ci[FI_NINT*i + FI_PAIR], h);
    iptr cf = MJH_G(B, con_flex, e) + MJH_CONFLEX*c;
    const int kind = ci[FI_NINT*i + FI_KIND], obj = ci[FI_NINT*i + FI_OBJ];
    // kind 0: geom : vertex, 1: geom : element, 2: element (FI_GEOM) of flex f0 (f if not given) : element (FI_OBJ) of flex f
    cf[0] = f; cf[1] = kind ? obj : -1; cf[2] = kind ? -1 : obj;
    cf[3] = kind == 2 ? (f0 >= 0 ? f0 : f) : -1; cf[4] = kind == 2 ? ci[FI_NINT*i + FI_GEOM] : -1; cf[5] = -1;
  }
  overflow = wv_any(overflow);
  wv_sync();
  return nsel | (overflow << 16);
}

// returns the number of contacts written from slot `base` on | overflow << 16
MJH_DEVN int flex_collide_job(MREF M_, BREF B_, int e_, int seg, int base) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int f = M.colseg[3*seg + 2];
  const int a0 = M.flexjob_adr[seg], a1 = M.flexjob_adr[seg + 1];
  if (a1 == a0) return 0;
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr aabb = MJH_F(B, flexelem_aabb, e);
  rptr cand = MJH_G(B, flexcand, e);
  iptr ci = MJH_G(B, flexcand_i, e);
  iptr surv = ci + FI_NINT*s.nflexcand;
  const real radius = M.flex_radius[f];
  const int vadr = M.flex_vertadr[f], nvert = M.flex_vertnum[f];
  const int eadr = M.flex_elemadr[f];
  int n = 0;
#ifdef MJH_PROFILE
  // (profile builds, slots 53..56: planes | leaf culling | element narrowphase | filter, sort, emission)
  long long ptick = wv_clock();
  auto tick = [&](int slot) { const long long c_ = wv_clock(); if (wv_lane() == 0) MJH_G(B, prof, e)[slot] += (real)(c_ - ptick)*0.01; ptick = c_; };
#else
  auto tick = [](int) {};
#endif

  // ---- planes: every vertex (mj_collidePlaneFlex)
  for (int a = a0; a < a1; a++) {
    const int g = M.flexjob_geom[a];
    if (M.geom_type[g] != MJH_GEOM_PLANE) continue;
    const int p = s.npair + a;
    const real bound = M.pair_margin[p] + radius;           // margin + gap + radius
    const V3 pos = ld3(gx + 3*g);
    const V3 nrm = mcol(gm + 9*g, 2);
    // (four blocks of vertices per trip: a lone wavefront pays the position loads' round trip once instead of four times;
    // the hits are compacted block by block, in vertex order)
    for (int v0 = 0; v0 < nvert; v0 += 4*MJH_W) {
      int hit[4]; real dist[4]; V3 vp[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int v = v0 + u*MJH_W + wv_lane();
        hit[u] = 0; dist[u] = 0; vp[u] = V3{0, 0, 0};
        if (v < nvert) {
          vp[u] = ld3(vx + 3*(vadr + v));
          const V3 dif = vp[u] - pos;
          dist[u] = dif.x*nrm.x + dif.y*nrm.y + dif.z*nrm.z;
          hit[u] = !(dist[u] > bound);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int v = v0 + u*MJH_W + wv_lane();
        const unsigned long long m = wv_ballot(hit[u]);
        if (hit[u]) {
          const int c = n + wv_rank_lt(m);
          const real cd = dist[u] - radius;
          const real scl = -cd*0.5 - radius;
          cand[FC_NREAL*c + FC_DIST] = cd;
          cand[FC_NREAL*c + FC_POS] = vp[u].x + nrm.x*scl; cand[FC_NREAL*c + FC_POS + 1] = vp[u].y + nrm.y*scl; cand[FC_NREAL*c + FC_POS + 2] = vp[u].z + nrm.z*scl;
          cand[FC_NREAL*c + FC_NRM] = nrm.x; cand[FC_NREAL*c + FC_NRM + 1] = nrm.y; cand[FC_NREAL*c + FC_NRM + 2] = nrm.z;
          ci[FI_NINT*c + FI_GEOM] = g; ci[FI_NINT*c + FI_OBJ] = v; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 0;
        }
        n += __builtin_popcountll(m);
      }
    }
  }

  tick(53);
  // ---- other geoms: BVH leaves in reach, then GJK / EPA per surviving (geom, element)
  const int l0 = M.flex_leafadr[f], l1 = M.flex_leafadr[f + 1];
  for (int a = a0; a < a1; a++) {
    const int g = M.flexjob_geom[a];
    if (M.geom_type[g] == MJH_GEOM_PLANE) continue;
    const int p = s.npair + a;
    const real mg = M.pair_margin[p];                       // margin + gap
    const real sbound = M.geom_rbound[g] + mg;
    const int gbody = M.geom_bodyid[g];
    const real ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero3[3] = {0, 0, 0};
    int nsurv = 0;
    // (two blocks of leaves per trip, for the same reason; survivors compacted in leaf order)
    for (int k0 = l0; k0 < l1; k0 += 2*MJH_W) {
      int ok[2], el[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int k = k0 + u*MJH_W + wv_lane();
        ok[u] = 0; el[u] = -1;
        if (k < l1) {
          el[u] = M.flexleaf_elem[k];
          crptr bx = aabb + 6*el[u];
          const real sx = gx[3*g], sy = gx[3*g + 1], sz = gx[3*g + 2];
          // filterSphereBox (:236-244)
          ok[u] = !(sx + sbound < bx[0] - bx[3] || sy + sbound < bx[1] - bx[4] || sz + sbound < bx[2] - bx[5] ||
                    sx - sbound > bx[0] + bx[3] || sy - sbound > bx[1] + bx[4] || sz - sbound > bx[2] + bx[5]);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (ok[u]) ok[u] = bp_obb<0>(M.geom_aabb + 6*g, aabb + 6*el[u], gx + 3*g, gm + 9*g, zero3, ident, mg);
        // an element with a vertex on the geom's own body is skipped (mj_collideGeomElem :2387-2394)
        if (ok[u]) for (int i = 0; i < 4; i++) { const int v = M.flexelem_vert[4*el[u] + i]; if (v >= 0 && M.flexvert_bodyid[v] == gbody) ok[u] = 0; }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const unsigned long long m = wv_ballot(ok[u]);
        if (ok[u]) surv[nsurv + wv_rank_lt(m)] = el[u];
        nsurv += __builtin_popcountll(m);
      }
    }
    wv_sync();
    tick(54);
    const int nsub = M.flexjob_nsub[a];
    const int gtype = M.geom_type[g];
    if (nsub && M.flex_dim[f] == 1) {
      // line elements against a sphere / capsule: the raw capsule colliders on the capsule made of the element's two
      // vertices (mj_collideGeomElem :2401-2425); up to two contacts per element, compacted in (leaf, contact) order
      const V3 gp = ld3(gx + 3*g);
      for (int r0 = 0; r0 < nsurv; r0 += MJH_W) {
        const int r = r0 + wv_lane();
        int got = 0, el = -1;
        Hit ha{0, V3{0, 0, 0}, V3{0, 0, 0}, V3{0, 0, 0}}, hb = ha;
        if (r < nsurv) {
          el = surv[r];
          V3 cpos; real cmat[9], csize[2];
          flex_make_capsule(ld3(vx + 3*M.flexelem_vert[4*el]), ld3(vx + 3*M.flexelem_vert[4*el + 1]), radius, cpos, cmat, csize);
          if (gtype == MJH_GEOM_SPHERE) got = hit_sphere_capsule(ha, mg, gp, gm + 9*g, M.geom_size[3*g], cpos, cmat, csize);
          else got = hit_capsule_capsule(ha, hb, mg, gp, gm + 9*g, M.geom_size + 3*g, cpos, cmat, csize);
        }
        const int before = wv_exscan_i(got);
        const int total = wv_sum_i(got);
        for (int q = 0; q < got; q++) {
          const int c = n + before + q;
          const Hit& h = q ? hb : ha;
          cand[FC_NREAL*c + FC_DIST] = h.dist;
          cand[FC_NREAL*c + FC_POS] = h.pos.x; cand[FC_NREAL*c + FC_POS + 1] = h.pos.y; cand[FC_NREAL*c + FC_POS + 2] = h.pos.z;
          cand[FC_NREAL*c + FC_NRM] = h.nrm.x; cand[FC_NREAL*c + FC_NRM + 1] = h.nrm.y; cand[FC_NREAL*c + FC_NRM + 2] = h.nrm.z;
          ci[FI_NINT*c + FI_GEOM] = g; ci[FI_NINT*c + FI_OBJ] = el - eadr; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 1;
        }
        n += total;
      }
    } else if (nsub) {
      // triangles against a sphere / capsule / box: the closed-form tests, one (triangle, sub-test) per lane, hits compacted
      // in (leaf, sub-test) order -- the order in which the reference emits them
      const V3 gp = ld3(gx + 3*g);
      const int nitems = nsurv*nsub;
      for (int i0 = 0; i0 < nitems; i0 += MJH_W) {
        const int it = i0 + wv_lane();
        int got = 0, el = -1;
        Hit h{0, V3{0, 0, 0}, V3{0, 0, 0}, V3{0, 0, 0}};
        if (it < nitems) {
          el = surv[it / nsub];
          const int sub = it % nsub;
          const V3 t1 = ld3(vx + 3*M.flexelem_vert[4*el]), t2 = ld3(vx + 3*M.flexelem_vert[4*el + 1]), t3 = ld3(vx + 3*M.flexelem_vert[4*el + 2]);
          if (gtype == MJH_GEOM_SPHERE) got = tri_sphere(h, mg, gp, M.geom_size[3*g], t1, t2, t3, radius);
          else if (gtype == MJH_GEOM_CAPSULE) got = tri_capsule_item(h, sub, mg, gp, gm + 9*g, M.geom_size + 3*g, t1, t2, t3, radius);
          else got = tri_box_item(h, sub, mg, gp, gm + 9*g, M.geom_size + 3*g, t1, t2, t3, radius);
        }
        const unsigned long long m = wv_ballot(got > 0);
        if (got > 0) {
          const int c = n + wv_rank_lt(m);
          cand[FC_NREAL*c + FC_DIST] = h.dist;
          cand[FC_NREAL*c + FC_POS] = h.pos.x; cand[FC_NREAL*c + FC_POS + 1] = h.pos.y; cand[FC_NREAL*c + FC_POS + 2] = h.pos.z;
          cand[FC_NREAL*c + FC_NRM] = h.nrm.x; cand[FC_NREAL*c + FC_NRM + 1] = h.nrm.y; cand[FC_NREAL*c + FC_NRM + 2] = h.nrm.z;
          ci[FI_NINT*c + FI_GEOM] = g; ci[FI_NINT*c + FI_OBJ] = el - eadr; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 1;
        }
        n += __builtin_popcountll(m);
      }
    } else {
      const int fixnormal = M.flex_dim[f] == 2 && (gtype == MJH_GEOM_CYLINDER || gtype == MJH_GEOM_ELLIPSOID);
      for (int r0 = 0; r0 < nsurv; r0 += MJH_W) {
        const int r = r0 + wv_lane();
        const int el = r < nsurv ? surv[r] : -1;
        const int got = ccd_geom_elem_pair(M, B, e, el >= 0 ? g : -1, el, mg);
        const unsigned long long m = wv_ballot(got > 0);
        if (got > 0) {
          const crptr rec = ccd_out_records(M, B, e);
          const int c = n + wv_rank_lt(m);
          for (int q = 0; q < 7; q++) cand[FC_NREAL*c + q] = rec[q];
          if (fixnormal) {
            const V3 nn = gtype == MJH_GEOM_ELLIPSOID ? tri_fix_normal_ellipsoid(ld3(rec + 4), ld3(rec + 1), ld3(gx + 3*g), gm + 9*g, M.geom_size + 3*g)
                                                      : tri_fix_normal_cylinder(ld3(rec + 4), ld3(rec + 1), ld3(gx + 3*g), gm + 9*g, M.geom_size + 3*g);
            cand[FC_NREAL*c + FC_NRM] = nn.x; cand[FC_NREAL*c + FC_NRM + 1] = nn.y; cand[FC_NREAL*c + FC_NRM + 2] = nn.z;
          }
          ci[FI_NINT*c + FI_GEOM] = g; ci[FI_NINT*c + FI_OBJ] = el - eadr; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 1;
        }
        n += __builtin_popcountll(m);
      }
    }
    wv_sync();
    tick(55);
  }
  wv_sync();
  if (n == 0) return 0;

  int r;
  if (s.njobbvh && M.flexjob_leaf[a0] >= 0 && n > MJH_FLEX_MAXCON) {
    // a body with several geoms, more contacts than the pair may keep: the thinning works on the candidates in the order
    // the reference's walk over the body's and the flex's hierarchies emits them (plane contacts come before the walk)
    crptr bb = MJH_G(B, flexbvh_aabb, e);
    const int eadr0 = M.flex_elemadr[f];
    MJH_FOR_LANES(i, n) {
      real key = -1;
      if (ci[FI_NINT*i + FI_KIND] != 0) {
        int leaf1 = -1;
        for (int a = a0; a < a1; a++) if (M.flexjob_geom[a] == ci[FI_NINT*i + FI_GEOM]) leaf1 = M.flexjob_leaf[a];
        key = flex_walk_key(M, bb, leaf1, 1, M.flexelem_bvhleaf[eadr0 + ci[FI_NINT*i + FI_OBJ]]);
      }
      cand[FC_NREAL*i + FC_MIND] = key;
    }
    wv_sync();
    flex_order_by_key(cand, ci, n);
    r = flex_filter_emit(M, B, e, f, cand + FC_NREAL*n, ci + FI_NINT*n, n, base, 1);
  } else
  r = flex_filter_emit(M, B, e, f, cand, ci, n, base, 1);
  tick(56);
  return r;
}

// ------------------------------------------------------------------------------------------------
// Self-collisions of flex number `sidx` of the self-colliding flexes (mj_collision, engine_collision_driver.c:834-881;
// mj_collideFlexSAP :2315, mj_SAP :1439, mj_collideElems :2518).
//
// The reference sorts the float-rounded end points of the active elements' boxes along the longest axis of the flex's root
// box (stable sort: equal values keep the order min_0 max_0 min_1 max_1 ...), sweeps, and sends every pair of boxes that are
// open at the same time and overlap on the other two axes to mj_collideElems -- in the order the sweep meets them: by the
// sort position of the later box's lower end, then of the earlier box's.  Here every pair (i < j) of active elements is
// tested directly ("would the sweep report it?" is a comparison of four float keys), the surviving pairs -- few: elements
// that share a vertex body are skipped -- go through GJK / EPA one per lane, and their contacts are put in the sweep's
// order afterwards (the candidates' keys are unique), before filterFlexContacts looks at them.  mode 2 (no midphase /
// selfcollide = narrow): all pairs e1 < e2 in lexicographic order.  mode 3 (selfcollide = bvh, and auto on solid flexes): the
// flex's hierarchy against itself -- every pair of leaves whose boxes overlap, in the orientation and order of mj_collideTree's walk.
// ------------------------------------------------------------------------------------------------
MJH_DEVN int flex_self_collide(MREF M_, BREF B_, int e_, int sidx, int base) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int f = M.flexself_flex[sidx], p = M.flexself_pair[sidx], mode = M.flexself_mode[sidx];
  const int a0 = M.flexact_adr[f], nact = M.flexact_adr[f + 1] - a0;
  if (nact < 2) return 0;
  crptr aabb = MJH_F(B, flexelem_aabb, e);
  rptr cand = MJH_G(B, flexcand, e);
  iptr ci = MJH_G(B, flexcand_i, e);
  iptr warn = MJH_F(B, warning, e);
  const int half = s.nflexcand/2;
  iptr surv = ci + FI_NINT*s.nflexcand;                       // (a << 16 | b: active-element indices, a the box the sweep opens first)
  const int eadr = M.flex_elemadr[f];
  int axis = 0;
  if (mode == 1) {
    crptr root = MJH_G(B, flexbvh_aabb, e) + 6*M.flexbvh_adr[f];
    axis = (root[3] > root[4] && root[3] > root[5]) ? 0 : (root[4] > root[5] ? 1 : 2);
  }
  // ---- pairs of elements the reference's narrowphase would see
  int nsurv = 0, toomany = 0;
  for (int i = 0; i + 1 < nact; i++) {
    const int ei = M.flexact_elem[a0 + i];
    crptr bi = aabb + 6*ei;
    const real ilo[3] = {bi[0] - bi[3], bi[1] - bi[4], bi[2] - bi[5]}, ihi[3] = {bi[0] + bi[3], bi[1] + bi[4], bi[2] + bi[5]};
    const float iminf = (float)(axis == 0 ? ilo[0] : (axis == 1 ? ilo[1] : ilo[2])), imaxf = (float)(axis == 0 ? ihi[0] : (axis == 1 ? ihi[1] : ihi[2]));
    int vb[4];
    for (int q = 0; q < 4; q++) { const int v = M.flexelem_vert[4*ei + q]; vb[q] = v >= 0 ? (int)M.flexvert_bodyid[v] : -1; }
    for (int j0 = i + 1; j0 < nact; j0 += MJH_W) {
      const int j = j0 + wv_lane();
      int ok = 0, first_i = 1;
      if (j < nact) {
        const int ej = M.flexact_elem[a0 + j];
        crptr bj = aabb + 6*ej;
        const real jlo[3] = {bj[0] - bj[3], bj[1] - bj[4], bj[2] - bj[5]}, jhi[3] = {bj[0] + bj[3], bj[1] + bj[4], bj[2] + bj[5]};
        ok = 1;
        if (mode == 3) ok = M.flexelem_bvhleaf[ei] >= 0 && M.flexelem_bvhleaf[ej] >= 0;      // (the walk reaches the hierarchy's leaves only)
        if (mode == 1) {
          // the sweep: keys (float value, position in the unsorted end-point array: 2 id for a lower, 2 id + 1 for an upper end)
          const float jminf = (float)(axis == 0 ? jlo[0] : (axis == 1 ? jlo[1] : jlo[2])), jmaxf = (float)(axis == 0 ? jhi[0] : (axis == 1 ? jhi[1] : jhi[2]));
          first_i = iminf <= jminf;                                   // (i < j: equal values keep i's lower end first)
          ok = first_i ? (jminf < imaxf) : (iminf <= jmaxf);         // the later box opens before the earlier one closes
          for (int q = 0; q < 3 && ok; q++) {
            if (q == axis) continue;
            if (ilo[q] > jhi[q] || jlo[q] > ihi[q]) ok = 0;
          }
        }
        // filterBox with margin 0 (mj_collideElems :2531)
        for (int q = 0; q < 3 && ok; q++) if (ihi[q] + 0 < jlo[q] || jhi[q] + 0 < ilo[q]) ok = 0;
        // elements with vertices on the same body (:2536-2548)
        if (ok) for (int q = 0; q < 4; q++) {
          const int v = M.flexelem_vert[4*ej + q];
          const int b = v >= 0 ? (int)M.flexvert_bodyid[v] : -1;
          if (b >= 0 && (b == vb[0] || b == vb[1] || b == vb[2] || b == vb[3])) ok = 0;
        }
        if (ok && mode == 3) {
          // the orientation in which mj_collideTree's walk reaches the pair (node1 <= node2 all the way down), if any
          crptr bb3 = MJH_G(B, flexbvh_aabb, e);
          int v1 = 0, v2 = 0;
          flex_walk_key(M, bb3, M.flexelem_bvhleaf[ei], 0, M.flexelem_bvhleaf[ej], &v1);
          if (!v1) flex_walk_key(M, bb3, M.flexelem_bvhleaf[ej], 0, M.flexelem_bvhleaf[ei], &v2);
          first_i = v1;
          ok = v1 || v2;
        }
      }
      const unsigned long long m = wv_ballot(ok);
      if (m) {
        const int at = nsurv + wv_rank_lt(m);
        if (ok) { if (at < half) surv[at] = first_i ? ((i << 16) | j) : ((j << 16) | i); else toomany = 1; }
        nsurv += __builtin_popcountll(m);
      }
    }
  }
  if (wv_any(toomany)) {
    // more overlapping pairs than the table holds: not a state the reference would flag -- stop the environment instead
    if (wv_lane() == 0) warn[MJH_WARN_UNSUPPORTED]++;
    nsurv = half;
  }
  wv_sync();
  if (nsurv == 0) return 0;
  int n = 0;
  if (M.flex_dim[f] == 1) {
    // ---- line elements: mjraw_CapsuleCapsule on the two elements' capsules, margin 0 (mj_collideElems :2555-2566)
    crptr vx = MJH_F(B, flexvert_xpos, e);
    const real radius = M.flex_radius[f];
    for (int r0 = 0; r0 < nsurv; r0 += MJH_W) {
      const int r = r0 + wv_lane();
      const int pr = r < nsurv ? surv[r] : -1;
      const int ia = pr >= 0 ? (pr >> 16) : -1, ib = pr >= 0 ? (pr & 0xffff) : -1;
      int got = 0;
      Hit ha{0, V3{0, 0, 0}, V3{0, 0, 0}, V3{0, 0, 0}}, hb = ha;
      if (pr >= 0) {
        const int e1 = M.flexact_elem[a0 + ia], e2 = M.flexact_elem[a0 + ib];
        V3 p1, p2; real m1[9], m2[9], s1[2], s2[2];
        flex_make_capsule(ld3(vx + 3*M.flexelem_vert[4*e1]), ld3(vx + 3*M.flexelem_vert[4*e1 + 1]), radius, p1, m1, s1);
        flex_make_capsule(ld3(vx + 3*M.flexelem_vert[4*e2]), ld3(vx + 3*M.flexelem_vert[4*e2 + 1]), radius, p2, m2, s2);
        got = hit_capsule_capsule(ha, hb, 0, p1, m1, s1, p2, m2, s2);
      }
      const int before = wv_exscan_i(got);
      const int total = wv_sum_i(got);
      for (int q = 0; q < got; q++) {
        const int c = n + before + q;
        if (c >= half) continue;
        const Hit& h = q ? hb : ha;
        cand[FC_NREAL*c + FC_DIST] = h.dist;
        cand[FC_NREAL*c + FC_POS] = h.pos.x; cand[FC_NREAL*c + FC_POS + 1] = h.pos.y; cand[FC_NREAL*c + FC_POS + 2] = h.pos.z;
        cand[FC_NREAL*c + FC_NRM] = h.nrm.x; cand[FC_NREAL*c + FC_NRM + 1] = h.nrm.y; cand[FC_NREAL*c + FC_NRM + 2] = h.nrm.z;
        // (the second contact of a pair follows the first: FI_SEL breaks the tie of the ordering below)
        ci[FI_NINT*c + FI_GEOM] = ia; ci[FI_NINT*c + FI_OBJ] = ib; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 2; ci[FI_NINT*c + FI_SEL] = q;
      }
      n += total;
    }
    wv_sync();
  } else
  // ---- GJK / EPA per surviving pair (mjc_ConvexElem, one contact at most)
  for (int r0 = 0; r0 < nsurv; r0 += MJH_W) {
    const int r = r0 + wv_lane();
    const int pr = r < nsurv ? surv[r] : -1;
    const int ia = pr >= 0 ? (pr >> 16) : -1, ib = pr >= 0 ? (pr & 0xffff) : -1;
    const int got = ccd_elem_elem_pair(M, B, e, pr >= 0 ? (int)M.flexact_elem[a0 + ia] : -1, pr >= 0 ? (int)M.flexact_elem[a0 + ib] : -1, 0);
    const unsigned long long m = wv_ballot(got > 0);
    if (got > 0) {
      const crptr rec = ccd_out_records(M, B, e);
      const int c = n + wv_rank_lt(m);
      if (c < half) {
        for (int q = 0; q < 7; q++) cand[FC_NREAL*c + q] = rec[q];
        ci[FI_NINT*c + FI_GEOM] = ia; ci[FI_NINT*c + FI_OBJ] = ib; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 2; ci[FI_NINT*c + FI_SEL] = 0;
      }
    }
    n += __builtin_popcountll(m);
    wv_sync();
  }
  if (n > half) n = half;
  if (n == 0) return 0;
  if (mode == 3) {
    // ---- the order of mj_collideTree's walk over the hierarchy against itself (flex_walk_key), candidates of one pair together
    crptr bb3 = MJH_G(B, flexbvh_aabb, e);
    MJH_FOR_LANES(i, n)
      cand[FC_NREAL*i + FC_MIND] = flex_walk_key(M, bb3, M.flexelem_bvhleaf[M.flexact_elem[a0 + ci[FI_NINT*i + FI_GEOM]]], 0,
                                                 M.flexelem_bvhleaf[M.flexact_elem[a0 + ci[FI_NINT*i + FI_OBJ]]]);
    wv_sync();
    flex_order_by_key(cand, ci, n);
    rptr cand3 = cand + FC_NREAL*n;
    iptr ci3 = ci + FI_NINT*n;
    MJH_FOR_LANES(i, n) {
      ci3[FI_NINT*i + FI_GEOM] = M.flexact_elem[a0 + ci3[FI_NINT*i + FI_GEOM]] - eadr;
      ci3[FI_NINT*i + FI_OBJ] = M.flexact_elem[a0 + ci3[FI_NINT*i + FI_OBJ]] - eadr;
      ci3[FI_NINT*i + FI_PAIR] = p; ci3[FI_NINT*i + FI_KIND] = 2;
    }
    wv_sync();
    return flex_filter_emit(M, B, e, f, cand3, ci3, n, base, 0);
  }
  // ---- the sweep's order: by the key of the later box's lower end, then of the earlier box's (mode 2: by (e1, e2))
  rptr cand2 = cand + FC_NREAL*half;
  iptr ci2 = ci + FI_NINT*half;
  auto lower_key = [&](int ia, float* val) {
    crptr bx = aabb + 6*M.flexact_elem[a0 + ia];
    *val = mode == 1 ? (float)(axis == 0 ? bx[0] - bx[3] : (axis == 1 ? bx[1] - bx[4] : bx[2] - bx[5])) : 0.0f;
  };
  for (int i0 = 0; i0 < n; i0 += MJH_W) {
    const int i = i0 + wv_lane();
    if (i >= n) continue;
    const int ia = ci[FI_NINT*i + FI_GEOM], ib = ci[FI_NINT*i + FI_OBJ];
    float ka, kb;
    lower_key(ia, &ka); lower_key(ib, &kb);
    int rank = 0;
    for (int j = 0; j < n; j++) {
      const int ja = ci[FI_NINT*j + FI_GEOM], jb = ci[FI_NINT*j + FI_OBJ];
      float la, lb;
      lower_key(ja, &la); lower_key(jb, &lb);
      int before;
      if (ja == ia && jb == ib) before = ci[FI_NINT*j + FI_SEL] < ci[FI_NINT*i + FI_SEL];
      else if (mode == 1) {
        if (lb != kb) before = lb < kb; else if (jb != ib) before = jb < ib;
        else if (la != ka) before = la < ka; else before = ja < ia;
      } else {
        before = ja != ia ? ja < ia : jb < ib;
      }
      rank += before;
    }
    for (int q = 0; q < 7; q++) cand2[FC_NREAL*rank + q] = cand[FC_NREAL*i + q];
    // (the element ids the contact record carries: local to the flex)
    ci2[FI_NINT*rank + FI_GEOM] = M.flexact_elem[a0 + ia] - eadr; ci2[FI_NINT*rank + FI_OBJ] = M.flexact_elem[a0 + ib] - eadr;
    ci2[FI_NINT*rank + FI_PAIR] = p; ci2[FI_NINT*rank + FI_KIND] = 2;
  }
  wv_sync();
  return flex_filter_emit(M, B, e, f, cand2, ci2, n, base, 0);
}

// ------------------------------------------------------------------------------------------------
// Collisions between two different flexes, pair k of the model's list (mj_collideElems for every pair of elements whose boxes
// overlap; see the model build for what the order of the reference's tree walk would matter for).  Survivors as in the
// self-collision pass -- element a of the first flex in the upper half of the entry -- then GJK / EPA or capsule : capsule per
// survivor, contacts sorted by (element, element) (contactSort after mj_collideTree; without midphase the double loop emits
// them in that order already).
// ------------------------------------------------------------------------------------------------
MJH_DEVN int flex_pair_collide(MREF M_, BREF B_, int e_, int k, int base) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int f1 = M.flexff_flex[2*k], f2 = M.flexff_flex[2*k + 1], p = M.flexff_pair[k];
  const int a1 = M.flex_elemadr[f1], n1 = M.flex_elemnum[f1], a2 = M.flex_elemadr[f2], n2 = M.flex_elemnum[f2];
  if (n1 == 0 || n2 == 0) return 0;
  crptr aabb = MJH_F(B, flexelem_aabb, e);
  rptr cand = MJH_G(B, flexcand, e);
  iptr ci = MJH_G(B, flexcand_i, e);
  iptr warn = MJH_F(B, warning, e);
  const int half = s.nflexcand/2;
  iptr surv = ci + FI_NINT*s.nflexcand;
  const real mg = M.pair_margin[p];
  // ---- the second flex's bounding box (exact union of its element boxes): an element of the first flex that misses it
  //      (by more than the margin) misses every element
  real lo2[3] = {MJH_MAXVAL, MJH_MAXVAL, MJH_MAXVAL}, hi2[3] = {-MJH_MAXVAL, -MJH_MAXVAL, -MJH_MAXVAL};
  MJH_FOR_LANES(j, n2) {
    crptr bj = aabb + 6*(a2 + j);
    for (int q = 0; q < 3; q++) { lo2[q] = r_min(lo2[q], bj[q] - bj[q + 3]); hi2[q] = r_max(hi2[q], bj[q] + bj[q + 3]); }
  }
  // (per-lane partial boxes through the candidate table, every lane reduces all of them)
  for (int q = 0; q < 3; q++) { cand[6*wv_lane() + q] = lo2[q]; cand[6*wv_lane() + 3 + q] = hi2[q]; }
  wv_sync();
  for (int l = 0; l < MJH_W; l++)
    for (int q = 0; q < 3; q++) { lo2[q] = r_min(lo2[q], cand[6*l + q]); hi2[q] = r_max(hi2[q], cand[6*l + 3 + q]); }
  wv_sync();
  // (with midphase only the elements held by the hierarchy's leaves take part: the walk reaches no others)
  const int tree = M.flexff_mode[k] == 1;
  int nsurv = 0, toomany = 0;
  for (int i = 0; i < n1; i++) {
    const int ei = a1 + i;
    if (tree && M.flexelem_bvhleaf[ei] < 0) continue;
    crptr bi = aabb + 6*ei;
    const real ilo[3] = {bi[0] - bi[3], bi[1] - bi[4], bi[2] - bi[5]}, ihi[3] = {bi[0] + bi[3], bi[1] + bi[4], bi[2] + bi[5]};
    if (ihi[0] + mg < lo2[0] || hi2[0] + mg < ilo[0] || ihi[1] + mg < lo2[1] || hi2[1] + mg < ilo[1] || ihi[2] + mg < lo2[2] || hi2[2] + mg < ilo[2]) continue;
    int vb[4];
    for (int q = 0; q < 4; q++) { const int v = M.flexelem_vert[4*ei + q]; vb[q] = v >= 0 ? (int)M.flexvert_bodyid[v] : -1; }
    for (int j0 = 0; j0 < n2; j0 += MJH_W) {
      const int j = j0 + wv_lane();
      int ok = 0;
      if (j < n2) {
        const int ej = a2 + j;
        crptr bj = aabb + 6*ej;
        ok = !tree || M.flexelem_bvhleaf[ej] >= 0;
        // filterBox (engine_collision_driver.c:230: centre / half-size boxes, apart by more than the margin along an axis)
        for (int q = 0; q < 3 && ok; q++) {
          const real jlo = bj[q] - bj[q + 3], jhi = bj[q] + bj[q + 3];
          if (ihi[q] + mg < jlo || jhi + mg < ilo[q]) ok = 0;
        }
        if (ok) for (int q = 0; q < 4; q++) {
          const int v = M.flexelem_vert[4*ej + q];
          const int b = v >= 0 ? (int)M.flexvert_bodyid[v] : -1;
          if (b >= 0 && (b == vb[0] || b == vb[1] || b == vb[2] || b == vb[3])) ok = 0;
        }
      }
      const unsigned long long m = wv_ballot(ok);
      if (m) {
        const int at = nsurv + wv_rank_lt(m);
        if (ok) { if (at < half) surv[at] = (i << 16) | j; else toomany = 1; }
        nsurv += __builtin_popcountll(m);
      }
    }
  }
  if (wv_any(toomany)) { if (wv_lane() == 0) warn[MJH_WARN_UNSUPPORTED]++; nsurv = half; }
  wv_sync();
  if (nsurv == 0) return 0;
  int n = 0;
  if (M.flex_dim[f1] == 1) {
    crptr vx = MJH_F(B, flexvert_xpos, e);
    for (int r0 = 0; r0 < nsurv; r0 += MJH_W) {
      const int r = r0 + wv_lane();
      const int pr = r < nsurv ? surv[r] : -1;
      const int ia = pr >= 0 ? (pr >> 16) : -1, ib = pr >= 0 ? (pr & 0xffff) : -1;
      int got = 0;
      Hit ha{0, V3{0, 0, 0}, V3{0, 0, 0}, V3{0, 0, 0}}, hb = ha;
      if (pr >= 0) {
        const int e1 = a1 + ia, e2 = a2 + ib;
        V3 p1, p2; real m1[9], m2[9], s1[2], s2[2];
        flex_make_capsule(ld3(vx + 3*M.flexelem_vert[4*e1]), ld3(vx + 3*M.flexelem_vert[4*e1 + 1]), M.flex_radius[f1], p1, m1, s1);
        flex_make_capsule(ld3(vx + 3*M.flexelem_vert[4*e2]), ld3(vx + 3*M.flexelem_vert[4*e2 + 1]), M.flex_radius[f2], p2, m2, s2);
        got = hit_capsule_capsule(ha, hb, mg, p1, m1, s1, p2, m2, s2);
      }
      const int before = wv_exscan_i(got);
      const int total = wv_sum_i(got);
      for (int q = 0; q < got; q++) {
        const int c = n + before + q;
        if (c >= half) continue;
        const Hit& h = q ? hb : ha;
        cand[FC_NREAL*c + FC_DIST] = h.dist;
        cand[FC_NREAL*c + FC_POS] = h.pos.x; cand[FC_NREAL*c + FC_POS + 1] = h.pos.y; cand[FC_NREAL*c + FC_POS + 2] = h.pos.z;
        cand[FC_NREAL*c + FC_NRM] = h.nrm.x; cand[FC_NREAL*c + FC_NRM + 1] = h.nrm.y; cand[FC_NREAL*c + FC_NRM + 2] = h.nrm.z;
        ci[FI_NINT*c + FI_GEOM] = ia; ci[FI_NINT*c + FI_OBJ] = ib; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 2; ci[FI_NINT*c + FI_SEL] = q;
      }
      n += total;
    }
    wv_sync();
  } else
  for (int r0 = 0; r0 < nsurv; r0 += MJH_W) {
    const int r = r0 + wv_lane();
    const int pr = r < nsurv ? surv[r] : -1;
    const int ia = pr >= 0 ? (pr >> 16) : -1, ib = pr >= 0 ? (pr & 0xffff) : -1;
    const int got = ccd_elem_elem_pair(M, B, e, pr >= 0 ? a1 + ia : -1, pr >= 0 ? a2 + ib : -1, mg);
    const unsigned long long m = wv_ballot(got > 0);
    if (got > 0) {
      const crptr rec = ccd_out_records(M, B, e);
      const int c = n + wv_rank_lt(m);
      if (c < half) {
        for (int q = 0; q < 7; q++) cand[FC_NREAL*c + q] = rec[q];
        ci[FI_NINT*c + FI_GEOM] = ia; ci[FI_NINT*c + FI_OBJ] = ib; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 2; ci[FI_NINT*c + FI_SEL] = 0;
      }
    }
    n += __builtin_popcountll(m);
    wv_sync();
  }
  if (n > half) n = half;
  if (n == 0) return 0;
  // (survivors, and with them the candidates, are in (element, element) order: the order of the double loop without midphase)
  if (!tree) return flex_filter_emit(M, B, e, f2, cand, ci, n, base, 0, f1);
  // ---- the order of mj_collideTree's walk (flex_walk_key), then thinning, then the sort by (element, element)
  crptr bb = MJH_G(B, flexbvh_aabb, e);
  MJH_FOR_LANES(i, n)
    cand[FC_NREAL*i + FC_MIND] = flex_walk_key(M, bb, M.flexelem_bvhleaf[a1 + ci[FI_NINT*i + FI_GEOM]], 0, M.flexelem_bvhleaf[a2 + ci[FI_NINT*i + FI_OBJ]]);
  wv_sync();
  flex_order_by_key(cand, ci, n);
  rptr cand2 = cand + FC_NREAL*n;
  iptr ci2 = ci + FI_NINT*n;
  return flex_filter_emit(M, B, e, f2, cand2, ci2, n, base, 1, f1);
}

#endif   // !MJH_LANE_MODE
