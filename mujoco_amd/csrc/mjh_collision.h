// Collision stage of the batched step: candidate geom pairs -> contacts.
//
// The reference runs sweep-and-prune over bodies, then a BVH midphase, then the narrowphase
// (engine_collision_driver.c:595-886).  Broad- and midphase only CULL pairs conservatively, so the
// contact list equals "narrowphase over every geom pair that survives the model-constant filters"
// in the reference's order: ascending body-pair signature, then (g1,g2) within a body pair, then
// the collider's own emission order.  The host precomputes that ordered static
// pair list with its mixed contact parameters (mjh_model_build.h).
//
// Mapping onto a lane group (the MJH_W lanes that step one environment):
//   * the pair list is walked in chunks of MJH_W pairs, one pair per lane, through the bounding-sphere
//     filter; the survivors are compacted into the lanes in pair order and take ONE round of the
//     closed-form point colliders (plane/sphere/capsule: at most two contacts, kept in registers -- no
//     contact array in private memory); more than MJH_W survivors: chunk by chunk;
//   * pairs of the chunk that need a multi-contact collider (box, cylinder) are taken ONE AT A
//     TIME BY THE WHOLE GROUP: separating axes, box corners, polygon vertices and contact
//     candidates are spread over lanes, selections are ballots / in-order scans that reproduce
//     the reference's first-wins tie rules, and every surviving lane holds exactly one contact;
//   * slots: a contact's index is (contacts of earlier rounds) + (exclusive scan of the per-pair
//     counts) + its rank inside the pair, so lanes write their contacts straight into the
//     (LDS-resident) contact slots in reference order.
// All arithmetic that reaches a contact is evaluated in the reference's association
// (engine_collision_primitive.c, engine_collision_box.c; cited per routine), which is what makes
// contact counts and geometry agree bit for bit.
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)

struct V3 { real x, y, z; };
MJH_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MJH_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MJH_DEV V3 operator*(V3 a, real s) { return V3{a.x*s, a.y*s, a.z*s}; }
MJH_DEV real dot(V3 a, V3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
MJH_DEV V3 cross(V3 a, V3 b) { return V3{a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x}; }
template <class P> MJH_DEV V3 ld3(P p) { return V3{p[0], p[1], p[2]}; }
template <class P> MJH_DEV void st3(P p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
// column c of a row-major 3x3 (c may differ per lane: the matrix lives in memory)
template <class P> MJH_DEV V3 mcol(P m, int c) { return V3{m[c], m[3 + c], m[6 + c]}; }
// M v and M' v in the reference's association (mji_mulMatVec3 / mji_mulMatTVec3)
template <class P> MJH_DEV V3 mmul(P m, V3 v) {
  return V3{m[0]*v.x + m[1]*v.y + m[2]*v.z, m[3]*v.x + m[4]*v.y + m[5]*v.z, m[6]*v.x + m[7]*v.y + m[8]*v.z};
}
template <class P> MJH_DEV real mtrow(P m, int i, V3 v) { return m[i]*v.x + m[3 + i]*v.y + m[6 + i]*v.z; }
// unit vector + previous length (mju_normalize3: tiny vectors become the x axis)
MJH_DEV real unitize(V3& v) {
  const real n = sqrt(v.x*v.x + v.y*v.y + v.z*v.z);
  if (n < MJH_MINVAL) { v = V3{1, 0, 0}; }
  else { const real inv = 1/n; v = V3{v.x*inv, v.y*inv, v.z*inv}; }
  return n;
}
// component k of v / v with component k replaced (k may differ per lane; selects, not memory)
MJH_DEV real comp(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
MJH_DEV V3 with_comp(V3 v, int k, real s) { return V3{k == 0 ? s : v.x, k == 1 ? s : v.y, k == 2 ? s : v.z}; }
// vector with components (i, i1, i2) = (a, b, c) for a cyclic index triple
MJH_DEV V3 from_cyclic(int i, real a, real b, real c) {
  return i == 0 ? V3{a, b, c} : (i == 1 ? V3{c, a, b} : V3{b, c, a});
}

// mask of the lanes below lane k of a group (k < 64; only the low 32 lanes carry collider work)
MJH_DEV unsigned lanes_below(int k) { return k >= 32 ? ~0u : (1u << k) - 1; }

// one contact candidate held by a lane: distance, position, normal, optional tangent hint
struct Hit { real dist; V3 pos, nrm, tan; };

// ---- point colliders (one pair per lane) -----------------------------------------------------------

// plane : sphere (mjraw_PlaneSphere, engine_collision_primitive.c:28)
MJH_DEV int hit_plane_sphere(Hit& h, real margin, V3 ppos, V3 pnrm, V3 centre, real radius) {
  const real height = dot(centre - ppos, pnrm);
  if (height > margin + radius) return 0;
  h.dist = height - radius;
  h.nrm = pnrm;
  h.pos = centre + pnrm*(-h.dist/2 - radius);
  h.tan = V3{0, 0, 0};
  return 1;
}

// sphere : sphere with the degenerate-centre fallback onto the geoms' z axes
// (mjraw_SphereSphere, engine_collision_primitive.c:262)
template <class P1, class P2>
MJH_DEV int hit_sphere_sphere(Hit& h, real margin, V3 c1, P1 mat1, real r1, V3 c2, P2 mat2, real r2) {
  const V3 back = c1 - c2;
  const real d2 = dot(back, back);
  const real reach = margin + r1 + r2;
  if (d2 > reach*reach) return 0;
  h.dist = sqrt(d2) - r1 - r2;
  V3 n = c2 - c1;
  if (unitize(n) < MJH_MINVAL) {
    n = cross(mcol(mat1, 2), mcol(mat2, 2));
    unitize(n);
  }
  h.nrm = n;
  h.pos = n*(r1 + h.dist/2) + c1;
  h.tan = V3{0, 0, 0};
  return 1;
}

// plane : capsule = the two end spheres, tangent along the capsule (mjc_PlaneCapsule, :66)
template <class P2, class S2>
MJH_DEV int hit_plane_capsule(Hit& a, Hit& b, real margin, V3 ppos, V3 pnrm, V3 c2, P2 mat2, S2 size2) {
  const V3 axis = mcol(mat2, 2);
  const V3 half = axis*(real)size2[1];
  Hit lo;
  const int n1 = hit_plane_sphere(a, margin, ppos, pnrm, c2 + half, size2[0]);
  const int n2 = hit_plane_sphere(lo, margin, ppos, pnrm, c2 - half, size2[0]);
  a.tan = axis;
  lo.tan = axis;
  if (n1) b = lo; else a = lo;
  return n1 + n2;
}

// sphere : capsule = sphere against the nearest point of the segment (mjraw_SphereCapsule, :313)
template <class P1, class P2, class S2>
MJH_DEV int hit_sphere_capsule(Hit& h, real margin, V3 c1, P1 mat1, real r1, V3 c2, P2 mat2, S2 size2) {
  const V3 axis = mcol(mat2, 2);
  const real half = size2[1];
  const real t = r_clip(dot(axis, c1 - c2), -half, half);
  return hit_sphere_sphere(h, margin, c1, mat1, r1, axis*t + c2, mat2, size2[0]);
}

// capsule : capsule (mjraw_CapsuleCapsule, :425): closest points of the two segments; parallel
// axes fall back to four end-point tests that stop at two contacts
template <class P1, class S1, class P2, class S2>
MJH_DEV int hit_capsule_capsule(Hit& a, Hit& b, real margin, V3 c1, P1 mat1, S1 size1, V3 c2, P2 mat2, S2 size2) {
  const V3 u1 = mcol(mat1, 2)*(real)size1[1], u2 = mcol(mat2, 2)*(real)size2[1];
  const V3 back = c1 - c2;
  const real ma = dot(u1, u1), mb = -dot(u1, u2), mc = dot(u2, u2);
  const real u = -dot(u1, back), v = dot(u2, back);
  const real det = ma*mc - mb*mb;
  const real r1 = size1[0], r2 = size2[0];
  if (fabs(det) >= MJH_MINVAL) {
    real x1 = (mc*u - mb*v) / det;
    real x2 = (ma*v - mb*u) / det;
    if (x1 > 1) { x1 = 1; x2 = (v - mb) / mc; }
    else if (x1 < -1) { x1 = -1; x2 = (v + mb) / mc; }
    if (x2 > 1) { x2 = 1; x1 = r_clip((u - mb) / ma, -1, 1); }
    else if (x2 < -1) { x2 = -1; x1 = r_clip((u + mb) / ma, -1, 1); }
    return hit_sphere_sphere(a, margin, u1*x1 + c1, mat1, r1, u2*x2 + c2, mat2, r2);
  }
  // parallel: end points of capsule 1 against segment 2, then of capsule 2 against segment 1
  int n = 0;
  Hit t;
  for (int trial = 0; trial < 4 && n < 2; trial++) {
    V3 q1, q2;
    if (trial < 2) {
      const real sgn = trial == 0 ? 1 : -1;
      q1 = trial == 0 ? c1 + u1 : c1 - u1;
      q2 = u2*r_clip((v - sgn*mb) / mc, -1, 1) + c2;
    } else {
      const real sgn = trial == 2 ? 1 : -1;
      q2 = trial == 2 ? c2 + u2 : c2 - u2;
      q1 = u1*r_clip((u - sgn*mb) / ma, -1, 1) + c1;
    }
    if (hit_sphere_sphere(t, margin, q1, mat1, r1, q2, mat2, r2)) {
      if (n == 0) a = t; else b = t;
      n++;
    }
  }
  return n;
}

// sphere : box (mjraw_SphereBox, engine_collision_box.c:35): clamp the centre into the box; a centre
// inside the box is pushed out through the nearest face
template <class P2, class S2>
MJH_DEVN int hit_sphere_box(Hit& h, real margin, V3 c1, real r1, V3 c2, P2 mat2, S2 size2) {
  const V3 off = c1 - c2;
  const V3 local{mtrow(mat2, 0, off), mtrow(mat2, 1, off), mtrow(mat2, 2, off)};
  const V3 ext = ld3(size2);
  const V3 nearest{r_clip(local.x, -ext.x, ext.x), r_clip(local.y, -ext.y, ext.y), r_clip(local.z, -ext.z, ext.z)};
  V3 dir = nearest - local;
  real gap = unitize(dir);
  if (gap - r1 > margin) return 0;
  V3 lpos;
  if (gap <= MJH_MINVAL) {
    // face k = 2*axis + side, the first strictly closer one wins
    real best = (ext.x + ext.y + ext.z)*2;
    int face = 0;
    for (int k = 0; k < 6; k++) {
      const real d = fabs(((k & 1) ? 1 : -1)*comp(ext, k >> 1) - comp(local, k >> 1));
      if (best > d) { best = d; face = k; }
    }
    const V3 out = with_comp(V3{0, 0, 0}, face >> 1, (face & 1) ? -1 : 1);
    lpos = local + out*((r1 - best)/2);
    h.nrm = mmul(mat2, out);
    gap = -best;
  } else {
    const V3 deep = local + dir*r1;
    lpos = (V3{0, 0, 0} + nearest*0.5) + deep*0.5;
    h.nrm = mmul(mat2, dir);
  }
  h.pos = mmul(mat2, lpos) + c2;
  h.dist = gap - r1;
  h.tan = V3{0, 0, 0};
  return 1;
}

// sphere : cylinder (mjc_SphereCylinder, engine_collision_primitive.c:345): the sphere meets the
// barrel (a sphere on the axis), a cap (a plane) or the rim (a point)
template <class P1, class P2, class S2>
MJH_DEVN int hit_sphere_cylinder(Hit& h, real margin, V3 c1, P1 mat1, real r1, V3 c2, P2 mat2, S2 size2) {
  const real radius = size2[0], half = size2[1];
  const V3 axis = mcol(mat2, 2);
  const V3 off = c1 - c2;
  const real along = dot(axis, off);
  const V3 onaxis = axis*along;
  const V3 radial = off - onaxis;
  const real rad2 = dot(radial, radial);
  int barrel = fabs(along) < half, cap = rad2 < radius*radius;
  if (barrel && cap) {
    // inside both slabs: the shallower penetration decides
    if (half - fabs(along) < radius - sqrt(rad2)) barrel = 0; else cap = 0;
  }
  if (barrel) return hit_sphere_sphere(h, margin, c1, mat1, r1, onaxis + c2, mat2, radius);
  const real side = along > 0 ? half : -half;
  if (cap) {
    // plane through the cap centre; its normal is the axis turned outwards
    const V3 capn = along > 0 ? axis : V3{-mat2[2], -mat2[5], -mat2[8]};
    const int n = hit_plane_sphere(h, margin, V3{c2.x + axis.x*side, c2.y + axis.y*side, c2.z + axis.z*side}, capn, c1, r1);
    if (n) h.nrm = h.nrm*(real)-1;
    return n;
  }
  const V3 rim = (axis*side + radial*(radius / sqrt(rad2))) + c2;
  return hit_sphere_sphere(h, margin, c1, mat1, r1, rim, mat2, (real)0);
}

#if !MJH_LANE_MODE
// ---- group-cooperative colliders (one pair per lane group) ----------------------------------------
// Each returns the number of contacts of the pair (uniform over the group); a lane with has != 0
// holds contact number `rank` of the pair in h.

// plane : box (mjc_PlaneBox, engine_collision_primitive.c:210): lane k < 8 tests corner k; the
// first four corners below the plane (in corner order) are the contacts
template <class P2, class S2>
MJH_DEVN int coop_plane_box(Hit& h, int& has, int& rank, real margin, V3 ppos, V3 pnrm, V3 c2, P2 mat2, S2 size2) {
  const int k = wv_lane();
  const real height = dot(c2 - ppos, pnrm);
  const V3 corner = mmul(mat2, V3{(k & 1) ? (real)size2[0] : -(real)size2[0],
                                  (k & 2) ? (real)size2[1] : -(real)size2[1],
                                  (k & 4) ? (real)size2[2] : -(real)size2[2]});
  const real rel = dot(pnrm, corner);
  const int below = k < 8 && !(height + rel > margin || rel > 0);
  const unsigned m = (unsigned)wv_ballot(below);
  rank = __builtin_popcount(m & lanes_below(k));
  has = below && rank < 4;
  h.dist = height + rel;
  h.nrm = pnrm;
  h.pos = (corner + c2) + pnrm*(-h.dist/2);
  h.tan = V3{0, 0, 0};
  const int total = __builtin_popcount(m);
  return total < 4 ? total : 4;
}

// plane : cylinder (mjc_PlaneCylinder, engine_collision_primitive.c:101): the lowest point of the
// near rim, the matching point of the far rim, and two points of the near rim 120 degrees away;
// lane k < 4 evaluates candidate k, nothing is reported unless the lowest point is in reach
template <class P2, class S2>
MJH_DEVN int coop_plane_cylinder(Hit& h, int& has, int& rank, real margin, V3 ppos, V3 pnrm, V3 c2, P2 mat2, S2 size2) {
  const int k = wv_lane();
  V3 axis = mcol(mat2, 2);
  real tilt = dot(pnrm, axis);
  if (tilt > 0) { axis = axis*(real)-1; tilt = -tilt; }      // axis points into the plane
  const real height = dot(c2 - ppos, pnrm);
  // direction from the axis to the rim point nearest the plane, scaled to the radius
  V3 down = axis*tilt - pnrm;
  const real len2 = dot(down, down);
  if (len2 >= MJH_MINVAL*MJH_MINVAL) down = down*((real)size2[0]/sqrt(len2));
  else down = V3{mat2[0]*size2[0], mat2[3]*size2[0], mat2[6]*size2[0]};     // upright: any rim point
  const real drop = dot(down, pnrm);
  const V3 halfax = axis*(real)size2[1];
  const real reach = tilt*size2[1];
  const real d_near = height + reach + drop, d_far = height - reach + drop, d_side = height + reach + (-drop*0.5);
  const int in0 = d_near <= margin, in1 = in0 && d_far <= margin, in2 = in0 && d_side <= margin;
  V3 lateral = cross(down, halfax);
  unitize(lateral);
  lateral = lateral*(size2[0]*sqrt(3.0)/2);
  if (k == 0) { h.dist = d_near; h.pos = ((c2 + down) + halfax) + pnrm*(-d_near*0.5); }
  else if (k == 1) { h.dist = d_far; h.pos = ((c2 + down) - halfax) + pnrm*(-d_far*0.5); }
  else {
    const V3 base = k == 2 ? c2 + lateral : c2 - lateral;
    h.dist = d_side;
    h.pos = ((base + halfax) + down*(real)-0.5) + pnrm*(-d_side*0.5);
  }
  h.nrm = pnrm;
  h.tan = V3{0, 0, 0};
  has = (k == 0 && in0) || (k == 1 && in1) || ((k == 2 || k == 3) && in2);
  rank = k < 2 ? k : (in1 ? k : k - 1);
  return in0 + in1 + 2*in2;
}

// capsule : box (mjraw_CapsuleBox, engine_collision_box.c:114-590): two sphere-box contacts, one at
// the point of the capsule's segment closest to the box, one further along the segment where it
// leaves the box's shadow.
//   1. closest feature: lanes 0 and 1 drop the segment's two end points onto the box (a face is
//      closest when at most one coordinate had to be clamped); lanes 2..13 take one box edge each
//      (axis j = (lane - 2)/4, the four edges parallel to it in corner order) and solve the
//      clamped segment-segment problem against it.
//   2. the winner is picked by an in-order scan of the lanes' squared distances: end points by
//      strict <, edges must win by more than MINVAL (the reference's guard against an axis that is
//      numerically parallel to the box).
//   3. where the second sphere goes is a case analysis on the winner (box corner / box edge / box
//      face) that every lane evaluates on the broadcast winner data.
//   4. lanes 0 and 1 each collide one sphere with the box.
template <class P1, class S1, class P2, class S2>
MJH_DEVN int coop_capsule_box(Hit& h, int& has, int& rank, real margin, V3 c1, P1 mat1, S1 size1, V3 c2, P2 mat2, S2 size2) {
  const int lane = wv_lane();
  has = 0; rank = 0;
  const real radius = size1[0], half = size1[1];
  const V3 ext = ld3(size2);
  const V3 off = c1 - c2;
  const V3 cen{mtrow(mat2, 0, off), mtrow(mat2, 1, off), mtrow(mat2, 2, off)};     // capsule centre, box frame
  const V3 dirw = mcol(mat1, 2);
  const V3 dir{mtrow(mat2, 0, dirw), mtrow(mat2, 1, dirw), mtrow(mat2, 2, dirw)};  // capsule axis, box frame
  const V3 seg = dir*half;
  const int octant = (seg.x > 0 ? 1 : 0) | (seg.y > 0 ? 2 : 0) | (seg.z > 0 ? 4 : 0);
  const real far = margin + 2*(radius + half + ext.x + ext.y + ext.z);

  // ---- 1. my candidate: squared distance `cand`, segment parameter tpar in [-1, 1], and for edges the
  // edge parameter epar, the clamp pattern and the corner the edge starts from
  real cand = 0, tpar = 0, epar = 0;
  int ok = 0, face = -1, pattern = 0, corner = 0, eaxis = 0;
  if (lane < 2) {
    const real sgn = lane == 0 ? -1 : 1;
    const V3 tip = cen + seg*sgn;
    int outside = 0;
    V3 clamped = tip;
    for (int j = 0; j < 3; j++) {
      const real t = comp(tip, j), lim = comp(ext, j);
      if (t < -lim) { outside++; face = j; clamped = with_comp(clamped, j, -lim); }
      else if (t > lim) { outside++; face = j; clamped = with_comp(clamped, j, lim); }
    }
    const V3 gap = clamped - tip;
    cand = dot(gap, gap);
    tpar = sgn;
    ok = outside <= 1;
  } else if (lane < 14) {
    const int q = lane - 2;
    const int j = q >> 2, r = q & 3;
    const int i = (r & ((1 << j) - 1)) | ((r >> j) << (j + 1));        // r with a zero inserted at bit j
    const V3 start = with_comp(V3{(i & 1) ? ext.x : -ext.x, (i & 2) ? ext.y : -ext.y, (i & 4) ? ext.z : -ext.z}, j, (real)0);
    const V3 rel = start - cen;
    const real lim = comp(ext, j);
    const real ma = lim*lim, mb = -lim*comp(seg, j), mc = half*half;
    const real u = -lim*comp(rel, j), v = dot(seg, rel);
    const real det = ma*mc - mb*mb;
    if (!(fabs(det) < MJH_MINVAL)) {
      const real idet = 1/det;
      real xe = (mc*u - mb*v)*idet;       // along the edge
      real xs = (ma*v - mb*u)*idet;       // along the segment
      int ende = 1, ends = 1;             // 0 / 1 / 2: lower end, interior, upper end
      if (xe > 1) { xe = 1; ende = 2; xs = (v - mb)*(1/mc); }
      else if (xe < -1) { xe = -1; ende = 0; xs = (v + mb)*(1/mc); }
      if (xs > 1 || xs < -1) {
        const int up = xs > 1;
        xs = up ? 1 : -1;
        ends = up ? 2 : 0;
        xe = (up ? (u - mb) : (u + mb))*(1/ma);
        if (xe > 1) { xe = 1; ende = 2; } else if (xe < -1) { xe = -1; ende = 0; }
      }
      V3 gap = rel + seg*(-xs);
      gap = with_comp(gap, j, comp(gap, j) + lim*xe);
      cand = dot(gap, gap);
      tpar = xs; epar = xe;
      pattern = ende*3 + ends;
      corner = i + (1 << j)*(pattern/6);
      eaxis = j;
      ok = 1;
    }
  }

  // ---- 2. in-order scan
  const unsigned okmask = (unsigned)wv_ballot(ok);
  real best = far;
  int win = -1;
  for (int a = 0; a < 14; a++) {
    const real d = wv_bcast(cand, a);
    if (((okmask >> a) & 1) && d < (a < 2 ? best : best - MJH_MINVAL)) { best = d; win = a; }
  }
  if (win < 0) return 0;
  tpar = wv_bcast(tpar, win);
  epar = wv_bcast(epar, win);
  face = wv_bcast_i(face, win);
  pattern = wv_bcast_i(pattern, win);
  corner = wv_bcast_i(corner, win);
  eaxis = wv_bcast_i(eaxis, win);

  // ---- 3. offset of the second sphere along the segment (none: stays below -3)
  real second = -4;
  auto shorten = [&](real lim) { if (lim < second) second = lim; };
  if (win >= 2 && pattern/3 != 1) {
    // a box corner is closest.  The octants of the capsule axis and of the corner tell whether the
    // capsule points at / away from the corner (nothing more to find) or runs along an edge or a face
    int relo = octant ^ corner;
    if (relo != 0 && relo != 7) {
      const int single = (relo & (relo - 1)) == 0;
      const real sense = single ? 1 : -1;
      if (!single) relo = 7 - relo;
      const real toend = single ? 1 - tpar : 1 + tpar, tostart = single ? 1 + tpar : 1 - tpar;
      const int ax = relo == 1 ? 0 : (relo == 2 ? 1 : 2);
      const int ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (comp(dir, ax)*comp(dir, ax) > 0.5) {          // along the box edge
        second = toend;
        shorten(2*comp(ext, ax) / fabs(comp(seg, ax)));
        second *= sense;
      } else {                                          // across a face
        second = tostart;
        shorten(2*comp(ext, ax1) / fabs(comp(seg, ax1)));
        shorten(2*comp(ext, ax2) / fabs(comp(seg, ax2)));
        second *= -sense;
      }
    }
  } else if (win >= 2) {
    // the interior of a box edge is closest: a T configuration has no second point, a crossing does
    const int relo = (octant ^ corner) & (7 - (1 << eaxis));
    if (relo == 1 || relo == 2 || relo == 4) {
      const int ax = eaxis;
      int ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (fabs(comp(dir, ax1)) > fabs(comp(dir, ax2))) ax1 = ax2;     // the face the capsule is flatter to
      ax2 = 3 - ax - ax1;
      const int fwd = (relo >> ax2) & 1;
      const real sense = fwd ? 1 : -1;
      second = fwd ? 1 - tpar : 1 + tpar;
      shorten(2*comp(ext, ax2) / fabs(comp(seg, ax2)));
      const real room = (((octant >> ax) & 1) == fwd) ? 1 - epar : 1 + epar;
      shorten(comp(ext, ax)*room / fabs(comp(seg, ax)));
      second *= sense;
    }
  } else if (face >= 0) {
    // an end point is closest to a face: walk to where the segment leaves the face's outline
    const real sense = win == 0 ? 1 : -1;
    second = 2;
    const V3 from = cen + seg*(-sense);
    for (int i = 0; i < 3; i++) {
      if (i == face) continue;
      const real hi = (comp(ext, i) - comp(from, i)) / comp(seg, i) * sense;
      if (hi > 0) shorten(hi);
      const real lo = (-comp(ext, i) - comp(from, i)) / comp(seg, i) * sense;
      if (lo > 0) shorten(lo);
    }
    second *= sense;
  }

  // ---- 4. the spheres
  const int two = second > -3;
  int n = 0;
  Hit hs;
  if (lane == 0 || (lane == 1 && two)) {
    const V3 local = cen + seg*(lane == 0 ? tpar : second + tpar);
    n = hit_sphere_box(hs, margin, mmul(mat2, local) + c2, radius, c2, mat2, size2);
  }
  const int n0 = wv_bcast_i(n, 0), n1 = wv_bcast_i(n, 1);
  h = hs;
  has = lane < 2 && n > 0;
  rank = lane == 0 ? 0 : n0;
  return n0 + n1;
}

// box : box (mjc_BoxBox, engine_collision_box.c:697-1066).
//   1. separating axes: lane a < 15 evaluates axis a (3 + 3 face normals, 9 edge cross products)
//      straight from the two rotation matrices in memory; any positive separation ends the test;
//      the winner is picked by an in-order scan of the lanes' values (first-wins ties, the edge
//      bias and the edge -> face substitution of the reference).
//   2. edge-edge: the (up to four) sign-ambiguous vertex choices go to lanes 0..3, the closest
//      pair of points wins, one contact.
//   3. face: the incident face's 4 corners sit on lanes 0..3 and are clipped against the four
//      side planes of the reference face, one vertex per lane (Sutherland-Hodgman: a lane emits
//      its vertex and/or the crossing towards its successor, an exclusive scan gives the new
//      positions); duplicates are dropped by an in-order ballot loop; every accepted vertex is a
//      contact (<= 8).
#define MJH_BB_SEPEPS 1e-13
#define MJH_BB_PAREPS 1e-16
#define MJH_BB_SGNEPS 1e-9
#define MJH_BB_DUPEPS 1e-14
#define MJH_BB_EDGEBIAS 1e-6

template <class P1, class S1, class P2, class S2>
MJH_DEVN int coop_box_box(Hit& h, int& has, int& rank, real margin, V3 c1, P1 mat1, S1 size1, V3 c2, P2 mat2, S2 size2) {
  const int lane = wv_lane();
  has = 0; rank = 0;
  h.tan = V3{0, 0, 0};
  // entry (r, c) of mat1' * mat2, evaluated from memory so r, c may differ per lane
  auto R = [&](int r, int c) -> real { return mat1[r]*mat2[c] + mat1[3 + r]*mat2[3 + c] + mat1[6 + r]*mat2[6 + c]; };
  const V3 d21 = c2 - c1, d12 = c1 - c2;
  const V3 p21{mtrow(mat1, 0, d21), mtrow(mat1, 1, d21), mtrow(mat1, 2, d21)};   // box 2 centre in frame 1
  const V3 p12{mtrow(mat2, 0, d12), mtrow(mat2, 1, d12), mtrow(mat2, 2, d12)};   // box 1 centre in frame 2
  const real septol = margin + MJH_BB_SEPEPS*(size1[0] + size1[1] + size1[2] + size2[0] + size2[1] + size2[2]);

  // ---- 1. separation along my axis
  real sep = 0;
  int valid = lane < 15;
  if (lane < 3) {
    const int i = lane;
    const real r2 = fabs(R(i, 0))*size2[0] + fabs(R(i, 1))*size2[1] + fabs(R(i, 2))*size2[2];
    sep = fabs(comp(p21, i)) - size1[i] - r2;
  } else if (lane < 6) {
    const int j = lane - 3;
    const real r1 = fabs(R(0, j))*size1[0] + fabs(R(1, j))*size1[1] + fabs(R(2, j))*size1[2];
    sep = fabs(comp(p12, j)) - size2[j] - r1;
  } else if (lane < 15) {
    const int i = (lane - 6) / 3, j = (lane - 6) % 3;
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    real a1 = -R(i2, j), a2 = R(i1, j);
    const real n2 = a1*a1 + a2*a2;
    if (n2 < MJH_BB_PAREPS) valid = 0;          // edges (nearly) parallel: no axis
    else {
      const real inv = 1/sqrt(n2);
      a1 *= inv;
      a2 *= inv;
      const real r1 = size1[i1]*fabs(a1) + size1[i2]*fabs(a2);
      const real b1 = a1*R(i1, j1) + a2*R(i2, j1);
      const real b2 = a1*R(i1, j2) + a2*R(i2, j2);
      const real r2 = size2[j1]*fabs(b1) + size2[j2]*fabs(b2);
      sep = fabs(a1*comp(p21, i1) + a2*comp(p21, i2)) - r1 - r2;
    }
  }
  if (wv_any(valid && sep > septol)) return 0;
  // in-order scan: faces take the first strict maximum, an edge must beat the best so far by the
  // bias AND the best face
  real best = -MJH_MAXVAL;
  int code = -1;
  for (int a = 0; a < 6; a++) {
    const real s = wv_bcast(sep, a);
    if (s > best) { best = s; code = a; }
  }
  const real best_face = best;
  const int code_face = code;
  const unsigned vmask = (unsigned)wv_ballot(valid);
  for (int a = 6; a < 15; a++) {
    const real s = wv_bcast(sep, a);
    if (((vmask >> a) & 1) && s - MJH_BB_EDGEBIAS*fabs(s) > best && s > best_face) { best = s; code = a; }
  }
  if (code < 0) return 0;

  // edge axis of pair (i, j) in frame 1
  auto edge_axis = [&](int i, int j) -> V3 {
    V3 ax = from_cyclic(i, (real)0, -R((i + 2) % 3, j), R((i + 1) % 3, j));
    unitize(ax);
    return ax;
  };
  if (code >= 6) {
    // an edge axis nearly parallel to the best face normal gives way to the face (:806-826)
    const V3 ax = edge_axis((code - 6) / 3, (code - 6) % 3);
    real along;
    if (code_face < 3) along = fabs(comp(ax, code_face));
    else { const int f = code_face - 3; along = fabs(ax.x*R(0, f) + ax.y*R(1, f) + ax.z*R(2, f)); }
    if (along > 0.99 && best < best_face + 0.05*fabs(best_face) + MJH_MINVAL) { code = code_face; best = best_face; }
  }

  // ---- 2. edge-edge contact (:830-945)
  if (code >= 6) {
    const int i = (code - 6) / 3, j = (code - 6) % 3;
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    V3 ax = edge_axis(i, j);
    if (dot(ax, p21) < 0) ax = V3{-ax.x, -ax.y, -ax.z};
    const V3 ax2{ax.x*R(0, 0) + ax.y*R(1, 0) + ax.z*R(2, 0),
                 ax.x*R(0, 1) + ax.y*R(1, 1) + ax.z*R(2, 1),
                 ax.x*R(0, 2) + ax.y*R(1, 2) + ax.z*R(2, 2)};      // the axis in frame 2
    // a supporting edge is ambiguous when the axis is (nearly) orthogonal to one of its offsets
    int amb1 = -1, amb2 = -1;
    if (fabs(comp(ax, i1)) < MJH_BB_SGNEPS) amb1 = i1; else if (fabs(comp(ax, i2)) < MJH_BB_SGNEPS) amb1 = i2;
    if (fabs(comp(ax2, j1)) < MJH_BB_SGNEPS) amb2 = j1; else if (fabs(comp(ax2, j2)) < MJH_BB_SGNEPS) amb2 = j2;
    const V3 e2{R(0, j), R(1, j), R(2, j)};            // direction of box 2's edge in frame 1
    const real b = comp(e2, i);
    const real denom = 1 - b*b;
    // lane q < 4: choice (q >> 1) of edge 1, (q & 1) of edge 2
    const int f1 = (lane >> 1) & 1, f2 = lane & 1;
    const int exists = lane < 4 && (!f1 || amb1 >= 0) && (!f2 || amb2 >= 0);
    real o1 = comp(ax, i1) >= 0 ? (real)size1[i1] : -(real)size1[i1];
    real o2 = comp(ax, i2) >= 0 ? (real)size1[i2] : -(real)size1[i2];
    if (f1 && amb1 == i1) o1 = -o1;
    if (f1 && amb1 == i2) o2 = -o2;
    const V3 m1 = from_cyclic(i, (real)0, o1, o2);      // middle of edge 1
    real q1 = comp(ax2, j1) >= 0 ? -(real)size2[j1] : (real)size2[j1];
    real q2 = comp(ax2, j2) >= 0 ? -(real)size2[j2] : (real)size2[j2];
    if (f2 && amb2 == j1) q1 = -q1;
    if (f2 && amb2 == j2) q2 = -q2;
    const V3 l2 = from_cyclic(j, (real)0, q1, q2);
    const V3 m2 = V3{R(0, 0)*l2.x + R(0, 1)*l2.y + R(0, 2)*l2.z,
                     R(1, 0)*l2.x + R(1, 1)*l2.y + R(1, 2)*l2.z,
                     R(2, 0)*l2.x + R(2, 1)*l2.y + R(2, 2)*l2.z} + p21;      // middle of edge 2, frame 1
    const V3 mm = m2 - m1;
    const real g1 = comp(mm, i), g2 = dot(e2, mm);
    real s = denom < MJH_MINVAL ? 0 : (g1 - b*g2) / denom;
    s = r_clip(s, -size1[i], size1[i]);
    const real t = r_clip(b*s - g2, -size2[j], size2[j]);
    s = r_clip(g1 + b*t, -size1[i], size1[i]);
    const V3 w1 = with_comp(m1, i, comp(m1, i) + s);
    const V3 w2 = m2 + e2*t;
    const V3 gapv = w2 - w1;
    const real gap2 = dot(gapv, gapv);
    // first strictly smaller gap in choice order wins
    int win = -1;
    real bestgap = MJH_MAXVAL;
    const unsigned em = (unsigned)wv_ballot(exists);
    for (int q = 0; q < 4; q++) {
      const real g = wv_bcast(gap2, q);
      if (((em >> q) & 1) && g < bestgap) { bestgap = g; win = q; }
    }
    V3 a1{0, 0, 0}, a2{0, 0, 0};
    if (win >= 0) {
      a1 = V3{wv_bcast(w1.x, win), wv_bcast(w1.y, win), wv_bcast(w1.z, win)};
      a2 = V3{wv_bcast(w2.x, win), wv_bcast(w2.y, win), wv_bcast(w2.z, win)};
    }
    const real dist = dot(a2 - a1, ax);
    if (dist > septol) return 0;
    const V3 mid{0.5*(a1.x + a2.x), 0.5*(a1.y + a2.y), 0.5*(a1.z + a2.z)};
    h.dist = dist;
    h.pos = mmul(mat1, mid) + c1;
    h.nrm = mmul(mat1, ax);
    has = lane == 0;
    rank = 0;
    return 1;
  }

  // ---- 3. face contact (:949-1066): reference face `a` of box `ref1 ? 1 : 2`
  const int ref1 = code < 3;
  const int a = ref1 ? code : code - 3;
  const int ax = (a + 1) % 3, ay = (a + 2) % 3;
  auto Rinc = [&](int r, int c) -> real { return ref1 ? R(r, c) : R(c, r); };   // incident box in the reference frame
  const V3 poi = ref1 ? p21 : p12;
  const real ref_a = ref1 ? (real)size1[a] : (real)size2[a];
  const real ref_x = ref1 ? (real)size1[ax] : (real)size2[ax];
  const real ref_y = ref1 ? (real)size1[ay] : (real)size2[ay];
  const real sgn = comp(poi, a) >= 0 ? 1 : -1;
  // incident face: the one most anti-parallel to the reference normal
  int binc = 0;
  for (int k = 1; k < 3; k++) if (fabs(Rinc(a, k)) > fabs(Rinc(a, binc))) binc = k;
  const real tinc = sgn*Rinc(a, binc) > 0 ? -1 : 1;
  const int bu = (binc + 1) % 3, bv = (binc + 2) % 3;
  const real inc_b = ref1 ? (real)size2[binc] : (real)size1[binc];
  const real inc_u = ref1 ? (real)size2[bu] : (real)size1[bu];
  const real inc_v = ref1 ? (real)size2[bv] : (real)size1[bv];
  // face centre and half edges in (x, y, depth) of the reference face
  V3 fc{comp(poi, ax) + tinc*inc_b*Rinc(ax, binc), comp(poi, ay) + tinc*inc_b*Rinc(ay, binc), comp(poi, a) + tinc*inc_b*Rinc(a, binc)};
  V3 fu{inc_u*Rinc(ax, bu), inc_u*Rinc(ay, bu), inc_u*Rinc(a, bu)};
  V3 fv{inc_v*Rinc(ax, bv), inc_v*Rinc(ay, bv), inc_v*Rinc(a, bv)};
  fc.z = sgn*fc.z - ref_a;
  fu.z *= sgn;
  fv.z *= sgn;
  // my vertex of the polygon (lanes 0..nv-1)
  const real su = (lane == 0 || lane == 3) ? 1 : -1, sv = (lane < 2) ? 1 : -1;
  V3 vert{fc.x + su*fu.x + sv*fv.x, fc.y + su*fu.y + sv*fv.y, fc.z + su*fu.z + sv*fv.z};
  int nvert = 4;
  for (int side = 0; side < 4; side++) {
    const real sign = (side & 1) ? -1 : 1, limit = side < 2 ? ref_x : ref_y;
    const int mine = lane < nvert;
    const real dp = sign*(side < 2 ? vert.x : vert.y) - limit;
    if (!wv_any(mine && !(dp <= 0))) continue;              // nothing outside this plane
    const int nxt = (lane + 1 == nvert) ? 0 : lane + 1;
    const real dq = wv_shfl(dp, nxt);
    const V3 vq{wv_shfl(vert.x, nxt), wv_shfl(vert.y, nxt), wv_shfl(vert.z, nxt)};
    const int keep = mine && dp <= 0;
    const int cut = mine && ((dp < 0 && dq > 0) || (dp > 0 && dq < 0));
    const real t = dp / (dp - dq);
    const V3 crossing{vert.x + t*(vq.x - vert.x), vert.y + t*(vq.y - vert.y), vert.z + t*(vq.z - vert.z)};
    const V3 out0 = keep ? vert : crossing, out1 = crossing;
    const int cnt = keep + cut;
    const int start = wv_exscan_i(cnt);
    const int total = wv_sum_i(cnt);
    // gather: output vertex `lane` comes from the input lane whose [start, start+cnt) holds it
    int src = 0, second = 0;
    for (int k = 0; k < nvert; k++) {
      const int sk = wv_bcast_i(start, k), ck = wv_bcast_i(cnt, k);
      if (lane >= sk && lane < sk + ck) { src = k; second = lane - sk; }
    }
    const V3 g0{wv_shfl(out0.x, src), wv_shfl(out0.y, src), wv_shfl(out0.z, src)};
    const V3 g1{wv_shfl(out1.x, src), wv_shfl(out1.y, src), wv_shfl(out1.z, src)};
    vert = second ? g1 : g0;
    nvert = total;
  }
  // contacts: vertices not above the margin, in polygon order, minus near-duplicates of earlier ones
  const real dupe2 = MJH_BB_DUPEPS*(ref_x*ref_x + ref_y*ref_y);
  int accepted = 0;
  for (int k = 0; k < nvert; k++) {
    const real xk = wv_bcast(vert.x, k), yk = wv_bcast(vert.y, k), zk = wv_bcast(vert.z, k);
    const real dx = vert.x - xk, dy = vert.y - yk;
    const int twin = accepted && lane < k && dx*dx + dy*dy < dupe2;
    const int dup = wv_any(twin);
    if (lane == k && !(zk > margin) && !dup) accepted = 1;
  }
  const unsigned am = (unsigned)wv_ballot(accepted);
  const int ncontact = __builtin_popcount(am);
  if (!ncontact) return 0;
  has = accepted;
  rank = __builtin_popcount(am & lanes_below(lane));
  const real nsign = ref1 ? sgn : -sgn;
  const V3 lp = from_cyclic(a, sgn*(ref_a + 0.5*vert.z), vert.x, vert.y);
  h.dist = vert.z;
  if (ref1) { h.pos = mmul(mat1, lp) + c1; h.nrm = mcol(mat1, a)*nsign; }
  else { h.pos = mmul(mat2, lp) + c2; h.nrm = mcol(mat2, a)*nsign; }
  return ncontact;
}
#endif  // !MJH_LANE_MODE

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

// Oriented-box cull for the GJK / EPA pairs: 1 = the geoms' local bounding boxes (geom_aabb: centre and half sizes
// in the geom frame) are separated by clearly more than `margin` along one of the 15 candidate axes.  Every convex shape
// lies inside its box, so the true distance is at least that separation.  mjc_Convex keeps a contact only if the
// distance GJK / EPA report is below the margin, and that distance is accurate to opt.ccd_tolerance -- NOT an upper
// bound: two boxes a nanometre apart come back as touching with a slightly negative depth (stacked_boxes.xml).  The
// cull therefore leaves a band of 1000 tolerances (1 mm at the default 1e-6) above the margin to the narrowphase: a
// culled pair cannot produce a contact, a pair that passes is handled exactly as before -- the contact list is unchanged,
// the narrowphase just sees fewer pairs on contact-rich scenes.
template <class P0, class P1>
MJH_DEV int filter_obb(MREF M, P0 gx, P1 gm, int g1, int g2, real margin) {
  auto a1 = M.geom_aabb + 6*g1;
  auto a2 = M.geom_aabb + 6*g2;
  crptr R1 = gm + 9*g1; crptr R2 = gm + 9*g2;
  real c1[3], c2[3], t[3];
  for (int k = 0; k < 3; k++) {
    c1[k] = gx[3*g1 + k] + (R1[3*k]*a1[0] + R1[3*k + 1]*a1[1] + R1[3*k + 2]*a1[2]);
    c2[k] = gx[3*g2 + k] + (R2[3*k]*a2[0] + R2[3*k + 1]*a2[1] + R2[3*k + 2]*a2[2]);
    t[k] = c2[k] - c1[k];
  }
  const real h1[3] = {a1[3], a1[4], a1[5]}, h2[3] = {a2[3], a2[4], a2[5]};
#ifndef MJH_OBB_SLACK_TOLS
#define MJH_OBB_SLACK_TOLS 1000
#endif
  const real bound = margin + MJH_OBB_SLACK_TOLS*M.o.ccd_tolerance + 1e-9;
  // C[i][j] = axis i of box 1 . axis j of box 2 (axes = columns of the rotation matrices)
  real C[3][3], Cabs[3][3], t1[3], t2[3];
  for (int i = 0; i < 3; i++) {
    t1[i] = t[0]*R1[i] + t[1]*R1[3 + i] + t[2]*R1[6 + i];
    t2[i] = t[0]*R2[i] + t[1]*R2[3 + i] + t[2]*R2[6 + i];
    for (int j = 0; j < 3; j++) {
      C[i][j] = R1[i]*R2[j] + R1[3 + i]*R2[3 + j] + R1[6 + i]*R2[6 + j];
      Cabs[i][j] = fabs(C[i][j]);
    }
  }
  for (int i = 0; i < 3; i++) {
    const real rb = h2[0]*Cabs[i][0] + h2[1]*Cabs[i][1] + h2[2]*Cabs[i][2];
    if (fabs(t1[i]) - (h1[i] + rb) > bound) return 1;
  }
  for (int j = 0; j < 3; j++) {
    const real ra = h1[0]*Cabs[0][j] + h1[1]*Cabs[1][j] + h1[2]*Cabs[2][j];
    if (fabs(t2[j]) - (ra + h2[j]) > bound) return 1;
  }
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      // axis L = a_i x b_j: |L|^2 = 1 - C[i][j]^2 (unit axes); nearly parallel edges give no usable axis
      const real len2 = 1 - C[i][j]*C[i][j];
      if (len2 < 1e-6) continue;
      const real sep = fabs(t1[i2]*C[i1][j] - t1[i1]*C[i2][j]) -
                       (h1[i1]*Cabs[i2][j] + h1[i2]*Cabs[i1][j] + h2[j1]*Cabs[i][j2] + h2[j2]*Cabs[i][j1]);
      if (sep > 0 && sep*sep > bound*bound*len2*(1 + 1e-9)) return 1;
    }
  }
  return 0;
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
// Broad- and midphase, reproduced as per-pair predicates.
// The reference decides which geom pairs reach the narrowphase with (a) sweep-and-prune over body
// bounding intervals in a PCA frame whose sweep-axis end points are rounded to float (mj_broadphase /
// mj_SAP / makeAAMM, engine_collision_driver.c:1250-1735) and (b), for bodies with several geoms,
// a descent through two bounding-volume hierarchies with oriented-box tests (mj_collideTree /
// mj_collideOBB, :898-1240).  Both only cull, but not always conservatively to the last bit (geoms
// that touch to within rounding), and contact COUNTS must be exact -- so the culls are evaluated
// here with the reference's own arithmetic instead of being argued away:
//   * the sweep order is a pure function of the (float value, array position) of the interval end
//     points, so "pair reported by the sweep" is a closed-form predicate of the two intervals;
//   * the chain of BVH node pairs a geom pair has to survive is static (mjh_model_build.h).
// They are asked lazily, by the narrowphase lanes that are about to emit a contact (bp_lazy_cull):
// most steps of most models never need the PCA frame (a serial eigen-decomposition) or a BVH walk.
// stage_collision runs right after kinematics, while the inertial frames are still resident.
// ------------------------------------------------------------------------------------------------

// eigen-frame of a symmetric 3x3 (mju_eig3, engine_util_solve.c:1096-1189): Jacobi rotations
// accumulated in a quaternion, eigenvalues sorted decreasingly by quarter turns
MJH_DEV void bp_eig3(real* eigvec, const real* mat) {
  const real eps = MJH_MINVAL*1000;
  real quat[4] = {1, 0, 0, 0}, eigval[3] = {0, 0, 0};
  real D[9], tmp[9];
  for (int iter = 0; iter < 500; iter++) {
    q_tomat(eigvec, quat);
    // tmp = eigvec' * mat ; D = tmp * eigvec
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
      tmp[3*r + c] = eigvec[r]*mat[c] + eigvec[3 + r]*mat[3 + c] + eigvec[6 + r]*mat[6 + c];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
      D[3*r + c] = tmp[3*r]*eigvec[c] + tmp[3*r + 1]*eigvec[3 + c] + tmp[3*r + 2]*eigvec[6 + c];
    eigval[0] = D[0]; eigval[1] = D[4]; eigval[2] = D[8];
    // largest off-diagonal element: (0,1) about z, (0,2) about y, (1,2) about x
    int rotk;
    real off, drr, dcc;
    if (fabs(D[1]) > fabs(D[2]) && fabs(D[1]) > fabs(D[5])) { rotk = 2; off = D[1]; drr = D[0]; dcc = D[4]; }
    else if (fabs(D[2]) > fabs(D[5])) { rotk = 1; off = D[2]; drr = D[0]; dcc = D[8]; }
    else { rotk = 0; off = D[5]; drr = D[4]; dcc = D[8]; }
    if (fabs(off) < eps) break;
    const real tau = (dcc - drr)/(2*off);
    real t;
    if (tau >= 0) t = 1.0/(tau + sqrt(1 + tau*tau));
    else t = -1.0/(-tau + sqrt(1 + tau*tau));
    const real c = 1.0/sqrt(1 + t*t);
    if (c > 1.0 - eps) break;
    real sn = (tau >= 0 ? -sqrt(0.5 - 0.5*c) : sqrt(0.5 - 0.5*c));
    if (rotk == 1) sn = -sn;
    real rot[4] = {sqrt(1.0 - sn*sn), rotk == 0 ? sn : (real)0, rotk == 1 ? sn : (real)0, rotk == 2 ? sn : (real)0};
    q_normalize(rot);
    q_mul(quat, quat, rot);
    q_normalize(quat);
  }
  for (int j = 0; j < 3; j++) {
    const int j1 = j % 2;
    const real lo = j1 == 0 ? eigval[0] : eigval[1], hi = j1 == 0 ? eigval[1] : eigval[2];
    if (lo + eps < hi) {
      if (j1 == 0) { eigval[0] = hi; eigval[1] = lo; } else { eigval[1] = hi; eigval[2] = lo; }
      const real h = 0.707106781186548;
      const int ax = (j1 + 2) % 3;
      real rot[4] = {h, ax == 0 ? h : (real)0, ax == 1 ? h : (real)0, ax == 2 ? h : (real)0};
      q_mul(quat, quat, rot);
      q_normalize(quat);
    }
  }
  q_tomat(eigvec, quat);
}

// oriented-box overlap of two boxes given as (centre[3], half[3]) in frames (pos, mat), Gottschalk's
// 6 face axes (mj_collideOBB, :898-991).  CACHED: the BVH-node form, which goes through the products
// of the frame axes; else the leaf form, which goes through the box centres.  1 = may overlap.
template <int CACHED, class A1, class A2, class P1, class M1, class P2, class M2>
MJH_DEV int bp_obb(A1 box1, A2 box2, P1 pos1, M1 mat1, P2 pos2, M2 mat2, real margin) {
  const int inf1a = box1[3] >= MJH_MAXVAL, inf1b = box1[4] >= MJH_MAXVAL, inf1c = box1[5] >= MJH_MAXVAL;
  const int inf2a = box2[3] >= MJH_MAXVAL, inf2b = box2[4] >= MJH_MAXVAL, inf2c = box2[5] >= MJH_MAXVAL;
  if ((inf1a && inf1b && inf1c) || (inf2a && inf2b && inf2c)) return 1;
  const int infinite1 = inf1a || inf1b || inf1c, infinite2 = inf2a || inf2b || inf2c;
  // normal[i][j] = column j of mat_i
  for (int j = 0; j < 2; j++) {          // box whose face normals are tested
    if (j == 0 ? infinite2 : infinite1) continue;
    for (int k = 0; k < 3; k++) {
      const V3 nk = j == 0 ? mcol(mat1, k) : mcol(mat2, k);
      real proj[2], radius[2];
      for (int i = 0; i < 2; i++) {
        const V3 a0 = i == 0 ? mcol(mat1, 0) : mcol(mat2, 0);
        const V3 a1 = i == 0 ? mcol(mat1, 1) : mcol(mat2, 1);
        const V3 a2 = i == 0 ? mcol(mat1, 2) : mcol(mat2, 2);
        const real bc0 = i == 0 ? (real)box1[0] : (real)box2[0], bc1 = i == 0 ? (real)box1[1] : (real)box2[1], bc2 = i == 0 ? (real)box1[2] : (real)box2[2];
        const real bh0 = i == 0 ? (real)box1[3] : (real)box2[3], bh1 = i == 0 ? (real)box1[4] : (real)box2[4], bh2 = i == 0 ? (real)box1[5] : (real)box2[5];
        // product[adr + l] = normal[i][l] . normal[j][k]
        const real pr0 = dot(a0, nk), pr1 = dot(a1, nk), pr2 = dot(a2, nk);
        if (CACHED) {
          const V3 xp = i == 0 ? ld3(pos1) : ld3(pos2);
          proj[i] = bc0*pr0 + bc1*pr1 + bc2*pr2 + dot(xp, nk);
        } else {
          const V3 xp = i == 0 ? ld3(pos1) : ld3(pos2);
          const V3 ctr = (i == 0 ? mmul(mat1, V3{bc0, bc1, bc2}) : mmul(mat2, V3{bc0, bc1, bc2})) + xp;
          proj[i] = dot(ctr, nk);
        }
        radius[i] = fabs(bh0*pr0) + fabs(bh1*pr1) + fabs(bh2*pr2);
      }
      if (radius[0] + radius[1] + margin < fabs(proj[1] - proj[0])) return 0;
    }
  }
  return 1;
}

// PCA frame of the non-world geom positions (mj_broadphase, :1627-1660); 0 if there is none
template <class GX>
MJH_DEV int bp_frame(MREF M, GX gx, real* frame) {
  const MJH_CONST_AS DSizes& s = M.s;
  real cen[3] = {0, 0, 0};
  int cnt = 0;
  for (int g = 0; g < s.ngeom; g++) if (M.geom_bodyid[g]) { cen[0] += gx[3*g]; cen[1] += gx[3*g + 1]; cen[2] += gx[3*g + 2]; cnt++; }
  if (cnt == 0 || s.nbp < 2) return 0;
  const real inv = 1.0/cnt;
  cen[0] *= inv; cen[1] *= inv; cen[2] *= inv;
  real cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int g = 0; g < s.ngeom; g++) if (M.geom_bodyid[g]) {
    const real d0 = gx[3*g] - cen[0], d1 = gx[3*g + 1] - cen[1], d2 = gx[3*g + 2] - cen[2];
    const real D00 = d0*d0, D01 = d0*d1, D02 = d0*d2, D11 = d1*d1, D12 = d1*d2, D22 = d2*d2;
    cov[0] += D00; cov[1] += D01; cov[2] += D02; cov[3] += D01; cov[4] += D11; cov[5] += D12;
    cov[6] += D02; cov[7] += D12; cov[8] += D22;
  }
  for (int k = 0; k < 9; k++) cov[k] *= inv;
  bp_eig3(frame, cov);
  return 1;
}

// bounding interval of a body along the three frame axes (makeAAMM, :1250-1350)
template <class GX, class GM>
MJH_DEV void bp_body_interval(MREF M, GX gx, GM gm, int body, const real* frame, real* lo, real* hi) {
  const int g0 = M.body_geomadr[body], g1 = g0 + M.body_geomnum[body];
  for (int g = g0; g < g1; g++) {
    const real margin = M.geom_bpmargin[g];
    auto box = M.geom_aabb + 6*g;
    crptr xp = gx + 3*g; crptr xm = gm + 9*g;
    const V3 ctr = mmul(xm, V3{box[0], box[1], box[2]}) + ld3(xp);
    const real rhalf = M.geom_rbound[g];
    for (int j = 0; j < 3; j++) {
      const V3 fj{frame[3*j], frame[3*j + 1], frame[3*j + 2]};
      const real bcen = dot(ctr, fj);
      const real bhalf = fabs(box[3]*dot(mcol(xm, 0), fj)) + fabs(box[4]*dot(mcol(xm, 1), fj)) + fabs(box[5]*dot(mcol(xm, 2), fj));
      const real rcen = dot(ld3(xp), fj);
      const real l = r_max(rcen - rhalf, bcen - bhalf) - margin;
      const real h = r_min(rcen + rhalf, bcen + bhalf) + margin;
      if (g == g0) { lo[j] = l; hi[j] = h; }
      else { lo[j] = r_min(lo[j], l); hi[j] = r_max(hi[j], h); }
    }
  }
}

// would mj_SAP report the body pair at positions (a < b) of the collidable list?  Neither interval
// ends before the other begins in the order of the stable sort of (float end point, array position:
// 2*id for a minimum, 2*id + 1 for a maximum), and the other two axes overlap in double precision
// (:1439-1533)
template <class GX, class GM>
MJH_DEV int bp_sap_reports(MREF M, GX gx, GM gm, const real* frame, int sap) {
  const int a = sap & 0xffff, b = sap >> 16;
  real alo[3], ahi[3], blo[3], bhi[3];
  bp_body_interval(M, gx, gm, M.bp_body[a], frame, alo, ahi);
  bp_body_interval(M, gx, gm, M.bp_body[b], frame, blo, bhi);
  const float amin = (float)alo[0], amax = (float)ahi[0], bmin = (float)blo[0], bmax = (float)bhi[0];
  const int a_ends_first = amax < bmin || (amax == bmin && 2*a + 1 < 2*b);
  const int b_ends_first = bmax < amin || (bmax == amin && 2*b + 1 < 2*a);
  return !a_ends_first && !b_ends_first &&
         !(alo[1] > bhi[1] || blo[1] > ahi[1] || alo[2] > bhi[2] || blo[2] > ahi[2]);
}

// The lazy broad / midphase check of stage_collision, out of line so that the narrowphase loop does
// not carry its registers: a lane with p >= 0 is about to emit a contact of static pair p and asks
// whether the reference would have let that pair reach its narrowphase.
//   * sweep-and-prune (pairs with pair_sap >= 0): needs the PCA frame, computed by the first call of
//     a step that needs it and parked in the environment's global scratch
//     (state: -1 not yet, 0 no frame, 1 parked);
//   * BVH midphase (route 2): the pair's static chain of node-pair tests in the bodies' inertial
//     frames, then the leaf test on the geoms' own boxes with the summed margins (:1075-1083).
// Returns (state' + 1) | reached << 2.
MJH_DEVN int bp_lazy_cull(MREF M_, BREF B_, int e_, int p, int state) {
  MJH_ENTER(M_, B_, e_);
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  const int sap = p >= 0 ? (int)M.pair_sap[p] : -1;
  int ok = p >= 0;
  if (wv_any(sap >= 0)) {
    rptr park = MJH_G(B, scratch, e);
    real frame[9];
    if (state < 0) {
      state = bp_frame(M, gx, frame);
      if (state && wv_lane() == 0) for (int k = 0; k < 9; k++) park[k] = frame[k];
      wv_sync();
    } else if (state) {
      for (int k = 0; k < 9; k++) frame[k] = park[k];
    }
    if (sap >= 0) ok = state && bp_sap_reports(M, gx, gm, frame, sap);
  }
  if (ok && M.pair_route[p] == 2) {
    crptr xipos = MJH_F(B, xipos, e);
    crptr ximat = MJH_F(B, ximat, e);
    const int ga = M.pair_geom1[p], gb = M.pair_geom2[p];
    // the chain was built with body 1 = the lower body id; the geoms are stored type-ordered
    int b1 = M.geom_bodyid[ga], b2 = M.geom_bodyid[gb], g1 = ga, g2 = gb;
    if (b1 > b2) { const int t = b1; b1 = b2; b2 = t; g1 = gb; g2 = ga; }
    const real bmargin = M.pair_bodymargin[p], lmargin = M.pair_margin[p];
    for (int q = M.pair_mid_adr[p]; q < M.pair_mid_adr[p + 1] && ok; q++)
      ok = bp_obb<1>(M.bvh_aabb + 6*M.pair_mid[2*q], M.bvh_aabb + 6*M.pair_mid[2*q + 1],
                     xipos + 3*b1, ximat + 9*b1, xipos + 3*b2, ximat + 9*b2, bmargin);
    // (the leaf's bounding-sphere test with the summed margins is the narrowphase's own filter)
    if (ok) ok = bp_obb<0>(M.geom_aabb + 6*g1, M.geom_aabb + 6*g2, gx + 3*g1, gm + 9*g1, gx + 3*g2, gm + 9*g2, lmargin);
  }
  return (state + 1) | (ok << 2);
}

// contact record `c` of the environment <- one hit of static pair p
// (mj_narrowphase fill + mj_setContact, engine_collision_driver.c:2050-2075, :1839-1875)
MJH_DEV void store_contact(MREF M, BREF B, int e, int c, int p, const Hit& h) {
  MJH_CON(B, con_dist, e, 1, c)[0] = h.dist;
  st3(MJH_CON(B, con_pos, e, 3, c), h.pos);
  real fr[9] = {h.nrm.x, h.nrm.y, h.nrm.z, h.tan.x, h.tan.y, h.tan.z, 0, 0, 0};
  make_frame(fr);
  rptr cframe = MJH_CON(B, con_frame, e, 9, c);
  for (int q = 0; q < 9; q++) cframe[q] = fr[q];
  MJH_CON(B, con_pair, e, 1, c)[0] = p;
  iptr cgeom = MJH_CON(B, con_geom, e, 2, c);
  cgeom[0] = M.pair_geom1[p];
  cgeom[1] = M.pair_geom2[p];
  // (an adhesive contact stays active in the gap, as one frictionless row whose reference acceleration pulls: :1853-1862)
  const int ingap = (h.dist >= M.pair_includemargin[p]) ? 1 : 0;
  const int adhesive = MJH_HAS(MJH_FT_PASSIVEMISC) && M.o.has_adhesion && M.pair_adhesion[p] != 0;
  MJH_CON(B, con_dim, e, 1, c)[0] = (adhesive && ingap) ? 1 : (int)M.pair_dim[p];
  MJH_CON(B, con_exclude, e, 1, c)[0] = (ingap && !adhesive) ? 1 : 0;
  MJH_CON(B, con_efcadr, e, 1, c)[0] = -1;
  MJH_CON(B, con_mu, e, 1, c)[0] = 0;
  if (MJH_HAS(MJH_FT_FLEX) && M.s.nconflex) { iptr cf = MJH_G(B, con_flex, e) + MJH_CONFLEX*c; for (int q = 0; q < MJH_CONFLEX; q++) cf[q] = -1; }
}

// general convex pairs (GJK / EPA / multicontact): one pair per lane
#include "mjh_convex.h"

#if !MJH_LANE_MODE
// The group-cooperative colliders of one chunk of MJH_W pairs, in pair order (bit q of `todo`: the pair
// lane q holds in `mypair`).  `slot` is the contact slot my own pair's first contact would take; each cooperative pair
// before mine pushes it back by that pair's count.  Returns (how far my slot moved) | (contacts
// emitted) << 12 | overflow << 24.  Out of line: the point colliders around the call site are the hot
// path of most models, and keeping the clipping code out of their function keeps its register
// pressure where the lean build's is.
// (one pair: loads, collider, contact stores.  Returns its contact count | overflow << 8.  A function of
// its own so that nothing but loop counters is live across the collider calls of the loop below.)
MJH_DEVN int collide_coop_pair(MREF M_, BREF B_, int e_, int pq, int first) {
  MJH_ENTER(M_, B_, e_);
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  const int g1 = M.pair_geom1[pq], g2 = M.pair_geom2[pq];
  const real margin = M.pair_margin[pq];
  crptr mat1 = gm + 9*g1; auto size1 = M.geom_size + 3*g1;
  crptr mat2 = gm + 9*g2; auto size2 = M.geom_size + 3*g2;
  const V3 c1 = ld3(gx + 3*g1), c2 = ld3(gx + 3*g2);
  Hit hc;
  int has = 0, rank = 0, cnt = 0;
  const int func = M.pair_func[pq];
  if (func == MJH_COL_PLANE_BOX) cnt = coop_plane_box(hc, has, rank, margin, c1, mcol(mat1, 2), c2, mat2, size2);
  else if (func == MJH_COL_PLANE_CYLINDER) cnt = coop_plane_cylinder(hc, has, rank, margin, c1, mcol(mat1, 2), c2, mat2, size2);
  else if (func == MJH_COL_BOX_BOX) cnt = coop_box_box(hc, has, rank, margin, c1, mat1, size1, c2, mat2, size2);
  else if (func == MJH_COL_CAPSULE_BOX) cnt = coop_capsule_box(hc, has, rank, margin, c1, mat1, size1, c2, mat2, size2);
  int overflow = 0;
  if (has) {
    const int c = first + rank;
    if (c >= M.s.nconmax) overflow = 1; else store_contact(M, B, e, c, pq, hc);
  }
  return cnt | (overflow << 8);
}

MJH_DEVN int collide_coop_pairs(MREF M_, BREF B_, int e_, int mypair, unsigned todo_lo, unsigned todo_hi, int slot) {
  MJH_ENTER(M_, B_, e_);
  unsigned long long todo = ((unsigned long long)todo_hi << 32) | todo_lo;
  int moved = 0, emitted = 0, overflow = 0;
  while (todo) {
    const int q = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int r = collide_coop_pair(M, B, e, wv_bcast_i(mypair, q), wv_bcast_i(slot + moved, q));
    const int cnt = r & 0xff;
    overflow |= r >> 8;
    if (wv_lane() > q) moved += cnt;
    emitted += cnt;
  }
  return moved | (emitted << 12) | (wv_any(overflow) << 24);
}
#endif

#if !MJH_LANE_MODE
MJH_DEVN int flex_collide_job(MREF M_, BREF B_, int e_, int seg, int base);     // mjh_flexcol.h
MJH_DEVN int flex_self_collide(MREF M_, BREF B_, int e_, int sidx, int base);
MJH_DEVN int flex_pair_collide(MREF M_, BREF B_, int e_, int k, int base);
#endif
// ------------------------------------------------------------------------------------------------
// mj_collision over the static pair list
// ------------------------------------------------------------------------------------------------
MJH_DEVN void stage_collision(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  iptr counts = MJH_F(B, counts, e);
  const int dsbl = M.o.disableflags;
  if ((dsbl & (1<<0)) || (dsbl & (1<<4)) || (s.npair == 0 && s.ncolseg <= 1 && s.nflexself == 0 && s.nflexff == 0)) {
    if (wv_lane() == 0) counts[MJH_C_NCON] = 0;
    wv_sync();
    return;
  }
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  iptr warn = MJH_F(B, warning, e);
#if !MJH_LANE_MODE
  if (MJH_HAS(MJH_FT_COLCONVEX) && s.ccd_any) { if (wv_lane() == 0) rc_header(M, B, e)[193] = 0; wv_sync(); }
#endif

  int base = 0;        // contacts emitted by earlier chunks (uniform over the group)
  int overflow = 0;
  // sweep-and-prune predicate of the reference, evaluated only for pairs that would emit a contact;
  // the PCA frame is computed the first time a group needs it (-1: not yet)
  int frame_state = -1;

  // One round of narrowphase: lane's pair p (-1: none), already through the bounding-sphere filter.
  auto narrow = [&](const int p) {
    Hit ha, hb;          // the (at most two) contacts of my pair's point collider
    int n = 0;
    int coop = 0;        // my pair needs a group-cooperative collider
    int convex = 0;      // my pair takes the GJK / EPA narrowphase (1) or plane-convex (2): its contacts sit in my workspace records
    int unsupported = 0;
    if (p >= 0) {
      const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
      const real margin = M.pair_margin[p];       // margin + gap: collider threshold
      crptr mat1 = gm + 9*g1; auto size1 = M.geom_size + 3*g1;
      crptr mat2 = gm + 9*g2; auto size2 = M.geom_size + 3*g2;
      const V3 c1 = ld3(gx + 3*g1), c2 = ld3(gx + 3*g2);
      const int func = M.pair_func[p];
      switch (func) {
        case MJH_COL_PLANE_SPHERE:
          n = hit_plane_sphere(ha, margin, c1, mcol(mat1, 2), c2, size2[0]); break;
        case MJH_COL_PLANE_CAPSULE:
          n = hit_plane_capsule(ha, hb, margin, c1, mcol(mat1, 2), c2, mat2, size2); break;
        case MJH_COL_SPHERE_SPHERE:
          n = hit_sphere_sphere(ha, margin, c1, mat1, size1[0], c2, mat2, size2[0]); break;
        case MJH_COL_SPHERE_CAPSULE:
          n = hit_sphere_capsule(ha, margin, c1, mat1, size1[0], c2, mat2, size2); break;
        case MJH_COL_CAPSULE_CAPSULE:
          n = hit_capsule_capsule(ha, hb, margin, c1, mat1, size1, c2, mat2, size2); break;
        default:
          if (MJH_HAS(MJH_FT_COLCONVEX)) {
            // (out-of-line colliders write through a reference: a local of their own keeps `ha` in registers)
            if (func == MJH_COL_SPHERE_BOX) { Hit hx; n = hit_sphere_box(hx, margin, c1, size1[0], c2, mat2, size2); ha = hx; }
            else if (func == MJH_COL_SPHERE_CYLINDER) { Hit hx; n = hit_sphere_cylinder(hx, margin, c1, mat1, size1[0], c2, mat2, size2); ha = hx; }
            else if (func == MJH_COL_UNSUPPORTED || MJH_LANE_MODE) unsupported = 1;
            else if (func == MJH_COL_CONVEX) convex = 1;
            else if (func == MJH_COL_PLANE_CONVEX) convex = 2;
            else coop = 1;
          }
          break;
      }
    }
#if !MJH_LANE_MODE
    if (MJH_HAS(MJH_FT_COLCONVEX) && s.ccd_any) {
      // every lane runs its own pair's GJK / EPA to completion (mjh_convex.h)
      if (wv_any(convex == 1)) { const int r = ccd_convex_pair(M, B, e, convex == 1 ? p : -1); if (convex == 1) n = r; }
      if (wv_any(convex == 2)) { const int r = ccd_plane_convex_pair(M, B, e, convex == 2 ? p : -1); if (convex == 2) n = r; }
    }
#endif
    if (s.nbp) {
      // would the reference's broad / midphase have let the pair reach its narrowphase at all?
      // (asked only by lanes about to emit a contact of a pair that is subject to one of the culls)
      int ask = (p >= 0 && (n > 0 || coop) && (M.pair_sap[p] >= 0 || M.pair_route[p] == 2)) ? p : -1;
      if (ask >= 0 && n > 0) {
        // A contact at distance `dist` below the pair's margin means the two geoms' projections on ANY
        // axis come within dist of each other, so the (margin-inflated) bounds the culls compare overlap
        // by at least margin - dist.  If that slack exceeds what rounding the sweep's end points to
        // float (2^-24 relative, two end points within body reach of the geom centres) can eat, the
        // pair certainly reached the reference's narrowphase: no need to ask.
        const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
        real deepest = n > 1 ? r_min(ha.dist, hb.dist) : ha.dist;
#if !MJH_LANE_MODE
        if (MJH_HAS(MJH_FT_COLCONVEX) && convex) {
          const crptr rec = ccd_out_records(M, B, e);
          deepest = rec[0];
          for (int k = 1; k < n; k++) deepest = r_min(deepest, rec[7*k]);
        }
#endif
        const real slack = M.pair_margin[p] - deepest;
        const real reach = fabs(gx[3*g1]) + fabs(gx[3*g1 + 1]) + fabs(gx[3*g1 + 2]) + fabs(gx[3*g2]) + fabs(gx[3*g2 + 1]) +
                           fabs(gx[3*g2 + 2]) + M.body_bpext[M.geom_bodyid[g1]] + M.body_bpext[M.geom_bodyid[g2]];
        if (slack > 1e-6*(1 + reach)) ask = -1;
      }
      if (wv_any(ask >= 0)) {
        const int r = bp_lazy_cull(M, B, e, ask, frame_state);
        frame_state = (r & 3) - 1;
        if (ask >= 0 && !(r >> 2)) { n = 0; coop = 0; }
      }
    }
    // a pair whose collider mjhip does not have reached the narrowphase: the result could differ
    // from the reference's, so the environment is flagged (and frozen by the rollout loop)
    if (MJH_HAS(MJH_FT_COLCONVEX) && wv_any(unsupported) && wv_lane() == 0) warn[MJH_WARN_UNSUPPORTED]++;

    int before = wv_exscan_i(n);     // contacts of the round's earlier pairs (point colliders so far)
    int total = wv_sum_i(n);
#if !MJH_LANE_MODE
    if (MJH_HAS(MJH_FT_COLCONVEX)) {
      const unsigned long long todo = wv_ballot(coop);
      if (todo) {
        const int r = collide_coop_pairs(M, B, e, p, (unsigned)todo, (unsigned)(todo >> 32), base + before);
        before += r & 0xfff;
        total += (r >> 12) & 0xfff;
        overflow |= r >> 24;
      }
    }
#endif
    if (n > 0) {
      const int c = base + before;
#if !MJH_LANE_MODE
      if (MJH_HAS(MJH_FT_COLCONVEX) && convex) {
        const crptr rec = ccd_out_records(M, B, e);
        for (int k = 0; k < n; k++) {
          if (c + k >= s.nconmax) { overflow = 1; break; }
          Hit hx{rec[7*k], ld3(rec + 7*k + 1), ld3(rec + 7*k + 4), V3{0, 0, 0}};
          store_contact(M, B, e, c + k, p, hx);
        }
      } else
#endif
      {
        if (c >= s.nconmax) overflow = 1; else store_contact(M, B, e, c, p, ha);
        if (n > 1) { if (c + 1 >= s.nconmax) overflow = 1; else store_contact(M, B, e, c + 1, p, hb); }
      }
    }
    base += total;
  };
  // the static pairs are taken in ranges [plo, phi): one range for models without flexes, else the ranges between the
  // body : flex jobs (colseg, mjh_flexcol.h)
  int plo = 0, phi = s.npair;
  auto passes_filter = [&](int p) -> int {
    if (p >= phi) return 0;
    if (filter_sphere(M, gx, gm, M.pair_geom1[p], M.pair_geom2[p], M.pair_margin[p])) return 0;
#ifndef MJH_NO_OBB_CULL
    if (MJH_HAS(MJH_FT_COLCONVEX) && M.pair_func[p] == MJH_COL_CONVEX &&
        filter_obb(M, gx, gm, M.pair_geom1[p], M.pair_geom2[p], M.pair_margin[p])) return 0;
#endif
    return 1;
  };

  auto run_pairs = [&]() {
#if !MJH_LANE_MODE
  // Phase 1: the bounding-sphere filter over the whole pair list, survivors compacted into the lanes in
  // pair order (lane r takes the r-th survivor: the position of the r-th set bit of the chunk's ballot).
  // Phase 2: ONE narrowphase round over the survivors -- a humanoid has ~130 candidate pairs and a
  // handful in reach, and the narrowphase (divergent over the collider kinds) was run per chunk of 64.
  // More than MJH_W survivors: chunk by chunk as before.
  int mine = -1, nsurv = 0;
  for (int p0 = plo; p0 < phi && nsurv <= MJH_W; p0 += MJH_W) {
    const unsigned long long m = wv_ballot(passes_filter(p0 + wv_lane()));
    const int cnt = __builtin_popcountll(m);
    const int k = wv_lane() - nsurv;
    if (k >= 0 && k < cnt) {
      // position of the k-th set bit of m
      unsigned long long mm = m;
      int kk = k, pos = 0;
      for (int w = 32; w >= 1; w >>= 1) {
        const int c = __builtin_popcountll(mm & ((1ull << w) - 1));
        if (kk >= c) { kk -= c; mm >>= w; pos += w; }
      }
      mine = p0 + pos;
    }
    nsurv += cnt;
  }
  // position of the k-th set bit of m
  auto kth_bit = [](unsigned long long m, int k) -> int {
    int pos = 0;
    for (int w = 32; w >= 1; w >>= 1) {
      const int c = __builtin_popcountll(m & ((1ull << w) - 1));
      if (k >= c) { k -= c; m >>= w; pos += w; }
    }
    return pos;
  };
  const int nchunk = (phi - plo + MJH_W - 1)/MJH_W;
  if (nsurv <= MJH_W) {
    if (nsurv > 0) narrow(mine);
  } else if (2*nchunk <= B.n_iscratch) {
    // more survivors than lanes (contact-rich scenes: the 3x3x3 cube has ~200 of 325 pairs in reach): the
    // chunks' survivor ballots are parked, then the survivors are dealt to the lanes in pair order, a full
    // wavefront per narrowphase round -- the colliders (GJK / EPA above all) run with every lane busy
    // instead of once per chunk of the static pair list with whatever that chunk happens to hold
    iptr park = MJH_G(B, iscratch, e);
    int total = 0;
    for (int ch = 0; ch < nchunk; ch++) {
      const unsigned long long m = wv_ballot(passes_filter(plo + ch*MJH_W + wv_lane()));
      if (wv_lane() == 0) { park[2*ch] = (int)(unsigned)m; park[2*ch + 1] = (int)(unsigned)(m >> 32); }
      total += __builtin_popcountll(m);
    }
    wv_sync();
    // the k-th pair in reach (pair order)
    auto pick_of = [&](int k) -> int {
      if (k >= total) return -1;
      for (int ch = 0; ch < nchunk; ch++) {
        const unsigned long long m = ((unsigned long long)(unsigned)park[2*ch + 1] << 32) | (unsigned)park[2*ch];
        const int cnt = __builtin_popcountll(m);
        if (k < cnt) return plo + ch*MJH_W + kth_bit(m, k);
        k -= cnt;
      }
      return -1;
    };
    // polyhedral convex pairs: their distance phase for ALL pairs in reach at once, ahead of the rounds (a lane that has
    // finished its pair takes the next one: mjh_convex.h, ccd_poly_distance); the rounds look the results up
    int prepassed = 0;
    if (MJH_HAS(MJH_FT_COLCONVEX) && s.ccd_npoly > MJH_W) {
      const PolyTab tab = rc_poly_tables(M, B, e);
      int npoly = 0;
      for (int r0 = 0; r0 < total; r0 += MJH_W) {
        const int pick = pick_of(r0 + wv_lane());
        const int poly = pick >= 0 && M.pair_func[pick] == MJH_COL_CONVEX && rc_max_contacts(M, pick) > 1;
        const unsigned long long m = wv_ballot(poly);
        if (poly) { const int t = npoly + wv_rank_lt(m); tab.list[t] = pick; tab.slot[pick] = t; }
        npoly += __builtin_popcountll(m);
      }
      wv_sync();
      if (npoly > MJH_W) {
        ccd_poly_distance(M, B, e, npoly);
        if (wv_lane() == 0) rc_header(M, B, e)[193] = 1;
        wv_sync();
        prepassed = 1;
      }
    }
    for (int r0 = 0; r0 < total; r0 += MJH_W) narrow(pick_of(r0 + wv_lane()));
    if (prepassed) { if (wv_lane() == 0) rc_header(M, B, e)[193] = 0; wv_sync(); }
  } else
#endif
  for (int p0 = plo; p0 < phi; p0 += MJH_W) {
    const int p = p0 + wv_lane();
    narrow(passes_filter(p) ? p : -1);
  }
  };   // run_pairs
#if !MJH_LANE_MODE
  if (MJH_HAS(MJH_FT_FLEX) && s.ncolseg > 1) {
    for (int seg = 0; seg < s.ncolseg; seg++) {
      phi = M.colseg[3*seg];
      if (phi > plo) run_pairs();
      plo = phi;
      if (M.colseg[3*seg + 1] >= 0) {
        const int r = flex_collide_job(M, B, e, seg, base);
        base += r & 0xffff;
        overflow |= r >> 16;
      }
    }
  } else
#endif
  run_pairs();
#if !MJH_LANE_MODE
  // flex : flex pairs (the last of the body / flex pairs, mj_collision :795-813), then flex self-collisions (:834-881)
  if (MJH_HAS(MJH_FT_FLEX))
    for (int k = 0; k < s.nflexff; k++) {
      const int r = flex_pair_collide(M, B, e, k, base);
      base += r & 0xffff;
      overflow |= r >> 16;
    }
  if (MJH_HAS(MJH_FT_FLEX))
    for (int k = 0; k < s.nflexself; k++) {
      const int r = flex_self_collide(M, B, e, k, base);
      base += r & 0xffff;
      overflow |= r >> 16;
    }
#endif
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
  if (MJH_HAS(MJH_FT_FLEX) && s.nconside) flex_contact_nodes(M, B, e, overflow ? s.nconmax : base);
}
