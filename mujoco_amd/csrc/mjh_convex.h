// General convex narrowphase (what the reference reaches through mjc_Convex / mjc_PlaneConvex / mjc_ConvexElem):
// distance + penetration queries on the Minkowski difference of two convex shapes, and the face-clipping
// multi-contact recovery for polyhedral pairs.
//
// Reference (results are reproduced exactly; citations per routine below): engine_collision_convex.c -- mjc_Convex
// :881, mjc_PlaneConvex :1004, mjc_penetration :87, the support mappings :201-516 -- and engine_collision_gjk.c --
// gjk :198, gjkIntersect :420, subdistance :586-880, polytope2/3/4 :948-1215, epa :1358, multicontact :2123,
// mjc_ccd :2318.
//
// MAPPING: ONE GEOM PAIR PER 16-LANE ROW -- four pairs in flight per wavefront, the 16 lanes of a row cooperating
// on their pair; everything the pair owns sits in a ~3 KB row workspace that the LDS plan places in the
// environment's LDS block (field `ccd_row`, alive during the collision stage only).
//   * support mapping: lanes 0-7 of the row serve shape A, lanes 8-15 shape B, at the same time.  A mesh is searched
//     with one candidate vertex per lane -- a hill-climbing step looks at all hull neighbours of the current vertex
//     at once, the exhaustive search takes eight vertices per pass -- and a "first maximum" lane reduction (DPP quad
//     permutes + row mirrors) picks what the reference's sequential scan would pick;
//   * closest point of the simplex to the origin: the 11 sub-simplices of a tetrahedron (6 segments, 4 triangles,
//     the tetrahedron) are evaluated by 11 lanes in three waves of work, then the reference's precedence rules are
//     applied to the finished table -- the recursion (tetrahedron -> faces -> edges) is gone;
//   * origin-containment refinement: one lane per face of the tetrahedron;
//   * expanding polytope: faces are a structure of arrays; the closest-face search, the visibility of every face
//     from the new vertex, the duplicate-vertex test and the construction of the new cone of faces (one horizon edge
//     per lane, ordered compaction into the priority map) are lane-parallel.  Only the silhouette walk stays a
//     scalar walk (over an explicit stack of edge crossings), because the ORDER in which it meets faces and edges
//     decides face numbering and tie-breaks of later iterations;
//   * multi-contact: candidate face normals, the aligned-pair search (first hit in lexicographic order = lowest
//     set bit of a ballot), Sutherland-Hodgman clipping with one polygon edge per lane and prefix-sum output slots,
//     and one witness pair per lane.
// The polytope of a cubelet pair needs ~1 KB; the row workspace holds the first RC_VFAST vertices / RC_FFAST faces
// and the (rare) overflow continues in a per-environment global page, so the reference's capacity (6 x
// ccd_iterations faces) is kept without paying for it in LDS.
//
// Arithmetic follows the reference expression by expression (association, comparison direction, first-wins tie
// rules): contact counts, iteration counts and contact frames agree bit for bit
// (tests/test_convex_hostsim.py, tests/test_gpu_parity.py).
// (included once per SPMD mode by mjh_modes.h, after mjh_collision.h -- no include guard)

#if !MJH_LANE_MODE

#ifdef MJH_HOSTSIM
#define RC_COUNT(k) do { if ((wv_lane() & 15) == 0) ::mjhsim::rc_stats()[k]++; } while (0)      // (mjh_spmd.h: emulation-only work counters)
#else
#define RC_COUNT(k) do {} while (0)
#endif

#define RC_TINY2 (MJH_MINVAL*MJH_MINVAL)
#define RC_HUGE2 (MJH_MAXVAL*MJH_MAXVAL)
#define RC_DBLMAX 1.7976931348623157e308         // mjMAX_LIMIT
#define RC_FLTMAX 3.4028234663852886e38          // (double)FLT_MAX
#define RC_FACE_ALIGN 0.996                      // mjFACE_TOL, engine_collision_gjk.h:42
#define RC_EDGE_ALIGN 0.0888                     // mjEDGE_TOL
#define RC_NONE 0x7fffffff
#define MJH_GEOM_FLEX 100

// how a shape answers a support query
enum { SK_POINT = 0, SK_SPHERE, SK_SEGMENT, SK_CAPSULE, SK_ELLIPSOID, SK_CYLINDER, SK_BOX, SK_MESH_ALL, SK_MESH_CLIMB,
       SK_FLEXELEM };
enum { RC_MAXWIT = 4, RC_MAXOUT = 5, RC_RECORD = 7 };

// ---- row workspace ------------------------------------------------------------------------------------------------
// fast page (LDS by plan), reals: two shape frames (pos[3] mat[9] size[3] margin centre[3] -), the simplex (4 x: point
// on A, point on B), scratch, then the polytope (vertices, face normals / squared distances as a structure of arrays)
// with the multi-contact buffers laid over it; ints: simplex vertex ids, scratch, polytope (vertex ids, per-face
// vertex triple / three neighbours / map slot / visibility, the priority map, the horizon, the crossing stack).
// mirror of the sizes: mjh_model_build.h (rc_workspace)
enum { RC_VFAST = 10, RC_FFAST = 24, RC_MFAST = 24, RC_HFAST = 12, RC_KFAST = 12 };
enum { RO_FRAME = 0, RO_SIM = 40, RO_SCR = 64, RO_POLY = 96,
       RO_VX = RO_POLY, RO_FN = RO_VX + 6*RC_VFAST, RO_FAST_END = RO_FN + 4*RC_FFAST };
enum { IO_SIM = 0, IO_SCR = 8, IO_POLY = 32,
       IO_VID = IO_POLY, IO_FI = IO_VID + 2*RC_VFAST, IO_MAP = IO_FI + 6*RC_FFAST, IO_HZ = IO_MAP + RC_MFAST,
       IO_STK = IO_HZ + 2*RC_HFAST, IO_FAST_END = IO_STK + RC_KFAST };
enum { FR_POS = 0, FR_MAT = 3, FR_SIZE = 12, FR_MARGIN = 15, FR_CENTRE = 16, FR_N = 20 };
enum { FI_VERTS = 0, FI_ADJ0 = 1, FI_ADJ1 = 2, FI_ADJ2 = 3, FI_SLOT = 4, FI_VIS = 5 };   // per-face int arrays
enum { FS_UNMAPPED = -1, FS_DELETED = -2 };

struct RowMem {
  real* R; int* I;          // fast page
  real* RS; int* IS;        // overflow page (global)
  int nslow_v, nslow_f;     // capacities of the overflow page: vertices, faces (= map entries = horizon edges)
};
MJH_DEV real* rm_vert(const RowMem& m, int v) { return v < RC_VFAST ? m.R + RO_VX + 6*v : m.RS + 6*(v - RC_VFAST); }
MJH_DEV int* rm_vid(const RowMem& m, int v) { return v < RC_VFAST ? m.I + IO_VID + 2*v : m.IS + 2*(v - RC_VFAST); }
// component c (0..2 normal, 3 squared distance) of face f
MJH_DEV real& rm_fr(const RowMem& m, int c, int f) {
  return f < RC_FFAST ? m.R[RO_FN + c*RC_FFAST + f] : m.RS[6*m.nslow_v + c*m.nslow_f + (f - RC_FFAST)];
}
MJH_DEV int& rm_fi(const RowMem& m, int c, int f) {
  return f < RC_FFAST ? m.I[IO_FI + c*RC_FFAST + f] : m.IS[2*m.nslow_v + c*m.nslow_f + (f - RC_FFAST)];
}
MJH_DEV int& rm_map(const RowMem& m, int i) { return i < RC_MFAST ? m.I[IO_MAP + i] : m.IS[2*m.nslow_v + 6*m.nslow_f + (i - RC_MFAST)]; }
MJH_DEV int& rm_hz(const RowMem& m, int i) { return i < RC_HFAST ? m.I[IO_HZ + 2*i] : m.IS[2*m.nslow_v + 7*m.nslow_f + 2*(i - RC_HFAST)]; }
MJH_DEV int& rm_stk(const RowMem& m, int i) { return i < RC_KFAST ? m.I[IO_STK + i] : m.IS[2*m.nslow_v + 9*m.nslow_f + (i - RC_KFAST)]; }

MJH_DEV int rw_l() { return wv_lane() & 15; }
MJH_DEV V3 rw_neg(V3 v) { return V3{-1*v.x, -1*v.y, -1*v.z}; }
MJH_DEV V3 rw_scl(V3 v, real s) { return V3{s*v.x, s*v.y, s*v.z}; }
MJH_DEV real rw_len(V3 v) { return sqrt(dot(v, v)); }
MJH_DEV real rw_det(V3 a, V3 b, V3 c) { return a.x*(b.y*c.z - b.z*c.y) + a.y*(b.z*c.x - b.x*c.z) + a.z*(b.x*c.y - b.y*c.x); }
// Minkowski-difference point of a stored vertex (point on A, point on B)
MJH_DEV V3 rw_mink(const real* v) { return V3{v[0] - v[3], v[1] - v[4], v[2] - v[5]}; }

// ---- lane reductions inside a row: the value a sequential "if (x > best)" / "if (x < best)" scan would keep ----------
// STEPS = 3: over the caller's 8-lane half, 4: over the row.  Lanes without a candidate pass (-inf / +inf, RC_NONE).
template <int STEPS> MJH_DEV void rw_first_max(real& v, int& i) {
#define RW_STEP(S) { const real v2 = wv_row_xchg<S>(v); const int i2 = wv_row_xchg_i<S>(i); if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; } }
  RW_STEP(0) RW_STEP(1) RW_STEP(2)
  if (STEPS == 4) RW_STEP(3)
#undef RW_STEP
}
template <int STEPS> MJH_DEV void rw_first_min(real& v, int& i) {
#define RW_STEP(S) { const real v2 = wv_row_xchg<S>(v); const int i2 = wv_row_xchg_i<S>(i); if (v2 < v || (v2 == v && i2 < i)) { v = v2; i = i2; } }
  RW_STEP(0) RW_STEP(1) RW_STEP(2)
  if (STEPS == 4) RW_STEP(3)
#undef RW_STEP
}
MJH_DEV real rw_min_all(real v) {
  real o = wv_row_xchg<0>(v); v = o < v ? o : v;
  o = wv_row_xchg<1>(v); v = o < v ? o : v;
  o = wv_row_xchg<2>(v); v = o < v ? o : v;
  o = wv_row_xchg<3>(v); v = o < v ? o : v;
  return v;
}
MJH_DEV int rw_popc_below(unsigned m, int l) { return __builtin_popcount(m & ((1u << l) - 1)); }

// ---- shapes and their support mappings (engine_collision_convex.c:201-516) --------------------------------------------
struct Shape { int type, kind, mesh, vcache, gcache; real margin; };   // geom type, support kind, mesh id (corner count of a
                                                            // flex element), last support vertex, its hull-graph node, margin
struct Far { V3 pa, pb; int ia, ib, ga, gb; };              // farthest point of A along d / of B along -d, vertex ids, graph nodes

template <class PM> MJH_DEV V3 rw_to_local(PM mat, V3 d) {
  return V3{mat[0]*d.x + mat[3]*d.y + mat[6]*d.z, mat[1]*d.x + mat[4]*d.y + mat[7]*d.z, mat[2]*d.x + mat[5]*d.y + mat[8]*d.z};
}
template <class PM, class PP> MJH_DEV V3 rw_to_world(PM mat, V3 l, PP pos) {
  V3 r{mat[0]*l.x + mat[1]*l.y + mat[2]*l.z, mat[3]*l.x + mat[4]*l.y + mat[5]*l.z, mat[6]*l.x + mat[7]*l.y + mat[8]*l.z};
  r.x += pos[0]; r.y += pos[1]; r.z += pos[2];
  return r;
}
template <class PM> MJH_DEV V3 rw_rot(PM mat, real l1, real l2, real l3) {
  return V3{mat[0]*l1 + mat[1]*l2 + mat[2]*l3, mat[3]*l1 + mat[4]*l2 + mat[5]*l3, mat[6]*l1 + mat[7]*l2 + mat[8]*l3};
}
template <class PM, class PP> MJH_DEV V3 rw_rot_add(PM mat, PP pos, real l1, real l2, real l3) {
  V3 r = rw_rot(mat, l1, l2, l3);
  r.x += pos[0]; r.y += pos[1]; r.z += pos[2];
  return r;
}
MJH_DEV real rw_vdot(MREF M, V3 a, int vbase) {
  return a.x*(real)M.mesh_vert[vbase] + a.y*(real)M.mesh_vert[vbase + 1] + a.z*(real)M.mesh_vert[vbase + 2];
}
MJH_DEV V3 rw_mesh_vert(MREF M, int vbase) {
  return V3{(real)M.mesh_vert[vbase], (real)M.mesh_vert[vbase + 1], (real)M.mesh_vert[vbase + 2]};
}

// Hill climbing over a mesh's hull graph towards local direction ld, from graph node `cur`: a step examines all
// neighbours of the current node at once (one per lane of the caller's 8-lane half, list order = lane order) and moves
// to the first best one if it beats the current node; stops when no neighbour does.  Returns the final node.
MJH_DEV int rw_hill_climb(MREF M, int mesh, V3 ld, int cur) {
  const int L = rw_l(), side = L >> 3, h = L & 7;
  const int vadr = 3*M.mesh_vertadr[mesh];
  const int gadr = M.mesh_graphadr[mesh];
  const int numvert = M.mesh_graph[gadr], numface = M.mesh_graph[gadr + 1];
  const int edgeadr = gadr + 2, globalid = gadr + 2 + numvert, localid = gadr + 2 + 2*numvert;
  const int listend = localid + numvert + 3*numface;
  real top = rw_vdot(M, ld, vadr + 3*M.mesh_graph[globalid + cur]);
  for (;;) {
    const int from = cur;
    int k0 = M.mesh_graph[edgeadr + from];
    for (;;) {
      const int at = localid + k0 + h;
      const int nb = at < listend ? M.mesh_graph[at] : -1;
      const unsigned ends = (wv_row_ballot(nb < 0) >> (8*side)) & 0xffu;
      const int nvalid = ends ? __builtin_ctz(ends) : 8;
      real v = h < nvalid ? rw_vdot(M, ld, vadr + 3*M.mesh_graph[globalid + nb]) : -HUGE_VAL;
      int iv = h < nvalid ? at : RC_NONE;
      rw_first_max<3>(v, iv);
      if (v > top) { top = v; cur = M.mesh_graph[iv]; }
      if (ends) break;
      k0 += 8;
    }
    if (cur == from) break;
  }
  return cur;
}

// Support point of the Minkowski difference A - B: lanes 0..7 of the row answer for A along d, lanes 8..15 for B along
// dn = -d, then the halves swap results.  Every lane returns the same Far.  (support, engine_collision_gjk.c:337)
MJH_DEVN_HOT Far rc_farthest(MREF M_, const real* frames, Shape a, Shape b, V3 d, V3 dn) {
  MREF M = wv_uniform_ref(M_);
  const int L = rw_l(), side = L >> 3, h = L & 7;
  const real* f = frames + (side ? FR_N : 0);
  const real* pos = f + FR_POS; const real* mat = f + FR_MAT; const real* size = f + FR_SIZE;
  const int kind = side ? b.kind : a.kind, mesh = side ? b.mesh : a.mesh;
  int vid = side ? b.vcache : a.vcache, gid = side ? b.gcache : a.gcache;
  const V3 dir = side ? dn : d;
  V3 out;
  if (kind == SK_POINT) out = ld3(pos);
  else if (kind == SK_SPHERE) {
    const real radius = size[0];
    out = V3{radius*dir.x + pos[0], radius*dir.y + pos[1], radius*dir.z + pos[2]};
  }
  else if (kind == SK_SEGMENT) {
    const real length = size[1];
    const real t = mat[2]*dir.x + mat[5]*dir.y + mat[8]*dir.z;
    const real scl = t >= 0 ? length : -length;
    out = V3{mat[2]*scl + pos[0], mat[5]*scl + pos[1], mat[8]*scl + pos[2]};
  }
  else if (kind == SK_CAPSULE) {
    const real radius = size[0], length = size[1];
    const V3 ld = rw_to_local(mat, dir);
    V3 ls{ld.x*radius, ld.y*radius, ld.z*radius};
    ls.z += (ld.z >= 0 ? length : -length);
    out = rw_to_world(mat, ls, pos);
  }
  else if (kind == SK_ELLIPSOID) {
    const V3 ld = rw_to_local(mat, dir);
    V3 ls{ld.x*size[0], ld.y*size[1], ld.z*size[2]};
    const real norm2 = ls.x*ls.x + ls.y*ls.y + ls.z*ls.z;
    if (norm2 < RC_TINY2) {
      out = V3{mat[0]*size[0] + pos[0], mat[3]*size[0] + pos[1], mat[6]*size[0] + pos[2]};
    } else {
      const real norm_inv = 1/sqrt(norm2);
      ls.x *= norm_inv*size[0]; ls.y *= norm_inv*size[1]; ls.z *= norm_inv*size[2];
      out = rw_to_world(mat, ls, pos);
    }
  }
  else if (kind == SK_CYLINDER) {
    const V3 ld = rw_to_local(mat, dir);
    const real n2 = ld.x*ld.x + ld.y*ld.y;
    const real scl = n2 >= RC_TINY2 ? size[0]/sqrt(n2) : 0;
    const V3 ls{scl*ld.x, scl*ld.y, ld.z >= 0 ? size[1] : -size[1]};
    out = rw_to_world(mat, ls, pos);
  }
  else if (kind == SK_BOX) {
    const V3 ld = rw_to_local(mat, dir);
    const V3 ls{ld.x >= 0 ? size[0] : -size[0], ld.y >= 0 ? size[1] : -size[1], ld.z >= 0 ? size[2] : -size[2]};
    vid = ((ls.x > 0) ? 1 : 0) | ((ls.y > 0) ? 2 : 0) | ((ls.z > 0) ? 4 : 0);
    out = rw_to_world(mat, ls, pos);
  }
  else if (kind == SK_FLEXELEM) {
    // corner with the largest projection (first wins), pushed out by radius + margin/2 (mjc_flexSupport :480)
    const bool has = h < mesh;
    const V3 v = has ? ld3(f + 3*h) : V3{0, 0, 0};
    real best = has ? v.x*dir.x + v.y*dir.y + v.z*dir.z : -HUGE_VAL;
    int ib = has ? h : RC_NONE;
    rw_first_max<3>(best, ib);
    const V3 c = ld3(f + 3*ib);
    const real scl = size[0];
    out = V3{c.x + dir.x*scl, c.y + dir.y*scl, c.z + dir.z*scl};
  }
  else if (kind == SK_MESH_ALL) {
    // every vertex, eight per pass; the cached vertex of the previous query only has to be beaten (:354)
    const int vadr = 3*M.mesh_vertadr[mesh], nverts = M.mesh_vertnum[mesh];
    const V3 ld = rw_to_local(mat, dir);
    real top = -RC_FLTMAX;
    int itop = 0;
    if (vid >= 0) { itop = vid; top = rw_vdot(M, ld, vadr + 3*itop); }
    for (int k0 = 0; k0 < nverts; k0 += 8) {
      const int k = k0 + h;
      real v = k < nverts ? rw_vdot(M, ld, vadr + 3*k) : -HUGE_VAL;
      int iv = k < nverts ? k : RC_NONE;
      rw_first_max<3>(v, iv);
      if (v > top) { top = v; itop = iv; }
    }
    vid = itop;
    out = rw_to_world(mat, rw_mesh_vert(M, vadr + 3*itop), pos);
  }
  else {
    // hill climbing over the hull graph: a step examines all neighbours of the current vertex at once; seeded from a
    // 3x3x3 direction grid or the cached vertex, whichever is farther (mjc_hillclimbSupport :396)
    const int vadr = 3*M.mesh_vertadr[mesh];
    const int gadr = M.mesh_graphadr[mesh];
    const int numvert = M.mesh_graph[gadr];
    const int globalid = gadr + 2 + numvert;
    const V3 ld = rw_to_local(mat, dir);
    const int cx = (ld.x > 0.4) - (ld.x < -0.4) + 1;
    const int cy = (ld.y > 0.4) - (ld.y < -0.4) + 1;
    const int cz = (ld.z > 0.4) - (ld.z < -0.4) + 1;
    const int seed = M.mesh_extrema[27*mesh + cx*9 + cy*3 + cz];
    int cur = seed;
    if (gid >= 0) {
      const real vc = rw_vdot(M, ld, vadr + 3*M.mesh_graph[globalid + gid]);
      const real vs = rw_vdot(M, ld, vadr + 3*M.mesh_graph[globalid + seed]);
      cur = (vs > vc) ? seed : gid;
    }
    cur = rw_hill_climb(M, mesh, ld, cur);
    gid = cur;
    vid = M.mesh_graph[globalid + cur];
    out = rw_to_world(mat, rw_mesh_vert(M, vadr + 3*vid), pos);
  }
  const real margin = side ? b.margin : a.margin;
  if (margin > 0) {
    const real hm = 0.5*margin;
    out.x += dir.x*hm; out.y += dir.y*hm; out.z += dir.z*hm;
  }
  wv_row_converge();
  const V3 other{wv_row_xchg<3>(out.x), wv_row_xchg<3>(out.y), wv_row_xchg<3>(out.z)};
  const int ovid = wv_row_xchg_i<3>(vid), ogid = wv_row_xchg_i<3>(gid);
  Far r;
  r.pa = side ? other : out; r.pb = side ? out : other;
  r.ia = side ? ovid : vid; r.ib = side ? vid : ovid;
  r.ga = side ? ogid : gid; r.gb = side ? gid : ogid;
  return r;
}

// ---- closest point of a simplex to the origin (signed-volume sub-algorithm, engine_collision_gjk.c:506-880) -----------
// origin projected on the plane through p, q, r; 1 = degenerate (projectOriginPlane :506)
MJH_DEV int rw_plane_foot(V3& res, V3 p, V3 q, V3 r) {
  const V3 qp = q - p, rp = r - p, rq = r - q;
  V3 n = cross(rq, qp);
  real nv = dot(n, q), nn = dot(n, n);
  if (nn == 0) return 1;
  if (nv != 0 && nn > MJH_MINVAL) { res = rw_scl(n, nv/nn); return 0; }
  n = cross(qp, rp);
  nv = dot(n, p); nn = dot(n, n);
  if (nn == 0) return 1;
  if (nv != 0 && nn > MJH_MINVAL) { res = rw_scl(n, nv/nn); return 0; }
  n = cross(rp, rq);
  nv = dot(n, r); nn = dot(n, n);
  res = rw_scl(n, nv/nn);
  return 0;
}
MJH_DEV int rw_sign_match(real a, real b) {
  if (a > 0 && b > 0) return 1;
  if (a < 0 && b < 0) return -1;
  return 0;
}
// barycentric weights of the point of segment ab closest to the origin (S1D :548)
MJH_DEV void rw_segment_weights(V3 a, V3 b, real& wa, real& wb) {
  const V3 ba = b - a;
  const real t = -(dot(b, ba)/dot(ba, ba));
  const V3 foot{b.x + t*ba.x, b.y + t*ba.y, b.z + t*ba.z};
  real span = a.x - b.x, widest = span;
  int axis = 0;
  span = a.y - b.y;
  if (fabs(span) >= fabs(widest)) { widest = span; axis = 1; }
  span = a.z - b.z;
  if (fabs(span) >= fabs(widest)) { widest = span; axis = 2; }
  const real ca = comp(foot, axis) - comp(b, axis);
  const real cb = comp(a, axis) - comp(foot, axis);
  const int inside = rw_sign_match(widest, ca) && rw_sign_match(widest, cb);
  wa = inside ? ca/widest : 0;
  wb = inside ? cb/widest : 1;
}
// index of segment (a, b), a < b, in the lane table of rc_simplex_weights
MJH_DEV int rw_edge_slot(int a, int b) { return a == 0 ? b - 1 : (a == 1 ? b + 1 : 5); }

// Weights lam[0..n) of the simplex's n points (n = 2..4; the simplex sits in the row workspace) for the point of the
// simplex closest to the origin.  Lane table: 0..5 the segments (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); 6..9 the
// triangles without point 0, 1, 2, 3; 10 the tetrahedron.  Three rounds: segments, then triangles (a triangle whose
// interior does not hold the foot point falls back on its segments: smallest distance, first wins in the reference's
// order (j,k) (i,k) (i,j)), then the tetrahedron (falls back on its triangles the same way).  (subdistance :586)
MJH_DEV void rc_simplex_weights(const RowMem& m, int n, real* lam) {
  const int L = rw_l();
  const real* sim = m.R + RO_SIM;
  real* tab = m.R + RO_SCR;                 // [0,12): segment weights, [12,24): triangle weights, [24,28): the answer
  if (L < 6) {
    const int a = L < 3 ? 0 : (L < 5 ? 1 : 2);
    const int b = L < 3 ? L + 1 : (L < 5 ? L - 1 : 3);
    if (b < n) {
      real wa, wb;
      rw_segment_weights(rw_mink(sim + 6*a), rw_mink(sim + 6*b), wa, wb);
      tab[2*L] = wa; tab[2*L + 1] = wb;
    }
  }
  wv_row_sync();
  if (L >= 6 && L < 10) {
    const int t = L - 6;
    const int i = t == 0 ? 1 : 0, j = t <= 1 ? 2 : 1, k = t == 3 ? 2 : 3;
    if (k < n) {
      const V3 s1 = rw_mink(sim + 6*i), s2 = rw_mink(sim + 6*j), s3 = rw_mink(sim + 6*k);
      real w0, w1, w2;
      V3 foot;
      if (rw_plane_foot(foot, s1, s2, s3)) {
        const int es = rw_edge_slot(i, j);
        w0 = tab[2*es]; w1 = tab[2*es + 1]; w2 = 0;
      } else {
        // the coordinate plane in which the triangle's projection is largest (S2D :687)
        const real m23 = s2.y*s3.z - s2.z*s3.y - s1.y*s3.z + s1.z*s3.y + s1.y*s2.z - s1.z*s2.y;
        const real m13 = s2.x*s3.z - s2.z*s3.x - s1.x*s3.z + s1.z*s3.x + s1.x*s2.z - s1.z*s2.x;
        const real m12 = s2.x*s3.y - s2.y*s3.x - s1.x*s3.y + s1.y*s3.x + s1.x*s2.y - s1.y*s2.x;
        const real g1 = fabs(m23), g2 = fabs(m13), g3 = fabs(m12);
        const int drop = (g1 >= g2 && g1 >= g3) ? 0 : (g2 >= g3 ? 1 : 2);
        const real area = drop == 0 ? m23 : (drop == 1 ? m13 : m12);
        const int u = drop == 0 ? 1 : 0, v = drop == 2 ? 1 : 2;
        const real a0 = comp(s1, u), a1 = comp(s1, v), b0 = comp(s2, u), b1 = comp(s2, v), c0 = comp(s3, u), c1 = comp(s3, v);
        const real q0 = comp(foot, u), q1 = comp(foot, v);
        const real ka = q0*b1 + q1*c0 + b0*c1 - q0*c1 - q1*b0 - c0*b1;
        const real kb = q0*c1 + q1*a0 + c0*a1 - q0*a1 - q1*c0 - a0*c1;
        const real kc = q0*a1 + q1*b0 + a0*b1 - q0*b1 - q1*a0 - b0*a1;
        const int ina = rw_sign_match(area, ka), inb = rw_sign_match(area, kb), inc = rw_sign_match(area, kc);
        if (ina && inb && inc) {
          w0 = ka/area; w1 = kb/area; w2 = kc/area;
        } else {
          real nearest = RC_DBLMAX;
          w0 = w1 = w2 = 0;      // (the reference leaves the caller's zeros when no fallback improves: not reachable)
          if (!ina) {
            const int es = rw_edge_slot(j, k);
            const real la = tab[2*es], lb = tab[2*es + 1];
            const V3 x{la*s2.x + lb*s3.x, la*s2.y + lb*s3.y, la*s2.z + lb*s3.z};
            w0 = 0; w1 = la; w2 = lb;
            nearest = dot(x, x);
          }
          if (!inb) {
            const int es = rw_edge_slot(i, k);
            const real la = tab[2*es], lb = tab[2*es + 1];
            const V3 x{la*s1.x + lb*s3.x, la*s1.y + lb*s3.y, la*s1.z + lb*s3.z};
            const real dd = dot(x, x);
            if (dd < nearest) { w0 = la; w1 = 0; w2 = lb; nearest = dd; }
          }
          if (!inc) {
            const int es = rw_edge_slot(i, j);
            const real la = tab[2*es], lb = tab[2*es + 1];
            const V3 x{la*s1.x + lb*s2.x, la*s1.y + lb*s2.y, la*s1.z + lb*s2.z};
            const real dd = dot(x, x);
            if (dd < nearest) { w0 = la; w1 = lb; w2 = 0; }
          }
        }
      }
      tab[12 + 3*t] = w0; tab[12 + 3*t + 1] = w1; tab[12 + 3*t + 2] = w2;
    }
  }
  wv_row_sync();
  if (L == 10) {
    if (n == 2) { tab[24] = tab[0]; tab[25] = tab[1]; tab[26] = 0; tab[27] = 0; }
    else if (n == 3) { tab[24] = tab[21]; tab[25] = tab[22]; tab[26] = tab[23]; tab[27] = 0; }
    else {
      const V3 s1 = rw_mink(sim), s2 = rw_mink(sim + 6), s3 = rw_mink(sim + 12), s4 = rw_mink(sim + 18);
      const real k1 = -rw_det(s2, s3, s4);
      const real k2 = rw_det(s1, s3, s4);
      const real k3 = -rw_det(s1, s2, s4);
      const real k4 = rw_det(s1, s2, s3);
      const real vol = k1 + k2 + k3 + k4;
      const int in1 = rw_sign_match(vol, k1), in2 = rw_sign_match(vol, k2), in3 = rw_sign_match(vol, k3), in4 = rw_sign_match(vol, k4);
      real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
      if (in1 && in2 && in3 && in4) {
        r0 = k1/vol; r1 = k2/vol; r2 = k3/vol; r3 = k4/vol;
      } else {
        real nearest = RC_DBLMAX;
        if (!in1) {
          const real a = tab[12], b = tab[13], c = tab[14];
          const V3 x{a*s2.x + b*s3.x + c*s4.x, a*s2.y + b*s3.y + c*s4.y, a*s2.z + b*s3.z + c*s4.z};
          r0 = 0; r1 = a; r2 = b; r3 = c;
          nearest = dot(x, x);
        }
        if (!in2) {
          const real a = tab[15], b = tab[16], c = tab[17];
          const V3 x{a*s1.x + b*s3.x + c*s4.x, a*s1.y + b*s3.y + c*s4.y, a*s1.z + b*s3.z + c*s4.z};
          const real dd = dot(x, x);
          if (dd < nearest) { r0 = a; r1 = 0; r2 = b; r3 = c; nearest = dd; }
        }
        if (!in3) {
          const real a = tab[18], b = tab[19], c = tab[20];
          const V3 x{a*s1.x + b*s2.x + c*s4.x, a*s1.y + b*s2.y + c*s4.y, a*s1.z + b*s2.z + c*s4.z};
          const real dd = dot(x, x);
          if (dd < nearest) { r0 = a; r1 = b; r2 = 0; r3 = c; nearest = dd; }
        }
        if (!in4) {
          const real a = tab[21], b = tab[22], c = tab[23];
          const V3 x{a*s1.x + b*s2.x + c*s3.x, a*s1.y + b*s2.y + c*s3.y, a*s1.z + b*s2.z + c*s3.z};
          const real dd = dot(x, x);
          if (dd < nearest) { r0 = a; r1 = b; r2 = c; r3 = 0; }
        }
      }
      tab[24] = r0; tab[25] = r1; tab[26] = r2; tab[27] = r3;
    }
  }
  wv_row_sync();
  lam[0] = tab[24]; lam[1] = tab[25]; lam[2] = tab[26]; lam[3] = tab[27];
  wv_row_sync();
}

// ---- the pair a row is working on -------------------------------------------------------------------------------------
struct RowPair {
  RowMem m;
  Shape a, b;
  real tol; int iters;            // opt.ccd_tolerance, opt.ccd_iterations
  int maxcon; real cutoff;        // contacts wanted, distance beyond which the query may stop early
  int nsim, apart, nw, spent;     // simplex size, "shapes are separated", witness pairs found, distance iterations used
  real dist0; V3 w1, w2;          // first witness pair and its signed distance
  int tabled;                     // the witness pairs are in the row's witness table (multi-contact), not in w1 / w2
#ifdef MJH_PROFILE
  long long t_far, t_clip;        // (profiling builds) clock ticks inside support queries / multi-contact of the penetration phase
#endif
  int nv, nf, nm;                 // polytope: vertices, faces, entries of the priority map
  V3 centre;                      // a point inside the polytope (orients the faces)
};
MJH_DEV real* rp_frame(const RowPair& c, int k) { return c.m.R + RO_FRAME + FR_N*k; }
MJH_DEV void rp_take_caches(RowPair& c, const Far& f) { c.a.vcache = f.ia; c.a.gcache = f.ga; c.b.vcache = f.ib; c.b.gcache = f.gb; }
// one lane stores a support result as vertex record (dr, di); the caller synchronises
MJH_DEV void rp_store_vertex(real* dr, int* di, const Far& f) {
  if (rw_l() == 0) {
    dr[0] = f.pa.x; dr[1] = f.pa.y; dr[2] = f.pa.z; dr[3] = f.pb.x; dr[4] = f.pb.y; dr[5] = f.pb.z;
    di[0] = f.ia; di[1] = f.ib;
  }
}
// both shapes polyhedral and without margin: support points come from a finite set (discreteGeoms :179)
MJH_DEV int rp_polyhedral(const RowPair& c) {
  if (c.a.margin != 0 || c.b.margin != 0) return 0;
  return (c.a.type == MJH_GEOM_MESH || c.a.type == MJH_GEOM_BOX) && (c.b.type == MJH_GEOM_MESH || c.b.type == MJH_GEOM_BOX);
}
MJH_DEV int rw_pick4(int s0, int s1, int s2, int s3, int k) { return k == 0 ? s0 : (k == 1 ? s1 : (k == 2 ? s2 : s3)); }

// Does the tetrahedron spanned by the simplex contain the origin?  Refines the tetrahedron (a copy: the simplex is
// only replaced on success) by swapping the vertex opposite the nearest face for a new support point.  One lane per
// face.  1 yes, 0 the shapes are apart, -1 undecided.  (gjkIntersect :420)
MJH_DEV int rc_holds_origin(MREF M, RowPair& c) {
  const RowMem& m = c.m;
  const int L = rw_l();
  real* sim = m.R + RO_SIM; int* sid = m.I + IO_SIM;
  real* tab = m.R + RO_SCR;
  // the working copy lives where the polytope's first vertices will
  if (L < 8) for (int q = 0; q < 4; q++) { if (L < 6) rm_vert(m, q)[L] = sim[6*q + L]; else rm_vid(m, q)[L - 6] = sid[2*q + L - 6]; }
  wv_row_sync();
  int s0 = 0, s1 = 1, s2 = 2, s3 = 3;
  int k = c.spent;
  for (; k < c.iters; k++) {
    if (L < 4) {
      // face L of the (permuted) tetrahedron, wound so that its normal points away from the fourth vertex
      const int ia = L == 0 ? s2 : (L == 2 ? s1 : s0);
      const int ib = L == 0 ? s1 : (L == 1 ? s2 : (L == 2 ? s0 : s1));
      const int ic = L == 3 ? s2 : s3;
      const V3 p = rw_mink(rm_vert(m, ia));
      const V3 e1 = rw_mink(rm_vert(m, ic)) - p, e2 = rw_mink(rm_vert(m, ib)) - p;
      V3 nrm = cross(e1, e2);
      const real n2 = dot(nrm, nrm);
      real sd = RC_DBLMAX;
      if (n2 > RC_TINY2 && n2 < RC_HUGE2) { nrm = rw_scl(nrm, 1/sqrt(n2)); sd = dot(nrm, p); }
      tab[4*L] = sd; tab[4*L + 1] = nrm.x; tab[4*L + 2] = nrm.y; tab[4*L + 3] = nrm.z;
    }
    wv_row_sync();
    const real d0 = tab[0], d1 = tab[4], d2 = tab[8], d3 = tab[12];
    if (!d3 || !d2 || !d1 || !d0) { wv_row_sync(); c.spent = k; return -1; }
    const int lo = (d0 < d1) ? 0 : 1, hi = (d2 < d3) ? 2 : 3;
    const real dlo = lo ? d1 : d0, dhi = hi == 2 ? d2 : d3;
    const int near = (dlo < dhi) ? lo : hi;
    const real dnear = (dlo < dhi) ? dlo : dhi;
    if (dnear > 0) {
      c.nsim = 4;
      wv_row_sync();
      if (L < 8) for (int q = 0; q < 4; q++) {
        const int src = rw_pick4(s0, s1, s2, s3, q);
        if (L < 6) sim[6*q + L] = rm_vert(m, src)[L]; else sid[2*q + L - 6] = rm_vid(m, src)[L - 6];
      }
      wv_row_sync();
      c.spent = k;
      return 1;
    }
    const V3 nrm = ld3(tab + 4*near + 1);
    wv_row_sync();
    const Far f = rc_farthest(M, rp_frame(c, 0), c.a, c.b, nrm, V3{-nrm.x, -nrm.y, -nrm.z});
    rp_take_caches(c, f);
    const int slot = rw_pick4(s0, s1, s2, s3, near);
    rp_store_vertex(rm_vert(m, slot), rm_vid(m, slot), f);
    wv_row_sync();
    if (dot(nrm, f.pa - f.pb) < 0) { c.nsim = 0; c.spent = k; return 0; }
    // exchange the two vertices after `near` (keeps the tetrahedron's orientation)
    const int i = (near + 1) & 3, j = (near + 2) & 3;
    const int vi = rw_pick4(s0, s1, s2, s3, i), vj = rw_pick4(s0, s1, s2, s3, j);
    s0 = i == 0 ? vj : (j == 0 ? vi : s0);
    s1 = i == 1 ? vj : (j == 1 ? vi : s1);
    s2 = i == 2 ? vj : (j == 2 ? vi : s2);
    s3 = i == 3 ? vj : (j == 3 ? vi : s3);
  }
  c.spent = k;
  return -1;
}

// Distance query on the Minkowski difference: on return the simplex (nsim points), the closest points w1 / w2 and
// dist0 (0 when the simplex encloses the origin), or apart = 1.  (gjk :198)
MJH_DEV void rc_distance(MREF M, RowPair& c) {
  const RowMem& m = c.m;
  const int L = rw_l();
  real* sim = m.R + RO_SIM; int* sid = m.I + IO_SIM;
  const int want_dist = c.cutoff > 0;
  int try_containment = !want_dist;
  int n = 0, k = 0;
  real lam[4] = {0, 0, 0, 0};
  const real tol2 = c.tol*c.tol;
  c.apart = 0;
  const int finite = rp_polyhedral(c);
  const real slack = finite ? 0 : 0.5*tol2;
  const real reach = finite ? MJH_MINVAL : c.tol;
  V3 x = c.w1 - c.w2;
  real xlen = rw_len(x), xlen_before = 0;
  RC_COUNT(0);
  for (; k < c.iters; k++) {
    if (xlen < reach || fabs(xlen_before - xlen) < MJH_MINVAL) break;
    RC_COUNT(1);
    const V3 dn = rw_scl(x, 1/xlen);
    const Far f = rc_farthest(M, rp_frame(c, 0), c.a, c.b, rw_scl(dn, -1), dn);
    rp_take_caches(c, f);
    rp_store_vertex(sim + 6*n, sid + 2*n, f);
    wv_row_sync();
    const V3 s = f.pa - f.pb;
    if (dot(x, x - s) < slack) break;
    const real lower = dot(x, s);
    if ((!want_dist && lower > 0) || (want_dist && c.cutoff < RC_DBLMAX && lower > 0 && lower >= c.cutoff*xlen)) {
      c.apart = 1; c.spent = k; c.nsim = 0; c.nw = 0; c.dist0 = RC_DBLMAX;
      return;
    }
    if (n == 3 && try_containment) {
      RC_COUNT(2);
      c.spent = k;
      const int ans = rc_holds_origin(M, c);
      if (ans != -1) {
        c.nw = 0;
        c.apart = ans == 0;
        c.dist0 = ans > 0 ? 0 : RC_DBLMAX;
        return;
      }
      k = c.spent;
      try_containment = 0;
    }
    if (n == 0) { lam[0] = 1; lam[1] = lam[2] = lam[3] = 0; }
    else rc_simplex_weights(m, n + 1, lam);
    // keep the points that carry weight, in order
    if (L < 8) {
      int to = 0;
      for (int i = 0; i < 4; i++) {
        if (!lam[i]) continue;
        if (to != i) { if (L < 6) sim[6*to + L] = sim[6*i + L]; else sid[2*to + L - 6] = sid[2*i + L - 6]; }
        to++;
      }
    }
    n = 0;
    for (int i = 0; i < 4; i++) if (lam[i]) lam[n++] = lam[i];
    wv_row_sync();
    if (n < 1) {
      c.spent = k; c.nsim = 0; c.nw = 0; c.dist0 = RC_DBLMAX; c.apart = 1;
      return;
    }
    {
      const V3 p0 = rw_mink(sim), p1 = rw_mink(sim + 6), p2 = rw_mink(sim + 12), p3 = rw_mink(sim + 18);
      if (n == 1) x = V3{lam[0]*p0.x, lam[0]*p0.y, lam[0]*p0.z};
      else if (n == 2) x = V3{lam[0]*p0.x + lam[1]*p1.x, lam[0]*p0.y + lam[1]*p1.y, lam[0]*p0.z + lam[1]*p1.z};
      else if (n == 3) x = V3{lam[0]*p0.x + lam[1]*p1.x + lam[2]*p2.x, lam[0]*p0.y + lam[1]*p1.y + lam[2]*p2.y,
                              lam[0]*p0.z + lam[1]*p1.z + lam[2]*p2.z};
      else x = V3{lam[0]*p0.x + lam[1]*p1.x + lam[2]*p2.x + lam[3]*p3.x, lam[0]*p0.y + lam[1]*p1.y + lam[2]*p2.y + lam[3]*p3.y,
                  lam[0]*p0.z + lam[1]*p1.z + lam[2]*p2.z + lam[3]*p3.z};
    }
    xlen_before = xlen;
    xlen = rw_len(x);
    if (n == 4) break;
  }
  if (n > 0) {
    // the same combination of the points on A and of the points on B
    V3 acc1{0, 0, 0}, acc2{0, 0, 0};
    for (int side = 0; side < 2; side++) {
      const real* p = sim + 3*side;
      V3 r;
      if (n == 1) r = V3{lam[0]*p[0], lam[0]*p[1], lam[0]*p[2]};
      else if (n == 2) r = V3{lam[0]*p[0] + lam[1]*p[6], lam[0]*p[1] + lam[1]*p[7], lam[0]*p[2] + lam[1]*p[8]};
      else if (n == 3) r = V3{lam[0]*p[0] + lam[1]*p[6] + lam[2]*p[12], lam[0]*p[1] + lam[1]*p[7] + lam[2]*p[13],
                              lam[0]*p[2] + lam[1]*p[8] + lam[2]*p[14]};
      else r = V3{lam[0]*p[0] + lam[1]*p[6] + lam[2]*p[12] + lam[3]*p[18], lam[0]*p[1] + lam[1]*p[7] + lam[2]*p[13] + lam[3]*p[19],
                  lam[0]*p[2] + lam[1]*p[8] + lam[2]*p[14] + lam[3]*p[20]};
      if (side == 0) acc1 = r; else acc2 = r;
    }
    c.w1 = acc1; c.w2 = acc2;
  }
  // one more support query along x: are the shapes apart after all?
  {
    const V3 dn = rw_scl(x, 1/xlen);
    const Far f = rc_farthest(M, rp_frame(c, 0), c.a, c.b, rw_scl(dn, -1), dn);
    rp_take_caches(c, f);
    if (dot(x, f.pa - f.pb) > 0) c.apart = 1;
  }
  c.nw = 1;
  c.spent = k;
  c.nsim = n;
  c.dist0 = (n == 4 && !c.apart) ? 0 : xlen;
}

// ---- expanding polytope (engine_collision_gjk.c:882-1500) --------------------------------------------------------------
MJH_DEV V3 rp_point(const RowPair& c, int v) { return rw_mink(rm_vert(c.m, v)); }
MJH_DEV int rp_face_vert(const RowPair& c, int f, int k) { return (rm_fi(c.m, FI_VERTS, f) >> (10*k)) & 0x3FF; }
MJH_DEV V3 rp_face_normal(const RowPair& c, int f) { return V3{rm_fr(c.m, 0, f), rm_fr(c.m, 1, f), rm_fr(c.m, 2, f)}; }

// face f = (va, vb, vc) with neighbours (n0, n1, n2) across its three edges; its plane's foot point of the origin is
// stored as the face vector.  Returns the squared distance of the plane (0: degenerate).  (attachFace :1238)
MJH_DEV real rp_make_face(const RowPair& c, int f, int va, int vb, int vc, int n0, int n1, int n2) {
  const RowMem& m = c.m;
  rm_fi(m, FI_VERTS, f) = va + (vb << 10) + (vc << 20);
  rm_fi(m, FI_ADJ0, f) = n0; rm_fi(m, FI_ADJ1, f) = n1; rm_fi(m, FI_ADJ2, f) = n2;
  rm_fi(m, FI_SLOT, f) = FS_UNMAPPED;
  V3 fv;
  if (rw_plane_foot(fv, rp_point(c, vc), rp_point(c, vb), rp_point(c, va))) return 0;
  if (dot(fv, rp_point(c, va) - c.centre) < 0) fv = rw_scl(fv, -1);
  const real d2 = dot(fv, fv);
  rm_fr(m, 0, f) = fv.x; rm_fr(m, 1, f) = fv.y; rm_fr(m, 2, f) = fv.z; rm_fr(m, 3, f) = d2;
  return d2;
}
// polytope vertices 0..n-1 := simplex points 0..n-1
MJH_DEV void rp_adopt_simplex(RowPair& c, int n) {
  const int L = rw_l();
  const real* sim = c.m.R + RO_SIM; const int* sid = c.m.I + IO_SIM;
  if (L < 8) for (int q = 0; q < n; q++) { if (L < 6) rm_vert(c.m, q)[L] = sim[6*q + L]; else rm_vid(c.m, q)[L - 6] = sid[2*q + L - 6]; }
  c.nv = n;
  wv_row_sync();
}
// simplex := the triangle (va, vb, vc) of the polytope; the polytope is emptied (replaceSimplex3 :1020)
MJH_DEV void rp_restart_from_triangle(RowPair& c, int va, int vb, int vc) {
  const int L = rw_l();
  real* sim = c.m.R + RO_SIM; int* sid = c.m.I + IO_SIM;
  wv_row_sync();
  if (L < 8) for (int q = 0; q < 3; q++) {
    const int src = q == 0 ? va : (q == 1 ? vb : vc);
    if (L < 6) sim[6*q + L] = rm_vert(c.m, src)[L]; else sid[2*q + L - 6] = rm_vid(c.m, src)[L - 6];
  }
  c.nsim = 3; c.nf = 0; c.nv = 0; c.nm = 0;
  wv_row_sync();
}
// support point along d (length dlen) appended to the polytope (epaSupport :384)
MJH_DEV int rp_grow(MREF M, RowPair& c, V3 d, real dlen, Far& f) {
  V3 dir{1, 0, 0}, dn{-1, 0, 0};
  if (dlen > MJH_MINVAL) { dir = V3{d.x/dlen, d.y/dlen, d.z/dlen}; dn = rw_scl(dir, -1); }
#ifdef MJH_PROFILE
  const long long t0_ = wv_clock();
#endif
  f = rc_farthest(M, rp_frame(c, 0), c.a, c.b, dir, dn);
#ifdef MJH_PROFILE
  c.t_far += wv_clock() - t0_;
#endif
  rp_take_caches(c, f);
  const int v = c.nv++;
  rp_store_vertex(rm_vert(c.m, v), rm_vid(c.m, v), f);
  wv_row_sync();
  return v;
}
// the initial faces: entry f of a table packs (va, vb, vc, n0, n1, n2), 3 bits each, built by lane f.  Returns the
// lowest face whose plane (numerically) holds the origin, -1 when all are fine; the priority map is then 0..nfaces-1.
#define RC_FACE(va, vb, vc, n0, n1, n2) ((va) | ((vb) << 3) | ((vc) << 6) | ((n0) << 9) | ((n1) << 12) | ((n2) << 15))
MJH_DEV int rp_seed_faces(RowPair& c, int nfaces, int entry) {
  const int L = rw_l();
  real d2 = 1;
  if (L < nfaces)
    d2 = rp_make_face(c, L, entry & 7, (entry >> 3) & 7, (entry >> 6) & 7, (entry >> 9) & 7, (entry >> 12) & 7, (entry >> 15) & 7);
  const unsigned bad = wv_row_ballot(L < nfaces && d2 < RC_TINY2);
  wv_row_sync();
  if (bad) return __builtin_ctz(bad);
  if (L < nfaces) { rm_map(c.m, L) = L; rm_fi(c.m, FI_SLOT, L) = L; }
  c.nf = nfaces; c.nm = nfaces;
  wv_row_sync();
  return -1;
}
MJH_DEV int rw_table6(int k, int e0, int e1, int e2, int e3, int e4, int e5) {
  return k == 0 ? e0 : (k == 1 ? e1 : (k == 2 ? e2 : (k == 3 ? e3 : (k == 4 ? e4 : e5))));
}
// is the origin on the same side of plane (p0 p1 p2) as p3?
MJH_DEV int rw_same_side(V3 p0, V3 p1, V3 p2, V3 p3) {
  const V3 n = cross(p1 - p0, p2 - p0);
  const real s3 = dot(n, p3 - p0);
  const real so = dot(n, rw_scl(p0, -1));
  return (s3 > 0 && so > 0) || (s3 < 0 && so < 0);
}
MJH_DEV int rw_tetra_holds_origin(V3 p0, V3 p1, V3 p2, V3 p3) {
  return rw_same_side(p0, p1, p2, p3) && rw_same_side(p1, p2, p3, p0) && rw_same_side(p2, p3, p0, p1) && rw_same_side(p3, p0, p1, p2);
}
// affine coordinates of p in the plane of triangle (v1 v2 v3), via the largest coordinate projection (triAffineCoord :1033)
MJH_DEV void rw_affine(real* l, V3 v1, V3 v2, V3 v3, V3 p) {
  const real m23 = v2.y*v3.z - v2.z*v3.y - v1.y*v3.z + v1.z*v3.y + v1.y*v2.z - v1.z*v2.y;
  const real m13 = v2.x*v3.z - v2.z*v3.x - v1.x*v3.z + v1.z*v3.x + v1.x*v2.z - v1.z*v2.x;
  const real m12 = v2.x*v3.y - v2.y*v3.x - v1.x*v3.y + v1.y*v3.x + v1.x*v2.y - v1.y*v2.x;
  const real g1 = fabs(m23), g2 = fabs(m13), g3 = fabs(m12);
  const int drop = (g1 >= g2 && g1 >= g3) ? 0 : (g2 >= g3 ? 1 : 2);
  const real area = drop == 0 ? m23 : (drop == 1 ? m13 : m12);
  const int u = drop == 0 ? 1 : 0, v = drop == 2 ? 1 : 2;
  const real px = comp(p, u), py = comp(p, v);
  const real ax = comp(v1, u), ay = comp(v1, v), bx = comp(v2, u), by = comp(v2, v), cx = comp(v3, u), cy = comp(v3, v);
  const real ka = px*by + py*cx + bx*cy - px*cy - py*bx - cx*by;
  const real kb = px*cy + py*ax + cx*ay - px*ay - py*cx - ax*cy;
  const real kc = px*ay + py*bx + ax*by - px*by - py*ax - bx*ay;
  l[0] = ka/area; l[1] = kb/area; l[2] = kc/area;
}
MJH_DEV int rw_on_triangle(V3 v1, V3 v2, V3 v3, V3 p) {
  real l[3];
  rw_affine(l, v1, v2, v3, p);
  if (l[0] < 0 || l[1] < 0 || l[2] < 0) return 0;
  const V3 pr{v1.x*l[0] + v2.x*l[1] + v3.x*l[2], v1.y*l[0] + v2.y*l[1] + v3.y*l[2], v1.z*l[0] + v2.z*l[1] + v3.z*l[2]};
  return rw_len(pr - p) < MJH_MINVAL;
}

// Triangle simplex -> double tetrahedron (two support points along +- the triangle's normal).  0 = built.  (polytope3 :1083)
MJH_DEV int rc_polytope_from_triangle(MREF M, RowPair& c) {
  const real* sim = c.m.R + RO_SIM;
  const V3 v1 = rw_mink(sim), v2 = rw_mink(sim + 6), v3 = rw_mink(sim + 12);
  c.centre = rw_scl((v1 + v2) + v3, 1.0/3.0);
  const V3 n = cross(v2 - v1, v3 - v1);
  const real nlen = rw_len(n);
  if (nlen < MJH_MINVAL) return 5;
  rp_adopt_simplex(c, 3);
  Far f;
  const int below = rp_grow(M, c, rw_scl(n, -1), nlen, f);
  const int above = rp_grow(M, c, n, nlen, f);
  const V3 v4 = rp_point(c, above), v5 = rp_point(c, below);
  if (rw_on_triangle(v1, v2, v3, v4)) return 6;
  if (rw_on_triangle(v1, v2, v3, v5)) return 7;
  if (c.dist0 > 10*MJH_MINVAL && !rw_tetra_holds_origin(v1, v2, v3, v4) && !rw_tetra_holds_origin(v1, v2, v3, v5)) return 8;
  // (above = 4, below = 3)
  const int entry = rw_table6(rw_l(), RC_FACE(4, 0, 1, 1, 3, 2), RC_FACE(4, 2, 0, 2, 4, 0), RC_FACE(4, 1, 2, 0, 5, 1),
                              RC_FACE(3, 1, 0, 5, 0, 4), RC_FACE(3, 0, 2, 3, 1, 5), RC_FACE(3, 2, 1, 4, 2, 3));
  if (rp_seed_faces(c, 6, entry) >= 0) return 9;
  return 0;
}
// Segment simplex -> double tetrahedron around the segment (three support points 120 degrees apart).  (polytope2 :948)
MJH_DEV int rc_polytope_from_segment(MREF M, RowPair& c) {
  const real* sim = c.m.R + RO_SIM;
  const V3 v1 = rw_mink(sim), v2 = rw_mink(sim + 6);
  c.centre = rw_scl(v1 + v2, 0.5);
  const V3 axis = v2 - v1;
  real least = RC_DBLMAX;
  int thin = 0;
  for (int i = 0; i < 3; i++) if (fabs(comp(axis, i)) < least) { least = fabs(comp(axis, i)); thin = i; }
  const V3 d1 = cross(with_comp(V3{0, 0, 0}, thin, 1), axis);
  // rotation by 120 degrees about the segment (rotmat :915)
  real R[9];
  {
    const real len = rw_len(axis);
    const real u1 = axis.x/len, u2 = axis.y/len, u3 = axis.z/len;
    const real sn = 0.86602540378, cs = -0.5;
    R[0] = cs + u1*u1*(1 - cs);
    R[1] = u1*u2*(1 - cs) - u3*sn;
    R[2] = u1*u3*(1 - cs) + u2*sn;
    R[3] = u2*u1*(1 - cs) + u3*sn;
    R[4] = cs + u2*u2*(1 - cs);
    R[5] = u2*u3*(1 - cs) - u1*sn;
    R[6] = u1*u3*(1 - cs) - u2*sn;
    R[7] = u2*u3*(1 - cs) + u1*sn;
    R[8] = cs + u3*u3*(1 - cs);
  }
  const V3 d2 = mmul(R, d1);
  const V3 d3 = mmul(R, d2);
  rp_adopt_simplex(c, 2);
  Far f;
  const int i3 = rp_grow(M, c, d1, rw_len(d1), f);
  const int i4 = rp_grow(M, c, d2, rw_len(d2), f);
  const int i5 = rp_grow(M, c, d3, rw_len(d3), f);
  const V3 v3 = rp_point(c, i3), v4 = rp_point(c, i4), v5 = rp_point(c, i5);
  const int entry = rw_table6(rw_l(), RC_FACE(0, 2, 3, 1, 3, 2), RC_FACE(0, 4, 2, 2, 4, 0), RC_FACE(0, 3, 4, 0, 5, 1),
                              RC_FACE(1, 3, 2, 5, 0, 4), RC_FACE(1, 2, 4, 3, 1, 5), RC_FACE(1, 4, 3, 4, 2, 3));
  const int bad = rp_seed_faces(c, 6, entry);
  if (bad >= 0) {
    const int e = rw_table6(bad, RC_FACE(0, 2, 3, 0, 0, 0), RC_FACE(0, 4, 2, 0, 0, 0), RC_FACE(0, 3, 4, 0, 0, 0),
                            RC_FACE(1, 3, 2, 0, 0, 0), RC_FACE(1, 2, 4, 0, 0, 0), RC_FACE(1, 4, 3, 0, 0, 0));
    rp_restart_from_triangle(c, e & 7, (e >> 3) & 7, (e >> 6) & 7);
    return rc_polytope_from_triangle(M, c);
  }
  // the three new points must wind around the segment (rayTriangle :932)
  const V3 e12 = v2 - v1, e13 = v3 - v1, e14 = v4 - v1, e15 = v5 - v1;
  const real vol1 = rw_det(e13, e14, e12), vol2 = rw_det(e14, e15, e12), vol3 = rw_det(e15, e13, e12);
  if (!((vol1 >= 0 && vol2 >= 0 && vol3 >= 0) || (vol1 <= 0 && vol2 <= 0 && vol3 <= 0))) return 2;
  return 0;
}
// Tetrahedron simplex -> the polytope itself.  (polytope4 :1167)
MJH_DEV int rc_polytope_from_tetrahedron(MREF M, RowPair& c) {
  rp_adopt_simplex(c, 4);
  const V3 p0 = rp_point(c, 0), p1 = rp_point(c, 1), p2 = rp_point(c, 2), p3 = rp_point(c, 3);
  c.centre = rw_scl(((p0 + p1) + p2) + p3, 0.25);
  const int L = rw_l();
  const int entry = L == 0 ? RC_FACE(0, 1, 2, 1, 3, 2) : (L == 1 ? RC_FACE(0, 3, 1, 2, 3, 0) : (L == 2 ? RC_FACE(0, 2, 3, 0, 3, 1) : RC_FACE(3, 2, 1, 2, 0, 1)));
  const int bad = rp_seed_faces(c, 4, entry);
  if (bad >= 0) {
    const int e = bad == 0 ? RC_FACE(0, 1, 2, 0, 0, 0) : (bad == 1 ? RC_FACE(0, 3, 1, 0, 0, 0) : (bad == 2 ? RC_FACE(0, 2, 3, 0, 0, 0) : RC_FACE(3, 2, 1, 0, 0, 0)));
    rp_restart_from_triangle(c, e & 7, (e >> 3) & 7, (e >> 6) & 7);
    return rc_polytope_from_triangle(M, c);
  }
  if (!rw_tetra_holds_origin(p0, p1, p2, p3)) return 10;
  return 0;
}

// remove face f from the priority map (its slot is taken by the last entry); one lane
MJH_DEV void rp_unmap(const RowMem& m, int f, int& nm) {
  const int slot = rm_fi(m, FI_SLOT, f);
  if (slot >= 0) {
    const int last = rm_map(m, --nm);
    rm_map(m, slot) = last;
    rm_fi(m, FI_SLOT, last) = slot;
  }
  rm_fi(m, FI_SLOT, f) = FS_DELETED;
}
// Silhouette of the faces visible from the new vertex (their FI_VIS flags are set), starting at `start`: visible faces
// are retired from the map in the order a depth-first flood over the face adjacency meets them, silhouette edges are
// listed in the order it leaves the visible region.  The flood runs on an explicit stack of edge crossings (face, edge)
// -- crossing = "step over edge `edge` of `face` into its neighbour" -- popped last-in-first-out, so a crossing is
// judged when the flood would get to it.  One lane; returns the number of silhouette edges.  (horizon :1285-1336)
MJH_DEV int rp_silhouette(const RowPair& c, int start, int& nm, int cap) {
  const RowMem& m = c.m;
  int sp = 0, nh = 0;
  rp_unmap(m, start, nm);
  rm_stk(m, sp++) = start*4 + 2; rm_stk(m, sp++) = start*4 + 1; rm_stk(m, sp++) = start*4 + 0;
  while (sp > 0) {
    const int cr = rm_stk(m, --sp);
    const int from = cr >> 2, e = cr & 3;
    const int into = rm_fi(m, FI_ADJ0 + e, from);
    if (rm_fi(m, FI_SLOT, into) == FS_DELETED) continue;
    // the edge as the neighbour numbers it: the one that starts at the far end vertex of edge e
    const int pivot = rp_face_vert(c, from, e == 2 ? 0 : e + 1);
    const int ie = rp_face_vert(c, into, 0) == pivot ? 0 : (rp_face_vert(c, into, 1) == pivot ? 1 : 2);
    if (rm_fi(m, FI_VIS, into)) {
      rp_unmap(m, into, nm);
      rm_stk(m, sp++) = into*4 + (ie + 2)%3;
      rm_stk(m, sp++) = into*4 + (ie + 1)%3;
    } else {
      if (nh < cap) { rm_hz(m, nh) = into; (&rm_hz(m, nh))[1] = ie; }
      nh++;
    }
  }
  return nh;
}

// Expanding polytope: returns the face closest to the origin when the expansion stops, -1 on failure; the witness
// pair of that face goes to w1 / w2 / dist0.  (epa :1358)
MJH_DEV int rc_expand(MREF M, RowPair& c) {
  const RowMem& m = c.m;
  const int L = rw_l();
  real upper = RC_DBLMAX, upper2 = RC_DBLMAX, lower2;
  int face = -1;
  const int finite = rp_polyhedral(c);
  const real enough = finite ? MJH_MINVAL : c.tol;
  const int rounds = c.iters < 1000 ? c.iters : 1000;
  const int maxfaces = 6*c.iters;
  for (int k = 0; k < rounds; k++) {
    const int before = face;
    lower2 = RC_DBLMAX;
    // the mapped face nearest to the origin (first of equals in map order)
    for (int i0 = 0; i0 < c.nm; i0 += 16) {
      const int i = i0 + L;
      real v = i < c.nm ? rm_fr(m, 3, rm_map(m, i)) : HUGE_VAL;
      int at = i < c.nm ? i : RC_NONE;
      rw_first_min<4>(v, at);
      if (v < lower2) { lower2 = v; face = rm_map(m, at); }
    }
    if (lower2 > upper2 || face < 0) { face = before; break; }
    if (lower2 <= 0) break;
    RC_COUNT(4);
    const real lower = sqrt(lower2);
    const V3 fv = rp_face_normal(c, face);
    Far f;
    const int wi = rp_grow(M, c, fv, lower, f);
    const V3 w = f.pa - f.pb;
    const real upper_k = dot(fv, w)/lower;
    if (upper_k < upper) { upper = upper_k; upper2 = upper*upper; }
    if (upper - lower < enough) {
      if (k == 0 && upper < lower - 1e-10) face = -1;
      break;
    }
    if (finite) {
      // a support point met before: the polytope cannot grow any more
      int seen = 0;
      for (int i0 = 0; i0 < c.nv - 1; i0 += 16) {
        const int i = i0 + L;
        seen |= wv_row_ballot(i < c.nv - 1 && rm_vid(m, i)[0] == f.ia && rm_vid(m, i)[1] == f.ib) != 0;
      }
      if (seen) break;
    }
    // which faces does the new vertex see?
    for (int f0 = 0; f0 < c.nf; f0 += 16) {
      const int q = f0 + L;
      if (q < c.nf) rm_fi(m, FI_VIS, q) = rm_fi(m, FI_SLOT, q) > FS_DELETED && (dot(rp_face_normal(c, q), w) - rm_fr(m, 3, q) > MJH_MINVAL);
    }
    wv_row_sync();
    int nh = 0, nm = c.nm;
    if (L == 0) nh = rp_silhouette(c, face, nm, maxfaces);
    wv_row_sync();
    nh = wv_row_get_i(nh, 0);
    c.nm = wv_row_get_i(nm, 0);
    if (nh < 3) { face = -1; break; }
    if (nh > maxfaces - c.nf || nh > maxfaces) break;
    // the cone of new faces over the silhouette, one edge per lane; faces whose plane distance lies between the bounds
    // join the map in edge order; a face through the origin ends the expansion
    const int base = c.nf;
    int failed = 0;
    for (int i0 = 0; i0 < nh && !failed; i0 += 16) {
      const int i = i0 + L;
      real d2 = 1;
      if (i < nh) {
        const int hf = rm_hz(m, i), he = (&rm_hz(m, i))[1];
        const int v1 = rp_face_vert(c, hf, he), v2 = rp_face_vert(c, hf, he == 2 ? 0 : he + 1);
        rm_fi(m, FI_ADJ0 + he, hf) = base + i;
        d2 = rp_make_face(c, base + i, wi, v2, v1, i == 0 ? base + nh - 1 : base + i - 1, hf, base + (i + 1 == nh ? 0 : i + 1));
      }
      const unsigned zero = wv_row_ballot(i < nh && d2 == 0);
      const unsigned live = zero ? ((1u << __builtin_ctz(zero)) - 1) : 0xffffu;
      const unsigned keep = wv_row_ballot(i < nh && d2 >= lower2 && d2 <= upper2) & live;
      if ((keep >> L) & 1) {
        const int slot = c.nm + rw_popc_below(keep, L);
        rm_map(m, slot) = base + i;
        rm_fi(m, FI_SLOT, base + i) = slot;
      }
      c.nm += __builtin_popcount(keep);
      if (zero) failed = 1;
    }
    c.nf += nh;
    wv_row_sync();
    if (failed) { face = -1; break; }
    if (!c.nm || face < 0) break;
  }
  if (face >= 0) {
    // witness points: the face's foot point in affine coordinates of its three vertices (epaWitness :1339)
    const int va = rp_face_vert(c, face, 0), vb = rp_face_vert(c, face, 1), vc = rp_face_vert(c, face, 2);
    real l[3];
    rw_affine(l, rp_point(c, va), rp_point(c, vb), rp_point(c, vc), rp_face_normal(c, face));
    const real* pa = rm_vert(m, va); const real* pb = rm_vert(m, vb); const real* pc = rm_vert(m, vc);
    c.w1 = V3{l[0]*pa[0] + l[1]*pb[0] + l[2]*pc[0], l[0]*pa[1] + l[1]*pb[1] + l[2]*pc[1], l[0]*pa[2] + l[1]*pb[2] + l[2]*pc[2]};
    c.w2 = V3{l[0]*pa[3] + l[1]*pb[3] + l[2]*pc[3], l[0]*pa[4] + l[1]*pb[4] + l[2]*pc[4], l[0]*pa[5] + l[1]*pb[5] + l[2]*pc[5]};
    c.dist0 = -sqrt(rm_fr(m, 3, face));
    c.nw = 1;
  } else {
    c.nw = 0;
    c.dist0 = 0;
  }
  return face;
}

// ---- multi-contact recovery for polyhedral pairs (engine_collision_gjk.c:1503-2310) ------------------------------------
// buffers laid over the (finished) polytope: candidate normals of A and B, edge end points, the two faces, two
// polygon buffers for the clipping passes; face indices of the normals
struct ClipMem { real* n1; real* n2; real* ev; real* f1; real* f2; real* pa; real* pb; int* id1; int* id2; int P, D; };
MJH_DEV ClipMem rp_clip_mem(MREF M, const RowPair& c) {
  const int P = M.s.ccd_P, D = M.s.ccd_D;
  real* r = c.m.R + RO_POLY;
  int* ip = c.m.I + IO_POLY;
  return ClipMem{r, r + 3*D, r + 6*D, r + 9*D, r + 9*D + 3*P, r + 9*D + 6*P, r + 9*D + 12*P, ip, ip + D, P, D};
}
MJH_DEV V3 rw_poly_normal(MREF M, const real* frame, int mesh, int poly) {
  const int base = 3*(M.mesh_polyadr[mesh] + poly);
  return rw_rot(frame + FR_MAT, M.mesh_polynormal[base], M.mesh_polynormal[base + 1], M.mesh_polynormal[base + 2]);
}
// is face id `x` in the polygon list [adr, adr + n) of a mesh vertex?
MJH_DEV int rw_in_polymap(MREF M, int x, int adr, int n) {
  int hit = 0;
  for (int j = 0; j < n; j++) hit |= M.mesh_polymap[adr + j] == x;
  return hit;
}
// Face normals of a mesh around the feature spanned by `dim` distinct vertices vi[]: the face through three vertices,
// the (up to two) faces through an edge, every face around a vertex.  Lanes scan a vertex's face list.  (meshNormals :1787)
MJH_DEV int rc_mesh_normals(MREF M, const real* frame, int mesh, real* res, int* resind, int dim, const int* vi, int D) {
  const int L = rw_l();
  const int vadr = M.mesh_vertadr[mesh];
  const int a1 = M.mesh_polymapadr[vadr + vi[0]], n1 = M.mesh_polymapnum[vadr + vi[0]];
  if (dim == 1) {
    const int n = n1 < D ? n1 : D;
    for (int i = L; i < n; i += 16) {
      const int poly = M.mesh_polymap[a1 + i];
      st3(res + 3*i, rw_poly_normal(M, frame, mesh, poly));
      resind[i] = poly;
    }
    wv_row_sync();
    return n;
  }
  // faces shared by vertices 0 and 1: the first two members of vertex 0's list that vertex 1 lists too
  const int a2 = M.mesh_polymapadr[vadr + vi[1]], n2 = M.mesh_polymapnum[vadr + vi[1]];
  int shared[2] = {0, 0}, ns = 0;
  for (int i0 = 0; i0 < n1 && ns < 2; i0 += 16) {
    const int i = i0 + L;
    const int poly = i < n1 ? M.mesh_polymap[a1 + i] : -1;
    unsigned hits = wv_row_ballot(i < n1 && rw_in_polymap(M, poly, a2, n2));
    while (hits && ns < 2) {
      const int at = __builtin_ctz(hits);
      hits &= hits - 1;
      shared[ns++] = M.mesh_polymap[a1 + i0 + at];
    }
  }
  if (ns == 0) return 0;
  if (dim == 2) {
    if (L < ns) { st3(res + 3*L, rw_poly_normal(M, frame, mesh, shared[L == 0 ? 0 : 1])); resind[L] = shared[L == 0 ? 0 : 1]; }
    wv_row_sync();
    return ns;
  }
  // dim == 3: of those, the first that vertex 2 lists as well
  const int a3 = M.mesh_polymapadr[vadr + vi[2]], n3 = M.mesh_polymapnum[vadr + vi[2]];
  int pick = -1;
  if (rw_in_polymap(M, shared[0], a3, n3)) pick = shared[0];
  else if (ns > 1 && rw_in_polymap(M, shared[1], a3, n3)) pick = shared[1];
  if (pick < 0) return 0;
  if (L == 0) { st3(res, rw_poly_normal(M, frame, mesh, pick)); resind[0] = pick; }
  wv_row_sync();
  return 1;
}
// the box face whose outward axis is within the alignment tolerance of direction n; six lanes, lowest face wins (boxNormals2 :1898)
MJH_DEV int rc_box_face_along(const real* frame, real* res, int* resind, V3 n) {
  const int L = rw_l();
  const real* mat = frame + FR_MAT;
  V3 ln = rw_to_local(mat, n);
  ln = rw_scl(ln, 1/sqrt(dot(ln, ln)));
  const V3 axis = with_comp(V3{0, 0, 0}, (L >> 1) % 3, (L & 1) ? -1 : 1);
  const unsigned hits = wv_row_ballot(L < 6 && dot(ln, axis) > RC_FACE_ALIGN);
  if (!hits) return 0;
  const int i = __builtin_ctz(hits);
  if (L == i) { st3(res, rw_rot(mat, axis.x, axis.y, axis.z)); resind[0] = i; }
  wv_row_sync();
  return 1;
}
// Face normals of a box around the feature spanned by `dim` distinct corners (corner id: bit k set = +size[k]).  (boxNormals :1924)
MJH_DEV int rc_box_normals(const real* frame, real* res, int* resind, int dim, const int* vi, V3 toward) {
  const int L = rw_l();
  const real* mat = frame + FR_MAT;
  const int v1 = vi[0], v2 = vi[1], v3 = vi[2];
  if (dim == 3) {
    // the axes on which all three corners agree
    const int x = ((v1 & 1) && (v2 & 1) && (v3 & 1)) - (!(v1 & 1) && !(v2 & 1) && !(v3 & 1));
    const int y = ((v1 & 2) && (v2 & 2) && (v3 & 2)) - (!(v1 & 2) && !(v2 & 2) && !(v3 & 2));
    const int z = ((v1 & 4) && (v2 & 4) && (v3 & 4)) - (!(v1 & 4) && !(v2 & 4) && !(v3 & 4));
    int cn = 0, first = 0;
    if (x) { first = 0; cn++; }
    if (y) { if (!cn) first = 2; cn++; }
    if (z) { if (!cn) first = 4; cn++; }
    if (x + y + z == -1) first++;
    if (L == 0) { st3(res, rw_rot(mat, x, y, z)); resind[0] = first; }
    wv_row_sync();
    return cn == 1 ? 1 : rc_box_face_along(frame, res, resind, toward);
  }
  if (dim == 2) {
    const int x = ((v1 & 1) && (v2 & 1)) - (!(v1 & 1) && !(v2 & 1));
    const int y = ((v1 & 2) && (v2 & 2)) - (!(v1 & 2) && !(v2 & 2));
    const int z = ((v1 & 4) && (v2 & 4)) - (!(v1 & 4) && !(v2 & 4));
    int cn = 0;
    if (L == 0) {
      if (x) { st3(res, rw_rot(mat, x, 0, 0)); resind[cn++] = (x > 0) ? 0 : 1; }
      if (y) { st3(res + 3*cn, rw_rot(mat, 0, y, 0)); resind[cn++] = (y > 0) ? 2 : 3; }
      if (z) { st3(res + 3, rw_rot(mat, 0, 0, z)); resind[cn++] = (z > 0) ? 4 : 5; }
    }
    cn = (x != 0) + (y != 0) + (z != 0);
    wv_row_sync();
    return cn == 2 ? 2 : rc_box_face_along(frame, res, resind, toward);
  }
  if (dim == 1) {
    if (L < 3) {
      const int up = (v1 >> L) & 1;
      st3(res + 3*L, rw_rot(mat, L == 0 ? (up ? 1 : -1) : 0, L == 1 ? (up ? 1 : -1) : 0, L == 2 ? (up ? 1 : -1) : 0));
      resind[L] = 2*L + (up ? 0 : 1);
    }
    wv_row_sync();
    return 3;
  }
  return 0;
}
// Unit directions of the edges that leave a feature vertex, and their far end points: the feature edge itself
// (dim 2), else every edge at the vertex -- of a mesh: the predecessor of the vertex in each adjacent face (one face
// per lane); of a box: its three edges.  (meshEdgeNormals :1852, boxEdgeNormals :1973)
MJH_DEV int rc_edge_dirs(MREF M, const real* frame, int type, int mesh, real* res, real* ends, int dim, V3 p1, V3 p2, int v1i, int D) {
  const int L = rw_l();
  if (dim == 2) {
    if (L == 0) {
      st3(ends, p2);
      V3 r = p2 - p1;
      unitize(r);
      st3(res, r);
    }
    wv_row_sync();
    return 1;
  }
  if (dim != 1) return 0;
  const real* mat = frame + FR_MAT; const real* pos = frame + FR_POS; const real* size = frame + FR_SIZE;
  if (type == MJH_GEOM_BOX) {
    if (L < 3) {
      const real x = (v1i & 1) ? size[0] : -size[0];
      const real y = (v1i & 2) ? size[1] : -size[1];
      const real z = (v1i & 4) ? size[2] : -size[2];
      const V3 far = rw_rot_add(mat, pos, L == 0 ? -x : x, L == 1 ? -y : y, L == 2 ? -z : z);
      st3(ends + 3*L, far);
      V3 r = far - p1;
      unitize(r);
      st3(res + 3*L, r);
    }
    wv_row_sync();
    return 3;
  }
  const int vadr = M.mesh_vertadr[mesh], padr = M.mesh_polyadr[mesh];
  const int a1 = M.mesh_polymapadr[vadr + v1i];
  const int n1r = M.mesh_polymapnum[vadr + v1i];
  const int n1 = n1r < D ? n1r : D;
  for (int i = L; i < n1; i += 16) {
    const int poly = M.mesh_polymap[a1 + i];
    const int adr = M.mesh_polyvertadr[padr + poly], nvert = M.mesh_polyvertnum[padr + poly];
    int at = -1;
    for (int j = nvert - 1; j >= 0; j--) if (M.mesh_polyvert[adr + j] == v1i) at = j;       // first occurrence
    if (at >= 0) {
      const int k = (at == 0) ? nvert - 1 : at - 1;
      const int vb = 3*(vadr + M.mesh_polyvert[adr + k]);
      const V3 far = rw_rot_add(mat, pos, M.mesh_vert[vb], M.mesh_vert[vb + 1], M.mesh_vert[vb + 2]);
      st3(ends + 3*i, far);
      V3 r = far - p1;
      unitize(r);
      st3(res + 3*i, r);
    }
  }
  wv_row_sync();
  return n1;
}
// corner k (counter-clockwise seen from outside) of box face idx (boxFace :2011)
MJH_DEV int rw_box_corner_signs(int idx, int k) {
  // 3 bits per corner (bit 0: +x, bit 1: +y, bit 2: +z), four corners per face
  const int tab = idx == 0 ? (7 | (3 << 3) | (1 << 6) | (5 << 9)) :
                  idx == 1 ? (2 | (6 << 3) | (4 << 6) | (0 << 9)) :
                  idx == 2 ? (2 | (3 << 3) | (7 << 6) | (6 << 9)) :
                  idx == 3 ? (4 | (5 << 3) | (1 << 6) | (0 << 9)) :
                  idx == 4 ? (6 | (7 << 3) | (5 << 6) | (4 << 9)) :
                             (3 | (2 << 3) | (0 << 6) | (1 << 9));
  return (tab >> (3*k)) & 7;
}
// the polygon of face idx of a box / mesh in world coordinates, one vertex per lane (mesh: reverse list order, at most
// P vertices).  Returns the vertex count.  (boxFace :2011, meshFace :2068)
MJH_DEV int rc_face_polygon(MREF M, const real* frame, int type, int mesh, real* res, int idx, int P) {
  const int L = rw_l();
  const real* mat = frame + FR_MAT; const real* pos = frame + FR_POS; const real* size = frame + FR_SIZE;
  int n = 0;
  if (type == MJH_GEOM_BOX) {
    if (idx >= 0 && idx <= 5) {
      if (L < 4) {
        const int sg = rw_box_corner_signs(idx, L);
        st3(res + 3*L, rw_rot_add(mat, pos, (sg & 1) ? size[0] : -size[0], (sg & 2) ? size[1] : -size[1], (sg & 4) ? size[2] : -size[2]));
      }
      n = 4;
    }
  } else if (type == MJH_GEOM_MESH) {
    const int vadr = M.mesh_vertadr[mesh], padr = M.mesh_polyadr[mesh];
    const int adr = M.mesh_polyvertadr[padr + idx];
    const int nvert = M.mesh_polyvertnum[padr + idx];
    n = nvert < P ? nvert : P;
    for (int j = L; j < n; j += 16) {
      const int vb = 3*(vadr + M.mesh_polyvert[adr + nvert - 1 - j]);
      st3(res + 3*j, rw_rot_add(mat, pos, M.mesh_vert[vb], M.mesh_vert[vb + 1], M.mesh_vert[vb + 2]));
    }
  }
  wv_row_sync();
  return n;
}
// number of distinct vertices among the three ids; duplicates are squeezed out of ids and points (simplexDim :2112)
MJH_DEV int rw_feature_dim(int* vi, V3* p) {
  if (vi[0] == vi[1]) {
    if (vi[0] == vi[2]) return 1;
    vi[1] = vi[2];
    p[1] = p[2];
    return 2;
  }
  return (vi[2] == vi[0] || vi[2] == vi[1]) ? 2 : 3;
}
MJH_DEV real rw_quad_area(V3 a, V3 b, V3 c, V3 d) {
  const V3 ad = d - a, db = b - d, bc = c - b, ca = a - c;
  return 0.5*rw_len(cross(ad, db) + cross(bc, ca));
}
// four vertices of a convex polygon spanning a quadrilateral of (locally) maximal area: corners advance around the
// polygon while the area grows (polygonQuad :1523)
MJH_DEV void rw_widest_quad(int* res, const real* poly, int nvert) {
  int ca = 0, cb = 1, cc = 2, cd = 3;
  res[0] = ca; res[1] = cb; res[2] = cc; res[3] = cd;
  real best = rw_quad_area(ld3(poly), ld3(poly + 3), ld3(poly + 6), ld3(poly + 9));
  for (; ca < nvert; ca++) {
    for (;;) {
      const int dn = cd == nvert - 1 ? 0 : cd + 1;
      real trial = rw_quad_area(ld3(poly + 3*ca), ld3(poly + 3*cb), ld3(poly + 3*cc), ld3(poly + 3*dn));
      if (trial <= best) break;
      best = trial; cd = dn;
      res[0] = ca; res[1] = cb; res[2] = cc; res[3] = cd;
      for (;;) {
        const int cn = cc == nvert - 1 ? 0 : cc + 1;
        trial = rw_quad_area(ld3(poly + 3*ca), ld3(poly + 3*cb), ld3(poly + 3*cn), ld3(poly + 3*cd));
        if (trial <= best) break;
        best = trial; cc = cn;
        res[0] = ca; res[1] = cb; res[2] = cc; res[3] = cd;
      }
      for (;;) {
        const int bn = cb == nvert - 1 ? 0 : cb + 1;
        trial = rw_quad_area(ld3(poly + 3*ca), ld3(poly + 3*bn), ld3(poly + 3*cc), ld3(poly + 3*cd));
        if (trial <= best) break;
        best = trial; cb = bn;
        res[0] = ca; res[1] = cb; res[2] = cc; res[3] = cd;
      }
    }
    if (cb == ca) {
      cb = cb == nvert - 1 ? 0 : cb + 1;
      if (cc == cb) {
        cc = cc == nvert - 1 ? 0 : cc + 1;
        if (cd == cc) cd = cd == nvert - 1 ? 0 : cd + 1;
      }
    }
  }
}

// Clip polygon `subject` (ns vertices) against the side planes of polygon `window` (nwin vertices, plane normal n) and
// keep what lies on or below the window's plane; the survivors become witness pairs (point on the subject's shape,
// its projection along `toward` onto the window's plane), written to the witness table.  Sutherland-Hodgman with one
// polygon edge per lane; output slots by prefix sums of the lanes' point counts.  (polygonClip :1616)
MJH_DEV void rc_clip(RowPair& c, const ClipMem& b, const real* window, int nwin, const real* subject, int ns, V3 n, V3 toward) {
  if (nwin < 3) return;
  const int L = rw_l();
  const int cap = 2*b.P;
  real* cur = b.pa; real* nxt = b.pb;
  for (int i = L; i < 3*ns; i += 16) cur[i] = subject[i];
  wv_row_sync();
  int np = ns;
  for (int e = 0; e < nwin; e++) {
    // side plane through window edge e, containing n (planeNormal :1581)
    const V3 ea = ld3(window + 3*e), eb = ld3(window + 3*(e < nwin - 1 ? e + 1 : 0));
    V3 pe = cross(eb - ea, (ea + n) - ea);
    unitize(pe);
    const real pd = dot(pe, ea);
    int made = 0;
    for (int i0 = 0; i0 < np; i0 += 16) {
      const int i = i0 + L;
      int cnt = 0;
      V3 o1{0, 0, 0}, o2{0, 0, 0};
      if (i < np) {
        const V3 p = ld3(cur + 3*i), q = ld3(cur + 3*(i < np - 1 ? i + 1 : 0));
        const V3 pq = q - p;
        const int in_p = dot(p - ea, pe) > -MJH_MINVAL, in_q = dot(q - ea, pe) > -MJH_MINVAL;
        if (in_p && in_q) { o1 = q; cnt = 1; }
        else if (in_p || in_q) {
          const real along = dot(pe, pq);
          if (along != 0.0) {
            const real t = (pd - dot(pe, p))/along;
            if (t >= 0.0 && t <= 1.0) { o1 = V3{p.x + t*pq.x, p.y + t*pq.y, p.z + t*pq.z}; cnt = 1; }
          }
          if (in_q) { if (cnt) o2 = q; else o1 = q; cnt++; }
        }
      }
      const unsigned some = wv_row_ballot(cnt >= 1), two = wv_row_ballot(cnt == 2);
      const int at = made + rw_popc_below(some, L) + rw_popc_below(two, L);
      if (cnt >= 1 && at < cap) st3(nxt + 3*at, o1);
      if (cnt == 2 && at + 1 < cap) st3(nxt + 3*(at + 1), o2);
      made += __builtin_popcount(some) + __builtin_popcount(two);
    }
    wv_row_sync();
    real* t = cur; cur = nxt; nxt = t;
    np = made < cap ? made : cap;
  }
  // drop what is above the window's plane (ordered)
  const V3 w0 = ld3(window);
  int kept = 0;
  for (int i0 = 0; i0 < np; i0 += 16) {
    const int i = i0 + L;
    const V3 v = i < np ? ld3(cur + 3*i) : V3{0, 0, 0};
    const unsigned ok = wv_row_ballot(i < np && dot(v - w0, n) <= 0);
    if ((ok >> L) & 1) st3(nxt + 3*(kept + rw_popc_below(ok, L)), v);
    kept += __builtin_popcount(ok);
  }
  wv_row_sync();
  const real* poly = nxt;
  np = kept;
  if (np < 1) return;
  // which survivors become contacts
  int pick[4] = {0, 1, 2, 3};
  int nout;
  if (c.maxcon < 5 && np > 4) {
    rw_widest_quad(pick, poly, np);
    nout = 4;
  } else if (ns == 2 && np > 2) {
    // the two survivors farthest apart: every lane scans the partners of one vertex
    real far = -HUGE_VAL;
    int mate = RC_NONE, who = RC_NONE;
    real top = 0;
    int bi = 0, bj = 1;
    for (int i0 = 0; i0 < np; i0 += 16) {
      const int i = i0 + L;
      far = -HUGE_VAL; mate = RC_NONE; who = RC_NONE;
      if (i < np) {
        const V3 pi = ld3(poly + 3*i);
        for (int j = i + 1; j < np; j++) {
          const V3 df = ld3(poly + 3*j) - pi;
          const real d2 = dot(df, df);
          if (d2 > far) { far = d2; mate = j; }
        }
        if (mate != RC_NONE) who = i;
      }
      real v = far; int iv = who;
      rw_first_max<4>(v, iv);
      const int its_mate = wv_row_get_i(mate, iv == RC_NONE ? 0 : (iv & 15));
      if (iv != RC_NONE && v > top) { top = v; bi = iv; bj = its_mate; }
    }
    pick[0] = bi; pick[1] = bj;
    nout = 2;
  } else {
    nout = np < RC_MAXWIT ? np : RC_MAXWIT;
  }
  // witness pairs, one per lane (witnessOnFace :1605)
  real* tab = c.m.R + RO_SCR;
  if (L < nout) {
    const V3 v = ld3(poly + 3*(L == 0 ? pick[0] : (L == 1 ? pick[1] : (L == 2 ? pick[2] : pick[3]))));
    const real dist = dot(v - w0, n);
    const real s = -fabs(dist);
    tab[7*L] = dist;
    st3(tab + 7*L + 1, V3{v.x + s*toward.x, v.y + s*toward.y, v.z + s*toward.z});
    st3(tab + 7*L + 4, v);
  }
  wv_row_sync();
  c.nw = nout;
  c.tabled = 1;
}

// Several contacts for a polyhedral pair in penetration: find a pair of (anti-)parallel faces -- or a face and an
// edge lying in it -- around the features the final polytope face touches, and clip one against the other.
// (multicontact :2123)
MJH_DEV void rc_multicontact(MREF M, RowPair& c, int face) {
  const int L = rw_l();
  const int t1 = c.a.type, t2 = c.b.type;
  if (t1 == MJH_GEOM_MESH && !M.mesh_polynum[c.a.mesh]) return;
  if (t2 == MJH_GEOM_MESH && !M.mesh_polynum[c.b.mesh]) return;
  // the features: the (up to three distinct) vertices of each shape behind the polytope face
  int ia[3], ib[3];
  V3 pa[3], pb[3];
  for (int k = 0; k < 3; k++) {
    const int v = rp_face_vert(c, face, k);
    ia[k] = rm_vid(c.m, v)[0]; ib[k] = rm_vid(c.m, v)[1];
    pa[k] = ld3(rm_vert(c.m, v)); pb[k] = ld3(rm_vert(c.m, v) + 3);
  }
  wv_row_sync();                      // (the buffers below take the polytope's place)
  const ClipMem b = rp_clip_mem(M, c);
  const real* fa = rp_frame(c, 0); const real* fb = rp_frame(c, 1);
  int dim1 = rw_feature_dim(ia, pa);
  int dim2 = rw_feature_dim(ib, pb);
  const V3 a2b = c.w2 - c.w1, b2a = c.w1 - c.w2;
  int nn1 = 0, nn2 = 0;
  if (t1 == MJH_GEOM_BOX) nn1 = rc_box_normals(fa, b.n1, b.id1, dim1, ia, b2a);
  else if (t1 == MJH_GEOM_MESH) nn1 = rc_mesh_normals(M, fa, c.a.mesh, b.n1, b.id1, dim1, ia, b.D);
  if (t2 == MJH_GEOM_BOX) nn2 = rc_box_normals(fb, b.n2, b.id2, dim2, ib, a2b);
  else if (t2 == MJH_GEOM_MESH) nn2 = rc_mesh_normals(M, fb, c.b.mesh, b.n2, b.id2, dim2, ib, b.D);
  wv_row_sync();
  // first pair (i, j), i-major, of opposing face normals (alignedFaces :2086)
  int hit1 = -1, hit2 = -1;
  {
    const int total = nn1*nn2;
    for (int q0 = 0; q0 < total && hit1 < 0; q0 += 16) {
      const int q = q0 + L;
      const int i = nn2 ? q / nn2 : 0, j = nn2 ? q - i*nn2 : 0;
      const unsigned hits = wv_row_ballot(q < total && dot(ld3(b.n1 + 3*i), ld3(b.n2 + 3*j)) < -RC_FACE_ALIGN);
      if (hits) { const int at = q0 + __builtin_ctz(hits); hit1 = at / nn2; hit2 = at - hit1*nn2; }
    }
  }
  int edge_of_a = 0, edge_of_b = 0;
  if (hit1 < 0) {
    // no face pair: an edge of the lower-dimensional feature lying in a face of the other shape (alignedFaceEdge :2099)
    if (dim1 < 3 && dim1 <= dim2) {
      nn1 = rc_edge_dirs(M, fa, t1, c.a.mesh, b.n1, b.ev, dim1, pa[0], pa[1], ia[0], b.D);
      if (t1 != MJH_GEOM_BOX && t1 != MJH_GEOM_MESH) nn1 = 0;
      const int total = nn2*nn1;
      for (int q0 = 0; q0 < total && hit1 < 0; q0 += 16) {
        const int q = q0 + L;
        const int fi = nn1 ? q / nn1 : 0, ej = nn1 ? q - fi*nn1 : 0;
        const unsigned hits = wv_row_ballot(q < total && fabs(dot(ld3(b.n1 + 3*ej), ld3(b.n2 + 3*fi))) < RC_EDGE_ALIGN);
        if (hits) { const int at = q0 + __builtin_ctz(hits); hit2 = at / nn1; hit1 = at - hit2*nn1; }
      }
      if (hit1 < 0) return;
      edge_of_a = 1;
    } else if (dim2 < 3) {
      nn2 = rc_edge_dirs(M, fb, t2, c.b.mesh, b.n2, b.ev, dim2, pb[0], pb[1], ib[0], b.D);
      if (t2 != MJH_GEOM_BOX && t2 != MJH_GEOM_MESH) nn2 = 0;
      const int total = nn1*nn2;
      for (int q0 = 0; q0 < total && hit1 < 0; q0 += 16) {
        const int q = q0 + L;
        const int fi = nn2 ? q / nn2 : 0, ej = nn2 ? q - fi*nn2 : 0;
        const unsigned hits = wv_row_ballot(q < total && fabs(dot(ld3(b.n2 + 3*ej), ld3(b.n1 + 3*fi))) < RC_EDGE_ALIGN);
        if (hits) { const int at = q0 + __builtin_ctz(hits); hit2 = at / nn2; hit1 = at - hit2*nn2; }
      }
      if (hit1 < 0) return;
      edge_of_b = 1;
    } else {
      return;
    }
  }
  // (hit1, hit2) = (index into the first list searched, into the second): faces (i of A, j of B); or (edge, face)
  const int i = hit1, j = hit2;
  int nf1 = 0, nf2 = 0;
  if (edge_of_a) {
    if (L == 0) { st3(b.f1, pa[0]); st3(b.f1 + 3, ld3(b.ev + 3*i)); }
    wv_row_sync();
    nf1 = 2;
  } else {
    nf1 = rc_face_polygon(M, fa, t1, c.a.mesh, b.f1, edge_of_b ? b.id1[j] : b.id1[i], b.P);
  }
  if (edge_of_b) {
    if (L == 0) { st3(b.f2, pb[0]); st3(b.f2 + 3, ld3(b.ev + 3*i)); }
    wv_row_sync();
    nf2 = 2;
  } else {
    nf2 = rc_face_polygon(M, fb, t2, c.b.mesh, b.f2, b.id2[j], b.P);
  }
  if (edge_of_a) {
    // A's edge against B's face: the witnesses come out B-first
    const V3 nj = ld3(b.n2 + 3*j);
    const int had = c.nw;
    c.nw = -1;
    rc_clip(c, b, b.f2, nf2, b.f1, nf1, nj, rw_scl(nj, -1.0));
    if (c.nw < 0) { c.nw = had; const V3 t = c.w1; c.w1 = c.w2; c.w2 = t; return; }      // (the reference swaps whatever witnesses there are)
    real* tab = c.m.R + RO_SCR;
    if (L < c.nw) for (int q = 0; q < 3; q++) { const real t = tab[7*L + 1 + q]; tab[7*L + 1 + q] = tab[7*L + 4 + q]; tab[7*L + 4 + q] = t; }
    wv_row_sync();
    return;
  }
  if (edge_of_b) {
    const V3 nj = ld3(b.n1 + 3*j);
    const int had = c.nw;
    c.nw = -1;
    rc_clip(c, b, b.f1, nf1, b.f2, nf2, nj, rw_scl(nj, -1.0));
    if (c.nw < 0) c.nw = had;
    return;
  }
  {
    const int had = c.nw;
    c.nw = -1;
    rc_clip(c, b, b.f1, nf1, b.f2, nf2, ld3(b.n1 + 3*i), ld3(b.n2 + 3*j));
    if (c.nw < 0) c.nw = had;
  }
}

// ---- one query: distance, then penetration depth and contacts (mjc_ccd :2318) ------------------------------------------
// returns the smallest witness distance (negative: penetration)
MJH_DEV real rc_solve(MREF M, RowPair& c) {
  const real* fa = rp_frame(c, 0); const real* fb = rp_frame(c, 1);
  // (mjc_center: a geom's position, the bounding-box centre of a flex element)
  const V3 centre_a = ld3(fa + (c.a.kind == SK_FLEXELEM ? FR_CENTRE : FR_POS));
  const V3 centre_b = ld3(fb + (c.b.kind == SK_FLEXELEM ? FR_CENTRE : FR_POS));
  c.w1 = centre_a; c.w2 = centre_b;
  c.spent = 0;
  c.cutoff = 0;
  c.tabled = 0;
  const int t1 = c.a.type, t2 = c.b.type;
  if (t1 == MJH_GEOM_SPHERE || t2 == MJH_GEOM_SPHERE || t1 == MJH_GEOM_CAPSULE || t2 == MJH_GEOM_CAPSULE) {
    // spheres shrink to points and capsules to segments; the result is inflated again
    const int kind_a = c.a.kind, kind_b = c.b.kind;
    const real margin_a = c.a.margin, margin_b = c.b.margin;
    real full1 = 0, full2 = 0;
    if (t1 == MJH_GEOM_SPHERE) { full1 = fa[FR_SIZE] + 0.5*margin_a; c.a.kind = SK_POINT; c.a.margin = 0; }
    else if (t1 == MJH_GEOM_CAPSULE) { full1 = fa[FR_SIZE] + 0.5*margin_a; c.a.kind = SK_SEGMENT; c.a.margin = 0; }
    if (t2 == MJH_GEOM_SPHERE) { full2 = fb[FR_SIZE] + 0.5*margin_b; c.b.kind = SK_POINT; c.b.margin = 0; }
    else if (t2 == MJH_GEOM_CAPSULE) { full2 = fb[FR_SIZE] + 0.5*margin_b; c.b.kind = SK_SEGMENT; c.b.margin = 0; }
    c.cutoff += full1 + full2;
    rc_distance(M, c);
    c.cutoff = 0;
    c.a.margin = margin_a; c.b.margin = margin_b;
    c.a.kind = kind_a; c.b.kind = kind_b;
    if (c.dist0 > c.tol) {
      V3 n = c.w2 - c.w1;
      unitize(n);
      if (full1) { c.w1.x += full1*n.x; c.w1.y += full1*n.y; c.w1.z += full1*n.z; }
      if (full2) { c.w2.x -= full2*n.x; c.w2.y -= full2*n.y; c.w2.z -= full2*n.z; }
      c.dist0 -= (full1 + full2);
      if (c.dist0 > c.cutoff) c.dist0 = RC_DBLMAX;
      return c.dist0;
    }
    c.spent = 0;
    c.w1 = centre_a; c.w2 = centre_b;
  }
  rc_distance(M, c);
  if (c.dist0 <= c.tol && c.nsim > 1 && !c.apart) {
    c.dist0 = 0;
    c.nf = c.nm = c.nv = 0;
    RC_COUNT(3);
    const int failed = c.nsim == 2 ? rc_polytope_from_segment(M, c)
                     : (c.nsim == 3 ? rc_polytope_from_triangle(M, c) : rc_polytope_from_tetrahedron(M, c));
    if (!failed) {
      const int face = rc_expand(M, c);
      if (c.maxcon > 1 && face >= 0) { RC_COUNT(5); rc_multicontact(M, c, face); }
    }
  }
  if (!c.tabled) return c.dist0;
  const real* tab = c.m.R + RO_SCR;
  const int L = rw_l();
  return rw_min_all(L < c.nw ? tab[7*L] : HUGE_VAL);
}

// contact records (dist, pos[3], normal[3]) of the query into rec[first...]; returns their number (mjc_penetration :87)
MJH_DEV int rc_contacts(MREF M, RowPair& c, real* rec, int first, int maxcon, real margin) {
  c.maxcon = maxcon;
  const real deepest = rc_solve(M, c);
  wv_row_sync();
  if (!(deepest < 0)) return 0;
  const int L = rw_l();
  const int n = c.nw;
  if (L < n) {
    real d; V3 x1, x2;
    if (c.tabled) { const real* t = c.m.R + RO_SCR + 7*L; d = t[0]; x1 = ld3(t + 1); x2 = ld3(t + 4); }
    else { d = c.dist0; x1 = c.w1; x2 = c.w2; }
    real* o = rec + RC_RECORD*(first + L);
    o[0] = margin + d;
    V3 pos = x1 + x2;
    st3(o + 1, V3{pos.x*0.5, pos.y*0.5, pos.z*0.5});
    V3 nrm = x1 - x2;
    unitize(nrm);
    st3(o + 4, nrm);
  }
  wv_row_sync();
  return n;
}

// geom g as a shape; its frame goes to slot k of the row workspace (mjc_initCCDObj :726)
template <class GX, class GM>
MJH_DEV Shape rp_load_geom(MREF M, RowPair& c, int k, GX gx, GM gm, int g, real margin) {
  const int L = rw_l();
  real* f = rp_frame(c, k);
  if (L < 3) { f[FR_POS + L] = gx[3*g + L]; f[FR_SIZE + L] = M.geom_size[3*g + L]; }
  if (L < 9) f[FR_MAT + L] = gm[9*g + L];
  Shape s;
  s.type = M.geom_type[g];
  s.vcache = -1; s.gcache = -1; s.mesh = -1;
  s.margin = margin;
  s.kind = SK_POINT;
  if (s.type == MJH_GEOM_SPHERE) s.kind = SK_SPHERE;
  else if (s.type == MJH_GEOM_CAPSULE) s.kind = SK_CAPSULE;
  else if (s.type == MJH_GEOM_ELLIPSOID) s.kind = SK_ELLIPSOID;
  else if (s.type == MJH_GEOM_CYLINDER) s.kind = SK_CYLINDER;
  else if (s.type == MJH_GEOM_BOX) s.kind = SK_BOX;
  else if (s.type == MJH_GEOM_MESH) {
    s.mesh = M.geom_dataid[g];
    s.kind = (M.mesh_graphadr[s.mesh] < 0 || M.mesh_vertnum[s.mesh] < 10) ? SK_MESH_ALL : SK_MESH_CLIMB;    // mjMESH_HILLCLIMB_MIN
  }
  return s;
}

// ---- the environment's workspace: row pages in the LDS-planned field ccd_row, the rest in the global buffer ccd_ws -----
// ccd_ws of environment e: [header: 64 pairs, 64 lanes, 64 counts, 64 arguments] [records: 64 x RC_MAXOUT x 7 reals] [one overflow page per row]
MJH_DEV int* rc_header(MREF M, BREF B, int e) { return (int*)((char*)B.ccd_ws + (size_t)e*(size_t)M.s.ccd_env_bytes); }
MJH_DEV real* rc_records(MREF M, BREF B, int e, int slot) {
  return (real*)((char*)B.ccd_ws + (size_t)e*(size_t)M.s.ccd_env_bytes + 256*sizeof(int)) + RC_MAXOUT*RC_RECORD*slot;
}
// tables of the polyhedral pairs' distance phase (behind the rows' fallback pages): slot of every static pair in the
// current list (-1: not listed), overlap flag and parked simplex (32 reals) per list entry, the list itself
struct PolyTab { int* slot; int* flag; int* list; real* park; };
MJH_DEV PolyTab rc_poly_tables(MREF M, BREF B, int e) {
  const MJH_CONST_AS DSizes& s = M.s;
  char* q = (char*)B.ccd_ws + (size_t)e*(size_t)s.ccd_env_bytes + 256*sizeof(int) + 64*RC_MAXOUT*RC_RECORD*sizeof(real)
            + (size_t)s.ccd_rows*((size_t)s.ccd_slow_bytes + (size_t)s.ccd_row_reals*sizeof(real));
  PolyTab t;
  t.park = (real*)q; q += (size_t)s.ccd_npoly*32*sizeof(real);
  t.slot = (int*)q; q += (size_t)s.npair*sizeof(int);
  t.flag = (int*)q; q += (size_t)s.ccd_npoly*sizeof(int);
  t.list = (int*)q;
  return t;
}
MJH_DEV void rc_attach(MREF M, BREF B, int e, RowPair& c) {
  const MJH_CONST_AS DSizes& s = M.s;
  const int row = wv_lane() >> 4;          // (row of the group: up to s.ccd_rows in a multi-wavefront workgroup)
  char* block = (char*)B.ccd_ws + (size_t)e*(size_t)s.ccd_env_bytes;
  char* pages = block + 256*sizeof(int) + 64*RC_MAXOUT*RC_RECORD*sizeof(real);
  // fast page: the LDS-planned field ccd_row, else (no plan, or no room in it) the tail of the environment's global block
  real* fast = (B.l_ccd_row >= 0 ? (real*)(MJH_LDS(B) + B.l_ccd_row) : (real*)(pages + (size_t)s.ccd_rows*(size_t)s.ccd_slow_bytes))
               + (size_t)row*s.ccd_row_reals;
#if defined(MJH_CCD_ASSUME_LDS) && !defined(MJH_HOSTSIM) && defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(__builtin_amdgcn_is_shared(fast));
#endif
  c.m.R = fast;
  c.m.I = (int*)(fast + s.ccd_row_freal);
  char* slow = pages + (size_t)row*(size_t)s.ccd_slow_bytes;
  c.m.RS = (real*)slow;
  c.m.nslow_v = 5 + s.ccd_N;
  c.m.nslow_f = 6*s.ccd_N;
  c.m.IS = (int*)(slow + (size_t)(6*c.m.nslow_v + 4*c.m.nslow_f)*sizeof(real));
  c.tol = M.o.ccd_tolerance;
  c.iters = s.ccd_N;
  c.maxcon = 1; c.cutoff = 0;
  c.nsim = c.apart = c.nw = c.spent = c.tabled = 0;
  c.nv = c.nf = c.nm = 0;
  c.dist0 = 0;
  c.w1 = c.w2 = c.centre = V3{0, 0, 0};
#ifdef MJH_PROFILE
  c.t_far = c.t_clip = 0;
#endif
}
// the calling lane's contact records (dist, pos[3], normal[3]) x RC_MAXOUT
MJH_DEV crptr ccd_out_records(MREF M, BREF B, int e) { return crptr{rc_records(M, B, e, wv_lane()), 1}; }

// mjc_Convex (:881) for the row's pair p: returns the number of contacts left in rec
MJH_DEV int rc_geom_pair(MREF M, BREF B, int e, RowPair& c, int p, real* rec) {
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  const int L = rw_l();
  const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
  const real margin = M.pair_margin[p];
  c.a = rp_load_geom(M, c, 0, gx, gm, g1, margin);
  c.b = rp_load_geom(M, c, 1, gx, gm, g2, margin);
  wv_row_sync();
  const int t1 = c.a.type, t2 = c.b.type;
  const int multiccd = !(M.o.disableflags & (1 << 19));
  // maxContacts (:855)
  int maxcon = 1;
  if (!(margin > 0) && multiccd && (t1 == MJH_GEOM_BOX || t1 == MJH_GEOM_MESH) && (t2 == MJH_GEOM_BOX || t2 == MJH_GEOM_MESH))
    maxcon = (t1 == MJH_GEOM_BOX && t2 == MJH_GEOM_BOX) ? 8 : 4;
  int ncon = rc_contacts(M, c, rec, 0, maxcon, margin);
  if (maxcon > 1) return ncon;
  if (ncon == 1 && multiccd && t1 != MJH_GEOM_ELLIPSOID && t1 != MJH_GEOM_SPHERE && t2 != MJH_GEOM_ELLIPSOID && t2 != MJH_GEOM_SPHERE) {
    // more contacts by perturbation: both geoms turned by +-1e-3 rad about the two tangents of the first contact
    real frame[9] = {rec[4], rec[5], rec[6], 0, 0, 0, 0, 0, 0};
    make_frame(frame);
    const real apart = 1e-3*r_min(M.geom_rbound[g1], M.geom_rbound[g2]);
    const V3 origin = ld3(rec + 1);
    for (int turn = 0; turn < 4; turn++) {
      const real* axis = frame + 3 + 3*(turn >> 1);
      // mji_axisAngle2Quat with angle -+1e-3: sin / cos of 5e-4 evaluated by the host's libm at upload
      const real sn = (turn & 1) == 0 ? -M.o.ccd_sin : M.o.ccd_sin;
      const real quat[4] = {M.o.ccd_cos, axis[0]*sn, axis[1]*sn, axis[2]*sn};
      real rot[9], back[9];
      q_tomat(rot, quat);
      for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) back[3*r + q] = rot[3*q + r];
      // A turns one way, B the other, both about the contact point (mju_rotateFrame :834)
      real nm[2][9]; V3 np[2];
      for (int side = 0; side < 2; side++) {
        const real* R = side == 0 ? rot : back;
        const real* xmat = rp_frame(c, side) + FR_MAT;
        const real* xpos = rp_frame(c, side) + FR_POS;
        for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++)
          nm[side][3*r + q] = R[3*r]*xmat[q] + R[3*r + 1]*xmat[3 + q] + R[3*r + 2]*xmat[6 + q];
        const V3 rel = origin - ld3(xpos);
        V3 vec = mmul(R, rel);
        vec = vec - rel;
        np[side] = V3{xpos[0] - vec.x, xpos[1] - vec.y, xpos[2] - vec.z};
      }
      wv_row_sync();
      if (L < 2) {
        real* f = rp_frame(c, L);
        for (int q = 0; q < 9; q++) f[FR_MAT + q] = L == 0 ? nm[0][q] : nm[1][q];
        st3(f + FR_POS, L == 0 ? np[0] : np[1]);
      }
      wv_row_sync();
      const int n = rc_contacts(M, c, rec, ncon, 1, margin);
      if (n) {
        // keep it if it is not where an earlier contact is (mjc_isDistinctContact :822)
        int distinct = 1;
        const V3 last = ld3(rec + RC_RECORD*ncon + 1);
        for (int i = 0; i < ncon; i++) {
          const V3 df = ld3(rec + RC_RECORD*i + 1) - last;
          if (sqrt(df.x*df.x + df.y*df.y + df.z*df.z) <= apart) { distinct = 0; break; }
        }
        wv_row_sync();
        if (distinct) { if (L == 0) rec[RC_RECORD*ncon] = rec[0]; ncon++; }
      }
      wv_row_sync();
      if (L < 3) { rp_frame(c, 0)[FR_POS + L] = gx[3*g1 + L]; rp_frame(c, 1)[FR_POS + L] = gx[3*g2 + L]; }
      if (L < 9) { rp_frame(c, 0)[FR_MAT + L] = gm[9*g1 + L]; rp_frame(c, 1)[FR_MAT + L] = gm[9*g2 + L]; }
      wv_row_sync();
    }
  }
  return ncon;
}

// mjc_ConvexElem (:1559): geom g against solid flex element `elem` (a tetrahedron): one contact at most
MJH_DEV int rc_geom_elem(MREF M, BREF B, int e, RowPair& c, int g, int elem, real margin, real* rec) {
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr aabb = MJH_F(B, flexelem_aabb, e);
  const int L = rw_l();
  c.a = rp_load_geom(M, c, 0, gx, gm, g, margin);
  const int fl = M.flexelem_flex[elem];
  const int n = M.flex_dim[fl] + 1;
  real* f = rp_frame(c, 1);
  if (L < 3*n) f[L] = vx[3*M.flexelem_vert[4*elem + L/3] + L%3];
  if (L == 12) f[FR_SIZE] = M.flex_radius[fl] + 0.5*margin;
  if (L >= 13) f[FR_CENTRE + L - 13] = aabb[6*elem + L - 13];
  c.b.type = MJH_GEOM_FLEX; c.b.kind = SK_FLEXELEM; c.b.vcache = -1; c.b.gcache = -1; c.b.mesh = n; c.b.margin = 0;
  wv_row_sync();
  return rc_contacts(M, c, rec, 0, 1, margin);
}

// mjc_ConvexElem for two flex elements (mj_collideElems, engine_collision_driver.c:2568): both shapes are elements
MJH_DEV int rc_elem_elem(MREF M, BREF B, int e, RowPair& c, int elem1, int elem2, real margin, real* rec) {
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr aabb = MJH_F(B, flexelem_aabb, e);
  const int L = rw_l();
  for (int side = 0; side < 2; side++) {
    const int elem = side ? elem2 : elem1;
    const int fl = M.flexelem_flex[elem];
    const int n = M.flex_dim[fl] + 1;
    real* f = rp_frame(c, side);
    if (L < 3*n) f[L] = vx[3*M.flexelem_vert[4*elem + L/3] + L%3];
    if (L == 12) f[FR_SIZE] = M.flex_radius[fl] + 0.5*margin;
    if (L >= 13) f[FR_CENTRE + L - 13] = aabb[6*elem + L - 13];
    Shape& sh = side ? c.b : c.a;
    sh.type = MJH_GEOM_FLEX; sh.kind = SK_FLEXELEM; sh.vcache = -1; sh.gcache = -1; sh.mesh = n; sh.margin = 0;
  }
  wv_row_sync();
  return rc_contacts(M, c, rec, 0, 1, margin);
}

// number of contacts a polyhedral pair may return (maxContacts :855); 1: the pair takes the single-contact path
MJH_DEV int rc_max_contacts(MREF M, int p) {
  const int t1 = M.geom_type[M.pair_geom1[p]], t2 = M.geom_type[M.pair_geom2[p]];
  const int multiccd = !(M.o.disableflags & (1 << 19));
  if (!(M.pair_margin[p] > 0) && multiccd && (t1 == MJH_GEOM_BOX || t1 == MJH_GEOM_MESH) && (t2 == MJH_GEOM_BOX || t2 == MJH_GEOM_MESH))
    return (t1 == MJH_GEOM_BOX && t2 == MJH_GEOM_BOX) ? 8 : 4;
  return 1;
}
// Penetration phase of a polyhedral pair whose distance phase (ccd_poly_distance, below) found the shapes overlapping:
// polytope, expansion, multi-contact from the parked simplex; contact records into rec (= the parking slot)
MJH_DEV int rc_poly_pair_penetration(MREF M, BREF B, int e, RowPair& c, int p, const real* park, real* rec) {
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  const int L = rw_l();
  const real margin = M.pair_margin[p];
  c.a = rp_load_geom(M, c, 0, gx, gm, M.pair_geom1[p], margin);
  c.b = rp_load_geom(M, c, 1, gx, gm, M.pair_geom2[p], margin);
  real* sim = c.m.R + RO_SIM; int* sid = c.m.I + IO_SIM;
  const int* pi = (const int*)(park + 24);
  for (int q = L; q < 24; q += 16) sim[q] = park[q];
  if (L < 8) sid[L] = pi[L];
  c.nsim = pi[8]; c.a.vcache = pi[9]; c.a.gcache = pi[10]; c.b.vcache = pi[11]; c.b.gcache = pi[12];
  wv_row_sync();
  c.maxcon = rc_max_contacts(M, p);
  c.cutoff = 0; c.tabled = 0; c.apart = 0; c.nw = 0;
  c.dist0 = 0;
  c.nf = c.nm = c.nv = 0;
  RC_COUNT(3);
  const int failed = c.nsim == 2 ? rc_polytope_from_segment(M, c)
                   : (c.nsim == 3 ? rc_polytope_from_triangle(M, c) : rc_polytope_from_tetrahedron(M, c));
  if (failed) return 0;
  const int face = rc_expand(M, c);
#ifdef MJH_PROFILE
  const long long t1_ = wv_clock();
#endif
  if (c.maxcon > 1 && face >= 0) { RC_COUNT(5); rc_multicontact(M, c, face); }
#ifdef MJH_PROFILE
  c.t_clip += wv_clock() - t1_;
#endif
  real deepest = c.dist0;
  if (c.tabled) deepest = rw_min_all(L < c.nw ? (c.m.R + RO_SCR)[7*L] : HUGE_VAL);
  wv_row_sync();
  if (!(deepest < 0)) return 0;
  const int n = c.nw;
  if (L < n) {
    real d; V3 x1, x2;
    if (c.tabled) { const real* t = c.m.R + RO_SCR + 7*L; d = t[0]; x1 = ld3(t + 1); x2 = ld3(t + 4); }
    else { d = c.dist0; x1 = c.w1; x2 = c.w2; }
    real* o = rec + RC_RECORD*L;
    o[0] = margin + d;
    const V3 pos = x1 + x2;
    st3(o + 1, V3{pos.x*0.5, pos.y*0.5, pos.z*0.5});
    V3 nrm = x1 - x2;
    unitize(nrm);
    st3(o + 4, nrm);
  }
  wv_row_sync();
  return n;
}

// ---- lane-parallel distance phase for polyhedral pairs ------------------------------------------------------------------
// The distance query of a box / mesh pair without margin needs no workspace beyond its simplex, and two out of three
// pairs in reach of each other turn out to be apart after one to three iterations: run as four pairs per wavefront the
// query is pure instruction-issue cost (SQ counters, profiles/r04*: 0.3 M vector instructions per cube step, as many as
// the one-pair-per-lane code of round 3 that kept its workspace in global memory).  So the distance phase of THESE
// pairs runs with ONE PAIR PER LANE AND THE SIMPLEX IN REGISTERS -- no memory traffic at all except the model's mesh
// tables -- and only the pairs that overlap (their simplex parked in the owner's record slot) go on to the
// row-cooperative penetration phase above.  The sub-simplex evaluation is the same table as rc_simplex_weights, filled
// lazily: an entry (tetrahedron, 4 triangles, 6 segments) is computed when some lane of the wavefront needs it.
struct LaneVert { V3 pa, pb; int ia, ib; };
MJH_DEV V3 lv_mink(const LaneVert& v) { return V3{v.pa.x - v.pb.x, v.pa.y - v.pb.y, v.pa.z - v.pb.z}; }
MJH_DEV V3 lv_pick(int k, V3 a, V3 b, V3 c, V3 d) {
  return V3{k == 0 ? a.x : (k == 1 ? b.x : (k == 2 ? c.x : d.x)), k == 0 ? a.y : (k == 1 ? b.y : (k == 2 ? c.y : d.y)),
            k == 0 ? a.z : (k == 1 ? b.z : (k == 2 ? c.z : d.z))};
}
MJH_DEV LaneVert lv_pickv(int k, const LaneVert& a, const LaneVert& b, const LaneVert& c, const LaneVert& d) {
  LaneVert r;
  r.pa = lv_pick(k, a.pa, b.pa, c.pa, d.pa); r.pb = lv_pick(k, a.pb, b.pb, c.pb, d.pb);
  r.ia = k == 0 ? a.ia : (k == 1 ? b.ia : (k == 2 ? c.ia : d.ia));
  r.ib = k == 0 ? a.ib : (k == 1 ? b.ib : (k == 2 ? c.ib : d.ib));
  return r;
}
MJH_DEV void lv_put(int k, int on, LaneVert& a, LaneVert& b, LaneVert& c, LaneVert& d, const LaneVert& v) {
  if (on && k == 0) a = v;
  if (on && k == 1) b = v;
  if (on && k == 2) c = v;
  if (on && k == 3) d = v;
}
MJH_DEV real rw_pickr(const real* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3])); }
struct LaneShape { int g, kind, mesh, vcache, gcache; };

// farthest point of the lane's shape along dir (box, mesh by exhaustive search, mesh by hill climbing), lanes with on == 0 idle
template <class GX, class GM>
MJH_DEV V3 lp_far(MREF M, GX gx, GM gm, LaneShape& s, V3 dir, int on) {
  const int g = on ? s.g : 0;
  const V3 ld = rw_to_local(gm + 9*g, dir);
  V3 local{0, 0, 0};
  if (on && s.kind == SK_BOX) {
    const real sx = M.geom_size[3*g], sy = M.geom_size[3*g + 1], sz = M.geom_size[3*g + 2];
    local = V3{ld.x >= 0 ? sx : -sx, ld.y >= 0 ? sy : -sy, ld.z >= 0 ? sz : -sz};
    s.vcache = ((local.x > 0) ? 1 : 0) | ((local.y > 0) ? 2 : 0) | ((local.z > 0) ? 4 : 0);
  }
  const int all = on && s.kind == SK_MESH_ALL;
  if (wv_any(all)) {
    const int mesh = all ? s.mesh : 0;
    const int vadr = 3*M.mesh_vertadr[mesh], nverts = all ? M.mesh_vertnum[mesh] : 0;
    real top = -RC_FLTMAX;
    int itop = 0;
    if (all && s.vcache >= 0) { itop = s.vcache; top = rw_vdot(M, ld, vadr + 3*itop); }
    for (int k = 0; wv_any(k < nverts); k++) {
      if (k < nverts) { const real v = rw_vdot(M, ld, vadr + 3*k); if (v > top) { top = v; itop = k; } }
    }
    if (all) { s.vcache = itop; local = rw_mesh_vert(M, vadr + 3*itop); }
  }
  const int climb = on && s.kind == SK_MESH_CLIMB;
  if (wv_any(climb)) {
    const int mesh = climb ? s.mesh : 0;
    const int vadr = 3*M.mesh_vertadr[mesh];
    const int gadr = M.mesh_graphadr[mesh];
    const int numvert = climb ? M.mesh_graph[gadr] : 0;
    const int edgeadr = gadr + 2, globalid = gadr + 2 + numvert, localid = gadr + 2 + 2*numvert;
    int cur = 0;
    real top = 0;
    if (climb) {
      const int cx = (ld.x > 0.4) - (ld.x < -0.4) + 1;
      const int cy = (ld.y > 0.4) - (ld.y < -0.4) + 1;
      const int cz = (ld.z > 0.4) - (ld.z < -0.4) + 1;
      const int seed = M.mesh_extrema[27*mesh + cx*9 + cy*3 + cz];
      cur = seed;
      if (s.gcache >= 0) {
        const real vc = rw_vdot(M, ld, vadr + 3*M.mesh_graph[globalid + s.gcache]);
        const real vs = rw_vdot(M, ld, vadr + 3*M.mesh_graph[globalid + seed]);
        cur = (vs > vc) ? seed : s.gcache;
      }
      top = rw_vdot(M, ld, vadr + 3*M.mesh_graph[globalid + cur]);
    }
    int moving = climb;
    while (wv_any(moving)) {
      const int from = cur;
      int k = moving ? M.mesh_graph[edgeadr + from] : 0;
      int scanning = moving;
      while (wv_any(scanning)) {
        if (scanning) {
          const int nb = M.mesh_graph[localid + k];
          if (nb < 0) scanning = 0;
          else {
            const real v = rw_vdot(M, ld, vadr + 3*M.mesh_graph[globalid + nb]);
            if (v > top) { top = v; cur = nb; }
            k++;
          }
        }
      }
      if (moving && cur == from) moving = 0;
    }
    if (climb) {
      s.gcache = cur;
      s.vcache = M.mesh_graph[globalid + cur];
      local = rw_mesh_vert(M, vadr + 3*s.vcache);
    }
  }
  return rw_to_world(gm + 9*g, local, gx + 3*g);
}
template <class GX, class GM>
MJH_DEV LaneVert lp_pair_far(MREF M, GX gx, GM gm, LaneShape& a, LaneShape& b, V3 d, V3 dn, int on) {
  LaneVert v;
  v.pa = lp_far(M, gx, gm, a, d, on);
  v.pb = lp_far(M, gx, gm, b, dn, on);
  v.ia = a.vcache; v.ib = b.vcache;
  return v;
}

// triangle (s1 s2 s3) of the sub-simplex table: weights if the origin's foot point lies inside, else which of its
// segments (bit 0: (s2,s3), 1: (s1,s3), 2: (s1,s2)) have to decide; degenerate: segment (s1,s2) alone (bit 3)
MJH_DEV int lp_triangle(V3 s1, V3 s2, V3 s3, real* w) {
  V3 foot;
  if (rw_plane_foot(foot, s1, s2, s3)) return 8;
  const real m23 = s2.y*s3.z - s2.z*s3.y - s1.y*s3.z + s1.z*s3.y + s1.y*s2.z - s1.z*s2.y;
  const real m13 = s2.x*s3.z - s2.z*s3.x - s1.x*s3.z + s1.z*s3.x + s1.x*s2.z - s1.z*s2.x;
  const real m12 = s2.x*s3.y - s2.y*s3.x - s1.x*s3.y + s1.y*s3.x + s1.x*s2.y - s1.y*s2.x;
  const real g1 = fabs(m23), g2 = fabs(m13), g3 = fabs(m12);
  const int drop = (g1 >= g2 && g1 >= g3) ? 0 : (g2 >= g3 ? 1 : 2);
  const real area = drop == 0 ? m23 : (drop == 1 ? m13 : m12);
  const int u = drop == 0 ? 1 : 0, v = drop == 2 ? 1 : 2;
  const real a0 = comp(s1, u), a1 = comp(s1, v), b0 = comp(s2, u), b1 = comp(s2, v), c0 = comp(s3, u), c1 = comp(s3, v);
  const real q0 = comp(foot, u), q1 = comp(foot, v);
  const real ka = q0*b1 + q1*c0 + b0*c1 - q0*c1 - q1*b0 - c0*b1;
  const real kb = q0*c1 + q1*a0 + c0*a1 - q0*a1 - q1*c0 - a0*c1;
  const real kc = q0*a1 + q1*b0 + a0*b1 - q0*b1 - q1*a0 - b0*a1;
  const int ina = rw_sign_match(area, ka), inb = rw_sign_match(area, kb), inc = rw_sign_match(area, kc);
  if (ina && inb && inc) { w[0] = ka/area; w[1] = kb/area; w[2] = kc/area; return 0; }
  return (ina ? 0 : 1) | (inb ? 0 : 2) | (inc ? 0 : 4);
}
// the triangle's answer from its segments' weights (same precedence as the row version): e23, e13, e12 = weights of
// segments (s2,s3), (s1,s3), (s1,s2)
MJH_DEV void lp_triangle_fallback(int need, V3 s1, V3 s2, V3 s3, const real* e23, const real* e13, const real* e12, real* w) {
  if (need & 8) { w[0] = e12[0]; w[1] = e12[1]; w[2] = 0; return; }
  real nearest = RC_DBLMAX;
  w[0] = w[1] = w[2] = 0;
  if (need & 1) {
    const V3 x{e23[0]*s2.x + e23[1]*s3.x, e23[0]*s2.y + e23[1]*s3.y, e23[0]*s2.z + e23[1]*s3.z};
    w[0] = 0; w[1] = e23[0]; w[2] = e23[1];
    nearest = dot(x, x);
  }
  if (need & 2) {
    const V3 x{e13[0]*s1.x + e13[1]*s3.x, e13[0]*s1.y + e13[1]*s3.y, e13[0]*s1.z + e13[1]*s3.z};
    const real dd = dot(x, x);
    if (dd < nearest) { w[0] = e13[0]; w[1] = 0; w[2] = e13[1]; nearest = dd; }
  }
  if (need & 4) {
    const V3 x{e12[0]*s1.x + e12[1]*s2.x, e12[0]*s1.y + e12[1]*s2.y, e12[0]*s1.z + e12[1]*s2.z};
    const real dd = dot(x, x);
    if (dd < nearest) { w[0] = e12[0]; w[1] = e12[1]; w[2] = 0; }
  }
}
// weights of the lane's simplex points P[0..n) (n = 2..4) for the point closest to the origin; lanes with on == 0 idle
MJH_DEV void lp_simplex_weights(const V3* P, int n, int on, real* lam) {
  // what the lane needs from the table: triangles t = 0..3 (the triangle without point t), segments 0..5 in the
  // order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
  int need_tri = 0, need_seg = 0, tet_fall = 0;
  real k1 = 0, k2 = 0, k3 = 0, k4 = 0, vol = 1;
  if (on && n == 4) {
    k1 = -rw_det(P[1], P[2], P[3]);
    k2 = rw_det(P[0], P[2], P[3]);
    k3 = -rw_det(P[0], P[1], P[3]);
    k4 = rw_det(P[0], P[1], P[2]);
    vol = k1 + k2 + k3 + k4;
    const int in1 = rw_sign_match(vol, k1), in2 = rw_sign_match(vol, k2), in3 = rw_sign_match(vol, k3), in4 = rw_sign_match(vol, k4);
    if (!(in1 && in2 && in3 && in4)) { tet_fall = (in1 ? 0 : 1) | (in2 ? 0 : 2) | (in3 ? 0 : 4) | (in4 ? 0 : 8); need_tri = tet_fall; }
  } else if (on && n == 3) need_tri = 8;
  else if (on && n == 2) need_seg = 1;
  real tw[4][3];
  int tneed[4] = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; t++) {
    tw[t][0] = tw[t][1] = tw[t][2] = 0;
    if (!wv_any((need_tri >> t) & 1)) continue;
    const int i = t == 0 ? 1 : 0, j = t <= 1 ? 2 : 1, k = t == 3 ? 2 : 3;
    if ((need_tri >> t) & 1) {
      tneed[t] = lp_triangle(P[i], P[j], P[k], tw[t]);
      if (tneed[t] & 1) need_seg |= 1 << rw_edge_slot(j, k);
      if (tneed[t] & 2) need_seg |= 1 << rw_edge_slot(i, k);
      if (tneed[t] & 12) need_seg |= 1 << rw_edge_slot(i, j);
    }
  }
  real sw[6][2];
#pragma unroll
  for (int q = 0; q < 6; q++) {
    sw[q][0] = sw[q][1] = 0;
    if (!wv_any((need_seg >> q) & 1)) continue;
    const int a = q < 3 ? 0 : (q < 5 ? 1 : 2), b = q < 3 ? q + 1 : (q < 5 ? q - 1 : 3);
    if ((need_seg >> q) & 1) rw_segment_weights(P[a], P[b], sw[q][0], sw[q][1]);
  }
#pragma unroll
  for (int t = 0; t < 4; t++) {
    if (!wv_any(tneed[t] != 0)) continue;
    const int i = t == 0 ? 1 : 0, j = t <= 1 ? 2 : 1, k = t == 3 ? 2 : 3;
    if (tneed[t]) lp_triangle_fallback(tneed[t], P[i], P[j], P[k], sw[rw_edge_slot(j, k)], sw[rw_edge_slot(i, k)], sw[rw_edge_slot(i, j)], tw[t]);
  }
  if (!on) return;
  if (n == 2) { lam[0] = sw[0][0]; lam[1] = sw[0][1]; lam[2] = 0; lam[3] = 0; }
  else if (n == 3) { lam[0] = tw[3][0]; lam[1] = tw[3][1]; lam[2] = tw[3][2]; lam[3] = 0; }
  else if (!tet_fall) { lam[0] = k1/vol; lam[1] = k2/vol; lam[2] = k3/vol; lam[3] = k4/vol; }
  else {
    real r0 = 0, r1 = 0, r2 = 0, r3 = 0, nearest = RC_DBLMAX;
    if (tet_fall & 1) {
      const real a = tw[0][0], b = tw[0][1], c = tw[0][2];
      const V3 x{a*P[1].x + b*P[2].x + c*P[3].x, a*P[1].y + b*P[2].y + c*P[3].y, a*P[1].z + b*P[2].z + c*P[3].z};
      r0 = 0; r1 = a; r2 = b; r3 = c;
      nearest = dot(x, x);
    }
    if (tet_fall & 2) {
      const real a = tw[1][0], b = tw[1][1], c = tw[1][2];
      const V3 x{a*P[0].x + b*P[2].x + c*P[3].x, a*P[0].y + b*P[2].y + c*P[3].y, a*P[0].z + b*P[2].z + c*P[3].z};
      const real dd = dot(x, x);
      if (dd < nearest) { r0 = a; r1 = 0; r2 = b; r3 = c; nearest = dd; }
    }
    if (tet_fall & 4) {
      const real a = tw[2][0], b = tw[2][1], c = tw[2][2];
      const V3 x{a*P[0].x + b*P[1].x + c*P[3].x, a*P[0].y + b*P[1].y + c*P[3].y, a*P[0].z + b*P[1].z + c*P[3].z};
      const real dd = dot(x, x);
      if (dd < nearest) { r0 = a; r1 = b; r2 = 0; r3 = c; nearest = dd; }
    }
    if (tet_fall & 8) {
      const real a = tw[3][0], b = tw[3][1], c = tw[3][2];
      const V3 x{a*P[0].x + b*P[1].x + c*P[2].x, a*P[0].y + b*P[1].y + c*P[2].y, a*P[0].z + b*P[1].z + c*P[2].z};
      const real dd = dot(x, x);
      if (dd < nearest) { r0 = a; r1 = b; r2 = c; r3 = 0; }
    }
    lam[0] = r0; lam[1] = r1; lam[2] = r2; lam[3] = r3;
  }
}

// layout of a parked simplex in a record slot: 24 reals (4 x point on A, point on B), then ints: 8 vertex ids, nsim,
// the four support caches
MJH_DEV void lp_park(real* park, const LaneVert& s0, const LaneVert& s1, const LaneVert& s2, const LaneVert& s3) {
  int* pi = (int*)(park + 24);
  const LaneVert* v[4] = {&s0, &s1, &s2, &s3};
  for (int q = 0; q < 4; q++) {
    st3(park + 6*q, v[q]->pa); st3(park + 6*q + 3, v[q]->pb);
    pi[2*q] = v[q]->ia; pi[2*q + 1] = v[q]->ib;
  }
}
MJH_DEV LaneVert lp_unpark(const real* park, int q) {
  const int* pi = (const int*)(park + 24);
  LaneVert v;
  v.pa = ld3(park + 6*q); v.pb = ld3(park + 6*q + 3); v.ia = pi[2*q]; v.ib = pi[2*q + 1];
  return v;
}

// Distance phase of the polyhedral pairs list[0 .. npairs): every lane takes a pair, and a lane that has finished its pair
// takes the next one of the list at once (the pairs need 1 to ~13 iterations each: without the refill a round of 64 lasts
// as long as its longest pair).  Per list entry t: flag[t] = the shapes overlap, and then the simplex, its size and the
// support caches parked at park + 32 t for rc_poly_pair_penetration.  (gjk :198 with gjkIntersect :420 for pairs without
// margin: dist_cutoff 0, discrete geoms)
MJH_DEVN_HOT void ccd_poly_distance(MREF M_, BREF B_, int e_, int npairs) {
  MJH_ENTER(M_, B_, e_);
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  const PolyTab tab = rc_poly_tables(M, B, e);
  real* hold = rc_records(M, B, e, wv_lane());          // the lane's own scratch record (containment test)
  const int iters = M.s.ccd_N;
  const real tol = M.o.ccd_tolerance;
  npairs = wv_uniform_i(npairs);
  LaneShape a, b;
  LaneVert s0, s1, s2, s3;
  V3 x{0, 0, 0};
  real xlen = 0, xlen_before = 0;
  int n = 0, k = 0, try_containment = 1;
  int apart = 0, nsim = 0;
  real dist0 = 0;
  real lam[4] = {0, 0, 0, 0};
  int cur = -1;
  // 0 iterating, 1 closing support query pending, 2 finished (result to be stored), 3 no pair
  int st = 3;
  auto start = [&](int t) {
    cur = t;
    const int p = tab.list[t];
    const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
    auto shape = [&](int g) {
      LaneShape sh;
      sh.g = g; sh.vcache = -1; sh.gcache = -1; sh.mesh = -1; sh.kind = SK_BOX;
      if (M.geom_type[g] == MJH_GEOM_MESH) {
        sh.mesh = M.geom_dataid[g];
        sh.kind = (M.mesh_graphadr[sh.mesh] < 0 || M.mesh_vertnum[sh.mesh] < 10) ? SK_MESH_ALL : SK_MESH_CLIMB;
      }
      return sh;
    };
    a = shape(g1); b = shape(g2);
    s0.pa = s0.pb = V3{0, 0, 0}; s0.ia = s0.ib = -1;
    s1 = s0; s2 = s0; s3 = s0;
    x = ld3(gx + 3*g1) - ld3(gx + 3*g2);
    xlen = rw_len(x); xlen_before = 0;
    n = 0; k = 0; try_containment = 1; apart = 0; nsim = 0; dist0 = 0;
    lam[0] = lam[1] = lam[2] = lam[3] = 0;
    st = 0;
  };
  a.g = b.g = 0; a.kind = b.kind = SK_BOX; a.mesh = b.mesh = -1; a.vcache = a.gcache = b.vcache = b.gcache = -1;
  s0.pa = s0.pb = V3{0, 0, 0}; s0.ia = s0.ib = -1;
  s1 = s0; s2 = s0; s3 = s0;
  if (wv_lane() < npairs) start(wv_lane());
  int next_free = npairs < MJH_WAVE ? npairs : MJH_WAVE;
  while (wv_any(st != 3)) {
    if (st == 0) RC_COUNT(1);
    if (st == 0 && (!(k < iters) || xlen < MJH_MINVAL || fabs(xlen_before - xlen) < MJH_MINVAL)) st = 1;
    int it = st == 0;
    const int closing = st == 1;
    const V3 dn = rw_scl(x, 1/xlen);
    const LaneVert f = lp_pair_far(M, gx, gm, a, b, rw_scl(dn, -1), dn, it || closing);
    if (closing) {
      // the closing support query along x: apart after all?
      if (dot(x, lv_mink(f)) > 0) apart = 1;
      nsim = n;
      dist0 = (n == 4 && !apart) ? 0 : xlen;
      st = 2;
    }
    lv_put(n, it, s0, s1, s2, s3, f);
    const V3 s = lv_mink(f);
    if (it && dot(x, x - s) < 0) { st = 1; it = 0; }
    if (it && dot(x, s) > 0) { apart = 1; nsim = 0; dist0 = RC_DBLMAX; st = 2; it = 0; }
    // the tetrahedron test once the simplex has four points
    const int ct = it && n == 3 && try_containment;
    if (wv_any(ct)) {
      if (ct) lp_park(hold, s0, s1, s2, s3);
      int p0 = 0, p1 = 1, p2 = 2, p3 = 3, kk = k, ans = -1, run = ct;
      while (wv_any(run)) {
        if (run && !(kk < iters)) run = 0;
        const V3 m0 = lv_mink(s0), m1 = lv_mink(s1), m2 = lv_mink(s2), m3 = lv_mink(s3);
        real sd[4]; V3 nr[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int ia = q == 0 ? p2 : (q == 2 ? p1 : p0);
          const int ib = q == 0 ? p1 : (q == 1 ? p2 : (q == 2 ? p0 : p1));
          const int ic = q == 3 ? p2 : p3;
          const V3 pt = lv_pick(ia, m0, m1, m2, m3);
          V3 nrm = cross(lv_pick(ic, m0, m1, m2, m3) - pt, lv_pick(ib, m0, m1, m2, m3) - pt);
          const real n2 = dot(nrm, nrm);
          sd[q] = RC_DBLMAX;
          if (n2 > RC_TINY2 && n2 < RC_HUGE2) { nrm = rw_scl(nrm, 1/sqrt(n2)); sd[q] = dot(nrm, pt); }
          nr[q] = nrm;
        }
        if (run && (!sd[3] || !sd[2] || !sd[1] || !sd[0])) run = 0;
        const int lo = (sd[0] < sd[1]) ? 0 : 1, hi = (sd[2] < sd[3]) ? 2 : 3;
        const real dlo = lo ? sd[1] : sd[0], dhi = hi == 2 ? sd[2] : sd[3];
        const int near = (dlo < dhi) ? lo : hi;
        const real dnear = (dlo < dhi) ? dlo : dhi;
        if (run && dnear > 0) {
          const LaneVert t0 = lv_pickv(p0, s0, s1, s2, s3), t1 = lv_pickv(p1, s0, s1, s2, s3),
                         t2 = lv_pickv(p2, s0, s1, s2, s3), t3 = lv_pickv(p3, s0, s1, s2, s3);
          s0 = t0; s1 = t1; s2 = t2; s3 = t3;
          ans = 1; run = 0;
        }
        const V3 nrm = near == 0 ? nr[0] : (near == 1 ? nr[1] : (near == 2 ? nr[2] : nr[3]));
        const LaneVert g = lp_pair_far(M, gx, gm, a, b, nrm, V3{-nrm.x, -nrm.y, -nrm.z}, run);
        lv_put(rw_pick4(p0, p1, p2, p3, near), run, s0, s1, s2, s3, g);
        if (run && dot(nrm, lv_mink(g)) < 0) { ans = 0; run = 0; }
        if (run) {
          const int i = (near + 1) & 3, j = (near + 2) & 3;
          const int vi = rw_pick4(p0, p1, p2, p3, i), vj = rw_pick4(p0, p1, p2, p3, j);
          p0 = i == 0 ? vj : (j == 0 ? vi : p0);
          p1 = i == 1 ? vj : (j == 1 ? vi : p1);
          p2 = i == 2 ? vj : (j == 2 ? vi : p2);
          p3 = i == 3 ? vj : (j == 3 ? vi : p3);
          kk++;
        }
      }
      if (ct) {
        RC_COUNT(2);
        if (ans != -1) {
          apart = ans == 0; dist0 = ans > 0 ? 0 : RC_DBLMAX; nsim = ans > 0 ? 4 : 0;
          st = 2; it = 0;
        } else {
          // undecided: the simplex as it was, the iteration count where the test stopped
          s0 = lp_unpark(hold, 0); s1 = lp_unpark(hold, 1); s2 = lp_unpark(hold, 2); s3 = lp_unpark(hold, 3);
          k = kk;
          try_containment = 0;
        }
      }
    }
    if (wv_any(it && n > 0)) {
      const V3 P[4] = {lv_mink(s0), lv_mink(s1), lv_mink(s2), lv_mink(s3)};
      lp_simplex_weights(P, n + 1, it && n > 0, lam);
    }
    if (it) {
      if (n == 0) { lam[0] = 1; lam[1] = lam[2] = lam[3] = 0; }
      // keep the points that carry weight, in order: slot d takes the d-th point with a non-zero weight (selects only:
      // an array indexed by a running count would live in scratch memory)
      const int k0 = lam[0] != 0, k1 = lam[1] != 0, k2 = lam[2] != 0, k3 = lam[3] != 0;
      const int r1 = k0, r2 = k0 + k1, r3 = k0 + k1 + k2;          // rank of point i among the kept ones
      const int m = r3 + k3;
      const int i0 = k0 ? 0 : (k1 ? 1 : (k2 ? 2 : 3));
      const int i1 = (k1 && r1 == 1) ? 1 : ((k2 && r2 == 1) ? 2 : 3);
      const int i2 = (k2 && r2 == 2) ? 2 : 3;
      const real l0 = rw_pickr(lam, i0), l1 = rw_pickr(lam, i1), l2 = rw_pickr(lam, i2), l3 = lam[3];
      const LaneVert t0 = lv_pickv(i0, s0, s1, s2, s3), t1 = lv_pickv(i1, s0, s1, s2, s3), t2 = lv_pickv(i2, s0, s1, s2, s3);
      if (m > 0) { s0 = t0; lam[0] = l0; }
      if (m > 1) { s1 = t1; lam[1] = l1; }
      if (m > 2) { s2 = t2; lam[2] = l2; }
      if (m > 3) lam[3] = l3;
      n = m;
      if (n < 1) { nsim = 0; dist0 = RC_DBLMAX; apart = 1; st = 2; }
      else {
        const V3 p0 = lv_mink(s0), p1 = lv_mink(s1), p2 = lv_mink(s2), p3 = lv_mink(s3);
        if (n == 1) x = V3{lam[0]*p0.x, lam[0]*p0.y, lam[0]*p0.z};
        else if (n == 2) x = V3{lam[0]*p0.x + lam[1]*p1.x, lam[0]*p0.y + lam[1]*p1.y, lam[0]*p0.z + lam[1]*p1.z};
        else if (n == 3) x = V3{lam[0]*p0.x + lam[1]*p1.x + lam[2]*p2.x, lam[0]*p0.y + lam[1]*p1.y + lam[2]*p2.y,
                                lam[0]*p0.z + lam[1]*p1.z + lam[2]*p2.z};
        else x = V3{lam[0]*p0.x + lam[1]*p1.x + lam[2]*p2.x + lam[3]*p3.x, lam[0]*p0.y + lam[1]*p1.y + lam[2]*p2.y + lam[3]*p3.y,
                    lam[0]*p0.z + lam[1]*p1.z + lam[2]*p2.z + lam[3]*p3.z};
        xlen_before = xlen;
        xlen = rw_len(x);
        if (n == 4) st = 1; else k++;
      }
    }
    // finished pairs: store the result, take the next pair of the list
    const int fin = st == 2;
    const unsigned long long finished = wv_ballot(fin);
    if (fin) {
      RC_COUNT(0);
      const int overlap = dist0 <= tol && nsim > 1 && !apart;
      tab.flag[cur] = overlap;
      if (overlap) {
        real* park = tab.park + 32*(size_t)cur;
        lp_park(park, s0, s1, s2, s3);
        int* pi = (int*)(park + 24);
        pi[8] = nsim; pi[9] = a.vcache; pi[10] = a.gcache; pi[11] = b.vcache; pi[12] = b.gcache;
      }
      const int t = next_free + wv_rank_lt(finished);
      if (t < npairs) start(t); else { st = 3; cur = -1; }
    }
    next_free += __builtin_popcountll(finished);
  }
  wv_sync();
}

// Every lane of the wavefront brings (at most) one pair: the pairs are listed, row r takes entries r, r + 4, ... of the
// list, the results go to the owning lane's records.  Polyhedral pairs (box / mesh against box / mesh, no margin: the
// pairs that may return several contacts) go through TWO passes: the distance query for all of them, then -- the
// overlapping ones compacted, so that all four rows are busy -- polytope expansion and face clipping.  (In the 3x3x3
// cube a third of the pairs in reach overlap, and the second pass is four times the work of the first.)
// Returns the calling lane's contact count.
MJH_DEVN_HOT int ccd_convex_pair(MREF M_, BREF B_, int e_, int p) {
  MJH_ENTER(M_, B_, e_);
  int* head = rc_header(M, B, e);
  const int row = wv_lane() >> 4;
  const unsigned long long have = wv_ballot(p >= 0);
  const int total = __builtin_popcountll(have);
  if (p >= 0) { const int t = wv_rank_lt(have); head[t] = p; head[64 + t] = wv_lane(); }
  head[128 + wv_lane()] = 0;
  wv_sync();
#ifdef MJH_PROFILE
  // sub-stage accumulators (us): 25 single-contact pairs, 46 lane-parallel distance phase, 47 row-cooperative penetration phase
  long long ptick = wv_clock();
  auto tick = [&](int slot) { const long long c_ = wv_clock(); if (wv_lane() == 0) MJH_G(B, prof, e)[slot] += (real)(c_ - ptick)*0.01; ptick = c_; };
#else
  auto tick = [](int) {};
#endif
  RowPair c;
  rc_attach(M, B, e, c);
  // first the distance phase of the polyhedral pairs, THEN the single-contact pairs: the lane-parallel distance phase uses
  // the record of the lane it runs on as scratch (ccd_poly_distance: `hold`), and lane k works on the k-th polyhedral pair,
  // not on its own -- run after the single-contact pairs it overwrote the finished contact of a curved pair owned by
  // that lane (round 6: discardvisual.xml, a sphere : mesh pair next to a mesh : mesh pair read dist 1.2 instead of -0.199)
  // one polyhedral pair per lane, simplex in registers -- unless stage_collision ran it for all
  // pairs in reach at once (ccd_poly_prepass: head[193] set), then the results are looked up
  const PolyTab tab = rc_poly_tables(M, B, e);
  int entry = -1;                               // my pair's entry in the tables
  {
    const int poly = p >= 0 && rc_max_contacts(M, p) > 1;
    const unsigned long long polys = wv_ballot(poly);
    if (polys) {
      if (head[193]) { if (poly) entry = tab.slot[p]; }
      else {
        if (poly) { entry = wv_rank_lt(polys); tab.list[entry] = p; }
        wv_sync();
        ccd_poly_distance(M, B, e, __builtin_popcountll(polys));
      }
      if (poly) head[128 + wv_lane()] = tab.flag[entry] ? -1 : 0;
    }
  }
  wv_sync();
  // (single-contact pairs -- curved shapes, margins -- take the whole query at once, four at a time)
  for (int t = row; t < total; t += 4) {
    const int owner = head[64 + t], pp = head[t];
    if (rc_max_contacts(M, pp) > 1) continue;
    const int n = rc_geom_pair(M, B, e, c, pp, rc_records(M, B, e, owner));
    if (rw_l() == 0) head[128 + owner] = n;
  }
  wv_converge();
  wv_sync();
  tick(46);
  // second pass over the pairs marked -1
  const unsigned long long deep = wv_ballot(head[128 + wv_lane()] < 0);
  const int ndeep = __builtin_popcountll(deep);
  if (ndeep) {
    wv_sync();
    if ((deep >> wv_lane()) & 1) { const int t = wv_rank_lt(deep); head[t] = p; head[64 + t] = wv_lane() | (entry << 8); }
    wv_sync();
    for (int t = row; t < ndeep; t += 4) {
      const int owner = head[64 + t] & 63;
      const int n = rc_poly_pair_penetration(M, B, e, c, head[t], tab.park + 32*(size_t)(head[64 + t] >> 8), rc_records(M, B, e, owner));
      if (rw_l() == 0) head[128 + owner] = n;
    }
    wv_converge();
    wv_sync();
  }
#ifdef MJH_PROFILE
  if (wv_lane() == 0) { MJH_G(B, prof, e)[25] += (real)c.t_far*0.01; MJH_G(B, prof, e)[21] += (real)c.t_clip*0.01; }
#endif
  tick(47);
  return p >= 0 ? head[128 + wv_lane()] : 0;
}

// The listed (geom, flex element) pairs, one per row: entry t is taken by row t mod (rows of the group).  The group is the
// wavefront (4 rows) -- or, in a multi-wavefront workgroup, the workgroup (mjh_modes.h: this function is one of the
// stages every wavefront runs, 4 rows each).
MJH_DEVN void rc_elem_rows(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  int* head = rc_header(M, B, e);
  const int total = head[192];
  RowPair c;
  rc_attach(M, B, e, c);
  for (int t = wv_lane() >> 4; t < total; t += MJH_W/16) {
    const int owner = (head[t] >> 24) & 63;
    real* rec = rc_records(M, B, e, owner);
    const real mg = rec[0];
    wv_row_sync();
    // (entry: geom | owner << 24, element; geom = 0xffffff: two elements, the first one's id in the owner's second record slot)
    const int g = head[t] & 0xffffff;
    const int n = g == 0xffffff ? rc_elem_elem(M, B, e, c, (int)rec[1], head[64 + t], mg, rec)
                                : rc_geom_elem(M, B, e, c, g, head[64 + t], mg, rec);
    if (rw_l() == 0) head[128 + owner] = n;
  }
  MJH_GROUP_JOIN();
}
// (geom, flex element) pairs, one per lane of the calling wavefront; g < 0: the lane has none.  Returns the lane's contact count.
MJH_DEVN_HOT int ccd_geom_elem_pair(MREF M_, BREF B_, int e_, int g, int elem, real margin) {
  MJH_ENTER(M_, B_, e_);
  int* head = rc_header(M, B, e);
  const unsigned long long have = wv_ballot(g >= 0);
  const int total = __builtin_popcountll(have);
  // (the list entries: geom, element, owning lane; margins in the first record slot of the owner)
  if (g >= 0) {
    const int t = wv_rank_lt(have);
    head[t] = g | (wv_lane() << 24); head[64 + t] = elem;
    rc_records(M, B, e, wv_lane())[0] = margin;
  }
  if (wv_lane() == 0) head[192] = total;
  wv_sync();
  MJH_WIDE(MJH_MWS_ELEMS, rc_elem_rows(M, B, e));
  return g >= 0 ? head[128 + wv_lane()] : 0;
}
// pairs of flex elements (global ids), one per lane of the calling wavefront; elem1 < 0: the lane has none
MJH_DEVN_HOT int ccd_elem_elem_pair(MREF M_, BREF B_, int e_, int elem1, int elem2, real margin) {
  MJH_ENTER(M_, B_, e_);
  int* head = rc_header(M, B, e);
  const unsigned long long have = wv_ballot(elem1 >= 0);
  const int total = __builtin_popcountll(have);
  if (elem1 >= 0) {
    const int t = wv_rank_lt(have);
    head[t] = 0xffffff | (wv_lane() << 24); head[64 + t] = elem2;
    real* rec = rc_records(M, B, e, wv_lane());
    rec[0] = margin; rec[1] = (real)elem1;
  }
  if (wv_lane() == 0) head[192] = total;
  wv_sync();
  MJH_WIDE(MJH_MWS_ELEMS, rc_elem_rows(M, B, e));
  return elem1 >= 0 ? head[128 + wv_lane()] : 0;
}

// mjc_PlaneConvex (:1004): plane against ellipsoid / mesh -- the point of the geom deepest below the plane, plus up to
// two more mesh vertices below it (candidates one per lane, taken in list order).  No distance query is needed.
MJH_DEV int rc_plane_pair(MREF M, BREF B, int e, RowPair& c, int p, real* rec) {
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  const int L = rw_l();
  const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
  const real margin = M.pair_margin[p];
  const V3 pos1 = ld3(gx + 3*g1), pos2 = ld3(gx + 3*g2);
  crptr mat1 = gm + 9*g1;
  const V3 normal{mat1[2], mat1[5], mat1[8]};
  c.a = rp_load_geom(M, c, 0, gx, gm, g2, 0);
  wv_row_sync();
  const real* f = rp_frame(c, 0);
  const real* mat2 = f + FR_MAT; const real* size = f + FR_SIZE;
  const V3 down{-mat1[2], -mat1[5], -mat1[8]};
  const V3 ld = rw_to_local(mat2, down);
  // the geom's farthest point against the plane normal (mjccd_support :518)
  V3 res;
  int node = -1;                      // mesh: vertex (exhaustive search) or hull-graph node (hill climbing) that was found
  if (c.a.type == MJH_GEOM_ELLIPSOID) {
    res = V3{ld.x*size[0], ld.y*size[1], ld.z*size[2]};
    unitize(res);
    res = V3{res.x*size[0], res.y*size[1], res.z*size[2]};
  } else {
    const int mesh = c.a.mesh;
    const int vadr = 3*M.mesh_vertadr[mesh];
    int vbest = -1;
    if (c.a.kind == SK_MESH_ALL) {
      const int nvert = M.mesh_vertnum[mesh];
      real top = -1E+10;
      for (int k0 = 0; k0 < nvert; k0 += 16) {
        const int k = k0 + L;
        real v = k < nvert ? rw_vdot(M, ld, vadr + 3*k) : -HUGE_VAL;
        int iv = k < nvert ? k : RC_NONE;
        rw_first_max<4>(v, iv);
        if (v > top) { top = v; vbest = iv; }
      }
      node = vbest;
    } else {
      const int gadr = M.mesh_graphadr[mesh];
      node = rw_hill_climb(M, mesh, ld, 0);
      wv_row_converge();
      vbest = M.mesh_graph[gadr + 2 + M.mesh_graph[gadr] + node];
    }
    res = vbest < 0 ? V3{0, 0, 0} : rw_mesh_vert(M, vadr + 3*vbest);
  }
  res = V3{res.x + ld.x*c.a.margin/2, res.y + ld.y*c.a.margin/2, res.z + ld.z*c.a.margin/2};
  res = mmul(mat2, res);
  const V3 sup{res.x + f[FR_POS], res.y + f[FR_POS + 1], res.z + f[FR_POS + 2]};
  const real depth = dot(normal, sup - pos1);
  if (depth > margin) return 0;
  const real half = -0.5*depth;
  const V3 first{sup.x + normal.x*half, sup.y + normal.y*half, sup.z + normal.z*half};
  if (L == 0) { rec[0] = depth; st3(rec + 1, first); st3(rec + 4, normal); }
  int count = 1;
  if (M.geom_dataid[g2] != -1) {
    const int mesh = M.geom_dataid[g2];
    const int vadr = 3*M.mesh_vertadr[mesh];
    const real threshold = dot(normal, pos2 - pos1) - margin;
    const real rbound = M.geom_rbound[g2];
    // candidate list: all vertices (no hull graph), else the hull neighbours of the deepest vertex
    int list0 = 0, nlist = 0, graph = 0, globalid = 0;
    if (M.mesh_graphadr[mesh] < 0) nlist = M.mesh_vertnum[mesh];
    else if (node >= 0) {
      const int gadr = M.mesh_graphadr[mesh];
      const int numvert = M.mesh_graph[gadr], numface = M.mesh_graph[gadr + 1];
      globalid = gadr + 2 + numvert;
      const int localid = gadr + 2 + 2*numvert;
      list0 = localid + M.mesh_graph[gadr + 2 + node];
      nlist = localid + numvert + 3*numface - list0;      // (the list ends at its terminator)
      graph = 1;
    }
    for (int i0 = 0; i0 < nlist && count < 3; i0 += 16) {
      const int i = i0 + L;
      int v = -1;
      if (i < nlist) v = graph ? M.mesh_graph[list0 + i] : i;
      const unsigned ends = graph ? wv_row_ballot(i < nlist && v < 0) : 0u;
      const int valid = i < nlist && (!ends || L < __builtin_ctz(ends));
      int ok = 0;
      V3 pnt{0, 0, 0};
      if (valid) {
        const int vb = vadr + 3*(graph ? M.mesh_graph[globalid + v] : v);
        const real along = ld.x*(real)M.mesh_vert[vb] + ld.y*(real)M.mesh_vert[vb + 1] + ld.z*(real)M.mesh_vert[vb + 2];
        if (along > threshold && (graph || v != node)) {
          // (addplanemesh :970: not within 0.3 rbound of the first contact)
          pnt = mmul(mat2, rw_mesh_vert(M, vb)) + pos2;
          const V3 df = pnt - first;
          ok = !(sqrt(df.x*df.x + df.y*df.y + df.z*df.z) < 0.3*rbound);
        }
      }
      unsigned take = wv_row_ballot(ok);
      while (take && count < 3) {
        const int at = __builtin_ctz(take);
        take &= take - 1;
        if (L == at) {
          real* oc = rec + RC_RECORD*count;
          oc[0] = dot(normal, pnt - pos1);
          const real hh = -0.5*oc[0];
          st3(oc + 1, V3{pnt.x + normal.x*hh, pnt.y + normal.y*hh, pnt.z + normal.z*hh});
          st3(oc + 4, normal);
        }
        count++;
      }
      if (ends) break;
    }
  }
  wv_row_sync();
  return count;
}

MJH_DEVN_HOT int ccd_plane_convex_pair(MREF M_, BREF B_, int e_, int p) {
  MJH_ENTER(M_, B_, e_);
  int* head = rc_header(M, B, e);
  const unsigned long long have = wv_ballot(p >= 0);
  const int total = __builtin_popcountll(have);
  if (p >= 0) { const int t = wv_rank_lt(have); head[t] = p; head[64 + t] = wv_lane(); }
  wv_sync();
  RowPair c;
  rc_attach(M, B, e, c);
  for (int t = wv_lane() >> 4; t < total; t += 4) {
    const int owner = head[64 + t];
    const int n = rc_plane_pair(M, B, e, c, head[t], rc_records(M, B, e, owner));
    if (rw_l() == 0) head[128 + owner] = n;
  }
  wv_converge();
  wv_sync();
  return p >= 0 ? head[128 + wv_lane()] : 0;
}

#endif  // !MJH_LANE_MODE
