// General convex narrowphase: GJK + EPA + multi-contact face clipping (mjc_Convex / mjc_PlaneConvex).
//
// Reference: engine_collision_convex.c (mjc_Convex :881, mjc_PlaneConvex :1004, support functions
// :201-460, mjccd_support :518), engine_collision_gjk.c (gjk :198, gjkIntersect :420, subdistance
// :586-880, polytope2/3/4 :948-1215, epa :1358, multicontact :2123, mjc_ccd :2318).
//
// Mapping: ONE GEOM PAIR PER LANE.  The algorithm is a chain of data-dependent branches on a
// handful of 3-vectors; the parallelism of a contact-rich scene is ACROSS pairs (BASELINE config 4,
// the 3x3x3 cube: ~100 mesh-mesh pairs reach the narrowphase every step), so the 64 lanes of the
// wavefront that owns the environment each run their own pair's GJK/EPA to completion, in lockstep
// where their control flow agrees and masked where it does not.  A wave-cooperative form (one pair at
// a time, lanes over support candidates / polytope faces) was rejected on numbers: a cubelet has 24
// vertices and its hill-climbing support touches 3-5 of them, an EPA run visits ~10 faces -- there is
// nothing for 64 lanes to share, and the pairs would queue behind each other.
//
// State: every lane owns a private slice of the batch's `ccd_ws` workspace (global memory; sized by
// the model: opt.ccd_iterations, npolygonmax, nmeshdegmax): the two objects, the GJK simplex, the
// EPA polytope (vertices, faces, the face map, the horizon) and, overlaid on the polytope once EPA
// is done, the face-clipping buffers.  The slices of an environment's 64 lanes are INTERLEAVED word
// by word (word w of lane l at [w][l]): lanes that touch the same slot -- object frames, the early
// simplex / polytope entries, the clipping buffers: most of the traffic -- read and write one
// contiguous 512-byte row instead of 64 cache lines 20 KB apart (measured on the cube: the latter
// made the narrowphase L1-throughput bound).  Nothing here touches LDS or other lanes.
//
// Arithmetic follows the reference expression by expression (association, comparison direction,
// first-wins tie rules), which is what makes contact counts and iteration counts agree exactly.
// (included once per SPMD mode by mjh_modes.h, after mjh_collision.h -- no include guard)

#if !MJH_LANE_MODE

#define MJH_CCD_MINVAL2 (MJH_MINVAL*MJH_MINVAL)
#define MJH_CCD_MAXVAL2 (MJH_MAXVAL*MJH_MAXVAL)
#define MJH_CCD_MAX 1.7976931348623157e308       // mjMAX_LIMIT = DBL_MAX
#define MJH_CCD_FLTMAX 3.4028234663852886e38     // (double)FLT_MAX
#define MJH_CCD_FACE_TOL 0.996                   // mjFACE_TOL, engine_collision_gjk.h:42
#define MJH_CCD_EDGE_TOL 0.0888                  // mjEDGE_TOL

enum { CCD_SUP_POINT = 0, CCD_SUP_SPHERE, CCD_SUP_LINE, CCD_SUP_CAPSULE, CCD_SUP_ELLIPSOID, CCD_SUP_CYLINDER,
       CCD_SUP_BOX, CCD_SUP_MESH, CCD_SUP_HILLCLIMB, CCD_SUP_FLEXELEM };
// object slots: reals pos[3] mat[9] size[3] margin centre[3] -; ints below
// (a flex element keeps its corner positions in pos | mat (4 x 3), its radius + margin/2 in size[0], its corner count in
// the CI_MESH slot and the centre mjc_center returns -- the centre of its bounding box -- in centre)
enum { CO_POS = 0, CO_MAT = 3, CO_SIZE = 12, CO_MARGIN = 15, CO_CENTER = 16, CO_NREAL = 20 };
#define MJH_GEOM_FLEX 100
enum { CI_TYPE = 0, CI_SUP = 1, CI_VERTINDEX = 2, CI_MESHINDEX = 3, CI_MESH = 4, CI_NINT = 6 };
// vertex slots: reals vert[3] (Minkowski difference) vert1[3] vert2[3]; ints index1 index2
enum { CV_NREAL = 9, CV_NINT = 2 };
// face slots: reals v[3] dist2; ints verts(packed 3 x 10 bit) adj[3] index
enum { CF_NREAL = 4, CF_NINT = 5 };
enum { CCD_MAXWIT = 4, CCD_MAXOUT = 5 };

struct CcdObj { rptr r; iptr i; };
struct CcdVtx { rptr r; iptr i; };

// the lane's workspace, carved up (host mirror of the sizes: mjh_model_build.h ccd_sizes)
struct Ccd {
  CcdObj o1, o2;
  rptr x1, x2, dist;               // witness points / distances (CCD_MAXWIT)
  rptr simr; iptr simi;            // GJK simplex: 4 vertices
  rptr tmpr; iptr tmpi;            // scratch: 5 vertices (gjkIntersect's copy + the final separation probe)
  rptr out;                        // contacts handed back: CCD_MAXOUT x (dist, pos[3], normal[3])
  rptr vr; iptr vi;                // polytope vertices
  rptr fr; iptr fi;                // polytope faces
  iptr map;                        // face map
  iptr hidx, hedge;                // horizon
  iptr stack;                      // horizon depth-first stack
  rptr mcr; iptr mci;              // multicontact buffers (overlay the polytope)
  // configuration / status scalars
  int N, P, D;                     // ccd_iterations, polygon size bound, vertex degree bound
  int maxfaces, maxhorizon;
  real tolerance;
  int max_contacts;
  real dist_cutoff;
  int separated, nx, nsimplex, gjk_iterations;
  int nverts, nfaces, nmap, nedges;
  V3 center;
  V3 horizon_w;
};

MJH_DEV CcdVtx ccd_vtx(rptr r, iptr i, int k) { return CcdVtx{r + CV_NREAL*k, i + CV_NINT*k}; }
MJH_DEV void ccd_vcopy(CcdVtx d, CcdVtx s) {
  for (int k = 0; k < CV_NREAL; k++) d.r[k] = s.r[k];
  d.i[0] = s.i[0]; d.i[1] = s.i[1];
}
MJH_DEV V3 ccd_scl(V3 v, real s) { return V3{s*v.x, s*v.y, s*v.z}; }     // scl3: s*v[k]
MJH_DEV real ccd_det3(V3 a, V3 b, V3 c) { return a.x*(b.y*c.z - b.z*c.y) + a.y*(b.z*c.x - b.x*c.z) + a.z*(b.x*c.y - b.y*c.x); }
MJH_DEV real ccd_norm(V3 v) { return sqrt(dot(v, v)); }
MJH_DEV real ccd_abs(real x) { return fabs(x); }

// mat' * dir and mat * l + pos (mulMatTVec3 / localToGlobal, engine_collision_convex.c:179-197)
template <class PM> MJH_DEV V3 ccd_to_local(PM mat, V3 d) {
  return V3{mat[0]*d.x + mat[3]*d.y + mat[6]*d.z, mat[1]*d.x + mat[4]*d.y + mat[7]*d.z, mat[2]*d.x + mat[5]*d.y + mat[8]*d.z};
}
template <class PM, class PP> MJH_DEV V3 ccd_to_global(PM mat, V3 l, PP pos) {
  V3 r{mat[0]*l.x + mat[1]*l.y + mat[2]*l.z, mat[3]*l.x + mat[4]*l.y + mat[5]*l.z, mat[6]*l.x + mat[7]*l.y + mat[8]*l.z};
  r.x += pos[0]; r.y += pos[1]; r.z += pos[2];
  return r;
}
// globalcoord (engine_collision_gjk.c:1756): mat * (l1,l2,l3) (+ pos)
template <class PM> MJH_DEV V3 ccd_globalrot(PM mat, real l1, real l2, real l3) {
  return V3{mat[0]*l1 + mat[1]*l2 + mat[2]*l3, mat[3]*l1 + mat[4]*l2 + mat[5]*l3, mat[6]*l1 + mat[7]*l2 + mat[8]*l3};
}
template <class PM, class PP> MJH_DEV V3 ccd_globalcoord(PM mat, PP pos, real l1, real l2, real l3) {
  V3 r = ccd_globalrot(mat, l1, l2, l3);
  r.x += pos[0]; r.y += pos[1]; r.z += pos[2];
  return r;
}
MJH_DEV real ccd_dot3f(MREF M, V3 a, int vbase) {
  return a.x*(real)M.mesh_vert[vbase] + a.y*(real)M.mesh_vert[vbase + 1] + a.z*(real)M.mesh_vert[vbase + 2];
}

// ---- support functions (engine_collision_convex.c:201-460) ------------------------------------------
MJH_DEVN_HOT void ccd_obj_support(MREF M, CcdObj o, V3 dir, rptr res) {
  const crptr pos = o.r + CO_POS; const crptr mat = o.r + CO_MAT; const crptr size = o.r + CO_SIZE;
  V3 out;
  switch (o.i[CI_SUP]) {
    case CCD_SUP_POINT: out = ld3(pos); break;
    case CCD_SUP_SPHERE: {
      const real radius = size[0];
      out = V3{radius*dir.x + pos[0], radius*dir.y + pos[1], radius*dir.z + pos[2]};
      break;
    }
    case CCD_SUP_FLEXELEM: {
      // mjc_flexSupport (engine_collision_convex.c:480-506): the corner with the largest projection (first wins), pushed
      // out along dir by the flex radius plus half the margin
      const int n = o.i[CI_MESH];
      out = ld3(o.r);
      real best = out.x*dir.x + out.y*dir.y + out.z*dir.z;
      for (int i = 1; i < n; i++) {
        const V3 v = ld3(o.r + 3*i);
        const real d = v.x*dir.x + v.y*dir.y + v.z*dir.z;
        if (d > best) { best = d; out = v; }
      }
      const real scl = size[0];
      out = V3{out.x + dir.x*scl, out.y + dir.y*scl, out.z + dir.z*scl};
      break;
    }
    case CCD_SUP_LINE: {
      const real length = size[1];
      const real d = mat[2]*dir.x + mat[5]*dir.y + mat[8]*dir.z;
      const real scl = d >= 0 ? length : -length;
      out = V3{mat[2]*scl + pos[0], mat[5]*scl + pos[1], mat[8]*scl + pos[2]};
      break;
    }
    case CCD_SUP_CAPSULE: {
      const real radius = size[0], length = size[1];
      const V3 ld = ccd_to_local(mat, dir);
      V3 ls{ld.x*radius, ld.y*radius, ld.z*radius};
      ls.z += (ld.z >= 0 ? length : -length);
      out = ccd_to_global(mat, ls, pos);
      break;
    }
    case CCD_SUP_ELLIPSOID: {
      const V3 ld = ccd_to_local(mat, dir);
      V3 ls{ld.x*size[0], ld.y*size[1], ld.z*size[2]};
      const real norm2 = ls.x*ls.x + ls.y*ls.y + ls.z*ls.z;
      if (norm2 < MJH_CCD_MINVAL2) {
        out = V3{mat[0]*size[0] + pos[0], mat[3]*size[0] + pos[1], mat[6]*size[0] + pos[2]};
        break;
      }
      const real norm_inv = 1/sqrt(norm2);
      ls.x *= norm_inv*size[0]; ls.y *= norm_inv*size[1]; ls.z *= norm_inv*size[2];
      out = ccd_to_global(mat, ls, pos);
      break;
    }
    case CCD_SUP_CYLINDER: {
      const V3 ld = ccd_to_local(mat, dir);
      const real n2 = ld.x*ld.x + ld.y*ld.y;
      const real scl = n2 >= MJH_CCD_MINVAL2 ? size[0]/sqrt(n2) : 0;
      const V3 ls{scl*ld.x, scl*ld.y, ld.z >= 0 ? size[1] : -size[1]};
      out = ccd_to_global(mat, ls, pos);
      break;
    }
    case CCD_SUP_BOX: {
      const V3 ld = ccd_to_local(mat, dir);
      const V3 ls{ld.x >= 0 ? size[0] : -size[0], ld.y >= 0 ? size[1] : -size[1], ld.z >= 0 ? size[2] : -size[2]};
      int vi = (ls.x > 0) ? 1 : 0;
      vi |= (ls.y > 0) ? 2 : 0;
      vi |= (ls.z > 0) ? 4 : 0;
      o.i[CI_VERTINDEX] = vi;
      out = ccd_to_global(mat, ls, pos);
      break;
    }
    case CCD_SUP_MESH: {
      // exhaustive search, first maximum wins, warm-started from the cached vertex (:354)
      const int mesh = o.i[CI_MESH];
      const int vadr = 3*M.mesh_vertadr[mesh], nverts = M.mesh_vertnum[mesh];
      const V3 ld = ccd_to_local(mat, dir);
      real max = -MJH_CCD_FLTMAX;
      int imax = 0;
      if (o.i[CI_VERTINDEX] >= 0) { imax = o.i[CI_VERTINDEX]; max = ccd_dot3f(M, ld, vadr + 3*imax); }
      for (int k = 0; k < nverts; k++) {
        const real vdot = ccd_dot3f(M, ld, vadr + 3*k);
        if (vdot > max) { max = vdot; imax = k; }
      }
      o.i[CI_VERTINDEX] = imax;
      const V3 lv{(real)M.mesh_vert[vadr + 3*imax], (real)M.mesh_vert[vadr + 3*imax + 1], (real)M.mesh_vert[vadr + 3*imax + 2]};
      out = ccd_to_global(mat, lv, pos);
      break;
    }
    default: {
      // hill climbing over the hull graph, seeded from a 3x3x3 direction grid or the cached vertex (:396)
      const int mesh = o.i[CI_MESH];
      const int vadr = 3*M.mesh_vertadr[mesh];
      const int gadr = M.mesh_graphadr[mesh];
      const int numvert = M.mesh_graph[gadr];
      const int edgeadr = gadr + 2, globalid = gadr + 2 + numvert, localid = gadr + 2 + 2*numvert;
      const V3 ld = ccd_to_local(mat, dir);
      const int cx = (ld.x > 0.4) - (ld.x < -0.4) + 1;
      const int cy = (ld.y > 0.4) - (ld.y < -0.4) + 1;
      const int cz = (ld.z > 0.4) - (ld.z < -0.4) + 1;
      const int grid_idx = M.mesh_extrema[27*mesh + cx*9 + cy*3 + cz];
      int imax;
      if (o.i[CI_MESHINDEX] >= 0) {
        const real cached = ccd_dot3f(M, ld, vadr + 3*M.mesh_graph[globalid + o.i[CI_MESHINDEX]]);
        const real seed = ccd_dot3f(M, ld, vadr + 3*M.mesh_graph[globalid + grid_idx]);
        imax = (seed > cached) ? grid_idx : o.i[CI_MESHINDEX];
      } else {
        imax = grid_idx;
      }
      real max = ccd_dot3f(M, ld, vadr + 3*M.mesh_graph[globalid + imax]);
      int prev = -1;
      while (imax != prev) {
        prev = imax;
        int sub;
        for (int k = M.mesh_graph[edgeadr + imax]; (sub = M.mesh_graph[localid + k]) >= 0; k++) {
          const real vdot = ccd_dot3f(M, ld, vadr + 3*M.mesh_graph[globalid + sub]);
          if (vdot > max) { max = vdot; imax = sub; }
        }
      }
      o.i[CI_MESHINDEX] = imax;
      const int gi = M.mesh_graph[globalid + imax];
      o.i[CI_VERTINDEX] = gi;
      const V3 lv{(real)M.mesh_vert[vadr + 3*gi], (real)M.mesh_vert[vadr + 3*gi + 1], (real)M.mesh_vert[vadr + 3*gi + 2]};
      out = ccd_to_global(mat, lv, pos);
      break;
    }
  }
  st3(res, out);
}

// support point of the Minkowski difference (support, engine_collision_gjk.c:337)
MJH_DEV void ccd_support(MREF M, Ccd& c, CcdVtx v, V3 dir, V3 dir_neg) {
  ccd_obj_support(M, c.o1, dir, v.r + 3);
  if (c.o1.r[CO_MARGIN] > 0) {
    const real margin = 0.5*c.o1.r[CO_MARGIN];
    v.r[3] += dir.x*margin; v.r[4] += dir.y*margin; v.r[5] += dir.z*margin;
  }
  ccd_obj_support(M, c.o2, dir_neg, v.r + 6);
  if (c.o2.r[CO_MARGIN] > 0) {
    const real margin = 0.5*c.o2.r[CO_MARGIN];
    v.r[6] += dir_neg.x*margin; v.r[7] += dir_neg.y*margin; v.r[8] += dir_neg.z*margin;
  }
  v.r[0] = v.r[3] - v.r[6]; v.r[1] = v.r[4] - v.r[7]; v.r[2] = v.r[5] - v.r[8];
  v.i[0] = c.o1.i[CI_VERTINDEX];
  v.i[1] = c.o2.i[CI_VERTINDEX];
}
MJH_DEV void ccd_gjk_support(MREF M, Ccd& c, CcdVtx v, V3 xk, real xnorm) {
  const V3 dir_neg = ccd_scl(xk, 1/xnorm);
  const V3 dir = ccd_scl(dir_neg, -1);
  ccd_support(M, c, v, dir, dir_neg);
}
// epaSupport (:384): new polytope vertex along d
MJH_DEV int ccd_epa_support(MREF M, Ccd& c, V3 d, real dnorm) {
  V3 dir{1, 0, 0}, dir_neg{-1, 0, 0};
  if (dnorm > MJH_MINVAL) {
    dir = V3{d.x/dnorm, d.y/dnorm, d.z/dnorm};
    dir_neg = ccd_scl(dir, -1);
  }
  const int n = c.nverts++;
  ccd_support(M, c, ccd_vtx(c.vr, c.vi, n), dir, dir_neg);
  return n;
}

// ---- distance sub-algorithm (signed volumes; :520-880) ------------------------------------------------
MJH_DEV int ccd_same_sign(real a, real b) {
  if (a > 0 && b > 0) return 1;
  if (a < 0 && b < 0) return -1;
  return 0;
}
MJH_DEV V3 ccd_lincomb2(const real* l, V3 a, V3 b) { return V3{l[0]*a.x + l[1]*b.x, l[0]*a.y + l[1]*b.y, l[0]*a.z + l[1]*b.z}; }
MJH_DEV V3 ccd_lincomb3(const real* l, V3 a, V3 b, V3 d) {
  return V3{l[0]*a.x + l[1]*b.x + l[2]*d.x, l[0]*a.y + l[1]*b.y + l[2]*d.y, l[0]*a.z + l[1]*b.z + l[2]*d.z};
}
// lincomb (:475): n-term combination of the rows of a vertex array (row stride `st`, column offset in p)
template <class PV> MJH_DEV V3 ccd_lincomb(const real* l, int n, PV p, int st) {
  V3 r{0, 0, 0};
  if (n == 1) r = V3{l[0]*p[0], l[0]*p[1], l[0]*p[2]};
  else if (n == 2) r = ccd_lincomb2(l, ld3(p), ld3(p + st));
  else if (n == 3) r = ccd_lincomb3(l, ld3(p), ld3(p + st), ld3(p + 2*st));
  else if (n == 4) {
    const V3 a = ld3(p), b = ld3(p + st), d = ld3(p + 2*st), e = ld3(p + 3*st);
    r = V3{l[0]*a.x + l[1]*b.x + l[2]*d.x + l[3]*e.x, l[0]*a.y + l[1]*b.y + l[2]*d.y + l[3]*e.y,
           l[0]*a.z + l[1]*b.z + l[2]*d.z + l[3]*e.z};
  }
  return r;
}

// projectOriginPlane (:506): 1 = degenerate
MJH_DEV int ccd_project_plane(V3& res, V3 v1, V3 v2, V3 v3) {
  const V3 diff21 = v2 - v1, diff31 = v3 - v1, diff32 = v3 - v2;
  V3 n = cross(diff32, diff21);
  real nv = dot(n, v2), nn = dot(n, n);
  if (nn == 0) return 1;
  if (nv != 0 && nn > MJH_MINVAL) { res = ccd_scl(n, nv/nn); return 0; }
  n = cross(diff21, diff31);
  nv = dot(n, v1); nn = dot(n, n);
  if (nn == 0) return 1;
  if (nv != 0 && nn > MJH_MINVAL) { res = ccd_scl(n, nv/nn); return 0; }
  n = cross(diff31, diff32);
  nv = dot(n, v3); nn = dot(n, n);
  res = ccd_scl(n, nv/nn);
  return 0;
}

MJH_DEVN_HOT void ccd_S1D(real* lambda, crptr s1, crptr s2) {
  // projectOriginLine (:548)
  const V3 a = ld3(s1), b = ld3(s2);
  const V3 diff = b - a;
  const real scl = -(dot(b, diff)/dot(diff, diff));
  const V3 po{b.x + scl*diff.x, b.y + scl*diff.y, b.z + scl*diff.z};
  real mu = a.x - b.x, mu_max = mu;
  int index = 0;
  mu = a.y - b.y;
  if (ccd_abs(mu) >= ccd_abs(mu_max)) { mu_max = mu; index = 1; }
  mu = a.z - b.z;
  if (ccd_abs(mu) >= ccd_abs(mu_max)) { mu_max = mu; index = 2; }
  const real C1 = comp(po, index) - comp(b, index);
  const real C2 = comp(a, index) - comp(po, index);
  const int same = ccd_same_sign(mu_max, C1) && ccd_same_sign(mu_max, C2);
  lambda[0] = same ? C1/mu_max : 0;
  lambda[1] = same ? C2/mu_max : 1;
}

MJH_DEVN_HOT void ccd_S2D(real* lambda, crptr p1, crptr p2, crptr p3) {
  const V3 s1 = ld3(p1), s2 = ld3(p2), s3 = ld3(p3);
  V3 po;
  if (ccd_project_plane(po, s1, s2, s3)) {
    ccd_S1D(lambda, p1, p2);
    lambda[2] = 0;
    return;
  }
  const real M_14 = s2.y*s3.z - s2.z*s3.y - s1.y*s3.z + s1.z*s3.y + s1.y*s2.z - s1.z*s2.y;
  const real M_24 = s2.x*s3.z - s2.z*s3.x - s1.x*s3.z + s1.z*s3.x + s1.x*s2.z - s1.z*s2.x;
  const real M_34 = s2.x*s3.y - s2.y*s3.x - s1.x*s3.y + s1.y*s3.x + s1.x*s2.y - s1.y*s2.x;
  real M_max;
  real a0, a1, b0, b1, c0, c1, q0, q1;       // 2D projections of s1, s2, s3, p_o
  const real mu1 = ccd_abs(M_14), mu2 = ccd_abs(M_24), mu3 = ccd_abs(M_34);
  if (mu1 >= mu2 && mu1 >= mu3) {
    M_max = M_14; a0 = s1.y; a1 = s1.z; b0 = s2.y; b1 = s2.z; c0 = s3.y; c1 = s3.z; q0 = po.y; q1 = po.z;
  } else if (mu2 >= mu3) {
    M_max = M_24; a0 = s1.x; a1 = s1.z; b0 = s2.x; b1 = s2.z; c0 = s3.x; c1 = s3.z; q0 = po.x; q1 = po.z;
  } else {
    M_max = M_34; a0 = s1.x; a1 = s1.y; b0 = s2.x; b1 = s2.y; c0 = s3.x; c1 = s3.y; q0 = po.x; q1 = po.y;
  }
  const real C31 = q0*b1 + q1*c0 + b0*c1 - q0*c1 - q1*b0 - c0*b1;
  const real C32 = q0*c1 + q1*a0 + c0*a1 - q0*a1 - q1*c0 - a0*c1;
  const real C33 = q0*a1 + q1*b0 + a0*b1 - q0*b1 - q1*a0 - b0*a1;
  const int comp1 = ccd_same_sign(M_max, C31), comp2 = ccd_same_sign(M_max, C32), comp3 = ccd_same_sign(M_max, C33);
  if (comp1 && comp2 && comp3) {
    lambda[0] = C31/M_max; lambda[1] = C32/M_max; lambda[2] = C33/M_max;
    return;
  }
  real dmin = MJH_CCD_MAX;
  if (!comp1) {
    real l1[2];
    ccd_S1D(l1, p2, p3);
    const V3 x = ccd_lincomb2(l1, s2, s3);
    const real d = dot(x, x);
    lambda[0] = 0; lambda[1] = l1[0]; lambda[2] = l1[1];
    dmin = d;
  }
  if (!comp2) {
    real l1[2];
    ccd_S1D(l1, p1, p3);
    const V3 x = ccd_lincomb2(l1, s1, s3);
    const real d = dot(x, x);
    if (d < dmin) { lambda[0] = l1[0]; lambda[1] = 0; lambda[2] = l1[1]; dmin = d; }
  }
  if (!comp3) {
    real l1[2];
    ccd_S1D(l1, p1, p2);
    const V3 x = ccd_lincomb2(l1, s1, s2);
    const real d = dot(x, x);
    if (d < dmin) { lambda[0] = l1[0]; lambda[1] = l1[1]; lambda[2] = 0; }
  }
}

MJH_DEVN_HOT void ccd_S3D(real* lambda, crptr p1, crptr p2, crptr p3, crptr p4) {
  const V3 s1 = ld3(p1), s2 = ld3(p2), s3 = ld3(p3), s4 = ld3(p4);
  const real C41 = -ccd_det3(s2, s3, s4);
  const real C42 = ccd_det3(s1, s3, s4);
  const real C43 = -ccd_det3(s1, s2, s4);
  const real C44 = ccd_det3(s1, s2, s3);
  const real m_det = C41 + C42 + C43 + C44;
  const int comp1 = ccd_same_sign(m_det, C41), comp2 = ccd_same_sign(m_det, C42),
            comp3 = ccd_same_sign(m_det, C43), comp4 = ccd_same_sign(m_det, C44);
  if (comp1 && comp2 && comp3 && comp4) {
    lambda[0] = C41/m_det; lambda[1] = C42/m_det; lambda[2] = C43/m_det; lambda[3] = C44/m_det;
    return;
  }
  real dmin = MJH_CCD_MAX;
  if (!comp1) {
    real l2[3];
    ccd_S2D(l2, p2, p3, p4);
    const V3 x = ccd_lincomb3(l2, s2, s3, s4);
    const real d = dot(x, x);
    lambda[0] = 0; lambda[1] = l2[0]; lambda[2] = l2[1]; lambda[3] = l2[2];
    dmin = d;
  }
  if (!comp2) {
    real l2[3];
    ccd_S2D(l2, p1, p3, p4);
    const V3 x = ccd_lincomb3(l2, s1, s3, s4);
    const real d = dot(x, x);
    if (d < dmin) { lambda[0] = l2[0]; lambda[1] = 0; lambda[2] = l2[1]; lambda[3] = l2[2]; dmin = d; }
  }
  if (!comp3) {
    real l2[3];
    ccd_S2D(l2, p1, p2, p4);
    const V3 x = ccd_lincomb3(l2, s1, s2, s4);
    const real d = dot(x, x);
    if (d < dmin) { lambda[0] = l2[0]; lambda[1] = l2[1]; lambda[2] = 0; lambda[3] = l2[2]; dmin = d; }
  }
  if (!comp4) {
    real l2[3];
    ccd_S2D(l2, p1, p2, p3);
    const V3 x = ccd_lincomb3(l2, s1, s2, s3);
    const real d = dot(x, x);
    if (d < dmin) { lambda[0] = l2[0]; lambda[1] = l2[1]; lambda[2] = l2[2]; lambda[3] = 0; }
  }
}

// ---- GJK (:198) ---------------------------------------------------------------------------------------------
MJH_DEV int ccd_discrete(const Ccd& c) {
  if (c.o1.r[CO_MARGIN] != 0 || c.o2.r[CO_MARGIN] != 0) return 0;
  const int g1 = c.o1.i[CI_TYPE], g2 = c.o2.i[CI_TYPE];
  return (g1 == MJH_GEOM_MESH || g1 == MJH_GEOM_BOX) && (g2 == MJH_GEOM_MESH || g2 == MJH_GEOM_BOX);
}

// signedDistance (:409)
MJH_DEV real ccd_signed_distance(V3& normal, crptr v1, crptr v2, crptr v3) {
  const V3 a = ld3(v1);
  const V3 diff1 = ld3(v3) - a, diff2 = ld3(v2) - a;
  normal = cross(diff1, diff2);
  const real norm2 = dot(normal, normal);
  if (norm2 > MJH_CCD_MINVAL2 && norm2 < MJH_CCD_MAXVAL2) {
    normal = ccd_scl(normal, 1/sqrt(norm2));
    return dot(normal, a);
  }
  return MJH_CCD_MAX;
}

// gjkIntersect (:420): 1 in contact, 0 not, -1 inconclusive
MJH_DEVN_HOT int ccd_gjk_intersect(MREF M, Ccd& c) {
  for (int k = 0; k < 4; k++) ccd_vcopy(ccd_vtx(c.tmpr, c.tmpi, k), ccd_vtx(c.simr, c.simi, k));
  int s[4] = {0, 1, 2, 3};
  int k = c.gjk_iterations;
  const int kmax = c.N;
  for (; k < kmax; k++) {
    real dist[4];
    V3 normals[4];
    dist[0] = ccd_signed_distance(normals[0], c.tmpr + CV_NREAL*s[2], c.tmpr + CV_NREAL*s[1], c.tmpr + CV_NREAL*s[3]);
    dist[1] = ccd_signed_distance(normals[1], c.tmpr + CV_NREAL*s[0], c.tmpr + CV_NREAL*s[2], c.tmpr + CV_NREAL*s[3]);
    dist[2] = ccd_signed_distance(normals[2], c.tmpr + CV_NREAL*s[1], c.tmpr + CV_NREAL*s[0], c.tmpr + CV_NREAL*s[3]);
    dist[3] = ccd_signed_distance(normals[3], c.tmpr + CV_NREAL*s[0], c.tmpr + CV_NREAL*s[1], c.tmpr + CV_NREAL*s[2]);
    if (!dist[3] || !dist[2] || !dist[1] || !dist[0]) { c.gjk_iterations = k; return -1; }
    int i = (dist[0] < dist[1]) ? 0 : 1;
    int j = (dist[2] < dist[3]) ? 2 : 3;
    const int index = (dist[i] < dist[j]) ? i : j;
    if (dist[index] > 0) {
      c.nsimplex = 4;
      // (the four sources are distinct slots of the copy: no aliasing with the destination)
      for (int q = 0; q < 4; q++) ccd_vcopy(ccd_vtx(c.simr, c.simi, q), ccd_vtx(c.tmpr, c.tmpi, s[q]));
      c.gjk_iterations = k;
      return 1;
    }
    const V3 nrm = normals[index];
    const CcdVtx nv = ccd_vtx(c.tmpr, c.tmpi, s[index]);
    ccd_support(M, c, nv, nrm, V3{-nrm.x, -nrm.y, -nrm.z});
    if (dot(nrm, ld3(nv.r)) < 0) { c.nsimplex = 0; c.gjk_iterations = k; return 0; }
    i = (index + 1) & 3;
    j = (index + 2) & 3;
    const int swap = s[i];
    s[i] = s[j];
    s[j] = swap;
  }
  c.gjk_iterations = k;
  return -1;
}

MJH_DEVN_HOT void ccd_gjk(MREF M, Ccd& c) {
  const int get_dist = c.dist_cutoff > 0;
  int backup_gjk = !get_dist;
  int n = 0, k = 0;
  const int kmax = c.N;
  real lambda[4] = {0, 0, 0, 0};
  const real tol2 = c.tolerance*c.tolerance;
  c.separated = 0;
  const int discrete = ccd_discrete(c);
  const real epsilon = discrete ? 0 : 0.5*tol2;
  const real min_norm = discrete ? MJH_MINVAL : c.tolerance;
  V3 xk = ld3(c.x1) - ld3(c.x2);
  real x_norm = ccd_norm(xk), x_norm_prev = 0;
  for (; k < kmax; k++) {
    if (x_norm < min_norm || ccd_abs(x_norm_prev - x_norm) < MJH_MINVAL) break;
    const CcdVtx sk = ccd_vtx(c.simr, c.simi, n);
    ccd_gjk_support(M, c, sk, xk, x_norm);
    const V3 s_k = ld3(sk.r);
    const V3 diff = xk - s_k;
    if (dot(xk, diff) < epsilon) break;
    const real lower = dot(xk, s_k);
    if (!get_dist) {
      if (lower > 0) {
        c.separated = 1; c.gjk_iterations = k; c.nsimplex = 0; c.nx = 0; c.dist[0] = MJH_CCD_MAX;
        return;
      }
    } else if (c.dist_cutoff < MJH_CCD_MAX) {
      if (lower > 0 && lower >= c.dist_cutoff*x_norm) {
        c.separated = 1; c.gjk_iterations = k; c.nsimplex = 0; c.nx = 0; c.dist[0] = MJH_CCD_MAX;
        return;
      }
    }
    if (n == 3 && backup_gjk) {
      c.gjk_iterations = k;
      const int ret = ccd_gjk_intersect(M, c);
      if (ret != -1) {
        c.nx = 0;
        c.separated = ret == 0;
        c.dist[0] = ret > 0 ? 0 : MJH_CCD_MAX;
        return;
      }
      k = c.gjk_iterations;
      backup_gjk = 0;
    }
    // subdistance (:586)
    lambda[0] = lambda[1] = lambda[2] = lambda[3] = 0;
    if (n + 1 == 4) ccd_S3D(lambda, c.simr, c.simr + CV_NREAL, c.simr + 2*CV_NREAL, c.simr + 3*CV_NREAL);
    else if (n + 1 == 3) ccd_S2D(lambda, c.simr, c.simr + CV_NREAL, c.simr + 2*CV_NREAL);
    else if (n + 1 == 2) ccd_S1D(lambda, c.simr, c.simr + CV_NREAL);
    else lambda[0] = 1;
    n = 0;
    for (int i = 0; i < 4; i++) {
      if (!lambda[i]) continue;
      if (n != i) ccd_vcopy(ccd_vtx(c.simr, c.simi, n), ccd_vtx(c.simr, c.simi, i));
      lambda[n++] = lambda[i];
    }
    if (n < 1) {
      c.gjk_iterations = k; c.nsimplex = 0; c.nx = 0; c.dist[0] = MJH_CCD_MAX; c.separated = 1;
      return;
    }
    xk = ccd_lincomb(lambda, n, c.simr, CV_NREAL);
    x_norm_prev = x_norm;
    x_norm = ccd_norm(xk);
    if (n == 4) break;
  }
  if (n > 0) {
    st3(c.x1, ccd_lincomb(lambda, n, c.simr + 3, CV_NREAL));
    st3(c.x2, ccd_lincomb(lambda, n, c.simr + 6, CV_NREAL));
  }
  // final separation check
  const CcdVtx tmp = ccd_vtx(c.tmpr, c.tmpi, 4);
  ccd_gjk_support(M, c, tmp, xk, x_norm);
  if (dot(xk, ld3(tmp.r)) > 0) c.separated = 1;
  c.nx = 1;
  c.gjk_iterations = k;
  c.nsimplex = n;
  c.dist[0] = (n == 4 && !c.separated) ? 0 : x_norm;
}

// ---- EPA (:882-1500) --------------------------------------------------------------------------------------------
MJH_DEV int ccd_insert_vertex(Ccd& c, CcdVtx v) {
  const int n = c.nverts++;
  ccd_vcopy(ccd_vtx(c.vr, c.vi, n), v);
  return n;
}
MJH_DEV V3 ccd_pv(const Ccd& c, int v) { return ld3(c.vr + CV_NREAL*v); }

// attachFace (:1254): squared distance of the new face to the origin
MJH_DEVN_HOT real ccd_attach_face(Ccd& c, int v1, int v2, int v3, int adj1, int adj2, int adj3) {
  const int f = c.nfaces++;
  const iptr fi = c.fi + CF_NINT*f;
  const rptr fr = c.fr + CF_NREAL*f;
  fi[0] = v1 + (v2 << 10) + (v3 << 20);
  fi[1] = adj1; fi[2] = adj2; fi[3] = adj3;
  V3 fv;
  if (ccd_project_plane(fv, ccd_pv(c, v3), ccd_pv(c, v2), ccd_pv(c, v1))) return 0;
  const V3 outward = ccd_pv(c, v1) - c.center;
  if (dot(fv, outward) < 0) fv = ccd_scl(fv, -1);
  st3(fr, fv);
  fr[3] = dot(fv, fv);
  fi[4] = -1;
  return fr[3];
}
MJH_DEV void ccd_replace_simplex3(Ccd& c, int v1, int v2, int v3) {
  c.nsimplex = 3;
  // (sources live in the polytope, destinations in the simplex: copy through the scratch vertices in
  // case the compiler reorders -- they never alias, the copy is direct)
  ccd_vcopy(ccd_vtx(c.simr, c.simi, 0), ccd_vtx(c.vr, c.vi, v1));
  ccd_vcopy(ccd_vtx(c.simr, c.simi, 1), ccd_vtx(c.vr, c.vi, v2));
  ccd_vcopy(ccd_vtx(c.simr, c.simi, 2), ccd_vtx(c.vr, c.vi, v3));
  c.nfaces = 0; c.nverts = 0; c.nmap = 0;
}
MJH_DEV int ccd_same_side(V3 p0, V3 p1, V3 p2, V3 p3) {
  const V3 n = cross(p1 - p0, p2 - p0);
  const real dot1 = dot(n, p3 - p0);
  const real dot2 = dot(n, ccd_scl(p0, -1));
  if (dot1 > 0 && dot2 > 0) return 1;
  if (dot1 < 0 && dot2 < 0) return 1;
  return 0;
}
MJH_DEV int ccd_test_tetra(V3 p0, V3 p1, V3 p2, V3 p3) {
  return ccd_same_side(p0, p1, p2, p3) && ccd_same_side(p1, p2, p3, p0) && ccd_same_side(p2, p3, p0, p1) && ccd_same_side(p3, p0, p1, p2);
}
// triAffineCoord (:1033)
MJH_DEV void ccd_tri_affine(real* lambda, V3 v1, V3 v2, V3 v3, V3 p) {
  const real M_14 = v2.y*v3.z - v2.z*v3.y - v1.y*v3.z + v1.z*v3.y + v1.y*v2.z - v1.z*v2.y;
  const real M_24 = v2.x*v3.z - v2.z*v3.x - v1.x*v3.z + v1.z*v3.x + v1.x*v2.z - v1.z*v2.x;
  const real M_34 = v2.x*v3.y - v2.y*v3.x - v1.x*v3.y + v1.y*v3.x + v1.x*v2.y - v1.y*v2.x;
  real M_max;
  int x, y;
  const real mu1 = ccd_abs(M_14), mu2 = ccd_abs(M_24), mu3 = ccd_abs(M_34);
  if (mu1 >= mu2 && mu1 >= mu3) { M_max = M_14; x = 1; y = 2; }
  else if (mu2 >= mu3) { M_max = M_24; x = 0; y = 2; }
  else { M_max = M_34; x = 0; y = 1; }
  const real px = comp(p, x), py = comp(p, y);
  const real ax = comp(v1, x), ay = comp(v1, y), bx = comp(v2, x), by = comp(v2, y), cx = comp(v3, x), cy = comp(v3, y);
  const real C31 = px*by + py*cx + bx*cy - px*cy - py*bx - cx*by;
  const real C32 = px*cy + py*ax + cx*ay - px*ay - py*cx - ax*cy;
  const real C33 = px*ay + py*bx + ax*by - px*by - py*ax - bx*ay;
  lambda[0] = C31/M_max; lambda[1] = C32/M_max; lambda[2] = C33/M_max;
}
MJH_DEV int ccd_tri_point_intersect(V3 v1, V3 v2, V3 v3, V3 p) {
  real l[3];
  ccd_tri_affine(l, v1, v2, v3, p);
  if (l[0] < 0 || l[1] < 0 || l[2] < 0) return 0;
  const V3 pr{v1.x*l[0] + v2.x*l[1] + v3.x*l[2], v1.y*l[0] + v2.y*l[1] + v3.y*l[2], v1.z*l[0] + v2.z*l[1] + v3.z*l[2]};
  return ccd_norm(pr - p) < MJH_MINVAL;
}
MJH_DEV void ccd_fill_map(Ccd& c, int n) {
  for (int i = 0; i < n; i++) { c.map[i] = i; c.fi[CF_NINT*i + 4] = i; }
  c.nmap = n;
}

// polytope3 (:1083): hexahedron from a triangle.  Returns an mjEPAStatus code (0 = success)
MJH_DEVN_HOT int ccd_polytope3(MREF M, Ccd& c) {
  const V3 v1 = ld3(c.simr), v2 = ld3(c.simr + CV_NREAL), v3 = ld3(c.simr + 2*CV_NREAL);
  c.center = ccd_scl((v1 + v2) + v3, 1.0/3.0);
  const V3 n = cross(v2 - v1, v3 - v1);
  const real n_norm = ccd_norm(n);
  if (n_norm < MJH_MINVAL) return 5;           // mjEPA_P3_BAD_NORMAL
  const V3 n_neg = ccd_scl(n, -1);
  const int v1i = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 0));
  const int v2i = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 1));
  const int v3i = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 2));
  const int v5i = ccd_epa_support(M, c, n_neg, n_norm);
  const int v4i = ccd_epa_support(M, c, n, n_norm);
  const V3 v4 = ccd_pv(c, v4i), v5 = ccd_pv(c, v5i);
  if (ccd_tri_point_intersect(v1, v2, v3, v4)) return 6;     // P3_INVALID_V4
  if (ccd_tri_point_intersect(v1, v2, v3, v5)) return 7;     // P3_INVALID_V5
  if (c.dist[0] > 10*MJH_MINVAL && !ccd_test_tetra(v1, v2, v3, v4) && !ccd_test_tetra(v1, v2, v3, v5)) return 8;   // P3_MISSING_ORIGIN
  if (ccd_attach_face(c, v4i, v1i, v2i, 1, 3, 2) < MJH_CCD_MINVAL2) return 9;   // P3_ORIGIN_ON_FACE
  if (ccd_attach_face(c, v4i, v3i, v1i, 2, 4, 0) < MJH_CCD_MINVAL2) return 9;
  if (ccd_attach_face(c, v4i, v2i, v3i, 0, 5, 1) < MJH_CCD_MINVAL2) return 9;
  if (ccd_attach_face(c, v5i, v2i, v1i, 5, 0, 4) < MJH_CCD_MINVAL2) return 9;
  if (ccd_attach_face(c, v5i, v1i, v3i, 3, 1, 5) < MJH_CCD_MINVAL2) return 9;
  if (ccd_attach_face(c, v5i, v3i, v2i, 4, 2, 3) < MJH_CCD_MINVAL2) return 9;
  ccd_fill_map(c, 6);
  return 0;
}

// polytope2 (:948): hexahedron around a segment
MJH_DEVN_HOT int ccd_polytope2(MREF M, Ccd& c) {
  const V3 v1 = ld3(c.simr), v2 = ld3(c.simr + CV_NREAL);
  c.center = ccd_scl(v1 + v2, 0.5);
  const V3 diff = v2 - v1;
  real value = MJH_CCD_MAX;
  int index = 0;
  for (int i = 0; i < 3; i++) if (ccd_abs(comp(diff, i)) < value) { value = ccd_abs(comp(diff, i)); index = i; }
  const V3 e = with_comp(V3{0, 0, 0}, index, 1);
  const V3 d1 = cross(e, diff);
  // rotmat (:915): 120 degrees about diff
  real R[9];
  {
    const real nrm = ccd_norm(diff);
    const real u1 = diff.x/nrm, u2 = diff.y/nrm, u3 = diff.z/nrm;
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
  const int v1i = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 0));
  const int v2i = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 1));
  const int v3i = ccd_epa_support(M, c, d1, ccd_norm(d1));
  const int v4i = ccd_epa_support(M, c, d2, ccd_norm(d2));
  const int v5i = ccd_epa_support(M, c, d3, ccd_norm(d3));
  const V3 v3 = ccd_pv(c, v3i), v4 = ccd_pv(c, v4i), v5 = ccd_pv(c, v5i);
  if (ccd_attach_face(c, v1i, v3i, v4i, 1, 3, 2) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v1i, v3i, v4i); return ccd_polytope3(M, c); }
  if (ccd_attach_face(c, v1i, v5i, v3i, 2, 4, 0) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v1i, v5i, v3i); return ccd_polytope3(M, c); }
  if (ccd_attach_face(c, v1i, v4i, v5i, 0, 5, 1) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v1i, v4i, v5i); return ccd_polytope3(M, c); }
  if (ccd_attach_face(c, v2i, v4i, v3i, 5, 0, 4) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v2i, v4i, v3i); return ccd_polytope3(M, c); }
  if (ccd_attach_face(c, v2i, v3i, v5i, 3, 1, 5) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v2i, v3i, v5i); return ccd_polytope3(M, c); }
  if (ccd_attach_face(c, v2i, v5i, v4i, 4, 2, 3) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v2i, v5i, v4i); return ccd_polytope3(M, c); }
  // rayTriangle (:932): the hexahedron must be convex
  {
    const V3 diff12 = v2 - v1, diff13 = v3 - v1, diff14 = v4 - v1, diff15 = v5 - v1;
    const real vol1 = ccd_det3(diff13, diff14, diff12);
    const real vol2 = ccd_det3(diff14, diff15, diff12);
    const real vol3 = ccd_det3(diff15, diff13, diff12);
    const int hit = (vol1 >= 0 && vol2 >= 0 && vol3 >= 0) || (vol1 <= 0 && vol2 <= 0 && vol3 <= 0);
    if (!hit) return 2;                        // P2_NONCONVEX
  }
  ccd_fill_map(c, 6);
  return 0;
}

// polytope4 (:1167): the GJK tetrahedron itself
MJH_DEVN_HOT int ccd_polytope4(MREF M, Ccd& c) {
  const int v1 = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 0));
  const int v2 = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 1));
  const int v3 = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 2));
  const int v4 = ccd_insert_vertex(c, ccd_vtx(c.simr, c.simi, 3));
  c.center = ccd_scl(((ccd_pv(c, v1) + ccd_pv(c, v2)) + ccd_pv(c, v3)) + ccd_pv(c, v4), 0.25);
  if (ccd_attach_face(c, v1, v2, v3, 1, 3, 2) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v1, v2, v3); return ccd_polytope3(M, c); }
  if (ccd_attach_face(c, v1, v4, v2, 2, 3, 0) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v1, v4, v2); return ccd_polytope3(M, c); }
  if (ccd_attach_face(c, v1, v3, v4, 0, 3, 1) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v1, v3, v4); return ccd_polytope3(M, c); }
  if (ccd_attach_face(c, v4, v3, v2, 2, 0, 1) < MJH_CCD_MINVAL2) { ccd_replace_simplex3(c, v4, v3, v2); return ccd_polytope3(M, c); }
  if (!ccd_test_tetra(ccd_pv(c, v1), ccd_pv(c, v2), ccd_pv(c, v3), ccd_pv(c, v4))) return 10;   // P4_MISSING_ORIGIN
  ccd_fill_map(c, 4);
  return 0;
}

MJH_DEV void ccd_delete_face(Ccd& c, int f) {
  const iptr fi = c.fi + CF_NINT*f;
  if (fi[4] >= 0) {
    c.map[fi[4]] = c.map[--c.nmap];
    c.fi[CF_NINT*c.map[fi[4]] + 4] = fi[4];
  }
  fi[4] = -2;
}
MJH_DEV int ccd_face_vert(const Ccd& c, int f, int k) { return (c.fi[CF_NINT*f] >> (10*k)) & 0x3FF; }
MJH_DEV int ccd_get_edge(const Ccd& c, int f, int vertex) {
  if (ccd_face_vert(c, f, 0) == vertex) return 0;
  if (ccd_face_vert(c, f, 1) == vertex) return 1;
  return 2;
}
MJH_DEV void ccd_add_edge(Ccd& c, int index, int edge) {
  if (c.nedges < c.maxhorizon) { c.hedge[c.nedges] = edge; c.hidx[c.nedges] = index; }
  c.nedges++;
}

// horizonRec (:1295) as an explicit depth-first walk: returns 1 if `face` is visible from w.  A frame is
// (face, entry edge e, progress k): k = 0 not yet tested, 1..2 the edge being expanded, 3 done.
MJH_DEVN_HOT int ccd_horizon_rec(Ccd& c, int face0, int e0) {
  int sp = 0;
  c.stack[0] = face0; c.stack[1] = e0 | (0 << 8);
  int result = 0;       // return value of the frame that just finished
  int returning = 0;
  while (sp >= 0) {
    const int face = c.stack[2*sp];
    const int e = c.stack[2*sp + 1] & 0xff;
    int k = c.stack[2*sp + 1] >> 8;
    const ciptr fi = c.fi + CF_NINT*face;
    if (returning) {
      // child of edge slot k-1 came back
      returning = 0;
      const int i = (e + (k - 1)) % 3;
      if (!result) {
        const int adj = fi[1 + i];
        ccd_add_edge(c, adj, ccd_get_edge(c, adj, ccd_face_vert(c, face, (i + 1) % 3)));
      }
    } else if (k == 0) {
      const crptr fr = c.fr + CF_NREAL*face;
      if (!(dot(ld3(fr), c.horizon_w) - fr[3] > MJH_MINVAL)) { result = 0; returning = 1; sp--; continue; }
      ccd_delete_face(c, face);
      k = 1;
    }
    // expand the remaining edges
    int pushed = 0;
    while (k < 3) {
      const int i = (e + k) % 3;
      const int adj = fi[1 + i];
      k++;
      if (c.fi[CF_NINT*adj + 4] > -2) {
        const int adj_edge = ccd_get_edge(c, adj, ccd_face_vert(c, face, (i + 1) % 3));
        c.stack[2*sp + 1] = e | (k << 8);
        sp++;
        c.stack[2*sp] = adj; c.stack[2*sp + 1] = adj_edge;
        pushed = 1;
        break;
      }
    }
    if (pushed) continue;
    result = 1; returning = 1; sp--;
  }
  return result;
}

// horizon (:1322)
MJH_DEV void ccd_horizon(Ccd& c, int face) {
  ccd_delete_face(c, face);
  const ciptr fi = c.fi + CF_NINT*face;
  int adj = fi[1];
  int adj_edge = ccd_get_edge(c, adj, ccd_face_vert(c, face, 1));
  if (!ccd_horizon_rec(c, adj, adj_edge)) ccd_add_edge(c, adj, adj_edge);
  adj = fi[2];
  adj_edge = ccd_get_edge(c, adj, ccd_face_vert(c, face, 2));
  if (c.fi[CF_NINT*adj + 4] > -2 && !ccd_horizon_rec(c, adj, adj_edge)) ccd_add_edge(c, adj, adj_edge);
  adj = fi[3];
  adj_edge = ccd_get_edge(c, adj, ccd_face_vert(c, face, 0));
  if (c.fi[CF_NINT*adj + 4] > -2 && !ccd_horizon_rec(c, adj, adj_edge)) ccd_add_edge(c, adj, adj_edge);
}

// epa (:1358): index of the face that approximates the penetration depth, -1 if none
MJH_DEVN_HOT int ccd_epa(MREF M, Ccd& c) {
  real upper = MJH_CCD_MAX, upper2 = MJH_CCD_MAX, lower2;
  int face = -1, pface = -1;
  const int discrete = ccd_discrete(c);
  const real tolerance = discrete ? MJH_MINVAL : c.tolerance;
  const int kmax = c.N < 1000 ? c.N : 1000;
  int k;
  for (k = 0; k < kmax; k++) {
    pface = face;
    lower2 = MJH_CCD_MAX;
    for (int i = 0; i < c.nmap; i++) {
      const real d2 = c.fr[CF_NREAL*c.map[i] + 3];
      if (d2 < lower2) { face = c.map[i]; lower2 = d2; }
    }
    if (lower2 > upper2 || face < 0) { face = pface; break; }
    if (lower2 <= 0) break;                     // (reference: warning "origin lies on affine hull of face")
    const real lower = sqrt(lower2);
    const V3 fv = ld3(c.fr + CF_NREAL*face);
    const int wi = ccd_epa_support(M, c, fv, lower);
    const CcdVtx w = ccd_vtx(c.vr, c.vi, wi);
    const real upper_k = dot(fv, ld3(w.r))/lower;
    if (upper_k < upper) { upper = upper_k; upper2 = upper*upper; }
    if (upper - lower < tolerance) {
      if (k == 0 && upper < lower - 1e-10) face = -1;
      break;
    }
    if (discrete) {
      int i = 0;
      const int nverts = c.nverts - 1;
      for (; i < nverts; i++) if (w.i[0] == c.vi[CV_NINT*i] && w.i[1] == c.vi[CV_NINT*i + 1]) break;
      if (i != nverts) break;
    }
    c.horizon_w = ld3(w.r);
    ccd_horizon(c, face);
    if (c.nedges < 3) { face = -1; break; }
    const int nfaces = c.nfaces, nedges = c.nedges;
    if (nedges > c.maxfaces - c.nfaces || nedges > c.maxhorizon) break;       // (reference: out-of-memory warning)
    int hidx = c.hidx[0], hedge = c.hedge[0];
    int v1 = ccd_face_vert(c, hidx, hedge), v2 = ccd_face_vert(c, hidx, (hedge + 1) % 3);
    c.fi[CF_NINT*hidx + 1 + hedge] = nfaces;
    real dist2 = ccd_attach_face(c, wi, v2, v1, nfaces + nedges - 1, hidx, nfaces + 1);
    if (dist2 == 0) { face = -1; break; }
    if (dist2 >= lower2 && dist2 <= upper2) {
      const int i = c.nmap++;
      c.map[i] = c.nfaces - 1;
      c.fi[CF_NINT*(c.nfaces - 1) + 4] = i;
    }
    for (int i = 1; i < nedges; i++) {
      const int cur = nfaces + i;
      const int next = nfaces + (i + 1) % nedges;
      hidx = c.hidx[i]; hedge = c.hedge[i];
      v1 = ccd_face_vert(c, hidx, hedge);
      v2 = ccd_face_vert(c, hidx, (hedge + 1) % 3);
      c.fi[CF_NINT*hidx + 1 + hedge] = cur;
      dist2 = ccd_attach_face(c, wi, v2, v1, cur - 1, hidx, next);
      if (dist2 == 0) { face = -1; break; }
      if (dist2 >= lower2 && dist2 <= upper2) {
        const int idx = c.nmap++;
        c.map[idx] = c.nfaces - 1;
        c.fi[CF_NINT*(c.nfaces - 1) + 4] = idx;
      }
    }
    c.nedges = 0;
    if (!c.nmap || face < 0) break;
  }
  if (face >= 0) {
    // epaWitness (:1339)
    const int a = ccd_face_vert(c, face, 0), b = ccd_face_vert(c, face, 1), d = ccd_face_vert(c, face, 2);
    real l[3];
    ccd_tri_affine(l, ccd_pv(c, a), ccd_pv(c, b), ccd_pv(c, d), ld3(c.fr + CF_NREAL*face));
    st3(c.x1, ccd_lincomb3(l, ld3(c.vr + CV_NREAL*a + 3), ld3(c.vr + CV_NREAL*b + 3), ld3(c.vr + CV_NREAL*d + 3)));
    st3(c.x2, ccd_lincomb3(l, ld3(c.vr + CV_NREAL*a + 6), ld3(c.vr + CV_NREAL*b + 6), ld3(c.vr + CV_NREAL*d + 6)));
    c.dist[0] = -sqrt(c.fr[CF_NREAL*face + 3]);
    c.nx = 1;
  } else {
    c.nx = 0;
    c.dist[0] = 0;
  }
  return face;
}

// ---- multi-contact recovery (:1503-2310) ----------------------------------------------------------------------------
MJH_DEV real ccd_area4(V3 a, V3 b, V3 c, V3 d) {
  const V3 ad = d - a, db = b - d, bc = c - b, ca = a - c;
  const V3 g = cross(ad, db) + cross(bc, ca);
  return 0.5*ccd_norm(g);
}
// polygonQuad (:1523): indices of a maximum-area quadrilateral of a convex polygon
MJH_DEV void ccd_polygon_quad(int* res, crptr polygon, int nvert) {
  auto P = [&](int i) { return ld3(polygon + 3*i); };
  auto nxt = [&](int i) { return i == nvert - 1 ? 0 : i + 1; };
  int a = 0, b = 1, cc = 2, d = 3;
  res[0] = a; res[1] = b; res[2] = cc; res[3] = d;
  real m = ccd_area4(P(a), P(b), P(cc), P(d)), m_next;
  for (; a < nvert; a++) {
    while (1) {
      m_next = ccd_area4(P(a), P(b), P(cc), P(nxt(d)));
      if (m_next <= m) break;
      m = m_next;
      d = nxt(d);
      res[0] = a; res[1] = b; res[2] = cc; res[3] = d;
      while (1) {
        m_next = ccd_area4(P(a), P(b), P(nxt(cc)), P(d));
        if (m_next <= m) break;
        m = m_next;
        cc = nxt(cc);
        res[0] = a; res[1] = b; res[2] = cc; res[3] = d;
      }
      while (1) {
        m_next = ccd_area4(P(a), P(nxt(b)), P(cc), P(d));
        if (m_next <= m) break;
        m = m_next;
        b = nxt(b);
        res[0] = a; res[1] = b; res[2] = cc; res[3] = d;
      }
    }
    if (b == a) {
      b = nxt(b);
      if (cc == b) {
        cc = nxt(cc);
        if (d == cc) d = nxt(d);
      }
    }
  }
}
// witnessOnFace (:1605)
MJH_DEV real ccd_witness_on_face(rptr w1, rptr w2, V3 v, V3 p, V3 n, V3 dir) {
  const V3 d = v - p;
  const real dist = dot(d, n);
  const real s = -ccd_abs(dist);
  st3(w1, V3{v.x + s*dir.x, v.y + s*dir.y, v.z + s*dir.z});
  st3(w2, v);
  return dist;
}

// polygonClip (:1617): clip face2 against the side planes of face1 (Sutherland-Hodgman)
MJH_DEVN_HOT void ccd_polygon_clip(Ccd& c, crptr face1, int nface1, crptr face2, int nface2, V3 n, V3 dir,
                                   rptr buffer) {
  if (nface1 < 3) return;
  const int P = c.P;
  rptr polygon = buffer;
  rptr clipped = polygon + 6*P;
  const rptr pn = clipped + 6*P;
  const rptr pd = pn + 3*P;
  // planeNormal (:1581) of every edge of face1
  for (int i = 0; i < nface1; i++) {
    const V3 v1 = ld3(face1 + 3*i), v2 = ld3(face1 + 3*(i < nface1 - 1 ? i + 1 : 0));
    const V3 v3 = v1 + n;
    V3 r = cross(v2 - v1, v3 - v1);
    unitize(r);
    st3(pn + 3*i, r);
    pd[i] = dot(r, v1);
  }
  int npolygon = nface2, nclipped = 0;
  for (int i = 0; i < 3*nface2; i++) polygon[i] = face2[i];
  for (int e = 0; e < nface1; e++) {
    const V3 fa = ld3(face1 + 3*e), pe = ld3(pn + 3*e);
    for (int i = 0; i < npolygon; i++) {
      const V3 Pp = ld3(polygon + 3*i);
      const V3 Q = ld3(polygon + 3*((i < npolygon - 1) ? i + 1 : 0));
      const V3 PQ = Q - Pp;
      const int inside1 = dot(Pp - fa, pe) > -MJH_MINVAL;     // halfspace (:1596)
      const int inside2 = dot(Q - fa, pe) > -MJH_MINVAL;
      if (!inside1 && !inside2) continue;
      if (inside1 && inside2) { if (nclipped < 2*P) st3(clipped + 3*nclipped, Q); nclipped++; continue; }
      const real tmp = dot(pe, PQ);
      if (tmp != 0.0) {
        const real t = (pd[e] - dot(pe, Pp))/tmp;
        if (t >= 0.0 && t <= 1.0) {
          if (nclipped < 2*P) st3(clipped + 3*nclipped, V3{Pp.x + t*PQ.x, Pp.y + t*PQ.y, Pp.z + t*PQ.z});
          nclipped++;
        }
      }
      if (inside2) { if (nclipped < 2*P) st3(clipped + 3*nclipped, Q); nclipped++; }
    }
    const rptr t = polygon; polygon = clipped; clipped = t;
    npolygon = nclipped < 2*P ? nclipped : 2*P;
    nclipped = 0;
  }
  // drop vertices above face1
  const int m = npolygon;
  npolygon = 0;
  const V3 f0 = ld3(face1);
  for (int i = 0; i < m; i++) {
    const V3 v = ld3(polygon + 3*i);
    if (dot(v - f0, n) <= 0) {
      if (npolygon != i) st3(polygon + 3*npolygon, v);
      npolygon++;
    }
  }
  if (npolygon < 1) return;
  if (c.max_contacts < 5 && npolygon > 4) {
    c.nx = 4;
    int rect[4];
    ccd_polygon_quad(rect, polygon, npolygon);
    for (int i = 0; i < 4; i++)
      c.dist[i] = ccd_witness_on_face(c.x1 + 3*i, c.x2 + 3*i, ld3(polygon + 3*rect[i]), f0, n, dir);
    return;
  }
  if (nface2 == 2 && npolygon > 2) {
    int best1 = 0, best2 = 1;
    real d = 0;
    for (int i = 0; i < npolygon; i++)
      for (int j = i + 1; j < npolygon; j++) {
        const V3 df = ld3(polygon + 3*j) - ld3(polygon + 3*i);
        const real d2 = dot(df, df);
        if (d2 > d) { d = d2; best1 = i; best2 = j; }
      }
    c.dist[0] = ccd_witness_on_face(c.x1, c.x2, ld3(polygon + 3*best1), f0, n, dir);
    c.dist[1] = ccd_witness_on_face(c.x1 + 3, c.x2 + 3, ld3(polygon + 3*best2), f0, n, dir);
    c.nx = 2;
    return;
  }
  npolygon = npolygon < CCD_MAXWIT ? npolygon : CCD_MAXWIT;
  for (int i = 0; i < npolygon; i++)
    c.dist[i] = ccd_witness_on_face(c.x1 + 3*i, c.x2 + 3*i, ld3(polygon + 3*i), f0, n, dir);
  c.nx = npolygon;
}

// intersect (:1772): up to two common entries of two polymap slices
MJH_DEV int ccd_intersect_map(MREF M, int* res, int adr1, int n, int adr2, int m) {
  int count = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++)
      if (M.mesh_polymap[adr1 + i] == M.mesh_polymap[adr2 + j]) {
        res[count++] = M.mesh_polymap[adr1 + i];
        if (count == 2) return 2;
      }
  return count;
}
MJH_DEV int ccd_intersect_arr(MREF M, int* res, const int* arr1, int n, int adr2, int m) {
  int count = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++)
      if (arr1[i] == M.mesh_polymap[adr2 + j]) {
        res[count++] = arr1[i];
        if (count == 2) return 2;
      }
  return count;
}
MJH_DEV V3 ccd_polynormal(MREF M, CcdObj o, int poly) {
  const int base = 3*(M.mesh_polyadr[o.i[CI_MESH]] + poly);
  return ccd_globalrot(o.r + CO_MAT, M.mesh_polynormal[base], M.mesh_polynormal[base + 1], M.mesh_polynormal[base + 2]);
}
// meshNormals (:1787)
MJH_DEV int ccd_mesh_normals(MREF M, const Ccd& c, rptr res, iptr resind, int dim, CcdObj o, const int* vi) {
  const int vadr = M.mesh_vertadr[o.i[CI_MESH]];
  const int a1 = M.mesh_polymapadr[vadr + vi[0]], n1 = M.mesh_polymapnum[vadr + vi[0]];
  if (dim == 3) {
    const int a2 = M.mesh_polymapadr[vadr + vi[1]], n2 = M.mesh_polymapnum[vadr + vi[1]];
    const int a3 = M.mesh_polymapadr[vadr + vi[2]], n3 = M.mesh_polymapnum[vadr + vi[2]];
    int edgeset[2], faceset[2];
    int n = ccd_intersect_map(M, edgeset, a1, n1, a2, n2);
    if (n == 0) return 0;
    n = ccd_intersect_arr(M, faceset, edgeset, n, a3, n3);
    if (n == 0) return 0;
    st3(res, ccd_polynormal(M, o, faceset[0]));
    resind[0] = faceset[0];
    return 1;
  }
  if (dim == 2) {
    const int a2 = M.mesh_polymapadr[vadr + vi[1]], n2 = M.mesh_polymapnum[vadr + vi[1]];
    int edgeset[2];
    const int n = ccd_intersect_map(M, edgeset, a1, n1, a2, n2);
    if (n == 0) return 0;
    for (int i = 0; i < n; i++) { st3(res + 3*i, ccd_polynormal(M, o, edgeset[i])); resind[i] = edgeset[i]; }
    return n;
  }
  if (dim == 1) {
    const int n = n1 < c.D ? n1 : c.D;
    for (int i = 0; i < n; i++) {
      const int index = M.mesh_polymap[a1 + i];
      st3(res + 3*i, ccd_polynormal(M, o, index));
      resind[i] = index;
    }
    return n;
  }
  return 0;
}
// meshEdgeNormals (:1852)
MJH_DEV int ccd_mesh_edge_normals(MREF M, const Ccd& c, rptr res, rptr endverts, int dim, CcdObj o, const real* v, int v1i) {
  const V3 v1 = ld3(v), v2 = ld3(v + 3);
  if (dim == 2) {
    st3(endverts, v2);
    V3 r = v2 - v1;
    unitize(r);
    st3(res, r);
    return 1;
  }
  if (dim == 1) {
    const int mesh = o.i[CI_MESH];
    const int vadr = M.mesh_vertadr[mesh], padr = M.mesh_polyadr[mesh];
    const int a1 = M.mesh_polymapadr[vadr + v1i];
    const int n1r = M.mesh_polymapnum[vadr + v1i];
    const int n1 = n1r < c.D ? n1r : c.D;
    for (int i = 0; i < n1; i++) {
      const int idx = M.mesh_polymap[a1 + i];
      const int adr = M.mesh_polyvertadr[padr + idx], nvert = M.mesh_polyvertnum[padr + idx];
      for (int j = 0; j < nvert; j++) {
        if (M.mesh_polyvert[adr + j] == v1i) {
          const int k = (j == 0) ? nvert - 1 : j - 1;
          const int vb = 3*(vadr + M.mesh_polyvert[adr + k]);
          const V3 ev = ccd_globalcoord(o.r + CO_MAT, o.r + CO_POS, M.mesh_vert[vb], M.mesh_vert[vb + 1], M.mesh_vert[vb + 2]);
          st3(endverts + 3*i, ev);
          V3 r = ev - v1;
          unitize(r);
          st3(res + 3*i, r);
          break;
        }
      }
    }
    return n1;
  }
  return 0;
}
// boxNormals2 (:1898)
MJH_DEV int ccd_box_normals2(rptr res, iptr resind, crptr mat, V3 n) {
  V3 ln{mat[0]*n.x + mat[3]*n.y + mat[6]*n.z, mat[1]*n.x + mat[4]*n.y + mat[7]*n.z, mat[2]*n.x + mat[5]*n.y + mat[8]*n.z};
  ln = ccd_scl(ln, 1/sqrt(dot(ln, ln)));
  for (int i = 0; i < 6; i++) {
    const V3 nr = with_comp(V3{0, 0, 0}, i >> 1, (i & 1) ? -1 : 1);
    if (dot(ln, nr) > MJH_CCD_FACE_TOL) {
      st3(res, ccd_globalrot(mat, nr.x, nr.y, nr.z));
      resind[0] = i;
      return 1;
    }
  }
  return 0;
}
// boxNormals (:1924)
MJH_DEV int ccd_box_normals(rptr res, iptr resind, int dim, CcdObj o, const int* vi, V3 dir) {
  const int v1 = vi[0], v2 = vi[1], v3 = vi[2];
  const crptr mat = o.r + CO_MAT;
  if (dim == 3) {
    int cn = 0;
    const int x = ((v1 & 1) && (v2 & 1) && (v3 & 1)) - (!(v1 & 1) && !(v2 & 1) && !(v3 & 1));
    const int y = ((v1 & 2) && (v2 & 2) && (v3 & 2)) - (!(v1 & 2) && !(v2 & 2) && !(v3 & 2));
    const int z = ((v1 & 4) && (v2 & 4) && (v3 & 4)) - (!(v1 & 4) && !(v2 & 4) && !(v3 & 4));
    st3(res, ccd_globalrot(mat, x, y, z));
    const int sgn = x + y + z;
    if (x) resind[cn++] = 0;
    if (y) resind[cn++] = 2;
    if (z) resind[cn++] = 4;
    if (sgn == -1) resind[0]++;
    return cn == 1 ? 1 : ccd_box_normals2(res, resind, mat, dir);
  }
  if (dim == 2) {
    int cn = 0;
    const int x = ((v1 & 1) && (v2 & 1)) - (!(v1 & 1) && !(v2 & 1));
    const int y = ((v1 & 2) && (v2 & 2)) - (!(v1 & 2) && !(v2 & 2));
    const int z = ((v1 & 4) && (v2 & 4)) - (!(v1 & 4) && !(v2 & 4));
    if (x) { st3(res, ccd_globalrot(mat, x, 0, 0)); resind[cn++] = (x > 0) ? 0 : 1; }
    if (y) { st3(res + 3*cn, ccd_globalrot(mat, 0, y, 0)); resind[cn++] = (y > 0) ? 2 : 3; }
    if (z) { st3(res + 3, ccd_globalrot(mat, 0, 0, z)); resind[cn++] = (z > 0) ? 4 : 5; }
    return cn == 2 ? 2 : ccd_box_normals2(res, resind, mat, dir);
  }
  if (dim == 1) {
    const real x = (v1 & 1) ? 1 : -1, y = (v1 & 2) ? 1 : -1, z = (v1 & 4) ? 1 : -1;
    st3(res, ccd_globalrot(mat, x, 0, 0));
    st3(res + 3, ccd_globalrot(mat, 0, y, 0));
    st3(res + 6, ccd_globalrot(mat, 0, 0, z));
    resind[0] = (x > 0) ? 0 : 1;
    resind[1] = (y > 0) ? 2 : 3;
    resind[2] = (z > 0) ? 4 : 5;
    return 3;
  }
  return 0;
}
// boxEdgeNormals (:1973)
MJH_DEV int ccd_box_edge_normals(rptr res, rptr endverts, int dim, CcdObj o, const real* v, int v1i) {
  const V3 v1 = ld3(v), v2 = ld3(v + 3);
  const crptr mat = o.r + CO_MAT; const crptr pos = o.r + CO_POS; const crptr size = o.r + CO_SIZE;
  if (dim == 2) {
    st3(endverts, v2);
    V3 r = v2 - v1;
    unitize(r);
    st3(res, r);
    return 1;
  }
  if (dim == 1) {
    const real x = (v1i & 1) ? size[0] : -size[0];
    const real y = (v1i & 2) ? size[1] : -size[1];
    const real z = (v1i & 4) ? size[2] : -size[2];
    for (int k = 0; k < 3; k++) {
      const V3 ev = ccd_globalcoord(mat, pos, k == 0 ? -x : x, k == 1 ? -y : y, k == 2 ? -z : z);
      st3(endverts + 3*k, ev);
      V3 r = ev - v1;
      unitize(r);
      st3(res + 3*k, r);
    }
    return 3;
  }
  return 0;
}
// boxFace (:2011): the four corners of face idx, counter-clockwise seen from outside
MJH_DEV int ccd_box_face(rptr res, CcdObj o, int idx) {
  const crptr mat = o.r + CO_MAT; const crptr pos = o.r + CO_POS; const crptr size = o.r + CO_SIZE;
  // sign patterns (x, y, z) of the four corners of each face, in the reference's order
  const signed char pat[6][4][3] = {
    {{1, 1, 1}, {1, 1, -1}, {1, -1, -1}, {1, -1, 1}},
    {{-1, 1, -1}, {-1, 1, 1}, {-1, -1, 1}, {-1, -1, -1}},
    {{-1, 1, -1}, {1, 1, -1}, {1, 1, 1}, {-1, 1, 1}},
    {{-1, -1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1}},
    {{-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}},
    {{1, 1, -1}, {-1, 1, -1}, {-1, -1, -1}, {1, -1, -1}}};
  if (idx < 0 || idx > 5) return 0;
  for (int k = 0; k < 4; k++)
    st3(res + 3*k, ccd_globalcoord(mat, pos, pat[idx][k][0] > 0 ? size[0] : -size[0], pat[idx][k][1] > 0 ? size[1] : -size[1],
                                   pat[idx][k][2] > 0 ? size[2] : -size[2]));
  return 4;
}
// meshFace (:2068): polygon idx in reverse vertex order
MJH_DEV int ccd_mesh_face(MREF M, const Ccd& c, rptr res, CcdObj o, int idx) {
  const int mesh = o.i[CI_MESH];
  const int vadr = M.mesh_vertadr[mesh], padr = M.mesh_polyadr[mesh];
  const int adr = M.mesh_polyvertadr[padr + idx];
  const int nvert = M.mesh_polyvertnum[padr + idx];
  int j = 0;
  for (int i = nvert - 1; i >= 0; i--) {
    const int vb = 3*(vadr + M.mesh_polyvert[adr + i]);
    if (j < c.P) st3(res + 3*j, ccd_globalcoord(o.r + CO_MAT, o.r + CO_POS, M.mesh_vert[vb], M.mesh_vert[vb + 1], M.mesh_vert[vb + 2]));
    j++;
  }
  return nvert < c.P ? nvert : c.P;
}
// simplexDim (:2112)
MJH_DEV int ccd_simplex_dim(int* vi, real* v) {
  if (vi[0] == vi[1]) {
    if (vi[0] == vi[2]) return 1;
    vi[1] = vi[2];
    v[3] = v[6]; v[4] = v[7]; v[5] = v[8];
    return 2;
  }
  return (vi[2] == vi[0] || vi[2] == vi[1]) ? 2 : 3;
}

// multicontact (:2123)
MJH_DEVN_HOT void ccd_multicontact(MREF M, Ccd& c, int face) {
  const int t1 = c.o1.i[CI_TYPE], t2 = c.o2.i[CI_TYPE];
  if (t1 == MJH_GEOM_MESH && !M.mesh_polynum[c.o1.i[CI_MESH]]) return;
  if (t2 == MJH_GEOM_MESH && !M.mesh_polynum[c.o2.i[CI_MESH]]) return;
  const int fv[3] = {ccd_face_vert(c, face, 0), ccd_face_vert(c, face, 1), ccd_face_vert(c, face, 2)};
  int v1i[3], v2i[3];
  real v1[9], v2[9];
  for (int k = 0; k < 3; k++) {
    v1i[k] = c.vi[CV_NINT*fv[k]]; v2i[k] = c.vi[CV_NINT*fv[k] + 1];
    for (int q = 0; q < 3; q++) { v1[3*k + q] = c.vr[CV_NREAL*fv[k] + 3 + q]; v2[3*k + q] = c.vr[CV_NREAL*fv[k] + 6 + q]; }
  }
  // buffers (overlay the polytope, whose data was saved above)
  const int D = c.D, P = c.P;
  const iptr idx1 = c.mci, idx2 = idx1 + D;
  const rptr n1 = c.mcr, n2 = n1 + 3*D, endverts = n2 + 3*D;
  const rptr face1 = endverts + 3*D, face2 = face1 + 3*P, polygon = face2 + 3*P;
  int nface1 = ccd_simplex_dim(v1i, v1);
  int nface2 = ccd_simplex_dim(v2i, v2);
  int nnorms1 = 0, nnorms2 = 0;
  const V3 dir = ld3(c.x2) - ld3(c.x1);
  const V3 dir_neg = ld3(c.x1) - ld3(c.x2);
  if (t1 == MJH_GEOM_BOX) nnorms1 = ccd_box_normals(n1, idx1, nface1, c.o1, v1i, dir_neg);
  else if (t1 == MJH_GEOM_MESH) nnorms1 = ccd_mesh_normals(M, c, n1, idx1, nface1, c.o1, v1i);
  if (t2 == MJH_GEOM_BOX) nnorms2 = ccd_box_normals(n2, idx2, nface2, c.o2, v2i, dir);
  else if (t2 == MJH_GEOM_MESH) nnorms2 = ccd_mesh_normals(M, c, n2, idx2, nface2, c.o2, v2i);
  int res0 = 0, res1 = 0, edgecon1 = 0, edgecon2 = 0;
  // alignedFaces (:2086)
  int aligned = 0;
  for (int i = 0; i < nnorms1 && !aligned; i++)
    for (int j = 0; j < nnorms2; j++)
      if (dot(ld3(n1 + 3*i), ld3(n2 + 3*j)) < -MJH_CCD_FACE_TOL) { res0 = i; res1 = j; aligned = 1; break; }
  if (!aligned) {
    if (nface1 < 3 && nface1 <= nface2) {
      nnorms1 = 0;
      if (t1 == MJH_GEOM_BOX) nnorms1 = ccd_box_edge_normals(n1, endverts, nface1, c.o1, v1, v1i[0]);
      else if (t1 == MJH_GEOM_MESH) nnorms1 = ccd_mesh_edge_normals(M, c, n1, endverts, nface1, c.o1, v1, v1i[0]);
      // alignedFaceEdge(res, n1, nnorms1, n2, nnorms2) (:2099): faces outer, edges inner
      int found = 0;
      for (int i = 0; i < nnorms2 && !found; i++)
        for (int j = 0; j < nnorms1; j++)
          if (ccd_abs(dot(ld3(n1 + 3*j), ld3(n2 + 3*i))) < MJH_CCD_EDGE_TOL) { res0 = j; res1 = i; found = 1; break; }
      if (!found) return;
      edgecon1 = 1;
    } else if (nface2 < 3) {
      nnorms2 = 0;
      if (t2 == MJH_GEOM_BOX) nnorms2 = ccd_box_edge_normals(n2, endverts, nface2, c.o2, v2, v2i[0]);
      else if (t2 == MJH_GEOM_MESH) nnorms2 = ccd_mesh_edge_normals(M, c, n2, endverts, nface2, c.o2, v2, v2i[0]);
      int found = 0;
      for (int i = 0; i < nnorms1 && !found; i++)
        for (int j = 0; j < nnorms2; j++)
          if (ccd_abs(dot(ld3(n2 + 3*j), ld3(n1 + 3*i))) < MJH_CCD_EDGE_TOL) { res0 = j; res1 = i; found = 1; break; }
      if (!found) return;
      edgecon2 = 1;
    } else {
      return;
    }
  }
  const int i = res0, j = res1;
  if (edgecon1) {
    st3(face1, ld3(v1));
    st3(face1 + 3, ld3(endverts + 3*i));
    nface1 = 2;
  } else {
    const int ind = edgecon2 ? idx1[j] : idx1[i];
    if (t1 == MJH_GEOM_BOX) nface1 = ccd_box_face(face1, c.o1, ind);
    else if (t1 == MJH_GEOM_MESH) nface1 = ccd_mesh_face(M, c, face1, c.o1, ind);
  }
  if (edgecon2) {
    st3(face2, ld3(v2));
    st3(face2 + 3, ld3(endverts + 3*i));
    nface2 = 2;
  } else {
    if (t2 == MJH_GEOM_BOX) nface2 = ccd_box_face(face2, c.o2, idx2[j]);
    else if (t2 == MJH_GEOM_MESH) nface2 = ccd_mesh_face(M, c, face2, c.o2, idx2[j]);
  }
  if (edgecon1) {
    const V3 nj = ld3(n2 + 3*j);
    ccd_polygon_clip(c, face2, nface2, face1, nface1, nj, ccd_scl(nj, -1.0), polygon);
    for (int k = 0; k < c.nx; k++) {
      const V3 tmp = ld3(c.x1 + 3*k);
      st3(c.x1 + 3*k, ld3(c.x2 + 3*k));
      st3(c.x2 + 3*k, tmp);
    }
    return;
  }
  if (edgecon2) {
    const V3 nj = ld3(n1 + 3*j);
    ccd_polygon_clip(c, face1, nface1, face2, nface2, nj, ccd_scl(nj, -1.0), polygon);
    return;
  }
  ccd_polygon_clip(c, face1, nface1, face2, nface2, ld3(n1 + 3*i), ld3(n2 + 3*j), polygon);
}

// ---- mjc_ccd (:2318) --------------------------------------------------------------------------------------------
// returns the smallest witness distance (negative: penetration)
MJH_DEVN_HOT real ccd_run(MREF M, Ccd& c) {
  // (mjc_center: a geom's position, the bounding-box centre of a flex element)
  const int ctr1 = c.o1.i[CI_SUP] == CCD_SUP_FLEXELEM ? CO_CENTER : CO_POS;
  const int ctr2 = c.o2.i[CI_SUP] == CCD_SUP_FLEXELEM ? CO_CENTER : CO_POS;
  st3(c.x1, ld3(c.o1.r + ctr1));
  st3(c.x2, ld3(c.o2.r + ctr2));
  c.gjk_iterations = 0;
  c.dist_cutoff = 0;
  const int t1 = c.o1.i[CI_TYPE], t2 = c.o2.i[CI_TYPE];
  if (t1 == MJH_GEOM_SPHERE || t2 == MJH_GEOM_SPHERE || t1 == MJH_GEOM_CAPSULE || t2 == MJH_GEOM_CAPSULE) {
    // shrink spheres to points and capsules to segments, inflate the result
    const int sup1 = c.o1.i[CI_SUP], sup2 = c.o2.i[CI_SUP];
    const real margin1 = c.o1.r[CO_MARGIN], margin2 = c.o2.r[CO_MARGIN];
    real full1 = 0, full2 = 0;
    if (t1 == MJH_GEOM_SPHERE) { full1 = c.o1.r[CO_SIZE] + 0.5*margin1; c.o1.i[CI_SUP] = CCD_SUP_POINT; c.o1.r[CO_MARGIN] = 0; }
    else if (t1 == MJH_GEOM_CAPSULE) { full1 = c.o1.r[CO_SIZE] + 0.5*margin1; c.o1.i[CI_SUP] = CCD_SUP_LINE; c.o1.r[CO_MARGIN] = 0; }
    if (t2 == MJH_GEOM_SPHERE) { full2 = c.o2.r[CO_SIZE] + 0.5*margin2; c.o2.i[CI_SUP] = CCD_SUP_POINT; c.o2.r[CO_MARGIN] = 0; }
    else if (t2 == MJH_GEOM_CAPSULE) { full2 = c.o2.r[CO_SIZE] + 0.5*margin2; c.o2.i[CI_SUP] = CCD_SUP_LINE; c.o2.r[CO_MARGIN] = 0; }
    c.dist_cutoff += full1 + full2;
    ccd_gjk(M, c);
    c.dist_cutoff = 0;
    c.o1.r[CO_MARGIN] = margin1; c.o2.r[CO_MARGIN] = margin2;
    c.o1.i[CI_SUP] = sup1; c.o2.i[CI_SUP] = sup2;
    if (c.dist[0] > c.tolerance) {
      // inflate (:2281)
      V3 n = ld3(c.x2) - ld3(c.x1);
      unitize(n);
      if (full1) { c.x1[0] += full1*n.x; c.x1[1] += full1*n.y; c.x1[2] += full1*n.z; }
      if (full2) { c.x2[0] -= full2*n.x; c.x2[1] -= full2*n.y; c.x2[2] -= full2*n.z; }
      c.dist[0] -= (full1 + full2);
      if (c.dist[0] > c.dist_cutoff) c.dist[0] = MJH_CCD_MAX;
      return c.dist[0];
    }
    c.gjk_iterations = 0;
    st3(c.x1, ld3(c.o1.r + ctr1));
    st3(c.x2, ld3(c.o2.r + ctr2));
  }
  ccd_gjk(M, c);
  if (c.dist[0] <= c.tolerance && c.nsimplex > 1 && !c.separated) {
    c.dist[0] = 0;
    c.nfaces = c.nmap = c.nverts = c.nedges = 0;
    int ret;
    if (c.nsimplex == 2) ret = ccd_polytope2(M, c);
    else if (c.nsimplex == 3) ret = ccd_polytope3(M, c);
    else ret = ccd_polytope4(M, c);
    if (!ret) {
      const int face = ccd_epa(M, c);
      if (c.max_contacts > 1 && face >= 0) ccd_multicontact(M, c, face);
    }
  }
  real min_dist = c.dist[0];
  for (int i = 1; i < c.nx; i++) if (c.dist[i] < min_dist) min_dist = c.dist[i];
  return min_dist;
}

// mjc_penetration (:87): contacts into out + 7*first, returns their number
MJH_DEV int ccd_penetration(MREF M, Ccd& c, int first, int nconmax, real margin) {
  c.max_contacts = nconmax;
  if (ccd_run(M, c) < 0) {
    const int nw = c.nx;
    for (int i = 0; i < nw; i++) {
      const rptr o = c.out + 7*(first + i);
      o[0] = margin + c.dist[i];
      V3 pos = ld3(c.x1 + 3*i) + ld3(c.x2 + 3*i);
      pos = V3{pos.x*0.5, pos.y*0.5, pos.z*0.5};
      st3(o + 1, pos);
      V3 nrm = ld3(c.x1 + 3*i) - ld3(c.x2 + 3*i);
      unitize(nrm);
      st3(o + 4, nrm);
    }
    return nw;
  }
  return 0;
}

// mjc_initCCDObj (:726)
template <class GX, class GM>
MJH_DEV void ccd_init_obj(MREF M, CcdObj o, GX gx, GM gm, int g, real margin) {
  for (int k = 0; k < 3; k++) { o.r[CO_SIZE + k] = M.geom_size[3*g + k]; o.r[CO_POS + k] = gx[3*g + k]; }
  for (int k = 0; k < 9; k++) o.r[CO_MAT + k] = gm[9*g + k];
  o.r[CO_MARGIN] = margin;
  const int type = M.geom_type[g];
  o.i[CI_TYPE] = type;
  o.i[CI_VERTINDEX] = -1;
  o.i[CI_MESHINDEX] = -1;
  o.i[CI_MESH] = -1;
  int sup = CCD_SUP_POINT;
  if (type == MJH_GEOM_SPHERE) sup = CCD_SUP_SPHERE;
  else if (type == MJH_GEOM_CAPSULE) sup = CCD_SUP_CAPSULE;
  else if (type == MJH_GEOM_ELLIPSOID) sup = CCD_SUP_ELLIPSOID;
  else if (type == MJH_GEOM_CYLINDER) sup = CCD_SUP_CYLINDER;
  else if (type == MJH_GEOM_BOX) sup = CCD_SUP_BOX;
  else if (type == MJH_GEOM_MESH) {
    const int mesh = M.geom_dataid[g];
    o.i[CI_MESH] = mesh;
    sup = (M.mesh_graphadr[mesh] < 0 || M.mesh_vertnum[mesh] < 10) ? CCD_SUP_MESH : CCD_SUP_HILLCLIMB;    // mjMESH_HILLCLIMB_MIN
  }
  o.i[CI_SUP] = sup;
}

// carve the lane's workspace slice: word w of lane l of environment e sits at [e][w][l]
MJH_DEV void ccd_carve(MREF M, BREF B, int e, Ccd& c) {
  const MJH_CONST_AS DSizes& s = M.s;
  char* base = (char*)B.ccd_ws + (size_t)e*MJH_WAVE*(size_t)s.ccd_lane_bytes;
  rptr r{(real*)base + wv_lane(), MJH_WAVE};
  iptr ip{(int*)(base + (size_t)s.ccd_nreal*MJH_WAVE*sizeof(real)) + wv_lane(), MJH_WAVE};
  const int N = s.ccd_N;
  c.N = N; c.P = s.ccd_P; c.D = s.ccd_D;
  c.maxfaces = 6*N; c.maxhorizon = 6*N;
  c.tolerance = M.o.ccd_tolerance;
  c.o1.r = r; r = r + CO_NREAL;
  c.o2.r = r; r = r + CO_NREAL;
  c.x1 = r; r = r + 3*CCD_MAXWIT;
  c.x2 = r; r = r + 3*CCD_MAXWIT;
  c.dist = r; r = r + CCD_MAXWIT;
  c.simr = r; r = r + 4*CV_NREAL;
  c.tmpr = r; r = r + 5*CV_NREAL;
  c.out = r; r = r + 7*CCD_MAXOUT;
  c.vr = r; c.mcr = r; r = r + (5 + N)*CV_NREAL;
  c.fr = r;
  c.o1.i = ip; ip = ip + CI_NINT;
  c.o2.i = ip; ip = ip + CI_NINT;
  c.simi = ip; ip = ip + 4*CV_NINT;
  c.tmpi = ip; ip = ip + 5*CV_NINT;
  c.hidx = ip; ip = ip + 6*N;
  c.hedge = ip; ip = ip + 6*N;
  c.stack = ip; ip = ip + 2*(6*N + 1);
  c.vi = ip; c.mci = ip; ip = ip + (5 + N)*CV_NINT;
  c.fi = ip; ip = ip + 6*N*CF_NINT;
  c.map = ip;
  c.separated = 0; c.nx = 0; c.nsimplex = 0; c.gjk_iterations = 0;
  c.nverts = c.nfaces = c.nmap = c.nedges = 0;
  c.max_contacts = 1; c.dist_cutoff = 0;
  c.center = V3{0, 0, 0}; c.horizon_w = V3{0, 0, 0};
}

// the calling lane's contact records (dist, pos[3], normal[3]) x CCD_MAXOUT
MJH_DEV crptr ccd_out_records(MREF M, BREF B, int e) {
  const real* base = (const real*)((const char*)B.ccd_ws + (size_t)e*MJH_WAVE*(size_t)M.s.ccd_lane_bytes);
  return crptr{base + (size_t)(2*CO_NREAL + 7*CCD_MAXWIT + 4*CV_NREAL + 5*CV_NREAL)*MJH_WAVE + wv_lane(), MJH_WAVE};
}

// mjc_Convex (:881): returns the number of contacts left in the lane's `out` records
MJH_DEVN_HOT int ccd_convex_pair(MREF M_, BREF B_, int e_, int p) {
  MJH_ENTER(M_, B_, e_);
  if (p < 0) return 0;
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  Ccd c;
  ccd_carve(M, B, e, c);
  const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
  const real margin = M.pair_margin[p];
  ccd_init_obj(M, c.o1, gx, gm, g1, margin);
  ccd_init_obj(M, c.o2, gx, gm, g2, margin);
  const int t1 = c.o1.i[CI_TYPE], t2 = c.o2.i[CI_TYPE];
  const int multiccd = !(M.o.disableflags & (1 << 19));
  // maxContacts (:855)
  int max_contacts = 1;
  if (!(margin > 0) && multiccd && (t1 == MJH_GEOM_BOX || t1 == MJH_GEOM_MESH) && (t2 == MJH_GEOM_BOX || t2 == MJH_GEOM_MESH))
    max_contacts = (t1 == MJH_GEOM_BOX && t2 == MJH_GEOM_BOX) ? 8 : 4;
  int ncon = ccd_penetration(M, c, 0, max_contacts, margin);
  if (max_contacts > 1) return ncon;
  if (ncon == 1 && multiccd && t1 != MJH_GEOM_ELLIPSOID && t1 != MJH_GEOM_SPHERE && t2 != MJH_GEOM_ELLIPSOID && t2 != MJH_GEOM_SPHERE) {
    // perturbation multi-contact: rotate both geoms by +-1e-3 rad about the two tangents of the first contact
    real frame[9] = {c.out[4], c.out[5], c.out[6], 0, 0, 0, 0, 0, 0};
    make_frame(frame);
    const real tolerance = 1e-3*r_min(M.geom_rbound[g1], M.geom_rbound[g2]);
    const V3 origin = ld3(c.out + 1);
    for (int axis_id = 0; axis_id < 2; axis_id++) {
      for (int angle_id = 0; angle_id < 2; angle_id++) {
        const real* axis = frame + 3 + 3*axis_id;
        // mji_axisAngle2Quat with angle -+1e-3: sin / cos of 5e-4 evaluated by the host's libm at upload
        const real sn = angle_id == 0 ? -M.o.ccd_sin : M.o.ccd_sin;
        const real quat[4] = {M.o.ccd_cos, axis[0]*sn, axis[1]*sn, axis[2]*sn};
        real rot[9], invrot[9];
        q_tomat(rot, quat);
        for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) invrot[3*r + q] = rot[3*q + r];
        for (int side = 0; side < 2; side++) {
          // mju_rotateFrame (:834)
          const real* R = side == 0 ? rot : invrot;
          const rptr xmat = (side == 0 ? c.o1.r : c.o2.r) + CO_MAT;
          const rptr xpos = (side == 0 ? c.o1.r : c.o2.r) + CO_POS;
          real mat[9];
          for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++)
            mat[3*r + q] = R[3*r]*xmat[q] + R[3*r + 1]*xmat[3 + q] + R[3*r + 2]*xmat[6 + q];
          for (int k = 0; k < 9; k++) xmat[k] = mat[k];
          const V3 rel = origin - ld3(xpos);
          V3 vec = mmul(R, rel);
          vec = vec - rel;
          xpos[0] -= vec.x; xpos[1] -= vec.y; xpos[2] -= vec.z;
        }
        const int n = ccd_penetration(M, c, ncon, 1, margin);
        if (n) {
          // mjc_isDistinctContact (:822)
          int distinct = 1;
          const V3 last = ld3(c.out + 7*ncon + 1);
          for (int i = 0; i < ncon; i++) {
            const V3 df = ld3(c.out + 7*i + 1) - last;
            if (sqrt(df.x*df.x + df.y*df.y + df.z*df.z) <= tolerance) { distinct = 0; break; }
          }
          if (distinct) { c.out[7*ncon] = c.out[0]; ncon++; }
        }
        for (int k = 0; k < 3; k++) { c.o1.r[CO_POS + k] = gx[3*g1 + k]; c.o2.r[CO_POS + k] = gx[3*g2 + k]; }
        for (int k = 0; k < 9; k++) { c.o1.r[CO_MAT + k] = gm[9*g1 + k]; c.o2.r[CO_MAT + k] = gm[9*g2 + k]; }
      }
    }
  }
  return ncon;
}

// mjc_ConvexElem (:1559) for a geom against a solid flex element (tetrahedron, 4 corners): one contact at most, left in the
// lane's first `out` record.  g < 0: the lane has no pair.
MJH_DEVN_HOT int ccd_geom_elem_pair(MREF M_, BREF B_, int e_, int g, int elem, real margin) {
  MJH_ENTER(M_, B_, e_);
  if (g < 0) return 0;
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr aabb = MJH_F(B, flexelem_aabb, e);
  Ccd c;
  ccd_carve(M, B, e, c);
  ccd_init_obj(M, c.o1, gx, gm, g, margin);
  const int f = M.flexelem_flex[elem];
  const int n = M.flex_dim[f] + 1;
  for (int i = 0; i < n; i++) {
    const int v = M.flexelem_vert[4*elem + i];
    for (int k = 0; k < 3; k++) c.o2.r[3*i + k] = vx[3*v + k];
  }
  c.o2.r[CO_SIZE] = M.flex_radius[f] + 0.5*margin;
  c.o2.r[CO_MARGIN] = 0;
  for (int k = 0; k < 3; k++) c.o2.r[CO_CENTER + k] = aabb[6*elem + k];
  c.o2.i[CI_TYPE] = MJH_GEOM_FLEX;
  c.o2.i[CI_SUP] = CCD_SUP_FLEXELEM;
  c.o2.i[CI_VERTINDEX] = -1;
  c.o2.i[CI_MESHINDEX] = -1;
  c.o2.i[CI_MESH] = n;
  return ccd_penetration(M, c, 0, 1, margin);
}

// mjccd_support (:518) for the geoms mjc_PlaneConvex sees (ellipsoid, mesh): libccd-style support
MJH_DEV V3 ccd_legacy_support(MREF M, CcdObj o, V3 dir) {
  const crptr mat = o.r + CO_MAT; const crptr pos = o.r + CO_POS; const crptr size = o.r + CO_SIZE;
  const V3 ld = ccd_to_local(mat, dir);
  V3 res;
  if (o.i[CI_TYPE] == MJH_GEOM_ELLIPSOID) {
    res = V3{ld.x*size[0], ld.y*size[1], ld.z*size[2]};
    unitize(res);
    res = V3{res.x*size[0], res.y*size[1], res.z*size[2]};
  } else {
    const int mesh = o.i[CI_MESH];
    const int vadr = 3*M.mesh_vertadr[mesh];
    real tmp = -1E+10;
    int ibest = -1;
    if (o.i[CI_SUP] == CCD_SUP_MESH) {
      const int nvert = M.mesh_vertnum[mesh];
      for (int i = 0; i < nvert; i++) {
        const real vdot = ccd_dot3f(M, ld, vadr + 3*i);
        if (vdot > tmp) { tmp = vdot; ibest = i; }
      }
      o.i[CI_MESHINDEX] = ibest;
    } else {
      const int gadr = M.mesh_graphadr[mesh];
      const int numvert = M.mesh_graph[gadr];
      const int edgeadr = gadr + 2, globalid = gadr + 2 + numvert, localid = gadr + 2 + 2*numvert;
      ibest = o.i[CI_MESHINDEX] < 0 ? 0 : o.i[CI_MESHINDEX];
      tmp = ccd_dot3f(M, ld, vadr + 3*M.mesh_graph[globalid + ibest]);
      int change = 1;
      while (change) {
        change = 0;
        int i = M.mesh_graph[edgeadr + ibest], locid;
        while ((locid = M.mesh_graph[localid + i]) >= 0) {
          const real vdot = ccd_dot3f(M, ld, vadr + 3*M.mesh_graph[globalid + locid]);
          if (vdot > tmp) { tmp = vdot; ibest = locid; change = 1; }
          i++;
        }
      }
      o.i[CI_MESHINDEX] = ibest;
      ibest = M.mesh_graph[globalid + ibest];
    }
    if (ibest < 0) res = V3{0, 0, 0};
    else res = V3{(real)M.mesh_vert[vadr + 3*ibest], (real)M.mesh_vert[vadr + 3*ibest + 1], (real)M.mesh_vert[vadr + 3*ibest + 2]};
  }
  // + local_dir * margin / 2 with margin 0
  res = V3{res.x + ld.x*o.r[CO_MARGIN]/2, res.y + ld.y*o.r[CO_MARGIN]/2, res.z + ld.z*o.r[CO_MARGIN]/2};
  res = mmul(mat, res);
  return V3{res.x + pos[0], res.y + pos[1], res.z + pos[2]};
}

// mjc_PlaneConvex (:1004): plane against ellipsoid / mesh; up to three contacts
MJH_DEVN_HOT int ccd_plane_convex_pair(MREF M_, BREF B_, int e_, int p) {
  MJH_ENTER(M_, B_, e_);
  if (p < 0) return 0;
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  Ccd c;
  ccd_carve(M, B, e, c);
  const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
  const real margin = M.pair_margin[p];
  const V3 pos1 = ld3(gx + 3*g1), pos2 = ld3(gx + 3*g2);
  crptr mat1 = gm + 9*g1;
  const V3 normal{mat1[2], mat1[5], mat1[8]};
  ccd_init_obj(M, c.o1, gx, gm, g2, 0);
  const V3 cdir{-mat1[2], -mat1[5], -mat1[8]};
  const V3 sup = ccd_legacy_support(M, c.o1, cdir);
  const rptr o = c.out;
  o[0] = dot(normal, sup - pos1);
  if (o[0] > margin) return 0;
  const real h = -0.5*o[0];
  st3(o + 1, V3{sup.x + normal.x*h, sup.y + normal.y*h, sup.z + normal.z*h});
  st3(o + 4, normal);
  int count = 1;
  if (M.geom_dataid[g2] == -1) return count;
  const int mesh = M.geom_dataid[g2];
  const int vadr = 3*M.mesh_vertadr[mesh];
  const crptr mat2 = c.o1.r + CO_MAT;
  const V3 locdir = ccd_to_local(mat2, cdir);
  const real threshold = dot(normal, pos2 - pos1) - margin;
  const V3 first = ld3(o + 1);
  const real rbound = M.geom_rbound[g2];
  // addplanemesh (:970)
  auto add = [&](int vb) -> int {
    const V3 v{(real)M.mesh_vert[vb], (real)M.mesh_vert[vb + 1], (real)M.mesh_vert[vb + 2]};
    const V3 pnt = mmul(mat2, v) + pos2;
    const V3 df = pnt - first;
    if (sqrt(df.x*df.x + df.y*df.y + df.z*df.z) < 0.3*rbound) return 0;
    const rptr oc = c.out + 7*count;
    oc[0] = dot(normal, pnt - pos1);
    const real hh = -0.5*oc[0];
    st3(oc + 1, V3{pnt.x + normal.x*hh, pnt.y + normal.y*hh, pnt.z + normal.z*hh});
    st3(oc + 4, normal);
    return 1;
  };
  if (M.mesh_graphadr[mesh] < 0) {
    const int nvert = M.mesh_vertnum[mesh];
    for (int i = 0; i < nvert && count < 3; i++) {
      const real vdot = locdir.x*(real)M.mesh_vert[vadr + 3*i] + locdir.y*(real)M.mesh_vert[vadr + 3*i + 1] + locdir.z*(real)M.mesh_vert[vadr + 3*i + 2];
      if (vdot > threshold && i != c.o1.i[CI_MESHINDEX]) count += add(vadr + 3*i);
    }
  } else if (c.o1.i[CI_MESHINDEX] >= 0) {
    const int gadr = M.mesh_graphadr[mesh];
    const int numvert = M.mesh_graph[gadr];
    const int edgeadr = gadr + 2, globalid = gadr + 2 + numvert, localid = gadr + 2 + 2*numvert;
    int i = M.mesh_graph[edgeadr + c.o1.i[CI_MESHINDEX]], locid;
    while ((locid = M.mesh_graph[localid + i]) >= 0 && count < 3) {
      const int vb = vadr + 3*M.mesh_graph[globalid + locid];
      const real vdot = locdir.x*(real)M.mesh_vert[vb] + locdir.y*(real)M.mesh_vert[vb + 1] + locdir.z*(real)M.mesh_vert[vb + 2];
      if (vdot > threshold) count += add(vb);
      i++;
    }
  }
  return count;
}

#endif  // !MJH_LANE_MODE
