// Test-infrastructure stand-in for qhull's reentrant API header "qhull_ra.h" (qhull is not on
// disk and is a third-party dependency of the reference's model COMPILER, not of the mj_step path).
// Only the names user_mesh.cc (mjCMesh::MakeGraph, /root/reference/src/user/user_mesh.cc:1640-1880)
// uses are declared.
//
// Unlike the round-1/2 stub this one builds a real convex hull, so that models with convex mesh
// geoms (BASELINE config 4, model/cube/cube_3x3x3.xml) compile in the oracle build:
//   * qh_qhull      : incremental 3-D hull (initial tetrahedron, then one point at a time: faces that
//                     see the point strictly are removed, the horizon is re-triangulated to the
//                     point).  Points coplanar with a face do not see it, so a planar polygon of the
//                     hull comes out as a fan/strip of coplanar triangles -- what "qhull Qt" yields
//                     too, up to the choice of diagonals and the order of vertices and facets.
//   * the vertex / facet lists, vertex->neighbors, facet->vertices and toporient are filled the way
//     MakeGraph walks them.
// The ORDER of hull vertices, of a vertex's edges and of the facets differs from real qhull's.
// That order is compiled into mjModel (mesh_graph, mesh_poly*): it is an INPUT shared by the
// oracle and the GPU path, which both step the same mjModel (or the .mjb written from it), so
// parity of mj_step is unaffected; trajectories may differ from an official MuJoCo build's wherever
// the hill-climbing support function breaks exact ties by graph order.
// "TA<n>" (maxhullvert) is not supported: such models still report "qhull error".
#ifndef ORACLE_STUB_QHULL_RA_H_
#define ORACLE_STUB_QHULL_RA_H_
#include <csetjmp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <map>
#include <utility>
typedef double coordT;
typedef coordT pointT;
typedef unsigned int boolT;
#define qh_False 0
#define qh_True 1
#define qh_ALL 1
union setelemT { void* p; int i; };
struct setT { int maxsize; setelemT e[1]; };
struct vertexT { vertexT* next; pointT* point; setT* neighbors; };
struct facetT { facetT* next; setT* vertices; unsigned toporient; };
struct qhT {
  jmp_buf errexit;
  boolT NOerrexit;
  int num_vertices, num_facets;
  vertexT* vertex_list;
  facetT* facet_list;
  // stub state
  coordT* first_point;
  int num_points;
  int unsupported;
  std::vector<vertexT>* verts;
  std::vector<facetT>* facets;
  std::vector<setT*>* sets;
  std::vector<int>* tri;     // hull triangles (point ids), counter-clockwise seen from outside
};
#define FORALLvertices for (vertex = qh->vertex_list; vertex && vertex->next; vertex = vertex->next)
#define FORALLfacets for (facet = qh->facet_list; facet && facet->next; facet = facet->next)
#define FOREACHsetelement_(type, set, variable) \
  if (((variable = NULL), set)) \
    for (variable##p = (type**)&((set)->e[0].p); (variable = *variable##p++);)

inline void qh_zero(qhT* qh, FILE*) {
  qh->NOerrexit = 1; qh->num_vertices = qh->num_facets = 0; qh->vertex_list = 0; qh->facet_list = 0;
  qh->first_point = 0; qh->num_points = 0; qh->unsupported = 0;
  qh->verts = 0; qh->facets = 0; qh->sets = 0; qh->tri = 0;
}
inline void qh_init_A(qhT*, FILE*, FILE*, FILE*, int, char**) {}
inline void qh_initflags(qhT* qh, char* flags) { if (flags && strstr(flags, " TA")) qh->unsupported = 1; }
inline void qh_init_B(qhT* qh, coordT* points, int numpoints, int dim, boolT) {
  if (dim != 3 || qh->unsupported) longjmp(qh->errexit, 1);
  qh->first_point = points; qh->num_points = numpoints;
}
inline int qh_pointid(qhT* qh, pointT* point) { return (int)((point - qh->first_point) / 3); }

namespace oracle_hull {
struct Tri { int a, b, c; bool alive; };
inline double orient(const double* P, int a, int b, int c, int d) {
  // signed volume: > 0 when d lies on the side the normal (b-a)x(c-a) points to
  const double* pa = P + 3*a; const double* pb = P + 3*b; const double* pc = P + 3*c; const double* pd = P + 3*d;
  double u[3] = {pb[0]-pa[0], pb[1]-pa[1], pb[2]-pa[2]};
  double v[3] = {pc[0]-pa[0], pc[1]-pa[1], pc[2]-pa[2]};
  double w[3] = {pd[0]-pa[0], pd[1]-pa[1], pd[2]-pa[2]};
  double n[3] = {u[1]*v[2]-u[2]*v[1], u[2]*v[0]-u[0]*v[2], u[0]*v[1]-u[1]*v[0]};
  double nn = std::sqrt(n[0]*n[0] + n[1]*n[1] + n[2]*n[2]);
  if (nn == 0) return 0;
  return (n[0]*w[0] + n[1]*w[1] + n[2]*w[2]) / nn;    // distance of d from the plane
}
// returns false when the points are degenerate (MakeGraph has already excluded that)
inline bool build(const double* P, int n, std::vector<int>& out) {
  if (n < 4) return false;
  double lo[3] = {P[0], P[1], P[2]}, hi[3] = {P[0], P[1], P[2]};
  for (int i = 1; i < n; i++) for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], P[3*i+k]); hi[k] = std::fmax(hi[k], P[3*i+k]); }
  const double scale = std::fmax(hi[0]-lo[0], std::fmax(hi[1]-lo[1], hi[2]-lo[2]));
  if (!(scale > 0)) return false;
  const double eps = 1e-10 * scale;
  // initial tetrahedron: point 0, the farthest point from it, the farthest from that line, the
  // farthest from that plane
  int i0 = 0, i1 = -1, i2 = -1, i3 = -1;
  double best = 0;
  for (int i = 1; i < n; i++) {
    double d = 0; for (int k = 0; k < 3; k++) d += (P[3*i+k]-P[3*i0+k])*(P[3*i+k]-P[3*i0+k]);
    if (d > best) { best = d; i1 = i; }
  }
  if (i1 < 0) return false;
  best = 0;
  for (int i = 0; i < n; i++) {
    if (i == i0 || i == i1) continue;
    double u[3], v[3];
    for (int k = 0; k < 3; k++) { u[k] = P[3*i1+k]-P[3*i0+k]; v[k] = P[3*i+k]-P[3*i0+k]; }
    double c[3] = {u[1]*v[2]-u[2]*v[1], u[2]*v[0]-u[0]*v[2], u[0]*v[1]-u[1]*v[0]};
    double d = c[0]*c[0] + c[1]*c[1] + c[2]*c[2];
    if (d > best) { best = d; i2 = i; }
  }
  if (i2 < 0) return false;
  best = 0;
  for (int i = 0; i < n; i++) {
    if (i == i0 || i == i1 || i == i2) continue;
    double d = std::fabs(orient(P, i0, i1, i2, i));
    if (d > best) { best = d; i3 = i; }
  }
  if (i3 < 0 || best <= eps) return false;
  if (orient(P, i0, i1, i2, i3) > 0) { int t = i1; i1 = i2; i2 = t; }   // i3 behind (i0,i1,i2)
  std::vector<Tri> T;
  T.push_back({i0, i1, i2, true}); T.push_back({i0, i3, i1, true});
  T.push_back({i1, i3, i2, true}); T.push_back({i2, i3, i0, true});
  std::vector<char> used(n, 0);
  used[i0] = used[i1] = used[i2] = used[i3] = 1;
  for (int p = 0; p < n; p++) {
    if (used[p]) continue;
    std::vector<int> vis;
    for (size_t f = 0; f < T.size(); f++) if (T[f].alive && orient(P, T[f].a, T[f].b, T[f].c, p) > eps) vis.push_back((int)f);
    if (vis.empty()) continue;                        // inside or on the surface
    // directed edges of the visible faces; a horizon edge is one whose reverse is not among them
    std::map<std::pair<int,int>, int> edges;
    for (int f : vis) { edges[{T[f].a, T[f].b}] = 1; edges[{T[f].b, T[f].c}] = 1; edges[{T[f].c, T[f].a}] = 1; T[f].alive = false; }
    for (const auto& kv : edges) {
      const int a = kv.first.first, b = kv.first.second;
      if (edges.find({b, a}) == edges.end()) T.push_back({a, b, p, true});
    }
    used[p] = 1;
  }
  out.clear();
  for (const Tri& t : T) if (t.alive) { out.push_back(t.a); out.push_back(t.b); out.push_back(t.c); }
  if (out.size() < 12) return false;
  // MakeGraph sizes its edge table as V + 3F, which holds exactly for a closed 2-manifold: refuse
  // anything else (tolerance trouble on near-degenerate inputs) instead of letting it overrun that table
  std::map<std::pair<int,int>, int> dir;
  for (size_t f = 0; f < out.size()/3; f++)
    for (int k = 0; k < 3; k++) dir[{out[3*f + k], out[3*f + (k + 1) % 3]}]++;
  for (const auto& kv : dir)
    if (kv.second != 1 || dir.find({kv.first.second, kv.first.first}) == dir.end()) return false;
  std::vector<char> isv(n, 0);
  int nvh = 0;
  for (int v : out) if (!isv[v]) { isv[v] = 1; nvh++; }
  return nvh - (int)dir.size()/2 + (int)out.size()/3 == 2;
}
inline setT* make_set(qhT* qh, const std::vector<void*>& items) {
  setT* s = (setT*)calloc(1, sizeof(setT) + sizeof(setelemT) * (items.size() + 1));
  s->maxsize = (int)items.size();
  for (size_t i = 0; i < items.size(); i++) s->e[i].p = items[i];
  s->e[items.size()].p = 0;
  qh->sets->push_back(s);
  return s;
}
}  // namespace oracle_hull

inline void qh_qhull(qhT* qh) {
  qh->tri = new std::vector<int>();
  qh->verts = new std::vector<vertexT>();
  qh->facets = new std::vector<facetT>();
  qh->sets = new std::vector<setT*>();
  if (!oracle_hull::build(qh->first_point, qh->num_points, *qh->tri)) longjmp(qh->errexit, 1);
}
inline void qh_triangulate(qhT*) {}
inline void qh_vertexneighbors(qhT* qh) {
  const std::vector<int>& tri = *qh->tri;
  const int nf = (int)tri.size() / 3;
  std::vector<int> hullid(qh->num_points, -1);
  int nv = 0;
  for (int p = 0; p < qh->num_points; p++) {
    bool on = false;
    for (int v : tri) if (v == p) { on = true; break; }
    if (on) hullid[p] = nv++;
  }
  // one sentinel element closes each list (FORALLvertices / FORALLfacets stop at ->next == 0)
  qh->verts->assign(nv + 1, vertexT{0, 0, 0});
  qh->facets->assign(nf + 1, facetT{0, 0, 0});
  std::vector<vertexT>& V = *qh->verts;
  std::vector<facetT>& F = *qh->facets;
  for (int p = 0; p < qh->num_points; p++) if (hullid[p] >= 0) V[hullid[p]].point = qh->first_point + 3*p;
  for (int i = 0; i < nv; i++) V[i].next = &V[i + 1];
  for (int f = 0; f < nf; f++) {
    F[f].next = &F[f + 1];
    F[f].toporient = 0;   // MakeGraph stores the vertices in set order: already counter-clockwise
    F[f].vertices = oracle_hull::make_set(qh, {&V[hullid[tri[3*f]]], &V[hullid[tri[3*f + 1]]], &V[hullid[tri[3*f + 2]]]});
  }
  for (int p = 0; p < qh->num_points; p++) {
    if (hullid[p] < 0) continue;
    std::vector<void*> nb;
    for (int f = 0; f < nf; f++) if (tri[3*f] == p || tri[3*f + 1] == p || tri[3*f + 2] == p) nb.push_back(&F[f]);
    V[hullid[p]].neighbors = oracle_hull::make_set(qh, nb);
  }
  qh->vertex_list = V.data();
  qh->facet_list = F.data();
  qh->num_vertices = nv;
  qh->num_facets = nf;
}
inline void qh_freeqhull(qhT* qh, boolT) {
  if (qh->sets) { for (setT* s : *qh->sets) free(s); delete qh->sets; qh->sets = 0; }
  delete qh->verts; qh->verts = 0;
  delete qh->facets; qh->facets = 0;
  delete qh->tri; qh->tri = 0;
  qh->vertex_list = 0; qh->facet_list = 0;
}
inline void qh_memfreeshort(qhT*, int* curlong, int* totlong) { *curlong = 0; *totlong = 0; }
#endif
