// Test-infrastructure stub for qhull's reentrant API header "qhull_ra.h" (qhull is
// not on disk). Only the names user_mesh.cc (mjCMesh::MakeGraph) uses are declared.
// qh_init_B longjmps to the caller's error handler, i.e. models with mesh geoms
// that need a convex hull report "qhull error"; the round-1 oracle models
// (humanoid, slider_crank) have no meshes.
#ifndef ORACLE_STUB_QHULL_RA_H_
#define ORACLE_STUB_QHULL_RA_H_
#include <csetjmp>
#include <cstdio>
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
};
#define FORALLvertices for (vertex = qh->vertex_list; vertex && vertex->next; vertex = vertex->next)
#define FORALLfacets for (facet = qh->facet_list; facet && facet->next; facet = facet->next)
#define FOREACHsetelement_(type, set, variable) \
  if (((variable = NULL), set)) \
    for (variable##p = (type**)&((set)->e[0].p); (variable = *variable##p++);)
inline void qh_zero(qhT* qh, FILE*) { qh->NOerrexit = 1; qh->num_vertices = qh->num_facets = 0; qh->vertex_list = 0; qh->facet_list = 0; }
inline void qh_init_A(qhT*, FILE*, FILE*, FILE*, int, char**) {}
inline void qh_initflags(qhT*, char*) {}
inline void qh_init_B(qhT* qh, coordT*, int, int, boolT) { longjmp(qh->errexit, 1); }
inline void qh_qhull(qhT*) {}
inline void qh_triangulate(qhT*) {}
inline void qh_vertexneighbors(qhT*) {}
inline int qh_pointid(qhT*, pointT*) { return -1; }
inline void qh_freeqhull(qhT*, boolT) {}
inline void qh_memfreeshort(qhT*, int* curlong, int* totlong) { *curlong = 0; *totlong = 0; }
#endif
