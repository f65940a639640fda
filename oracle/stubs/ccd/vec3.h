// Test-infrastructure stub for <ccd/vec3.h> (libccd v2.1 is not on disk).
// Only the types/macros the reference's engine_collision_convex.{c,h} touch are
// declared; the libccd (MPR) code path is dead unless mjDSBL_NATIVECCD is set.
#ifndef ORACLE_STUB_CCD_VEC3_H_
#define ORACLE_STUB_CCD_VEC3_H_
#ifdef __cplusplus
extern "C" {
#endif
typedef double ccd_real_t;
typedef struct _ccd_vec3_t { ccd_real_t v[3]; } ccd_vec3_t;
static const ccd_vec3_t ccd_vec3_origin_storage = {{0, 0, 0}};
#define ccd_vec3_origin (&ccd_vec3_origin_storage)
static inline void ccdVec3Set(ccd_vec3_t* v, ccd_real_t x, ccd_real_t y, ccd_real_t z) {
  v->v[0] = x; v->v[1] = y; v->v[2] = z;
}
static inline int ccdVec3Eq(const ccd_vec3_t* a, const ccd_vec3_t* b) {
  return a->v[0] == b->v[0] && a->v[1] == b->v[1] && a->v[2] == b->v[2];
}
#ifdef __cplusplus
}
#endif
#endif
