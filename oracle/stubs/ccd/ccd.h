// Test-infrastructure stub for <ccd/ccd.h> (libccd v2.1 is not on disk).
// ccdMPRPenetration reports "no penetration": the legacy libccd fallback is
// unreachable with default options (native CCD is used unless mjDSBL_NATIVECCD).
#ifndef ORACLE_STUB_CCD_CCD_H_
#define ORACLE_STUB_CCD_CCD_H_
#include <ccd/vec3.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void (*ccd_support_fn)(const void* obj, const ccd_vec3_t* dir, ccd_vec3_t* vec);
typedef void (*ccd_first_dir_fn)(const void* obj1, const void* obj2, ccd_vec3_t* dir);
typedef void (*ccd_center_fn)(const void* obj1, ccd_vec3_t* center);
typedef struct _ccd_t {
  ccd_first_dir_fn first_dir;
  ccd_support_fn support1, support2;
  ccd_center_fn center1, center2;
  unsigned long max_iterations;
  ccd_real_t epa_tolerance, mpr_tolerance, dist_tolerance;
} ccd_t;
static inline void ccdFirstDirDefault(const void* o1, const void* o2, ccd_vec3_t* dir) {
  (void)o1; (void)o2; ccdVec3Set(dir, 1, 0, 0);
}
#define CCD_INIT(c) do { (c)->first_dir = ccdFirstDirDefault; (c)->support1 = 0; (c)->support2 = 0; \
  (c)->center1 = 0; (c)->center2 = 0; (c)->max_iterations = (unsigned long)-1; \
  (c)->epa_tolerance = 1e-4; (c)->mpr_tolerance = 1e-4; (c)->dist_tolerance = 1e-6; } while (0)
static inline int ccdMPRPenetration(const void* o1, const void* o2, const ccd_t* c,
                                    ccd_real_t* depth, ccd_vec3_t* dir, ccd_vec3_t* pos) {
  (void)o1; (void)o2; (void)c; (void)depth; (void)dir; (void)pos;
  return -1;
}
#ifdef __cplusplus
}
#endif
#endif
