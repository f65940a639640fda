// Oracle-side introspection helper (TEST INFRASTRUCTURE).
//
// Compiled against the reference headers in place (-I/root/reference/include) so the
// Python test harness can reach mjModel / mjData fields by NAME without a hand-copied
// struct layout: the tables below are expanded from the reference's own X-macros
// (include/mujoco/mjxmacro.h:162 MJMODEL_SIZES, :740 MJMODEL_POINTERS, :842
// MJDATA_POINTERS, :1023 MJDATA_ARENA_POINTERS, :1032 MJDATA_SCALAR, :1070
// MJDATA_VECTOR, :23 MJOPTION_FIELDS).
//
// type codes: 'd' mjtNum(double), 'i' int, 'f' float, 'b' mjtByte/mjtBool/char,
//             'q' int64 (mjtSize/size_t/uintptr_t), 'C' mjContact, 'S' other struct.

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <mujoco/mjdata.h>
#include <mujoco/mjmodel.h>
#include <mujoco/mjxmacro.h>

#define TYPECODE(T) _Generic((T*)0,                                             \
    double*: 'd', float*: 'f', int*: 'i', unsigned char*: 'b', _Bool*: 'b', char*: 'b', \
    int64_t*: 'q', uint64_t*: 'q', mjContact*: 'C', default: 'S')

// ---- mjModel sizes ------------------------------------------------------------------
long long mjo_model_size(const mjModel* m, const char* name) {
#define X(n) if (!strcmp(name, #n)) return (long long)m->n;
  MJMODEL_SIZES
#undef X
  return -1;
}

// ---- mjModel pointer fields ---------------------------------------------------------
// returns 1 if found; *ptr = field pointer, *nr/*nc = rows/cols, *type = code, *itemsize
int mjo_model_field(const mjModel* m, const char* name, void** ptr, long long* nr,
                    long long* nc, char* type, int* itemsize) {
  MJMODEL_POINTERS_PREAMBLE(m)
#define X(T, n, r, c)                                                            \
  if (!strcmp(name, #n)) {                                                       \
    *ptr = (void*)m->n; *nr = (long long)(m->r); *nc = (long long)(c);           \
    *type = TYPECODE(T); *itemsize = (int)sizeof(T); return 1;                   \
  }
#undef MJ_M
#define MJ_M(n) n
  MJMODEL_POINTERS
#undef X
  return 0;
}

// list model pointer field names, '\n'-separated, into buf
int mjo_model_field_names(char* buf, int nbuf) {
  int pos = 0;
  buf[0] = 0;
#define X(T, n, r, c)                                                            \
  { int l = (int)strlen(#n); if (pos + l + 2 < nbuf) { memcpy(buf + pos, #n, l); pos += l; buf[pos++] = '\n'; buf[pos] = 0; } }
  MJMODEL_POINTERS
#undef X
  return pos;
}

// ---- mjOption / mjStatistic ---------------------------------------------------------
int mjo_option_field(const mjModel* m, const char* name, void** ptr, int* n, char* type) {
#define X(T, nm, cnt) if (!strcmp(name, #nm)) { *ptr = (void*)&m->opt.nm; *n = 1; *type = TYPECODE(T); return 1; }
#define XVEC(T, nm, cnt) if (!strcmp(name, #nm)) { *ptr = (void*)m->opt.nm; *n = (int)(cnt); *type = TYPECODE(T); return 1; }
  MJOPTION_FIELDS
#undef X
#undef XVEC
  return 0;
}

int mjo_stat_field(const mjModel* m, const char* name, void** ptr, int* n) {
#define X(nm, cnt) if (!strcmp(name, #nm)) { *ptr = (void*)&m->stat.nm; *n = 1; return 1; }
#define XVEC(nm, cnt) if (!strcmp(name, #nm)) { *ptr = (void*)m->stat.nm; *n = (int)(cnt); return 1; }
  MJSTATISTIC_FIELDS
#undef X
#undef XVEC
  return 0;
}

// ---- mjData -------------------------------------------------------------------------
int mjo_data_field(const mjModel* m, const mjData* d, const char* name, void** ptr,
                   long long* nr, long long* nc, char* type, int* itemsize) {
  MJMODEL_POINTERS_PREAMBLE(m)
  (void)nuser_body; (void)nuser_jnt; (void)nuser_geom; (void)nuser_site; (void)nuser_cam;
  (void)nuser_tendon; (void)nuser_actuator; (void)nuser_sensor; (void)nq; (void)nv; (void)na; (void)nu; (void)nmocap;
#define X(T, n, r, c)                                                            \
  if (!strcmp(name, #n)) {                                                       \
    *ptr = (void*)d->n; *nr = (long long)(m->r); *nc = (long long)(c);           \
    *type = TYPECODE(T); *itemsize = (int)sizeof(T); return 1;                   \
  }
  MJDATA_POINTERS
#undef X
#undef MJ_M
#undef MJ_D
#define MJ_M(n) m->n
#define MJ_D(n) d->n
#define X(T, n, r, c)                                                            \
  if (!strcmp(name, #n)) {                                                       \
    *ptr = (void*)d->n; *nr = (long long)(r); *nc = (long long)(c);              \
    *type = TYPECODE(T); *itemsize = (int)sizeof(T); return 1;                   \
  }
  MJDATA_ARENA_POINTERS
#undef X
#define X(T, n, r, c)                                                            \
  if (!strcmp(name, #n)) {                                                       \
    *ptr = (void*)d->n; *nr = (long long)(r); *nc = (long long)(c);              \
    *type = TYPECODE(T); *itemsize = (int)sizeof(T); return 1;                   \
  }
  MJDATA_VECTOR
#undef X
  return 0;
}

int mjo_data_scalar(const mjData* d, const char* name, void** ptr, char* type) {
#define X(T, n) if (!strcmp(name, #n)) { *ptr = (void*)&d->n; *type = TYPECODE(T); return 1; }
  MJDATA_SCALAR
#undef X
  return 0;
}

int mjo_data_field_names(char* buf, int nbuf) {
  int pos = 0;
  buf[0] = 0;
#define X(T, n, r, c)                                                            \
  { int l = (int)strlen(#n); if (pos + l + 2 < nbuf) { memcpy(buf + pos, #n, l); pos += l; buf[pos++] = '\n'; buf[pos] = 0; } }
  MJDATA_POINTERS
  MJDATA_ARENA_POINTERS
#undef X
  return pos;
}

// ---- struct sizes / offsets the harness needs ---------------------------------------
int mjo_sizeof(const char* what) {
  if (!strcmp(what, "mjContact")) return (int)sizeof(mjContact);
  if (!strcmp(what, "mjData")) return (int)sizeof(mjData);
  if (!strcmp(what, "mjModel")) return (int)sizeof(mjModel);
  if (!strcmp(what, "mjWarningStat")) return (int)sizeof(mjWarningStat);
  if (!strcmp(what, "mjSolverStat")) return (int)sizeof(mjSolverStat);
  if (!strcmp(what, "mjOption")) return (int)sizeof(mjOption);
  return -1;
}

int mjo_contact_offset(const char* field) {
#define F(n) if (!strcmp(field, #n)) return (int)offsetof(mjContact, n);
  F(dist) F(pos) F(frame) F(includemargin) F(friction) F(solref) F(solreffriction) F(solimp)
  F(mu) F(H) F(dim) F(geom1) F(geom2) F(geom) F(flex) F(elem) F(vert) F(exclude) F(efc_address)
#undef F
  return -1;
}

int mjo_warning_number(const mjData* d, int i) { return d->warning[i].number; }
