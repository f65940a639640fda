// Reader for MuJoCo's binary model format (.mjb), product side.
//
// Restates the layout written by mj_saveModel and read by mj_loadModelBuffer
// (/root/reference/src/engine/engine_io.c:514-700): a 5-int header {54321, sizeof(mjtNum), #sizes,
// mjVERSION, #pointers}, the mjModel size fields in MJMODEL_SIZES order, the mjOption / mjVisual /
// mjStatistic structs, two flag bytes, then every MJMODEL_POINTERS array back to back.  The field
// tables come from MuJoCo's public X-macro header, so the reader tracks whatever MuJoCo version
// the library is compiled against.  The in-memory struct reproduces MuJoCo's own buffer layout
// (64-byte aligned arrays, engine_io.c:142-165) so `nbuffer` doubles as an integrity check.
#pragma once

#include <mujoco/mujoco.h>
#include <mujoco/mjxmacro.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mjhmjb {

static inline size_t skip64(size_t off) { return (64 - (off % 64)) % 64; }

static void release(mjModel* m) {
  if (!m) return;
  free(m->buffer);
  free(m);
}

static mjModel* load(const char* path, std::string* err) {
  FILE* f = fopen(path, "rb");
  if (!f) { *err = std::string("mjhip_load_mjb: cannot open ") + path; return nullptr; }
  fseek(f, 0, SEEK_END);
  long fsz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> buf((size_t)fsz);
  if (fsz <= 0 || fread(buf.data(), 1, (size_t)fsz, f) != (size_t)fsz) {
    fclose(f);
    *err = std::string("mjhip_load_mjb: cannot read ") + path;
    return nullptr;
  }
  fclose(f);
  size_t pos = 0;
  auto rd = [&](void* dst, size_t n) -> bool {
    if (pos + n > buf.size()) return false;
    memcpy(dst, buf.data() + pos, n);
    pos += n;
    return true;
  };

  int nsize = 0, nptr = 0;
#define X(name) nsize++;
  MJMODEL_SIZES
#undef X
#define X(type, name, nr, nc) nptr++;
#define XNV X
  MJMODEL_POINTERS
#undef X
#undef XNV

  int header[5];
  if (!rd(header, sizeof(header))) { *err = "mjhip_load_mjb: truncated header"; return nullptr; }
  if (header[0] != 54321) { *err = "mjhip_load_mjb: not an MJB file"; return nullptr; }
  if (header[1] != (int)sizeof(mjtNum)) { *err = "mjhip_load_mjb: floating point precision mismatch"; return nullptr; }
  if (header[2] != nsize || header[4] != nptr || header[3] != mjVERSION_HEADER) {
    *err = "mjhip_load_mjb: file was written by a different MuJoCo version (header " +
           std::to_string(header[3]) + ", library " + std::to_string(mjVERSION_HEADER) + ")";
    return nullptr;
  }

  mjModel* m = (mjModel*)calloc(1, sizeof(mjModel));
  if (!m) { *err = "mjhip_load_mjb: out of memory"; return nullptr; }
  bool ok = true;
#define X(name) ok = ok && rd(&m->name, sizeof(m->name));
  MJMODEL_SIZES
#undef X
  ok = ok && rd(&m->opt, sizeof(mjOption)) && rd(&m->vis, sizeof(mjVisual)) && rd(&m->stat, sizeof(mjStatistic));
  ok = ok && rd(&m->flg_gravcomp, sizeof(mjtBool)) && rd(&m->flg_surfacevel, sizeof(mjtBool));
  if (!ok) { free(m); *err = "mjhip_load_mjb: truncated size/option block"; return nullptr; }

  // every size field is a count: a negative one is rejected before anything is sized from it (the
  // arrays are bounded by the file size below; narena / nbuffer are byte counts, not array lengths)
  {
    const char* bad = nullptr;
    // (njmax / nconmax are legacy limits where -1 means "unlimited"; they size nothing here)
#define X(name) if ((long long)m->name < 0 && !bad && strcmp(#name, "njmax") && strcmp(#name, "nconmax")) bad = #name;
    MJMODEL_SIZES
#undef X
    if (bad) { *err = std::string("mjhip_load_mjb: corrupted file (negative size field ") + bad + ")"; free(m); return nullptr; }
  }

  // buffer layout; no array can be larger than the file it is read from (this also bounds the
  // products: every factor is <= fsz < 2^63 / 8 for any real file, checked per term)
  size_t total = 0;
  {
    bool bad = false;
    auto term = [&](size_t esz, long long nr, long long nc) -> size_t {
      if (nr < 0 || nc < 0) { bad = true; return 0; }
      if (nr == 0 || nc == 0) return 0;
      if ((unsigned long long)nr > (unsigned long long)fsz / (unsigned long long)nc ||
          (unsigned long long)(nr*nc) > (unsigned long long)fsz / esz) { bad = true; return 0; }
      return esz*(size_t)nr*(size_t)nc;
    };
    MJMODEL_POINTERS_PREAMBLE(m)
#define X(type, name, nr, nc) total += skip64(total) + term(sizeof(type), (long long)(m->nr), (long long)(nc));
#define XNV X
    MJMODEL_POINTERS
#undef X
#undef XNV
    if (bad) { free(m); *err = "mjhip_load_mjb: corrupted file (array larger than the file)"; return nullptr; }
  }
  if ((mjtSize)total != m->nbuffer) {
    free(m);
    *err = "mjhip_load_mjb: corrupted file (nbuffer mismatch)";
    return nullptr;
  }
  void* base = nullptr;
  if (posix_memalign(&base, 64, total ? total : 64)) { free(m); *err = "mjhip_load_mjb: out of memory"; return nullptr; }
  memset(base, 0, total);
  m->buffer = base;
  {
    size_t off = 0;
    MJMODEL_POINTERS_PREAMBLE(m)
#define X(type, name, nr, nc)                                              \
    off += skip64(off);                                                    \
    m->name = (type*)((char*)base + off);                                  \
    { size_t nb = sizeof(type)*(size_t)(m->nr)*(size_t)(nc);               \
      ok = ok && rd((void*)m->name, nb);                                   \
      off += nb; }
#define XNV X
    MJMODEL_POINTERS
#undef X
#undef XNV
  }
  if (!ok || pos != buf.size()) {
    release(m);
    *err = "mjhip_load_mjb: file size does not match the model sizes";
    return nullptr;
  }
  // flg_adhesion is not stored in the file; recompute it like the compiler does
  m->flg_adhesion = 0;
  for (int i = 0; i < m->ngeom; i++) if (m->geom_adhesion[i] != 0) m->flg_adhesion = 1;
  for (int i = 0; i < m->npair; i++) if (m->pair_adhesion[i] != 0) m->flg_adhesion = 1;
  return m;
}

}  // namespace mjhmjb
