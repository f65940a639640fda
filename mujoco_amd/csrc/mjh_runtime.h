// Host runtime shared by the two builds of the C ABI (include/mjhip.h):
//   product   : mjh_hip.hip     -> Backend = HIP runtime + __global__ kernels (libmjhip.so)
//   test-only : tests/hostsim   -> Backend = malloc + the wavefront emulation of mjh_spmd.h
// Everything here is plain C++: model upload, batch allocation (one SoA-across-envs arena per
// batch), field registry, rollout argument marshalling, the mjhip_rollout drop-in.
//
// The including translation unit must define, before including this file, a struct `Backend` with:
//   static const char* name();
//   static int  device_count();
//   static bool set_device(int dev, std::string* err);
//   static void* alloc(size_t bytes);            // device memory, nullptr on failure
//   static void free(void* p);
//   static bool h2d(void* dst, const void* src, size_t bytes, void* stream);
//   static bool d2h(void* dst, const void* src, size_t bytes, void* stream);
//   static bool zero(void* dst, size_t bytes, void* stream);
//   static bool sync(void* stream);
//   static int  current_device();
//   static bool launch_balance(const DBatch* B, int nenv, int nstep, void* stream);   // launch order of the next rollout launch
//   (M, B below are DEVICE pointers to the descriptor structs)
//   (lds = bytes of LDS per one-wavefront workgroup demanded by the batch descriptor's plan, 0 = none)
//   (variant = MJH_VAR_* of mjh_modes.h: the kernel mapping that steps the batch; lds is per environment)
//   static bool launch_forward(const DModel* M, const DBatch* B, int nenv, int stages, int lds, int soa, int variant, void* stream);
//   static bool launch_rollout(const DModel* M, const DBatch* B, int nenv, const RolloutArgs&, int lds, int variant, void* stream);
//   static const char* rollout_kernel_name(int variant, int nenv);   // name of the kernel launch_rollout launches
//   static bool launch_reset(const DModel* M, const DBatch* B, int nenv, void* stream);
//   lane-mode kernels of the SoA pipeline (epw = environments per wavefront):
//   static bool launch_smooth(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs&, void* stream);
//   static bool launch_integrate(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs&, void* stream);
//   static bool launch_lane_forward(const DModel* M, const DBatch* B, int nenv, int epw, int stages, void* stream);
//   static bool launch_lane_reset(const DModel* M, const DBatch* B, int nenv, int epw, void* stream);
//   static int  max_lds();                       // largest LDS block one workgroup may ask for
//   static int  num_cus();                       // compute units of the current device
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mjhip.h"
#include "mjh_model_build.h"
#include "mjh_mjb.h"
#include "mjh_modes.h"

#include <algorithm>
#include <atomic>

#ifndef MJHIP_DEFAULT_LAYOUT
#define MJHIP_DEFAULT_LAYOUT MJHIP_LAYOUT_AOS
#endif
#ifndef MJHIP_DEFAULT_LDS_BYTES
#define MJHIP_DEFAULT_LDS_BYTES 10240   // 16 one-wavefront workgroups per CU (160 KB LDS): 4096 environments resident at once
#endif

static thread_local std::string g_mjhip_err;
static void set_err(const std::string& e) { g_mjhip_err = e; }

struct mjhipModel_ {
  HostModel H;
  DModel D;                 // device pointers (host copy of the struct)
  DModel* D_dev = nullptr;  // the same struct in device memory: what the kernels receive
  std::vector<void*> allocs;
  int device = 0;
  std::vector<char> signature;   // bytes identifying the source mjModel (for the rollout cache)
};

struct FieldInfo { void* ptr; int count; int is_int; };

struct mjhipStage_;
struct mjhipBatch_ {
  mjhipModel_* model = nullptr;
  int device = 0;              // the GPU that holds the arena; every entry point selects it
  bool xfrc_on = false;        // xfrc_applied may be non-zero (mj_xfrcAccumulate runs)
  mjhipStage_* stage = nullptr;   // pooled device staging of host-pointer rollouts
  DBatch D;                    // all-global descriptor (l_* = -1): inspection / debug kernels
  DBatch* D_dev = nullptr;
  DBatch L;                    // descriptor with the LDS residency plan (rollout / step kernels)
  DBatch* L_dev = nullptr;
  std::string plan_report;     // human-readable plan (mjhip_batch_lds_report)
  int soa = 0;                 // 0: fields [nenv][count]; else nenvpad: fields [count][nenvpad]
  int nenvpad = 0;
  int epw = 64;                // environments per wavefront of the lane-mode kernels (<= 64)
  int variant = MJH_VAR_GENERIC;   // kernel mapping (MJH_VAR_*, mjh_modes.h) that steps this batch
  int lds_request = 0;             // LDS budget per environment last asked for (before clamping)
  bool balance = true;             // order the next rollout launch by the work estimate of the last
  void* arena = nullptr;
  size_t arena_bytes = 0;
  void* ccd_ws = nullptr;          // GJK / EPA workspace (models with convex pairs)
  std::map<std::string, FieldInfo> fields;
  int nenv = 0;
};

// `slot` is the address of a DModel table pointer (a constant-address-space pointer in the device
// build); the device address is stored into it bytewise
template <class T>
static bool upload_vec(mjhipModel_* M, const std::vector<T>& v, void* slot, std::string* err) {
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  void* p = Backend::alloc(bytes);
  if (!p) { *err = "mjhip: device allocation failed (model)"; return false; }
  M->allocs.push_back(p);
  if (!v.empty() && !Backend::h2d(p, v.data(), v.size()*sizeof(T), nullptr)) {
    *err = "mjhip: host->device copy failed (model)";
    return false;
  }
  memcpy(slot, &p, sizeof p);
  return true;
}


// ---- LDS residency plan ---------------------------------------------------------------------------
// Interval overlay of the per-field lifetimes of mjh_types.h inside a block of `budget` bytes.
// Fields alive during constraint assembly/solve are packed first from offset 0; the bytes above
// them form the dynamic region handed to efc_layout().  Fields that die before MJH_T_MAKE may sit
// anywhere, including on top of that region.  A field that does not fit stays global.
struct PlanField { const char* name; int* l; int* io; int bytes; int t0, t1; int off; int born = 0; };

// span_first..span_last: the timeline points the kernel using this plan executes.  Only fields
// whose lifetime intersects the span are placed.  `skip`: fields alive across the span that the
// kernel never touches (left in HBM); `readonly`: persistent fields the kernel reads but does not
// modify (loaded at entry, not stored at exit).
static bool plan_lds(mjhipBatch_* Bt, int budget, unsigned executed,
                     const std::vector<std::string>& skip, const std::vector<std::string>& readonly,
                     std::string* report) {
  // executed: bit t set = the kernel using this plan runs timeline point t
  auto touches = [&](int t0, int t1) { for (int t = t0; t <= t1; t++) if ((executed >> t) & 1) return true; return false; };
  int span_first = 0, span_last = 0;
  for (int t = 0; t < 32; t++) if ((executed >> t) & 1) { if (!span_first) span_first = t; span_last = t; }
  DBatch& L = Bt->L;
  L = Bt->D;
  const DSizes& s = Bt->model->H.s;
  std::vector<PlanField> f;
  auto listed = [](const std::vector<std::string>& v, const char* n) {
    return std::find(v.begin(), v.end(), std::string(n)) != v.end();
  };
  // (models on the explicit-index constraint path -- mjh_csr.h, hundreds to thousands of dofs: their dof / body sized
  // fields are streamed lane-parallel and gain nothing from LDS, while the solver's ordered sums need the block as
  // staging space: fields above 2 KB stay in global memory)
  // (with most of a CU's block to itself -- launches of one workgroup per CU -- a big field whose lifetime ends before
  // constraint assembly, or starts after the solve, may overlay the solver's region: the per-body arrays of the smooth
  // stages are then read at LDS instead of global-memory latency, which a lone wavefront cannot hide)
  const size_t big = s.csr ? 2048 : (size_t)1 << 30;
  const bool big_overlay = s.csr && budget >= 128*1024 && !getenv("MJHIP_NO_BIG_OVERLAY");
  auto fits = [&](size_t bytes, int t0, int t1) {
    if (bytes <= big) return true;
    return big_overlay && (t1 < MJH_T_MAKE || t0 > MJH_T_CONSTRAINT) && t0 != MJH_T_BEGIN;
  };
#define X(name, cnt, lcnt, t0, t1) if ((t0) != MJH_T_GLB && (int)(lcnt) > 0 && fits((size_t)(lcnt)*sizeof(real), (t0), (t1)) && touches((t0), (t1)) && !listed(skip, #name)) \
    f.push_back(PlanField{#name, &L.l_##name, &L.io_##name, (int)(((size_t)(lcnt)*sizeof(real) + 7) & ~(size_t)7), (t0), (t1), -1});
  MJH_BATCH_REAL_FIELDS(X)
#undef X
#define X(name, cnt, lcnt, t0, t1) if ((t0) != MJH_T_GLB && (int)(lcnt) > 0 && fits((size_t)(lcnt)*sizeof(int), (t0), (t1)) && touches((t0), (t1)) && !listed(skip, #name)) \
    f.push_back(PlanField{#name, &L.l_##name, &L.io_##name, (int)(((size_t)(lcnt)*sizeof(int) + 7) & ~(size_t)7), (t0), (t1), -1});
  MJH_BATCH_INT_FIELDS(X)
#undef X
  // the convex narrowphase's row workspaces: a one-wavefront mapping uses rows 0..3 only (row = lane >> 4); the model sizes
  // the field for a multi-wavefront workgroup's 4 MJH_MW rows (flex models) -- eight times what the other mappings can
  // touch, taken ahead of every other field
  if (Bt->variant != MJH_VAR_MULTIWAVE)
    for (auto& x : f)
      if (!strcmp(x.name, "ccd_row")) x.bytes = std::min(x.bytes, (int)((4*(size_t)s.ccd_row_reals*sizeof(real) + 7) & ~(size_t)7));
  budget &= ~7;
  // a field this kernel does not produce is copied in at kernel entry: it occupies its bytes from the
  // kernel's first stage on, whatever its nominal first write is
  for (auto& x : f) { x.born = (executed >> x.t0) & 1; if (!x.born) x.t0 = std::min(x.t0, span_first); }
  auto live_in = [](const PlanField& a, int t0, int t1) { return a.t0 <= t1 && t0 <= a.t1; };
  auto place = [&](PlanField& x, int limit) -> bool {
    // candidate offsets: 0 and the end of every placed field x conflicts with
    std::vector<int> cand{0};
    for (auto& y : f) if (y.off >= 0 && live_in(y, x.t0, x.t1)) cand.push_back(y.off + y.bytes);
    std::sort(cand.begin(), cand.end());
    for (int c : cand) {
      if (c + x.bytes > limit) break;
      bool ok = true;
      for (auto& y : f) if (y.off >= 0 && live_in(y, x.t0, x.t1) && c < y.off + y.bytes && y.off < c + x.bytes) { ok = false; break; }
      if (ok) { x.off = c; return true; }
    }
    return false;
  };
  std::vector<int> order(f.size());
  for (size_t i = 0; i < f.size(); i++) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    int la = f[a].t1 - f[a].t0, lb = f[b].t1 - f[b].t0;
    if (la != lb) return la > lb;            // long-lived first
    return f[a].bytes > f[b].bytes;
  });
  bool ok = budget > 0;
  int dyn_off = 0, dyn2_off = 0;
  // phase A1: alive from MJH_T_PROJECT to MJH_T_CONSTRAINT (packed lowest)
  for (int i : order) {
    PlanField& x = f[i];
    if (!live_in(x, MJH_T_PROJECT, MJH_T_CONSTRAINT)) continue;
    bool persistent = (x.t0 == MJH_T_BEGIN && x.t1 == MJH_T_END);
    if (ok && place(x, budget)) dyn2_off = std::max(dyn2_off, x.off + x.bytes);
    else if (persistent) ok = false;         // the state itself must fit, else no plan at all
  }
  dyn_off = dyn2_off;
  // phase A2: alive at MJH_T_MAKE but dead afterwards (cdof, subtree_com, contact slots, tendon
  // rows): they sit between the two dynamic regions, the lower of which reuses their bytes
  for (int i : order) {
    PlanField& x = f[i];
    if (live_in(x, MJH_T_PROJECT, MJH_T_CONSTRAINT) || !live_in(x, MJH_T_MAKE, MJH_T_MAKE)) continue;
    if (ok && place(x, budget)) dyn_off = std::max(dyn_off, x.off + x.bytes);
  }
  // phase B: everything else, free to overlay the dynamic region (it is dead by MJH_T_MAKE or
  // born after MJH_T_CONSTRAINT)
  // (largest byte x lifetime area first: the short-lived per-body arrays of the smooth stages are the
  // big ones, and first-fit by lifetime alone left holes they did not fit into)
  std::vector<int> orderB(order);
  std::stable_sort(orderB.begin(), orderB.end(), [&](int a, int b) {
    // (the convex narrowphase's row workspaces first: every access of its inner loops lands there, while the tree
    // fields they would displace are written once and read once or twice)
    const bool ha = !strcmp(f[a].name, "ccd_row"), hb = !strcmp(f[b].name, "ccd_row");
    if (ha != hb) return ha;
    return (long long)f[a].bytes*(f[a].t1 - f[a].t0 + 1) > (long long)f[b].bytes*(f[b].t1 - f[b].t0 + 1);
  });
  for (int i : orderB) {
    PlanField& x = f[i];
    if (live_in(x, MJH_T_MAKE, MJH_T_CONSTRAINT)) continue;
    if (ok) place(x, budget);
  }
  char line[256];
  if (!ok) {
    L = Bt->D;
    if (report) *report = "no LDS plan (budget " + std::to_string(budget) + " B): all fields global\n";
    return false;
  }
  for (auto& x : f) {
    *x.l = x.off;
    if (x.off < 0) continue;
    const bool persistent = (x.t0 == MJH_T_BEGIN);
    const bool born_here = x.born != 0;                           // this kernel runs the stage that writes it
    const bool live_in = !born_here;                              // produced by another kernel / launch
    const bool live_out = x.t1 > span_last;                       // consumed after it
    int io = 0;
    if (live_in) io |= 1;
    if (live_out && (born_here || (persistent && !listed(readonly, x.name)))) io |= 2;
    *x.io = io;
  }
  L.lds_bytes = budget;
  L.dyn_off = dyn_off;
  L.dyn2_off = dyn2_off;
  L.nconlds = s.nconlds;
  if (report) {
    report->clear();
    snprintf(line, sizeof line, "LDS plan: %d B per workgroup, static [0,%d), dynamic constraint regions [%d,%d) from make + [%d,%d) from project\n",
             budget, dyn_off, dyn_off, budget, dyn2_off, dyn_off);
    *report += line;
    for (auto& x : f) {
      snprintf(line, sizeof line, "  %-18s %6d B  t[%2d,%2d]  %s%d%s%s\n", x.name, x.bytes, x.t0, x.t1,
               x.off >= 0 ? "lds@" : "global ", x.off, (x.off >= 0 && (*x.io & 1)) ? " in" : "",
               (x.off >= 0 && (*x.io & 2)) ? " out" : "");
      *report += line;
    }
  }
  return true;
}

extern "C" {

MJHIP_API const char* mjhip_backend(void) { return Backend::name(); }
MJHIP_API const char* mjhip_last_error(void) { return g_mjhip_err.c_str(); }
MJHIP_API int mjhip_device_count(void) { return Backend::device_count(); }

MJHIP_API mjhipModel* mjhip_model_create(const struct mjModel_* m, int nconmax, int nefcmax) {
  if (!m) { set_err("mjhip_model_create: null mjModel"); return nullptr; }
  if (Backend::device_count() <= 0) {
    set_err("mjhip: no HIP device visible -- libmjhip has no CPU fallback");
    return nullptr;
  }
  mjhipModel_* M = new mjhipModel_();
  M->device = Backend::current_device();
  std::string err;
  mjhb::BuildCaps caps;
  // capacities: explicit arguments, else $MJHIP_NCONMAX / $MJHIP_NEFCMAX, else chosen from the model
  if (nconmax <= 0) if (const char* ev = getenv("MJHIP_NCONMAX")) nconmax = atoi(ev);
  if (nefcmax <= 0) if (const char* ev = getenv("MJHIP_NEFCMAX")) nefcmax = atoi(ev);
  caps.nconmax = nconmax; caps.nefcmax = nefcmax;
  if (const char* ev = getenv("MJHIP_EFC_BYTES")) caps.efc_bytes = atoll(ev);
  if (!mjhb::build((const mjModel*)m, caps, &M->H, &err) || !mjhb::check_sizes(M->H, &err)) {
    set_err(err);
    delete M;
    return nullptr;
  }
  // ($MJHIP_BROADPHASE=0: skip the reproduction of the reference's broad / midphase culls -- every
  // static pair goes to the bounding-sphere filter and the narrowphase; conservative, but geoms that
  // touch to within rounding can then differ from the reference in contact count.  For A/B timing.)
  if (const char* ev = getenv("MJHIP_BROADPHASE")) if (atoi(ev) == 0) M->H.s.nbp = 0;
  if (const char* ev = getenv("MJHIP_PGS_WIDE")) if (atoi(ev) == 0) M->H.s.pgs_nmax = 64;   // A/B: 64 < nefc <= 128 on the memory-based sweep
  M->D.s = M->H.s;
  M->D.o = M->H.o;
  bool ok = true;
#define X(name, cnt) ok = ok && upload_vec<int>(M, M->H.name, (void*)&M->D.name, &err);
  MJH_MODEL_INT_FIELDS(X)
#undef X
#define X(name, cnt) ok = ok && upload_vec<real>(M, M->H.name, (void*)&M->D.name, &err);
  MJH_MODEL_REAL_FIELDS(X)
#undef X
  if (ok) {
    M->D_dev = (DModel*)Backend::alloc(sizeof(DModel));
    ok = M->D_dev && Backend::h2d(M->D_dev, &M->D, sizeof(DModel), nullptr);
    if (M->D_dev) M->allocs.push_back(M->D_dev);
  }
  if (!ok || !Backend::sync(nullptr)) {
    set_err(err.empty() ? "mjhip: model upload failed" : err);
    mjhip_model_destroy(M);
    return nullptr;
  }
  // signature for the rollout cache: sizes + option block + the constant buffer contents
  const mjModel* mm = (const mjModel*)m;
  M->signature.assign((const char*)mm->buffer, (const char*)mm->buffer + mm->nbuffer);
  M->signature.insert(M->signature.end(), (const char*)&mm->opt, (const char*)&mm->opt + sizeof(mm->opt));
  return M;
}

MJHIP_API void mjhip_model_destroy(mjhipModel* M) {
  if (!M) return;
  for (void* p : M->allocs) Backend::free(p);
  delete M;
}

MJHIP_API int mjhip_model_size(const mjhipModel* M, const char* name) {
  if (!M || !name) return -1;
  const DSizes& s = M->H.s;
#define SZ(n) if (!strcmp(name, #n)) return s.n;
  SZ(nq) SZ(nv) SZ(nu) SZ(na) SZ(nbody) SZ(njnt) SZ(ngeom) SZ(nsite) SZ(ntendon) SZ(npair)
  SZ(features) SZ(nsensor) SZ(nsensordata) SZ(nconmax) SZ(nefcmax) SZ(nstate) SZ(nC) SZ(nJten) SZ(ntree) SZ(nlevel) SZ(nmoment) SZ(ccd_any) SZ(ccd_env_bytes) SZ(ccd_row_reals) SZ(nmesh) SZ(sparse) SZ(nJmax) SZ(nLp) SZ(nflex) SZ(nflexvert) SZ(nflexedge) SZ(nflexelem) SZ(csr) SZ(neqrow) SZ(ndoffric) SZ(njntlim) SZ(xn) SZ(xncap) SZ(nfv) SZ(efm) SZ(ne0) SZ(ne0L) SZ(ne0lev1) SZ(ne0lev2)
#undef SZ
  set_err(std::string("mjhip_model_size: unknown size ") + name);
  return -1;
}

MJHIP_API struct mjModel_* mjhip_load_mjb(const char* path) {
  std::string err;
  mjModel* m = mjhmjb::load(path, &err);
  if (!m) set_err(err);
  return (struct mjModel_*)m;
}
MJHIP_API void mjhip_free_mjb(struct mjModel_* m) { mjhmjb::release((mjModel*)m); }

MJHIP_API int mjhip_set_option(struct mjModel_* mm, const char* name, double value) {
  mjModel* m = (mjModel*)mm;
  if (!m || !name) return -1;
#define OPTD(n) if (!strcmp(name, #n)) { m->opt.n = value; return 0; }
#define OPTI(n) if (!strcmp(name, #n)) { m->opt.n = (int)value; return 0; }
  OPTD(timestep) OPTD(impratio) OPTD(tolerance) OPTD(ls_tolerance) OPTD(noslip_tolerance)
  OPTI(integrator) OPTI(cone) OPTI(jacobian) OPTI(solver) OPTI(iterations) OPTI(ls_iterations)
  OPTI(noslip_iterations) OPTI(disableflags) OPTI(enableflags)
#undef OPTD
#undef OPTI
  set_err(std::string("mjhip_set_option: unknown option ") + name);
  return -2;
}

static int default_variant(const mjhipModel_* M, int soa, int nenv);

MJHIP_API mjhipBatch* mjhip_batch_create(mjhipModel* M, int nenv, int device) {
  // $MJHIP_LAYOUT = aos | soa overrides the default layout
  int layout = MJHIP_DEFAULT_LAYOUT;
  if (const char* ev = getenv("MJHIP_LAYOUT")) layout = (!strcmp(ev, "aos") || !strcmp(ev, "0")) ? MJHIP_LAYOUT_AOS : MJHIP_LAYOUT_SOA;
  return mjhip_batch_create_layout(M, nenv, device, layout);
}

MJHIP_API mjhipBatch* mjhip_batch_create_layout(mjhipModel* M, int nenv, int device, int layout) {
  if (!M || nenv <= 0) { set_err("mjhip_batch_create: bad arguments"); return nullptr; }
  std::string err;
  if (!Backend::set_device(device, &err)) { set_err(err); return nullptr; }
  if (M->device != device) {
    set_err("mjhip_batch_create: the model was uploaded to device " + std::to_string(M->device) + ", the batch is asked for device " +
            std::to_string(device) + " (select the device before mjhip_model_create)");
    return nullptr;
  }
  mjhipBatch_* Bt = new mjhipBatch_();
  Bt->model = M;
  Bt->device = device;
  Bt->nenv = nenv;
  Bt->nenvpad = (nenv + 63) & ~63;
  Bt->soa = (layout == MJHIP_LAYOUT_SOA) ? Bt->nenvpad : 0;
  if (const char* ev = getenv("MJHIP_EPW")) { int v = atoi(ev); if (v >= 1 && v <= 64) Bt->epw = v; }
  const size_t nalloc = (size_t)Bt->nenvpad;
  const DSizes& s = M->H.s;
  // one arena; every field aligned to 256 bytes
  size_t off = 0;
  auto place = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  std::vector<std::pair<size_t, size_t>> spans;
#define X(name, cnt, lcnt, t0, t1) { size_t n = (size_t)std::max(1, (int)(cnt)); spans.push_back({place(n*sizeof(real)*nalloc), n}); }
  MJH_BATCH_REAL_FIELDS(X)
#undef X
#define X(name, cnt, lcnt, t0, t1) { size_t n = (size_t)std::max(1, (int)(cnt)); spans.push_back({place(n*sizeof(int)*nalloc), n}); }
  MJH_BATCH_INT_FIELDS(X)
#undef X
  Bt->arena_bytes = off;
  Bt->arena = Backend::alloc(off);
  if (!Bt->arena) {
    set_err("mjhip: device allocation failed (batch arena of " + std::to_string(off) + " bytes)");
    delete Bt;
    return nullptr;
  }
  Backend::zero(Bt->arena, off, nullptr);
  char* base = (char*)Bt->arena;
  size_t k = 0;
  memset(&Bt->D, 0, sizeof(DBatch));
  Bt->D.nenv = nenv;
  Bt->D.soa = Bt->soa;
#define X(name, cnt, lcnt, t0, t1) Bt->D.name = (real*)(base + spans[k].first); Bt->D.n_##name = (int)spans[k].second; \
  Bt->D.l_##name = -1; Bt->fields[#name] = FieldInfo{(void*)Bt->D.name, ((cnt) > 0 ? (int)(cnt) : 0), 0}; k++;
  MJH_BATCH_REAL_FIELDS(X)
#undef X
#define X(name, cnt, lcnt, t0, t1) Bt->D.name = (int*)(base + spans[k].first); Bt->D.n_##name = (int)spans[k].second; \
  Bt->D.l_##name = -1; Bt->fields[#name] = FieldInfo{(void*)Bt->D.name, ((cnt) > 0 ? (int)(cnt) : 0), 1}; k++;
  MJH_BATCH_INT_FIELDS(X)
#undef X
  // GJK / EPA workspace (mjh_convex.h): one private block per lane of every environment's wavefront;
  // raw memory, never read before it is written
  if (s.ccd_any) {
    const size_t bytes = (size_t)nalloc*(size_t)s.ccd_env_bytes;
    Bt->ccd_ws = Backend::alloc(bytes);
    if (!Bt->ccd_ws) {
      set_err("mjhip: device allocation failed (convex-collision workspace of " + std::to_string(bytes) + " bytes)");
      mjhip_batch_destroy(Bt);
      return nullptr;
    }
    Bt->D.ccd_ws = Bt->ccd_ws;
  }
  Bt->D_dev = (DBatch*)Backend::alloc(sizeof(DBatch));
  Bt->L_dev = (DBatch*)Backend::alloc(sizeof(DBatch));
  if (!Bt->D_dev || !Bt->L_dev || !Backend::h2d(Bt->D_dev, &Bt->D, sizeof(DBatch), nullptr)) {
    set_err("mjhip: device allocation failed (batch descriptor)");
    mjhip_batch_destroy(Bt);
    return nullptr;
  }
  {
    std::vector<int> ident((size_t)Bt->nenvpad);
    for (size_t i = 0; i < ident.size(); i++) ident[i] = (int)i;
    if (!Backend::h2d(Bt->D.perm, ident.data(), ident.size()*sizeof(int), nullptr) || !Backend::sync(nullptr)) {
      set_err("mjhip: batch initialisation failed"); mjhip_batch_destroy(Bt); return nullptr;
    }
  }
  // launch balancing ($MJHIP_BALANCE=0 keeps the identity order; measured in profiles/r02_balance)
  if (const char* ev = getenv("MJHIP_BALANCE")) Bt->balance = atoi(ev) != 0;
  if (const char* ev = getenv("MJHIP_MFMA")) { Bt->D.mfma = atoi(ev) != 0; Backend::h2d(Bt->D_dev, &Bt->D, sizeof(DBatch), nullptr); }
  if (const char* ev = getenv("MJHIP_PGS")) { Bt->D.pgs_mode = (ev[0] == 'r' || ev[0] == '1') ? 1 : 0; Backend::h2d(Bt->D_dev, &Bt->D, sizeof(DBatch), nullptr); }
  // kernel variant: the leanest mapping whose feature set covers the model ($MJHIP_VARIANT overrides)
  Bt->variant = default_variant(M, Bt->soa, Bt->nenv);
  // ($MJHIP_VARIANT is a preference: batches it cannot serve -- SoA layout, models that need
  // features the lean kernels lack -- keep their default)
  if (const char* ev = getenv("MJHIP_VARIANT")) (void)mjhip_batch_set_variant(Bt, ev);
  // residency plan: MJHIP_LDS_BYTES overrides the default per-workgroup LDS budget (0 disables)
  {
    // default: the CU's 160 KB shared by the one-wavefront workgroups a full launch puts on it -- 10 KB at 4096
    // environments on 256 CUs (all of them resident at once), 20 KB at 2048 (the cube's solver then keeps its sparse
    // factor in LDS: profiles/r03*/cube_lds_sweep.txt), never less than the 10 KB the stage lifetimes were tuned for
    int budget = MJHIP_DEFAULT_LDS_BYTES;
    {
      const int cus = Backend::num_cus();
      const int per_cu = (nenv + cus - 1)/cus;
      int b = (160*1024/(per_cu > 0 ? per_cu : 1)) & ~1023;
      if (b > Backend::max_lds()) b = Backend::max_lds();
      if (b > budget) budget = b;
    }
    if (const char* ev = getenv("MJHIP_LDS_BYTES")) budget = atoi(ev);
    if (mjhip_batch_plan_lds(Bt, budget) < 0) { mjhip_batch_destroy(Bt); return nullptr; }
  }
  if (mjhip_batch_reset(Bt) != 0) { mjhip_batch_destroy(Bt); return nullptr; }
  return Bt;
}

struct StageBuf;
static void stage_release(mjhipStage_* st);
MJHIP_API void mjhip_batch_destroy(mjhipBatch* Bt) {
  if (!Bt) return;
  std::string err_;
  Backend::set_device(Bt->device, &err_);
  if (Bt->stage) stage_release(Bt->stage);
  if (Bt->arena) Backend::free(Bt->arena);
  if (Bt->ccd_ws) Backend::free(Bt->ccd_ws);
  if (Bt->D_dev) Backend::free(Bt->D_dev);
  if (Bt->L_dev) Backend::free(Bt->L_dev);
  delete Bt;
}

MJHIP_API int mjhip_batch_nenv(const mjhipBatch* Bt) { return Bt ? Bt->nenv : -1; }

// which kernel mappings can step this model at all
static bool variant_ok(const mjhipModel_* M, int soa, int variant, std::string* why) {
  if (variant < 0 || variant >= MJH_NVARIANT) { if (why) *why = "unknown variant"; return false; }
  if (variant == MJH_VAR_GENERIC) return true;
  if (soa) { if (why) *why = "the lean and multi-wavefront kernel variants step environment-major (AoS) batches only"; return false; }
  if (variant == MJH_VAR_MULTIWAVE) return true;
  const int missing = M->H.s.features & ~mjh_variant_features(variant);
  if (missing) {
    if (why) { char b[96]; snprintf(b, sizeof b, "the model needs features 0x%x that the lean kernels do not carry", missing); *why = b; }
    return false;
  }
  return true;
}
static int default_variant(const mjhipModel_* M, int soa, int nenv) {
  // flex models at launches of at most one workgroup per CU: MJH_MW wavefronts per environment ($MJHIP_MW=0 turns
  // the choice off: A/B runs)
  {
    static const int mw = [] { const char* ev = getenv("MJHIP_MW"); return ev ? atoi(ev) : 1; }();
    if (mw && !soa && M->H.s.nflex > 0 && nenv <= Backend::num_cus()) return MJH_VAR_MULTIWAVE;
  }
  // one environment per wavefront: measured fastest at the batch sizes of interest (4096 envs on
  // 1024 SIMDs: profiles/r02_variants -- with every environment resident the step is bound by the
  // dependent chain of one environment, and the two / four environments of a shared wavefront run
  // their solver loops to the longer of their iteration counts); those variants were deleted in round 3
  if (variant_ok(M, soa, MJH_VAR_LEAN, nullptr)) return MJH_VAR_LEAN;
  return MJH_VAR_GENERIC;
}

MJHIP_API int mjhip_batch_set_variant(mjhipBatch* Bt, const char* name) {
  if (!Bt || !name) return -1;
  int v = -1;
  for (int k = 0; k < MJH_NVARIANT; k++) if (!strcmp(name, mjh_variant_name(k))) v = k;
  if (!strcmp(name, "auto")) v = default_variant(Bt->model, Bt->soa, Bt->nenv);
  std::string why;
  if (!variant_ok(Bt->model, Bt->soa, v, &why)) {
    set_err(std::string("mjhip_batch_set_variant(") + name + "): " + why);
    return -2;
  }
  Bt->variant = v;
  // re-plan with the budget originally asked for (plan_lds clamps it per variant)
  if (Bt->L_dev) return mjhip_batch_plan_lds(Bt, Bt->lds_request) < 0 ? -3 : 0;
  return 0;
}
// AR = Y Y' on the matrix cores (tolerance parity); default off, $MJHIP_MFMA=1 turns it on at creation
MJHIP_API int mjhip_batch_set_mfma(mjhipBatch* Bt, int on) {
  if (!Bt) return -1;
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  Bt->D.mfma = Bt->L.mfma = on ? 1 : 0;
  if (!Backend::h2d(Bt->D_dev, &Bt->D, sizeof(DBatch), nullptr) || !Backend::h2d(Bt->L_dev, &Bt->L, sizeof(DBatch), nullptr) ||
      !Backend::sync(nullptr)) { set_err("mjhip_batch_set_mfma: descriptor upload failed"); return -2; }
  return 0;
}
// PGS sweep: 0 = the reference's, bit for bit (default); 1 = residual-update form (tolerance parity); $MJHIP_PGS=residual at creation
MJHIP_API int mjhip_batch_set_pgs_mode(mjhipBatch* Bt, int mode) {
  if (!Bt || mode < 0 || mode > 1) { set_err("mjhip_batch_set_pgs_mode: bad arguments"); return -1; }
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  Bt->D.pgs_mode = Bt->L.pgs_mode = mode;
  if (!Backend::h2d(Bt->D_dev, &Bt->D, sizeof(DBatch), nullptr) || !Backend::h2d(Bt->L_dev, &Bt->L, sizeof(DBatch), nullptr) ||
      !Backend::sync(nullptr)) { set_err("mjhip_batch_set_pgs_mode: descriptor upload failed"); return -2; }
  return 0;
}
MJHIP_API const char* mjhip_batch_variant(const mjhipBatch* Bt) { return Bt ? mjh_variant_name(Bt->variant) : ""; }
MJHIP_API const char* mjhip_batch_kernel(const mjhipBatch* Bt) { return Bt ? Backend::rollout_kernel_name(Bt->variant, Bt->nenv) : ""; }

MJHIP_API int mjhip_batch_plan_lds(mjhipBatch* Bt, int lds_bytes) {
  if (!Bt) return -1;
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  if (lds_bytes < 0) lds_bytes = 0;
  Bt->lds_request = lds_bytes;
  const int max_lds = Backend::max_lds() / mjh_variant_nsub(Bt->variant) - (Bt->variant == MJH_VAR_MULTIWAVE ? MJH_MW_LDS_TAIL : 0);
  if (lds_bytes > max_lds) lds_bytes = max_lds;

  // equality constraints read kinematics / velocity quantities long after their usual lifetimes
  // (rows at make, Jdot*v at reference): such models keep those fields in their global homes
  std::vector<std::string> eqskip;
  if (Bt->model->H.s.neq > 0)
    eqskip = {"xpos", "xquat", "xmat", "site_xpos", "cdof", "subtree_com", "cvel", "cdof_dot"};
  // implicitfast differentiates the actuator / damper forces at integration time, and standalone
  // free bodies need their frames for the gyroscopic derivative
  if (Bt->model->H.o.integrator == MJH_INT_IMPLICIT)          // (mjd_rne_vel at integration time)
    for (const char* f : {"cdof", "cdof_dot", "cvel", "cinert", "subtree_com"}) eqskip.push_back(f);
  if (Bt->model->H.o.integrator == MJH_INT_IMPLICITFAST || Bt->model->H.o.integrator == MJH_INT_IMPLICIT) {
    for (const char* f : {"xpos", "xmat", "xipos", "ximat", "ten_J", "ten_velocity", "actuator_moment",
                          "moment_rownnz", "moment_colind", "actuator_force", "actuator_length", "actuator_velocity"})
      eqskip.push_back(f);
    if (Bt->model->H.o.has_fluid) { eqskip.push_back("cdof"); eqskip.push_back("subtree_com"); }   // the fluid forces' velocity derivative
    if (Bt->model->H.s.ngeom_fluid) { eqskip.push_back("geom_xpos"); eqskip.push_back("geom_xmat"); }
  }
  // sensors are evaluated after the solve and read kinematic / velocity / actuator / contact
  // quantities past their usual lifetimes: those fields stay in their global homes; the constraint
  // arrays (efc_*) are still intact at that point and the rest of the plan is unaffected
  if (Bt->model->H.s.nsensor > 0) {
    for (const char* f : {"xpos", "xquat", "xmat", "xipos", "ximat", "geom_xpos", "geom_xmat", "site_xpos", "site_xmat",
                          "subtree_com", "cinert", "cdof", "cdof_dot", "cvel", "ten_length", "ten_velocity", "ten_J",
                          "actuator_length", "actuator_velocity", "actuator_force", "qfrc_actuator",
                          "con_dist", "con_pos", "con_frame", "con_mu", "con_pair", "con_geom", "con_dim",
                          "con_exclude", "con_efcadr",
                          // born at finish: they would overlay the constraint arrays the sensors still read
                          "qH2", "qH2DiagInv", "qe"})
      eqskip.push_back(f);
  }
  if (Bt->model->H.o.has_gravcomp) eqskip.push_back("xipos");    // read by the passive stage
  if (Bt->model->H.o.has_refsite) eqskip.push_back("xquat");     // site transmissions with a reference site: relative orientation
  if (Bt->model->H.o.has_tendon_wrap) { eqskip.push_back("geom_xpos"); eqskip.push_back("geom_xmat"); }   // mju_wrap at the tendon stage
  if (Bt->model->H.o.has_ten_armature) eqskip.push_back("site_xpos");   // mj_tendonDot reads the sites at the bias-force stage
  if (Bt->model->H.o.has_surfacevel) {   // contact geometry is read again by the reference stage
    for (const char* f : {"geom_xpos", "geom_xmat", "con_pos", "con_frame", "con_pair", "con_geom", "con_dim", "con_efcadr"})
      eqskip.push_back(f);
  }
  if (Bt->model->H.o.has_adhesion) {     // the reference stage looks the adhesive rows up through the contacts
    for (const char* f : {"con_pair", "con_dim", "con_efcadr"}) eqskip.push_back(f);
  }
  if (Bt->model->H.o.has_fluid) { eqskip.push_back("xipos"); eqskip.push_back("ximat"); }
  if (Bt->model->H.s.ngeom_fluid) { eqskip.push_back("geom_xpos"); eqskip.push_back("geom_xmat"); }   // ellipsoid fluid model: geom frames at the passive stage
  if (Bt->xfrc_on) eqskip.push_back("xipos");                  // Cartesian forces act at the body COMs (stage_acceleration)
  if (Bt->soa) {
    // the constraint kernel of the per-step pipeline: collision .. PGS
    std::vector<std::string> skip = {"time", "act", "ctrl", "qfrc_applied", "qfrc_smooth", "qLDiagInv"};
    skip.insert(skip.end(), eqskip.begin(), eqskip.end());
    unsigned ex = 1u << MJH_T_COLLISION;
    for (int t = MJH_T_MAKE; t <= MJH_T_CONSTRAINT; t++) ex |= 1u << t;
    plan_lds(Bt, lds_bytes, ex, skip, {"qpos", "qvel", "qacc_warmstart"}, &Bt->plan_report);
  } else {
    // the single wave-per-environment kernel: the whole step
    unsigned ex = 0;
    for (int t = MJH_T_KIN; t <= MJH_T_EULER; t++) ex |= 1u << t;
    plan_lds(Bt, lds_bytes, ex, eqskip, {}, &Bt->plan_report);
  }
  Bt->L.xfrc_on = Bt->xfrc_on ? 1 : 0;
  if (!Backend::h2d(Bt->L_dev, &Bt->L, sizeof(DBatch), nullptr) || !Backend::sync(nullptr)) {
    set_err("mjhip_batch_plan_lds: descriptor upload failed");
    return -2;
  }
  return Bt->L.lds_bytes ? Bt->L.lds_bytes - Bt->L.dyn_off : 0;
}

MJHIP_API const char* mjhip_batch_lds_report(const mjhipBatch* Bt) { return Bt ? Bt->plan_report.c_str() : ""; }

MJHIP_API int mjhip_batch_reset(mjhipBatch* Bt) {
  if (!Bt) return -1;
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  bool ok = Bt->soa ? Backend::launch_lane_reset(Bt->model->D_dev, Bt->D_dev, Bt->nenv, Bt->epw, nullptr)
                    : Backend::launch_reset(Bt->model->D_dev, Bt->D_dev, Bt->nenv, nullptr);
  if (!ok || !Backend::sync(nullptr)) {
    set_err("mjhip_batch_reset: kernel launch failed");
    return -2;
  }
  return 0;
}

MJHIP_API int mjhip_batch_field(mjhipBatch* Bt, const char* name, void** device_ptr,
                                int* count_per_env, int* is_int) {
  if (!Bt || !name) return -1;
  auto it = Bt->fields.find(name);
  if (it == Bt->fields.end()) { set_err(std::string("mjhip: unknown batch field ") + name); return -2; }
  if (device_ptr) *device_ptr = it->second.ptr;
  if (count_per_env) *count_per_env = it->second.count;
  if (is_int) *is_int = it->second.is_int;
  return 0;
}

static size_t field_stride(const mjhipBatch_* Bt, const FieldInfo& f) {
  return (size_t)std::max(1, f.count) * (f.is_int ? sizeof(int) : sizeof(real));
}

// host <-> device copies of one field; the host side is always [nenv][count]
MJHIP_API int mjhip_batch_get(mjhipBatch* Bt, const char* name, void* host_dst) {
  void* p; int cnt, isint;
  if (mjhip_batch_field(Bt, name, &p, &cnt, &isint)) return -2;
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  if (cnt == 0) return 0;
  const size_t esz = isint ? sizeof(int) : sizeof(real);
  if (!Bt->soa) {
    size_t bytes = (size_t)cnt * esz * Bt->nenv;
    if (!Backend::d2h(host_dst, p, bytes, nullptr) || !Backend::sync(nullptr)) {
      set_err("mjhip_batch_get: copy failed"); return -3;
    }
    return 0;
  }
  const size_t np = (size_t)Bt->nenvpad;
  std::vector<char> tmp((size_t)cnt * np * esz);
  if (!Backend::d2h(tmp.data(), p, tmp.size(), nullptr) || !Backend::sync(nullptr)) {
    set_err("mjhip_batch_get: copy failed"); return -3;
  }
  for (int e = 0; e < Bt->nenv; e++)
    for (int i = 0; i < cnt; i++)
      memcpy((char*)host_dst + ((size_t)e*cnt + i)*esz, tmp.data() + ((size_t)i*np + e)*esz, esz);
  return 0;
}

static int enable_xfrc(mjhipBatch_* Bt);
MJHIP_API int mjhip_batch_set(mjhipBatch* Bt, const char* name, const void* host_src) {
  void* p; int cnt, isint;
  if (mjhip_batch_field(Bt, name, &p, &cnt, &isint)) return -2;
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  if (!strcmp(name, "xfrc_applied") && enable_xfrc(Bt)) return -3;
  if (cnt == 0) return 0;
  const size_t esz = isint ? sizeof(int) : sizeof(real);
  if (!Bt->soa) {
    size_t bytes = (size_t)cnt * esz * Bt->nenv;
    if (!Backend::h2d(p, host_src, bytes, nullptr) || !Backend::sync(nullptr)) {
      set_err("mjhip_batch_set: copy failed"); return -3;
    }
    return 0;
  }
  const size_t np = (size_t)Bt->nenvpad;
  std::vector<char> tmp((size_t)cnt * np * esz, 0);
  for (int e = 0; e < Bt->nenv; e++)
    for (int i = 0; i < cnt; i++)
      memcpy(tmp.data() + ((size_t)i*np + e)*esz, (const char*)host_src + ((size_t)e*cnt + i)*esz, esz);
  if (!Backend::h2d(p, tmp.data(), tmp.size(), nullptr) || !Backend::sync(nullptr)) {
    set_err("mjhip_batch_set: copy failed"); return -3;
  }
  return 0;
}

// one mj_step of every environment on an SoA batch: smooth (lane) -> constraint (wave) -> integrate (lane)
static bool pipeline_step(mjhipBatch_* Bt, const RolloutArgs& A, void* stream) {
  const DModel* M = Bt->model->D_dev;
  if (!Backend::launch_smooth(M, Bt->D_dev, Bt->nenv, Bt->epw, A, stream)) return false;
  const bool lds = Bt->L.lds_bytes > 0;
  if (!Backend::launch_forward(M, lds ? Bt->L_dev : Bt->D_dev, Bt->nenv, MJH_STAGES_CONSTRAINT_MASK | MJH_STAGE_IFACTIVE,
                               lds ? Bt->L.lds_bytes : 0, 1, MJH_VAR_GENERIC, stream)) return false;
  return Backend::launch_integrate(M, Bt->D_dev, Bt->nenv, Bt->epw, A, stream);
}

MJHIP_API int mjhip_batch_forward(mjhipBatch* Bt, int stages, void* stream) {
  if (!Bt) return -1;
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  if (stages < 0) stages = MJH_STAGE_ALL;   // -1: mj_forward
  if ((stages & MJH_STAGE_ALL) == MJH_STAGE_ALL) stages |= MJH_STAGE_SENSOR;   // mj_forward evaluates the sensors
  // MJHIP_STAGE_LDS: run on the LDS residency plan and write every stage's fields back to their
  // global homes (debug / parity tests of the resident path); default: everything global
  const bool lds = (stages & MJH_STAGE_LDS) && Bt->L.lds_bytes;
  stages &= ~(MJH_STAGE_LDS | MJH_STAGE_WRITEBACK | MJH_STAGE_IFACTIVE);
  bool ok = true;
  if (!Bt->soa) {
    if (lds) stages |= MJH_STAGE_WRITEBACK;
    ok = Backend::launch_forward(Bt->model->D_dev, lds ? Bt->L_dev : Bt->D_dev, Bt->nenv, stages,
                                 lds ? Bt->L.lds_bytes : 0, 0, Bt->variant, stream);
  } else {
    // the pipeline's split: lane-mode kernel for the smooth stages, wave-mode kernel for the
    // constraint stages (on its LDS plan + write-back if asked), lane-mode kernel for the tail
    const int s1 = stages & MJH_STAGES_SMOOTH_MASK, s2 = stages & MJH_STAGES_CONSTRAINT_MASK;
    const int s3 = stages & (MJH_STAGE_FINISH | MJH_STAGE_EULER);
    if (s1) ok = ok && Backend::launch_lane_forward(Bt->model->D_dev, Bt->D_dev, Bt->nenv, Bt->epw, s1, stream);
    if (s2) ok = ok && Backend::launch_forward(Bt->model->D_dev, lds ? Bt->L_dev : Bt->D_dev, Bt->nenv,
                                               s2 | (lds ? MJH_STAGE_WRITEBACK : 0), lds ? Bt->L.lds_bytes : 0, 1,
                                               MJH_VAR_GENERIC, stream);
    if (s3) ok = ok && Backend::launch_lane_forward(Bt->model->D_dev, Bt->D_dev, Bt->nenv, Bt->epw, s3, stream);
  }
  if (!ok) { set_err("mjhip_batch_forward: kernel launch failed"); return -2; }
  return 0;
}

// mj_step1 / mj_step2 (engine_forward.c:1884, :1916): the step split around the point where a
// controller may read positions / velocities / position- and velocity-stage sensors and write ctrl
// (mj_step1 ends with mj_sensorPos + mj_sensorVel; mj_sensorAcc runs inside mj_step2).  Between the two
// calls every intermediate lives in its global field (inspectable with mjhip_batch_get).
static int step_half(mjhipBatch_* Bt, int stages, const char* who, void* stream) {
  if (!Bt) return -1;
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  if (Bt->soa) { set_err(std::string(who) + ": needs the AoS (wave-per-environment) layout"); return -2; }
  if (!Backend::launch_forward(Bt->model->D_dev, Bt->D_dev, Bt->nenv, stages, 0, 0, Bt->variant, stream)) {
    set_err(std::string(who) + ": kernel launch failed");
    return -2;
  }
  return 0;
}
MJHIP_API int mjhip_batch_step1(mjhipBatch* Bt, void* stream) {
  return step_half(Bt, MJH_STAGE_CHECKPV | MJH_STAGE_KINEMATICS | MJH_STAGE_INERTIA | MJH_STAGE_COLLISION | MJH_STAGE_MAKE |
                       MJH_STAGE_PROJECT | MJH_STAGE_TRANSMISSION | MJH_STAGE_VELOCITY | MJH_STAGE_REFERENCE | MJH_STAGE_SENSPV,
                   "mjhip_batch_step1", stream);
}
MJHIP_API int mjhip_batch_step2(mjhipBatch* Bt, void* stream) {
  return step_half(Bt, MJH_STAGE_ACTUATION | MJH_STAGE_CONSTRAINT | MJH_STAGE_FINISH | MJH_STAGE_SENSACC |
                       MJH_STAGE_CHECKACC | MJH_STAGE_INTEGRATE, "mjhip_batch_step2", stream);
}

MJHIP_API int mjhip_batch_step(mjhipBatch* Bt, int nstep, void* stream) {
  if (!Bt || nstep < 0) return -1;
  { std::string e_; if (!Backend::set_device(Bt->device, &e_)) { set_err(e_); return -1; } }
  RolloutArgs A;
  memset(&A, 0, sizeof(A));
  A.nstep = nstep;
  A.has_ctrl = 1; A.has_qfrc = 1;    // keep the resident ctrl / qfrc_applied / mocap poses
  A.mpos_off = 0; A.mquat_off = 0;
  A.xfrc_off = 0; A.eq_off = 0; A.ud_off = 0;
  A.init = 0;
  bool ok = true;
  if (Bt->soa) {
    for (int t = 0; t < nstep && ok; t++) { A.t0 = t; ok = pipeline_step(Bt, A, stream); }
  } else {
    ok = Backend::launch_rollout(Bt->model->D_dev, Bt->L_dev, Bt->nenv, A, Bt->L.lds_bytes, Bt->variant, stream);
    if (ok && Bt->balance) ok = Backend::launch_balance(Bt->L_dev, Bt->nenv, nstep, stream);
  }
  if (!ok) { set_err("mjhip_batch_step: kernel launch failed"); return -2; }
  return 0;
}

// mjtState bits (include/mujoco/mjtype.h:504-527)
// layout of one control vector = the mjtState bit order of mj_getState (engine_support.c:214):
// every mjSTATE_USER element can be driven (ctrl, qfrc_applied, xfrc_applied, eq_active, mocap_pos,
// mocap_quat, userdata), exactly what _unsafe_rollout hands to mj_setState (rollout.cc:160)
struct ControlLayout { int n, qfrc, xfrc, eq, mpos, mquat, ud; };
static bool control_layout(const DSizes& s, unsigned spec, ControlLayout* L, std::string* err) {
  const unsigned user = mjSTATE_CTRL | mjSTATE_QFRC_APPLIED | mjSTATE_XFRC_APPLIED | mjSTATE_EQ_ACTIVE |
                        mjSTATE_MOCAP_POS | mjSTATE_MOCAP_QUAT | mjSTATE_USERDATA;
  if (spec & ~user) {
    *err = "mjhip: control_spec may only contain mjSTATE_USER bits (ctrl, qfrc_applied, xfrc_applied, eq_active, "
           "mocap_pos, mocap_quat, userdata)";
    return false;
  }
  int n = 0;
  if (spec & mjSTATE_CTRL) n += s.nu;
  L->qfrc = n;
  if (spec & mjSTATE_QFRC_APPLIED) n += s.nv;
  L->xfrc = (spec & mjSTATE_XFRC_APPLIED) ? n : -1;
  if (spec & mjSTATE_XFRC_APPLIED) n += 6*s.nbody;
  L->eq = (spec & mjSTATE_EQ_ACTIVE) ? n : -1;
  if (spec & mjSTATE_EQ_ACTIVE) n += s.neq;
  L->mpos = (spec & mjSTATE_MOCAP_POS) ? n : -1;
  if (spec & mjSTATE_MOCAP_POS) n += 3*s.nmocap;
  L->mquat = (spec & mjSTATE_MOCAP_QUAT) ? n : -1;
  if (spec & mjSTATE_MOCAP_QUAT) n += 4*s.nmocap;
  L->ud = (spec & mjSTATE_USERDATA) ? n : -1;
  if (spec & mjSTATE_USERDATA) n += s.nuserdata;
  L->n = n;
  return true;
}

// Cartesian forces become possible inputs: mj_xfrcAccumulate runs and the plan keeps xipos readable
static int enable_xfrc(mjhipBatch_* Bt) {
  if (Bt->xfrc_on) return 0;
  Bt->xfrc_on = true;
  Bt->D.xfrc_on = 1;
  if (!Backend::h2d(Bt->D_dev, &Bt->D, sizeof(DBatch), nullptr)) { set_err("mjhip: descriptor upload failed"); return -2; }
  return mjhip_batch_plan_lds(Bt, Bt->lds_request) < 0 ? -2 : 0;
}

// grow-only device staging buffers of one batch: the rollout arrays of host-pointer calls
struct StageBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) Backend::free(p);
    cap = bytes + bytes/4;
    p = Backend::alloc(cap);
    if (!p) { cap = 0; return false; }
    return true;
  }
  void release() { if (p) Backend::free(p); p = nullptr; cap = 0; }
};
struct mjhipStage_ { StageBuf state0, warm, control, state, sens; void* copy_in = nullptr; void* copy_out = nullptr; };
static void stage_release(mjhipStage_* st) {
  st->state0.release(); st->warm.release(); st->control.release(); st->state.release(); st->sens.release();
  Backend::stream_destroy(st->copy_in); Backend::stream_destroy(st->copy_out);
  delete st;
}

MJHIP_API int mjhip_batch_rollout_sensors(mjhipBatch* Bt, int nstep, unsigned control_spec,
                                          const double* state0, const double* warmstart0,
                                          const double* control, double* state, double* sensordata,
                                          int on_device, void* stream);
MJHIP_API int mjhip_batch_rollout(mjhipBatch* Bt, int nstep, unsigned control_spec,
                                  const double* state0, const double* warmstart0,
                                  const double* control, double* state, int on_device,
                                  void* stream) {
  return mjhip_batch_rollout_sensors(Bt, nstep, control_spec, state0, warmstart0, control, state, nullptr,
                                     on_device, stream);
}

// nlaunch: environments [0, nlaunch) of the batch take part (0 = all of them)
static int rollout_impl(mjhipBatch_* Bt, int nlaunch, int nstep, unsigned control_spec,
                        const double* state0, const double* warmstart0, const double* control,
                        double* state, double* sensordata, int on_device, void* stream) {
  if (!Bt || nstep < 0 || nlaunch < 0 || nlaunch > Bt->nenv) { set_err("mjhip_batch_rollout: bad arguments"); return -1; }
  std::string err;
  if (!Backend::set_device(Bt->device, &err)) { set_err(err); return -1; }
  if (sensordata && Bt->model->H.s.nsensordata == 0) sensordata = nullptr;
  if (sensordata && Bt->soa) { set_err("mjhip_batch_rollout: sensordata needs the AoS (wave-per-environment) layout"); return -2; }
  if (nlaunch && nlaunch < Bt->nenv && Bt->soa) { set_err("mjhip_batch_rollout: partial launches need the AoS layout"); return -2; }
  const DSizes& s = Bt->model->H.s;
  ControlLayout CL;
  if (!control_layout(s, control_spec, &CL, &err)) { set_err(err); return -2; }
  if ((control_spec & mjSTATE_XFRC_APPLIED) && enable_xfrc(Bt)) return -2;
  const size_t nenv = nlaunch ? (size_t)nlaunch : (size_t)Bt->nenv;
  RolloutArgs A;
  memset(&A, 0, sizeof(A));
  A.nstep = nstep;
  A.has_ctrl = (control_spec & mjSTATE_CTRL) ? 1 : 0;
  A.has_qfrc = (control_spec & mjSTATE_QFRC_APPLIED) ? 1 : 0;
  A.ncontrol = CL.n;
  A.qfrc_off = CL.qfrc;
  A.mpos_off = CL.mpos; A.mquat_off = CL.mquat;
  A.xfrc_off = CL.xfrc; A.eq_off = CL.eq; A.ud_off = CL.ud;
  A.nlaunch = (nlaunch && nlaunch < Bt->nenv) ? nlaunch : 0;
  A.init = (on_device & MJHIP_ROLLOUT_CONTINUE) ? 0 : 1;
  A.pitch = nstep; A.tbase = 0;
  on_device &= MJHIP_ROLLOUT_ON_DEVICE;
  // Host arrays, long rollouts: the rollout is launched in chunks of steps so that the copies overlap the kernels -- the
  // controls of chunk k + 1 go up and the states / sensor data of chunk k - 1 come down while chunk k runs (two copy
  // streams next to the caller's).  The device arrays keep the caller's layout [env][step][...], so a chunk is a strided
  // block on both sides (2-D copies); the kernels address it through (pitch, tbase).  $MJHIP_ROLLOUT_CHUNK sets the chunk
  // length (0, the default: one launch, copies before and after it).
  // (measured on MI355X, humanoid 4096 x 250 steps, profiles/r05/api_rate.txt: ONE launch with the copies before and after it
  // reaches 0.89 of the device-resident rate -- pageable 1-D copies run at ~50 GB/s --, while 2 / 3 / 5 chunks reach 0.76 / 0.68 /
  // 0.61: every launch ends with its stragglers, and the strided 2-D copies are slower than the 12 ms they hide.  So the
  // default is one launch; the chunked path stays selectable.)
  const int chunk_steps = [] { const char* ev = getenv("MJHIP_ROLLOUT_CHUNK"); return ev ? atoi(ev) : 0; }();
  if (!on_device && !Bt->soa && chunk_steps > 0 && nstep >= 2*chunk_steps && (state || sensordata)) {
    if (!Bt->stage) Bt->stage = new mjhipStage_();
    mjhipStage_* S = Bt->stage;
    if (!S->copy_in) S->copy_in = Backend::stream_create();
    if (!S->copy_out) S->copy_out = Backend::stream_create();
    if (S->copy_in && S->copy_out) {
      auto up = [&](StageBuf& b, const double* src, size_t n, const real** dst) -> bool {
        if (!src || n == 0) { *dst = nullptr; return true; }
        if (!b.ensure(n*sizeof(real))) return false;
        *dst = (const real*)b.p;
        return Backend::h2d(b.p, src, n*sizeof(real), stream);
      };
      bool ok = up(S->state0, state0, nenv*s.nstate, &A.state0) && up(S->warm, warmstart0, nenv*s.nv, &A.warmstart0);
      const size_t crow = (size_t)nstep*CL.n*sizeof(real), srow = (size_t)nstep*s.nstate*sizeof(real),
                   drow = (size_t)nstep*s.nsensordata*sizeof(real);
      if (ok && control && CL.n) { ok = S->control.ensure(nenv*crow); A.control = (const real*)S->control.p; }
      if (ok && state) { ok = S->state.ensure(nenv*srow); A.state = (real*)S->state.p; }
      if (ok && sensordata) { ok = S->sens.ensure(nenv*drow); A.sensordata = (real*)S->sens.p; }
      if (!ok) { set_err("mjhip_batch_rollout: staging allocation/copy failed"); return -3; }
      const int nchunk = (nstep + chunk_steps - 1)/chunk_steps;
      auto ctrl_up = [&](int k) -> bool {
        if (!A.control || k >= nchunk) return true;
        const int t0 = k*chunk_steps, c = std::min(chunk_steps, nstep - t0);
        const size_t off = (size_t)t0*CL.n*sizeof(real);
        return Backend::copy2d((char*)S->control.p + off, crow, (const char*)control + off, crow, (size_t)c*CL.n*sizeof(real), nenv, 0, S->copy_in);
      };
      auto state_down = [&](int k) -> bool {
        const int t0 = k*chunk_steps, c = std::min(chunk_steps, nstep - t0);
        bool okd = true;
        if (state) okd = Backend::copy2d((char*)state + (size_t)t0*s.nstate*sizeof(real), srow, (const char*)S->state.p + (size_t)t0*s.nstate*sizeof(real),
                                         srow, (size_t)c*s.nstate*sizeof(real), nenv, 1, S->copy_out);
        if (okd && sensordata) okd = Backend::copy2d((char*)sensordata + (size_t)t0*s.nsensordata*sizeof(real), drow,
                                                     (const char*)S->sens.p + (size_t)t0*s.nsensordata*sizeof(real), drow,
                                                     (size_t)c*s.nsensordata*sizeof(real), nenv, 1, S->copy_out);
        return okd;
      };
      // Copies between the device and PAGEABLE host memory (numpy arrays) keep the calling thread until they are done, so
      // the order of the calls matters: kernel k is launched FIRST, then -- while it runs -- chunk k - 1 comes down and the
      // controls of chunk k + 1 go up.  copy_in starts behind whatever the caller's stream still has queued (an earlier
      // call's kernel may read the staging buffers); copy_out's copy of chunk k waits for kernel k through an event
      // recorded right after that launch.
      ok = Backend::stream_follow(S->copy_in, stream) && ctrl_up(0);
      bool launched = true;
      for (int k = 0; k < nchunk && ok; k++) {
        const int t0 = k*chunk_steps, c = std::min(chunk_steps, nstep - t0);
        ok = Backend::stream_follow(stream, S->copy_in);            // kernel k waits for the uploads queued so far (chunk k's)
        A.nstep = c; A.tbase = t0;
        // (chunks after the first add their work to the environment's cost word -- rollout_env, tbase > 0 -- so that the
        //  launch order and the priority reference of the next call come from the whole rollout, as in the one-launch path)
        if (ok) ok = launched = Backend::launch_rollout(Bt->model->D_dev, Bt->L_dev, (int)nenv, A, Bt->L.lds_bytes, Bt->variant, stream);
        A.init = 0;
        if (ok && k > 0) ok = state_down(k - 1);                     // (copy_out already waits for kernel k - 1)
        if (ok) ok = Backend::stream_follow(S->copy_out, stream);   // what copy_out is given next waits for kernel k
        if (ok) ok = ctrl_up(k + 1);
      }
      if (ok) ok = state_down(nchunk - 1);
      if (ok && Bt->balance && !A.nlaunch) ok = launched = Backend::launch_balance(Bt->L_dev, Bt->nenv, nstep, stream);
      ok = Backend::sync(stream) && ok;
      ok = Backend::sync(S->copy_out) && ok;
      ok = Backend::sync(S->copy_in) && ok;
      if (!launched) { set_err("mjhip_batch_rollout: kernel launch failed"); return -4; }       // (same codes as the one-launch path)
      if (!ok) { set_err("mjhip_batch_rollout: device<->host copy failed"); return -5; }
      return 0;
    }
  }
  if (on_device) {
    A.state0 = state0; A.warmstart0 = warmstart0; A.control = control; A.state = state;
    A.sensordata = sensordata;
  } else {
    // host pointers: pooled device staging (grow-only, owned by the batch), asynchronous copies
    if (!Bt->stage) Bt->stage = new mjhipStage_();
    auto up = [&](StageBuf& b, const double* src, size_t n, const real** dst) -> bool {
      if (!src || n == 0) { *dst = nullptr; return true; }
      if (!b.ensure(n*sizeof(real))) return false;
      *dst = (const real*)b.p;
      return Backend::h2d(b.p, src, n*sizeof(real), stream);
    };
    bool ok = up(Bt->stage->state0, state0, nenv*s.nstate, &A.state0) &&
              up(Bt->stage->warm, warmstart0, nenv*s.nv, &A.warmstart0) &&
              up(Bt->stage->control, control, CL.n ? nenv*(size_t)nstep*CL.n : 0, &A.control);
    if (ok && state && nstep > 0) {
      ok = Bt->stage->state.ensure(nenv*(size_t)nstep*s.nstate*sizeof(real));
      A.state = (real*)Bt->stage->state.p;
    }
    if (ok && sensordata && nstep > 0) {
      ok = Bt->stage->sens.ensure(nenv*(size_t)nstep*s.nsensordata*sizeof(real));
      A.sensordata = (real*)Bt->stage->sens.p;
    }
    if (!ok) { set_err("mjhip_batch_rollout: staging allocation/copy failed"); return -3; }
  }
  bool launched = true;
  if (Bt->soa) {
    for (int t = 0; t < nstep && launched; t++) {
      A.t0 = t;
      launched = pipeline_step(Bt, A, stream);
      A.init = 0;
    }
  } else {
    launched = Backend::launch_rollout(Bt->model->D_dev, Bt->L_dev, (int)nenv, A, Bt->L.lds_bytes, Bt->variant, stream);
    // (a partial launch runs in identity order and leaves the launch order of full launches alone)
    if (launched && Bt->balance && !A.nlaunch) launched = Backend::launch_balance(Bt->L_dev, Bt->nenv, nstep, stream);
  }
  if (!launched) { set_err("mjhip_batch_rollout: kernel launch failed"); return -4; }
  if (!on_device) {
    bool ok = true;
    if (state && nstep > 0) ok = Backend::d2h(state, A.state, nenv*(size_t)nstep*s.nstate*sizeof(real), stream);
    if (sensordata && nstep > 0)
      ok = Backend::d2h(sensordata, A.sensordata, nenv*(size_t)nstep*s.nsensordata*sizeof(real), stream) && ok;
    ok = Backend::sync(stream) && ok;
    if (!ok) { set_err("mjhip_batch_rollout: device->host copy failed"); return -5; }
  }
  return 0;
}

MJHIP_API int mjhip_batch_rollout_sensors(mjhipBatch* Bt, int nstep, unsigned control_spec,
                                          const double* state0, const double* warmstart0,
                                          const double* control, double* state, double* sensordata,
                                          int on_device, void* stream) {
  return rollout_impl(Bt, 0, nstep, control_spec, state0, warmstart0, control, state, sensordata, on_device, stream);
}

MJHIP_API int mjhip_batch_sync(mjhipBatch* Bt, void* stream) {
  std::string err;
  if (Bt && !Backend::set_device(Bt->device, &err)) { set_err(err); return -1; }
  if (!Backend::sync(stream)) { set_err("mjhip_batch_sync: failed"); return -1; }
  return 0;
}

// sum over environments [0, n) of the warning counters that mean "this trajectory is NOT what the
// reference would have computed": capacity overflow (the reference's arena grows) and colliders
// mjhip lacks.  Returns <0 on error.
MJHIP_API int mjhip_batch_trouble(mjhipBatch* Bt, int n, int* ncapacity, int* nunsupported) {
  if (!Bt) return -1;
  if (n <= 0 || n > Bt->nenv) n = Bt->nenv;
  std::vector<int> w((size_t)Bt->nenv*8);
  if (mjhip_batch_get(Bt, "warning", w.data())) return -2;
  int cap = 0, uns = 0;
  for (int e = 0; e < n; e++) {
    cap += w[(size_t)e*8 + MJH_WARN_CONTACTFULL] + w[(size_t)e*8 + MJH_WARN_CNSTRFULL];
    uns += w[(size_t)e*8 + MJH_WARN_UNSUPPORTED];
  }
  if (ncapacity) *ncapacity = cap;
  if (nunsupported) *nunsupported = uns;
  return 0;
}

// ---- the drop-in: _unsafe_rollout / _unsafe_rollout_threaded contract ----------------------------------
// nbatch rollouts, possibly of DIFFERENT models of equal sizes (rollout.cc:100-118 uses m[r]):
//   * rollouts are grouped by model (pointer, then content); every group is a device batch;
//   * groups are sharded over the visible GPUs ($MJHIP_DEVICES caps their number): one host thread
//     and one cached (device model, batch) per GPU, each writing its own rows of the caller's
//     output arrays -- no exchange between GPUs;
//   * device models / batches are cached per GPU (a few most recently used models), staging
//     buffers are pooled per batch.
static_assert(mjNWARNING <= 8, "the batch keeps 8 warning counters per environment");

struct RollEntry {
  std::vector<char> sig;
  mjhipModel_* model = nullptr;
  mjhipBatch_* batch = nullptr;
  unsigned long long stamp = 0;
};
struct RollDevice { std::vector<RollEntry> cache; std::mutex mu; };   // mu: one rollout call at a time per GPU
struct RollJob {
  const mjModel* m = nullptr;
  std::vector<int> rows;       // rollout indices, ascending
  int device = 0;
  int rc = 0;
  int ncap = 0, nuns = 0;
  std::string err;
};
// Concurrency: concurrent mjhip_rollout calls serialise per GPU, not globally -- a call locks the
// devices it uses in ascending order (no deadlock) and two calls on disjoint devices overlap.  The
// device table has a fixed size so that its elements (and their mutexes) never move.
enum { MJH_MAX_DEVICES = 64 };
static RollDevice g_roll_dev[MJH_MAX_DEVICES];
static std::atomic<unsigned long long> g_roll_stamp{0};
// the caller's d[0] inputs, copied before any device thread starts: the thread that owns the last
// rollout writes d[0] at its end while the others may still be seeding their batches
struct RollSeed {
  std::vector<real> ctrl, qfrc_applied, xfrc_applied, mocap_pos, mocap_quat, userdata;
  std::vector<int> eq_active;
};

static bool same_model(const mjModel* a, const mjModel* b) {
  return a == b || (a->nbuffer == b->nbuffer && !memcmp(&a->opt, &b->opt, sizeof(a->opt)) &&
                    !memcmp(a->buffer, b->buffer, (size_t)a->nbuffer));
}

MJHIP_API void mjhip_rollout_clear_cache(void) {
  std::string err;
  const int ndev_ = std::min((int)MJH_MAX_DEVICES, std::max(0, Backend::device_count()));
  for (size_t dv = 0; dv < (size_t)ndev_; dv++) {
    std::lock_guard<std::mutex> lock(g_roll_dev[dv].mu);
    Backend::set_device((int)dv, &err);
    for (auto& c : g_roll_dev[dv].cache) {
      if (c.batch) mjhip_batch_destroy(c.batch);
      if (c.model) mjhip_model_destroy(c.model);
    }
    g_roll_dev[dv].cache.clear();
  }
}

// one job on its device (called from that device's host thread)
static void roll_run_job(RollJob& J, const RollSeed& seed0, struct mjData_* const* dp, int nbatch, int nstep, unsigned control_spec,
                         const double* state0, const double* warmstart0, const double* control,
                         double* state, double* sensordata) {
  auto fail = [&](int rc, const std::string& msg) { J.rc = rc; J.err = msg; };
  std::string err;
  if (!Backend::set_device(J.device, &err)) return fail(-3, err);
  RollDevice& RD = g_roll_dev[(size_t)J.device];
  const mjModel* m = J.m;
  const int n = (int)J.rows.size();
  // cached (model, batch) of this GPU
  RollEntry* E = nullptr;
  const size_t sigsz = (size_t)m->nbuffer + sizeof(m->opt);
  for (auto& c : RD.cache)
    if (c.sig.size() == sigsz && !memcmp(c.sig.data(), m->buffer, (size_t)m->nbuffer) &&
        !memcmp(c.sig.data() + m->nbuffer, &m->opt, sizeof(m->opt))) { E = &c; break; }
  if (!E) {
    size_t maxent = 4;
    if (const char* ev = getenv("MJHIP_CACHE_MODELS")) maxent = (size_t)std::max(1, atoi(ev));
    if (RD.cache.size() >= maxent) {
      size_t old = 0;
      for (size_t k = 1; k < RD.cache.size(); k++) if (RD.cache[k].stamp < RD.cache[old].stamp) old = k;
      if (RD.cache[old].batch) mjhip_batch_destroy(RD.cache[old].batch);
      if (RD.cache[old].model) mjhip_model_destroy(RD.cache[old].model);
      RD.cache.erase(RD.cache.begin() + (long)old);
    }
    RollEntry ne;
    ne.model = mjhip_model_create((const struct mjModel_*)m, 0, 0);
    if (!ne.model) return fail(-3, g_mjhip_err);
    ne.sig = ne.model->signature;
    RD.cache.push_back(ne);
    E = &RD.cache.back();
  }
  E->stamp = ++g_roll_stamp;
  if (!E->batch || E->batch->nenv < n) {
    if (E->batch) mjhip_batch_destroy(E->batch);
    // the drop-in always uses the environment-major layout (sensordata, partial launches)
    E->batch = mjhip_batch_create_layout(E->model, n, J.device, MJHIP_LAYOUT_AOS);
    if (!E->batch) return fail(-4, g_mjhip_err);
  }
  mjhipBatch_* Bt = E->batch;
  const DSizes& s = E->model->H.s;
  ControlLayout CL;
  if (!control_layout(s, control_spec, &CL, &err)) return fail(-2, err);
  const size_t nstate = (size_t)s.nstate, nsens = (size_t)s.nsensordata, ncontrol = (size_t)CL.n;

  // inputs of the control spec with no control array: the rollout steps with the caller's values
  // (the reference only clears inputs that are NOT in the spec, rollout.cc:85-115)
  mjData* d0 = (mjData*)dp[0];
  if (d0) {
    auto seed = [&](const char* name, const std::vector<real>& src, int cnt) -> bool {
      if (cnt <= 0 || (int)src.size() < cnt) return true;
      std::vector<real> rep((size_t)Bt->nenv*cnt);
      for (int e = 0; e < Bt->nenv; e++) memcpy(rep.data() + (size_t)e*cnt, src.data(), (size_t)cnt*sizeof(real));
      return mjhip_batch_set(Bt, name, rep.data()) == 0;
    };
    bool ok = true;
    if (!control) {
      if (control_spec & mjSTATE_CTRL) ok = ok && seed("ctrl", seed0.ctrl, s.nu);
      if (control_spec & mjSTATE_QFRC_APPLIED) ok = ok && seed("qfrc_applied", seed0.qfrc_applied, s.nv);
      if (control_spec & mjSTATE_XFRC_APPLIED) ok = ok && seed("xfrc_applied", seed0.xfrc_applied, 6*s.nbody);
      if (control_spec & mjSTATE_MOCAP_POS) ok = ok && seed("mocap_pos", seed0.mocap_pos, 3*s.nmocap);
      if (control_spec & mjSTATE_MOCAP_QUAT) ok = ok && seed("mocap_quat", seed0.mocap_quat, 4*s.nmocap);
      if ((control_spec & mjSTATE_EQ_ACTIVE) && s.neq > 0) {
        std::vector<int> rep((size_t)Bt->nenv*s.neq);
        for (int e = 0; e < Bt->nenv; e++) for (int k = 0; k < s.neq; k++) rep[(size_t)e*s.neq + k] = seed0.eq_active[(size_t)k];
        ok = ok && mjhip_batch_set(Bt, "eq_active", rep.data()) == 0;
      }
    }
    // userdata: the reference never clears it (rollout.cc:85-115 has no userdata branch), so unless a
    // control array supplies it every rollout steps with -- and d[0] ends with -- the caller's values;
    // the cached batch may hold an earlier call's
    if (!(control && (control_spec & mjSTATE_USERDATA))) ok = ok && seed("userdata", seed0.userdata, s.nuserdata);
    if (!ok) return fail(-5, g_mjhip_err);
  }

  // rows of the caller's arrays: a contiguous block is copied in place, anything else is packed
  bool contiguous = true;
  for (int k = 1; k < n; k++) if (J.rows[k] != J.rows[0] + k) contiguous = false;
  const size_t r0 = (size_t)J.rows[0];
  std::vector<double> p_state0, p_warm, p_control, p_state, p_sens;
  const double *in_state0 = state0 + r0*nstate, *in_warm = warmstart0 ? warmstart0 + r0*s.nv : nullptr;
  const double* in_control = control ? control + r0*(size_t)nstep*ncontrol : nullptr;
  double* out_state = state ? state + r0*(size_t)nstep*nstate : nullptr;
  double* out_sens = (sensordata && nsens) ? sensordata + r0*(size_t)nstep*nsens : nullptr;
  if (!contiguous) {
    auto pack = [&](std::vector<double>& dst, const double* src, size_t width) -> const double* {
      if (!src || !width) return nullptr;
      dst.resize((size_t)n*width);
      for (int k = 0; k < n; k++) memcpy(dst.data() + (size_t)k*width, src + (size_t)J.rows[k]*width, width*sizeof(double));
      return dst.data();
    };
    in_state0 = pack(p_state0, state0, nstate);
    in_warm = pack(p_warm, warmstart0, (size_t)s.nv);
    in_control = pack(p_control, control, (size_t)nstep*ncontrol);
    if (state) { p_state.resize((size_t)n*nstep*nstate); out_state = p_state.data(); }
    if (sensordata && nsens) { p_sens.resize((size_t)n*nstep*nsens); out_sens = p_sens.data(); }
  }
  int rc = rollout_impl(Bt, n, nstep, control_spec, in_state0, in_warm, in_control, out_state, out_sens, 0, nullptr);
  if (rc) return fail(rc, g_mjhip_err);
  if (!contiguous) {
    for (int k = 0; k < n; k++) {
      if (state) memcpy(state + (size_t)J.rows[k]*nstep*nstate, p_state.data() + (size_t)k*nstep*nstate, (size_t)nstep*nstate*sizeof(double));
      if (sensordata && nsens) memcpy(sensordata + (size_t)J.rows[k]*nstep*nsens, p_sens.data() + (size_t)k*nstep*nsens, (size_t)nstep*nsens*sizeof(double));
    }
  }
  if (mjhip_batch_trouble(Bt, n, &J.ncap, &J.nuns)) return fail(-6, g_mjhip_err);

  // d[0] <- last step of the LAST rollout (rollout.cc:73)
  if (J.rows.back() == nbatch - 1 && d0) {
    const int last = n - 1;
    bool ok = true;
    auto pull = [&](const char* name, void* dst, int cnt, size_t esz) {
      if (cnt <= 0 || !dst) return;
      void* p; int fc, isint;
      if (mjhip_batch_field(Bt, name, &p, &fc, &isint) || fc < cnt) { ok = false; return; }
      ok = Backend::d2h(dst, (const char*)p + (size_t)last*fc*esz, (size_t)cnt*esz, nullptr) && ok;
    };
    pull("time", &d0->time, 1, sizeof(real));
    pull("qpos", d0->qpos, s.nq, sizeof(real));
    pull("qvel", d0->qvel, s.nv, sizeof(real));
    pull("act", d0->act, s.na, sizeof(real));
    pull("ctrl", d0->ctrl, s.nu, sizeof(real));
    pull("qfrc_applied", d0->qfrc_applied, s.nv, sizeof(real));
    pull("xfrc_applied", d0->xfrc_applied, 6*s.nbody, sizeof(real));
    pull("mocap_pos", d0->mocap_pos, 3*s.nmocap, sizeof(real));
    pull("mocap_quat", d0->mocap_quat, 4*s.nmocap, sizeof(real));
    pull("userdata", d0->userdata, s.nuserdata, sizeof(real));
    pull("qacc_warmstart", d0->qacc_warmstart, s.nv, sizeof(real));
    pull("qacc", d0->qacc, s.nv, sizeof(real));
    pull("sensordata", d0->sensordata, s.nsensordata, sizeof(real));
    int w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<int> eqa((size_t)std::max(1, s.neq));
    pull("warning", w, 8, sizeof(int));
    pull("eq_active", eqa.data(), s.neq, sizeof(int));
    ok = Backend::sync(nullptr) && ok;
    if (!ok) return fail(-7, "mjhip_rollout: reading back the final state failed");
    for (int k = 0; k < mjNWARNING; k++) d0->warning[k].number = w[k];
    for (int k = 0; k < s.neq; k++) d0->eq_active[k] = (mjtByte)eqa[(size_t)k];
  }
}

MJHIP_API int mjhip_rollout(const struct mjModel_* const* mp, struct mjData_* const* dp, int nbatch,
                            int nstep, unsigned control_spec, const double* state0,
                            const double* warmstart0, const double* control, double* state,
                            double* sensordata) {
  if (!mp || !dp || nbatch <= 0 || nstep < 0 || !state0) { set_err("mjhip_rollout: bad arguments"); return -1; }
  const int ndev_all = Backend::device_count();
  if (ndev_all <= 0) { set_err("mjhip: no HIP device visible -- libmjhip has no CPU fallback"); return -3; }
  int ndev = ndev_all;
  if (const char* ev = getenv("MJHIP_DEVICES")) ndev = std::max(1, std::min(ndev_all, atoi(ev)));
  ndev = std::min(ndev, (int)MJH_MAX_DEVICES);
  const int home = Backend::current_device();

  // group the rollouts by model; sizes must agree (the caller's arrays have one row width)
  const mjModel* m0 = (const mjModel*)mp[0];
  std::vector<RollJob> groups;
  {
    const mjModel* prev = nullptr;
    int prev_g = -1;
    for (int r = 0; r < nbatch; r++) {
      const mjModel* mr = (const mjModel*)mp[r];
      if (!mr) { set_err("mjhip_rollout: null model"); return -1; }
      int g = -1;
      if (mr == prev) g = prev_g;
      else {
        if (mr->nq != m0->nq || mr->nv != m0->nv || mr->nu != m0->nu || mr->na != m0->na || mr->nbody != m0->nbody ||
            mr->nsensordata != m0->nsensordata || mr->nmocap != m0->nmocap || mr->neq != m0->neq ||
            mr->nuserdata != m0->nuserdata) {
          set_err("mjhip_rollout: the models of one call must have identical sizes");
          return -2;
        }
        for (size_t k = 0; k < groups.size() && g < 0; k++) if (same_model(groups[k].m, mr)) g = (int)k;
        if (g < 0) { groups.emplace_back(); groups.back().m = mr; g = (int)groups.size() - 1; }
      }
      groups[(size_t)g].rows.push_back(r);
      prev = mr; prev_g = g;
    }
  }
  // shard: a group large enough is cut into one contiguous piece per GPU, small groups go round-robin
  std::vector<RollJob> jobs;
  int rr = 0;
  for (auto& G : groups) {
    const int n = (int)G.rows.size();
    const int pieces = (ndev > 1 && n >= 2*ndev) ? ndev : 1;
    for (int k = 0; k < pieces; k++) {
      const int lo = (int)((long long)n*k/pieces), hi = (int)((long long)n*(k + 1)/pieces);
      RollJob J;
      J.m = G.m;
      J.rows.assign(G.rows.begin() + lo, G.rows.begin() + hi);
      J.device = pieces > 1 ? k : (rr++ % ndev);
      jobs.push_back(std::move(J));
    }
  }

  // snapshot of the caller's d[0] inputs (see RollSeed)
  RollSeed seed0;
  if (const mjData* d0 = (const mjData*)dp[0]) {
    auto snap = [](std::vector<real>& dst, const mjtNum* src, size_t cnt) { if (src && cnt) dst.assign(src, src + cnt); };
    snap(seed0.ctrl, d0->ctrl, (size_t)m0->nu);
    snap(seed0.qfrc_applied, d0->qfrc_applied, (size_t)m0->nv);
    snap(seed0.xfrc_applied, d0->xfrc_applied, 6*(size_t)m0->nbody);
    snap(seed0.mocap_pos, d0->mocap_pos, 3*(size_t)m0->nmocap);
    snap(seed0.mocap_quat, d0->mocap_quat, 4*(size_t)m0->nmocap);
    snap(seed0.userdata, d0->userdata, (size_t)m0->nuserdata);
    if (d0->eq_active) for (int k = 0; k < m0->neq; k++) seed0.eq_active.push_back(d0->eq_active[k]);
  }
  auto run_device = [&](int dev) {
    bool any = false;
    for (auto& J : jobs) if (J.device == dev) any = true;
    if (!any) return;
    std::lock_guard<std::mutex> lock(g_roll_dev[dev].mu);
    for (auto& J : jobs) if (J.device == dev)
      roll_run_job(J, seed0, dp, nbatch, nstep, control_spec, state0, warmstart0, control, state, sensordata);
  };
  int used = 0;
  for (int dv = 0; dv < ndev; dv++) for (auto& J : jobs) if (J.device == dv) { used++; break; }
  if (used <= 1) {
    for (int dv = 0; dv < ndev; dv++) run_device(dv);
  } else {
    std::vector<std::thread> th;
    for (int dv = 0; dv < ndev; dv++) th.emplace_back(run_device, dv);
    for (auto& t : th) t.join();
  }
  std::string err;
  Backend::set_device(home, &err);
  int ncap = 0, nuns = 0;
  for (auto& J : jobs) {
    if (J.rc) { set_err(J.err); return J.rc; }
    ncap += J.ncap; nuns += J.nuns;
  }
  // trajectories that are not what the reference would have computed are reported, not hidden:
  // positive return codes (the outputs are complete: such environments were frozen and back-filled)
  if (nuns) {
    set_err("mjhip_rollout: " + std::to_string(nuns) + " environment(s) reached a geom pair that has no GPU collider "
            "(mjhip warning slot 7) and were frozen");
    return 2;
  }
  if (ncap) {
    set_err("mjhip_rollout: " + std::to_string(ncap) + " contact / constraint capacity overflow(s); the affected "
            "environments were frozen -- raise $MJHIP_NCONMAX / $MJHIP_NEFCMAX");
    return 1;
  }
  return 0;
}

}  // extern "C"
