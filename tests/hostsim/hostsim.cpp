// TEST INFRASTRUCTURE -- not part of the product, never loaded by mujoco_amd.
//
// Builds the SAME kernel sources as libmjhip.so (mujoco_amd/csrc/mjh_*.h) for the host with
// -DMJH_HOSTSIM: each "wavefront" is emulated by 64 cooperatively scheduled ucontext fibers that
// switch at every wv_sync()/cross-lane primitive (mjh_spmd.h).  It exports the C ABI of
// include/mjhip.h from libmjhip_hostsim.so so the parity tests can drive the kernel logic against
// the oracle in a container without a GPU.  Running the lanes in reverse order (env var
// MJH_HOSTSIM_REVERSE=1) must give bit-identical results: that is the race detector.
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>

#include "../../mujoco_amd/csrc/mjh_spmd.h"

#include "../../mujoco_amd/csrc/mjh_math.h"
#include "../../mujoco_amd/csrc/mjh_types.h"

namespace mjhsim {
thread_local WaveSim* g_wave = nullptr;
thread_local char g_lds[MJH_LDS_MAX];     // the emulated workgroup's LDS block
}

#include "../../mujoco_amd/csrc/mjh_modes.h"

namespace {

constexpr size_t kStack = 256 * 1024;

struct Runner {
  mjhsim::WaveSim w;
  std::function<void()> body;
  Runner() {
    w.stacks = (char*)malloc(kStack * MJH_WAVE);
    w.reverse = getenv("MJH_HOSTSIM_REVERSE") && atoi(getenv("MJH_HOSTSIM_REVERSE"));
  }
  ~Runner() { free(w.stacks); }
  static void trampoline(unsigned lo, unsigned hi) {
    Runner* r = (Runner*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    r->body();
    r->w.done[r->w.cur] = 1;
    // returning switches to uc_link (the scheduler)
  }
  void run(int env, const std::function<void()>& fn) {
    body = fn;
    w.env = env;
    mjhsim::g_wave = &w;
    for (int l = 0; l < MJH_WAVE; l++) {
      w.done[l] = 0;
      getcontext(&w.ctx[l]);
      w.ctx[l].uc_stack.ss_sp = w.stacks + kStack * l;
      w.ctx[l].uc_stack.ss_size = kStack;
      w.ctx[l].uc_link = &w.sched;
      uintptr_t p = (uintptr_t)this;
      makecontext(&w.ctx[l], (void (*)())trampoline, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
    }
    int remaining = MJH_WAVE;
    while (remaining) {
      remaining = 0;
      for (int k = 0; k < MJH_WAVE; k++) {
        int l = w.reverse ? MJH_WAVE - 1 - k : k;
        if (w.done[l]) continue;
        w.cur = l;
        swapcontext(&w.sched, &w.ctx[l]);
        if (!w.done[l]) remaining++;
      }
    }
  }
};

thread_local Runner* g_runner = nullptr;
Runner* runner() {
  if (!g_runner) g_runner = new Runner();
  return g_runner;
}

}  // namespace

struct Backend {
  static const char* name() { return "hostsim"; }
  static int device_count() { return 1; }
  static bool set_device(int, std::string*) { return true; }
  static void* alloc(size_t bytes) { void* p = nullptr; if (posix_memalign(&p, 256, bytes ? bytes : 256)) return nullptr; return p; }
  static void free(void* p) { ::free(p); }
  static bool h2d(void* dst, const void* src, size_t n, void*) { memcpy(dst, src, n); return true; }
  static bool d2h(void* dst, const void* src, size_t n, void*) { memcpy(dst, src, n); return true; }
  static bool zero(void* dst, size_t n, void*) { memset(dst, 0, n); return true; }
  static bool sync(void*) { return true; }
  static int max_lds() { return 64 * 1024; }
  // a fresh workgroup sees garbage in LDS: poison it so stale-data bugs cannot hide
  static void poison_lds(int lds) { memset(mjhsim::g_lds, 0xff, lds > 0 ? (size_t)lds : 0); }
  static bool launch_forward(const DModel* M, const DBatch* B, int nenv, int stages, int lds, int soa, void*) {
    for (int e = 0; e < nenv; e++) {
      poison_lds(lds);
      if (soa) runner()->run(e, [&]() { ws::forward_or_euler(*M, *B, wv_env(), stages); });
      else runner()->run(e, [&]() { wv::forward_or_euler(*M, *B, wv_env(), stages); });
    }
    return true;
  }
  static bool launch_rollout(const DModel* M, const DBatch* B, int nenv, const RolloutArgs& A, int lds, void*) {
    // (launch order: the emulation runs the permutation back to front to show results do not depend on it)
    for (int w = nenv - 1; w >= 0; w--) { poison_lds(lds); runner()->run(B->perm[w], [&]() { wv::rollout_env(*M, *B, wv_env(), A); }); }
    return true;
  }
  // lane mode needs no wavefront emulation: every environment is an ordinary serial call
  static bool launch_smooth(const DModel* M, const DBatch* B, int nenv, int, const RolloutArgs& A, void*) {
    for (int e = 0; e < nenv; e++) ln::smooth_env(*M, *B, e, A);
    return true;
  }
  static bool launch_integrate(const DModel* M, const DBatch* B, int nenv, int, const RolloutArgs& A, void*) {
    for (int e = 0; e < nenv; e++) ln::integrate_env(*M, *B, e, A);
    return true;
  }
  static bool launch_lane_forward(const DModel* M, const DBatch* B, int nenv, int, int stages, void*) {
    for (int e = 0; e < nenv; e++) ln::forward_or_euler(*M, *B, e, stages);
    return true;
  }
  static bool launch_lane_reset(const DModel* M, const DBatch* B, int nenv, int, void*) {
    for (int e = 0; e < nenv; e++) ln::reset_env(*M, *B, e);
    return true;
  }
  static bool launch_reset(const DModel* M, const DBatch* B, int nenv, void*) {
    for (int e = 0; e < nenv; e++) runner()->run(e, [&]() { wv::reset_env(*M, *B, wv_env()); });
    return true;
  }
};

#include "../../mujoco_amd/csrc/mjh_runtime.h"
