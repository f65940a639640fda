// TEST INFRASTRUCTURE -- not part of the product, never loaded by mujoco_amd.
//
// Builds the SAME kernel sources as libmjhip.so (mujoco_amd/csrc/mjh_*.h) for the host with
// -DMJH_HOSTSIM: each "wavefront" is emulated by 64 cooperatively scheduled ucontext fibers that
// switch at every wv_sync()/cross-lane primitive (mjh_spmd.h).  It exports the C ABI of
// include/mjhip.h from libmjhip_hostsim.so so the parity tests can drive the kernel logic against
// the oracle in a container without a GPU.  Running the lanes in reverse order (env var
// MJH_HOSTSIM_REVERSE=1) must give bit-identical results: that is the race detector.

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include <sys/mman.h>
#include <unistd.h>
#include <algorithm>

#include "../../mujoco_amd/csrc/mjh_spmd.h"

#include "../../mujoco_amd/csrc/mjh_math.h"
#include "../../mujoco_amd/csrc/mjh_types.h"

namespace mjhsim {
thread_local WaveSim* g_wave = nullptr;
thread_local char* g_lds = nullptr;       // the emulated workgroup's LDS block (set per launch: lds_block)
}

#include "../../mujoco_amd/csrc/mjh_modes.h"

namespace {

constexpr size_t kStack = 256 * 1024;

struct Runner {
  mjhsim::WaveSim w;
  std::function<void()> body;
  Runner() {
    w.stacks = (char*)malloc(kStack * MJH_WAVE * MJH_MW);
    w.reverse = getenv("MJH_HOSTSIM_REVERSE") && atoi(getenv("MJH_HOSTSIM_REVERSE"));
  }
  ~Runner() { free(w.stacks); }
  // first frame of every lane fiber: runs the kernel body, marks the lane done and hands control
  // back to the scheduler for good
  static void fiber_entry();
  void run(int env, const std::function<void()>& fn, int nfib = MJH_WAVE) {
    body = fn;
    w.env = env;
    w.nfib = nfib;
    mjhsim::g_wave = &w;
    unsigned csr[2] = {0, 0};
    asm volatile("stmxcsr %0\n\tfnstcw %1" : "=m"(csr[0]), "=m"(csr[1]));
    for (int l = 0; l < MJH_WAVE*MJH_MW; l++) w.done[l] = 1;
    for (int l = 0; l < nfib; l++) {
      w.done[l] = 0;
      w.parked[l] = 0;
      w.arrive_all[l] = 0;
      w.arrive_row[l] = 0;
      if (l < MJH_WAVE) w.arrive_wave[l] = 0;
      // initial frame popped by mjh_ctx_switch: [mxcsr|x87cw] r15 r14 r13 r12 rbx rbp, return address
      uintptr_t top = ((uintptr_t)(w.stacks + kStack * (l + 1))) & ~(uintptr_t)15;
      uintptr_t* sp = (uintptr_t*)top;
      *--sp = 0;                                  // the entry function's (never used) return address
      *--sp = (uintptr_t)&Runner::fiber_entry;
      for (int k = 0; k < 6; k++) *--sp = 0;
      *--sp = (uintptr_t)csr[0] | ((uintptr_t)csr[1] << 32);
      w.ctx_sp[l] = sp;
    }
    w.round = 0; w.wave_done = w.all_done = 0;
    for (int r = 0; r < MJH_WAVE*MJH_MW/16; r++) w.row_done[r] = 0;
    int remaining = nfib;
    while (remaining) {
      remaining = 0;
      w.round++;
      for (int k = 0; k < nfib; k++) {
        int l = w.reverse ? nfib - 1 - k : k;
        if (w.done[l]) continue;
        if (w.parked[l]) { remaining++; continue; }
        w.cur = l;
        mjhsim::mjh_ctx_switch(&w.sched_sp, w.ctx_sp[l]);
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

void Runner::fiber_entry() {
  Runner* r = g_runner;
  r->body();
  r->w.done[r->w.cur] = 1;
  mjhsim::mjh_ctx_switch(&r->w.ctx_sp[r->w.cur], r->w.sched_sp);
  abort();     // a finished lane is never resumed
}

}  // namespace

struct Backend {
  static const char* name() { return "hostsim"; }
  // ($MJH_HOSTSIM_DEVICES > 1 lets the CPU tests drive the multi-GPU sharding of mjhip_rollout)
  static int device_count() { const char* ev = getenv("MJH_HOSTSIM_DEVICES"); return ev ? std::max(1, atoi(ev)) : 1; }
  static int& cur_dev() { static thread_local int d = 0; return d; }
  static int current_device() { return cur_dev(); }
  static bool set_device(int d, std::string* err) {
    if (d < 0 || d >= device_count()) { *err = "hostsim: no such device"; return false; }
    cur_dev() = d;
    return true;
  }
  static void* alloc(size_t bytes) { void* p = nullptr; if (posix_memalign(&p, 256, bytes ? bytes : 256)) return nullptr; return p; }
  static void free(void* p) { ::free(p); }
  static bool h2d(void* dst, const void* src, size_t n, void*) { memcpy(dst, src, n); return true; }
  static bool d2h(void* dst, const void* src, size_t n, void*) { memcpy(dst, src, n); return true; }
  static bool copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int, void*) {
    for (size_t r = 0; r < height; r++) memcpy((char*)dst + r*dpitch, (const char*)src + r*spitch, width);
    return true;
  }
  static void* stream_create() { return (void*)1; }      // (no streams on the host: a token, so that the chunked path runs)
  static void stream_destroy(void*) {}
  static bool stream_follow(void*, void*) { return true; }
  static bool zero(void* dst, size_t n, void*) { memset(dst, 0, n); return true; }
  static bool sync(void*) { return true; }
  static int max_lds() { return 160 * 1024; }   // (gfx950: a workgroup may take the whole CU block)
  static int num_cus() { return 256; }
  // a fresh workgroup sees garbage in LDS: poison it so stale-data bugs cannot hide
  // The block of a launch that allocates `lds` bytes ends (up to alignment) at a PROT_NONE page, and the alignment slack
  // in between holds a canary that is checked when the next launch re-uses the region: a flat access beyond a
  // workgroup's LDS allocation raises HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on the device (profiles/r04/
  // negative_results.txt #6 met one that the emulation, with its fixed 160 KB array, let through).
  struct LdsRegion { char* base = nullptr; char* guard = nullptr; size_t slack = 0; };
  static LdsRegion& lds_region() {
    static thread_local LdsRegion R;
    if (!R.base) {
      const size_t page = (size_t)sysconf(_SC_PAGESIZE);
      const size_t bytes = ((size_t)MJH_LDS_MAX + 512 + page - 1)/page*page;
      const size_t guard_bytes = ((size_t)2*MJH_LDS_MAX + page - 1)/page*page;      // (any offset a 160 KB plan could produce)
      R.base = (char*)mmap(nullptr, bytes + guard_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (R.base == (char*)MAP_FAILED) { perror("hostsim: mmap"); abort(); }
      R.guard = R.base + bytes;
      if (mprotect(R.guard, guard_bytes, PROT_NONE)) { perror("hostsim: mprotect"); abort(); }
    }
    return R;
  }
  static void poison_lds(int lds) {
    LdsRegion& R = lds_region();
    if (mjhsim::g_lds && R.slack) {
      for (size_t i = 0; i < R.slack; i++)
        if ((unsigned char)R.guard[-(long)R.slack + (long)i] != 0xA5) { fprintf(stderr, "hostsim: write beyond the LDS allocation (slack byte %zu)\n", i); abort(); }
    }
    const size_t n = lds > 0 ? (size_t)lds : 0;
    const size_t aligned = (n + 255)/256*256;
    mjhsim::g_lds = R.guard - aligned;
    R.slack = aligned - n;
    memset(mjhsim::g_lds, 0xff, n);
    memset(mjhsim::g_lds + n, 0xA5, R.slack);
  }
  // run NS's body for the nsub environments of every emulated wavefront (lanes of group g step
  // environment wave*nsub + g; a group past the end of the batch exits at once, like the kernels)
  // multi-wavefront workgroups (mjh_modes.h: wn + wq): fibers 0..63 are wave 0, the others the helper wavefronts
  template <class F>
  static void run_multiwave(int nenv, int lds, const DModel* M, const DBatch* B, F main_body) {
    for (int w = nenv - 1; w >= 0; w--) {
      poison_lds(lds + MJH_MW_LDS_TAIL);
      runner()->run(w, [&]() {
        if (mjhsim::lane() >= MJH_WAVE) { mw_helper_loop(*M, *B); return; }
        main_body(wv_env());
        mw_release_helpers(*B);
      }, MJH_WAVE*MJH_MW);
    }
  }
  template <class F>
  static void run_waves(int nenv, int nsub, int lds, F body) {
    for (int w = (nenv + nsub - 1)/nsub - 1; w >= 0; w--) {     // (back to front: order must not matter)
      poison_lds(lds*nsub);
      runner()->run(w, [&]() {
        const int e = wv_env()*nsub + mjhsim::lane()/(MJH_WAVE/nsub);
        if (e < nenv) body(e);
      });
    }
  }
  static bool launch_forward(const DModel* M, const DBatch* B, int nenv, int stages, int lds, int soa, int variant, void*) {
    if (soa) run_waves(nenv, 1, lds, [&](int e) { ws::forward_or_euler(*M, *B, e, stages); });
    else if (variant == MJH_VAR_LEAN) run_waves(nenv, 1, lds, [&](int e) { wl::forward_or_euler(*M, *B, e, stages); });
    else if (variant == MJH_VAR_MULTIWAVE) run_multiwave(nenv, lds, M, B, [&](int e) { wn::forward_or_euler(*M, *B, e, stages); });
    else run_waves(nenv, 1, lds, [&](int e) { wv::forward_or_euler(*M, *B, e, stages); });
    return true;
  }
  static bool launch_rollout(const DModel* M, const DBatch* B, int nenv, const RolloutArgs& A, int lds, int variant, void*) {
    // (slot w of the launch steps environment perm[w], like the kernels)
    if (variant == MJH_VAR_LEAN) run_waves(nenv, 1, lds, [&](int w) { wl::rollout_env(*M, *B, A.nlaunch ? w : B->perm[w], A); });
    else if (variant == MJH_VAR_MULTIWAVE) run_multiwave(nenv, lds, M, B, [&](int w) { wn::rollout_env(*M, *B, A.nlaunch ? w : B->perm[w], A); });
    else run_waves(nenv, 1, lds, [&](int w) { wv::rollout_env(*M, *B, A.nlaunch ? w : B->perm[w], A); });
    return true;
  }
  // the launch order of the next rollout launch: environments by decreasing work estimate
  static const char* rollout_kernel_name(int variant, int) { return variant == MJH_VAR_LEAN ? "hostsim:wl" : variant == MJH_VAR_MULTIWAVE ? "hostsim:wn" : "hostsim:wv"; }
  static bool launch_balance(const DBatch* B, int nenv, int, void*) {
    std::vector<int> idx(nenv);
    for (int i = 0; i < nenv; i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return B->cost[a] > B->cost[b]; });
    for (int i = 0; i < nenv; i++) B->perm[i] = idx[i];
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

// register-only context switch (x86-64 SysV): callee-saved integer registers, MXCSR and the x87
// control word live on the outgoing stack; glibc's swapcontext would add two sigprocmask system
// calls per switch, which dominated the emulation's run time
asm(R"(
.text
.globl mjh_ctx_switch
.type mjh_ctx_switch,@function
mjh_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size mjh_ctx_switch, .-mjh_ctx_switch
)");

// test hook: the device's sine / cosine routine (mjh_math.h) compiled for the host -- its explicit fma
// calls are correctly rounded on both sides, so this is bit for bit what the GPU evaluates
extern "C" __attribute__((visibility("default"))) void mjh_test_sincos(int n, const double* x, double* sn, double* cs) {
  for (int i = 0; i < n; i++) mjh_sincos(x[i], sn + i, cs + i);
}
// likewise the device's atan2 and exp
extern "C" __attribute__((visibility("default"))) void mjh_test_atan2_exp(int n, const double* y, const double* x, double* at, double* ex) {
  for (int i = 0; i < n; i++) { at[i] = mjh_atan2(y[i], x[i]); ex[i] = mjh_exp(x[i]); }
}
// test hook for the emulation's LDS bounds: a launch that allocates `lds` bytes, then one access at byte `offset` of the
// block.  Beyond the allocation (rounded up to 256) the read faults like a flat access beyond a workgroup's LDS
// allocation does on the device (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION); a write into the rounding slack is caught
// by the canary check of the next launch.
extern "C" __attribute__((visibility("default"))) int mjh_test_lds_probe(int lds, int offset, int write) {
  Backend::poison_lds(lds);
  volatile char* p = mjh_lds() + offset;
  if (write) *p = 1;
  const int v = *p;
  Backend::poison_lds(lds);          // (checks the canary the first launch left)
  return v;
}
