// libmjhip.so -- the product build of the mjhip C ABI (include/mjhip.h) for MI355X (gfx950).
//
// One HIP block == one 64-lane wavefront == one environment.  Kernels are thin __global__
// wrappers around the stage functions of mjh_smooth/collision/constraint/solver/step.h; the host
// runtime (model upload, batch arena, rollout marshalling) is mjh_runtime.h.
//
// Build (see __graft_entry__.build): every mjh_*.hip of this directory is compiled with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -c -I<mujoco include dir>
// (in parallel) and the objects are linked into libmjhip.so
// -ffp-contract=off is part of the numerical contract: the parity oracle is the reference engine
// built without FMA contraction, and the kernels reproduce its operation order.
#include <hip/hip_runtime.h>

#include <string>

// this translation unit: host runtime + the generic kernels (namespace wv: one wavefront per
// environment, every feature); the other mappings are compiled in mjh_kern_*.hip
#define MJH_BUILD_WV 1
#include "mjh_kernels.h"

MJH_DEFINE_WAVE_KERNELS(wv, 1, 4, 0)
MJH_DECLARE_WAVE_LAUNCHERS(wl)
MJH_DECLARE_WAVE_LAUNCHERS(wv2)     // mjh_kern_wide.hip: the generic kernels with a 256-VGPR budget
MJH_DECLARE_WAVE_LAUNCHERS(wn)      // mjh_kern_mw.hip: MJH_MW wavefronts per environment
extern "C" bool mjh_launch_forward_soa(const DModel* M, const DBatch* B, int nenv, int stages, int lds, void* stream);
extern "C" bool mjh_launch_smooth(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs* A, void* stream);
extern "C" bool mjh_launch_integrate(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs* A, void* stream);
extern "C" bool mjh_launch_lane_forward(const DModel* M, const DBatch* B, int nenv, int epw, int stages, void* stream);
extern "C" bool mjh_launch_lane_reset(const DModel* M, const DBatch* B, int nenv, int epw, void* stream);

// Launch order of the next rollout launch: counting sort of the environments by the work estimate
// of the last one (256 buckets, one workgroup), most expensive first, then dealt to the SIMDs in
// serpentine order.  Measured on MI355X (tools/wg_map.py): one-wavefront workgroup w of a launch lands on
// SIMD w mod nsimd (nsimd = 4 x CUs = 1024), so with every wave slot taken SIMD s steps workgroups s,
// s + nsimd, s + 2 nsimd, ...  Dealing rank r of block q to slot r (q even) or nsimd - 1 - r (q odd)
// gives every SIMD one environment of each cost quartile AND nearly equal sums; a SIMD is work
// conserving, so it finishes when the sum of its wavefronts' work is done.
__global__ __launch_bounds__(1024) void mjh_k_balance(const DBatch* __restrict__ B, int nsimd, int mode, int nstep) {
  __shared__ int hist[256];
  __shared__ int maxc;
  const int n = B->nenv, tid = (int)threadIdx.x;
  // (mode bit 2: deal by the MEASURED wall time of the previous launch instead of the work estimate -- $MJHIP_BALANCE_COST=wall)
  const int* cost = (mode & 4) ? B->wall : B->cost;
  mode &= 3;
  int* perm = B->perm;
  if (tid < 256) hist[tid] = 0;
  if (tid == 0) maxc = 1;
  __syncthreads();
  int m = 1;
  for (int e = tid; e < n; e += 1024) m = max(m, cost[e]);
  atomicMax(&maxc, m);
  __syncthreads();
  // (mean cost per environment of the launch just finished: the reference level of the next launch's issue priorities)
  {
    __shared__ long long tot;
    if (tid == 0) tot = 0;
    __syncthreads();
    long long part = 0;
    for (int e = tid; e < n; e += 1024) part += B->cost[e];
    atomicAdd((unsigned long long*)&tot, (unsigned long long)part);
    __syncthreads();
    // (both words written HERE, after the launch they describe: a wavefront of the launch itself writing the step count
    //  raced with later-dispatched wavefronts of the same launch reading it)
    if (tid == 0) { B->prio_ref[0] = (int)(tot/(n > 0 ? n : 1)); B->prio_ref[1] = nstep; }
  }
  const float scale = 255.0f / (float)maxc;
  for (int e = tid; e < n; e += 1024) atomicAdd(&hist[255 - (int)(cost[e] * scale)], 1);
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int b = 0; b < 256; b++) { int c = hist[b]; hist[b] = acc; acc += c; } }
  __syncthreads();
  for (int e = tid; e < n; e += 1024) {
    const int rank = atomicAdd(&hist[255 - (int)(cost[e] * scale)], 1);
    const int q = rank / nsimd, r = rank - q*nsimd;
    int w = rank;
    if (mode == 2 && n == 4*nsimd) {
      // the nsimd heaviest get a SIMD each, and the three lightest of the rest as company
      if (rank >= nsimd) { const int u = n - 1 - rank; w = (u % 3 + 1)*nsimd + u / 3; }
    } else if ((q & 1) && (q + 1)*nsimd <= n) w = q*nsimd + (nsimd - 1 - r);      // full odd blocks run backwards
    perm[w] = e;
  }
}

__global__ __launch_bounds__(MJH_WAVE) void mjh_k_reset(const DModel* __restrict__ M, const DBatch* __restrict__ B) {
  wv::reset_env(wv_const_ref(M), wv_const_ref(B), (int)blockIdx.x);
}

struct Backend {
  static const char* name() { return "hip-gfx950"; }
  static int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
  }
  static int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return 0;
    return d;
  }
  static bool set_device(int dev, std::string* err) {
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) { *err = std::string("mjhip: hipSetDevice failed: ") + hipGetErrorString(e); return false; }
    return true;
  }
  static void* alloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) return nullptr;
    return p;
  }
  static void free(void* p) { (void)hipFree(p); }
  static bool h2d(void* dst, const void* src, size_t n, void* stream) {
    return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess;
  }
  static bool d2h(void* dst, const void* src, size_t n, void* stream) {
    return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess;
  }
  // `height` rows of `width` bytes between arrays of different row pitch (kind: 0 host->device, 1 device->host)
  static bool copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int kind, void* stream) {
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice,
                            (hipStream_t)stream) == hipSuccess;
  }
  // a second stream for copies that overlap the rollout kernel, and "stream b waits for what stream a has queued so far"
  static void* stream_create() { hipStream_t st = nullptr; return hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess ? (void*)st : nullptr; }
  static void stream_destroy(void* st) { if (st) (void)hipStreamDestroy((hipStream_t)st); }
  static bool stream_follow(void* b, void* a) {
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
    bool ok = hipEventRecord(ev, (hipStream_t)a) == hipSuccess && hipStreamWaitEvent((hipStream_t)b, ev, 0) == hipSuccess;
    (void)hipEventDestroy(ev);
    return ok;
  }
  static bool zero(void* dst, size_t n, void* stream) {
    return hipMemsetAsync(dst, 0, n, (hipStream_t)stream) == hipSuccess;
  }
  static bool sync(void* stream) { return hipStreamSynchronize((hipStream_t)stream) == hipSuccess; }
  // largest dynamic LDS block a workgroup may request: what the device reports per block, at most the CU's 160 KB,
  // at least the 64 KB every kernel gets without opting in (the launchers raise the per-kernel limit above that:
  // mjh_raise_lds); $MJHIP_MAX_LDS overrides (A/B runs)
  static int max_lds() {
    static const int v = [] {
      if (const char* ev = getenv("MJHIP_MAX_LDS")) return atoi(ev);
      int dev = 0, b = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&b, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) b = 0;
      if (b > 160*1024) b = 160*1024;
      if (b < 64*1024) b = 64*1024;
      return b;
    }();
    return v;
  }
  // compute units of the current device (256 on MI355X)
  static int num_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
  }
  // generic kernels: a launch that cannot put more than two wavefronts on a SIMD takes the 256-VGPR build
  // ($MJHIP_WIDE_REGS=0/1 forces the choice: A/B runs)
  static bool wide_regs(int nenv) {
    static const int force = [] { const char* ev = getenv("MJHIP_WIDE_REGS"); return ev ? atoi(ev) : -1; }();
    if (force >= 0) return force != 0;
    return nenv <= 2*4*num_cus();
  }
  // `variant` (MJH_VAR_*, mjh_modes.h): which mapping of the stage sources steps the batch; lds = LDS
  // bytes per ENVIRONMENT (a workgroup of a sub-wave variant allocates one block per lane group)
  static bool launch_forward(const DModel* M, const DBatch* B, int nenv, int stages, int lds, int soa, int variant, void* stream) {
    if (soa) return mjh_launch_forward_soa(M, B, nenv, stages, lds, stream);
    switch (variant) {
      case MJH_VAR_LEAN: return mjh_launch_forward_wl(M, B, nenv, stages, lds, stream);
      case MJH_VAR_MULTIWAVE: return mjh_launch_forward_wn(M, B, nenv, stages, lds, stream);
      default: return wide_regs(nenv) ? mjh_launch_forward_wv2(M, B, nenv, stages, lds, stream)
                                      : mjh_launch_forward_wv(M, B, nenv, stages, lds, stream);
    }
  }
  static bool launch_rollout(const DModel* M, const DBatch* B, int nenv, const RolloutArgs& A, int lds, int variant, void* stream) {
    switch (variant) {
      case MJH_VAR_LEAN: return mjh_launch_rollout_wl(M, B, nenv, &A, lds, stream);
      case MJH_VAR_MULTIWAVE: return mjh_launch_rollout_wn(M, B, nenv, &A, lds, stream);
      default: return wide_regs(nenv) ? mjh_launch_rollout_wv2(M, B, nenv, &A, lds, stream)
                                      : mjh_launch_rollout_wv(M, B, nenv, &A, lds, stream);
    }
  }
  static const char* rollout_kernel_name(int variant, int nenv) {
    return variant == MJH_VAR_LEAN ? "mjh_k_rollout_wl" : variant == MJH_VAR_MULTIWAVE ? "mjh_k_rollout_wn" :
           wide_regs(nenv) ? "mjh_k_rollout_wv2" : "mjh_k_rollout_wv";
  }
  static bool launch_balance(const DBatch* B, int nenv, int nstep, void* stream) {
    (void)nenv;
    // (function-local statics: initialised once, thread-safely -- the per-GPU host threads of mjhip_rollout all land here)
    struct Cfg { int nsimd, mode; };
    static const Cfg cfg = [] {
      Cfg c{0, 1};
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
      c.nsimd = 4*cus;
      if (const char* ev = getenv("MJHIP_BALANCE_SNAKE")) { c.mode = atoi(ev); if (c.mode == 0) c.nsimd = 1 << 30; }     // A/B: 0 plain descending order, 2 heaviest + three lightest
      if (const char* ev = getenv("MJHIP_BALANCE_COST")) { if (ev[0] == 'w') c.mode |= 4; }
      return c;
    }();
    const int nsimd = cfg.nsimd, mode = cfg.mode;
    hipLaunchKernelGGL(mjh_k_balance, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, nsimd, mode, nstep);
    return hipGetLastError() == hipSuccess;
  }
  static bool launch_smooth(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs& A, void* stream) {
    return mjh_launch_smooth(M, B, nenv, epw, &A, stream);
  }
  static bool launch_integrate(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs& A, void* stream) {
    return mjh_launch_integrate(M, B, nenv, epw, &A, stream);
  }
  static bool launch_lane_forward(const DModel* M, const DBatch* B, int nenv, int epw, int stages, void* stream) {
    return mjh_launch_lane_forward(M, B, nenv, epw, stages, stream);
  }
  static bool launch_lane_reset(const DModel* M, const DBatch* B, int nenv, int epw, void* stream) {
    return mjh_launch_lane_reset(M, B, nenv, epw, stream);
  }
  static bool launch_reset(const DModel* M, const DBatch* B, int nenv, void* stream) {
    hipLaunchKernelGGL(mjh_k_reset, dim3(nenv), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B);
    return hipGetLastError() == hipSuccess;
  }
};

#include "mjh_runtime.h"
