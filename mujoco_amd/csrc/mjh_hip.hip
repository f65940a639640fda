// libmjhip.so -- the product build of the mjhip C ABI (include/mjhip.h) for MI355X (gfx950).
//
// One HIP block == one 64-lane wavefront == one environment.  Kernels are thin __global__
// wrappers around the stage functions of mjh_smooth/collision/constraint/solver/step.h; the host
// runtime (model upload, batch arena, rollout marshalling) is mjh_runtime.h.
//
// Build (see __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared \
//         -I<mujoco include dir> mjh_hip.hip -o libmjhip.so
// -ffp-contract=off is part of the numerical contract: the parity oracle is the reference engine
// built without FMA contraction, and the kernels reproduce its operation order.
#include <hip/hip_runtime.h>

#include <string>

#include "mjh_modes.h"

// The model / batch descriptors (tables of device pointers, ~1 KB each) live in device memory and
// are read through the scalar cache on demand; passing them by value made the compiler hoist
// every pointer into SGPRs for the whole kernel (hundreds of spills).
__global__ __launch_bounds__(MJH_WAVE) __attribute__((amdgpu_waves_per_eu(MJH_WAVES_PER_EU, MJH_WAVES_PER_EU))) void mjh_k_forward(const DModel* __restrict__ M,
                                                          const DBatch* __restrict__ B, int stages) {
  wv::forward_or_euler(wv_const_ref(M), wv_const_ref(B), (int)blockIdx.x, stages);
}

// the same stage-masked kernel for SoA batches (strided views): constraint kernel of the pipeline
__global__ __launch_bounds__(MJH_WAVE) __attribute__((amdgpu_waves_per_eu(MJH_WAVES_PER_EU, MJH_WAVES_PER_EU))) void mjh_k_forward_soa(const DModel* __restrict__ M,
                                                          const DBatch* __restrict__ B, int stages) {
  ws::forward_or_euler(wv_const_ref(M), wv_const_ref(B), (int)blockIdx.x, stages);
}

__global__ __launch_bounds__(MJH_WAVE) __attribute__((amdgpu_waves_per_eu(MJH_WAVES_PER_EU, MJH_WAVES_PER_EU))) void mjh_k_rollout(const DModel* __restrict__ M,
                                                          const DBatch* __restrict__ B, RolloutArgs A) {
  // workgroups are dispatched in blockIdx order: perm lists the environments by decreasing cost
  wv::rollout_env(wv_const_ref(M), wv_const_ref(B), B->perm[blockIdx.x], A);
}

// Longest-processing-time-first launch order for the next rollout launch: counting sort of the
// environments by the cost measured in the last one (256 buckets, one workgroup).
__global__ __launch_bounds__(1024) void mjh_k_balance(const DBatch* __restrict__ B) {
  __shared__ int hist[256];
  __shared__ int maxc;
  const int n = B->nenv, tid = (int)threadIdx.x;
  const int* cost = B->cost;
  int* perm = B->perm;
  if (tid < 256) hist[tid] = 0;
  if (tid == 0) maxc = 1;
  __syncthreads();
  int m = 1;
  for (int e = tid; e < n; e += 1024) m = max(m, cost[e]);
  atomicMax(&maxc, m);
  __syncthreads();
  const float scale = 255.0f / (float)maxc;
  for (int e = tid; e < n; e += 1024) atomicAdd(&hist[255 - (int)(cost[e] * scale)], 1);
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int b = 0; b < 256; b++) { int c = hist[b]; hist[b] = acc; acc += c; } }
  __syncthreads();
  for (int e = tid; e < n; e += 1024) perm[atomicAdd(&hist[255 - (int)(cost[e] * scale)], 1)] = e;
}

__global__ __launch_bounds__(MJH_WAVE) __attribute__((amdgpu_waves_per_eu(MJH_WAVES_PER_EU, MJH_WAVES_PER_EU))) void mjh_k_reset(const DModel* __restrict__ M,
                                                        const DBatch* __restrict__ B) {
  wv::reset_env(wv_const_ref(M), wv_const_ref(B), (int)blockIdx.x);
}

// ---- lane-mode kernels of the SoA pipeline: one lane per environment, epw environments per wavefront
#define MJH_LANE_KERNEL __global__ __launch_bounds__(MJH_WAVE)
#define MJH_LANE_ENV() const int lane_ = (int)threadIdx.x; if (lane_ >= epw) return; \
                       const int e = (int)blockIdx.x * epw + lane_; if (e >= B->nenv) return;

MJH_LANE_KERNEL void mjh_k_smooth(const DModel* __restrict__ M, const DBatch* __restrict__ B, RolloutArgs A, int epw) {
  MJH_LANE_ENV();
  ln::smooth_env(wv_const_ref(M), wv_const_ref(B), e, A);
}
MJH_LANE_KERNEL void mjh_k_integrate(const DModel* __restrict__ M, const DBatch* __restrict__ B, RolloutArgs A, int epw) {
  MJH_LANE_ENV();
  ln::integrate_env(wv_const_ref(M), wv_const_ref(B), e, A);
}
MJH_LANE_KERNEL void mjh_k_lane_forward(const DModel* __restrict__ M, const DBatch* __restrict__ B, int stages, int epw) {
  MJH_LANE_ENV();
  ln::forward_or_euler(wv_const_ref(M), wv_const_ref(B), e, stages);
}
MJH_LANE_KERNEL void mjh_k_lane_reset(const DModel* __restrict__ M, const DBatch* __restrict__ B, int epw) {
  MJH_LANE_ENV();
  ln::reset_env(wv_const_ref(M), wv_const_ref(B), e);
}

struct Backend {
  static const char* name() { return "hip-gfx950"; }
  static int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
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
  static bool zero(void* dst, size_t n, void* stream) {
    return hipMemsetAsync(dst, 0, n, (hipStream_t)stream) == hipSuccess;
  }
  static bool sync(void* stream) { return hipStreamSynchronize((hipStream_t)stream) == hipSuccess; }
  // largest dynamic LDS block a workgroup may request (gfx950: 160 KB per CU; one workgroup is
  // allowed 64 KB without opting in, which is already far beyond the useful range here)
  static int max_lds() { return 64 * 1024; }
  static bool launch_forward(const DModel* M, const DBatch* B, int nenv, int stages, int lds, int soa, void* stream) {
    if (soa) hipLaunchKernelGGL(mjh_k_forward_soa, dim3(nenv), dim3(MJH_WAVE), (size_t)lds, (hipStream_t)stream, M, B, stages);
    else hipLaunchKernelGGL(mjh_k_forward, dim3(nenv), dim3(MJH_WAVE), (size_t)lds, (hipStream_t)stream, M, B, stages);
    return hipGetLastError() == hipSuccess;
  }
  static bool launch_rollout(const DModel* M, const DBatch* B, int nenv, const RolloutArgs& A, int lds, void* stream) {
    hipLaunchKernelGGL(mjh_k_rollout, dim3(nenv), dim3(MJH_WAVE), (size_t)lds, (hipStream_t)stream, M, B, A);
    if (hipGetLastError() != hipSuccess) return false;
    hipLaunchKernelGGL(mjh_k_balance, dim3(1), dim3(1024), 0, (hipStream_t)stream, B);
    return hipGetLastError() == hipSuccess;
  }
  static dim3 lane_grid(int nenv, int epw) { return dim3((nenv + epw - 1) / epw); }
  static bool launch_smooth(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs& A, void* stream) {
    hipLaunchKernelGGL(mjh_k_smooth, lane_grid(nenv, epw), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B, A, epw);
    return hipGetLastError() == hipSuccess;
  }
  static bool launch_integrate(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs& A, void* stream) {
    hipLaunchKernelGGL(mjh_k_integrate, lane_grid(nenv, epw), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B, A, epw);
    return hipGetLastError() == hipSuccess;
  }
  static bool launch_lane_forward(const DModel* M, const DBatch* B, int nenv, int epw, int stages, void* stream) {
    hipLaunchKernelGGL(mjh_k_lane_forward, lane_grid(nenv, epw), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B, stages, epw);
    return hipGetLastError() == hipSuccess;
  }
  static bool launch_lane_reset(const DModel* M, const DBatch* B, int nenv, int epw, void* stream) {
    hipLaunchKernelGGL(mjh_k_lane_reset, lane_grid(nenv, epw), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B, epw);
    return hipGetLastError() == hipSuccess;
  }
  static bool launch_reset(const DModel* M, const DBatch* B, int nenv, void* stream) {
    hipLaunchKernelGGL(mjh_k_reset, dim3(nenv), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B);
    return hipGetLastError() == hipSuccess;
  }
};

#include "mjh_runtime.h"
