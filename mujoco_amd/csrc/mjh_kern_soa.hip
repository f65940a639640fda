// libmjhip.so, translation unit of the SoA pipeline: namespace ws (wave mapping on strided views:
// the constraint kernel) and namespace ln (one lane per environment: smooth / integrate kernels).
#define MJH_BUILD_WS 1
#define MJH_BUILD_LN 1
#include "mjh_kernels.h"

__global__ __launch_bounds__(MJH_WAVE) __attribute__((amdgpu_waves_per_eu(4, 4)))
void mjh_k_forward_soa(const DModel* __restrict__ M, const DBatch* __restrict__ B, int stages) {
  ws::forward_or_euler(wv_const_ref(M), wv_const_ref(B), (int)blockIdx.x, stages);
}

// ---- lane-mode kernels: one lane per environment, epw environments per wavefront
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

static dim3 lane_grid(int nenv, int epw) { return dim3((nenv + epw - 1) / epw); }
extern "C" bool mjh_launch_forward_soa(const DModel* M, const DBatch* B, int nenv, int stages, int lds, void* stream) {
  if (!mjh_raise_lds((const void*)mjh_k_forward_soa, (size_t)lds)) return false;
  hipLaunchKernelGGL(mjh_k_forward_soa, dim3(nenv), dim3(MJH_WAVE), (size_t)lds, (hipStream_t)stream, M, B, stages);
  return hipGetLastError() == hipSuccess;
}
extern "C" bool mjh_launch_smooth(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs* A, void* stream) {
  hipLaunchKernelGGL(mjh_k_smooth, lane_grid(nenv, epw), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B, *A, epw);
  return hipGetLastError() == hipSuccess;
}
extern "C" bool mjh_launch_integrate(const DModel* M, const DBatch* B, int nenv, int epw, const RolloutArgs* A, void* stream) {
  hipLaunchKernelGGL(mjh_k_integrate, lane_grid(nenv, epw), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B, *A, epw);
  return hipGetLastError() == hipSuccess;
}
extern "C" bool mjh_launch_lane_forward(const DModel* M, const DBatch* B, int nenv, int epw, int stages, void* stream) {
  hipLaunchKernelGGL(mjh_k_lane_forward, lane_grid(nenv, epw), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B, stages, epw);
  return hipGetLastError() == hipSuccess;
}
extern "C" bool mjh_launch_lane_reset(const DModel* M, const DBatch* B, int nenv, int epw, void* stream) {
  hipLaunchKernelGGL(mjh_k_lane_reset, lane_grid(nenv, epw), dim3(MJH_WAVE), 0, (hipStream_t)stream, M, B, epw);
  return hipGetLastError() == hipSuccess;
}
