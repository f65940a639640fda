// Kernel + launcher definitions shared by the .hip translation units of libmjhip.so.
//
// libmjhip.so compiles each SPMD mapping of the stage sources (mjh_modes.h) in its own translation
// unit so they build in parallel; a unit instantiates MJH_DEFINE_WAVE_KERNELS for its namespace and
// exports plain-C launchers that the host runtime (mjh_hip.hip + mjh_runtime.h) calls.
//
// One HIP block == one 64-lane wavefront == NSUB environments (NSUB lane groups of 64/NSUB lanes).
#pragma once
#include <hip/hip_runtime.h>

#include "mjh_modes.h"

// The model / batch descriptors (tables of device pointers, ~1 KB each) live in device memory and
// are read through the scalar cache on demand; passing them by value made the compiler hoist
// every pointer into SGPRs for the whole kernel (hundreds of spills).
// WPE = waves per SIMD the register budget is sized for: 4 (128 VGPRs) keeps 4096 one-environment
// wavefronts co-resident; the two-environment mapping needs only half as many wavefronts and gets
// 256 VGPRs.  SUBEXPR = index of the calling lane's group inside its wavefront.
// A workgroup may take up to the CU's whole 160 KB of LDS (one-wavefront workgroups of a launch with at most one
// workgroup per CU: flexes at BASELINE config 5's 256 environments keep the CG solver's dof vectors there); beyond the
// 64 KB every kernel may ask for, the limit has to be raised per kernel (and per device: called before every such launch).
static inline bool mjh_raise_lds(const void* kernel, size_t bytes) {
  if (bytes <= 64*1024) return true;
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

#define MJH_DEFINE_WAVE_KERNELS(NS, NSUB, WPE, SUBEXPR) MJH_DEFINE_WAVE_KERNELS_AS(NS, NS, NSUB, WPE, SUBEXPR)
// (NM: suffix of the kernel / launcher names -- the generic namespace is compiled twice, for two register budgets)
#define MJH_DEFINE_WAVE_KERNELS_AS(NS, NM, NSUB, WPE, SUBEXPR)                                                        \
  __global__ __launch_bounds__(MJH_WAVE) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))                        \
  void mjh_k_forward_##NM(const DModel* __restrict__ M, const DBatch* __restrict__ B, int stages) {            \
    const int e = (int)blockIdx.x*(NSUB) + (SUBEXPR);                                                          \
    if (e >= B->nenv) return;                                                                                  \
    NS::forward_or_euler(wv_const_ref(M), wv_const_ref(B), e, stages);                                         \
  }                                                                                                            \
  __global__ __launch_bounds__(MJH_WAVE) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))                        \
  void mjh_k_rollout_##NM(const DModel* __restrict__ M, const DBatch* __restrict__ B, RolloutArgs A) {         \
    const int w = (int)blockIdx.x*(NSUB) + (SUBEXPR);                                                          \
    if (w >= (A.nlaunch ? A.nlaunch : B->nenv)) return;                                                        \
    /* perm lists the environments by decreasing cost: workgroups are dispatched in blockIdx order  */        \
    /* round-robin over the XCDs / CUs, so every SIMD receives a mix of cheap and expensive ones,   */        \
    /* and the environments that share a wavefront (NSUB > 1) have similar solver work              */        \
    NS::rollout_env(wv_const_ref(M), wv_const_ref(B), A.nlaunch ? w : B->perm[w], A);                          \
  }                                                                                                            \
  extern "C" bool mjh_launch_forward_##NM(const DModel* M, const DBatch* B, int nenv, int stages, int lds,     \
                                          void* stream) {                                                      \
    if (!mjh_raise_lds((const void*)mjh_k_forward_##NM, (size_t)lds*(NSUB))) return false;                     \
    hipLaunchKernelGGL(mjh_k_forward_##NM, dim3((nenv + (NSUB) - 1)/(NSUB)), dim3(MJH_WAVE),                   \
                       (size_t)lds*(NSUB), (hipStream_t)stream, M, B, stages);                                 \
    return hipGetLastError() == hipSuccess;                                                                    \
  }                                                                                                            \
  extern "C" bool mjh_launch_rollout_##NM(const DModel* M, const DBatch* B, int nenv, const RolloutArgs* A,    \
                                          int lds, void* stream) {                                             \
    if (!mjh_raise_lds((const void*)mjh_k_rollout_##NM, (size_t)lds*(NSUB))) return false;                     \
    hipLaunchKernelGGL(mjh_k_rollout_##NM, dim3((nenv + (NSUB) - 1)/(NSUB)), dim3(MJH_WAVE),                   \
                       (size_t)lds*(NSUB), (hipStream_t)stream, M, B, *A);                                     \
    return hipGetLastError() == hipSuccess;                                                                    \
  }

// Multi-wavefront workgroups (mjh_modes.h: wn + wq): MJH_MW wavefronts per environment; wave 0 runs the step, the
// others wait for the stages it posts.  WPE = wavefronts per SIMD the register budget is sized for.
#define MJH_DEFINE_MULTIWAVE_KERNELS(WPE)                                                                      \
  __global__ __launch_bounds__(MJH_WAVE*MJH_MW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))                 \
  void mjh_k_forward_wn(const DModel* __restrict__ M, const DBatch* __restrict__ B, int stages) {              \
    const int e = (int)blockIdx.x;                                                                             \
    if (e >= B->nenv) return;                                                                                  \
    if (threadIdx.x >= MJH_WAVE) { mw_helper_loop(wv_const_ref(M), wv_const_ref(B)); return; }                 \
    wn::forward_or_euler(wv_const_ref(M), wv_const_ref(B), e, stages);                                         \
    mw_release_helpers(wv_const_ref(B));                                                                       \
  }                                                                                                            \
  __global__ __launch_bounds__(MJH_WAVE*MJH_MW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))                 \
  void mjh_k_rollout_wn(const DModel* __restrict__ M, const DBatch* __restrict__ B, RolloutArgs A) {           \
    const int w = (int)blockIdx.x;                                                                             \
    if (w >= (A.nlaunch ? A.nlaunch : B->nenv)) return;                                                        \
    if (threadIdx.x >= MJH_WAVE) { mw_helper_loop(wv_const_ref(M), wv_const_ref(B)); return; }                 \
    wn::rollout_env(wv_const_ref(M), wv_const_ref(B), A.nlaunch ? w : B->perm[w], A);                          \
    mw_release_helpers(wv_const_ref(B));                                                                       \
  }                                                                                                            \
  extern "C" bool mjh_launch_forward_wn(const DModel* M, const DBatch* B, int nenv, int stages, int lds,       \
                                        void* stream) {                                                        \
    const size_t bytes = (size_t)lds + MJH_MW_LDS_TAIL;                                                        \
    if (!mjh_raise_lds((const void*)mjh_k_forward_wn, bytes)) return false;                                    \
    hipLaunchKernelGGL(mjh_k_forward_wn, dim3(nenv), dim3(MJH_WAVE*MJH_MW), bytes, (hipStream_t)stream, M, B, stages); \
    return hipGetLastError() == hipSuccess;                                                                    \
  }                                                                                                            \
  extern "C" bool mjh_launch_rollout_wn(const DModel* M, const DBatch* B, int nenv, const RolloutArgs* A,      \
                                        int lds, void* stream) {                                               \
    const size_t bytes = (size_t)lds + MJH_MW_LDS_TAIL;                                                        \
    if (!mjh_raise_lds((const void*)mjh_k_rollout_wn, bytes)) return false;                                    \
    hipLaunchKernelGGL(mjh_k_rollout_wn, dim3(nenv), dim3(MJH_WAVE*MJH_MW), bytes, (hipStream_t)stream, M, B, *A); \
    return hipGetLastError() == hipSuccess;                                                                    \
  }

#define MJH_DECLARE_WAVE_LAUNCHERS(NS)                                                                         \
  extern "C" bool mjh_launch_forward_##NS(const DModel* M, const DBatch* B, int nenv, int stages, int lds,     \
                                          void* stream);                                                       \
  extern "C" bool mjh_launch_rollout_##NS(const DModel* M, const DBatch* B, int nenv, const RolloutArgs* A,    \
                                          int lds, void* stream);
