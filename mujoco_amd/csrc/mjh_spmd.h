// SPMD layer for the mjhip kernels: one 64-lane wavefront per environment.
//
// All physics kernels are written once, in HIP device style (lane = threadIdx.x, block = one
// wavefront = one environment, __syncthreads() between phases, wave-level reductions through
// the helpers below).  Two builds exist:
//
//   * hipcc --offload-arch=gfx950   : the product.  Lanes are the 64 lanes of a CDNA4 wavefront,
//                                     wv_sync() is s_barrier + waitcnt, reductions use DPP/shuffles.
//   * g++ -DMJH_HOSTSIM (tests only): a deterministic fiber emulation of one wavefront (64 ucontext
//                                     fibers, switched at every wv_sync()).  It exists so kernel
//                                     logic can be parity-checked against the oracle in a container
//                                     with no GPU; it is compiled only by tests/hostsim and is
//                                     never loaded by the product library.
#pragma once

#include <math.h>
#include <stdint.h>

#define MJH_WAVE 64
// wavefronts of a multi-wavefront workgroup (mjh_modes.h: namespaces wn / wq; low-occupancy launches of flex models)
#ifndef MJH_MW
#define MJH_MW 8
#endif

// issue priority of rollout wavefronts by solver work (mjh_step.h: rollout_env; -DMJH_NO_STEP_PRIO: measurement builds)
#if defined(MJH_HOSTSIM) || defined(MJH_NO_STEP_PRIO)
#define MJH_STEP_PRIO 0
#else
#define MJH_STEP_PRIO 1
#endif

#ifdef MJH_HOSTSIM
// ------------------------------------------------------------------------------------------------
// host emulation of a wavefront (tests only)
// ------------------------------------------------------------------------------------------------
#include <stdlib.h>
#include <string.h>

#define MJH_DEV static inline
#define MJH_MEM inline
// wave-uniform, read-only views of model data (identity on the host)
#define MJH_CONST_AS
template <class T> static inline const T* wv_uniform_ptr(const T* p) { return p; }
template <class T> static inline const T& wv_uniform_ref(const T& r) { return r; }
template <class T> static inline const T& wv_const_ref(const T* p) { return *p; }
static inline int wv_uniform_i(int v) { return v; }
static inline uint64_t wv_uniform_u64(uint64_t v) { return v; }
#define MJH_DEVN_WAVE static __attribute__((noinline))
#define MJH_DEVN_HOT static __attribute__((noinline))
#define MJH_DEVN_LANE static inline
#define MJH_GLOBAL static void
#define MJH_SHARED static

namespace mjhsim {
struct WaveSim {
  void* sched_sp;            // saved stack pointers of the scheduler and of the 64 lane fibers
  void* ctx_sp[MJH_WAVE*MJH_MW];    // (mjh_ctx_switch, hostsim.cpp: a register-only switch, no syscalls)
  char* stacks;
  int cur;            // lane currently running (0 .. nfib-1; wavefront = cur / 64)
  int nfib;           // fibers of the emulated workgroup: 64, or 64*MJH_MW for the multi-wavefront kernels
  int done[MJH_WAVE*MJH_MW];
  long long arrive_all[MJH_WAVE*MJH_MW];   // workgroup barrier of the multi-wavefront kernels
  int parked[MJH_WAVE*MJH_MW];             // helper fibers waiting at that barrier: the scheduler skips them
  long long round;                         // scheduler passes so far
  long long row_done[MJH_WAVE*MJH_MW/16], wave_done, all_done;   // pass in which the last fiber reached the current barrier
  int env;            // blockIdx.x
  int reverse;        // run lanes 63..0 instead of 0..63 (race detector)
  // scratch for cross-lane primitives
  double dscratch[MJH_WAVE*MJH_MW];     // (sized for the workgroup: the row primitives also run on helper wavefronts)
  double dscratch2[MJH_WAVE*MJH_MW];
  long long iscratch[MJH_WAVE*MJH_MW];
  long long iscratch2[MJH_WAVE*MJH_MW];
  long long arrive_row[MJH_WAVE*MJH_MW];    // wv_row_converge / wv_converge arrival counts
  long long arrive_wave[MJH_WAVE];
};
extern thread_local WaveSim* g_wave;
static inline int lane() { return g_wave->cur; }
static inline int env() { return g_wave->env; }
extern "C" void mjh_ctx_switch(void** from_sp, void* to_sp);
static inline void yield() { mjh_ctx_switch(&g_wave->ctx_sp[g_wave->cur], g_wave->sched_sp); }
}  // namespace mjhsim

MJH_DEV int wv_lane() { return mjhsim::lane(); }
MJH_DEV int wv_env() { return mjhsim::env(); }
#include <stdio.h>
// work counters of the convex narrowphase (mjh_convex.h), printed at exit when $MJH_RC_STATS is set
namespace mjhsim { inline long long* rc_stats() {
  static long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  static bool reg = false;
  if (!reg) { reg = true; if (getenv("MJH_RC_STATS")) atexit([]() { long long* q = rc_stats();
    fprintf(stderr, "rc_stats: queries %lld distance-iterations %lld containment-tests %lld polytopes %lld expansions %lld multicontacts %lld; fused CG passes %lld\n",
            q[0], q[1], q[2], q[3], q[4], q[5], q[6]); }); }
  return c; } }
MJH_DEV void wv_sync() { mjhsim::yield(); }

// broadcast v from lane src to all lanes (src must be wave-uniform)
MJH_DEV double wv_bcast(double v, int src) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = w->dscratch[src];
  mjhsim::yield();
  return r;
}
MJH_DEV int wv_bcast_i(int v, int src) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = v;
  mjhsim::yield();
  int r = (int)w->iscratch[src];
  mjhsim::yield();
  return r;
}
// 64-bit mask of lanes with pred != 0
MJH_DEV uint64_t wv_ballot(int pred) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = pred ? 1 : 0;
  mjhsim::yield();
  uint64_t m = 0;
  for (int l = 0; l < MJH_WAVE; l++) if (w->iscratch[l]) m |= (1ull << l);
  mjhsim::yield();
  return m;
}
// integer sum over all lanes
MJH_DEV int wv_sum_i(int v) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = v;
  mjhsim::yield();
  long long s = 0;
  for (int l = 0; l < MJH_WAVE; l++) s += w->iscratch[l];
  mjhsim::yield();
  return (int)s;
}
// exclusive prefix sum over lanes
MJH_DEV int wv_exscan_i(int v) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = v;
  mjhsim::yield();
  long long s = 0;
  for (int l = 0; l < w->cur; l++) s += w->iscratch[l];
  mjhsim::yield();
  return (int)s;
}
// value held by lane (lane ^ mask)
MJH_DEV double wv_shfl_xor(double v, int mask) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = w->dscratch[w->cur ^ mask];
  mjhsim::yield();
  return r;
}
// value held by lane src (src may differ per lane)
MJH_DEV double wv_shfl(double v, int src) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = w->dscratch[src & (MJH_WAVE - 1)];
  mjhsim::yield();
  return r;
}
MJH_DEV int wv_any(int pred) { return wv_ballot(pred) != 0; }
// int held by lane src (src may differ per lane)
MJH_DEV int wv_shfl_i(int v, int src) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = v;
  mjhsim::yield();
  int r = (int)w->iscratch[src & (MJH_WAVE - 1)];
  mjhsim::yield();
  return r;
}
// value held by lane (lane + K) of the caller's 16-lane row; 0 beyond the row's end
template <int K>
MJH_DEV double wv_row_shl(double v) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = ((w->cur & 15) + K < 16) ? w->dscratch[w->cur + K] : 0.0;
  mjhsim::yield();
  return r;
}
// value held by lane K of the caller's 16-lane row
template <int K>
MJH_DEV double wv_row_bcast(double v) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = w->dscratch[(w->cur & ~15) + K];
  mjhsim::yield();
  return r;
}
// v is uniform inside each 16-lane row (value r_c in row c): every lane gets (r0 + r2) + (r1 + r3),
// mju_dot's final association of its four partial sums
MJH_DEV double wv_rows_sum(double v) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = (w->dscratch[0] + w->dscratch[32]) + (w->dscratch[16] + w->dscratch[48]);
  mjhsim::yield();
  return r;
}
MJH_DEV long long wv_clock() { return 0; }
MJH_DEV int wv_sub() { return 0; }
// *p = min(*p, v), atomically with respect to the other lanes (the emulation runs them one at a time)
MJH_DEV void wv_atomic_min_i(int* p, int v) { if (v < *p) *p = v; }
MJH_DEV int wv_atomic_add_i(int* p, int v) { const int old = *p; *p = old + v; return old; }
MJH_DEV void wv_atomic_or_i(int* p, int v) { *p |= v; }


// value of lane (lane ^ 16): the neighbouring 16-lane row
MJH_DEV double sw_row_swap(double v) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = w->dscratch[w->cur ^ 16];
  mjhsim::yield();
  return r;
}

// ---- ordered reductions: the reference sums in a FIXED sequential order (plain loops, or mju_dot's four
// interleaved accumulators); these reproduce that order with the addends spread over lanes
// init (+|-) v[lo] (+|-) v[lo+1] ... (+|-) v[hi-1], left to right (lo, hi wave-uniform)
MJH_DEV double wv_chain(double init, double v, int lo, int hi, int sub) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = init;
  for (int l = lo; l < hi; l++) r = sub ? r - w->dscratch[l] : r + w->dscratch[l];
  mjhsim::yield();
  return r;
}
// init + v[l] over the lanes l selected by the wave-uniform mask, ascending (an ordered sum whose zero addends,
// which cannot change it, were dropped by the caller's mask)
MJH_DEV double wv_chain_mask(double init, double v, uint64_t mask) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = init;
  for (int l = 0; l < MJH_WAVE; l++) if ((mask >> l) & 1) r += w->dscratch[l];
  mjhsim::yield();
  return r;
}
// six chains at once: acc[q] += v[q] of lanes 0..n-1 in lane order
MJH_DEV void wv_chain6(double* acc, const double* v, int n) {
  for (int q = 0; q < 6; q++) acc[q] = wv_chain(acc[q], v[q], 0, n, 0);
}
// three chains over one mask (addends that are exact zeros may be included: they cannot change a sum)
MJH_DEV void wv_chain3_mask(double* acc, const double* v, uint64_t mask) {
  for (int q = 0; q < 3; q++) acc[q] = wv_chain_mask(acc[q], v[q], mask);
}
// mju_dot's accumulators: r[c] += v[4g + c] for g = 0 .. ngroup-1, in order of g
MJH_DEV void wv_dot4_acc(double* r, double v, int ngroup) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  for (int g = 0; g < ngroup; g++) for (int c = 0; c < 4; c++) r[c] += w->dscratch[4*g + c];
  mjhsim::yield();
}

// 64-bit word held by lane src (src wave-uniform)
MJH_DEV uint64_t wv_bcast_u64(uint64_t v, int src) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = (long long)v;
  mjhsim::yield();
  uint64_t r = (uint64_t)w->iscratch[src];
  mjhsim::yield();
  return r;
}
// 128-bit value (two words) held by lane src (src wave-uniform)
MJH_DEV void wv_bcast_u128(uint64_t lo, uint64_t hi, int src, uint64_t* rlo, uint64_t* rhi) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = (long long)lo;
  w->iscratch2[w->cur] = (long long)hi;
  mjhsim::yield();
  *rlo = (uint64_t)w->iscratch[src];
  *rhi = (uint64_t)w->iscratch2[src];
  mjhsim::yield();
}
// number of set bits of m below the calling lane's position
MJH_DEV int wv_rank_lt(uint64_t m) { const int l = mjhsim::lane(); return l ? __builtin_popcountll(m & ((1ull << l) - 1)) : 0; }
// mju_dot / mju_dotSparse over the lanes selected by a wave-uniform 128-bit mask: element t lives in p0 of lane t
// (t < 64) or in p1 of lane t - 64.  The selected elements, in ascending order, are the operands of the
// reference's loop: four interleaved accumulators over groups of four, (r0 + r2) + (r1 + r3), then the
// remainder -- added as ONE grouped term (mju_dot, engine_util_blas.c:517-525) when dense_tail, else one by one
// (mju_dotSparse, engine_util_sparse.h:218-221)
MJH_DEV double wv_dot4m(double p0, double p1, uint64_t mlo, uint64_t mhi, int dense_tail) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = p0;
  w->dscratch2[w->cur] = p1;
  mjhsim::yield();
  double v[128];
  int cnt = 0;
  for (int l = 0; l < 64; l++) if ((mlo >> l) & 1) v[cnt++] = w->dscratch[l];
  for (int l = 0; l < 64; l++) if ((mhi >> l) & 1) v[cnt++] = w->dscratch2[l];
  double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  int i = 0;
  for (; i <= cnt - 4; i += 4) { r0 += v[i]; r1 += v[i+1]; r2 += v[i+2]; r3 += v[i+3]; }
  double res = (r0 + r2) + (r1 + r3);
  const int rem = cnt - i;
  if (dense_tail) {
    if (rem == 3) res += v[i] + v[i+1] + v[i+2];
    else if (rem == 2) res += v[i] + v[i+1];
    else if (rem == 1) res += v[i];
  } else {
    for (; i < cnt; i++) res += v[i];
  }
  mjhsim::yield();
  return res;
}

// ---- 16-lane rows as independent work groups (mjh_convex.h: one geom pair per row) ------------------------------
// A row's lanes run in lockstep with each other but the four rows of a wavefront may sit in different iterations of a
// data-dependent loop: on the device that is ordinary divergence (the hardware serialises, reconvergence is
// structural); the emulation runs every lane as a fiber, so the row primitives below exchange data through per-lane
// slots that only row-mates read, and wv_row_converge / wv_converge are real barriers here (no-ops on the device).
// value of the partner lane: S = 0: lane ^ 1, 1: lane ^ 2, 2: mirrored inside the 8-lane half, 3: mirrored inside the row
template <int S>
MJH_DEV int wv_row_partner(int l) { return S == 0 ? (l ^ 1) : S == 1 ? (l ^ 2) : S == 2 ? ((l & ~7) | (7 - (l & 7))) : ((l & ~15) | (15 - (l & 15))); }
template <int S>
MJH_DEV double wv_row_xchg(double v) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = w->dscratch[wv_row_partner<S>(w->cur)];
  mjhsim::yield();
  return r;
}
template <int S>
MJH_DEV int wv_row_xchg_i(int v) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = v;
  mjhsim::yield();
  int r = (int)w->iscratch[wv_row_partner<S>(w->cur)];
  mjhsim::yield();
  return r;
}
// value / int held by lane src (0..15) of the caller's row
MJH_DEV double wv_row_get(double v, int src) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->dscratch[w->cur] = v;
  mjhsim::yield();
  double r = w->dscratch[(w->cur & ~15) | (src & 15)];
  mjhsim::yield();
  return r;
}
MJH_DEV int wv_row_get_i(int v, int src) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = v;
  mjhsim::yield();
  int r = (int)w->iscratch[(w->cur & ~15) | (src & 15)];
  mjhsim::yield();
  return r;
}
// 16-bit mask of the row's lanes with pred != 0 (bit k: lane k of the row)
MJH_DEV unsigned wv_row_ballot(int pred) {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  w->iscratch[w->cur] = pred ? 1 : 0;
  mjhsim::yield();
  unsigned m = 0;
  const int b = w->cur & ~15;
  for (int l = 0; l < 16; l++) if (w->iscratch[b + l]) m |= (1u << l);
  mjhsim::yield();
  return m;
}
// memory written by a row-mate before the call is visible after it (LDS / global exchange inside a row)
MJH_DEV void wv_row_sync() { mjhsim::yield(); }
// barriers of the emulation: every lane of the row / the wavefront has arrived
// (every fiber leaves a barrier in the SAME scheduler pass -- two passes after the last one arrived -- so that the
// one-switch-per-phase lockstep the other primitives rely on holds again afterwards)
MJH_DEV void wv_row_converge() {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  const int b = w->cur & ~15;
  const long long mine = ++w->arrive_row[w->cur];
  bool last = true;
  for (int l = 0; l < 16; l++) if (w->arrive_row[b + l] < mine) last = false;
  if (last) w->row_done[b >> 4] = w->round;
  for (;;) {
    bool all = true;
    for (int l = 0; l < 16; l++) if (w->arrive_row[b + l] < mine) all = false;
    if (all && w->round >= w->row_done[b >> 4] + 2) break;
    mjhsim::yield();
  }
}
MJH_DEV void wv_converge() {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  const long long mine = ++w->arrive_wave[w->cur];
  bool last = true;
  for (int l = 0; l < MJH_WAVE; l++) if (w->arrive_wave[l] < mine) last = false;
  if (last) w->wave_done = w->round;
  for (;;) {
    bool all = true;
    for (int l = 0; l < MJH_WAVE; l++) if (w->arrive_wave[l] < mine) all = false;
    if (all && w->round >= w->wave_done + 2) break;
    mjhsim::yield();
  }
}

#else
// ------------------------------------------------------------------------------------------------
// CDNA4 device build
// ------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>

#define MJH_DEV __device__ __forceinline__
#define MJH_MEM __device__ __forceinline__
// Wave-uniform, read-only views of model data.  A stage function that was not inlined receives
// its pointers in VGPRs, so the compiler must assume they differ per lane: every model constant
// becomes a vector load and every loop a divergent loop.  These helpers move the pointer to
// SGPRs (v_readfirstlane) and retype it to the constant address space, which makes the loads
// scalar (s_load) and their results provably uniform.
#define MJH_CONST_AS __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const MJH_CONST_AS T* wv_uniform_ptr(const T* p) {
  unsigned long long a = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (const MJH_CONST_AS T*)(((unsigned long long)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ const MJH_CONST_AS T* wv_uniform_ptr(const MJH_CONST_AS T* p) {
  unsigned long long a = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (const MJH_CONST_AS T*)(((unsigned long long)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ const MJH_CONST_AS T& wv_uniform_ref(const MJH_CONST_AS T& r) { return *wv_uniform_ptr(&r); }
// kernel argument (global pointer, already uniform) -> constant-address-space reference
template <class T>
__device__ __forceinline__ const MJH_CONST_AS T& wv_const_ref(const T* p) {
  return *(const MJH_CONST_AS T*)(unsigned long long)p;
}
__device__ __forceinline__ int wv_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// out-of-line device function: gives the big stages their own register allocation scope
// register budget: 4 waves/SIMD (<=128 VGPRs) so that 4096 one-wave environments are co-resident
#ifndef MJH_WAVES_PER_EU
#define MJH_WAVES_PER_EU 4
#endif
#ifdef MJH_INLINE_STAGES
#define MJH_DEVN_WAVE __device__ __forceinline__ static
#else
#define MJH_DEVN_WAVE __device__ __noinline__ static
#endif
// the register-resident serial sweeps (PGS, L'DL factor / solves): always out of line -- they are
// leaves that fit the caller-saved registers (nothing to save on entry) and their inner loops must
// not inherit the register pressure of whatever surrounds the call
#define MJH_DEVN_HOT __device__ __noinline__ static
#define MJH_DEVN_LANE __device__ __forceinline__ static
#define MJH_GLOBAL __global__ void
#define MJH_SHARED __shared__

MJH_DEV int wv_lane() { return (int)threadIdx.x; }
MJH_DEV int wv_env() { return (int)blockIdx.x; }
// block == one wavefront: s_barrier is (nearly) free, the waitcnt/fence part is what we need
MJH_DEV void wv_sync() { __syncthreads(); }

MJH_DEV double wv_bcast(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
MJH_DEV int wv_bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
MJH_DEV uint64_t wv_ballot(int pred) { return __ballot(pred); }
MJH_DEV int wv_sum_i(int v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
MJH_DEV int wv_exscan_i(int v) {
  int x = v;
  int lane = (int)threadIdx.x;
  for (int d = 1; d < 64; d <<= 1) {
    int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  return x - v;
}
MJH_DEV double wv_shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
MJH_DEV double wv_shfl(double v, int src) { return __shfl(v, src, 64); }
MJH_DEV int wv_any(int pred) { return __any(pred); }
MJH_DEV int wv_shfl_i(int v, int src) { return __shfl(v, src, 64); }
// value held by lane (lane + K) of the caller's 16-lane DPP row, 0 beyond its end (row_shl:K)
template <int K>
MJH_DEV double wv_row_shl(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + K, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// value held by lane K of the caller's 16-lane DPP row (v_mov_b32_dpp row_newbcast:K, no LDS)
template <int K>
MJH_DEV double wv_row_bcast(double v) {
#ifdef MJH_DPP32
  int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + K, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + K, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
#else
  // one v_mov_b64_dpp: gfx90a+ has the 64-bit DPP move for exactly this control (row_newbcast);
  // every lane of a row broadcast is written, so the "old" operand is never observed
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true);
#endif
}
// v is uniform inside each 16-lane row (value r_c in row c): every lane gets (r0 + r2) + (r1 + r3),
// mju_dot's final association of its four partial sums.  gfx950's v_permlane32_swap exchanges the
// upper 32 lanes of one register with the lower 32 of another (rows 2,3 <-> rows 0,1), then
// v_permlane16_swap the odd with the even rows: two VALU exchanges and two adds, no SGPR round
// trip.  (IEEE addition commutes, so which operand of a swap pair lands where is immaterial.)
MJH_DEV double wv_rows_sum(double v) {
  typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const u32x2_ l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const u32x2_ h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double t = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
  const unsigned tlo = (unsigned)__double2loint(t), thi = (unsigned)__double2hiint(t);
  const u32x2_ l2 = __builtin_amdgcn_permlane16_swap(tlo, tlo, false, false);
  const u32x2_ h2 = __builtin_amdgcn_permlane16_swap(thi, thi, false, false);
  return __hiloint2double((int)h2.x, (int)l2.x) + __hiloint2double((int)h2.y, (int)l2.y);
}
// constant-rate (100 MHz) timestamp, for -DMJH_PROFILE builds
MJH_DEV long long wv_clock() { return (long long)wall_clock64(); }
MJH_DEV int wv_sub() { return 0; }
// *p = min(*p, v), atomically with respect to the other lanes (flat address: LDS or global)
MJH_DEV void wv_atomic_min_i(int* p, int v) { atomicMin(p, v); }
MJH_DEV int wv_atomic_add_i(int* p, int v) { return atomicAdd(p, v); }
MJH_DEV void wv_atomic_or_i(int* p, int v) { atomicOr(p, v); }

// value of lane (lane ^ 16): the neighbouring 16-lane row.  v_permlane16_swap exchanges
// the odd rows of its first operand with the even rows of its second; with both operands = v each
// lane ends up with {own value, partner row's value} in the two results -- the one whose bits differ
// from the own value is the partner's (if neither differs they are equal anyway).
MJH_DEV double sw_row_swap(double v) {
  typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const u32x2_ l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const u32x2_ h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const bool x_is_own = (l.x == lo) && (h.x == hi);
  return __hiloint2double((int)(x_is_own ? h.y : h.x), (int)(x_is_own ? l.y : l.x));
}

// ---- ordered reductions: the reference sums in a FIXED sequential order (plain loops, or mju_dot's four
// interleaved accumulators); these reproduce that order with the addends spread over lanes.  One
// v_readlane pair + one dependent v_add_f64 per addend: a serial chain, but of register operands.
MJH_DEV double wv_chain(double init, double v, int lo, int hi, int sub) {
  double r = init;
  if (sub) { for (int l = lo; l < hi; l++) r = r - wv_bcast(v, l); }
  else { for (int l = lo; l < hi; l++) r = r + wv_bcast(v, l); }
  return r;
}
// init + v[l] over the lanes l selected by the wave-uniform mask, ascending: a scalar walk over the set bits
MJH_DEV double wv_chain_mask(double init, double v, uint64_t mask) {
  mask = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mask >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mask);
  double r = init;
  while (mask) {
    const int l = __builtin_ctzll(mask);
    mask &= mask - 1;
    r += wv_bcast(v, l);
  }
  return r;
}
// three chains over one mask: one walk over the set bits, the three dependent additions of a step overlap
MJH_DEV void wv_chain3_mask(double* acc, const double* v, uint64_t mask) {
  mask = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mask >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mask);
  double a0 = acc[0], a1 = acc[1], a2 = acc[2];
  while (mask) {
    const int l = __builtin_ctzll(mask);
    mask &= mask - 1;
    a0 += wv_bcast(v[0], l); a1 += wv_bcast(v[1], l); a2 += wv_bcast(v[2], l);
  }
  acc[0] = a0; acc[1] = a1; acc[2] = a2;
}
// six chains at once (independent accumulators: the chains overlap in the pipeline)
MJH_DEV void wv_chain6(double* acc, const double* v, int n) {
  double a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3], a4 = acc[4], a5 = acc[5];
  for (int l = 0; l < n; l++) {
    a0 += wv_bcast(v[0], l); a1 += wv_bcast(v[1], l); a2 += wv_bcast(v[2], l);
    a3 += wv_bcast(v[3], l); a4 += wv_bcast(v[4], l); a5 += wv_bcast(v[5], l);
  }
  acc[0] = a0; acc[1] = a1; acc[2] = a2; acc[3] = a3; acc[4] = a4; acc[5] = a5;
}
// mju_dot's accumulators: r[c] += v[4g + c] for g = 0 .. ngroup-1, in order of g
MJH_DEV void wv_dot4_acc(double* r, double v, int ngroup) {
  double r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
  for (int g = 0; g < ngroup; g++) {
    r0 += wv_bcast(v, 4*g); r1 += wv_bcast(v, 4*g + 1); r2 += wv_bcast(v, 4*g + 2); r3 += wv_bcast(v, 4*g + 3);
  }
  r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
}

MJH_DEV uint64_t wv_bcast_u64(uint64_t v, int src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
MJH_DEV uint64_t wv_uniform_u64(uint64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
MJH_DEV void wv_bcast_u128(uint64_t lo, uint64_t hi, int src, uint64_t* rlo, uint64_t* rhi) {
  *rlo = wv_bcast_u64(lo, src);
  *rhi = wv_bcast_u64(hi, src);
}
// number of set bits of m below the calling lane's position (v_mbcnt)
MJH_DEV int wv_rank_lt(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// mju_dot / mju_dotSparse over the lanes selected by a wave-uniform 128-bit mask (see the host build's
// comment): a scalar walk over the set bits, one v_readlane pair + one dependent v_add_f64 per element
MJH_DEV double wv_dot4m(double p0, double p1, uint64_t mlo, uint64_t mhi, int dense_tail) {
  mlo = wv_uniform_u64(mlo);
  mhi = wv_uniform_u64(mhi);
  const int cnt = __builtin_popcountll(mlo) + __builtin_popcountll(mhi);
  auto next = [&]() -> double {
    if (mlo) { const int b = __builtin_ctzll(mlo); mlo &= mlo - 1; return wv_bcast(p0, b); }
    const int b = __builtin_ctzll(mhi); mhi &= mhi - 1; return wv_bcast(p1, b);
  };
  double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  for (int g = cnt >> 2; g > 0; g--) { r0 += next(); r1 += next(); r2 += next(); r3 += next(); }
  double res = (r0 + r2) + (r1 + r3);
  const int rem = cnt & 3;
  if (dense_tail) {
    if (rem == 3) { const double a = next(), b = next(), c = next(); res += a + b + c; }
    else if (rem == 2) { const double a = next(), b = next(); res += a + b; }
    else if (rem == 1) res += next();
  } else {
    for (int k = 0; k < rem; k++) res += next();
  }
  return res;
}

// ---- 16-lane rows as independent work groups (mjh_convex.h: one geom pair per row; see the host build's comment)
// partner exchanges as DPP moves: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
template <int S>
MJH_DEV int wv_row_xchg_i(int v) {
  constexpr int ctrl = S == 0 ? 0xB1 : S == 1 ? 0x4E : S == 2 ? 0x141 : 0x140;
  return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, false);
}
template <int S>
MJH_DEV double wv_row_xchg(double v) {
  return __hiloint2double(wv_row_xchg_i<S>(__double2hiint(v)), wv_row_xchg_i<S>(__double2loint(v)));
}
MJH_DEV int wv_row_get_i(int v, int src) { return __shfl(v, (int)(threadIdx.x & 48u) | (src & 15), 64); }
MJH_DEV double wv_row_get(double v, int src) { return __shfl(v, (int)(threadIdx.x & 48u) | (src & 15), 64); }
MJH_DEV unsigned wv_row_ballot(int pred) { return (unsigned)(__ballot(pred) >> (threadIdx.x & 48u)) & 0xffffu; }
// a wavefront's LDS / memory instructions complete in program order: only the compiler has to be kept from moving
// accesses across the exchange point
MJH_DEV void wv_row_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
MJH_DEV void wv_row_converge() { __builtin_amdgcn_wave_barrier(); }
MJH_DEV void wv_converge() { __builtin_amdgcn_wave_barrier(); }

#endif  // MJH_HOSTSIM

