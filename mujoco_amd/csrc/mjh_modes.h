// The stage sources are written once against a small SPMD vocabulary (wv_lane, wv_sub, wv_sync,
// wv_bcast, wv_ballot, wv_sum_i, wv_exscan_i, wv_any, MJH_FOR_LANES, MJH_W) and a compile-time
// feature set (MJH_HAS, mjh_types.h), and compiled in several mappings, one namespace each:
//
//   wv : "one wavefront per environment", every feature.  MJH_W = 64 work items of a phase run on
//        the 64 lanes of the wavefront that owns the environment; phases are separated by
//        wv_sync(); cross-lane primitives are DPP/readlane/shuffles (mjh_spmd.h).
//   ws : the same mapping on SoA batches (strided views): constraint kernel of the 3-kernel pipeline.
//   ln : "one lane per environment".  MJH_W = 1: every GPU lane walks the serial algorithm of its
//        own environment, a wavefront steps 64 environments in lockstep; model constants are
//        wave-uniform (scalar loads), mjData fields are read SoA-across-environments.
//   wl : wv restricted to the LEAN feature set (PGS, pyramidal cones, Euler, primitive colliders,
//        no sensors / equalities / ...): the kernel a model like humanoid.xml actually needs,
//        without the stack frames and register pressure of everything else.
//
// A translation unit selects what it compiles with MJH_BUILD_<NS> (none given: everything, which
// is what the host emulation does); libmjhip.so compiles the namespaces in separate .hip files so
// they build in parallel.
#pragma once

#include "mjh_spmd.h"
#include "mjh_math.h"
#include "mjh_types.h"

#if !defined(MJH_BUILD_WV) && !defined(MJH_BUILD_WS) && !defined(MJH_BUILD_LN) && !defined(MJH_BUILD_WL) && !defined(MJH_BUILD_WN)
#define MJH_BUILD_WV 1
#define MJH_BUILD_WS 1
#define MJH_BUILD_LN 1
#define MJH_BUILD_WL 1
#define MJH_BUILD_WN 1
#endif

// kernel variants (mjhipBatch_::variant, RolloutArgs consumers): which namespace steps a batch
// (round 2 also carried "lean2" / "lean4": two / four environments per wavefront.  Measured 2-3x slower
// at every batch size of interest (profiles/r02_variants) and deleted in round 3.)
// MJH_VAR_MULTIWAVE: MJH_MW wavefronts per environment (namespaces wn + wq below), the generic feature set; chosen for
// flex models at launches of at most one workgroup per CU (BASELINE config 5)
enum { MJH_VAR_GENERIC = 0, MJH_VAR_LEAN = 1, MJH_VAR_MULTIWAVE = 2, MJH_NVARIANT = 3 };
static inline int mjh_variant_nsub(int) { return 1; }
static inline int mjh_variant_features(int v) { return v == MJH_VAR_LEAN ? MJH_FT_LEAN : MJH_FT_ALL; }
static inline const char* mjh_variant_name(int v) {
  return v == MJH_VAR_GENERIC ? "generic" : (v == MJH_VAR_LEAN ? "lean" : "multiwave");
}
// stages every wavefront of a multi-wavefront workgroup runs together (mw_exec below)
enum { MJH_MWS_EXIT = 0, MJH_MWS_KIN = 1, MJH_MWS_COMPOS, MJH_MWS_FLEXEDGES, MJH_MWS_TAVEL, MJH_MWS_COMVEL, MJH_MWS_PASSIVE,
       MJH_MWS_RNE, MJH_MWS_ELEMS, MJH_MWS_CSRPASS, MJH_MWS_CRB, MJH_MWS_FACTOR, MJH_MWS_ACCEL, MJH_MWS_EULER, MJH_MWS_CSRVALS };
// bytes at the end of a multi-wavefront workgroup's LDS block that the residency plan leaves alone: the command word
// wave 0 posts for the helper wavefronts (first 64 bytes) and the argument block of a stage that takes one (CsrPass)
#define MJH_MW_LDS_TAIL 256
#define MJH_MW_LDS_ARGS 64

// ------------------------------------------------------------------------------------------------
// wave mode (one environment per wavefront)
// ------------------------------------------------------------------------------------------------
#define MJH_W 64
#define MJH_LANE_MODE 0
#define MJH_DEVN MJH_DEVN_WAVE
// a stage call that a multi-wavefront workgroup runs on all of its wavefronts (namespace wn redefines this)
#define MJH_WIDE(id, call) call
// the same for a stage that takes an argument block (posted through LDS in a multi-wavefront workgroup)
#define MJH_WIDE_ARGS(id, A, call) call
#define MJH_HELPERS_ARGS(id, A, call) call
#define MJH_WIDE_IF(cond, id, call) call
// end of a section in which the rows of the group ran apart: everyone is back, memory is visible
#define MJH_GROUP_JOIN() do { wv_converge(); wv_sync(); } while (0)
#define MJH_FOR_LANES(i, n) for (int i = wv_lane(); i < (n); i += MJH_W)
// entry of an out-of-line stage function: its arguments arrive in VGPRs; all three are wave-uniform
// namespace wv serves environment-major batches only (B.soa == 0): telling the compiler makes every
// strided view a unit-stride view (no index multiply per access)
#if defined(MJH_HOSTSIM) || !defined(MJH_STAGE_LAUNDER)
#define MJH_ENTER(M_, B_, e_) MREF M = wv_uniform_ref(M_); BREF B = wv_uniform_ref(B_); const int e = wv_uniform_i(e_); \
                              if (B.soa != 0) __builtin_unreachable()
#else
// (experiment -DMJH_STAGE_LAUNDER: every stage reaches the descriptors through pointers the compiler cannot relate to the
// previous stage's, so a descriptor word is re-read from the scalar cache per stage instead of being carried in SGPRs)
#define MJH_ENTER(M_, B_, e_) const MJH_CONST_AS DModel* mp_ = &wv_uniform_ref(M_); const MJH_CONST_AS DBatch* bp_ = &wv_uniform_ref(B_); \
                              asm volatile("" : "+s"(mp_), "+s"(bp_)); \
                              MREF M = *mp_; BREF B = *bp_; const int e = wv_uniform_i(e_); \
                              if (B.soa != 0) __builtin_unreachable()
#endif
#if MJH_BUILD_WV
#define MJH_FEATURES MJH_FT_ALL
namespace wv {
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#endif
#if MJH_BUILD_WL
#define MJH_FEATURES MJH_FT_LEAN
namespace wl {
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#endif
#undef MJH_ENTER
// namespace ws: the same wave mapping on SoA batches (constraint kernel of the per-step pipeline)
#define MJH_ENTER(M_, B_, e_) MREF M = wv_uniform_ref(M_); BREF B = wv_uniform_ref(B_); const int e = wv_uniform_i(e_)
#if MJH_BUILD_WS
#define MJH_FEATURES MJH_FT_ALL
namespace ws {
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#endif
#undef MJH_W
#undef MJH_FOR_LANES
#undef MJH_ENTER
#undef MJH_WIDE
#undef MJH_WIDE_ARGS
#undef MJH_HELPERS_ARGS
#undef MJH_WIDE_IF

#undef MJH_LANE_MODE
#undef MJH_DEVN


// ------------------------------------------------------------------------------------------------
// multi-wavefront workgroups: MJH_MW wavefronts per environment
//
// A launch of at most one workgroup per CU (256 environments of a flex model: BASELINE config 5) leaves three of a
// CU's four SIMDs idle under the one-wavefront mapping.  Here a workgroup is MJH_MW wavefronts:
//   * wave 0 runs the whole step (namespace wn: the wave mapping above, except that wv_sync() is a memory fence and
//     not a workgroup barrier -- the other wavefronts are not there to meet it);
//   * the helper wavefronts wait at a workgroup barrier; when wave 0 reaches a stage whose work is nothing but
//     independent items between barriers (MJH_FOR_LANES loops separated by wv_sync(): kinematics, comPos, the flex
//     position / edge / passive passes, comVel, rne -- no wavefront collective inside), it posts the stage's id in
//     LDS, meets the helpers at the barrier, and ALL wavefronts run the stage from namespace wq, where the "group" is
//     the workgroup: MJH_W = 64 MJH_MW lanes, wv_lane() = thread index in the workgroup, wv_sync() = s_barrier.
//     Same items, same arithmetic per item: results are bit-identical to the one-wavefront mapping.
// The LDS block (and its residency plan) belongs to the workgroup, so every wavefront sees the same fields.
// ------------------------------------------------------------------------------------------------
#if MJH_BUILD_WN
#define MJH_W (MJH_WAVE*MJH_MW)
#define MJH_LANE_MODE 0
#define MJH_DEVN MJH_DEVN_WAVE
#define MJH_FOR_LANES(i, n) for (int i = wv_lane(); i < (n); i += MJH_W)
#define MJH_WIDE(id, call) call
#define MJH_WIDE_ARGS(id, A, call) call
#define MJH_HELPERS_ARGS(id, A, call) call
#define MJH_WIDE_IF(cond, id, call) call
#undef MJH_GROUP_JOIN
#define MJH_GROUP_JOIN() wv_sync()
#define MJH_ENTER(M_, B_, e_) MREF M = wv_uniform_ref(M_); BREF B = wv_uniform_ref(B_); const int e = wv_uniform_i(e_); \
                              if (B.soa != 0) __builtin_unreachable()
#define MJH_FEATURES MJH_FT_ALL
// the workgroup barrier at which the helper wavefronts wait for a stage, and the command word in LDS
#ifdef MJH_HOSTSIM
// (emulation: helper fibers PARK at the barrier -- the scheduler skips them -- until the last fiber of wave 0 arrives;
// everyone leaves in the same scheduler pass, see wv_converge)
MJH_DEV void mw_barrier() {
  mjhsim::WaveSim* w = mjhsim::g_wave;
  const long long mine = ++w->arrive_all[w->cur];
  bool last = true;
  for (int l = 0; l < w->nfib; l++) if (w->arrive_all[l] < mine) last = false;
  if (last) {
    w->all_done = w->round;
    for (int l = 0; l < w->nfib; l++) w->parked[l] = 0;
  } else if (w->cur >= MJH_WAVE) {
    w->parked[w->cur] = 1;
  }
  for (;;) {
    bool all = true;
    for (int l = 0; l < w->nfib; l++) if (w->arrive_all[l] < mine) all = false;
    if (all && w->round >= w->all_done + 2) break;
    mjhsim::yield();
  }
}
#else
MJH_DEV void mw_barrier() { __syncthreads(); }
#endif
namespace wq {
#ifdef MJH_HOSTSIM
MJH_DEV int wv_lane() { return mjhsim::lane(); }
// (a counted barrier: the rows of the convex narrowphase leave their loops at different times)
MJH_DEV void wv_sync() { ::mw_barrier(); }
#else
MJH_DEV int wv_lane() { return (int)threadIdx.x; }
MJH_DEV void wv_sync() { __syncthreads(); }
#endif
#include "mjh_flexinterp.h"
#include "mjh_flex.h"
#include "mjh_smooth.h"
#include "mjh_collision.h"
#include "mjh_flexcol.h"
#include "mjh_csrpass.h"
}
#undef MJH_FEATURES
#undef MJH_W
#undef MJH_FOR_LANES
#undef MJH_WIDE
#undef MJH_WIDE_ARGS
#undef MJH_HELPERS_ARGS
#undef MJH_WIDE_IF
#undef MJH_ENTER
#undef MJH_GROUP_JOIN
#define MJH_GROUP_JOIN() do { wv_converge(); wv_sync(); } while (0)

// a posted stage, run by every wavefront of the workgroup
MJH_DEV void mw_exec(MREF M, BREF B, int e, int id) {
  switch (id) {
    case MJH_MWS_KIN: wq::stage_kinematics(M, B, e); if (M.s.nflex) wq::stage_flex_pos(M, B, e); break;
    case MJH_MWS_COMPOS: wq::stage_compos(M, B, e); break;
    case MJH_MWS_FLEXEDGES: wq::stage_flex_edges(M, B, e); break;
    case MJH_MWS_TAVEL: wq::stage_ten_act_velocity(M, B, e); break;
    case MJH_MWS_COMVEL: wq::stage_comvel(M, B, e); break;
    case MJH_MWS_PASSIVE: wq::stage_passive(M, B, e); break;
    case MJH_MWS_RNE: wq::stage_rne(M, B, e); break;
    case MJH_MWS_ELEMS: wq::rc_elem_rows(M, B, e); break;
    case MJH_MWS_CRB: wq::stage_crb(M, B, e, 0); break;
    case MJH_MWS_FACTOR: wq::stage_factor_m(M, B, e); break;
    case MJH_MWS_ACCEL: wq::stage_acceleration(M, B, e); break;
    case MJH_MWS_EULER: wq::euler_advance(M, B, e); break;
    case MJH_MWS_CSRVALS: { const CsrRowArgs A = *(const CsrRowArgs*)(mjh_lds() + B.lds_bytes + MJH_MW_LDS_ARGS); wq::csr_row_values(M, B, e, A); break; }
    case MJH_MWS_CSRPASS: { const CsrPass A = *(const CsrPass*)(mjh_lds() + B.lds_bytes + MJH_MW_LDS_ARGS); wq::csr_pass(M, A); break; }
    default: break;
  }
}
MJH_DEV volatile int* mw_command(BREF B) { return (volatile int*)(mjh_lds() + B.lds_bytes); }
// wave 0: post stage `id` for environment e and run it together with the helpers
MJH_DEV void mw_run(MREF M, BREF B, int e, int id) {
  volatile int* cmd = mw_command(B);
  if (wv_lane() == 0) { cmd[0] = id; cmd[1] = e; }
  mw_barrier();
  mw_exec(M, B, e, id);
}
// wave 0: post a stage together with its argument block
template <class T>
MJH_DEV void mw_run_args(MREF M, BREF B, int e, int id, const T& A) {
  static_assert(sizeof(T) <= MJH_MW_LDS_TAIL - MJH_MW_LDS_ARGS, "argument block larger than the LDS tail");
  if (wv_lane() == 0) *(T*)(mjh_lds() + B.lds_bytes + MJH_MW_LDS_ARGS) = A;
  mw_run(M, B, e, id);
}
// wave 0: post the explicit-index CG solver's pass for the HELPER wavefronts alone and wait for them.  Wave 0 holds the
// solver's state in registers; running its share of the pass inline cost it a spill / reload of that state around every
// pass (out of line: the same through the call, measured slower still) -- it now only posts the block and meets the two
// barriers, and the pass runs on 64 (MJH_MW - 1) lanes.
MJH_DEV void mw_run_helpers(MREF M, BREF B, int e, int id, CsrPass& A) {
  A.lane0 = MJH_WAVE; A.width = MJH_WAVE*(MJH_MW - 1);
  if (wv_lane() == 0) *(CsrPass*)(mjh_lds() + B.lds_bytes + MJH_MW_LDS_ARGS) = A;
  volatile int* cmd = mw_command(B);
  if (wv_lane() == 0) { cmd[0] = id; cmd[1] = e; }
  mw_barrier();
  mw_barrier();
}
// the helper wavefronts' whole program
MJH_DEV void mw_helper_loop(MREF M, BREF B) {
  volatile int* cmd = mw_command(B);
  for (;;) {
    mw_barrier();
    const int id = wv_uniform_i(cmd[0]), e = wv_uniform_i(cmd[1]);
    if (id == MJH_MWS_EXIT) break;
    mw_exec(M, B, e, id);
  }
}
MJH_DEV void mw_release_helpers(BREF B) {
  volatile int* cmd = mw_command(B);
  if (wv_lane() == 0) cmd[0] = MJH_MWS_EXIT;
  mw_barrier();
}

#define MJH_W 64
#define MJH_LANE_MODE 0
#define MJH_DEVN MJH_DEVN_WAVE
#define MJH_FOR_LANES(i, n) for (int i = wv_lane(); i < (n); i += MJH_W)
#define MJH_WIDE(id, call) mw_run(M, B, e, id)
#define MJH_WIDE_ARGS(id, A, call) mw_run_args(M, B, e, id, A)
#define MJH_HELPERS_ARGS(id, A, call) mw_run_helpers(M, B, e, id, A)
// (a stage that is workgroup-wide only under a condition -- e.g. not when a register-resident one-wavefront routine applies)
#define MJH_WIDE_IF(cond, id, call) do { if (cond) mw_run(M, B, e, id); else { call; } } while (0)
#define MJH_ENTER(M_, B_, e_) MREF M = wv_uniform_ref(M_); BREF B = wv_uniform_ref(B_); const int e = wv_uniform_i(e_); \
                              if (B.soa != 0) __builtin_unreachable()
#define MJH_FEATURES MJH_FT_ALL
namespace wn {
// (wave 0 of a multi-wavefront workgroup: phases are separated by a memory fence -- its own accesses complete, in
// order -- not by the workgroup barrier the helpers would have to meet)
#ifdef MJH_HOSTSIM
MJH_DEV void wv_sync() { mjhsim::yield(); }
#else
MJH_DEV void wv_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
#endif
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#undef MJH_W
#undef MJH_FOR_LANES
#undef MJH_WIDE
#undef MJH_WIDE_ARGS
#undef MJH_HELPERS_ARGS
#undef MJH_WIDE_IF
#undef MJH_ENTER
#undef MJH_LANE_MODE
#undef MJH_DEVN
#endif   // MJH_BUILD_WN
// ------------------------------------------------------------------------------------------------
// lane mode
// ------------------------------------------------------------------------------------------------
#define MJH_W 1
#define MJH_LANE_MODE 1
#define MJH_DEVN MJH_DEVN_LANE
#define MJH_WIDE(id, call) call
#define MJH_WIDE_ARGS(id, A, call) call
#define MJH_HELPERS_ARGS(id, A, call) call
#define MJH_WIDE_IF(cond, id, call) call
#define MJH_FOR_LANES(i, n) for (int i = 0; i < (n); i++)
// (stage functions are inlined into the kernel here: descriptors are already uniform, e is per lane)
#define MJH_ENTER(M_, B_, e_) MREF M = M_; BREF B = B_; const int e = e_
#if MJH_BUILD_LN
#define MJH_FEATURES MJH_FT_ALL
namespace ln {
// the SPMD vocabulary for a width-1 "wave": these hide the wavefront primitives of mjh_spmd.h
MJH_DEV int wv_lane() { return 0; }
MJH_DEV int wv_sub() { return 0; }
MJH_DEV void wv_sync() {}
MJH_DEV double wv_bcast(double v, int) { return v; }
MJH_DEV int wv_bcast_i(int v, int) { return v; }
MJH_DEV uint64_t wv_ballot(int pred) { return pred ? 1ull : 0ull; }
MJH_DEV int wv_sum_i(int v) { return v; }
MJH_DEV int wv_exscan_i(int) { return 0; }
MJH_DEV int wv_any(int pred) { return pred != 0; }
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#endif
#undef MJH_W
#undef MJH_LANE_MODE
#undef MJH_DEVN
#undef MJH_FOR_LANES
#undef MJH_ENTER
#undef MJH_WIDE
#undef MJH_WIDE_ARGS
#undef MJH_HELPERS_ARGS
#undef MJH_WIDE_IF
