// mj_step for one environment per wavefront: checks -> forward -> Euler/advance, and the rollout
// loop around it (python/mujoco/rollout.cc:74-178 restated per environment).
#pragma once

#include "mjh_types.h"
#include "mjh_smooth.h"
#include "mjh_collision.h"
#include "mjh_constraint.h"
#include "mjh_solver.h"

// stage bits for partial forward evaluation (tests and per-stage profiling)
enum {
  MJH_STAGE_KINEMATICS = 1<<0,   // kinematics, comPos, tendon
  MJH_STAGE_INERTIA    = 1<<1,   // crb, factorM
  MJH_STAGE_COLLISION  = 1<<2,
  MJH_STAGE_MAKE       = 1<<3,   // makeConstraint
  MJH_STAGE_PROJECT    = 1<<4,   // Y, AR
  MJH_STAGE_TRANSMISSION = 1<<5,
  MJH_STAGE_VELOCITY   = 1<<6,   // ten/act velocity, comVel, passive, reference, rne
  MJH_STAGE_ACTUATION  = 1<<7,   // actuation + acceleration
  MJH_STAGE_CONSTRAINT = 1<<8,   // fwdConstraint
  MJH_STAGE_ALL        = (1<<9) - 1,
  MJH_STAGE_EULER      = 1<<9,   // mj_Euler + mj_advance (not part of mj_forward; for per-stage runs)
  MJH_STAGE_LDS        = 1<<21,  // host flag of mjhip_batch_forward: use the LDS residency plan (+ write-back)
  MJH_STAGE_WRITEBACK  = 1<<20,  // debug: copy LDS-resident fields to their global homes after every stage
};

// mj_resetData as far as the state vector is concerned (engine_io.c:1289-1420)
MJH_DEV void reset_env(const DModel& M, const DBatch& B, int e) {
  const DSizes& s = M.s;
  real* qpos = MJH_F(B, qpos, e);
  MJH_FOR_LANES(i, s.nq) qpos[i] = M.qpos0[i];
  real* qvel = MJH_F(B, qvel, e);
  real* ws = MJH_F(B, qacc_warmstart, e);
  real* fa = MJH_F(B, qfrc_applied, e);
  MJH_FOR_LANES(i, s.nv) { qvel[i] = 0; ws[i] = 0; fa[i] = 0; }
  real* act = MJH_F(B, act, e);
  MJH_FOR_LANES(i, s.na) act[i] = 0;
  real* ctrl = MJH_F(B, ctrl, e);
  MJH_FOR_LANES(i, s.nu) ctrl[i] = 0;
  real* xf = MJH_G(B, xfrc_applied, e);
  MJH_FOR_LANES(i, 6*s.nbody) xf[i] = 0;
  int* warn = MJH_F(B, warning, e);
  if (wv_lane() == 0) {
    MJH_F(B, time, e)[0] = 0;
    for (int k = 0; k < 8; k++) warn[k] = 0;
  }
  wv_sync();
}

// mj_checkPos / mj_checkVel / mj_checkAcc (engine_forward.c:54-113): returns 1 if x had a bad value
MJH_DEV int check_bad(const DModel& M, const DBatch& B, int e, const real* x, int n, int which) {
  int bad = 0;
  MJH_FOR_LANES(i, n) if (r_isbad(x[i])) bad = 1;
  bad = wv_any(bad);
  if (bad) {
    wv_sync();
    if (!(M.o.disableflags & (1<<16))) reset_env(M, B, e);
    int* warn = MJH_F(B, warning, e);
    if (wv_lane() == 0) warn[which] += 1;
    wv_sync();
  }
  return bad;
}

// ---- LDS residency helpers -----------------------------------------------------------------------
// copy a field between its global home and its LDS slot (no-op for fields the plan left global)
template <class T>
MJH_DEV void lds_copy_in(T* g, int n, int l, int e) {
  if (l < 0) return;
  T* dst = (T*)(mjh_lds() + l);
  const T* src = g + (size_t)e*(size_t)n;
  MJH_FOR_LANES(i, n) dst[i] = src[i];
}
template <class T>
MJH_DEV void lds_copy_out(T* g, int n, int l, int e, int cnt) {
  if (l < 0) return;
  const T* src = (const T*)(mjh_lds() + l);
  T* dst = g + (size_t)e*(size_t)n;
  MJH_FOR_LANES(i, cnt) dst[i] = src[i];
}

// kernel entry: persistent state, global home -> LDS
MJH_DEV void lds_enter(const DModel& M, const DBatch& B, int e) {
  if (!B.lds_bytes) return;
  const DSizes& s = M.s;
  (void)s;
#define X(name, cnt, lcnt, t0, t1) if ((t0) == MJH_T_BEGIN && (t1) == MJH_T_END) lds_copy_in(B.name, B.n_##name, B.l_##name, e);
  MJH_BATCH_REAL_FIELDS(X)
  MJH_BATCH_INT_FIELDS(X)
#undef X
  wv_sync();
}
// kernel exit: persistent + exported fields, LDS -> global home
MJH_DEV void lds_exit(const DModel& M, const DBatch& B, int e) {
  if (!B.lds_bytes) return;
  const DSizes& s = M.s;
  (void)s;
  wv_sync();
#define X(name, cnt, lcnt, t0, t1) if ((t1) == MJH_T_END) lds_copy_out(B.name, B.n_##name, B.l_##name, e, (int)(lcnt));
  MJH_BATCH_REAL_FIELDS(X)
  MJH_BATCH_INT_FIELDS(X)
#undef X
  wv_sync();
}
// debug write-back (tests): after timeline point t, every LDS-resident field live at t is copied to
// its global home, so the host can inspect intermediates of the LDS path field by field
MJH_DEVN void lds_writeback(const DModel& M, const DBatch& B, int e, int t) {
  if (!B.lds_bytes) return;
  const DSizes& s = M.s;
  (void)s;
  wv_sync();
#define X(name, cnt, lcnt, t0, t1) if ((t0) != MJH_T_GLB && (t0) <= t && t <= (t1)) lds_copy_out(B.name, B.n_##name, B.l_##name, e, (int)(lcnt));
  MJH_BATCH_REAL_FIELDS(X)
  MJH_BATCH_INT_FIELDS(X)
#undef X
  if (t >= MJH_T_MAKE && t <= MJH_T_CONSTRAINT) efc_writeback(M, B, e);
  wv_sync();
}

// mj_forwardSkip(mjSTAGE_NONE, skipsensor) restricted by a stage mask   (engine_forward.c:1783-1836)
// Stages run in the order of the MJH_T_* timeline: the velocity-dependent smooth terms are taken
// before constraint assembly (none of them reads a constraint quantity), which shortens the LDS
// lifetimes of the per-body spatial arrays; every stage computes exactly what the reference does.
#ifdef MJH_PROFILE
// per-stage wall time of lane 0, accumulated per environment into the global field `prof` (us)
#define MJH_TIMED(t, call) do { long long c0_ = wv_clock(); call; \
    if (wv_lane() == 0) MJH_G(B, prof, e)[t] += (real)(wv_clock() - c0_) * 0.01; } while (0)
#else
#define MJH_TIMED(t, call) do { call; } while (0)
#endif
#define MJH_RUN(t, call) do { MJH_TIMED(t, call); if (stages & MJH_STAGE_WRITEBACK) lds_writeback(M, B, e, t); } while (0)
MJH_DEVN void forward(const DModel& M, const DBatch& B, int e, int stages) {
  const int pgs = (M.o.solver == MJH_SOL_PGS);
  if (stages & MJH_STAGE_KINEMATICS) {
    MJH_RUN(MJH_T_KIN, stage_kinematics(M, B, e));
    MJH_RUN(MJH_T_COMPOS, stage_compos(M, B, e));
    MJH_RUN(MJH_T_TENDON, stage_tendon(M, B, e));
  }
  if (stages & MJH_STAGE_INERTIA) {
    MJH_RUN(MJH_T_CRB, stage_crb(M, B, e));
    MJH_RUN(MJH_T_FACTOR, stage_factor_m(M, B, e));
  }
  if (stages & MJH_STAGE_COLLISION) MJH_RUN(MJH_T_COLLISION, stage_collision(M, B, e));
  if (stages & MJH_STAGE_TRANSMISSION) MJH_RUN(MJH_T_TRANSMISSION, stage_transmission(M, B, e));
  if (stages & MJH_STAGE_VELOCITY) {
    MJH_RUN(MJH_T_TAVEL, stage_ten_act_velocity(M, B, e));
    MJH_RUN(MJH_T_COMVEL, stage_comvel(M, B, e));
    MJH_RUN(MJH_T_PASSIVE, stage_passive(M, B, e));
    MJH_RUN(MJH_T_RNE, stage_rne(M, B, e));
  }
  if (stages & MJH_STAGE_ACTUATION) {
    MJH_RUN(MJH_T_ACTUATION, stage_actuation(M, B, e));
    MJH_RUN(MJH_T_ACCEL, stage_acceleration(M, B, e));
  }
  if (stages & MJH_STAGE_MAKE) MJH_RUN(MJH_T_MAKE, stage_make_constraint(M, B, e));
  if ((stages & MJH_STAGE_PROJECT) && pgs) MJH_RUN(MJH_T_PROJECT, stage_project(M, B, e));
  if (stages & MJH_STAGE_VELOCITY) MJH_RUN(MJH_T_REFERENCE, stage_reference(M, B, e));
  if (stages & MJH_STAGE_CONSTRAINT) MJH_RUN(MJH_T_CONSTRAINT, stage_fwd_constraint(M, B, e));
}

MJH_DEVN void euler_advance(const DModel& M, const DBatch& B, int e);
// body of the forward kernel (mjhip_batch_forward): stage-masked mj_forward (+ optional Euler step)
MJH_DEV void forward_or_euler(const DModel& M, const DBatch& B, int e, int stages) {
  lds_enter(M, B, e);
  forward(M, B, e, stages);
  if (stages & MJH_STAGE_EULER) {
    euler_advance(M, B, e);
    if (stages & MJH_STAGE_WRITEBACK) lds_writeback(M, B, e, MJH_T_EULER);
  }
  lds_exit(M, B, e);
}

// mj_EulerSkip + mj_advance                        (engine_forward.c:1398-1476, :1261-1395)
// The implicit-damping matrix qH = M + h*diag(B) is factorised in the slots of qLD/qLDiagInv,
// which are dead once the constraint solve has produced qacc; M itself was parked in the global
// field qH by stage_factor_m.
MJH_DEVN void euler_advance(const DModel& M, const DBatch& B, int e) {
  const DSizes& s = M.s;
  const int nv = s.nv;
  const real h = M.o.timestep;
  real* qvel = MJH_F(B, qvel, e);
  real* qpos = MJH_F(B, qpos, e);
  const real* qacc = MJH_F(B, qacc, e);
  real* qe = MJH_F(B, qe, e);               // integrated acceleration [nv]

  if (M.o.euler_damp) {
    const real* Mq = MJH_G(B, qH, e);
    real* qH = MJH_F(B, qLD, e);
    real* qHDiagInv = MJH_F(B, qLDiagInv, e);
    MJH_FOR_LANES(k, s.nC) qH[k] = Mq[k];
    wv_sync();
    MJH_FOR_LANES(i, nv) {
      real dd = poly_force_deriv(M.dof_damping_eff[i], M.dof_dampingpoly_eff + 2*i, qvel[i], 1);
      qH[M.M_rowadr[i] + M.M_rownnz[i] - 1] += h * dd;
    }
    wv_sync();
    factor_ld(M, qH, qHDiagInv);
    const real* fs = MJH_F(B, qfrc_smooth, e);
    const real* fc = MJH_F(B, qfrc_constraint, e);
    MJH_FOR_LANES(i, nv) qe[i] = fs[i] + fc[i];
    wv_sync();
    solve_ld(M, qe, qH, qHDiagInv);
  } else {
    MJH_FOR_LANES(i, nv) qe[i] = qacc[i];
    wv_sync();
  }

  // mj_advance: qvel += h*qacc ; qpos integrates the NEW qvel ; time ; warmstart
  MJH_FOR_LANES(i, nv) qvel[i] += qe[i]*h;
  wv_sync();
  MJH_FOR_LANES(j, s.njnt) {
    int padr = M.jnt_qposadr[j], vadr = M.jnt_dofadr[j];
    int jt = M.jnt_type[j];
    if (jt == MJH_JNT_FREE) {
      for (int i = 0; i < 3; i++) qpos[padr + i] += h * qvel[vadr + i];
      padr += 3; vadr += 3;
    }
    if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
      q_integrate(qpos + padr, qvel + vadr, h);
    } else {
      qpos[padr] += h * qvel[vadr];
    }
  }
  real* ws = MJH_F(B, qacc_warmstart, e);
  MJH_FOR_LANES(i, nv) ws[i] = qacc[i];
  if (wv_lane() == 0) MJH_F(B, time, e)[0] += h;
  wv_sync();
}

// mj_step                                          (engine_forward.c:1846-1880)
MJH_DEV void step_env(const DModel& M, const DBatch& B, int e) {
  check_bad(M, B, e, MJH_F(B, qpos, e), M.s.nq, MJH_WARN_BADQPOS);
  check_bad(M, B, e, MJH_F(B, qvel, e), M.s.nv, MJH_WARN_BADQVEL);
  for (int attempt = 0; attempt < 2; attempt++) {
    forward(M, B, e, MJH_STAGE_ALL);
    int bad = check_bad(M, B, e, MJH_F(B, qacc, e), M.s.nv, MJH_WARN_BADQACC);
    // bad qacc: state was reset; the reference re-runs mj_forward before integrating
    if (!bad || (M.o.disableflags & (1<<16))) break;
  }
  MJH_TIMED(MJH_T_EULER, euler_advance(M, B, e));
}

// pack FULLPHYSICS state [time, qpos, qvel, act]    (mj_getState, engine_support.c:214)
MJH_DEV void get_state(const DModel& M, const DBatch& B, int e, real* out) {
  const DSizes& s = M.s;
  if (wv_lane() == 0) out[0] = MJH_F(B, time, e)[0];
  const real* qpos = MJH_F(B, qpos, e);
  const real* qvel = MJH_F(B, qvel, e);
  const real* act = MJH_F(B, act, e);
  MJH_FOR_LANES(i, s.nq) out[1 + i] = qpos[i];
  MJH_FOR_LANES(i, s.nv) out[1 + s.nq + i] = qvel[i];
  MJH_FOR_LANES(i, s.na) out[1 + s.nq + s.nv + i] = act[i];
}

MJH_DEV void set_state(const DModel& M, const DBatch& B, int e, const real* in) {
  const DSizes& s = M.s;
  if (wv_lane() == 0) MJH_F(B, time, e)[0] = in[0];
  real* qpos = MJH_F(B, qpos, e);
  real* qvel = MJH_F(B, qvel, e);
  real* act = MJH_F(B, act, e);
  MJH_FOR_LANES(i, s.nq) qpos[i] = in[1 + i];
  MJH_FOR_LANES(i, s.nv) qvel[i] = in[1 + s.nq + i];
  MJH_FOR_LANES(i, s.na) act[i] = in[1 + s.nq + s.nv + i];
}

// arguments of the rollout kernel (device pointers; layouts of python/mujoco/rollout.cc:51-69)
struct RolloutArgs {
  int nstep;
  int has_ctrl;            // control_spec contains mjSTATE_CTRL
  int has_qfrc;            // control_spec contains mjSTATE_QFRC_APPLIED
  int ncontrol;            // mj_stateSize(control_spec)
  int qfrc_off;            // offset of qfrc_applied inside one control vector
  int init;                // 1: load state0/warmstart0, clear warnings (start of a rollout)
  const real* state0;      // [nenv][nstate]        or null
  const real* warmstart0;  // [nenv][nv]            or null -> zeros
  const real* control;     // [nenv][nstep][ncontrol] or null
  real* state;             // [nenv][nstep][nstate] or null
  int env_offset;          // first env of this launch inside state0/control/state
};

// _unsafe_rollout for one environment                (python/mujoco/rollout.cc:74-178)
MJH_DEV void rollout_env(const DModel& M, const DBatch& B, int e, const RolloutArgs& A) {
  const DSizes& s = M.s;
  const size_t r = (size_t)(A.env_offset + e);
  lds_enter(M, B, e);
  if (A.init) {
    if (A.state0) set_state(M, B, e, A.state0 + r*s.nstate);
    real* ws = MJH_F(B, qacc_warmstart, e);
    MJH_FOR_LANES(i, s.nv) ws[i] = A.warmstart0 ? A.warmstart0[r*s.nv + i] : 0;
    int* warn = MJH_F(B, warning, e);
    if (wv_lane() == 0) for (int k = 0; k < 8; k++) warn[k] = 0;
    if (!A.has_ctrl) { real* c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = 0; }
    if (!A.has_qfrc) { real* f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = 0; }
    wv_sync();
  }
  const int* warn = MJH_F(B, warning, e);
#ifdef MJH_PROFILE
  const long long c_start = wv_clock();
#endif
  for (int t = 0; t < A.nstep; t++) {
    // any warning freezes the trajectory: back-fill the rest with the current state (:135-155)
    int nw = 0;
    for (int k = 0; k < 8; k++) nw |= warn[k];
    const size_t step = r*(size_t)A.nstep + t;
    if (!nw) {
      if (A.control) {
        const real* u = A.control + step*A.ncontrol;
        if (A.has_ctrl) { real* c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = u[i]; }
        if (A.has_qfrc) { real* f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = u[A.qfrc_off + i]; }
        wv_sync();
      }
      step_env(M, B, e);
    }
    if (A.state) get_state(M, B, e, A.state + step*s.nstate);
    wv_sync();
  }
  lds_exit(M, B, e);
#ifdef MJH_PROFILE
  if (wv_lane() == 0) {
    real* pr = MJH_G(B, prof, e);
    const long long c_end = wv_clock();
    pr[31] += (real)(c_end - c_start) * 0.01; pr[30] += A.nstep;
    pr[28] = (real)c_start * 0.01; pr[29] = (real)c_end * 0.01;      // residency census (last launch)
#ifndef MJH_HOSTSIM
    pr[27] = (real)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_ID
    pr[26] = (real)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // XCC_ID
#endif
  }
#endif
}
