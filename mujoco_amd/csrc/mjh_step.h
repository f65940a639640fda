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
  real* xf = MJH_F(B, xfrc_applied, e);
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

// mj_forwardSkip(mjSTAGE_NONE, skipsensor) restricted by a stage mask   (engine_forward.c:1783-1836)
MJH_DEVN void forward(const DModel& M, const DBatch& B, int e, int stages) {
  if (stages & MJH_STAGE_KINEMATICS) {
    stage_kinematics(M, B, e);
    stage_compos(M, B, e);
    stage_tendon(M, B, e);
  }
  if (stages & MJH_STAGE_INERTIA) {
    stage_crb(M, B, e);
    stage_factor_m(M, B, e);
  }
  if (stages & MJH_STAGE_COLLISION) stage_collision(M, B, e);
  if (stages & MJH_STAGE_MAKE) stage_make_constraint(M, B, e);
  if ((stages & MJH_STAGE_PROJECT) && M.o.solver == MJH_SOL_PGS) stage_project(M, B, e);
  if (stages & MJH_STAGE_TRANSMISSION) stage_transmission(M, B, e);
  if (stages & MJH_STAGE_VELOCITY) {
    stage_ten_act_velocity(M, B, e);
    stage_comvel(M, B, e);
    stage_passive(M, B, e);
    stage_reference(M, B, e);
    stage_rne(M, B, e);
  }
  if (stages & MJH_STAGE_ACTUATION) {
    stage_actuation(M, B, e);
    stage_acceleration(M, B, e);
  }
  if (stages & MJH_STAGE_CONSTRAINT) stage_fwd_constraint(M, B, e);
}

MJH_DEVN void euler_advance(const DModel& M, const DBatch& B, int e);
MJH_DEV void forward_or_euler(const DModel& M, const DBatch& B, int e, int stages) {
  forward(M, B, e, stages);
  if (stages & MJH_STAGE_EULER) euler_advance(M, B, e);
}

// mj_EulerSkip + mj_advance                        (engine_forward.c:1398-1476, :1261-1395)
MJH_DEVN void euler_advance(const DModel& M, const DBatch& B, int e) {
  const DSizes& s = M.s;
  const int nv = s.nv;
  const real h = M.o.timestep;
  real* qvel = MJH_F(B, qvel, e);
  real* qpos = MJH_F(B, qpos, e);
  const real* qacc = MJH_F(B, qacc, e);
  real* qe = MJH_F(B, scratch, e);          // integrated acceleration [nv]

  if (M.o.euler_damp) {
    const real* Mq = MJH_F(B, M, e);
    real* qH = MJH_F(B, qH, e);
    MJH_FOR_LANES(k, s.nC) qH[k] = Mq[k];
    wv_sync();
    MJH_FOR_LANES(i, nv) {
      real dd = poly_force_deriv(M.dof_damping_eff[i], M.dof_dampingpoly_eff + 2*i, qvel[i], 1);
      qH[M.M_rowadr[i] + M.M_rownnz[i] - 1] += h * dd;
    }
    wv_sync();
    factor_ld(M, qH, MJH_F(B, qHDiagInv, e));
    const real* fs = MJH_F(B, qfrc_smooth, e);
    const real* fc = MJH_F(B, qfrc_constraint, e);
    MJH_FOR_LANES(i, nv) qe[i] = fs[i] + fc[i];
    wv_sync();
    solve_ld(M, qe, qH, MJH_F(B, qHDiagInv, e));
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
  euler_advance(M, B, e);
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
  if (A.init) {
    if (A.state0) set_state(M, B, e, A.state0 + r*s.nstate);
    real* ws = MJH_F(B, qacc_warmstart, e);
    MJH_FOR_LANES(i, s.nv) ws[i] = A.warmstart0 ? A.warmstart0[r*s.nv + i] : 0;
    int* warn = MJH_F(B, warning, e);
    if (wv_lane() == 0) for (int k = 0; k < 8; k++) warn[k] = 0;
    if (!A.has_ctrl) { real* c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = 0; }
    if (!A.has_qfrc) { real* f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = 0; }
    real* xf = MJH_F(B, xfrc_applied, e);
    MJH_FOR_LANES(i, 6*s.nbody) xf[i] = 0;
    wv_sync();
  }
  const int* warn = MJH_F(B, warning, e);
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
}
