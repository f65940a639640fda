// mj_step for one environment per wavefront: checks -> forward -> Euler/advance, and the rollout
// loop around it (python/mujoco/rollout.cc:74-178 restated per environment).
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)


// mj_resetData as far as the state vector is concerned (engine_io.c:1289-1420)
MJH_DEV void reset_env(MREF M, BREF B, int e) {
  const MJH_CONST_AS DSizes& s = M.s;
  rptr qpos = MJH_F(B, qpos, e);
  MJH_FOR_LANES(i, s.nq) qpos[i] = M.qpos0[i];
  rptr qvel = MJH_F(B, qvel, e);
  rptr ws = MJH_F(B, qacc_warmstart, e);
  rptr fa = MJH_F(B, qfrc_applied, e);
  MJH_FOR_LANES(i, s.nv) { qvel[i] = 0; ws[i] = 0; fa[i] = 0; }
  rptr act = MJH_F(B, act, e);
  MJH_FOR_LANES(i, s.na) act[i] = 0;
  rptr ctrl = MJH_F(B, ctrl, e);
  MJH_FOR_LANES(i, s.nu) ctrl[i] = 0;
  rptr xf = MJH_G(B, xfrc_applied, e);
  MJH_FOR_LANES(i, 6*s.nbody) xf[i] = 0;
  iptr eqa = MJH_G(B, eq_active, e);
  MJH_FOR_LANES(i, s.neq) eqa[i] = M.eq_active0[i];
  iptr warn = MJH_F(B, warning, e);
  if (wv_lane() == 0) {
    MJH_F(B, time, e)[0] = 0;
    for (int k = 0; k < 8; k++) warn[k] = 0;
  }
  wv_sync();
}

// mj_checkPos / mj_checkVel / mj_checkAcc (engine_forward.c:54-113): returns 1 if x had a bad value
template <class P0>
MJH_DEV int check_bad(MREF M, BREF B, int e, P0 x, int n, int which) {
  int bad = 0;
  MJH_FOR_LANES(i, n) if (r_isbad(x[i])) bad = 1;
  bad = wv_any(bad);
  if (bad) {
    wv_sync();
    if (!(M.o.disableflags & (1<<16))) reset_env(M, B, e);
    iptr warn = MJH_F(B, warning, e);
    if (wv_lane() == 0) warn[which] += 1;
    wv_sync();
  }
  return bad;
}

// ---- LDS residency helpers -----------------------------------------------------------------------
// copy a field between its global home and its LDS slot (no-op for fields the plan left global)
template <class T>
MJH_DEV void lds_copy_in(T* g, int n, int l, int soa, int e, int cnt) {
  if (l < 0) return;
  T* dst = (T*)(mjh_lds() + l);
  SP<T> src = mjh_gp(g, n, soa, e);
  MJH_FOR_LANES(i, cnt) dst[i] = src[i];
}
template <class T>
MJH_DEV void lds_copy_out(T* g, int n, int l, int soa, int e, int cnt) {
  if (l < 0) return;
  const T* src = (const T*)(mjh_lds() + l);
  SP<T> dst = mjh_gp(g, n, soa, e);
  MJH_FOR_LANES(i, cnt) dst[i] = src[i];
}

// kernel entry: persistent state, global home -> LDS
MJH_DEV void lds_enter(MREF M, BREF B, int e) {
  if (!B.lds_bytes) return;
  const MJH_CONST_AS DSizes& s = M.s;
  (void)s;
#define X(name, cnt, lcnt, t0, t1) if (B.io_##name & 1) lds_copy_in(B.name, B.n_##name, B.l_##name, B.soa, e, (int)(lcnt));
  MJH_BATCH_REAL_FIELDS(X)
  MJH_BATCH_INT_FIELDS(X)
#undef X
  wv_sync();
}
// kernel exit: persistent + exported fields, LDS -> global home
MJH_DEV void lds_exit(MREF M, BREF B, int e) {
  if (!B.lds_bytes) return;
  const MJH_CONST_AS DSizes& s = M.s;
  (void)s;
  wv_sync();
#define X(name, cnt, lcnt, t0, t1) if (B.io_##name & 2) lds_copy_out(B.name, B.n_##name, B.l_##name, B.soa, e, (int)(lcnt));
  MJH_BATCH_REAL_FIELDS(X)
  MJH_BATCH_INT_FIELDS(X)
#undef X
  wv_sync();
}
// debug write-back (tests): after timeline point t, every LDS-resident field live at t is copied to
// its global home, so the host can inspect intermediates of the LDS path field by field
MJH_DEVN void lds_writeback(MREF M_, BREF B_, int e_, int t) {
  MJH_ENTER(M_, B_, e_);
  if (!B.lds_bytes) return;
  const MJH_CONST_AS DSizes& s = M.s;
  (void)s;
  wv_sync();
#define X(name, cnt, lcnt, t0, t1) if ((t0) != MJH_T_GLB && (t0) <= t && t <= (t1)) lds_copy_out(B.name, B.n_##name, B.l_##name, B.soa, e, (int)(lcnt));
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
MJH_DEVN void forward(MREF M_, BREF B_, int e_, int stages) {
  MJH_ENTER(M_, B_, e_);
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
  if (stages & MJH_STAGE_COLLISION) MJH_RUN(MJH_T_COLLISION, stage_collision(M, B, e));
  if (stages & MJH_STAGE_MAKE) {
    MJH_RUN(MJH_T_MAKE, stage_make_constraint(M, B, e));
    stage_island(M, B, e);
  }
  if ((stages & MJH_STAGE_PROJECT) && pgs) MJH_RUN(MJH_T_PROJECT, stage_project(M, B, e));
  if (stages & MJH_STAGE_REFERENCE) MJH_RUN(MJH_T_REFERENCE, stage_reference(M, B, e));
  if (stages & MJH_STAGE_CONSTRAINT) MJH_RUN(MJH_T_CONSTRAINT, stage_fwd_constraint(M, B, e));
  if (stages & MJH_STAGE_FINISH) MJH_RUN(MJH_T_FINISH, stage_finish(M, B, e));
}

// mj_integratePos: qpos <- qpos (+) qvel*h, joint by joint            (engine_support.c:639-690)
template <class P0, class P1>
MJH_DEV void integrate_pos(MREF M, P0 qpos, P1 qvel, real h) {
  MJH_FOR_LANES(j, M.s.njnt) {
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
}

MJH_DEVN void forward(MREF M_, BREF B_, int e_, int stages);

// mj_RungeKutta(m, d, 4) + mj_advance                 (engine_forward.c:1486-1587, :1261-1395)
// mj_forward for stage 0 has already run (mj_step); the three further evaluations warm-start from
// the same qacc_warmstart, which only mj_advance overwrites.
MJH_DEVN void rk4_advance(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nq = s.nq, nv = s.nv, nx = nq + nv;
  const real h = M.o.timestep;
  const real A[9] = {0.5, 0, 0,  0, 0.5, 0,  0, 0, 1};
  const real Bw[4] = {1.0/6.0, 1.0/3.0, 1.0/3.0, 1.0/6.0};
  rptr qpos = MJH_F(B, qpos, e);
  rptr qvel = MJH_F(B, qvel, e);
  rptr tm = MJH_F(B, time, e);
  rptr X = MJH_G(B, rk_X, e);
  rptr F = MJH_G(B, rk_F, e);
  rptr dX = MJH_G(B, rk_dX, e);
  const real time0 = tm[0];
  real T[3];
  for (int i = 1; i < 4; i++) {
    real C = 0;
    for (int j = 0; j < i; j++) C += A[(i-1)*3 + j];
    T[i-1] = time0 + C*h;
  }
  {
    crptr qacc = MJH_F(B, qacc, e);
    MJH_FOR_LANES(k, nq) X[k] = qpos[k];
    MJH_FOR_LANES(k, nv) { X[nq + k] = qvel[k]; F[k] = qacc[k]; }
    wv_sync();
  }
  for (int i = 1; i < 4; i++) {
    // dX = sum_j A(i,j) * (X[j].qvel, F[j])   (every j < i is added, zero weights included)
    MJH_FOR_LANES(k, nv) {
      real dv = 0, da = 0;
      for (int j = 0; j < i; j++) {
        dv += X[j*nx + nq + k] * A[(i-1)*3 + j];
        da += F[j*nv + k] * A[(i-1)*3 + j];
      }
      dX[k] = dv;
      dX[nv + k] = da;
    }
    // X[i] = X[0] (+) dX
    MJH_FOR_LANES(k, nx) X[i*nx + k] = X[k];
    wv_sync();
    integrate_pos(M, X + i*nx, dX, h);
    MJH_FOR_LANES(k, nv) X[i*nx + nq + k] += dX[nv + k] * h;
    wv_sync();
    MJH_FOR_LANES(k, nq) qpos[k] = X[i*nx + k];
    MJH_FOR_LANES(k, nv) qvel[k] = X[i*nx + nq + k];
    if (wv_lane() == 0) tm[0] = T[i-1];
    wv_sync();
    forward(M, B, e, MJH_STAGE_ALL);
    crptr qacc = MJH_F(B, qacc, e);
    MJH_FOR_LANES(k, nv) F[i*nv + k] = qacc[k];
    wv_sync();
  }
  // final combination with B, state reset, mj_advance
  MJH_FOR_LANES(k, nv) {
    real dv = 0, da = 0;
    for (int j = 0; j < 4; j++) {
      dv += X[j*nx + nq + k] * Bw[j];
      da += F[j*nv + k] * Bw[j];
    }
    dX[k] = dv;
    dX[nv + k] = da;
  }
  MJH_FOR_LANES(k, nq) qpos[k] = X[k];
  MJH_FOR_LANES(k, nv) qvel[k] = X[nq + k];
  wv_sync();
  MJH_FOR_LANES(k, nv) qvel[k] += dX[nv + k] * h;
  wv_sync();
  integrate_pos(M, qpos, dX, h);
  rptr ws = MJH_F(B, qacc_warmstart, e);
  crptr qacc = MJH_F(B, qacc, e);
  MJH_FOR_LANES(k, nv) ws[k] = qacc[k];
  if (wv_lane() == 0) tm[0] = time0 + h;
  wv_sync();
}

MJH_DEVN void euler_advance(MREF M, BREF B, int e);
// body of the forward kernel (mjhip_batch_forward): stage-masked mj_forward (+ optional Euler step)
MJH_DEV void forward_or_euler(MREF M, BREF B, int e, int stages) {
  // pipeline use: the constraint kernel skips environments frozen by a warning
  if ((stages & MJH_STAGE_IFACTIVE) && !MJH_G(B, active, e)[0]) return;
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
MJH_DEVN void euler_advance(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  const real h = M.o.timestep;
  rptr qvel = MJH_F(B, qvel, e);
  rptr qpos = MJH_F(B, qpos, e);
  crptr qacc = MJH_F(B, qacc, e);
  rptr qe = MJH_F(B, qe, e);               // integrated acceleration [nv]

  if (M.o.euler_damp) {
    crptr Mq = MJH_G(B, qH, e);
    rptr qH = MJH_F(B, qLD, e);
    rptr qHDiagInv = MJH_F(B, qLDiagInv, e);
    MJH_FOR_LANES(k, s.nC) qH[k] = Mq[k];
    wv_sync();
    MJH_FOR_LANES(i, nv) {
      real dd = poly_force_deriv(M.dof_damping_eff[i], M.dof_dampingpoly_eff + 2*i, qvel[i], 1);
      qH[M.M_rowadr[i] + M.M_rownnz[i] - 1] += h * dd;
    }
    wv_sync();
    factor_ld(M, qH, qHDiagInv);
    crptr fs = MJH_F(B, qfrc_smooth, e);
    crptr fc = MJH_F(B, qfrc_constraint, e);
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
  integrate_pos(M, qpos, qvel, h);
  rptr ws = MJH_F(B, qacc_warmstart, e);
  MJH_FOR_LANES(i, nv) ws[i] = qacc[i];
  if (wv_lane() == 0) MJH_F(B, time, e)[0] += h;
  wv_sync();
}

// mj_step                                          (engine_forward.c:1846-1880)
MJH_DEV void step_env(MREF M, BREF B, int e) {
  check_bad(M, B, e, MJH_F(B, qpos, e), M.s.nq, MJH_WARN_BADQPOS);
  check_bad(M, B, e, MJH_F(B, qvel, e), M.s.nv, MJH_WARN_BADQVEL);
  for (int attempt = 0; attempt < 2; attempt++) {
    forward(M, B, e, MJH_STAGE_ALL);
    int bad = check_bad(M, B, e, MJH_F(B, qacc, e), M.s.nv, MJH_WARN_BADQACC);
    // bad qacc: state was reset; the reference re-runs mj_forward before integrating
    if (!bad || (M.o.disableflags & (1<<16))) break;
  }
  if (M.o.integrator == MJH_INT_RK4) MJH_TIMED(MJH_T_EULER, rk4_advance(M, B, e));
  else MJH_TIMED(MJH_T_EULER, euler_advance(M, B, e));
}

// pack FULLPHYSICS state [time, qpos, qvel, act]    (mj_getState, engine_support.c:214)
template <class P0>
MJH_DEV void get_state(MREF M, BREF B, int e, P0 out) {
  const MJH_CONST_AS DSizes& s = M.s;
  if (wv_lane() == 0) out[0] = MJH_F(B, time, e)[0];
  crptr qpos = MJH_F(B, qpos, e);
  crptr qvel = MJH_F(B, qvel, e);
  crptr act = MJH_F(B, act, e);
  MJH_FOR_LANES(i, s.nq) out[1 + i] = qpos[i];
  MJH_FOR_LANES(i, s.nv) out[1 + s.nq + i] = qvel[i];
  MJH_FOR_LANES(i, s.na) out[1 + s.nq + s.nv + i] = act[i];
}

template <class P0>
MJH_DEV void set_state(MREF M, BREF B, int e, P0 in) {
  const MJH_CONST_AS DSizes& s = M.s;
  if (wv_lane() == 0) MJH_F(B, time, e)[0] = in[0];
  rptr qpos = MJH_F(B, qpos, e);
  rptr qvel = MJH_F(B, qvel, e);
  rptr act = MJH_F(B, act, e);
  MJH_FOR_LANES(i, s.nq) qpos[i] = in[1 + i];
  MJH_FOR_LANES(i, s.nv) qvel[i] = in[1 + s.nq + i];
  MJH_FOR_LANES(i, s.na) act[i] = in[1 + s.nq + s.nv + i];
}

// _unsafe_rollout for one environment                (python/mujoco/rollout.cc:74-178)
MJH_DEV void rollout_env(MREF M, BREF B, int e, const RolloutArgs& A) {
  const MJH_CONST_AS DSizes& s = M.s;
  const size_t r = (size_t)(A.env_offset + e);
  const long long c_begin = wv_clock();
  lds_enter(M, B, e);
  if (A.init) {
    if (A.state0) set_state(M, B, e, A.state0 + r*s.nstate);
    rptr ws = MJH_F(B, qacc_warmstart, e);
    MJH_FOR_LANES(i, s.nv) ws[i] = A.warmstart0 ? A.warmstart0[r*s.nv + i] : 0;
    iptr warn = MJH_F(B, warning, e);
    if (wv_lane() == 0) for (int k = 0; k < 8; k++) warn[k] = 0;
    if (!A.has_ctrl) { rptr c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = 0; }
    if (!A.has_qfrc) { rptr f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = 0; }
    wv_sync();
  }
  ciptr warn = MJH_F(B, warning, e);
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
        if (A.has_ctrl) { rptr c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = u[i]; }
        if (A.has_qfrc) { rptr f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = u[A.qfrc_off + i]; }
        wv_sync();
      }
      step_env(M, B, e);
    }
    if (A.state) get_state(M, B, e, A.state + step*s.nstate);
    wv_sync();
  }
  lds_exit(M, B, e);
  // what this environment cost: the next launch starts the expensive ones first (mjh_k_balance)
  if (wv_lane() == 0) MJH_G(B, cost, e)[0] = (int)((wv_clock() - c_begin) >> 4);
#ifdef MJH_PROFILE
  if (wv_lane() == 0) {
    rptr pr = MJH_G(B, prof, e);
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


// ------------------------------------------------------------------------------------------------
// The per-step pipeline (SoA batches): one mj_step of every environment is three kernels
//   smooth_env     (lane mode)  rollout prologue, checks, every constraint-free stage
//   forward_or_euler(wave mode) collision .. PGS with MJH_STAGE_IFACTIVE   (kernel mjh_k_forward)
//   integrate_env  (lane mode)  qacc, checkAcc, Euler, advance, state output
// Together they are rollout_env's loop body; A.t0 is the step's index in control/state.
// ------------------------------------------------------------------------------------------------

MJH_DEV void smooth_env(MREF M, BREF B, int e, const RolloutArgs& A) {
  const MJH_CONST_AS DSizes& s = M.s;
  const size_t r = (size_t)(A.env_offset + e);
  if (A.init) {
    if (A.state0) set_state(M, B, e, A.state0 + r*s.nstate);
    rptr ws = MJH_F(B, qacc_warmstart, e);
    MJH_FOR_LANES(i, s.nv) ws[i] = A.warmstart0 ? A.warmstart0[r*s.nv + i] : 0;
    iptr warn = MJH_F(B, warning, e);
    if (wv_lane() == 0) for (int k = 0; k < 8; k++) warn[k] = 0;
    if (!A.has_ctrl) { rptr c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = 0; }
    if (!A.has_qfrc) { rptr f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = 0; }
    wv_sync();
  }
  // any warning freezes the trajectory (python/mujoco/rollout.cc:135-155)
  ciptr warn = MJH_F(B, warning, e);
  int nw = 0;
  for (int k = 0; k < 8; k++) nw |= warn[k];
  if (wv_lane() == 0) MJH_G(B, active, e)[0] = nw ? 0 : 1;
  if (nw) return;
  if (A.control) {
    const real* u = A.control + (r*(size_t)A.nstep + A.t0)*A.ncontrol;
    if (A.has_ctrl) { rptr c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = u[i]; }
    if (A.has_qfrc) { rptr f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = u[A.qfrc_off + i]; }
    wv_sync();
  }
  check_bad(M, B, e, MJH_F(B, qpos, e), s.nq, MJH_WARN_BADQPOS);
  check_bad(M, B, e, MJH_F(B, qvel, e), s.nv, MJH_WARN_BADQVEL);
  forward(M, B, e, MJH_STAGES_SMOOTH_MASK);
}

MJH_DEV void integrate_env(MREF M, BREF B, int e, const RolloutArgs& A) {
  const MJH_CONST_AS DSizes& s = M.s;
  const size_t r = (size_t)(A.env_offset + e);
  if (MJH_G(B, active, e)[0]) {
    stage_finish(M, B, e);
    int bad = check_bad(M, B, e, MJH_F(B, qacc, e), s.nv, MJH_WARN_BADQACC);
    // bad qacc: the state was reset; the reference re-runs mj_forward before integrating
    // (engine_forward.c:1863-1870).  Rare, so the whole forward pass is redone right here.
    if (bad && !(M.o.disableflags & (1<<16))) forward(M, B, e, MJH_STAGE_ALL);
    if (M.o.integrator == MJH_INT_RK4) MJH_TIMED(MJH_T_EULER, rk4_advance(M, B, e));
    else MJH_TIMED(MJH_T_EULER, euler_advance(M, B, e));
  }
  if (A.state) get_state(M, B, e, A.state + (r*(size_t)A.nstep + A.t0)*s.nstate);
}
