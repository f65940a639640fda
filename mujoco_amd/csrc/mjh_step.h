// mj_step for one environment per wavefront: checks -> forward -> Euler/advance, and the rollout
// loop around it (python/mujoco/rollout.cc:74-178 restated per environment).
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)


// mocap poses back to the model's body_pos / body_quat (mj_resetData, engine_io.c)
// (positions and orientations are reset independently: rollout.cc:98-109 does so per control-spec bit)
MJH_DEV void reset_mocap(MREF M, BREF B, int e, int do_pos = 1, int do_quat = 1) {
  const MJH_CONST_AS DSizes& s = M.s;
  if (!MJH_HAS(MJH_FT_MOCAP) || !s.nmocap) return;
  rptr mp = MJH_G(B, mocap_pos, e);
  rptr mq = MJH_G(B, mocap_quat, e);
  MJH_FOR_LANES(i, s.nbody) {
    const int mid = M.body_mocapid[i];
    if (mid < 0) continue;
    if (do_pos) for (int k = 0; k < 3; k++) mp[3*mid + k] = M.body_pos[3*i + k];
    if (do_quat) for (int k = 0; k < 4; k++) mq[4*mid + k] = M.body_quat[4*i + k];
  }
}

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
  rptr ud = MJH_G(B, userdata, e);
  MJH_FOR_LANES(i, s.nuserdata) ud[i] = 0;
  reset_mocap(M, B, e);
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
MJH_DEV void lds_copy_in(T* g, int n, int l, int soa, int e, int cnt, char* lds) {
  if (l < 0) return;
  T* dst = (T*)(lds + l);
  SP<T> src = mjh_gp(g, n, soa, e);
  MJH_FOR_LANES(i, cnt) dst[i] = src[i];
}
template <class T>
MJH_DEV void lds_copy_out(T* g, int n, int l, int soa, int e, int cnt, char* lds) {
  if (l < 0) return;
  const T* src = (const T*)(lds + l);
  SP<T> dst = mjh_gp(g, n, soa, e);
  MJH_FOR_LANES(i, cnt) dst[i] = src[i];
}

// kernel entry: persistent state, global home -> LDS
MJH_DEV void lds_enter(MREF M, BREF B, int e) {
  if (!B.lds_bytes) return;
  const MJH_CONST_AS DSizes& s = M.s;
  (void)s;
#define X(name, cnt, lcnt, t0, t1) if (B.io_##name & 1) lds_copy_in(B.name, B.n_##name, B.l_##name, B.soa, e, (int)(lcnt), MJH_LDS(B));
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
#define X(name, cnt, lcnt, t0, t1) if (B.io_##name & 2) lds_copy_out(B.name, B.n_##name, B.l_##name, B.soa, e, (int)(lcnt), MJH_LDS(B));
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
#define X(name, cnt, lcnt, t0, t1) if ((t0) != MJH_T_GLB && (t0) <= t && t <= (t1)) lds_copy_out(B.name, B.n_##name, B.l_##name, B.soa, e, (int)(lcnt), MJH_LDS(B));
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
  const int pgs = !MJH_HAS(MJH_FT_PRIMAL) || (M.o.solver == MJH_SOL_PGS);
  if (stages & MJH_STAGE_KINEMATICS) {
    MJH_RUN(MJH_T_KIN, MJH_WIDE(MJH_MWS_KIN, { stage_kinematics(M, B, e); if (MJH_HAS(MJH_FT_FLEX) && M.s.nflex) stage_flex_pos(M, B, e); }));
  }
  // (collision needs nothing but the frames kinematics just produced)
  if (stages & MJH_STAGE_COLLISION) MJH_RUN(MJH_T_COLLISION, stage_collision(M, B, e));
  if (stages & MJH_STAGE_KINEMATICS) {
    MJH_RUN(MJH_T_COMPOS, MJH_WIDE(MJH_MWS_COMPOS, stage_compos(M, B, e)));
    MJH_RUN(MJH_T_TENDON, { stage_tendon(M, B, e); if (MJH_HAS(MJH_FT_FLEX) && M.s.nflexedge) MJH_WIDE(MJH_MWS_FLEXEDGES, stage_flex_edges(M, B, e)); });
  }
  if (stages & MJH_STAGE_TRANSMISSION) MJH_RUN(MJH_T_TRANSMISSION, stage_transmission(M, B, e));
  if (stages & MJH_STAGE_VELOCITY) {
    MJH_RUN(MJH_T_TAVEL, MJH_WIDE(MJH_MWS_TAVEL, stage_ten_act_velocity(M, B, e)));
    MJH_RUN(MJH_T_COMVEL, MJH_WIDE(MJH_MWS_COMVEL, stage_comvel(M, B, e)));
    MJH_RUN(MJH_T_PASSIVE, MJH_WIDE(MJH_MWS_PASSIVE, stage_passive(M, B, e)));
    MJH_RUN(MJH_T_RNE, MJH_WIDE(MJH_MWS_RNE, stage_rne(M, B, e)));
  }
  if (stages & MJH_STAGE_INERTIA) {
    // (models beyond the register-resident L'DL routines -- flexes: the generic, collective-free forms on every wavefront)
    MJH_RUN(MJH_T_CRB, MJH_WIDE_IF(!M.s.ld_fast, MJH_MWS_CRB, stage_crb(M, B, e, (stages & MJH_STAGE_NOPARK) != 0)));
    MJH_RUN(MJH_T_FACTOR, MJH_WIDE_IF(!M.s.ld_fast, MJH_MWS_FACTOR, stage_factor_m(M, B, e)));
  }
#if !MJH_LANE_MODE
  // implicit effective metric M + K of a flex model under CG with an implicit integrator (mjd_effBuild at the end of
  // mj_fwdPosition, mjd_effShift at the end of mj_fwdVelocity: both need nothing later than M here)
  if (MJH_HAS(MJH_FT_FLEX) && M.s.efm && (stages & MJH_STAGE_INERTIA)) MJH_TIMED(MJH_T_FACTOR, stage_eff_build(M, B, e));
#endif
  if (stages & MJH_STAGE_ACTUATION) {
    MJH_RUN(MJH_T_ACTUATION, stage_actuation(M, B, e));
    MJH_RUN(MJH_T_ACCEL, MJH_WIDE_IF(!M.s.ld_fast, MJH_MWS_ACCEL, stage_acceleration(M, B, e)));
#if !MJH_LANE_MODE
    // (M + K) qacc_smooth = qfrc_smooth + efm_c (mj_fwdAcceleration, engine_forward.c:1033-1040)
    if (MJH_HAS(MJH_FT_FLEX) && M.s.efm) MJH_TIMED(MJH_T_ACCEL, stage_eff_accel(M, B, e));
#endif
  }
  if (stages & MJH_STAGE_MAKE) {
    MJH_RUN(MJH_T_MAKE, stage_make_constraint(M, B, e));
    MJH_TIMED(49, {
#if !MJH_LANE_MODE
    if (MJH_HAS(MJH_FT_PRIMAL) && M.s.csr) stage_csr_rows(M, B, e);      // (the island scan reads the compressed rows)
#endif
    stage_island(M, B, e);
#if !MJH_LANE_MODE
    if (MJH_HAS(MJH_FT_PRIMAL) && M.s.sparse) stage_sparsify(M, B, e);
#endif
    });
  }
  // (mj_isDual, engine_core_constraint.c:167: efc_AR is built for the PGS solver and whenever the noslip pass will run)
  if ((stages & MJH_STAGE_PROJECT) && (pgs || (MJH_HAS(MJH_FT_PRIMAL) && M.o.noslip_iterations > 0))) MJH_RUN(MJH_T_PROJECT, stage_project(M, B, e));
  if (stages & MJH_STAGE_REFERENCE) MJH_RUN(MJH_T_REFERENCE, stage_reference(M, B, e));
  if (stages & MJH_STAGE_CONSTRAINT) MJH_RUN(MJH_T_CONSTRAINT, stage_fwd_constraint(M, B, e));
  if (stages & MJH_STAGE_FINISH) MJH_RUN(MJH_T_FINISH, stage_finish(M, B, e));
  if (MJH_HAS(MJH_FT_SENSOR) && (stages & (MJH_STAGE_SENSOR | MJH_STAGE_SENSPV | MJH_STAGE_SENSACC)) && M.s.nsensor)
    MJH_TIMED(52, stage_sensors(M, B, e, (stages & MJH_STAGE_SENSOR) ? 7 : (((stages & MJH_STAGE_SENSPV) ? 3 : 0) | ((stages & MJH_STAGE_SENSACC) ? 4 : 0))));
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
  // activations: X part [4][na], F part [4][na], combined act_dot [na]
  const int na = s.na;
  rptr act = MJH_F(B, act, e);
  crptr act_dot = MJH_F(B, act_dot, e);
  rptr XA = MJH_G(B, rk_act, e);
  rptr FA = XA + 4*na;
  rptr dA = XA + 8*na;
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
    MJH_FOR_LANES(k, na) { XA[k] = act[k]; FA[k] = act_dot[k]; }
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
    MJH_FOR_LANES(k, na) {
      real da = 0;
      for (int j = 0; j < i; j++) da += FA[j*na + k] * A[(i-1)*3 + j];
      XA[i*na + k] = XA[k] + da*h;
      act[k] = XA[i*na + k];
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
    MJH_FOR_LANES(k, na) FA[i*na + k] = act_dot[k];
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
  MJH_FOR_LANES(k, na) {
    real da = 0;
    for (int j = 0; j < 4; j++) da += FA[j*na + k] * Bw[j];
    dA[k] = da;
    act[k] = XA[k];
  }
  MJH_FOR_LANES(k, nq) qpos[k] = X[k];
  MJH_FOR_LANES(k, nv) qvel[k] = X[nq + k];
  wv_sync();
  advance_act(M, B, e, dA);
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
MJH_DEVN void implicitfast_advance(MREF M, BREF B, int e);
template <class P0> MJH_DEV int check_bad(MREF M, BREF B, int e, P0 x, int n, int which);
// body of the forward kernel (mjhip_batch_forward, mjhip_batch_step1/2): stage-masked mj_forward with
// the optional checks and integration of mj_step1 / mj_step2 (engine_forward.c:1884-1939)
MJH_DEV void forward_or_euler(MREF M, BREF B, int e, int stages) {
  // pipeline use: the constraint kernel skips environments frozen by a warning
  if ((stages & MJH_STAGE_IFACTIVE) && !MJH_G(B, active, e)[0]) return;
  lds_enter(M, B, e);
  if (stages & MJH_STAGE_CHECKPV) {
    check_bad(M, B, e, MJH_F(B, qpos, e), M.s.nq, MJH_WARN_BADQPOS);
    check_bad(M, B, e, MJH_F(B, qvel, e), M.s.nv, MJH_WARN_BADQVEL);
  }
  forward(M, B, e, stages);
  if (stages & MJH_STAGE_CHECKACC) check_bad(M, B, e, MJH_F(B, qacc, e), M.s.nv, MJH_WARN_BADQACC);
  if (stages & MJH_STAGE_INTEGRATE) {
    // mj_step2: implicit integrators as configured, everything else (RK4 included) is Euler
    if (MJH_HAS(MJH_FT_IMPLICIT) && M.o.integrator >= MJH_INT_IMPLICIT) implicitfast_advance(M, B, e);
    else euler_advance(M, B, e);
  } else if (stages & MJH_STAGE_EULER) {
    euler_advance(M, B, e);
    if (stages & MJH_STAGE_WRITEBACK) lds_writeback(M, B, e, MJH_T_EULER);
  }
  lds_exit(M, B, e);
}

// 3x3 blocks of d(qfrc_bias)/d(qvel) of a standalone free body: the rotational columns of the 6x6
// Jacobian are [-mass*lin ; rot]            (freeBias_vel_blocks, engine_derivative.c:711-786)
MJH_DEV void free_bias_blocks(real mass, const real* R, const real* Xi, const real* inertia,
                              const real* sv, const real* qvel_rot, real* lin, real* rot) {
  real w[3];
  m3_mulvec(w, R, qvel_rot);
  // world inertia about the COM: Xi * diag(inertia) * Xi'
  real XI[9], Iw[9];
  for (int i = 0; i < 3; i++) for (int c = 0; c < 3; c++) XI[3*i+c] = Xi[3*i+c]*inertia[c];
  Iw[0] = XI[0]*Xi[0] + XI[1]*Xi[1] + XI[2]*Xi[2];
  Iw[4] = XI[3]*Xi[3] + XI[4]*Xi[4] + XI[5]*Xi[5];
  Iw[8] = XI[6]*Xi[6] + XI[7]*Xi[7] + XI[8]*Xi[8];
  Iw[1] = Iw[3] = XI[0]*Xi[3] + XI[1]*Xi[4] + XI[2]*Xi[5];
  Iw[2] = Iw[6] = XI[0]*Xi[6] + XI[1]*Xi[7] + XI[2]*Xi[8];
  Iw[5] = Iw[7] = XI[3]*Xi[6] + XI[4]*Xi[7] + XI[5]*Xi[8];
  real ws[3], Iww[3];
  v3_cross(ws, w, sv);
  m3_mulvec(Iww, Iw, w);
  // K = s w' - (w.s) I + [w x s]_x
  const real wds = w[0]*sv[0] + w[1]*sv[1] + w[2]*sv[2];
  real K[9];
  K[0] = sv[0]*w[0] - wds;   K[1] = sv[0]*w[1] - ws[2];  K[2] = sv[0]*w[2] + ws[1];
  K[3] = sv[1]*w[0] + ws[2]; K[4] = sv[1]*w[1] - wds;    K[5] = sv[1]*w[2] - ws[0];
  K[6] = sv[2]*w[0] - ws[1]; K[7] = sv[2]*w[1] + ws[0];  K[8] = sv[2]*w[2] - wds;
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
    lin[3*r+c] = K[3*r]*R[c] + K[3*r+1]*R[3+c] + K[3*r+2]*R[6+c];
  // C = -mass [s]_x K + [w]_x Iw - [Iw w]_x
  real C[9];
  for (int c = 0; c < 3; c++) {
    real sk0 = sv[1]*K[6+c] - sv[2]*K[3+c];
    real sk1 = sv[2]*K[c] - sv[0]*K[6+c];
    real sk2 = sv[0]*K[3+c] - sv[1]*K[c];
    real wi0 = w[1]*Iw[6+c] - w[2]*Iw[3+c];
    real wi1 = w[2]*Iw[c] - w[0]*Iw[6+c];
    real wi2 = w[0]*Iw[3+c] - w[1]*Iw[c];
    C[c]     = -mass*sk0 + wi0 + (c == 1 ? Iww[2] : (c == 2 ? -Iww[1] : 0));
    C[3 + c] = -mass*sk1 + wi1 + (c == 0 ? -Iww[2] : (c == 2 ? Iww[0] : 0));
    C[6 + c] = -mass*sk2 + wi2 + (c == 0 ? Iww[1] : (c == 1 ? -Iww[0] : 0));
  }
  real T[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
    T[3*r+c] = R[r]*C[c] + R[3+r]*C[3+c] + R[6+r]*C[6+c];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
    rot[3*r+c] = T[3*r]*R[c] + T[3*r+1]*R[3+c] + T[3*r+2]*R[6+c];
}

// velocity derivative of one actuator's force, 0 when it does not contribute
// (mjd_actuator_vel, engine_derivative.c:2350-2490: affine gain/bias, stateless actuators)
MJH_DEV real actuator_vel_deriv(MREF M, int a, real force, real ctrl_or_act, real len, real vel) {
  if (M.actuator_forcelimited[a]) {
    if (force <= M.actuator_forcerange[2*a] || force >= M.actuator_forcerange[2*a+1]) return 0;
  }
  real bias_vel = 0, gain_vel = 0;
  if (M.actuator_biastype[a] == MJH_BIAS_AFFINE) bias_vel = M.actuator_biasprm[10*a + 2];
  if (M.actuator_gaintype[a] == MJH_GAIN_AFFINE) gain_vel = M.actuator_gainprm[10*a + 2];
  else if (M.actuator_gaintype[a] == MJH_GAIN_MUSCLE)
    gain_vel = muscle_gain(len, vel, M.actuator_lengthrange + 2*a, M.actuator_acc0[a], M.actuator_gainprm + 10*a, 1);
  if (gain_vel != 0) bias_vel += gain_vel * ctrl_or_act;
  return bias_vel;
}

// ------------------------------------------------------------------------------------------------
// The fully implicit integrator's solve (round 6).  mjd_smooth_vel with flg_bias (engine_derivative.c:3145-3166),
// mjd_rne_vel / mjd_comVel_vel (:527-720), mju_factorLUSparse / mju_solveLUSparse (engine_util_solve.c:938-1060).
//
// Mapping.  qDeriv's actuator / passive part: one lane per entry of the pattern D (the evaluator implicitfast uses on
// M's pattern).  The derivative of the bias force works on body-by-dof arrays of 6-vectors (pattern B: a body's row
// holds the dofs of its chain, its own, its subtree's).  The two forward passes go level by level, one lane per body
// of the level (a body reads its parent's row and writes its own and its dofs' rows); the backward accumulation of
// Dcfrcbody takes the bodies in descending order, as the reference does -- several children add into one parent row --,
// with the lanes over a row's components.  A 6 x 6 product of the reference (mju_mulMatMat with the transposed
// derivative matrix) is, per row of the work array, "matrix times that 6-vector" summed in component order with the
// vector's zero components skipped: mat6_apply.  The LU factorisation eliminates column by column from the last;
// the rows that hold the pivot's column are independent: one lane per such row.  First landing: correctness and the
// reference's rounding, not speed (a lane per body, global-memory work arrays).
// ------------------------------------------------------------------------------------------------
// out = A v with A row-major 6 x 6: out[c] = sum over k in order of A[c][k] v[k], terms with v[k] == 0 skipped
MJH_DEV void mat6_apply(real* out, const real* A, const real* v) {
  for (int c = 0; c < 6; c++) out[c] = 0;
  for (int k = 0; k < 6; k++) {
    const real t = v[k];
    if (t) for (int c = 0; c < 6; c++) out[c] += A[6*c + k]*t;
  }
}
// 3 x 3 block of D at (r0, c0): sgn [b]x, the matrix of the cross product b x . (sgn = -1: of . x b)
MJH_DEV void mat6_skew(real* D, int r0, int c0, const real* b, real sgn) {
  D[6*r0 + c0 + 1] = -sgn*b[2];       D[6*r0 + c0 + 2] = sgn*b[1];
  D[6*(r0 + 1) + c0] = sgn*b[2];      D[6*(r0 + 1) + c0 + 2] = -sgn*b[0];
  D[6*(r0 + 2) + c0] = -sgn*b[1];     D[6*(r0 + 2) + c0 + 1] = sgn*b[0];
}
// d crossMotion(vel, v) / d vel = [[-[v_ang]x, 0], [-[v_lin]x, -[v_ang]x]]     (mjd_crossMotion_vel)
MJH_DEV void d_cross_motion_vel(real* D, const real* v) {
  for (int k = 0; k < 36; k++) D[k] = 0;
  mat6_skew(D, 0, 0, v, -1); mat6_skew(D, 3, 0, v + 3, -1); mat6_skew(D, 3, 3, v, -1);
}
// d crossForce(vel, f) / d vel = [[-[f_ang]x, -[f_lin]x], [-[f_lin]x, 0]]       (mjd_crossForce_vel)
MJH_DEV void d_cross_force_vel(real* D, const real* f) {
  for (int k = 0; k < 36; k++) D[k] = 0;
  mat6_skew(D, 0, 0, f, -1); mat6_skew(D, 0, 3, f + 3, -1); mat6_skew(D, 3, 0, f + 3, -1);
}
// d crossForce(vel, f) / d f = [[[vel_ang]x, [vel_lin]x], [0, [vel_ang]x]]       (mjd_crossForce_frc)
MJH_DEV void d_cross_force_frc(real* D, const real* vel) {
  for (int k = 0; k < 36; k++) D[k] = 0;
  mat6_skew(D, 0, 0, vel, 1); mat6_skew(D, 0, 3, vel + 3, 1); mat6_skew(D, 3, 3, vel, 1);
}
// the spatial inertia as a 6 x 6 matrix (mjd_mulInertVec_vel): [[I, [c]x], [-[c]x, m 1]] from the ten numbers of cinert
MJH_DEV void d_mul_inert_vec(real* D, const real* ci) {
  for (int k = 0; k < 36; k++) D[k] = 0;
  D[0] = ci[0]; D[1] = ci[3]; D[2] = ci[4];
  D[6] = ci[3]; D[7] = ci[1]; D[8] = ci[5];
  D[12] = ci[4]; D[13] = ci[5]; D[14] = ci[2];
  mat6_skew(D, 0, 3, ci + 6, 1); mat6_skew(D, 3, 0, ci + 6, -1);
  D[21] = ci[9]; D[28] = ci[9]; D[35] = ci[9];
}

template <class QE>
MJH_DEV void implicit_full_solve(MREF M, BREF B, int e, QE& qderiv_entry) {
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv, nD = s.nD, nB = s.nB;
  const real h = M.o.timestep;
  rptr qD = MJH_G(B, qDeriv, e);
  rptr qLU = MJH_G(B, qLU, e);
  rptr Dd = MJH_G(B, Dcdofdot, e);
  rptr Dv = MJH_G(B, Dcvel, e);
  rptr Da = MJH_G(B, Dcacc, e);
  rptr Df = MJH_G(B, Dcfrcbody, e);
  crptr cdof = MJH_F(B, cdof, e);
  crptr cdofdot = MJH_F(B, cdof_dot, e);
  crptr cvel = MJH_F(B, cvel, e);
  crptr cinert = MJH_F(B, cinert, e);
  crptr qvel = MJH_F(B, qvel, e);
  crptr Mq = MJH_G(B, M, e);
  rptr qe = MJH_F(B, qe, e);

  // ---- qDeriv = d(qfrc_actuator + qfrc_passive) / d(qvel); work arrays cleared
  MJH_FOR_LANES(k, nD) qD[k] = qderiv_entry((int)M.D_rowid[k], (int)M.D_colind[k]);
  MJH_FOR_LANES(k, 6*nD) Dd[k] = 0;
  MJH_FOR_LANES(k, 6*nB) { Dv[k] = 0; Da[k] = 0; Df[k] = 0; }
  wv_sync();

  // ---- mjd_comVel_vel: Dcvel (per body) and Dcdofdot (per dof), parents before children
  for (int L = 1; L < s.nlevel; L++) {
    const int a0 = M.body_level_adr[L], a1 = M.body_level_adr[L + 1];
    MJH_FOR_LANES(kb, a1 - a0) {
      const int i = M.body_level_ids[a0 + kb];
      const int bi = M.B_rowadr[i];
      { const int nc = 6*M.B_ncopy[i], bp = M.B_rowadr[M.body_parentid[i]]; for (int t = 0; t < nc; t++) Dv[6*bi + t] = Dv[6*bp + t]; }
      const int d0 = M.body_dofadr[i], d1 = d0 + M.body_dofnum[i];
      for (int j = d0; j < d1; j++) {
        int Jadr = M.D_diag[j];                    // (number of dof ancestors of dof j; M's own pattern drops the zero couplings of simple dofs)
        const int jt = M.jnt_type[M.dof_jntid[j]];
        real mat[36], v6[6], o6[6];
        if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
          if (jt == MJH_JNT_FREE) {
            for (int q = 0; q < 3; q++) for (int c = 0; c < 6; c++) Dv[6*(bi + Jadr + q) + c] += cdof[6*(j + q) + c];
            j += 3; Jadr += 3;
          }
          for (int dj = 0; dj < 3; dj++) {
            for (int c = 0; c < 6; c++) v6[c] = cdof[6*(j + dj) + c];
            d_cross_motion_vel(mat, v6);
            const int da = M.D_rowadr[j + dj];
            for (int t = 0; t < Jadr + dj; t++) {
              for (int c = 0; c < 6; c++) v6[c] = Dv[6*(bi + t) + c];
              mat6_apply(o6, mat, v6);
              for (int c = 0; c < 6; c++) Dd[6*(da + t) + c] = o6[c];
            }
          }
          for (int q = 0; q < 3; q++) for (int c = 0; c < 6; c++) Dv[6*(bi + Jadr + q) + c] += cdof[6*(j + q) + c];
          j += 2;
        } else {
          for (int c = 0; c < 6; c++) v6[c] = cdof[6*j + c];
          d_cross_motion_vel(mat, v6);
          const int da = M.D_rowadr[j];
          for (int t = 0; t < Jadr; t++) {
            for (int c = 0; c < 6; c++) v6[c] = Dv[6*(bi + t) + c];
            mat6_apply(o6, mat, v6);
            for (int c = 0; c < 6; c++) Dd[6*(da + t) + c] = o6[c];
          }
          for (int c = 0; c < 6; c++) Dv[6*(bi + Jadr) + c] += cdof[6*j + c];
        }
      }
    }
    wv_sync();
  }

  // ---- forward pass of mjd_rne_vel: Dcacc, then Dcfrcbody = D(cinert cacc + cvel x* (cinert cvel))
  for (int L = 1; L < s.nlevel; L++) {
    const int a0 = M.body_level_adr[L], a1 = M.body_level_adr[L + 1];
    MJH_FOR_LANES(kb, a1 - a0) {
      const int i = M.body_level_ids[a0 + kb];
      const int bi = M.B_rowadr[i], bn = M.B_rownnz[i];
      { const int nc = 6*M.B_ncopy[i], bp = M.B_rowadr[M.body_parentid[i]]; for (int t = 0; t < nc; t++) Da[6*bi + t] = Da[6*bp + t]; }
      const int d0 = M.body_dofadr[i], d1 = d0 + M.body_dofnum[i];
      for (int j = d0; j < d1; j++) {
        const int Jadr = M.D_diag[j];
        for (int c = 0; c < 6; c++) Da[6*(bi + Jadr) + c] += cdofdot[6*j + c];
        const int da = M.D_rowadr[j];
        const real qv = qvel[j];
        for (int t = 0; t < 6*bn; t++) Da[6*bi + t] += Dd[6*da + t]*qv;
      }
      real dmul[36], mat[36], mat1[36], ci[10], cv[6], tmp[6], v6[6], o6[6];
      for (int c = 0; c < 10; c++) ci[c] = cinert[10*i + c];
      for (int c = 0; c < 6; c++) cv[c] = cvel[6*i + c];
      d_mul_inert_vec(dmul, ci);
      for (int t = 0; t < bn; t++) {
        for (int c = 0; c < 6; c++) v6[c] = Da[6*(bi + t) + c];
        mat6_apply(o6, dmul, v6);
        for (int c = 0; c < 6; c++) Df[6*(bi + t) + c] = o6[c];
      }
      sp_mul_inert(tmp, ci, cv);
      d_cross_force_vel(mat, tmp);
      d_cross_force_frc(mat1, cv);
      // mat += mat1 dmul  (mju_mulMatMat: row i of the product = sum over k of dmul's row k times mat1[i][k], zero mat1[i][k] skipped)
      for (int r = 0; r < 6; r++) {
        real acc[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 6; k++) { const real t = mat1[6*r + k]; if (t) for (int c = 0; c < 6; c++) acc[c] += dmul[6*k + c]*t; }
        for (int c = 0; c < 6; c++) mat[6*r + c] += acc[c];
      }
      for (int t = 0; t < bn; t++) {
        for (int c = 0; c < 6; c++) v6[c] = Dv[6*(bi + t) + c];
        mat6_apply(o6, mat, v6);
        for (int c = 0; c < 6; c++) Df[6*(bi + t) + c] += o6[c];
      }
    }
    wv_sync();
  }

  // ---- backward pass: a body's row joins its parent's (bodies in descending order, as the reference adds them)
  for (int b = s.nbody - 1; b > 0; b--) {
    const int bb = M.B_rowadr[b], bn = M.B_rownnz[b], bp = M.B_rowadr[M.body_parentid[b]];
    if (bn == 0 || M.B_pmap[bb] < 0) continue;
    MJH_FOR_LANES(w, 6*bn) {
      const int t = w / 6, c = w - 6*t;
      Df[6*(bp + M.B_pmap[bb + t]) + c] += Df[6*(bb + t) + c];
    }
    wv_sync();
  }

  // ---- qDeriv -= D(cdof . cfrc_body): entry t of dof j's row takes mju_dot(Dcfrcbody[body of j][t], cdof_j, 6)
  MJH_FOR_LANES(k, nD) {
    const int j = M.D_rowid[k], t = k - M.D_rowadr[j];
    const int bi = M.B_rowadr[M.dof_bodyid[j]];
    real a[6], c6[6];
    for (int c = 0; c < 6; c++) { a[c] = Df[6*(bi + t) + c]; c6[c] = cdof[6*j + c]; }
    const real r0 = a[0]*c6[0], r1 = a[1]*c6[1], r2 = a[2]*c6[2], r3 = a[3]*c6[3];
    real res = (r0 + r2) + (r1 + r3);
    res += a[4]*c6[4] + a[5]*c6[5];
    qD[k] -= res;
  }
  wv_sync();

  // ---- qLU = M (lower to full) - h qDeriv
  MJH_FOR_LANES(k, nD) {
    const int im = M.D_mapM[k];
    real v = im >= 0 ? (real)Mq[im] : 0;
    v += qD[k]*(-h);
    qLU[k] = v;
  }
  wv_sync();

  // ---- mju_factorLUSparse: pivots from the last row; row j < i holds column i iff its last remaining entry is i
  //      (rows above a pivot are independent of one another)
  for (int i = nv - 1; i >= 0; i--) {
    const int ra = M.D_rowadr[i], di = M.D_diag[i];
    const real piv = qLU[ra + di];
    MJH_FOR_LANES(j, i) {
      // (entries of row j beyond column i have been eliminated: what remains of it ends at the last column <= i)
      const int rj = M.D_rowadr[j], nj = M.D_rownnz[j];
      int last = nj - 1;
      while (last >= 0 && M.D_colind[rj + last] > i) last--;
      if (last < 0 || M.D_colind[rj + last] != i) continue;
      const real lji = qLU[rj + last] / piv;
      qLU[rj + last] = lji;
      int ic = ra;
      for (int jc = rj; jc < rj + last; jc++) {
        // (row i's remaining entries, columns < i, are a subset of row j's: the reference stops on fill-in)
        if (ic < ra + di && M.D_colind[ic] == M.D_colind[jc]) { qLU[jc] -= qLU[ic]*lji; ic++; }
      }
    }
    wv_sync();
  }

  // ---- mju_solveLUSparse on qfrc_smooth + qfrc_constraint: (U + I) y = b from the last row, L x = y from the first;
  //      one lane walks the rows (every row needs the ones before it), the row's sparse dot is mju_dotSparse's
  crptr fs = MJH_F(B, qfrc_smooth, e);
  crptr fc = MJH_F(B, qfrc_constraint, e);
  MJH_FOR_LANES(i, nv) qe[i] = fs[i] + fc[i];
  wv_sync();
  if (wv_lane() == 0) {
    for (int i = nv - 1; i >= 0; i--) {
      const int d1 = M.D_diag[i] + 1, nn = M.D_rownnz[i] - d1, adr = M.D_rowadr[i] + d1;
      real r = qe[i];
      if (nn > 0) r -= dot_sparse_ref(qLU + adr, qe, nn, M.D_colind + adr);
      qe[i] = r;
    }
    for (int i = 0; i < nv; i++) {
      const int d = M.D_diag[i], adr = M.D_rowadr[i];
      real r = qe[i];
      if (d > 0) r -= dot_sparse_ref(qLU + adr, qe, d, M.D_colind + adr);
      qe[i] = r / qLU[adr + d];
    }
  }
  wv_sync();
}

// mj_implicitSkip (implicitfast branch) + mj_advance        (engine_forward.c:1649-1770)
// qH = M - h * lower(qDeriv), qDeriv = d(qfrc_actuator + qfrc_passive)/d(qvel) accumulated per
// M entry in the reference's order (actuators, dof damping, tendon damping); standalone free
// bodies keep their rows of M and get the unsymmetric 6x6 solve with the gyroscopic derivative.
MJH_DEVN void implicitfast_advance(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv;
  const real h = M.o.timestep;
  const int dsbl = M.o.disableflags;
  rptr qvel = MJH_F(B, qvel, e);
  rptr qpos = MJH_F(B, qpos, e);
  crptr qacc = MJH_F(B, qacc, e);
  rptr qe = MJH_F(B, qe, e);
  crptr Mq = MJH_G(B, M, e);
  rptr qH = MJH_F(B, qLD, e);
  rptr qHDiagInv = MJH_F(B, qLDiagInv, e);
  crptr mom = MJH_F(B, actuator_moment, e);
  ciptr mnnz = MJH_F(B, moment_rownnz, e);
  ciptr mcol = MJH_F(B, moment_colind, e);
  crptr force = MJH_F(B, actuator_force, e);
  crptr ctrl = MJH_F(B, ctrl, e);
  crptr actv = MJH_F(B, act, e);
  crptr actd = MJH_F(B, act_dot, e);
  crptr tJ = MJH_F(B, ten_J, e);
  crptr tvel = MJH_F(B, ten_velocity, e);
  crptr alen = MJH_F(B, actuator_length, e);
  crptr avel = MJH_F(B, actuator_velocity, e);
  const int act_on = !(dsbl & (1<<11)) && s.nu > 0;
  const int passive_on = !((dsbl & (1<<5)) && (dsbl & (1<<6)));
  const int damper_on = passive_on && !(dsbl & (1<<6));
  const int fluid_on = MJH_HAS(MJH_FT_PASSIVEMISC) && passive_on && M.o.has_fluid && (M.o.viscosity > 0 || M.o.density > 0);
  crptr cdof = MJH_F(B, cdof, e);
  crptr com = MJH_F(B, subtree_com, e);
  crptr xpos = MJH_F(B, xpos, e);
  crptr xipos = MJH_F(B, xipos, e);
  crptr xmat = MJH_F(B, xmat, e);
  crptr ximat = MJH_F(B, ximat, e);

  // qDeriv(i, j) = d(qfrc_actuator + qfrc_passive)(i) / d(qvel)(j), accumulated in the reference's
  // order: actuators, dof damping, tendon damping  (mjd_actuator_vel, mjd_passive_vel; an entry of
  // J'BJ is J(j) * (J(i) * B), addJTBJSparse engine_derivative.c:934-965)
  auto qderiv_entry = [&](int i, int j) -> real {
    real q = 0;
    if (act_on) {
      for (int a = 0; a < s.nu; a++) {
        if (M.o.has_act_disabled && M.actuator_disabled[a]) continue;      // (mjd_actuator_vel skips disabled actuators, :2365)
        // the gain multiplies ctrl, or the (next, if actearly) activation (:2475-2490)
        real u = ctrl[a];
        if (M.actuator_dyntype[a] != MJH_DYN_NONE) {
          const int aa = M.actuator_actadr[a];
          u = M.actuator_actearly[a] ? next_activation(M, a, actv[aa], actd[aa]) : (real)actv[aa];
        }
        real bv = actuator_vel_deriv(M, a, force[a], u, alen[a], avel[a]);
        if (bv == 0) continue;
        const int adr = M.actuator_momentadr[a];
        real mi = 0, mj = 0;
        int hi = 0, hj = 0;
        for (int c = 0; c < mnnz[a]; c++) {
          if (mcol[adr + c] == i) { mi = mom[adr + c]; hi = 1; }
          if (mcol[adr + c] == j) { mj = mom[adr + c]; hj = 1; }
        }
        if (hi && hj) q += mj * (mi*bv);
      }
    }
    // fluid forces, inertia-box model (mjd_passive_vel :3051-3072 -> mjd_inertiaBoxFluid :2884-3038): per body with mass, J'BJ
    // of the body's COM Jacobian rotated into the inertial frame (mju_mulMatTMat: the three terms of a row in order, zero
    // entries of ximat skipped), one scalar B per local axis -- viscous torque (rows 0-2), viscous force (3-5), then the
    // quadratic drag on rows 0..5; the coefficients were left next to the fluid wrench by the passive stage
    if (fluid_on) {
      crptr bf = MJH_G(B, fluid_frc, e);
      crptr cdi = cdof + 6*i, cdj = cdof + 6*j;
      for (int b = 0; b < s.nbody; b++) {
        if (M.body_mass[b] < MJH_MINVAL) continue;
        if (!((M.body_dofanc[b*s.nvw + (i >> 5)] >> (i & 31)) & 1) || !((M.body_dofanc[b*s.nvw + (j >> 5)] >> (j & 31)) & 1)) continue;
        if (s.ngeom_fluid && M.body_ellipsoid[b]) {
          // ellipsoid model (mjd_ellipsoidFluid :2778-2880): per geom J' B J with the geom's Jacobian (rotation rows, then
          // translation rows, rotated into the geom frame) and the full 6 x 6 B the passive stage left in fluid_geom;
          // addJTBJ's order: rows a of B, columns c, zero entries of B skipped
          crptr gx = MJH_F(B, geom_xpos, e);
          crptr gmat = MJH_F(B, geom_xmat, e);
          crptr gw = MJH_G(B, fluid_geom, e);
          for (int g = M.body_geomadr[b]; g < M.body_geomadr[b] + M.body_geomnum[b]; g++) {
            if (M.geom_fluid[12*g] == 0) continue;
            real off[3], ci[3], cj[3], Ji[6], Jj[6];
            v3_sub(off, gx + 3*g, com + 3*M.body_rootid[b]);
            v3_cross(ci, cdi, off);
            v3_cross(cj, cdj, off);
            const real gi[6] = {cdi[0], cdi[1], cdi[2], cdi[3] + ci[0], cdi[4] + ci[1], cdi[5] + ci[2]};
            const real gj[6] = {cdj[0], cdj[1], cdj[2], cdj[3] + cj[0], cdj[4] + cj[1], cdj[5] + cj[2]};
            crptr xm = gmat + 9*g;
            for (int r = 0; r < 3; r++) {
              real ai = 0, aj = 0, li = 0, lj = 0;
              for (int c = 0; c < 3; c++) {
                const real t = xm[3*c + r];
                if (t) { ai += gi[c]*t; aj += gj[c]*t; li += gi[3 + c]*t; lj += gj[3 + c]*t; }
              }
              Ji[r] = ai; Jj[r] = aj; Ji[3 + r] = li; Jj[3 + r] = lj;
            }
            crptr Bm = gw + 42*g + 6;
            for (int a = 0; a < 6; a++)
              for (int c = 0; c < 6; c++) {
                const real bv = Bm[6*a + c];
                if (bv) q += Jj[c]*(Ji[a]*bv);
              }
          }
          continue;
        }
        crptr cf = bf + 6*s.nbody + 8*b;
        real off[3], ci[3], cj[3], Ji[6], Jj[6];
        v3_sub(off, xipos + 3*b, com + 3*M.body_rootid[b]);
        v3_cross(ci, cdi, off);
        v3_cross(cj, cdj, off);
        const real gi[6] = {cdi[0], cdi[1], cdi[2], cdi[3] + ci[0], cdi[4] + ci[1], cdi[5] + ci[2]};
        const real gj[6] = {cdj[0], cdj[1], cdj[2], cdj[3] + cj[0], cdj[4] + cj[1], cdj[5] + cj[2]};
        crptr xm = ximat + 9*b;
        for (int r = 0; r < 3; r++) {
          real ai = 0, aj = 0, li = 0, lj = 0;
          for (int c = 0; c < 3; c++) {
            const real t = xm[3*c + r];
            if (t) { ai += gi[c]*t; aj += gj[c]*t; li += gi[3 + c]*t; lj += gj[3 + c]*t; }
          }
          Ji[r] = ai; Jj[r] = aj; Ji[3 + r] = li; Jj[3 + r] = lj;
        }
        if (M.o.viscosity > 0) {
          for (int r = 0; r < 3; r++) if (cf[0]) q += Jj[r]*(Ji[r]*cf[0]);
          for (int r = 3; r < 6; r++) if (cf[1]) q += Jj[r]*(Ji[r]*cf[1]);
        }
        if (M.o.density > 0)
          for (int r = 0; r < 6; r++) if (cf[2 + r]) q += Jj[r]*(Ji[r]*cf[2 + r]);
      }
    }
    if (damper_on) {
      if (i == j) q -= poly_force_deriv(M.dof_damping_eff[i], M.dof_dampingpoly_eff + 2*i, qvel[i], 1);
      if (MJH_HAS(MJH_FT_FLEX) && s.nflexdof) {
        // flex edge damping (mjd_passive_vel :3090-3111): J' B J of every non-rigid edge with B = -flex_edgedamping, flex by
        // flex, edge by edge; the edges that hold dof i come from the by-column table (ascending edge order)
        crptr fJ = MJH_F(B, flexedge_J, e);
        for (int a = M.flexJ_cscadr[i]; a < M.flexJ_cscadr[i + 1]; a++) {
          const int ai = M.flexJ_cscind[a], ed = M.flexedge_J_rowid[ai];
          const int f = M.flexedge_flex[ed];
          const real Bf = -M.flex_edgedamping[f];
          if (M.flex_rigid[f] || !Bf || M.flexedge_rigid[ed]) continue;
          const int adr = M.flexedge_J_rowadr[ed], nn = M.flexedge_J_rownnz[ed];
          for (int c = 0; c < nn; c++) if (M.flexedge_J_colind[adr + c] == j) q += fJ[adr + c]*(fJ[ai]*Bf);
        }
      }
      for (int t = 0; t < s.ntendon; t++) {
        real dp[2] = {M.tendon_dampingpoly_eff[2*t], M.tendon_dampingpoly_eff[2*t+1]};
        real bt = -poly_force_deriv(M.tendon_damping_eff[t], dp, tvel[t], 1);
        if (bt == 0) continue;
        const int adr = M.ten_J_rowadr[t];
        real ji = 0, jj = 0;
        int hi = 0, hj = 0;
        for (int c = 0; c < M.ten_J_rownnz[t]; c++) {
          if (M.ten_J_colind[adr + c] == i) { ji = tJ[adr + c]; hi = 1; }
          if (M.ten_J_colind[adr + c] == j) { jj = tJ[adr + c]; hj = 1; }
        }
        if (hi && hj) q += jj * (ji*bt);
      }
    }
    return q;
  };

  if (M.o.integrator == MJH_INT_IMPLICIT) {
    // ---- the fully implicit integrator (mj_implicitSkip :1680-1690, :1718-1733): qDeriv on its own pattern with the
    //      derivative of the bias force, qLU = M - h qDeriv factorised without pivoting, one sparse LU solve
    implicit_full_solve(M, B, e, qderiv_entry);
    advance_act(M, B, e, MJH_F(B, act_dot, e));
    MJH_FOR_LANES(i, nv) qvel[i] += qe[i]*h;
    wv_sync();
    integrate_pos(M, qpos, qvel, h);
    rptr ws2 = MJH_F(B, qacc_warmstart, e);
    MJH_FOR_LANES(i, nv) ws2[i] = qacc[i];
    if (wv_lane() == 0) MJH_F(B, time, e)[0] += h;
    wv_sync();
    return;
  }
  MJH_FOR_LANES(k, s.nC) {
    const int i = M.M_rowid[k], j = M.M_colind[k];
    const real q = qderiv_entry(i, j);
    // rows of standalone free bodies stay M
    const int jn = M.dof_jntid[i];
    if (M.jnt_freebody[jn]) qH[k] = Mq[k];
    else qH[k] = Mq[k] + q*(-h);
  }
  wv_sync();
  factor_ld(M, qH, qHDiagInv);
  crptr fs = MJH_F(B, qfrc_smooth, e);
  crptr fc = MJH_F(B, qfrc_constraint, e);
  rptr qf = MJH_F(B, rk_F, e);             // qfrc = qfrc_smooth + qfrc_constraint (kept for the 6x6 solves)
  MJH_FOR_LANES(i, nv) { real f = fs[i] + fc[i]; qf[i] = f; qe[i] = f; }
  wv_sync();
  solve_ld(M, qe, qH, qHDiagInv);

  // standalone free bodies: A = M_block - h*qDeriv_block + h*d(bias)/dv, LU with partial pivoting
  // (mjd_freeMhat engine_derivative.c:844-893, mju_factorLU6/solveLU6 engine_util_solve.c:857-931)
  MJH_FOR_LANES(jn, s.njnt) {
    if (!M.jnt_freebody[jn]) continue;
    const int b = M.jnt_bodyid[jn], adr = M.jnt_dofadr[jn];
    real A[36];
    for (int k = 0; k < 36; k++) A[k] = 0;
    for (int r = 0; r < 6; r++) {
      const int ra = M.M_rowadr[adr + r];
      for (int k = 0; k < M.M_rownnz[adr + r]; k++) {
        int c = M.M_colind[ra + k] - adr;
        A[6*r + c] = Mq[ra + k];
        A[6*c + r] = Mq[ra + k];
      }
    }
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) A[6*r + c] -= h * qderiv_entry(adr + r, adr + c);
    real sv[3], R[9], Xi[9], inr[3], wq[3], lin[9], rot[9];
    for (int k = 0; k < 3; k++) { sv[k] = xipos[3*b + k] - xpos[3*b + k]; inr[k] = M.body_inertia[3*b + k]; wq[k] = qvel[adr + 3 + k]; }
    for (int k = 0; k < 9; k++) { R[k] = xmat[9*b + k]; Xi[k] = ximat[9*b + k]; }
    const real mass = M.body_mass[b];
    free_bias_blocks(mass, R, Xi, inr, sv, wq, lin, rot);
    const real hm = -h * mass;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
      A[6*r + 3 + c] += hm * lin[3*r + c];
      A[6*(3 + r) + 3 + c] += h * rot[3*r + c];
    }
    // LU with row pivoting
    int piv[6];
    int ok = 1;
    for (int k = 0; k < 6 && ok; k++) {
      piv[k] = k;
      real mx = fabs(A[6*k + k]);
      int mr = k;
      for (int i = k + 1; i < 6; i++) { real v = fabs(A[6*i + k]); if (v > mx) { mx = v; mr = i; } }
      if (mx < MJH_MINVAL) { ok = 0; break; }
      if (mr != k) {
        piv[k] = mr;
        for (int c = 0; c < 6; c++) { real t = A[6*k + c]; A[6*k + c] = A[6*mr + c]; A[6*mr + c] = t; }
      }
      const real di = 1.0 / A[6*k + k];
      for (int i = k + 1; i < 6; i++) {
        A[6*i + k] *= di;
        const real aik = A[6*i + k];
        for (int c = k + 1; c < 6; c++) A[6*i + c] -= aik * A[6*k + c];
      }
    }
    if (ok) {
      real x[6];
      for (int i = 0; i < 6; i++) x[i] = qf[adr + i];
      for (int i = 0; i < 6; i++) {
        if (piv[i] != i) { real t = x[i]; x[i] = x[piv[i]]; x[piv[i]] = t; }
        for (int c = 0; c < i; c++) x[i] -= A[6*i + c] * x[c];
      }
      for (int i = 5; i >= 0; i--) {
        for (int c = i + 1; c < 6; c++) x[i] -= A[6*i + c] * x[c];
        x[i] /= A[6*i + i];
      }
      for (int i = 0; i < 6; i++) qe[adr + i] = x[i];
    }
  }
  wv_sync();

  advance_act(M, B, e, MJH_F(B, act_dot, e));
  MJH_FOR_LANES(i, nv) qvel[i] += qe[i]*h;
  wv_sync();
  integrate_pos(M, qpos, qvel, h);
  rptr ws = MJH_F(B, qacc_warmstart, e);
  MJH_FOR_LANES(i, nv) ws[i] = qacc[i];
  if (wv_lane() == 0) MJH_F(B, time, e)[0] += h;
  wv_sync();
}

// mj_implicitSkip under mj_flexCG (engine_forward.c:1675, :1727-1729, :1766): the constraint solver's qacc already carries
// the implicit flex force -- no qDeriv, no factorisation: mj_advance with qacc itself
MJH_DEVN void flexcg_advance(MREF M_, BREF B_, int e_) {
  MJH_ENTER(M_, B_, e_);
  const int nv = M.s.nv;
  const real h = M.o.timestep;
  rptr qvel = MJH_F(B, qvel, e);
  rptr qpos = MJH_F(B, qpos, e);
  crptr qacc = MJH_F(B, qacc, e);
  advance_act(M, B, e, MJH_F(B, act_dot, e));
  MJH_FOR_LANES(i, nv) qvel[i] += qacc[i]*h;
  wv_sync();
  integrate_pos(M, qpos, qvel, h);
  rptr ws = MJH_F(B, qacc_warmstart, e);
  MJH_FOR_LANES(i, nv) ws[i] = qacc[i];
  if (wv_lane() == 0) MJH_F(B, time, e)[0] += h;
  wv_sync();
}

// mj_step                                          (engine_forward.c:1846-1880)
MJH_DEV void step_env(MREF M, BREF B, int e) {
  // (profile builds, slots 48..52: state checks | compressed rows + islands | state / sensor output | control input | sensors)
  MJH_TIMED(48, { check_bad(M, B, e, MJH_F(B, qpos, e), M.s.nq, MJH_WARN_BADQPOS);
                  check_bad(M, B, e, MJH_F(B, qvel, e), M.s.nv, MJH_WARN_BADQVEL); });
  for (int attempt = 0; attempt < 2; attempt++) {
    forward(M, B, e, MJH_STAGE_ALL | MJH_STAGE_SENSOR | MJH_STAGE_NOPARK);
    int bad;
    MJH_TIMED(48, bad = check_bad(M, B, e, MJH_F(B, qacc, e), M.s.nv, MJH_WARN_BADQACC));
    // bad qacc: state was reset; the reference re-runs mj_forward before integrating
    if (!bad || (M.o.disableflags & (1<<16))) break;
  }
  if (MJH_HAS(MJH_FT_RK4) && M.o.integrator == MJH_INT_RK4) MJH_TIMED(MJH_T_EULER, rk4_advance(M, B, e));
  else if (MJH_HAS(MJH_FT_IMPLICIT) && MJH_HAS(MJH_FT_FLEX) && M.s.efm) MJH_TIMED(MJH_T_EULER, flexcg_advance(M, B, e));
  else if (MJH_HAS(MJH_FT_IMPLICIT) && M.o.integrator >= MJH_INT_IMPLICIT) MJH_TIMED(MJH_T_EULER, implicitfast_advance(M, B, e));
  else MJH_TIMED(MJH_T_EULER, MJH_WIDE_IF(!M.s.ld_fast, MJH_MWS_EULER, euler_advance(M, B, e)));
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

// user inputs that are not part of the control spec are cleared / reset at the start of a rollout
// (python/mujoco/rollout.cc:85-115); inputs that are part of it keep their current values until
// the control array (if any) overwrites them
MJH_DEV void rollout_clear_inputs(MREF M, BREF B, int e, const RolloutArgs& A) {
  const MJH_CONST_AS DSizes& s = M.s;
  if (!A.has_ctrl) { rptr c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = 0; }
  if (!A.has_qfrc) { rptr f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = 0; }
  if (A.xfrc_off < 0) { rptr x = MJH_G(B, xfrc_applied, e); MJH_FOR_LANES(i, 6*s.nbody) x[i] = 0; }
  if (A.mpos_off < 0 || A.mquat_off < 0) reset_mocap(M, B, e, A.mpos_off < 0, A.mquat_off < 0);
  if (A.eq_off < 0) { iptr q = MJH_G(B, eq_active, e); MJH_FOR_LANES(i, s.neq) q[i] = M.eq_active0[i]; }
}

// one control vector -> the user-input fields, in mjtState bit order (mj_setState, engine_support.c:282)
MJH_DEV void rollout_load_control(MREF M, BREF B, int e, const RolloutArgs& A, const real* u) {
  const MJH_CONST_AS DSizes& s = M.s;
  if (A.has_ctrl) { rptr c = MJH_F(B, ctrl, e); MJH_FOR_LANES(i, s.nu) c[i] = u[i]; }
  if (A.has_qfrc) { rptr f = MJH_F(B, qfrc_applied, e); MJH_FOR_LANES(i, s.nv) f[i] = u[A.qfrc_off + i]; }
  if (A.xfrc_off >= 0) { rptr x = MJH_G(B, xfrc_applied, e); MJH_FOR_LANES(i, 6*s.nbody) x[i] = u[A.xfrc_off + i]; }
  // eq_active is a byte array in mjData: the state value is converted like the reference's assignment
  if (A.eq_off >= 0) { iptr q = MJH_G(B, eq_active, e); MJH_FOR_LANES(i, s.neq) q[i] = (int)(unsigned char)u[A.eq_off + i]; }
  if (A.mpos_off >= 0) { rptr p = MJH_G(B, mocap_pos, e); MJH_FOR_LANES(i, 3*s.nmocap) p[i] = u[A.mpos_off + i]; }
  if (A.mquat_off >= 0) { rptr q = MJH_G(B, mocap_quat, e); MJH_FOR_LANES(i, 4*s.nmocap) q[i] = u[A.mquat_off + i]; }
  if (A.ud_off >= 0) { rptr d = MJH_G(B, userdata, e); MJH_FOR_LANES(i, s.nuserdata) d[i] = u[A.ud_off + i]; }
}

// _unsafe_rollout for one environment                (python/mujoco/rollout.cc:74-178)
MJH_DEV void rollout_env(MREF M_, BREF B_, int e, const RolloutArgs& A) {
  MREF M = M_;
  BREF B = B_;
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
    rollout_clear_inputs(M, B, e, A);
    wv_sync();
  }
  ciptr warn = MJH_F(B, warning, e);
#ifdef MJH_PROFILE
  const long long c_start = wv_clock();
#endif
  int work = A.tbase > 0 ? wv_uniform_i(MJH_G(B, cost, e)[0]) : 0;      // (a later chunk of one host rollout continues the count)
#if MJH_STEP_PRIO && !MJH_LANE_MODE
  // Issue priority (s_setprio: the SIMD's arbiter serves the higher level first).  A launch ends with its slowest
  // wavefront, and the slowest ones are environments in long solves that stay expensive for many consecutive steps; while
  // such a wavefront shares its SIMD with three cheap environments it should run at its own speed, the cheap ones filling
  // the gaps -- the SIMD's total work does not change, the launch's tail does (profiles/r05/step_priority_ab.txt: +8..10 %
  // on the humanoid).  Levels: by the solver work of the step just taken against the batch's mean work per step of the
  // previous launch (prio_ref, written by mjh_k_balance: 2x / 4x / 6x the mean); on the first step by the position in the
  // launch order, which lists the environments by decreasing cost of the previous launch.  No reference yet: no priorities.
  int prio_ref = 0, prio_floor = 0;
  (void)prio_floor;
  {
    ciptr pr = MJH_G(B, prio_ref, 0);
    const int tot = wv_uniform_i(pr[0]), ns = wv_uniform_i(pr[1]);
    prio_ref = (ns > 0 && (int)blockDim.x == MJH_WAVE) ? tot/ns : 0;      // (one-wavefront workgroups only: a multi-wavefront
                                                                           //  environment's helpers would fall behind its wave 0)
#ifndef MJH_PRIO_MODE
#define MJH_PRIO_MODE 0
#endif
    if (prio_ref > 0 && !A.nlaunch) {
      const int w = (int)blockIdx.x, n = B.nenv;
#if MJH_PRIO_MODE == 0
      if (w*64 < n) __builtin_amdgcn_s_setprio(3);
      else if (w*16 < n) __builtin_amdgcn_s_setprio(2);
      else if (w*4 < n) __builtin_amdgcn_s_setprio(1);
#else
      // (measurement builds: the launch order deals one environment of each cost quartile to every SIMD -- quartile = level)
      if (w*4 < n) { prio_floor = 3; __builtin_amdgcn_s_setprio(3); }
      else if (w*2 < n) { prio_floor = 2; __builtin_amdgcn_s_setprio(2); }
      else if (w*4 < 3*n) { prio_floor = 1; __builtin_amdgcn_s_setprio(1); }
#endif
    }
  }
#endif
  for (int t = 0; t < A.nstep; t++) {
#if !defined(MJH_HOSTSIM) && !defined(MJH_NO_LAUNDER)
    // The two descriptors are re-read through "new" pointers every step.  With every stage inlined the compiler
    // otherwise hoists hundreds of loop-invariant table pointers and sizes out of the step loop, keeps them in SGPRs
    // across all of it, runs out, and spills them (and, through the spill lanes, VGPRs) to scratch: re-loading a
    // descriptor word from the scalar cache where it is used is cheaper than carrying it.
    const MJH_CONST_AS DModel* Mp_ = &M_;
    const MJH_CONST_AS DBatch* Bp_ = &B_;
    asm volatile("" : "+s"(Mp_), "+s"(Bp_));
    MREF M = *Mp_;
    BREF B = *Bp_;
#else
    MREF M = M_;
    BREF B = B_;
#endif
    // any warning freezes the trajectory: back-fill the rest with the current state (:135-155)
    int nw = 0;
    for (int k = 0; k < 8; k++) nw |= warn[k];
    const size_t step = r*(size_t)A.pitch + A.tbase + t;
    if (!nw) {
      if (A.control) {
        MJH_TIMED(51, { rollout_load_control(M, B, e, A, A.control + step*A.ncontrol); wv_sync(); });
      }
      step_env(M, B, e);
      ciptr cnt = MJH_F(B, counts, e);
      work += 64 + cnt[MJH_C_NEFC]*(cnt[MJH_C_NITER] + 4);
#if MJH_STEP_PRIO && !MJH_LANE_MODE
#if MJH_PRIO_MODE == 1
      if (false)
#endif
      if (prio_ref > 0) {
        const int wk = wv_uniform_i(cnt[MJH_C_NEFC]*(cnt[MJH_C_NITER] + 4));
        // (thresholds in quarters of the batch's mean work per step: 8 / 16 / 24 = 2x / 4x / 6x; -DMJH_PRIO_Q1.. for A/B builds)
#ifndef MJH_PRIO_Q1
#define MJH_PRIO_Q1 8
#define MJH_PRIO_Q2 16
#define MJH_PRIO_Q3 24
#endif
        int lvl = 4*wk >= MJH_PRIO_Q3*prio_ref ? 3 : (4*wk >= MJH_PRIO_Q2*prio_ref ? 2 : (4*wk >= MJH_PRIO_Q1*prio_ref ? 1 : 0));
#if MJH_PRIO_MODE == 2
        lvl = lvl > prio_floor ? lvl : prio_floor;
#endif
        if (lvl == 3) __builtin_amdgcn_s_setprio(3);
        else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
        else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
#if MJH_PRIO_MODE != 1
      else __builtin_amdgcn_s_setprio(0);      // (no reference level: undo what a long PGS solve raised, mjh_solver.h solve_pgs_fast)
#endif
#endif
    }
    MJH_TIMED(50, {
    if (A.state) get_state(M, B, e, A.state + step*s.nstate);
    if (A.sensordata) {
      crptr sd = MJH_G(B, sensordata, e);
      MJH_FOR_LANES(i, s.nsensordata) A.sensordata[step*s.nsensordata + i] = sd[i];
    }
    wv_sync();
    });
  }
  lds_exit(M, B, e);
  // what this environment cost: the next launch deals the environments to the SIMDs by it
  // (mjh_k_balance); wall time of the wavefront for the tail statistics (tools/tail_stats.py)
  if (wv_lane() == 0) {
    MJH_G(B, cost, e)[0] = work;
    MJH_G(B, wall, e)[0] = (int)((wv_clock() - c_begin) >> 4);
  }
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
    rollout_clear_inputs(M, B, e, A);
    wv_sync();
  }
  // any warning freezes the trajectory (python/mujoco/rollout.cc:135-155)
  ciptr warn = MJH_F(B, warning, e);
  int nw = 0;
  for (int k = 0; k < 8; k++) nw |= warn[k];
  if (wv_lane() == 0) MJH_G(B, active, e)[0] = nw ? 0 : 1;
  if (nw) return;
  if (A.control) {
    rollout_load_control(M, B, e, A, A.control + (r*(size_t)A.nstep + A.t0)*A.ncontrol);
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
    if (MJH_HAS(MJH_FT_RK4) && M.o.integrator == MJH_INT_RK4) MJH_TIMED(MJH_T_EULER, rk4_advance(M, B, e));
    else if (MJH_HAS(MJH_FT_IMPLICIT) && MJH_HAS(MJH_FT_FLEX) && M.s.efm) MJH_TIMED(MJH_T_EULER, flexcg_advance(M, B, e));
  else if (MJH_HAS(MJH_FT_IMPLICIT) && M.o.integrator >= MJH_INT_IMPLICIT) MJH_TIMED(MJH_T_EULER, implicitfast_advance(M, B, e));
    else MJH_TIMED(MJH_T_EULER, MJH_WIDE_IF(!M.s.ld_fast, MJH_MWS_EULER, euler_advance(M, B, e)));
  }
  if (A.state) get_state(M, B, e, A.state + (r*(size_t)A.nstep + A.t0)*s.nstate);
}
