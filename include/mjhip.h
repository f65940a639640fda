/* mjhip -- MI355X-native batched mj_step: C ABI of libmjhip.so
 *
 * Drop-in boundary for ONE path of MuJoCo: stepping many independent environments of one model.
 * The caller keeps owning its mjModel / mjData (created by MuJoCo itself); this library mirrors
 * the model constants and a batch of mjData states in GPU memory (SoA across environments) and
 * advances them with hand-written HIP kernels (one wavefront per environment).
 *
 * Every entry point cites the reference interface it replaces (paths relative to the MuJoCo
 * source tree).  Signatures use only plain pointers and sizes; `mjModel`/`mjData` are the
 * reference's own structs (include/mujoco/mjmodel.h, mjdata.h), passed by address and only read
 * through their public fields.
 *
 * Error handling: functions returning int return 0 on success and a negative code on failure;
 * functions returning a handle return NULL on failure.  mjhip_last_error() gives the message of
 * the last failure on the calling thread.  Numerical trouble inside a step is NOT an error: it is
 * reported exactly like the reference does, through per-environment warning counters
 * (mjData.warning[], include/mujoco/mjdata.h:74) readable as the batch field "warning".
 */
#ifndef MJHIP_H_
#define MJHIP_H_

#ifdef __cplusplus
extern "C" {
#endif

struct mjModel_;
struct mjData_;
typedef struct mjhipModel_ mjhipModel;   /* device copy of the model constants            */
typedef struct mjhipBatch_ mjhipBatch;   /* nenv device-resident mjData mirrors of a model */

#define MJHIP_API __attribute__((visibility("default")))

/* library / backend identification: "hip-gfx950" for the product, "hostsim" for the test-only
 * wavefront emulation built under tests/hostsim (never shipped). */
MJHIP_API const char* mjhip_backend(void);
MJHIP_API const char* mjhip_last_error(void);

/* number of visible GPUs (0 when the HIP runtime finds none: every compute entry point then fails
 * loudly -- there is no CPU fallback in the product). */
MJHIP_API int mjhip_device_count(void);

/* ---- model ------------------------------------------------------------------------------------
 * Replaces: nothing in the reference has a device model; this is the upload step that precedes the
 * calls below.  Reads the arrays of `m` listed in mujoco_amd/csrc/mjh_types.h.  Fails (NULL) with a
 * message naming the feature if the model uses something the GPU path does not implement.
 * nconmax / nefcmax: per-environment contact / constraint-row capacity (0 = choose from the
 * model); overflow raises mjWARN_CONTACTFULL / mjWARN_CNSTRFULL like a full arena does in the
 * reference (engine_collision_driver.c:2028, engine_core_constraint.c:145). */
MJHIP_API mjhipModel* mjhip_model_create(const struct mjModel_* m, int nconmax, int nefcmax);
MJHIP_API void mjhip_model_destroy(mjhipModel* model);
/* sizes: name is one of nq nv nu na nbody njnt ngeom nsite ntendon npair nconmax nefcmax nstate, or
 * "features" (bit set of optional features the model needs from a kernel variant) */
MJHIP_API int mjhip_model_size(const mjhipModel* model, const char* name);

/* Product-side reader of the reference's binary model format (mj_loadModel / mj_saveModel,
 * src/engine/engine_io.c:514-700) so that bench/smoke can obtain an mjModel without MuJoCo being
 * installed.  The returned struct is laid out exactly as MuJoCo's; free with mjhip_free_mjb. */
MJHIP_API struct mjModel_* mjhip_load_mjb(const char* path);
MJHIP_API void mjhip_free_mjb(struct mjModel_* m);
/* set one scalar of m->opt (mjOption, include/mujoco/mjmodel.h:83-126) by name, e.g. "solver",
 * "iterations", "tolerance", "timestep", "integrator", "cone", "disableflags"; what the reference's
 * bindings do with `model.opt.solver = ...`.  Must precede mjhip_model_create. */
MJHIP_API int mjhip_set_option(struct mjModel_* m, const char* name, double value);

/* ---- batch ------------------------------------------------------------------------------------
 * Replaces: mj_makeData x nenv (src/engine/engine_io.c) + one mjData per worker thread of
 * python/mujoco/rollout.cc.  All environments start at mj_resetData state (qpos0, zero velocity). */
MJHIP_API mjhipBatch* mjhip_batch_create(mjhipModel* model, int nenv, int device);
/* Explicit device layout of the batch (mjhip_batch_create uses $MJHIP_LAYOUT or the default):
 *   MJHIP_LAYOUT_AOS  every field [nenv][count]; one wavefront steps one environment through the
 *                     whole mj_step in a single kernel (LDS-resident working set);
 *   MJHIP_LAYOUT_SOA  every field [count][nenvpad] (SoA across environments, nenvpad = nenv rounded
 *                     up to 64); a step is three kernels: the constraint-free stages with one LANE
 *                     per environment (coalesced, scalar model constants), collision..PGS with one
 *                     wavefront per environment, then integration with one lane per environment.
 * Host-side get/set and the rollout arrays are [nenv][...] in both layouts. */
#define MJHIP_LAYOUT_AOS 0
#define MJHIP_LAYOUT_SOA 1
MJHIP_API mjhipBatch* mjhip_batch_create_layout(mjhipModel* model, int nenv, int device, int layout);
MJHIP_API void mjhip_batch_destroy(mjhipBatch* batch);
MJHIP_API int mjhip_batch_nenv(const mjhipBatch* batch);
MJHIP_API int mjhip_batch_reset(mjhipBatch* batch);   /* mj_resetData for every env */

/* Field access.  A field is one array [nenv][count] (see mjh_types.h, MJH_BATCH_*_FIELDS); names
 * follow mjData (qpos, qvel, ctrl, xpos, qLD, efc_J, ...) plus "counts" = {ncon,nefc,ne,nf,nl,
 * solver_niter,nisland,0} and "warning".  mjhip_batch_field returns the DEVICE pointer (zero-copy
 * interop with e.g. torch tensors); get/set copy a whole field from/to HOST memory. */
MJHIP_API int mjhip_batch_field(mjhipBatch* batch, const char* name, void** device_ptr,
                                int* count_per_env, int* is_int);
MJHIP_API int mjhip_batch_get(mjhipBatch* batch, const char* name, void* host_dst);
MJHIP_API int mjhip_batch_set(mjhipBatch* batch, const char* name, const void* host_src);

/* Kernel variant that steps the batch (no reference counterpart).  The stage sources are compiled in
 * several mappings (mujoco_amd/csrc/mjh_modes.h):
 *   "generic"  one wavefront per environment, every supported model feature;
 *   "lean"     the same mapping compiled for the lean feature set only (PGS, pyramidal cones, Euler,
 *              plane/sphere/capsule colliders, no sensors/equalities/...): no stack frames or register
 *              pressure from features the model does not use.
 * mjhip_batch_create picks the leanest variant that covers the model ("auto"; $MJHIP_VARIANT
 * overrides).  All variants produce bit-identical results.  set: 0 on success, <0 if the model needs
 * a feature the variant lacks. */
MJHIP_API int mjhip_batch_set_variant(mjhipBatch* batch, const char* name);
MJHIP_API const char* mjhip_batch_variant(const mjhipBatch* batch);
/* Name of the HIP kernel a rollout of this batch launches (the generic mapping exists for two register budgets; profiles
 * and bench.py's roofline name the kernel they measured). */
MJHIP_API const char* mjhip_batch_kernel(const mjhipBatch* batch);
/* Opt-in: build AR = Y Y' + diag(R) (mj_makeAR, src/engine/engine_core_constraint.c:3009 -- the one
 * dense contraction of the step) with v_mfma_f64_16x16x4_f64 instead of the reference-ordered
 * vector sums.  Results then agree with the reference to rounding (tolerance parity: states within
 * 1e-6, solver iteration counts within one), not bit for bit.  Default off ($MJHIP_MFMA=1 turns it
 * on for new batches); measured in profiles/r02_mfma. */
MJHIP_API int mjhip_batch_set_mfma(mjhipBatch* batch, int on);
/* Opt-in: the PGS sweep (mj_solPGS, src/engine/engine_solver.c:457-741) in residual-update form.  mode 0 (default): every
 * row visit takes a fresh mju_dot in the reference's association -- forces, states and iteration counts bit for bit.
 * mode 1: every constraint keeps its residual b + AR f and a row visit folds the one changed force into all of them with a
 * multiply-add -- the same sweep, projections, cost guard, momentum, restart and termination, but a residual's rounding
 * differs, so results agree with the reference to rounding (next states within 1e-6, contact / constraint counts exact,
 * solver_niter may differ) and not bit for bit.  About 2x less solver time on humanoid.xml.
 * $MJHIP_PGS=residual selects mode 1 for new batches.  Applies to solves of at most 64 rows with pyramidal cones. */
MJHIP_API int mjhip_batch_set_pgs_mode(mjhipBatch* batch, int mode);

/* LDS residency plan of the batch kernels (no reference counterpart: the reference keeps mjData in
 * host DRAM).  Each environment is stepped by one 64-lane wavefront that owns `lds_bytes` of LDS;
 * the plan decides which mjData fields live there while they are in use (see mjh_types.h).
 * mjhip_batch_create plans with $MJHIP_LDS_BYTES or a default; 0 disables residency (all fields
 * in HBM).  Returns the bytes left for the per-step constraint arrays, or <0 on error.
 * mjhip_batch_lds_report: printable description of the current plan. */
MJHIP_API int mjhip_batch_plan_lds(mjhipBatch* batch, int lds_bytes);
MJHIP_API const char* mjhip_batch_lds_report(const mjhipBatch* batch);

/* mj_forward restricted to a stage mask (bits MJH_STAGE_* of mjh_step.h; -1 = all).
 * Replaces mj_forward / mj_forwardSkip (src/engine/engine_forward.c:1783-1842) for every env.
 * By default every intermediate is materialised in its global field (inspection); OR-ing
 * MJHIP_STAGE_LDS into `stages` runs the stages on the LDS residency plan, exactly as the
 * step/rollout kernels do, and copies each field back to its global home after every stage. */
#define MJHIP_STAGE_ALL 0xdff
#define MJHIP_STAGE_EULER 0x200
#define MJHIP_STAGE_LDS (1 << 21)
MJHIP_API int mjhip_batch_forward(mjhipBatch* batch, int stages, void* hip_stream);

/* nstep x mj_step for every env with the device-resident ctrl / qfrc_applied
 * (closed-loop RL stepping).  Replaces mj_step (src/engine/engine_forward.c:1846). */
MJHIP_API int mjhip_batch_step(mjhipBatch* batch, int nstep, void* hip_stream);

/* mj_step1 / mj_step2 for every env (src/engine/engine_forward.c:1884, :1916): step1 = checkPos/Vel,
 * fwdPosition, fwdVelocity; the caller may then read any field (mjhip_batch_get / _field) and set
 * ctrl; step2 = fwdActuation, fwdAcceleration, fwdConstraint, sensors, checkAcc, integration (the
 * model's implicit integrator, else Euler -- RK4 is not available in the split, like the
 * reference).  step1 + step2 == mjhip_batch_step(batch, 1) for Euler / implicitfast models. */
MJHIP_API int mjhip_batch_step1(mjhipBatch* batch, void* hip_stream);
MJHIP_API int mjhip_batch_step2(mjhipBatch* batch, void* hip_stream);

/* Open-loop rollout of every env in the batch: the contract of _unsafe_rollout
 * (python/mujoco/rollout.cc:71-178) with one environment per rollout.
 *   state0      [nenv][nstate]            FULLPHYSICS initial states, or NULL (keep current)
 *   warmstart0  [nenv][nv]                or NULL (zeros)
 *   control     [nenv][nstep][ncontrol]   or NULL; ncontrol = mj_stateSize(control_spec)
 *   state       [nenv][nstep][nstate]     output, or NULL
 * control_spec: any mjSTATE_USER bits (mjtState, include/mujoco/mjtype.h:504-527): ctrl, qfrc_applied,
 * xfrc_applied, eq_active, mocap_pos, mocap_quat, userdata, laid out in bit order like mj_setState
 * (engine_support.c:282).  Inputs outside the spec are cleared / reset at the start of a rollout
 * (rollout.cc:85-115); inputs inside it keep their current values when control is NULL.
 * on_device: bit flags.  MJHIP_ROLLOUT_ON_DEVICE: the four pointers are DEVICE pointers (no PCIe
 * traffic in the call).  MJHIP_ROLLOUT_CONTINUE: keep the batch's current state, warm start and
 * warning counters instead of loading state0 / warmstart0 (chunked rollouts). */
#define MJHIP_ROLLOUT_ON_DEVICE 1
#define MJHIP_ROLLOUT_CONTINUE 2
MJHIP_API int mjhip_batch_rollout(mjhipBatch* batch, int nstep, unsigned control_spec,
                                  const double* state0, const double* warmstart0,
                                  const double* control, double* state, int on_device,
                                  void* hip_stream);
/* Same, plus the sensor readings of every step (the `sensordata` output of _unsafe_rollout,
 * python/mujoco/rollout.cc:77,:130-133):  sensordata [nenv][nstep][nsensordata] or NULL. */
MJHIP_API int mjhip_batch_rollout_sensors(mjhipBatch* batch, int nstep, unsigned control_spec,
                                          const double* state0, const double* warmstart0,
                                          const double* control, double* state, double* sensordata,
                                          int on_device, void* hip_stream);
MJHIP_API int mjhip_batch_sync(mjhipBatch* batch, void* hip_stream);
/* Sum, over environments [0, n) (n <= 0: all), of the warning counters that mean "this trajectory is
 * not what the reference would have computed": contact / constraint capacity overflow (the
 * reference's arena grows on demand) and geom pairs without a GPU collider (mjhip's warning slot 7).
 * Such environments are frozen like after any warning; this makes it visible.  0 on success. */
MJHIP_API int mjhip_batch_trouble(mjhipBatch* batch, int n, int* ncapacity, int* nunsupported);

/* ---- the drop-in ------------------------------------------------------------------------------
 * Same arguments and semantics as the reference's _unsafe_rollout / _unsafe_rollout_threaded /
 * Rollout::rollout (python/mujoco/rollout.cc:74, :181, :250): nbatch rollouts of nstep steps, HOST
 * pointers.  m[r] may differ per rollout as long as the sizes agree (rollout.cc:100-118;
 * rollout_test.py:363): rollouts are grouped by model (pointer, then content) and every group is
 * one device batch.  d[0] receives the last step of the LAST rollout (rollout.cc:73): time, qpos,
 * qvel, act, the user inputs, qacc_warmstart, qacc, sensordata, warning counters; with control ==
 * NULL the inputs named by control_spec are TAKEN from d[0] (the reference steps with whatever its
 * mjData holds).  The work is sharded over all visible GPUs ($MJHIP_DEVICES caps the count), one
 * host thread per GPU writing disjoint rows of the caller's arrays; device models and batches are
 * cached per GPU keyed by model content ($MJHIP_CACHE_MODELS entries, default 4), staging buffers
 * are pooled.  Returns 0 on success, <0 on failure, and -- with complete outputs -- 1 if some
 * environment overflowed the contact / constraint capacity ($MJHIP_NCONMAX / $MJHIP_NEFCMAX raise
 * it), 2 if one reached a geom pair that has no GPU collider: such environments were frozen and
 * back-filled where the reference would have kept simulating (message in mjhip_last_error). */
MJHIP_API int mjhip_rollout(const struct mjModel_* const* m, struct mjData_* const* d, int nbatch,
                            int nstep, unsigned control_spec, const double* state0,
                            const double* warmstart0, const double* control, double* state,
                            double* sensordata);

/* drop every cached device model / batch of mjhip_rollout (all GPUs) */
MJHIP_API void mjhip_rollout_clear_cache(void);

#ifdef __cplusplus
}
#endif
#endif  /* MJHIP_H_ */
