// Host-side construction of the device model tables from an mjModel (pure C++, no HIP).
//
// Direct copies of mjModel arrays + everything that depends only on the model and that the
// reference recomputes every step: tree levels / child lists for level-synchronous traversal,
// per-body dof-ancestor masks (mj_jac chains, engine_core_util.c:176), effective armature and
// damping incl. actuator contributions (mj_actuatorArmature/Damping, engine_core_util.c:1119-1220),
// and the static candidate geom-pair list in the reference's contact order with mixed contact
// parameters (mj_collision / mj_contactParam, engine_collision_driver.c:595-886,1740-1835).
//
// Anything the GPU path does not implement is rejected HERE, loudly, with the feature named.
#pragma once

#include <mujoco/mujoco.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <functional>
#include <vector>
#include <map>

#include "mjh_types.h"

struct HostModel {
  DSizes s;
  DOptions o;
#define X(name, cnt) std::vector<int> name;
  MJH_MODEL_INT_FIELDS(X)
#undef X
#define X(name, cnt) std::vector<real> name;
  MJH_MODEL_REAL_FIELDS(X)
#undef X
};

static_assert(sizeof(mjtNum) == sizeof(real), "mjhip is fp64 only");
static_assert(mjJNT_FREE == MJH_JNT_FREE && mjJNT_BALL == MJH_JNT_BALL &&
              mjJNT_SLIDE == MJH_JNT_SLIDE && mjJNT_HINGE == MJH_JNT_HINGE, "joint enum");
static_assert(mjGEOM_PLANE == MJH_GEOM_PLANE && mjGEOM_SPHERE == MJH_GEOM_SPHERE &&
              mjGEOM_CAPSULE == MJH_GEOM_CAPSULE && mjGEOM_BOX == MJH_GEOM_BOX, "geom enum");
static_assert(mjCNSTR_FRICTION_DOF == MJH_CNSTR_FRICTION_DOF &&
              mjCNSTR_LIMIT_JOINT == MJH_CNSTR_LIMIT_JOINT &&
              mjCNSTR_LIMIT_TENDON == MJH_CNSTR_LIMIT_TENDON &&
              mjCNSTR_CONTACT_FRICTIONLESS == MJH_CNSTR_CONTACT_FRICTIONLESS &&
              mjCNSTR_CONTACT_PYRAMIDAL == MJH_CNSTR_CONTACT_PYRAMIDAL &&
              mjCNSTR_CONTACT_ELLIPTIC == MJH_CNSTR_CONTACT_ELLIPTIC, "constraint enum");
static_assert(mjCNSTRSTATE_SATISFIED == MJH_STATE_SATISFIED && mjCNSTRSTATE_QUADRATIC == MJH_STATE_QUADRATIC &&
              mjCNSTRSTATE_LINEARNEG == MJH_STATE_LINEARNEG && mjCNSTRSTATE_LINEARPOS == MJH_STATE_LINEARPOS,
              "constraint state enum");
static_assert(mjWARN_CONTACTFULL == MJH_WARN_CONTACTFULL && mjWARN_CNSTRFULL == MJH_WARN_CNSTRFULL &&
              mjWARN_BADQPOS == MJH_WARN_BADQPOS && mjWARN_BADQVEL == MJH_WARN_BADQVEL &&
              mjWARN_BADQACC == MJH_WARN_BADQACC && mjWARN_BADCTRL == MJH_WARN_BADCTRL &&
              mjNWARNING <= 7, "warning enum");
static_assert(mjSAMEFRAME_BODY == MJH_SAMEFRAME_BODY && mjSAMEFRAME_INERTIA == MJH_SAMEFRAME_INERTIA &&
              mjSAMEFRAME_BODYROT == MJH_SAMEFRAME_BODYROT && mjSAMEFRAME_INERTIAROT == MJH_SAMEFRAME_INERTIAROT,
              "sameframe enum");
static_assert(mjDSBL_CONSTRAINT == 1<<0 && mjDSBL_FRICTIONLOSS == 1<<2 && mjDSBL_LIMIT == 1<<3 &&
              mjDSBL_CONTACT == 1<<4 && mjDSBL_SPRING == 1<<5 && mjDSBL_DAMPER == 1<<6 &&
              mjDSBL_GRAVITY == 1<<7 && mjDSBL_CLAMPCTRL == 1<<8 && mjDSBL_WARMSTART == 1<<9 &&
              mjDSBL_ACTUATION == 1<<11 && mjDSBL_REFSAFE == 1<<12 && mjDSBL_AUTORESET == 1<<16,
              "disable bits");
static_assert(mjNPOLY == 2 && mjNREF == 2 && mjNIMP == 5 && mjNGAIN == 10 && mjNBIAS == 10, "sizes");

namespace mjhb {

template <class T, class U>
static void copy_arr(std::vector<T>& dst, const U* src, size_t n) {
  dst.resize(n);
  for (size_t i = 0; i < n; i++) dst[i] = (T)src[i];
}

static inline bool filter_bitmask(int ct1, int ca1, int ct2, int ca2) {
  return !(ct1 & ca2) && !(ct2 & ca1);
}

// mj_actuatorDamping / mj_actuatorArmature for joints and tendons
static real actuator_contrib(const mjModel* m, int is_tendon, int id, int want_armature, real* poly) {
  int actuatorid = is_tendon ? m->tendon_actuatorid[id] : m->jnt_actuatorid[id];
  if (actuatorid == -1) return 0;
  real out = 0;
  auto add = [&](int k) {
    real g = m->actuator_gear[6*m->actuator_outadr[k]];
    real gear2 = g*g;
    if (want_armature) {
      out += m->actuator_armature[k] * gear2;
    } else {
      out += m->actuator_damping[k] * gear2;
      for (int j = 0; j < mjNPOLY; j++) poly[j] += m->actuator_dampingpoly[mjNPOLY*k + j] * gear2;
    }
  };
  if (actuatorid >= 0) {
    // single contributor: the reference assigns (not accumulates) damping/armature here
    real g = m->actuator_gear[6*m->actuator_outadr[actuatorid]];
    real gear2 = g*g;
    if (want_armature) return m->actuator_armature[actuatorid] * gear2;
    out = m->actuator_damping[actuatorid] * gear2;
    for (int j = 0; j < mjNPOLY; j++) poly[j] += m->actuator_dampingpoly[mjNPOLY*actuatorid + j] * gear2;
    return out;
  }
  for (int k = 0; k < m->nactuator; k++) {
    if (m->actuator_trnid[2*k] != id) continue;
    if (!is_tendon && m->actuator_trntype[k] != mjTRN_JOINT && m->actuator_trntype[k] != mjTRN_JOINTINPARENT) continue;
    if (is_tendon && m->actuator_trntype[k] != mjTRN_TENDON) continue;
    add(k);
  }
  return out;
}

// narrowphase routine of a type-ordered geom pair (mjCOLLISIONFUNC, engine_collision_driver.c:45-56) and the
// contact bound mj_maxContact gives it (:66-146); -1: the reference collides the types but mjhip does not
// (height fields, signed distance fields)
static int pair_func(int t1, int t2, bool has_margin, bool multiccd, int* maxcon) {
  if (t1 == mjGEOM_PLANE && t2 == mjGEOM_SPHERE) { *maxcon = 1; return MJH_COL_PLANE_SPHERE; }
  if (t1 == mjGEOM_PLANE && t2 == mjGEOM_CAPSULE) { *maxcon = 2; return MJH_COL_PLANE_CAPSULE; }
  if (t1 == mjGEOM_SPHERE && t2 == mjGEOM_SPHERE) { *maxcon = 1; return MJH_COL_SPHERE_SPHERE; }
  if (t1 == mjGEOM_SPHERE && t2 == mjGEOM_CAPSULE) { *maxcon = 1; return MJH_COL_SPHERE_CAPSULE; }
  if (t1 == mjGEOM_CAPSULE && t2 == mjGEOM_CAPSULE) { *maxcon = 2; return MJH_COL_CAPSULE_CAPSULE; }
  if (t1 == mjGEOM_PLANE && t2 == mjGEOM_CYLINDER) { *maxcon = 4; return MJH_COL_PLANE_CYLINDER; }
  if (t1 == mjGEOM_PLANE && t2 == mjGEOM_BOX) { *maxcon = 4; return MJH_COL_PLANE_BOX; }
  if (t1 == mjGEOM_SPHERE && t2 == mjGEOM_BOX) { *maxcon = 1; return MJH_COL_SPHERE_BOX; }
  if (t1 == mjGEOM_SPHERE && t2 == mjGEOM_CYLINDER) { *maxcon = 1; return MJH_COL_SPHERE_CYLINDER; }
  if (t1 == mjGEOM_BOX && t2 == mjGEOM_BOX) { *maxcon = 8; return MJH_COL_BOX_BOX; }
  if (t1 == mjGEOM_CAPSULE && t2 == mjGEOM_BOX) { *maxcon = 2; return MJH_COL_CAPSULE_BOX; }
  if (t1 == mjGEOM_PLANE && t2 == mjGEOM_ELLIPSOID) { *maxcon = 1; return MJH_COL_PLANE_CONVEX; }
  if (t1 == mjGEOM_PLANE && t2 == mjGEOM_MESH) { *maxcon = 3; return MJH_COL_PLANE_CONVEX; }
  // everything else between convex primitives and meshes goes to mjc_Convex (GJK / EPA)
  auto convex = [](int t) { return t == mjGEOM_SPHERE || t == mjGEOM_CAPSULE || t == mjGEOM_CYLINDER ||
                                   t == mjGEOM_ELLIPSOID || t == mjGEOM_BOX || t == mjGEOM_MESH; };
  if (convex(t1) && convex(t2)) {
    if (t1 == mjGEOM_SPHERE || t1 == mjGEOM_ELLIPSOID || t2 == mjGEOM_SPHERE || t2 == mjGEOM_ELLIPSOID) *maxcon = 1;
    else if (!multiccd) *maxcon = 1;
    else if (t1 == mjGEOM_CAPSULE || t2 == mjGEOM_CAPSULE || t1 == mjGEOM_CYLINDER || t2 == mjGEOM_CYLINDER) *maxcon = 5;
    else *maxcon = has_margin ? 5 : 4;
    return MJH_COL_CONVEX;
  }
  *maxcon = 0;
  return -1;
}

// does the reference have a collision function for (t1<=t2)?  (mjCOLLISIONFUNC, engine_collision_driver.c:45-56)
static bool ref_collides(int t1, int t2) {
  if (t1 == mjGEOM_PLANE) return t2 >= mjGEOM_SPHERE;   // plane-plane, plane-hfield: none
  if (t1 == mjGEOM_HFIELD) return t2 >= mjGEOM_SPHERE;
  return true;
}

struct BuildCaps { int nconmax = 0; int nefcmax = 0; long long efc_bytes = 0; };

#define MJH_REJECT(cond, msg) do { if (cond) { *err = std::string("mjhip: unsupported model feature: ") + (msg); return false; } } while (0)

static bool build(const mjModel* m, const BuildCaps& caps, HostModel* H, std::string* err) {
  DSizes& s = H->s;
  DOptions& o = H->o;
  memset(&s, 0, sizeof(s));
  memset(&o, 0, sizeof(o));

  // ---------------- feature gate ------------------------------------------------------------------
  MJH_REJECT(m->nv == 0, "model without degrees of freedom");
  for (int i = 0; i < m->neq; i++) {
    MJH_REJECT(m->eq_type[i] != mjEQ_CONNECT && m->eq_type[i] != mjEQ_WELD && m->eq_type[i] != mjEQ_JOINT &&
               m->eq_type[i] != mjEQ_TENDON && m->eq_type[i] != mjEQ_FLEX && m->eq_type[i] != mjEQ_FLEXVERT, "flex strain equality constraints");
    if (m->eq_type[i] == mjEQ_CONNECT || m->eq_type[i] == mjEQ_WELD)
      MJH_REJECT(m->eq_objtype[i] != mjOBJ_BODY && m->eq_objtype[i] != mjOBJ_SITE, "connect/weld between objects other than bodies or sites");
    const mjtNum* r = m->eq_solref + 2*i;
    MJH_REJECT((r[0] > 0) != (r[1] > 0), "mixed-sign solref on an equality constraint");
  }
  // flexes (mjh_flex.h): vertex-based deformables -- elasticity, bending, edge spring-dampers; interpolated (nodal)
  // flexes, flex equality constraints and penalty ("passive") flex contacts are not built
  for (int f = 0; f < m->nflex; f++) {
    if (m->flex_interp[f] != 0) {
      // interpolated flexes (nodes are bodies, vertices follow them): volume cells of order 1 or 2; every node either the
      // origin of a body with three dofs of its own (the force path of mj_flexPassiveInterp :185-199 that writes the body's
      // three dofs) or fixed to the world; explicit elasticity (Euler / RK4 -- the implicit integrators add mjd_flexInterp terms)
      MJH_REJECT(m->flex_interp[f] < 0 || m->flex_interp[f] > 2, "interpolated flexes in shell mode or of order above 2");
      MJH_REJECT(m->opt.integrator != mjINT_EULER && m->opt.integrator != mjINT_RK4, "interpolated flexes with an implicit integrator");
      MJH_REJECT(m->flex_selfcollide[f] != mjFLEXSELF_NONE && (m->flex_contype[f] & m->flex_conaffinity[f]), "self-collisions of an interpolated flex");
      for (int i = m->flex_nodeadr[f]; i < m->flex_nodeadr[f] + m->flex_nodenum[f]; i++) {
        const int b = m->flex_nodebodyid[i];
        const bool origin = m->flex_centered[f] || (m->flex_node[3*i] == 0 && m->flex_node[3*i+1] == 0 && m->flex_node[3*i+2] == 0);
        const bool fixed = m->body_dofnum[m->body_weldid[b]] == 0 && m->body_weldid[b] == 0;
        MJH_REJECT(!fixed && !(origin && m->body_dofnum[b] == 3 && m->body_simple[b] == 2),
                   "interpolated flex nodes other than bodies with three sliders of their own or fixed to the world");
      }
    }
    MJH_REJECT(m->flex_edgeequality[f] != 0 && m->flex_edgeequality[f] != 1 && m->flex_edgeequality[f] != 2, "flex strain equality constraints");
    MJH_REJECT(m->flex_edgeequality[f] == 2 && (m->flex_dim[f] != 2 || m->flex_interp[f] != 0 || m->flex_rigid[f]),
               "flex vertex equality constraints on a flex that is not a deformable shell");
    MJH_REJECT(m->flex_passive[f] != 0, "passive (penalty) flex contacts");
    MJH_REJECT(m->flex_dim[f] < 1 || m->flex_dim[f] > 3, "flex dimension outside 1..3");
  }
  MJH_REJECT(m->nplugin > 0, "plugins");
  for (int i = 0; i < m->nsensor; i++) {
    MJH_REJECT(m->sensor_history[2*i] != 0 || m->sensor_delay[i] != 0, "sensor history / delay");
  }
  MJH_REJECT(m->nsensor > 0 && m->nflex > 0, "sensors in models with flexes");
  MJH_REJECT(m->nhistory > 0, "history buffers / delays");
  MJH_REJECT(m->flg_adhesion && m->nflex > 0, "contact adhesion in models with flexes");
  MJH_REJECT(m->opt.integrator != mjINT_EULER && m->opt.integrator != mjINT_RK4 && m->opt.integrator != mjINT_IMPLICITFAST &&
             m->opt.integrator != mjINT_IMPLICIT, "unknown integrator");
  MJH_REJECT(m->opt.solver != mjSOL_PGS && m->opt.solver != mjSOL_NEWTON && m->opt.solver != mjSOL_CG, "unknown solver type");
  // mj_solNoSlip (engine_solver.c:764-958) runs on the reference's dense constraint path here (mju_dot residuals over
  // dense efc_AR rows); the sparse path's AR is a compressed matrix with its own summation order
  MJH_REJECT(m->opt.noslip_iterations > 0 && (m->opt.jacobian == mjJAC_SPARSE || (m->opt.jacobian == mjJAC_AUTO && m->nv >= 60) || m->nflex > 0),
             "noslip iterations outside the dense constraint path (sparse Jacobian, flexes)");
  // mj_isSparse (engine_core_util.c:29): with jacobian=sparse, or auto and nv >= 60, the reference
  // runs its sparse code paths.  They compute the same quantities with sums taken over the non-zeros
  // only; this path always evaluates the dense form, so such models agree with the reference to
  // rounding (not bit for bit) -- the parity tests hold them to the 1e-6 bar.
  // mjENBL_ENERGY and mjENBL_FWDINV only fill diagnostics (d->energy, d->solver_fwdinv) that are not
  // part of the rollout's outputs (energy sensors are rejected with the sensor list): accepted
  MJH_REJECT(m->opt.enableflags & (mjENBL_SLEEP | mjENBL_DIAGEXACT), "enable flags sleep / diagexact");
  {
    const bool fluid = m->opt.density != 0 || m->opt.viscosity != 0;
    MJH_REJECT(!fluid && (m->opt.wind[0] != 0 || m->opt.wind[1] != 0 || m->opt.wind[2] != 0), "wind without a fluid medium");

    // (round 6: the ellipsoid fluid model -- mj_ellipsoidFluidModel, engine_passive.c:1213-1270 -- per geom; tables below)
  }
  MJH_REJECT(m->nactuator != m->nu, "multi-input actuators (nactuator != nu)");
  MJH_REJECT(m->nout != m->nu, "multi-output actuators (nout != nu)");
  for (int i = 0; i < m->nactuator; i++) {
    MJH_REJECT(m->actuator_ctrlnum[i] != 1 || m->actuator_ctrladr[i] != i, "actuator control blocks other than one scalar");
    MJH_REJECT(m->actuator_outnum[i] != 1 || m->actuator_outadr[i] != i, "actuator output blocks other than one scalar");
    {
      const int dt = m->actuator_dyntype[i];
      MJH_REJECT(dt != mjDYN_NONE && dt != mjDYN_INTEGRATOR && dt != mjDYN_FILTER && dt != mjDYN_FILTEREXACT && dt != mjDYN_MUSCLE,
                 "actuator dynamics other than none/integrator/filter/filterexact/muscle (dcmotor, pid, user)");
      MJH_REJECT(dt == mjDYN_NONE ? m->actuator_actnum[i] != 0 : m->actuator_actnum[i] != 1,
                 "actuators with more than one activation variable");
    }
    MJH_REJECT(m->actuator_gaintype[i] != mjGAIN_FIXED && m->actuator_gaintype[i] != mjGAIN_AFFINE && m->actuator_gaintype[i] != mjGAIN_MUSCLE,
               "actuator gain types other than fixed/affine/muscle");
    MJH_REJECT(m->actuator_biastype[i] != mjBIAS_NONE && m->actuator_biastype[i] != mjBIAS_AFFINE && m->actuator_biastype[i] != mjBIAS_MUSCLE,
               "actuator bias types other than none/affine/muscle");
    MJH_REJECT(m->actuator_plugin[i] >= 0, "actuator plugins");
    MJH_REJECT(m->actuator_delay[i] != 0, "actuator delays");
    int tt = m->actuator_trntype[i];
    MJH_REJECT(tt != mjTRN_JOINT && tt != mjTRN_JOINTINPARENT && tt != mjTRN_SLIDERCRANK && tt != mjTRN_TENDON && tt != mjTRN_SITE && tt != mjTRN_BODY,
               "actuator transmissions other than joint / slider-crank / tendon / site / body (SO3)");
    if (tt == mjTRN_TENDON) continue;      // (tendon-level armature / force limits are checked with the tendons)
    if (tt == mjTRN_BODY) {
      // adhesion actuators: the moment averages the normal rows of the body's contacts (mj_mulJacTVec over efc_J: dense or
      // compressed, every moment entry sums the rows in order)
      MJH_REJECT(m->nflex > 0, "body transmissions (adhesion actuators) in models with flexes");
      MJH_REJECT(m->actuator_armature[i] != 0 || m->actuator_damping[i] != 0 ||
                 m->actuator_dampingpoly[mjNPOLY*i] != 0 || m->actuator_dampingpoly[mjNPOLY*i + 1] != 0,
                 "actuator-level armature/damping on a body transmission");
      continue;
    }
    if (tt == mjTRN_SITE) {
      if (m->actuator_trnid[2*i + 1] != -1) o.has_refsite = 1;      // (round 6: relative pose against a reference site)
      MJH_REJECT(m->actuator_armature[i] != 0 || m->actuator_damping[i] != 0 ||
                 m->actuator_dampingpoly[mjNPOLY*i] != 0 || m->actuator_dampingpoly[mjNPOLY*i + 1] != 0,
                 "actuator-level armature/damping on a site transmission");
      continue;
    }
    if (tt == mjTRN_SLIDERCRANK) {
      MJH_REJECT(m->actuator_armature[i] != 0 || m->actuator_damping[i] != 0 ||
                 m->actuator_dampingpoly[mjNPOLY*i] != 0 || m->actuator_dampingpoly[mjNPOLY*i + 1] != 0,
                 "actuator-level armature/damping on a slider-crank transmission");
      continue;
    }
    int jt = m->jnt_type[m->actuator_trnid[2*i]];
    if (jt == mjJNT_BALL || jt == mjJNT_FREE) {
      // 3D / 6D gear (engine_core_smooth.c:1331-1392); position servos on ball joints wrap their
      // set point (wrapPeriod, engine_forward.c:297-328), which is not implemented
      MJH_REJECT(jt == mjJNT_BALL && m->actuator_gaintype[i] == mjGAIN_FIXED && m->actuator_biastype[i] == mjBIAS_AFFINE &&
                 m->actuator_gainprm[mjNGAIN*i] == -m->actuator_biasprm[mjNBIAS*i + 1], "position servos on ball joints");
      MJH_REJECT(m->actuator_armature[i] != 0 || m->actuator_damping[i] != 0 ||
                 m->actuator_dampingpoly[mjNPOLY*i] != 0 || m->actuator_dampingpoly[mjNPOLY*i + 1] != 0,
                 "actuator-level armature/damping on a ball or free joint");
    }
    // servo wrap period (wrapPeriod, engine_forward.c:305-342) is zero for hinge/slide joint transmissions
  }
  for (int i = 0; i < m->ntendon; i++) {
    // (tendons wrapping around spheres / cylinders: round 6, mjh_math.h mjh_wrap + stage_tendon)
    for (int w = m->tendon_adr[i]; w < m->tendon_adr[i] + m->tendon_num[i]; w++)
      if (m->wrap_type[w] == mjWRAP_SPHERE || m->wrap_type[w] == mjWRAP_CYLINDER) o.has_tendon_wrap = 1;
    if (m->tendon_frictionloss[i] > 0) {
      const mjtNum* r = m->tendon_solref_fri + 2*i;
      MJH_REJECT((r[0] > 0) != (r[1] > 0), "mixed-sign solref on tendon friction");
    }
    // (tendon armature, round 6: mj_tendonArmature / mj_tendonBias -- stage_crb, stage_rne; the fully implicit integrator,
    //  whose derivative of the bias term would be needed, is rejected as a whole)
    MJH_REJECT((m->tendon_armature[i] != 0 || actuator_contrib(m, 1, i, 1, nullptr) != 0) && m->nv > 128 &&
               (m->opt.jacobian == mjJAC_SPARSE || m->opt.jacobian == mjJAC_AUTO),
               "tendon armature in a model with more than 128 degrees of freedom and a sparse Jacobian");
  }
  for (int i = 0; i < m->njnt; i++) {
    if (m->jnt_limited[i]) {
      const mjtNum* r = m->jnt_solref + mjNREF*i;
      MJH_REJECT((r[0] > 0) != (r[1] > 0), "mixed-sign solref on a joint limit");
    }
  }
  for (int i = 0; i < m->nv; i++) {
    if (m->dof_frictionloss[i] != 0) {
      const mjtNum* r = m->dof_solref + mjNREF*i;
      MJH_REJECT((r[0] > 0) != (r[1] > 0), "mixed-sign solref on dof friction");
    }
  }

  // ---------------- sizes & options -----------------------------------------------------------------
  s.nq = m->nq; s.nv = m->nv; s.nu = m->nu; s.na = m->na; s.nbody = m->nbody; s.njnt = m->njnt;
  s.neq = m->neq;
  s.ngeom = m->ngeom; s.nsite = m->nsite; s.ntendon = m->ntendon; s.nwrap = m->nwrap;
  s.nC = m->nC; s.nJten = m->nJten; s.ntree = m->ntree;
  s.nvw = (m->nv + 31)/32;
  s.nstate = 1 + m->nq + m->nv + m->na;
  s.nmoment = m->nu;   // (recomputed below once the transmission types are known)

  o.timestep = m->opt.timestep; o.impratio = m->opt.impratio; o.tolerance = m->opt.tolerance; o.ls_tolerance = m->opt.ls_tolerance; o.ls_iterations = m->opt.ls_iterations;
  for (int k = 0; k < 3; k++) o.gravity[k] = m->opt.gravity[k];
  o.meaninertia = m->stat.meaninertia;
  o.integrator = m->opt.integrator; o.cone = m->opt.cone; o.solver = m->opt.solver;
  o.iterations = m->opt.iterations;
  o.noslip_iterations = m->opt.noslip_iterations > 0 ? m->opt.noslip_iterations : 0; o.noslip_tolerance = m->opt.noslip_tolerance;
  o.disableflags = m->opt.disableflags; o.enableflags = m->opt.enableflags;

  // ---------------- direct copies ----------------------------------------------------------------------
  copy_arr(H->body_parentid, m->body_parentid, m->nbody);
  copy_arr(H->body_rootid, m->body_rootid, m->nbody);
  copy_arr(H->body_weldid, m->body_weldid, m->nbody);
  copy_arr(H->body_simple, m->body_simple, m->nbody);
  copy_arr(H->body_mocapid, m->body_mocapid, m->nbody);
  copy_arr(H->body_jntnum, m->body_jntnum, m->nbody);
  copy_arr(H->body_jntadr, m->body_jntadr, m->nbody);
  copy_arr(H->body_dofnum, m->body_dofnum, m->nbody);
  copy_arr(H->body_dofadr, m->body_dofadr, m->nbody);
  copy_arr(H->body_geomnum, m->body_geomnum, m->nbody);
  copy_arr(H->body_geomadr, m->body_geomadr, m->nbody);
  copy_arr(H->body_sameframe, m->body_sameframe, m->nbody);
  copy_arr(H->jnt_type, m->jnt_type, m->njnt);
  copy_arr(H->jnt_qposadr, m->jnt_qposadr, m->njnt);
  copy_arr(H->jnt_dofadr, m->jnt_dofadr, m->njnt);
  copy_arr(H->jnt_bodyid, m->jnt_bodyid, m->njnt);
  copy_arr(H->jnt_limited, m->jnt_limited, m->njnt);
  // standalone free bodies (mj_isFreeBody, engine_derivative.c:822-838): implicitfast solves their
  // 6x6 block separately
  H->jnt_freebody.assign(m->njnt, 0);
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j];
    if (m->jnt_type[j] != mjJNT_FREE || m->body_jntnum[b] != 1) continue;
    int adr = m->jnt_dofadr[j];
    if (m->tree_dofnum[m->dof_treeid[adr]] != 6 || m->body_subtreemass[b] != m->body_mass[b]) continue;
    H->jnt_freebody[j] = 1;
  }
  // the fully implicit integrator (round 6; mj_implicitSkip, engine_forward.c:1649-1770): the patterns of qDeriv and of
  // mjd_rne_vel's body-by-dof work arrays
  s.nD = 0; s.nB = 0;
  H->D_rowadr.clear(); H->D_rownnz.clear(); H->D_diag.clear(); H->D_colind.clear(); H->D_rowid.clear(); H->D_mapM.clear();
  H->B_rowadr.clear(); H->B_rownnz.clear(); H->B_ncopy.clear(); H->B_pmap.clear();
  if (m->opt.integrator == mjINT_IMPLICIT && m->nv > 0) {
    s.nD = m->nD; s.nB = m->nB;
    copy_arr(H->D_rowadr, m->D_rowadr, m->nv);
    copy_arr(H->D_rownnz, m->D_rownnz, m->nv);
    copy_arr(H->D_diag, m->D_diag, m->nv);
    copy_arr(H->D_colind, m->D_colind, m->nD);
    copy_arr(H->D_mapM, m->mapM2D, m->nD);
    H->D_rowid.assign(m->nD, 0);
    for (int i = 0; i < m->nv; i++) {
      for (int k = 0; k < m->D_rownnz[i]; k++) H->D_rowid[m->D_rowadr[i] + k] = i;
      // (mjd_rne_vel: "Dcacc[row i] and Dcdofdot[row j] have identical sparsity", and the last pass subtracts a B row from a D row)
      MJH_REJECT(m->D_rownnz[i] != m->B_rownnz[m->dof_bodyid[i]], "internal: qDeriv row and body row of different lengths");
      MJH_REJECT(m->D_colind[m->D_rowadr[i] + m->D_diag[i]] != i, "internal: qDeriv diagonal index");
    }
    copy_arr(H->B_rowadr, m->B_rowadr, m->nbody);
    copy_arr(H->B_rownnz, m->B_rownnz, m->nbody);
    H->B_ncopy.assign(m->nbody, 0);
    H->B_pmap.assign(m->nB, -1);
    for (int n = 1; n < m->nbody; n++) {
      const int np = m->body_parentid[n];
      if (m->body_weldid[np] == 0) continue;
      int ndof = 0;
      for (int a = m->body_weldid[np]; a > 0; a = m->body_weldid[m->body_parentid[a]]) ndof += m->body_dofnum[a];
      H->B_ncopy[n] = ndof;
      MJH_REJECT(ndof > m->B_rownnz[n] || ndof > m->B_rownnz[np], "internal: body row shorter than its chain");
      int ip = 0;
      for (int i = 0; i < m->B_rownnz[n]; i++) {
        const int c = m->B_colind[m->B_rowadr[n] + i];
        while (ip < m->B_rownnz[np] && m->B_colind[m->B_rowadr[np] + ip] < c) ip++;
        MJH_REJECT(ip >= m->B_rownnz[np] || m->B_colind[m->B_rowadr[np] + ip] != c, "internal: body row not a subset of its parent's");
        H->B_pmap[m->B_rowadr[n] + i] = ip;
      }
    }
  }
  H->M_rowid.assign(m->nC, 0);
  for (int i = 0; i < m->nv; i++)
    for (int k = 0; k < m->M_rownnz[i]; k++) H->M_rowid[m->M_rowadr[i] + k] = i;
  copy_arr(H->jnt_actfrclimited, m->jnt_actfrclimited, m->njnt);
  copy_arr(H->dof_bodyid, m->dof_bodyid, m->nv);
  copy_arr(H->dof_jntid, m->dof_jntid, m->nv);
  copy_arr(H->dof_parentid, m->dof_parentid, m->nv);
  copy_arr(H->dof_simplenum, m->dof_simplenum, m->nv);
  copy_arr(H->M_rownnz, m->M_rownnz, m->nv);
  copy_arr(H->M_rowadr, m->M_rowadr, m->nv);
  copy_arr(H->M_colind, m->M_colind, m->nC);
  {
    // column lists of M's strict lower triangle (see M_cscadr in mjh_types.h)
    std::vector<std::vector<int>> col(m->nv);
    for (int i = 0; i < m->nv; i++)
      for (int k = 0; k < m->M_rownnz[i] - 1; k++) col[m->M_colind[m->M_rowadr[i] + k]].push_back(m->M_rowadr[i] + k);
    H->M_cscadr.assign((size_t)m->nv + 1, 0);
    H->M_cscind.assign((size_t)m->nC, 0);
    int fill = 0;
    for (int t = 0; t < m->nv; t++) {
      H->M_cscadr[t] = fill;
      for (int a : col[t]) H->M_cscind[fill++] = a;
    }
    H->M_cscadr[m->nv] = fill;
  }
  copy_arr(H->geom_type, m->geom_type, m->ngeom);
  copy_arr(H->geom_bodyid, m->geom_bodyid, m->ngeom);
  copy_arr(H->geom_sameframe, m->geom_sameframe, m->ngeom);
  copy_arr(H->site_bodyid, m->site_bodyid, m->nsite);
  // equality constraints: rows per equality (connect 3, weld 6, joint/tendon 1)
  copy_arr(H->eq_type, m->eq_type, m->neq);
  copy_arr(H->eq_obj1id, m->eq_obj1id, m->neq);
  copy_arr(H->eq_obj2id, m->eq_obj2id, m->neq);
  H->eq_objsite.resize(m->neq);
  H->eq_active0.resize(m->neq);
  H->eq_rowadr.assign((size_t)m->neq + 1, 0);
  for (int i = 0; i < m->neq; i++) {
    H->eq_objsite[i] = (m->eq_objtype[i] == mjOBJ_SITE) ? 1 : 0;
    H->eq_active0[i] = m->eq_active0[i] ? 1 : 0;
    int size = m->eq_type[i] == mjEQ_CONNECT ? 3 : (m->eq_type[i] == mjEQ_WELD ? 6 : 1);
    if (m->eq_type[i] == mjEQ_FLEX) {
      // one row per non-rigid edge of the flex (mj_instantiateEquality, engine_core_constraint.c:982-1010)
      const int f = m->eq_obj1id[i];
      size = 0;
      for (int ed = m->flex_edgeadr[f]; ed < m->flex_edgeadr[f] + m->flex_edgenum[f]; ed++) if (!m->flexedge_rigid[ed]) size++;
    }
    if (m->eq_type[i] == mjEQ_FLEXVERT) {
      // two rows per vertex of the flex (mj_instantiateEquality :1013-1038); a row without entries would be dropped by
      // mj_addConstraint's empty guard while mj_diagApprox still counts it: not reproduced
      const int f = m->eq_obj1id[i];
      MJH_REJECT(m->flex_edgeequality[f] != 2, "a flex vertex equality constraint on a flex without vertex constraints");
      size = 2*m->flex_vertnum[f];
      for (int v = m->flex_vertadr[f]; v < m->flex_vertadr[f] + m->flex_vertnum[f]; v++)
        MJH_REJECT(m->flexvert_J_rownnz[2*v] == 0, "flex vertex equality constraints with a vertex whose row is empty");
    }
    if (m->eq_type[i] == mjEQ_CONNECT || m->eq_type[i] == mjEQ_WELD) {
      // both bodies static: the Jacobian block is identically zero and mj_addConstraint drops the
      // whole constraint (empty-block guard, engine_core_constraint.c:424-447)
      int b1 = m->eq_obj1id[i], b2 = m->eq_obj2id[i];
      if (m->eq_objtype[i] == mjOBJ_SITE) { b1 = m->site_bodyid[b1]; b2 = m->site_bodyid[b2]; }
      if (m->body_treeid[b1] < 0 && m->body_treeid[b2] < 0) size = 0;
    }
    H->eq_rowadr[i + 1] = H->eq_rowadr[i] + size;
  }
  {
    bool anyflex = false;
    for (int i = 0; i < m->neq; i++) if (m->eq_type[i] == mjEQ_FLEX || m->eq_type[i] == mjEQ_FLEXVERT) anyflex = true;
    s.neqrow = anyflex ? H->eq_rowadr[m->neq] : 0;
    H->eqrow_edge.assign((size_t)s.neqrow, -1);
    for (int i = 0; i < m->neq && anyflex; i++) {
      if (m->eq_type[i] == mjEQ_FLEXVERT) {
        // (vertex constraints: the flexvert row 2 v + j of every row)
        const int f = m->eq_obj1id[i];
        int k = H->eq_rowadr[i];
        for (int v = m->flex_vertadr[f]; v < m->flex_vertadr[f] + m->flex_vertnum[f]; v++) { H->eqrow_edge[k++] = 2*v; H->eqrow_edge[k++] = 2*v + 1; }
        continue;
      }
      if (m->eq_type[i] != mjEQ_FLEX) continue;
      const int f = m->eq_obj1id[i];
      int k = H->eq_rowadr[i];
      for (int ed = m->flex_edgeadr[f]; ed < m->flex_edgeadr[f] + m->flex_edgenum[f]; ed++) if (!m->flexedge_rigid[ed]) H->eqrow_edge[k++] = ed;
    }
  }
  copy_arr(H->eq_solref, m->eq_solref, 2*m->neq);
  copy_arr(H->eq_solimp, m->eq_solimp, 5*m->neq);
  copy_arr(H->eq_data, m->eq_data, mjNEQDATA*m->neq);
  copy_arr(H->tendon_length0, m->tendon_length0, m->ntendon);
  copy_arr(H->site_sameframe, m->site_sameframe, m->nsite);
  copy_arr(H->tendon_adr, m->tendon_adr, m->ntendon);
  copy_arr(H->tendon_num, m->tendon_num, m->ntendon);
  copy_arr(H->tendon_limited, m->tendon_limited, m->ntendon);
  copy_arr(H->ten_J_rownnz, m->ten_J_rownnz, m->ntendon);
  copy_arr(H->ten_J_rowadr, m->ten_J_rowadr, m->ntendon);
  copy_arr(H->ten_J_colind, m->ten_J_colind, m->nJten);
  copy_arr(H->wrap_type, m->wrap_type, m->nwrap);
  copy_arr(H->wrap_objid, m->wrap_objid, m->nwrap);
  copy_arr(H->actuator_trntype, m->actuator_trntype, m->nu);
  copy_arr(H->actuator_trnid, m->actuator_trnid, 2*m->nu);
  copy_arr(H->actuator_gaintype, m->actuator_gaintype, m->nu);
  copy_arr(H->actuator_biastype, m->actuator_biastype, m->nu);
  copy_arr(H->actuator_ctrllimited, m->actuator_ctrllimited, m->nu);
  copy_arr(H->actuator_forcelimited, m->actuator_forcelimited, m->nu);
  copy_arr(H->actuator_dyntype, m->actuator_dyntype, m->nu);
  copy_arr(H->actuator_actadr, m->actuator_actadr, m->nu);
  copy_arr(H->actuator_actlimited, m->actuator_actlimited, m->nu);
  copy_arr(H->actuator_actearly, m->actuator_actearly, m->nu);
  // mj_actuatorDisabled (engine_support.c:695): groups 0..30 named in opt.disableactuator
  H->actuator_disabled.assign(m->nu, 0);
  for (int i = 0; i < m->nu; i++) {
    const int g = m->actuator_group[i];
    if (g >= 0 && g <= 30 && (m->opt.disableactuator & (1 << g))) { H->actuator_disabled[i] = 1; o.has_act_disabled = 1; }
  }
  copy_arr(H->tendon_actfrclimited, m->tendon_actfrclimited, m->ntendon);
  copy_arr(H->tendon_actfrcrange, m->tendon_actfrcrange, 2*m->ntendon);
  for (int i = 0; i < m->ntendon; i++) if (m->tendon_actfrclimited[i]) o.has_ten_actfrc = 1;
  copy_arr(H->actuator_actrange, m->actuator_actrange, 2*m->nu);
  H->actuator_dyntau.resize(m->nu);
  for (int i = 0; i < m->nu; i++) H->actuator_dyntau[i] = m->actuator_dynprm[mjNDYN*i];
  H->actuator_dynprm.resize(3*m->nu);
  for (int i = 0; i < m->nu; i++) for (int k = 0; k < 3; k++) H->actuator_dynprm[3*i + k] = m->actuator_dynprm[mjNDYN*i + k];
  copy_arr(H->actuator_lengthrange, m->actuator_lengthrange, 2*m->nu);
  copy_arr(H->actuator_acc0, m->actuator_acc0, m->nu);

  copy_arr(H->qpos0, m->qpos0, m->nq);
  copy_arr(H->qpos_spring, m->qpos_spring, m->nq);
  copy_arr(H->body_pos, m->body_pos, 3*m->nbody);
  copy_arr(H->body_quat, m->body_quat, 4*m->nbody);
  copy_arr(H->body_ipos, m->body_ipos, 3*m->nbody);
  copy_arr(H->body_iquat, m->body_iquat, 4*m->nbody);
  copy_arr(H->body_mass, m->body_mass, m->nbody);
  copy_arr(H->body_gravcomp, m->body_gravcomp, m->nbody);
  // ellipsoid fluid model: interaction coefficients and semi-axes of every geom, the bodies that use it
  s.ngeom_fluid = 0;
  H->geom_fluid.clear(); H->geom_semiaxes.clear(); H->body_ellipsoid.clear();
  if (m->opt.density != 0 || m->opt.viscosity != 0)
    for (int g = 0; g < m->ngeom; g++) if (m->geom_fluid[mjNFLUID*g] > 0) s.ngeom_fluid = m->ngeom;
  if (s.ngeom_fluid) {
    static_assert(mjNFLUID == 12, "geom_fluid layout");
    copy_arr(H->geom_fluid, m->geom_fluid, (size_t)mjNFLUID*m->ngeom);
    H->geom_semiaxes.assign(3*(size_t)m->ngeom, 0);
    for (int g = 0; g < m->ngeom; g++) {
      const mjtNum* sz = m->geom_size + 3*g;
      real* ax = &H->geom_semiaxes[3*g];
      // (mju_geomSemiAxes, engine_util_misc.c:423: sphere r r r; capsule r r h + r; cylinder r r h; else the sizes)
      const int t = m->geom_type[g];
      ax[0] = sz[0];
      ax[1] = (t == mjGEOM_SPHERE || t == mjGEOM_CAPSULE || t == mjGEOM_CYLINDER) ? sz[0] : sz[1];
      ax[2] = t == mjGEOM_SPHERE ? sz[0] : (t == mjGEOM_CAPSULE ? sz[1] + sz[0] : (t == mjGEOM_CYLINDER ? sz[1] : sz[2]));
    }
    H->body_ellipsoid.assign(m->nbody, 0);
    for (int b = 0; b < m->nbody; b++)
      for (int g = m->body_geomadr[b]; g < m->body_geomadr[b] + m->body_geomnum[b]; g++)
        if (m->geom_fluid[mjNFLUID*g] > 0) H->body_ellipsoid[b] = 1;
  }
  o.has_gravcomp = m->flg_gravcomp ? 1 : 0;
  // actuator-level gravity compensation (round 6; engine_forward.c:981-996, engine_passive.c:1112-1122): the compensation
  // force of such a joint's dofs joins qfrc_actuator (before the joint-level force limits) instead of qfrc_passive
  s.nv_actgc = 0;
  H->dof_actgravcomp.clear();
  for (int j = 0; j < m->njnt; j++) if (m->jnt_actgravcomp[j] && m->flg_gravcomp) s.nv_actgc = m->nv;
  if (s.nv_actgc) {
    H->dof_actgravcomp.assign(m->nv, 0);
    for (int i = 0; i < m->nv; i++) H->dof_actgravcomp[i] = m->jnt_actgravcomp[m->dof_jntid[i]] ? 1 : 0;
  }
  o.has_surfacevel = m->flg_surfacevel ? 1 : 0;
  o.has_adhesion = m->flg_adhesion ? 1 : 0;
  copy_arr(H->geom_surfacevel, m->geom_surfacevel, 6*m->ngeom);
  o.has_fluid = (m->opt.density != 0 || m->opt.viscosity != 0) ? 1 : 0;
  o.density = m->opt.density; o.viscosity = m->opt.viscosity;
  for (int k = 0; k < 3; k++) o.wind[k] = m->opt.wind[k];
  s.nbody_fluid = o.has_fluid ? m->nbody : 0;
  copy_arr(H->body_subtreemass, m->body_subtreemass, m->nbody);
  copy_arr(H->body_inertia, m->body_inertia, 3*m->nbody);
  copy_arr(H->body_invweight0, m->body_invweight0, 2*m->nbody);
  copy_arr(H->jnt_pos, m->jnt_pos, 3*m->njnt);
  copy_arr(H->jnt_axis, m->jnt_axis, 3*m->njnt);
  copy_arr(H->jnt_stiffness, m->jnt_stiffness, m->njnt);
  copy_arr(H->jnt_stiffnesspoly, m->jnt_stiffnesspoly, 2*m->njnt);
  copy_arr(H->jnt_range, m->jnt_range, 2*m->njnt);
  copy_arr(H->jnt_margin, m->jnt_margin, m->njnt);
  copy_arr(H->jnt_solref, m->jnt_solref, 2*m->njnt);
  copy_arr(H->jnt_solimp, m->jnt_solimp, 5*m->njnt);
  copy_arr(H->jnt_actfrcrange, m->jnt_actfrcrange, 2*m->njnt);
  copy_arr(H->dof_invweight0, m->dof_invweight0, m->nv);
  copy_arr(H->dof_M0, m->dof_M0, m->nv);
  copy_arr(H->dof_frictionloss, m->dof_frictionloss, m->nv);
  copy_arr(H->dof_solref, m->dof_solref, 2*m->nv);
  copy_arr(H->dof_solimp, m->dof_solimp, 5*m->nv);
  copy_arr(H->geom_pos, m->geom_pos, 3*m->ngeom);
  copy_arr(H->geom_quat, m->geom_quat, 4*m->ngeom);
  copy_arr(H->geom_size, m->geom_size, 3*m->ngeom);
  copy_arr(H->geom_rbound, m->geom_rbound, m->ngeom);
  copy_arr(H->site_pos, m->site_pos, 3*m->nsite);
  copy_arr(H->site_quat, m->site_quat, 4*m->nsite);
  copy_arr(H->site_size, m->site_size, 3*m->nsite);
  copy_arr(H->site_type, m->site_type, m->nsite);
  copy_arr(H->tendon_range, m->tendon_range, 2*m->ntendon);
  copy_arr(H->tendon_margin, m->tendon_margin, m->ntendon);
  copy_arr(H->tendon_solref_lim, m->tendon_solref_lim, 2*m->ntendon);
  copy_arr(H->tendon_solimp_lim, m->tendon_solimp_lim, 5*m->ntendon);
  copy_arr(H->tendon_solref_fri, m->tendon_solref_fri, 2*m->ntendon);
  copy_arr(H->tendon_solimp_fri, m->tendon_solimp_fri, 5*m->ntendon);
  copy_arr(H->tendon_invweight0, m->tendon_invweight0, m->ntendon);
  copy_arr(H->tendon_stiffness, m->tendon_stiffness, m->ntendon);
  copy_arr(H->tendon_stiffnesspoly, m->tendon_stiffnesspoly, 2*m->ntendon);
  copy_arr(H->tendon_lengthspring, m->tendon_lengthspring, 2*m->ntendon);
  copy_arr(H->tendon_frictionloss, m->tendon_frictionloss, m->ntendon);
  copy_arr(H->wrap_prm, m->wrap_prm, m->nwrap);
  copy_arr(H->actuator_gear, m->actuator_gear, 6*m->nu);
  copy_arr(H->actuator_ctrlrange, m->actuator_ctrlrange, 2*m->nu);
  copy_arr(H->actuator_forcerange, m->actuator_forcerange, 2*m->nu);
  copy_arr(H->actuator_gainprm, m->actuator_gainprm, 10*m->nu);
  copy_arr(H->actuator_biasprm, m->actuator_biasprm, 10*m->nu);
  copy_arr(H->actuator_cranklength, m->actuator_cranklength, m->nu);

  // ---------------- derived: dofs ------------------------------------------------------------------------
  H->dof_jnttype.resize(m->nv);
  for (int i = 0; i < m->nv; i++) H->dof_jnttype[i] = m->jnt_type[m->dof_jntid[i]];
  H->dof_armature_eff.resize(m->nv);
  H->dof_damping_eff.resize(m->nv);
  H->dof_dampingpoly_eff.assign(2*m->nv, 0);
  int euler_damp = 0;
  for (int i = 0; i < m->nv; i++) {
    int j = m->dof_jntid[i];
    H->dof_armature_eff[i] = m->dof_armature[i] + actuator_contrib(m, 0, j, 1, nullptr);
    real poly[2] = {m->dof_dampingpoly[2*i], m->dof_dampingpoly[2*i+1]};
    H->dof_damping_eff[i] = m->dof_damping[i] + actuator_contrib(m, 0, j, 0, poly);
    H->dof_dampingpoly_eff[2*i] = poly[0];
    H->dof_dampingpoly_eff[2*i+1] = poly[1];
    // gate of the implicit-damping branch (engine_forward.c:1409-1420)
    if (m->dof_damping[i] > 0 || m->dof_dampingpoly[2*i] != 0 || m->dof_dampingpoly[2*i+1] != 0 ||
        m->jnt_actuatorid[j] != -1) euler_damp = 1;
  }
  if ((m->opt.disableflags & mjDSBL_EULERDAMP) || (m->opt.disableflags & mjDSBL_DAMPER)) euler_damp = 0;
  o.euler_damp = euler_damp;
  H->tendon_damping_eff.resize(m->ntendon);
  H->tendon_dampingpoly_eff.assign(2*m->ntendon, 0);
  H->tendon_armature_eff.assign(m->ntendon, 0);
  o.has_ten_armature = 0;
  for (int i = 0; i < m->ntendon; i++) {
    // mj_tendonArmature / mj_tendonBias: tendon_armature + mj_actuatorArmature(mjOBJ_TENDON) (engine_core_smooth.c:1860, :2617)
    H->tendon_armature_eff[i] = m->tendon_armature[i] + actuator_contrib(m, 1, i, 1, nullptr);
    if (H->tendon_armature_eff[i] != 0) o.has_ten_armature = 1;
    real poly[2] = {m->tendon_dampingpoly[2*i], m->tendon_dampingpoly[2*i+1]};
    H->tendon_damping_eff[i] = m->tendon_damping[i] + actuator_contrib(m, 1, i, 0, poly);
    H->tendon_dampingpoly_eff[2*i] = poly[0];
    H->tendon_dampingpoly_eff[2*i+1] = poly[1];
  }
  // sparse actuator_moment: static row capacity 1 for joint transmissions, nv for slider-cranks
  H->actuator_momentadr.resize(m->nu + 1);
  H->actuator_momentadr[0] = 0;
  for (int i = 0; i < m->nu; i++)
    H->actuator_momentadr[i + 1] = H->actuator_momentadr[i] + ((m->actuator_trntype[i] == mjTRN_SLIDERCRANK || m->actuator_trntype[i] == mjTRN_SITE || m->actuator_trntype[i] == mjTRN_BODY) ? m->nv :
        ((m->actuator_trntype[i] == mjTRN_JOINT || m->actuator_trntype[i] == mjTRN_JOINTINPARENT) ?
          (m->jnt_type[m->actuator_trnid[2*i]] == mjJNT_BALL ? 3 : (m->jnt_type[m->actuator_trnid[2*i]] == mjJNT_FREE ? 6 : 1)) :
        (m->actuator_trntype[i] == mjTRN_TENDON ? std::max(1, m->ten_J_rownnz[m->actuator_trnid[2*i]]) : 1)));
  s.nmoment = H->actuator_momentadr[m->nu];
  {
    H->dof_act_adr.assign(m->nv + 1, 0);
    H->dof_act_ids.assign(m->nu + 1, 0);
    int fill = 0;
    for (int j = 0; j < m->nv; j++) {
      H->dof_act_adr[j] = fill;
      for (int i = 0; i < m->nu; i++) {
        const int tt = m->actuator_trntype[i];
        if (tt != mjTRN_JOINT && tt != mjTRN_JOINTINPARENT) continue;
        const int id = m->actuator_trnid[2*i];
        if (m->jnt_type[id] != mjJNT_HINGE && m->jnt_type[id] != mjJNT_SLIDE) continue;
        if (m->jnt_dofadr[id] == j) H->dof_act_ids[fill++] = i;
      }
    }
    H->dof_act_adr[m->nv] = fill;
  }

  // ---------------- derived: tree levels, children, dof ancestors --------------------------------------
  std::vector<int> depth(m->nbody, 0);
  int nlevel = 1;
  for (int i = 1; i < m->nbody; i++) {
    depth[i] = depth[m->body_parentid[i]] + 1;
    nlevel = std::max(nlevel, depth[i] + 1);
  }
  s.nlevel = nlevel;
  H->body_level_adr.assign(nlevel + 1, 0);
  for (int i = 0; i < m->nbody; i++) H->body_level_adr[depth[i] + 1]++;
  for (int L = 0; L < nlevel; L++) H->body_level_adr[L+1] += H->body_level_adr[L];
  H->body_level_ids.assign(m->nbody, 0);
  {
    std::vector<int> fill(H->body_level_adr.begin(), H->body_level_adr.end() - 1);
    for (int i = 0; i < m->nbody; i++) H->body_level_ids[fill[depth[i]]++] = i;
  }
  H->body_child_adr.assign(m->nbody + 1, 0);
  for (int i = 1; i < m->nbody; i++) H->body_child_adr[m->body_parentid[i] + 1]++;
  for (int i = 0; i < m->nbody; i++) H->body_child_adr[i+1] += H->body_child_adr[i];
  H->body_child_ids.assign(m->nbody, 0);
  {
    // children in DECREASING body id: the order in which `for b = nbody-1..1` visits them
    std::vector<int> fill(H->body_child_adr.begin(), H->body_child_adr.end() - 1);
    for (int i = m->nbody - 1; i >= 1; i--) H->body_child_ids[fill[m->body_parentid[i]]++] = i;
  }
  H->body_dofanc.assign((size_t)m->nbody * s.nvw, 0);
  for (int b = 0; b < m->nbody; b++) {
    int w = m->body_weldid[b];
    if (m->body_dofnum[w] == 0) continue;
    int i = m->body_dofadr[w] + m->body_dofnum[w] - 1;
    while (i >= 0) {
      H->body_dofanc[(size_t)b*s.nvw + (i >> 5)] |= (1u << (i & 31));
      i = m->dof_parentid[i];
    }
  }

  // ---------------- static candidate pairs in reference contact order --------------------------------
  const int dsbl_filterparent = m->opt.disableflags & mjDSBL_FILTERPARENT;
  const int midphase = !(m->opt.disableflags & mjDSBL_MIDPHASE);
  const int override_ = m->opt.enableflags & mjENBL_OVERRIDE;
  struct GP { int g1, g2, ipair; int b1 = -1, b2 = -1, route = 0; };
  std::vector<int> pair_b1, pair_b2;     // body pair of every emitted geom pair (-1: predefined)
  int maxcon_total = 0;
  // predefined <pair>s are merged into the list in signature order, ahead of their body pair's own
  // geom pairs (mj_collision, engine_collision_driver.c:651-664); a geom pair that duplicates a
  // predefined pair of the same body pair is dropped (filterCollisionPair :550-557)
  int pairadr = 0;
  auto emit_pairs = [&](const std::vector<GP>& gps) -> bool {
      for (const GP& gp : gps) {
        int g1 = gp.g1, g2 = gp.g2;
        int maxcon = 0;
        int func = -2;
        // mj_contactParam (engine_collision_driver.c:1740-1835)
        int condim;
        real solref[2], solimp[5], fri[3];
        int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
        const mjtNum *sr1 = m->geom_solref + 2*g1, *sr2 = m->geom_solref + 2*g2;
        const mjtNum *si1 = m->geom_solimp + 5*g1, *si2 = m->geom_solimp + 5*g2;
        const mjtNum *f1 = m->geom_friction + 3*g1, *f2 = m->geom_friction + 3*g2;
        if (p1 > p2) {
          condim = m->geom_condim[g1];
          for (int k = 0; k < 2; k++) solref[k] = sr1[k];
          for (int k = 0; k < 5; k++) solimp[k] = si1[k];
          for (int k = 0; k < 3; k++) fri[k] = f1[k];
        } else if (p1 < p2) {
          condim = m->geom_condim[g2];
          for (int k = 0; k < 2; k++) solref[k] = sr2[k];
          for (int k = 0; k < 5; k++) solimp[k] = si2[k];
          for (int k = 0; k < 3; k++) fri[k] = f2[k];
        } else {
          condim = std::max(m->geom_condim[g1], m->geom_condim[g2]);
          real m1 = m->geom_solmix[g1], m2 = m->geom_solmix[g2], mix;
          if (m1 >= mjMINVAL && m2 >= mjMINVAL) mix = m1 / (m1 + m2);
          else if (m1 < mjMINVAL && m2 < mjMINVAL) mix = 0.5;
          else if (m1 < mjMINVAL) mix = 0.0;
          else mix = 1.0;
          if (sr1[0] > 0 && sr2[0] > 0) {
            for (int k = 0; k < 2; k++) solref[k] = mix*sr1[k] + (1-mix)*sr2[k];
          } else {
            for (int k = 0; k < 2; k++) solref[k] = std::min(sr1[k], sr2[k]);
          }
          for (int k = 0; k < 5; k++) solimp[k] = mix*si1[k] + (1-mix)*si2[k];
          for (int k = 0; k < 3; k++) fri[k] = std::max(f1[k], f2[k]);
        }
        real friction[5] = {fri[0], fri[0], fri[1], fri[2], fri[2]};
        // adhesion: each surface contributes its own attraction; the side with the higher priority alone (:1763-1779)
        real adhesion = p1 > p2 ? m->geom_adhesion[g1] : p1 < p2 ? m->geom_adhesion[g2] : m->geom_adhesion[g1] + m->geom_adhesion[g2];
        real margin = m->geom_margin[g1] + m->geom_margin[g2];
        real gap = m->geom_gap[g1] + m->geom_gap[g2];
        real solreffriction[2] = {0, 0};
        if (gp.ipair >= 0) {
          // predefined pair: its own parameters (engine_collision_driver.c:2048-2057, getMargin/getGap :161-175)
          const int ip = gp.ipair;
          condim = m->pair_dim[ip];
          for (int k = 0; k < 2; k++) solref[k] = m->pair_solref[2*ip + k];
          for (int k = 0; k < 5; k++) solimp[k] = m->pair_solimp[5*ip + k];
          for (int k = 0; k < 5; k++) friction[k] = m->pair_friction[5*ip + k];
          for (int k = 0; k < 2; k++) solreffriction[k] = m->pair_solreffriction[2*ip + k];
          margin = m->pair_margin[ip];
          gap = m->pair_gap[ip];
          adhesion = m->pair_adhesion[ip];
          MJH_REJECT((solreffriction[0] > 0) != (solreffriction[1] > 0), "mixed-sign pair solreffriction");
        }
        if (override_) {
          // mj_assignMargin/Ref/Imp/Friction (engine_core_constraint.c:177-218)
          margin = m->opt.o_margin;
          for (int k = 0; k < 2; k++) solref[k] = m->opt.o_solref[k];
          for (int k = 0; k < 5; k++) solimp[k] = m->opt.o_solimp[k];
          for (int k = 0; k < 5; k++) friction[k] = m->opt.o_friction[k];
        }
        for (int k = 0; k < 5; k++) friction[k] = std::max((real)mjMINMU, friction[k]);
        MJH_REJECT((solref[0] > 0) != (solref[1] > 0), "mixed-sign contact solref");
        func = pair_func(m->geom_type[g1], m->geom_type[g2], margin + gap > 0, !(m->opt.disableflags & mjDSBL_MULTICCD), &maxcon);
        MJH_REJECT(func < 0, "collision with height-field / signed-distance-field geoms");
        if (func == MJH_COL_CONVEX || func == MJH_COL_PLANE_CONVEX) {
          MJH_REJECT(m->opt.disableflags & mjDSBL_NATIVECCD, "the libccd convex collision pipeline (mjDSBL_NATIVECCD; libccd is a third-party library)");
          for (int g : {g1, g2})
            if (m->geom_type[g] == mjGEOM_MESH)
              MJH_REJECT(m->geom_dataid[g] < 0 || m->mesh_vertnum[m->geom_dataid[g]] < 1, "mesh geom without vertices");
          s.ccd_any = 1;
        }
        H->pair_geom1.push_back(g1);
        H->pair_geom2.push_back(g2);
        H->pair_dim.push_back(condim);
        H->pair_maxcon.push_back(maxcon);
        H->pair_func.push_back(func);
        H->pair_route.push_back(gp.route);
        pair_b1.push_back(gp.b1);
        pair_b2.push_back(gp.b2);
        H->pair_margin.push_back(margin + gap);
        H->pair_includemargin.push_back(margin);
        H->pair_adhesion.push_back(adhesion);
        for (int k = 0; k < 5; k++) H->pair_friction.push_back(friction[k]);
        for (int k = 0; k < 2; k++) H->pair_solref.push_back(solref[k]);
        for (int k = 0; k < 2; k++) H->pair_solreffriction.push_back(solreffriction[k]);
        for (int k = 0; k < 5; k++) H->pair_solimp.push_back(solimp[k]);
        maxcon_total += maxcon;
      }
      return true;
  };
  struct FlexJob { int pair_end, body, flex; };
  std::vector<FlexJob> flexjobs;
  for (int b1 = 0; b1 < m->nbody; b1++) {
    const bool b1_on = m->body_contype[b1] || m->body_conaffinity[b1];
    if (b1 > 0 && m->nflex) {
      // body : flex pairs of the previous body follow its body : body pairs (signature order, bodyflex id of a flex = nbody + f;
      // mj_collision, engine_collision_driver.c:637-700).  The sweep-and-prune cull of such a pair compares margin-inflated
      // bounds and is conservative: every pair is taken to the per-vertex / per-element tests.
      const int b = b1 - 1;
      if ((m->body_contype[b] || m->body_conaffinity[b]) && m->body_geomnum[b] > 0)
        for (int f = 0; f < m->nflex; f++)
          if ((m->flex_contype[f] || m->flex_conaffinity[f]) &&
              !filter_bitmask(m->body_contype[b], m->body_conaffinity[b], m->flex_contype[f], m->flex_conaffinity[f]))
            flexjobs.push_back({(int)H->pair_geom1.size(), b, f});
    }
    for (int b2 = b1 + 1; b2 < m->nbody; b2++) {
      const unsigned sig_here = ((unsigned)b1 << 16) + (unsigned)b2;
      const int startadr = pairadr;
      {
        std::vector<GP> ex;
        for (; pairadr < m->npair && (unsigned)m->pair_signature[pairadr] <= sig_here; pairadr++) {
          int a = m->pair_geom1[pairadr], b = m->pair_geom2[pairadr];
          if (m->geom_type[a] > m->geom_type[b]) std::swap(a, b);
          ex.push_back({a, b, pairadr});
        }
        if (!emit_pairs(ex)) return false;
      }
      if (!b1_on || !(m->body_contype[b2] || m->body_conaffinity[b2])) continue;
      int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
      int pw1 = m->body_weldid[m->body_parentid[w1]], pw2 = m->body_weldid[m->body_parentid[w2]];
      // filterBodyPair (engine_collision_driver.c:288-319), nothing asleep
      if (w1 == w2) continue;
      if (m->body_dofnum[w1] == 0 && m->body_dofnum[w2] == 0) continue;
      if (!dsbl_filterparent && w1 != 0 && w2 != 0 && (w1 == pw2 || w2 == pw1)) continue;
      if (filter_bitmask(m->body_contype[b1], m->body_conaffinity[b1],
                         m->body_contype[b2], m->body_conaffinity[b2])) continue;
      unsigned sig = ((unsigned)b1 << 16) + (unsigned)b2;
      bool excluded = false;
      for (int x = 0; x < m->nexclude; x++) if ((unsigned)m->exclude_signature[x] == sig) excluded = true;
      if (excluded) continue;

      std::vector<GP> gps;
      for (int g1 = m->body_geomadr[b1]; g1 < m->body_geomadr[b1] + m->body_geomnum[b1]; g1++) {
        for (int g2 = m->body_geomadr[b2]; g2 < m->body_geomadr[b2] + m->body_geomnum[b2]; g2++) {
          if (filter_bitmask(m->geom_contype[g1], m->geom_conaffinity[g1],
                             m->geom_contype[g2], m->geom_conaffinity[g2])) continue;
          int a = g1, b = g2;
          if (m->geom_type[a] > m->geom_type[b]) std::swap(a, b);   // pushGeomGeom
          if (!ref_collides(m->geom_type[a], m->geom_type[b])) continue;
          bool dup = false;
          for (int k = startadr; k < pairadr; k++)
            if ((unsigned)m->pair_signature[k] == sig_here &&
                ((m->pair_geom1[k] == g1 && m->pair_geom2[k] == g2) || (m->pair_geom1[k] == g2 && m->pair_geom2[k] == g1))) dup = true;
          if (dup) continue;
          gps.push_back({a, b, -1, b1, b2, 1});
        }
      }
      // midphase route: contacts are sorted by the STORED (type-ordered) geom ids (contactcompare :410-440)
      bool single = m->body_geomnum[b1] == 1 && m->body_geomnum[b2] == 1;
      if (!single && midphase && m->body_bvhadr[b1] >= 0 && m->body_bvhadr[b2] >= 0) {
        for (GP& x : gps) x.route = 2;
        std::stable_sort(gps.begin(), gps.end(), [](const GP& x, const GP& y) {
          return x.g1 != y.g1 ? x.g1 < y.g1 : x.g2 < y.g2; });
      }
      if (!emit_pairs(gps)) return false;
    }
  }
  {
    // predefined pairs beyond the last body pair
    std::vector<GP> ex;
    for (; pairadr < m->npair; pairadr++) {
      int a = m->pair_geom1[pairadr], b = m->pair_geom2[pairadr];
      if (m->geom_type[a] > m->geom_type[b]) std::swap(a, b);
      ex.push_back({a, b, pairadr});
    }
    if (!emit_pairs(ex)) return false;
  }
  if (m->nflex) {
    const int b = m->nbody - 1;
    if ((m->body_contype[b] || m->body_conaffinity[b]) && m->body_geomnum[b] > 0)
      for (int f = 0; f < m->nflex; f++)
        if ((m->flex_contype[f] || m->flex_conaffinity[f]) &&
            !filter_bitmask(m->body_contype[b], m->body_conaffinity[b], m->flex_contype[f], m->flex_conaffinity[f]))
          flexjobs.push_back({(int)H->pair_geom1.size(), b, f});
  }
  s.npair = (int)H->pair_geom1.size();

  // ---------------- body : flex collision jobs (mjh_flexcol.h) -------------------------------------------------
  // The geoms of a job's body get parameter records appended to the pair tables (index npair + k): mj_contactParam with
  // the flex on the second side (engine_collision_driver.c:1740-1835).
  {
    H->colseg.clear(); H->flexjob_adr.clear(); H->flexjob_geom.clear(); H->flexjob_nsub.clear();
    H->flexjob_leaf.clear(); H->jobbvh_child.clear(); H->jobbvh_parent.clear(); H->jobbvh_surface.clear();
    bool need_bvh = false;
    for (const FlexJob& j : flexjobs) {
      const int f = j.flex, b = j.body;
      // (a body with several geoms: its hierarchy's nodes, for the walk order -- see flex_collide_job)
      // (needed only when at least TWO geoms of the body can emit (geom, element) pairs: for one geom the pairs come in the
      //  flex hierarchy's own depth-first order whatever the body's hierarchy splits in between -- the children of a flex
      //  node are pushed in a fixed order, :1196-1240 --, and planes never reach the walk's colliders (:1128).  A world body
      //  that holds nothing but planes -- jelly.xml's walls -- must not make every step recompute the flex's hierarchy:
      //  that cost the jelly benchmark 3 % in round 5, profiles/r06/flex_bisect.txt)
      int nwalk = 0;
      for (int g = m->body_geomadr[b]; g < m->body_geomadr[b] + m->body_geomnum[b]; g++)
        if (m->geom_type[g] != mjGEOM_PLANE && m->geom_type[g] != mjGEOM_SDF &&
            !filter_bitmask(m->geom_contype[g], m->geom_conaffinity[g], m->flex_contype[f], m->flex_conaffinity[f])) nwalk++;
      int treebase = -1;
      if (midphase && m->body_bvhadr[b] >= 0 && m->body_bvhnum[b] > 1 && nwalk > 1) {
        treebase = (int)H->jobbvh_parent.size();
        const int adr = m->body_bvhadr[b], num = m->body_bvhnum[b];
        H->jobbvh_parent.resize(treebase + num, -1);
        for (int i = 0; i < num; i++) {
          const int c0 = m->bvh_child[2*(adr + i)], c1 = m->bvh_child[2*(adr + i) + 1];
          MJH_REJECT((c0 < 0) != (c1 < 0), "internal: body bounding volume hierarchy node with one child");
          H->jobbvh_child.push_back(c0 < 0 ? -1 : treebase + c0);
          H->jobbvh_child.push_back(c1 < 0 ? -1 : treebase + c1);
          if (c0 >= 0) { H->jobbvh_parent[treebase + c0] = treebase + i; H->jobbvh_parent[treebase + c1] = treebase + i; }
          // (the "surface" mj_collideTree compares, :1207-1215: from entries 3..5 minus entries 0..2 of the node's box)
          const mjtNum* bx = m->bvh_aabb + 6*(adr + i);
          const mjtNum x1 = bx[3] - bx[0], y1 = bx[4] - bx[1], z1 = bx[5] - bx[2];
          H->jobbvh_surface.push_back(x1*y1 + y1*z1 + z1*x1);
        }
        need_bvh = true;
      }
      MJH_REJECT(midphase == 0, "flex collisions with the midphase disabled");
      MJH_REJECT(m->flex_bvhadr[f] < 0 || m->body_bvhadr[b] < 0, "flex collisions without bounding volume hierarchies");
      MJH_REJECT(m->flex_rigid[f], "collisions of rigid flexes");
      const bool dofless = m->body_dofnum[m->body_weldid[b]] == 0;
      H->colseg.push_back(j.pair_end); H->colseg.push_back(b); H->colseg.push_back(f);
      H->flexjob_adr.push_back((int)H->flexjob_geom.size());
      for (int g = m->body_geomadr[b]; g < m->body_geomadr[b] + m->body_geomnum[b]; g++) {
        const int type = m->geom_type[g];
        int nsub = 0;
        MJH_REJECT(type == mjGEOM_SDF || type == mjGEOM_HFIELD, "flex collisions with height-field / signed-distance-field geoms");
        if (type == mjGEOM_PLANE) {
          if (!dofless) continue;                 // (planes on moving bodies never reach a flex collider, mj_collideTree :1030-1038, :1128)
        } else {
          if (filter_bitmask(m->geom_contype[g], m->geom_conaffinity[g], m->flex_contype[f], m->flex_conaffinity[f])) continue;
          // mj_collideGeomElem (engine_collision_driver.c:2372): triangles against spheres / capsules / boxes have closed
          // forms (mjraw_SphereTriangle / CapsuleTriangle / BoxTriangle), every other pairing goes through GJK / EPA with the
          // element as a convex object (mjc_ConvexElem).  Not built: line elements against spheres / capsules / boxes (the raw
          // capsule colliders on a capsule made of the two vertices)
          const int dim = m->flex_dim[f];
          MJH_REJECT(dim == 1 && type == mjGEOM_BOX, "collisions of line flexes with boxes");
          int sub = 0;                            // closed form: candidate contacts per (geom, triangle)
          if (dim == 2 && type == mjGEOM_SPHERE) sub = 1;
          else if (dim == 2 && type == mjGEOM_CAPSULE) sub = 5;
          else if (dim == 2 && type == mjGEOM_BOX) sub = 11;
          else if (dim == 1 && type == mjGEOM_SPHERE) sub = 1;      // (mjraw_SphereCapsule / CapsuleCapsule on the element's capsule)
          else if (dim == 1 && type == mjGEOM_CAPSULE) sub = 2;
          nsub = sub;
          if (!sub) {
            MJH_REJECT(m->opt.disableflags & mjDSBL_NATIVECCD, "the libccd convex collision pipeline (mjDSBL_NATIVECCD; libccd is a third-party library)");
            if (type == mjGEOM_MESH)
              MJH_REJECT(m->geom_dataid[g] < 0 || m->mesh_vertnum[m->geom_dataid[g]] < 1, "mesh geom without vertices");
            s.ccd_any = 1;
          }
        }
        // mj_contactParam(g, -1, -1, f)
        int condim; real solref[2], solimp[5], fri[3];
        const int p1 = m->geom_priority[g], p2 = m->flex_priority[f];
        const mjtNum *sr1 = m->geom_solref + 2*g, *sr2 = m->flex_solref + 2*f;
        const mjtNum *si1 = m->geom_solimp + 5*g, *si2 = m->flex_solimp + 5*f;
        const mjtNum *f1 = m->geom_friction + 3*g, *f2 = m->flex_friction + 3*f;
        if (p1 > p2) {
          condim = m->geom_condim[g];
          for (int k = 0; k < 2; k++) solref[k] = sr1[k];
          for (int k = 0; k < 5; k++) solimp[k] = si1[k];
          for (int k = 0; k < 3; k++) fri[k] = f1[k];
        } else if (p1 < p2) {
          condim = m->flex_condim[f];
          for (int k = 0; k < 2; k++) solref[k] = sr2[k];
          for (int k = 0; k < 5; k++) solimp[k] = si2[k];
          for (int k = 0; k < 3; k++) fri[k] = f2[k];
        } else {
          condim = std::max(m->geom_condim[g], m->flex_condim[f]);
          real m1 = m->geom_solmix[g], m2 = m->flex_solmix[f], mix;
          if (m1 >= mjMINVAL && m2 >= mjMINVAL) mix = m1 / (m1 + m2);
          else if (m1 < mjMINVAL && m2 < mjMINVAL) mix = 0.5;
          else if (m1 < mjMINVAL) mix = 0.0;
          else mix = 1.0;
          if (sr1[0] > 0 && sr2[0] > 0) { for (int k = 0; k < 2; k++) solref[k] = mix*sr1[k] + (1-mix)*sr2[k]; }
          else { for (int k = 0; k < 2; k++) solref[k] = std::min(sr1[k], sr2[k]); }
          for (int k = 0; k < 5; k++) solimp[k] = mix*si1[k] + (1-mix)*si2[k];
          for (int k = 0; k < 3; k++) fri[k] = std::max(f1[k], f2[k]);
        }
        real friction[5] = {fri[0], fri[0], fri[1], fri[2], fri[2]};
        real margin = m->geom_margin[g] + m->flex_margin[f];
        const real gap = m->geom_gap[g] + m->flex_gap[f];
        if (override_) {
          margin = m->opt.o_margin;
          for (int k = 0; k < 2; k++) solref[k] = m->opt.o_solref[k];
          for (int k = 0; k < 5; k++) solimp[k] = m->opt.o_solimp[k];
          for (int k = 0; k < 5; k++) friction[k] = m->opt.o_friction[k];
        }
        for (int k = 0; k < 5; k++) friction[k] = std::max((real)mjMINMU, friction[k]);
        MJH_REJECT((solref[0] > 0) != (solref[1] > 0), "mixed-sign contact solref");
        H->flexjob_geom.push_back(g);
        H->flexjob_nsub.push_back(nsub);
        {
          int leaf = -1;
          if (treebase >= 0)
            for (int i = 0; i < m->body_bvhnum[b]; i++)
              if (m->bvh_child[2*(m->body_bvhadr[b] + i)] < 0 && m->bvh_nodeid[m->body_bvhadr[b] + i] == g) leaf = treebase + i;
          MJH_REJECT(treebase >= 0 && leaf < 0, "internal: geom without a leaf in its body's bounding volume hierarchy");
          H->flexjob_leaf.push_back(leaf);
        }
        H->pair_geom1.push_back(g);
        H->pair_geom2.push_back(-1);
        H->pair_dim.push_back(condim);
        H->pair_margin.push_back(margin + gap);
        H->pair_includemargin.push_back(margin);
        H->pair_adhesion.push_back(0);
        for (int k = 0; k < 5; k++) H->pair_friction.push_back(friction[k]);
        for (int k = 0; k < 2; k++) H->pair_solref.push_back(solref[k]);
        for (int k = 0; k < 2; k++) H->pair_solreffriction.push_back(0);
        for (int k = 0; k < 5; k++) H->pair_solimp.push_back(solimp[k]);
      }
    }
    H->colseg.push_back(s.npair); H->colseg.push_back(-1); H->colseg.push_back(-1);
    H->flexjob_adr.push_back((int)H->flexjob_geom.size());
    H->flexjob_adr.push_back((int)H->flexjob_geom.size());
    s.ncolseg = (int)H->colseg.size()/3;
    // internal collisions: not built.  Self-collisions of line / shell
    // flexes (mj_collision, engine_collision_driver.c:834-881): by sweep-and-prune over the element boxes
    // (mj_collideFlexSAP :2315) or over all pairs of active elements; the walk of the flex's bounding volume hierarchy
    // against itself (solid flexes under selfcollide = auto, or selfcollide = bvh) is not built.
    // Pairs of different flexes (mj_collision :795-813 without midphase, mj_collideTree :1166-1181 with it): every pair of
    // elements whose boxes overlap goes to mj_collideElems.  With midphase the walk over the two bounding volume hierarchies
    // meets the pairs in depth-first order (the thinning of more than mjMAXCONPAIR contacts -- filterFlexContacts -- works on
    // array positions, so the order matters); the contacts are sorted by (element, element) afterwards.  The position of a
    // leaf pair in that order is a function of its own path from the roots (flex_pair_collide, mjh_flexcol.h); mode 1.
    // Without midphase (mode 2) the double loop emits in (element, element) order and nothing is sorted.  Contacts of these
    // pairs follow all geom contacts.
    H->flexff_flex.clear(); H->flexff_pair.clear(); H->flexff_mode.clear();
    for (int f1 = 0; f1 < m->nflex; f1++)
      for (int f2 = f1 + 1; f2 < m->nflex; f2++) {
        if (!(m->flex_contype[f1] || m->flex_conaffinity[f1]) || !(m->flex_contype[f2] || m->flex_conaffinity[f2])) continue;
        if (filter_bitmask(m->flex_contype[f1], m->flex_conaffinity[f1], m->flex_contype[f2], m->flex_conaffinity[f2])) continue;
        MJH_REJECT(m->flex_rigid[f1] || m->flex_rigid[f2], "collisions of rigid flexes");
        MJH_REJECT((m->flex_dim[f1] == 1) != (m->flex_dim[f2] == 1), "collisions between a line flex and a shell / solid flex");
        MJH_REJECT(m->flex_gap[f1] + m->flex_gap[f2] != 0, "collisions between two flexes with a contact gap");
        MJH_REJECT(m->flex_dim[f1] != 1 && (m->opt.disableflags & mjDSBL_NATIVECCD), "the libccd convex collision pipeline (mjDSBL_NATIVECCD; libccd is a third-party library)");
        MJH_REJECT(m->flex_elemnum[f1] >= 0x10000 || m->flex_elemnum[f2] >= 0x10000, "flex : flex collisions with 65536 or more elements");
        if (m->flex_dim[f1] != 1) s.ccd_any = 1;
        // mj_contactParam(-1, -1, f1, f2)
        int condim; real solref[2], solimp[5], fri[3];
        const int p1 = m->flex_priority[f1], p2 = m->flex_priority[f2];
        const mjtNum *sr1 = m->flex_solref + 2*f1, *sr2 = m->flex_solref + 2*f2;
        const mjtNum *si1 = m->flex_solimp + 5*f1, *si2 = m->flex_solimp + 5*f2;
        const mjtNum *fr1 = m->flex_friction + 3*f1, *fr2 = m->flex_friction + 3*f2;
        if (p1 > p2) {
          condim = m->flex_condim[f1];
          for (int k = 0; k < 2; k++) solref[k] = sr1[k];
          for (int k = 0; k < 5; k++) solimp[k] = si1[k];
          for (int k = 0; k < 3; k++) fri[k] = fr1[k];
        } else if (p1 < p2) {
          condim = m->flex_condim[f2];
          for (int k = 0; k < 2; k++) solref[k] = sr2[k];
          for (int k = 0; k < 5; k++) solimp[k] = si2[k];
          for (int k = 0; k < 3; k++) fri[k] = fr2[k];
        } else {
          condim = std::max(m->flex_condim[f1], m->flex_condim[f2]);
          real m1 = m->flex_solmix[f1], m2 = m->flex_solmix[f2], mix;
          if (m1 >= mjMINVAL && m2 >= mjMINVAL) mix = m1 / (m1 + m2);
          else if (m1 < mjMINVAL && m2 < mjMINVAL) mix = 0.5;
          else if (m1 < mjMINVAL) mix = 0.0;
          else mix = 1.0;
          if (sr1[0] > 0 && sr2[0] > 0) { for (int k = 0; k < 2; k++) solref[k] = mix*sr1[k] + (1-mix)*sr2[k]; }
          else { for (int k = 0; k < 2; k++) solref[k] = std::min(sr1[k], sr2[k]); }
          for (int k = 0; k < 5; k++) solimp[k] = mix*si1[k] + (1-mix)*si2[k];
          for (int k = 0; k < 3; k++) fri[k] = std::max(fr1[k], fr2[k]);
        }
        real friction[5] = {fri[0], fri[0], fri[1], fri[2], fri[2]};
        real margin = m->flex_margin[f1] + m->flex_margin[f2];
        if (override_) {
          margin = m->opt.o_margin;
          for (int k = 0; k < 2; k++) solref[k] = m->opt.o_solref[k];
          for (int k = 0; k < 5; k++) solimp[k] = m->opt.o_solimp[k];
          for (int k = 0; k < 5; k++) friction[k] = m->opt.o_friction[k];
        }
        for (int k = 0; k < 5; k++) friction[k] = std::max((real)mjMINMU, friction[k]);
        MJH_REJECT((solref[0] > 0) != (solref[1] > 0), "mixed-sign contact solref");
        H->flexff_flex.push_back(f1); H->flexff_flex.push_back(f2);
        H->flexff_pair.push_back((int)H->pair_geom1.size());
        {
          const bool tree = midphase && m->flex_bvhadr[f1] >= 0 && m->flex_bvhadr[f2] >= 0;
          H->flexff_mode.push_back(tree ? 1 : 2);
          if (tree) need_bvh = true;
        }
        H->flexjob_geom.push_back(-1);
        H->flexjob_nsub.push_back(0);
        H->flexjob_leaf.push_back(-1);
        H->pair_geom1.push_back(-1);
        H->pair_geom2.push_back(-1);
        H->pair_dim.push_back(condim);
        H->pair_margin.push_back(margin);
        H->pair_includemargin.push_back(margin);
        H->pair_adhesion.push_back(0);
        for (int k = 0; k < 5; k++) H->pair_friction.push_back(friction[k]);
        for (int k = 0; k < 2; k++) H->pair_solref.push_back(solref[k]);
        for (int k = 0; k < 2; k++) H->pair_solreffriction.push_back(0);
        for (int k = 0; k < 5; k++) H->pair_solimp.push_back(solimp[k]);
      }
    s.nflexff = (int)H->flexff_pair.size();
    H->flexself_flex.clear(); H->flexself_pair.clear(); H->flexself_mode.clear();
    for (int f1 = 0; f1 < m->nflex; f1++) {
      if (!(m->flex_contype[f1] || m->flex_conaffinity[f1])) continue;
      if (m->flex_rigid[f1] || !(m->flex_contype[f1] & m->flex_conaffinity[f1])) continue;
      MJH_REJECT(m->flex_internal[f1], "flex internal collisions");
      const int sc = m->flex_selfcollide[f1];
      if (sc == mjFLEXSELF_NONE) continue;
      int mode = 2;
      if (midphase && sc != mjFLEXSELF_NARROW && m->flex_bvhadr[f1] >= 0) {
        // (mode 1: the sweep-and-prune of mj_collideFlexSAP; mode 3: mj_collideTree of the flex's hierarchy against itself)
        mode = (sc == mjFLEXSELF_BVH || (sc == mjFLEXSELF_AUTO && m->flex_dim[f1] == 3)) ? 3 : 1;
        need_bvh = true;
      }
      MJH_REJECT(m->flex_dim[f1] != 1 && (m->opt.disableflags & mjDSBL_NATIVECCD), "the libccd convex collision pipeline (mjDSBL_NATIVECCD; libccd is a third-party library)");
      MJH_REJECT(m->flex_elemnum[f1] >= 0x10000, "flex self-collisions with 65536 or more elements");
      if (m->flex_dim[f1] != 1) s.ccd_any = 1;
      // parameter record: mj_contactParam(-1, -1, f, f) -- both sides the same flex, so the mixing rules return the flex's own
      // parameters (0.5 x + 0.5 x = x exactly); margin and gap are ignored in self-collisions (mj_collideElems :2524)
      const int f = f1;
      real friction[5] = {(real)m->flex_friction[3*f], (real)m->flex_friction[3*f], (real)m->flex_friction[3*f + 1],
                          (real)m->flex_friction[3*f + 2], (real)m->flex_friction[3*f + 2]};
      real solref[2], solimp[5];
      const real mix = m->flex_solmix[f] >= mjMINVAL ? m->flex_solmix[f]/(m->flex_solmix[f] + m->flex_solmix[f]) : 0.5;
      if (m->flex_solref[2*f] > 0) { for (int k = 0; k < 2; k++) solref[k] = mix*m->flex_solref[2*f + k] + (1 - mix)*m->flex_solref[2*f + k]; }
      else { for (int k = 0; k < 2; k++) solref[k] = m->flex_solref[2*f + k]; }
      for (int k = 0; k < 5; k++) solimp[k] = mix*m->flex_solimp[5*f + k] + (1 - mix)*m->flex_solimp[5*f + k];
      if (override_) {
        for (int k = 0; k < 2; k++) solref[k] = m->opt.o_solref[k];
        for (int k = 0; k < 5; k++) solimp[k] = m->opt.o_solimp[k];
        for (int k = 0; k < 5; k++) friction[k] = m->opt.o_friction[k];
      }
      for (int k = 0; k < 5; k++) friction[k] = std::max((real)mjMINMU, friction[k]);
      MJH_REJECT((solref[0] > 0) != (solref[1] > 0), "mixed-sign contact solref");
      H->flexself_flex.push_back(f);
      H->flexself_pair.push_back((int)H->pair_geom1.size());
      H->flexself_mode.push_back(mode);
      H->flexjob_geom.push_back(-1);
      H->flexjob_nsub.push_back(0);
      H->flexjob_leaf.push_back(-1);
      H->pair_geom1.push_back(-1);
      H->pair_geom2.push_back(-1);
      H->pair_dim.push_back(m->flex_condim[f]);
      H->pair_margin.push_back(0);
      H->pair_includemargin.push_back(0);
      H->pair_adhesion.push_back(0);
      for (int k = 0; k < 5; k++) H->pair_friction.push_back(friction[k]);
      for (int k = 0; k < 2; k++) H->pair_solref.push_back(solref[k]);
      for (int k = 0; k < 2; k++) H->pair_solreffriction.push_back(0);
      for (int k = 0; k < 5; k++) H->pair_solimp.push_back(solimp[k]);
    }
    s.nflexself = (int)H->flexself_flex.size();
    s.nflexpair = (int)H->flexjob_geom.size();
    s.njobbvh = (int)H->jobbvh_parent.size();
    // active elements of every flex, in element order (mj_isElemActive :347)
    H->flexact_adr.assign((size_t)m->nflex + 1, 0);
    H->flexact_elem.clear();
    for (int f = 0; f < m->nflex; f++) {
      H->flexact_adr[f] = (int)H->flexact_elem.size();
      if (s.nflexself)
        for (int t = 0; t < m->flex_elemnum[f]; t++)
          if (m->flex_dim[f] < 3 || m->flex_elemlayer[m->flex_elemadr[f] + t] < m->flex_activelayers[f]) H->flexact_elem.push_back(m->flex_elemadr[f] + t);
    }
    H->flexact_adr[m->nflex] = (int)H->flexact_elem.size();
    s.nflexact = (int)H->flexact_elem.size();
    // the flexes' bounding volume hierarchies (only where the sweep axis of a self-collision is read off the root box)
    H->flexbvh_adr.assign((size_t)m->nflex + 1, 0);
    H->flexbvh_child.clear(); H->flexbvh_elem.clear(); H->flexbvh_order.clear(); H->flexbvh_hadr.clear();
    {
      std::vector<int> height;
      for (int f = 0; f < m->nflex; f++) {
        const int base = (int)H->flexbvh_elem.size();
        H->flexbvh_adr[f] = base;
        const int adr = m->flex_bvhadr[f], num = need_bvh && adr >= 0 ? m->flex_bvhnum[f] : 0;
        for (int i = 0; i < num; i++) {
          const int c0 = m->bvh_child[2*(adr + i)], c1 = m->bvh_child[2*(adr + i) + 1], id = m->bvh_nodeid[adr + i];
          MJH_REJECT(id < 0 && (c0 <= i || c1 <= i || c0 >= num || c1 >= num), "internal: flex bounding volume hierarchy (children)");
          H->flexbvh_child.push_back(id >= 0 ? -1 : base + c0);
          H->flexbvh_child.push_back(id >= 0 ? -1 : base + c1);
          H->flexbvh_elem.push_back(id >= 0 ? m->flex_elemadr[f] + id : -1);
        }
        height.resize(base + num, 0);
        for (int i = num - 1; i >= 0; i--)
          if (H->flexbvh_elem[base + i] < 0)
            height[base + i] = 1 + std::max(height[H->flexbvh_child[2*(base + i)]], height[H->flexbvh_child[2*(base + i) + 1]]);
      }
      H->flexbvh_adr[m->nflex] = (int)H->flexbvh_elem.size();
      s.nflexbvh = (int)H->flexbvh_elem.size();
      // parents, and the leaf of every element (the path of a leaf pair through mj_collideTree's walk)
      H->flexbvh_parent.assign(s.nflexbvh, -1);
      s.nselfbvh = 0;
      for (int md : H->flexself_mode) if (md == 3) s.nselfbvh++;
      H->flexelem_bvhleaf.assign((s.nflexff || s.njobbvh || s.nselfbvh) ? m->nflexelem : 0, -1);
      for (int i = 0; i < s.nflexbvh; i++) {
        if (H->flexbvh_elem[i] < 0) { H->flexbvh_parent[H->flexbvh_child[2*i]] = i; H->flexbvh_parent[H->flexbvh_child[2*i + 1]] = i; }
        else if (s.nflexff || s.njobbvh || s.nselfbvh) H->flexelem_bvhleaf[H->flexbvh_elem[i]] = i;
      }
      for (int k = 0; k + 1 < s.ncolseg; k++)
        if (H->flexjob_adr[k] < H->flexjob_adr[k + 1] && H->flexjob_leaf[H->flexjob_adr[k]] >= 0) {
          int depth = 0;
          for (int a = H->flexjob_adr[k]; a < H->flexjob_adr[k + 1]; a++) {
            int dd = 0;
            for (int nd = H->flexjob_leaf[a]; nd >= 0; nd = H->jobbvh_parent[nd]) dd++;
            depth = std::max(depth, dd);
          }
          MJH_REJECT(depth > 30 || height[H->flexbvh_adr[H->colseg[3*k + 2]]] + depth > 50,
                     "body : flex collisions with bounding volume hierarchies deeper than 50 levels together");
        }
      for (int k = 0; k < (int)H->flexself_mode.size(); k++)
        if (H->flexself_mode[k] == 3)
          MJH_REJECT(height[H->flexbvh_adr[H->flexself_flex[k]]] > 26, "flex self-collisions through a bounding volume hierarchy deeper than 26 levels");
      for (int k = 0; k < s.nflexff; k++)
        if (H->flexff_mode[k] == 1)
          for (int side = 0; side < 2; side++) {
            const int f = H->flexff_flex[2*k + side];
            MJH_REJECT(height[H->flexbvh_adr[f]] > 26, "flex : flex collisions with a bounding volume hierarchy deeper than 26 levels");
          }
      int hmax = 0;
      for (int h : height) hmax = std::max(hmax, h);
      s.nflexbvhh = s.nflexbvh ? hmax + 1 : 0;
      for (int h = 0; h < s.nflexbvhh; h++) {
        H->flexbvh_hadr.push_back((int)H->flexbvh_order.size());
        for (int i = 0; i < s.nflexbvh; i++) if (height[i] == h) H->flexbvh_order.push_back(i);
      }
      H->flexbvh_hadr.push_back((int)H->flexbvh_order.size());
    }
    // BVH leaves of every flex in depth-first order (second child first: mj_collideTree pushes child 0 then child 1 and pops the last)
    H->flex_leafadr.assign((size_t)m->nflex + 1, 0);
    H->flexleaf_elem.clear();
    for (int f = 0; f < m->nflex; f++) {
      H->flex_leafadr[f] = (int)H->flexleaf_elem.size();
      const int adr = m->flex_bvhadr[f];
      if (adr < 0 || m->flex_bvhnum[f] <= 0) continue;
      std::vector<int> stack{0};
      while (!stack.empty()) {
        const int node = stack.back(); stack.pop_back();
        const int c0 = m->bvh_child[2*(adr + node)], c1 = m->bvh_child[2*(adr + node) + 1];
        if (c0 < 0 && c1 < 0) {
          if (m->bvh_nodeid[adr + node] >= 0) H->flexleaf_elem.push_back(m->flex_elemadr[f] + m->bvh_nodeid[adr + node]);
          continue;
        }
        if (c0 >= 0) stack.push_back(c0);
        if (c1 >= 0) stack.push_back(c1);
      }
    }
    H->flex_leafadr[m->nflex] = (int)H->flexleaf_elem.size();
    s.nflexleaf = (int)H->flexleaf_elem.size();
    // smallest tree of every flex whose stiffness couples its vertices (dim >= 2, bending or a non-zero stiffness), else -1
    H->flex_mintree.assign(m->nflex, -1);
    for (int f = 0; f < m->nflex; f++) {
      if (m->flex_rigid[f] || m->flex_dim[f] < 2) continue;
      const int sadr = m->flex_stiffnessadr[f];
      if (m->flex_bendingadr[f] < 0 && (sadr < 0 || m->flex_stiffness[sadr] == 0)) continue;
      int mt = -1;
      if (m->flex_interp[f])
        for (int i = m->flex_nodeadr[f]; i < m->flex_nodeadr[f] + m->flex_nodenum[f]; i++) {
          const int tr = m->body_treeid[m->flex_nodebodyid[i]];
          if (tr >= 0 && (mt < 0 || tr < mt)) mt = tr;
        }
      for (int v = m->flex_vertadr[f]; v < m->flex_vertadr[f] + m->flex_vertnum[f]; v++) {
        const int tr = m->flex_vertbodyid[v] < 0 ? -1 : m->body_treeid[m->flex_vertbodyid[v]];
        if (tr >= 0 && (mt < 0 || tr < mt)) mt = tr;
      }
      H->flex_mintree[f] = mt;
    }
    copy_arr(H->flex_contype, m->flex_contype, m->nflex);
    copy_arr(H->flex_conaffinity, m->flex_conaffinity, m->nflex);
    s.ngeomflex = m->nflex ? m->ngeom : 0;
    copy_arr(H->geom_contype, m->geom_contype, s.ngeomflex);
    copy_arr(H->geom_conaffinity, m->geom_conaffinity, s.ngeomflex);
    // candidate capacity of one job: every vertex against every plane of the body, every BVH leaf against every other geom
    int cand = 0;
    for (int k = 0; k + 1 < s.ncolseg; k++) {
      const int f = H->colseg[3*k + 2];
      int c = 0;
      for (int a = H->flexjob_adr[k]; a < H->flexjob_adr[k + 1]; a++)
        c += m->geom_type[H->flexjob_geom[a]] == mjGEOM_PLANE ? m->flex_vertnum[f]
                                                                : (H->flex_leafadr[f + 1] - H->flex_leafadr[f])*std::max(1, H->flexjob_nsub[a]);
      // (a multi-geom body's candidates are put in walk order in the second half of the table)
      if (H->flexjob_adr[k] < H->flexjob_adr[k + 1] && H->flexjob_leaf[H->flexjob_adr[k]] >= 0) c *= 2;
      cand = std::max(cand, c);
    }
    // self-collisions: the pairs of elements whose boxes overlap (beyond the capacity the environment raises the mjhip-only
    // UNSUPPORTED warning); their contacts are put in order in the second half of the table
    for (int k = 0; k < s.nflexself; k++) cand = std::max(cand, 2*16*(int)m->flex_elemnum[H->flexself_flex[k]]);
    for (int k = 0; k < s.nflexff; k++) cand = std::max(cand, 2*16*(int)std::max(m->flex_elemnum[H->flexff_flex[2*k]], m->flex_elemnum[H->flexff_flex[2*k + 1]]));
    s.nflexcand = cand;
  }

  // ---------------- broad / midphase emulation tables (mjh_collision.h: stage_broadphase) ----------------
  // The reference culls BODY pairs by sweep-and-prune over float-rounded bounding intervals in a
  // PCA frame (mj_broadphase / mj_SAP, engine_collision_driver.c:1439-1735) and, for bodies with
  // several geoms, walks two BVHs testing oriented boxes (mj_collideTree / mj_collideOBB, :898-1240).
  // Both are functions of the current poses; which tests a given geom pair has to pass is static:
  //   * sweep-and-prune: the pair of positions of its bodies in the collidable-body list, unless one
  //     body is always paired (world geoms, static bodies with planes, :1588-1624);
  //   * midphase: the chain of BVH node pairs from (root, root) down to the pair's two leaves -- the
  //     descent rule (a leaf never descends, else the box with the larger surface does, :1185-1236)
  //     only reads model constants.
  {
    std::vector<int> pos(m->nbody, -1);
    for (int b = 1; b < m->nbody; b++)
      if (m->body_contype[b] || m->body_conaffinity[b]) { pos[b] = (int)H->bp_body.size(); H->bp_body.push_back(b); }
    auto always = [&](int b) {
      if (b == 0) return m->body_geomnum[0] > 0;
      if (m->body_dofnum[m->body_weldid[b]] != 0) return false;
      for (int g = m->body_geomadr[b]; g < m->body_geomadr[b] + m->body_geomnum[b]; g++)
        if (m->geom_type[g] == mjGEOM_PLANE) return true;
      return false;
    };
    H->pair_sap.assign(s.npair, -1);
    H->pair_mid_adr.assign((size_t)s.npair + 1, 0);
    H->pair_bodymargin.assign(s.npair, 0);
    bool any_sap = false;
    for (int p = 0; p < s.npair; p++) {
      H->pair_mid_adr[p] = (int)H->pair_mid.size()/2;
      const int b1 = pair_b1[p], b2 = pair_b2[p];
      if (b1 < 0) continue;                                   // predefined pair: never culled by body
      if (!always(b1) && !always(b2)) {
        MJH_REJECT(pos[b1] < 0 || pos[b2] < 0 || H->bp_body.size() >= 0x8000, "internal: broadphase body list");
        H->pair_sap[p] = std::min(pos[b1], pos[b2]) | (std::max(pos[b1], pos[b2]) << 16);
        any_sap = true;
      }
      if (H->pair_route[p] != 2) continue;
      const real bm = m->body_margin[b1] + m->body_margin[b2];
      H->pair_bodymargin[p] = override_ ? (real)m->opt.o_margin : bm;
      // which geom of the (type-ordered) pair belongs to which body
      const int ga = m->geom_bodyid[H->pair_geom1[p]] == b1 ? H->pair_geom1[p] : H->pair_geom2[p];
      const int gb = ga == H->pair_geom1[p] ? H->pair_geom2[p] : H->pair_geom1[p];
      const int adr1 = m->body_bvhadr[b1], adr2 = m->body_bvhadr[b2];
      const int* child1 = m->bvh_child + 2*adr1;
      const int* child2 = m->bvh_child + 2*adr2;
      // does the subtree of `node` hold the leaf of geom g?
      std::function<bool(const int*, int, int, int)> holds = [&](const int* child, int adr, int node, int g) -> bool {
        if (node < 0) return false;
        if (child[2*node] < 0 && child[2*node + 1] < 0) return m->bvh_nodeid[adr + node] == g;
        return holds(child, adr, child[2*node], g) || holds(child, adr, child[2*node + 1], g);
      };
      int n1 = 0, n2 = 0;
      for (int guard = 0; guard < 4096; guard++) {
        const bool leaf1 = child1[2*n1] < 0 && child1[2*n1 + 1] < 0;
        const bool leaf2 = child2[2*n2] < 0 && child2[2*n2 + 1] < 0;
        if (leaf1 && leaf2) break;
        H->pair_mid.push_back(adr1 + n1);
        H->pair_mid.push_back(adr2 + n2);
        bool down1;
        if (!leaf1 && leaf2) down1 = true;
        else if (leaf1 && !leaf2) down1 = false;
        else {
          const mjtNum* a1 = m->bvh_aabb + 6*(adr1 + n1);
          const mjtNum* a2 = m->bvh_aabb + 6*(adr2 + n2);
          const mjtNum x1 = a1[3] - a1[0], y1 = a1[4] - a1[1], z1 = a1[5] - a1[2];
          const mjtNum x2 = a2[3] - a2[0], y2 = a2[4] - a2[1], z2 = a2[5] - a2[2];
          down1 = (x1*y1 + y1*z1 + z1*x1) > (x2*y2 + y2*z2 + z2*x2);
        }
        if (down1) n1 = holds(child1, adr1, child1[2*n1], ga) ? child1[2*n1] : child1[2*n1 + 1];
        else n2 = holds(child2, adr2, child2[2*n2], gb) ? child2[2*n2] : child2[2*n2 + 1];
        MJH_REJECT(n1 < 0 || n2 < 0, "internal: BVH chain");
      }
      MJH_REJECT(m->bvh_nodeid[adr1 + n1] != ga || m->bvh_nodeid[adr2 + n2] != gb, "internal: BVH leaves");
    }
    H->pair_mid_adr[s.npair] = (int)H->pair_mid.size()/2;
    s.nmid = (int)H->pair_mid.size()/2;
    s.nbp = (int)H->bp_body.size();
    bool any_mid = false;
    for (int p = 0; p < s.npair; p++) if (H->pair_route[p] == 2) any_mid = true;
    if (!any_sap && !any_mid) { s.nbp = 0; H->bp_body.clear(); }       // nothing to cull: the stage is skipped
    s.nbvh = (int)m->nbvhstatic;
    copy_arr(H->bvh_aabb, m->bvh_aabb, 6*(int)m->nbvhstatic);
    copy_arr(H->geom_aabb, m->geom_aabb, 6*m->ngeom);
    H->geom_bpmargin.resize(m->ngeom);
    for (int g = 0; g < m->ngeom; g++)
      H->geom_bpmargin[g] = override_ ? (real)(0.5*m->opt.o_margin) : (real)(m->geom_margin[g] + m->geom_gap[g]);
    // reach of a body around any of its geom centres: an upper bound on how far an end point of the
    // body's bounding interval can lie from the projection of that centre (geoms are rigidly attached,
    // so centre-to-centre distances are pose independent).  Bounds the float rounding of the sweep.
    H->body_bpext.assign(m->nbody, 0);
    for (int b = 0; b < m->nbody; b++) {
      real far = 0, pad = 0;
      const int g0 = m->body_geomadr[b], g1 = g0 + m->body_geomnum[b];
      for (int g = g0; g < g1; g++) {
        pad = std::max(pad, (real)(m->geom_rbound[g] + std::fabs(H->geom_bpmargin[g])));
        if (m->geom_rbound[g] <= 0) pad = 1e30;            // planes: no finite interval, always ask
        for (int h = g0; h < g1; h++) {
          real d2 = 0;
          for (int k = 0; k < 3; k++) { real d = m->geom_pos[3*g + k] - m->geom_pos[3*h + k]; d2 += d*d; }
          far = std::max(far, (real)std::sqrt(d2));
        }
      }
      H->body_bpext[b] = far + pad;
    }
    s.npassw = (s.npair + 31)/32;
    s.bp_any_sap = any_sap ? 1 : 0;
    s.bp_any_mid = any_mid ? 1 : 0;
  }

  // ---------------- convex meshes + GJK / EPA workspace (mjh_convex.h) -----------------------------------
  {
    const bool any = s.ccd_any != 0;
    s.nmesh = any ? (int)m->nmesh : 0;
    s.nmeshvert = any ? (int)m->nmeshvert : 0;
    s.nmeshgraph = any ? (int)m->nmeshgraph : 0;
    s.nmeshpoly = any ? (int)m->nmeshpoly : 0;
    s.nmeshpolyvert = any ? (int)m->nmeshpolyvert : 0;
    s.nmeshpolymap = any ? (int)m->nmeshpolymap : 0;
    copy_arr(H->geom_dataid, m->geom_dataid, m->ngeom);
    copy_arr(H->mesh_vertadr, m->mesh_vertadr, s.nmesh);
    copy_arr(H->mesh_vertnum, m->mesh_vertnum, s.nmesh);
    copy_arr(H->mesh_graphadr, m->mesh_graphadr, s.nmesh);
    copy_arr(H->mesh_polynum, m->mesh_polynum, s.nmesh);
    copy_arr(H->mesh_polyadr, m->mesh_polyadr, s.nmesh);
    copy_arr(H->mesh_graph, m->mesh_graph, s.nmeshgraph);
    copy_arr(H->mesh_extrema, m->mesh_extrema, 27*(size_t)s.nmesh);
    copy_arr(H->mesh_polyvertadr, m->mesh_polyvertadr, s.nmeshpoly);
    copy_arr(H->mesh_polyvertnum, m->mesh_polyvertnum, s.nmeshpoly);
    copy_arr(H->mesh_polyvert, m->mesh_polyvert, s.nmeshpolyvert);
    copy_arr(H->mesh_polymapadr, m->mesh_polymapadr, s.nmeshvert);
    copy_arr(H->mesh_polymapnum, m->mesh_polymapnum, s.nmeshvert);
    copy_arr(H->mesh_polymap, m->mesh_polymap, s.nmeshpolymap);
    copy_arr(H->mesh_vert, m->mesh_vert, 3*(size_t)s.nmeshvert);          // float -> double: exact
    copy_arr(H->mesh_polynormal, m->mesh_polynormal, 3*(size_t)s.nmeshpoly);
    o.ccd_tolerance = m->opt.ccd_tolerance;
    o.ccd_sin = std::sin(0.5*1e-3);
    o.ccd_cos = std::cos(0.5*1e-3);
    if (any) {
      MJH_REJECT(m->opt.ccd_iterations < 1 || m->opt.ccd_iterations > 200, "opt.ccd_iterations outside 1..200 (bounds the per-lane EPA workspace)");
      s.ccd_N = m->opt.ccd_iterations;
      s.ccd_P = std::max(4, (int)m->npolygonmax);
      s.ccd_D = std::max(3, (int)m->nmeshdegmax);
      // mirror of the row workspace layout (mjh_convex.h: RO_* / IO_* offsets, rc_attach)
      const int N = s.ccd_N, P = s.ccd_P, D = s.ccd_D;
      const int VFAST = 10, FFAST = 24, MFAST = 24, HFAST = 12, KFAST = 12;
      const int poly_r = 6*VFAST + 4*FFAST, clip_r = 9*D + 18*P;
      s.ccd_row_freal = 96 + std::max(poly_r, clip_r);
      const int poly_i = 2*VFAST + 6*FFAST + MFAST + 2*HFAST + KFAST, clip_i = 2*D;
      const int row_i = 32 + std::max(poly_i, clip_i);
      s.ccd_row_reals = s.ccd_row_freal + (row_i + 1)/2;
      // overflow page of a row: full-capacity vertex / face / map / horizon / crossing-stack arrays
      const int nsv = 5 + N, nsf = 6*N;
      s.ccd_slow_bytes = (6*nsv + 4*nsf)*(int)sizeof(real) + ((2*nsv + 6*nsf + nsf + 2*nsf + 2*nsf + 16 + 1) & ~1)*(int)sizeof(int);
      // (global-memory sizing only: the LDS plan of a one-wavefront mapping places 4 rows whatever this says -- plan_lds, mjh_runtime.h)
      s.ccd_rows = m->nflex > 0 ? 4*MJH_MW : 4;
      // (header, contact records, overflow pages, and the rows' fast pages for launches whose LDS plan has no room for them)
      // polyhedral pairs (rc_max_contacts > 1, mjh_convex.h): tables of their distance phase (rc_poly_tables)
      s.ccd_npoly = 0;
      {
        const bool multiccd = !(m->opt.disableflags & (1 << 19));
        for (int p = 0; p < s.npair; p++) {
          const int t1 = m->geom_type[H->pair_geom1[p]], t2 = m->geom_type[H->pair_geom2[p]];
          if (multiccd && !(H->pair_margin[p] > 0) && (t1 == mjGEOM_BOX || t1 == mjGEOM_MESH) && (t2 == mjGEOM_BOX || t2 == mjGEOM_MESH))
            s.ccd_npoly++;
        }
      }
      const int poly_tables = s.ccd_npoly*32*(int)sizeof(real) + ((s.npair + 2*s.ccd_npoly + 1) & ~1)*(int)sizeof(int);
      s.ccd_env_bytes = 256*(int)sizeof(int) + 64*5*7*(int)sizeof(real) + s.ccd_rows*s.ccd_slow_bytes +
                        s.ccd_rows*s.ccd_row_reals*(int)sizeof(real) + poly_tables;
    }
  }

  // ---------------- capacities ---------------------------------------------------------------------------
  int nlimit = 0;
  for (int i = 0; i < m->njnt; i++) if (m->jnt_limited[i]) nlimit += (m->jnt_type[i] == mjJNT_BALL) ? 1 : 2;
  for (int i = 0; i < m->ntendon; i++) if (m->tendon_limited[i]) nlimit += 2;
  int nfric = 0;
  for (int i = 0; i < m->nv; i++) if (m->dof_frictionloss[i] != 0) nfric++;
  s.ndoffric = nfric;
  s.njntlim = 0;
  for (int i = 0; i < m->njnt; i++) if (m->jnt_limited[i]) s.njntlim++;
  for (int i = 0; i < m->ntendon; i++) if (m->tendon_frictionloss[i] > 0) nfric++;
  // contact capacity: the bound mj_maxContact gives the static pair list (the reference's arena grows
  // on demand; a fixed 512 stands in for "as many as a scene of this size can touch at once")
  // (a body : flex job leaves at most mjMAXCONPAIR contacts, filterFlexContacts :447-515)
  for (int k = 0; k + 1 < s.ncolseg; k++) maxcon_total += std::min(s.nflexcand, (int)mjMAXCONPAIR);
  maxcon_total += (s.nflexself + s.nflexff)*(int)mjMAXCONPAIR;              // (and so does a flex's collision with itself, mj_collision :878)
  s.nconmax = caps.nconmax > 0 ? caps.nconmax : std::max(1, std::min(maxcon_total, 512));
  s.nconflex = m->nflex ? s.nconmax : 0;
  s.nconlds = std::min(s.nconmax, 8);
  // tree ids (constraint islands, engine_island.c)
  H->body_treeid.assign(m->body_treeid, m->body_treeid + m->nbody);
  H->dof_treeid.assign(m->dof_treeid, m->dof_treeid + m->nv);
  H->tree_dofadr.assign(m->tree_dofadr, m->tree_dofadr + m->ntree);
  H->tree_dofnum.assign(m->tree_dofnum, m->tree_dofnum + m->ntree);
  // L'DL fast path tables (mjh_smooth.h: factor_ld / solve_ld)
  {
    const int nv = m->nv;
    int maxdepth = 0;
    for (int i = 0; i < nv; i++) maxdepth = std::max(maxdepth, m->M_rownnz[i] - 1);
    s.ld_fast = (nv <= 64 && m->nC <= 1024 && maxdepth <= 16) ? 1 : 0;
    H->dof_ancmask.assign(2*(size_t)nv, 0);
    H->ld_prog_adr.assign((size_t)nv + 1, 0);
    H->ld_prog.clear();
    if (s.ld_fast) {
      for (int i = 0; i < nv; i++) {
        unsigned long long mask = 0;
        for (int a = 0; a < m->M_rownnz[i] - 1; a++) mask |= 1ull << m->M_colind[m->M_rowadr[i] + a];
        H->dof_ancmask[2*i] = (int)(unsigned)(mask & 0xffffffffu);
        H->dof_ancmask[2*i + 1] = (int)(unsigned)(mask >> 32);
      }
      // items are stored pivot by pivot in the order the factorisation visits them (k = nv-1 .. 0)
      std::vector<std::vector<int>> per(nv);
      for (int k = 0; k < nv; k++) {
        int start = m->M_rowadr[k], diag = m->M_rownnz[k] - 1;
        for (int a = 0; a < diag; a++) {
          int i = m->M_colind[start + a];
          for (int el = 0; el <= a; el++)
            per[k].push_back((m->M_rowadr[i] + el) | ((start + el) << 10) | ((start + a) << 20));
        }
      }
      for (int k = 0; k < nv; k++) {
        H->ld_prog_adr[k] = (int)H->ld_prog.size();
        H->ld_prog.insert(H->ld_prog.end(), per[k].begin(), per[k].end());
      }
      H->ld_prog_adr[nv] = (int)H->ld_prog.size();
    }
    if (H->ld_prog.empty()) H->ld_prog.push_back(0);
    s.nldprog = (int)H->ld_prog.size();
    H->ld_rows.clear();
    for (int i = 0; i < nv; i++) if (m->M_rownnz[i] > 1) H->ld_rows.push_back(i);
    s.nldrows = (int)H->ld_rows.size();
  }
  int rows_per_con = 1;
  for (int c : H->pair_dim)
    rows_per_con = std::max(rows_per_con, c == 1 ? 1 : (m->opt.cone == mjCONE_PYRAMIDAL ? 2*(c-1) : c));
  // ---- sensors (engine_sensor.c): kinds translated to the device enum, frame objects to MJH_OBJ_*
  s.nmocap = m->nmocap;
  s.nuserdata = m->nuserdata;
  s.nsensor = m->nsensor;
  s.nsensordata = m->nsensordata;
  s.nbody_sens = m->nsensor ? m->nbody : 0;
  s.sens_rnepost = 0; s.sens_subtreevel = 0; s.sens_energy = 0;
  H->sensor_type.assign(m->nsensor, 0);
  H->sensor_objtype.assign(m->nsensor, MJH_OBJ_NONE);
  H->sensor_reftype.assign(m->nsensor, MJH_OBJ_NONE);
  copy_arr(H->sensor_datatype, m->sensor_datatype, m->nsensor);
  copy_arr(H->sensor_objid, m->sensor_objid, m->nsensor);
  copy_arr(H->sensor_refid, m->sensor_refid, m->nsensor);
  copy_arr(H->sensor_dim, m->sensor_dim, m->nsensor);
  copy_arr(H->sensor_adr, m->sensor_adr, m->nsensor);
  copy_arr(H->sensor_cutoff, m->sensor_cutoff, m->nsensor);
  copy_arr(H->sensor_needstage, m->sensor_needstage, m->nsensor);
  H->sensor_intprm0.resize(m->nsensor);
  for (int i = 0; i < m->nsensor; i++) H->sensor_intprm0[i] = m->sensor_intprm[i*mjNSENS];
  H->sensor_intprm1.resize(m->nsensor);
  for (int i = 0; i < m->nsensor; i++) H->sensor_intprm1[i] = m->sensor_intprm[i*mjNSENS + 1];
  // geoms a ray never sees: fully transparent colour or material (ray_eliminate, engine_ray.c:74-82)
  H->geom_rayskip.assign(m->ngeom, 0);
  for (int g = 0; g < m->ngeom; g++) {
    const int mat = m->geom_matid[g];
    if ((mat < 0 && m->geom_rgba[4*g + 3] == 0) || (mat >= 0 && m->mat_rgba[4*mat + 3] == 0)) H->geom_rayskip[g] = 1;
  }
  bool use_cameras = false;
  for (int i = 0; i < m->nsensor; i++) {
    int t = -1;
    switch (m->sensor_type[i]) {
      case mjSENS_JOINTPOS: t = MJH_SENS_JOINTPOS; break;
      case mjSENS_JOINTVEL: t = MJH_SENS_JOINTVEL; break;
      case mjSENS_TENDONPOS: t = MJH_SENS_TENDONPOS; break;
      case mjSENS_TENDONVEL: t = MJH_SENS_TENDONVEL; break;
      case mjSENS_ACTUATORPOS: t = MJH_SENS_ACTUATORPOS; break;
      case mjSENS_ACTUATORVEL: t = MJH_SENS_ACTUATORVEL; break;
      case mjSENS_ACTUATORFRC: t = MJH_SENS_ACTUATORFRC; break;
      case mjSENS_JOINTACTFRC: t = MJH_SENS_JOINTACTFRC; break;
      case mjSENS_BALLQUAT: t = MJH_SENS_BALLQUAT; break;
      case mjSENS_BALLANGVEL: t = MJH_SENS_BALLANGVEL; break;
      case mjSENS_JOINTLIMITPOS: t = MJH_SENS_JOINTLIMITPOS; break;
      case mjSENS_JOINTLIMITVEL: t = MJH_SENS_JOINTLIMITVEL; break;
      case mjSENS_JOINTLIMITFRC: t = MJH_SENS_JOINTLIMITFRC; break;
      case mjSENS_TENDONLIMITPOS: t = MJH_SENS_TENDONLIMITPOS; break;
      case mjSENS_TENDONLIMITVEL: t = MJH_SENS_TENDONLIMITVEL; break;
      case mjSENS_TENDONLIMITFRC: t = MJH_SENS_TENDONLIMITFRC; break;
      case mjSENS_FRAMEPOS: t = MJH_SENS_FRAMEPOS; break;
      case mjSENS_FRAMEQUAT: t = MJH_SENS_FRAMEQUAT; break;
      case mjSENS_FRAMEXAXIS: t = MJH_SENS_FRAMEXAXIS; break;
      case mjSENS_FRAMEYAXIS: t = MJH_SENS_FRAMEYAXIS; break;
      case mjSENS_FRAMEZAXIS: t = MJH_SENS_FRAMEZAXIS; break;
      case mjSENS_FRAMELINVEL: t = MJH_SENS_FRAMELINVEL; break;
      case mjSENS_FRAMEANGVEL: t = MJH_SENS_FRAMEANGVEL; break;
      case mjSENS_FRAMELINACC: t = MJH_SENS_FRAMELINACC; s.sens_rnepost = 1; break;
      case mjSENS_FRAMEANGACC: t = MJH_SENS_FRAMEANGACC; s.sens_rnepost = 1; break;
      case mjSENS_SUBTREECOM: t = MJH_SENS_SUBTREECOM; break;
      case mjSENS_SUBTREELINVEL: t = MJH_SENS_SUBTREELINVEL; s.sens_subtreevel = 1; break;
      case mjSENS_SUBTREEANGMOM: t = MJH_SENS_SUBTREEANGMOM; s.sens_subtreevel = 1; break;
      case mjSENS_CLOCK: t = MJH_SENS_CLOCK; break;
      case mjSENS_GEOMDIST: case mjSENS_GEOMNORMAL: case mjSENS_GEOMFROMTO: {
        // mj_geomDistance (engine_support.c:553) through the closed-form colliders: planes, spheres, capsules, and a sphere
        // against a box / cylinder.  Pairs that take the convex pipeline (mjc_ccd with a distance cutoff) or a
        // wave-cooperative box collider are not evaluated by the sensor stage.
        const int ot = m->sensor_objtype[i], rt = m->sensor_reftype[i];
        bool ok = (ot == mjOBJ_GEOM || ot == mjOBJ_BODY) && (rt == mjOBJ_GEOM || rt == mjOBJ_BODY);
        if (ok) {
          const int n1 = ot == mjOBJ_BODY ? m->body_geomnum[m->sensor_objid[i]] : 1, a1 = ot == mjOBJ_BODY ? m->body_geomadr[m->sensor_objid[i]] : m->sensor_objid[i];
          const int n2 = rt == mjOBJ_BODY ? m->body_geomnum[m->sensor_refid[i]] : 1, a2 = rt == mjOBJ_BODY ? m->body_geomadr[m->sensor_refid[i]] : m->sensor_refid[i];
          for (int g1 = a1; g1 < a1 + n1 && ok; g1++)
            for (int g2 = a2; g2 < a2 + n2 && ok; g2++) {
              int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
              if (t1 > t2) std::swap(t1, t2);
              const bool point = (t1 == mjGEOM_PLANE || t1 == mjGEOM_SPHERE || t1 == mjGEOM_CAPSULE) && (t2 == mjGEOM_PLANE || t2 == mjGEOM_SPHERE || t2 == mjGEOM_CAPSULE);
              const bool sph = t1 == mjGEOM_SPHERE && (t2 == mjGEOM_BOX || t2 == mjGEOM_CYLINDER);
              if (!point && !sph) ok = false;
            }
        }
        if (ok) {
          t = m->sensor_type[i] == mjSENS_GEOMDIST ? MJH_SENS_GEOMDIST : (m->sensor_type[i] == mjSENS_GEOMNORMAL ? MJH_SENS_GEOMNORMAL : MJH_SENS_GEOMFROMTO);
          H->sensor_objtype[i] = ot == mjOBJ_BODY ? MJH_OBJ_BODY : MJH_OBJ_GEOM;
          H->sensor_reftype[i] = rt == mjOBJ_BODY ? MJH_OBJ_BODY : MJH_OBJ_GEOM;
        }
        break;
      }
      case mjSENS_E_POTENTIAL: t = MJH_SENS_E_POTENTIAL; s.sens_energy |= 1; break;
      case mjSENS_E_KINETIC: t = MJH_SENS_E_KINETIC; s.sens_energy |= 2; break;
      case mjSENS_VELOCIMETER: t = MJH_SENS_VELOCIMETER; break;
      case mjSENS_GYRO: t = MJH_SENS_GYRO; break;
      case mjSENS_ACCELEROMETER: t = MJH_SENS_ACCELEROMETER; s.sens_rnepost = 1; break;
      case mjSENS_FORCE: t = MJH_SENS_FORCE; s.sens_rnepost = 1; break;
      case mjSENS_TORQUE: t = MJH_SENS_TORQUE; s.sens_rnepost = 1; break;
      case mjSENS_MAGNETOMETER: t = MJH_SENS_MAGNETOMETER; break;
      case mjSENS_INSIDESITE: t = MJH_SENS_INSIDESITE; break;
      case mjSENS_TENDONACTFRC: t = MJH_SENS_TENDONACTFRC; break;
      case mjSENS_CAMPROJECTION:
        // (round 6: a site projected into a camera's image, engine_sensor.c:541-545; the focal lengths are model constants)
        if (m->sensor_objtype[i] == mjOBJ_SITE && m->sensor_reftype[i] == mjOBJ_CAMERA) { t = MJH_SENS_CAMPROJECTION; use_cameras = true; }
        break;
      case mjSENS_RANGEFINDER:
        // site-attached rangefinders (camera-attached ones cast a ray per pixel: not evaluated)
        if (m->sensor_objtype[i] == mjOBJ_SITE) {
          t = MJH_SENS_RANGEFINDER;
          // (rays against primitives only: mj_rayMesh / mj_rayHfield / mj_raySdf are not evaluated)
          for (int g = 0; g < m->ngeom; g++)
            MJH_REJECT(!H->geom_rayskip[g] && m->geom_bodyid[g] != m->site_bodyid[m->sensor_objid[i]] &&
                       (m->geom_type[g] == mjGEOM_MESH || m->geom_type[g] == mjGEOM_HFIELD || m->geom_type[g] == mjGEOM_SDF),
                       "rangefinders in models with visible mesh / height-field / signed-distance-field geoms");
        }
        break;
      case mjSENS_CONTACT:
        // contact sensors: matching by site / geom / body / subtree, the four reductions (engine_sensor.c:1027-1150)
        t = MJH_SENS_CONTACT;
        break;
      case mjSENS_TOUCH: {
        const int st = m->site_type[m->sensor_objid[i]];
        // (the zone test is mju_rayGeom on the site: every primitive the rangefinders intersect)
        if (st == mjGEOM_SPHERE || st == mjGEOM_ELLIPSOID || st == mjGEOM_BOX || st == mjGEOM_CAPSULE || st == mjGEOM_CYLINDER) t = MJH_SENS_TOUCH;
        break;
      }
      default: break;
    }
    MJH_REJECT(t < 0, "sensor types other than joint/tendon/actuator/ball/limit/frame/subtree/clock/IMU/force/torque/magnetometer/insidesite/"
                      "touch / site rangefinders / contact / camprojection / energy / geom distance between planes, spheres and "
                      "capsules (camera rangefinders, geom distance through the convex pipeline or the box colliders, tactile, user, plugin)");
    H->sensor_type[i] = t;
    auto frame_obj = [&](int ot, int* out) -> bool {
      if (ot == mjOBJ_BODY) *out = MJH_OBJ_BODY;
      else if (ot == mjOBJ_XBODY) *out = MJH_OBJ_XBODY;
      else if (ot == mjOBJ_GEOM) *out = MJH_OBJ_GEOM;
      else if (ot == mjOBJ_SITE) *out = MJH_OBJ_SITE;
      else if (ot == mjOBJ_CAMERA) { *out = MJH_OBJ_CAMERA; use_cameras = true; }     // (round 6: camera frames, mj_camlight)
      else return false;
      return true;
    };
    if (t == MJH_SENS_CAMPROJECTION) { H->sensor_objtype[i] = MJH_OBJ_SITE; H->sensor_reftype[i] = MJH_OBJ_CAMERA; }
    if (t == MJH_SENS_INSIDESITE)
      MJH_REJECT(!frame_obj(m->sensor_objtype[i], &H->sensor_objtype[i]), "insidesite sensors attached to objects without a frame");
    if (t == MJH_SENS_CONTACT) {
      // (either side: nothing -- mjOBJ_UNKNOWN --, a site's volume, a geom, a body, a subtree)
      auto side = [&](int ot, int* out) -> bool { if (ot == mjOBJ_UNKNOWN) { *out = MJH_OBJ_NONE; return true; } return frame_obj(ot, out); };
      MJH_REJECT(!side(m->sensor_objtype[i], &H->sensor_objtype[i]) || !side(m->sensor_reftype[i], &H->sensor_reftype[i]),
                 "contact sensors matched against objects other than sites / geoms / bodies / subtrees");
    }
    if (t >= MJH_SENS_FRAMEPOS && t <= MJH_SENS_FRAMEANGACC) {
      MJH_REJECT(!frame_obj(m->sensor_objtype[i], &H->sensor_objtype[i]), "frame sensors attached to objects without a frame");
      if (m->sensor_refid[i] >= 0)
        MJH_REJECT(!frame_obj(m->sensor_reftype[i], &H->sensor_reftype[i]), "frame sensors with a reference object without a frame");
    }
  }
  // cameras that sensors use: the inputs of mj_camlight (engine_core_smooth.c:354-432) and cam_project's constants
  s.ncam_s = (use_cameras && m->ncam > 0) ? m->ncam : 0;
  H->cam_bodyid.clear(); H->cam_mode.clear(); H->cam_targetbodyid.clear(); H->cam_pos.clear(); H->cam_quat.clear();
  H->cam_mat0.clear(); H->cam_pos0.clear(); H->cam_poscom0.clear(); H->cam_proj.clear();
  if (s.ncam_s) {
    copy_arr(H->cam_bodyid, m->cam_bodyid, m->ncam);
    copy_arr(H->cam_mode, m->cam_mode, m->ncam);
    copy_arr(H->cam_targetbodyid, m->cam_targetbodyid, m->ncam);
    copy_arr(H->cam_pos, m->cam_pos, 3*(size_t)m->ncam);
    copy_arr(H->cam_quat, m->cam_quat, 4*(size_t)m->ncam);
    copy_arr(H->cam_mat0, m->cam_mat0, 9*(size_t)m->ncam);
    copy_arr(H->cam_pos0, m->cam_pos0, 3*(size_t)m->ncam);
    copy_arr(H->cam_poscom0, m->cam_poscom0, 3*(size_t)m->ncam);
    H->cam_proj.assign(4*(size_t)m->ncam, 0);
    for (int c = 0; c < m->ncam; c++) {
      const float* ss = m->cam_sensorsize + 2*c;
      const float* in = m->cam_intrinsic + 4*c;
      const int* res = m->cam_resolution + 2*c;
      mjtNum fx, fy;
      if (ss[0] && ss[1]) { fx = in[0] / ss[0] * res[0]; fy = in[1] / ss[1] * res[1]; }
      else { fx = fy = .5 / std::tan(m->cam_fovy[c] * mjPI / 360.) * res[1]; }
      H->cam_proj[4*c] = fx; H->cam_proj[4*c + 1] = fy; H->cam_proj[4*c + 2] = (mjtNum)res[0]; H->cam_proj[4*c + 3] = (mjtNum)res[1];
    }
  }
  for (int k = 0; k < 3; k++) o.magnetic[k] = m->opt.magnetic[k];
  s.nconH = (m->opt.cone != mjCONE_PYRAMIDAL && m->opt.solver != mjSOL_PGS) ? s.nconmax : 0;
  int nefc_bound = H->eq_rowadr[m->neq] + nfric + nlimit + rows_per_con*s.nconmax;
  // row capacity: the model's bound, cut back (not below 256 rows) until the per-environment constraint
  // arrays -- efc_J, efc_Y, the dense efc_AR under the dual solver, ~24 row vectors -- fit the budget
  // ($MJHIP_EFC_BYTES, default: the reference arena size clamped to 4..16 MiB; rows beyond the capacity raise mjWARN_CNSTRFULL like a full arena)
  // longest compressed contact row (stored before duplicates are merged): the dof chains of two bodies; a geom's chain and
  // the chains of the corners of a flex element; the chains of the corners of two elements of a flex that collides with itself
  auto csr_row_bound = [&]() {
    auto chain_len = [&](int b) {
      int cnt = 0;
      for (int w = 0; w < s.nvw; w++) cnt += __builtin_popcount((unsigned)H->body_dofanc[(size_t)m->body_weldid[b]*s.nvw + w]);
      return cnt;
    };
    int chainmax = 1;
    for (int b = 0; b < m->nbody; b++) chainmax = std::max(chainmax, chain_len(b));
    int bound = 2*chainmax;
    std::vector<int> sidemax(m->nflex, 0);
    for (int f = 0; f < m->nflex; f++) {
      int vchain = 0;
      for (int v = m->flex_vertadr[f]; v < m->flex_vertadr[f] + m->flex_vertnum[f]; v++) vchain = std::max(vchain, m->flex_vertbodyid[v] < 0 ? 0 : chain_len(m->flex_vertbodyid[v]));
      bound = std::max(bound, chainmax + (m->flex_dim[f] + 1)*vchain);
      sidemax[f] = (m->flex_dim[f] + 1)*vchain;
      if (m->flex_interp[f]) { const int o = m->flex_interp[f] + 1; bound = std::max(bound, chainmax + 3*o*o*o); sidemax[f] = 3*o*o*o; }
      for (int k = 0; k < s.nflexself; k++) if (H->flexself_flex[k] == f) bound = std::max(bound, 2*(m->flex_dim[f] + 1)*vchain);
    }
    for (int k = 0; k < s.nflexff; k++) bound = std::max(bound, sidemax[H->flexff_flex[2*k]] + sidemax[H->flexff_flex[2*k + 1]]);
    return bound;
  };
  {
    const bool dual = m->opt.solver == mjSOL_PGS || m->opt.noslip_iterations > 0;      // (mj_isDual: efc_AR exists)
    // (large models under CG keep the constraint Jacobian compressed -- mjh_csr.h -- and do not stream the dense rows)
    const bool ref_sparse0 = m->opt.jacobian == mjJAC_SPARSE || (m->opt.jacobian == mjJAC_AUTO && m->nv >= 60);
    // (equality rows on this path: flex edge / flex vertex constraints -- their rows are the model's flexedge_J rows -- and
    // connect, weld and joint equalities, whose rows are cut to their dof chains; tendon couplings are not)
    bool eq_ok = true, eq_flex = false, eq_flexvert = false;
    for (int i = 0; i < m->neq; i++) {
      if (m->eq_type[i] == mjEQ_FLEX) eq_flex = true;
      else if (m->eq_type[i] == mjEQ_FLEXVERT) eq_flexvert = true;
      else if (m->eq_type[i] != mjEQ_CONNECT && m->eq_type[i] != mjEQ_WELD && m->eq_type[i] != mjEQ_JOINT) eq_ok = false;   // (tendon couplings: not on this path)
    }
    // (tendons: none that could produce a constraint row -- limits, friction loss, couplings; a tendon that only carries an
    // actuator's transmission does not touch the rows)
    bool tendon_rows = false;
    for (int t = 0; t < m->ntendon; t++) if (m->tendon_limited[t] || m->tendon_frictionloss[t] != 0) tendon_rows = true;
    s.csr = (ref_sparse0 && m->nv > 128 && (m->opt.solver == mjSOL_CG || m->opt.solver == mjSOL_NEWTON) && eq_ok && !tendon_rows &&
             !(m->opt.disableflags & mjDSBL_ISLAND)) ? 1 : 0;
    if (s.csr) {
      // a row's merged dof chain is assembled in a 64-entry per-lane array (mjh_csr.h): models beyond that bound keep the
      // dense rows
      if (csr_row_bound() > MJH_CSR_CHAIN_MAX) s.csr = 0;
    }
    // Newton beyond 128 dofs runs on the explicit-index rows (mjh_newtonx.h): the factor as a packed lower triangle
    s.xn = (s.csr && m->opt.solver == mjSOL_NEWTON) ? 1 : 0;
    MJH_REJECT(m->opt.solver == mjSOL_NEWTON && m->nv > 128 && !s.csr,
               "the Newton solver with more than 128 degrees of freedom outside the explicit-index row path (sparse Jacobian, islands "
               "enabled, no tendon limits / friction / couplings, contacts up to condim 3)");
    MJH_REJECT(s.xn && m->nv > 2048, "the Newton solver with more than 2048 degrees of freedom");
    s.xncap = s.xn ? m->nv*(m->nv + 1)/2 : 0;
    s.xnw = s.xn ? (m->nv + 31)/32 : 0;
    s.xnell = (s.xn && m->opt.cone == mjCONE_ELLIPTIC) ? 1 : 0;
    // the dof chains csr_body_chain would walk through dof_parentid, as a table (independent loads on the device)
    H->body_chainadr.assign(m->nbody + 1, 0);
    H->body_chain.clear();
    if (s.csr) {
      for (int b = 0; b < m->nbody; b++) {
        H->body_chainadr[b] = (int)H->body_chain.size();
        const int w = m->body_weldid[b];
        if (m->body_dofnum[w]) for (int j = m->body_dofadr[w] + m->body_dofnum[w] - 1; j >= 0; j = m->dof_parentid[j]) H->body_chain.push_back(j);
      }
      H->body_chainadr[m->nbody] = (int)H->body_chain.size();
    }
    if (H->body_chain.empty()) H->body_chain.push_back(0);
    s.nchain = (int)H->body_chain.size();
    // flex edge constraints have no dense row: they need one of the compressed-row paths under a primal solver
    {
      const bool mask_path = ref_sparse0 && m->nv <= 128 && m->opt.solver != mjSOL_PGS;
      // (dense rows -- the dual solver, or a model the reference runs densely -- take the edge's row scattered into a
      // cleared row)
      const bool dense_rows = dual || !ref_sparse0;
      MJH_REJECT(eq_flexvert && !s.csr, "flex vertex equality constraints outside the explicit-index row path (CG, sparse Jacobian, more than 128 dofs)");
      MJH_REJECT(eq_flex && !s.csr && !mask_path && !dense_rows, "flex edge equality constraints outside the compressed-Jacobian paths (sparse Jacobian "
                                                  "under CG / Newton up to 128 dofs; beyond: CG / Newton, no tendon limit / friction / coupling rows)");
    }
    // default budget: what the reference's own arena (mjModel.narena, engine_memory.c:107-138) could hold, between 4 and
    // 16 MiB per environment (64 MiB for the compressed-row CG path) -- stacked_boxes.xml under PGS needs ~1000 rows
    const double MiB = 1024.0*1024.0;
    const double arena = std::min(std::max((double)m->narena, 4*MiB), 16*MiB);
    const double budget = caps.efc_bytes > 0 ? (double)caps.efc_bytes : (s.csr ? 64*MiB : arena);
    auto bytes = [&](int n) { return 8.0*n*(2.0*m->nv + (dual ? n : 0) + 24); };
    int n = std::max(1, std::min(nefc_bound, 4096));
    while (n > 256 && bytes(n) > budget) n = std::max(256, n*7/8);
    s.nefcmax = caps.nefcmax > 0 ? caps.nefcmax : n;
    s.nefcAR = dual ? s.nefcmax : 0;
  }
  // sparse constraint path (mj_isSparse, engine_core_util.c:32): the reference's sparse routines are followed
  // operation for operation (mjh_sparse.h); dof sets are 128-bit masks
  {
    const bool ref_sparse = m->opt.jacobian == mjJAC_SPARSE || (m->opt.jacobian == mjJAC_AUTO && m->nv >= 60);
    s.sparse = (ref_sparse && m->nv <= 128) ? 1 : 0;
    MJH_REJECT(s.sparse && s.nflexpair > 0 && m->opt.solver == mjSOL_PGS,
               "flex collisions under PGS in a model that takes the compressed-Jacobian path (sparse Jacobian with at most 128 dofs)");
    s.nJmax = 0; s.nLp = 0; s.nLpc = 0; s.nARw = 0;
    if (s.sparse) {
      // longest row pattern: two body chains (contacts, connect / weld), two tendons, a ball joint limit
      int chainmax = 1;
      for (int b = 0; b < m->nbody; b++) {
        int cnt = 0;
        for (int w = 0; w < s.nvw; w++) cnt += __builtin_popcount((unsigned)H->body_dofanc[(size_t)b*s.nvw + w]);
        chainmax = std::max(chainmax, cnt);
      }
      int tenmax = 0;
      for (int t = 0; t < m->ntendon; t++) tenmax = std::max(tenmax, (int)m->ten_J_rownnz[t]);
      const int rowmax = std::min((int)m->nv, std::max(std::max(2*chainmax, 2*tenmax), 3));
      s.nJmax = s.nefcmax*rowmax;
      s.nLp = m->opt.solver == mjSOL_NEWTON ? m->nv*(m->nv + 1)/2 : 0;
      s.nLpc = (m->opt.cone != mjCONE_PYRAMIDAL) ? s.nLp : 0;
      if (m->opt.solver == mjSOL_PGS) s.nARw = (s.nefcmax + 63)/64;
    }
    if (s.csr) {
      s.csr_rowmax = std::min((int)m->nv, csr_row_bound());
      // (capacity of the compressed rows: the longest row -- a contact's merged chains, or a flex vertex constraint's row
      // over the dofs of the vertex and its edge neighbours -- times the row capacity)
      int rowcap = s.csr_rowmax;
      for (int f = 0; f < m->nflex; f++)
        if (m->flex_edgeequality[f] == 2)
          for (int v = m->flex_vertadr[f]; v < m->flex_vertadr[f] + m->flex_vertnum[f]; v++) rowcap = std::max(rowcap, (int)m->flexvert_J_rownnz[2*v]);
      s.nJmax = s.nefcmax*rowcap;
    }
  }
  // PGS visitation orders for nefc = 1..64 (..128 when the capacity allows more than 64 rows) (engine_solver.c:241-265, :498-502): PCG32 with
  // state = 0, inc = 1 and one warm-up draw per solver call; every iteration Fisher-Yates-shuffles
  // the order array left by the previous iteration with j = next % (i+1), i = n-1 .. 1
  {
    s.pgs_iters = std::max(0, std::min((int)m->opt.iterations, 128));
    H->pgs_order_adr.assign(130, 0);
    H->pgs_order.clear();
    s.pgs_nmax = s.nefcmax > 64 ? 128 : 64;
    for (int n = 0; n <= s.pgs_nmax; n++) {
      H->pgs_order_adr[n] = (int)H->pgs_order.size();
      uint64_t state = 0, inc = 1;
      auto next = [&]() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + (inc | 1);
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (uint32_t)((xorshifted >> rot) | (xorshifted << ((-rot) & 31)));
      };
      next();
      std::vector<int> order(n);
      for (int i = 0; i < n; i++) order[i] = i;
      for (int it = 0; it < s.pgs_iters; it++) {
        for (int i = n - 1; i > 0; i--) {
          uint32_t j = next() % (uint32_t)(i + 1);
          std::swap(order[i], order[j]);
        }
        H->pgs_order.insert(H->pgs_order.end(), order.begin(), order.end());
      }
    }
    for (int n = s.pgs_nmax + 1; n < 130; n++) H->pgs_order_adr[n] = (int)H->pgs_order.size();
    if (H->pgs_order.empty()) H->pgs_order.push_back(0);
    s.npgsorder = (int)H->pgs_order.size();
  }

  // ---------------- flexes (mjh_flex.h) --------------------------------------------------------------------
  {
    const int nf = m->nflex;
    s.nflex = nf; s.nflexvert = m->nflexvert; s.nflexedge = m->nflexedge; s.nflexelem = m->nflexelem;
    s.nflexelemdata = m->nflexelemdata; s.nflexstiffness = m->nflexstiffness; s.nflexbending = m->nflexbending;
    s.nJfe = m->nJfe; s.nflexdof = nf ? m->nv : 0;
    bool anybend = false;
    for (int f = 0; f < nf; f++) if (m->flex_dim[f] == 2 && m->flex_bendingadr[f] >= 0) anybend = true;
    s.nflexbend = anybend ? m->nflexedge : 0;
    copy_arr(H->flex_dim, m->flex_dim, nf);
    copy_arr(H->flex_vertadr, m->flex_vertadr, nf);
    copy_arr(H->flex_vertnum, m->flex_vertnum, nf);
    copy_arr(H->flex_edgeadr, m->flex_edgeadr, nf);
    copy_arr(H->flex_edgenum, m->flex_edgenum, nf);
    copy_arr(H->flex_elemadr, m->flex_elemadr, nf);
    copy_arr(H->flex_elemnum, m->flex_elemnum, nf);
    copy_arr(H->flex_rigid, m->flex_rigid, nf);
    copy_arr(H->flex_centered, m->flex_centered, nf);
    copy_arr(H->flex_stiffnessadr, m->flex_stiffnessadr, nf);
    copy_arr(H->flex_bendingadr, m->flex_bendingadr, nf);
    copy_arr(H->flexvert_bodyid, m->flex_vertbodyid, m->nflexvert);
    copy_arr(H->flexedge_rigid, m->flexedge_rigid, m->nflexedge);
    copy_arr(H->flex_edgeequality, m->flex_edgeequality, nf);
    copy_arr(H->flexedge_invweight0, m->flexedge_invweight0, m->nflexedge);
    copy_arr(H->flexedge_J_rownnz, m->flexedge_J_rownnz, m->nflexedge);
    copy_arr(H->flexedge_J_rowadr, m->flexedge_J_rowadr, m->nflexedge);
    copy_arr(H->flexedge_J_colind, m->flexedge_J_colind, m->nJfe);
    {
      // flex vertex constraints (mjh_flex.h: flex_vert_rows)
      bool anyvert = false;
      for (int f = 0; f < nf; f++) if (m->flex_edgeequality[f] == 2) anyvert = true;
      s.nfv = anyvert ? m->nflexvert : 0;
      s.nJfv2 = anyvert ? 2*(int)m->nJfv : 0;
      s.nfvedge = anyvert ? 2*m->nflexedge : 0;
      s.nfvdx = anyvert ? m->nflexedge : 0;
      H->fv_rownnz.clear(); H->fv_rowadr.clear(); H->fv_colind.clear(); H->fv_edgeadr.clear(); H->fv_edgenum.clear(); H->fv_edge.clear();
      H->fv_metric.clear(); H->fv_dx.clear();
      if (anyvert) {
        copy_arr(H->fv_rownnz, m->flexvert_J_rownnz, 2*m->nflexvert);
        copy_arr(H->fv_rowadr, m->flexvert_J_rowadr, 2*m->nflexvert);
        copy_arr(H->fv_colind, m->flexvert_J_colind, s.nJfv2);
        copy_arr(H->fv_edgeadr, m->flex_vertedgeadr, m->nflexvert);
        copy_arr(H->fv_edgenum, m->flex_vertedgenum, m->nflexvert);
        copy_arr(H->fv_edge, m->flex_vertedge, 2*m->nflexedge);
        copy_arr(H->fv_metric, m->flex_vertmetric, 4*m->nflexvert);
        H->fv_dx.assign((size_t)3*m->nflexedge, 0);
        for (int f = 0; f < nf; f++) {
          if (m->flex_edgeequality[f] != 2) continue;
          const int vbase = m->flex_vertadr[f], ebase = m->flex_edgeadr[f];
          for (int ed = 0; ed < m->flex_edgenum[f]; ed++) {
            // rest edge vector, scaled: the vertices are stored in units of the half sizes (mj_flex :765-772)
            const int v1 = m->flex_edge[2*(ebase + ed)], v2 = m->flex_edge[2*(ebase + ed) + 1];
            mjtNum dx[3];
            for (int x = 0; x < 3; x++) dx[x] = m->flex_vert0[3*(vbase + v2) + x] - m->flex_vert0[3*(vbase + v1) + x];
            for (int x = 0; x < 3; x++) dx[x] *= 2*m->flex_size[3*f + x];
            for (int x = 0; x < 3; x++) H->fv_dx[3*(ebase + ed) + x] = dx[x];
          }
          for (int v = vbase; v < vbase + m->flex_vertnum[f]; v++) {
            MJH_REJECT(m->flexvert_J_rownnz[2*v] != m->flexvert_J_rownnz[2*v + 1], "internal: flex vertex constraint rows with different patterns");
            MJH_REJECT(m->flex_vertedgeadr[v] < 0 || m->flex_vertedgeadr[v] + m->flex_vertedgenum[v] > 2*m->nflexedge, "internal: flex vertex adjacency");
          }
        }
      }
    }
    copy_arr(H->flex_vert, m->flex_vert, 3*m->nflexvert);
    copy_arr(H->flexedge_length0, m->flexedge_length0, m->nflexedge);
    copy_arr(H->flex_stiffness, m->flex_stiffness, m->nflexstiffness);
    copy_arr(H->flex_bending, m->flex_bending, m->nflexbending);
    copy_arr(H->flex_damping, m->flex_damping, nf);
    copy_arr(H->flex_edgestiffness, m->flex_edgestiffness, nf);
    copy_arr(H->flex_edgedamping, m->flex_edgedamping, nf);
    copy_arr(H->flex_radius, m->flex_radius, nf);
    // interpolated flexes: nodes, cells, and per node the cells' contributions in scatter order (mj_flexPassiveInterp :131-180)
    {
      bool any = false;
      for (int f = 0; f < nf; f++) any = any || m->flex_interp[f] != 0;
      s.nflexnode = any ? m->nflexnode : 0;
      s.nflexivert = any ? m->nflexvert : 0;
      s.nconside = 0;
      H->flex_interp.assign(nf, 0);
      H->flex_cellnum.assign((size_t)3*nf, 0);
      H->flex_nodeadr.assign((size_t)nf + 1, 0);
      H->flexnode_bodyid.assign(s.nflexnode, 0);
      H->flexnode_flex.assign(s.nflexnode, 0);
      H->flexcell_flex.clear(); H->flexcell_kadr.clear(); H->flexcell_node.clear();
      H->flexnode_celladr.assign((size_t)s.nflexnode + 1, 0);
      H->flexnode_cell.clear();
      H->flex_node.assign((size_t)3*s.nflexnode, 0); H->flex_node0.assign((size_t)3*s.nflexnode, 0);
      H->flex_vert0.assign((size_t)3*s.nflexivert, 0);
      if (any) {
        copy_arr(H->flex_interp, m->flex_interp, nf);
        copy_arr(H->flex_cellnum, m->flex_cellnum, 3*nf);
        copy_arr(H->flexnode_bodyid, m->flex_nodebodyid, m->nflexnode);
        copy_arr(H->flex_node, m->flex_node, 3*m->nflexnode);
        copy_arr(H->flex_node0, m->flex_node0, 3*m->nflexnode);
        copy_arr(H->flex_vert0, m->flex_vert0, 3*m->nflexvert);
        std::vector<std::vector<int>> items(m->nflexnode);
        int maxnpc = 0;
        for (int f = 0; f < nf; f++) {
          H->flex_nodeadr[f] = m->flex_nodeadr[f];
          const int order = m->flex_interp[f];
          if (!order) continue;
          const int na = m->flex_nodeadr[f];
          for (int i = 0; i < m->flex_nodenum[f]; i++) {
            H->flexnode_flex[na + i] = f;
            if (m->flex_centered[f]) { H->flex_node[3*(na + i)] = 0; H->flex_node[3*(na + i) + 1] = 0; H->flex_node[3*(na + i) + 2] = 0; }
          }
          const int cx = m->flex_cellnum[3*f], cy = m->flex_cellnum[3*f + 1], cz = m->flex_cellnum[3*f + 2];
          const int npc = (order + 1)*(order + 1)*(order + 1);
          maxnpc = std::max(maxnpc, npc);
          MJH_REJECT(m->flex_nodenum[f] != (cx*order + 1)*(cy*order + 1)*(cz*order + 1), "an interpolated flex whose node grid does not match its cells");
          const int sadr = m->flex_stiffnessadr[f];
          const bool stretch = sadr >= 0 && m->flex_stiffness[sadr] != 0 && m->flex_edgeequality[f] != 3 && !m->flex_rigid[f] && m->flex_dim[f] != 1;
          if (!stretch) continue;
          const int ny_g = cy*order + 1, nz_g = cz*order + 1;
          for (int fe = 0; fe < cx*cy*cz; fe++) {
            const int cell = (int)H->flexcell_flex.size();
            const mjtNum* k_elem = m->flex_stiffness + sadr + (size_t)fe*3*npc*3*npc;
            H->flexcell_flex.push_back(f);
            H->flexcell_kadr.push_back(k_elem[0] == 0 ? -1 : (int)(k_elem - m->flex_stiffness));
            const int ci = fe/(cy*cz), cj = (fe/cz) % cy, ck = fe % cz;
            int local = 0;
            for (int li = 0; li <= order; li++)
              for (int lj = 0; lj <= order; lj++)
                for (int lk = 0; lk <= order; lk++) {
                  const int gidx = (ci*order + li)*ny_g*nz_g + (cj*order + lj)*nz_g + (ck*order + lk);
                  H->flexcell_node.push_back(na + gidx);
                  if (k_elem[0] != 0) items[na + gidx].push_back((cell << 5) | local);
                  local++;
                }
            H->flexcell_node.resize((size_t)27*(cell + 1), 0);
          }
        }
        for (int i = 0; i < m->nflexnode; i++) {
          H->flexnode_celladr[i] = (int)H->flexnode_cell.size();
          H->flexnode_cell.insert(H->flexnode_cell.end(), items[i].begin(), items[i].end());
        }
        H->flexnode_celladr[m->nflexnode] = (int)H->flexnode_cell.size();
        // a contact's body list: the nodes of a cell on an interpolated side, the corners of an element (or a geom) otherwise
        s.nconside = 2*std::max(4, maxnpc);
      }
      H->flex_nodeadr[nf] = s.nflexnode;
      s.nflexcell = (int)H->flexcell_flex.size();
      s.nflexnodecell = (int)H->flexnode_cell.size();
    }
    // (centered flexes and vertices at the body origin copy the body position: flag them once, mj_flex :567-572)
    for (int f = 0; f < nf; f++)
      for (int v = m->flex_vertadr[f]; v < m->flex_vertadr[f] + m->flex_vertnum[f]; v++)
        if (m->flex_centered[f]) { H->flex_vert[3*v] = 0; H->flex_vert[3*v+1] = 0; H->flex_vert[3*v+2] = 0; }
    H->flexvert_flex.assign(m->nflexvert, 0);
    H->flexedge_flex.assign(m->nflexedge, 0);
    H->flexedge_vert.assign((size_t)2*m->nflexedge, 0);
    H->flexedge_flap.assign((size_t)2*s.nflexbend, -1);
    H->flexelem_flex.assign(m->nflexelem, 0);
    H->flexelem_vert.assign((size_t)4*m->nflexelem, -1);
    H->flexelem_edge.assign((size_t)6*m->nflexelem, -1);
    H->flexedge_J_rowid.assign(m->nJfe, 0);
    std::vector<std::vector<int>> velem(m->nflexvert), vbend(m->nflexvert);
    for (int f = 0; f < nf; f++) {
      const int va = m->flex_vertadr[f], ea = m->flex_edgeadr[f], ta = m->flex_elemadr[f], dim = m->flex_dim[f];
      const int nedge = dim == 1 ? 1 : dim == 2 ? 3 : 6;
      for (int v = 0; v < m->flex_vertnum[f]; v++) H->flexvert_flex[va + v] = f;
      for (int e = 0; e < m->flex_edgenum[f]; e++) {
        H->flexedge_flex[ea + e] = f;
        H->flexedge_vert[2*(ea + e)] = va + m->flex_edge[2*(ea + e)];
        H->flexedge_vert[2*(ea + e) + 1] = va + m->flex_edge[2*(ea + e) + 1];
        if (s.nflexbend && dim == 2 && m->flex_bendingadr[f] >= 0) {
          const int* flap = m->flex_edgeflap + 2*(ea + e);
          H->flexedge_flap[2*(ea + e)] = flap[0] < 0 ? -1 : va + flap[0];
          H->flexedge_flap[2*(ea + e) + 1] = flap[1] < 0 ? -1 : va + flap[1];
          if (flap[1] != -1) {    // (boundary edges carry no bending force, mj_flexPassiveBend :479)
            const int vv[4] = {H->flexedge_vert[2*(ea + e)], H->flexedge_vert[2*(ea + e) + 1], va + flap[0], va + flap[1]};
            for (int i = 0; i < 4; i++) vbend[vv[i]].push_back(((ea + e) << 2) | i);
          }
        }
      }
      for (int t = 0; t < m->flex_elemnum[f]; t++) {
        const int* ed = m->flex_elem + m->flex_elemdataadr[f] + t*(dim + 1);
        for (int i = 0; i <= dim; i++) {
          H->flexelem_vert[4*(ta + t) + i] = va + ed[i];
          velem[va + ed[i]].push_back(((ta + t) << 2) | i);
        }
        if (dim >= 2)
          for (int k = 0; k < nedge; k++) H->flexelem_edge[6*(ta + t) + k] = ea + m->flex_elemedge[m->flex_elemedgeadr[f] + t*nedge + k];
        H->flexelem_flex[ta + t] = f;
      }
    }
    H->flexvert_elemadr.assign((size_t)m->nflexvert + 1, 0);
    H->flexvert_bendadr.assign((size_t)m->nflexvert + 1, 0);
    H->flexvert_elem.clear(); H->flexvert_bend.clear();
    for (int v = 0; v < m->nflexvert; v++) {
      H->flexvert_elemadr[v] = (int)H->flexvert_elem.size();
      H->flexvert_bendadr[v] = (int)H->flexvert_bend.size();
      H->flexvert_elem.insert(H->flexvert_elem.end(), velem[v].begin(), velem[v].end());
      H->flexvert_bend.insert(H->flexvert_bend.end(), vbend[v].begin(), vbend[v].end());
    }
    H->flexvert_elemadr[m->nflexvert] = (int)H->flexvert_elem.size();
    H->flexvert_bendadr[m->nflexvert] = (int)H->flexvert_bend.size();
    H->flexvert_elem.resize(s.nflexelemdata, 0);
    // vertices on articulated bodies: per dof, the vertices whose chain holds it (mj_flexPassiveStretch :632-648 applies the
    // stretch force of such a vertex through mj_applyFT, i.e. to every dof of its body's chain)
    {
      s.flex_sliders = 1;
      std::vector<std::vector<int>> dv(m->nv);
      for (int v = 0; v < m->nflexvert; v++) {
        const int b = m->flex_vertbodyid[v];
        if (b < 0 || m->body_simple[b] == 2) continue;    // (vertices of an interpolated flex have no body)
        const int w = m->body_weldid[b];
        if (m->body_dofnum[w] == 0) continue;
        s.flex_sliders = 0;
        for (int j = m->body_dofadr[w] + m->body_dofnum[w] - 1; j >= 0; j = m->dof_parentid[j]) dv[j].push_back(v);
      }
      H->flexdof_vadr.assign((size_t)s.nflexdof + 1, 0);
      H->flexdof_vert.clear();
      for (int j = 0; j < s.nflexdof; j++) {
        H->flexdof_vadr[j] = (int)H->flexdof_vert.size();
        H->flexdof_vert.insert(H->flexdof_vert.end(), dv[j].begin(), dv[j].end());
      }
      if (s.nflexdof) H->flexdof_vadr[s.nflexdof] = (int)H->flexdof_vert.size();
      s.nflexdofv = (int)H->flexdof_vert.size();
    }

    // ---------------- implicit effective metric (mjh_effmetric.h) ----------------
    // mj_flexCG (engine_forward.c:1640): CG + implicit / implicitfast + pyramidal cones + a flex with implicit stiffness: the
    // solve runs in the metric M + K, K = (h^2 + h damping)(K_bend + K_stretch) assembled per step into a dof-level CSR
    // (mjd_flexStiff_assemble, engine_derivative.c:1810).  Its STRUCTURE is static for standard flexes: vertex slots (bodies
    // with three dofs), neighbours through bending flaps and element cliques, sorted by dof address -- built here together
    // with, per 3 x 3 block, the list of its contributions in the reference's order of accumulation.
    s.efm = 0;
    H->efm_rownnz.clear(); H->efm_rowadr.clear(); H->efm_colind.clear(); H->efm_slotvert.clear(); H->efm_slotdiag.clear();
    H->e0_dof.clear(); H->e0_rownnz.clear(); H->e0_rowadr.clear(); H->e0_colind.clear(); H->e0_L.clear(); H->e0_cov.clear();
    H->e0_cscind.clear(); H->e0_cscrow.clear(); H->e0_l1row.clear(); H->e0_l2row.clear();
    H->efm_vertslot.clear(); H->efmblk_slot.clear(); H->efmblk_pos.clear(); H->efmblk_cadr.assign(1, 0); H->efmblk_c.clear(); H->efmblk_cij.clear();
    {
      bool implicit_stiff = false, has_stretch = false;
      for (int f = 0; f < nf; f++) {
        if (m->flex_rigid[f]) continue;
        if (m->flex_interp[f] != 0) continue;      // (explicit integrators only, checked with the other gates)
        if (m->flex_dim[f] == 2 && m->flex_bendingadr[f] >= 0) implicit_stiff = true;
        if (m->flex_dim[f] >= 2 && m->flex_stiffnessadr[f] >= 0 && m->flex_stiffness[m->flex_stiffnessadr[f]] != 0) { implicit_stiff = true; has_stretch = true; }
      }
      const bool flexcg = m->opt.solver == mjSOL_CG && (m->opt.integrator == mjINT_IMPLICIT || m->opt.integrator == mjINT_IMPLICITFAST) &&
                          m->opt.cone != mjCONE_ELLIPTIC && implicit_stiff;
      if (flexcg) {
        // (bending alone: the reference keeps the stencil operator and a constant sparse factor from mj_setConst -- not built;
        // vertices with one or two dofs of their own: mjd_flexStretch_mul reads three consecutive dofs at their address)
        MJH_REJECT(m->nv <= 128, "the implicit effective metric (mj_flexCG) in models of at most 128 degrees of freedom");
        for (int v = 0; v < m->nflexvert; v++) {
          const int b = m->flex_vertbodyid[v], f = 0;
          (void)f;
          MJH_REJECT(m->body_dofnum[b] != 0 && !(m->body_dofnum[b] == 3 && m->body_simple[b] == 2),
                     "the implicit effective metric with flex vertices that are not three sliders of their own (or pinned)");
        }
        s.efm = has_stretch ? 1 : 2;
        if (!has_stretch) {
          // bending alone: no per-step CSR (mjd_effBuild leaves nefmK = 0); the matrix-vector product is the stencil operator
          // (mjd_flexBend_mul) and the preconditioner the CONSTANT factor of mj_setConst on the dofs it covers
          // (effBlockApply's flg_bend branch, engine_derivative.c:3262-3301)
          const int nbd = (int)m->nefm0dof, nL = (int)m->nefm0L;
          MJH_REJECT(nbd <= 0 || nL < nbd, "internal: bending-only effective metric without the constant factor of mj_setConst");
          s.ne0 = nbd; s.ne0L = nL; s.ne0off = nL - nbd;
          H->e0_dof.assign(m->efm0_dofid, m->efm0_dofid + nbd);
          H->e0_rownnz.assign(m->efm0_L_rownnz, m->efm0_L_rownnz + nbd);
          H->e0_rowadr.assign(m->efm0_L_rowadr, m->efm0_L_rowadr + nbd);
          H->e0_colind.assign(m->efm0_L_colind, m->efm0_L_colind + nL);
          H->e0_L.assign(m->efm0_L, m->efm0_L + nL);
          H->e0_cov.assign(m->nv, 0);
          for (int i = 0; i < nbd; i++) { MJH_REJECT(H->e0_dof[i] < 0 || H->e0_dof[i] >= m->nv, "internal: constant metric factor (dof id)"); H->e0_cov[H->e0_dof[i]] = 1; }
          // columns: the rows i > c that hold an entry in column c, descending (the order in which the first sweep of
          // mju_cholSolveSparse subtracts from x[c]); levels of the two sweeps
          std::vector<std::vector<std::pair<int, int>>> col(nbd);
          for (int i = 0; i < nbd; i++) {
            const int adr = H->e0_rowadr[i], nnz = H->e0_rownnz[i];
            MJH_REJECT(nnz < 1 || adr < 0 || adr + nnz > nL || H->e0_colind[adr + nnz - 1] != i, "internal: constant metric factor (row layout)");
            for (int j = 0; j < nnz - 1; j++) {
              const int c = H->e0_colind[adr + j];
              MJH_REJECT(c < 0 || c >= i, "internal: constant metric factor (column index)");
              col[c].push_back({i, adr + j});
            }
          }
          H->e0_cscadr.assign(nbd + 1, 0);
          H->e0_cscind.clear(); H->e0_cscrow.clear();
          std::vector<int> lev1(nbd, 0), lev2(nbd, 0);
          for (int c = nbd - 1; c >= 0; c--) {
            int l = 0;
            for (auto& it : col[c]) l = std::max(l, lev1[it.first] + 1);
            lev1[c] = l;
          }
          for (int c = 0; c < nbd; c++) {
            H->e0_cscadr[c] = (int)H->e0_cscind.size();
            for (int q = (int)col[c].size() - 1; q >= 0; q--) { H->e0_cscrow.push_back(col[c][q].first); H->e0_cscind.push_back(col[c][q].second); }
          }
          H->e0_cscadr[nbd] = (int)H->e0_cscind.size();
          for (int i = 0; i < nbd; i++) {
            int l = 0;
            const int adr = H->e0_rowadr[i], nnz = H->e0_rownnz[i];
            for (int j = 0; j < nnz - 1; j++) l = std::max(l, lev2[H->e0_colind[adr + j]] + 1);
            lev2[i] = l;
          }
          auto schedule = [&](const std::vector<int>& lev, std::vector<int>& ladr, std::vector<int>& lrow) {
            int nl = 0;
            for (int i = 0; i < nbd; i++) nl = std::max(nl, lev[i] + 1);
            ladr.assign(nl + 1, 0);
            for (int i = 0; i < nbd; i++) ladr[lev[i] + 1]++;
            for (int l = 0; l < nl; l++) ladr[l + 1] += ladr[l];
            lrow.assign(nbd, 0);
            std::vector<int> fill(ladr.begin(), ladr.end() - 1);
            for (int i = 0; i < nbd; i++) lrow[fill[lev[i]]++] = i;
            return nl;
          };
          s.ne0lev1 = schedule(lev1, H->e0_l1adr, H->e0_l1row);
          s.ne0lev2 = schedule(lev2, H->e0_l2adr, H->e0_l2row);
          H->efm_rownnz.assign(m->nv, 0);
          H->efm_rowadr.assign(m->nv, 0);
          s.nefmrow = m->nv; s.nefmK = 0; s.nefmslot = 0; s.nefmvert = 0; s.nefmblk = 0; s.nefmcon = 0;
          H->efmblk_cadr.assign(1, 0);
        }
        if (has_stretch) {
        std::vector<int> vslot(m->nflexvert, -1), vdof;
        auto active = [&](int f) {
          if (m->flex_interp[f] || m->flex_rigid[f] || m->flex_dim[f] < 2) return false;
          const bool bend = m->flex_bendingadr[f] >= 0;
          const bool stretch = m->flex_stiffnessadr[f] >= 0 && m->flex_stiffness[m->flex_stiffnessadr[f]] != 0;
          return bend || stretch;
        };
        for (int f = 0; f < nf; f++) {
          if (!active(f)) continue;
          for (int lv = 0; lv < m->flex_vertnum[f]; lv++) {
            const int gv = m->flex_vertadr[f] + lv;
            if (m->body_dofnum[m->flex_vertbodyid[gv]] == 3) { vslot[gv] = (int)vdof.size(); vdof.push_back(m->body_dofadr[m->flex_vertbodyid[gv]]); H->efm_slotvert.push_back(gv); }
          }
        }
        const int nvert = (int)vdof.size();
        // candidate neighbours (with duplicates), in the reference's order; contributions keyed by (slot, neighbour slot)
        struct Con { int kind_id, cij; };
        std::vector<std::vector<int>> cand(nvert);
        std::map<std::pair<int, int>, std::vector<Con>> cons;
        for (int f = 0; f < nf; f++) {
          if (!active(f)) continue;
          const int dim = m->flex_dim[f], nvrt = dim + 1, va = m->flex_vertadr[f];
          if (m->flex_bendingadr[f] >= 0 && dim == 2) {
            for (int e = 0; e < m->flex_edgenum[f]; e++) {
              const int* edge = m->flex_edge + 2*(e + m->flex_edgeadr[f]);
              const int* flap = m->flex_edgeflap + 2*(e + m->flex_edgeadr[f]);
              if (flap[1] == -1) continue;
              const int v[4] = {edge[0], edge[1], flap[0], flap[1]};
              for (int i = 0; i < 4; i++) {
                const int si = vslot[va + v[i]];
                if (si < 0) continue;
                for (int j = 0; j < 4; j++) {
                  const int sj = vslot[va + v[j]];
                  if (sj < 0) continue;
                  cand[si].push_back(sj);
                  cons[{si, sj}].push_back({0 | ((m->flex_edgeadr[f] + e) << 1), (i << 2) | j});
                }
              }
            }
          }
          if (m->flex_stiffnessadr[f] >= 0 && m->flex_stiffness[m->flex_stiffnessadr[f]] != 0) {
            const int* elem = m->flex_elem + m->flex_elemdataadr[f];
            for (int t = 0; t < m->flex_elemnum[f]; t++) {
              const int* vert = elem + (dim + 1)*t;
              for (int i = 0; i < nvrt; i++) {
                const int si = vslot[va + vert[i]];
                if (si < 0) continue;
                for (int j = 0; j < nvrt; j++) {
                  const int sj = vslot[va + vert[j]];
                  if (sj < 0) continue;
                  cand[si].push_back(sj);
                  cons[{si, sj}].push_back({1 | ((m->flex_elemadr[f] + t) << 1), (i << 2) | j});
                }
              }
            }
          }
        }
        // NOTE: the reference accumulates bending of flex f, then stretch of flex f, flex by flex, into zeroed values: a
        // block belongs to one flex, so its list above is already in that order
        std::vector<int> nadr(nvert + 1, 0), neigh;
        for (int sl = 0; sl < nvert; sl++) {
          std::vector<int>& c = cand[sl];
          std::stable_sort(c.begin(), c.end(), [&](int a, int b) { return vdof[a] < vdof[b]; });     // (insertion sort by dof address: stable)
          nadr[sl] = (int)neigh.size();
          for (size_t i = 0; i < c.size(); i++) if ((int)neigh.size() == nadr[sl] || neigh.back() != c[i]) neigh.push_back(c[i]);
        }
        nadr[nvert] = (int)neigh.size();
        H->efm_rownnz.assign(m->nv, 0);
        H->efm_rowadr.assign(m->nv, 0);
        for (int sl = 0; sl < nvert; sl++) for (int k = 0; k < 3; k++) H->efm_rownnz[vdof[sl] + k] = 3*(nadr[sl + 1] - nadr[sl]);
        for (int i = 1; i < m->nv; i++) H->efm_rowadr[i] = H->efm_rowadr[i - 1] + H->efm_rownnz[i - 1];
        int nnz = 0;
        for (int i = 0; i < m->nv; i++) nnz += H->efm_rownnz[i];
        H->efm_colind.assign(nnz, 0);
        H->efm_slotdiag.assign(nvert, -1);
        for (int sl = 0; sl < nvert; sl++) {
          for (int k = 0; k < 3; k++) {
            int adr = H->efm_rowadr[vdof[sl] + k];
            for (int j = nadr[sl]; j < nadr[sl + 1]; j++) for (int c = 0; c < 3; c++) H->efm_colind[adr++] = vdof[neigh[j]] + c;
          }
          for (int j = nadr[sl]; j < nadr[sl + 1]; j++) {
            const int pos = j - nadr[sl];
            if (neigh[j] == sl) H->efm_slotdiag[sl] = pos;
            H->efmblk_slot.push_back(sl);
            H->efmblk_pos.push_back(pos);
            for (const Con& c : cons[{sl, neigh[j]}]) { H->efmblk_c.push_back(c.kind_id); H->efmblk_cij.push_back(c.cij); }
            H->efmblk_cadr.push_back((int)H->efmblk_c.size());
          }
          MJH_REJECT(H->efm_slotdiag[sl] < 0, "internal: effective-metric diagonal block");
        }
        // the covered dofs come in the triples of the slots, ascending (effBlocks walks the covered rows)
        for (int sl = 1; sl < nvert; sl++) MJH_REJECT(vdof[sl] <= vdof[sl - 1], "internal: effective-metric slots not in dof order");
        H->efm_vertslot = vslot;
        s.nefmrow = m->nv; s.nefmK = nnz; s.nefmslot = nvert; s.nefmvert = m->nflexvert;
        s.nefmblk = (int)H->efmblk_slot.size(); s.nefmcon = (int)H->efmblk_c.size();
        H->e0_cov.assign(m->nv, 0);
        }
      }
      if (!s.efm) { s.nefmrow = s.nefmK = s.nefmslot = s.nefmvert = s.nefmblk = s.nefmcon = 0; H->efmblk_cadr.assign(1, 0); }
      if (s.efm != 2) {
        s.ne0 = s.ne0L = s.ne0off = s.ne0lev1 = s.ne0lev2 = 0;
        H->e0_cscadr.assign(1, 0); H->e0_l1adr.assign(1, 0); H->e0_l2adr.assign(1, 0);
      }
    }
    H->flexvert_bend.resize((size_t)4*s.nflexbend, 0);
    // flexedge_J by column, entries in ascending edge order (the order mj_springdamper adds edge forces to a dof, :770-787)
    {
      std::vector<std::vector<int>> col(nf ? m->nv : 0);
      for (int e = 0; e < m->nflexedge; e++)
        for (int j = m->flexedge_J_rowadr[e]; j < m->flexedge_J_rowadr[e] + m->flexedge_J_rownnz[e]; j++) {
          H->flexedge_J_rowid[j] = e;
          col[m->flexedge_J_colind[j]].push_back(j);
        }
      H->flexJ_cscadr.assign((size_t)s.nflexdof + 1, 0);
      H->flexJ_cscind.clear();
      for (int i = 0; i < (nf ? m->nv : 0); i++) {
        H->flexJ_cscadr[i] = (int)H->flexJ_cscind.size();
        H->flexJ_cscind.insert(H->flexJ_cscind.end(), col[i].begin(), col[i].end());
      }
      if (nf) H->flexJ_cscadr[m->nv] = (int)H->flexJ_cscind.size();
      H->flexJ_cscind.resize(m->nJfe, 0);
      H->flexJ_cscedge.assign(m->nJfe, 0);
      for (int a = 0; a < m->nJfe; a++) H->flexJ_cscedge[a] = H->flexedge_J_rowid[H->flexJ_cscind[a]];
      H->flexedge_k.assign(m->nflexedge, 0);
      H->flexedge_d.assign(m->nflexedge, 0);
      for (int ed = 0; ed < m->nflexedge; ed++) {
        const int f = H->flexedge_flex[ed];
        if (m->flex_rigid[f] || m->flexedge_rigid[ed]) continue;
        H->flexedge_k[ed] = m->flex_edgestiffness[f];
        H->flexedge_d[ed] = m->flex_edgedamping[f];
      }
    }
  }

  // ---------------- features this model needs from a kernel variant (MJH_FT_*, mjh_types.h) ------------
  // every `MJH_HAS(x) && condition` of the stage sources has its condition mirrored here
  {
    int ft = 0;
    // (the sparse constraint path lives in the generic kernel, whatever the solver)
    if (m->opt.solver != mjSOL_PGS || s.sparse || m->opt.noslip_iterations > 0) ft |= MJH_FT_PRIMAL;      // (noslip: carried by the generic kernels only)
    if (m->opt.cone != mjCONE_PYRAMIDAL) ft |= MJH_FT_ELLIPTIC;
    if (m->neq > 0) ft |= MJH_FT_EQUALITY;
    if (m->opt.integrator == mjINT_RK4) ft |= MJH_FT_RK4;
    if (m->opt.integrator == mjINT_IMPLICITFAST || m->opt.integrator == mjINT_IMPLICIT) ft |= MJH_FT_IMPLICIT;
    if (m->nsensor > 0) ft |= MJH_FT_SENSOR;
    for (int p = 0; p < s.npair; p++) {
      const int f = H->pair_func[p];
      if (f != MJH_COL_PLANE_SPHERE && f != MJH_COL_PLANE_CAPSULE && f != MJH_COL_SPHERE_SPHERE &&
          f != MJH_COL_SPHERE_CAPSULE && f != MJH_COL_CAPSULE_CAPSULE) ft |= MJH_FT_COLCONVEX;
      if (H->pair_dim[p] > 3) ft |= MJH_FT_CONDIM46;
    }
    if (m->na > 0) ft |= MJH_FT_ACT;
    for (int i = 0; i < m->ntendon; i++) if (m->wrap_type[m->tendon_adr[i]] != mjWRAP_JOINT) ft |= MJH_FT_TENDONSPATIAL;
    for (int i = 0; i < m->nu; i++) {
      const int tt = m->actuator_trntype[i];
      if (tt != mjTRN_JOINT && tt != mjTRN_JOINTINPARENT) ft |= MJH_FT_TRNMISC;
      else {
        const int jt = m->jnt_type[m->actuator_trnid[2*i]];
        if (jt == mjJNT_BALL || jt == mjJNT_FREE) ft |= MJH_FT_TRNMISC;
      }
      if (m->actuator_gaintype[i] != mjGAIN_FIXED || m->actuator_biastype[i] != mjBIAS_NONE ||
          m->actuator_forcelimited[i] || H->actuator_disabled[i]) ft |= MJH_FT_GAINBIAS;
    }
    for (int i = 0; i < m->njnt; i++) if (m->jnt_actfrclimited[i]) ft |= MJH_FT_GAINBIAS;
    if (o.has_ten_actfrc) ft |= MJH_FT_GAINBIAS;
    if (o.has_gravcomp || o.has_fluid || o.has_surfacevel || o.has_ten_armature || o.has_adhesion) ft |= MJH_FT_PASSIVEMISC;
    if (m->nmocap > 0) ft |= MJH_FT_MOCAP;
    if (m->ntree > 1) ft |= MJH_FT_ISLANDS;
    if (m->nflex > 0) ft |= MJH_FT_FLEX;
    s.features = ft;
  }
  return true;
}

// verify that every vector has exactly the size its X-macro declares
static bool check_sizes(const HostModel& H, std::string* err) {
  const DSizes& s = H.s;
#define X(name, cnt) if ((long long)H.name.size() != (long long)(cnt)) { *err = std::string("mjhip internal: size of ") + #name; return false; }
  MJH_MODEL_INT_FIELDS(X)
  MJH_MODEL_REAL_FIELDS(X)
#undef X
  return true;
}

}  // namespace mjhb
