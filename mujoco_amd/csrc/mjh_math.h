// Small fixed-size fp64 math used by the mjhip kernels.
//
// Operation ORDER matters here: the parity target is the reference engine compiled without FMA
// contraction, so every expression below is written with the same association as the reference
// helper it stands in for (cited per function; files relative to /root/reference/src/engine).
// The kernels are compiled with -ffp-contract=off.
#pragma once

#include "mjh_spmd.h"

#define MJH_MINVAL 1E-15   // mjMINVAL, include/mujoco/mjmodel.h:24
#define MJH_MAXVAL 1E+10   // mjMAXVAL, mjmodel.h:25
#define MJH_PI 3.14159265358979323846

typedef double real;

// strided view of a vector: element i lives at p[i*s].  Every per-environment array of the batch is
// handed to the stage code as one of these, so the same stage source runs on environment-major
// data (s = 1; also LDS-resident slices) and on SoA-across-environments data (s = nenvpad).
template <class T>
struct SP {
  T* p;
  int s;
  template <class I> MJH_MEM T& operator[](I i) const { return p[(long long)i * s]; }
  template <class I> MJH_MEM SP operator+(I k) const { return SP{p + (long long)k * s, s}; }
  MJH_MEM operator SP<const T>() const { return SP<const T>{p, s}; }
};
typedef SP<real> rptr;
typedef SP<const real> crptr;
typedef SP<int> iptr;
typedef SP<const int> ciptr;


// ---- 128-bit sets of dofs (sparse constraint path: row patterns of J, H and the Cholesky factor; nv <= 128)
struct M128 { uint64_t lo, hi; };
MJH_DEV M128 m128_zero() { return M128{0, 0}; }
MJH_DEV M128 m128_bit(int i) { return i < 64 ? M128{1ull << i, 0} : M128{0, 1ull << (i - 64)}; }
MJH_DEV int m128_test(M128 m, int i) { return (int)(((i < 64 ? m.lo : m.hi) >> (i & 63)) & 1); }
MJH_DEV M128 m128_or(M128 a, M128 b) { return M128{a.lo | b.lo, a.hi | b.hi}; }
MJH_DEV M128 m128_and(M128 a, M128 b) { return M128{a.lo & b.lo, a.hi & b.hi}; }
MJH_DEV M128 m128_xor(M128 a, M128 b) { return M128{a.lo ^ b.lo, a.hi ^ b.hi}; }
// bits [0, i)
MJH_DEV M128 m128_below(int i) {
  if (i <= 0) return M128{0, 0};
  if (i < 64) return M128{(1ull << i) - 1, 0};
  if (i == 64) return M128{~0ull, 0};
  if (i < 128) return M128{~0ull, (1ull << (i - 64)) - 1};
  return M128{~0ull, ~0ull};
}
MJH_DEV int m128_any(M128 m) { return (m.lo | m.hi) != 0; }
MJH_DEV int m128_count(M128 m) { return __builtin_popcountll(m.lo) + __builtin_popcountll(m.hi); }
// number of members below i = position of member i in the ascending list
MJH_DEV int m128_rank(M128 m, int i) { return m128_count(m128_and(m, m128_below(i))); }
MJH_DEV int m128_lowest(M128 m) { return m.lo ? __builtin_ctzll(m.lo) : 64 + __builtin_ctzll(m.hi); }
MJH_DEV int m128_highest(M128 m) { return m.hi ? 127 - __builtin_clzll(m.hi) : 63 - __builtin_clzll(m.lo); }
MJH_DEV M128 m128_drop_lowest(M128 m) { return m.lo ? M128{m.lo & (m.lo - 1), m.hi} : M128{0, m.hi & (m.hi - 1)}; }
// four 32-bit words of an int array <-> mask
template <class P0>
MJH_DEV M128 m128_ld(P0 w) {
  return M128{((uint64_t)(unsigned)w[1] << 32) | (unsigned)w[0], ((uint64_t)(unsigned)w[3] << 32) | (unsigned)w[2]};
}
template <class P0>
MJH_DEV void m128_st(P0 w, M128 m) {
  w[0] = (int)(unsigned)m.lo; w[1] = (int)(unsigned)(m.lo >> 32); w[2] = (int)(unsigned)m.hi; w[3] = (int)(unsigned)(m.hi >> 32);
}
// nw (<= 4) 32-bit words, the rest zero
template <class P0>
MJH_DEV M128 m128_ldw(P0 w, int nw) {
  unsigned a[4] = {0, 0, 0, 0};
  for (int k = 0; k < nw && k < 4; k++) a[k] = (unsigned)w[k];
  return M128{((uint64_t)a[1] << 32) | a[0], ((uint64_t)a[3] << 32) | a[2]};
}
MJH_DEV M128 wv_uniform_m128(M128 m) { return M128{wv_uniform_u64(m.lo), wv_uniform_u64(m.hi)}; }
MJH_DEV M128 wv_bcast_m128(M128 m, int src) { M128 r; wv_bcast_u128(m.lo, m.hi, src, &r.lo, &r.hi); return r; }
// position of the calling lane's dof (slot 0: dof = lane, slot 1: dof = lane + 64) among the members of m
MJH_DEV int m128_rank_lane0(M128 m) { return wv_rank_lt(m.lo); }
MJH_DEV int m128_rank_lane1(M128 m) { return __builtin_popcountll(m.lo) + wv_rank_lt(m.hi); }

template <class P0>
MJH_DEV void v3_zero(P0 r) { r[0] = 0; r[1] = 0; r[2] = 0; }
template <class P0, class P1>
MJH_DEV void v3_copy(P0 r, P1 a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
template <class P0, class P1>
MJH_DEV void v3_scl(P0 r, P1 a, real s) { r[0] = a[0]*s; r[1] = a[1]*s; r[2] = a[2]*s; }
template <class P0, class P1, class P2>
MJH_DEV void v3_add(P0 r, P1 a, P2 b) { r[0] = a[0]+b[0]; r[1] = a[1]+b[1]; r[2] = a[2]+b[2]; }
template <class P0, class P1, class P2>
MJH_DEV void v3_sub(P0 r, P1 a, P2 b) { r[0] = a[0]-b[0]; r[1] = a[1]-b[1]; r[2] = a[2]-b[2]; }
template <class P0, class P1>
MJH_DEV void v3_addto(P0 r, P1 a) { r[0] += a[0]; r[1] += a[1]; r[2] += a[2]; }
template <class P0, class P1>
MJH_DEV void v3_subfrom(P0 r, P1 a) { r[0] -= a[0]; r[1] -= a[1]; r[2] -= a[2]; }
// r += a*s                                     (mji_addToScl3, engine_inline.h:108)
template <class P0, class P1>
MJH_DEV void v3_addtoscl(P0 r, P1 a, real s) { r[0] += a[0]*s; r[1] += a[1]*s; r[2] += a[2]*s; }
// a.b, left-to-right                           (mju_dot3, engine_util_blas.c:140)
template <class P0, class P1>
MJH_DEV real v3_dot(P0 a, P1 b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
template <class P0>
MJH_DEV real v3_norm(P0 a) { return sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }
// normalize, return previous length            (mju_normalize3, engine_util_blas.c:115)
template <class P0>
MJH_DEV real v3_normalize(P0 v) {
  real n = sqrt(v[0]*v[0] + v[1]*v[1] + v[2]*v[2]);
  if (n < MJH_MINVAL) {
    v[0] = 1; v[1] = 0; v[2] = 0;
  } else {
    real inv = 1/n;
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
  }
  return n;
}
// cross product                                (mji_cross, engine_inline.h:418)
template <class P0, class P1, class P2>
MJH_DEV void v3_cross(P0 r, P1 a, P2 b) {
  r[0] = a[1]*b[2] - a[2]*b[1];
  r[1] = a[2]*b[0] - a[0]*b[2];
  r[2] = a[0]*b[1] - a[1]*b[0];
}
// r = M v, row-major 3x3                       (mji_mulMatVec3, engine_inline.h:147)
template <class P0, class P1, class P2>
MJH_DEV void m3_mulvec(P0 r, P1 m, P2 v) {
  r[0] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  r[1] = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  r[2] = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
}
// r = M' v                                     (mji_mulMatTVec3, engine_inline.h:157)
template <class P0, class P1, class P2>
MJH_DEV void m3_multvec(P0 r, P1 m, P2 v) {
  r[0] = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  r[1] = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  r[2] = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
}

// ---- sine and cosine ---------------------------------------------------------------------------
// The device libm's sin / cos carry a Payne-Hanek path for huge arguments whose tables and spills are
// paid -- in scratch memory and registers -- by every kernel that calls them, although joint angles
// are O(1).  mjh_sincos replaces them on the device: Cody-Waite reduction by pi/2 in three 53-bit pieces
// with error-free products (explicit fma), good to |x| ~ 1e10 (mjMAXVAL, beyond which mj_checkPos has
// already reset the state), then the classic minimax kernels on [-pi/4, pi/4] (fdlibm k_sin.c / k_cos.c
// coefficients) fed with the reduced argument as head + tail.  Error < 1 ulp, like libm's; no tables, no
// scratch.  The explicit fma calls are correctly rounded on host and device alike, so the routine is
// bit-reproducible across both (tests/hostsim exports it: mjh_test_sincos).
MJH_DEV void mjh_sincos(real x, real* sn, real* cs) {
  const real P1 = 0x1.921fb54442d18p+0, P2 = 0x1.1a62633145c07p-54, P3 = -0x1.f1976b7ed8fbcp-110;
  const real k = rint(x * 0x1.45f306dc9c883p-1);
  // x - k*pi/2 as head + tail: k*P1 = p1 + e1 and k*P2 = p2 + e2 exactly; x - p1 is exact (Sterbenz)
  const real p1 = k*P1, e1 = fma(k, P1, -p1);
  const real p2 = k*P2, e2 = fma(k, P2, -p2);
  const real r0 = x - p1;
  const real s1 = r0 - e1;
  real bb = s1 - r0;
  const real t1 = (r0 - (s1 - bb)) + (-e1 - bb);
  const real s2 = s1 - p2;
  bb = s2 - s1;
  const real t2 = (s1 - (s2 - bb)) + (-p2 - bb);
  const real lo = ((t1 + t2) - e2) - k*P3;
  const real rh = s2 + lo;
  const real rl = lo - (rh - s2);
  // kernels
  const real z = rh*rh;
  const real v = z*rh;
  const real rs = 8.33333333332248946124e-03 + z*(-1.98412698298579493134e-04 + z*(2.75573137070700676789e-06 +
                  z*(-2.50507602534068634195e-08 + z*1.58969099521155010221e-10)));
  const real ksin = rh - ((z*(0.5*rl - v*rs) - rl) - v*-1.66666666666666324348e-01);
  const real rc = z*(4.16666666666666019037e-02 + z*(-1.38888888888741095749e-03 + z*(2.48015872894767294178e-05 +
                  z*(-2.75573143513906633035e-07 + z*(2.08757232129817482790e-09 + z*-1.13596475577881948265e-11)))));
  const real hz = 0.5*z;
  const real w = 1.0 - hz;
  const real kcos = w + (((1.0 - w) - hz) + (z*rc - rh*rl));
  const int q = (int)((long long)k & 3);
  const real a = (q & 1) ? kcos : ksin;      // |sin|-role value of this quadrant
  const real b = (q & 1) ? ksin : kcos;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
}
// ---- atan2 and exp ------------------------------------------------------------------------------
// Like mjh_sincos: what the DEVICE evaluates instead of its library's routines, so that the compiled reference can
// be linked against the very same code (oracle/devmath_shim.cc) and the two sides agree bit for bit.  Plain
// multiplications, additions and divisions only (all correctly rounded on both sides, -ffp-contract=off): the
// classic fdlibm algorithms (s_atan.c / e_atan2.c / e_exp.c: argument reduction to a table of four arctangents,
// resp. to |r| <= ln2/2, then a minimax polynomial): exp < 0.8 ulp, atan2 < 1.5 ulp (the rounding of y/x on top of atan's);
// tests/test_convex_hostsim.py::test_device_atan2_exp_accuracy.
MJH_DEV real mjh_atan(real x) {
  const real hi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
  const real lo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
  const real aT[11] = {3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
                       -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
                       6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
                       -3.65315727442169155270e-02, 1.62858201153657823623e-02};
  if (x != x) return x;
  const int neg = x < 0;
  real a = fabs(x);
  if (a >= 0x1p66) { const real z = hi[3] + lo[3]; return neg ? -z : z; }
  int id;
  if (a < 0.4375) {
    if (a < 0x1p-29) return x;
    id = -1;
  } else if (a < 1.1875) {
    if (a < 0.6875) { id = 0; a = (2.0*a - 1.0)/(2.0 + a); }
    else { id = 1; a = (a - 1.0)/(a + 1.0); }
  } else if (a < 2.4375) { id = 2; a = (a - 1.5)/(1.0 + 1.5*a); }
  else { id = 3; a = -1.0/a; }
  const real z = a*a, w = z*z;
  const real s1 = z*(aT[0] + w*(aT[2] + w*(aT[4] + w*(aT[6] + w*(aT[8] + w*aT[10])))));
  const real s2 = w*(aT[1] + w*(aT[3] + w*(aT[5] + w*(aT[7] + w*aT[9]))));
  if (id < 0) { const real r = a - a*(s1 + s2); return neg ? -r : r; }
  const real r = hi[id] - ((a*(s1 + s2) - lo[id]) - a);
  return neg ? -r : r;
}
MJH_DEV real mjh_atan2(real y, real x) {
  const real pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  if (x != x || y != y) return x + y;
  if (x == 1.0) return mjh_atan(y);
  const int sy = signbit(y) ? 1 : 0, sx = signbit(x) ? 1 : 0, m = sy + 2*sx;
  if (y == 0) return m == 0 ? y : (m == 1 ? y : (m == 2 ? pi : -pi));
  if (x == 0) return sy ? -pi/2 : pi/2;
  if (isinf(x)) {
    if (isinf(y)) return m == 0 ? pi/4 : (m == 1 ? -pi/4 : (m == 2 ? 3.0*(pi/4) : -3.0*(pi/4)));
    return m == 0 ? (real)0 : (m == 1 ? -(real)0 : (m == 2 ? pi : -pi));
  }
  if (isinf(y)) return sy ? -pi/2 : pi/2;
  // |y/x| beyond 2^60: pi/2 (+ its tail); x < 0 with |y/x| below 2^-60: 0
  const real ay = fabs(y), ax = fabs(x);
  real z;
  if (ay > ax*0x1p60) z = pi/2 + 0.5*pi_lo;
  else if (sx && ay*0x1p60 < ax) z = 0;
  else z = mjh_atan(ay/ax);
  if (m == 0) return z;
  if (m == 1) return -z;
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}
MJH_DEV real mjh_exp(real x) {
  const real ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
  const real P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
             P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 7.09782712893383973096e+02) return __builtin_huge_val();
  if (x < -7.45133219101941108420e+02) return 0;
  real hi = x, lo = 0;
  int k = 0;
  const real ax = fabs(x);
  if (ax > 0.34657359027997264) {            // 0.5 ln2
    k = ax < 1.0397207708399179 ? (x < 0 ? -1 : 1) : (int)(invln2*x + (x < 0 ? -0.5 : 0.5));     // 1.5 ln2
    hi = x - k*ln2HI;
    lo = k*ln2LO;
    x = hi - lo;
  } else if (ax < 0x1p-28) return 1.0 + x;
  const real t = x*x;
  const real c = x - t*(P1 + t*(P2 + t*(P3 + t*(P4 + t*P5))));
  if (k == 0) return 1.0 - ((x*c)/(c - 2.0) - x);
  const real y = 1.0 - ((lo - (x*c)/(2.0 - c)) - hi);
  return ldexp(y, k);
}
// atan2 / exp as the kernels take them: libm on the host emulation (compared bit for bit with the reference, which
// calls the host's libm), the routines above on the device
MJH_DEV real r_atan2(real y, real x) {
#ifdef MJH_HOSTSIM
  return atan2(y, x);
#else
  return mjh_atan2(y, x);
#endif
}
MJH_DEV real r_exp(real x) {
#ifdef MJH_HOSTSIM
  return exp(x);
#else
  return mjh_exp(x);
#endif
}

// sin and cos of x: libm on the host emulation (it is compared bit for bit with the reference, which
// calls the host's libm), mjh_sincos on the device
MJH_DEV void r_sincos(real x, real* sn, real* cs) {
#ifdef MJH_HOSTSIM
  *sn = sin(x); *cs = cos(x);
#else
  mjh_sincos(x, sn, cs);
#endif
}

template <class P0, class P1>
MJH_DEV void q_copy(P0 r, P1 q) { r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3]; }
// normalize quaternion in place                (mju_normalize4 / mji__normalize4, engine_inline.h:228)
template <class P0>
MJH_DEV real q_normalize(P0 q) {
  real n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < MJH_MINVAL) {
    q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0;
  } else if (fabs(n - 1) > MJH_MINVAL) {
    real inv = 1/n;
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
  }
  return n;
}
// r = qa * qb (r may alias)                    (mju_mulQuat, engine_util_spatial.c:66)
template <class P0, class P1, class P2>
MJH_DEV void q_mul(P0 r, P1 a, P2 b) {
  real t0 = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  real t1 = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  real t2 = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  real t3 = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
// rotate vector by quaternion, r must not alias q   (mji_rotVecQuat, engine_inline.h:252)
template <class P0, class P1, class P2>
MJH_DEV void q_rotvec(P0 r, P1 v, P2 q) {
  if (q[0] == 1 && q[1] == 0 && q[2] == 0 && q[3] == 0) {
    real v0 = v[0], v1 = v[1], v2 = v[2];
    r[0] = v0; r[1] = v1; r[2] = v2;
  } else {
    real v0 = v[0], v1 = v[1], v2 = v[2];
    real t0 = q[0]*v0 + q[2]*v2 - q[3]*v1;
    real t1 = q[0]*v1 + q[3]*v0 - q[1]*v2;
    real t2 = q[0]*v2 + q[1]*v1 - q[2]*v0;
    r[0] = v0 + 2 * (q[2]*t2 - q[3]*t1);
    r[1] = v1 + 2 * (q[3]*t0 - q[1]*t2);
    r[2] = v2 + 2 * (q[1]*t1 - q[2]*t0);
  }
}
// axis-angle to quaternion                     (mji_axisAngle2Quat, engine_inline.h:308)
template <class P0, class P1>
MJH_DEV void q_axisangle(P0 r, P1 axis, real angle) {
  if (angle == 0) {
    r[0] = 1; r[1] = 0; r[2] = 0; r[3] = 0;
  } else {
    real s, c;
    r_sincos(angle*0.5, &s, &c);
    r[0] = c;
    r[1] = axis[0]*s;
    r[2] = axis[1]*s;
    r[3] = axis[2]*s;
  }
}
// quaternion to rotation matrix                (mju_quat2Mat, engine_util_spatial.c:145)
template <class P0, class P1>
MJH_DEV void q_tomat(P0 r, P1 q) {
  if (q[0] == 1 && q[1] == 0 && q[2] == 0 && q[3] == 0) {
    r[0] = 1; r[1] = 0; r[2] = 0;
    r[3] = 0; r[4] = 1; r[5] = 0;
    r[6] = 0; r[7] = 0; r[8] = 1;
  } else {
    real q00 = q[0]*q[0], q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
    real q11 = q[1]*q[1], q12 = q[1]*q[2], q13 = q[1]*q[3];
    real q22 = q[2]*q[2], q23 = q[2]*q[3], q33 = q[3]*q[3];
    r[0] = q00 + q11 - q22 - q33;
    r[4] = q00 - q11 + q22 - q33;
    r[8] = q00 - q11 - q22 + q33;
    r[1] = 2*(q12 - q03);
    r[2] = 2*(q13 + q02);
    r[3] = 2*(q12 + q03);
    r[5] = 2*(q23 - q01);
    r[6] = 2*(q13 - q02);
    r[7] = 2*(q23 + q01);
  }
}
// orientation-difference quaternion -> 3D velocity   (mji_quat2Vel, engine_inline.h:330)
template <class P0, class P1>
MJH_DEV void q_tovel(P0 r, P1 q, real dt) {
  real axis[3] = {q[1], q[2], q[3]};
  real sin_a_2 = v3_normalize(axis);
  real speed = 2 * r_atan2(sin_a_2, q[0]);
  if (speed > MJH_PI) speed -= 2*MJH_PI;
  speed /= dt;
  v3_scl(r, axis, speed);
}
// r = vel such that qb*quat(r) = qa             (mji_subQuat, engine_inline.h:347)
template <class P0, class P1, class P2>
MJH_DEV void q_sub(P0 r, P1 qa, P2 qb) {
  real qneg[4] = {qb[0], -qb[1], -qb[2], -qb[3]};
  real qdif[4];
  q_mul(qdif, qneg, qa);
  q_tovel(r, qdif, 1);
}
// integrate quaternion by angular velocity      (mju_quatIntegrate, engine_util_spatial.c:234)
template <class P0, class P1>
MJH_DEV void q_integrate(P0 q, P1 vel, real scale) {
  real tmp[3] = {vel[0], vel[1], vel[2]};
  real angle = scale * v3_normalize(tmp);
  real qrot[4];
  q_axisangle(qrot, tmp, angle);
  q_normalize(q);
  q_mul(q, q, qrot);
}

// ---- spatial (6D, rotation:translation) -------------------------------------------------------

// motion cross product                          (mji_crossMotion, engine_inline.h:428)
template <class P0, class P1, class P2>
MJH_DEV void sp_cross_motion(P0 r, P1 vel, P2 v) {
  r[0] = -vel[2]*v[1] + vel[1]*v[2];
  r[1] =  vel[2]*v[0] - vel[0]*v[2];
  r[2] = -vel[1]*v[0] + vel[0]*v[1];
  r[3] = -vel[2]*v[4] + vel[1]*v[5];
  r[4] =  vel[2]*v[3] - vel[0]*v[5];
  r[5] = -vel[1]*v[3] + vel[0]*v[4];
  r[3] += -vel[5]*v[1] + vel[4]*v[2];
  r[4] +=  vel[5]*v[0] - vel[3]*v[2];
  r[5] += -vel[4]*v[0] + vel[3]*v[1];
}
// force cross product                           (mji_crossForce, engine_inline.h:445)
template <class P0, class P1, class P2>
MJH_DEV void sp_cross_force(P0 r, P1 vel, P2 f) {
  r[0] = -vel[2]*f[1] + vel[1]*f[2];
  r[1] =  vel[2]*f[0] - vel[0]*f[2];
  r[2] = -vel[1]*f[0] + vel[0]*f[1];
  r[3] = -vel[2]*f[4] + vel[1]*f[5];
  r[4] =  vel[2]*f[3] - vel[0]*f[5];
  r[5] = -vel[1]*f[3] + vel[0]*f[4];
  r[0] += -vel[5]*f[4] + vel[4]*f[5];
  r[1] +=  vel[5]*f[3] - vel[3]*f[5];
  r[2] += -vel[4]*f[3] + vel[3]*f[4];
}
// 6D dot in mju_dot's association               (mji_dot6, engine_inline.h:462)
template <class P0, class P1>
MJH_DEV real sp_dot6(P0 a, P1 b) {
  return ((a[0]*b[0] + a[2]*b[2]) + (a[1]*b[1] + a[3]*b[3])) + (a[4]*b[4] + a[5]*b[5]);
}
// r = I(10) * v(6)                              (mju_mulInertVec, engine_util_spatial.c:439)
template <class P0, class P1, class P2>
MJH_DEV void sp_mul_inert(P0 r, P1 i, P2 v) {
  r[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  r[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  r[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  r[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  r[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  r[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
// body inertia in the com-based frame           (mju_inertCom, engine_util_spatial.c:405)
template <class P0, class P1, class P2, class P3>
MJH_DEV void sp_inert_com(P0 r, P1 inert, P2 mat, P3 dif, real mass) {
  real t[9] = {mat[0]*inert[0], mat[3]*inert[0], mat[6]*inert[0],
               mat[1]*inert[1], mat[4]*inert[1], mat[7]*inert[1],
               mat[2]*inert[2], mat[5]*inert[2], mat[8]*inert[2]};
  r[0] = mat[0]*t[0] + mat[1]*t[3] + mat[2]*t[6];
  r[1] = mat[3]*t[1] + mat[4]*t[4] + mat[5]*t[7];
  r[2] = mat[6]*t[2] + mat[7]*t[5] + mat[8]*t[8];
  r[3] = mat[0]*t[1] + mat[1]*t[4] + mat[2]*t[7];
  r[4] = mat[0]*t[2] + mat[1]*t[5] + mat[2]*t[8];
  r[5] = mat[3]*t[2] + mat[4]*t[5] + mat[5]*t[8];
  r[0] += mass*(dif[1]*dif[1] + dif[2]*dif[2]);
  r[1] += mass*(dif[0]*dif[0] + dif[2]*dif[2]);
  r[2] += mass*(dif[0]*dif[0] + dif[1]*dif[1]);
  r[3] -= mass*dif[0]*dif[1];
  r[4] -= mass*dif[0]*dif[2];
  r[5] -= mass*dif[1]*dif[2];
  r[6] = mass*dif[0];
  r[7] = mass*dif[1];
  r[8] = mass*dif[2];
  r[9] = mass;
}

// ---- misc -------------------------------------------------------------------------------------

MJH_DEV real r_clip(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
MJH_DEV real r_max(real a, real b) { return a > b ? a : b; }
MJH_DEV real r_min(real a, real b) { return a < b ? a : b; }
// mju_isBad, engine_util_misc.c:2022
MJH_DEV int r_isbad(real x) { return (x != x) || (x > MJH_MAXVAL) || (x < -MJH_MAXVAL); }

// dense dot product in mju_dot's association (4 interleaved partial sums, then (r0+r2)+(r1+r3),
// then the 1..3 element tail as ONE expression) -- engine_util_blas.c:493-527.
// stride-aware so rows/columns of env-local matrices can be passed directly.
template <class P0, class P1>
MJH_DEV real dot_ref(P0 a, P1 b, int n) {
  real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  int i = 0;
  for (; i <= n - 4; i += 4) {
    r0 += a[i]*b[i];
    r1 += a[i+1]*b[i+1];
    r2 += a[i+2]*b[i+2];
    r3 += a[i+3]*b[i+3];
  }
  real res = (r0 + r2) + (r1 + r3);
  int rem = n - i;
  if (rem == 3) {
    res += a[i]*b[i] + a[i+1]*b[i+1] + a[i+2]*b[i+2];
  } else if (rem == 2) {
    res += a[i]*b[i] + a[i+1]*b[i+1];
  } else if (rem == 1) {
    res += a[i]*b[i];
  }
  return res;
}
// sparse . dense in mju_dotSparse's association (tail added one by one) -- engine_util_sparse.h:197
template <class P0, class P1, class P2>
MJH_DEV real dot_sparse_ref(P0 a, P1 x, int nnz, P2 ind) {
  real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  int i = 0;
  for (; i <= nnz - 4; i += 4) {
    r0 += a[i]*x[ind[i]];
    r1 += a[i+1]*x[ind[i+1]];
    r2 += a[i+2]*x[ind[i+2]];
    r3 += a[i+3]*x[ind[i+3]];
  }
  real res = (r0 + r2) + (r1 + r3);
  for (; i < nnz; i++) res += a[i]*x[ind[i]];
  return res;
}

// ---- tendon wrapping around spheres and cylinders ------------------------------------------------------------------
// mju_wrap and its helpers (engine_util_misc.c:36-413), operation for operation: plain scalar code, one tendon per
// caller.  acos / asin: libm on the host emulation; on the device through mjh_atan2 (which oracle/devmath_shim.cc binds
// the reference's acos / asin to as well, for the device-libm build of the oracle).
MJH_DEV real mjh_acos(real x) { return mjh_atan2(sqrt((1 - x)*(1 + x)), x); }
MJH_DEV real mjh_asin(real x) { return mjh_atan2(x, sqrt((1 - x)*(1 + x))); }
MJH_DEV real r_acos(real x) {
#ifdef MJH_HOSTSIM
  return acos(x);
#else
  return mjh_acos(x);
#endif
}
MJH_DEV real r_asin(real x) {
#ifdef MJH_HOSTSIM
  return asin(x);
#else
  return mjh_asin(x);
#endif
}
// mju_dot / mju_normalize for n = 2 (the four-accumulator dot product degenerates to 0 + (a0 b0 + a1 b1))
MJH_DEV real wr_dot2(const real* a, const real* b) { return a[0]*b[0] + a[1]*b[1]; }
MJH_DEV real wr_normalize2(real* v) {
  const real n = sqrt(wr_dot2(v, v));
  if (n < MJH_MINVAL) { v[0] = 1; v[1] = 0; }
  else { const real inv = 1/n; v[0] *= inv; v[1] *= inv; }
  return n;
}
// do the 2D segments (p1, p2) and (p3, p4) cross?                                   (is_intersect, :36-51)
MJH_DEV int wr_intersect(const real* p1, const real* p2, const real* p3, const real* p4) {
  const real det = (p4[1] - p3[1])*(p2[0] - p1[0]) - (p4[0] - p3[0])*(p2[1] - p1[1]);
  if (fabs(det) < MJH_MINVAL) return 0;
  const real a = ((p4[0] - p3[0])*(p1[1] - p3[1]) - (p4[1] - p3[1])*(p1[0] - p3[0])) / det;
  const real b = ((p2[0] - p1[0])*(p1[1] - p3[1]) - (p2[1] - p1[1])*(p1[0] - p3[0])) / det;
  return a >= 0 && a <= 1 && b >= 0 && b <= 1;
}
// arc length on the circle between two of its points                                (length_circle, :55-71)
MJH_DEV real wr_arc(const real* p0, const real* p1, int ind, real radius) {
  real p0n[2] = {p0[0], p0[1]}, p1n[2] = {p1[0], p1[1]};
  wr_normalize2(p0n);
  wr_normalize2(p1n);
  real angle = r_acos(wr_dot2(p0n, p1n));
  const real cross = p0[1]*p1[0] - p0[0]*p1[1];
  if ((cross > 0 && ind) || (cross < 0 && !ind)) angle = 2*MJH_PI - angle;
  return radius*angle;
}
// 2D wrap around a circle at the origin: tangent points in pnt[4], arc length or -1    (wrap_circle, :78-151)
MJH_DEV real wr_circle(real* pnt, const real* end, const real* side, real radius) {
  const real sqlen0 = end[0]*end[0] + end[1]*end[1];
  const real sqlen1 = end[2]*end[2] + end[3]*end[3];
  const real sqrad = radius*radius;
  if (sqlen0 < sqrad || sqlen1 < sqrad || radius < MJH_MINVAL) return -1;
  const real dif[2] = {end[2] - end[0], end[3] - end[1]};
  const real dd = dif[0]*dif[0] + dif[1]*dif[1];
  if (dd < MJH_MINVAL) return -1;
  real a = -(dif[0]*end[0] + dif[1]*end[1])/dd;
  if (a < 0) a = 0; else if (a > 1) a = 1;
  const real tmp0[2] = {a*dif[0] + end[0], a*dif[1] + end[1]};
  if (tmp0[0]*tmp0[0] + tmp0[1]*tmp0[1] > sqrad && (!side || wr_dot2(side, tmp0) >= 0)) return -1;
  const real sqrt0 = sqrt(sqlen0 - sqrad), sqrt1 = sqrt(sqlen1 - sqrad);
  real sol[2][2][2], good[2];
  for (int i = 0; i < 2; i++) {
    const int sgn = i == 0 ? 1 : -1;
    sol[i][0][0] = (end[0]*sqrad + sgn*radius*end[1]*sqrt0)/sqlen0;
    sol[i][0][1] = (end[1]*sqrad - sgn*radius*end[0]*sqrt0)/sqlen0;
    sol[i][1][0] = (end[2]*sqrad - sgn*radius*end[3]*sqrt1)/sqlen1;
    sol[i][1][1] = (end[3]*sqrad + sgn*radius*end[2]*sqrt1)/sqlen1;
    real tmp[2];
    if (side) {
      tmp[0] = sol[i][0][0] + sol[i][1][0]; tmp[1] = sol[i][0][1] + sol[i][1][1];
      wr_normalize2(tmp);
      good[i] = wr_dot2(tmp, side);
    } else {
      tmp[0] = sol[i][0][0] - sol[i][1][0]; tmp[1] = sol[i][0][1] - sol[i][1][1];
      good[i] = -wr_dot2(tmp, tmp);
    }
    if (wr_intersect(end, sol[i][0], end + 2, sol[i][1])) good[i] = -10000;
  }
  const int i = good[0] > good[1] ? 0 : 1;
  pnt[0] = sol[i][0][0]; pnt[1] = sol[i][0][1]; pnt[2] = sol[i][1][0]; pnt[3] = sol[i][1][1];
  if (wr_intersect(end, pnt, end + 2, pnt + 2)) return -1;
  return wr_arc(sol[i][0], sol[i][1], i, radius);
}
// 2D wrap on the INSIDE of a circle: one touching point (twice) in pnt[4]; 0, or -1 without a wrap   (wrap_inside, :158-272)
MJH_DEV real wr_inside(real* pnt, const real* end, real radius) {
  const int maxiter = 20;
  const real zinit = 1 - 1e-7, tolerance = 1e-6;
  const real len0 = sqrt(wr_dot2(end, end)), len1 = sqrt(wr_dot2(end + 2, end + 2));
  const real dif[2] = {end[2] - end[0], end[3] - end[1]};
  const real dd = dif[0]*dif[0] + dif[1]*dif[1];
  if (len0 <= radius || len1 <= radius || radius < MJH_MINVAL || len0 < MJH_MINVAL || len1 < MJH_MINVAL) return -1;
  if (dd > MJH_MINVAL) {
    const real a = -(dif[0]*end[0] + dif[1]*end[1]) / dd;
    if (a > 0 && a < 1) {
      const real tmp[2] = {end[0] + dif[0]*a, end[1] + dif[1]*a};
      if (sqrt(wr_dot2(tmp, tmp)) <= radius) return -1;
    }
  }
  pnt[0] = 0.5*(end[0] + end[2]);
  pnt[1] = 0.5*(end[1] + end[3]);
  wr_normalize2(pnt);
  pnt[0] = pnt[0]*radius; pnt[1] = pnt[1]*radius;
  pnt[2] = pnt[0]; pnt[3] = pnt[1];
  const real A = radius/len0, Bc = radius/len1;
  const real cosG = (len0*len0 + len1*len1 - dd) / (2*len0*len1);
  if (cosG < -1 + MJH_MINVAL) return -1;
  else if (cosG > 1 - MJH_MINVAL) return 0;
  const real G = r_acos(cosG);
  real z = zinit;
  real f = r_asin(A*z) + r_asin(Bc*z) - 2*r_asin(z) + G;
  if (f > 0) return 0;
  int iter;
  for (iter = 0; iter < maxiter && fabs(f) > tolerance; iter++) {
    const real s0 = sqrt(1 - z*z*A*A), s1 = sqrt(1 - z*z*Bc*Bc), s2 = sqrt(1 - z*z);
    const real df = A/(MJH_MINVAL > s0 ? MJH_MINVAL : s0) + Bc/(MJH_MINVAL > s1 ? MJH_MINVAL : s1) - 2/(MJH_MINVAL > s2 ? MJH_MINVAL : s2);
    if (df > -MJH_MINVAL) return 0;
    const real z1 = z - f/df;
    if (z1 > z) return 0;
    z = z1;
    f = r_asin(A*z) + r_asin(Bc*z) - 2*r_asin(z) + G;
    if (f > tolerance) return 0;
  }
  if (iter >= maxiter) return 0;
  real vec[2], ang;
  if (end[0]*end[3] - end[1]*end[2] > 0) { vec[0] = end[0]; vec[1] = end[1]; ang = r_asin(z) - r_asin(A*z); }
  else { vec[0] = end[2]; vec[1] = end[3]; ang = r_asin(z) - r_asin(Bc*z); }
  wr_normalize2(vec);
  real sn, cs;
  r_sincos(ang, &sn, &cs);
  pnt[0] = radius*(cs*vec[0] - sn*vec[1]);
  pnt[1] = radius*(sn*vec[0] + cs*vec[1]);
  pnt[2] = pnt[0]; pnt[3] = pnt[1];
  return 0;
}
// mju_wrap (:283-413): the two points where the path x0 -> x1 meets the wrapping geom (type 4: sphere, 5: cylinder, the
// mjtWrap values) in wpnt[6], arc length between them, or -1 when the straight segment does not touch it
MJH_DEV real mjh_wrap(real* wpnt, const real* x0, const real* x1, const real* xpos, const real* xmat, real radius,
                      int type, const real* side) {
  real tmp[3], p[2][3];
  for (int k = 0; k < 3; k++) tmp[k] = x0[k] - xpos[k];
  m3_multvec(p[0], xmat, tmp);
  for (int k = 0; k < 3; k++) tmp[k] = x1[k] - xpos[k];
  m3_multvec(p[1], xmat, tmp);
  if (v3_norm(p[0]) < MJH_MINVAL || v3_norm(p[1]) < MJH_MINVAL) return -1;
  real axis[2][3];
  if (type == 4) {
    v3_copy(axis[0], p[0]);
    v3_normalize(axis[0]);
    real normal[3];
    v3_cross(normal, p[0], p[1]);
    const real nrm = v3_normalize(normal);
    if (nrm < MJH_MINVAL) {
      int i = 0;
      if (fabs(axis[0][1]) > fabs(axis[0][0]) && fabs(axis[0][1]) > fabs(axis[0][2])) i = 1;
      if (fabs(axis[0][2]) > fabs(axis[0][0]) && fabs(axis[0][2]) > fabs(axis[0][1])) i = 2;
      axis[1][0] = 1; axis[1][1] = 1; axis[1][2] = 1;
      axis[1][i] = 0;
      v3_cross(normal, axis[0], axis[1]);
      v3_normalize(normal);
    }
    v3_cross(axis[1], normal, axis[0]);
    v3_normalize(axis[1]);
  } else {
    axis[0][0] = 1; axis[0][1] = 0; axis[0][2] = 0;
    axis[1][0] = 0; axis[1][1] = 1; axis[1][2] = 0;
  }
  real s[3] = {0, 0, 0}, d[4], sd[2] = {0, 0};
  d[0] = v3_dot(p[0], axis[0]);
  d[1] = v3_dot(p[0], axis[1]);
  d[2] = v3_dot(p[1], axis[0]);
  d[3] = v3_dot(p[1], axis[1]);
  if (side) {
    for (int k = 0; k < 3; k++) tmp[k] = side[k] - xpos[k];
    m3_multvec(s, xmat, tmp);
    sd[0] = v3_dot(s, axis[0]);
    sd[1] = v3_dot(s, axis[1]);
    wr_normalize2(sd);
    sd[0] = sd[0]*radius; sd[1] = sd[1]*radius;
  }
  real wlen, pnt[4];
  if (side && v3_norm(s) < radius) wlen = wr_inside(pnt, d, radius);
  else wlen = wr_circle(pnt, d, side ? sd : (const real*)nullptr, radius);
  if (wlen < 0) return -1;
  real res[6];
  for (int i = 0; i < 2; i++) {
    for (int k = 0; k < 3; k++) res[3*i + k] = axis[0][k]*pnt[2*i];
    for (int k = 0; k < 3; k++) tmp[k] = axis[1][k]*pnt[2*i + 1];
    for (int k = 0; k < 3; k++) res[3*i + k] += tmp[k];
  }
  if (type == 5) {
    const real L0 = sqrt((p[0][0] - res[0])*(p[0][0] - res[0]) + (p[0][1] - res[1])*(p[0][1] - res[1]));
    const real L1 = sqrt((p[1][0] - res[3])*(p[1][0] - res[3]) + (p[1][1] - res[4])*(p[1][1] - res[4]));
    res[2] = p[0][2] + (p[1][2] - p[0][2])*L0 / (L0 + wlen + L1);
    res[5] = p[0][2] + (p[1][2] - p[0][2])*(L0 + wlen) / (L0 + wlen + L1);
    const real height = fabs(res[5] - res[2]);
    wlen = sqrt(wlen*wlen + height*height);
  }
  m3_mulvec(wpnt, xmat, res);
  m3_mulvec(wpnt + 3, xmat, res + 3);
  for (int k = 0; k < 3; k++) { wpnt[k] += xpos[k]; wpnt[3 + k] += xpos[k]; }
  return wlen;
}
