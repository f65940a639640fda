// Primal solvers: Newton (mj_solNewton) and conjugate gradient (mj_solCG) -- mj_solPrimal,
// engine_solver.c:2344-2563, restated OPERATION FOR OPERATION for the dense-Jacobian case:
//   PrimalUpdateConstraint / Grad / Mgrad :1370-1448      PrimalPrepare :1451-1524
//   PrimalEval :1675-1822 (shifted costs: cost(alpha) - cost(0))      PrimalSearch :1856-2054
//   MakeHessian :2057-2143 (mju_sqrMatTD_impl + mju_addToSymSparse)   FactorizeHessian :2149-2216
//   HessianCone :2219-2281   HessianIncremental :2285-2340 (mju_cholUpdate, engine_util_solve.c:104)
//   mju_cholFactor / mju_cholSolve (engine_util_solve.c:27-102)       mju_mulSymVecSparse (engine_util_sparse.c)
//
// The Hessian H = M + J' D J is built and factorised ONCE per solve; afterwards every constraint that
// enters or leaves the quadratic zone costs one rank-one update / downdate of the factor (O(nv^2)), as in
// the reference -- round 2 rebuilt and refactorised H every iteration, which is what made this solver
// slow and its iteration counts approximate.
//
// Mapping (one wavefront per environment):
//   * the factor L lives column-major, lower triangle packed (column k holds rows k..n-1) so that "lane = row i"
//     walks a column with unit stride: the dot products of mju_cholFactor, the column sweeps of mju_cholUpdate and the
//     back substitution all read coalesced; L sits in LDS when the residency plan leaves room;
//   * vectors over dofs live in registers (lane = dof, a second register for dofs 64..127);
//   * the reference's sums are sequential: dot products in mju_dot's four-accumulator order, constraint
//     costs in row order.  The addends are computed in parallel (lane = row / column) and combined by
//     the ordered reductions of mjh_spmd.h (wv_chain, wv_chain6, wv_dot4_acc) -- serial chains of
//     register operands, ~1 us per 200 rows -- so every branch the solver takes (line-search brackets,
//     termination tests, which constraints change state) is the reference's own.
//
// Parity: bit for bit on models the reference solves with its dense path (mj_isSparse false: nv < 60) and
// as ONE problem (no islands, or one island that spans every dof).  Elsewhere the same arithmetic runs in
// a different summation order (the reference's sparse routines; island-compressed vectors), which keeps
// the states within the solver tolerance and usually the iteration counts too.
// (included once per SPMD mode by mjh_modes.h -- no include guard, no includes of its own)

#if MJH_LANE_MODE
// one lane per environment never runs the constraint solve (the SoA pipeline sends the constraint stages
// to the wave-per-environment kernel): flag the environment instead of carrying a second implementation
MJH_DEVN void solve_newton(MREF M_, BREF B_, int e_) { MJH_ENTER(M_, B_, e_); MJH_F(B, warning, e)[MJH_WARN_UNSUPPORTED]++; }
MJH_DEVN void solve_cg(MREF M_, BREF B_, int e_) { MJH_ENTER(M_, B_, e_); MJH_F(B, warning, e)[MJH_WARN_UNSUPPORTED]++; }
#else

// frictionCost / frictionCostDif (engine_solver.c:1536-1570)
MJH_DEV real nt_friction_cost(real x, real f, real Rf, real D) {
  if (-Rf < x && x < Rf) return 0.5*D*x*x;
  else if (x <= -Rf) return f*(-0.5*Rf - x);
  else return f*(-0.5*Rf + x);
}
MJH_DEV real nt_friction_costdif(real start, real x, real f, real Rf, real D) {
  const int s0 = (-Rf < start && start < Rf) ? 0 : (start <= -Rf ? -1 : 1);
  const int s1 = (-Rf < x && x < Rf) ? 0 : (x <= -Rf ? -1 : 1);
  if (s0 == 0 && s1 == 0) return 0.5*D*(x - start)*(x + start);
  if (s0 == -1 && s1 == -1) return f*(start - x);
  if (s0 == 1 && s1 == 1) return f*(x - start);
  return nt_friction_cost(x, f, Rf, D) - nt_friction_cost(start, f, Rf, D);
}

// ellipticCostDif (engine_solver.c:1573-1672): q = the block's 9 quad slots
template <class Q>
MJH_DEV real nt_elliptic_costdif(Q q, real alpha, real mu, real Dm) {
  const real U0 = q[3], V0 = q[4], UU = q[5], UV = q[6], VV = q[7];
  int zone0;
  real T0 = 0;
  if (UU <= 0) zone0 = (U0 < 0) ? 2 : 1;
  else {
    T0 = sqrt(UU);
    if (U0 >= mu*T0) zone0 = 1;
    else if (mu*U0 + T0 <= 0) zone0 = 2;
    else zone0 = 3;
  }
  const real N = U0 + alpha*V0;
  const real Tsqr = UU + alpha*(2*UV + alpha*VV);
  int zone_a;
  real T = 0;
  if (Tsqr <= 0) zone_a = (N < 0) ? 2 : 1;
  else {
    T = sqrt(Tsqr);
    if (N >= mu*T) zone_a = 1;
    else if (mu*N + T <= 0) zone_a = 2;
    else zone_a = 3;
  }
  if (zone0 == 1 && zone_a == 1) return 0;
  if (zone0 == 2 && zone_a == 2) return alpha*alpha*q[2] + alpha*q[1];
  if (zone0 == 3 && zone_a == 3) {
    const real Tsqr_delta = alpha*(2*UV + alpha*VV);
    const real T_delta = Tsqr_delta/(T + T0);
    const real r_delta = alpha*V0 - mu*T_delta;
    const real r0 = U0 - mu*T0;
    return 0.5*Dm*r_delta*(2*r0 + r_delta);
  }
  if (zone0 == 3 && zone_a == 2) {
    const real dq = alpha*(alpha*q[2] + q[1]);
    const real b0 = mu*U0 + T0;
    return dq + 0.5*Dm*b0*b0;
  }
  if (zone0 == 2 && zone_a == 3) {
    const real dq = alpha*(alpha*q[2] + q[1]);
    const real bb = mu*N + T;
    return dq - 0.5*Dm*bb*bb;
  }
  if (zone0 == 1 && zone_a == 2) return alpha*alpha*q[2] + alpha*q[1] + q[0];
  if (zone0 == 1 && zone_a == 3) { const real r = N - mu*T; return 0.5*Dm*r*r; }
  if (zone0 == 3 && zone_a == 1) { const real r0 = U0 - mu*T0; return -0.5*Dm*r0*r0; }
  if (zone0 == 2 && zone_a == 1) return -q[0];
  return 0;
}

// ---- dense factor, nv <= 64: out-of-line solve and rank-one update (own register scope; the factor through a local
// -- ds_read / ds_write -- pointer when it sits in LDS; vectors by value; every pivot in the register of the lane named
// after its row, so no lane reads a pivot from memory while its owner rewrites it).  Packed lower triangle,
// column-major: element (row i, column k), i >= k, at k*nv - k(k-1)/2 + (i - k).
MJH_DEV int nt_lx(int nv, int k, int i) { return k*nv - k*(k - 1)/2 + (i - k); }
// mju_cholSolve: forward substitution by row dots (mju_dot's order), back substitution by sequential subtraction
template <class PL>
MJH_DEVN_HOT real dn_solve(PL L, int nv, real y0) {
  const int lane = wv_lane();
  const real dg = lane < nv ? (real)L[nt_lx(nv, lane, lane)] : (real)1;
  for (int i = 0; i < nv; i++) {
    const real p0 = lane < i ? (real)(L[nt_lx(nv, lane, i)]*y0) : (real)0;
    real yi = wv_bcast(y0, i);
    if (i) {
      real r[4] = {0, 0, 0, 0};
      const int G = i >> 2;
      wv_dot4_acc(r, p0, G);
      real res = (r[0] + r[2]) + (r[1] + r[3]);
      const int c = 4*G, rem = i - c;
      if (rem == 3) res += wv_bcast(p0, c) + wv_bcast(p0, c + 1) + wv_bcast(p0, c + 2);
      else if (rem == 2) res += wv_bcast(p0, c) + wv_bcast(p0, c + 1);
      else if (rem == 1) res += wv_bcast(p0, c);
      yi -= res;
    }
    yi /= wv_bcast(dg, i);
    if (lane == i) y0 = yi;
  }
  for (int i = nv - 1; i >= 0; i--) {
    const real p0 = (lane > i && lane < nv) ? (real)(L[nt_lx(nv, i, lane)]*y0) : (real)0;
    real yi = wv_bcast(y0, i);
    yi = wv_chain(yi, p0, i + 1, nv, 1);
    yi /= wv_bcast(dg, i);
    if (lane == i) y0 = yi;
  }
  return y0;
}
// the same for an island that leaves dofs out (`mask`: the island's dofs): the reference solves the island-local system, so
// the row dots of the forward substitution group the island's finished entries by their position inside the island; rows
// outside the island carry a zero right-hand side and are skipped
template <class PL>
MJH_DEVN real dn_solve_isl(PL L, int nv, real y0, uint64_t mask) {
  const int lane = wv_lane();
  const real dg = lane < nv ? (real)L[nt_lx(nv, lane, lane)] : (real)1;
  for (int i = 0; i < nv; i++) {
    if (!((mask >> i) & 1)) continue;
    const real p0 = lane < i ? (real)(L[nt_lx(nv, lane, i)]*y0) : (real)0;
    real yi = wv_bcast(y0, i);
    yi -= wv_dot4m(p0, (real)0, mask & ((1ull << i) - 1), 0ull, 1);
    yi /= wv_bcast(dg, i);
    if (lane == i) y0 = yi;
  }
  for (int i = nv - 1; i >= 0; i--) {
    if (!((mask >> i) & 1)) continue;
    const real p0 = (lane > i && lane < nv) ? (real)(L[nt_lx(nv, i, lane)]*y0) : (real)0;
    real yi = wv_bcast(y0, i);
    yi = wv_chain(yi, p0, i + 1, nv, 1);
    yi /= wv_bcast(dg, i);
    if (lane == i) y0 = yi;
  }
  return y0;
}
// mju_cholUpdate(L, x, flg_plus); returns the number of clamped pivots
template <class PL>
MJH_DEVN_HOT int dn_update(PL L, int nv, real x0, int flg_plus) {
  const int lane = wv_lane();
  real dg = lane < nv ? (real)L[nt_lx(nv, lane, lane)] : (real)1;
  int clamped = 0;
  for (int k = 0; k < nv; k++) {
    const real xk = wv_bcast(x0, k);
    if (xk == 0) continue;
    const real Lkk = wv_bcast(dg, k);
    real tmp = Lkk*Lkk + (flg_plus ? xk*xk : -xk*xk);
    if (tmp < MJH_MINVAL) { tmp = MJH_MINVAL; clamped++; }
    const real r = sqrt(tmp);
    const real c = r/Lkk;
    const real cinv = 1/c;
    const real sx = xk/Lkk;
    if (lane == k) { dg = r; L[nt_lx(nv, k, k)] = r; }
    else if (lane > k && lane < nv) {
      const int a = nt_lx(nv, k, lane);
      const real l0 = L[a];
      const real lik = flg_plus ? (l0 + sx*x0)*cinv : (l0 - sx*x0)*cinv;
      L[a] = lik;
      x0 = c*x0 - sx*lik;
    }
  }
  wv_sync();
  return clamped;
}

struct NtPoint { real alpha, cost, d0, d1; };

// ELL = 0: instantiation without elliptic-cone code (the common pyramidal case keeps its register budget)
// nd <= 16 dot products over the dofs idof[0..n) at once: product d in the four lanes 4d..4d+3, lane 4d + a running mju_dot's
// accumulator a (elements a, a + 4, ... of the list, in order), then (r0 + r2) + (r1 + r3) plus the sum of the tail
// (engine_util_blas.c:mju_dot).  A chain of n/4 dependent additions is the floor for a sum in the reference's order; the
// independent products of a solver iteration share it.
// elements per lane whose loads are issued together in the nv-long passes of the explicit-index path: with one or two
// wavefronts on a SIMD nothing else hides a global-memory round trip, so the 256-VGPR build (mjh_kern_wide.hip) takes
// eight at a time
#ifdef MJH_WIDE_REGS
#define MJH_NVU 8
#else
#define MJH_NVU 4
#endif
// One accumulator chain of mju_dot: r += x[a + 4 i]*y[a + 4 i] (or r += p[a + 4 i]), i = 0 .. L-1, in order.  The additions
// depend on each other; the reads do not: they are issued a batch ahead (six pairs: with the next batch in flight the
// wait for the oldest stays inside the 4-bit lgkmcnt), so a link costs the latency of an addition, not of a memory access.
template <class PX, class PY>
MJH_DEV real csr_chain2(PX x, PY y, int a, int L) {
  // (two register sets used alternately, each refilled in place as soon as it has been consumed: no copies, and the
  // refill past the end re-reads the last batch instead of branching, so every wait is a partial one)
  real r = 0;
  int i = 0;
  if (L >= 6) {
    const int last = L - 6;
    real xa[6], ya[6], xb[6], yb[6];
#pragma unroll
    for (int u = 0; u < 6; u++) { xa[u] = x[a + 4*u]; ya[u] = y[a + 4*u]; }
    for (; i + 12 <= L; i += 12) {
#pragma unroll
      for (int u = 0; u < 6; u++) { xb[u] = x[a + 4*(i + 6 + u)]; yb[u] = y[a + 4*(i + 6 + u)]; }
#pragma unroll
      for (int u = 0; u < 6; u++) r += xa[u]*ya[u];
      const int nx = i + 12 < last ? i + 12 : last;
#pragma unroll
      for (int u = 0; u < 6; u++) { xa[u] = x[a + 4*(nx + u)]; ya[u] = y[a + 4*(nx + u)]; }
#pragma unroll
      for (int u = 0; u < 6; u++) r += xb[u]*yb[u];
    }
    if (i + 6 <= L) {
#pragma unroll
      for (int u = 0; u < 6; u++) r += xa[u]*ya[u];
      i += 6;
    }
  }
  for (; i < L; i++) r += x[a + 4*i]*y[a + 4*i];
  return r;
}
template <class PP>
MJH_DEV real csr_chain1(PP p, int a, int L) {
  real r = 0;
  int i = 0;
  if (L >= 8) {
    const int last = L - 8;
    real xa[8], xb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) xa[u] = p[a + 4*u];
    for (; i + 16 <= L; i += 16) {
#pragma unroll
      for (int u = 0; u < 8; u++) xb[u] = p[a + 4*(i + 8 + u)];
#pragma unroll
      for (int u = 0; u < 8; u++) r += xa[u];
      const int nx = i + 16 < last ? i + 16 : last;
#pragma unroll
      for (int u = 0; u < 8; u++) xa[u] = p[a + 4*(nx + u)];
#pragma unroll
      for (int u = 0; u < 8; u++) r += xb[u];
    }
    if (i + 8 <= L) {
#pragma unroll
      for (int u = 0; u < 8; u++) r += xa[u];
      i += 8;
    }
  }
  for (; i < L; i++) r += p[a + 4*i];
  return r;
}
// r + p[0] + p[1] + ... + p[n-1], strictly left to right (a single running sum of the reference), every lane on its own
// (uniform result); the reads run a batch ahead of the additions like csr_chain1's
template <class PP>
MJH_DEV real csr_chain_serial(PP p, int n, real r) {
  int i = 0;
  if (n >= 8) {
    const int last = n - 8;
    real xa[8], xb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) xa[u] = p[u];
    for (; i + 16 <= n; i += 16) {
#pragma unroll
      for (int u = 0; u < 8; u++) xb[u] = p[i + 8 + u];
#pragma unroll
      for (int u = 0; u < 8; u++) r += xa[u];
      const int nx = i + 16 < last ? i + 16 : last;
#pragma unroll
      for (int u = 0; u < 8; u++) xa[u] = p[nx + u];
#pragma unroll
      for (int u = 0; u < 8; u++) r += xb[u];
    }
    if (i + 8 <= n) {
#pragma unroll
      for (int u = 0; u < 8; u++) r += xa[u];
      i += 8;
    }
  }
  for (; i < n; i++) r += p[i];
  return r;
}
// element k of a strided walk through a vector (csr_chain_serial over one column of a row-major block)
template <class PP>
struct CsrStrided {
  PP p; int s;
  template <class I> MJH_MEM real operator[](I i) const { return p[i*s]; }
};
// 1 if the contiguous slice lies in the workgroup's LDS block (also on the host emulation, where mjh_in_lds is 0)
MJH_DEV int csr_lds_resident(const crptr& v) {
  const long long off = mjh_lds_offset((const void*)v.p);
  return v.s == 1 && off >= 0 && off < 160*1024;
}
template <class IP>
MJH_DEV void csr_dots(int n, IP idof, int ident, int nd, const crptr* xs, const crptr* ys, real* out, real* stage, int stage_cap) {
  // A product whose two vectors are LDS-resident (launches with one workgroup per CU give the solver the CU's whole
  // 160 KB) over the identity list is summed DIRECT: the chain lane reads x[k], y[k] itself -- two LDS reads that do not
  // depend on the running sum.  The other products are formed lane-parallel (coalesced reads) into a staging block --
  // LDS when the plan leaves room, else global memory -- so that a chain lane's loop is a stream of independent loads
  // feeding one dependent addition each (reading x[idof[k]] from global memory inside the chain put two dependent
  // memory latencies into every link).  stage_cap: reals.
  const int lane = wv_lane();
  const int d = lane >> 2, a = lane & 3;
  const int n4 = n & ~3;
  unsigned direct = 0;
  if (ident) for (int q = 0; q < nd; q++) if (csr_lds_resident(xs[q]) && csr_lds_resident(ys[q])) direct |= 1u << q;
  const int per = stage_cap/(n > 0 ? n : 1) < 16 ? stage_cap/(n > 0 ? n : 1) : 16;
  // ---- direct products: one pass for all of them
  if (direct) {
    const real* px = nullptr; const real* py = nullptr;
    for (int q = 0; q < nd; q++) if (d == q && ((direct >> q) & 1)) { px = xs[q].p; py = ys[q].p; }
    real r = 0;
    if (px) {
      // (local-address-space reads: LDS returns in order, so the additions start with the first pair that arrives
      // instead of waiting for the whole unrolled batch as flat loads must)
      r = csr_chain2(mjh_local(px), mjh_local(py), a, n4 >> 2);
    }
    const real r2 = wv_shfl(r, lane ^ 2);
    const real s02 = r + r2;
    const real s13 = wv_shfl(s02, lane ^ 1);
    real res = s02 + s13;
    if (px && a == 0 && n > n4) {
      real tail = px[n4]*py[n4];
      for (int k = n4 + 1; k < n; k++) tail += px[k]*py[k];
      res += tail;
    }
    for (int q = 0; q < nd; q++) if ((direct >> q) & 1) out[q] = wv_bcast(res, 4*q);
  }
  // ---- staged products, `per` at a time
  int done = 0;
  while (done < nd) {
    int qs[16]; int nb = 0;
    while (done < nd && nb < per) { if (!((direct >> done) & 1)) qs[nb++] = done; done++; }
    if (!nb) break;
    // (four rounds of loads are issued before the first product is stored: with one wavefront per SIMD nothing else
    // hides the latency of these reads, and a store in between would keep the compiler from hoisting the next loads)
    for (int k0 = lane; k0 < n; k0 += MJH_NVU*MJH_W) {
      int ii[MJH_NVU];
#pragma unroll
      for (int u = 0; u < MJH_NVU; u++) { const int k = k0 + u*MJH_W; ii[u] = k < n ? (ident ? k : (int)idof[k]) : 0; }
      for (int q = 0; q < nb; q++) {
        const crptr x = xs[qs[q]], y = ys[qs[q]];
        real xv[MJH_NVU], yv[MJH_NVU];
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { xv[u] = x[ii[u]]; yv[u] = y[ii[u]]; }
        real* p = stage + (size_t)q*n;
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { const int k = k0 + u*MJH_W; if (k < n) p[k] = xv[u]*yv[u]; }
      }
    }
    wv_sync();
    real r = 0;
    if (d < nb) {
      const real* p = stage + (size_t)d*n;
      const long long soff = mjh_lds_offset((const void*)stage);
      r = (soff >= 0 && soff < 160*1024) ? csr_chain1(mjh_local(p), a, n4 >> 2) : csr_chain1(p, a, n4 >> 2);
      const real r2 = wv_shfl(r, lane ^ 2);           // (lane a = 0 adds chain 2, lane 1 chain 3: r0 + r2, r1 + r3)
      const real s02 = r + r2;
      const real s13 = wv_shfl(s02, lane ^ 1);
      real res = s02 + s13;                            // (in lane 4d: (r0 + r2) + (r1 + r3))
      if (a == 0 && n > n4) {
        // (mju_dot adds the sum of the remaining one to three products, engine_util_blas.c:517-525)
        real tail = p[n4];
        for (int k = n4 + 1; k < n; k++) tail += p[k];
        res += tail;
      }
      r = res;
    } else {
      // (collectives are entered by every lane)
      const real r2 = wv_shfl(r, lane ^ 2); const real s13 = wv_shfl(r2, lane ^ 1); (void)s13;
    }
    for (int q = 0; q < nb; q++) out[qs[q]] = wv_bcast(r, 4*q);
    wv_sync();
  }
}

// nd <= 16 ordered sums whose ADDENDS sit in consecutive vectors of n reals (the fused passes of mjh_csrpass.h leave the
// element-wise products there): sum d in the four lanes 4d..4d+3, one accumulator chain of mju_dot each -- a link is one read
// and one dependent addition --, then (r0 + r2) + (r1 + r3) plus the sum of the tail, as in csr_dots
MJH_DEV void csr_sums(int n, int nd, const real* prod, real* out) {
  const int lane = wv_lane();
  const int d = lane >> 2, a = lane & 3;
  const int n4 = n & ~3;
  const real* p = prod + (size_t)(d < nd ? d : 0)*n;
  real r = 0;
  if (d < nd) {
    const long long soff = mjh_lds_offset((const void*)prod);
    r = (soff >= 0 && soff < 160*1024) ? csr_chain1(mjh_local(p), a, n4 >> 2) : csr_chain1(p, a, n4 >> 2);
  }
  const real r2 = wv_shfl(r, lane ^ 2);           // (lane a = 0 adds chain 2, lane 1 chain 3: r0 + r2, r1 + r3)
  const real s02 = r + r2;
  const real s13 = wv_shfl(s02, lane ^ 1);
  real res = s02 + s13;                            // (in lane 4d: (r0 + r2) + (r1 + r3))
  if (d < nd && a == 0 && n > n4) {
    // (mju_dot adds the sum of the remaining one to three products, engine_util_blas.c:517-525)
    real tail = p[n4];
    for (int k = n4 + 1; k < n; k++) tail += p[k];
    res += tail;
  }
  for (int q = 0; q < nd; q++) out[q] = wv_bcast(res, 4*q);
  wv_sync();
}

// SPA = 1: the reference's sparse path (mj_isSparse): compressed J / J', packed sparse factor -- mjh_sparse.h describes the
// data model; the blocks marked "sparse" below restate engine_util_sparse.c / engine_util_solve.c operation for operation
// XN = 1: the instantiation that carries Newton on the explicit-index rows (mjh_newtonx.h; SPA = 2 only) -- CG on those rows
// keeps its own instance without that code (measured: 3.4 % of the flex configuration's throughput, profiles/r05/negative_results.txt)
template <int ELL, int SPA, int XN = 0>
MJH_DEVN void solve_primal(MREF M_, BREF B_, int e_, int flg_newton) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int nv = s.nv, nmax = s.nefcmax;
  iptr counts = MJH_F(B, counts, e);
  const int nefc = counts[MJH_C_NEFC], ne = counts[MJH_C_NE], nf = counts[MJH_C_NF];
  Efc P;
  efc_layout(M, B, e, nefc, P);
  crptr J = P.J;
  crptr aref = P.aref;
  crptr qas = MJH_F(B, qacc_smooth, e);
  crptr qfs = MJH_F(B, qfrc_smooth, e);
  crptr qws = MJH_F(B, qacc_warmstart, e);
  rptr qacc = MJH_F(B, qacc, e);
  rptr qfc = MJH_F(B, qfrc_constraint, e);
  crptr Ms = MJH_G(B, M, e);              // mass matrix, CSR lower triangle (diagonal last in each row)
  const int lane = wv_lane();
#ifdef MJH_PROFILE
  // sub-stage accumulators (us): 32 set-up / warm start, 33 Hessian + factorisation, 34 factor solves, 35 incremental
  // updates, 36 line search, 37 constraint update + gradient
  long long ptick = wv_clock();
  auto tick = [&](int slot) { const long long c_ = wv_clock(); if (lane == 0) MJH_G(B, prof, e)[slot] += (real)(c_ - ptick)*0.01; ptick = c_; };
#else
  auto tick = [](int) {};
#endif
  const real tol = M.o.tolerance;
  const int elliptic = ELL ? (M.o.cone != 0) : 0;

  // ---- storage
  // factor(s): column-major n x n; in LDS when the plan's unused tail takes them (primal solvers leave
  // the dual arrays' bytes free), else their global homes
  // (sparse: compressed by pattern, mjh_sparse.h; dense: packed lower triangle, column-major -- element (row i, column k),
  // i >= k, at LX(k, i); both sit in the LDS slot efc_layout reserves for the solver when the plan has room)
  rptr Lt = P.spL;
  rptr Lc = P.spLc;                               // Lcone (elliptic)
  rptr vec = P.vec;
  auto LX = [nv](int k, int i) { return k*nv - k*(k - 1)/2 + (i - k); };
  rptr Ma = vec, grad = vec + nv, Mgrad = vec + 2*nv, search = vec + 3*nv, Mv = vec + 4*nv;
  rptr gradold = vec + 5*nv, Mgradold = vec + 6*nv, tmpv = vec + 7*nv;
  // the passes over the dofs as fused, possibly workgroup-wide functions of an argument block (mjh_csrpass.h): CG on an
  // environment-major batch with a diagonal mass matrix; per island, when it spans every dof (`fused` below).  The dof
  // vectors are then laid out for those passes: grad and search next to SIX PRODUCT vectors in the solver's block (LDS by
  // plan) -- the addends of the iteration's ordered sums, which the passes leave there --, and Ma, Mv, Mgrad, which only
  // the passes touch, in global memory.  (An island that does not span every dof takes the unfused code on the same
  // pointers; its three difference vectors use the product vectors' bytes.)
  // implicit effective metric (mj_flexCG; mjh_effmetric.h): after the warm start the solve is ONE problem over every dof
  // (mj_fwdConstraint forces the monolithic solver, engine_forward.c:1187), Ma / Mv carry + K, the preconditioner is the
  // block one, and qfrc_smooth is shifted by efm_c (PrimalAllocate, engine_solver.c:1358-1365)
  const int efm = SPA == 2 && !ELL && !flg_newton && MJH_HAS(MJH_FT_FLEX) && s.efm;
  int efm_on = 0;
  const int fuse_ok = SPA == 2 && !flg_newton && s.nC == nv && !B.soa && !efm;
  real* prodv = nullptr;
  if (fuse_ok) {
    const rptr gp = MJH_G(B, csr_prod, e);
    Ma = gp; Mv = gp + nv; Mgrad = gp + 2*nv;
    grad = vec; search = vec + nv; prodv = vec.p + 2*nv;
    gradold = vec + 2*nv; Mgradold = vec + 3*nv; tmpv = vec + 4*nv;
  }
  rptr jar = P.jar, Jv = P.ARf;
  rptr scr = MJH_G(B, scratch, e);
  rptr quad = scr;                                 // [3*nefc] (+ cone extras in the block's slots)
  rptr Dact = scr + 5*nmax;                        // D of the rows in the quadratic zone, else 0
  rptr LTJ = scr + 6*nmax;                         // [6*nv] (HessianCone)
  iptr oldstate = MJH_G(B, iscratch, e);
  rptr conH = MJH_G(B, con_H, e);

  // Newton on the explicit-index rows (mjh_newtonx.h): the factor(s) as packed lower triangles in global memory
  const int xn = XN && SPA == 2 && flg_newton;
  XnWork XW;
  if (XN && SPA == 2) {
    XW = xn_work(M, B, e);
    if (xn) { Lt = MJH_G(B, xn_L, e); Lc = MJH_G(B, xn_Lc, e); }
  }

  // ---- islands (engine_forward.c:1187-1212).  Exact when the solve is one problem over every dof; with
  // several islands, or an island that leaves trees out, dofs / rows outside the island are masked
  int nisl_raw = counts[MJH_C_NISLAND];
  int nisl = nisl_raw > 1 ? nisl_raw : 1;
  int multi_tree = (s.ntree > 1) && (nisl_raw > 0);
  ciptr tree_island = MJH_G(B, island_work, e) + s.nefcmax + s.ntree;   // left by stage_island
  int isl = 0;
  auto in_dof = [&](int i) { return !multi_tree || tree_island[M.dof_treeid[i]] == isl; };
  auto in_row = [&](int r) { return nisl_raw <= 1 || P.island[r] == isl; };

  // ---- building blocks ------------------------------------------------------------------------------------
  // mju_dot(a, b, nv) evaluated by every lane on its own (uniform result, no exchange): nv serial
  // multiply-adds in four chains
  // (sparse path: the reference's vectors are island-local -- the island's dofs in ascending order, contiguous -- so
  // mju_dot groups them by their position inside the island: ordered reduction over the island's dof mask)
  M128 isl_dofs = m128_below(nv);
  // (dense path, an island that leaves trees out: the reference's dense vectors and matrices are island-local too --
  // PrimalPointers, engine_solver.c:1148-1290 -- so every mju_dot groups its operands by their position INSIDE the island.
  // dpart marks that case; the dots below then walk the island's dof mask instead of [0, nv).)
  int dpart = 0;
  // mju_dot over the island's dofs, evaluated by one lane on its own: fn(i) = the product at dof i; elements below `lim`
  auto dot_isl = [&](int lim, auto fn) -> real {
    uint64_t lo = isl_dofs.lo, hi = isl_dofs.hi;
    if (lim < 64) { lo &= lim > 0 ? ((1ull << lim) - 1) : 0ull; hi = 0; }
    else if (lim < 128) hi &= lim > 64 ? ((1ull << (lim - 64)) - 1) : 0ull;
    const int cnt = __builtin_popcountll(lo) + __builtin_popcountll(hi);
    auto next = [&]() -> int {
      if (lo) { const int b = __builtin_ctzll(lo); lo &= lo - 1; return b; }
      const int b = __builtin_ctzll(hi); hi &= hi - 1; return 64 + b;
    };
    real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int g = cnt >> 2; g > 0; g--) {
      const int i0 = next(), i1 = next(), i2 = next(), i3 = next();
      r0 += fn(i0); r1 += fn(i1); r2 += fn(i2); r3 += fn(i3);
    }
    real res = (r0 + r2) + (r1 + r3);
    const int rem = cnt & 3;
    if (rem == 3) { const int i0 = next(), i1 = next(), i2 = next(); res += fn(i0) + fn(i1) + fn(i2); }
    else if (rem == 2) { const int i0 = next(), i1 = next(); res += fn(i0) + fn(i1); }
    else if (rem == 1) { const int i0 = next(); res += fn(i0); }
    return res;
  };
  // (SPA = 2, mjh_csr.h: the island's dofs as a list, ascending -- the reference's island-local vectors; sums take the four
  // accumulator chains of mju_dot over positions in that list, one chain per lane, several products at a time: csr_dots)
  iptr idof = MJH_G(B, csr_idof, e);
  int nidof = nv;
  // staging block of csr_dots: the unused tail of the LDS regions when it holds at least one product vector, else global
  real* dstage = nullptr; int dstage_cap = 0;
  if (SPA == 2) {
    int fb = P.free_bytes;
    // (the line search's quadratic coefficients are read row by row in every evaluation: they take the top of the free
    // tail when it has room for them next to one product vector)
    const int qb = 3*nefc*(int)sizeof(real);
    if (fb >= qb + nv*(int)sizeof(real)) { fb -= qb; quad = SP<real>{(real*)(P.free_p + fb), 1}; }
    // (qfrc_smooth is read twice per iteration -- PrimalPrepare's sums, the gradient: with room for it next to one
    // product vector the solver works on an LDS copy, and every ordered sum of the iteration is a direct one)
    // (the staging block below may then fall back to global memory: with qfrc_smooth in LDS no sum of the iteration is staged)
    // (fused passes: the sums read product vectors, never qfrc_smooth itself)
    if (!fuse_ok && !efm && fb >= nv*(int)sizeof(real)) {
      fb -= nv*(int)sizeof(real);
      real* c = (real*)(P.free_p + fb);
      for (int i0 = lane; i0 < nv; i0 += MJH_NVU*MJH_W) {
        real t[MJH_NVU];
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) t[u] = qfs[i0 + u*MJH_W < nv ? i0 + u*MJH_W : 0];
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) if (i0 + u*MJH_W < nv) c[i0 + u*MJH_W] = t[u];
      }
      qfs = SP<const real>{c, 1};
      wv_sync();
    }
    // (Newton on the explicit-index rows: the arrays its scalar walks and row sweeps go through -- elimination-tree
    // parents, visit flags, list lengths, seeds, the dense vector of a solve / update and the products of a row dot --
    // take the top of the tail when it has room for them next to one product vector; the factor stays in global memory)
    if (xn) {
      const int need = 4*nv*(int)sizeof(int) + 4*nv*(int)sizeof(real);
      if (fb >= need + nv*(int)sizeof(real)) {
        fb -= need;
        char* const q = P.free_p + fb;
        XW.x = SP<real>{(real*)q, 1}; XW.stage = XW.x + nv; XW.diag = XW.x + 2*nv; XW.y = XW.x + 3*nv;
        int* const iw = (int*)(q + 4*nv*(int)sizeof(real));
        XW.parent = SP<int>{iw, 1}; XW.flag = XW.parent + nv; XW.ltn = XW.parent + 2*nv; XW.seed = XW.parent + 3*nv;
        // (the dense vectors of a batch of rank-one updates, when there is room for them too)
        const int needb = MJH_XN_KB*nv*(int)sizeof(real);
        if (fb >= needb + nv*(int)sizeof(real)) { fb -= needb; XW.xb = SP<real>{(real*)(P.free_p + fb), 1}; }
      }
    }
    if (fb >= nv*(int)sizeof(real)) { dstage = (real*)P.free_p; dstage_cap = fb/(int)sizeof(real); }
    else if (fuse_ok) { dstage = &MJH_G(B, csr_prod, e)[0] + 3*nv; dstage_cap = 3*nv; }     // (the first three hold Ma, Mv, Mgrad)
    else { dstage = &MJH_G(B, csr_prod, e)[0]; dstage_cap = 6*nv; }
  }
  CsrPass pa;
  if (SPA == 2) {
    pa.op = 0; pa.nv = nv; pa.flag = 0; pa.pad_ = 0; pa.alpha = 0; pa.lane0 = 0; pa.width = MJH_W;
    pa.grad = grad.p; pa.search = search.p; pa.prod = prodv; pa.stage = prodv + 5*nv;
    pa.Ma = Ma.p; pa.Mv = Mv.p; pa.Mgrad = Mgrad.p; pa.qacc = qacc.p; pa.qfc = qfc.p;
    pa.qfs = qfs.p; pa.dinv = MJH_F(B, qLDiagInv, e).p; pa.Ms = Ms.p; pa.qws = qws.p; pa.qas = qas.p;
    pa.spJT = P.spJT.p; pa.force = P.force.p; pa.JTadr = P.JTadr.p; pa.JTrow = P.JTrow.p;
    pa.tree_island = tree_island.p;
  }
  auto run_pass = [&](int op, int flag, real alpha) {
    pa.op = op; pa.flag = flag; pa.alpha = alpha;
#ifdef MJH_HOSTSIM
    if (lane == 0) ::mjhsim::rc_stats()[6]++;
#endif
    MJH_HELPERS_ARGS(MJH_MWS_CSRPASS, pa, csr_pass(M, pa));
  };
  // r + stage[0] + stage[1] + ... in order (one running sum of the reference over addends a pass left in the staging vector)
  auto stage_sum = [&](real r) -> real {
    const real* st = pa.stage;
    const long long soff = mjh_lds_offset((const void*)st);
    return (soff >= 0 && soff < 160*1024) ? csr_chain_serial(mjh_local(st), nv, r) : csr_chain_serial(st, nv, r);
  };
  auto dotv = [&](crptr a, crptr b) -> real {
    if (SPA == 2) { real out[1]; const crptr xs[1] = {a}, ys[1] = {b}; csr_dots(nidof, idof, nidof == nv, 1, xs, ys, out, dstage, dstage_cap); return out[0]; }
    if (SPA) {
      const real p0 = (lane < nv) ? (real)(a[lane]*b[lane]) : (real)0;
      const real p1 = (lane + MJH_W < nv) ? (real)(a[lane + MJH_W]*b[lane + MJH_W]) : (real)0;
      return wv_dot4m(p0, p1, isl_dofs.lo, isl_dofs.hi, 1);
    }
    if (!multi_tree) return dot_ref(a, b, nv);
    if (dpart) return dot_isl(nv, [&](int i) -> real { return a[i]*b[i]; });
    real r0 = 0, r1 = 0, r2 = 0, r3 = 0;      // the island spans every dof: mju_dot over [0, nv)
    int i = 0;
    for (; i <= nv - 4; i += 4) { r0 += a[i]*b[i]; r1 += a[i+1]*b[i+1]; r2 += a[i+2]*b[i+2]; r3 += a[i+3]*b[i+3]; }
    real res = (r0 + r2) + (r1 + r3);
    // (mju_dot adds the SUM of the last one to three products: an island that spans every dof -- a flex whose vertices are
    // trees of their own -- then gets the reference's bits; engine_util_blas.c:517-525)
    const int rem = nv - i;
    if (rem == 3) res += a[i]*b[i] + a[i+1]*b[i+1] + a[i+2]*b[i+2];
    else if (rem == 2) res += a[i]*b[i] + a[i+1]*b[i+1];
    else if (rem == 1) res += a[i]*b[i];
    return res;
  };
  // out = M v in mju_mulSymVecSparse's order: diagonal, own row right to left, then the column's
  // entries by ascending row
  auto mul_M = [&](rptr out, crptr v) {
    if (SPA == 2 && s.nC == nv && !efm) {
      // (diagonal mass matrix -- every dof a slider of its own body: row t is the single entry Ms[t])
      for (int t0 = lane; t0 < nv; t0 += MJH_NVU*MJH_W) {
        real a[MJH_NVU], b[MJH_NVU];
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { const int tt = t0 + u*MJH_W < nv ? t0 + u*MJH_W : 0; a[u] = Ms[tt]; b[u] = v[tt]; }
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) if (t0 + u*MJH_W < nv) out[t0 + u*MJH_W] = a[u]*b[u];
      }
      wv_sync();
      return;
    }
    if (SPA == 2 && efm) {
#if !MJH_LANE_MODE
      eff_mul_M(M, B, e, out, v);
      if (efm_on) eff_mul_add(M, B, e, out, v);
#endif
      return;
    }
    MJH_FOR_LANES(t, nv) {
      const int adr = M.M_rowadr[t], diag = M.M_rownnz[t] - 1;
      real acc = Ms[adr + diag]*v[t];
      for (int k = diag - 1; k >= 0; k--) acc += Ms[adr + k]*v[M.M_colind[adr + k]];
      for (int q = M.M_cscadr[t]; q < M.M_cscadr[t + 1]; q++) { const int a = M.M_cscind[q]; acc += Ms[a]*v[M.M_rowid[a]]; }
      out[t] = acc;
    }
    wv_sync();
  };
  // out = J v (mju_mulMatVec: one mju_dot per row), optionally - aref
  auto mul_J = [&](rptr out, crptr v, int sub_aref) {
    MJH_FOR_LANES(r, nefc) {
      if (!SPA && dpart) {
        // (the island's rows over the island's columns: the reference's island-local dense J)
        if (!in_row(r)) continue;
        crptr Jr = J + (size_t)r*nv;
        const real acc = dot_isl(nv, [&](int i) -> real { return Jr[i]*v[i]; });
        out[r] = sub_aref ? acc - aref[r] : acc;
        continue;
      }
      const real acc = SPA == 2 ? csr_row_dot(P, r, v) : SPA ? sp_row_dot(P, r, v) : dot_ref(J + (size_t)r*nv, v, nv);
      out[r] = sub_aref ? acc - aref[r] : acc;
    }
    wv_sync();
  };
  auto is_cone_row = [&](int r) { return elliptic && r >= ne + nf && P.type[r] == MJH_CNSTR_CONTACT_ELLIPTIC; };

  // PrimalUpdateConstraint (without the cost, which the solver never reads back): force, state, cone
  // Hessians, qfrc_constraint = J' force (mju_mulMatTVec: rows added in order), ncone
  int ncone = 0;
  auto update_rows = [&]() {
    int cones = 0;
    MJH_FOR_LANES(r, nefc) {
      if (!in_row(r)) continue;
      if (is_cone_row(r)) {
        if (cone_leader(P, r)) {
          const int dim = cone_dim(P, r, nefc);
          cone_update(P, r, dim, jar, conH + 36*P.id[r], flg_newton);
          if (P.state[r] == MJH_STATE_CONE) cones += dim;
        }
        continue;
      }
      real f = -P.D[r]*jar[r];
      int st;
      if (r < ne) st = MJH_STATE_QUADRATIC;
      else if (r < ne + nf) {
        if (jar[r] <= -P.R[r]*P.floss[r]) { f = P.floss[r]; st = MJH_STATE_LINEARNEG; }
        else if (jar[r] >= P.R[r]*P.floss[r]) { f = -P.floss[r]; st = MJH_STATE_LINEARPOS; }
        else st = MJH_STATE_QUADRATIC;
      } else {
        if (jar[r] >= 0) { f = 0; st = MJH_STATE_SATISFIED; }
        else st = MJH_STATE_QUADRATIC;
      }
      P.force[r] = f;
      P.state[r] = st;
    }
    ncone = ELL ? wv_sum_i(cones) : 0;
    wv_sync();
  };
  auto update_constraint = [&]() {
    update_rows();
    if (SPA) {
      // sparse: mju_mulMatVecSparse(J', force) -- one mju_dotSparse per dof over the rows that contain it
      // (SPA = 2: lanes over the island's dof list instead of testing every dof of the model)
      if (SPA == 2) {
        // (lanes over the island's dof list, four rounds of address loads in flight; most dofs of a flex hold no row at all)
        for (int k0 = lane; k0 < nidof; k0 += MJH_NVU*MJH_W) {
          int jj[MJH_NVU], b0[MJH_NVU], b1[MJH_NVU];
#pragma unroll
          for (int u = 0; u < MJH_NVU; u++) { const int k = k0 + u*MJH_W; jj[u] = k < nidof ? (nidof == nv ? k : (int)idof[k]) : -1; }
#pragma unroll
          for (int u = 0; u < MJH_NVU; u++) { b0[u] = jj[u] >= 0 ? (int)P.JTadr[jj[u]] : 0; b1[u] = jj[u] >= 0 ? (int)P.JTadr[jj[u] + 1] : 0; }
#pragma unroll
          for (int u = 0; u < MJH_NVU; u++) {
            if (jj[u] < 0) continue;
            const int a0 = b0[u], n = b1[u] - a0;
            crptr v = P.spJT + a0;
            ciptr ri = P.JTrow + a0;
            real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            int k = 0;
            for (; k <= n - 4; k += 4) {
              r0 += v[k]*P.force[ri[k]]; r1 += v[k + 1]*P.force[ri[k + 1]];
              r2 += v[k + 2]*P.force[ri[k + 2]]; r3 += v[k + 3]*P.force[ri[k + 3]];
            }
            real res = (r0 + r2) + (r1 + r3);
            for (; k < n; k++) res += v[k]*P.force[ri[k]];
            qfc[jj[u]] = res;
            // (the gradient Ma - qfrc_smooth - qfrc_constraint is formed here, from the value in the register: update_grad
            // follows every call, and its pass would read qfrc_constraint back from global memory)
            grad[jj[u]] = Ma[jj[u]] - qfs[jj[u]] - res;
          }
        }
        wv_sync();
        return;
      }
      MJH_FOR_LANES(j, nv) {
        if (!in_dof(j)) continue;
        const int a0 = P.JTadr[j], n = P.JTadr[j + 1] - a0;
        crptr v = P.spJT + a0;
        ciptr ri = P.JTrow + a0;
        real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        int k = 0;
        for (; k <= n - 4; k += 4) {
          r0 += v[k]*P.force[ri[k]]; r1 += v[k + 1]*P.force[ri[k + 1]];
          r2 += v[k + 2]*P.force[ri[k + 2]]; r3 += v[k + 3]*P.force[ri[k + 3]];
        }
        real res = (r0 + r2) + (r1 + r3);
        for (; k < n; k++) res += v[k]*P.force[ri[k]];
        qfc[j] = res;
      }
      wv_sync();
      return;
    }
    MJH_FOR_LANES(j, nv) {
      real acc = 0;
      for (int r = 0; r < nefc; r++) if (in_row(r)) acc += J[(size_t)r*nv + j]*P.force[r];
      if (in_dof(j)) qfc[j] = acc;
    }
    wv_sync();
  };
  // grad = Ma - qfrc_smooth - qfrc_constraint (PrimalUpdateGrad)
  auto update_grad = [&]() {
    if (SPA == 2) return;          // (formed by update_constraint's pass over the island's dofs)
    if (SPA == 2) {
      // (dofs outside the island: zeroed once when the island starts, never written afterwards)
      for (int k0 = lane; k0 < nidof; k0 += MJH_NVU*MJH_W) {
        int jj[MJH_NVU]; real a[MJH_NVU], b[MJH_NVU], c[MJH_NVU];
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { const int k = k0 + u*MJH_W; jj[u] = k < nidof ? (nidof == nv ? k : (int)idof[k]) : 0; }
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { a[u] = Ma[jj[u]]; b[u] = qfs[jj[u]]; c[u] = qfc[jj[u]]; }
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) if (k0 + u*MJH_W < nidof) grad[jj[u]] = a[u] - b[u] - c[u];
      }
      wv_sync();
      return;
    }
    MJH_FOR_LANES(j, nv) grad[j] = in_dof(j) ? (Ma[j] - qfs[j] - qfc[j]) : (real)0;
    wv_sync();
  };
  // Mgrad = M \ grad (CG preconditioner; Newton's convergence certificate)
  auto precondition = [&]() {
    if (SPA == 2 && efm && efm_on) {
#if !MJH_LANE_MODE
      eff_block_apply(M, B, e, Mgrad, grad, MJH_G(B, efm_work, e));      // mjd_effPrec
#endif
      return;
    }
    if (SPA == 2 && s.nC == nv) {
      // (diagonal mass matrix: mj_solveLD reduces to x * qLDiagInv)
      crptr dinv = MJH_F(B, qLDiagInv, e);
      for (int t0 = lane; t0 < nv; t0 += MJH_NVU*MJH_W) {
        real a[MJH_NVU], b[MJH_NVU];
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { const int tt = t0 + u*MJH_W < nv ? t0 + u*MJH_W : 0; a[u] = grad[tt]; b[u] = dinv[tt]; }
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) if (t0 + u*MJH_W < nv) Mgrad[t0 + u*MJH_W] = a[u]*b[u];
      }
      wv_sync();
      return;
    }
    MJH_FOR_LANES(i, nv) Mgrad[i] = grad[i];
    wv_sync();
    solve_ld(M, Mgrad, MJH_F(B, qLD, e), MJH_F(B, qLDiagInv, e));
  };

  // ---- dense Cholesky machinery (Newton) -------------------------------------------------------------------
  // element idx of a dof vector held in registers: slot idx / 64 of lane idx % 64
  auto dof_get = [&](real y0, real y1, int idx) -> real { return idx < MJH_W ? wv_bcast(y0, idx) : wv_bcast(y1, idx - MJH_W); };
  // mju_dot over elements [0, cnt) whose products the lanes hold (p0: idx = lane, p1: idx = lane + 64)
  auto dot_lanes = [&](real p0, real p1, int cnt) -> real {
    real r[4] = {0, 0, 0, 0};
    const int G = cnt >> 2;
    const int g0 = G < 16 ? G : 16;
    wv_dot4_acc(r, p0, g0);
    if (G > 16) wv_dot4_acc(r, p1, G - 16);
    real res = (r[0] + r[2]) + (r[1] + r[3]);
    const int i = 4*G, rem = cnt - i;
    if (rem == 3) res += dof_get(p0, p1, i) + dof_get(p0, p1, i + 1) + dof_get(p0, p1, i + 2);
    else if (rem == 2) res += dof_get(p0, p1, i) + dof_get(p0, p1, i + 1);
    else if (rem == 1) res += dof_get(p0, p1, i);
    return res;
  };

  // MakeHessian, dense (mju_sqrMatTD_impl with diag = Dact, then + M): entry (i, k), k <= i, is
  // sum_j J[j][k]*(J[j][i]*Dact[j]) over the rows j in order, skipping rows with Dact[j] = 0 or J[j][i] = 0
  auto make_hessian = [&](rptr L) {
    MJH_FOR_LANES(r, nefc) Dact[r] = (in_row(r) && P.state[r] == MJH_STATE_QUADRATIC) ? (real)P.D[r] : (real)0;
    wv_sync();
    // lanes over the columns k of one row i at a time would idle for short rows: flatten the lower
    // triangle instead (w -> (i, k)), row-major so that consecutive lanes read consecutive J columns
    const int ntri = nv*(nv + 1)/2;
    MJH_FOR_LANES(w, ntri) {
      int i = (int)((sqrt(8.0*w + 1.0) - 1.0)*0.5);
      while (i*(i + 1)/2 > w) i--;
      while ((i + 1)*(i + 2)/2 <= w) i++;
      const int k = w - i*(i + 1)/2;
      real acc = 0;
      for (int j = 0; j < nefc; j++) {
        const real dj = Dact[j];
        if (dj == 0) continue;
        const real tmp = J[(size_t)j*nv + i];
        if (tmp == 0) continue;
        acc += J[(size_t)j*nv + k]*(tmp*dj);
      }
      L[LX(k, i)] = acc;
    }
    wv_sync();
    // mju_addToSymSparse: + M on the lower triangle
    MJH_FOR_LANES(a, s.nC) { const int i = M.M_rowid[a], k = M.M_colind[a]; L[LX(k, i)] += Ms[a]; }
    wv_sync();
  };

  // mju_cholFactor(L, nv, mjMINVAL) in place on the column-major factor: column j needs, for every row
  // i >= j, the dot of rows i and j over the finished columns [0, j).  Rows lane and lane + 64.
  const int two = nv > MJH_W;                    // dof vectors occupy a second register slot
  auto chol_factor = [&](rptr L) {
    // mju_dot over the finished columns c < j of rows i and j
    auto rowdot = [&](int i, int j) -> real {
      // (column of an island that leaves trees out: the finished columns of the ISLAND, grouped by their position in it;
      //  rows and columns of other trees hold exact zeros against the island's and are factorised in the global order)
      if (dpart && m128_test(isl_dofs, j)) return dot_isl(j, [&](int c) -> real { return L[LX(c, i)]*L[LX(c, j)]; });
      real r0 = 0, r1 = 0, r2 = 0, r3 = 0;
      int c = 0;
      for (; c <= j - 4; c += 4) {
        r0 += L[LX(c, i)]*L[LX(c, j)]; r1 += L[LX(c + 1, i)]*L[LX(c + 1, j)];
        r2 += L[LX(c + 2, i)]*L[LX(c + 2, j)]; r3 += L[LX(c + 3, i)]*L[LX(c + 3, j)];
      }
      real res = (r0 + r2) + (r1 + r3);
      const int rem = j - c;
      if (rem == 3) res += L[LX(c, i)]*L[LX(c, j)] + L[LX(c + 1, i)]*L[LX(c + 1, j)] + L[LX(c + 2, i)]*L[LX(c + 2, j)];
      else if (rem == 2) res += L[LX(c, i)]*L[LX(c, j)] + L[LX(c + 1, i)]*L[LX(c + 1, j)];
      else if (rem == 1) res += L[LX(c, i)]*L[LX(c, j)];
      return res;
    };
    for (int j = 0; j < nv; j++) {
      real d0 = 0, d1 = 0;
      if (j > 0) {
        if (lane >= j && lane < nv) d0 = rowdot(lane, j);
        if (two && lane + MJH_W >= j && lane + MJH_W < nv) d1 = rowdot(lane + MJH_W, j);
      }
      real tmp = L[LX(j, j)];
      if (j) tmp -= dof_get(d0, d1, j);
      const int deficient = tmp < MJH_MINVAL;
      if (deficient) tmp = MJH_MINVAL;
      const real djj = sqrt(tmp);
      const real inv = 1/djj;
      wv_sync();
      if (lane == j) L[LX(j, j)] = djj;
      else if (lane > j && lane < nv) L[LX(j, lane)] = deficient ? (real)0 : (L[LX(j, lane)] - d0)*inv;
      if (two) {
        const int i = lane + MJH_W;
        if (i == j) L[LX(j, j)] = djj;
        else if (i > j && i < nv) L[LX(j, i)] = deficient ? (real)0 : (L[LX(j, i)] - d1)*inv;
      }
      wv_sync();
    }
  };

  // mju_cholSolve(Mgrad, L, grad, nv): forward substitution by row dots, back substitution by sequential
  // subtraction; the vector stays in registers (y0: dof lane, y1: dof lane + 64)
  auto chol_solve = [&](crptr L) {
    if (!two) {
      real y = lane < nv ? (real)grad[lane] : (real)0;
      if (dpart) y = mjh_in_lds(L) ? dn_solve_isl(mjh_local(L.p), nv, y, isl_dofs.lo) : dn_solve_isl(L, nv, y, isl_dofs.lo);
      else
      y = mjh_in_lds(L) ? dn_solve(mjh_local(L.p), nv, y) : dn_solve(L, nv, y);
      if (lane < nv) Mgrad[lane] = y;
      wv_sync();
      return;
    }
    real y0 = lane < nv ? (real)grad[lane] : (real)0;
    real y1 = (two && lane + MJH_W < nv) ? (real)grad[lane + MJH_W] : (real)0;
    for (int i = 0; i < nv; i++) {
      const real p0 = lane < i ? L[LX(lane, i)]*y0 : (real)0;
      const real p1 = (two && lane + MJH_W < i) ? L[LX((lane + MJH_W), i)]*y1 : (real)0;
      real yi = dof_get(y0, y1, i);
      if (dpart) {
        if (!m128_test(isl_dofs, i)) continue;       // (a dof outside the island: its gradient entry is zero and stays zero)
        const uint64_t blo = i < 64 ? (i ? ((1ull << i) - 1) : 0ull) : ~0ull;
        const uint64_t bhi = i > 64 ? ((1ull << (i - 64)) - 1) : 0ull;
        yi -= wv_dot4m(p0, p1, isl_dofs.lo & blo, isl_dofs.hi & bhi, 1);
      } else
      if (i) yi -= dot_lanes(p0, p1, i);
      yi /= L[LX(i, i)];
      if (i < MJH_W) { if (lane == i) y0 = yi; } else { if (lane == i - MJH_W) y1 = yi; }
    }
    for (int i = nv - 1; i >= 0; i--) {
      const real p0 = (lane > i && lane < nv) ? L[LX(i, lane)]*y0 : (real)0;
      const real p1 = (two && lane + MJH_W > i && lane + MJH_W < nv) ? L[LX(i, lane + MJH_W)]*y1 : (real)0;
      real yi = dof_get(y0, y1, i);
      // res[i] -= L[j][i]*res[j], j = i+1 .. n-1 in order
      if (i + 1 < MJH_W) yi = wv_chain(yi, p0, i + 1, nv < MJH_W ? nv : MJH_W, 1);
      if (two) yi = wv_chain(yi, p1, i + 1 > MJH_W ? i + 1 - MJH_W : 0, nv - MJH_W, 1);
      yi /= L[LX(i, i)];
      if (i < MJH_W) { if (lane == i) y0 = yi; } else { if (lane == i - MJH_W) y1 = yi; }
    }
    if (lane < nv) Mgrad[lane] = y0;
    if (two && lane + MJH_W < nv) Mgrad[lane + MJH_W] = y1;
    wv_sync();
  };

  // mju_cholUpdate(L, x, nv, flg_plus): x in registers (x0: dof lane, x1: dof lane + 64); returns the rank
  auto chol_update = [&](rptr L, real x0, real x1, int flg_plus) -> int {
    if (!two) return nv - (mjh_in_lds(L) ? dn_update(mjh_local(L.p), nv, x0, flg_plus) : dn_update(L, nv, x0, flg_plus));
    int rank = nv;
    for (int k = 0; k < nv; k++) {
      const real xk = dof_get(x0, x1, k);
      if (xk == 0) continue;
      const real Lkk = L[LX(k, k)];
      real tmp = Lkk*Lkk + (flg_plus ? xk*xk : -xk*xk);
      if (tmp < MJH_MINVAL) { tmp = MJH_MINVAL; rank--; }
      const real r = sqrt(tmp);
      const real c = r/Lkk;
      const real cinv = 1/c;
      const real sx = xk/Lkk;
      wv_sync();                                  // (every lane has read L[k][k] before its owner overwrites it)
      if (lane == k) L[LX(k, k)] = r;
      else if (lane > k && lane < nv) {
        const real lik = flg_plus ? (L[LX(k, lane)] + sx*x0)*cinv : (L[LX(k, lane)] - sx*x0)*cinv;
        L[LX(k, lane)] = lik;
        x0 = c*x0 - sx*lik;
      }
      if (two) {
        const int i = lane + MJH_W;
        if (i == k) L[LX(k, k)] = r;
        else if (i > k && i < nv) {
          const real lik = flg_plus ? (L[LX(k, i)] + sx*x1)*cinv : (L[LX(k, i)] - sx*x1)*cinv;
          L[LX(k, i)] = lik;
          x1 = c*x1 - sx*lik;
        }
      }
    }
    wv_sync();
    return rank;
  };


  // ---- sparse Cholesky machinery (Newton on the reference's sparse path): mjh_sparse.h ------------------------
  // the factor is stored compressed by its pattern; it works in the LDS slot when its fill fits, else in its global home
  int sp_nL = 0;
  auto sp_factorize = [&]() {
    MJH_FOR_LANES(r, nefc) Dact[r] = (in_row(r) && P.state[r] == MJH_STATE_QUADRATIC) ? (real)P.D[r] : (real)0;
    wv_sync();
    sp_nL = sp_symbolic(M, B, e, P, isl_dofs);
    Lt = sp_nL <= P.spL_cap ? P.spL : P.spL_home;
    if (mjh_in_lds(Lt)) sp_numeric(M, B, e, P, mjh_local(Lt.p), isl_dofs, Dact, Ms);
    else sp_numeric(M, B, e, P, Lt, isl_dofs, Dact, Ms);
  };
  auto sp_chol_solve = [&](rptr L) {
    real y0 = (lane < nv && m128_test(isl_dofs, lane)) ? (real)grad[lane] : (real)0;
    real y1 = (lane + MJH_W < nv && m128_test(isl_dofs, lane + MJH_W)) ? (real)grad[lane + MJH_W] : (real)0;
    const SpVec2 y = mjh_in_lds(L) ? sp_solve(M, P, mjh_local(L.p), isl_dofs, y0, y1) : sp_solve(M, P, L, isl_dofs, y0, y1);
    if (lane < nv) Mgrad[lane] = y.y0;
    if (lane + MJH_W < nv) Mgrad[lane + MJH_W] = y.y1;
    wv_sync();
  };
  // returns the rank, like mju_cholUpdateSparse
  auto sp_chol_update = [&](rptr L, real x0, real x1, M128 xm, int flg_plus) -> int {
    const int clamped = mjh_in_lds(L) ? sp_update(M, P, mjh_local(L.p), x0, x1, xm, flg_plus) : sp_update(M, P, L, x0, x1, xm, flg_plus);
    return nv - clamped;
  };
  // row i of J, scaled, spread over the dof lanes
  auto sp_row_lanes = [&](int i, real scl, real& x0, real& x1, M128& pm) {
    pm = m128_ld(P.rowmask + 4*i);
    const int adr = P.rowadr[i];
    x0 = (lane < nv && m128_test(pm, lane)) ? (real)(P.spJ[adr + m128_rank(pm, lane)]*scl) : (real)0;
    x1 = (lane + MJH_W < nv && m128_test(pm, lane + MJH_W)) ? (real)(P.spJ[adr + m128_rank(pm, lane + MJH_W)]*scl) : (real)0;
  };
  // explicit-index Newton: rank-one updates are queued (their dense vectors side by side) and applied MJH_XN_KB at a time in
  // one sweep over the factor (xn_update_batch); a clamped pivot anywhere means the reference refactorises from scratch
  int xq_n = 0, xq_plus = 0, xq_start = -1;
  auto xq_flush = [&](rptr L) -> int {
    if (!xq_n) return 0;
    const int clamped = xn_update_batch(XW, L, xq_n, xq_plus, xq_start);
    xq_n = 0; xq_plus = 0; xq_start = -1;
    return clamped;
  };
  // entry q of the vector = valfn(q) at column cols[q], q < m (ascending columns); returns the clamped count of a flush
  auto xq_push = [&](rptr L, ciptr cols, int m, auto valfn, int plus) -> int {
    if (m <= 0) return 0;
    const rptr xv = XW.xb + (long long)xq_n*nv;
    MJH_FOR_LANES(j, nv) xv[j] = 0;
    wv_sync();
    MJH_FOR_LANES(q, m) xv[cols[q]] = valfn(q);
    wv_sync();
    if (plus) xq_plus |= 1 << xq_n;
    const int last = cols[m - 1];
    if (last > xq_start) xq_start = last;
    xq_n++;
    return xq_n == MJH_XN_KB ? xq_flush(L) : 0;
  };
  // HessianCone: Lcone = L, then one rank-one update per row of L_local' J of every contact in the cone zone
  auto hessian_cone = [&]() {
    if (XN && SPA == 2) {
      // (only the island's rows of the packed triangle are in use)
      for (int k = 0; k < nidof; k++) { const int r = idof[k]; const long long a = xn_row(r); for (int j = lane; j <= r; j += MJH_W) Lc[a + j] = Lt[a + j]; }
    } else
    MJH_FOR_LANES(w, SPA ? sp_nL : nv*(nv + 1)/2) Lc[w] = Lt[w];
    wv_sync();
    for (int i = 0; i < nefc; i++) {
      if (!in_row(i) || P.state[i] != MJH_STATE_CONE) continue;
      const int dim = cone_dim(P, i, nefc);
      real local[36];
      crptr Hc = conH + 36*P.id[i];
      for (int q = 0; q < dim*dim; q++) local[q] = Hc[q];
      // mju_cholFactor(local, dim, mjMINVAL) (row-major, tiny: every lane does it)
      for (int j = 0; j < dim; j++) {
        real tmp = local[j*(dim + 1)];
        if (j) tmp -= dot_ref(local + j*dim, local + j*dim, j);
        const int deficient = tmp < MJH_MINVAL;
        if (deficient) tmp = MJH_MINVAL;
        local[j*(dim + 1)] = sqrt(tmp);
        if (deficient) { for (int r = j + 1; r < dim; r++) local[r*dim + j] = 0; }
        else {
          tmp = 1/local[j*(dim + 1)];
          for (int r = j + 1; r < dim; r++) local[r*dim + j] = (local[r*dim + j] - dot_ref(local + r*dim, local + j*dim, j))*tmp;
        }
      }
      if (XN && SPA == 2) {
        // explicit-index rows: LTJ over the contact's shared columns (HessianCone :2245-2262: row c of LTJ accumulates
        // J[i + r] * local[r][c] over r = c .. dim-1 in order, from zero), one mju_cholUpdateSparse per row of LTJ
        const int a0 = P.rowadr[i], m = P.rowadr[i + 1] - a0;
        for (int c = 0; c < dim; c++)
          xq_push(Lc, P.colind + a0, m, [&](int q) -> real {
            real acc = 0;
            for (int r = c; r < dim; r++) acc += P.spJ[P.rowadr[i + r] + q]*local[r*dim + c];
            return acc;
          }, 1);
        i += dim - 1;
        continue;
      }
      if (SPA) {
        // sparse: LTJ over the contact's shared pattern, one mju_cholUpdateSparse per column of L_local
        const M128 pm = m128_ld(P.rowmask + 4*i);
        const int k0 = m128_rank(pm, lane), k1 = m128_rank(pm, lane + MJH_W);
        const int in0 = lane < nv && m128_test(pm, lane), in1 = lane + MJH_W < nv && m128_test(pm, lane + MJH_W);
        for (int c = 0; c < dim; c++) {
          real x0 = 0, x1 = 0;
          for (int r = c; r < dim; r++) {
            const int adr = P.rowadr[i + r];
            if (in0) x0 += P.spJ[adr + k0]*local[r*dim + c];
            if (in1) x1 += P.spJ[adr + k1]*local[r*dim + c];
          }
          sp_chol_update(Lc, x0, x1, pm, 1);
        }
        i += dim - 1;
        continue;
      }
      // LTJ[c] = sum_{r >= c} J[i+r] * local[r][c], rows added in order of r
      MJH_FOR_LANES(j, nv) {
        for (int c = 0; c < dim; c++) {
          real acc = 0;
          for (int r = c; r < dim; r++) acc += J[(size_t)(i + r)*nv + j]*local[r*dim + c];
          LTJ[c*nv + j] = acc;
        }
      }
      wv_sync();
      for (int r = 0; r < dim; r++) {
        const real x0 = lane < nv ? (real)LTJ[r*nv + lane] : (real)0;
        const real x1 = (two && lane + MJH_W < nv) ? (real)LTJ[r*nv + lane + MJH_W] : (real)0;
        chol_update(Lc, x0, x1, 1);
      }
      i += dim - 1;
    }
    if (XN && SPA == 2) xq_flush(Lc);
  };
  // FactorizeHessian
  auto factorize = [&](int recompute) {
    if (XN && SPA == 2) {
      MJH_FOR_LANES(r, nefc) Dact[r] = (in_row(r) && P.state[r] == MJH_STATE_QUADRATIC) ? (real)P.D[r] : (real)0;
      wv_sync();
      xn_factorize(M, P, XW, Lt, idof, nidof, nefc, Dact, Ms, nisl_raw > 1 ? isl : -1);
      if (ELL && ncone) hessian_cone();
      return;
    }
    if (SPA) {
      sp_factorize();
      if (ELL && ncone) hessian_cone();
      return;
    }
    if (recompute) make_hessian(Lt);
    chol_factor(Lt);
    if (ELL && ncone) hessian_cone();
  };
  // HessianIncremental
  auto hessian_incremental = [&]() {
    if (XN && SPA == 2) {
      // one mju_cholUpdateSparse per row that entered or left the quadratic zone, in row order, with J[i] * sqrt(D[i]) --
      // queued and applied MJH_XN_KB at a time
      for (int i = 0; i < nefc; i++) {
        if (!in_row(i)) continue;
        const int was = oldstate[i] == MJH_STATE_QUADRATIC, is = P.state[i] == MJH_STATE_QUADRATIC;
        if (was == is) continue;
        const int a0 = P.rowadr[i], m = P.rowadr[i + 1] - a0;
        const real sq = sqrt(P.D[i]);
        if (xq_push(Lt, P.colind + a0, m, [&](int q) -> real { return P.spJ[a0 + q]*sq; }, is ? 1 : 0)) { xq_flush(Lt); factorize(1); return; }
      }
      if (xq_flush(Lt)) { factorize(1); return; }
      if (ELL && ncone) hessian_cone();
      return;
    }
    if (SPA) {
      // the rows that entered or left the quadratic zone, in order, MJH_SP_KB at a time through one sweep over the
      // factor (sp_update_batch); a clamped pivot anywhere means the reference refactorises from scratch
      SpBatch bt;
      bt.n = 0; bt.plus = 0;
      for (int u = 0; u < MJH_SP_KB/2; u++) bt.rows[u] = 0;
      auto flush = [&]() -> int {
        if (!bt.n) return 0;
        const int clamped = mjh_in_lds(Lt) ? sp_update_batch(M, P, mjh_local(Lt.p), bt) : sp_update_batch(M, P, Lt, bt);
        bt.n = 0; bt.plus = 0;
        for (int u = 0; u < MJH_SP_KB/2; u++) bt.rows[u] = 0;
        return clamped;
      };
      for (int r0 = 0; r0 < nefc; r0 += MJH_W) {
        const int r = r0 + lane;
        int changed = 0;
        if (r < nefc && in_row(r)) changed = (oldstate[r] == MJH_STATE_QUADRATIC) != (P.state[r] == MJH_STATE_QUADRATIC);
        for (unsigned long long chg = wv_ballot(changed); chg; chg &= chg - 1) {
          const int i = r0 + __builtin_ctzll(chg);
          const int u = bt.n;
#pragma unroll
          for (int q = 0; q < MJH_SP_KB/2; q++) if (q == (u >> 1)) bt.rows[q] |= i << (16*(u & 1));
          if (P.state[i] == MJH_STATE_QUADRATIC) bt.plus |= 1 << u;
          bt.n = u + 1;
          if (bt.n == MJH_SP_KB && flush()) { factorize(1); return; }
        }
      }
      if (flush()) { factorize(1); return; }
      if (ELL && ncone) hessian_cone();
      return;
    }
    for (int i = 0; i < nefc; i++) {
      if (!in_row(i)) continue;
      const int was = oldstate[i] == MJH_STATE_QUADRATIC, is = P.state[i] == MJH_STATE_QUADRATIC;
      if (was == is) continue;
      const real sq = sqrt(P.D[i]);
      const real x0 = lane < nv ? J[(size_t)i*nv + lane]*sq : (real)0;
      const real x1 = (two && lane + MJH_W < nv) ? J[(size_t)i*nv + lane + MJH_W]*sq : (real)0;
      const int rank = chol_update(Lt, x0, x1, is ? 1 : 0);
      if (rank < nv) { factorize(1); return; }
    }
    if (ELL && ncone) hessian_cone();
  };
  auto newton_mgrad = [&]() {
    if (XN && SPA == 2) { xn_solve(XW, (ELL && ncone) ? (crptr)Lc : (crptr)Lt, idof, nidof, grad, Mgrad); return; }
    if (SPA) sp_chol_solve((ELL && ncone) ? Lc : Lt);
    else chol_solve((ELL && ncone) ? (crptr)Lc : (crptr)Lt);
  };

  // ---- warm start: best of (qacc_warmstart, qacc_smooth)        (warmstart, engine_forward.c:1056-1132)
  // (jar = J qacc_warmstart - aref and efc_b = J qacc_smooth - aref were left by stage_fwd_constraint)
  const int trace_scale = !(M.o.disableflags & (1<<18)) && nisl_raw > 0 && !efm;
  if (fuse_ok) {
    int use_smooth = 1;
    if (!(M.o.disableflags & (1<<9))) {
      run_pass(CSR_OP_WARM, 0, 0);
      real cost_ws = constraint_update(B, e, P, jar, 1, elliptic);
      cost_ws = stage_sum(cost_ws);
      wv_sync();
      const real cost_smooth = constraint_update(B, e, P, P.b, 1, elliptic);
      use_smooth = cost_ws > cost_smooth;
    }
    run_pass(CSR_OP_START, use_smooth | (multi_tree ? 2 : 0) | (trace_scale ? 4 : 0), 0);
  } else
  if (!(M.o.disableflags & (1<<9))) {
    mul_M(Ma, qws);
    real cost_ws = constraint_update(B, e, P, jar, 1, elliptic);
    if (SPA == 2) {
      // (the addends lane-parallel into the staging block, then the reference's single running sum over them)
      for (int i0 = lane; i0 < nv; i0 += MJH_NVU*MJH_W) {
        real a[MJH_NVU], b[MJH_NVU], c[MJH_NVU], d[MJH_NVU];
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { const int i = i0 + u*MJH_W < nv ? i0 + u*MJH_W : 0; a[u] = Ma[i]; b[u] = qfs[i]; c[u] = qws[i]; d[u] = qas[i]; }
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { const int i = i0 + u*MJH_W; if (i < nv) dstage[i] = 0.5*(a[u] - b[u])*(c[u] - d[u]); }
      }
      wv_sync();
      {
        const long long soff = mjh_lds_offset((const void*)dstage);
        cost_ws = (soff >= 0 && soff < 160*1024) ? csr_chain_serial(mjh_local((const real*)dstage), nv, cost_ws)
                                                  : csr_chain_serial((const real*)dstage, nv, cost_ws);
      }
    } else
    for (int i = 0; i < nv; i++) cost_ws += 0.5*(Ma[i] - qfs[i])*(qws[i] - qas[i]);
    wv_sync();
    const real cost_smooth = constraint_update(B, e, P, P.b, 1, elliptic);
    const int use_smooth = cost_ws > cost_smooth;
    if (SPA == 2) {
      for (int i0 = lane; i0 < nv; i0 += MJH_NVU*MJH_W) {
        real c[MJH_NVU];
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { const int i = i0 + u*MJH_W < nv ? i0 + u*MJH_W : 0; c[u] = use_smooth ? qas[i] : qws[i]; }
#pragma unroll
        for (int u = 0; u < MJH_NVU; u++) { const int i = i0 + u*MJH_W; if (i < nv) qacc[i] = c[u]; }
      }
    } else
    MJH_FOR_LANES(i, nv) qacc[i] = use_smooth ? qas[i] : qws[i];
    wv_sync();
    if (multi_tree) {
      // dofs of unconstrained trees start (and stay) at qacc_smooth
      MJH_FOR_LANES(i, nv) if (tree_island[M.dof_treeid[i]] < 0) qacc[i] = qas[i];
      wv_sync();
    }
  } else {
    MJH_FOR_LANES(i, nv) qacc[i] = qas[i];
    wv_sync();
  }

  // ---- mj_solPrimal -------------------------------------------------------------------------------------------
  if (efm) {
    // from here on the effective metric: monolithic, shifted smooth force (the warm start above used M and qfrc_smooth)
    efm_on = 1;
    nisl_raw = 0; nisl = 1; multi_tree = 0;
    rptr qeff = MJH_G(B, efm_work, e) + 5*nv;
    crptr cs = MJH_G(B, efm_c, e);
    crptr qf0 = MJH_F(B, qfrc_smooth, e);
    MJH_FOR_LANES(i, nv) qeff[i] = qf0[i] + cs[i];
    wv_sync();
    qfs = qeff;
  }
  if (!fuse_ok) mul_M(Ma, qacc);
  mul_J(jar, qacc, 1);
  if (multi_tree) { MJH_FOR_LANES(j, nv) qfc[j] = 0; wv_sync(); }
  int niter0 = 0;
  tick(32);
  for (isl = 0; isl < nisl; isl++) {
    int fused = 0;
    if (fuse_ok) {
      // does the island span every dof?  (every tree belongs to it; without islands the solve is one problem anyway)
      int miss = 0;
      if (multi_tree) MJH_FOR_LANES(t, s.ntree) miss |= tree_island[t] != isl;
      fused = !wv_any(miss);
      if (fused) nidof = nv;
    }
    if (SPA == 2 && !fused) {
      nidof = 0;
      for (int j0 = 0; j0 < nv; j0 += MJH_W) {
        const int j = j0 + lane;
        const int in = j < nv && in_dof(j);
        const unsigned long long m = wv_ballot(in);
        if (in) idof[nidof + wv_rank_lt(m)] = j;
        nidof += __builtin_popcountll(m);
      }
      MJH_FOR_LANES(j, nv) grad[j] = 0;
      wv_sync();
    } else if (SPA) {
      isl_dofs.lo = wv_ballot(lane < nv && in_dof(lane));
      isl_dofs.hi = wv_ballot(lane + MJH_W < nv && in_dof(lane + MJH_W));
    } else if (multi_tree) {
      // dense path: does this island leave dofs out?  Then its sums run over the island-local vectors of the reference
      isl_dofs.lo = wv_ballot(lane < nv && in_dof(lane));
      isl_dofs.hi = wv_ballot(lane + MJH_W < nv && in_dof(lane + MJH_W));
      dpart = __builtin_popcountll(isl_dofs.lo) + __builtin_popcountll(isl_dofs.hi) != nv;
      if (dpart) mul_J(jar, qacc, 1);          // (Jaref of the island's rows: J qacc in the island's own grouping)
    }
    if (fused) { update_rows(); run_pass(CSR_OP_GRAD, 0, 0); }
    else { update_constraint(); update_grad(); }
    tick(37);

    // scale: 1/(meaninertia*nv) monolithic, 1/trace(M over the island's dofs) per island
    real scale;
    if (trace_scale) {
      real tr = 0;
      if (fused) {
        // (the diagonal of M, left in the staging vector by CSR_OP_START, summed in order; the staging vector of a second
        // island of this kind would have been overwritten -- there is only one island that spans every dof)
        tr = stage_sum(0);
      } else
      if (SPA == 2) {
        MJH_FOR_LANES(k, nidof) { const int i = idof[k]; dstage[k] = Ms[M.M_rowadr[i] + M.M_rownnz[i] - 1]; }
        wv_sync();
        for (int k = 0; k < nidof; k++) tr += dstage[k];
        wv_sync();
      } else
      for (int i = 0; i < nv; i++) if (in_dof(i)) tr += Ms[M.M_rowadr[i] + M.M_rownnz[i] - 1];
      scale = 1/tr;
    } else {
      scale = 1/(M.o.meaninertia*(real)(nv > 1 ? nv : 1));
    }

    // convergence certificate with M^-1
    if (!fused) precondition();
    real gm_gg[2];
    if (fused) csr_sums(nv, 2, prodv, gm_gg);
    else if (SPA == 2) { const crptr xs[2] = {grad, grad}, ys[2] = {Mgrad, grad}; csr_dots(nidof, idof, nidof == nv, 2, xs, ys, gm_gg, dstage, dstage_cap); }
    else { gm_gg[0] = dotv(grad, Mgrad); gm_gg[1] = dotv(grad, grad); }
    const int flg_gap = r_max(0, 0.5*scale*gm_gg[0]) < tol;
    const int flg_gradient = scale*sqrt(gm_gg[1]) < tol;
    int done = flg_gap && (!flg_newton || flg_gradient);
    tick(32);
    if (!done && flg_newton) {
      factorize(1);
      tick(33);
      newton_mgrad();
      tick(34);
      done = flg_gradient && (r_max(0, 0.5*scale*dotv(grad, Mgrad)) < tol);
    }
    if (!done) {
      if (fused) run_pass(CSR_OP_DIR, 1, 0);
      else { MJH_FOR_LANES(i, nv) search[i] = -1*Mgrad[i]; wv_sync(); }
    }

    int iter = 0;
    const int maxiter = M.o.iterations;
    while (!done && iter < maxiter) {
      // ================= PrimalSearch
      real alpha = 0, ls_improvement = 0;
      real snorm, q3[3];
      if (SPA == 2) {
        // (a chain of nv/4 dependent additions costs the same for one product as for sixteen: |search|^2 shares the pass
        // of PrimalPrepare's three sums, so M search is formed first -- it is not read again if the norm ends the solve)
        if (!fused) mul_M(Mv, search);
        tick(43);
        real q4[4];
        if (fused) csr_sums(nv, 4, prodv, q4);
        else {
          const crptr xs[4] = {search, search, qfs, search}, ys[4] = {search, Ma, search, Mv};
          csr_dots(nidof, idof, nidof == nv, 4, xs, ys, q4, dstage, dstage_cap);
        }
        snorm = sqrt(q4[0]); q3[0] = q4[1]; q3[1] = q4[2]; q3[2] = q4[3];
        tick(45);
      } else snorm = sqrt(dotv(search, search));
      if (!(snorm < MJH_MINVAL)) {
        const real gtol = tol*M.o.ls_tolerance*snorm/scale;
        tick(42);
        if (SPA != 2) mul_M(Mv, search);
        mul_J(Jv, search, 0);
        tick(44);
        // PrimalPrepare
        if (SPA != 2) { q3[0] = dotv(search, Ma); q3[1] = dotv(qfs, search); q3[2] = dotv(search, Mv); }
        const real qg1 = q3[0] - q3[1];
        const real qg2 = 0.5*q3[2];
        MJH_FOR_LANES(r, nefc) {
          if (!in_row(r)) continue;
          if (is_cone_row(r) && !cone_leader(P, r)) continue;
          const real DJ0 = P.D[r]*jar[r];
          real q0 = jar[r]*DJ0, q1 = Jv[r]*DJ0, q2 = Jv[r]*P.D[r]*Jv[r];
          if (is_cone_row(r)) {
            const int dim = cone_dim(P, r, nefc);
            const real mu = P.cone[r];
            real U[6], V[6], UU = 0, UV = 0, VV = 0;
            for (int j = 1; j < dim; j++) {
              const real DJj = P.D[r + j]*jar[r + j];
              q0 += jar[r + j]*DJj;
              q1 += Jv[r + j]*DJj;
              q2 += Jv[r + j]*P.D[r + j]*Jv[r + j];
            }
            U[0] = jar[r]*mu; V[0] = Jv[r]*mu;
            for (int j = 1; j < dim; j++) { U[j] = jar[r + j]*P.cone[r + j]; V[j] = Jv[r + j]*P.cone[r + j]; }
            for (int j = 1; j < dim; j++) { UU += U[j]*U[j]; UV += U[j]*V[j]; VV += V[j]*V[j]; }
            quad[3*r + 3] = U[0]; quad[3*r + 4] = V[0]; quad[3*r + 5] = UU; quad[3*r + 6] = UV; quad[3*r + 7] = VV;
            quad[3*r + 8] = P.D[r]/((mu*mu)*(1 + (mu*mu)));
          }
          quad[3*r] = q0*0.5; quad[3*r + 1] = q1; quad[3*r + 2] = q2*0.5;
        }
        wv_sync();
        tick(38);          // (profile builds: direction products and quadratic coefficients, apart from the evaluations)

        int lsiter = 0;
        // PrimalEval: the six running sums of the reference's loop over rows -- cost, deriv[0..1],
        // quadTotal[0..2] -- as ordered chains over the rows' contributions
        // (explicit-index path with room in the LDS tail: the rows' addends are written to a row-major block and the six
        // sums are taken by six lanes, one read and one addition per row, instead of a v_readlane walk per contributing
        // row and sum.  Rows that contribute an exact zero are added too: no accumulator here can be -0.0 -- mju_dot never
        // returns it, so neither initial value is -- hence x + (+-0) = x for every partial sum.)
        // (one 64-row group at a time: 3 KB -- the explicit-index path takes them from the staging block in the LDS tail, the
        // sparse Newton path from the slot efc_layout reserves ahead of the factor)
        real* const ev = SPA == 2 ? dstage : P.ev;
        const long long ev_off = mjh_lds_offset((const void*)ev);
        const int ev_ok = ev != nullptr && ev_off >= 0 && ev_off < 160*1024 && (SPA != 2 || dstage_cap >= 6*MJH_W);
        real ev_a = 0;
        auto eval = [&](NtPoint& p) {
          const real al = p.alpha;
          real acc[6] = {0, 0, 0, 0, qg1, qg2};          // cost, d0, d1, qT0, qT1, qT2
          ev_a = lane == 4 ? qg1 : (lane == 5 ? qg2 : (real)0);
          for (int r0 = 0; r0 < nefc; r0 += MJH_W) {
            const int r = r0 + lane;
            real c[6] = {0, 0, 0, 0, 0, 0};
            if (r < nefc && in_row(r)) {
              if (r < ne) { c[4] = quad[3*r + 1]; c[5] = quad[3*r + 2]; }
              else if (r < ne + nf) {
                const real start = jar[r], dir = Jv[r], x = start + al*dir;
                const real f = P.floss[r], D = P.D[r], Rf = P.R[r]*f;
                c[0] = nt_friction_costdif(start, x, f, Rf, D);
                if (-Rf < x && x < Rf) { c[1] = D*x*dir; c[2] = D*dir*dir; }
                else if (x <= -Rf) c[1] = -f*dir;
                else c[1] = f*dir;
              } else if (is_cone_row(r)) {
                if (cone_leader(P, r)) {
                  crptr q = quad + 3*r;
                  const real mu = P.cone[r];
                  const real U0 = q[3], V0 = q[4], UU = q[5], UV = q[6], VV = q[7], Dm = q[8];
                  c[0] = nt_elliptic_costdif(q, al, mu, Dm);
                  const real N = U0 + al*V0;
                  const real Tsqr = UU + al*(2*UV + al*VV);
                  if (Tsqr <= 0) {
                    if (N < 0) { c[1] = 2*al*q[2] + q[1]; c[2] = 2*q[2]; }
                  } else {
                    const real T = sqrt(Tsqr);
                    if (N >= mu*T) {}
                    else if (mu*N + T <= 0) { c[1] = 2*al*q[2] + q[1]; c[2] = 2*q[2]; }
                    else {
                      const real N1 = V0;
                      const real T1 = (UV + al*VV)/T;
                      const real T2 = VV/T - (UV + al*VV)*T1/(T*T);
                      c[1] = Dm*(N - mu*T)*(N1 - mu*T1);
                      c[2] = Dm*((N1 - mu*T1)*(N1 - mu*T1) + (N - mu*T)*(-mu*T2));
                    }
                  }
                }
              } else {
                const real start = jar[r], x = start + al*Jv[r];
                const real cost0 = start < 0 ? (real)quad[3*r] : (real)0;
                if (x < 0) { c[3] = quad[3*r] - cost0; c[4] = quad[3*r + 1]; c[5] = quad[3*r + 2]; }
                else c[0] = -cost0;
              }
            }
            // (rows that contribute an exact zero to a sum cannot change it: only the others are chained --
            // friction rows feed cost / derivatives, active contacts the quadratic totals, seldom both)
            // (the three quadratic totals come from the same rows -- the active contacts: one walk, three overlapping chains)
            if (ev_ok) {
              if (r < nefc) { for (int q = 0; q < 6; q++) ev[6*(r - r0) + q] = c[q]; }
              wv_sync();
              if (lane < 6) {
                const int nr = nefc - r0 < MJH_W ? nefc - r0 : MJH_W;
#ifdef MJH_HOSTSIM
                ev_a = csr_chain_serial(CsrStrided<const real*>{(const real*)ev + lane, 6}, nr, ev_a);
#else
                ev_a = csr_chain_serial(CsrStrided<LP<const real>>{mjh_local((const real*)ev + lane), 6}, nr, ev_a);
#endif
              }
              wv_sync();
              continue;
            }
            for (int q = 0; q < 3; q++) acc[q] = wv_chain_mask(acc[q], c[q], wv_ballot(c[q] != 0));
            wv_chain3_mask(acc + 3, c + 3, wv_ballot(c[3] != 0 || c[4] != 0 || c[5] != 0));
          }
          if (ev_ok) { for (int q = 0; q < 6; q++) acc[q] = wv_bcast(ev_a, q); }
          real cost = acc[0], d0 = acc[1], d1 = acc[2];
          cost += al*al*acc[5] + al*acc[4] + acc[3];
          d0 += 2*al*acc[5] + acc[4];
          d1 += 2*acc[5];
          if (d1 <= 0) d1 = MJH_MINVAL;
          p.cost = cost; p.d0 = d0; p.d1 = d1;
          lsiter++;
#ifdef MJH_PROFILE
          if (lane == 0) MJH_G(B, prof, e)[40] += 1;    // line-search evaluations
#endif
        };
        const int lsmax = M.o.ls_iterations;
        NtPoint p0, p1, p2, pmid, p1next, p2next;
        p0.alpha = 0; eval(p0);
        p1.alpha = p0.alpha - p0.d0/p0.d1; eval(p1);
        int found = 0;
        if (fabs(p1.d0) < gtol && (p1.alpha == 0 || p1.cost < 0)) { alpha = p1.alpha; ls_improvement = -p1.cost; found = 1; }
        if (!found) {
          const int dir = (p1.d0 < 0) ? 1 : -1;
          p2 = p0;
          while (p1.d0*dir <= -gtol && lsiter < lsmax) {            // one-sided search
            p2 = p1;
            p1.alpha -= p1.d0/p1.d1; eval(p1);
            if (fabs(p1.d0) < gtol && p1.cost < 0) { alpha = p1.alpha; ls_improvement = -p1.cost; found = 1; break; }
          }
          if (!found && lsiter >= lsmax) { alpha = p1.alpha; ls_improvement = -p1.cost; found = 1; }
        }
        if (!found) {
          p2next = p1;
          p1next.alpha = p1.alpha - p1.d0/p1.d1; eval(p1next);
          auto update_bracket = [&](NtPoint& p, const NtPoint* cand, NtPoint& pnext) {
            int flag = 0;
            for (int i = 0; i < 3; i++) {
              if (p.d0 < 0 && cand[i].d0 < 0 && p.d0 < cand[i].d0) { p = cand[i]; flag = 1; }
              else if (p.d0 > 0 && cand[i].d0 > 0 && p.d0 > cand[i].d0) { p = cand[i]; flag = 2; }
            }
            if (flag) { pnext.alpha = p.alpha - p.d0/p.d1; eval(pnext); }
            return flag;
          };
          while (lsiter < lsmax) {                                    // bracketed search
            pmid.alpha = 0.5*(p1.alpha + p2.alpha); eval(pmid);
            const NtPoint cand[3] = {p1next, p2next, pmid};
            int best = -1;
            real bestcost = 0;
            for (int i = 0; i < 3; i++)
              if (fabs(cand[i].d0) < gtol && (best == -1 || cand[i].cost < bestcost)) { bestcost = cand[i].cost; best = i; }
            if (best >= 0) { alpha = cand[best].alpha; ls_improvement = -cand[best].cost; found = 1; break; }
            const int b1 = update_bracket(p1, cand, p1next);
            const int b2 = update_bracket(p2, cand, p2next);
            if (!b1 && !b2) { alpha = pmid.alpha; ls_improvement = -pmid.cost; found = 1; break; }
          }
          if (!found) {
            if (p1.cost <= p2.cost && p1.cost < 0) { alpha = p1.alpha; ls_improvement = -p1.cost; }
            else if (p2.cost <= p1.cost && p2.cost < 0) { alpha = p2.alpha; ls_improvement = -p2.cost; }
            else alpha = 0;
          }
        }
      }
      tick(36);
#ifdef MJH_PROFILE
      if (lane == 0) MJH_G(B, prof, e)[39] += 1;      // line searches
#endif
      if (alpha == 0) break;

      // ================= move, update
      if (fused) {
        // (rows first, then ONE pass over the dofs: move, J' force, gradient, preconditioner, Hager-Zhang differences)
        MJH_FOR_LANES(r, nefc) jar[r] += Jv[r]*alpha;
        MJH_FOR_LANES(r, nefc) oldstate[r] = P.state[r];
        wv_sync();
        update_rows();
        run_pass(CSR_OP_STEP, 0, alpha);
      } else {
      if (SPA == 2) {
        for (int i0 = lane; i0 < nv; i0 += MJH_NVU*MJH_W) {
          real q[MJH_NVU], sv[MJH_NVU], ma[MJH_NVU], mv[MJH_NVU], g[MJH_NVU], mg[MJH_NVU];
#pragma unroll
          for (int u = 0; u < MJH_NVU; u++) {
            const int i = i0 + u*MJH_W < nv ? i0 + u*MJH_W : 0;
            q[u] = qacc[i]; sv[u] = search[i]; ma[u] = Ma[i]; mv[u] = Mv[i]; g[u] = grad[i]; mg[u] = Mgrad[i];
          }
#pragma unroll
          for (int u = 0; u < MJH_NVU; u++) {
            const int i = i0 + u*MJH_W;
            if (i < nv) { qacc[i] = q[u] + sv[u]*alpha; Ma[i] = ma[u] + mv[u]*alpha; gradold[i] = g[u]; Mgradold[i] = mg[u]; }
          }
        }
      } else {
      MJH_FOR_LANES(i, nv) { qacc[i] += search[i]*alpha; Ma[i] += Mv[i]*alpha; }
      if (!flg_newton) MJH_FOR_LANES(i, nv) { gradold[i] = grad[i]; Mgradold[i] = Mgrad[i]; }
      }
      MJH_FOR_LANES(r, nefc) jar[r] += Jv[r]*alpha;
      MJH_FOR_LANES(r, nefc) oldstate[r] = P.state[r];
      wv_sync();
      update_constraint();
      }
      tick(37);
      if (flg_newton) hessian_incremental();
      tick(35);
      if (!fused) {
        update_grad();
        if (flg_newton) newton_mgrad(); else precondition();
      }
      tick(34);
      const real improvement = scale*ls_improvement;
      // (CG on the explicit-index path: the six sums of the Hager-Zhang update -- |grad|^2 among them -- are taken in one
      // pass before the termination test instead of |grad|^2 now and the others afterwards)
      real hz[6];
      if (SPA == 2 && !flg_newton) {
        if (!fused)
        for (int i0 = lane; i0 < nv; i0 += MJH_NVU*MJH_W) {
          real g[MJH_NVU], go[MJH_NVU], mg[MJH_NVU], mgo[MJH_NVU];
#pragma unroll
          for (int u = 0; u < MJH_NVU; u++) { const int i = i0 + u*MJH_W < nv ? i0 + u*MJH_W : 0; g[u] = grad[i]; go[u] = gradold[i]; mg[u] = Mgrad[i]; mgo[u] = Mgradold[i]; }
#pragma unroll
          for (int u = 0; u < MJH_NVU; u++) { const int i = i0 + u*MJH_W; if (i < nv) { tmpv[i] = g[u] - go[u]; gradold[i] = mg[u] - mgo[u]; } }
        }
        wv_sync();
        if (fused) csr_sums(nv, 6, prodv, hz);
        else {
          const crptr xs[6] = {search, tmpv, tmpv, search, search, grad}, ys[6] = {tmpv, gradold, Mgrad, grad, search, grad};
          csr_dots(nidof, idof, nidof == nv, 6, xs, ys, hz, dstage, dstage_cap);
        }
      }
      const real gradient = scale*sqrt((SPA == 2 && !flg_newton) ? hz[5] : dotv(grad, grad));
      tick(41);          // (profile builds, slots 41..45: gradient norm | direction update | |search|, M search | J search | PrimalPrepare sums)
      const real decrement = flg_newton ? r_max(0, 0.5*scale*dotv(grad, Mgrad)) : 0;
      iter++;
      if ((improvement > 0 && improvement < tol) || gradient < tol || (flg_newton && decrement < tol)) break;
      if (flg_newton) {
        MJH_FOR_LANES(i, nv) search[i] = -1*Mgrad[i];
      } else {
        // Hager-Zhang conjugate direction (engine_solver.c:2489-2521)
        if (SPA != 2) {
          MJH_FOR_LANES(i, nv) { tmpv[i] = grad[i] - gradold[i]; gradold[i] = Mgrad[i] - Mgradold[i]; }   // graddif, Mgraddif
          wv_sync();
        }
        crptr graddif = tmpv, Mgraddif = gradold;
        real beta;
        if (SPA != 2) hz[0] = dotv(search, graddif);
        const real d_dot_y = hz[0];
        if (d_dot_y < MJH_MINVAL) beta = 0;
        else {
          const real y_dot_My = SPA == 2 ? hz[1] : dotv(graddif, Mgraddif);
          const real y_dot_Mgrad = SPA == 2 ? hz[2] : dotv(graddif, Mgrad);
          const real d_dot_grad = SPA == 2 ? hz[3] : dotv(search, grad);
          const real beta_hz = (y_dot_Mgrad - 2*(y_dot_My/d_dot_y)*d_dot_grad)/d_dot_y;
          const real d_norm = sqrt(SPA == 2 ? hz[4] : dotv(search, search));
          const real grad_norm = sqrt(SPA == 2 ? hz[5] : dotv(grad, grad));
          const real eta_k = -1.0/r_max(MJH_MINVAL, d_norm*r_min(0.01, grad_norm));
          beta = r_max(eta_k, beta_hz);
        }
        wv_sync();
        if (fused) run_pass(CSR_OP_DIR, 0, beta);
        else if (SPA == 2) {
          for (int i0 = lane; i0 < nv; i0 += MJH_NVU*MJH_W) {
            real mg[MJH_NVU], sv[MJH_NVU];
#pragma unroll
            for (int u = 0; u < MJH_NVU; u++) { const int i = i0 + u*MJH_W < nv ? i0 + u*MJH_W : 0; mg[u] = Mgrad[i]; sv[u] = search[i]; }
#pragma unroll
            for (int u = 0; u < MJH_NVU; u++) { const int i = i0 + u*MJH_W; if (i < nv) search[i] = -mg[u] + beta*sv[u]; }
          }
        } else
        MJH_FOR_LANES(i, nv) search[i] = -Mgrad[i] + beta*search[i];
      }
      wv_sync();
    }
    if (isl == 0) niter0 = iter;
  }   // islands
  if (lane == 0) counts[MJH_C_NITER] = niter0;
  wv_sync();
}

MJH_DEVN void solve_newton(MREF M_, BREF B_, int e_) {
  const int ell = MJH_HAS(MJH_FT_ELLIPTIC) && M_.o.cone != 0;
  if (M_.s.csr) { if (ell) solve_primal<1, 2, 1>(M_, B_, e_, 1); else solve_primal<0, 2, 1>(M_, B_, e_, 1); }
  else if (M_.s.sparse) { if (ell) solve_primal<1, 1>(M_, B_, e_, 1); else solve_primal<0, 1>(M_, B_, e_, 1); }
  else { if (ell) solve_primal<1, 0>(M_, B_, e_, 1); else solve_primal<0, 0>(M_, B_, e_, 1); }
}
MJH_DEVN void solve_cg(MREF M_, BREF B_, int e_) {
  const int ell = MJH_HAS(MJH_FT_ELLIPTIC) && M_.o.cone != 0;
  if (M_.s.csr) { if (ell) solve_primal<1, 2>(M_, B_, e_, 0); else solve_primal<0, 2>(M_, B_, e_, 0); }
  else if (M_.s.sparse) { if (ell) solve_primal<1, 1>(M_, B_, e_, 0); else solve_primal<0, 1>(M_, B_, e_, 0); }
  else { if (ell) solve_primal<1, 0>(M_, B_, e_, 0); else solve_primal<0, 0>(M_, B_, e_, 0); }
}
#endif  // !MJH_LANE_MODE
