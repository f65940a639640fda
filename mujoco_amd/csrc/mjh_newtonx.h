// Newton's Hessian machinery beyond 128 dofs, on the explicit-index rows of mjh_csr.h (SPA = 2 of mjh_newton.h) -- what
// the reference does with its sparse matrices when mj_isSparse holds (engine_solver.c: MakeHessian :2057, FactorizeHessian
// :2149, HessianCone :2219, HessianIncremental :2285):
//
//   H = J' D J + M                 mju_sqrMatTDSparseSymbolic / Numeric (engine_util_sparse.c:747, :910), mju_addToMatSparse
//   reverse Cholesky H = L' L      mju_cholFactorSymbolic / Numeric (engine_util_solve.c:198, :312)
//   solve, rank-one update         mju_cholSolveSparse (:387), mju_cholUpdateSparse (:431)
//
// The mask form of mjh_sparse.h keeps a dof set in two 64-bit registers and a row of the factor in the lanes; that ends at
// 128 dofs.  Here (nv <= 2048):
//   * H and the factor share ONE packed lower triangle (row r at r (r + 1) / 2) in global memory.  Entries outside the
//     symbolic pattern hold exact zeros, and every sweep below runs over ALL columns of a row, lane = column: the terms the
//     reference's sparse loops do not visit are +-0 here, which changes no sum.  (The rank-one update relies on the same fact
//     the reference states as "change in sparsity pattern of mat is not allowed": the update vector stays inside the
//     pattern of every row it rotates.)
//   * what IS order-dependent follows the reference's structures, which are therefore built: the structural pattern of H
//     as a bit matrix (one atomic OR per pair of entries of a J row), the elimination tree and, per row r, the list of the
//     rows c > r with L[c][r] != 0 in the order mju_cholFactorSymbolic's tree walks reach them -- the order in which
//     mju_cholFactorNumeric subtracts their outer products from row r -- and the pattern of every row of the factor as a
//     bit row: the rank of a column in it is the position of the entry in the reference's compressed row, which decides the
//     accumulator of mju_dotSparse it goes to.
//   * the walks are a scalar recursion over memory (parent / flag arrays): lane 0 runs them between barriers, the seeds
//     (column r of H's pattern, ascending) are gathered by the whole wavefront.
// Island-local problems: the reference factors the island's block with island-local indices, which preserve the order of
// the dofs -- the same sweeps over the island's dof list (ascending, csr_idof) with global indices.
// Parity: bit for bit (tests/test_flex_hostsim.py, the model sweep over model/flex under Newton).
// (included once per SPMD mode by mjh_stages.inc: no include guard)

#if !MJH_LANE_MODE

#define MJH_XN_KB 8
struct XnWork {
  int nv, nw;
  iptr parent, flag, ltn, seed;      // [nv] each
  iptr LT;                           // visiting lists: row r at xn_ltoff(nv, r)
  iptr hbits, lbits;                 // [nv][nw]
  rptr xb;                           // [MJH_XN_KB][nv]: the dense vectors of a batch of rank-one updates
  rptr x, stage, diag, y;            // [nv] each: dense vector of a solve / update | products of a row dot, L[c][r] of a visiting
                                     // list | the pivots of the factor being solved with | the vector of a solve after its first sweep
};
MJH_DEV long long xn_row(int r) { return (long long)r*(r + 1)/2; }
MJH_DEV long long xn_ltoff(int nv, int r) { return (long long)(nv - 1 - r)*(nv - 2 - r)/2; }
MJH_DEV XnWork xn_work(MREF M, BREF B, int e) {
  XnWork W;
  W.nv = M.s.nv; W.nw = M.s.xnw;
  const iptr iw = MJH_G(B, xn_iw, e);
  W.parent = iw; W.flag = iw + W.nv; W.ltn = iw + 2*W.nv; W.seed = iw + 3*W.nv;
  W.LT = MJH_G(B, xn_LT, e);
  W.hbits = MJH_G(B, xn_bits, e); W.lbits = W.hbits + (long long)W.nv*W.nw;
  W.x = MJH_G(B, xn_rw, e); W.stage = W.x + W.nv; W.diag = W.x + 2*W.nv; W.y = W.x + 3*W.nv;
  W.xb = W.x + 4*W.nv;
  return W;
}

// FactorizeHessian(recompute): H of the island's rows / dofs into L, the symbolic structures, the numeric factorisation.
// idof[0..n): the island's dofs, ascending; in_row(k): row k belongs to the island; Dact: D of the rows in the quadratic zone
// (isl >= 0: the rows with efc_island == isl, else every row)
MJH_DEVN_HOT void xn_factorize(MREF M_, const Efc& P, const XnWork& W, rptr L, ciptr idof, int n, int nefc, crptr Dact, crptr Ms, int isl) {
  const MJH_CONST_AS DModel& M = wv_uniform_ref(M_);
  const int nv = W.nv, nw = W.nw, lane = wv_lane();
  auto in_row = [&](int k) { return isl < 0 || P.island[k] == isl; };
#ifdef MJH_XN_DEBUG
  if (lane == 0) fprintf(stderr, "xn_factorize n %d nefc %d isl %d\n", n, nefc, isl);
#endif
  // ---- clear the island's rows of H / L and of the two bit matrices
  MJH_FOR_LANES(k, n) { W.parent[idof[k]] = -1; W.flag[idof[k]] = -1; W.ltn[idof[k]] = 0; }
  for (int k = 0; k < n; k++) {
    const int r = idof[k];
    const long long a = xn_row(r);
    for (int j = lane; j <= r; j += MJH_W) L[a + j] = 0;
    for (int w = lane; w < nw; w += MJH_W) { W.hbits[(long long)r*nw + w] = 0; W.lbits[(long long)r*nw + w] = 0; }
  }
  wv_sync();
  // ---- structural pattern of H (mju_sqrMatTDSparseSymbolic: rows i > c that share a row of J with c -- every pair of
  //      entries of a row; supernodes change the bookkeeping, not the pattern) and of M (mju_addToMatSparse)
  MJH_FOR_LANES(k, nefc) {
    if (!in_row(k)) continue;
    const int a0 = P.rowadr[k], m = P.rowadr[k + 1] - a0;
    for (int a = 1; a < m; a++) {
      const int i = P.colind[a0 + a];
      for (int b = 0; b < a; b++) { const int c = P.colind[a0 + b]; wv_atomic_or_i(&W.hbits[(long long)i*nw + (c >> 5)], 1 << (c & 31)); }
    }
  }
  MJH_FOR_LANES(k, n) {
    const int r = idof[k];
    const int ma = M.M_rowadr[r], mn = M.M_rownnz[r];
    for (int q = 0; q < mn - 1; q++) { const int c = M.M_colind[ma + q]; wv_atomic_or_i(&W.hbits[(long long)r*nw + (c >> 5)], 1 << (c & 31)); }
  }
  wv_sync();
  // ---- H values (mju_sqrMatTDSparseNumeric: entry (r, c), c <= r, accumulates (D[k] J[k][r]) J[k][c] over the rows k
  //      that hold dof r, ascending; rows with a zero scale are skipped).  Row k of J adds its outer product: a lane per
  //      pair of entries, one row at a time
  for (int k = 0; k < nefc; k++) {
    if (!in_row(k)) continue;
    const real dk = Dact[k];
    if (dk == 0) continue;
    const int a0 = P.rowadr[k], m = P.rowadr[k + 1] - a0;
    const int npair = m*(m + 1)/2;
    MJH_FOR_LANES(p, npair) {
      int a = (int)((sqrt(8.0*p + 1.0) - 1.0)*0.5);
      while (a*(a + 1)/2 > p) a--;
      while ((a + 1)*(a + 2)/2 <= p) a++;
      const int b = p - a*(a + 1)/2;                 // entries b <= a: columns c <= r
      const real scale = dk*P.spJ[a0 + a];
      if (scale != 0) L[xn_row(P.colind[a0 + a]) + P.colind[a0 + b]] += scale*P.spJ[a0 + b];
    }
    wv_sync();
  }
  MJH_FOR_LANES(k, n) {
    const int r = idof[k];
    const int ma = M.M_rowadr[r], mn = M.M_rownnz[r];
    for (int q = 0; q < mn; q++) L[xn_row(r) + M.M_colind[ma + q]] += Ms[ma + q];
  }
  wv_sync();
#ifdef MJH_XN_DEBUG
  if (lane == 0) fprintf(stderr, "xn_factorize H done\n");
#endif
  // ---- mju_cholFactorSymbolic: rows descending; seeds = the rows i > r with H[i][r] != 0, ascending; from each seed up
  //      the elimination tree until a row already visited for r
  for (int kr = n - 1; kr >= 0; kr--) {
    const int r = idof[kr];
    int nseed = 0;
    for (int k0 = kr + 1; k0 < n; k0 += MJH_W) {
      const int k = k0 + lane;
      int hit = 0, i = 0;
      if (k < n) { i = idof[k]; hit = (W.hbits[(long long)i*nw + (r >> 5)] >> (r & 31)) & 1; }
      const unsigned long long m = wv_ballot(hit);
      if (hit) W.seed[nseed + wv_rank_lt(m)] = i;
      nseed += __builtin_popcountll(m);
    }
    wv_sync();
    if (lane == 0) {
      W.flag[r] = r;
      const long long off = xn_ltoff(nv, r);
      int cnt = 0;
      for (int q = 0; q < nseed; q++) {
        int i = W.seed[q];
        while (W.flag[i] != r) {
          if (W.parent[i] == -1) W.parent[i] = r;
          W.LT[off + cnt++] = i;
          W.flag[i] = r;
          i = W.parent[i];
        }
      }
      W.ltn[r] = cnt;
    }
    wv_sync();
  }
#ifdef MJH_XN_DEBUG
  if (lane == 0) fprintf(stderr, "xn_factorize symbolic done\n");
#endif
  // (the factor's row patterns: row i holds column r when i is on r's visiting list -- set after the walks, a lane per list
  // entry, so that the walk itself is a chain of LDS accesses without a read-modify-write of global memory in it)
  for (int kr = 0; kr < n; kr++) {
    const int r = idof[kr];
    const long long off = xn_ltoff(nv, r);
    const int cnt = W.ltn[r];
    for (int q = lane; q < cnt; q += MJH_W) wv_atomic_or_i(&W.lbits[(long long)W.LT[off + q]*nw + (r >> 5)], 1 << (r & 31));
  }
  wv_sync();
  // ---- mju_cholFactorNumeric: rows descending; dense[j] -= L[c][r] L[c][j] over the visiting list, then the pivot.
  //      The list's rows and their L[c][r] are fetched once, lane-parallel (seed / stage); a lane then walks the list for its
  //      columns with four rows' loads in flight, the subtractions in list order; the dense row is W.x
  for (int kr = n - 1; kr >= 0; kr--) {
    const int r = idof[kr];
    const long long ar = xn_row(r), off = xn_ltoff(nv, r);
    const int cnt = W.ltn[r];
    for (int q = lane; q < cnt; q += MJH_W) { const int c = W.LT[off + q]; W.seed[q] = c; W.stage[q] = L[xn_row(c) + r]; }
    wv_sync();
    for (int j = lane; j <= r; j += MJH_W) {
      real d = L[ar + j];
      int q = 0;
      for (; q + 4 <= cnt; q += 4) {
        real lc[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { lc[u] = W.stage[q + u]; v[u] = L[xn_row(W.seed[q + u]) + j]; }
#pragma unroll
        for (int u = 0; u < 4; u++) d -= lc[u]*v[u];
      }
      for (; q < cnt; q++) d -= W.stage[q]*L[xn_row(W.seed[q]) + j];
      W.x[j] = d;
    }
    wv_sync();
    real diag = W.x[r];
    if (diag < MJH_MINVAL) diag = MJH_MINVAL;
    const real Lrr = sqrt(diag);
    const real inv = 1.0/Lrr;
    for (int j = lane; j <= r; j += MJH_W) L[ar + j] = j == r ? Lrr : (real)(W.x[j]*inv);
    wv_sync();
  }
}

// mju_cholSolveSparse: out = (L' L)^-1 in over the island's dofs (zero elsewhere).  The pivots are fetched once, lane-parallel;
// the first sweep leaves its result in y (x[i] is read by every lane at row i's turn and never again), so a row costs one
// barrier there
MJH_DEVN_HOT void xn_solve(const XnWork& W, crptr L, ciptr idof, int n, crptr in, rptr out) {
  const int nv = W.nv, nw = W.nw, lane = wv_lane();
  rptr x = W.x, y = W.y;
  MJH_FOR_LANES(j, nv) { x[j] = 0; y[j] = 0; }
  wv_sync();
  MJH_FOR_LANES(k, n) { const int i = idof[k]; x[i] = in[i]; W.diag[i] = L[xn_row(i) + i]; }
  wv_sync();
  // x <- L^-T x, rows descending: x[j] -= L[i][j] x[i]
  for (int k = n - 1; k >= 0; k--) {
    const int i = idof[k];
    const long long ai = xn_row(i);
    real xi = x[i];
    if (xi == 0) continue;                  // (uniform: every lane reads the same element; y[i] stays 0)
    xi /= W.diag[i];
    if (lane == 0) y[i] = xi;
    for (int j = lane; j < i; j += MJH_W) x[j] -= L[ai + j]*xi;
    wv_sync();
  }
  // x <- L^-1 x, rows ascending: x[i] -= mju_dotSparse(row i, x) -- the stored entries' products by their position in the
  // compressed row (rank of the column in the row's pattern), four accumulators, (r0 + r2) + (r1 + r3), then the tail
  for (int k = 0; k < n; k++) {
    const int i = idof[k];
    const long long ai = xn_row(i);
    int cnt = 0;
    for (int j0 = 0; j0 < i; j0 += MJH_W) {
      const int w0 = j0 >> 5;
      unsigned long long word = (unsigned)W.lbits[(long long)i*nw + w0];
      if (w0 + 1 < nw) word |= (unsigned long long)(unsigned)W.lbits[(long long)i*nw + w0 + 1] << 32;
      const int j = j0 + lane;
      if (j < i && ((word >> lane) & 1)) W.stage[cnt + __builtin_popcountll(word & ((1ull << lane) - 1))] = L[ai + j]*y[j];
      cnt += __builtin_popcountll(word);
    }
    real yi = y[i];
    if (cnt) {
      wv_sync();
      const int G = cnt >> 2, a = lane & 3;
      real r = 0;
      for (int g = 0; g < G; g++) r += W.stage[4*g + a];
      real res = (wv_bcast(r, 0) + wv_bcast(r, 2)) + (wv_bcast(r, 1) + wv_bcast(r, 3));
      for (int q = 4*G; q < cnt; q++) res += W.stage[q];
      yi -= res;
    }
    yi /= W.diag[i];
    wv_sync();
    if (lane == 0) y[i] = yi;
    wv_sync();
  }
  MJH_FOR_LANES(j, nv) out[j] = y[j];
  wv_sync();
}

// mju_cholUpdateSparse(L, x, flg_plus) with x = scl * (the m entries vals[0..m) at columns cols[0..m), ascending); returns
// the number of clamped pivots.  Rows from the last non-zero of x downwards; a row whose x entry is zero is skipped.
MJH_DEVN_HOT int xn_update(const XnWork& W, rptr L, ciptr cols, crptr vals, int m, int flg_plus) {
  const int lane = wv_lane();
#ifdef MJH_XN_DEBUG
  if (lane == 0) fprintf(stderr, "xn_update m %d last col %d plus %d\n", m, m > 0 ? (int)cols[m - 1] : -1, flg_plus);
#endif
  if (m <= 0) return 0;
  rptr x = W.x;
  const int start = cols[m - 1];
  for (int j = lane; j <= start; j += MJH_W) x[j] = 0;
  wv_sync();
  MJH_FOR_LANES(q, m) x[cols[q]] = vals[q];
  wv_sync();
  int clamped = 0;
  int row = start;
  while (row >= 0) {
    // next row at or below `row` with a non-zero entry of x (64 rows per look)
    {
      const int j = row - lane;
      const unsigned long long nzm = wv_ballot(j >= 0 && x[j] != 0);
      if (!nzm) { row -= MJH_W; continue; }
      row -= __builtin_ctzll(nzm);
    }
    const long long ar = xn_row(row);
    const real diag = L[ar + row], xr = x[row];
    real tmp = diag*diag + (flg_plus ? xr*xr : -xr*xr);
    if (tmp < MJH_MINVAL) { tmp = MJH_MINVAL; clamped++; }
    const real rr = sqrt(tmp);
    const real c = diag/rr;
    const real sn = -xr/rr;
    const real ss = flg_plus ? -sn : sn;
    wv_sync();
    for (int j = lane; j <= row; j += MJH_W) {
      if (j == row) { L[ar + row] = rr; continue; }
      const real mv = L[ar + j], xj = x[j];
      L[ar + j] = c*mv + ss*xj;
      x[j] = sn*mv + c*xj;
    }
    wv_sync();
    row--;
  }
  return clamped;
}

// Several rank-one updates in ONE sweep over the rows (the reference applies them one after the other, each as a sweep from
// its last non-zero row down: HessianIncremental / HessianCone -> mju_cholUpdateSparse).  Update u rotates row r using only
// row r of the factor and its own vector; update u + 1 may therefore rotate row r as soon as update u has, before update u
// goes on to row r - 1: visiting the rows once, top down, and applying at each row the pending rotations in update order
// performs the reference's operations on the reference's operands, and a row of the factor is read and written once for
// all of them (as sp_update_batch does for the mask form).  The rotation parameters of a row depend on its pivot and on the
// vectors' entries AT that row only, which no rotation of this row changes: they are computed first, in update order.
// xb[u][0..nv): the dense vectors (zero outside their patterns), nb <= MJH_XN_KB of them, bit u of plus: update (1) or
// downdate (0); start: the last row that can hold a non-zero.  Returns the number of clamped pivots.
MJH_DEVN_HOT int xn_update_batch(const XnWork& W, rptr L, int nb, int plus, int start) {
  const int lane = wv_lane(), nv = W.nv;
  rptr xb = W.xb;
  int clamped = 0;
  int row = start;
  while (row >= 0) {
    {
      const int j = row - lane;
      int any = 0;
      if (j >= 0) {
#pragma unroll
        for (int u = 0; u < MJH_XN_KB; u++) if (u < nb) any |= xb[(long long)u*nv + j] != 0;
      }
      const unsigned long long nzm = wv_ballot(any);
      if (!nzm) { row -= MJH_W; continue; }
      row -= __builtin_ctzll(nzm);
    }
    const long long ar = xn_row(row);
    real d = L[ar + row];
    real cu[MJH_XN_KB], su[MJH_XN_KB], qu[MJH_XN_KB];
    int act = 0;
#pragma unroll
    for (int u = 0; u < MJH_XN_KB; u++) {
      cu[u] = 1; su[u] = 0; qu[u] = 0;
      if (u >= nb) continue;
      const real xr = xb[(long long)u*nv + row];
      if (xr == 0) continue;
      const int up = (plus >> u) & 1;
      real tmp = d*d + (up ? xr*xr : -xr*xr);
      if (tmp < MJH_MINVAL) { tmp = MJH_MINVAL; clamped++; }
      const real rr = sqrt(tmp);
      cu[u] = d/rr;
      su[u] = -xr/rr;
      qu[u] = up ? -su[u] : su[u];
      d = rr;
      act |= 1 << u;
    }
    wv_sync();
    for (int j = lane; j < row; j += MJH_W) {
      real mv = L[ar + j];
#pragma unroll
      for (int u = 0; u < MJH_XN_KB; u++) {
        if (!((act >> u) & 1)) continue;
        const real xj = xb[(long long)u*nv + j];
        const real nm = cu[u]*mv + qu[u]*xj;
        xb[(long long)u*nv + j] = su[u]*mv + cu[u]*xj;
        mv = nm;
      }
      L[ar + j] = mv;
    }
    if (lane == 0) L[ar + row] = d;
    wv_sync();
    row--;
  }
  return clamped;
}

#endif   // !MJH_LANE_MODE
