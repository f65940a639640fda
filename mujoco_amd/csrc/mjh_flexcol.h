// Flex collisions: one body against one flex (a "job"), in the place the reference's bodyflex order gives the pair
// (mj_collision / mj_collideTree, engine_collision_driver.c:637-730, :996-1150).
//
//   planes of a static body   every vertex sphere against the plane          mj_collidePlaneFlex :2086-2140
//   other geoms               every BVH leaf (active element) whose box the geom's bounds reach: GJK / EPA of the geom
//                             against the element                           mj_collideGeomElem :2372-2515, mjc_ConvexElem
//   filterFlexContacts        more than mjMAXCONPAIR candidates: deepest first, then farthest-point sampling :447-515
//   contactSort               by (geom, vertex / element)                   :410-443
//
// Mapping.  Lanes take vertices / BVH leaves; hits are compacted in order into the job's candidate table (global memory),
// the leaves that pass the box tests are compacted first so that a narrowphase round runs with every lane busy.  The
// midphase walk is replaced by the leaf tests it ends in (the inner boxes contain their leaves); the candidates of a
// geom are generated in the order the walk visits the leaves, which only matters when filterFlexContacts has to break
// a tie.  (included once per SPMD mode by mjh_stages.inc, after mjh_collision.h: no include guard)

#if !MJH_LANE_MODE

#define MJH_FLEX_MAXCON 50          // mjMAXCONPAIR
enum { FC_DIST = 0, FC_POS = 1, FC_NRM = 4, FC_MIND = 7, FC_NREAL = 8 };
enum { FI_GEOM = 0, FI_OBJ = 1, FI_PAIR = 2, FI_KIND = 3, FI_SEL = 4, FI_NINT = 5 };

// (value, index) maximum over the wavefront, smallest index among equals; lanes without a candidate pass index < 0
MJH_DEV int flex_argmax(real v, int idx) {
  real bv = idx >= 0 ? v : (real)-MJH_MAXVAL*MJH_MAXVAL;
  int bi = idx >= 0 ? idx : 0x7fffffff;
  for (int mask = MJH_W/2; mask >= 1; mask >>= 1) {
    const real ov = wv_shfl_xor(bv, mask);
    const int oi = wv_shfl_i(bi, wv_lane() ^ mask);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
  }
  return bi == 0x7fffffff ? -1 : bi;
}

// returns the number of contacts written from slot `base` on | overflow << 16
MJH_DEVN int flex_collide_job(MREF M_, BREF B_, int e_, int seg, int base) {
  MJH_ENTER(M_, B_, e_);
  const MJH_CONST_AS DSizes& s = M.s;
  const int f = M.colseg[3*seg + 2];
  const int a0 = M.flexjob_adr[seg], a1 = M.flexjob_adr[seg + 1];
  if (a1 == a0) return 0;
  crptr gx = MJH_F(B, geom_xpos, e);
  crptr gm = MJH_F(B, geom_xmat, e);
  crptr vx = MJH_F(B, flexvert_xpos, e);
  crptr aabb = MJH_F(B, flexelem_aabb, e);
  rptr cand = MJH_G(B, flexcand, e);
  iptr ci = MJH_G(B, flexcand_i, e);
  iptr surv = ci + FI_NINT*s.nflexcand;
  const real radius = M.flex_radius[f];
  const int vadr = M.flex_vertadr[f], nvert = M.flex_vertnum[f];
  const int eadr = M.flex_elemadr[f];
  int n = 0;
#ifdef MJH_PROFILE
  // (profile builds, slots 53..56: planes | leaf culling | element narrowphase | filter, sort, emission)
  long long ptick = wv_clock();
  auto tick = [&](int slot) { const long long c_ = wv_clock(); if (wv_lane() == 0) MJH_G(B, prof, e)[slot] += (real)(c_ - ptick)*0.01; ptick = c_; };
#else
  auto tick = [](int) {};
#endif

  // ---- planes: every vertex (mj_collidePlaneFlex)
  for (int a = a0; a < a1; a++) {
    const int g = M.flexjob_geom[a];
    if (M.geom_type[g] != MJH_GEOM_PLANE) continue;
    const int p = s.npair + a;
    const real bound = M.pair_margin[p] + radius;           // margin + gap + radius
    const V3 pos = ld3(gx + 3*g);
    const V3 nrm = mcol(gm + 9*g, 2);
    // (four blocks of vertices per trip: a lone wavefront pays the position loads' round trip once instead of four times;
    // the hits are compacted block by block, in vertex order)
    for (int v0 = 0; v0 < nvert; v0 += 4*MJH_W) {
      int hit[4]; real dist[4]; V3 vp[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int v = v0 + u*MJH_W + wv_lane();
        hit[u] = 0; dist[u] = 0; vp[u] = V3{0, 0, 0};
        if (v < nvert) {
          vp[u] = ld3(vx + 3*(vadr + v));
          const V3 dif = vp[u] - pos;
          dist[u] = dif.x*nrm.x + dif.y*nrm.y + dif.z*nrm.z;
          hit[u] = !(dist[u] > bound);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int v = v0 + u*MJH_W + wv_lane();
        const unsigned long long m = wv_ballot(hit[u]);
        if (hit[u]) {
          const int c = n + wv_rank_lt(m);
          const real cd = dist[u] - radius;
          const real scl = -cd*0.5 - radius;
          cand[FC_NREAL*c + FC_DIST] = cd;
          cand[FC_NREAL*c + FC_POS] = vp[u].x + nrm.x*scl; cand[FC_NREAL*c + FC_POS + 1] = vp[u].y + nrm.y*scl; cand[FC_NREAL*c + FC_POS + 2] = vp[u].z + nrm.z*scl;
          cand[FC_NREAL*c + FC_NRM] = nrm.x; cand[FC_NREAL*c + FC_NRM + 1] = nrm.y; cand[FC_NREAL*c + FC_NRM + 2] = nrm.z;
          ci[FI_NINT*c + FI_GEOM] = g; ci[FI_NINT*c + FI_OBJ] = v; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 0;
        }
        n += __builtin_popcountll(m);
      }
    }
  }

  tick(53);
  // ---- other geoms: BVH leaves in reach, then GJK / EPA per surviving (geom, element)
  const int l0 = M.flex_leafadr[f], l1 = M.flex_leafadr[f + 1];
  for (int a = a0; a < a1; a++) {
    const int g = M.flexjob_geom[a];
    if (M.geom_type[g] == MJH_GEOM_PLANE) continue;
    const int p = s.npair + a;
    const real mg = M.pair_margin[p];                       // margin + gap
    const real sbound = M.geom_rbound[g] + mg;
    const int gbody = M.geom_bodyid[g];
    const real ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero3[3] = {0, 0, 0};
    int nsurv = 0;
    // (two blocks of leaves per trip, for the same reason; survivors compacted in leaf order)
    for (int k0 = l0; k0 < l1; k0 += 2*MJH_W) {
      int ok[2], el[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int k = k0 + u*MJH_W + wv_lane();
        ok[u] = 0; el[u] = -1;
        if (k < l1) {
          el[u] = M.flexleaf_elem[k];
          crptr bx = aabb + 6*el[u];
          const real sx = gx[3*g], sy = gx[3*g + 1], sz = gx[3*g + 2];
          // filterSphereBox (:236-244)
          ok[u] = !(sx + sbound < bx[0] - bx[3] || sy + sbound < bx[1] - bx[4] || sz + sbound < bx[2] - bx[5] ||
                    sx - sbound > bx[0] + bx[3] || sy - sbound > bx[1] + bx[4] || sz - sbound > bx[2] + bx[5]);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (ok[u]) ok[u] = bp_obb<0>(M.geom_aabb + 6*g, aabb + 6*el[u], gx + 3*g, gm + 9*g, zero3, ident, mg);
        // an element with a vertex on the geom's own body is skipped (mj_collideGeomElem :2387-2394)
        if (ok[u]) for (int i = 0; i < 4; i++) { const int v = M.flexelem_vert[4*el[u] + i]; if (v >= 0 && M.flexvert_bodyid[v] == gbody) ok[u] = 0; }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const unsigned long long m = wv_ballot(ok[u]);
        if (ok[u]) surv[nsurv + wv_rank_lt(m)] = el[u];
        nsurv += __builtin_popcountll(m);
      }
    }
    wv_sync();
    tick(54);
    for (int r0 = 0; r0 < nsurv; r0 += MJH_W) {
      const int r = r0 + wv_lane();
      const int el = r < nsurv ? surv[r] : -1;
      const int got = ccd_geom_elem_pair(M, B, e, el >= 0 ? g : -1, el, mg);
      const unsigned long long m = wv_ballot(got > 0);
      if (got > 0) {
        const crptr rec = ccd_out_records(M, B, e);
        const int c = n + wv_rank_lt(m);
        for (int q = 0; q < 7; q++) cand[FC_NREAL*c + q] = rec[q];
        ci[FI_NINT*c + FI_GEOM] = g; ci[FI_NINT*c + FI_OBJ] = el - eadr; ci[FI_NINT*c + FI_PAIR] = p; ci[FI_NINT*c + FI_KIND] = 1;
      }
      n += __builtin_popcountll(m);
    }
    wv_sync();
    tick(55);
  }
  wv_sync();
  if (n == 0) return 0;

  // ---- filterFlexContacts: more candidates than a pair may keep.  The reference works on array positions: it swaps the
  //      chosen contact forward but leaves the `selected` / `min_dist` entries where they are -- reproduced as is.
  int nsel = n;
  if (n > MJH_FLEX_MAXCON) {
    MJH_FOR_LANES(i, n) { cand[FC_NREAL*i + FC_MIND] = MJH_MAXVAL; ci[FI_NINT*i + FI_SEL] = 0; }
    wv_sync();
    int best;
    {
      real bv = 0; int bi = -1;
      MJH_FOR_LANES(i, n) { const real v = -cand[FC_NREAL*i + FC_DIST]; if (bi < 0 || v > bv) { bv = v; bi = i; } }
      best = flex_argmax(bv, bi);
    }
    nsel = 0;
    while (nsel < MJH_FLEX_MAXCON && best >= 0) {
      if (wv_lane() == 0) ci[FI_NINT*best + FI_SEL] = 1;
      wv_sync();
      const V3 bp = ld3(cand + FC_NREAL*best + FC_POS);
      real bv = 0; int bi = -1;
      MJH_FOR_LANES(i, n) {
        if (ci[FI_NINT*i + FI_SEL]) continue;
        const real dx = cand[FC_NREAL*i + FC_POS] - bp.x, dy = cand[FC_NREAL*i + FC_POS + 1] - bp.y, dz = cand[FC_NREAL*i + FC_POS + 2] - bp.z;
        const real d2 = dx*dx + dy*dy + dz*dz;
        real md = cand[FC_NREAL*i + FC_MIND];
        if (d2 < md) { md = d2; cand[FC_NREAL*i + FC_MIND] = md; }
        if (bi < 0 || md > bv) { bv = md; bi = i; }
      }
      int next = flex_argmax(bv, bi);
      wv_sync();
      if (nsel < MJH_FLEX_MAXCON - 1) {
        if (wv_lane() == 0 && best != nsel) {
          for (int q = 0; q < 7; q++) { const real t = cand[FC_NREAL*nsel + q]; cand[FC_NREAL*nsel + q] = cand[FC_NREAL*best + q]; cand[FC_NREAL*best + q] = t; }
          for (int q = 0; q < 4; q++) { const int t = ci[FI_NINT*nsel + q]; ci[FI_NINT*nsel + q] = ci[FI_NINT*best + q]; ci[FI_NINT*best + q] = t; }
        }
        if (next == nsel) next = best;
        wv_sync();
      }
      nsel++;
      best = next;
    }
  }

  // ---- contactSort (stable, by geom then vertex / element) and emission
  int overflow = 0;
  for (int i0 = 0; i0 < nsel; i0 += MJH_W) {
    const int i = i0 + wv_lane();
    if (i >= nsel) continue;
    const long long key = ((long long)ci[FI_NINT*i + FI_GEOM] << 32) | (unsigned)ci[FI_NINT*i + FI_OBJ];
    int rank = 0;
    for (int j = 0; j < nsel; j++) {
      const long long kj = ((long long)ci[FI_NINT*j + FI_GEOM] << 32) | (unsigned)ci[FI_NINT*j + FI_OBJ];
      if (kj < key || (kj == key && j < i)) rank++;
    }
    const int c = base + rank;
    if (c >= s.nconmax) { overflow = 1; continue; }
    Hit h{cand[FC_NREAL*i + FC_DIST], ld3(cand + FC_NREAL*i + FC_POS), ld3(cand + FC_NREAL*i + FC_NRM), V3{0, 0, 0}};
    store_contact(M, B, e, c, ci[FI_NINT*i + FI_PAIR], h);
    iptr cf = MJH_G(B, con_flex, e) + 3*c;
    const int kind = ci[FI_NINT*i + FI_KIND], obj = ci[FI_NINT*i + FI_OBJ];
    cf[0] = f; cf[1] = kind ? obj : -1; cf[2] = kind ? -1 : obj;
  }
  overflow = wv_any(overflow);
  wv_sync();
  tick(56);
  return nsel | (overflow << 16);
}

#endif   // !MJH_LANE_MODE
