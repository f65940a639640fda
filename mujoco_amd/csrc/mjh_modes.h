// The stage sources are written once against a small SPMD vocabulary (wv_lane, wv_sync, wv_bcast,
// wv_ballot, wv_sum_i, wv_exscan_i, wv_any, MJH_FOR_LANES, MJH_W) and compiled in two mappings:
//
//   namespace wv : "one wavefront per environment".  MJH_W = 64 work items of a phase run on the 64
//                  lanes of the wavefront that owns the environment; phases are separated by
//                  wv_sync(); cross-lane primitives are DPP/readlane/shuffles (mjh_spmd.h).  Used
//                  where one environment has enough fine-grained parallelism and data-dependent
//                  length: collision, constraint assembly, AR = Y Y' and the PGS sweep.
//   namespace ln : "one lane per environment".  MJH_W = 1: every GPU lane walks the serial
//                  algorithm of its own environment, so a wavefront steps 64 environments in
//                  lockstep; model constants are wave-uniform (scalar loads), mjData fields are
//                  read SoA-across-environments (lane stride 8 bytes: fully coalesced), there are
//                  no barriers and no cross-lane traffic.  Used for the tree recursions and the
//                  L'DL factor/solves, whose control flow is identical for all environments of
//                  a model.
#pragma once

#include "mjh_spmd.h"
#include "mjh_math.h"
#include "mjh_types.h"

// ------------------------------------------------------------------------------------------------
// wave mode
// ------------------------------------------------------------------------------------------------
#define MJH_W 64
#define MJH_LANE_MODE 0
#define MJH_DEVN MJH_DEVN_WAVE
#define MJH_FOR_LANES(i, n) for (int i = wv_lane(); i < (n); i += MJH_W)
// entry of an out-of-line stage function: its arguments arrive in VGPRs; all three are wave-uniform
// namespace wv serves environment-major batches only (B.soa == 0): telling the compiler makes every
// strided view a unit-stride view (no index multiply per access)
#define MJH_ENTER(M_, B_, e_) MREF M = wv_uniform_ref(M_); BREF B = wv_uniform_ref(B_); const int e = wv_uniform_i(e_); \
                              if (B.soa != 0) __builtin_unreachable()
namespace wv {
#include "mjh_stages.inc"
}
#undef MJH_ENTER
// namespace ws: the same wave mapping on SoA batches (constraint kernel of the per-step pipeline)
#define MJH_ENTER(M_, B_, e_) MREF M = wv_uniform_ref(M_); BREF B = wv_uniform_ref(B_); const int e = wv_uniform_i(e_)
namespace ws {
#include "mjh_stages.inc"
}
#undef MJH_W
#undef MJH_LANE_MODE
#undef MJH_DEVN
#undef MJH_FOR_LANES
#undef MJH_ENTER

// ------------------------------------------------------------------------------------------------
// lane mode
// ------------------------------------------------------------------------------------------------
#define MJH_W 1
#define MJH_LANE_MODE 1
#define MJH_DEVN MJH_DEVN_LANE
#define MJH_FOR_LANES(i, n) for (int i = 0; i < (n); i++)
// (stage functions are inlined into the kernel here: descriptors are already uniform, e is per lane)
#define MJH_ENTER(M_, B_, e_) MREF M = M_; BREF B = B_; const int e = e_
namespace ln {
// the SPMD vocabulary for a width-1 "wave": these hide the wavefront primitives of mjh_spmd.h
MJH_DEV int wv_lane() { return 0; }
MJH_DEV void wv_sync() {}
MJH_DEV double wv_bcast(double v, int) { return v; }
MJH_DEV int wv_bcast_i(int v, int) { return v; }
MJH_DEV uint64_t wv_ballot(int pred) { return pred ? 1ull : 0ull; }
MJH_DEV int wv_sum_i(int v) { return v; }
MJH_DEV int wv_exscan_i(int) { return 0; }
MJH_DEV int wv_any(int pred) { return pred != 0; }
#include "mjh_stages.inc"
}
#undef MJH_W
#undef MJH_LANE_MODE
#undef MJH_DEVN
#undef MJH_FOR_LANES
#undef MJH_ENTER
