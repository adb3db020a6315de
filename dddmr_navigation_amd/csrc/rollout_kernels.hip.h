// rollout_kernels.hip.h -- CDNA4 (gfx950) kernels of the local-planner tick.
//
// One tick = 3 launches on one stream:
//   k_bin_count  : crop the aggregate cloud to the local costmap tile, count
//                  points per cell; the last workgroup scans the counters
//                  (replaces the per-tick kd-tree build,
//                  mpc_critics/include/mpc_critics/model_shared_data.h:78-81)
//   k_bin_scatter: counting-sort scatter -> cell-sorted float4 points
//   k_score      : fused rollout (trajectory_generators theories) + all critics
//                  (mpc_critics/models/*.cpp) + packed-key argmin; the last
//                  workgroup decodes the winner (local_planner.cpp:447-480)
//                  into host-mapped memory
//
// No MFMA anywhere: this is gather / compare work.
//
// Numerics follow the reference's mixed float/double arithmetic; float
// expressions that decide a boolean (box test, radius test, NN distance) are
// evaluated without fused multiply-add so they round exactly like the x86-64
// build of the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "../../include/dddmr_rollout.h"

// The reference is an x86-64 build without FMA contraction: every multiply and add below
// rounds separately, in float and in double (hipcc's default would fuse them).
#pragma clang fp contract(off)

namespace dddmr {

#ifndef DDDMR_PSPLIT_MAX
#define DDDMR_PSPLIT_MAX 4   // the path critic's 1-NN search is shared by up to 2^4 lanes per pair
#endif
#ifndef DDDMR_SCORE_WPE
#define DDDMR_SCORE_WPE 2   // min waves per SIMD the register allocator must allow for the 512-lane k_score
#endif
// The 256-lane k_score runs big shards in several rounds of workgroups and is bound by how many waves hide each
// other's latency: 5 waves per SIMD (96 VGPRs, the pose math of phase D1 spills ~25 loop-invariant doubles to
// scratch) beat 4 (121 VGPRs, no spills) by 3 % at C3 and 7 % at C4; the single-round 512-lane shape loses 10 %
// with it (C2) and keeps 4.
#ifndef DDDMR_SCORE_WPE_256
#define DDDMR_SCORE_WPE_256 5
#endif
// k_score is instantiated for 256- and 512-lane workgroups (template parameter
// kScoreThreads): 512 lanes halve the collision walk of the heaviest tile and win when
// the shard fits one round of resident workgroups (C2); 256 lanes keep more, smaller
// workgroups in flight and win on big batches (C3, C4).
// Hand-offs inside a launch (binning ticket -> cell scan; winner slots -> ticket -> slot reduction).  What crosses
// workgroups is written ONLY by device-scope atomics or relaxed agent-scope atomic stores (global_store ... sc1: written
// through past the XCD's L2) and read back by agent-scope atomic loads (sc1); the writer then waits for its own
// operations (`s_waitcnt vmcnt(0)`, an asm statement with a memory clobber: the compiler moves nothing across it) before
// one RELAXED device-scope add on the ticket.  In HSA memory-model terms this is a release on the ticket restricted to the
// locations that matter.  Spelled as such (-DDDDMR_HANDOFF_ACQREL: __ATOMIC_ACQ_REL on the ticket) the compiler emits
// `buffer_wbl2 sc1` + `buffer_inv sc1` around the atomic -- a write-back of the XCD's whole L2 per workgroup, measured at
// +8.5 / +8.2 us per C2 / C3 tick on MI355X (profiles/r03_handoff.txt; the marking layer's last launch lost 120 us to the same
// fence, profiles/r03_C5M_fused_kernel_stats.csv history in DESIGN.md).  The relaxed form is kept; the instruction
// sequence it relies on (sc1 stores, s_waitcnt vmcnt(0), global_atomic_add ... sc1) is committed with the ROCm version it
// was checked on in profiles/r03_handoff.txt and re-checked by tools/check_handoff_isa.py after any toolchain update.
#ifdef DDDMR_HANDOFF_ACQREL
#define DDDMR_HANDOFF_ORDER __ATOMIC_ACQ_REL
#else
#define DDDMR_HANDOFF_ORDER __ATOMIC_RELAXED
#endif

constexpr int kBinPer = 4;            // points per lane and pass of a binning workgroup
constexpr int kRolloutMax = 128;      // trajectories per rollout workgroup (phase A: a lane each, C: two), at most
constexpr int kBinThreads = 1024;     // k_bin_count workgroup (its last workgroup scans 4096 cells per step)
constexpr int kMaxTile = 16;          // trajectories per workgroup (upper bound)
constexpr int kMaxPlan = 512;         // prune-plan poses kept in LDS
constexpr int kInlineAxes = 192;      // sample-axis floats that travel inside the kernel arguments
constexpr int64_t kKeyNone = INT64_MAX;
// best_key[]: words 0..2 are the reduced argmin words of multi-round shards (packed key, cost bits, sampled
// collision count); from word kSlotBase on, four words per k_score workgroup for single-round shards (see the
// tail of k_score): cost bits | global index | vx, vy | w, generated << 40, collided << 32
constexpr int kSlotBase = 8;
constexpr int kSlotWords = 4;
constexpr double kMaxAcceptedCost = 9999999.0;   // local_planner.cpp:452
// a cell-sorted point: 12 bytes -- the collision walk is bound by the bytes the lanes
// pull through the vector L1, and the intensity word is never read
struct Pt3 { float x, y, z; };
constexpr int kKeyIndexBits = 24;     // 16.7 M samples per tick

// Per-tick constants, passed by value (kernarg segment => scalar loads).
struct DevTick {
  // --- sampling (initialise() result) ---
  int kind;          // dddmr_theory_kind
  int fixed_steps;   // bench mode, 0 = reference rule
  int list_mode;     // 1: explicit sample list in `samples`, 0: axis grid
  int n_global, begin, n_local;
  int nx, ny, nth;   // axis lengths (grid mode); sample = x-major, y, theta-minor
  int ay_ofs, ath_ofs;
  int max_steps;     // capacity per trajectory (LDS rows are max_steps+1 long)
  double sim_time, sim_gran, ang_gran;
  double min_vel_x, max_vel_x, min_vel_theta, min_vel_trans, max_vel_trans;
  double allowed_max;
  // --- robot pose (tf2::transformToEigen(robot_pose_)) ---
  double R[9];       // row-major rotation
  double t[3];
  float cub[24];     // 8 cuboid vertices, reference push order
  // --- prune plan ---
  int m;             // poses
  int pad0;
  double planR[9];   // rotation of the LAST plan pose
  double planT[3];   // position of the LAST plan pose
  // --- critic stack ---
  int n_critics;
  int ckind[DDDMR_MAX_CRITICS];
  int pad1;
  double cw[DDDMR_MAX_CRITICS], ctw[DDDMR_MAX_CRITICS], cow[DDDMR_MAX_CRITICS];
  double heading_dev;
  // --- cloud + local costmap tile (uniform grid, z fastest, then x, then y) ---
  int n_points;      // aggregate observation size (the "< 5 points" rule uses this)
  int gnx, gny, gnz;
  int n_cells;
  float gmin[3];
  float inv_cell;
  float inv_cell_z;  // z cells keep the small size when x/y cells grow: they only spread the counting atomics
  float rmin[3], rmax[3];   // region accepted by the binning pass
  int tile;          // trajectories per workgroup
  int want_collision, want_minmax;
  int tab_entries;   // row-run index entries staged in LDS by k_score (0 = read from L2)
  // sample axes inline in the kernarg segment when they fit (no per-tick H2D copy):
  // x at [0,nx), y at [nx,nx+ny), theta at [nx+ny, nx+ny+nth)
  int axes_inline;
  int use_assign;    // 1: tiles take their trajectories from assign[] (load feedback), 0: strided
  int n_tiles;       // k_score workgroups of this tick (a multiple of assign_groups)
  int nb_tiles;      // the first nb_tiles workgroups take part in every round of the deal, the others only from round r0 on
  int r0;            //   (nb_tiles = n_tiles, r0 = 0: every workgroup gets `tile` trajectories; see tile_slot())
  int assign_groups; // assignment workgroups
  int rows_cap;      // cell rows one cuboid AABB can span with this tick's cell size (<= kRows)
  int rt;            // trajectories per rollout workgroup
  int bin_blocks;    // binning workgroups of the k_bin_count launch, followed by
  int roll_blocks;   // the rollout workgroups and (use_assign) one assignment workgroup
  int box_fast;      // the cuboid is a body-frame box in the reference's vertex order (host-checked)
  int rec_pose;      // OBB records carry the pose (some pair may need the 1 m radius test, or min-max critic)
  int final_kernel;  // 1: the winner is decoded by k_finalize after k_score (multi-round shards), 0: by k_score's last workgroup
  int probe;         // 1: the collision walk starts with a probe round (pays when most trajectories collide)
  uint32_t seq;      // tick sequence number echoed into DevResult::seq
  float axes_inl[kInlineAxes];
};

// Load feedback: every tick k_score files what each trajectory cost it (collision
// work items actually walked, path-critic work, pairs); the next tick's assignment
// block (rides along with k_bin_count, like the rollout) deals the trajectories to
// the tiles heaviest-first in snake order, so that all workgroups carry the same
// load and the launch has no long tail.  Any assignment gives identical results.
constexpr int kLoadClasses = 1024;
constexpr int kAssignMax = 1 << 18;    // larger shards keep the strided assignment (64 groups of 4096)

struct DevResult {    // written by the last k_score workgroup into host-mapped memory
  int64_t key;
  double cost;
  float vx, vy, wz;
  int32_t index;      // global sample index or -1
  uint32_t n_binned;
  uint32_t overflow;  // 1 if a trajectory needed more than max_steps
  uint32_t seq;       // tick sequence number, stored LAST: the host may poll it instead of a stream sync
  uint32_t n_collided; // trajectories of the shard the collision critics rejected (feedback for DevTick::probe)
};

__host__ __device__ inline int64_t pack_key(double cost, uint32_t gidx) {
  // positive doubles order like their bit patterns; keep the top 40 bits of the
  // cost and put (max - index) below so that min(key) = min cost, ties -> highest
  // index (local_planner.cpp:460-463 keeps the LAST minimal trajectory).
  // (getBestTrajectory starts from minimum_cost = 9999999 and accepts cost_ <= minimum_cost,
  // local_planner.cpp:452,460: a larger cost is never accepted)
  if (!(cost >= 0.0 && cost <= kMaxAcceptedCost)) return kKeyNone;
  union { double d; uint64_t u; } c;
  c.d = cost;
  const uint64_t idx_mask = (1ull << kKeyIndexBits) - 1;
  return (int64_t)((c.u & ~idx_mask) | (idx_mask - (uint64_t)gidx));
}
// The full bit pattern of an acceptable cost (non-negative doubles order like their bits):
// reduced beside the packed key so that the winner is exact, see k_score's decode.
__host__ __device__ inline int64_t cost_bits(double cost) {
  if (!(cost >= 0.0 && cost <= kMaxAcceptedCost)) return kKeyNone;
  union { double d; int64_t i; } c;
  c.d = cost;
  return c.i;
}
__host__ __device__ inline int32_t key_index(int64_t key) {
  if (key == kKeyNone) return -1;
  const uint64_t idx_mask = (1ull << kKeyIndexBits) - 1;
  return (int32_t)(idx_mask - ((uint64_t)key & idx_mask));
}

// ---------------------------------------------------------------------------
// float helpers that must not be contracted into fma
// ---------------------------------------------------------------------------
// (HIP's __fmul_rn / __fadd_rn are plain operators compiled under the translation
// unit's -ffp-contract=fast, i.e. they still carry the `contract` flag and the
// backend may fuse them; the pragma below is what actually keeps them apart.)
__device__ __forceinline__ float fmul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float fadd(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float fsub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// FLANN L2_Simple<float>: result = 0; result += diff*diff per dimension.
__device__ __forceinline__ float l2_simple(float ax, float ay, float az, float bx, float by, float bz) {
  float d = fsub(ax, bx);
  float r = fmul(d, d);
  d = fsub(ay, by);
  r = fadd(r, fmul(d, d));
  d = fsub(az, bz);
  r = fadd(r, fmul(d, d));
  return r;
}
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return fadd(fadd(fmul(ax, bx), fmul(ay, by)), fmul(az, bz));
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (register-to-register, a few
// cycles each) instead of LDS-crossbar shuffles: Hillis-Steele inside the 16-lane rows
// (row_shr 1, 2, 4, 8; lanes without a source add 0), then the row totals are carried over
// with row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}

// DPP row_shl:N (lane i reads lane i+N of its 16-lane row; lanes without a source keep their own
// value) for 32- and 64-bit payloads: the building block of reductions towards lane 0 of a group.
template <int N>
__device__ __forceinline__ int dpp_row_shl(int v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x100 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v) {
  return __int_as_float(dpp_row_shl<N>(__float_as_int(v)));
}
template <int N>
__device__ __forceinline__ long long dpp_row_shl(long long v) {
  const int lo = dpp_row_shl<N>((int)(v & 0xFFFFFFFFll)), hi = dpp_row_shl<N>((int)(v >> 32));
  return ((long long)hi << 32) | (long long)(unsigned int)lo;
}
template <int N>
__device__ __forceinline__ double dpp_row_shl(double v) {
  return __longlong_as_double(dpp_row_shl<N>(__double_as_longlong(v)));
}

// sin and cos of a heading in double.  The rollout's phase B is bound by exactly this
// (one call per pose), and ocml's sincos spends most of its instructions on argument
// ranges a heading never has.  fdlibm's algorithm for |x| < 2^20 pi/2: Cody-Waite
// reduction with pi/2 in 33-bit pieces (118 bits, exact products), then the
// __kernel_sin / __kernel_cos polynomials on [-pi/4, pi/4] with the reduction's tail;
// < 0.8 ulp (ocml and glibc: <= 1 ulp; tests/test_parity_gpu.py checks it against long-double
// values through dddmr_rollout_selftest_sincos).  Larger arguments fall back to ocml.
// Separate multiplies and adds on purpose: the fused form was measured (round 2) and is slower here --
// v_fma_f64 itself issues at the rate of v_mul_f64 / v_add_f64 (tools/ubench/f64_rate.hip: ~5 cycles per
// wave-instruction each), but the compiler picks v_fmac_f64, whose addend must sit in VGPRs while
// v_mul / v_add read the coefficients from SGPR pairs; under k_bin_count's 64-VGPR cap (two 1024-lane
// workgroups per CU) that spilled 14 registers and phase B went from 14.5 to 24 kilo-cycles.
__device__ __forceinline__ void sincos_heading(const double x, double* sn, double* cs) {
  if (fabs(x) > 1.0e5) {
    sincos(x, sn, cs);
    return;
  }
  const double fn = rint(x * 6.36619772367581382433e-01);
  const int n = (int)fn;
#define DDDMR_MADD(a, b, c) ((a) * (b) + (c))
  const double t = x - fn * 1.57079632673412561417e+00;     // exact product: 33-bit piece x |fn| < 2^20
  double w = fn * 6.07710050630396597660e-11;
  const double r = t - w;
  w = fn * 2.02226624879595063154e-21 - ((t - r) - w);
  const double y0 = r - w;
  const double y1 = (r - y0) - w;
  const double z = y0 * y0;
  const double v = z * y0;
  const double ps = DDDMR_MADD(z, DDDMR_MADD(z, DDDMR_MADD(z, DDDMR_MADD(z, 1.58969099521155010221e-10,
                               -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                               -1.98412698298579493134e-04), 8.33333333332248946124e-03);
  const double s = y0 - ((z * (0.5 * y1 - v * ps) - y1) - v * -1.66666666666666324348e-01);
  const double pc = z * DDDMR_MADD(z, DDDMR_MADD(z, DDDMR_MADD(z, DDDMR_MADD(z, DDDMR_MADD(z,
                                   -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                   -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                   -1.38888888888741095749e-03), 4.16666666666666019037e-02);
#undef DDDMR_MADD
  const double hz = 0.5 * z;
  const double w2 = 1.0 - hz;
  const double c = w2 + (((1.0 - w2) - hz) + (z * pc - y0 * y1));
  // quadrant n & 3: 0 (s, c)  1 (c, -s)  2 (-s, -c)  3 (-c, s) -- one swap, then sign bits
  const bool odd = (n & 1) != 0;
  const double a = odd ? c : s, b = odd ? s : c;
  const unsigned long long sa = (unsigned long long)((n >> 1) & 1) << 63;          // sin negative in quadrants 2, 3
  const unsigned long long sb = (unsigned long long)(((n + 1) >> 1) & 1) << 63;    // cos negative in quadrants 1, 2
  *sn = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(a) ^ sa));
  *cs = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(b) ^ sb));
}

// cos/sin(M_PI_2 + theta) in double from sin/cos(theta) (omni_simple...cpp:501-502).  The
// argument the reference hands to libm is t = fl(fl(pi/2) + theta) = pi/2 + theta - e
// with e = (pi/2 - fl(pi/2)) + (rounding error of the sum, exact by Fast2Sum), so
// cos t = -sin(theta - e), sin t = cos(theta - e): first-order corrections of the values
// already at hand (|e| < 3e-16, the neglected term < 1e-31) instead of a second sincos.
__device__ __forceinline__ void sincos_quarter_ahead(const double ang, const double sn, const double cs, double* s2,
                                                     double* c2) {
  const double t = M_PI_2 + ang;
  const double err = fabs(ang) <= M_PI_2 ? (M_PI_2 - t) + ang : (ang - t) + M_PI_2;   // (pi/2_fl + theta) - t
  const double e = 6.123233995736766e-17 + err;
  *c2 = -sn + e * cs;
  *s2 = cs + e * sn;
}

// dddmr_rollout_selftest_sincos: the routine above, one angle per lane
__global__ void k_selftest_sincos(const double* __restrict__ ang, int n, double* __restrict__ sn, double* __restrict__ cs) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  double s, c;
  sincos_heading(ang[i], &s, &c);
  sn[i] = s;
  cs[i] = c;
}

// ---------------------------------------------------------------------------
// binning: cloud -> cell-sorted local costmap tile
// ---------------------------------------------------------------------------
__device__ __forceinline__ int cell_of(const DevTick& k, float x, float y, float z) {
  int cx = (int)floorf((x - k.gmin[0]) * k.inv_cell);
  int cy = (int)floorf((y - k.gmin[1]) * k.inv_cell);
  int cz = (int)floorf((z - k.gmin[2]) * k.inv_cell_z);
  cx = min(max(cx, 0), k.gnx - 1);
  cy = min(max(cy, 0), k.gny - 1);
  cz = min(max(cz, 0), k.gnz - 1);
  return (cy * k.gnx + cx) * k.gnz + cz;
}

// Exclusive scan of the cell counters by ONE workgroup (any size that is a
// multiple of 64, <= 1024); also zeroes the counters for the next tick.
__device__ inline void scan_cells(const DevTick& k, uint32_t* __restrict__ cell_count,
                                  uint32_t* __restrict__ cell_start, uint32_t* wave_sum, uint32_t* carry_s) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wid = tid >> 6, nw = nthr >> 6;
  if (tid == 0) *carry_s = 0;
  __syncthreads();
  const int n = k.n_cells;
  for (int base = 0; base < n; base += nthr * 4) {
    const int i0 = base + tid * 4;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // counters were built with device-scope atomics by workgroups on any XCD
      v[j] = (i0 + j < n) ? __hip_atomic_load(&cell_count[i0 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      if (i0 + j < n) cell_count[i0 + j] = 0u;
    }
    const uint32_t tsum = v[0] + v[1] + v[2] + v[3];
    const uint32_t incl = wave_incl_scan_u32(tsum);
    if (lane == 63) wave_sum[wid] = incl;
    __syncthreads();
    // offsets of the <= 16 waves: one LDS read, a DPP scan, two readlanes
    const uint32_t wincl = wave_incl_scan_u32(lane < nw ? wave_sum[lane] : 0u);
    const int w_u = __builtin_amdgcn_readfirstlane(wid);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wincl, 15);      // lanes >= nw add 0
    const uint32_t wprev = (uint32_t)__builtin_amdgcn_readlane((int)wincl, w_u > 0 ? w_u - 1 : 0);
    const uint32_t wofs = w_u > 0 ? wprev : 0u;
    const uint32_t carry = *carry_s;
    uint32_t run = carry + wofs + incl - tsum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0 + j < n) cell_start[i0 + j] = run;
      run += v[j];
    }
    __syncthreads();
    if (tid == 0) *carry_s = carry + total;
    __syncthreads();
  }
  if (tid == 0) cell_start[n] = *carry_s;
}

// Empty cloud: only the scan/reset part.
__global__ __launch_bounds__(256) void k_bin_reset(DevTick k, uint32_t* __restrict__ cell_count,
                                                   uint32_t* __restrict__ cell_start,
                                                   int64_t* __restrict__ best_key,
                                                   uint32_t* __restrict__ overflow) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) {
    best_key[0] = kKeyNone;
    best_key[1] = kKeyNone;
    best_key[2] = 0;
    *overflow = 0;
  }
  scan_cells(k, cell_count, cell_start, wave_sum, &carry_s);
}

__global__ __launch_bounds__(256) void k_bin_scatter(DevTick k, const float4* __restrict__ cloud,
                                                     const uint2* __restrict__ pt_slot,
                                                     const uint32_t* __restrict__ cell_start,
                                                     Pt3* __restrict__ sorted, uint32_t* __restrict__ row_tab) {
  const int stride = gridDim.x * blockDim.x;
  // The costmap's row-run index, compact: entry (cy, cx) = first sorted point of cell column (cx, cy), cx = gnx being
  // the end of the row.  Every k_score workgroup stages it; gathering it there from cell_start (stride gnz words)
  // cost each of them ~1000 scattered 64-byte sectors.
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k.tab_entries; i += stride) {
    const int cy = i / (k.gnx + 1), cx = i - cy * (k.gnx + 1);
    row_tab[i] = cell_start[(cy * k.gnx + cx) * k.gnz];
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k.n_points; i += stride) {
    const uint2 slot = pt_slot[i];
    if (slot.x != 0xFFFFFFFFu) {
      const float4 p = cloud[i];
      sorted[cell_start[slot.x] + slot.y] = Pt3{p.x, p.y, p.z};
    }
  }
}

// ---------------------------------------------------------------------------
// k_rollout: body-frame forward simulation of every sample of the shard
// (generateTrajectory + computeNewPositions of the three theories).  It depends on
// neither the robot pose nor the cloud, so it runs as extra workgroups of the
// k_bin_count launch (which leaves most of the chip idle).  One workgroup = k.rt trajectories:
//   A  one lane per trajectory: sample decode, generation gates, step count,
//      theta recurrence theta <- float(double(theta) + w*dt)  (2 dependent ops/step)
//   B  all lanes: double sin/cos of every theta_k (and of pi/2 + theta_k for omni);
//      the position increment of step k only depends on theta_k and is formed here
//   C  one lane per trajectory: x,y running sum p <- float(double(p) + inc)
// Output per (trajectory, step): body-frame x,y after the step (float2) and
// cos/sin of the heading after the step (double2, the pose's AngleAxisd rotation).
// ---------------------------------------------------------------------------
#ifdef DDDMR_PHASE_STAMPS
constexpr int kStampSlots = 20;      // 0..11 s_memtime per phase, 12 / 13 s_memrealtime (100 MHz, chip-wide) at start / end, 14 XCC | CU id, 15..18 the last workgroup's hand-off
__device__ unsigned long long g_stamps[16384 * kStampSlots];
#define DDDMR_STAMP(i)                                                                         \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 16384) {                                                \
      g_stamps[blockIdx.x * kStampSlots + (i)] = __builtin_amdgcn_s_memtime();                   \
      if ((i) == 0) {                                                                            \
        g_stamps[blockIdx.x * kStampSlots + 12] = __builtin_amdgcn_s_memrealtime();              \
        g_stamps[blockIdx.x * kStampSlots + 14] =                                                \
            ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) | \
            (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     \
      }                                                                                          \
      if ((i) == 7) g_stamps[blockIdx.x * kStampSlots + 13] = __builtin_amdgcn_s_memrealtime();  \
    }                                                                                            \
  } while (0)
#define DDDMR_STAMP_RAW(i)                                                                     \
  do {                                                                                         \
    if ((threadIdx.x & 63) == 0 && threadIdx.x == 0 && blockIdx.x < 16384)                       \
      g_stamps[blockIdx.x * kStampSlots + (i)] = __builtin_amdgcn_s_memtime();                   \
  } while (0)
// rollout / assignment / binning workgroups of the k_bin_count launch: 8 slots per block
__device__ unsigned long long g_rstamps[4096 * 8];
#define DDDMR_RSTAMP(i)                                                                        \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                 \
      g_rstamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();                            \
      if ((i) == 0) g_rstamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();            \
      if ((i) == 3) g_rstamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();            \
    }                                                                                            \
  } while (0)
#else
#define DDDMR_STAMP(i) do { } while (0)
#define DDDMR_STAMP_RAW(i) do { } while (0)
#define DDDMR_RSTAMP(i) do { } while (0)
#endif

struct TrajInfo {      // per trajectory, 32 bytes
  float vx, vy, w;
  int steps;           // 0 = not generated
  double dt;
  int over;            // step count exceeded the context's max_steps (capacity error)
  int pad;
};

__host__ __device__ inline size_t rollout_lds_bytes(int rt, int max_steps) {
  return (size_t)rt * (size_t)(max_steps + 1) * 16 + 16;   // one 16-byte slot per (trajectory, step): theta, then the increment
}

template <int kThreads>
__device__ __forceinline__ void rollout_block(const DevTick& k, const int block, const float* __restrict__ axes,
                                              const float4* __restrict__ samples, TrajInfo* __restrict__ info,
                                              double2* __restrict__ st_sc, float2* __restrict__ st_xy,
                                              unsigned char* roll_lds) {
  const int S1 = k.max_steps + 1;
  const int rt = k.rt;
  // [rt][S1] slots of 16 bytes.  Phase A leaves theta_s in ONE float of the slot, phase B (the lane that read it)
  // overwrites the slot with the step's increment: 16 bytes per pair instead of 20, and the LDS footprint decides
  // whether two 1024-lane workgroups of k_bin_count share a CU.  Trajectory j keeps its theta in word (j >> 4) & 3,
  // so the 64 lanes of phase A (slot stride S1 * 4 words, S1 odd) write 64 different banks.
  double2* inc = reinterpret_cast<double2*>(roll_lds);
  __shared__ int steps_s[kRolloutMax];
  __shared__ float vel_s[kRolloutMax][3];
  __shared__ double dt_s[kRolloutMax];
  const int tid = threadIdx.x;
  const int l0 = block * rt;                          // first local trajectory of this workgroup
  const int nt = min(rt, k.n_local - l0);
  const bool omni = (k.kind == DDDMR_THEORY_OMNI_SIMPLE);
  DDDMR_RSTAMP(0);

  // ---- phase A ----
  if (tid < nt) {
    const int li = l0 + tid;
    const int gi = k.begin + li;
    float vx, vy, w;
    if (k.list_mode) {
      const float4 sm = samples[gi];
      vx = sm.x; vy = sm.y; w = sm.z;
    } else {
      const int ith = gi % k.nth;
      const int r = gi / k.nth;
      const int iy = r % k.ny;
      const int ix = r / k.ny;
      if (k.axes_inline) {
        vx = k.axes_inl[ix];
        vy = k.axes_inl[k.ay_ofs + iy];
        w = k.axes_inl[k.ath_ofs + ith];
      } else {
        vx = axes[ix];
        vy = axes[k.ay_ofs + iy];
        w = axes[k.ath_ofs + ith];
      }
    }
    int over = 0;
    const double eps = 1e-4;
    bool ok = true;
    double vmag, sim_time = k.sim_time;
    if (k.kind == DDDMR_THEORY_DD_SIMPLE) {
      // dd_simple_trajectory_generator_theory.cpp:364-371
      vmag = fabs((double)vx);
      if ((k.min_vel_x >= 0 && vmag + eps < k.min_vel_x) &&
          (k.min_vel_theta >= 0 && fabs((double)w) + eps < k.min_vel_theta)) ok = false;
      if (k.max_vel_x >= 0 && vmag - eps > k.max_vel_x) ok = false;
    } else if (omni) {
      // omni_simple_trajectory_generator_theory.cpp:387-411
      vmag = hypot((double)vx, (double)vy);
      if ((k.min_vel_trans >= 0 && vmag + eps < k.min_vel_trans) &&
          (k.min_vel_theta >= 0 && fabs((double)w) + eps < k.min_vel_theta)) ok = false;
      if (k.max_vel_trans >= 0 && vmag - eps > k.max_vel_trans) ok = false;
      if (k.allowed_max > 0.0 && vmag - eps > k.allowed_max) ok = false;
    } else {
      // dd_rotate_inplace_theory.cpp:337: one full turn
      vmag = fabs((double)vx);
      sim_time = 6.28 / fabs((double)w);
    }
    int ns = 0;
    if (ok) {
      if (k.fixed_steps > 0) {
        ns = k.fixed_steps;
      } else {
        const double sd = vmag * sim_time;
        const double sa = fabs((double)w) * sim_time;
        ns = (int)ceil(fmax(sd / k.sim_gran, sa / k.ang_gran));
      }
      if (ns > k.max_steps) {       // capacity error, reported to the host by k_score
        over = 1;
        ns = 0;
      }
    }
    const double dt = ns > 0 ? sim_time / (double)ns : 0.0;
    TrajInfo ti;
    ti.vx = vx; ti.vy = vy; ti.w = w;
    ti.steps = ns;
    ti.dt = dt;
    ti.over = over;
    ti.pad = 0;
    info[li] = ti;
    steps_s[tid] = ns;
    vel_s[tid][0] = vx; vel_s[tid][1] = vy; vel_s[tid][2] = w;
    dt_s[tid] = dt;
    // theta_{k+1} = float(theta_k + w*dt)   (computeNewPositions, dd_simple...cpp:457-464)
    float* row = reinterpret_cast<float*>(inc + (size_t)tid * S1) + ((tid >> 4) & 3);
    float a = 0.f;
    row[0] = 0.f;
    const double wdt = (double)w * dt;
    for (int s = 1; s <= ns; ++s) {
      a = (float)((double)a + wdt);
      row[4 * s] = a;
    }
  }
  __syncthreads();
  DDDMR_RSTAMP(1);

  // ---- phase B ----
  // (j, s) of a lane's items advance by a fixed stride: one integer division per lane, not one per item
  const int stride_j = kThreads / S1, stride_s = kThreads - stride_j * S1;
  int bj = tid / S1, bs = tid - bj * S1;
  for (int idx = tid; idx < nt * S1; idx += kThreads) {
    const int j = bj, s = bs;
    bj += stride_j;
    bs += stride_s;
    if (bs >= S1) { bs -= S1; ++bj; }
    const int ns = steps_s[j];
    if (s <= ns) {
      const double ang = (double)reinterpret_cast<const float*>(inc + idx)[(j >> 4) & 3];
      double sn, cs;
      sincos_heading(ang, &sn, &cs);
      // heading after step s-1 = theta_s: the pose's rotation (dd_simple...cpp:416)
      if (s >= 1) st_sc[(size_t)(l0 + j) * k.max_steps + (s - 1)] = make_double2(cs, sn);
      if (s < ns) {
        const float vx = vel_s[j][0], vy = vel_s[j][1];
        const float cf = (float)cs, sf = (float)sn;      // cos/sin(float) overloads
        double ix = (double)fmul(vx, cf), iy = (double)fmul(vx, sf);
        if (omni) {
          double s2, c2;
          sincos_quarter_ahead(ang, sn, cs, &s2, &c2);   // cos/sin(M_PI_2 + theta), omni_simple...cpp:501-502
          ix += (double)vy * c2;
          iy += (double)vy * s2;
        }
        inc[idx] = make_double2(ix * dt_s[j], iy * dt_s[j]);
      }
    }
  }
  __syncthreads();
  DDDMR_RSTAMP(2);

  // ---- phase C ----  (x, y overwrite the consumed increment slots; a coalesced copy-out follows)
  // The x and the y recurrence of a trajectory are independent chains of cvt / add / cvt: lanes [0, 128) run the x
  // chains, lanes [128, 256) the y chains, i.e. DIFFERENT waves (on different SIMDs), which halves the dependent
  // issue per step.  Each lane reads its double (words 0-1 or 2-3 of the slot) and writes the float position over
  // the low word of what it read, so the two chains never touch each other's bytes.
  static_assert(kThreads >= 2 * kRolloutMax, "phase C needs a lane per chain");
  {
    const int cj = tid & (kRolloutMax - 1), comp = tid >> 7;      // kRolloutMax == 128
    if (tid < 2 * kRolloutMax && cj < nt) {
      const int ns = steps_s[cj];
      double* ir = reinterpret_cast<double*>(inc + (size_t)cj * S1) + comp;
      float p = 0.f;
      int s = 0;
      for (; s + 4 <= ns; s += 4) {
        const double i0 = ir[2 * s], i1 = ir[2 * s + 2], i2 = ir[2 * s + 4], i3 = ir[2 * s + 6];
        p = (float)((double)p + i0); *reinterpret_cast<float*>(ir + 2 * s) = p;
        p = (float)((double)p + i1); *reinterpret_cast<float*>(ir + 2 * s + 2) = p;
        p = (float)((double)p + i2); *reinterpret_cast<float*>(ir + 2 * s + 4) = p;
        p = (float)((double)p + i3); *reinterpret_cast<float*>(ir + 2 * s + 6) = p;
      }
      for (; s < ns; ++s) {
        p = (float)((double)p + ir[2 * s]);
        *reinterpret_cast<float*>(ir + 2 * s) = p;
      }
    }
  }
  __syncthreads();
  bj = tid / S1;
  bs = tid - bj * S1;
  for (int idx = tid; idx < nt * S1; idx += kThreads) {
    const int j = bj, s = bs;
    bj += stride_j;
    bs += stride_s;
    if (bs >= S1) { bs -= S1; ++bj; }
    if (s < steps_s[j]) {
      const float* slot = reinterpret_cast<const float*>(inc + idx);
      st_xy[(size_t)(l0 + j) * k.max_steps + s] = make_float2(slot[0], slot[2]);
    }
  }
  DDDMR_RSTAMP(3);
}

// Assignment blocks: counting sort of the shard's trajectories by last tick's load,
// heaviest first, then dealt to the tiles in snake order (rank r -> round r / tiles,
// tile r % tiles, reversed in odd rounds).  Slot [round * n_tiles + tile] of assign[]
// is what k_score reads.  Big shards are cut into k.assign_groups independent
// groups, one workgroup each: group g owns the trajectories AND the tiles congruent
// to g (n_tiles is a multiple of the group count), so every group deals a
// representative sample of the shard to its own tiles.  Whatever the loads hold
// (first tick: zeros), the result is a permutation.
// Slots of the deal.  Round r of the deal is nb_tiles slots wide while r < r0 and n_tiles wide from then on: workgroup b <
// nb_tiles owns one slot of every round (up to `tile` trajectories), a workgroup beyond that only of the rounds r >= r0
// (tile - r0 trajectories).  With DDDMR_TAIL_ROUND=1 a shard that needs several rounds of resident workgroups is laid out
// so that the workgroups with all their trajectories fill whole rounds and ONE last round of short workgroups takes the
// rest (measured: no gain, rollout_engine.hip; the default keeps nb_tiles = n_tiles, r0 = 0).  Slots are
// numbered round by round, so slot s of round r belongs to workgroup s - base(r), and base(r) is a multiple of the
// assignment groups: group g owns exactly the slots = g (mod groups), as before.
__device__ __forceinline__ int tile_round_base(const DevTick& k, int r) {
  return r < k.r0 ? r * k.nb_tiles : k.r0 * k.nb_tiles + (r - k.r0) * k.n_tiles;
}
__device__ __forceinline__ int tile_slot(const DevTick& k, int b, int j) {     // j-th slot of workgroup b
  return tile_round_base(k, b < k.nb_tiles ? j : j + k.r0) + b;
}
constexpr int kAssignPer = 4;          // trajectories per lane of an assignment block (held in registers)
template <int kThreads>
__device__ __forceinline__ void assign_block(const DevTick& k, const int grp, const uint32_t* __restrict__ load,
                                             uint32_t* __restrict__ assign) {
  __shared__ uint32_t hist[kLoadClasses];
  __shared__ uint32_t wsum[kThreads / 64];
  __shared__ uint32_t mx_s, mn_s;
  static_assert(kLoadClasses == kThreads, "one class per lane in the scan");
  const int tid = threadIdx.x;
  const int G = k.assign_groups;
  const int n = k.n_local, tl = k.n_tiles / G, tlb = k.nb_tiles / G;     // slots of this group per late / early round
  const int ng = (n - grp + G - 1) / G;              // trajectories (and slots) of this group
  // r-th slot of the group in deal order (its slots are the indices = grp (mod G), see tile_slot()): round, position
  // in the round, and the snake: odd rounds run backwards while the round is complete
  auto group_slot = [&](int r) {
    const int early = k.r0 * tlb;
    int start, width, round;
    if (r < early) { round = r / tlb; start = round * tlb; width = tlb; }
    else { round = k.r0 + (r - early) / tl; start = early + (round - k.r0) * tl; width = tl; }
    const int pos = r - start;
    const int bl = ((round & 1) && start + width <= ng) ? width - 1 - pos : pos;
    return grp + G * (start + bl);
  };
  DDDMR_RSTAMP(0);
  hist[tid] = 0;
  if (tid == 0) { mx_s = 1; mn_s = 0xFFFFFFFFu; }
  uint32_t v[kAssignPer];
  uint32_t mx = 1, mn = 0xFFFFFFFFu;
#pragma unroll
  for (int e = 0; e < kAssignPer; ++e) {
    const int m = tid + e * kThreads;
    v[e] = m < ng ? load[grp + G * m] : 0u;
    if (m < ng) { mx = max(mx, v[e]); mn = min(mn, v[e]); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
  }
  __syncthreads();
  if ((tid & 63) == 0) { atomicMax(&mx_s, mx); atomicMin(&mn_s, mn); }
  __syncthreads();
  if (mx_s <= mn_s) {
    // all loads equal (e.g. no cloud): nothing to balance, deal in index order
#pragma unroll
    for (int e = 0; e < kAssignPer; ++e) {
      const int m = tid + e * kThreads;
      if (m < ng) assign[grp + G * m] = (uint32_t)(grp + G * m);
    }
    return;
  }
  const float scale = (float)(kLoadClasses - 1) / (float)mx_s;
#pragma unroll
  for (int e = 0; e < kAssignPer; ++e) {
    const int m = tid + e * kThreads;
    if (m < ng) atomicAdd(&hist[kLoadClasses - 1 - min(kLoadClasses - 1, (int)((float)v[e] * scale))], 1u);
  }
  __syncthreads();
  // exclusive scan of the classes, one per lane
  const uint32_t cnt = hist[tid];
  const uint32_t inc = wave_incl_scan_u32(cnt);
  if ((tid & 63) == 63) wsum[tid >> 6] = inc;
  __syncthreads();
  uint32_t base = inc - cnt;
  for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
  hist[tid] = base;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kAssignPer; ++e) {
    const int m = tid + e * kThreads;
    if (m < ng) {
      const int c = kLoadClasses - 1 - min(kLoadClasses - 1, (int)((float)v[e] * scale));
      const int r = (int)atomicAdd(&hist[c], 1u);
      assign[group_slot(r)] = (uint32_t)(grp + G * m);
    }
  }
  DDDMR_RSTAMP(3);
}

__global__ __launch_bounds__(kBinThreads) void k_assign(DevTick k, const uint32_t* __restrict__ load,
                                                        uint32_t* __restrict__ assign) {
  assign_block<kBinThreads>(k, (int)blockIdx.x, load, assign);
}

// Stand-alone launch (empty cloud: there is no k_bin_count to ride along with).
__global__ __launch_bounds__(256) void k_rollout(DevTick k, const float* __restrict__ axes,
                                                 const float4* __restrict__ samples,
                                                 TrajInfo* __restrict__ info, double2* __restrict__ st_sc,
                                                 float2* __restrict__ st_xy) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  rollout_block<256>(k, (int)blockIdx.x, axes, samples, info, st_sc, st_xy, dyn_lds);
}

// Crop the cloud to the local costmap tile and count points per cell; the LAST
// binning workgroup to finish (device-scope ticket) scans the counters, so binning is
// two launches, not three.  Also resets the argmin key / capacity flag of the tick.
// The workgroups after the k.bin_blocks binning ones run the tile assignment (a few)
// and the body-frame rollout:
// a cloud of ~10^4 points keeps only ~10 binning workgroups busy, the rollout fills
// the rest of the chip for free and needs no cross-stream dependency.
__global__ __launch_bounds__(kBinThreads, 8) void k_bin_count(DevTick k, const float4* __restrict__ cloud,
                                                   uint32_t* __restrict__ cell_count,
                                                   uint32_t* __restrict__ cell_start,
                                                   uint2* __restrict__ pt_slot, uint32_t* __restrict__ ticket,
                                                   int64_t* __restrict__ best_key,
                                                   uint32_t* __restrict__ overflow, const float* __restrict__ axes,
                                                   const float4* __restrict__ samples, TrajInfo* __restrict__ info,
                                                   double2* __restrict__ st_sc, float2* __restrict__ st_xy,
                                                   const uint32_t* __restrict__ load, uint32_t* __restrict__ assign) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  // Longest first: the assignment workgroups (only launched with use_assign), the rollout, then the binning
  // workgroups -- when the launch does not fit the chip in one round, the short ones fill the tail.
  const int n_assign = k.use_assign ? k.assign_groups : 0;
  if ((int)blockIdx.x < n_assign) {
    assign_block<kBinThreads>(k, (int)blockIdx.x, load, assign);
    return;
  }
  if ((int)blockIdx.x < n_assign + k.roll_blocks) {
    rollout_block<kBinThreads>(k, (int)blockIdx.x - n_assign, axes, samples, info, st_sc, st_xy, dyn_lds);
    return;
  }
  const int bin_block = (int)blockIdx.x - n_assign - k.roll_blocks;
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  __shared__ uint32_t is_last;
  const int stride = k.bin_blocks * blockDim.x;
  DDDMR_RSTAMP(0);
  // Every lane bins up to kBinPer points per pass, all their loads and counting atomics in
  // flight together: a quarter of the workgroups (dispatching a 1024-lane workgroup costs
  // ~12 ns, and the launch also carries the rollout's) at the latency of one point.
  for (int base = bin_block * blockDim.x + threadIdx.x; base < k.n_points; base += stride * kBinPer) {
    float4 p[kBinPer];
    uint2 slot[kBinPer];
#pragma unroll
    for (int m = 0; m < kBinPer; ++m) {
      const int i = base + m * stride;
      p[m] = i < k.n_points ? cloud[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int m = 0; m < kBinPer; ++m) {
      const int i = base + m * stride;
      slot[m] = make_uint2(0xFFFFFFFFu, 0u);
      const bool in = i < k.n_points && p[m].x >= k.rmin[0] && p[m].x <= k.rmax[0] && p[m].y >= k.rmin[1] &&
                      p[m].y <= k.rmax[1] && p[m].z >= k.rmin[2] && p[m].z <= k.rmax[2];
      if (in) {
        const int c = cell_of(k, p[m].x, p[m].y, p[m].z);
        slot[m].x = (uint32_t)c;
        slot[m].y = atomicAdd(&cell_count[c], 1u);
      }
    }
#pragma unroll
    for (int m = 0; m < kBinPer; ++m) {
      const int i = base + m * stride;
      if (i < k.n_points) pt_slot[i] = slot[m];
    }
  }
  // Ticket.  The only data handed to the last workgroup are the cell counters, and
  // those are touched exclusively by device-scope atomics (returned => performed)
  // and read back with device-scope atomic loads, so no cache write-back /
  // invalidate is needed: every wave drains its atomics, barrier, one relaxed
  // device-scope add per workgroup.  (DDDMR_HANDOFF_ORDER: see the hand-off note above k_score.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, DDDMR_HANDOFF_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == (uint32_t)k.bin_blocks - 1) ? 1u : 0u;
    if (is_last) {
      *ticket = 0;            // next tick
      best_key[0] = kKeyNone;
      best_key[1] = kKeyNone;
      best_key[2] = 0;            // collided-trajectory count of the tick
      *overflow = 0;
    }
  }
  __syncthreads();
  DDDMR_RSTAMP(1);
  if (is_last) scan_cells(k, cell_count, cell_start, wave_sum, &carry_s);
  DDDMR_RSTAMP(3);
}

// ---------------------------------------------------------------------------
// fused rollout + critics + argmin
//
// One workgroup scores a tile of `tile` consecutive trajectories (~256
// (trajectory, step) pairs).  Phases:
//   A  theta recurrence (one lane per trajectory; 2 dependent ops per step)
//   B  double sin/cos of every theta_k, all lanes
//   C  x,y recurrence (one lane per trajectory)
//   D1 one lane per pair: pose, path critics, cuboid -> OBB record in LDS,
//      candidate cell range
//   D2 one lane per (pair, cell row): the row's contiguous run of cell-sorted
//      points; workgroup exclusive scan of the 8-point work items
//   D3 load-balanced walk: every lane takes an equal share of the flattened
//      item list; trajectory-level early exit through LDS flags
//   E  stacked scoring + packed-key wave min-reduction + one atomicMin
// ---------------------------------------------------------------------------
// Diagnostic build only (make diag): per-workgroup phase timestamps, written to a
// buffer nothing else reads (cdna_hip_programming.md 7, In-kernel stamps).

struct TrajHead {     // per-trajectory header in LDS
  float vx, vy, w;
  int steps;          // 0 = not generated
  double dt;
  int pair_base;      // first flattened (traj,step) pair of this trajectory
  int hit_box;        // CollisionModel verdict
  int hit_mm;         // CollisionMinMaxModel verdict
  int pad;
  double pp_dist;     // PurePursuitModel: |translation| of the pose difference
  double pp_yaw;      // PurePursuitModel: folded yaw of the pose difference
  double stick_sum;   // StickPathModel: sum of the per-step 1-NN distances
  int li;             // local trajectory index (row of the k_rollout state arrays)
  int walked;         // collision work items walked for this trajectory (load feedback)
};

// OBB record: [0..2] centre, [3..11] axes, [12..14] half extents, [15] cx0|cx1, [16] cy0|cy1,
// [17] trajectory | radius-skip flag, then (only when some pair can need the radius test)
// [18..20] pose, then (only for CollisionMinMaxModel) the world AABB
constexpr int kRecBase = 18;
// (an ODD stride -- 19 words, no two lanes of D1 / D2 on one LDS bank -- was measured and changes nothing: C3 k_score
// 127.1 -> 128.3 us; the walk's record reads are mostly same-pair broadcasts)
__host__ __device__ inline int rec_words_of(bool rec_pose, bool want_mm) { return kRecBase + (rec_pose ? 3 : 0) + (want_mm ? 6 : 0); }
constexpr int kRows = 8;            // y-rows of cells one cuboid AABB may span (host sizes the cells for it)
#ifndef DDDMR_ITEM
#define DDDMR_ITEM 16
#endif
constexpr int kItem = DDDMR_ITEM;   // points per work item of the collision walk (8 -> 16: -5 % at C2, neutral at C3/C4)
constexpr int kTabCap = 4096;       // (gnx+1)*gny row-run boundaries staged in LDS when they fit

// dynamic LDS carve, see k_score (rows are max_steps+1 long):
__host__ __device__ inline size_t score_lds_bytes(int tile, int max_steps, int m, int rec_words, int tab_entries,
                                                  int rows_cap) {
  const size_t S1 = (size_t)max_steps + 1;
  const size_t Q = (size_t)tile * (size_t)max_steps;
  size_t b = 0;
  b += sizeof(TrajHead) * (size_t)tile;
  b = (b + 15) & ~(size_t)15;
  b += 16 * (size_t)(m > 0 ? m : 1);                 // plan
  b += 4 * (size_t)tile * S1;                        // dist
  b += 16 * Q;                                       // pose positions (float) of every pair
  b += 4 * (size_t)rec_words * Q;                    // OBB records
  b += 4 * (size_t)tab_entries;                      // costmap row-run index (cell_start slice), 0 = not staged
  b = (b + 15) & ~(size_t)15;
  b += 4 * (Q * (size_t)rows_cap + 1) + 8 * (Q * (size_t)rows_cap) + 16;   // pref, seg_p, seg_len
  b += 64;                                           // scan scratch
  return (b + 15) & ~(size_t)15;
}

// Two points per instruction: gfx950's packed FP32 ops (v_pk_add_f32 / v_pk_mul_f32)
// do the same IEEE operations as the scalar form, two lanes-values at a time, and
// with contraction off no multiply-add is fused -- so the results are bit-identical
// to l2_simple / dot3 above at half the VALU issue cost.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 l2_simple2(f2 ax, f2 ay, f2 az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  f2 d = ax - bx;
  f2 r = d * d;
  d = ay - by;
  r = r + d * d;
  d = az - bz;
  r = r + d * d;
  return r;
}
__device__ __forceinline__ f2 dot3_2(f2 dx, f2 dy, f2 dz, float a, float b, float c) {
#pragma clang fp contract(off)
  return (dx * a + dy * b) + dz * c;
}
// box (and radius) verdicts of two cloud points against one OBB record:
// bit 0 / bit 1 = point a / b collides (collision_model.cpp:122-139)
__device__ __forceinline__ int box_test2(const float* r, const Pt3 a, const Pt3 b, bool use_radius) {
#pragma clang fp contract(off)
  const f2 x = {a.x, b.x}, y = {a.y, b.y}, z = {a.z, b.z};
  const f2 dx = x - r[0], dy = y - r[1], dz = z - r[2];
  const f2 xv = __builtin_elementwise_abs(dot3_2(dx, dy, dz, r[3], r[4], r[5]));
  const f2 yv = __builtin_elementwise_abs(dot3_2(dx, dy, dz, r[6], r[7], r[8]));
  const f2 zv = __builtin_elementwise_abs(dot3_2(dx, dy, dz, r[9], r[10], r[11]));
  bool h0 = xv.x <= r[12] && yv.x <= r[13] && zv.x <= r[14];
  bool h1 = xv.y <= r[12] && yv.y <= r[13] && zv.y <= r[14];
  if (use_radius) {
    const f2 d2 = l2_simple2(x, y, z, r[15], r[16], r[17]);   // FLANN: point - query squared the same
    h0 = h0 && d2.x < 1.0f;
    h1 = h1 && d2.y < 1.0f;
  }
  return (h0 ? 1 : 0) | (h1 ? 2 : 0);
}

// translation of trans_gbl2traj = pos_af3 * [Rz(theta), (x, y, 0)] (one definition so
// that every phase rounds it identically)
__device__ __forceinline__ void pose_translation(const DevTick& k, float2 bxy, double T[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) T[i] = k.R[3 * i + 0] * (double)bxy.x + k.R[3 * i + 1] * (double)bxy.y + k.t[i];
}

// OBB record words [3..14]: the cuboid's vertices blb, brb, blt, flb as phase D1 left them -> axes (x: blb->flb,
// y: blb->brb, z: blb->blt, collision_model.cpp:97-110) and half extents (:112-115), in place.
__device__ __forceinline__ void obb_axes_in_place(float* r) {
  float v[4][3];
#pragma unroll
  for (int a = 0; a < 4; ++a) { v[a][0] = r[3 + 3 * a]; v[a][1] = r[4 + 3 * a]; v[a][2] = r[5 + 3 * a]; }
  const int vi[3] = {3, 1, 2};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float ex = fsub(v[vi[a]][0], v[0][0]), ey = fsub(v[vi[a]][1], v[0][1]), ez = fsub(v[vi[a]][2], v[0][2]);
    const float len = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
    // The reference divides in double, (float)((double)e / (2. * h)) with h = (double)len / 2.:
    // 2h == len exactly, and rounding a double quotient of two floats to float equals
    // the correctly rounded float division (53 >= 2*24 + 2 bits: double rounding is
    // innocuous for division) -- so the IEEE float divide gives the same bits for less.
    r[12 + a] = len * 0.5f;                  // len/2 is exact in float
    r[3 + 3 * a + 0] = ex / len;
    r[3 + 3 * a + 1] = ey / len;
    r[3 + 3 * a + 2] = ez / len;
  }
}

// Winner decode (local_planner.cpp:447-480), run by ONE wave once every k_score workgroup has filed its keys:
// either wave 0 of the workgroup that drew the last ticket (small shards: no extra launch) or k_finalize.
//
// Exactness: the reference compares full doubles (`cost_ <= minimum_cost`, :460).  The key's winner i is the
// highest index among the costs that share the minimum's top 40 bits; if cost[i] IS the minimum (its bits equal
// the reduced cost word -- always, unless two costs differ by less than 3.7e-9 relative without being equal) then
// i is also the highest index among the exactly minimal costs, i.e. the reference's answer.  Otherwise the wave
// rescans the shard's costs for the exact minimum (rare; 64 lanes, device-scope loads).
__device__ __forceinline__ void decode_winner(const DevTick& k, const int lane, const int64_t* __restrict__ best_key,
                                              const double* __restrict__ costs, const float4* __restrict__ samples_out,
                                              const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ overflow,
                                              DevResult* __restrict__ result, int64_t* __restrict__ words_out) {
  const int64_t kmin = __hip_atomic_load(best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t cmin = __hip_atomic_load(best_key + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long* cost_words = reinterpret_cast<const unsigned long long*>(costs);
  int li = -1;
  if (kmin != kKeyNone) {
    li = key_index(kmin) - k.begin;
    const long long cb = (long long)__hip_atomic_load(cost_words + li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cb != (long long)cmin) {
      int found = -1;
      for (int i = lane; i < k.n_local; i += 64)
        if ((long long)__hip_atomic_load(cost_words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (long long)cmin) found = i;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) found = max(found, __shfl_xor(found, o, 64));
      li = found;
    }
  }
  if (lane == 0) {
    DevResult r;
    r.index = li >= 0 ? k.begin + li : -1;
    r.cost = -1.0;
    r.vx = r.vy = r.wz = 0.f;
    r.key = kKeyNone;
    if (li >= 0) {
      r.cost = __longlong_as_double((long long)cmin);
      r.key = pack_key(r.cost, (uint32_t)r.index);
      const float* so = reinterpret_cast<const float*>(samples_out + li);
      r.vx = __hip_atomic_load(so + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r.vy = __hip_atomic_load(so + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r.wz = __hip_atomic_load(so + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    r.n_binned = cell_start[k.n_cells];
    r.overflow = __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.seq = 0;
    {
      // sampled share of collided trajectories, scaled to the shard
      const unsigned long long w2 = (unsigned long long)__hip_atomic_load(best_key + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long sampled = w2 >> 32, collided = w2 & 0xFFFFFFFFull;
      r.n_collided = sampled ? (uint32_t)(collided * (unsigned long long)k.n_local / sampled) : 0u;
    }
    *result = r;
    if (words_out) {
      // multi-rank context: this shard's (cost bits, -index) slot of the all-reduce that follows on
      // the stream; `result` then is a device-side staging record and k_resolve tells the host
      words_out[0] = li >= 0 ? (int64_t)cmin : kKeyNone;
      words_out[1] = li >= 0 ? -(int64_t)r.index : kKeyNone;
    } else {
      __threadfence_system();
      __hip_atomic_store(&result->seq, k.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Single-round shards: wave 0 of the last workgroup reduces the workgroups' slots (see the tail of k_score) --
// minimum cost as full doubles, equal costs -> highest index -- and publishes the result.  The loads of up to
// 8 slots per lane (a launch of <= 512 workgroups) are all issued before the first use, beside the two other words
// the result needs: ONE memory round trip, then six shuffle steps.
__device__ __forceinline__ void reduce_slots(const DevTick& k, const int lane, const int n_slots,
                                             const unsigned long long* __restrict__ slots,
                                             const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ overflow,
                                             DevResult* __restrict__ result, int64_t* __restrict__ words_out) {
  constexpr int kPer = 8;
  const uint32_t n_binned = cell_start[k.n_cells];
  const uint32_t over = __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  long long bc = (long long)kKeyNone;
  long long bi = -1;
  unsigned long long b2 = 0, b3 = 0;
  uint32_t n_coll = 0;
  for (int base = 0; base < n_slots; base += 64 * kPer) {
    unsigned long long w[kPer][kSlotWords];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int i = base + u * 64 + lane;
      const unsigned long long* sl = slots + (size_t)(i < n_slots ? i : 0) * kSlotWords;
#pragma unroll
      for (int q = 0; q < kSlotWords; ++q) w[u][q] = __hip_atomic_load(sl + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int i = base + u * 64 + lane;
      if (i < n_slots) {
        const long long c = (long long)w[u][0], gi = (long long)w[u][1];
        n_coll += (uint32_t)((w[u][3] >> 32) & 0xFFu);
        if (c < bc || (c == bc && gi > bi)) { bc = c; bi = gi; b2 = w[u][2]; b3 = w[u][3]; }
      }
    }
  }
  DDDMR_STAMP_RAW(16);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long oc = __shfl_xor(bc, o, 64), oi = __shfl_xor(bi, o, 64);
    const unsigned long long o2 = __shfl_xor(b2, o, 64), o3 = __shfl_xor(b3, o, 64);
    n_coll += __shfl_xor(n_coll, o, 64);
    if (oc < bc || (oc == bc && oi > bi)) { bc = oc; bi = oi; b2 = o2; b3 = o3; }
  }
  if (lane == 0) {
    const bool any = bc != (long long)kKeyNone;
    DevResult r;
    r.index = any ? (int32_t)bi : -1;
    r.cost = any ? __longlong_as_double(bc) : -1.0;
    r.key = any ? pack_key(r.cost, (uint32_t)r.index) : kKeyNone;
    r.vx = any ? __uint_as_float((uint32_t)b2) : 0.f;
    r.vy = any ? __uint_as_float((uint32_t)(b2 >> 32)) : 0.f;
    r.wz = any ? __uint_as_float((uint32_t)b3) : 0.f;
    r.n_binned = n_binned;
    r.overflow = over;
    r.seq = 0;
    r.n_collided = n_coll;          // exact here (every workgroup reports), sampled on multi-round shards
    DDDMR_STAMP_RAW(17);
    *result = r;
    if (words_out) {
      // multi-rank context: this shard's (cost bits, -index) slot of the all-reduce that follows on
      // the stream; `result` then is a device-side staging record and k_resolve tells the host
      words_out[0] = any ? (int64_t)bc : kKeyNone;
      words_out[1] = any ? -(int64_t)r.index : kKeyNone;
    } else {
      __threadfence_system();
      __hip_atomic_store(&result->seq, k.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    DDDMR_STAMP_RAW(18);
  }
}

// kLean: the common case -- no min-max critic, no pair that needs the 1 m radius test (the cuboid lies
// inside the search ball) and a cuboid that is a body-frame box -- as compile-time facts: the walk loses its
// radius / min-max code, the records their optional words, phase D1 its general-vertex-list path.
// kProbe: the collision walk starts with a probe round (compile-time: as a run-time flag it cost the walk its
// schedule -- 121 -> 107 VGPRs and +10 % time).
template <int kScoreThreads, bool kLean, bool kProbe>
__global__ __launch_bounds__(kScoreThreads, kScoreThreads >= 512 ? DDDMR_SCORE_WPE : DDDMR_SCORE_WPE_256) void k_score(
    DevTick k, const TrajInfo* __restrict__ info, const double2* __restrict__ st_sc, const float2* __restrict__ st_xy,
    const float4* __restrict__ plan_xyz, const uint32_t* __restrict__ cell_start,
    const Pt3* __restrict__ sorted, double* __restrict__ costs, int32_t* __restrict__ steps_out,
    float4* __restrict__ samples_out, int64_t* __restrict__ best_key, uint32_t* __restrict__ overflow,
    uint32_t* __restrict__ ticket, DevResult* __restrict__ result, const uint32_t* __restrict__ assign,
    uint32_t* __restrict__ traj_load, int64_t* __restrict__ words_out, const uint32_t* __restrict__ row_tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tile = k.tile;
  const int S1 = k.max_steps + 1;
  const int Qcap = tile * k.max_steps;
  const bool need_box = k.want_collision != 0, need_mm = !kLean && k.want_minmax != 0;
  const bool rec_pose = !kLean && k.rec_pose != 0;
  const bool box_fast = kLean || k.box_fast != 0;      // (the lean variant is only launched for body-frame boxes)
  const int rec_words = rec_words_of(rec_pose, need_mm);
  const int mm_ofs = kRecBase + (rec_pose ? 3 : 0);
  size_t ofs = 0;
  TrajHead* head = reinterpret_cast<TrajHead*>(lds_raw);
  ofs += sizeof(TrajHead) * (size_t)tile;
  ofs = (ofs + 15) & ~(size_t)15;
  float4* plan = reinterpret_cast<float4*>(lds_raw + ofs);
  ofs += 16 * (size_t)(k.m > 0 ? k.m : 1);
  float4* ppos = reinterpret_cast<float4*>(lds_raw + ofs);   // pose position of pair q (phase D1 -> P)
  ofs += 16 * (size_t)Qcap;
  float* dist = reinterpret_cast<float*>(lds_raw + ofs);
  ofs += 4 * (size_t)tile * S1;
  float* rec = reinterpret_cast<float*>(lds_raw + ofs);
  ofs += 4 * (size_t)rec_words * Qcap;
  uint32_t* tab = reinterpret_cast<uint32_t*>(lds_raw + ofs);
  ofs += 4 * (size_t)k.tab_entries;
  ofs = (ofs + 15) & ~(size_t)15;
  // collision segments (phases D2..D3)
  uint32_t* pref = reinterpret_cast<uint32_t*>(lds_raw + ofs);
  uint32_t* seg_p = pref + ((size_t)Qcap * k.rows_cap + 1);
  uint32_t* seg_len = seg_p + (size_t)Qcap * k.rows_cap;
  ofs += 4 * ((size_t)Qcap * k.rows_cap + 1) + 8 * ((size_t)Qcap * k.rows_cap) + 16;
  ofs = (ofs + 7) & ~(size_t)7;
  unsigned long long* wsum64 = reinterpret_cast<unsigned long long*>(lds_raw + ofs);

  __shared__ int alive_pairs_s;
  __shared__ uint32_t t_item0[kMaxTile + 1];   // first collision item of each trajectory of the tile
  __shared__ uint32_t t_ubase[kMaxTile + 1];   // ... and of the undecided ones, packed, per walk round
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  // Tile b scores the trajectories in slots b, b + n_tiles, b + 2 n_tiles, ... of the
  // assignment (load feedback, see assign_block) or, without one, the local
  // trajectories of those indices: neighbours in sample order head the same way and
  // would make whole tiles cheap (open space) or expensive (along a wall); striding
  // mixes them.
  const int tb = (int)blockIdx.x;
  int nt = 0;                                                  // <= tile (see tile_slot())
  for (int j = 0; j < tile; ++j) nt += ((tb < k.nb_tiles || j + k.r0 < tile) && tile_slot(k, tb, j) < k.n_local) ? 1 : 0;

  DDDMR_STAMP(0);
  // ---- stage the prune plan (float xyz, model_shared_data.h:83-91) ----
  for (int i = tid; i < k.m; i += kScoreThreads) plan[i] = plan_xyz[i];

  // ---- phase A: trajectory headers from k_rollout ----
  if (tid < nt) {
    const int slot = tile_slot(k, tb, tid);
    const int li = k.use_assign ? (int)assign[slot] : slot;
    const TrajInfo ti = info[li];
    if (ti.over) atomicOr(overflow, 1u);
    TrajHead h;
    h.vx = ti.vx; h.vy = ti.vy; h.w = ti.w;
    h.steps = ti.steps;
    h.dt = ti.dt;
    h.pair_base = 0;
    h.hit_box = 0; h.hit_mm = 0; h.pad = 0;
    h.pp_dist = 0.0; h.pp_yaw = 0.0; h.stick_sum = 0.0;
    h.li = li;
    h.walked = 0;
    head[tid] = h;
  }
  // the costmap's row-run index is staged meanwhile (independent loads)
  const int tab_n = k.tab_entries;          // (gnx+1)*gny, or 0 when the index does not fit
  const bool tab_staged = tab_n > 0;
  if (tab_staged && k.n_points >= 5 && (need_box || need_mm)) {
    for (int i = tid; i < tab_n; i += kScoreThreads) tab[i] = row_tab[i];       // (built by k_bin_scatter)
  }
  __syncthreads();
  DDDMR_STAMP(1);   // end of phase A
  // pair offsets
  if (tid < 64) {                                   // exclusive prefix of the step counts: one DPP wave scan
    const uint32_t st = tid < nt ? (uint32_t)head[tid].steps : 0u;
    const uint32_t incl = wave_incl_scan_u32(st);
    if (tid < nt) head[tid].pair_base = (int)(incl - st);
  }
  __syncthreads();
  DDDMR_STAMP(2);
  DDDMR_STAMP(3);
  // ---- phase D1: one (trajectory, step) pair per lane ----
  int total_pairs = 0;
  if (nt > 0) total_pairs = head[nt - 1].pair_base + head[nt - 1].steps;
  const bool cloud_ok = k.n_points >= 5;   // collision_model.cpp:53-55
  const bool do_coll = cloud_ok && (need_box || need_mm);
  for (int q = tid; q < total_pairs; q += kScoreThreads) {
    int j = 0;
    while (j + 1 < nt && head[j + 1].pair_base <= q) ++j;
    const int s = q - head[j].pair_base;
    const size_t so = (size_t)head[j].li * k.max_steps + s;
    const double2 cs = st_sc[so];                    // heading after the step
    const float2 bxy = st_xy[so];                    // body-frame position after the step
    const double c = cs.x, sn = cs.y;
    // trans_gbl2traj = pos_af3 * [Rz(theta), (x, y, 0)]
    double L[9], T[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double r0 = k.R[3 * i + 0], r1 = k.R[3 * i + 1], r2 = k.R[3 * i + 2];
      L[3 * i + 0] = r0 * c + r1 * sn;
      L[3 * i + 1] = r1 * c - r0 * sn;
      L[3 * i + 2] = r2;
    }
    pose_translation(k, bxy, T);
    const float px = (float)T[0], py = (float)T[1], pz = (float)T[2];   // trajectory.cpp:69-75
    ppos[q] = make_float4(px, py, pz, 0.f);

    // ---- cuboid -> OBB record (collision_model.cpp:85-115) ----
    if (do_coll) {
      // pcl::transformPointCloud(cuboid, Affine3d): double multiply-add, cast to float.
      // Vertex order blb,brb,blt,flb,...: keep [0..3] for the axes, fold the rest.
      float v[4][3];
      float mnx = 3.402823466e+38f, mny = mnx, mnz = mnx, mxx = -mnx, mxy = -mnx, mxz = -mnx;
      float ccx = 0.f, ccy = 0.f, ccz = 0.f;
      float vmax2 = 0.f;   // farthest vertex from the pose: if < 1 the box lies inside the search ball
      auto world_vertex = [&](int vtx, float& wx, float& wy, float& wz) {
        const double cx = k.cub[3 * vtx + 0], cy = k.cub[3 * vtx + 1], cz = k.cub[3 * vtx + 2];
        wx = (float)(L[0] * cx + L[1] * cy + L[2] * cz + T[0]);
        wy = (float)(L[3] * cx + L[4] * cy + L[5] * cz + T[1]);
        wz = (float)(L[6] * cx + L[7] * cy + L[8] * cz + T[2]);
        mnx = fminf(mnx, wx); mxx = fmaxf(mxx, wx);
        mny = fminf(mny, wy); mxy = fmaxf(mxy, wy);
        mnz = fminf(mnz, wz); mxz = fmaxf(mxz, wz);
        ccx = fadd(ccx, wx); ccy = fadd(ccy, wy); ccz = fadd(ccz, wz);   // centre sum in vertex order
        if (rec_pose) {
          const float ux = wx - px, uy = wy - py, uz = wz - pz;
          vmax2 = fmaxf(vmax2, ux * ux + uy * uy + uz * uz);
        }
      };
      if (box_fast && !rec_pose) {
        // The cuboid is a box in the body frame (x back/front, y left/right, z bottom/top --
        // checked bit for bit on the host), so the 24 products L(i,c) * coordinate take only
        // 18 distinct values and the partial sums L(i,0) x + L(i,1) y only 12: same
        // operations in the same order as the general path, each done once.
        // vertex order blb brb blt flb brt frt flt frb -> (x, y, z) selectors
        //   0:(0,0,0) 1:(0,1,0) 2:(0,0,1) 3:(1,0,0) 4:(0,1,1) 5:(1,1,1) 6:(1,0,1) 7:(1,1,0)
        const double X0 = k.cub[0], X1 = k.cub[9], Y0 = k.cub[1], Y1 = k.cub[4], Z0 = k.cub[2], Z1 = k.cub[8];
        auto row = [&](const int i, float& mn, float& mx, float& cc, float& v0, float& v1, float& v2, float& v3) {
          const double a0 = L[3 * i] * X0, a1 = L[3 * i] * X1;
          const double b0 = L[3 * i + 1] * Y0, b1 = L[3 * i + 1] * Y1;
          const double c0 = L[3 * i + 2] * Z0, c1 = L[3 * i + 2] * Z1;
          const double s00 = a0 + b0, s01 = a0 + b1, s10 = a1 + b0, s11 = a1 + b1;
          const double t = T[i];
          const float w0 = (float)((s00 + c0) + t), w1 = (float)((s01 + c0) + t), w2 = (float)((s00 + c1) + t),
                      w3 = (float)((s10 + c0) + t), w4 = (float)((s01 + c1) + t), w5 = (float)((s11 + c1) + t),
                      w6 = (float)((s10 + c1) + t), w7 = (float)((s11 + c0) + t);
          mn = fminf(fminf(fminf(w0, w1), fminf(w2, w3)), fminf(fminf(w4, w5), fminf(w6, w7)));
          mx = fmaxf(fmaxf(fmaxf(w0, w1), fmaxf(w2, w3)), fmaxf(fmaxf(w4, w5), fmaxf(w6, w7)));
          cc = fadd(fadd(fadd(fadd(fadd(fadd(fadd(fadd(0.f, w0), w1), w2), w3), w4), w5), w6), w7);   // vertex order
          v0 = w0; v1 = w1; v2 = w2; v3 = w3;
        };
        row(0, mnx, mxx, ccx, v[0][0], v[1][0], v[2][0], v[3][0]);
        row(1, mny, mxy, ccy, v[0][1], v[1][1], v[2][1], v[3][1]);
        row(2, mnz, mxz, ccz, v[0][2], v[1][2], v[2][2], v[3][2]);
      } else {
#pragma unroll
        for (int vtx = 0; vtx < 4; ++vtx) world_vertex(vtx, v[vtx][0], v[vtx][1], v[vtx][2]);
#pragma unroll 1
        for (int vtx = 4; vtx < 8; ++vtx) {
          float wx, wy, wz;
          world_vertex(vtx, wx, wy, wz);
        }
      }
      // What the collision critic tests is { d : |d . a_i| <= h_i, i = 1..3 } around the mean of the 8 vertices, with
      // a_i, h_i from the edges e_i = v_i - v_0 (collision_model.cpp:85-115).  For a body-frame box the e_i are
      // orthogonal and that region is the cuboid itself, centre +- e_1/2 +- e_2/2 +- e_3/2.  For any other vertex list
      // the three slabs meet in the DUAL parallelepiped, centre +- g_1 +- g_2 +- g_3 with g_i = (e_j x e_k) |e_i|^2 /
      // (2 det[e_1 e_2 e_3]), which reaches beyond the vertices' bounding box -- in the soak scenario that found this by
      // 3.3 cm, where a point collided in the reference and was never looked at here.  g_i bound the candidate cells
      // (obx, oby) and the radius-skip flag below.
      float obx, oby;      // extent of that region along world x and y
      {
        const float e1x = v[1][0] - v[0][0], e1y = v[1][1] - v[0][1], e1z = v[1][2] - v[0][2];
        const float e2x = v[2][0] - v[0][0], e2y = v[2][1] - v[0][1], e2z = v[2][2] - v[0][2];
        const float e3x = v[3][0] - v[0][0], e3y = v[3][1] - v[0][1], e3z = v[3][2] - v[0][2];
        // ... and, where the 1 m radius test may matter, than the farthest corner of the tested region
        auto far_corner = [&](float g1x, float g1y, float g1z, float g2x, float g2y, float g2z, float g3x, float g3y, float g3z) {
          const float ox = ccx / 8.f - px, oy = ccy / 8.f - py, oz = ccz / 8.f - pz;
#pragma unroll
          for (int corner = 0; corner < 8; ++corner) {
            const float a = (corner & 1) ? 1.0f : -1.0f, b = (corner & 2) ? 1.0f : -1.0f, c3 = (corner & 4) ? 1.0f : -1.0f;
            const float ux = ox + a * g1x + b * g2x + c3 * g3x, uy = oy + a * g1y + b * g2y + c3 * g3y,
                        uz = oz + a * g1z + b * g2z + c3 * g3z;
            vmax2 = fmaxf(vmax2, ux * ux + uy * uy + uz * uz);
          }
        };
        if (box_fast) {
          obx = 0.5f * (fabsf(e1x) + fabsf(e2x) + fabsf(e3x));
          oby = 0.5f * (fabsf(e1y) + fabsf(e2y) + fabsf(e3y));
          if (rec_pose) far_corner(0.5f * e1x, 0.5f * e1y, 0.5f * e1z, 0.5f * e2x, 0.5f * e2y, 0.5f * e2z, 0.5f * e3x, 0.5f * e3y, 0.5f * e3z);
        } else {
          const float c1x = e2y * e3z - e2z * e3y, c1y = e2z * e3x - e2x * e3z, c1z = e2x * e3y - e2y * e3x;
          const float c2x = e3y * e1z - e3z * e1y, c2y = e3z * e1x - e3x * e1z, c2z = e3x * e1y - e3y * e1x;
          const float c3x = e1y * e2z - e1z * e2y, c3y = e1z * e2x - e1x * e2z, c3z = e1x * e2y - e1y * e2x;
          const float det = e1x * c1x + e1y * c1y + e1z * c1z;
          // (a degenerate vertex list makes the region unbounded: the 1 m search ball then is the only bound)
          const float inv = fabsf(det) > 1e-12f ? 0.5f / det : 3.0e+30f;
          const float s1 = (e1x * e1x + e1y * e1y + e1z * e1z) * inv, s2 = (e2x * e2x + e2y * e2y + e2z * e2z) * inv,
                      s3 = (e3x * e3x + e3y * e3y + e3z * e3z) * inv;
          obx = fabsf(c1x * s1) + fabsf(c2x * s2) + fabsf(c3x * s3);
          oby = fabsf(c1y * s1) + fabsf(c2y * s2) + fabsf(c3y * s3);
          if (rec_pose) far_corner(c1x * s1, c1y * s1, c1z * s1, c2x * s2, c2y * s2, c2z * s2, c3x * s3, c3y * s3, c3z * s3);
        }
        obx = obx * 1.00001f + 1e-4f;
        oby = oby * 1.00001f + 1e-4f;
      }
      float* r = rec + (size_t)q * rec_words;
      r[0] = ccx / 8.f; r[1] = ccy / 8.f; r[2] = ccz / 8.f;
      // The box axes and half extents (three square roots, nine divisions) are only needed by pairs that find
      // candidate points at all -- a minority: the four vertices they derive from are parked in the record and
      // phase D2 finishes the record of a pair once it knows the pair has work (obb_axes_in_place).
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        r[3 + 3 * a + 0] = v[a][0];
        r[3 + 3 * a + 1] = v[a][1];
        r[3 + 3 * a + 2] = v[a][2];
      }
      // Candidate cells: the bounding box of what the critics test -- the region above for the collision critic, the
      // vertices' own bounding box for the min-max critic -- clipped to the 1 m search ball's box, plus 0.1 mm + 1e-5
      // of the extent for the float rounding of the reference's normalised axes (more cells never change a result; a
      // missing one loses a collision).
      // (everything relative to the tile's corner BEFORE the extents are added: `centre - extent` in map coordinates
      // rounds to the float grid of the centre -- 0.24 mm at 4 km, more than the 0.1 mm margin above -- while
      // `(centre - corner) - extent` is carried to ~1e-6 m, like the cell a point was binned into)
      const float gx = k.gmin[0], gy = k.gmin[1];
      const float ocx = ccx / 8.f - gx, ocy = ccy / 8.f - gy, prx = px - gx, pry = py - gy;
      const float lox = fmaxf(fminf(mnx - gx, ocx - obx), prx - 1.0f), hix = fminf(fmaxf(mxx - gx, ocx + obx), prx + 1.0f);
      const float loy = fmaxf(fminf(mny - gy, ocy - oby), pry - 1.0f), hiy = fminf(fmaxf(mxy - gy, ocy + oby), pry + 1.0f);
      int cx0 = (int)floorf(lox * k.inv_cell), cx1 = (int)floorf(hix * k.inv_cell);
      int cy0 = (int)floorf(loy * k.inv_cell), cy1 = (int)floorf(hiy * k.inv_cell);
      cx0 = max(cx0, 0); cy0 = max(cy0, 0);
      cx1 = min(cx1, k.gnx - 1); cy1 = min(cy1, k.gny - 1);
      if (cx0 > cx1 || cy0 > cy1) { cx0 = 1; cx1 = 0; cy0 = 1; cy1 = 0; }
      reinterpret_cast<int*>(r)[15] = (cx0 & 0xFFFF) | (cx1 << 16);
      reinterpret_cast<int*>(r)[16] = (cy0 & 0xFFFF) | (cy1 << 16);
      // A point inside the (convex) box is no farther from the pose than the farthest
      // vertex, so with all vertices well inside the 1 m ball the radius test is moot.
      reinterpret_cast<int*>(r)[17] = j | ((vmax2 < 0.99f || !rec_pose) ? 0x10000 : 0);
      if (rec_pose) { r[18] = px; r[19] = py; r[20] = pz; }
      if (need_mm) {
        float* rm = r + mm_ofs;
        rm[0] = mnx; rm[1] = mny; rm[2] = mnz; rm[3] = mxx; rm[4] = mxy; rm[5] = mxz;
      }
      if (cy1 - cy0 + 1 > k.rows_cap) atomicOr(overflow, 2u);   // host sizes the cells so this cannot happen
    }
  }
  // ---- pure pursuit on the last pose (pure_pursuit_model.cpp:86-113): one lane per
  // trajectory, after the pair loop so that only one wave pays for the atan2 ----
  if (tid < nt && head[tid].steps > 0) {
    const int j = tid;
    const size_t so = (size_t)head[j].li * k.max_steps + (head[j].steps - 1);
    const double2 cs = st_sc[so];
    const float2 bxy = st_xy[so];
    const double c = cs.x, sn = cs.y;
    double L[9], T[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double r0 = k.R[3 * i + 0], r1 = k.R[3 * i + 1], r2 = k.R[3 * i + 2];
      L[3 * i + 0] = r0 * c + r1 * sn;
      L[3 * i + 1] = r1 * c - r0 * sn;
      L[3 * i + 2] = r2;
    }
    pose_translation(k, bxy, T);
    // D = inverse(T_traj) * T_plan ; rotation part L^T * planR, translation L^T (planT - T)
    const double d0 = k.planT[0] - T[0], d1 = k.planT[1] - T[1], d2 = k.planT[2] - T[2];
    const double tx = L[0] * d0 + L[3] * d1 + L[6] * d2;
    const double ty = L[1] * d0 + L[4] * d1 + L[7] * d2;
    const double tz = L[2] * d0 + L[5] * d1 + L[8] * d2;
    const double D00 = L[0] * k.planR[0] + L[3] * k.planR[3] + L[6] * k.planR[6];
    const double D10 = L[1] * k.planR[0] + L[4] * k.planR[3] + L[7] * k.planR[6];
    const double D20 = L[2] * k.planR[0] + L[5] * k.planR[3] + L[8] * k.planR[6];
    double yaw = 0.0;
    if (fabs(D20) < 1.0) yaw = atan2(D10, D00);   // getEulerYPR, solution 1
    // weights are applied in phase E (they are per-critic)
    head[j].pp_dist = sqrt(tx * tx + ty * ty + tz * tz);
    head[j].pp_yaw = fmod(yaw + 3.1416, 3.1416);
  }
  __syncthreads();

  DDDMR_STAMP(4);   // end of phase D1
  if (do_coll && total_pairs > 0) {
    // ---- phase D2: row segments per pair -----------------------------------
    // z is the fastest cell axis, then x: the cells [cx0..cx1] x all z of one y-row
    // are ONE contiguous run [start[(cy*gnx+cx0)*gnz], start[(cy*gnx+cx1+1)*gnz]).
    // Slot q*kRows + r holds row r of pair q.  Work is cut into ITEMS of up to
    // kItem consecutive points of one segment so that every lane of the walk
    // executes the same unrolled body.
    // One lane per pair looks its (<= kRows) cell rows up; ONE packed 64-bit
    // exclusive scan over the pairs gives both the index of a pair's first NON-EMPTY
    // segment (high word) and the number of items in front of it (low word); each
    // lane then writes its own segments.  Empty rows (open space: most of them) are
    // dropped so the walk never has to step over them.
    unsigned long long carry = 0;
    if (tid <= kMaxTile) t_item0[tid] = 0u;           // per-trajectory item counts (a barrier follows inside the loop)
    for (int base = 0; base < total_pairs; base += kScoreThreads) {
      const int q = base + tid;
      uint32_t rb[kRows], rl[kRows];
#pragma unroll
      for (int r = 0; r < kRows; ++r) { rb[r] = 0; rl[r] = 0; }
      unsigned long long cnt = 0;
      if (q < total_pairs) {
        const int* ri = reinterpret_cast<const int*>(rec + (size_t)q * rec_words);
        const int cx0 = (short)(ri[15] & 0xFFFF), cx1 = ri[15] >> 16;
        const int cy0 = (short)(ri[16] & 0xFFFF), cy1 = ri[16] >> 16;
        if (cx0 <= cx1) {
#pragma unroll
          for (int r = 0; r < kRows; ++r) {
            const int cy = cy0 + r;
            if (cy <= cy1) {
              uint32_t b, e;
              if (tab_staged) {
                b = tab[cy * (k.gnx + 1) + cx0];
                e = tab[cy * (k.gnx + 1) + cx1 + 1];
              } else {
                b = cell_start[(cy * k.gnx + cx0) * k.gnz];
                e = cell_start[(cy * k.gnx + cx1 + 1) * k.gnz];
              }
              rb[r] = b;
              rl[r] = e - b;
              if (e > b) cnt += (1ull << 32) | (unsigned long long)((e - b + kItem - 1) / kItem);
            }
          }
        }
      }
      // (segments in the high word, items in the low word: two 32-bit DPP scans)
      const unsigned long long incl = ((unsigned long long)wave_incl_scan_u32((uint32_t)(cnt >> 32)) << 32) |
                                      (unsigned long long)wave_incl_scan_u32((uint32_t)cnt);
      if (lane == 63) wsum64[wid] = incl;
      __syncthreads();
      // offsets of the waves: one LDS read per lane, a DPP scan over the <= 8 wave totals, two readlanes
      unsigned long long wofs, tot;
      {
        constexpr int kW = kScoreThreads / 64;
        const unsigned long long v = lane < kW ? wsum64[lane] : 0ull;
        const uint32_t ihi = wave_incl_scan_u32((uint32_t)(v >> 32)), ilo = wave_incl_scan_u32((uint32_t)v);
        const int w_u = __builtin_amdgcn_readfirstlane(wid);
        tot = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ihi, kW - 1) << 32) |
              (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ilo, kW - 1);
        const int prev = w_u > 0 ? w_u - 1 : 0;
        const unsigned long long pw = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ihi, prev) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ilo, prev);
        wofs = w_u > 0 ? pw : 0ull;
      }
      const unsigned long long ex = carry + wofs + incl - cnt;
      if (cnt) {
        obb_axes_in_place(rec + (size_t)q * rec_words);   // this pair has candidate points: finish its OBB record
        // items of this pair towards its trajectory's total (-> first item of every trajectory below)
        atomicAdd(&t_item0[reinterpret_cast<const int*>(rec + (size_t)q * rec_words)[17] & 0xFFFF], (uint32_t)cnt);
        uint32_t ci = (uint32_t)(ex >> 32), itn = (uint32_t)ex;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          if (rl[r]) {
            pref[ci] = itn;
            seg_p[ci] = rb[r];
            seg_len[ci] = (rl[r] << 12) | (uint32_t)q;     // pair index < 4096, run length < 2^20
            ++ci;
            itn += (rl[r] + kItem - 1) / kItem;
          }
        }
      }
      carry += tot;
      __syncthreads();
    }
    const int nseg = (int)(carry >> 32);
    const uint32_t total = (uint32_t)carry;
    if (tid == 0) pref[nseg] = total;
    if (tid < 64) {                                   // per-trajectory item counts -> first item (exclusive scan)
      const uint32_t c = tid < nt ? t_item0[tid] : 0u;
      const uint32_t incl = wave_incl_scan_u32(c);
      if (tid < nt) t_item0[tid] = incl - c;
      if (tid == nt - 1) t_item0[nt] = incl;          // == total
    }
    __syncthreads();

    DDDMR_STAMP(5);   // end of phase D2
    // ---- phase D3: load-balanced item walk -----------------------------------
#ifdef DDDMR_PHASE_STAMPS
    if (tid == 0 && blockIdx.x < 16384) g_stamps[blockIdx.x * kStampSlots + 9] = total;
#endif
    auto walk_item = [&](const uint32_t it, const int sg) {
      const uint32_t sl_len = seg_len[sg];
      const int q = (int)(sl_len & 0xFFFu);
      const float* rq = rec + (size_t)q * rec_words;
      const int jw = reinterpret_cast<const int*>(rq)[17];
      const int j = jw & 0xFFFF;
      const bool use_radius = (jw & 0x10000) == 0;
      const bool hb = !need_box || head[j].hit_box != 0;
      const bool hm = !need_mm || head[j].hit_mm != 0;
      if (hb && hm) return;                       // trajectory already decided
      atomicAdd(&head[j].walked, 1);              // load feedback: items walked for this trajectory
      const uint32_t off = (it - pref[sg]) * kItem;
      const uint32_t p0 = seg_p[sg] + off;
      const uint32_t n = min((uint32_t)kItem, (sl_len >> 12) - off);
      // kItem independent loads in flight (the sorted array is padded by kItem)
      Pt3 pt[kItem];
      const Pt3* __restrict__ src = sorted + p0;     // one 64-bit address, the 16 loads use immediate offsets
#pragma unroll
      for (int u = 0; u < kItem; ++u) pt[u] = src[u];
      float r[18];
  #pragma unroll
      for (int u = 0; u < 15; ++u) r[u] = rq[u];
      r[15] = r[16] = r[17] = 0.f;             // pose, only read when the radius test is live
      if (use_radius || need_mm) { r[15] = rq[18]; r[16] = rq[19]; r[17] = rq[20]; }
      bool fb = false, fm = false;
      if (need_box) {
        int hits = 0;
  #pragma unroll
        for (int u = 0; u < kItem; u += 2) {
          const int h2 = box_test2(r, pt[u], pt[u + 1], use_radius);
          hits |= h2 << u;                             // one bit per point ...
        }
        fb = (hits & ((1 << n) - 1)) != 0;             // ... and the padding of a partial item masked once
      }
      if (need_mm) {
  #pragma unroll
        for (int u = 0; u < kItem; ++u) {
          // radiusSearch(pose, 1.0): FLANN keeps dist^2 < r^2
          const bool in = (uint32_t)u < n && l2_simple(r[15], r[16], r[17], pt[u].x, pt[u].y, pt[u].z) < 1.0f;
          const float* rm = rq + mm_ofs;
          fm |= in && (pt[u].x >= rm[0] && pt[u].x <= rm[3] && pt[u].y >= rm[1] && pt[u].y <= rm[4] &&
                       pt[u].z >= rm[2] && pt[u].z <= rm[5]);
        }
      }
      if (fb) atomicOr(&head[j].hit_box, 1);
      if (fm) atomicOr(&head[j].hit_mm, 1);
    };
    // Two rounds.  Round 0 probes: every lane walks ONE item, the lanes evenly spread
    // over the list, which settles most colliding trajectories at once (~60 probes
    // each).  Round 1 deals the items of the still undecided trajectories evenly to
    // the lanes, so that nobody idles behind lanes whose share was skipped.
    auto find_seg = [&](uint32_t item, int lo) {      // largest segment >= lo with pref[seg] <= item
      int hi = nseg;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pref[mid] <= item) lo = mid; else hi = mid;
      }
      return lo;
    };
    const int n_rounds = (kProbe && total > (uint32_t)kScoreThreads) ? 2 : 1;
    for (int round = 0; round < n_rounds; ++round) {
      if (tid < 64) {                               // exclusive prefix of the undecided trajectories' item counts
        uint32_t cnt = 0;
        if (tid < nt) {
          const bool decided = round > 0 && (!need_box || head[tid].hit_box != 0) && (!need_mm || head[tid].hit_mm != 0);
          if (!decided) cnt = t_item0[tid + 1] - t_item0[tid];
        }
        const uint32_t incl = wave_incl_scan_u32(cnt);
        if (tid < nt) t_ubase[tid] = incl - cnt;
        if (tid == nt - 1) t_ubase[nt] = incl;
      }
      __syncthreads();
      const uint32_t tot_r = t_ubase[nt];
      const uint32_t chunk = (tot_r + kScoreThreads - 1) / kScoreThreads;
      uint32_t p = (uint32_t)tid * chunk;
      const uint32_t p1 = min(tot_r, p + ((round == 0 && n_rounds == 2) ? 1u : chunk));
      if (p < p1) {
        int j = 0;
        while (t_ubase[j + 1] <= p) ++j;
        uint32_t it = t_item0[j] + (p - t_ubase[j]);
        int sg = find_seg(it, 0);
        for (;;) {
          walk_item(it, sg);
          if (++p >= p1) break;
          if (p >= t_ubase[j + 1]) {                  // next undecided trajectory
            do { ++j; } while (t_ubase[j + 1] <= p);
            it = t_item0[j] + (p - t_ubase[j]);
            sg = find_seg(it, sg);
          } else {
            ++it;
            while (pref[sg + 1] <= it) ++sg;
          }
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();

  DDDMR_STAMP(6);   // end of phase D3
  // ---- phase P: path critics, only for trajectories that did not collide ----
  // A collision verdict makes the trajectory's cost -1 whatever the path critics
  // would return (first negative return wins and StickPath / TowardGlobalPlan never
  // return a negative value when the plan has >= 3 poses), exactly like the
  // reference never reaches them after CollisionModel returned -1
  // (stacked_scoring_model.cpp:83-86).  With most samples colliding in cluttered
  // scenes this skips most of the 1-NN searches.
  if (tid < 64) {
    uint32_t st = 0;
    if (tid < nt) {
      const bool dead = cloud_ok && ((need_box && head[tid].hit_box) || (need_mm && head[tid].hit_mm));
      st = dead ? 0u : (uint32_t)head[tid].steps;
    }
    const uint32_t incl = wave_incl_scan_u32(st);
    if (tid < nt) head[tid].pad = (int)(incl - st);         // first surviving pair of trajectory tid
    if (tid == max(nt, 1) - 1) alive_pairs_s = nt > 0 ? (int)incl : 0;
  }
  __syncthreads();
  {
    // G lanes share one pair, each scanning a slice of the plan; the minimum is exact
    // whatever the split, so the result does not depend on G.
    const int alive = alive_pairs_s;
    int lg = 0;
    while (lg < DDDMR_PSPLIT_MAX && (alive << (lg + 1)) <= kScoreThreads) ++lg;
    const int G = 1 << lg;
    const int per = (k.m + G - 1) >> lg;
    for (int base = 0; base < (alive << lg); base += kScoreThreads) {
      const int idx = base + tid;
      const int q2 = idx >> lg, g = idx & (G - 1);
      const bool valid = q2 < alive;
      int j = 0, s = 0;
      float px = 0.f, py = 0.f, pz = 0.f;
      if (valid) {
        while (j + 1 < nt && head[j + 1].pad <= q2) ++j;    // (dead trajectories have empty ranges)
        s = q2 - head[j].pad;
        const float4 pp = ppos[head[j].pair_base + s];    // the pose D1 composed for this pair
        px = pp.x; py = pp.y; pz = pp.z;
      }
      // exact 1-NN distance to the prune plan (FLANN float distance)
      float best = 3.402823466e+38f;
      if (valid && lg == 0) {
        // One lane per pose, every lane walks the WHOLE plan: the plan point is the same for all lanes, so it comes
        // through the scalar unit (uniform index into the kernel argument -> s_load, operands in SGPRs) instead of
        // 64 lanes reading one LDS address -- broadcast 16-byte LDS reads were what bound this phase.
        int i = 0;
        for (; i + 4 <= k.m; i += 4) {
          const float4 p0 = plan_xyz[i], p1 = plan_xyz[i + 1], p2 = plan_xyz[i + 2], p3 = plan_xyz[i + 3];
          const f2 da = l2_simple2(f2{p0.x, p1.x}, f2{p0.y, p1.y}, f2{p0.z, p1.z}, px, py, pz);
          const f2 db = l2_simple2(f2{p2.x, p3.x}, f2{p2.y, p3.y}, f2{p2.z, p3.z}, px, py, pz);
          best = fminf(fminf(best, fminf(da.x, da.y)), fminf(db.x, db.y));
        }
        for (; i < k.m; ++i) {
          const float4 pp = plan_xyz[i];
          best = fminf(best, l2_simple(pp.x, pp.y, pp.z, px, py, pz));
        }
      } else if (valid) {
        int i = g * per;
        const int i1 = min(k.m, i + per);
        for (; i + 4 <= i1; i += 4) {
          const float4 p0 = plan[i], p1 = plan[i + 1], p2 = plan[i + 2], p3 = plan[i + 3];
          const f2 da = l2_simple2(f2{p0.x, p1.x}, f2{p0.y, p1.y}, f2{p0.z, p1.z}, px, py, pz);
          const f2 db = l2_simple2(f2{p2.x, p3.x}, f2{p2.y, p3.y}, f2{p2.z, p3.z}, px, py, pz);
          best = fminf(fminf(best, fminf(da.x, da.y)), fminf(db.x, db.y));
        }
        for (; i < i1; ++i) {
          const float4 pp = plan[i];
          best = fminf(best, l2_simple(pp.x, pp.y, pp.z, px, py, pz));
        }
      }
      // minimum over the G (<= 16, aligned) lanes of a pose, gathered in its first lane
      if (G > 1) best = fminf(best, dpp_row_shl<1>(best));
      if (G > 2) best = fminf(best, dpp_row_shl<2>(best));
      if (G > 4) best = fminf(best, dpp_row_shl<4>(best));
      if (G > 8) best = fminf(best, dpp_row_shl<8>(best));
      if (valid && g == 0) dist[(size_t)j * S1 + s] = sqrtf(best);
    }
  }
  __syncthreads();
  DDDMR_STAMP(8);   // end of phase P

  // StickPath's sum of per-step distances (stick_path_model.cpp:62-73): one wave per
  // trajectory, lanes stride the steps, fixed reduction tree in double (the
  // reference adds in step order; the difference is <= 1e-15 relative).
  for (int j = wid; j < nt; j += kScoreThreads / 64) {
    const int ns = head[j].steps;
    const float* dr = dist + (size_t)j * S1;
    double acc = 0.0;
    for (int s2 = lane; s2 < ns; s2 += 64) acc += (double)dr[s2];
    // fixed tree: inside the 16-lane rows towards their first lane (DPP), then the four row sums
    acc += dpp_row_shl<8>(acc);
    acc += dpp_row_shl<4>(acc);
    acc += dpp_row_shl<2>(acc);
    acc += dpp_row_shl<1>(acc);
    const double r1 = __shfl(acc, 16, 64), r2 = __shfl(acc, 32, 64), r3 = __shfl(acc, 48, 64);
    if (lane == 0) head[j].stick_sum = (acc + r1) + (r2 + r3);
  }
  __syncthreads();
  DDDMR_STAMP(10);  // end of the StickPath sums

  // ---- phase E: stacked scoring (stacked_scoring_model.cpp:75-93) + argmin ----
  int64_t key = kKeyNone, cbits = kKeyNone;
  int my_gi = -1;
  if (tid < nt) {
    const TrajHead h = head[tid];
    const int li = head[tid].li;
    const int gi = k.begin + li;
    my_gi = gi;
    double cost = DDDMR_COST_NOT_GENERATED;
    if (h.steps > 0) {
      cost = 0.0;
      const float* dr = dist + (size_t)tid * S1;
      // A collided trajectory skipped phase P: its dist[] row and stick_sum hold nothing.  The
      // stack may list a path critic BEFORE the collision critic (any plugin order is legal,
      // mpc_critics_ros.cpp:60-81); its non-negative return is then added and thrown away by
      // the collision critic's -1 further down the stack, so 0 stands in for it here.
      const bool dead = cloud_ok && ((need_box && h.hit_box) || (need_mm && h.hit_mm));
      // (unrolled over the <= 8 stack slots: k.ckind[m] / k.cw[m] with a compile-time m are plain kernel-argument
      // reads the scalar unit fetches up front, instead of one dependent scalar load chain per loop trip)
      bool done = false;
#pragma unroll
      for (int m = 0; m < DDDMR_MAX_CRITICS; ++m) {
        if (m >= k.n_critics || done) continue;
        double r = 0.0;
        switch (k.ckind[m]) {
          case DDDMR_CRITIC_COLLISION:
            r = (cloud_ok && h.hit_box) ? -1.0 : 0.0;
            break;
          case DDDMR_CRITIC_COLLISION_MIN_MAX:
            r = (cloud_ok && h.hit_mm) ? -1.0 : 0.0;
            break;
          case DDDMR_CRITIC_STICK_PATH:
            if (k.m < 3) {
              r = 10.0;
            } else if (dead) {
              r = 0.0;
            } else {
              const double acc = h.stick_sum;
              r = acc / (double)k.m;
            }
            break;
          case DDDMR_CRITIC_PURE_PURSUIT:
            if (k.m == 0 || h.steps < 2) r = -4.0;
            else r = k.ctw[m] * h.pp_dist + k.cow[m] * h.pp_yaw;
            break;
          case DDDMR_CRITIC_TOWARD_GLOBAL_PLAN:
            if (k.m < 3) r = 10.0;
            else if (dead) r = 0.0;
            else r = (double)dr[h.steps - 1] * k.cw[m];
            break;
          case DDDMR_CRITIC_SHORTEST_ANGLE: {
            const double thv = (double)h.w;
            if (k.heading_dev >= 0) r = (thv >= 0) ? k.cw[m] : k.cw[m] * 2;
            else r = (thv >= 0) ? k.cw[m] * 2 : k.cw[m];
          } break;
          case DDDMR_CRITIC_TWIRLING:
            r = fabs((double)h.w) * k.cw[m];
            break;
          default:
            r = 0.0;
        }
        if (r < 0) { cost = r; done = true; }
        else cost += r;
      }
      if (k.final_kernel) key = pack_key(cost, (uint32_t)gi);
      cbits = cost_bits(cost);
    }
    // per-trajectory outputs: plain stores (nothing on the device reads them before the launch ends: k_finalize
    // comes after the launch boundary, and a single-round shard's last workgroup works from the workgroup slots)
    costs[li] = cost;
    samples_out[li] = make_float4(h.vx, h.vy, h.w, 0.f);
    steps_out[li] = h.steps;
    // what this trajectory cost (in collision work items): the walk, the path critics'
    // 1-NN searches when it got that far, and the per-pair phases
    const bool collided = (need_box && h.hit_box) || (need_mm && h.hit_mm);
    traj_load[li] = (uint32_t)h.walked + (uint32_t)h.steps * (2u + (collided ? 0u : (uint32_t)((k.m + 15) >> 4)));
  }
  DDDMR_STAMP(11);  // end of the stacked scoring + per-trajectory stores
  // ---- argmin hand-off ----
  // Multi-round shards (k.final_kernel): wave-0 min-reduction of two words -- the packed key (top 40 bits of the
  // cost | inverted index: one atomicMin yields minimum cost AND, among costs equal in those bits, the highest
  // index) and the cost's full bit pattern, which makes k_finalize's decode exact -- and one pair of atomicMins.
  if (k.final_kernel) {
    if (tid < 64) {
      static_assert(kMaxTile <= 16, "only lanes 0..kMaxTile-1 hold keys");
      {
        long long kk = (long long)key, cc = (long long)cbits, o;
        o = dpp_row_shl<8>(kk); kk = o < kk ? o : kk;
        o = dpp_row_shl<4>(kk); kk = o < kk ? o : kk;
        o = dpp_row_shl<2>(kk); kk = o < kk ? o : kk;
        o = dpp_row_shl<1>(kk); kk = o < kk ? o : kk;
        o = dpp_row_shl<8>(cc); cc = o < cc ? o : cc;
        o = dpp_row_shl<4>(cc); cc = o < cc ? o : cc;
        o = dpp_row_shl<2>(cc); cc = o < cc ? o : cc;
        o = dpp_row_shl<1>(cc); cc = o < cc ? o : cc;
        key = (int64_t)kk;                                   // lane 0: minimum of lanes 0..15
        cbits = (int64_t)cc;
      }
      // Same-address device-scope atomics serialise (~4 ns each): a workgroup first LOOKS at the running minima
      // (plain device-scope loads, served in parallel) and only issues an atomicMin for a word it can still lower.
      // The minima only ever decrease, so a stale (larger) value can cause a superfluous atomic, never a missing
      // one (C3 k_score 126.5 -> 122 us).
      if (tid == 0 && key != kKeyNone) {
        const long long cur_k = (long long)__hip_atomic_load(best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long cur_c = (long long)__hip_atomic_load(best_key + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((long long)key < cur_k) atomicMin((long long*)best_key, (long long)key);
        if ((long long)cbits < cur_c) atomicMin((long long*)best_key + 1, (long long)cbits);
      }
      // How many trajectories the collision critics rejected -- a SAMPLE (every 16th workgroup; tiles are dealt as
      // representative mixes): the next tick's walk starts with a probe round only when most do (the probe settles
      // colliding trajectories early and is pure overhead otherwise).  Word 2 counts the sampled collided
      // trajectories in its low half and the sampled trajectories in its high half.
      if ((blockIdx.x & 15u) == 0u) {
        const bool gen = tid < nt && head[tid].steps > 0;
        const bool coll = gen && cloud_ok && ((need_box && head[tid].hit_box) || (need_mm && head[tid].hit_mm));
        const unsigned long long cm = __ballot(coll), gm = __ballot(gen);
        if (tid == 0 && gm)
          atomicAdd((unsigned long long*)best_key + 2, ((unsigned long long)__popcll(gm) << 32) | (unsigned long long)__popcll(cm));
      }
    }
    // nothing more to do here -- k_finalize decodes after the launch, so a workgroup neither waits for its stores
    // nor pays a device-scope ticket round trip while it holds a slot that the next workgroup is waiting for.
    DDDMR_STAMP(7);
    return;
  }
  // Single-round shards: all workgroups of the launch finish within a few microseconds of each other, and what they
  // used to do then -- two atomicMins and a ticket on the SAME addresses -- queued up behind one another (C2: 3.6 us
  // of a workgroup's 23 us on average, 8.6 us for the unluckiest, which is what the launch waits for).  Now every
  // workgroup leaves its exact local winner in its OWN slot (four 8-byte write-through stores: cost bits, global
  // index, the sample, its collision counts) and draws a ticket; the workgroup that draws the last one reduces the
  // slots (minimum cost as full doubles, equal costs -> highest index: local_planner.cpp:456-463) and publishes the
  // result.  Placement-independent hand-off: slots are written with device-scope stores and read with device-scope
  // loads.
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(best_key) + kSlotBase;
  if (tid < 64) {
    long long cc = (long long)cbits;
    int gi_w = my_gi;
    {
      long long oc; int oi; bool take;
      oc = dpp_row_shl<8>(cc); oi = dpp_row_shl<8>(gi_w); take = oc < cc || (oc == cc && oi > gi_w); cc = take ? oc : cc; gi_w = take ? oi : gi_w;
      oc = dpp_row_shl<4>(cc); oi = dpp_row_shl<4>(gi_w); take = oc < cc || (oc == cc && oi > gi_w); cc = take ? oc : cc; gi_w = take ? oi : gi_w;
      oc = dpp_row_shl<2>(cc); oi = dpp_row_shl<2>(gi_w); take = oc < cc || (oc == cc && oi > gi_w); cc = take ? oc : cc; gi_w = take ? oi : gi_w;
      oc = dpp_row_shl<1>(cc); oi = dpp_row_shl<1>(gi_w); take = oc < cc || (oc == cc && oi > gi_w); cc = take ? oc : cc; gi_w = take ? oi : gi_w;
    }
    const int win_gi = __builtin_amdgcn_readfirstlane(gi_w);         // lane 0: the tile's winner (cost bits in its cc)
    const bool gen = tid < nt && head[tid].steps > 0;
    const bool coll = gen && cloud_ok && ((need_box && head[tid].hit_box) || (need_mm && head[tid].hit_mm));
    const unsigned long long cm = __ballot(coll), gm = __ballot(gen);
    const unsigned long long wm = __ballot(tid < nt && my_gi == win_gi);
    if (tid == 0) {
      float vx = 0.f, vy = 0.f, w = 0.f;
      if (cc != (long long)kKeyNone && wm) {
        const int wl = __ffsll((long long)wm) - 1;
        vx = head[wl].vx; vy = head[wl].vy; w = head[wl].w;
      }
      unsigned long long* sl = slots + (size_t)blockIdx.x * kSlotWords;
      __hip_atomic_store(sl + 0, (unsigned long long)cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sl + 1, (unsigned long long)(long long)win_gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sl + 2, (unsigned long long)__float_as_uint(vx) | ((unsigned long long)__float_as_uint(vy) << 32),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sl + 3, (unsigned long long)__float_as_uint(w) | ((unsigned long long)__popcll(cm) << 32) |
                                     ((unsigned long long)__popcll(gm) << 40),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // wave 0 alone goes on (the other waves are done: nothing they wrote is read on the device again): its slot
    // stores drained, one relaxed device-scope ticket, and the wave that draws the last one reduces
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, DDDMR_HANDOFF_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    DDDMR_STAMP_RAW(15);
    if (t == gridDim.x - 1) {
      if (lane == 0) *ticket = 0;
      reduce_slots(k, lane, (int)gridDim.x, slots, cell_start, overflow, result, words_out);
    }
  }
  DDDMR_STAMP(7);
}

// Winner decode as its own one-wave launch (multi-round shards, DevTick::final_kernel).
__global__ __launch_bounds__(64) void k_finalize(DevTick k, const int64_t* __restrict__ best_key, const double* __restrict__ costs,
                                                 const float4* __restrict__ samples_out, const uint32_t* __restrict__ cell_start,
                                                 const uint32_t* __restrict__ overflow, DevResult* __restrict__ result,
                                                 int64_t* __restrict__ words_out) {
  decode_winner(k, (int)threadIdx.x, best_key, costs, samples_out, cell_start, overflow, result, words_out);
}

// n_local == 0 (a rank without samples): result only.
__global__ void k_empty_result(DevTick k, const uint32_t* __restrict__ cell_start,
                               DevResult* __restrict__ res, int64_t* __restrict__ words_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  DevResult r;
  r.key = kKeyNone;
  r.index = -1;
  r.cost = -1.0;
  r.vx = r.vy = r.wz = 0.f;
  r.n_binned = cell_start[k.n_cells];
  r.overflow = 0;
  r.seq = 0;
  r.n_collided = 0;
  *res = r;
  if (words_out) {
    words_out[0] = kKeyNone;
    words_out[1] = kKeyNone;
    return;
  }
  __threadfence_system();
  __hip_atomic_store(&res->seq, k.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The command of global sample gi: the sample decode of the rollout's phase A.
__device__ __forceinline__ void sample_of_index(const DevTick& k, int gi, const float* __restrict__ axes,
                                                const float4* __restrict__ samples, float* vx, float* vy, float* w) {
  if (k.list_mode) {
    const float4 sm = samples[gi];
    *vx = sm.x; *vy = sm.y; *w = sm.z;
    return;
  }
  const int ith = gi % k.nth;
  const int r = gi / k.nth;
  const int iy = r % k.ny;
  const int ix = r / k.ny;
  if (k.axes_inline) {
    *vx = k.axes_inl[ix]; *vy = k.axes_inl[k.ay_ofs + iy]; *w = k.axes_inl[k.ath_ofs + ith];
  } else {
    *vx = axes[ix]; *vy = axes[k.ay_ofs + iy]; *w = axes[k.ath_ofs + ith];
  }
}

// Multi-rank contexts (dddmr_rollout_comm_init): runs on the tick's stream right after the
// ncclAllReduce(min) of the ranks' (cost bits, -index) slots.  Minimum cost as full doubles, equal
// costs -> highest index = the reference's `<=` scan over the whole batch (local_planner.cpp:456-463);
// the winner's command follows from its index; the result goes to host-mapped memory like k_score's.
__global__ void k_resolve(DevTick k, const int64_t* __restrict__ slots, int n_ranks, const DevResult* __restrict__ local,
                          const float* __restrict__ axes, const float4* __restrict__ samples,
                          DevResult* __restrict__ res) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t c = kKeyNone, ni = kKeyNone;
  for (int r = 0; r < n_ranks; ++r) {
    const int64_t cr = slots[2 * r], ir = slots[2 * r + 1];
    if (cr == kKeyNone) continue;
    if (cr < c || (cr == c && ir < ni)) { c = cr; ni = ir; }
  }
  DevResult r;
  r.key = kKeyNone;
  r.index = -1;
  r.cost = -1.0;
  r.vx = r.vy = r.wz = 0.f;
  if (c != kKeyNone) {
    r.index = (int32_t)(-ni);
    r.cost = __longlong_as_double((long long)c);
    r.key = pack_key(r.cost, (uint32_t)r.index);
    sample_of_index(k, r.index, axes, samples, &r.vx, &r.vy, &r.wz);
  }
  r.n_binned = local->n_binned;
  r.overflow = local->overflow;
  r.seq = 0;
  r.n_collided = local->n_collided;
  *res = r;
  __threadfence_system();
  __hip_atomic_store(&res->seq, k.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// position + Eigen::Quaterniond(T.linear()) of one pose -> x y z qx qy qz qw
__device__ __forceinline__ void write_pose(const double L[9], const double T[3], double* o) {
  double q[4];  // x y z w
  double tr = L[0] + L[4] + L[8];
  if (tr > 0.0) {
    double t = sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (L[7] - L[5]) * t;
    q[1] = (L[2] - L[6]) * t;
    q[2] = (L[3] - L[1]) * t;
  } else {
    int i = 0;
    if (L[4] > L[0]) i = 1;
    if (L[8] > L[4 * i]) i = 2;
    const int j = (i + 1) % 3, kk = (j + 1) % 3;
    double t = sqrt(L[4 * i] - L[4 * j] - L[4 * kk] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (L[3 * kk + j] - L[3 * j + kk]) * t;
    q[j] = (L[3 * j + i] + L[3 * i + j]) * t;
    q[kk] = (L[3 * kk + i] + L[3 * i + kk]) * t;
  }
  o[0] = T[0]; o[1] = T[1]; o[2] = T[2];
  o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
}

// Poses of one trajectory for visualisation (local_planner.cpp:472-478 publishes
// the best trajectory): position + Quaterniond(T.linear()) per step, recomputed
// by one lane with the same recurrence as k_score.
// cub_out (may be null): the 8 cuboid vertices carried along every pose, [step][8][3] floats --
// Trajectory::getCuboid(i) = pcl::transformPointCloud(cuboid, Affine3d) (dd_simple...cpp:443).
__global__ void k_trajectory_poses(DevTick k, int li, const float4* __restrict__ samples_out,
                                   const int32_t* __restrict__ steps, double* __restrict__ poses,
                                   float* __restrict__ cub_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float4 smp = samples_out[li];
  const int ns = steps[li];
  if (ns <= 0) return;
  double sim_time = k.sim_time;
  if (k.kind == DDDMR_THEORY_DD_ROTATE_INPLACE) sim_time = 6.28 / fabs((double)smp.z);
  const double dt = sim_time / (double)ns;
  float px = 0.f, py = 0.f, pth = 0.f;
  for (int s = 0; s < ns; ++s) {
    double sn, cs;
    sincos_heading((double)pth, &sn, &cs);
    double ix = (double)fmul(smp.x, (float)cs), iy = (double)fmul(smp.x, (float)sn);
    if (k.kind == DDDMR_THEORY_OMNI_SIMPLE) {
      double s2, c2;
      sincos_quarter_ahead((double)pth, sn, cs, &s2, &c2);
      ix += (double)smp.y * c2;
      iy += (double)smp.y * s2;
    }
    px = (float)((double)px + ix * dt);
    py = (float)((double)py + iy * dt);
    pth = (float)((double)pth + (double)smp.z * dt);
    sincos_heading((double)pth, &sn, &cs);
    double L[9], T[3];
    for (int i = 0; i < 3; ++i) {
      const double r0 = k.R[3 * i + 0], r1 = k.R[3 * i + 1], r2 = k.R[3 * i + 2];
      L[3 * i + 0] = r0 * cs + r1 * sn;
      L[3 * i + 1] = r1 * cs - r0 * sn;
      L[3 * i + 2] = r2;
      T[i] = r0 * (double)px + r1 * (double)py + k.t[i];
    }
    write_pose(L, T, poses + 7 * (size_t)s);
    if (cub_out) {
      for (int v = 0; v < 8; ++v) {
        const double cx = k.cub[3 * v + 0], cy = k.cub[3 * v + 1], cz = k.cub[3 * v + 2];
        cub_out[(size_t)s * 24 + 3 * v + 0] = (float)(L[0] * cx + L[1] * cy + L[2] * cz + T[0]);
        cub_out[(size_t)s * 24 + 3 * v + 1] = (float)(L[3] * cx + L[4] * cy + L[5] * cz + T[1]);
        cub_out[(size_t)s * 24 + 3 * v + 2] = (float)(L[6] * cx + L[7] * cy + L[8] * cz + T[2]);
      }
    }
  }
}

// Pose arrays of many trajectories at once (the reference's `trajectory` and
// `accepted_trajectory` debug topics, local_planner.cpp:549-569 and :447-470): one lane
// per (trajectory, step) straight from the rollout state of the last tick; off[li] is
// the first output pose of trajectory li, or -1 when it is not wanted.
__global__ __launch_bounds__(256) void k_pose_arrays(DevTick k, const int32_t* __restrict__ off,
                                                     const int32_t* __restrict__ steps,
                                                     const double2* __restrict__ st_sc, const float2* __restrict__ st_xy,
                                                     double* __restrict__ poses) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int li = (int)(idx / (size_t)k.max_steps), s = (int)(idx % (size_t)k.max_steps);
  if (li >= k.n_local || s >= steps[li] || off[li] < 0) return;
  const double2 cs = st_sc[idx];
  const float2 bxy = st_xy[idx];
  double L[9], T[3];
  for (int i = 0; i < 3; ++i) {
    const double r0 = k.R[3 * i + 0], r1 = k.R[3 * i + 1], r2 = k.R[3 * i + 2];
    L[3 * i + 0] = r0 * cs.x + r1 * cs.y;
    L[3 * i + 1] = r1 * cs.x - r0 * cs.y;
    L[3 * i + 2] = r2;
  }
  pose_translation(k, bxy, T);
  write_pose(L, T, poses + 7 * ((size_t)off[li] + (size_t)s));
}

}  // namespace dddmr
