// rollout_kernels.hip.h -- CDNA4 (gfx950) kernels of the local-planner tick.
//
// One tick = 5 launches on one stream:
//   k_bin_count  : crop the aggregate cloud to the local costmap tile and count
//                  points per cell               (replaces the per-tick kd-tree
//                  build, mpc_critics/include/mpc_critics/model_shared_data.h:78-81)
//   k_bin_scan   : exclusive scan of the cell counts (one workgroup)
//   k_bin_scatter: counting-sort scatter -> cell-sorted float4 points
//   k_score      : fused rollout (trajectory_generators theories) + all critics
//                  (mpc_critics/models/*.cpp) + per-workgroup argmin
//   k_finalize   : decode the winner (local_planner.cpp:447-480)
//
// No MFMA anywhere: this is gather / compare work.
//
// Numerics follow the reference's mixed float/double arithmetic; float
// expressions that decide a boolean (box test, radius test, NN distance) are
// evaluated without fused multiply-add so they round exactly like the x86-64
// build of the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dddmr_rollout.h"

namespace dddmr {

constexpr int kScoreThreads = 256;
constexpr int kMaxTile = 16;          // trajectories per workgroup (upper bound)
constexpr int kMaxPlan = 512;         // prune-plan poses kept in LDS
constexpr int64_t kKeyNone = INT64_MAX;
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
  float rmin[3], rmax[3];   // region accepted by the binning pass
  int tile;          // trajectories per workgroup
  int want_collision, want_minmax;
};

struct DevResult {    // written by k_finalize, copied to the host
  int64_t key;
  double cost;
  float vx, vy, wz;
  int32_t index;      // global sample index or -1
  uint32_t n_binned;
  uint32_t overflow;  // 1 if a trajectory needed more than max_steps
};

__host__ __device__ inline int64_t pack_key(double cost, uint32_t gidx) {
  // positive doubles order like their bit patterns; keep the top 40 bits of the
  // cost and put (max - index) below so that min(key) = min cost, ties -> highest
  // index (local_planner.cpp:460-463 keeps the LAST minimal trajectory).
  if (!(cost >= 0.0)) return kKeyNone;
  union { double d; uint64_t u; } c;
  c.d = cost;
  const uint64_t idx_mask = (1ull << kKeyIndexBits) - 1;
  return (int64_t)((c.u & ~idx_mask) | (idx_mask - (uint64_t)gidx));
}
__host__ __device__ inline int32_t key_index(int64_t key) {
  if (key == kKeyNone) return -1;
  const uint64_t idx_mask = (1ull << kKeyIndexBits) - 1;
  return (int32_t)(idx_mask - ((uint64_t)key & idx_mask));
}

// ---------------------------------------------------------------------------
// float helpers that must not be contracted into fma
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }

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

// ---------------------------------------------------------------------------
// binning: cloud -> cell-sorted local costmap tile
// ---------------------------------------------------------------------------
__device__ __forceinline__ int cell_of(const DevTick& k, float x, float y, float z) {
  int cx = (int)floorf((x - k.gmin[0]) * k.inv_cell);
  int cy = (int)floorf((y - k.gmin[1]) * k.inv_cell);
  int cz = (int)floorf((z - k.gmin[2]) * k.inv_cell);
  cx = min(max(cx, 0), k.gnx - 1);
  cy = min(max(cy, 0), k.gny - 1);
  cz = min(max(cz, 0), k.gnz - 1);
  return (cy * k.gnx + cx) * k.gnz + cz;
}

__global__ __launch_bounds__(256) void k_bin_count(DevTick k, const float4* __restrict__ cloud,
                                                   uint32_t* __restrict__ cell_count,
                                                   uint2* __restrict__ pt_slot) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k.n_points; i += stride) {
    const float4 p = cloud[i];
    uint2 slot = make_uint2(0xFFFFFFFFu, 0u);
    const bool in = p.x >= k.rmin[0] && p.x <= k.rmax[0] && p.y >= k.rmin[1] && p.y <= k.rmax[1] &&
                    p.z >= k.rmin[2] && p.z <= k.rmax[2];
    if (in) {
      const int c = cell_of(k, p.x, p.y, p.z);
      slot.x = (uint32_t)c;
      slot.y = atomicAdd(&cell_count[c], 1u);
    }
    pt_slot[i] = slot;
  }
}

// one workgroup of 1024 threads; also zeroes the counters for the next tick and
// resets the argmin key.
__global__ __launch_bounds__(1024) void k_bin_scan(DevTick k, uint32_t* __restrict__ cell_count,
                                                   uint32_t* __restrict__ cell_start,
                                                   int64_t* __restrict__ best_key,
                                                   uint32_t* __restrict__ overflow) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  if (tid == 0) {
    carry_s = 0;
    *best_key = kKeyNone;
    *overflow = 0;
  }
  __syncthreads();
  const int n = k.n_cells;
  for (int base = 0; base < n; base += 1024 * 4) {
    const int i0 = base + tid * 4;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = (i0 + j < n) ? cell_count[i0 + j] : 0u;
      if (i0 + j < n) cell_count[i0 + j] = 0u;
    }
    const uint32_t tsum = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wave_sum[wid] = incl;
    __syncthreads();
    uint32_t wofs = 0;
    for (int w = 0; w < wid; ++w) wofs += wave_sum[w];
    uint32_t total = 0;
    for (int w = 0; w < 16; ++w) total += wave_sum[w];
    const uint32_t carry = carry_s;
    uint32_t run = carry + wofs + incl - tsum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0 + j < n) cell_start[i0 + j] = run;
      run += v[j];
    }
    __syncthreads();
    if (tid == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (tid == 0) cell_start[n] = carry_s;
}

__global__ __launch_bounds__(256) void k_bin_scatter(DevTick k, const float4* __restrict__ cloud,
                                                     const uint2* __restrict__ pt_slot,
                                                     const uint32_t* __restrict__ cell_start,
                                                     float4* __restrict__ sorted) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k.n_points; i += stride) {
    const uint2 slot = pt_slot[i];
    if (slot.x != 0xFFFFFFFFu) sorted[cell_start[slot.x] + slot.y] = cloud[i];
  }
}

// ---------------------------------------------------------------------------
// fused rollout + critics + argmin
// ---------------------------------------------------------------------------
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
};

// dynamic LDS carve (all rows are max_steps+1 long):
//   TrajHead head[tile]; float4 plan[m]; double2 sc[tile][S1]; float th[tile][S1];
//   float2 xy[tile][S1]; float dist[tile][S1];
__host__ __device__ inline size_t score_lds_bytes(int tile, int max_steps, int m) {
  const size_t S1 = (size_t)max_steps + 1;
  size_t b = 0;
  b += sizeof(TrajHead) * (size_t)tile;
  b = (b + 15) & ~(size_t)15;
  b += 16 * (size_t)(m > 0 ? m : 1);
  b += 16 * (size_t)tile * S1;   // sc
  b += 4 * (size_t)tile * S1;    // th
  b += 8 * (size_t)tile * S1;    // xy
  b += 4 * (size_t)tile * S1;    // dist
  return (b + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(kScoreThreads) void k_score(
    DevTick k, const float* __restrict__ axes, const float4* __restrict__ samples,
    const float4* __restrict__ plan_xyz, const uint32_t* __restrict__ cell_start,
    const float4* __restrict__ sorted, double* __restrict__ costs, int32_t* __restrict__ steps_out,
    float4* __restrict__ samples_out, int64_t* __restrict__ best_key, uint32_t* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tile = k.tile;
  const int S1 = k.max_steps + 1;
  size_t ofs = 0;
  TrajHead* head = reinterpret_cast<TrajHead*>(lds_raw);
  ofs += sizeof(TrajHead) * (size_t)tile;
  ofs = (ofs + 15) & ~(size_t)15;
  float4* plan = reinterpret_cast<float4*>(lds_raw + ofs);
  ofs += 16 * (size_t)(k.m > 0 ? k.m : 1);
  double2* sc = reinterpret_cast<double2*>(lds_raw + ofs);
  ofs += 16 * (size_t)tile * S1;
  float* th = reinterpret_cast<float*>(lds_raw + ofs);
  ofs += 4 * (size_t)tile * S1;
  float2* xy = reinterpret_cast<float2*>(lds_raw + ofs);
  ofs += 8 * (size_t)tile * S1;
  float* dist = reinterpret_cast<float*>(lds_raw + ofs);

  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * tile;                 // first local trajectory of this tile
  const int nt = min(tile, k.n_local - t0);         // trajectories in this tile

  // ---- stage the prune plan (float xyz, model_shared_data.h:83-91) ----
  for (int i = tid; i < k.m; i += kScoreThreads) plan[i] = plan_xyz[i];

  // ---- phase A: sample, generation gates, step count, theta recurrence ----
  if (tid < nt) {
    const int li = t0 + tid;
    const int gi = k.begin + li;
    float vx, vy, w;
    if (k.list_mode) {
      const float4 s = samples[gi];
      vx = s.x; vy = s.y; w = s.z;
    } else {
      const int ith = gi % k.nth;
      const int r = gi / k.nth;
      const int iy = r % k.ny;
      const int ix = r / k.ny;
      vx = axes[ix];
      vy = axes[k.ay_ofs + iy];
      w = axes[k.ath_ofs + ith];
    }
    const double eps = 1e-4;
    bool ok = true;
    double vmag, sim_time = k.sim_time;
    if (k.kind == DDDMR_THEORY_DD_SIMPLE) {
      // dd_simple_trajectory_generator_theory.cpp:364-371
      vmag = fabs((double)vx);
      if ((k.min_vel_x >= 0 && vmag + eps < k.min_vel_x) &&
          (k.min_vel_theta >= 0 && fabs((double)w) + eps < k.min_vel_theta)) ok = false;
      if (k.max_vel_x >= 0 && vmag - eps > k.max_vel_x) ok = false;
    } else if (k.kind == DDDMR_THEORY_OMNI_SIMPLE) {
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
      if (ns > k.max_steps) {       // capacity error, reported to the host
        atomicOr(overflow, 1u);
        ns = 0;
      }
    }
    const double dt = ns > 0 ? sim_time / (double)ns : 0.0;
    TrajHead h;
    h.vx = vx; h.vy = vy; h.w = w;
    h.steps = ns;
    h.dt = dt;
    h.pair_base = 0;
    h.hit_box = 0; h.hit_mm = 0; h.pad = 0;
    h.pp_dist = 0.0; h.pp_yaw = 0.0;
    head[tid] = h;
    // theta_{k+1} = float(theta_k + w*dt)   (computeNewPositions, :457-464)
    float* row = th + (size_t)tid * S1;
    float a = 0.f;
    row[0] = 0.f;
    const double wdt = (double)w * dt;
    for (int s = 1; s <= ns; ++s) {
      a = (float)((double)a + wdt);
      row[s] = a;
    }
  }
  __syncthreads();

  // pair offsets (tile <= 16: serial prefix by one lane)
  if (tid == 0) {
    int acc = 0;
    for (int j = 0; j < nt; ++j) {
      head[j].pair_base = acc;
      acc += head[j].steps;
    }
  }

  // ---- phase B: double sin/cos of every theta_k (k = 0..steps) ----
  // theta_k feeds both the next position update (cos/sin of the float state) and
  // the pose's AngleAxisd(theta) rotation (dd_simple...cpp:416).
  for (int idx = tid; idx < nt * S1; idx += kScoreThreads) {
    const int j = idx / S1, s = idx - j * S1;
    if (s <= head[j].steps) {
      double sn, cs;
      sincos((double)th[(size_t)j * S1 + s], &sn, &cs);
      sc[(size_t)j * S1 + s] = make_double2(cs, sn);
    }
  }
  __syncthreads();

  // ---- phase C: x,y recurrence in the body frame ----
  if (tid < nt) {
    const TrajHead h = head[tid];
    const double2* scr = sc + (size_t)tid * S1;
    float2* xr = xy + (size_t)tid * S1;
    float px = 0.f, py = 0.f;
    const bool omni = (k.kind == DDDMR_THEORY_OMNI_SIMPLE);
    for (int s = 0; s < h.steps; ++s) {
      const double2 cs = scr[s];
      const float cf = (float)cs.x, sf = (float)cs.y;  // cos/sin(float) overloads
      double ix, iy;
      if (omni) {
        // cos(M_PI_2 + theta), sin(M_PI_2 + theta) in double (omni...cpp:501-502)
        double s2, c2;
        sincos(M_PI_2 + (double)th[(size_t)tid * S1 + s], &s2, &c2);
        ix = (double)fmul(h.vx, cf) + (double)h.vy * c2;
        iy = (double)fmul(h.vx, sf) + (double)h.vy * s2;
      } else {
        ix = (double)fmul(h.vx, cf);
        iy = (double)fmul(h.vx, sf);
      }
      px = (float)((double)px + ix * h.dt);
      py = (float)((double)py + iy * h.dt);
      xr[s] = make_float2(px, py);
    }
  }
  __syncthreads();

  // ---- phase D: one (trajectory, step) pair per lane ----
  int total_pairs = 0;
  if (nt > 0) total_pairs = head[nt - 1].pair_base + head[nt - 1].steps;
  const bool cloud_ok = k.n_points >= 5;   // collision_model.cpp:53-55
  for (int q = tid; q < total_pairs; q += kScoreThreads) {
    int j = 0;
    while (j + 1 < nt && head[j + 1].pair_base <= q) ++j;
    const int s = q - head[j].pair_base;
    const double2 cs = sc[(size_t)j * S1 + s + 1];   // pose after the step
    const float2 bxy = xy[(size_t)j * S1 + s];
    const double c = cs.x, sn = cs.y;
    // trans_gbl2traj = pos_af3 * [Rz(theta), (x, y, 0)]
    double L[9], T[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double r0 = k.R[3 * i + 0], r1 = k.R[3 * i + 1], r2 = k.R[3 * i + 2];
      L[3 * i + 0] = r0 * c + r1 * sn;
      L[3 * i + 1] = r1 * c - r0 * sn;
      L[3 * i + 2] = r2;
      T[i] = r0 * (double)bxy.x + r1 * (double)bxy.y + k.t[i];
    }
    const float px = (float)T[0], py = (float)T[1], pz = (float)T[2];   // trajectory.cpp:69-75

    // ---- path critics: exact 1-NN distance to the prune plan ----
    {
      float best = 3.402823466e+38f;
      for (int i = 0; i < k.m; ++i) {
        const float4 pp = plan[i];
        best = fminf(best, l2_simple(pp.x, pp.y, pp.z, px, py, pz));
      }
      dist[(size_t)j * S1 + s] = sqrtf(best);
    }

    // ---- pure pursuit on the last pose (pure_pursuit_model.cpp:86-113) ----
    if (s == head[j].steps - 1) {
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
      const double yf = fmod(yaw + 3.1416, 3.1416);
      // weights are applied in phase E (they are per-critic)
      head[j].pp_dist = sqrt(tx * tx + ty * ty + tz * tz);
      head[j].pp_yaw = yf;
    }

    // ---- collision critics ----
    if (cloud_ok && (k.want_collision | k.want_minmax)) {
      // pcl::transformPointCloud(cuboid, Affine3d): double multiply-add, cast to float
      float vxs[8], vys[8], vzs[8];
      float mnx = 3.402823466e+38f, mny = mnx, mnz = mnx, mxx = -mnx, mxy = -mnx, mxz = -mnx;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const double cx = k.cub[3 * v + 0], cy = k.cub[3 * v + 1], cz = k.cub[3 * v + 2];
        vxs[v] = (float)(L[0] * cx + L[1] * cy + L[2] * cz + T[0]);
        vys[v] = (float)(L[3] * cx + L[4] * cy + L[5] * cz + T[1]);
        vzs[v] = (float)(L[6] * cx + L[7] * cy + L[8] * cz + T[2]);
        mnx = fminf(mnx, vxs[v]); mxx = fmaxf(mxx, vxs[v]);
        mny = fminf(mny, vys[v]); mxy = fmaxf(mxy, vys[v]);
        mnz = fminf(mnz, vzs[v]); mxz = fmaxf(mxz, vzs[v]);
      }
      // collision_model.cpp:85-115: centre, axes, half extents (float / double mix)
      float ccx = 0.f, ccy = 0.f, ccz = 0.f;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        ccx = fadd(ccx, vxs[v]); ccy = fadd(ccy, vys[v]); ccz = fadd(ccz, vzs[v]);
      }
      ccx = ccx / 8.f; ccy = ccy / 8.f; ccz = ccz / 8.f;
      float ax[3][3];
      float half[3];
      const int vi[3] = {3, 1, 2};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float ex = fsub(vxs[vi[a]], vxs[0]), ey = fsub(vys[vi[a]], vys[0]), ez = fsub(vzs[vi[a]], vzs[0]);
        const float len = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
        const double h = (double)len / 2.;
        half[a] = (float)h;                    // len/2 is exact in float
        ax[a][0] = (float)((double)ex / (2. * h));
        ax[a][1] = (float)((double)ey / (2. * h));
        ax[a][2] = (float)((double)ez / (2. * h));
      }
      // candidate cells: cuboid AABB clipped to the 1 m search ball's AABB
      const float lox = fmaxf(mnx, px - 1.0f), hix = fminf(mxx, px + 1.0f);
      const float loy = fmaxf(mny, py - 1.0f), hiy = fminf(mxy, py + 1.0f);
      int cx0 = (int)floorf((lox - k.gmin[0]) * k.inv_cell), cx1 = (int)floorf((hix - k.gmin[0]) * k.inv_cell);
      int cy0 = (int)floorf((loy - k.gmin[1]) * k.inv_cell), cy1 = (int)floorf((hiy - k.gmin[1]) * k.inv_cell);
      cx0 = max(cx0, 0); cy0 = max(cy0, 0);
      cx1 = min(cx1, k.gnx - 1); cy1 = min(cy1, k.gny - 1);
      bool hit_box = false, hit_mm = false;
      const bool need_box = k.want_collision != 0, need_mm = k.want_minmax != 0;
      if (cx0 <= cx1) {
        for (int cy = cy0; cy <= cy1; ++cy) {
          // z is the fastest cell axis, then x: one contiguous run per y-row
          const uint32_t b = cell_start[(cy * k.gnx + cx0) * k.gnz];
          const uint32_t e = cell_start[(cy * k.gnx + cx1 + 1) * k.gnz];
          for (uint32_t i = b; i < e; ++i) {
            const float4 p = sorted[i];
            // radiusSearch(pose, 1.0): FLANN keeps dist^2 < r^2
            if (!(l2_simple(px, py, pz, p.x, p.y, p.z) < 1.0f)) continue;
            if (need_box) {
              const float dx = fsub(p.x, ccx), dy = fsub(p.y, ccy), dz = fsub(p.z, ccz);
              const float xv = fabsf(dot3(dx, dy, dz, ax[0][0], ax[0][1], ax[0][2]));
              const float yv = fabsf(dot3(dx, dy, dz, ax[1][0], ax[1][1], ax[1][2]));
              const float zv = fabsf(dot3(dx, dy, dz, ax[2][0], ax[2][1], ax[2][2]));
              hit_box |= (xv <= half[0] && yv <= half[1] && zv <= half[2]);
            }
            if (need_mm) {
              hit_mm |= (p.x >= mnx && p.x <= mxx && p.y >= mny && p.y <= mxy && p.z >= mnz && p.z <= mxz);
            }
          }
          if ((hit_box || !need_box) && (hit_mm || !need_mm)) break;
        }
      }
      if (hit_box) atomicOr(&head[j].hit_box, 1);
      if (hit_mm) atomicOr(&head[j].hit_mm, 1);
    }
  }
  __syncthreads();

  // ---- phase E: stacked scoring (stacked_scoring_model.cpp:75-93) + argmin ----
  int64_t key = kKeyNone;
  if (tid < nt) {
    const TrajHead h = head[tid];
    const int li = t0 + tid;
    const int gi = k.begin + li;
    double cost = DDDMR_COST_NOT_GENERATED;
    if (h.steps > 0) {
      cost = 0.0;
      const float* dr = dist + (size_t)tid * S1;
      for (int m = 0; m < k.n_critics; ++m) {
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
            } else {
              double acc = 0.0;
              for (int s = 0; s < h.steps; ++s) acc += (double)dr[s];
              r = acc / (double)k.m;
            }
            break;
          case DDDMR_CRITIC_PURE_PURSUIT:
            if (k.m == 0 || h.steps < 2) r = -4.0;
            else r = k.ctw[m] * h.pp_dist + k.cow[m] * h.pp_yaw;
            break;
          case DDDMR_CRITIC_TOWARD_GLOBAL_PLAN:
            if (k.m < 3) r = 10.0;
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
        if (r < 0) { cost = r; break; }
        cost += r;
      }
      key = pack_key(cost, (uint32_t)gi);
    }
    costs[li] = cost;
    steps_out[li] = h.steps;
    samples_out[li] = make_float4(h.vx, h.vy, h.w, 0.f);
  }
  // wave-0 shuffle min-reduction of the packed keys, one atomic per workgroup
  if (tid < 64) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int64_t other = __shfl_xor(key, o, 64);
      key = other < key ? other : key;
    }
    if (tid == 0 && key != kKeyNone) atomicMin((long long*)best_key, (long long)key);
  }
}

__global__ void k_finalize(DevTick k, const int64_t* __restrict__ best_key,
                           const double* __restrict__ costs, const float4* __restrict__ samples_out,
                           const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ overflow,
                           DevResult* __restrict__ res) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  DevResult r;
  r.key = *best_key;
  r.index = key_index(r.key);
  r.cost = -1.0;
  r.vx = r.vy = r.wz = 0.f;
  if (r.index >= 0) {
    const int li = r.index - k.begin;
    r.cost = costs[li];
    const float4 s = samples_out[li];
    r.vx = s.x; r.vy = s.y; r.wz = s.z;
  }
  r.n_binned = cell_start[k.n_cells];
  r.overflow = *overflow;
  *res = r;
}

// Poses of one trajectory for visualisation (local_planner.cpp:472-478 publishes
// the best trajectory): position + Quaterniond(T.linear()) per step, recomputed
// by one lane with the same recurrence as k_score.
__global__ void k_trajectory_poses(DevTick k, int li, const float4* __restrict__ samples_out,
                                   const int32_t* __restrict__ steps, double* __restrict__ poses) {
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
    sincos((double)pth, &sn, &cs);
    double ix = (double)fmul(smp.x, (float)cs), iy = (double)fmul(smp.x, (float)sn);
    if (k.kind == DDDMR_THEORY_OMNI_SIMPLE) {
      double s2, c2;
      sincos(M_PI_2 + (double)pth, &s2, &c2);
      ix += (double)smp.y * c2;
      iy += (double)smp.y * s2;
    }
    px = (float)((double)px + ix * dt);
    py = (float)((double)py + iy * dt);
    pth = (float)((double)pth + (double)smp.z * dt);
    sincos((double)pth, &sn, &cs);
    double L[9], T[3];
    for (int i = 0; i < 3; ++i) {
      const double r0 = k.R[3 * i + 0], r1 = k.R[3 * i + 1], r2 = k.R[3 * i + 2];
      L[3 * i + 0] = r0 * cs + r1 * sn;
      L[3 * i + 1] = r1 * cs - r0 * sn;
      L[3 * i + 2] = r2;
      T[i] = r0 * (double)px + r1 * (double)py + k.t[i];
    }
    // Eigen::Quaterniond(matrix)
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
    double* o = poses + 7 * (size_t)s;
    o[0] = T[0]; o[1] = T[1]; o[2] = T[2];
    o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
  }
}

}  // namespace dddmr
