// marking.hip.h -- global-mode marking / clearing layer on the device.
//
// Replaces, for is_local_planner = false, one StackedPerception::doClear_then_Mark pass of the lidar
// plugin (citations relative to /root/reference/src/dddmr_perception_3d/):
//   selfClear              plugins/multilayer_spinning_lidar.cpp:456-628   -> k_mk_clear, k_mk_unmark
//   selfMark               :306-455                                        -> k_mk_cc_*, k_mk_cluster_*, k_mk_*_keys,
//                                                                             k_mk_group_reduce, k_mk_slots, k_mk_commit,
//                                                                             k_mk_dgraph
//   isinLidarObservation   :682-746                                        -> in_lidar_observation()
//   getCastingPointCloud   :630-651                                        -> the ray march inside k_mk_clear
//   Marking::addPCPtr / removePCPtr / computeMinDistanceFromObstacle2GroundNodes
//                          plugins/cluster_marking.cpp:49-138              -> k_mk_commit + k_mk_dgraph / k_mk_unmark
//   DynamicGraph           src/graph/dynamic_graph.cpp:38-61               -> the dgraph array (atomic min on the double's bits)
//
// Data layout in HBM (persistent across updates):
//   store     open-addressing hash table, voxel key (x, y, z ints, the reference's std::map keys) -> slot = table
//             position; per slot: alive flag, plane, range of its GENERATOR points in the pool.  A cleared marking
//             keeps its slot like the reference keeps the map entry with a null pc_; a new cluster at the same voxel
//             reuses it.
//   pool      generator points of every stored marking: the cluster projected onto the robot's ground plane and
//             voxel-downsampled at 0.1 m -- exactly the points computeMinDistanceFromObstacle2GroundNodes searches the
//             ground kd-tree with.  nodes_of_min_distance_ is a pure function of them and the static ground cloud, so
//             it is recomputed when the marking is removed instead of being stored.
//   dgraph    double[n_ground + 1], lethal uint8[n_ground + 1]
//   grids     uniform grids (cell-sorted copies) of the ground cloud, the static map cloud and the last two observations:
//             they answer every kd-tree radius query of the reference exactly (FLANN float distance, strict <).
//
// Every grouping step whose summation order shows in the result (cluster centroid, 0.2 m and 0.1 m VoxelGrid
// centroids: PCL adds floats in index order) is a STABLE radix sort (rocPRIM) followed by one lane per group adding
// in order, so the centroids are bit-identical to a sequential pass over the same point order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/rocprim.hpp>

#include "rollout_kernels.hip.h"

#pragma clang fp contract(off)

namespace dddmr {

struct PointGrid {            // uniform grid over a point set; cell = (cz * ny + cy) * nx + cx
  float ox = 0, oy = 0, oz = 0;
  float inv_xy = 1, inv_z = 1;
  int nx = 1, ny = 1, nz = 1;
  uint32_t n = 0;
  uint32_t* cell_start = nullptr;   // [nx * ny * nz + 1]
  float4* sorted = nullptr;         // x y z, w = original index (bits)
};

__device__ __forceinline__ int grid_cx(const PointGrid& g, float x) { return min(max((int)floorf((x - g.ox) * g.inv_xy), 0), g.nx - 1); }
__device__ __forceinline__ int grid_cy(const PointGrid& g, float y) { return min(max((int)floorf((y - g.oy) * g.inv_xy), 0), g.ny - 1); }
__device__ __forceinline__ int grid_cz(const PointGrid& g, float z) { return min(max((int)floorf((z - g.oz) * g.inv_z), 0), g.nz - 1); }
// Cell of q + d (a search ball's edge), the origin subtracted FIRST: q - origin is a small number carried exactly or to
// ~1e-6 m, whereas q + d rounds to the float grid of q -- 0.24 mm at 4 km, more than the 0.1 mm by which the callers
// widen their balls, so that a neighbour could fall into a cell outside the range (found with the scene shifted by
// kilometres: tests/test_marking_gpu.py::test_marking_far_from_the_map_origin).
__device__ __forceinline__ int grid_cx(const PointGrid& g, float q, float d) { return min(max((int)floorf(((q - g.ox) + d) * g.inv_xy), 0), g.nx - 1); }
__device__ __forceinline__ int grid_cy(const PointGrid& g, float q, float d) { return min(max((int)floorf(((q - g.oy) + d) * g.inv_xy), 0), g.ny - 1); }
__device__ __forceinline__ int grid_cz(const PointGrid& g, float q, float d) { return min(max((int)floorf(((q - g.oz) + d) * g.inv_z), 0), g.nz - 1); }

__global__ __launch_bounds__(256) void k_grid_count(PointGrid g, const float4* __restrict__ pts, uint32_t* __restrict__ counts,
                                                    uint2* __restrict__ slot) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n) return;
  const float4 p = pts[i];
  const uint32_t c = (uint32_t)((grid_cz(g, p.z) * g.ny + grid_cy(g, p.y)) * g.nx + grid_cx(g, p.x));
  slot[i] = make_uint2(c, atomicAdd(&counts[c], 1u));
}
__global__ __launch_bounds__(256) void k_grid_scatter(PointGrid g, const float4* __restrict__ pts, const uint2* __restrict__ slot) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n) return;
  const float4 p = pts[i];
  const uint2 s = slot[i];
  g.sorted[g.cell_start[s.x] + s.y] = make_float4(p.x, p.y, p.z, __int_as_float((int)i));
}

// All points of the cells the ball's bounding box touches (the caller applies FLANN's float distance test).
template <class F>
__device__ __forceinline__ void grid_for_each(const PointGrid& g, float qx, float qy, float qz, float r, F&& f) {
  const int x0 = grid_cx(g, qx, -r), x1 = grid_cx(g, qx, r);
  const int y0 = grid_cy(g, qy, -r), y1 = grid_cy(g, qy, r);
  const int z0 = grid_cz(g, qz, -r), z1 = grid_cz(g, qz, r);
  const int nys = y1 - y0 + 1, nrows = nys * (z1 - z0 + 1);
  if (nrows <= 4 && nys <= 2) {
    // The usual case (a ball no wider than a cell: at most 2 x 2 rows of cells): the bounds of all rows and the first point
    // of each are loaded together, then walked in the same order -- two dependent round trips per query instead of two
    // per row (the ray tests and the "still observed" test are chains of these).
    uint32_t b[4], e[4];
    float4 first[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int q = min(rr, nrows - 1);
      const int cz = z0 + (nys == 2 ? (q >> 1) : q), cy = y0 + (nys == 2 ? (q & 1) : 0);
      const int row = (cz * g.ny + cy) * g.nx;
      b[rr] = g.cell_start[row + x0];
      e[rr] = g.cell_start[row + x1 + 1];
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (rr >= nrows) e[rr] = b[rr];
      first[rr] = b[rr] < e[rr] ? g.sorted[b[rr]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (b[rr] < e[rr]) {
        if (f(first[rr])) return;
        for (uint32_t k = b[rr] + 1u; k < e[rr]; ++k)
          if (f(g.sorted[k])) return;
      }
    }
    return;
  }
  for (int cz = z0; cz <= z1; ++cz)
    for (int cy = y0; cy <= y1; ++cy) {
      const uint32_t b = g.cell_start[(cz * g.ny + cy) * g.nx + x0], e = g.cell_start[(cz * g.ny + cy) * g.nx + x1 + 1];
      for (uint32_t k = b; k < e; ++k)
        if (f(g.sorted[k])) return;
    }
}
// The same walk with the 64 lanes of a wave striding each row's run (one query per wave).
template <class F>
__device__ __forceinline__ void grid_for_each_wave(const PointGrid& g, float qx, float qy, float qz, float r, int lane, F&& f) {
  const int x0 = grid_cx(g, qx, -r), x1 = grid_cx(g, qx, r);
  const int y0 = grid_cy(g, qy, -r), y1 = grid_cy(g, qy, r);
  const int z0 = grid_cz(g, qz, -r), z1 = grid_cz(g, qz, r);
  for (int cz = z0; cz <= z1; ++cz)
    for (int cy = y0; cy <= y1; ++cy) {
      const uint32_t b = g.cell_start[(cz * g.ny + cy) * g.nx + x0], e = g.cell_start[(cz * g.ny + cy) * g.nx + x1 + 1];
      for (uint32_t k = b + (uint32_t)lane; k < e; k += 64) f(g.sorted[k]);
    }
}
// pcl::KdTreeFLANN::radiusSearch count with the squared radius already cast to float
__device__ __forceinline__ int grid_radius_count(const PointGrid& g, float qx, float qy, float qz, float r, float r2, int stop_at) {
  int cnt = 0;
  grid_for_each(g, qx, qy, qz, r, [&](const float4 p) {
    if (l2_simple(p.x, p.y, p.z, qx, qy, qz) < r2) ++cnt;
    return cnt >= stop_at;
  });
  return cnt;
}

struct MarkParams {
  // dddmr_marking_config
  double res, hres, marking_height, window;
  double fov_top, fov_bottom, ps, pe, ns, ne;
  double ignore_ratio, inscribed, inflation;
  float tol, tol2;             // cluster tolerance and static_cast<float>(tol * tol)
  int min_cluster;
  // transforms of this update
  double sn[3], sd;            // sensor plane normal (quatRotate(q_gbl2s, z)) and offset
  double st[3];                // trans_gbl2s_ translation
  double Rs[9];                // rotation matrix of trans_gbl2s_ (tf2 Matrix3x3::setRotation of its quaternion)
  float mc[4];                 // plane through base_link, normal = base z (ModelCoefficients, :401-409)
  int wx0, wx1, wy0, wy1, wz0, wz1;   // selfClear's window in voxel keys, [min, max)
  uint32_t n_obs;              // points of this update's observation
  uint32_t n_prev;             // points of the previous observation (pcl_msg_gbl_), 0 = none
  uint32_t table_mask;         // store size - 1
  uint32_t pool_cap;
  uint32_t n_ground;
  uint32_t seq;                // update sequence number
  uint32_t n_alive_prev;       // entries of MarkStore::alive_list
  float pad;                   // by how much the ray probes' and the still-observed test's search boxes are widened (1e-4 m)
};

struct MarkCounters {         // device counters of one update (copied back for dddmr_marking_stats)
  uint32_t n_clusters, n_marked, n_in_window, n_cleared, n_alive, pool_used, overflow, n_groups2, n_groups3, n_clusters_kept;
  uint32_t n_removed;
  uint32_t n_dup;   // clusters of this update that found their voxel already claimed by another one (marking_fix_ties)
  uint32_t n_new_keys;   // voxels that entered the store for the first time in this update (store garbage collection)
  uint32_t n_rehashed;   // alive markings moved by this update's garbage collection
  uint32_t fallback;     // fused route: a partition or its voxel sort keys did not fit, the mark phase has to take the general route
  uint32_t n_clear;      // fused route: stored markings inside the window and the sensor's view (entries of the ray-test list)
  uint32_t n_revived;    // fused route: markings that went from not alive to alive in the commit
  uint32_t n_unmark_pts; // fused route: generator points of removed markings that have to be walked point by point
  uint32_t n_cleared_shard[32];   // fused route: removals, counted in shards (their sum = n_cleared)
};

// isinLidarObservation (:682-746).  The reference builds a rotation that turns the x axis onto the viewing
// direction, multiplies it with the inverse sensor rotation and reads the yaw back through a
// matrix -> quaternion -> matrix round trip; column 0 of that product is the viewing direction in the sensor
// frame, so the yaw is formed from it directly (differences ~1e-16 rad, far below any threshold distance).
__device__ inline bool in_lidar_observation(const MarkParams& k, const float pcx, const float pcy, const float pcz) {
  const double p2plane = pcx * k.sn[0] + pcy * k.sn[1] + pcz * k.sn[2] + k.sd;
  const double dx = pcx - k.st[0], dy = pcy - k.st[1], dz = pcz - k.st[2];
  const double p2s = sqrt(dx * dx + dy * dy + dz * dz);
  const double result = asin(p2plane / p2s) * 180.0 / 3.1415926535;
  if (result < k.fov_bottom || result > k.fov_top) return false;
  const double ax = dx / p2s, ay = dy / p2s, az = dz / p2s;
  // A viewing direction exactly along the GLOBAL x axis makes the reference's rotation axis (axis x (1,0,0)) the
  // zero vector: tf2::Quaternion(axis, angle) divides by its length, the yaw comes out NaN, every comparison below
  // fails and the function falls through to `return true` (:741-746).
  if (ay == 0.0 && az == 0.0) return true;
  const double sx = k.Rs[0] * ax + k.Rs[3] * ay + k.Rs[6] * az;      // Rs^T * axis
  const double sy = k.Rs[1] * ax + k.Rs[4] * ay + k.Rs[7] * az;
  const double sz = k.Rs[2] * ax + k.Rs[5] * ay + k.Rs[8] * az;
  double yaw = 0.0;
  if (fabs(sz) < 1.0) {
    const double pitch = -asin(sz);
    yaw = atan2(sy / cos(pitch), sx / cos(pitch));
  }
  const double r = fmod(yaw + M_PI, 2.0 * M_PI);                    // angles::shortest_angular_distance(0, yaw)
  yaw = r <= 0.0 ? r + M_PI : r - M_PI;
  yaw = yaw * 180.0 / 3.1415926535;
  if (yaw >= 0 && (yaw < k.ps || yaw > k.pe)) return false;
  else if (yaw < 0 && (yaw > k.ns || yaw < k.ne)) return false;
  return true;
}

__device__ __forceinline__ unsigned long long voxel_key(int x, int y, int z) {   // bit 63 set: 0 = empty slot
  return (1ull << 63) | ((unsigned long long)((uint32_t)(x + (1 << 20)) & 0x1FFFFFu) << 42) |
         ((unsigned long long)((uint32_t)(y + (1 << 20)) & 0x1FFFFFu) << 21) |
         (unsigned long long)((uint32_t)(z + (1 << 20)) & 0x1FFFFFu);
}
__host__ __device__ __forceinline__ void voxel_unkey(unsigned long long key, int* x, int* y, int* z) {
  *x = (int)((key >> 42) & 0x1FFFFFu) - (1 << 20);
  *y = (int)((key >> 21) & 0x1FFFFFu) - (1 << 20);
  *z = (int)(key & 0x1FFFFFu) - (1 << 20);
}
__device__ __forceinline__ uint32_t mk_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (uint32_t)k;
}

struct MarkStore {            // the persistent store (device pointers)
  unsigned long long* keys;   // [table] voxel key or 0
  uint32_t* alive;            // [table]
  uint32_t* pts_ofs;          // [table] first generator point in the pool
  uint32_t* pts_n;            // [table]
  uint32_t* removed_seq;      // [table] update that cleared the slot
  unsigned long long* owner;  // [table] priority of the cluster that takes the slot in this update (0 = none)
  uint32_t* alive_list;       // [table] slots alive at the end of the last update (what selfClear walks)
  uint32_t* removed_list;     // [table] slots this update's selfClear removed
  uint32_t* fov_flag;         // [table] per alive_list entry: 1 = inside window and sensor view (needs the ray test)
  float4* pool;               // generator points
  double* dgraph;             // [n_ground + 1]
  uint8_t* lethal;            // [n_ground + 1]
};

// ---------------------------------------------------------------------------------------------
// selfClear: one wave per store slot
// ---------------------------------------------------------------------------------------------
// Window + field-of-view test of every stored marking, one LANE each (the double asin / atan2 / cos of
// isinLidarObservation cost a whole wave as much as a lane).
__global__ __launch_bounds__(256) void k_mk_fov(MarkParams k, MarkStore s, MarkCounters* __restrict__ cnt) {
  const uint32_t w = blockIdx.x * 256 + threadIdx.x;
  if (w >= k.n_alive_prev) return;
  const uint32_t slot = s.alive_list[w];
  uint32_t flag = 0;
  if (s.alive[slot]) {
    int x, y, z;
    voxel_unkey(s.keys[slot], &x, &y, &z);
    // map iteration lower_bound(min) .. lower_bound(max): keys in [min, max) on every axis (:487-516)
    if (!(x < k.wx0 || x >= k.wx1 || y < k.wy0 || y >= k.wy1 || z < k.wz0 || z >= k.wz1)) {
      atomicAdd(&cnt->n_in_window, 1u);
      const float px = (float)(x * k.res), py = (float)(y * k.res), pz = (float)(z * k.hres);
      flag = in_lidar_observation(k, px, py, pz) ? 1u : 0u;         // outside the sensor's view: stays (:531-540)
    }
  }
  s.fov_flag[w] = flag;
}

#ifdef DDDMR_PHASE_STAMPS
__device__ int g_mk_exp_noprobe;                    // diagnostic build: 1 = the ray probes look nothing up (wrong results: timing only)
#endif
// -DDDDMR_CLEAR_CYCLES (on top of the diagnostic build): per-ray cycle sums -- their same-address atomics make the launch
// itself ~20x slower, so they are only good for shares, never next to kernel times
#ifdef DDDMR_CLEAR_CYCLES
__device__ unsigned long long g_mk_clear_cyc[8];   // [0] rays, [1] chunks probed, cycles of [2] phase A, [3] phase B, [4] near test, [5] removal
#define MKC_NOW() clock64()
#define MKC_ADD(i, v) do { if (lane == 0) atomicAdd(&g_mk_clear_cyc[i], (unsigned long long)(v)); } while (0)
#else
#define MKC_NOW() 0ll
#define MKC_ADD(i, v) do { (void)sizeof(v); } while (0)
#endif
// The ray test of one stored marking by one wave; true (wave-uniform) when the marking was removed.
// kListed: the removed slot goes on MarkStore::removed_list (what k_mk_unmark walks) and is counted in n_cleared / n_removed,
// two same-address device-scope atomics per removal (~12 ns each, one after the other: 1400 removals = 30 us of the
// launch).  The fused route keeps no such list and counts removals in 32 shards of n_cleared_shard instead.
template <bool kListed>
__device__ __forceinline__ bool mk_clear_wave(const MarkParams& k, const MarkStore& s, const PointGrid& prev, MarkCounters* __restrict__ cnt,
                                              const uint32_t slot, const int lane) {
  int x, y, z;
  voxel_unkey(s.keys[slot], &x, &y, &z);
  const float px = (float)(x * k.res), py = (float)(y * k.res), pz = (float)(z * k.hres);
  const bool observation_clear = !(k.n_prev > 5);
  bool blocked = false;
  const long long mkc0 = MKC_NOW();
  MKC_ADD(0, 1);
  if (!observation_clear) {
    // getCastingPointCloud: points every 5 cm from the sensor to the voxel, t accumulated in float
    const float dX = (float)(px - k.st[0]), dY = (float)(py - k.st[1]), dZ = (float)(pz - k.st[2]);
    float distance = sqrtf(dX * dX + dY * dY + dZ * dZ);
    const float len = distance;                                     // metres from the sensor to the voxel
    distance = (float)(distance / 0.05);
    const float dt = 1 / distance;
    float t0 = 0.f;                                                 // t of lane 0 in this chunk of 64 ray points
    for (int chunk = 0; chunk < 4096; ++chunk) {
      MKC_ADD(1, 1);
      // The reference's running float sum t += dt: lane i holds the i-th partial sum.  One chain for the wave: every step
      // each lane takes its left neighbour's value + dt (DPP wave_shr:1) and lane 0 is put back to t0, so after 63 steps
      // lane i has been through exactly i additions, in order.
      // (one instruction per step: v_add_f32_dpp with bound_ctrl off leaves lane 0, which has no left neighbour, unwritten;
      // the compiler's form of the same chain needs a v_cndmask per step to put t0 back.  s_nop 1: the two wait states a
      // DPP read of a VGPR written by the previous VALU instruction needs.)
      float t = t0;
#pragma unroll
      for (int j = 0; j < 63; ++j)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(dt));
      const bool live = t <= 1.0;
      bool stop = false, hit = false;
      if (live) {
        const float ax = (float)(k.st[0] + dX * t), ay = (float)(k.st[1] + dY * t), az = (float)(k.st[2] + dZ * t);
        // search radius min(intensity / 20 + 0.01, 0.1), intensity = distance to the voxel: 0.1 beyond 1.8 m.  A float
        // estimate of that distance (error < 1 mm for rays of tens of metres) settles all but the last 2 m of the ray;
        // there the reference's double arithmetic runs.
        const float est = len * (1.0f - t);
        double sd = 0.1;
        if (est < 2.0f) {
          const double ddx = px - ax, ddy = py - ay, ddz = pz - az;   // getDistanceBTWPoints
          const float intensity = (float)sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
          stop = intensity < 0.05;                                  // the last 5 cm are ignored (:563-564)
          sd = intensity / 20. + 0.01;
          sd = fmin(sd, 0.1);
        }
        if (!stop) {
          const float r2 = static_cast<float>(sd * sd);
#ifdef DDDMR_PHASE_STAMPS
          if (!g_mk_exp_noprobe)
#endif
          hit = grid_radius_count(prev, ax, ay, az, (float)sd + k.pad, r2, 1) > 0;
        }
      }
      const unsigned long long m_end = __ballot(!live || stop);    // first lane at which the reference's loop ends
      const unsigned long long m_hit = __ballot(hit);
      const unsigned long long before = m_end ? ((m_end & (0ull - m_end)) - 1ull) : ~0ull;   // lanes ahead of it
      if (m_hit & before) { blocked = true; break; }
      if (m_end) break;
      t0 = __shfl(t, 63, 64) + dt;
    }
  }
  const long long mkc2 = MKC_NOW();
  MKC_ADD(3, mkc2 - mkc0);
  if (blocked) return false;                                        // the ray is blocked: keep (:582-591)
  int near = 0;
  if (!observation_clear) {
    const float r2 = static_cast<float>(k.res * k.res);
    near = grid_radius_count(prev, px, py, pz, (float)k.res + k.pad, r2, 2);
  }
  const long long mkc3 = MKC_NOW();
  MKC_ADD(4, mkc3 - mkc2);
  if (near > 1) return false;                                       // still observed (:596-605)
  if (lane == 0) {                                                  // Marking::removePCPtr
    s.alive[slot] = 0;
    s.removed_seq[slot] = k.seq;
    if (kListed) {
      atomicAdd(&cnt->n_cleared, 1u);
      s.removed_list[atomicAdd(&cnt->n_removed, 1u)] = slot;
    } else {
      atomicAdd(&cnt->n_cleared_shard[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 31u], 1u);
    }
  }
  MKC_ADD(5, MKC_NOW() - mkc3);
  return true;
}
__global__ __launch_bounds__(256) void k_mk_clear(MarkParams k, MarkStore s, PointGrid prev, MarkCounters* __restrict__ cnt) {
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= k.n_alive_prev || !s.fov_flag[w]) return;
  (void)mk_clear_wave<true>(k, s, prev, cnt, s.alive_list[w], threadIdx.x & 63);
}

// removePCPtr's loop over nodes_of_min_distance_, recomputed from the marking's generator points: every ground node
// within inflation_radius (3-D) of one gets clearValue(node, 9999.0); erased from the lethal set where the xy distance
// is within the inscribed radius.  One wave per slot cleared in this update.
__global__ __launch_bounds__(256) void k_mk_unmark(MarkParams k, MarkStore s, PointGrid ground, const MarkCounters* __restrict__ cnt) {
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= cnt->n_removed) return;
  const uint32_t slot = s.removed_list[w];
  const uint32_t ofs = s.pts_ofs[slot], n = s.pts_n[slot];
  const float r = (float)k.inflation, r2 = static_cast<float>(k.inflation * k.inflation);
  for (uint32_t i = 0; i < n; ++i) {                                // (a marking has a handful of generator points)
    const float4 p = s.pool[ofs + i];
    grid_for_each_wave(ground, p.x, p.y, p.z, r + 1e-4f, lane, [&](const float4 g) {
      if (l2_simple(g.x, g.y, g.z, p.x, p.y, p.z) < r2) {
        const int node = __float_as_int(g.w);
        const float dx = p.x - g.x, dy = p.y - g.y;
        const float d = sqrtf(dx * dx + dy * dy);
        s.dgraph[node] = 9999.0;
        if (d <= k.inscribed) s.lethal[node] = 0;
      }
    });
  }
}

// ---------------------------------------------------------------------------------------------
// selfMark
// ---------------------------------------------------------------------------------------------
// Euclidean clustering = connected components of "closer than the tolerance" (pcl::extractEuclideanClusters):
// lock-free union-find, the larger root is hooked under the smaller one, so a component's root is its smallest
// point index -- the seed PCL would have started the cluster from.
__device__ __forceinline__ uint32_t cc_find(uint32_t* parent, uint32_t i) {
  uint32_t p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != i) {
    i = p;
    p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return i;
}
__global__ __launch_bounds__(256) void k_mk_cc_init(uint32_t n, uint32_t* __restrict__ parent) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) parent[i] = i;
}
__global__ __launch_bounds__(256) void k_mk_cc_union(MarkParams k, PointGrid obs, const float4* __restrict__ pts, uint32_t* parent) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k.n_obs) return;
  const float4 p = pts[i];
  grid_for_each(obs, p.x, p.y, p.z, k.tol + 1e-4f, [&](const float4 q) {
    const uint32_t j = (uint32_t)__float_as_int(q.w);
    if (j < i && l2_simple(q.x, q.y, q.z, p.x, p.y, p.z) < k.tol2) {
      uint32_t u = cc_find(parent, i), v = cc_find(parent, j);
      while (u != v) {
        if (u < v) { const uint32_t t = u; u = v; v = t; }          // u is the larger root
        const uint32_t old = atomicCAS(&parent[u], u, v);
        if (old == u) break;
        u = cc_find(parent, old);
        v = cc_find(parent, v);
      }
    }
    return false;
  });
}
// sort key of point i: (root index << 20) | i -- clusters in seed order, their points in index order
__global__ __launch_bounds__(256) void k_mk_cc_keys(uint32_t n, uint32_t* parent, unsigned long long* __restrict__ keys) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = ((unsigned long long)cc_find(parent, i) << 20) | i;
}
// flags of group starts in a sorted key array (invalid keys = ~0 sort to the end and start nothing)
__global__ __launch_bounds__(256) void k_mk_flags(uint32_t n, const unsigned long long* __restrict__ keys, int shift,
                                                  uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long a = keys[i];
  flags[i] = (a != ~0ull && (i == 0 || (keys[i - 1] >> shift) != (a >> shift))) ? 1u : 0u;
}

struct ClusterArrays {
  uint32_t* start;     // [n] first position (in sort-1 order) of the cluster
  uint32_t* size;      // [n]
  float4* centroid;    // [n] xyz
  uint32_t* state;     // [n] 0 rejected, 1 passed the size / ground tests, 2 accepted (in the sensor's view)
  uint32_t* ds_count;  // [n] points after the 0.2 m VoxelGrid
  uint32_t* gen_first; // [n] first generator point (sort-3 group index)
  uint32_t* gen_count; // [n]
  uint32_t* slot;      // [n] store slot of an accepted cluster
  int* vkey;           // [n][3] voxel key of the centroid
};

__global__ __launch_bounds__(256) void k_mk_cluster_starts(uint32_t n, const uint32_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ cid_incl, ClusterArrays c,
                                                           MarkCounters* __restrict__ cnt) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) c.start[cid_incl[i] - 1] = i;
  if (i == n - 1) { c.start[cid_incl[i]] = n; cnt->n_clusters = cid_incl[i]; }
}

// per cluster: centroid (floats added in index order, / size), min size, "centre attached to the ground" (:364-368)
__global__ __launch_bounds__(64) void k_mk_cluster_stage1(MarkParams k, MarkCounters* __restrict__ cnt, ClusterArrays c,
                                                          const unsigned long long* __restrict__ keys1,
                                                          const float4* __restrict__ pts, PointGrid ground) {
  const uint32_t ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= cnt->n_clusters) return;
  const uint32_t b = c.start[ci], e = c.start[ci + 1];
  float cx = 0.f, cy = 0.f, cz = 0.f;
  for (uint32_t m = b; m < e; ++m) {
    const float4 p = pts[(uint32_t)(keys1[m] & 0xFFFFFu)];
    cx += p.x; cy += p.y; cz += p.z;
  }
  const float sz = (float)(e - b);
  cx /= sz; cy /= sz; cz /= sz;
  c.size[ci] = e - b;
  c.centroid[ci] = make_float4(cx, cy, cz, 0.f);
  c.ds_count[ci] = 0;
  c.gen_count[ci] = 0;
  c.gen_first[ci] = 0xFFFFFFFFu;
  bool ok = (int)(e - b) >= k.min_cluster;
  if (ok) atomicAdd(&cnt->n_clusters_kept, 1u);        // what extractEuclideanClusters returns (min_pts_per_cluster)
  if (ok && grid_radius_count(ground, cx, cy, cz, 0.05f + 1e-4f, static_cast<float>(0.05 * 0.05), 1) > 0) ok = false;
  c.state[ci] = ok ? 1u : 0u;
}

// VoxelGrid keys.  A PCL voxel is floor(p * inverse_leaf) per axis (the grid's min_b only shifts the indices), and the
// output order inside one cloud is x fastest, then y, then z: key = cluster | z | y | x, 16 + 16 bits for x and y and 10
// for z around `org` = the window's centre - half the range: +-3.2 km in x / y and +-51 m in z at 0.1 m.  A point of a
// live cluster beyond that cannot be keyed: capacity flag 4, the update answers DDDMR_ERR_CAPACITY instead of dropping
// the point.
constexpr int kVgHalfXY = 32768, kVgHalfZ = 512;
__device__ __forceinline__ unsigned long long vg_key(uint32_t ci, float x, float y, float z, float inv, int ox, int oy, int oz,
                                                     MarkCounters* __restrict__ cnt) {
  const int ix = (int)floorf(x * inv) - ox, iy = (int)floorf(y * inv) - oy, iz = (int)floorf(z * inv) - oz;
  if ((unsigned)ix >= 65536u || (unsigned)iy >= 65536u || (unsigned)iz >= 1024u) { atomicOr(&cnt->overflow, 4u); return ~0ull; }
  return ((unsigned long long)ci << 42) | ((unsigned long long)iz << 32) | ((unsigned long long)iy << 16) | (unsigned long long)ix;
}
// 0.2 m VoxelGrid of the clusters that passed stage 1 (:370-374): key per point, in sort-1 order
__global__ __launch_bounds__(256) void k_mk_ds_keys(MarkParams k, const unsigned long long* __restrict__ keys1,
                                                    const uint32_t* __restrict__ cid_incl, ClusterArrays c,
                                                    const float4* __restrict__ pts, int ox, int oy, int oz,
                                                    unsigned long long* __restrict__ keys2, uint32_t* __restrict__ vals2,
                                                    MarkCounters* __restrict__ cnt) {
  const uint32_t m = blockIdx.x * 256 + threadIdx.x;
  if (m >= k.n_obs) return;
  const uint32_t ci = cid_incl[m] - 1;
  unsigned long long key = ~0ull;
  if (c.state[ci]) {
    const float4 p = pts[(uint32_t)(keys1[m] & 0xFFFFFu)];
    key = vg_key(ci, p.x, p.y, p.z, 1.0f / 0.2f, ox, oy, oz, cnt);
  }
  keys2[m] = key;
  vals2[m] = m;
}
// one lane per group of a sorted (key, value) array: floats added in order, / count (pcl CentroidPoint)
// mode 0: values are sort-1 positions of observation points; mode 1: values index `src` directly
__global__ __launch_bounds__(64) void k_mk_group_reduce(uint32_t n, const unsigned long long* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals, const uint32_t* __restrict__ flags,
                                                        const uint32_t* __restrict__ gid_incl, int mode,
                                                        const unsigned long long* __restrict__ keys1,
                                                        const float4* __restrict__ src, float4* __restrict__ out,
                                                        uint32_t* __restrict__ group_count, uint32_t* __restrict__ group_first,
                                                        uint32_t* __restrict__ n_groups) {
  const uint32_t m = blockIdx.x * 64 + threadIdx.x;
  if (m >= n) return;
  if (m == n - 1) *n_groups = gid_incl[m];
  if (!flags[m]) return;
  const unsigned long long key = keys[m];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  uint32_t e = m;
  for (; e < n && keys[e] == key; ++e) {
    const uint32_t v = vals[e];
    const float4 p = src[mode == 0 ? (uint32_t)(keys1[v] & 0xFFFFFu) : v];
    sx += p.x; sy += p.y; sz += p.z;
  }
  const float cntf = (float)(e - m);
  const uint32_t g = gid_incl[m] - 1, ci = (uint32_t)(key >> 42);
  out[g] = make_float4(sx / cntf, sy / cntf, sz / cntf, __int_as_float((int)ci));
  atomicAdd(&group_count[ci], 1u);
  atomicMin(&group_first[ci], g);
}

// per cluster that passed stage 1: "is it part of the static map" (:375-389), voxel key, in the sensor's view (:425-430)
__global__ __launch_bounds__(64) void k_mk_cluster_stage2(MarkParams k, const MarkCounters* __restrict__ cnt, ClusterArrays c,
                                                          PointGrid map, uint32_t n_map) {
  const uint32_t ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= cnt->n_clusters || c.state[ci] != 1u) return;
  const float4 cen = c.centroid[ci];
  const size_t nds = c.ds_count[ci];
  size_t hit = 0;
  if (k.ignore_ratio <= 0.999) {
    // the loop searches with the CENTROID for every downsampled point (:380): all hit or none do
    const bool near = n_map > 0 && grid_radius_count(map, cen.x, cen.y, cen.z, 0.1f + 1e-4f, static_cast<float>(0.1 * 0.1), 1) > 0;
    if (near)
      for (size_t a = 0; a < nds; ++a) {
        hit++;
        if (hit > nds * k.ignore_ratio) break;
      }
  }
  if (!(hit <= nds * k.ignore_ratio)) { c.state[ci] = 0u; return; }
  const int vx = (int)(cen.x / k.res), vy = (int)(cen.y / k.res), vz = (int)(cen.z / k.hres);
  c.vkey[3 * ci + 0] = vx; c.vkey[3 * ci + 1] = vy; c.vkey[3 * ci + 2] = vz;
  const float px = (float)(vx * k.res), py = (float)(vy * k.res), pz = (float)(vz * k.hres);
  c.state[ci] = in_lidar_observation(k, px, py, pz) ? 2u : 0u;
}

// ProjectInliers(SACMODEL_PLANE) of the downsampled points of accepted clusters + 0.1 m VoxelGrid keys
// (cluster_marking.cpp:54-64).  Eigen's SSE reduction order for the 4-float dot product: (a0 + a2) + (a1 + a3).
__global__ __launch_bounds__(256) void k_mk_proj_keys(MarkParams k, const uint32_t* __restrict__ n_ds, const float4* __restrict__ ds,
                                                      ClusterArrays c, int ox, int oy, int oz, float4* __restrict__ proj,
                                                      unsigned long long* __restrict__ keys3, uint32_t* __restrict__ vals3,
                                                      uint32_t n_pad, MarkCounters* __restrict__ cnt) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n_pad) return;
  unsigned long long key = ~0ull;
  if (g < *n_ds) {
    const float4 p = ds[g];
    const uint32_t ci = (uint32_t)__float_as_int(p.w);
    if (c.state[ci] == 2u) {
      float m0 = k.mc[0], m1 = k.mc[1], m2 = k.mc[2], m3 = 0.0f;
      const float nrm = sqrtf((m0 * m0 + m2 * m2) + (m1 * m1 + m3 * m3));
      m0 = m0 / nrm; m1 = m1 / nrm; m2 = m2 / nrm;
      const float dist = (m0 * p.x + m2 * p.z) + (m1 * p.y + k.mc[3] * 1.0f);
      const float qx = p.x - m0 * dist, qy = p.y - m1 * dist, qz = p.z - m2 * dist;
      proj[g] = make_float4(qx, qy, qz, p.w);
      key = vg_key(ci, qx, qy, qz, 1.0f / 0.1f, ox, oy, oz, cnt);
    }
  }
  keys3[g] = key;
  vals3[g] = g;
}

// Marking::addPCPtr, slot part: marking_[x][y][z] is created or found; when several clusters of one scan land on the
// same voxel the last one in PCL's order (clusters sorted by size, descending) keeps the slot.  Among clusters of
// EQUAL size that order is whatever libstdc++'s introsort leaves (std::sort over reverse iterators,
// pcl/segmentation/impl/extract_clusters.hpp): the priority below breaks such ties by cluster index, and the host
// replays the very sort for the updates that have a contested voxel at all (n_dup > 0, marking_fix_ties).
__global__ __launch_bounds__(64) void k_mk_slots(MarkParams k, const MarkCounters* __restrict__ cnt_in, ClusterArrays c, MarkStore s,
                                                 MarkCounters* __restrict__ cnt) {
  const uint32_t ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= cnt_in->n_clusters || c.state[ci] != 2u) return;
  const unsigned long long key = voxel_key(c.vkey[3 * ci], c.vkey[3 * ci + 1], c.vkey[3 * ci + 2]);
  uint32_t slot = mk_hash(key) & k.table_mask;
  bool found = false;
  for (uint32_t probe = 0; probe <= k.table_mask; ++probe) {
    const unsigned long long prev = atomicCAS(&s.keys[slot], 0ull, key);
    if (prev == 0ull) atomicAdd(&cnt->n_new_keys, 1u);
    if (prev == 0ull || prev == key) { found = true; break; }
    slot = (slot + 1) & k.table_mask;
  }
  if (!found) { atomicOr(&cnt->overflow, 1u); c.state[ci] = 3u; return; }   // store full: the cluster still updates the dGraph
  c.slot[ci] = slot;
  const unsigned long long pr = ((unsigned long long)((1u << 20) - min(c.size[ci], (1u << 20) - 1u)) << 20) | (unsigned long long)(ci + 1u);
  if (atomicMax(&s.owner[slot], pr) != 0ull) atomicAdd(&cnt->n_dup, 1u);
  atomicAdd(&cnt->n_marked, 1u);
}
// ... storage part: the winning cluster's generator points go to the pool
__global__ __launch_bounds__(64) void k_mk_commit(MarkParams k, const MarkCounters* __restrict__ cnt_in, ClusterArrays c, MarkStore s,
                                                  MarkCounters* __restrict__ cnt, uint32_t* __restrict__ pool_ofs) {
  const uint32_t ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= cnt_in->n_clusters) return;
  pool_ofs[ci] = 0xFFFFFFFFu;
  if (c.state[ci] != 2u) return;
  const uint32_t slot = c.slot[ci];
  const unsigned long long pr = ((unsigned long long)((1u << 20) - min(c.size[ci], (1u << 20) - 1u)) << 20) | (unsigned long long)(ci + 1u);
  if (s.owner[slot] != pr) return;
  const uint32_t n = c.gen_count[ci];
  const uint32_t ofs = atomicAdd(&cnt->pool_used, n);
  if (ofs + n > k.pool_cap) { atomicOr(&cnt->overflow, 2u); s.alive[slot] = 0; s.pts_n[slot] = 0; return; }
  pool_ofs[ci] = ofs;
  s.pts_ofs[slot] = ofs;
  s.pts_n[slot] = n;
  s.alive[slot] = 1;
}
// one lane per generator point of an accepted cluster: copy into the pool (winner of its voxel) and
// computeMinDistanceFromObstacle2GroundNodes + DynamicGraph::setValue + lethal_map_ (cluster_marking.cpp:66-123)
__global__ __launch_bounds__(256) void k_mk_dgraph(MarkParams k, const uint32_t* __restrict__ n_gen, const float4* __restrict__ gen,
                                                   ClusterArrays c, const uint32_t* __restrict__ pool_ofs, MarkStore s,
                                                   PointGrid ground) {
  const uint32_t h = blockIdx.x * 4 + (threadIdx.x >> 6);          // one WAVE per generator point, lanes over ground nodes
  const int lane = threadIdx.x & 63;
  if (h >= *n_gen) return;
  const float4 p = gen[h];
  const uint32_t ci = (uint32_t)__float_as_int(p.w);
  const uint32_t po = pool_ofs[ci];
  if (lane == 0 && po != 0xFFFFFFFFu) s.pool[po + (h - c.gen_first[ci])] = make_float4(p.x, p.y, p.z, 0.f);
  const float r = (float)k.inflation, r2 = static_cast<float>(k.inflation * k.inflation);
  grid_for_each_wave(ground, p.x, p.y, p.z, r + 1e-4f, lane, [&](const float4 g) {
    if (l2_simple(g.x, g.y, g.z, p.x, p.y, p.z) < r2) {
      const int node = __float_as_int(g.w);
      const float dx = p.x - g.x, dy = p.y - g.y;
      const float d = sqrtf(dx * dx + dy * dy);                    // z dropped on purpose (:86-88)
      // setValue: graph_[key] = min(graph_[key], d); non-negative doubles order like their bit patterns
      atomicMin(reinterpret_cast<unsigned long long*>(s.dgraph) + node, (unsigned long long)__double_as_longlong((double)d));
      if (d <= k.inscribed) s.lethal[node] = 1;
    }
  });
}
// marking_fix_ties: the generator points of the cluster that keeps a contested voxel in the reference's order replace
// the ones k_mk_commit stored (those become pool garbage).  One wave per (slot, cluster) pair.
__global__ __launch_bounds__(256) void k_mk_fix_owner(MarkParams k, uint32_t n_fix, const uint2* __restrict__ fix,
                                                      const float4* __restrict__ gen, ClusterArrays c, MarkStore s,
                                                      MarkCounters* __restrict__ cnt) {
  const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (f >= n_fix) return;
  const uint32_t slot = fix[f].x, ci = fix[f].y;
  const uint32_t n = c.gen_count[ci], first = c.gen_first[ci];
  uint32_t ofs = 0;
  if (lane == 0) ofs = atomicAdd(&cnt->pool_used, n);
  ofs = (uint32_t)__builtin_amdgcn_readfirstlane((int)ofs);
  if (ofs + n > k.pool_cap) {
    if (lane == 0) { atomicOr(&cnt->overflow, 2u); s.alive[slot] = 0; s.pts_n[slot] = 0; }
    return;
  }
  for (uint32_t i = lane; i < n; i += 64) {
    const float4 p = gen[first + i];
    s.pool[ofs + i] = make_float4(p.x, p.y, p.z, 0.f);
  }
  if (lane == 0) { s.pts_ofs[slot] = ofs; s.pts_n[slot] = n; s.alive[slot] = 1; }
}
__global__ __launch_bounds__(256) void k_mk_finish(MarkParams k, MarkStore s, MarkCounters* __restrict__ cnt) {
  const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
  if (slot > k.table_mask) return;
  s.owner[slot] = 0ull;
  if (s.alive[slot]) s.alive_list[atomicAdd(&cnt->n_alive, 1u)] = slot;      // next update's selfClear walks this list
}
// Store garbage collection.  A voxel whose marking was cleared keeps its key (the reference's marking_[x][y][z] entry
// stays too, with has_pc false: selfClear skips it, addPCPtr overwrites it), so a robot that keeps moving fills the
// table with keys nothing reads any more.  When half the table is used the alive markings move to a fresh table;
// the dropped keys change nothing the reference can observe.  One lane per old slot; the new alive list is built
// on the way (its order is irrelevant: selfClear treats every marking on its own).
__global__ __launch_bounds__(256) void k_mk_rehash(uint32_t table_mask, MarkStore s, unsigned long long* __restrict__ keys_new,
                                                   uint32_t* __restrict__ alive_new, uint32_t* __restrict__ pts_ofs_new,
                                                   uint32_t* __restrict__ pts_n_new, MarkCounters* __restrict__ cnt) {
  const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
  if (slot > table_mask || !s.alive[slot]) return;
  const unsigned long long key = s.keys[slot];
  uint32_t ns = mk_hash(key) & table_mask;
  for (uint32_t probe = 0; probe <= table_mask; ++probe) {          // (alive markings are fewer than slots: always ends)
    if (atomicCAS(&keys_new[ns], 0ull, key) == 0ull) break;
    ns = (ns + 1) & table_mask;
  }
  alive_new[ns] = 1;
  pts_ofs_new[ns] = s.pts_ofs[slot];
  pts_n_new[ns] = s.pts_n[slot];
  s.alive_list[atomicAdd(&cnt->n_rehashed, 1u)] = ns;
}
// pool compaction: generator points of the alive markings move to the front of the other pool buffer
__global__ __launch_bounds__(256) void k_mk_compact_sizes(uint32_t table, MarkStore s, uint32_t* __restrict__ sizes) {
  const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
  if (slot < table) sizes[slot] = s.alive[slot] ? s.pts_n[slot] : 0u;
}
__global__ __launch_bounds__(256) void k_mk_compact_move(uint32_t table, MarkStore s, const uint32_t* __restrict__ new_ofs,
                                                         float4* __restrict__ dst, MarkCounters* __restrict__ cnt) {
  const uint32_t slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (slot >= table || !s.alive[slot]) return;
  const uint32_t n = s.pts_n[slot], from = s.pts_ofs[slot], to = new_ofs[slot];
  for (uint32_t i = lane; i < n; i += 64) dst[to + i] = s.pool[from + i];
  if (lane == 0) {
    s.pts_ofs[slot] = to;
    atomicMax(&cnt->pool_used, to + n);
  }
}
__global__ void k_mk_fill_dgraph(uint32_t n, double* __restrict__ dgraph, double v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dgraph[i] = v;
}

}  // namespace dddmr
