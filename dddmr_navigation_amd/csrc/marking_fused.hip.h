// marking_fused.hip.h -- the marking / clearing update for observations of up to 16384 points in FOUR launches.
//
// The general route (marking.hip.h + rocPRIM sorts, any observation size) issues ~53 launches per update, ~30 of them
// at the 4.5-4.9 us launch floor, plus three mid-update copies: for the 6 k-point observation of a 16-line LiDAR the
// update is launch-bound (profiles/r02_C5M_kernel_stats.csv).  Here every grouping step (the three stable sorts, their
// flag / scan / reduce rounds, the per-cluster stages, the hash insert and the commit) runs inside ONE 1024-lane
// workgroup with the sort keys in registers and the exchange buffers in LDS, and the chip-wide steps share launches
// with each other:
//
//   launch 1  k_mkf_pre             block 0: grid of the new observation (zero, count, scan, scatter: one workgroup)
//                                   blocks 1..: window + field-of-view test of every stored marking
//   launch 2  k_mkf_clear_cc        selfClear ray tests (a wave per marking)  |  Euclidean clustering (union-find)
//   launch 3  k_mkf_unmark_groups   block 0: clusters -> centroids -> 0.2 m VoxelGrid -> static / FOV tests ->
//                                   projection + 0.1 m VoxelGrid -> store slots -> commit
//                                   blocks 1..: removePCPtr of the markings launch 2 cleared
//   launch 4  k_mkf_dgraph_finish   dGraph / lethal update of the new generator points  |  next update's alive list;
//                                   the last block publishes the counters to host-mapped memory and zeroes them
//
// What each step computes, and the reference lines it follows, is unchanged from marking.hip.h: the same float
// operations in the same order (cluster / voxel centroids are sequential float sums in point-index order, which is
// why the sorts must be stable), so both routes give bit-identical stores, dGraphs and lethal sets.
#pragma once
#include "marking.hip.h"

#pragma clang fp contract(off)

namespace dddmr {

constexpr int kFuseThreads = 1024;
constexpr uint32_t kFuseMaxObs = 16384;      // points of an observation the fused route takes (16 per lane)
constexpr uint32_t kFuseMaxCells = 32768;    // cells of the observation grid one workgroup scans
constexpr int kFuseKeyBits = 28;             // voxel sort keys: 4 passes of 7 bits at most (sentinel = all ones)

struct FuseBufs {
  const float4* pts;       // this update's observation (global frame)
  uint32_t* parent;        // [n] union-find
  float4* spts;            // [n] observation in cluster order (sort 1)
  float4* ds;              // [n] 0.2 m voxel centroids, w = cluster
  float4* gen;             // [n] generator points, w = cluster
  uint32_t* pool_ofs;      // [n] per cluster: first pool entry of its generator points, ~0 = not the keeper
  uint2* removed_on;       // [table] (pool offset, count) of the markings this update's selfClear removed
  uint32_t* ticket;        // [2]
  MarkCounters* host_out;  // host-mapped copy of the update's counters
};

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

// exclusive prefix of one value per lane over the 1024-lane block, in lane order (two barriers; wsum = 16 LDS words)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan_u32(v);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t x = wsum[j];
    tot += x;
    if (j < w) base += x;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// The block's elements live in registers, E per lane, "wave-blocked": position = wave * 64 E + slot * 64 + lane.
template <int E>
__device__ __forceinline__ uint32_t fpos(int s) {
  return (uint32_t)((threadIdx.x >> 6) * 64 * E + s * 64 + (threadIdx.x & 63));
}
// exclusive prefix over positions of one flag per element (bit s of `flags` = slot s)
template <int E>
__device__ __forceinline__ void pos_excl_scan(const uint32_t flags, uint32_t (&ex)[E], uint32_t* wsum, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t carry = 0;
#pragma unroll
  for (int s = 0; s < E; ++s) {
    const unsigned long long b = __ballot(((flags >> s) & 1u) != 0u);
    ex[s] = carry + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    carry += (uint32_t)__popcll(b);
  }
  if (lane == 0) wsum[w] = carry;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t x = wsum[j];
    tot += x;
    if (j < w) base += x;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < E; ++s) ex[s] += base;
  *total = tot;
}

// One stable LSD radix pass over a 7-bit digit.  Ranking: the lanes of a wave that hold the same digit find each
// other with seven ballots (their rank among themselves = lanes below in the match mask), a per-wave histogram
// column in LDS carries the count over the wave's slots, one block scan over (digit, wave) turns the columns into
// bases.  Keys and payloads go through the LDS exchange buffers and come back in position order, so after the pass
// xk / xp also hold the whole sorted sequence.  Keys of all ones keep to the end (digit 127 at every shift < 22).
template <int E>
__device__ __forceinline__ void radix_pass(uint32_t (&key)[E], uint32_t (&pay)[E], const int shift, uint32_t* hist, uint32_t* xk,
                                           uint16_t* xp, uint32_t* wsum) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  hist[tid] = 0u;
  hist[tid + 1024] = 0u;
  __syncthreads();
  uint32_t rank[E];
#pragma unroll
  for (int s = 0; s < E; ++s) {
    const uint32_t d = (key[s] >> shift) & 127u;
    unsigned long long m = ~0ull;
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const bool bit = ((d >> b) & 1u) != 0u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t pre = hist[d * 16 + w];
    rank[s] = pre + below;
    __builtin_amdgcn_wave_barrier();
    if (below == 0u) hist[d * 16 + w] = pre + (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {
    const uint32_t a = hist[2 * tid], b = hist[2 * tid + 1];
    uint32_t tot;
    const uint32_t ex = block_excl_scan(a + b, wsum, &tot);
    hist[2 * tid] = ex;
    hist[2 * tid + 1] = ex + a;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < E; ++s) {
    const uint32_t d = (key[s] >> shift) & 127u;
    const uint32_t p = hist[d * 16 + w] + rank[s];
    xk[p] = key[s];
    xp[p] = (uint16_t)pay[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < E; ++s) {
    const uint32_t p = fpos<E>(s);
    key[s] = xk[p];
    pay[s] = xp[p];
  }
  (void)lane;
}

// block-wide minimum / maximum of three ints (red = 6 * 16 LDS ints)
__device__ __forceinline__ void block_minmax3(int (&mn)[3], int (&mx)[3], int* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64));
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[a * 16 + w] = mn[a]; red[(3 + a) * 16 + w] = mx[a]; }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int lo = red[a * 16], hi = red[(3 + a) * 16];
#pragma unroll
    for (int j = 1; j < 16; ++j) { lo = min(lo, red[a * 16 + j]); hi = max(hi, red[(3 + a) * 16 + j]); }
    mn[a] = lo;
    mx[a] = hi;
  }
  __syncthreads();
}
__device__ __forceinline__ int bits_for(int range) { return range <= 0 ? 1 : 32 - __clz(range); }

// ProjectInliers(SACMODEL_PLANE) of one point (cluster_marking.cpp:54-64; k_mk_proj_keys)
__device__ __forceinline__ void project_on_base_plane(const MarkParams& k, const float4 p, float* qx, float* qy, float* qz) {
  float m0 = k.mc[0], m1 = k.mc[1], m2 = k.mc[2], m3 = 0.0f;
  const float nrm = sqrtf((m0 * m0 + m2 * m2) + (m1 * m1 + m3 * m3));
  m0 = m0 / nrm; m1 = m1 / nrm; m2 = m2 / nrm;
  const float dist = (m0 * p.x + m2 * p.z) + (m1 * p.y + k.mc[3] * 1.0f);
  *qx = p.x - m0 * dist; *qy = p.y - m1 * dist; *qz = p.z - m2 * dist;
}

// ---------------------------------------------------------------------------------------------
// launch 1, block 0: uniform grid of the observation built by one workgroup (<= 16384 points, <= 32768 cells)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fuse_build_grid(PointGrid g, const float4* __restrict__ pts, uint32_t* __restrict__ parent,
                                                uint32_t* wsum) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t n = g.n, cells1 = (uint32_t)(g.nx * g.ny * g.nz) + 1u;
  for (uint32_t j = tid; j < cells1; j += kFuseThreads) g.cell_start[j] = 0u;
  __threadfence();
  __syncthreads();
  uint32_t cell[16], rk[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const uint32_t i = (uint32_t)s * kFuseThreads + tid;
    cell[s] = 0u; rk[s] = 0u;
    if (i < n) {
      const float4 p = pts[i];
      cell[s] = (uint32_t)((grid_cz(g, p.z) * g.ny + grid_cy(g, p.y)) * g.nx + grid_cx(g, p.x));
      rk[s] = atomicAdd(&g.cell_start[cell[s]], 1u);
      parent[i] = i;
    }
  }
  __threadfence();
  __syncthreads();
  // in-place exclusive scan: wave w owns the cells [w R, (w + 1) R), 64 consecutive cells per step
  const uint32_t R = (((cells1 + 15u) / 16u) + 63u) & ~63u;
  const uint32_t lo = min((uint32_t)w * R, cells1), hi = min(lo + R, cells1);
  uint32_t acc = 0;
  for (uint32_t j = lo + lane; j < hi; j += 64) acc += ld_agent(&g.cell_start[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, 64);
  if (lane == 0) wsum[w] = acc;
  __syncthreads();
  uint32_t carry = 0;
  for (int j = 0; j < w; ++j) carry += wsum[j];
  for (uint32_t j0 = lo; j0 < hi; j0 += 64) {        // (wave-uniform trip count)
    const uint32_t j = j0 + lane;
    const uint32_t v = j < hi ? ld_agent(&g.cell_start[j]) : 0u;
    const uint32_t incl = wave_incl_scan_u32(v);
    if (j < hi) g.cell_start[j] = carry + incl - v;
    carry += wave_last(incl);
  }
  __threadfence();
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const uint32_t i = (uint32_t)s * kFuseThreads + tid;
    if (i < n) {
      const float4 p = pts[i];
      g.sorted[ld_agent(&g.cell_start[cell[s]]) + rk[s]] = make_float4(p.x, p.y, p.z, __int_as_float((int)i));
    }
  }
}

__global__ __launch_bounds__(kFuseThreads) void k_mkf_pre(MarkParams k, MarkStore s, PointGrid obs, const float4* __restrict__ pts,
                                                          uint32_t* __restrict__ parent, MarkCounters* __restrict__ cnt) {
  __shared__ uint32_t wsum[16];
  if (blockIdx.x == 0) {
    if (k.n_obs > 5u) fuse_build_grid(obs, pts, parent, wsum);
    return;
  }
  // window + field-of-view test of every stored marking (k_mk_fov), one lane each
  const uint32_t w = (blockIdx.x - 1u) * kFuseThreads + threadIdx.x;
  uint32_t flag = 0, inwin = 0;
  if (w < k.n_alive_prev) {
    const uint32_t slot = s.alive_list[w];
    if (s.alive[slot]) {
      int x, y, z;
      voxel_unkey(s.keys[slot], &x, &y, &z);
      if (!(x < k.wx0 || x >= k.wx1 || y < k.wy0 || y >= k.wy1 || z < k.wz0 || z >= k.wz1)) {
        inwin = 1;
        const float px = (float)(x * k.res), py = (float)(y * k.res), pz = (float)(z * k.hres);
        flag = in_lidar_observation(k, px, py, pz) ? 1u : 0u;
      }
    }
    s.fov_flag[w] = flag;
  }
  const unsigned long long b = __ballot(inwin != 0u);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt->n_in_window, (uint32_t)__popcll(b));
}

// ---------------------------------------------------------------------------------------------
// launch 2: selfClear (a wave per marking in view)  |  Euclidean clustering, four lanes per point
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cc_union_pair(uint32_t* parent, uint32_t i, uint32_t j) {
  uint32_t u = cc_find(parent, i), v = cc_find(parent, j);
  while (u != v) {
    if (u < v) { const uint32_t t = u; u = v; v = t; }          // u is the larger root
    const uint32_t old = atomicCAS(&parent[u], u, v);
    if (old == u) break;
    u = cc_find(parent, old);
    v = cc_find(parent, v);
  }
}

__global__ __launch_bounds__(256) void k_mkf_clear_cc(MarkParams k, MarkStore s, PointGrid prev, PointGrid obs,
                                                      const float4* __restrict__ pts, uint32_t* parent, uint2* __restrict__ removed_on,
                                                      MarkCounters* __restrict__ cnt, uint32_t nb_clear) {
  if (blockIdx.x < nb_clear) {
    mk_clear_wave(k, s, prev, cnt, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, removed_on);
    return;
  }
  // pcl::extractEuclideanClusters as connected components (k_mk_cc_union); the (z, y) rows of a point's tolerance box
  // are dealt to four lanes
  const uint32_t t = (blockIdx.x - nb_clear) * 256 + threadIdx.x;
  const uint32_t i = t >> 2;
  const int sub = (int)(t & 3u);
  if (i >= k.n_obs) return;
  const float4 p = pts[i];
  const float r = k.tol + 1e-4f;
  const int x0 = grid_cx(obs, p.x - r), x1 = grid_cx(obs, p.x + r);
  const int y0 = grid_cy(obs, p.y - r), y1 = grid_cy(obs, p.y + r);
  const int z0 = grid_cz(obs, p.z - r), z1 = grid_cz(obs, p.z + r);
  const int nys = y1 - y0 + 1, nrows = nys * (z1 - z0 + 1);
  for (int rr = sub; rr < nrows; rr += 4) {
    const int cz = z0 + rr / nys, cy = y0 + rr % nys;
    const uint32_t b = obs.cell_start[(cz * obs.ny + cy) * obs.nx + x0], e = obs.cell_start[(cz * obs.ny + cy) * obs.nx + x1 + 1];
    for (uint32_t q = b; q < e; ++q) {
      const float4 o = obs.sorted[q];
      const uint32_t j = (uint32_t)__float_as_int(o.w);
      if (j < i && l2_simple(o.x, o.y, o.z, p.x, p.y, p.z) < k.tol2) cc_union_pair(parent, i, j);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ground search of one ball by one wave: the (z, y) rows of the ball's box are looked up by one lane each, their
// runs of cell-sorted points flattened over the 64 lanes (k_mk_dgraph / k_mk_unmark walk row after row: two dependent
// round trips per row)
// ---------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void ground_ball_wave(const PointGrid& g, float qx, float qy, float qz, float r, int lane, F&& f) {
  const int x0 = grid_cx(g, qx - r), x1 = grid_cx(g, qx + r);
  const int y0 = grid_cy(g, qy - r), y1 = grid_cy(g, qy + r);
  const int z0 = grid_cz(g, qz - r), z1 = grid_cz(g, qz + r);
  const int nys = y1 - y0 + 1, nrows = nys * (z1 - z0 + 1);
  for (int r0 = 0; r0 < nrows; r0 += 64) {
    const int rr = r0 + lane;
    uint32_t b = 0, len = 0;
    if (rr < nrows) {
      const int row = ((z0 + rr / nys) * g.ny + (y0 + rr % nys)) * g.nx;
      b = g.cell_start[row + x0];
      len = g.cell_start[row + x1 + 1] - b;
    }
    const uint32_t incl = wave_incl_scan_u32(len);
    const uint32_t excl = incl - len, total = wave_last(incl);
    const int nr = min(64, nrows - r0);
    for (uint32_t q0 = 0; q0 < total; q0 += 64) {
      const uint32_t q = q0 + (uint32_t)lane;
      uint32_t src = 0;
      for (int j = 0; j < nr; ++j) {
        const uint32_t ej = (uint32_t)__builtin_amdgcn_readlane((int)excl, j), lj = (uint32_t)__builtin_amdgcn_readlane((int)len, j);
        const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)b, j);
        if (q >= ej && q < ej + lj) src = bj + (q - ej);
      }
      if (q < total) f(g.sorted[src]);
    }
  }
}

// Marking::removePCPtr of one removed marking (k_mk_unmark), generator points recorded by the clearing wave
__device__ __forceinline__ void fuse_unmark_wave(const MarkParams& k, const MarkStore& s, const PointGrid& ground, const uint2 on, int lane) {
  const float r = (float)k.inflation, r2 = static_cast<float>(k.inflation * k.inflation);
  for (uint32_t i = 0; i < on.y; ++i) {
    const float4 p = s.pool[on.x + i];
    ground_ball_wave(ground, p.x, p.y, p.z, r + 1e-4f, lane, [&](const float4 g) {
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
// launch 3, block 0: everything between the union-find and the dGraph update, one workgroup
// ---------------------------------------------------------------------------------------------
template <int E>
constexpr size_t fuse_groups_lds_bytes() {
  return (size_t)1024 * E * (4 + 2 + 2 + 1) + 2048 * 4 + 16 * 4 + 96 * 4;
}

template <int E>
__device__ __forceinline__ void fuse_groups(const MarkParams& k, const FuseBufs& fb, ClusterArrays c, MarkStore s, const PointGrid& ground,
                            const PointGrid& map, uint32_t n_map, MarkCounters* __restrict__ cnt, unsigned char* lds) {
  constexpr uint32_t N = 1024u * E;
  uint32_t* xk = reinterpret_cast<uint32_t*>(lds);              // [N] key exchange / voxel key by position
  uint16_t* xp = reinterpret_cast<uint16_t*>(xk + N);           // [N] payload exchange = the sorted sequence
  uint16_t* cid = xp + N;                                       // [N] cluster of a point, later of a 0.2 m voxel
  uint8_t* state = reinterpret_cast<uint8_t*>(cid + N);         // [N] per cluster
  uint32_t* hist = reinterpret_cast<uint32_t*>(state + N);      // [2048]
  uint32_t* wsum = hist + 2048;                                 // [16]
  int* red = reinterpret_cast<int*>(wsum + 16);                 // [96]
  uint16_t* rootcid = reinterpret_cast<uint16_t*>(xk);          // [N]     alias, G0 only
  uint16_t* start = reinterpret_cast<uint16_t*>(xk);            // [N + 1] alias, G1..G2

  const int tid = threadIdx.x;
  const uint32_t n = k.n_obs;
  uint32_t key[E], pay[E];

  // ---- G0: roots -> cluster ids in seed order (root = smallest point index = PCL's seed) ----
  uint32_t nc;
  {
    uint32_t root[E], ex[E], isr = 0u;
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t i = fpos<E>(s_);
      root[s_] = i < n ? cc_find(fb.parent, i) : 0xFFFFFFFFu;
      if (i < n && root[s_] == i) isr |= 1u << s_;
    }
    pos_excl_scan<E>(isr, ex, wsum, &nc);
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_)
      if ((isr >> s_) & 1u) rootcid[fpos<E>(s_)] = (uint16_t)ex[s_];
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t i = fpos<E>(s_);
      key[s_] = 0xFFFFFFFFu; pay[s_] = 0u;
      if (i < n) {
        const uint32_t cv = rootcid[root[s_]];
        cid[i] = (uint16_t)cv;
        key[s_] = cv; pay[s_] = i;
      }
    }
    __syncthreads();
  }
  // ---- G1: sort 1 = clusters in seed order, their points in index order (stable by cluster id) ----
  radix_pass<E>(key, pay, 0, hist, xk, xp, wsum);
  if (nc > 128u) radix_pass<E>(key, pay, 7, hist, xk, xp, wsum);
  __syncthreads();                                               // xk is free: cluster starts
#pragma unroll
  for (int s_ = 0; s_ < E; ++s_) {
    const uint32_t m = fpos<E>(s_);
    if (m < n) {
      const uint32_t prev = m ? (uint32_t)cid[xp[m - 1]] : 0xFFFFFFFFu;
      if (key[s_] != prev) start[key[s_]] = (uint16_t)m;
      fb.spts[m] = fb.pts[pay[s_]];
    }
  }
  if (tid == 0) start[nc] = (uint16_t)n;
  __threadfence();
  __syncthreads();
  // ---- G2: per cluster centroid (floats added in index order, / size), min size, "centre attached to the ground" ----
  {
    uint32_t kept = 0;
    for (uint32_t ci = tid; ci < nc; ci += kFuseThreads) {
      const uint32_t b = start[ci], e = start[ci + 1];
      float cx = 0.f, cy = 0.f, cz = 0.f;
      uint32_t m = b;
      for (; m + 4 <= e; m += 4) {
        const float4 p0 = fb.spts[m], p1 = fb.spts[m + 1], p2 = fb.spts[m + 2], p3 = fb.spts[m + 3];
        cx += p0.x; cy += p0.y; cz += p0.z;
        cx += p1.x; cy += p1.y; cz += p1.z;
        cx += p2.x; cy += p2.y; cz += p2.z;
        cx += p3.x; cy += p3.y; cz += p3.z;
      }
      for (; m < e; ++m) {
        const float4 p = fb.spts[m];
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
      if (ok) ++kept;
      if (ok && grid_radius_count(ground, cx, cy, cz, 0.05f + 1e-4f, static_cast<float>(0.05 * 0.05), 1) > 0) ok = false;
      state[ci] = ok ? 1u : 0u;
    }
    uint32_t tot;
    (void)block_excl_scan(kept, wsum, &tot);
    if (tid == 0) { cnt->n_clusters = nc; cnt->n_clusters_kept = tot; }
    __threadfence();
    __syncthreads();
  }
  // ---- G3: sort 2 = (cluster, 0.2 m voxel z|y|x, point index): voxel order first, then stable by cluster ----
  int bx, by, bz, mn[3], mx[3];
  {
    const float inv = 1.0f / 0.2f;
    mn[0] = mn[1] = mn[2] = 0x7FFFFFFF; mx[0] = mx[1] = mx[2] = (int)0x80000000;
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t i = fpos<E>(s_);
      if (i < n) {
        const float4 p = fb.pts[i];
        const int vx = (int)floorf(p.x * inv), vy = (int)floorf(p.y * inv), vz = (int)floorf(p.z * inv);
        mn[0] = min(mn[0], vx); mx[0] = max(mx[0], vx);
        mn[1] = min(mn[1], vy); mx[1] = max(mx[1], vy);
        mn[2] = min(mn[2], vz); mx[2] = max(mx[2], vz);
      }
    }
    block_minmax3(mn, mx, red);
    // (differences in 64 bits: points may be anywhere; a range that does not fit falls back to the general route)
    const long long rx = (long long)mx[0] - mn[0], ry = (long long)mx[1] - mn[1], rz = (long long)mx[2] - mn[2];
    if (rx >= (1ll << 27) || ry >= (1ll << 27) || rz >= (1ll << 27)) { if (tid == 0) cnt->fallback = 1u; return; }
    bx = bits_for((int)rx); by = bits_for((int)ry); bz = bits_for((int)rz);
    if (bx + by + bz > kFuseKeyBits) { if (tid == 0) cnt->fallback = 1u; return; }
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t i = fpos<E>(s_);
      key[s_] = 0xFFFFFFFFu; pay[s_] = 0u;
      if (i < n) {
        const float4 p = fb.pts[i];
        key[s_] = ((uint32_t)((int)floorf(p.z * inv) - mn[2]) << (bx + by)) | ((uint32_t)((int)floorf(p.y * inv) - mn[1]) << bx) |
                  (uint32_t)((int)floorf(p.x * inv) - mn[0]);
        pay[s_] = i;
      }
    }
  }
  for (int sh = 0; sh < bx + by + bz; sh += 7) radix_pass<E>(key, pay, sh, hist, xk, xp, wsum);
#pragma unroll
  for (int s_ = 0; s_ < E; ++s_) key[s_] = key[s_] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)cid[pay[s_]];
  radix_pass<E>(key, pay, 0, hist, xk, xp, wsum);
  if (nc > 128u) radix_pass<E>(key, pay, 7, hist, xk, xp, wsum);
  __syncthreads();
  // ---- G4: 0.2 m VoxelGrid of every cluster that passed G2: one lane per voxel adds its points in order ----
  uint32_t ng2;
  {
    const float inv = 1.0f / 0.2f;
    uint32_t gex[E], fl = 0u;
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t j = fpos<E>(s_);
      uint32_t vox = 0xFFFFFFFFu;
      if (j < n) {
        const float4 p = fb.pts[pay[s_]];
        vox = ((uint32_t)((int)floorf(p.z * inv) - mn[2]) << (bx + by)) | ((uint32_t)((int)floorf(p.y * inv) - mn[1]) << bx) |
              (uint32_t)((int)floorf(p.x * inv) - mn[0]);
      }
      xk[j] = vox;
    }
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t j = fpos<E>(s_);
      if (j < n && state[key[s_]] != 0u) {
        const bool first = j == 0u || (uint32_t)cid[xp[j - 1]] != key[s_] || xk[j - 1] != xk[j];
        if (first) fl |= 1u << s_;
      }
    }
    pos_excl_scan<E>(fl, gex, wsum, &ng2);
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      if (!((fl >> s_) & 1u)) continue;
      const uint32_t j = fpos<E>(s_), cj = key[s_], vox = xk[j];
      float sx = 0.f, sy = 0.f, sz = 0.f;
      uint32_t e = j;
      for (; e < n && xk[e] == vox && (uint32_t)cid[xp[e]] == cj; ++e) {
        const float4 p = fb.pts[xp[e]];
        sx += p.x; sy += p.y; sz += p.z;
      }
      const float cntf = (float)(e - j);
      fb.ds[gex[s_]] = make_float4(sx / cntf, sy / cntf, sz / cntf, __int_as_float((int)cj));
      atomicAdd(&c.ds_count[cj], 1u);
    }
    if (tid == 0) cnt->n_groups2 = ng2;
    __threadfence();
    __syncthreads();
  }
  // ---- G5: "is it part of the static map", voxel key, in the sensor's view (k_mk_cluster_stage2) ----
  for (uint32_t ci = tid; ci < nc; ci += kFuseThreads) {
    if (state[ci] != 1u) continue;
    const float4 cen = c.centroid[ci];
    const size_t nds = ld_agent(&c.ds_count[ci]);
    size_t hit = 0;
    if (k.ignore_ratio <= 0.999) {
      const bool near = n_map > 0 && grid_radius_count(map, cen.x, cen.y, cen.z, 0.1f + 1e-4f, static_cast<float>(0.1 * 0.1), 1) > 0;
      if (near)
        for (size_t a = 0; a < nds; ++a) {
          hit++;
          if (hit > nds * k.ignore_ratio) break;
        }
    }
    if (!(hit <= nds * k.ignore_ratio)) { state[ci] = 0u; continue; }
    const int vx = (int)(cen.x / k.res), vy = (int)(cen.y / k.res), vz = (int)(cen.z / k.hres);
    c.vkey[3 * ci + 0] = vx; c.vkey[3 * ci + 1] = vy; c.vkey[3 * ci + 2] = vz;
    const float px = (float)(vx * k.res), py = (float)(vy * k.res), pz = (float)(vz * k.hres);
    state[ci] = in_lidar_observation(k, px, py, pz) ? 2u : 0u;
  }
  __syncthreads();
  // ---- G6: sort 3 = (cluster, 0.1 m voxel of the projected voxel centroid, order of G4) over accepted clusters ----
  {
    const float inv = 1.0f / 0.1f;
    mn[0] = mn[1] = mn[2] = 0x7FFFFFFF; mx[0] = mx[1] = mx[2] = (int)0x80000000;
    uint32_t on = 0u;
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t g = fpos<E>(s_);
      if (g < ng2) {
        const float4 d = fb.ds[g];
        const uint32_t ci = (uint32_t)__float_as_int(d.w);
        cid[g] = (uint16_t)ci;
        if (state[ci] == 2u) {
          float qx, qy, qz;
          project_on_base_plane(k, d, &qx, &qy, &qz);
          on |= 1u << s_;
          const int vx = (int)floorf(qx * inv), vy = (int)floorf(qy * inv), vz = (int)floorf(qz * inv);
          mn[0] = min(mn[0], vx); mx[0] = max(mx[0], vx);
          mn[1] = min(mn[1], vy); mx[1] = max(mx[1], vy);
          mn[2] = min(mn[2], vz); mx[2] = max(mx[2], vz);
        }
      }
    }
    block_minmax3(mn, mx, red);
    bx = by = bz = 1;
    if (mn[0] <= mx[0]) {
      const long long rx = (long long)mx[0] - mn[0], ry = (long long)mx[1] - mn[1], rz = (long long)mx[2] - mn[2];
      if (rx >= (1ll << 27) || ry >= (1ll << 27) || rz >= (1ll << 27)) { if (tid == 0) cnt->fallback = 1u; return; }
      bx = bits_for((int)rx); by = bits_for((int)ry); bz = bits_for((int)rz);
      if (bx + by + bz > kFuseKeyBits) { if (tid == 0) cnt->fallback = 1u; return; }
    }
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      key[s_] = 0xFFFFFFFFu;
      pay[s_] = fpos<E>(s_) & (N - 1u);
      if ((on >> s_) & 1u) {
        float qx, qy, qz;
        project_on_base_plane(k, fb.ds[pay[s_]], &qx, &qy, &qz);
        key[s_] = ((uint32_t)((int)floorf(qz * inv) - mn[2]) << (bx + by)) | ((uint32_t)((int)floorf(qy * inv) - mn[1]) << bx) |
                  (uint32_t)((int)floorf(qx * inv) - mn[0]);
      }
    }
  }
  for (int sh = 0; sh < bx + by + bz; sh += 7) radix_pass<E>(key, pay, sh, hist, xk, xp, wsum);
#pragma unroll
  for (int s_ = 0; s_ < E; ++s_) key[s_] = key[s_] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)cid[pay[s_]];
  radix_pass<E>(key, pay, 0, hist, xk, xp, wsum);
  if (nc > 128u) radix_pass<E>(key, pay, 7, hist, xk, xp, wsum);
  __syncthreads();
  // ---- G7: 0.1 m VoxelGrid of the projected points -> generator points ----
  {
    const float inv = 1.0f / 0.1f;
    uint32_t gex[E], fl = 0u, ng3;
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t j = fpos<E>(s_);
      uint32_t vox = 0xFFFFFFFFu;
      if (key[s_] != 0xFFFFFFFFu) {
        float qx, qy, qz;
        project_on_base_plane(k, fb.ds[pay[s_]], &qx, &qy, &qz);
        vox = ((uint32_t)((int)floorf(qz * inv) - mn[2]) << (bx + by)) | ((uint32_t)((int)floorf(qy * inv) - mn[1]) << bx) |
              (uint32_t)((int)floorf(qx * inv) - mn[0]);
      }
      xk[j] = vox;
    }
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      const uint32_t j = fpos<E>(s_);
      if (key[s_] != 0xFFFFFFFFu) {
        const bool first = j == 0u || (uint32_t)cid[xp[j - 1]] != key[s_] || xk[j - 1] != xk[j];
        if (first) fl |= 1u << s_;
      }
    }
    pos_excl_scan<E>(fl, gex, wsum, &ng3);
#pragma unroll
    for (int s_ = 0; s_ < E; ++s_) {
      if (!((fl >> s_) & 1u)) continue;
      const uint32_t j = fpos<E>(s_), cj = key[s_], vox = xk[j];
      float sx = 0.f, sy = 0.f, sz = 0.f;
      uint32_t e = j;
      for (; e < N && xk[e] == vox && (uint32_t)cid[xp[e]] == cj; ++e) {     // (sentinels hold vox = ~0: never equal)
        float qx, qy, qz;
        project_on_base_plane(k, fb.ds[xp[e]], &qx, &qy, &qz);
        sx += qx; sy += qy; sz += qz;
      }
      const float cntf = (float)(e - j);
      const uint32_t h = gex[s_];
      fb.gen[h] = make_float4(sx / cntf, sy / cntf, sz / cntf, __int_as_float((int)cj));
      atomicAdd(&c.gen_count[cj], 1u);
      atomicMin(&c.gen_first[cj], h);
    }
    if (tid == 0) cnt->n_groups3 = ng3;
    __threadfence();
    __syncthreads();
  }
  // ---- G8: Marking::addPCPtr: slot of every accepted cluster (k_mk_slots), then the keeper's pool range (k_mk_commit) ----
  {
    uint32_t my_slot[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t ci = (uint32_t)q * kFuseThreads + tid;
      my_slot[q] = 0xFFFFFFFFu;
      if (ci >= nc) continue;
      fb.pool_ofs[ci] = 0xFFFFFFFFu;
      c.state[ci] = state[ci];
      if (state[ci] != 2u) continue;
      const unsigned long long vk = voxel_key(c.vkey[3 * ci], c.vkey[3 * ci + 1], c.vkey[3 * ci + 2]);
      uint32_t slot = mk_hash(vk) & k.table_mask;
      bool found = false;
      for (uint32_t probe = 0; probe <= k.table_mask; ++probe) {
        const unsigned long long prev = atomicCAS(&s.keys[slot], 0ull, vk);
        if (prev == 0ull) atomicAdd(&cnt->n_new_keys, 1u);
        if (prev == 0ull || prev == vk) { found = true; break; }
        slot = (slot + 1) & k.table_mask;
      }
      if (!found) { atomicOr(&cnt->overflow, 1u); state[ci] = 3u; c.state[ci] = 3u; continue; }
      c.slot[ci] = slot;
      my_slot[q] = slot;
      const unsigned long long pr = ((unsigned long long)((1u << 20) - min(c.size[ci], (1u << 20) - 1u)) << 20) | (unsigned long long)(ci + 1u);
      if (atomicMax(&s.owner[slot], pr) != 0ull) atomicAdd(&cnt->n_dup, 1u);
      atomicAdd(&cnt->n_marked, 1u);
    }
    __threadfence();
    __syncthreads();
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t ci = (uint32_t)q * kFuseThreads + tid, slot = my_slot[q];
      if (slot == 0xFFFFFFFFu) continue;
      const unsigned long long pr = ((unsigned long long)((1u << 20) - min(c.size[ci], (1u << 20) - 1u)) << 20) | (unsigned long long)(ci + 1u);
      if (ld_agent(&s.owner[slot]) != pr) continue;
      const uint32_t ng = ld_agent(&c.gen_count[ci]);
      const uint32_t ofs = atomicAdd(&cnt->pool_used, ng);
      if (ofs + ng > k.pool_cap) { atomicOr(&cnt->overflow, 2u); s.alive[slot] = 0; s.pts_n[slot] = 0; continue; }
      fb.pool_ofs[ci] = ofs;
      s.pts_ofs[slot] = ofs;
      s.pts_n[slot] = ng;
      s.alive[slot] = 1;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < E; ++q)
      if (my_slot[q] != 0xFFFFFFFFu) s.owner[my_slot[q]] = 0ull;       // next update starts with no owners
  }
}

template <int E>
__global__ __launch_bounds__(kFuseThreads) void k_mkf_unmark_groups(MarkParams k, FuseBufs fb, ClusterArrays c, MarkStore s, PointGrid ground,
                                                                    PointGrid map, uint32_t n_map, MarkCounters* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fuse_lds[];
  if (blockIdx.x == 0) {
    if (k.n_obs > 5u) fuse_groups<E>(k, fb, c, s, ground, map, n_map, cnt, fuse_lds);
    return;
  }
  const int lane = threadIdx.x & 63;
  const uint32_t stride = (gridDim.x - 1u) * 16u, n_removed = cnt->n_removed;
  for (uint32_t r = (blockIdx.x - 1u) * 16u + (threadIdx.x >> 6); r < n_removed; r += stride)
    fuse_unmark_wave(k, s, ground, fb.removed_on[r], lane);
}

// ---------------------------------------------------------------------------------------------
// launch 4: dGraph / lethal update of the new generator points (k_mk_dgraph)  |  alive list (k_mk_finish); the last
// block to finish publishes the counters and leaves them zeroed for the next update
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mkf_dgraph_finish(MarkParams k, FuseBufs fb, ClusterArrays c, MarkStore s, PointGrid ground,
                                                           MarkCounters* __restrict__ cnt, uint32_t nb_dg) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x < nb_dg) {
    const uint32_t n_gen = cnt->n_groups3;
    const float r = (float)k.inflation, r2 = static_cast<float>(k.inflation * k.inflation);
    for (uint32_t h = blockIdx.x * 4 + (threadIdx.x >> 6); h < n_gen; h += nb_dg * 4) {
      const float4 p = fb.gen[h];
      const uint32_t ci = (uint32_t)__float_as_int(p.w);
      const uint32_t po = fb.pool_ofs[ci];
      if (lane == 0 && po != 0xFFFFFFFFu) s.pool[po + (h - c.gen_first[ci])] = make_float4(p.x, p.y, p.z, 0.f);
      ground_ball_wave(ground, p.x, p.y, p.z, r + 1e-4f, lane, [&](const float4 g) {
        if (l2_simple(g.x, g.y, g.z, p.x, p.y, p.z) < r2) {
          const int node = __float_as_int(g.w);
          const float dx = p.x - g.x, dy = p.y - g.y;
          const float d = sqrtf(dx * dx + dy * dy);                    // z dropped on purpose (cluster_marking.cpp:86-88)
          atomicMin(reinterpret_cast<unsigned long long*>(s.dgraph) + node, (unsigned long long)__double_as_longlong((double)d));
          if (d <= k.inscribed) s.lethal[node] = 1;
        }
      });
    }
  } else {
    __shared__ uint32_t base;
    const uint32_t slot = (blockIdx.x - nb_dg) * 256 + threadIdx.x;
    const bool al = slot <= k.table_mask && s.alive[slot] != 0u;
    const unsigned long long b = __ballot(al);
    const int w = threadIdx.x >> 6;
    __shared__ uint32_t wc[4];
    if (lane == 0) wc[w] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = wc[0] + wc[1] + wc[2] + wc[3];
      base = tot ? atomicAdd(&cnt->n_alive, tot) : 0u;
    }
    __syncthreads();
    if (al) {
      uint32_t o = base;
      for (int j = 0; j < w; ++j) o += wc[j];
      o += __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
      s.alive_list[o] = slot;
    }
  }
  // ---- last block out: counters -> host-mapped record, device copy zeroed (the pool fill carries over) ----
  __shared__ uint32_t last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&fb.ticket[0], 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  constexpr int kWords = (int)(sizeof(MarkCounters) / sizeof(uint32_t));
  uint32_t* src = reinterpret_cast<uint32_t*>(cnt);
  uint32_t* dst = reinterpret_cast<uint32_t*>(fb.host_out);
  if ((int)threadIdx.x < kWords) {
    const uint32_t v = ld_agent(&src[threadIdx.x]);
    __hip_atomic_store(&dst[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (&src[threadIdx.x] != &cnt->pool_used) src[threadIdx.x] = 0u;
  }
  if (threadIdx.x == 0) fb.ticket[0] = 0u;
}

}  // namespace dddmr
