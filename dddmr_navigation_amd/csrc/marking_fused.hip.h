// marking_fused.hip.h -- the marking / clearing update for observations of up to 32768 points in FIVE launches.
//
// The general route (marking.hip.h + rocPRIM sorts, any observation size) issues ~60 launches per update, ~30 of them
// at the 4.5-4.9 us launch floor, plus three mid-update copies: for the observation of a 16-line LiDAR the update is
// launch-bound (profiles/r02_C5M_kernel_stats.csv).  Here the chip-wide steps share launches, and everything between
// the union-find and the dGraph update -- three stable sorts with their flag / scan / reduce rounds, the per-cluster
// tests, the hash insert -- runs in 64 independent 256-lane workgroups, each on the clusters whose seed point hashes
// to it, with points, sort keys and payloads resident in LDS:
//
//   grid launch       k_mkf_grid_fov        first blocks: the new observation's uniform grid, built in LDS from the points
//                                           alone (a quarter of the cells per workgroup)  |  every store slot: window +
//                                           field-of-view test -> list of ray tests
//   clear launch      k_mkf_clear_cc        selfClear ray tests (a wave per listed marking)  |  Euclidean clustering
//                                           (union-find)
//   seed launch       k_mkf_roots_unmark    every point's cluster seed (root of the union-find)
//   partition launch  k_mkf_groups          64 blocks, partition p = clusters with hash(seed) = p: centroids -> 0.2 m
//                                           VoxelGrid -> static / FOV tests -> projection + 0.1 m VoxelGrid -> store
//                                           slots  |  on the idle CUs: removePCPtr of the markings the clear launch
//                                           removed, ground node by ground node
//   commit launch     k_mkf_commit_dgraph   keeper of every claimed voxel -> pool  |  dGraph / lethal update, ground node
//                                           by ground node; the last block publishes the counters to host-mapped memory
//   (DDDMR_MKF_GRID=global: a sixth launch, k_mkf_count, counts the cells with global atomics first;
//    DDDMR_MKF_UNMARK=roots: removePCPtr in the seed launch)
//
// (A first version ran the grouping chain in ONE 1024-lane workgroup with the keys in registers: 670 us at 10.5 k
// points -- a wave64 VALU instruction occupies its SIMD for four cycles and one CU is 1/256 of the chip; the phase
// stamps of tools/marking_stamps.py showed every step, not only the sorts, waiting on that one CU.)
//
// What each step computes, and the reference lines it follows, is unchanged from marking.hip.h: the same float
// operations in the same order (cluster / voxel centroids are sequential float sums in point-index order, which is
// why the sorts must be stable), so both routes give bit-identical stores, dGraphs and lethal sets.  A cluster is
// named by its seed's point index here (marking.hip.h: by its rank among the seeds): the same order, which is all
// the contested-voxel priority looks at.
#pragma once
#include "marking.hip.h"

#pragma clang fp contract(off)

namespace dddmr {

constexpr uint32_t kFuseMaxObs = 32768;      // points of an observation the fused route takes (a 32-line lidar after the 0.1 m feed)
constexpr uint32_t kFuseRegObs = 16384;      // ... of which the grid builder keeps (cell, rank) in registers; beyond: parked in global memory
constexpr uint32_t kFuseMaxCells = 65536;    // cells of the observation grid (16-bit counters, two per LDS word: 132 KB)
constexpr int kFuseParts = 64;               // partitions of the clusters (by hash of the seed point index)
constexpr uint32_t kPartCap = 4096;          // points one partition workgroup takes (16 per lane)
constexpr int kPartThreads = 256;
constexpr int kFuseKeyBits = 28;             // voxel sort keys: 4 passes of 7 bits at most

#ifdef DDDMR_PHASE_STAMPS
// diagnostic build only: s_memtime at the phase boundaries of partition 0 and of the grid block (tools/marking_stamps.py)
__device__ unsigned long long g_mk_stamps[192];   // [0,64) phase stamps of block 0; [64,192) per partition: points << 40 | ticks
#define MKF_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_mk_stamps[i] = (unsigned long long)clock64(); } while (0)
#else
#define MKF_STAMP(i) do { } while (0)
#endif


constexpr uint32_t kBandMax = 128;      // y rows of ground cells the window's range may span (else: point by point)
constexpr uint32_t kBandCap = kFuseMaxObs;   // points one band takes (an update makes at most one generator point per observation point)

struct SplatRange {
  int cx0, cx1, cy0, cy1;     // ground-grid cells of the window (+ inflation radius), every z row
  uint32_t rows;              // (cy1 - cy0 + 1) * nz
  uint32_t segs;              // 64-node segments per row (from the longest row of the ground grid)
  int delta;                  // a node of row cy can only be within the radius of points of the rows cy - delta .. cy + delta
  uint32_t bands;             // cy1 - cy0 + 1, or 0: banding off (the range spans more than kBandMax rows)
};
struct BandList {             // points bucketed by the y row of ground cells they fall in (relative to cy0)
  float4* pts;                // [kBandMax * kBandCap]
  uint32_t* cnt;              // [kBandMax], all zero between updates
};
__device__ __forceinline__ bool ball_in_range(const PointGrid& g, const SplatRange& rg, float qx, float qy, float r) {
  return grid_cx(g, qx, -r) >= rg.cx0 && grid_cx(g, qx, r) <= rg.cx1 && grid_cy(g, qy, -r) >= rg.cy0 && grid_cy(g, qy, r) <= rg.cy1;
}
// true: the point went into its band (the node-by-node pass will see it); false: it has to be walked point by point
__device__ __forceinline__ bool band_push(const PointGrid& g, const SplatRange& rg, const BandList& b, const float4 p, const float r) {
  if (!rg.bands || !ball_in_range(g, rg, p.x, p.y, r)) return false;
  const uint32_t band = (uint32_t)(grid_cy(g, p.y) - rg.cy0);
  const uint32_t at = atomicAdd(&b.cnt[band], 1u);
  if (at >= kBandCap) return false;
  b.pts[(size_t)band * kBandCap + at] = p;
  return true;
}

struct FuseBufs {
  const float4* pts;       // this update's observation (global frame)
  uint32_t* parent;        // [n] union-find; after k_mkf_roots: every point's seed
  float4* ds;              // [n] 0.2 m voxel centroids, w = cluster (scratch of a partition between its two halves)
  float4* gen;             // [n] generator points, w = cluster
  uint32_t* clear_list;    // [table] slots inside the window and the sensor's view
  float4* unmark_pts;      // [pool] generator points of removed markings that have to be walked point by point
  BandList gen_bands, unmark_bands;
  SplatRange rg;
  uint32_t* ticket;        // [34] top counter, spare, 32 shards
  uint32_t* cell_count;    // [kFuseMaxCells] all zero between updates
  unsigned long long* slot;  // [n] (rank << 32) | cell of a point in the observation grid
  MarkCounters* host_out;  // host-mapped copy of the update's counters
  uint32_t grid_in_lds;    // the observation grid is built by the grid launch's first workgroups alone (no count launch)
  uint32_t* hi_rank;       // [8][kFuseMaxObs - kFuseRegObs] (cell, rank) of the points past kFuseRegObs, per grid workgroup
};

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// This wave's stores and atomics have been acknowledged (written through to the level every CU of the chip reads:
// device-scope atomics and sc1 loads go there).  What a hand-off inside a launch needs here is exactly that plus a
// barrier / ticket; an agent-scope fence (__threadfence) also writes the XCD's L2 back, which cost the last launch
// 120 us when all 2090 workgroups issued one (profiles/r03_C5M_fused_kernel_stats.csv history in DESIGN.md).
__device__ __forceinline__ void wait_own_memory_ops() { __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t part_of(uint32_t seed) { return (seed * 2654435761u) >> 26; }   // 64 partitions
__device__ __forceinline__ int bits_for(int range) { return range <= 0 ? 1 : 32 - __clz(range); }

// exclusive prefix of one value per lane over a block of W waves, in lane order (two barriers; wsum = W LDS words)
template <int W>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan_u32(v);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const uint32_t x = wsum[j];
    tot += x;
    if (j < w) base += x;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// ProjectInliers(SACMODEL_PLANE) of one point (cluster_marking.cpp:54-64; k_mk_proj_keys)
__device__ __forceinline__ void project_on_base_plane(const MarkParams& k, float x, float y, float z, float* qx, float* qy, float* qz) {
  float m0 = k.mc[0], m1 = k.mc[1], m2 = k.mc[2], m3 = 0.0f;
  const float nrm = sqrtf((m0 * m0 + m2 * m2) + (m1 * m1 + m3 * m3));
  m0 = m0 / nrm; m1 = m1 / nrm; m2 = m2 / nrm;
  const float dist = (m0 * x + m2 * z) + (m1 * y + k.mc[3] * 1.0f);
  *qx = x - m0 * dist; *qy = y - m1 * dist; *qz = z - m2 * dist;
}

// ---------------------------------------------------------------------------------------------
// count + grid launch (DDDMR_MKF_GRID=global): uniform grid of the observation (<= 16384 points, <= 65536 cells).  The blocks of the count launch take 256
// points each into the cells (device-scope atomics: the rank inside the cell comes back) and leave (cell, rank) per point;
// the grid launch's first workgroup scans the counters in LDS (16 bits each, two per word), writes the cell starts, leaves the
// counters zeroed for the next update and scatters the points.  (As the last-ticket block of the count launch the scan had to
// read the counters past its XCD's L2, one sc1 load after the other: 60 us; built in LDS by one workgroup from scratch,
// count included: 41 us.)  The order of the points inside a cell is whatever the atomics made it: no result depends on
// it (radius tests look at every point of a cell).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fuse_grid_count(const PointGrid& g, const float4* __restrict__ pts, uint32_t* __restrict__ parent,
                                                uint32_t* __restrict__ cell_count, unsigned long long* __restrict__ slot) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= g.n) return;
  const float4 p = pts[i];
  const uint32_t cell = (uint32_t)((grid_cz(g, p.z) * g.ny + grid_cy(g, p.y)) * g.nx + grid_cx(g, p.x));
  const uint32_t rank = atomicAdd(&cell_count[cell], 1u);
  slot[i] = ((unsigned long long)rank << 32) | cell;
  parent[i] = i;
}

__device__ __forceinline__ void fuse_grid_scan_scatter(const PointGrid& g, const float4* __restrict__ pts, uint32_t* __restrict__ cell_count,
                                                       const unsigned long long* __restrict__ slot, uint32_t* cnt2 /* [33 * 1024] */,
                                                       uint32_t* wsum) {
  const int tid = threadIdx.x;
  const uint32_t n = g.n, cells = (uint32_t)(g.nx * g.ny * g.nz);
  MKF_STAMP(32);
  // counters -> LDS, two cells per word, word w at w + w / 32 (lane t then owns 32 words at a stride of 33: no bank conflicts);
  // four cells per lane and step (the counter array is padded to a multiple of four)
  uint4* cc4 = reinterpret_cast<uint4*>(cell_count);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t q = (uint32_t)j * 1024u + tid, c0 = 4u * q;           // cells c0 .. c0 + 3 = words 2 q, 2 q + 1
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (c0 < cells) { v = cc4[q]; cc4[q] = make_uint4(0u, 0u, 0u, 0u); }
    const uint32_t w = 2u * q;
    cnt2[w + (w >> 5)] = v.x | (v.y << 16);
    cnt2[w + 1 + ((w + 1) >> 5)] = v.z | (v.w << 16);
  }
  __syncthreads();
  MKF_STAMP(33);
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const uint32_t v = cnt2[tid * 33 + j];
    sum += (v & 0xFFFFu) + (v >> 16);
  }
  uint32_t tot;
  uint32_t run = block_excl_scan<16>(sum, wsum, &tot);
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const uint32_t v = cnt2[tid * 33 + j];
    const uint32_t a = v & 0xFFFFu, b = v >> 16;
    cnt2[tid * 33 + j] = run | ((run + a) << 16);              // (starts <= 16384 fit 16 bits)
    run += a + b;
  }
  __syncthreads();
  uint4* cs4 = reinterpret_cast<uint4*>(g.cell_start);
#pragma unroll
  for (int j = 0; j < 16; ++j) {                               // coalesced copy-out of the starts
    const uint32_t q = (uint32_t)j * 1024u + tid, c0 = 4u * q, w = 2u * q;
    const uint32_t v0 = cnt2[w + (w >> 5)], v1 = cnt2[w + 1 + ((w + 1) >> 5)];
    if (c0 < cells) cs4[q] = make_uint4(v0 & 0xFFFFu, v0 >> 16, v1 & 0xFFFFu, v1 >> 16);   // (entries past `cells` hold n: the scan ran on)
  }
  if (tid == 0) g.cell_start[cells] = n;
  MKF_STAMP(34);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const uint32_t i = (uint32_t)s * 1024u + tid;
    if (i < n) {
      const float4 p = pts[i];
      const unsigned long long sl = slot[i];
      const uint32_t cell = (uint32_t)sl, wi = cell >> 1;
      const uint32_t st = (cnt2[wi + (wi >> 5)] >> ((cell & 1u) * 16u)) & 0xFFFFu;
      g.sorted[st + (uint32_t)(sl >> 32)] = make_float4(p.x, p.y, p.z, __int_as_float((int)i));
    }
  }
  MKF_STAMP(35);
}

// The same grid built from the points alone by `ng` workgroups (1, 2, 4 or 8), workgroup q owning the cells
// [q, q + 1) * 65536 / ng: every workgroup reads ALL points, ranks the ones of its cells by LDS atomics on the packed 16-bit
// counters (the returned old value is the rank) and counts the points of lower cells -- its base, so no workgroup waits for
// another -- then scans, copies out and scatters its own share.  (cell, rank) of a lane's <= 16 points stay in registers
// between the count and the scatter.  No count launch, no 256 KB of global counters to read back and re-zero; the single
// workgroup's 20 us (the launch's critical path: the slot blocks next to it take 9) divide by ng up to the count pass.
// kBig (observations of more than kFuseRegObs points): the points past kFuseRegObs go through the same count in batches of
// four, their (cell, rank) words parked in hi_rank -- a slice per workgroup, so nobody reads what another wrote.
template <bool kBig>
__device__ __forceinline__ void fuse_grid_build_lds(const PointGrid& g, const float4* __restrict__ pts, uint32_t* __restrict__ parent,
                                                    uint32_t* cnt2 /* [33 * 1024] */, uint32_t* wsum, const uint32_t part, const uint32_t ng,
                                                    uint32_t* __restrict__ hi_rank) {
  const int tid = threadIdx.x;
  const uint32_t n = g.n, cells = (uint32_t)(g.nx * g.ny * g.nz);
  const uint32_t wpt = 32u / ng;                               // counter words (two cells each) per lane
  const uint32_t lw = 31u - (uint32_t)__clz((int)wpt);         // log2(wpt): word w lives at w + (w >> lw)
  const uint32_t c_lo = part * (65536u / ng), c_hi = c_lo + 65536u / ng;
  MKF_STAMP(32);
  for (uint32_t j = 0; j <= wpt; ++j) cnt2[j * 1024u + tid] = 0u;
  __syncthreads();
  // all loads first, then all atomics, then the ranks: one latency each instead of sixteen in a row
  float4 pt[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const uint32_t i = (uint32_t)s * 1024u + tid;
    pt[s] = i < n ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  uint32_t pack[16];                                           // (cell - c_lo) | rank << 16; ~0: not mine
  uint32_t lower = 0;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const uint32_t i = (uint32_t)s * 1024u + tid;
    pack[s] = ~0u;
    if (i < n) {
      const uint32_t cell = (uint32_t)((grid_cz(g, pt[s].z) * g.ny + grid_cy(g, pt[s].y)) * g.nx + grid_cx(g, pt[s].x));
      lower += cell < c_lo ? 1u : 0u;
      if (cell >= c_lo && cell < c_hi) {
        const uint32_t lc = cell - c_lo, wi = lc >> 1, sh = (lc & 1u) * 16u;
        pack[s] = atomicAdd(&cnt2[wi + (wi >> lw)], 1u << sh);   // (the old word, for now)
        pt[s].w = __uint_as_float(lc);
      } else {
        pt[s].w = __uint_as_float(~0u);
      }
      if (part == 0) parent[i] = i;
    } else {
      pt[s].w = __uint_as_float(~0u);
    }
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const uint32_t lc = __float_as_uint(pt[s].w);
    pack[s] = lc == ~0u ? ~0u : (lc | (((pack[s] >> ((lc & 1u) * 16u)) & 0xFFFFu) << 16));
  }
  uint32_t* my_hi = hi_rank + (size_t)part * (kFuseMaxObs - kFuseRegObs);
  if (kBig) {
    for (uint32_t i0 = kFuseRegObs; i0 < n; i0 += 4096u) {
      float4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + (uint32_t)u * 1024u + tid;
        q[u] = i < n ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      uint32_t oldw[4], lcs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + (uint32_t)u * 1024u + tid;
        lcs[u] = ~0u; oldw[u] = 0u;
        if (i < n) {
          const uint32_t cell = (uint32_t)((grid_cz(g, q[u].z) * g.ny + grid_cy(g, q[u].y)) * g.nx + grid_cx(g, q[u].x));
          lower += cell < c_lo ? 1u : 0u;
          if (cell >= c_lo && cell < c_hi) {
            const uint32_t lc = cell - c_lo, wi = lc >> 1;
            oldw[u] = atomicAdd(&cnt2[wi + (wi >> lw)], 1u << ((lc & 1u) * 16u));
            lcs[u] = lc;
          }
          if (part == 0) parent[i] = i;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + (uint32_t)u * 1024u + tid;
        if (i < n) my_hi[i - kFuseRegObs] = lcs[u] == ~0u ? ~0u : (lcs[u] | (((oldw[u] >> ((lcs[u] & 1u) * 16u)) & 0xFFFFu) << 16));
      }
    }
  }
  uint32_t base;
  (void)block_excl_scan<16>(lower, wsum, &base);               // (ends with a barrier: the counts are complete too)
  MKF_STAMP(33);
  uint32_t sum = 0;
  for (uint32_t j = 0; j < wpt; ++j) {
    const uint32_t v = cnt2[tid * (wpt + 1u) + j];
    sum += (v & 0xFFFFu) + (v >> 16);
  }
  uint32_t tot;
  uint32_t run = base + block_excl_scan<16>(sum, wsum, &tot);
  for (uint32_t j = 0; j < wpt; ++j) {
    const uint32_t v = cnt2[tid * (wpt + 1u) + j];
    const uint32_t a = v & 0xFFFFu, b = v >> 16;
    cnt2[tid * (wpt + 1u) + j] = run | ((run + a) << 16);      // (starts <= kFuseMaxObs fit 16 bits)
    run += a + b;
  }
  __syncthreads();
  uint4* cs4 = reinterpret_cast<uint4*>(g.cell_start + c_lo);
  for (uint32_t j = 0; j < wpt / 2u; ++j) {                    // coalesced copy-out of the starts
    const uint32_t q = j * 1024u + tid, c0 = c_lo + 4u * q, w = 2u * q;
    const uint32_t v0 = cnt2[w + (w >> lw)], v1 = cnt2[w + 1u + ((w + 1u) >> lw)];
    if (c0 < cells) cs4[q] = make_uint4(v0 & 0xFFFFu, v0 >> 16, v1 & 0xFFFFu, v1 >> 16);   // (entries past `cells` hold n: the scan ran on)
  }
  if (tid == 0 && part == 0) g.cell_start[cells] = n;
  MKF_STAMP(34);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const uint32_t i = (uint32_t)s * 1024u + tid;
    if (pack[s] != ~0u) {
      const uint32_t lc = pack[s] & 0xFFFFu, wi = lc >> 1;
      const uint32_t st = (cnt2[wi + (wi >> lw)] >> ((lc & 1u) * 16u)) & 0xFFFFu;
      g.sorted[st + (pack[s] >> 16)] = make_float4(pt[s].x, pt[s].y, pt[s].z, __int_as_float((int)i));
    }
  }
  if (kBig) {
    for (uint32_t i0 = kFuseRegObs; i0 < n; i0 += 4096u) {
      float4 q[4];
      uint32_t wd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + (uint32_t)u * 1024u + tid;
        wd[u] = i < n ? my_hi[i - kFuseRegObs] : ~0u;          // (this lane's own stores)
        q[u] = i < n ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + (uint32_t)u * 1024u + tid;
        if (wd[u] != ~0u) {
          const uint32_t lc = wd[u] & 0xFFFFu, wi = lc >> 1;
          const uint32_t st = (cnt2[wi + (wi >> lw)] >> ((lc & 1u) * 16u)) & 0xFFFFu;
          g.sorted[st + (wd[u] >> 16)] = make_float4(q[u].x, q[u].y, q[u].z, __int_as_float((int)i));
        }
      }
    }
  }
  MKF_STAMP(35);
}

// count launch (DDDMR_MKF_GRID=global only): cell counts
__global__ __launch_bounds__(256) void k_mkf_count(PointGrid obs, FuseBufs fb) { fuse_grid_count(obs, fb.pts, fb.parent, fb.cell_count, fb.slot); }

// grid launch: first blocks: the observation grid  |  the others: every slot of the store
template <bool kBig>
__global__ __launch_bounds__(1024) void k_mkf_grid_fov(MarkParams k, MarkStore s, PointGrid obs, FuseBufs fb, MarkCounters* __restrict__ cnt,
                                                       uint32_t nb_grid) {
  __shared__ uint32_t cnt2[33 * 1024];
  __shared__ uint32_t wsum[16];
  if (blockIdx.x < nb_grid) {
    if (kBig || fb.grid_in_lds) fuse_grid_build_lds<kBig>(obs, fb.pts, fb.parent, cnt2, wsum, blockIdx.x, nb_grid, fb.hi_rank);
    else fuse_grid_scan_scatter(obs, fb.pts, fb.cell_count, fb.slot, cnt2, wsum);
    return;
  }
  // every slot of the store: no owner yet in this update; alive markings inside the window (integer test, one lane per
  // slot) are compacted in LDS, then the field-of-view test of k_mk_fov runs on the dense list (its double asin / atan2
  // cost a wave as much as a lane: on the sparse slots four times the waves did the same work); those in view go on the
  // ray-test list
  uint32_t* lst = cnt2;
  const uint32_t slot = (blockIdx.x - nb_grid) * 1024u + threadIdx.x;
  bool inwin = false;
  if (slot <= k.table_mask) {
    s.owner[slot] = 0ull;
    if (s.alive[slot]) {
      int x, y, z;
      voxel_unkey(s.keys[slot], &x, &y, &z);
      // map iteration lower_bound(min) .. lower_bound(max): keys in [min, max) on every axis (:487-516)
      inwin = !(x < k.wx0 || x >= k.wx1 || y < k.wy0 || y >= k.wy1 || z < k.wz0 || z >= k.wz1);
    }
  }
  uint32_t n_in;
  const uint32_t at = block_excl_scan<16>(inwin ? 1u : 0u, wsum, &n_in);
  if (inwin) lst[at] = slot;
  __syncthreads();
  if (threadIdx.x == 0 && n_in) atomicAdd(&cnt->n_in_window, n_in);
  if ((threadIdx.x & ~63u) >= n_in) return;                    // (whole waves leave)
  bool inview = false;
  uint32_t sl = 0;
  if (threadIdx.x < n_in) {
    sl = lst[threadIdx.x];
    int x, y, z;
    voxel_unkey(s.keys[sl], &x, &y, &z);
    const float px = (float)(x * k.res), py = (float)(y * k.res), pz = (float)(z * k.hres);
    inview = in_lidar_observation(k, px, py, pz);              // outside the sensor's view: stays (:531-540)
  }
  const unsigned long long bv = __ballot(inview);
  uint32_t base = 0;
  if ((threadIdx.x & 63) == 0 && bv) base = atomicAdd(&cnt->n_clear, (uint32_t)__popcll(bv));
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  if (inview) fb.clear_list[base + lanes_below(bv)] = sl;
}

// ---------------------------------------------------------------------------------------------
// clear launch: selfClear (a wave per listed marking)  |  Euclidean clustering, four lanes per point
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

__global__ __launch_bounds__(256) void k_mkf_clear_cc(MarkParams k, MarkStore s, PointGrid prev, PointGrid obs, PointGrid ground, FuseBufs fb,
                                                      MarkCounters* __restrict__ cnt, uint32_t nb_clear) {
  if (blockIdx.x < nb_clear) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w < cnt->n_clear) {
      const uint32_t slot = fb.clear_list[w];
      const int lane = threadIdx.x & 63;
      if (mk_clear_wave<false>(k, s, prev, cnt, slot, lane)) {
        // the generator points of the removed marking, for this update's removePCPtr (launches 4 / 5): into the band of
        // the row of ground cells they fall in, or onto the list of points to be walked one by one.  (Copied now: the
        // commit launch may hand the slot to a new cluster.)
        const uint32_t ofs = s.pts_ofs[slot], n = s.pts_n[slot];
        const float r = (float)k.inflation + 1e-4f;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
          const uint32_t i = i0 + (uint32_t)lane;
          bool walk = false;
          float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i < n) {
            p = s.pool[ofs + i];
            walk = !band_push(ground, fb.rg, fb.unmark_bands, p, r);
          }
          const unsigned long long bw = __ballot(walk);
          uint32_t base = 0;
          if (lane == 0 && bw) base = atomicAdd(&cnt->n_unmark_pts, (uint32_t)__popcll(bw));
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (walk) fb.unmark_pts[base + lanes_below(bw)] = p;
        }
      }
    }
    return;
  }
  // pcl::extractEuclideanClusters as connected components (k_mk_cc_union); the (z, y) rows of a point's tolerance box
  // are dealt to four lanes
  const uint32_t t = (blockIdx.x - nb_clear) * 256 + threadIdx.x;
  const uint32_t i = t >> 2;
  const int sub = (int)(t & 3u);
  if (i >= k.n_obs) return;
  const float4 p = fb.pts[i];
  const float r = k.tol + 1e-4f;
  const int x0 = grid_cx(obs, p.x, -r), x1 = grid_cx(obs, p.x, r);
  const int y0 = grid_cy(obs, p.y, -r), y1 = grid_cy(obs, p.y, r);
  const int z0 = grid_cz(obs, p.z, -r), z1 = grid_cz(obs, p.z, r);
  const int nys = y1 - y0 + 1, nrows = nys * (z1 - z0 + 1);
  for (int rr = sub; rr < nrows; rr += 4) {
    const int cz = z0 + rr / nys, cy = y0 + rr % nys;
    const uint32_t b = obs.cell_start[(cz * obs.ny + cy) * obs.nx + x0], e = obs.cell_start[(cz * obs.ny + cy) * obs.nx + x1 + 1];
    for (uint32_t q = b; q < e; ++q) {
      const float4 o = obs.sorted[q];
      const uint32_t j = (uint32_t)__float_as_int(o.w);
      if (j < i && l2_simple(o.x, o.y, o.z, p.x, p.y, p.z) < k.tol2) cc_union_pair(fb.parent, i, j);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// seed launch: every point's seed (the smallest point index of its component = the point PCL starts the cluster from);
// the per-cluster records of every point index start empty (a cluster is named by its seed's index)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fuse_roots(uint32_t n, uint32_t* parent, ClusterArrays c) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = cc_find(parent, i);
  __hip_atomic_store(&parent[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (an ancestor at any time: safe for the finds of others)
  c.size[i] = 0u;
  c.state[i] = 0u;
}

// ---------------------------------------------------------------------------------------------
// ground search of one ball by one wave: the (z, y) rows of the ball's box are looked up by one lane each, their
// runs of cell-sorted points flattened over the 64 lanes (k_mk_dgraph / k_mk_unmark walk row after row: two dependent
// round trips per row)
// ---------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void ground_ball_wave(const PointGrid& g, float qx, float qy, float qz, float r, int lane, F&& f) {
  const int x0 = grid_cx(g, qx, -r), x1 = grid_cx(g, qx, r);
  const int y0 = grid_cy(g, qy, -r), y1 = grid_cy(g, qy, r);
  const int z0 = grid_cz(g, qz, -r), z1 = grid_cz(g, qz, r);
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

// ---------------------------------------------------------------------------------------------
// dGraph / lethal updates, node by node.  The reference walks from every generator point to the ground nodes around it
// (computeMinDistanceFromObstacle2GroundNodes, removePCPtr); k_mk_dgraph / k_mk_unmark do the same with a wave per
// point -- half a million scattered device-scope atomics (or small stores) per update onto the few hundred cache lines of
// the window's nodes, which is what those launches waited for (PMC: waves waiting 99 % of their cycles, 2 % VALU).
// Turned around, every ground node of the window looks at the points that can reach it -- the points are dropped into
// bands (the y row of ground cells they fall in) as they are made, a node's row reads the 2 delta + 1 bands around it,
// staged through LDS 256 at a time, a broadcast read per pair -- and writes ONCE: the minimum of a set of floats / "any
// point within the radius" do not depend on the order, and the per-pair arithmetic is the reference's.  Points whose
// ball leaves the window's range of ground cells (a marking made from another pose, a cloud handed over uncropped) or
// that find their band full are walked point by point as before.
// ---------------------------------------------------------------------------------------------
// one block = up to four 64-node segments of one row of ground cells (a wave each) against every `n_part`-th 256-point
// chunk of the bands that can reach the row
template <bool kMark>
__device__ __forceinline__ void splat_band_block(const MarkParams& k, const MarkStore& s, const PointGrid& g, const SplatRange& rg,
                                                 const BandList& bl, const uint32_t row, const uint32_t seg_group, const uint32_t part,
                                                 const uint32_t n_part, float4* stage /* LDS [256] */) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nyr = rg.cy1 - rg.cy0 + 1;
  const int cy = rg.cy0 + (int)(row % (uint32_t)nyr), cz = (int)(row / (uint32_t)nyr);
  const uint32_t seg = seg_group * 4u + (uint32_t)w;
  bool have = false;
  float4 nd = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const int rb = (cz * g.ny + cy) * g.nx;
    const uint32_t b = g.cell_start[rb + rg.cx0], e = g.cell_start[rb + rg.cx1 + 1];
    const uint32_t at = b + seg * 64u + (uint32_t)lane;
    if (seg < rg.segs && at < e) { nd = g.sorted[at]; have = true; }
  }
  // Per pair the reference tests the 3-D distance (FLANN's L2_Simple: dx^2, + dy^2, + dz^2) against the squared radius and
  // then takes d = sqrtf(dx^2 + dy^2) -- the partial sum after two terms of the very same accumulation.  sqrtf is
  // monotone, so the minimum of d over the hits is sqrtf of the minimum partial sum, and "some d <= inscribed radius" is
  // "the minimum d <= it": the loop keeps one float minimum, branch-free (11 VALU per pair; with the sqrtf and the
  // compares under a per-pair branch a wave spent 36).
  const float r2 = static_cast<float>(k.inflation * k.inflation);
  float min_d2 = 3.0e38f;
  // the fills of the (at most 2 delta + 1) bands in reach, one lane each
  const int b_lo = max(cy - rg.delta, rg.cy0), b_hi = min(cy + rg.delta, rg.cy1);
  const uint32_t my_n = (b_lo + lane <= b_hi && lane < 64) ? min(bl.cnt[b_lo + lane - rg.cy0], kBandCap) : 0u;
  uint32_t ci = 0;
  for (int band = b_lo; band <= b_hi; ++band) {
    const uint32_t n_b = (uint32_t)__builtin_amdgcn_readlane((int)my_n, band - b_lo);
    const float4* src = bl.pts + (size_t)(band - rg.cy0) * kBandCap;
    for (uint32_t c0 = 0; c0 < n_b; c0 += 256u, ++ci) {
      if (ci % n_part != part) continue;
      __syncthreads();
      stage[tid] = c0 + tid < n_b ? src[c0 + tid] : make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);    // (padding: out of any ball)
      __syncthreads();
      const uint32_t m = min(256u, n_b - c0);
      if (have)
        for (uint32_t j = 0; j < m; j += 8) {                    // eight broadcast reads in flight per step
          float4 pp[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) pp[u] = stage[j + u];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            float d = fsub(nd.x, pp[u].x);
            float acc = fmul(d, d);
            d = fsub(nd.y, pp[u].y);
            acc = fadd(acc, fmul(d, d));
            const float d2 = acc;                                // dx^2 + dy^2
            d = fsub(nd.z, pp[u].z);
            acc = fadd(acc, fmul(d, d));                         // l2_simple(node, point)
            min_d2 = fminf(min_d2, acc < r2 ? d2 : 3.0e38f);
          }
        }
    }
  }
  if (!have || !(min_d2 < 3.0e38f)) return;
  const float dmin = sqrtf(min_d2);                              // z dropped on purpose (cluster_marking.cpp:86-88)
  const bool close = dmin <= k.inscribed;
  const int node = __float_as_int(nd.w);
  if (kMark) {
    // DynamicGraph::setValue: graph_[key] = min(graph_[key], d); non-negative doubles order like their bit patterns.
    // (several blocks may land on one node: atomic; values only fall during the launch, so a plain read that is already
    // <= d settles it without one)
    if ((double)dmin < s.dgraph[node])
      atomicMin(reinterpret_cast<unsigned long long*>(s.dgraph) + node, (unsigned long long)__double_as_longlong((double)dmin));
    if (close) s.lethal[node] = 1;
  } else {
    s.dgraph[node] = 9999.0;                                        // clearValue(node, 9999.0) (:125-138)
    if (close) s.lethal[node] = 0;
  }
}


// the walk from one point, for points whose ball leaves the window's range of ground cells
template <bool kMark>
__device__ __forceinline__ void splat_point_wave(const MarkParams& k, const MarkStore& s, const PointGrid& ground, const float4 p, int lane) {
  const float r = (float)k.inflation, r2 = static_cast<float>(k.inflation * k.inflation);
  ground_ball_wave(ground, p.x, p.y, p.z, r + 1e-4f, lane, [&](const float4 g) {
    if (l2_simple(g.x, g.y, g.z, p.x, p.y, p.z) < r2) {
      const int node = __float_as_int(g.w);
      const float dx = p.x - g.x, dy = p.y - g.y;
      const float d = sqrtf(dx * dx + dy * dy);
      if (kMark) {
        atomicMin(reinterpret_cast<unsigned long long*>(s.dgraph) + node, (unsigned long long)__double_as_longlong((double)d));
        if (d <= k.inscribed) s.lethal[node] = 1;
      } else {
        s.dgraph[node] = 9999.0;
        if (d <= k.inscribed) s.lethal[node] = 0;
      }
    }
  });
}

// block `bi` of the `nb` blocks that undo the dGraph / lethal entries of the markings this update cleared: the first nb_band
// node by node over the bands, the rest point by point over the points that found no band
__device__ __forceinline__ void fuse_unmark_block(const MarkParams& k, const FuseBufs& fb, const MarkStore& s, const PointGrid& ground,
                                                  const MarkCounters* __restrict__ cnt, uint32_t bi, const uint32_t nb, const uint32_t nb_band,
                                                  const uint32_t seg_groups, const uint32_t n_part, float4* stage) {
  if (bi < nb_band) {
    splat_band_block<false>(k, s, ground, fb.rg, fb.unmark_bands, bi / (seg_groups * n_part), (bi / n_part) % seg_groups, bi % n_part, n_part, stage);
    return;
  }
  bi -= nb_band;
  const uint32_t n_src = cnt->n_unmark_pts;
  const int lane = threadIdx.x & 63;
  const uint32_t stride = (nb - nb_band) * 4u;
  for (uint32_t h = bi * 4u + (threadIdx.x >> 6); h < n_src; h += stride) splat_point_wave<false>(k, s, ground, fb.unmark_pts[h], lane);
}

// seed launch: seeds  |  (DDDMR_MKF_UNMARK=roots, or nothing to mark) removePCPtr of the markings the clear launch removed: ground node by ground node, and point by point for
// the points that found no band
__global__ __launch_bounds__(256) void k_mkf_roots_unmark(MarkParams k, FuseBufs fb, ClusterArrays c, MarkStore s, PointGrid ground,
                                                          const MarkCounters* __restrict__ cnt, uint32_t nb_roots, uint32_t nb_band,
                                                          uint32_t seg_groups, uint32_t n_part) {
  __shared__ float4 stage[256];
  if (blockIdx.x < nb_roots) {
    fuse_roots(k.n_obs, fb.parent, c);
    return;
  }
  fuse_unmark_block(k, fb, s, ground, cnt, blockIdx.x - nb_roots, gridDim.x - nb_roots, nb_band, seg_groups, n_part, stage);
}

// ---------------------------------------------------------------------------------------------
// partition launch, blocks 0..63: one partition of the clusters, everything in LDS
// ---------------------------------------------------------------------------------------------
struct PartLds {              // carve-up of the dynamic LDS of a partition workgroup
  float *px, *py, *pz;        // [P] the partition's points (index order); second half: projected 0.2 m voxel centroids
  uint32_t *ka, *kb;          // [P] sort keys, ping-pong; the free one holds the voxel key by sorted position
  uint16_t *pa, *pb;          // [P] payloads, ping-pong
  uint16_t* rk;               // [P + 2] rank scratch of a pass; group index by sorted position
  uint16_t* start;            // [P + 2] first sorted position of a cluster
  uint16_t* mine;             // [P] observation index of a local point
  uint16_t* lcid;             // [P] local cluster of a local point / of a voxel centroid
  uint16_t* gci;              // [P] local cluster -> its seed's observation index
  uint8_t* state;             // [P] per local cluster
  uint32_t* hist;             // [4 * 128]
  uint32_t* wsum;             // [4]
  int* red;                   // [24]
  uint32_t* misc;             // [8]
};
constexpr size_t kPartLdsBytes = (size_t)kPartCap * (12 + 8 + 4 + 2 + 2 + 2 + 2 + 2 + 1) + 8 + 512 * 4 + 4 * 4 + 24 * 4 + 8 * 4 + 64;

__device__ __forceinline__ PartLds part_lds(unsigned char* lds) {
  PartLds L;
  constexpr uint32_t P = kPartCap;
  L.px = reinterpret_cast<float*>(lds); L.py = L.px + P; L.pz = L.py + P;
  L.ka = reinterpret_cast<uint32_t*>(L.pz + P); L.kb = L.ka + P;
  L.hist = L.kb + P; L.wsum = L.hist + 512; L.red = reinterpret_cast<int*>(L.wsum + 4); L.misc = reinterpret_cast<uint32_t*>(L.red + 24);
  L.pa = reinterpret_cast<uint16_t*>(L.misc + 8); L.pb = L.pa + P;
  L.rk = L.pb + P; L.start = L.rk + P + 2; L.mine = L.start + P + 2; L.lcid = L.mine + P; L.gci = L.lcid + P;
  L.state = reinterpret_cast<uint8_t*>(L.gci + P);
  return L;
}

// block-wide minimum / maximum of three ints over 4 waves (red = 24 LDS ints)
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
    for (int a = 0; a < 3; ++a) { red[a * 4 + w] = mn[a]; red[12 + a * 4 + w] = mx[a]; }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = min(min(red[a * 4], red[a * 4 + 1]), min(red[a * 4 + 2], red[a * 4 + 3]));
    mx[a] = max(max(red[12 + a * 4], red[12 + a * 4 + 1]), max(red[12 + a * 4 + 2], red[12 + a * 4 + 3]));
  }
  __syncthreads();
}

// One stable LSD radix pass over a 7-bit digit of m keys in LDS (kin / pin -> kout / pout).  Sorted position p belongs
// to wave p / (64 E), E = elements per lane.  The lanes of a wave that hold the same digit find each other with seven
// ballots (rank among themselves = lanes below in the match mask); a per-wave histogram column carries the count over
// the wave's rounds; one block scan over (digit, wave) turns the columns into bases.
__device__ __forceinline__ void lds_radix_pass(const uint32_t m, const int shift, const uint32_t* kin, const uint16_t* pin, uint32_t* kout,
                                               uint16_t* pout, uint16_t* rk, uint32_t* hist, uint32_t* wsum) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t E = (m + 255u) / 256u;
  hist[tid] = 0u;
  hist[tid + 256] = 0u;
  __syncthreads();
  for (uint32_t s = 0; s < E; ++s) {
    const uint32_t p = (uint32_t)w * 64u * E + s * 64u + (uint32_t)lane;
    if (p < m) {
      const uint32_t d = (kin[p] >> shift) & 127u;
      unsigned long long mm = __ballot(true);
#pragma unroll
      for (int b = 0; b < 7; ++b) {
        const bool bit = ((d >> b) & 1u) != 0u;
        const unsigned long long bal = __ballot(bit);
        mm &= bit ? bal : ~bal;
      }
      const uint32_t below = lanes_below(mm);
      const uint32_t pre = hist[w * 128 + d];
      rk[p] = (uint16_t)(pre + below);
      __builtin_amdgcn_wave_barrier();
      if (below == 0u) hist[w * 128 + d] = pre + (uint32_t)__popcll(mm);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {
    // entry j = digit * 4 + wave; lane t scans j = 2 t, 2 t + 1
    const uint32_t j0 = 2u * tid, j1 = j0 + 1u;
    const uint32_t a0 = (j0 & 3u) * 128u + (j0 >> 2), a1 = (j1 & 3u) * 128u + (j1 >> 2);
    const uint32_t a = hist[a0], b = hist[a1];
    uint32_t tot;
    const uint32_t ex = block_excl_scan<4>(a + b, wsum, &tot);
    hist[a0] = ex;
    hist[a1] = ex + a;
  }
  __syncthreads();
  for (uint32_t s = 0; s < E; ++s) {
    const uint32_t p = (uint32_t)w * 64u * E + s * 64u + (uint32_t)lane;
    if (p < m) {
      const uint32_t key = kin[p];
      const uint32_t dst = hist[w * 128 + ((key >> shift) & 127u)] + rk[p];
      kout[dst] = key;
      pout[dst] = pin[p];
    }
  }
  __syncthreads();
}

// stable sort of (key, payload) pairs on the key bits [0, nbits): the current buffers are swapped pass by pass
__device__ __forceinline__ void lds_radix_sort(const uint32_t m, const int nbits, uint32_t*& kc, uint16_t*& pc, uint32_t*& kf, uint16_t*& pf,
                                               uint16_t* rk, uint32_t* hist, uint32_t* wsum) {
  for (int sh = 0; sh < nbits; sh += 7) {
    lds_radix_pass(m, sh, kc, pc, kf, pf, rk, hist, wsum);
    uint32_t* tk = kc; kc = kf; kf = tk;
    uint16_t* tp = pc; pc = pf; pf = tp;
  }
}

// The same stable order for a partition of at most 256 elements (the usual case: ~165-220 points per partition at
// 10 k points), one element per lane: (cluster, key, position) packed into one 48-bit word (cluster < 256 here, key <
// 2^kFuseKeyBits, position < 256 -- all distinct), an element's place = the number of smaller words, counted with
// broadcast 128-bit LDS reads, 8 words per batch so the reads are in flight together (one by one the loop is LDS
// latency: measured 68 us for the kernel against 42 with the radix passes).  ~1 us against ~1.1 us for EACH of the 3-5
// radix passes, whose cost at this size is their barriers and scans.  Reads cluster[] and key[] by element; leaves
// payload, cluster and key by sorted position in pc / kc / kf.  packed = 256 u64 of scratch (L.hist).
__device__ __forceinline__ void lds_rank_sort(const uint32_t m, const uint16_t* cluster, const uint32_t* key, uint32_t* kc, uint16_t* pc,
                                              uint32_t* kf, uint32_t* scratch) {
  unsigned long long* packed = reinterpret_cast<unsigned long long*>(scratch);
  const uint32_t i = threadIdx.x;
  uint32_t hi = 0, lo = 0;
  unsigned long long mine = ~0ull;
  if (i < m) {
    hi = cluster[i];
    lo = key ? key[i] : 0u;
    mine = ((unsigned long long)hi << 36) | ((unsigned long long)lo << 8) | (unsigned long long)i;
  }
  packed[i] = mine;                                                         // (256 lanes: the tail is the sentinel)
  __syncthreads();
  uint32_t rank = 0;
  const uint32_t m8 = (m + 7u) & ~7u;
  for (uint32_t j = 0; j < m8; j += 8) {
    unsigned long long w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = packed[j + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) rank += w[u] < mine ? 1u : 0u;
  }
  if (i < m) { pc[rank] = (uint16_t)i; kc[rank] = hi; kf[rank] = lo; }
  __syncthreads();
}

// groups of equal (cluster, voxel) runs in a sorted sequence of m elements: kc = cluster by position, vk = voxel key by
// position.  rk[j] = number of group starts before position j (rk[m] = groups), returns the number of groups.
__device__ __forceinline__ uint32_t lds_group_index(const uint32_t m, const uint32_t* kc, const uint32_t* vk, const uint8_t* state, const bool all,
                                                    uint16_t* rk, uint32_t* wsum) {
  uint32_t run = 0;
  for (uint32_t j0 = 0; j0 < m; j0 += 256) {
    const uint32_t j = j0 + threadIdx.x;
    uint32_t fl = 0;
    if (j < m && (all || state[kc[j]] != 0u)) fl = (j == 0u || kc[j - 1] != kc[j] || vk[j - 1] != vk[j]) ? 1u : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<4>(fl, wsum, &tot);
    if (j < m) rk[j] = (uint16_t)(run + ex);
    run += tot;
  }
  if (threadIdx.x == 0) rk[m] = (uint16_t)run;
  __syncthreads();
  return run;
}

__device__ __forceinline__ void fuse_partition(const MarkParams& k, const FuseBufs& fb, ClusterArrays c, MarkStore s, const PointGrid& ground,
                                               const PointGrid& map, uint32_t n_map, MarkCounters* __restrict__ cnt, unsigned char* lds) {
  constexpr uint32_t P = kPartCap;
  PartLds L = part_lds(lds);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t n = k.n_obs, part = blockIdx.x;
  uint32_t *kc = L.ka, *kf = L.kb;
  uint16_t *pc = L.pa, *pf = L.pb;
  MKF_STAMP(0);
#ifdef DDDMR_PHASE_STAMPS
  const unsigned long long part_t0 = (unsigned long long)clock64();
#endif
  // ---- P0: the partition's points, in index order.  Every wave scans a quarter of the seeds; what it selects goes to
  //      its own staging list first (ka | kb as 4 x P shorts), the four lists are then concatenated ----
  uint32_t m;
  {
    uint16_t* stage = reinterpret_cast<uint16_t*>(L.ka) + (size_t)w * P;
    const uint32_t chunk = (((n + 3u) / 4u) + 63u) & ~63u;
    const uint32_t lo = min((uint32_t)w * chunk, n), hi = min(lo + chunk, n);
    uint32_t have = 0;
    for (uint32_t i0 = lo; i0 < hi; i0 += 64u * 32u) {         // (32 loads in flight: a wave's quarter of 10 k seeds in two trips)
      uint32_t r[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const uint32_t i = i0 + 64u * u + lane;
        r[u] = i < hi ? fb.parent[i] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const uint32_t i = i0 + 64u * u + lane;
        const bool sel = r[u] != 0xFFFFFFFFu && part_of(r[u]) == part;
        const unsigned long long b = __ballot(sel);
        if (sel && have + lanes_below(b) < P) stage[have + lanes_below(b)] = (uint16_t)i;
        have += (uint32_t)__popcll(b);
      }
    }
    if (lane == 0) L.wsum[w] = have;
    __syncthreads();
    const uint32_t h0 = L.wsum[0], h1 = L.wsum[1], h2 = L.wsum[2], h3 = L.wsum[3];
    m = h0 + h1 + h2 + h3;
    if (m > P || h0 > P || h1 > P || h2 > P || h3 > P) {     // a cluster (or a collision of clusters) beyond one workgroup
      if (tid == 0) cnt->fallback = 1u;
      return;
    }
    const uint32_t base = (w > 0 ? h0 : 0u) + (w > 1 ? h1 : 0u) + (w > 2 ? h2 : 0u);
    for (uint32_t j = lane; j < have; j += 64) L.mine[base + j] = stage[j];
    __syncthreads();
  }
  if (m == 0) return;
  // points and local clusters: a point's seed is a point of the same partition, found by binary search in `mine`
  uint32_t ncl;
  {
    uint32_t run = 0;
    for (uint32_t e0 = 0; e0 < m; e0 += 256) {
      const uint32_t e = e0 + tid;
      uint32_t isr = 0;
      if (e < m) {
        const uint32_t i = L.mine[e];
        const float4 p = fb.pts[i];
        L.px[e] = p.x; L.py[e] = p.y; L.pz[e] = p.z;
        const uint32_t r = fb.parent[i];
        kf[e] = r;
        isr = r == i ? 1u : 0u;
      }
      uint32_t tot;
      const uint32_t ex = block_excl_scan<4>(isr, L.wsum, &tot);
      if (isr) { L.lcid[e] = (uint16_t)(run + ex); L.gci[run + ex] = L.mine[e]; }
      run += tot;
    }
    ncl = run;
    __syncthreads();
    for (uint32_t e = tid; e < m; e += 256) {
      const uint32_t r = kf[e];
      uint32_t a = 0, b = m;                                  // first position with mine[pos] >= r
      while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if ((uint32_t)L.mine[mid] < r) a = mid + 1; else b = mid;
      }
      kc[e] = L.lcid[a];                                       // (seeds read their own entry)
      pc[e] = (uint16_t)e;
    }
    __syncthreads();
    for (uint32_t e = tid; e < m; e += 256) L.lcid[e] = (uint16_t)kc[e];
    __syncthreads();
  }
  MKF_STAMP(1);
  // ---- P1: sort 1 = clusters in seed order, their points in index order (stable by cluster) ----
  const int cbits = bits_for((int)ncl - 1);
  if (m <= 256u) lds_rank_sort(m, L.lcid, nullptr, kc, pc, kf, L.hist);
  else lds_radix_sort(m, cbits, kc, pc, kf, pf, L.rk, L.hist, L.wsum);
  for (uint32_t j = tid; j < m; j += 256)
    if (j == 0u || kc[j - 1] != kc[j]) L.start[kc[j]] = (uint16_t)j;
  if (tid == 0) L.start[ncl] = (uint16_t)m;
  __syncthreads();
  MKF_STAMP(2);
  // ---- P2: per cluster centroid (floats added in index order, / size), min size, "centre attached to the ground"
  //      (k_mk_cluster_stage1; :343-368) ----
  uint32_t kept = 0;
  for (uint32_t ci = tid; ci < ncl; ci += 256) {
    const uint32_t b = L.start[ci], e = L.start[ci + 1], gi = L.gci[ci];
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (uint32_t j = b; j < e; ++j) {
      const uint32_t q = pc[j];
      cx += L.px[q]; cy += L.py[q]; cz += L.pz[q];
    }
    const float sz = (float)(e - b);
    cx /= sz; cy /= sz; cz /= sz;
    c.size[gi] = e - b;
    c.centroid[gi] = make_float4(cx, cy, cz, 0.f);
    bool ok = (int)(e - b) >= k.min_cluster;
    if (ok) ++kept;
    if (ok && grid_radius_count(ground, cx, cy, cz, 0.05f + 1e-4f, static_cast<float>(0.05 * 0.05), 1) > 0) ok = false;
    L.state[ci] = ok ? 1u : 0u;
  }
  __syncthreads();
  MKF_STAMP(3);
  // ---- P3: sort 2 = (cluster, 0.2 m voxel z|y|x, point index): voxel order first, then stable by cluster ----
  int bx, by, bz, mn[3], mx[3];
  {
    const float inv = 1.0f / 0.2f;
    mn[0] = mn[1] = mn[2] = 0x7FFFFFFF; mx[0] = mx[1] = mx[2] = (int)0x80000000;
    for (uint32_t e = tid; e < m; e += 256) {
      const int vx = (int)floorf(L.px[e] * inv), vy = (int)floorf(L.py[e] * inv), vz = (int)floorf(L.pz[e] * inv);
      mn[0] = min(mn[0], vx); mx[0] = max(mx[0], vx);
      mn[1] = min(mn[1], vy); mx[1] = max(mx[1], vy);
      mn[2] = min(mn[2], vz); mx[2] = max(mx[2], vz);
    }
    block_minmax3(mn, mx, L.red);
    // (differences in 64 bits: points may be anywhere; a range that does not fit falls back to the general route)
    const long long rx = (long long)mx[0] - mn[0], ry = (long long)mx[1] - mn[1], rz = (long long)mx[2] - mn[2];
    if (rx >= (1ll << 27) || ry >= (1ll << 27) || rz >= (1ll << 27)) { if (tid == 0) cnt->fallback = 1u; return; }
    bx = bits_for((int)rx); by = bits_for((int)ry); bz = bits_for((int)rz);
    if (bx + by + bz > kFuseKeyBits) { if (tid == 0) cnt->fallback = 1u; return; }
    for (uint32_t e = tid; e < m; e += 256) {
      kc[e] = ((uint32_t)((int)floorf(L.pz[e] * inv) - mn[2]) << (bx + by)) | ((uint32_t)((int)floorf(L.py[e] * inv) - mn[1]) << bx) |
              (uint32_t)((int)floorf(L.px[e] * inv) - mn[0]);
      pc[e] = (uint16_t)e;
    }
    __syncthreads();
    if (m <= 256u) {
      lds_rank_sort(m, L.lcid, kc, kc, pc, kf, L.hist);
    } else {
      lds_radix_sort(m, bx + by + bz, kc, pc, kf, pf, L.rk, L.hist, L.wsum);
      for (uint32_t j = tid; j < m; j += 256) kc[j] = L.lcid[pc[j]];                       // re-key by cluster
      __syncthreads();
      lds_radix_sort(m, cbits, kc, pc, kf, pf, L.rk, L.hist, L.wsum);
    }
    for (uint32_t j = tid; j < m; j += 256) {                                               // voxel key by sorted position
      const uint32_t e = pc[j];
      kf[j] = ((uint32_t)((int)floorf(L.pz[e] * inv) - mn[2]) << (bx + by)) | ((uint32_t)((int)floorf(L.py[e] * inv) - mn[1]) << bx) |
              (uint32_t)((int)floorf(L.px[e] * inv) - mn[0]);
    }
    __syncthreads();
  }
  MKF_STAMP(4);
  // ---- P4: 0.2 m VoxelGrid of every cluster that passed P2: one lane per voxel adds its points in order (:370-374) ----
  const uint32_t ng2 = lds_group_index(m, kc, kf, L.state, false, L.rk, L.wsum);
  if (tid == 0) L.misc[0] = ng2 ? atomicAdd(&cnt->n_groups2, ng2) : 0u;
  __syncthreads();
  const uint32_t ds_base = L.misc[0];
  for (uint32_t j = tid; j < m; j += 256) {
    if (L.rk[j + 1] == L.rk[j]) continue;                       // not a group start
    const uint32_t cj = kc[j], vox = kf[j];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    uint32_t e = j;
    for (; e < m && kf[e] == vox && kc[e] == cj; ++e) {
      const uint32_t q = pc[e];
      sx += L.px[q]; sy += L.py[q]; sz += L.pz[q];
    }
    const float cntf = (float)(e - j);
    fb.ds[ds_base + L.rk[j]] = make_float4(sx / cntf, sy / cntf, sz / cntf, __int_as_float((int)cj));
  }
  wait_own_memory_ops();                                       // (read back below by other lanes of this workgroup: same CU, same L1)
  __syncthreads();
  MKF_STAMP(5);
  // ---- P5: "is it part of the static map", voxel key, in the sensor's view (k_mk_cluster_stage2; :375-430) ----
  for (uint32_t ci = tid; ci < ncl; ci += 256) {
    if (L.state[ci] != 1u) continue;
    const uint32_t gi = L.gci[ci];
    const float4 cen = c.centroid[gi];
    const size_t nds = (size_t)(L.rk[L.start[ci + 1]] - L.rk[L.start[ci]]);
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
    if (!(hit <= nds * k.ignore_ratio)) { L.state[ci] = 0u; continue; }
    const int vx = (int)(cen.x / k.res), vy = (int)(cen.y / k.res), vz = (int)(cen.z / k.hres);
    c.vkey[3 * gi + 0] = vx; c.vkey[3 * gi + 1] = vy; c.vkey[3 * gi + 2] = vz;
    const float px = (float)(vx * k.res), py = (float)(vy * k.res), pz = (float)(vz * k.hres);
    L.state[ci] = in_lidar_observation(k, px, py, pz) ? 2u : 0u;
  }
  __syncthreads();
  MKF_STAMP(6);
  // ---- P6: sort 3 = (cluster, 0.1 m voxel of the projected voxel centroid, order of P4) over accepted clusters
  //      (cluster_marking.cpp:54-64).  The projected points replace the partition's points in LDS. ----
  uint32_t m3;
  {
    uint32_t run = 0;
    for (uint32_t g0 = 0; g0 < ng2; g0 += 256) {
      const uint32_t g = g0 + tid;
      uint32_t on = 0;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g < ng2) {
        d = fb.ds[ds_base + g];
        on = L.state[(uint32_t)__float_as_int(d.w)] == 2u ? 1u : 0u;
      }
      uint32_t tot;
      const uint32_t ex = block_excl_scan<4>(on, L.wsum, &tot);
      if (on) {
        const uint32_t q = run + ex;
        float qx, qy, qz;
        project_on_base_plane(k, d.x, d.y, d.z, &qx, &qy, &qz);
        L.px[q] = qx; L.py[q] = qy; L.pz[q] = qz;
        L.lcid[q] = (uint16_t)__float_as_int(d.w);
      }
      run += tot;
    }
    m3 = run;
    __syncthreads();
  }
  uint32_t ng3 = 0;
  if (m3 > 0) {
    const float inv = 1.0f / 0.1f;
    mn[0] = mn[1] = mn[2] = 0x7FFFFFFF; mx[0] = mx[1] = mx[2] = (int)0x80000000;
    for (uint32_t e = tid; e < m3; e += 256) {
      const int vx = (int)floorf(L.px[e] * inv), vy = (int)floorf(L.py[e] * inv), vz = (int)floorf(L.pz[e] * inv);
      mn[0] = min(mn[0], vx); mx[0] = max(mx[0], vx);
      mn[1] = min(mn[1], vy); mx[1] = max(mx[1], vy);
      mn[2] = min(mn[2], vz); mx[2] = max(mx[2], vz);
    }
    block_minmax3(mn, mx, L.red);
    const long long rx = (long long)mx[0] - mn[0], ry = (long long)mx[1] - mn[1], rz = (long long)mx[2] - mn[2];
    if (rx >= (1ll << 27) || ry >= (1ll << 27) || rz >= (1ll << 27)) { if (tid == 0) cnt->fallback = 1u; return; }
    bx = bits_for((int)rx); by = bits_for((int)ry); bz = bits_for((int)rz);
    if (bx + by + bz > kFuseKeyBits) { if (tid == 0) cnt->fallback = 1u; return; }
    for (uint32_t e = tid; e < m3; e += 256) {
      kc[e] = ((uint32_t)((int)floorf(L.pz[e] * inv) - mn[2]) << (bx + by)) | ((uint32_t)((int)floorf(L.py[e] * inv) - mn[1]) << bx) |
              (uint32_t)((int)floorf(L.px[e] * inv) - mn[0]);
      pc[e] = (uint16_t)e;
    }
    __syncthreads();
    if (m3 <= 256u) {
      lds_rank_sort(m3, L.lcid, kc, kc, pc, kf, L.hist);
    } else {
      lds_radix_sort(m3, bx + by + bz, kc, pc, kf, pf, L.rk, L.hist, L.wsum);
      for (uint32_t j = tid; j < m3; j += 256) kc[j] = L.lcid[pc[j]];
      __syncthreads();
      lds_radix_sort(m3, cbits, kc, pc, kf, pf, L.rk, L.hist, L.wsum);
    }
    for (uint32_t j = tid; j < m3; j += 256) {
      const uint32_t e = pc[j];
      kf[j] = ((uint32_t)((int)floorf(L.pz[e] * inv) - mn[2]) << (bx + by)) | ((uint32_t)((int)floorf(L.py[e] * inv) - mn[1]) << bx) |
              (uint32_t)((int)floorf(L.px[e] * inv) - mn[0]);
    }
    __syncthreads();
    ng3 = lds_group_index(m3, kc, kf, L.state, true, L.rk, L.wsum);
  }
  MKF_STAMP(7);
  // ---- P7: 0.1 m VoxelGrid of the projected points -> generator points; first / count per cluster.  Every generator
  //      point also goes into the band of the row of ground cells it falls in (the commit launch reads them node by node): ranks
  //      inside the partition from LDS counters, then ONE device-scope atomic per band and partition (a returning
  //      atomic per point on the ~40 band counters, from 64 partitions at once, cost this phase 15 us) ----
  uint32_t* band_n = L.hist;                                     // [kBandMax] points of this partition per band
  uint32_t* band_at = L.hist + kBandMax;                         // [kBandMax] first entry of the partition in the band
  if (tid < (int)kBandMax) band_n[tid] = 0u;
  if (tid == 0) L.misc[1] = ng3 ? atomicAdd(&cnt->n_groups3, ng3) : 0u;
  __syncthreads();
  const uint32_t gen_base = L.misc[1];
  const float rball = (float)k.inflation + 1e-4f;
  for (uint32_t j = tid; j < m3; j += 256) {
    const uint32_t cj = kc[j];
    if (j == 0u || kc[j - 1] != cj) L.start[cj] = L.rk[j];       // (start: now the cluster's first generator point)
    pf[j] = 0xFFFFu;
    if (L.rk[j + 1] == L.rk[j]) continue;
    const uint32_t vox = kf[j];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    uint32_t e = j;
    for (; e < m3 && kf[e] == vox && kc[e] == cj; ++e) {
      const uint32_t q = pc[e];
      sx += L.px[q]; sy += L.py[q]; sz += L.pz[q];
    }
    const float cntf = (float)(e - j);
    const float4 gp = make_float4(sx / cntf, sy / cntf, sz / cntf, __int_as_float((int)L.gci[cj]));
    fb.gen[gen_base + L.rk[j]] = gp;
    if (fb.rg.bands && ball_in_range(ground, fb.rg, gp.x, gp.y, rball)) {     // (the commit launch walks the others point by point)
      const uint32_t band = (uint32_t)(grid_cy(ground, gp.y) - fb.rg.cy0);
      pf[j] = (uint16_t)band;
      L.mine[j] = (uint16_t)atomicAdd(&band_n[band], 1u);
    }
  }
  __syncthreads();
  if (tid < (int)kBandMax && band_n[tid]) band_at[tid] = atomicAdd(&fb.gen_bands.cnt[tid], band_n[tid]);
  __syncthreads();
  for (uint32_t j = tid; j < m3; j += 256) {
    const uint32_t band = pf[j];
    if (band == 0xFFFFu) continue;
    const uint32_t at = band_at[band] + L.mine[j];               // (< kBandCap: an update makes at most kFuseMaxObs generator points)
    fb.gen_bands.pts[(size_t)band * kBandCap + at] = fb.gen[gen_base + L.rk[j]];
  }
  __syncthreads();
  for (uint32_t j = tid; j < m3; j += 256) {
    const uint32_t cj = kc[j];
    if (j + 1 == m3 || kc[j + 1] != cj) {                       // last element of the cluster: rk[j + 1] = groups up to and including it
      const uint32_t gi = L.gci[cj];
      c.gen_first[gi] = gen_base + L.start[cj];
      c.gen_count[gi] = (uint32_t)L.rk[j + 1] - (uint32_t)L.start[cj];
    }
  }
  MKF_STAMP(8);
  // ---- P8: Marking::addPCPtr, slot part (k_mk_slots): marking_[x][y][z] is created or found; the claim with the
  //      highest priority (smallest cluster, then latest seed) keeps the voxel -- decided in the next launch ----
  uint32_t marked = 0, dup = 0, newk = 0;
  for (uint32_t ci = tid; ci < ncl; ci += 256) {
    const uint32_t gi = L.gci[ci];
    uint32_t st = L.state[ci];
    if (st == 2u) {
      const unsigned long long vk = voxel_key(c.vkey[3 * gi], c.vkey[3 * gi + 1], c.vkey[3 * gi + 2]);
      uint32_t slot = mk_hash(vk) & k.table_mask;
      bool found = false;
      for (uint32_t probe = 0; probe <= k.table_mask; ++probe) {
        const unsigned long long prev = atomicCAS(&s.keys[slot], 0ull, vk);
        if (prev == 0ull) ++newk;
        if (prev == 0ull || prev == vk) { found = true; break; }
        slot = (slot + 1) & k.table_mask;
      }
      if (!found) { atomicOr(&cnt->overflow, 1u); st = 3u; }   // store full: the cluster still updates the dGraph
      else {
        c.slot[gi] = slot;
        const unsigned long long pr = ((unsigned long long)((1u << 20) - min(c.size[gi], (1u << 20) - 1u)) << 20) | (unsigned long long)(gi + 1u);
        if (atomicMax(&s.owner[slot], pr) != 0ull) ++dup;
        ++marked;
      }
    }
    c.state[gi] = st;
  }
  // the partition's share of the update's counters
  {
    uint32_t t0, t1, t2, t3;
    (void)block_excl_scan<4>(kept, L.wsum, &t0);
    (void)block_excl_scan<4>(marked, L.wsum, &t1);
    (void)block_excl_scan<4>(dup, L.wsum, &t2);
    (void)block_excl_scan<4>(newk, L.wsum, &t3);
    if (tid == 0) {
      if (t0) atomicAdd(&cnt->n_clusters_kept, t0);
      if (t1) atomicAdd(&cnt->n_marked, t1);
      if (t2) atomicAdd(&cnt->n_dup, t2);
      if (t3) atomicAdd(&cnt->n_new_keys, t3);
    }
  }
  MKF_STAMP(9);
#ifdef DDDMR_PHASE_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 128) g_mk_stamps[64 + blockIdx.x] = ((unsigned long long)m << 40) | ((unsigned long long)clock64() - part_t0);
#endif
}

// partition launch: 64 partition blocks  |  removePCPtr blocks
__global__ __launch_bounds__(kPartThreads) void k_mkf_groups(MarkParams k, FuseBufs fb, ClusterArrays c, MarkStore s, PointGrid ground,
                                                             PointGrid map, uint32_t n_map, MarkCounters* __restrict__ cnt, uint32_t nb_band,
                                                             uint32_t seg_groups, uint32_t n_part_un) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fuse_lds[];
  __shared__ float4 stage[256];
  if (blockIdx.x >= (uint32_t)kFuseParts) {
    // removePCPtr of the markings the ray tests cleared, on the CUs the 64 partition workgroups leave idle (it touches the
    // dGraph and the lethal flags only, the partitions never do; every block of this launch owns the partitions' 140 KB
    // of LDS, so these run one per CU -- the partitions take 42 us, these are done in 15)
    fuse_unmark_block(k, fb, s, ground, cnt, blockIdx.x - (uint32_t)kFuseParts, gridDim.x - (uint32_t)kFuseParts, nb_band, seg_groups, n_part_un, stage);
    return;
  }
  fuse_partition(k, fb, c, s, ground, map, n_map, cnt, fuse_lds);
}

// ---------------------------------------------------------------------------------------------
// commit launch: the keeper of every claimed voxel stores its generator points (k_mk_commit)  |  dGraph / lethal update of
// the new generator points (k_mk_dgraph); the last block to finish publishes the counters and leaves them zeroed
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mkf_commit_dgraph(MarkParams k, FuseBufs fb, ClusterArrays c, MarkStore s, PointGrid ground,
                                                           MarkCounters* __restrict__ cnt, uint32_t nb_commit, uint32_t nb_band,
                                                           uint32_t seg_groups, uint32_t n_part) {
  __shared__ float4 stage[256];
  const int lane = threadIdx.x & 63;
  const bool fallback = cnt->fallback != 0u;
  if (fallback) {
    // the mark phase is redone on the general route: nothing of this attempt may reach the store
  } else if (blockIdx.x < nb_commit) {
    // one lane per point index = possible cluster; the keepers of a wave draw their pool ranges with ONE atomic (3600
    // keepers drawing one each from the same counter took 40 us: ~12 ns per same-address device-scope atomic)
    const uint32_t gi = blockIdx.x * 256 + threadIdx.x;
    bool keeper = false;
    uint32_t slot = 0, ng = 0, first = 0;
    if (gi < k.n_obs && c.state[gi] == 2u) {
      slot = c.slot[gi];
      const unsigned long long pr = ((unsigned long long)((1u << 20) - min(c.size[gi], (1u << 20) - 1u)) << 20) | (unsigned long long)(gi + 1u);
      if (s.owner[slot] == pr) { keeper = true; ng = c.gen_count[gi]; first = c.gen_first[gi]; }
    }
    const uint32_t incl = wave_incl_scan_u32(ng);
    const uint32_t tot = wave_last(incl);
    uint32_t base = 0;
    if (lane == 0 && tot) base = atomicAdd(&cnt->pool_used, tot);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    uint32_t revived = 0, killed = 0;
    if (keeper) {
      const uint32_t ofs = base + incl - ng;
      const uint32_t was = s.alive[slot];
      if (ofs + ng > k.pool_cap) {
        atomicOr(&cnt->overflow, 2u);
        s.alive[slot] = 0; s.pts_n[slot] = 0;
        killed = was ? 1u : 0u;
      } else {
        for (uint32_t i = 0; i < ng; ++i) {
          const float4 p = fb.gen[first + i];
          s.pool[ofs + i] = make_float4(p.x, p.y, p.z, 0.f);
        }
        s.pts_ofs[slot] = ofs;
        s.pts_n[slot] = ng;
        s.alive[slot] = 1;
        revived = was ? 0u : 1u;
      }
    }
    const unsigned long long br = __ballot(revived != 0u), bk = __ballot(killed != 0u);
    if (lane == 0 && (br | bk)) atomicAdd(&cnt->n_revived, (uint32_t)__popcll(br) - (uint32_t)__popcll(bk));
  } else {
    uint32_t bi = blockIdx.x - nb_commit;
    if (bi < nb_band) {
      // computeMinDistanceFromObstacle2GroundNodes + DynamicGraph::setValue + lethal_map_ (cluster_marking.cpp:66-123), node by node
      splat_band_block<true>(k, s, ground, fb.rg, fb.gen_bands, bi / (seg_groups * n_part), (bi / n_part) % seg_groups, bi % n_part, n_part, stage);
    } else {
      // ... and point by point for the generator points that went into no band
      bi -= nb_band;
      const uint32_t n_gen = cnt->n_groups3, stride = (gridDim.x - nb_commit - nb_band) * 4u;
      const float r = (float)k.inflation + 1e-4f;
      for (uint32_t h = bi * 4u + (threadIdx.x >> 6); h < n_gen; h += stride) {
        const float4 p = fb.gen[h];
        if (!fb.rg.bands || !ball_in_range(ground, fb.rg, p.x, p.y, r)) splat_point_wave<true>(k, s, ground, p, lane);
      }
    }
  }
  // ---- last block out: counters -> host-mapped record, device copy zeroed (the pool fill carries over) ----
  // (only the counters cross workgroups inside this launch: device-scope atomics, read back with sc1 loads).  Two-level
  // ticket: 32 shard counters, then one -- 2000 blocks drawing from ONE counter took 25 us (~12 ns per same-address
  // device-scope atomic, one after the other).
  __shared__ uint32_t last;
  wait_own_memory_ops();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t shard = blockIdx.x & 31u, in_shard = (gridDim.x - shard + 31u) / 32u;
    uint32_t l = 0;
    if (atomicAdd(&fb.ticket[2 + shard], 1u) == in_shard - 1u) {
      fb.ticket[2 + shard] = 0u;
      l = atomicAdd(&fb.ticket[0], 1u) == min(gridDim.x, 32u) - 1u ? 1u : 0u;
    }
    last = l;
  }
  __syncthreads();
  if (!last) return;
  constexpr int kWords = (int)(sizeof(MarkCounters) / sizeof(uint32_t));
  uint32_t* src = reinterpret_cast<uint32_t*>(cnt);
  uint32_t* dst = reinterpret_cast<uint32_t*>(fb.host_out);
  if ((int)threadIdx.x < kWords) {
    const uint32_t v = ld_agent(&src[threadIdx.x]);
    __hip_atomic_store(&dst[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (&src[threadIdx.x] != &cnt->pool_used) src[threadIdx.x] = 0u;
  }
  if (threadIdx.x < kBandMax) { fb.gen_bands.cnt[threadIdx.x] = 0u; fb.unmark_bands.cnt[threadIdx.x] = 0u; }
  if (threadIdx.x == 0) fb.ticket[0] = 0u;
}

}  // namespace dddmr
