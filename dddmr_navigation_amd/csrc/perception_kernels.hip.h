// perception_kernels.hip.h -- local-mode perception feed as a HIP voxel-hash.
//
// Replaces MultiLayerSpinningLidar::cbSensor
// (dddmr_perception_3d/plugins/multilayer_spinning_lidar.cpp:177-281) for the
// local planner: sensor->base transform (:232-233), PassThrough crop
// |x|,|y| <= window, 0 <= z <= marking_height (:240-251), VoxelGrid centroid
// downsample with a 0.1 m leaf (:253-256), base->global transform (:264-269).
// The result is written straight into the context's aggregate-observation
// buffer (StackedPerception::aggregateObservations, src/stacked_perception.cpp:128-140)
// so the scorer's binning pass reads it without a host round trip.
//
// Voxel membership is PCL's: voxel = floor(p * (1/leaf)) per axis in float.  The
// per-voxel centroid is accumulated with double atomics (PCL accumulates in
// float in an unspecified order, so only ~1e-6 agreement is meaningful).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <deque>

// The reference is an x86-64 build without FMA contraction: every multiply and add below
// rounds separately, in float and in double (hipcc's default would fuse them).
#pragma clang fp contract(off)

namespace dddmr {

struct FeedParams {
  double Rbs[9], tbs[3];  // base <- sensor
  double Rgb[9], tgb[3];  // global <- base
  int n;
  float window;           // perception_window_size_
  float height;           // marking_height_
};

struct FeedResult {          // host-mapped: written by the last k_feed_emit workgroup
  uint32_t n_out;
  uint32_t seq;              // stored last (system-scope release); the host polls it
};

struct PerceptionScratch {
  float* stage_dev = nullptr;          // device address of `stage` (raw scan records, stride_floats apart)
  uint32_t* claimed = nullptr;         // table slots claimed by this scan's voxels
  unsigned char* table = nullptr;      // [keys 8B | sums 3x8B | counts 4B] x slots, one memset clears it
  uint32_t* counters = nullptr;        // [0] n_out, [1] ticket, [2] claimed slots
  FeedResult* res_host = nullptr;      // pinned + mapped
  FeedResult* res_dev = nullptr;
  float* stage = nullptr;              // pinned staging for the raw scan
  size_t cap_points = 0;
  size_t cap_slots = 0;
  uint32_t seq = 0;
  // stitcher (cbSensor :185-200): the last stitcher_num raw scans, oldest first, packed xyz in `stage`
  int stitcher_num = 0;
  std::deque<uint32_t> stitched;       // point counts of the queued scans
};

// key 0 = empty slot (a real key always has bit 63 set)
__device__ __forceinline__ uint32_t hash_key(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

__global__ __launch_bounds__(256) void k_feed_insert(FeedParams f, const float* __restrict__ scan, int stride_floats,
                                                     unsigned long long* __restrict__ keys,
                                                     double* __restrict__ sums, uint32_t* __restrict__ counts,
                                                     uint32_t slot_mask, uint32_t* __restrict__ claimed,
                                                     uint32_t* __restrict__ counters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f.n) return;
  const float* sp = scan + (size_t)i * stride_floats;
  const float sx = sp[0], sy = sp[1], sz = sp[2];
  if (!(isfinite(sx) && isfinite(sy) && isfinite(sz))) return;
  // pcl::transformPointCloud(cloud, cloud, Affine3d): double multiply-add, float result
  const float x = (float)(f.Rbs[0] * sx + f.Rbs[1] * sy + f.Rbs[2] * sz + f.tbs[0]);
  const float y = (float)(f.Rbs[3] * sx + f.Rbs[4] * sy + f.Rbs[5] * sz + f.tbs[1]);
  const float z = (float)(f.Rbs[6] * sx + f.Rbs[7] * sy + f.Rbs[8] * sz + f.tbs[2]);
  // pcl::PassThrough keeps limit_min <= v <= limit_max
  if (x < -f.window || x > f.window || y < -f.window || y > f.window || z < 0.0f || z > f.height) return;
  // pcl::VoxelGrid: ijk = floor(p * inverse_leaf_size), leaf 0.1f -> inverse 10.0f
  const float inv_leaf = 1.0f / 0.1f;
  const int ix = (int)floorf(x * inv_leaf), iy = (int)floorf(y * inv_leaf), iz = (int)floorf(z * inv_leaf);
  const unsigned long long key = (1ull << 63) | ((unsigned long long)((uint32_t)(ix + (1 << 20)) & 0x1FFFFFu) << 42) |
                                 ((unsigned long long)((uint32_t)(iy + (1 << 20)) & 0x1FFFFFu) << 21) |
                                 (unsigned long long)((uint32_t)(iz + (1 << 20)) & 0x1FFFFFu);
  uint32_t slot = hash_key(key) & slot_mask;
  for (uint32_t probe = 0; probe <= slot_mask; ++probe) {
    const unsigned long long prev = atomicCAS(&keys[slot], 0ull, key);
    if (prev == 0ull || prev == key) {
      if (prev == 0ull) claimed[atomicAdd(&counters[2], 1u)] = slot;   // first point of a voxel: list its slot for the emit pass
      atomicAdd(&sums[3 * (size_t)slot + 0], (double)x);
      atomicAdd(&sums[3 * (size_t)slot + 1], (double)y);
      atomicAdd(&sums[3 * (size_t)slot + 2], (double)z);
      atomicAdd(&counts[slot], 1u);
      return;
    }
    slot = (slot + 1) & slot_mask;
  }
}

__global__ __launch_bounds__(256) void k_feed_emit(FeedParams f, unsigned long long* __restrict__ keys,
                                                   double* __restrict__ sums,
                                                   uint32_t* __restrict__ counts, const uint32_t* __restrict__ claimed,
                                                   float4* __restrict__ out, uint32_t* __restrict__ counters,
                                                   FeedResult* __restrict__ res, uint32_t seq) {
  // one lane per occupied voxel (the slots k_feed_insert listed), not per table slot
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  const bool occ = idx < counters[2];
  const uint32_t slot = occ ? claimed[idx] : 0u;
  // wave-aggregated append: one atomic per wave
  const unsigned long long mask = __ballot(occ);
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (mask) {
    if (lane == (__ffsll((long long)mask) - 1)) base = atomicAdd(&counters[0], (uint32_t)__popcll(mask));
    base = __shfl(base, __ffsll((long long)mask) - 1, 64);
  }
  if (occ) {
    const double n = (double)counts[slot];
    const float cx = (float)(sums[3 * (size_t)slot + 0] / n);
    const float cy = (float)(sums[3 * (size_t)slot + 1] / n);
    const float cz = (float)(sums[3 * (size_t)slot + 2] / n);
    const float gx = (float)(f.Rgb[0] * cx + f.Rgb[1] * cy + f.Rgb[2] * cz + f.tgb[0]);
    const float gy = (float)(f.Rgb[3] * cx + f.Rgb[4] * cy + f.Rgb[5] * cz + f.tgb[1]);
    const float gz = (float)(f.Rgb[6] * cx + f.Rgb[7] * cy + f.Rgb[8] * cz + f.tgb[2]);
    const uint32_t o = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    out[o] = make_float4(gx, gy, gz, 0.f);
    // leave the table empty for the next scan (saves a 2 MB memset per call)
    keys[slot] = 0ull;
    sums[3 * (size_t)slot + 0] = 0.0;
    sums[3 * (size_t)slot + 1] = 0.0;
    sums[3 * (size_t)slot + 2] = 0.0;
    counts[slot] = 0u;
  }
  // last workgroup publishes the count to the host (device-scope ticket; the count is
  // only touched by device-scope atomics)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(&counters[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      res->n_out = __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      counters[0] = 0;         // next call
      counters[1] = 0;
      counters[2] = 0;
      __threadfence_system();
      __hip_atomic_store(&res->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

inline size_t feed_table_bytes(size_t slots) { return slots * (8 + 24 + 4) + 64; }

inline int perception_alloc(PerceptionScratch& s, size_t max_points) {
  s.cap_points = max_points;
  size_t slots = 1024;
  while (slots < 2 * max_points) slots <<= 1;
  s.cap_slots = slots;
  if (hipMalloc(&s.claimed, max_points * sizeof(uint32_t)) != hipSuccess) return -1;
  if (hipMalloc(&s.table, feed_table_bytes(slots)) != hipSuccess) return -1;
  if (hipMalloc(&s.counters, 4 * sizeof(uint32_t)) != hipSuccess) return -1;
  if (hipMemset(s.table, 0, feed_table_bytes(slots)) != hipSuccess) return -1;   // k_feed_emit keeps it clean afterwards
  if (hipMemset(s.counters, 0, 4 * sizeof(uint32_t)) != hipSuccess) return -1;
  if (hipHostMalloc(&s.res_host, sizeof(FeedResult), hipHostMallocMapped) != hipSuccess) return -1;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&s.res_dev), s.res_host, 0) != hipSuccess) return -1;
  // the raw scan is read by k_feed_insert straight from this pinned, device-mapped buffer
  // (no separate H2D copy: the kernel's coalesced reads stream it over PCIe)
  if (hipHostMalloc(&s.stage, max_points * 4 * sizeof(float), hipHostMallocMapped) != hipSuccess) return -1;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&s.stage_dev), s.stage, 0) != hipSuccess) return -1;
  s.res_host->n_out = 0;
  s.res_host->seq = 0;
  return 0;
}

inline void perception_free(PerceptionScratch& s) {
  if (s.claimed) (void)hipFree(s.claimed);
  if (s.table) (void)hipFree(s.table);
  if (s.counters) (void)hipFree(s.counters);
  if (s.res_host) (void)hipHostFree(s.res_host);
  if (s.stage) (void)hipHostFree(s.stage);
  s = PerceptionScratch();
}

// scan: caller's records (stride_bytes apart, x y z first).  Output: out_dev (global frame).
// With a stitcher depth N > 0 the scan joins the queue of the last N raw scans (the oldest one leaves when
// the queue is full) and the WHOLE queue, oldest first, is fed through the current transforms -- cbSensor's
// pcl_stitcher_ deque (multilayer_spinning_lidar.cpp:185-200).
inline int perception_feed(PerceptionScratch& s, FeedParams f, const float* scan, size_t stride_bytes,
                           float4* out_dev, hipStream_t stream, uint32_t* n_out) {
  *n_out = 0;
  int stride_floats;
  if (s.stitcher_num > 0) {
    if ((int)s.stitched.size() >= s.stitcher_num) {            // pop_front: the later scans move up
      const size_t drop = s.stitched.front();
      s.stitched.pop_front();
      size_t rest = 0;
      for (uint32_t c : s.stitched) rest += c;
      std::memmove(s.stage, s.stage + 3 * drop, rest * 3 * sizeof(float));
    }
    size_t have = 0;
    for (uint32_t c : s.stitched) have += c;
    if (have + (size_t)f.n > s.cap_points) return -2;
    const size_t sf = stride_bytes / 4;
    for (size_t i = 0; i < (size_t)f.n; ++i) {
      s.stage[3 * (have + i) + 0] = scan[i * sf + 0];
      s.stage[3 * (have + i) + 1] = scan[i * sf + 1];
      s.stage[3 * (have + i) + 2] = scan[i * sf + 2];
    }
    s.stitched.push_back((uint32_t)f.n);
    f.n = (int)(have + (size_t)f.n);
    stride_floats = 3;
  }
  if (f.n == 0) return 0;
  size_t slots = 1024;
  while (slots < 2 * (size_t)f.n) slots <<= 1;
  if (slots > s.cap_slots) return -2;
  // stage the raw records in pinned memory: packed xyz(i) records go as they are,
  // wider ones (PCL: 16/32 bytes) are narrowed to 12 bytes on the way
  if (s.stitcher_num > 0) {
    // (already staged above)
  } else if (stride_bytes == 12 || stride_bytes == 16) {
    stride_floats = (int)(stride_bytes / 4);
    std::memcpy(s.stage, scan, (size_t)f.n * stride_bytes);
  } else {
    stride_floats = 3;
    const size_t sf = stride_bytes / 4;
    for (size_t i = 0; i < (size_t)f.n; ++i) {
      s.stage[3 * i + 0] = scan[i * sf + 0];
      s.stage[3 * i + 1] = scan[i * sf + 1];
      s.stage[3 * i + 2] = scan[i * sf + 2];
    }
  }
  // fixed layout over the full-capacity table; a call only uses its first `slots` entries
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(s.table);
  double* sums = reinterpret_cast<double*>(s.table + s.cap_slots * 8);
  uint32_t* counts = reinterpret_cast<uint32_t*>(s.table + s.cap_slots * 32);
  const uint32_t seq = ++s.seq ? s.seq : ++s.seq;
  hipLaunchKernelGGL(k_feed_insert, dim3((f.n + 255) / 256), dim3(256), 0, stream, f, s.stage_dev, stride_floats, keys,
                     sums, counts, (uint32_t)(slots - 1), s.claimed, s.counters);
  hipLaunchKernelGGL(k_feed_emit, dim3((unsigned)((f.n + 255) / 256)), dim3(256), 0, stream, f, keys, sums, counts,
                     s.claimed, out_dev, s.counters, s.res_dev, seq);
  if (hipGetLastError() != hipSuccess) return -5;
  // poll the host-mapped sequence number (bounded), then make sure the stream is idle
  volatile uint32_t* seq_p = &s.res_host->seq;
  bool seen = false;
  for (uint64_t spins = 0; spins < (1ull << 26); ++spins) {
    if (*seq_p == seq) { seen = true; break; }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  if (!seen && hipStreamSynchronize(stream) != hipSuccess) return -4;
  *n_out = s.res_host->n_out;
  return 0;
}

// ---------------------------------------------------------------------------
// PathBlockedStrategy::selfMark
// (dddmr_perception_3d/plugins/path_blocked_strategy.cpp:56-100): which forward points of
// the prune-plan cloud have an observation point within check_radius.  The reference
// builds a second kd-tree on the aggregate observation for M radius searches (:68-83);
// here every cloud point is tested against the forward plan points kept in LDS (a
// bounding box of the plan, grown by the radius, rejects nearly all of them first).
// FLANN's L2_Simple float distance, strict `<` against static_cast<float>(r * r).
// ---------------------------------------------------------------------------
constexpr int kBlockedMaxPlan = 1024;   // pcl_prune_plan_ points (the nearest pose appears twice)
struct BlockedParams {
  int n_points;
  int m;            // plan points
  float r2;
  float lo[3], hi[3];
};

__global__ __launch_bounds__(256) void k_path_blocked(BlockedParams b, const float4* __restrict__ cloud,
                                                      const float4* __restrict__ plan_xyzi,
                                                      uint32_t* __restrict__ flags /* [(m + 31) / 32] */) {
  __shared__ float4 plan[kBlockedMaxPlan];
  __shared__ uint32_t hit[kBlockedMaxPlan / 32];
  for (int i = threadIdx.x; i < b.m; i += blockDim.x) plan[i] = plan_xyzi[i];
  for (int i = threadIdx.x; i < kBlockedMaxPlan / 32; i += blockDim.x) hit[i] = 0u;
  __syncthreads();
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n_points; i += stride) {
    const float4 p = cloud[i];
    if (!(p.x >= b.lo[0] && p.x <= b.hi[0] && p.y >= b.lo[1] && p.y <= b.hi[1] && p.z >= b.lo[2] && p.z <= b.hi[2]))
      continue;
    for (int j = 0; j < b.m; ++j) {
      const float4 q = plan[j];
      if (q.w < 0.f) continue;                    // backward of the robot (:80-81)
      float d = q.x - p.x;
      float r = d * d;
      d = q.y - p.y;
      r = r + d * d;
      d = q.z - p.z;
      r = r + d * d;
      if (r < b.r2) atomicOr(&hit[j >> 5], 1u << (j & 31));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (b.m + 31) / 32; i += blockDim.x)
    if (hit[i]) atomicOr(&flags[i], hit[i]);
}

}  // namespace dddmr
