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

namespace dddmr {

struct FeedParams {
  double Rbs[9], tbs[3];  // base <- sensor
  double Rgb[9], tgb[3];  // global <- base
  int n;
  float window;           // perception_window_size_
  float height;           // marking_height_
};

struct PerceptionScratch {
  float4* scan_dev = nullptr;
  unsigned long long* keys = nullptr;  // voxel key per slot (EMPTY = ~0)
  double* sums = nullptr;              // [slots][3]
  uint32_t* counts = nullptr;          // [slots]
  uint32_t* n_out = nullptr;
  uint32_t* n_out_host = nullptr;      // pinned
  size_t cap_points = 0;
  size_t cap_slots = 0;
};

constexpr unsigned long long kEmptyKey = ~0ull;

__device__ __forceinline__ uint32_t hash_key(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

__global__ __launch_bounds__(256) void k_feed_insert(FeedParams f, const float4* __restrict__ scan,
                                                     unsigned long long* __restrict__ keys,
                                                     double* __restrict__ sums, uint32_t* __restrict__ counts,
                                                     uint32_t slot_mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f.n) return;
  const float4 s = scan[i];
  if (!(isfinite(s.x) && isfinite(s.y) && isfinite(s.z))) return;
  // pcl::transformPointCloud(cloud, cloud, Affine3d): double multiply-add, float result
  const float x = (float)(f.Rbs[0] * s.x + f.Rbs[1] * s.y + f.Rbs[2] * s.z + f.tbs[0]);
  const float y = (float)(f.Rbs[3] * s.x + f.Rbs[4] * s.y + f.Rbs[5] * s.z + f.tbs[1]);
  const float z = (float)(f.Rbs[6] * s.x + f.Rbs[7] * s.y + f.Rbs[8] * s.z + f.tbs[2]);
  // pcl::PassThrough keeps limit_min <= v <= limit_max
  if (x < -f.window || x > f.window || y < -f.window || y > f.window || z < 0.0f || z > f.height) return;
  // pcl::VoxelGrid: ijk = floor(p * inverse_leaf_size), leaf 0.1f -> inverse 10.0f
  const float inv_leaf = 1.0f / 0.1f;
  const int ix = (int)floorf(x * inv_leaf), iy = (int)floorf(y * inv_leaf), iz = (int)floorf(z * inv_leaf);
  const unsigned long long key = ((unsigned long long)(uint32_t)(ix + (1 << 20)) << 42) |
                                 ((unsigned long long)(uint32_t)(iy + (1 << 20)) << 21) |
                                 (unsigned long long)(uint32_t)(iz + (1 << 20));
  uint32_t slot = hash_key(key) & slot_mask;
  for (uint32_t probe = 0; probe <= slot_mask; ++probe) {
    const unsigned long long prev = atomicCAS(&keys[slot], kEmptyKey, key);
    if (prev == kEmptyKey || prev == key) {
      atomicAdd(&sums[3 * (size_t)slot + 0], (double)x);
      atomicAdd(&sums[3 * (size_t)slot + 1], (double)y);
      atomicAdd(&sums[3 * (size_t)slot + 2], (double)z);
      atomicAdd(&counts[slot], 1u);
      return;
    }
    slot = (slot + 1) & slot_mask;
  }
}

__global__ __launch_bounds__(256) void k_feed_emit(FeedParams f, const unsigned long long* __restrict__ keys,
                                                   const double* __restrict__ sums,
                                                   const uint32_t* __restrict__ counts, uint32_t n_slots,
                                                   float4* __restrict__ out, uint32_t* __restrict__ n_out) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_slots) return;
  if (keys[slot] == kEmptyKey) return;
  const double n = (double)counts[slot];
  const float cx = (float)(sums[3 * (size_t)slot + 0] / n);
  const float cy = (float)(sums[3 * (size_t)slot + 1] / n);
  const float cz = (float)(sums[3 * (size_t)slot + 2] / n);
  const float gx = (float)(f.Rgb[0] * cx + f.Rgb[1] * cy + f.Rgb[2] * cz + f.tgb[0]);
  const float gy = (float)(f.Rgb[3] * cx + f.Rgb[4] * cy + f.Rgb[5] * cz + f.tgb[1]);
  const float gz = (float)(f.Rgb[6] * cx + f.Rgb[7] * cy + f.Rgb[8] * cz + f.tgb[2]);
  const uint32_t o = atomicAdd(n_out, 1u);
  out[o] = make_float4(gx, gy, gz, 0.f);
}

inline int perception_alloc(PerceptionScratch& s, size_t max_points) {
  s.cap_points = max_points;
  size_t slots = 1024;
  while (slots < 2 * max_points) slots <<= 1;
  s.cap_slots = slots;
  if (hipMalloc(&s.scan_dev, max_points * sizeof(float4)) != hipSuccess) return -1;
  if (hipMalloc(&s.keys, slots * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (hipMalloc(&s.sums, slots * 3 * sizeof(double)) != hipSuccess) return -1;
  if (hipMalloc(&s.counts, slots * sizeof(uint32_t)) != hipSuccess) return -1;
  if (hipMalloc(&s.n_out, sizeof(uint32_t)) != hipSuccess) return -1;
  if (hipHostMalloc(&s.n_out_host, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return -1;
  return 0;
}

inline void perception_free(PerceptionScratch& s) {
  if (s.scan_dev) (void)hipFree(s.scan_dev);
  if (s.keys) (void)hipFree(s.keys);
  if (s.sums) (void)hipFree(s.sums);
  if (s.counts) (void)hipFree(s.counts);
  if (s.n_out) (void)hipFree(s.n_out);
  if (s.n_out_host) (void)hipHostFree(s.n_out_host);
  s = PerceptionScratch();
}

// scan_host: pinned float4[n] in the sensor frame.  Output: out_dev (global frame).
inline int perception_feed(PerceptionScratch& s, const FeedParams& f, const float4* scan_host,
                           float4* out_dev, hipStream_t stream, uint32_t* n_out) {
  *n_out = 0;
  if (f.n == 0) return 0;
  size_t slots = 1024;
  while (slots < 2 * (size_t)f.n) slots <<= 1;
  if (slots > s.cap_slots) return -2;
  if (hipMemcpyAsync(s.scan_dev, scan_host, (size_t)f.n * sizeof(float4), hipMemcpyHostToDevice, stream) != hipSuccess) return -3;
  if (hipMemsetAsync(s.keys, 0xFF, slots * sizeof(unsigned long long), stream) != hipSuccess) return -3;
  if (hipMemsetAsync(s.sums, 0, slots * 3 * sizeof(double), stream) != hipSuccess) return -3;
  if (hipMemsetAsync(s.counts, 0, slots * sizeof(uint32_t), stream) != hipSuccess) return -3;
  if (hipMemsetAsync(s.n_out, 0, sizeof(uint32_t), stream) != hipSuccess) return -3;
  hipLaunchKernelGGL(k_feed_insert, dim3((f.n + 255) / 256), dim3(256), 0, stream, f, s.scan_dev, s.keys,
                     s.sums, s.counts, (uint32_t)(slots - 1));
  hipLaunchKernelGGL(k_feed_emit, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, f, s.keys,
                     s.sums, s.counts, (uint32_t)slots, out_dev, s.n_out);
  if (hipMemcpyAsync(s.n_out_host, s.n_out, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) return -3;
  if (hipStreamSynchronize(stream) != hipSuccess) return -4;
  if (hipGetLastError() != hipSuccess) return -5;
  *n_out = *s.n_out_host;
  return 0;
}

}  // namespace dddmr
