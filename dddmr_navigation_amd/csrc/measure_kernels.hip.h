// measure_kernels.hip.h -- measurement aids: the stream ceiling the roofline's
// second denominator comes from.  Not on the tick's path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dddmr {

typedef float v4f __attribute__((ext_vector_type(4)));

// Grid-stride 16-byte copy, four independent loads in flight per lane and pass, non-temporal (streamed once:
// no point in keeping the lines in L2 / MALL).
__global__ __launch_bounds__(256) void k_stream_copy(const float4* __restrict__ src4, float4* __restrict__ dst4, size_t n) {
  const v4f* __restrict__ src = reinterpret_cast<const v4f*>(src4);
  v4f* __restrict__ dst = reinterpret_cast<v4f*>(dst4);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const v4f a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
              c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i);
    __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride);
    __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// Read-only stream (what a gather/compare kernel like k_score could at best approach): every lane
// folds what it reads into one word so the loads cannot be dropped; one store per lane.
__global__ __launch_bounds__(256) void k_stream_read(const float4* __restrict__ src4, float* __restrict__ sink, size_t n) {
  const v4f* __restrict__ src = reinterpret_cast<const v4f*>(src4);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const v4f a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
              c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    acc += (a.x + b.y) + (c.z + d.w);
  }
  for (; i < n; i += stride) acc += src[i].x;
  sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

}  // namespace dddmr
