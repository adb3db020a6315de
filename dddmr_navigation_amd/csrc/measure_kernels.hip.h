// measure_kernels.hip.h -- measurement aids: the stream ceiling the roofline's
// second denominator comes from.  Not on the tick's path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dddmr {

// Grid-stride float4 copy, four independent 16-byte loads in flight per lane and pass.
__global__ __launch_bounds__(256) void k_stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// Read-only stream (what a gather/compare kernel like k_score could at best approach): every lane
// folds what it reads into one word so the loads cannot be dropped; one store per lane.
__global__ __launch_bounds__(256) void k_stream_read(const float4* __restrict__ src, float* __restrict__ sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc += (a.x + b.y) + (c.z + d.w);
  }
  for (; i < n; i += stride) acc += src[i].x;
  sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

}  // namespace dddmr
