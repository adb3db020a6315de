// Micro-benchmark: issue rate of v_fma_f64 against v_mul_f64 + v_add_f64 on gfx950 (why the rollout's own sincos keeps
// separate multiplies and adds: tools/r02_exp notes in DESIGN.md 4).  Every lane runs kIters rounds of 8 independent
// chains; build: hipcc -O3 --offload-arch=gfx950 -o f64_rate f64_rate.hip ; run: ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
constexpr int kIters = 4096;
template <int kMode>
__global__ __launch_bounds__(256) void k(double* out, double a, double b) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a + (double)(threadIdx.x + i);
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (kMode == 0) x[i] = __builtin_fma(x[i], a, b);        // 1 v_fma_f64
      else if (kMode == 1) x[i] = x[i] * a + b;                // v_mul_f64 + v_add_f64 (contraction off)
      else if (kMode == 2) x[i] = x[i] * a;                    // 1 v_mul_f64
      else x[i] = x[i] + b;                                    // 1 v_add_f64
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int kMode>
static void run(const char* name, int ops_per_round, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {       // 256-lane workgroups: 1, 2, 4, 8 waves per SIMD
    const int blocks = 256 * wg_per_cu;
    hipLaunchKernelGGL(k<kMode>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<kMode>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)kIters * 8 * ops_per_round * wg_per_cu;   // 4 waves / 4 SIMDs per workgroup
    printf("%-22s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name,
           wg_per_cu, ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
  }
}
int main() {
  double* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  run<0>("v_fma_f64", 1, out);
  run<1>("v_mul_f64+v_add_f64", 2, out);
  run<2>("v_mul_f64", 1, out);
  run<3>("v_add_f64", 1, out);
  hipFree(out);
  return 0;
}
