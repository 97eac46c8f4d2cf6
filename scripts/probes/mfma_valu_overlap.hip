// Does f32 VALU work overlap with v_mfma_f32_16x16x4_f32 on one SIMD?  One wave per SIMD (1024 waves), a loop of 16 independent MFMAs
// with M independent v_add_f32 / v_fma_f32 threaded between them; prints cycles per loop trip for M = 0, 8, 16, 32, 64.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int M, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) probe(float* out, int iters, float seed) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x + i;
  float a = seed * 0.5f + threadIdx.x, b = seed + 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[p], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < M / 16; ++j) {
        const int k = (p * (M / 16) + j) & 7;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[k]) : "v"(v[k]), "v"(v[(k + 1) & 7]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int M, int WAVES>
static void run(float* d, const char* what) {
  const int iters = 2000, blocks = 256 * 4 / WAVES * (WAVES > 4 ? 2 : 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<M, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, 10, 1.0f);
  // the chip needs ~25 ms of load to reach its steady clocks (round 3: a single 0.4 ms launch -- rounds 1-2 -- reads the ramp clock, ~2.17 GHz)
  for (int w = 0; w < (getenv("FN2_PROBE_WARM") ? atoi(getenv("FN2_PROBE_WARM")) : 0); ++w)
    hipLaunchKernelGGL((probe<M, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, iters, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<M, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_trip_us = ms * 1e3 / iters;
  printf("%s: %d waves/block, %d blocks, %2d VALU per 16 MFMAs: %.3f us per trip = %.0f cycles at 2.4 GHz (16 MFMAs alone = 512)\n", what, WAVES, blocks, M,
         per_trip_us, per_trip_us * 2400);
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 24);
  run<0, 4>(d, "1 wave/SIMD ");
  run<16, 4>(d, "1 wave/SIMD ");
  run<32, 4>(d, "1 wave/SIMD ");
  run<64, 4>(d, "1 wave/SIMD ");
  run<128, 4>(d, "1 wave/SIMD ");
  run<0, 8>(d, "2 waves/SIMD");
  run<32, 8>(d, "2 waves/SIMD");
  run<64, 8>(d, "2 waves/SIMD");
  run<128, 8>(d, "2 waves/SIMD");
  return 0;
}
