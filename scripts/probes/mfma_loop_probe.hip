// What costs matrix time in a one-wave-per-SIMD MFMA loop?  The K loop of the small-map kernel (csrc/conv_plane.hip: NP = 5 pixel tiles x
// MW = 2 channel groups = 10 accumulators, per k-step 5 LDS operand reads + one 8-byte weight load + 10 MFMAs) rebuilt from nothing,
// one ingredient at a time:
//   A  MFMAs on loop-invariant registers
//   B  + the pixel operands come from LDS, read one k-step ahead, one read behind every MW MFMAs
//   C  + the weight operands come from global memory through a ring of 8 k-steps
//   D  B + C
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_loop_probe.hip -o /tmp/mlp && /tmp/mlp
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int NP = 5, MW = 2, RING = 8, STEPS = 18;      // a chunk of 18 k-steps, unrolled

template <int LDS, int WGT>
__global__ void __launch_bounds__(256) loop(const float* __restrict__ w, float* out, int chunks, float seed) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed + (i & 255) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float* lp = lds + (lane & 15) + 72 * (lane >> 4);
  const float* wl = w + (size_t)blockIdx.x * 65536 + lane * 2;
  f32x4 acc[MW][NP];
  for (int j = 0; j < MW; ++j) for (int p = 0; p < NP; ++p) acc[j][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  float b[2][NP];
  for (int p = 0; p < NP; ++p) b[0][p] = LDS ? lp[16 * p] : seed + p, b[1][p] = seed;
  f32x2 ring[RING];
  for (int i = 0; i < RING; ++i) ring[i] = WGT ? *reinterpret_cast<const f32x2*>(wl + 128 * i) : f32x2{seed, seed * 0.5f};
  for (int c = 0; c < chunks; ++c) {
    const float* wc = wl + 128 * STEPS * (c & 15);
#pragma unroll
    for (int ks = 0; ks < STEPS; ++ks) {
      if (WGT) ring[(ks + RING - 1) % RING] = *reinterpret_cast<const f32x2*>(wc + 128 * (ks + RING - 1));
      const f32x2 wv = ring[ks % RING];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int j = 0; j < MW; ++j) acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ks & 1][p], wv[j], acc[j][p], 0, 0, 0);
        if (LDS) b[(ks + 1) & 1][p] = lp[16 * p + 8 * ((ks + 1) % 9) + 1152 * ((ks + 1) / 9)];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < MW; ++j) for (int p = 0; p < NP; ++p) s += acc[j][p][0] + acc[j][p][1] + acc[j][p][2] + acc[j][p][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LDS, int WGT>
static void run(const float* w, float* d, const char* what) {
  const int chunks = 400;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((loop<LDS, WGT>), dim3(256), dim3(256), 0, 0, w, d, chunks, 1.0f);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((loop<LDS, WGT>), dim3(256), dim3(256), 0, 0, w, d, chunks, 1.0f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 10.0 * 256 * 4 * (double)chunks * STEPS * NP * MW * 2048.0;
  printf("%-70s %7.1f TFLOP/s\n", what, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *w, *d;
  (void)hipMalloc(&w, 256ull * 65536 * 4 + (1 << 20)); (void)hipMalloc(&d, 1 << 22);
  (void)hipMemset(w, 0, 256ull * 65536 * 4 + (1 << 20));
  run<0, 0>(w, d, "A  10 accumulators, loop-invariant operands");
  run<1, 0>(w, d, "B  pixel operands from LDS (one read behind every 2 MFMAs, a step ahead)");
  run<0, 1>(w, d, "C  weight operands from global memory (ring of 8 k-steps)");
  run<1, 1>(w, d, "D  both");
  return 0;
}
