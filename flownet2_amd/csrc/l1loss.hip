// L1Loss (plain L1 and the per-pixel L2 "EPE" form) for gfx950.
//
// Replaces L1LossLayer::Forward_gpu / Backward_gpu (reference: src/caffe/layers/l1loss_layer.cu:67-188,
// composition in l1loss_layer.cpp:11-90).  The reference runs >= 10 element-wise kernels, an Eltwise,
// two Power layers and a 1x1 convolution, 2-5 cudaDeviceSynchronize and two blocking cublasSdot per
// call.  Here the forward is ONE fused streaming pass (difference, NaN mask, plateau, per-pixel
// norm, wave64 DPP reduction -> LDS -> one partial per block) plus a 1-block finalise kernel; the
// loss and the normalisation coefficient stay on the device.  Reduction order is fixed (no
// atomics), so the loss is bit-reproducible run to run.
#include "fn2_common.hpp"

namespace fn2 {

constexpr int kL1Threads = 256;
constexpr int kL1MaxBlocks = 1024;

struct L1Args {
  int N, C, H, W;
  int l2_per_location, prescale, normalize;
  float epsilon, plateau;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// One block-level reduction of two doubles; result valid in thread 0.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* lds /* [2*4] */) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { lds[wid] = a; lds[4 + wid] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = lds[0] + lds[1] + lds[2] + lds[3];
    b = lds[4] + lds[5] + lds[6] + lds[7];
  }
}

// partial[2*b] = sum of per-location loss terms, partial[2*b+1] = number of non-NaN entries.
__global__ void __launch_bounds__(kL1Threads) l1loss_fwd_partial(const float* __restrict__ b0, const float* __restrict__ b1,
                                                                 double* __restrict__ partial, L1Args a) {
  __shared__ double lds[8];
  const size_t hw = (size_t)a.H * a.W;
  double dot = 0.0, nvalid = 0.0;
  if (a.l2_per_location) {
    const float wgt = a.prescale ? 1.f / (float)a.C : 1.f;    // l1loss_layer.cpp:47-51
    const float plat2 = a.plateau * a.plateau;                // l1loss_layer.cu:104
    const long long total = (long long)a.N * hw;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
      const size_t n = q / hw, s = q % hw;
      float acc = 0.f;
      int cnt = 0;
      for (int c = 0; c < a.C; ++c) {
        const size_t i = (n * a.C + c) * hw + s;
        float d = b1 ? (b0[i] - b1[i]) : b0[i];               // Eltwise coeff (+1,-1), cpp:19-26
        const bool ok = (d == d);                             // FindNotNaNs cu:20-24
        cnt += ok;
        d = ok ? d : 0.f;                                     // KillMasked cu:95-96
        acc += wgt * (d * d);                                 // Power^2 cu:99, 1x1 conv cu:100
      }
      if (a.plateau > 0.f && fabsf(acc) < plat2) acc = 0.f;   // cu:103-114
      dot += (double)sqrtf(acc + a.epsilon);                  // Power^0.5 shift eps cu:117, dot with ones cu:119
      nvalid += (double)cnt;
    }
  } else {
    const long long total = (long long)a.N * a.C * hw;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      float d = b1 ? (b0[i] - b1[i]) : b0[i];
      const bool ok = (d == d);
      nvalid += ok ? 1.0 : 0.0;
      bool keep = ok;
      if (a.plateau > 0.f && fabsf(d) < a.plateau) keep = false;   // MaskPlateauValues cu:52-56
      d = keep ? d : 0.f;                                          // KillMasked cu:132-134
      const float sign = d > 0.f ? 1.f : -1.f;                     // ComputeSign cu:11-15
      dot += (double)(d * sign);                                   // cu:139
    }
  }
  block_sum2(dot, nvalid, lds);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = dot;
    partial[2 * blockIdx.x + 1] = nvalid;
  }
}

// ws[0] = loss, ws[1] = normalize_coeff; loss_out[0] = loss.
__global__ void __launch_bounds__(kL1Threads) l1loss_fwd_final(const double* __restrict__ partial, int nblocks,
                                                               float* __restrict__ ws, float* __restrict__ loss_out, L1Args a) {
  __shared__ double lds[8];
  double dot = 0.0, nvalid = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    dot += partial[2 * b];
    nvalid += partial[2 * b + 1];
  }
  block_sum2(dot, nvalid, lds);
  if (threadIdx.x == 0) {
    const float norm = a.normalize ? (float)nvalid / (float)a.C : (float)a.N;   // cu:86-91
    const float loss = (float)dot / norm;                                       // cu:141
    ws[0] = loss;
    ws[1] = norm;
    if (loss_out) loss_out[0] = loss;
  }
}

__global__ void __launch_bounds__(kL1Threads) l1loss_bwd(const float* __restrict__ b0, const float* __restrict__ b1,
                                                         const float* __restrict__ ws, float top_diff,
                                                         float* __restrict__ d0, float* __restrict__ d1, L1Args a) {
  const size_t hw = (size_t)a.H * a.W;
  const float alpha = top_diff / ws[1];                        // cu:155
  if (a.l2_per_location) {
    const float wgt = a.prescale ? 1.f / (float)a.C : 1.f;
    const float plat2 = a.plateau * a.plateau;
    const long long total = (long long)a.N * hw;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
      const size_t n = q / hw, s = q % hw;
      float acc = 0.f;
      for (int c = 0; c < a.C; ++c) {
        const size_t i = (n * a.C + c) * hw + s;
        float d = b1 ? (b0[i] - b1[i]) : b0[i];
        d = (d == d) ? d : 0.f;
        acc += wgt * (d * d);
      }
      bool kill = false;
      if (a.plateau > 0.f && fabsf(acc) < plat2) { acc = 0.f; kill = true; }
      const float e = sqrtf(acc + a.epsilon);
      // sqrt_output diff = alpha (cu:158); Power backward, general branch (power_layer.cu:62-74):
      // top_data / (x + shift) * diff_scale(0.5) * top_diff; plateau mask cu:162-166.
      const float ds = kill ? 0.f : (e / (acc + a.epsilon)) * 0.5f * alpha;
      for (int c = 0; c < a.C; ++c) {
        const size_t i = (n * a.C + c) * hw + s;
        float d = b1 ? (b0[i] - b1[i]) : b0[i];
        const bool ok = (d == d);
        d = ok ? d : 0.f;
        float g = (2.f * d) * (wgt * ds);                       // conv backward, square backward (power_layer.cu:48-52)
        g = ok ? g : 0.f;                                       // KillMasked cu:179-180
        d0[i] = g;                                              // Eltwise backward, coeff +1
        if (d1) d1[i] = -g;                                     // coeff -1
      }
    }
  } else {
    const long long total = (long long)a.N * a.C * hw;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      float d = b1 ? (b0[i] - b1[i]) : b0[i];
      bool keep = (d == d);
      if (a.plateau > 0.f && fabsf(d) < a.plateau) keep = false;
      d = keep ? d : 0.f;
      const float sign = d > 0.f ? 1.f : -1.f;
      float g = alpha * sign;                                   // cu:175-176
      g = keep ? g : 0.f;                                       // cu:179-180
      d0[i] = g;
      if (d1) d1[i] = -g;
    }
  }
}

static int l1_args(const fn2_l1loss_params* p, int N, int C, int H, int W, L1Args* a) {
  if (!p) return fail(FN2_ERR_INVALID_ARG, "l1loss: params == NULL");
  if (N < 1 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "l1loss: bad shape [%d,%d,%d,%d]", N, C, H, W);
  a->N = N; a->C = C; a->H = H; a->W = W;
  a->l2_per_location = p->l2_per_location != 0;
  a->prescale = p->l2_prescale_by_channels != 0;
  a->normalize = p->normalize_by_num_entries != 0;
  a->epsilon = p->epsilon;
  a->plateau = p->plateau;
  return FN2_OK;
}

static int l1_blocks(const L1Args& a) {
  const long long work = a.l2_per_location ? (long long)a.N * a.H * a.W : (long long)a.N * a.C * a.H * a.W;
  return (int)blocks_for(work, kL1Threads, kL1MaxBlocks);
}

}  // namespace fn2

using namespace fn2;

// layout: float[0]=loss, float[1]=normalize_coeff, pad to 64 B, then double partial[2*kL1MaxBlocks]
FN2_API size_t fn2_l1loss_workspace_bytes(int, int, int, int) { return 64 + sizeof(double) * 2 * kL1MaxBlocks; }

FN2_API int fn2_l1loss_forward(const fn2_l1loss_params* p, const float* bottom0, const float* bottom1, float* loss_out,
                               int N, int C, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  L1Args a;
  int rc = l1_args(p, N, C, H, W, &a);
  if (rc) return rc;
  if (!bottom0) return fail(FN2_ERR_INVALID_ARG, "l1loss_forward: bottom[0] == NULL");
  if (!workspace || workspace_bytes < fn2_l1loss_workspace_bytes(N, C, H, W))
    return fail(FN2_ERR_WORKSPACE, "l1loss_forward: workspace too small (%zu < %zu)", workspace_bytes, fn2_l1loss_workspace_bytes(N, C, H, W));
  hipStream_t st = as_stream(stream);
  float* wsf = reinterpret_cast<float*>(workspace);
  double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 64);
  const int nb = l1_blocks(a);
  hipLaunchKernelGGL(l1loss_fwd_partial, dim3(nb), dim3(kL1Threads), 0, st, bottom0, bottom1, partial, a);
  hipLaunchKernelGGL(l1loss_fwd_final, dim3(1), dim3(kL1Threads), 0, st, partial, nb, wsf, loss_out, a);
  return check_launch("l1loss_forward");
}

FN2_API int fn2_l1loss_backward(const fn2_l1loss_params* p, const float* bottom0, const float* bottom1, float top_diff,
                                float* bottom0_diff, float* bottom1_diff, int N, int C, int H, int W, void* workspace,
                                size_t workspace_bytes, void* stream) {
  L1Args a;
  int rc = l1_args(p, N, C, H, W, &a);
  if (rc) return rc;
  if (!bottom0 || !bottom0_diff) return fail(FN2_ERR_INVALID_ARG, "l1loss_backward: NULL blob pointer");
  if (!workspace || workspace_bytes < fn2_l1loss_workspace_bytes(N, C, H, W))
    return fail(FN2_ERR_WORKSPACE, "l1loss_backward: workspace too small");
  const int nb = l1_blocks(a);
  hipLaunchKernelGGL(l1loss_bwd, dim3(nb), dim3(kL1Threads), 0, as_stream(stream), bottom0, bottom1,
                     reinterpret_cast<const float*>(workspace), top_diff, bottom0_diff, bottom1 ? bottom1_diff : nullptr, a);
  return check_launch("l1loss_backward");
}
