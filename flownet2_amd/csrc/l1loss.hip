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
// (bid, nblk): this workgroup's index and the number of workgroups of ITS blob pair -- the grid-stride loop and with it the summation
// order are a function of the blob shape only, whether the pair has a launch of its own or shares one with other scales.
__device__ __forceinline__ void l1_partial_body(const float* __restrict__ b0, const float* __restrict__ b1, double* __restrict__ partial,
                                                const L1Args& a, unsigned bid, unsigned nblk, double* lds) {
  const size_t hw = (size_t)a.H * a.W;
  double dot = 0.0, nvalid = 0.0;
  if (a.l2_per_location) {
    const float wgt = a.prescale ? 1.f / (float)a.C : 1.f;    // l1loss_layer.cpp:47-51
    const float plat2 = a.plateau * a.plateau;                // l1loss_layer.cu:104
    const long long total = (long long)a.N * hw;
    for (long long q = bid * (long long)blockDim.x + threadIdx.x; q < total; q += (long long)nblk * blockDim.x) {
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
    for (long long i = bid * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)nblk * blockDim.x) {
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
    partial[2 * bid] = dot;
    partial[2 * bid + 1] = nvalid;
  }
}

__global__ void __launch_bounds__(kL1Threads) l1loss_fwd_partial(const float* __restrict__ b0, const float* __restrict__ b1,
                                                                 double* __restrict__ partial, L1Args a) {
  __shared__ double lds[8];
  l1_partial_body(b0, b1, partial, a, blockIdx.x, gridDim.x, lds);
}

// ws[0] = loss, ws[1] = normalize_coeff; loss_out[0] = loss.
__device__ __forceinline__ void l1_final_body(const double* __restrict__ partial, int nblocks, float* __restrict__ ws, float* __restrict__ loss_out,
                                              const L1Args& a, double* lds) {
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

__global__ void __launch_bounds__(kL1Threads) l1loss_fwd_final(const double* __restrict__ partial, int nblocks,
                                                               float* __restrict__ ws, float* __restrict__ loss_out, L1Args a) {
  __shared__ double lds[8];
  l1_final_body(partial, nblocks, ws, loss_out, a, lds);
}

__device__ __forceinline__ void l1_bwd_body(const float* __restrict__ b0, const float* __restrict__ b1, const float* __restrict__ ws, float top_diff,
                                            float* __restrict__ d0, float* __restrict__ d1, const L1Args& a, unsigned bid, unsigned nblk) {
  const size_t hw = (size_t)a.H * a.W;
  const float alpha = top_diff / ws[1];                        // cu:155
  if (a.l2_per_location) {
    const float wgt = a.prescale ? 1.f / (float)a.C : 1.f;
    const float plat2 = a.plateau * a.plateau;
    const long long total = (long long)a.N * hw;
    for (long long q = bid * (long long)blockDim.x + threadIdx.x; q < total; q += (long long)nblk * blockDim.x) {
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
    for (long long i = bid * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)nblk * blockDim.x) {
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

__global__ void __launch_bounds__(kL1Threads) l1loss_bwd(const float* __restrict__ b0, const float* __restrict__ b1,
                                                         const float* __restrict__ ws, float top_diff,
                                                         float* __restrict__ d0, float* __restrict__ d1, L1Args a) {
  l1_bwd_body(b0, b1, ws, top_diff, d0, d1, a, blockIdx.x, gridDim.x);
}

// ---- all loss layers of a net in ONE launch per direction (FlowNet trains on five scales: 15 launches of the three kernels above).
// A workgroup belongs to one scale (blk0[s] <= blockIdx.x < blk0[s + 1]) and runs that scale's body with the (bid, nblk) its own launch
// would have had: partial sums, their order and the loss are bit-identical to fn2_l1loss_forward.  The workgroup that arrives LAST at
// its scale's counter (agent-scope release / acquire around an atomic: the partials of the other workgroups may sit in another XCD's
// L2) runs the finalise body -- which adds the partials in index order whoever runs it; the scale that finishes last adds up
// total = sum_s loss_weight[s] * loss[s] in scale order, each product and sum rounded to float (Net::ForwardFromTo, net.cpp:565-579 +
// Layer::Forward's caffe_cpu_dot(top, loss_weight), layer.hpp:434-440).  The counters are left at zero for the next call.
constexpr int kL1MaxScales = 8;
struct L1Scale {
  const float* b0; const float* b1; float* d0; float* d1;
  float* ws; double* partial;
  L1Args a;
  int blk0, nb;
  float loss_weight;
};
struct L1Multi {
  int n;
  L1Scale s[kL1MaxScales];
  unsigned* sync;           // [kL1MaxScales] arrival counters + [1] finished scales; zero between calls
  float* losses;            // [n] or null
  float* total;             // [1] or null
  const float* total_diff;  // backward: d(objective) / d(total), a device scalar (null: 1)
};

__device__ __forceinline__ int l1_scale_of(const L1Multi& m) {
  int s = 0;
#pragma unroll
  for (int i = 1; i < kL1MaxScales; ++i) s += (i < m.n && (int)blockIdx.x >= m.s[i].blk0) ? 1 : 0;
  return s;
}

__global__ void __launch_bounds__(kL1Threads) l1loss_fwd_multi(L1Multi m) {
#pragma clang fp contract(off)
  __shared__ double lds[8];
  __shared__ int s_last;
  const int si = l1_scale_of(m);
  const L1Scale& sc = m.s[si];
  l1_partial_body(sc.b0, sc.b1, sc.partial, sc.a, blockIdx.x - sc.blk0, sc.nb, lds);
  if (threadIdx.x == 0) {
    __threadfence();                                                  // release: this workgroup's partial before its arrival
    s_last = atomicAdd(&m.sync[si], 1u) == (unsigned)(sc.nb - 1);
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();                                                    // acquire: the other workgroups' partials
  l1_final_body(sc.partial, sc.nb, sc.ws, m.losses ? m.losses + si : nullptr, sc.a, lds);
  if (threadIdx.x == 0) {
    m.sync[si] = 0u;
    __threadfence();
    if (atomicAdd(&m.sync[kL1MaxScales], 1u) == (unsigned)(m.n - 1)) {
      __threadfence();
      float total = 0.f;
      for (int i = 0; i < m.n; ++i) {
        const float l = *reinterpret_cast<volatile float*>(m.s[i].ws);
        total = total + m.s[i].loss_weight * l;
      }
      if (m.total) m.total[0] = total;
      m.sync[kL1MaxScales] = 0u;
    }
  }
}

__global__ void __launch_bounds__(kL1Threads) l1loss_bwd_multi(L1Multi m) {
#pragma clang fp contract(off)
  const int si = l1_scale_of(m);
  const L1Scale& sc = m.s[si];
  const float g = m.total_diff ? m.total_diff[0] : 1.f;
  l1_bwd_body(sc.b0, sc.b1, sc.ws, sc.loss_weight * g, sc.d0, sc.d1, sc.a, blockIdx.x - sc.blk0, sc.nb);       // top_diff of the layer = loss_weight x d(total)
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

// ---- multi-scale entry points
static size_t l1_scale_ws_bytes() { return 64 + sizeof(double) * 2 * kL1MaxBlocks; }

FN2_API size_t fn2_l1loss_multi_workspace_bytes(int nscales) { return nscales > 0 ? (size_t)nscales * l1_scale_ws_bytes() : 0; }
FN2_API size_t fn2_l1loss_multi_sync_bytes(void) { return sizeof(unsigned) * (kL1MaxScales + 1 + 7); }

static int l1_multi_fill(const char* what, const fn2_l1loss_params* p, int nscales, const fn2_l1loss_scale* scales, void* workspace,
                         size_t workspace_bytes, bool backward, L1Multi* m, unsigned* total_blocks) {
  if (nscales < 1 || nscales > kL1MaxScales) return fail(FN2_ERR_INVALID_ARG, "%s: 1 .. %d scales (got %d)", what, kL1MaxScales, nscales);
  if (!scales) return fail(FN2_ERR_INVALID_ARG, "%s: scales == NULL", what);
  if (!workspace || workspace_bytes < fn2_l1loss_multi_workspace_bytes(nscales))
    return fail(FN2_ERR_WORKSPACE, "%s: workspace too small (%zu < %zu)", what, workspace_bytes, fn2_l1loss_multi_workspace_bytes(nscales));
  m->n = nscales;
  int blk = 0;
  for (int i = 0; i < nscales; ++i) {
    L1Scale& sc = m->s[i];
    int rc = l1_args(p, scales[i].N, scales[i].C, scales[i].H, scales[i].W, &sc.a);
    if (rc) return rc;
    if (!scales[i].bottom0 || (backward && !scales[i].bottom0_diff)) return fail(FN2_ERR_INVALID_ARG, "%s: NULL blob pointer (scale %d)", what, i);
    sc.b0 = scales[i].bottom0; sc.b1 = scales[i].bottom1;
    sc.d0 = scales[i].bottom0_diff; sc.d1 = scales[i].bottom1 ? scales[i].bottom1_diff : nullptr;
    char* base = reinterpret_cast<char*>(workspace) + (size_t)i * l1_scale_ws_bytes();
    sc.ws = reinterpret_cast<float*>(base);
    sc.partial = reinterpret_cast<double*>(base + 64);
    sc.nb = l1_blocks(sc.a);
    sc.blk0 = blk;
    sc.loss_weight = scales[i].loss_weight;
    blk += sc.nb;
  }
  *total_blocks = (unsigned)blk;
  return FN2_OK;
}

FN2_API int fn2_l1loss_forward_multi(const fn2_l1loss_params* p, int nscales, const fn2_l1loss_scale* scales, float* losses, float* total,
                                     void* workspace, size_t workspace_bytes, void* sync, void* stream) {
  L1Multi m{};
  unsigned blocks = 0;
  int rc = l1_multi_fill("l1loss_forward_multi", p, nscales, scales, workspace, workspace_bytes, false, &m, &blocks);
  if (rc) return rc;
  if (!sync) return fail(FN2_ERR_INVALID_ARG, "l1loss_forward_multi: sync == NULL (fn2_l1loss_multi_sync_bytes zero bytes, left zero by every call)");
  m.sync = reinterpret_cast<unsigned*>(sync);
  m.losses = losses; m.total = total; m.total_diff = nullptr;
  hipLaunchKernelGGL(l1loss_fwd_multi, dim3(blocks), dim3(kL1Threads), 0, as_stream(stream), m);
  return check_launch("l1loss_forward_multi");
}

FN2_API int fn2_l1loss_backward_multi(const fn2_l1loss_params* p, int nscales, const fn2_l1loss_scale* scales, const float* total_diff,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  L1Multi m{};
  unsigned blocks = 0;
  int rc = l1_multi_fill("l1loss_backward_multi", p, nscales, scales, workspace, workspace_bytes, true, &m, &blocks);
  if (rc) return rc;
  m.sync = nullptr; m.losses = nullptr; m.total = nullptr; m.total_diff = total_diff;
  hipLaunchKernelGGL(l1loss_bwd_multi, dim3(blocks), dim3(kL1Threads), 0, as_stream(stream), m);
  return check_launch("l1loss_backward_multi");
}
