// Cross-entropy over the MLM decoder's logits (reference objectives.py:88-143: F.cross_entropy(mlm_logits.view(-1, vocab),
// mlm_labels.view(-1), ignore_index=-100) on the [B * T, 30 522] logits) as two row-wise bandwidth-bound kernels on the bf16
// logits the decoder GEMM wrote: stock torch upcasts the logits to a contiguous fp32 copy (54 -> 107 MB), runs log_softmax and
// nll_loss forward / backward over it (four more passes) and copies the fp32 gradient back into the padded bf16 buffer the dgrad
// GEMM reads.  Here: forward = one read of the logits (online max / sum of exponentials per row -> lse, loss = lse - logit[label]),
// backward = one read + one write: dlogits = scale * (exp(logit - lse) - [col == label]) as bf16, straight into the
// [rows, ld_d] buffer (zero in the padding columns and in rows whose label is ignore_index) that the decoder's dgrad and wgrad
// GEMMs take as their operand.  Arithmetic in fp32 like the reference's autocast (cross_entropy runs in fp32).
#include "vlm_common.h"

#define CE_THREADS 256

__device__ __forceinline__ void ce_online(float x, float& m, float& s) {  // running (max, sum exp(x - max))
  if (x > m) {
    s = s * __expf(m - x) + 1.0f;
    m = x;
  } else {
    s += __expf(x - m);
  }
}

__global__ __launch_bounds__(CE_THREADS) void cross_entropy_fwd_kernel(const bf16_t* __restrict__ logits, int ld, int V,
                                                                      const int64_t* __restrict__ labels, int64_t ignore,
                                                                      float* __restrict__ loss_rows, float* __restrict__ lse_out) {
  __shared__ float sm[CE_THREADS / 64], ss[CE_THREADS / 64];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* p = logits + (size_t)row * ld;
  float m = -INFINITY, s = 0.f;
  const int V8 = V & ~7;
  for (int c = tid * 8; c < V8; c += CE_THREADS * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) ce_online((float)v[j], m, s);
  }
  if (tid < V - V8) ce_online((float)p[V8 + tid], m, s);
  // combine the threads' (m, s) pairs: wave, then workgroup
  const float wm = wave_max(m);
  s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));  // a thread (or a whole wave) without columns contributes nothing
  if (lane == 0) { sm[wave] = wm; ss[wave] = s; }
  __syncthreads();
  if (tid == 0) {
    float M = sm[0];
    for (int w = 1; w < CE_THREADS / 64; ++w) M = fmaxf(M, sm[w]);
    float S = 0.f;
    for (int w = 0; w < CE_THREADS / 64; ++w) S += sm[w] == -INFINITY ? 0.f : ss[w] * __expf(sm[w] - M);
    const float lse = M + __logf(S);
    lse_out[row] = lse;
    const int64_t lab = labels[row];
    const bool valid = lab != ignore && lab >= 0 && lab < V;
    loss_rows[row] = valid ? lse - (float)p[lab] : 0.f;
  }
}

__global__ __launch_bounds__(CE_THREADS) void cross_entropy_bwd_kernel(const bf16_t* __restrict__ logits, int ld, int V, int Vpad,
                                                                      const int64_t* __restrict__ labels, int64_t ignore,
                                                                      const float* __restrict__ lse, const float* __restrict__ scale,
                                                                      bf16_t* __restrict__ dlogits, int ld_d) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16_t* p = logits + (size_t)row * ld;
  bf16_t* d = dlogits + (size_t)row * ld_d;
  const int64_t lab = labels[row];
  const bool valid = lab != ignore && lab >= 0 && lab < V;
  const float sc = valid ? scale[0] : 0.f, l = lse[row];
  for (int c = tid * 8; c < Vpad; c += CE_THREADS * 8) {  // Vpad % 8 == 0, both rows 16-B aligned (launcher)
    bf16x8 o;
    if (!valid || c >= V) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.0f;
    }
    if (valid && c < V) {
      if (c + 8 <= V) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(sc * (__expf((float)v[j] - l) - ((int64_t)(c + j) == lab ? 1.0f : 0.0f)));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cc = c + j;
          o[j] = cc < V ? (bf16_t)(sc * (__expf((float)p[cc] - l) - ((int64_t)cc == lab ? 1.0f : 0.0f))) : (bf16_t)0.0f;
        }
      }
    }
    *reinterpret_cast<bf16x8*>(d + c) = o;
  }
}

extern "C" int vlm_cross_entropy_fwd(const void* logits_bf16, int ld, int rows, int V, const int64_t* labels, int64_t ignore_index,
                                     float* loss_rows, float* lse, void* stream) {
  if (rows == 0) return VLM_OK;
  if (!logits_bf16 || !labels || !loss_rows || !lse || rows < 0 || V <= 0 || ld < V || (ld & 7) || ((uintptr_t)logits_bf16 & 15))
    return VLM_ERR_ARG;
  hipLaunchKernelGGL(cross_entropy_fwd_kernel, dim3(rows), dim3(CE_THREADS), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16_t*>(logits_bf16), ld, V, labels, ignore_index, loss_rows, lse);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_cross_entropy_bwd(const void* logits_bf16, int ld, int rows, int V, const int64_t* labels, int64_t ignore_index,
                                     const float* lse, const float* scale_dev, void* dlogits_bf16, int ld_d, void* stream) {
  if (rows == 0) return VLM_OK;
  if (!logits_bf16 || !labels || !lse || !scale_dev || !dlogits_bf16 || rows < 0 || V <= 0 || ld < V || (ld & 7) || ld_d < V ||
      (ld_d & 7) || ((uintptr_t)logits_bf16 & 15) || ((uintptr_t)dlogits_bf16 & 15))
    return VLM_ERR_ARG;
  hipLaunchKernelGGL(cross_entropy_bwd_kernel, dim3(rows), dim3(CE_THREADS), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16_t*>(logits_bf16), ld, V, ld_d, labels, ignore_index, lse, scale_dev,
                     reinterpret_cast<bf16_t*>(dlogits_bf16), ld_d);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
