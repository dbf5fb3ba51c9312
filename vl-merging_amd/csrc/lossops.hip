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

// ======================================================================================================================
// Round 6: the rest of the loss tail as HIP kernels (objectives.py:248-300 compute_ifm, :372 compute_irtr, :146
// compute_itm_hardneg's cross-entropy; the [B, *] algebra that round 4-5 left to ~120 torch-native launches per step).
// Everything here is tiny (B = 22 samples, D = 768): the point is ONE launch where torch issued a dozen.
// ----------------------------------------------------------------------------------------------------------------------

// y = x / ||x||_2 per row in fp32 (x bf16 or fp32), inv[row] = 1 / ||x||: one wave per row.
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, int ld, int rows, int D, float* __restrict__ y,
                                                        float* __restrict__ inv) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const T* p = x + (size_t)row * ld;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = (float)p[c]; s += v * v; }
  s = wave_sum(s);
  const float n = sqrtf(s);  // (the reference divides by the norm: a zero row gives inf / nan there too)
  for (int c = lane; c < D; c += 64) y[(size_t)row * D + c] = (float)p[c] / n;
  if (lane == 0) inv[row] = 1.0f / n;
}

// dx = (g - y (g . y)) * inv, written in x's dtype: one wave per row.
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                        const float* __restrict__ inv, int rows, int D, T* __restrict__ dx, int ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* gp = g + (size_t)row * D;
  const float* yp = y + (size_t)row * D;
  float t = 0.f;
  for (int c = lane; c < D; c += 64) t += gp[c] * yp[c];
  t = wave_sum(t);
  const float iv = inv[row];
  for (int c = lane; c < D; c += 64) dx[(size_t)row * ld + c] = (T)((gp[c] - yp[c] * t) * iv);
}

extern "C" int vlm_l2norm_fwd(const void* x, int x_is_bf16, int ld, int rows, int D, float* y, float* inv, void* stream) {
  if (rows == 0) return VLM_OK;
  if (!x || !y || !inv || rows < 0 || D <= 0 || ld < D) return VLM_ERR_ARG;
  const dim3 grid((rows + 3) / 4), block(256);
  if (x_is_bf16) hipLaunchKernelGGL((l2norm_fwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, ld, rows, D, y, inv);
  else hipLaunchKernelGGL((l2norm_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, ld, rows, D, y, inv);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_l2norm_bwd(const float* g, const float* y, const float* inv, int rows, int D, void* dx, int dx_is_bf16, int ld,
                              void* stream) {
  if (rows == 0) return VLM_OK;
  if (!g || !y || !inv || !dx || rows < 0 || D <= 0 || ld < D) return VLM_ERR_ARG;
  const dim3 grid((rows + 3) / 4), block(256);
  if (dx_is_bf16) hipLaunchKernelGGL((l2norm_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, g, y, inv, rows, D, (bf16_t*)dx, ld);
  else hipLaunchKernelGGL((l2norm_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, g, y, inv, rows, D, (float*)dx, ld);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- symmetric contrastive loss on L2-normalised features (CLIP style; objectives.py:248-300 / :372-445) -----------------
// all_img / all_txt: [n, D] fp32, this rank's B rows first (the reference re-inserts its own slice at index 0 and lets
// gradients flow only through it, :269-286); s = exp(log_scale).
//   logits[i][j] = s <img_i, txt_j> ;  loss = (CE(logits, diag) + CE(logits^T, diag)) / 2   (means over the n rows)
// Kernel 1: logits (one workgroup per image row).  Kernel 2 (one workgroup): row / column statistics, the loss, the gradient
// matrix G = dloss / dlogits, d log_scale = sum G logits, and d img[i] = s sum_j G[i][j] txt_j, d txt[j] = s sum_i G[i][j] img_i
// for the OWN rows i, j < B.  Forward and gradient in one go: the loss is a scalar leaf of the graph, its upstream gradient only
// scales these (vlm_scale_by_scalar).
__global__ __launch_bounds__(256) void contrastive_logits_kernel(const float* __restrict__ img, const float* __restrict__ txt, int n, int D,
                                                                 const float* __restrict__ log_scale, float* __restrict__ logits) {
  extern __shared__ float row[];  // [D]
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < D; c += 256) row[c] = img[(size_t)i * D + c];
  __syncthreads();
  const float s = __expf(log_scale[0]);
  for (int j = wave; j < n; j += 4) {
    const float* t = txt + (size_t)j * D;
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) acc += row[c] * t[c];
    acc = wave_sum(acc);
    if (lane == 0) logits[(size_t)i * n + j] = s * acc;
  }
}

__global__ __launch_bounds__(1024) void contrastive_grad_kernel(const float* __restrict__ img, const float* __restrict__ txt, int n, int B,
                                                                int D, const float* __restrict__ log_scale, const float* __restrict__ logits,
                                                                float* __restrict__ G, float* __restrict__ out /* loss, d log_scale, s */,
                                                                float* __restrict__ d_img, float* __restrict__ d_txt) {
  extern __shared__ float sh[];  // row max, row lse, col max, col lse: 4 n floats; then 32 floats of reduction scratch
  float* rmax = sh; float* rlse = sh + n; float* cmax = sh + 2 * n; float* clse = sh + 3 * n; float* red = sh + 4 * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const float s = __expf(log_scale[0]);
  // statistics: wave w takes rows w, w + nw, ... and then the columns
  for (int k = wave; k < 2 * n; k += nw) {
    const bool col = k >= n;
    const int i = col ? k - n : k;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, col ? logits[(size_t)j * n + i] : logits[(size_t)i * n + j]);
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < n; j += 64) z += __expf((col ? logits[(size_t)j * n + i] : logits[(size_t)i * n + j]) - m);
    z = wave_sum(z);
    if (lane == 0) { (col ? cmax : rmax)[i] = m; (col ? clse : rlse)[i] = m + __logf(z); }
  }
  __syncthreads();
  // loss, G, d log_scale
  float lpart = 0.f, spart = 0.f;
  const float w = 0.5f / (float)n;
  for (int e = tid; e < n * n; e += blockDim.x) {
    const int i = e / n, j = e - i * n;
    const float l = logits[e];
    const float g = w * (__expf(l - rlse[i]) + __expf(l - clse[j]) - (i == j ? 2.0f : 0.f));
    G[e] = g;
    spart += g * l;
    if (i == j) lpart += w * (rlse[i] + clse[i] - 2.0f * l);
  }
  lpart = wave_sum(lpart);
  spart = wave_sum(spart);
  if (lane == 0) { red[wave] = lpart; red[16 + wave] = spart; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < nw; ++k) { a += red[k]; b += red[16 + k]; }
    out[0] = a;
    out[1] = b;  // d loss / d log_scale = sum G logits (d s = sum G logits / s, times ds / d log_scale = s)
    out[2] = s;
  }
  // feature gradients of the own rows
  for (int e = tid; e < B * D; e += blockDim.x) {
    const int i = e / D, c = e - i * D;
    float a = 0.f, b = 0.f;
    for (int j = 0; j < n; ++j) {
      a += G[(size_t)i * n + j] * txt[(size_t)j * D + c];
      b += G[(size_t)j * n + i] * img[(size_t)j * D + c];
    }
    d_img[e] = s * a;
    d_txt[e] = s * b;
  }
}

extern "C" size_t vlm_contrastive_ws_floats(int n) { return (size_t)n * n; }

extern "C" int vlm_contrastive(const float* all_img, const float* all_txt, int n, int B, int D, const float* log_scale, float* logits,
                               float* out3, float* d_img, float* d_txt, float* ws, void* stream) {
  if (n == 0) return VLM_OK;
  if (!all_img || !all_txt || !log_scale || !logits || !out3 || !d_img || !d_txt || !ws || n < 0 || B < 0 || B > n || D <= 0) return VLM_ERR_ARG;
  if ((size_t)D * 4 > 64 * 1024 || (size_t)(4 * n + 32) * 4 > 64 * 1024) return VLM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(contrastive_logits_kernel, dim3(n), dim3(256), (size_t)D * 4, s, all_img, all_txt, n, D, log_scale, logits);
  VLM_CHECK_LAUNCH();
  hipLaunchKernelGGL(contrastive_grad_kernel, dim3(1), dim3(1024), (size_t)(4 * n + 32) * 4, s, all_img, all_txt, n, B, D, log_scale,
                     logits, ws, out3, d_img, d_txt);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- small cross-entropy (the ITM head's [3B, 2] logits, objectives.py:146-245): loss (mean over rows) and dlogits = (softmax -
// onehot) / rows in one workgroup; logits bf16 or fp32 with a row stride ----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void small_ce_kernel(const T* __restrict__ logits, int ld, int rows, int V, const int64_t* __restrict__ labels,
                                                      float* __restrict__ loss, float* __restrict__ dlogits) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float part = 0.f;
  for (int r = tid; r < rows; r += 256) {
    const T* p = logits + (size_t)r * ld;
    float m = -INFINITY;
    for (int c = 0; c < V; ++c) m = fmaxf(m, (float)p[c]);
    float z = 0.f;
    for (int c = 0; c < V; ++c) z += __expf((float)p[c] - m);
    const float lse = m + __logf(z);
    const int64_t lab = labels[r];
    part += lse - (float)p[lab];
    for (int c = 0; c < V; ++c) dlogits[(size_t)r * V + c] = (__expf((float)p[c] - lse) - (c == lab ? 1.0f : 0.f)) / (float)rows;
  }
  part = wave_sum(part);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)rows;
}

extern "C" int vlm_small_cross_entropy(const void* logits, int is_bf16, int ld, int rows, int V, const int64_t* labels, float* loss,
                                       float* dlogits, void* stream) {
  if (!logits || !labels || !loss || !dlogits || rows <= 0 || V <= 0 || V > 4096 || ld < V) return VLM_ERR_ARG;
  if (is_bf16) hipLaunchKernelGGL((small_ce_kernel<bf16_t>), dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, rows, V, labels, loss, dlogits);
  else hipLaunchKernelGGL((small_ce_kernel<float>), dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ld, rows, V, labels, loss, dlogits);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- out_k[i] = in_k[i] * scalar_dev[0] for up to four buffers in ONE launch (a scalar loss's upstream gradient) ------------
struct scale_jobs_t { const float* in[4]; float* out[4]; int n[4]; };
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const scale_jobs_t jobs, const float* __restrict__ scalar) {
  const float s = scalar[0];
  const int k = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < jobs.n[k]; i += gridDim.x * 256) jobs.out[k][i] = jobs.in[k][i] * s;
}

extern "C" int vlm_scale_by_scalar(const float* const* in, float* const* out, const int* n, int count, const float* scalar_dev, void* stream) {
  if (count == 0) return VLM_OK;
  if (!in || !out || !n || !scalar_dev || count < 0 || count > 4) return VLM_ERR_ARG;
  scale_jobs_t j = {};
  int mx = 0;
  for (int k = 0; k < count; ++k) { j.in[k] = in[k]; j.out[k] = out[k]; j.n[k] = n[k]; if (n[k] > mx) mx = n[k]; if (n[k] < 0 || (n[k] && (!in[k] || !out[k]))) return VLM_ERR_ARG; }
  if (mx == 0) return VLM_OK;
  int gx = (mx + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(gx, count), dim3(256), 0, (hipStream_t)stream, j, scalar_dev);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- mean cross-entropy from the per-row losses of cross_entropy_fwd_kernel: loss = sum(loss_rows) / count, inv_count = 1 / count,
// count = rows whose label is counted (not ignore_index, inside [0, V)): replaces nine torch launches -------------------------
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ loss_rows, const int64_t* __restrict__ labels, int rows, int V,
                                                       int64_t ignore, float* __restrict__ out2) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f, c = 0.f;
  for (int r = tid; r < rows; r += 256) {
    const int64_t lab = labels[r];
    if (lab != ignore && lab >= 0 && lab < V) { s += loss_rows[r]; c += 1.0f; }
  }
  s = wave_sum(s);
  c = wave_sum(c);
  if (lane == 0) { red[wave] = s; red[4 + wave] = c; }
  __syncthreads();
  if (tid == 0) {
    const float S = red[0] + red[1] + red[2] + red[3], C = red[4] + red[5] + red[6] + red[7];
    out2[0] = S / C;      // no counted row: 0 / 0 = nan, like F.cross_entropy
    out2[1] = 1.0f / C;
  }
}

extern "C" int vlm_cross_entropy_reduce(const float* loss_rows, const int64_t* labels, int rows, int V, int64_t ignore_index, float* out2,
                                        void* stream) {
  if (!loss_rows || !labels || !out2 || rows < 0 || V <= 0) return VLM_ERR_ARG;
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_rows, labels, rows, V, ignore_index, out2);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
