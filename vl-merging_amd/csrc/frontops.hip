// The two ends of a pass as single launches (round 6): the token rows a pass starts from and the small pieces between the
// last block and the losses.  Everything here is launch-bound glue that stock torch issues as 5-15 kernels per call site;
// none of it is on the roofline of the step, all of it is on its launch count (profiles/r06_launches_per_step.txt).
//
//   text rows   BertEmbeddings.forward + token_type_embeddings(0) (reference vilt_module.py:51-63, :1111-1113):
//               out[r] = dropout(LayerNorm(word[ids[r]] + bert_type0)) + vilt_type0, written straight into the pass's token matrix
//   image rows  the lead (cls) rows and the patch-embed GEMM's bias of vision_transformer.py:952-991 + vilt_module.py:1114-1117
//   heads       tanh / GELU' times the upstream gradient as the bf16 operand of the head's dgrad / wgrad GEMMs (heads.py:8-53),
//               column sums for a head with a handful of outputs (ITMHead: 2)
//   negatives   the hard-negative draw of objectives.py:176-229 (softmax over the similarities, own pair excluded, one
//               categorical sample per row) as one launch from 2B uniforms
#include "vlm_common.h"

#define FR_MAXV 4  // float4 per lane: D <= 64 * 4 * 4 = 1024 (the backward's 16 D floats of LDS stay within 64 KB)

// ---- text rows -----------------------------------------------------------------------------------------------------------
// One wave per row.  u (optional): fp32 [n, D]; an element is KEPT where u >= p and multiplied by `scale` (nn.Dropout: p = drop
// probability, scale = 1 / (1 - p); an injected keep mask arrives as u in {0, 1} with p = 0.5).
__global__ __launch_bounds__(256) void text_rows_fwd_kernel(const int64_t* __restrict__ ids, int n, const float* __restrict__ word, int ldw,
                                                           const float* __restrict__ add0, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, const float* __restrict__ u, float p,
                                                           float scale, const float* __restrict__ add1, float* __restrict__ out, int ldo,
                                                           float* __restrict__ stats, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const int nv = D >> 2;
  const float4* w = reinterpret_cast<const float4*>(word + (size_t)ids[row] * ldw);
  float4 v[FR_MAXV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < FR_MAXV; ++k) {
    const int c = lane + 64 * k;
    if (c < nv) {
      float4 a = w[c];
      if (add0) { const float4 b = reinterpret_cast<const float4*>(add0)[c]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
      v[k] = a;
      s += a.x + a.y + a.z + a.w;
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < FR_MAXV; ++k) {
    const int c = lane + 64 * k;
    if (c < nv) {
      const float4 a = v[k];
      q += (a.x - mean) * (a.x - mean) + (a.y - mean) * (a.y - mean) + (a.z - mean) * (a.z - mean) + (a.w - mean) * (a.w - mean);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int k = 0; k < FR_MAXV; ++k) {
    const int c = lane + 64 * k;
    if (c < nv) {
      const float4 a = v[k], g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
      float4 y = {(a.x - mean) * rstd * g.x + b.x, (a.y - mean) * rstd * g.y + b.y, (a.z - mean) * rstd * g.z + b.z,
                  (a.w - mean) * rstd * g.w + b.w};
      if (u) {
        const float4 r = reinterpret_cast<const float4*>(u + (size_t)row * D)[c];
        y.x = r.x >= p ? y.x * scale : 0.f; y.y = r.y >= p ? y.y * scale : 0.f;
        y.z = r.z >= p ? y.z * scale : 0.f; y.w = r.w >= p ? y.w * scale : 0.f;
      }
      if (add1) { const float4 t = reinterpret_cast<const float4*>(add1)[c]; y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w; }
      reinterpret_cast<float4*>(out + (size_t)row * ldo)[c] = y;
    }
  }
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// Backward of the above.  A workgroup strides over the rows; its waves keep the four column sums (d add1 = sum g, d beta, d gamma,
// d add0 = sum dx) in registers, meet in LDS at the end and store one partial per workgroup: part[wg][4][D] (folded by
// fold_parts_kernel -- plain stores and a fixed order instead of contended float atomics: deterministic).  dx goes to the word
// rows by float atomics (ids repeat), rows with ids == padding_idx get none (nn.Embedding(padding_idx)).
__global__ __launch_bounds__(256) void text_rows_bwd_kernel(const float* __restrict__ g, int ldg, const int64_t* __restrict__ ids, int n,
                                                           const float* __restrict__ word, int ldw, const float* __restrict__ add0,
                                                           const float* __restrict__ gamma, const float* __restrict__ stats,
                                                           const float* __restrict__ u, float p, float scale, int D,
                                                           float* __restrict__ dword, int64_t padding_idx, float* __restrict__ part) {
  extern __shared__ float sh[];  // [4 waves][4 sums][D]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nv = D >> 2;
  float4 acc[4][FR_MAXV];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < FR_MAXV; ++k) acc[j][k] = float4{0.f, 0.f, 0.f, 0.f};
  for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
    const int64_t id = ids[row];
    const float4* w = reinterpret_cast<const float4*>(word + (size_t)id * ldw);
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float4 xh[FR_MAXV], t[FR_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < FR_MAXV; ++k) {
      const int c = lane + 64 * k;
      if (c < nv) {
        float4 a = w[c];
        if (add0) { const float4 b = reinterpret_cast<const float4*>(add0)[c]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        const float4 x = {(a.x - mean) * rstd, (a.y - mean) * rstd, (a.z - mean) * rstd, (a.w - mean) * rstd};
        float4 gy = reinterpret_cast<const float4*>(g + (size_t)row * ldg)[c];
        acc[0][k].x += gy.x; acc[0][k].y += gy.y; acc[0][k].z += gy.z; acc[0][k].w += gy.w;
        if (u) {
          const float4 r = reinterpret_cast<const float4*>(u + (size_t)row * D)[c];
          gy.x = r.x >= p ? gy.x * scale : 0.f; gy.y = r.y >= p ? gy.y * scale : 0.f;
          gy.z = r.z >= p ? gy.z * scale : 0.f; gy.w = r.w >= p ? gy.w * scale : 0.f;
        }
        acc[1][k].x += gy.x; acc[1][k].y += gy.y; acc[1][k].z += gy.z; acc[1][k].w += gy.w;
        acc[2][k].x += gy.x * x.x; acc[2][k].y += gy.y * x.y; acc[2][k].z += gy.z * x.z; acc[2][k].w += gy.w * x.w;
        const float4 gm = reinterpret_cast<const float4*>(gamma)[c];
        const float4 tt = {gy.x * gm.x, gy.y * gm.y, gy.z * gm.z, gy.w * gm.w};
        xh[k] = x; t[k] = tt;
        s1 += tt.x + tt.y + tt.z + tt.w;
        s2 += tt.x * x.x + tt.y * x.y + tt.z * x.z + tt.w * x.w;
      }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
    float* dw = dword + (size_t)id * ldw;
#pragma unroll
    for (int k = 0; k < FR_MAXV; ++k) {
      const int c = lane + 64 * k;
      if (c < nv) {
        const float4 x = xh[k], tt = t[k];
        const float4 dx = {rstd * (tt.x - m1 - x.x * m2), rstd * (tt.y - m1 - x.y * m2), rstd * (tt.z - m1 - x.z * m2),
                           rstd * (tt.w - m1 - x.w * m2)};
        acc[3][k].x += dx.x; acc[3][k].y += dx.y; acc[3][k].z += dx.z; acc[3][k].w += dx.w;
        if (dword && id != padding_idx) {
          atomicAdd(dw + 4 * c, dx.x); atomicAdd(dw + 4 * c + 1, dx.y); atomicAdd(dw + 4 * c + 2, dx.z); atomicAdd(dw + 4 * c + 3, dx.w);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < FR_MAXV; ++k) {
      const int c = lane + 64 * k;
      if (c < nv) reinterpret_cast<float4*>(sh + (size_t)(wave * 4 + j) * D)[c] = acc[j][k];
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 4 * D; e += 256)
    part[(size_t)blockIdx.x * 4 * D + e] = sh[e] + sh[4 * D + e] + sh[8 * D + e] + sh[12 * D + e];
}

// dst[k][c] += sum over the partials p and the vectors j selected by mask[k] (bit j) of part[p][j][c].  A workgroup owns 16
// columns; its 64 thread groups take every 64th partial each and meet in LDS: a thread reads nparts / 64 * nvec values instead of
// nparts * nvec (the first form, one thread per column over all 512 partials in 3 workgroups, took 280 us per call).
struct fold_dst_t { float* dst[4]; unsigned mask[4]; };
#define FOLD_COLS 16
#define FOLD_GROUPS 64
__global__ __launch_bounds__(1024) void fold_parts_kernel(const float* __restrict__ part, int nparts, int nvec, int D, fold_dst_t f, int ndst) {
  __shared__ float sh[FOLD_GROUPS][4][FOLD_COLS];
  const int cl = threadIdx.x % FOLD_COLS, grp = threadIdx.x / FOLD_COLS, c = blockIdx.x * FOLD_COLS + cl;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < D)
    for (int p = grp; p < nparts; p += FOLD_GROUPS)
      for (int j = 0; j < nvec; ++j) s[j] += part[((size_t)p * nvec + j) * D + c];
  for (int j = 0; j < 4; ++j) sh[grp][j][cl] = s[j];
  __syncthreads();
  if (grp < nvec && c < D) {  // thread group j finishes vector j (fixed order: deterministic)
    float t = 0.f;
    for (int g = 0; g < FOLD_GROUPS; ++g) t += sh[g][grp][cl];
    sh[0][grp][cl] = t;
  }
  __syncthreads();
  if (grp == 0 && c < D) {
    for (int k = 0; k < ndst; ++k) {
      if (!f.dst[k]) continue;
      float v = 0.f;
      for (int j = 0; j < nvec; ++j) if (f.mask[k] >> j & 1) v += sh[0][j][cl];
      f.dst[k][c] += v;
    }
  }
}

extern "C" int vlm_text_rows_fwd(const int64_t* ids, int n, const float* word, int ld_word, const float* add0, const float* gamma,
                                 const float* beta, float eps, const float* u, float p, float scale, const float* add1, float* out,
                                 int ld_out, float* stats, int D, void* stream) {
  if (n == 0) return VLM_OK;
  if (!ids || !word || !gamma || !beta || !out || !stats || n < 0 || D <= 0 || (D & 3) || D > 256 * FR_MAXV || ld_word < D || (ld_word & 3) ||
      ld_out < D || (ld_out & 3))
    return VLM_ERR_ARG;
  hipLaunchKernelGGL(text_rows_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, n, word, ld_word, add0, gamma, beta, eps, u,
                     p, scale, add1, out, ld_out, stats, D);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

#define FR_TEXT_PARTS 256
extern "C" size_t vlm_text_rows_bwd_ws_floats(int D) { return (size_t)FR_TEXT_PARTS * 4 * D; }

extern "C" int vlm_text_rows_bwd(const float* g, int ld_g, const int64_t* ids, int n, const float* word, int ld_word, const float* add0,
                                 const float* gamma, const float* stats, const float* u, float p, float scale, int D, float* d_word,
                                 int64_t padding_idx, float* d_add1, float* d_beta, float* d_gamma, float* d_add0, float* ws, void* stream) {
  if (n == 0) return VLM_OK;
  if (!g || !ids || !word || !gamma || !stats || !ws || n < 0 || D <= 0 || (D & 3) || D > 256 * FR_MAXV || ld_word < D || (ld_word & 3) ||
      ld_g < D || (ld_g & 3))
    return VLM_ERR_ARG;
  const int parts = (n + 3) / 4 < FR_TEXT_PARTS ? (n + 3) / 4 : FR_TEXT_PARTS;
  hipLaunchKernelGGL(text_rows_bwd_kernel, dim3(parts), dim3(256), (size_t)16 * D * sizeof(float), (hipStream_t)stream, g, ld_g, ids, n, word,
                     ld_word, add0, gamma, stats, u, p, scale, D, d_word, padding_idx, ws);
  VLM_CHECK_LAUNCH();
  fold_dst_t f = {{d_add1, d_beta, d_gamma, d_add0}, {1u, 2u, 4u, 8u}};
  hipLaunchKernelGGL(fold_parts_kernel, dim3((D + FOLD_COLS - 1) / FOLD_COLS), dim3(1024), 0, (hipStream_t)stream, ws, parts, 4, D, f, 4);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- image rows ----------------------------------------------------------------------------------------------------------
// out2[0][c] = conv_bias[c] + type_row[c] (the patch-embed GEMM's bias), out2[1][c] = cls[c] + type_row[c] (the lead row).
__global__ __launch_bounds__(256) void image_rows_prep_kernel(const float* __restrict__ conv_bias, const float* __restrict__ type_row,
                                                             const float* __restrict__ cls, int D, float* __restrict__ out2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  const float t = type_row ? type_row[c] : 0.f;
  out2[c] = (conv_bias ? conv_bias[c] : 0.f) + t;
  out2[D + c] = cls[c] + t;
}

__global__ __launch_bounds__(256) void image_lead_rows_kernel(float* __restrict__ x, int ldx, int B, int rows, int D, const float* __restrict__ lead) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c < D && b < B) x[(size_t)b * rows * ldx + c] = lead[c];
}

extern "C" int vlm_image_rows_prep(const float* conv_bias, const float* type_row, const float* cls, int D, float* out2, void* stream) {
  if (!cls || !out2 || D <= 0) return VLM_ERR_ARG;
  hipLaunchKernelGGL(image_rows_prep_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, conv_bias, type_row, cls, D, out2);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_image_lead_rows(float* x, int ld_x, int B, int rows, int D, const float* lead, void* stream) {
  if (B == 0) return VLM_OK;
  if (!x || !lead || B < 0 || rows <= 0 || D <= 0 || ld_x < D) return VLM_ERR_ARG;
  hipLaunchKernelGGL(image_lead_rows_kernel, dim3((D + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, ld_x, B, rows, D, lead);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// g fp32 [B * rows, D] -> g16 bf16 with the lead rows (row % rows == 0) zeroed; part[wg][0][D] = column sums over the patch rows,
// part[wg][1][D] = over the lead rows.  D % 4 == 0; a thread owns 4 columns, a workgroup a band of rows.
__global__ __launch_bounds__(256) void image_rows_bwd_kernel(const float* __restrict__ g, int ldg, int n, int rows, int D,
                                                            bf16_t* __restrict__ g16, float* __restrict__ part) {
  const int nv = D >> 2;
  for (int c = threadIdx.x; c < nv; c += 256) {
    float4 sp = {0.f, 0.f, 0.f, 0.f}, sl = {0.f, 0.f, 0.f, 0.f};
    for (int r = blockIdx.x; r < n; r += gridDim.x) {
      const float4 v = reinterpret_cast<const float4*>(g + (size_t)r * ldg)[c];
      const bool lead = r % rows == 0;
      if (lead) { sl.x += v.x; sl.y += v.y; sl.z += v.z; sl.w += v.w; }
      else { sp.x += v.x; sp.y += v.y; sp.z += v.z; sp.w += v.w; }
      bf16x4 o;
      o[0] = (bf16_t)(lead ? 0.f : v.x); o[1] = (bf16_t)(lead ? 0.f : v.y); o[2] = (bf16_t)(lead ? 0.f : v.z); o[3] = (bf16_t)(lead ? 0.f : v.w);
      reinterpret_cast<bf16x4*>(g16 + (size_t)r * D)[c] = o;
    }
    reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * D)[c] = sp;
    reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * D + D)[c] = sl;
  }
}

#define FR_IMAGE_PARTS 1024
extern "C" size_t vlm_image_rows_bwd_ws_floats(int D) { return (size_t)FR_IMAGE_PARTS * 2 * D; }

// d_bias += patch sums; d_type_row += patch sums + lead sums; d_cls += lead sums (any of the three may be null).
extern "C" int vlm_image_rows_bwd(const float* g, int ld_g, int B, int rows, int D, void* g16, float* d_bias, float* d_type_row, float* d_cls,
                                  float* ws, void* stream) {
  if (B == 0) return VLM_OK;
  if (!g || !g16 || !ws || B < 0 || rows <= 0 || D <= 0 || (D & 3) || ld_g < D || (ld_g & 3)) return VLM_ERR_ARG;
  const int n = B * rows, parts = n < FR_IMAGE_PARTS ? n : FR_IMAGE_PARTS;
  hipLaunchKernelGGL(image_rows_bwd_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, g, ld_g, n, rows, D, (bf16_t*)g16, ws);
  VLM_CHECK_LAUNCH();
  fold_dst_t f = {{d_bias, d_type_row, d_cls, nullptr}, {1u, 3u, 2u, 0u}};
  hipLaunchKernelGGL(fold_parts_kernel, dim3((D + FOLD_COLS - 1) / FOLD_COLS), dim3(1024), 0, (hipStream_t)stream, ws, parts, 2, D, f, 3);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- heads ---------------------------------------------------------------------------------------------------------------
// y = tanh(x) (bf16 -> fp32), Pooler (heads.py:8-19).
__global__ __launch_bounds__(256) void tanh_fwd_kernel(const bf16_t* __restrict__ x, int ldx, int M, int N, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int r = (int)(i / N), c = (int)(i % N);
  y[i] = tanhf((float)x[(size_t)r * ldx + c]);
}

// dy (bf16 [M, Np], columns >= N zero) = upstream gradient times the activation's derivative:
//   mode 0  GELU (exact, erf) from the saved pre-activation h (bf16):  g * (Phi(h) + h phi(h))
//   mode 1  tanh from the saved OUTPUT y (fp32):                        g * (1 - y^2)
//   mode 2  none: the cast / zero padding alone
template <typename G>
__global__ __launch_bounds__(256) void act_bwd_kernel(const G* __restrict__ g, int ldg, const void* __restrict__ saved, int lds_, int mode, int M, int N,
                                                     int Np, bf16_t* __restrict__ dy) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * Np) return;
  const int r = (int)(i / Np), c = (int)(i % Np);
  float v = 0.f;
  if (c < N) {
    const float gg = (float)g[(size_t)r * ldg + c];
    if (mode == 0) {
      const float h = (float)reinterpret_cast<const bf16_t*>(saved)[(size_t)r * lds_ + c];
      v = gg * (0.5f * (1.0f + erff(h * 0.70710678118654752f)) + h * 0.3989422804014327f * __expf(-0.5f * h * h));
    } else if (mode == 1) {
      const float y = reinterpret_cast<const float*>(saved)[(size_t)r * lds_ + c];
      v = gg * (1.0f - y * y);
    } else {
      v = gg;
    }
  }
  dy[i] = (bf16_t)v;
}

extern "C" int vlm_tanh_fwd(const void* x_bf16, int ld_x, int M, int N, float* y, void* stream) {
  if (M == 0 || N == 0) return VLM_OK;
  if (!x_bf16 || !y || M < 0 || N < 0 || ld_x < N) return VLM_ERR_ARG;
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(vlm_ceil_div((long)M * N, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, ld_x, M, N, y);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_act_bwd(const void* g, int g_is_f32, int ld_g, const void* saved, int ld_saved, int mode, int M, int N, int Np, void* dy_bf16,
                           void* stream) {
  if (M == 0 || Np == 0) return VLM_OK;
  if (!g || !dy_bf16 || M < 0 || N < 0 || Np < N || ld_g < N || mode < 0 || mode > 2 || (mode != 2 && (!saved || ld_saved < N))) return VLM_ERR_ARG;
  const dim3 grid(vlm_ceil_div((long)M * Np, 256)), block(256);
  if (g_is_f32) hipLaunchKernelGGL((act_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)g, ld_g, saved, ld_saved, mode, M, N, Np, (bf16_t*)dy_bf16);
  else hipLaunchKernelGGL((act_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)g, ld_g, saved, ld_saved, mode, M, N, Np, (bf16_t*)dy_bf16);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// out[c] += sum_r a[r][c] for a matrix with a handful of columns (N <= 64): one workgroup, a wave per column in turn.
__global__ __launch_bounds__(256) void colsum_small_kernel(const bf16_t* __restrict__ a, int lda, int M, int N, float* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = wave; c < N; c += 4) {
    float s = 0.f;
    for (int r = lane; r < M; r += 64) s += (float)a[(size_t)r * lda + c];
    s = wave_sum(s);
    if (lane == 0) out[c] += s;
  }
}

extern "C" int vlm_colsum_small(const void* a_bf16, int lda, int M, int N, float* out, void* stream) {
  if (M == 0 || N == 0) return VLM_OK;
  if (!a_bf16 || !out || M < 0 || N < 0 || N > 64 || lda < N) return VLM_ERR_ARG;
  hipLaunchKernelGGL(colsum_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a_bf16, lda, M, N, out);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- hard negatives ------------------------------------------------------------------------------------------------------
// idx[d][i] ~ Categorical(softmax(sim_d[i, :]) with entry i removed), d = 0: sim_a, d = 1: sim_b; one wave per row, inverse CDF
// on the uniform u[d][i].  A row whose remaining weights are all zero (n == 1) returns i.
__global__ __launch_bounds__(64) void sample_negatives_kernel(const float* __restrict__ sim_a, int ar, int ac, const float* __restrict__ sim_b, int br,
                                                             int bc, int B, int n, const float* __restrict__ u, int64_t* __restrict__ idx) {
  const int d = blockIdx.x / B, i = blockIdx.x % B, lane = threadIdx.x;
  const float* s = d ? sim_b + (size_t)i * br : sim_a + (size_t)i * ar;
  const int cs = d ? bc : ac;  // column stride: a matrix may be the transposed view of the other
  float m = -INFINITY;
  for (int j = lane; j < n; j += 64) if (j != i) m = fmaxf(m, s[(size_t)j * cs]);
  m = wave_max(m);
  float z = 0.f;
  for (int j = lane; j < n; j += 64) if (j != i) z += __expf(s[(size_t)j * cs] - m);
  z = wave_sum(z);
  const float target = u[blockIdx.x] * z;
  // chunks of 64 candidates: inclusive scan inside the chunk, running total across chunks
  float base = 0.f;
  int pick = -1, last = -1;
  for (int j0 = 0; j0 < n && pick < 0; j0 += 64) {
    const int j = j0 + lane;
    const float w = (j < n && j != i) ? __expf(s[(size_t)j * cs] - m) : 0.f;
    float c = w;
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(c, o, 64); if (lane >= o) c += t; }
    const bool hit = w > 0.f && base + c > target;
    const unsigned long long ballot = __ballot(hit);
    const unsigned long long nz = __ballot(w > 0.f);
    if (nz) last = j0 + 63 - __builtin_clzll(nz);
    if (ballot) pick = j0 + __builtin_ctzll(ballot);
    base += __shfl(c, 63, 64);
  }
  if (pick < 0) pick = last >= 0 ? last : i;  // u * z rounded up to the total: the last candidate with weight
  if (lane == 0) idx[blockIdx.x] = pick;
}

extern "C" int vlm_sample_negatives(const float* sim_a, int a_row_stride, int a_col_stride, const float* sim_b, int b_row_stride, int b_col_stride,
                                    int B, int n, const float* u, int64_t* idx, void* stream) {
  if (B == 0) return VLM_OK;
  if (!sim_a || !sim_b || !u || !idx || B < 0 || n <= 0) return VLM_ERR_ARG;
  hipLaunchKernelGGL(sample_negatives_kernel, dim3(2 * B), dim3(64), 0, (hipStream_t)stream, sim_a, a_row_stride, a_col_stride, sim_b, b_row_stride,
                     b_col_stride, B, n, u, idx);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- weighted sum of scalar losses -----------------------------------------------------------------------------------------
// out[0] = sum_k w[k] * (*term[k]); count <= 8.
struct wsum_t { const float* term[8]; float w[8]; };
__global__ void weighted_sum_kernel(wsum_t a, int count, float* __restrict__ out) {
  float s = 0.f;
  for (int k = 0; k < count; ++k) s += a.w[k] * a.term[k][0];
  out[0] = s;
}

extern "C" int vlm_weighted_sum(const float* const* terms, const float* weights, int count, float* out, void* stream) {
  if (!terms || !weights || !out || count <= 0 || count > 8) return VLM_ERR_ARG;
  wsum_t a;
  for (int k = 0; k < count; ++k) { if (!terms[k]) return VLM_ERR_ARG; a.term[k] = terms[k]; a.w[k] = weights[k]; }
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, count, out);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---- row scatter: the backward of taking row sets of a feature matrix ----------------------------------------------------------
// dx [R, D] = sum over the sources k of (g_k placed at rows first_k + j * step_k, j < count_k), zero elsewhere: the gradient of the
// views a pass's result exposes (text_feats = x[:B T], cls rows = every T-th row, a leading block of samples ...).  Autograd builds
// it as one zero fill + one strided copy per view plus an addition per extra view; here it is one pass over dx.  One wave per row.
struct scatter_src_t { const void* g; int is_f32, ld, first, step, count; };
struct scatter_args_t { scatter_src_t s[4]; int n; };

template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(T* __restrict__ dx, int ldx, int R, int D, scatter_args_t a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  int src_row[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    src_row[k] = -1;
    if (k < a.n) {
      const int d = row - a.s[k].first;
      if (d >= 0 && d % a.s[k].step == 0 && d / a.s[k].step < a.s[k].count) src_row[k] = d / a.s[k].step;
    }
  }
  for (int c = lane * 4; c < D; c += 256) {
    float4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (src_row[k] < 0) continue;
      if (a.s[k].is_f32) {
        const float4 g = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.s[k].g) + (size_t)src_row[k] * a.s[k].ld + c);
        v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
      } else {
        const bf16x4 g = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(a.s[k].g) + (size_t)src_row[k] * a.s[k].ld + c);
        v.x += (float)g[0]; v.y += (float)g[1]; v.z += (float)g[2]; v.w += (float)g[3];
      }
    }
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(dx) + (size_t)row * ldx + c) = v;
    } else {
      bf16x4 o;
      o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(dx) + (size_t)row * ldx + c) = o;
    }
  }
}

extern "C" int vlm_scatter_rows(void* dx, int dx_is_f32, int ld_dx, int R, int D, const vlm_scatter_src_t* src, int n_src, void* stream) {
  if (R == 0) return VLM_OK;
  if (!dx || R < 0 || D <= 0 || (D & 3) || ld_dx < D || (ld_dx & 3) || n_src < 0 || n_src > 4 || (n_src && !src)) return VLM_ERR_ARG;
  scatter_args_t a;
  a.n = n_src;
  for (int k = 0; k < n_src; ++k) {
    if (!src[k].g || src[k].ld < D || (src[k].ld & 3) || src[k].row_step <= 0 || src[k].count < 0 || src[k].first_row < 0) return VLM_ERR_ARG;
    a.s[k] = {src[k].g, src[k].g_is_f32, src[k].ld, src[k].first_row, src[k].row_step, src[k].count};
  }
  const dim3 grid((R + 3) / 4), block(256);
  if (dx_is_f32) hipLaunchKernelGGL((scatter_rows_kernel<float>), grid, block, 0, (hipStream_t)stream, (float*)dx, ld_dx, R, D, a);
  else hipLaunchKernelGGL((scatter_rows_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (bf16_t*)dx, ld_dx, R, D, a);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
