// Flat-buffer elementwise kernels: fused AdamW step (K16), fp32 -> bf16 shadow cast, patch im2col (K1 front end).
// All HBM-bound; 16-B vector accesses, grid-stride over float4 granules.
#include "vlm_common.h"

#define EW_THREADS 256
#ifndef EW_NT
#define EW_NT true
#endif

static int ew_grid(size_t n_vec) {
  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  size_t want = (n_vec + EW_THREADS - 1) / EW_THREADS;
  size_t cap = (size_t)cus * 8;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

// HuggingFace-transformers-4.x AdamW (reference vilt_utils.py:314-317 instantiates it; its source is not in
// /root/reference: "parity unpinned", restated from the published rule):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= step_size * m / (sqrt(v) + eps) ; then p -= lr*wd*p
// with step_size = lr*sqrt(1-b2^t)/(1-b1^t) computed on the host.  grad_scale folds the DDP 1/world average.
__global__ __launch_bounds__(EW_THREADS) void adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           bf16_t* __restrict__ pb, size_t n4, float lr, float b1,
                                                           float b2, float eps, float wd, float step_size,
                                                           float grad_scale, int zero_grad) {
  for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * EW_THREADS) {
    // gradient and moments stream through once per step: non-temporal (EW_NT), so that they do not push the parameters and
    // their bf16 shadows -- which the next forward pass reads first -- out of the caches
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];  // (nt on the fp32 master as well: inside the noise)
    f32x4 gg = EW_NT ? __builtin_nontemporal_load(reinterpret_cast<f32x4*>(g) + i) : reinterpret_cast<f32x4*>(g)[i];
    f32x4 mm = EW_NT ? __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i) : reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = EW_NT ? __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i) : reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gr = gg[r] * grad_scale;
      mm[r] = b1 * mm[r] + (1.0f - b1) * gr;
      vv[r] = b2 * vv[r] + (1.0f - b2) * gr * gr;
      const float denom = sqrtf(vv[r]) + eps;
      float x = pp[r] - step_size * (mm[r] / denom);
      if (wd != 0.0f) x = x - lr * wd * x;
      pp[r] = x;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    if (EW_NT) {
      __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m) + i);
      __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
    } else {
      reinterpret_cast<f32x4*>(m)[i] = mm;
      reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (zero_grad) reinterpret_cast<f32x4*>(g)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pb) {
      bf16x4 h = {(bf16_t)pp[0], (bf16_t)pp[1], (bf16_t)pp[2], (bf16_t)pp[3]};
      reinterpret_cast<bf16x4*>(pb)[i] = h;
    }
  }
}

__global__ __launch_bounds__(EW_THREADS) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                          size_t n4) {
  for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * EW_THREADS) {
    const f32x4 s = reinterpret_cast<const f32x4*>(src)[i];
    bf16x4 h = {(bf16_t)s[0], (bf16_t)s[1], (bf16_t)s[2], (bf16_t)s[3]};
    reinterpret_cast<bf16x4*>(dst)[i] = h;
  }
}

// image [B,3,H,W] fp32 (NCHW) -> patch matrix bf16 [B*rows_per_img, 3*P*P], column = c*P*P + i*P + j
// (the flattening of nn.Conv2d(3, D, k=P, s=P).weight, vision_transformer.py:714-728).  rows_per_img = lead +
// (H/P)*(W/P): `lead` zero rows in front of every image leave room for the cls token so that the GEMM output
// already has the [B, 1+patches, D] row layout of visual_embed (:974-975).
__global__ __launch_bounds__(EW_THREADS) void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out,
                                                            int B, int H, int W, int P, int lead) {
  const int gh = H / P, gw = W / P;
  const int K = 3 * P * P;
  const int k8 = K / 8;
  const int rows = lead + gh * gw;
  const size_t total = (size_t)B * rows * k8;
  for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * EW_THREADS) {
    const int kk = (int)(i % k8) * 8;
    const size_t rr = i / k8;
    const int r = (int)(rr % rows);
    const int b = (int)(rr / rows);
    bf16x8 h;
    if (r < lead) {
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (bf16_t)0.0f;
    } else {
      const int pidx = r - lead;
      const int py = pidx / gw, px = pidx % gw;
      const int c = kk / (P * P), ij = kk % (P * P);
      const int ii = ij / P, jj = ij % P;  // jj multiple of 8 (P % 8 == 0)
      const float* s = img + (((size_t)b * 3 + c) * H + (py * P + ii)) * W + px * P + jj;
      const f32x4 a = *reinterpret_cast<const f32x4*>(s);
      const f32x4 c2 = *reinterpret_cast<const f32x4*>(s + 4);
      h = (bf16x8){(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3],
                   (bf16_t)c2[0], (bf16_t)c2[1], (bf16_t)c2[2], (bf16_t)c2[3]};
    }
    *reinterpret_cast<bf16x8*>(out + rr * K + kk) = h;
  }
}

// G64 += (double)T32 : folds the fp32 MFMA result of one X^T X product into the float64 Gram accumulator
// (cache_gram_matrices.py:246-254 keeps the running sum in float64).
__global__ __launch_bounds__(EW_THREADS) void acc_f64_kernel(const float* __restrict__ src, double* __restrict__ dst,
                                                             size_t n4) {
  for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * EW_THREADS) {
    const f32x4 s = reinterpret_cast<const f32x4*>(src)[i];
    double* d = dst + 4 * i;
    d[0] += (double)s[0];
    d[1] += (double)s[1];
    d[2] += (double)s[2];
    d[3] += (double)s[3];
  }
}

extern "C" int vlm_accumulate_f32_f64(const float* src, double* dst, uint64_t n, void* stream) {
  if (n == 0) return VLM_OK;
  if (!src || !dst || (n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 31)) return VLM_ERR_ARG;
  const size_t n4 = n >> 2;
  hipLaunchKernelGGL(acc_f64_kernel, dim3(ew_grid(n4)), dim3(EW_THREADS), 0, (hipStream_t)stream, src, dst, n4);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, uint64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float step_size,
                              float grad_scale, int zero_grad, void* stream) {
  if (n == 0) return VLM_OK;
  if (!p || !g || !m || !v || (n & 3)) return VLM_ERR_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return VLM_ERR_ARG;
  if (p_bf16 && ((uintptr_t)p_bf16 & 7)) return VLM_ERR_ARG;
  const size_t n4 = n >> 2;
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n4)), dim3(EW_THREADS), 0, (hipStream_t)stream, p, g, m, v,
                     (bf16_t*)p_bf16, n4, lr, beta1, beta2, eps, weight_decay, step_size, grad_scale, zero_grad);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_cast_f32_bf16(const float* src, void* dst, uint64_t n, void* stream) {
  if (n == 0) return VLM_OK;
  if (!src || !dst || (n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return VLM_ERR_ARG;
  const size_t n4 = n >> 2;
  hipLaunchKernelGGL(cast_kernel, dim3(ew_grid(n4)), dim3(EW_THREADS), 0, (hipStream_t)stream, src, (bf16_t*)dst, n4);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// Backward of the word-embedding gather (BertEmbeddings.word_embeddings, vilt_module.py:63 / :1090: nn.Embedding with
// padding_idx = 0): dW[ids[t], :] += gy[t, :] for every token t whose id is not the padding index.  torch's
// embedding_dense_backward sorts the ids and builds a dense [vocab, D] gradient that autograd then adds to .grad (0.5 ms per
// step for 4 400 tokens); here one wave per token adds its row straight into the flat gradient buffer (float atomics on
// 256-B segments: the full-rate shape).
__global__ __launch_bounds__(EW_THREADS) void embedding_bwd_kernel(const float* __restrict__ gy, int ld, const int64_t* __restrict__ ids,
                                                                   int64_t n, int D, int64_t pad, float* __restrict__ dW, int ld_w,
                                                                   int64_t vocab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t t = (int64_t)blockIdx.x * (EW_THREADS / 64) + wave; t < n; t += (int64_t)gridDim.x * (EW_THREADS / 64)) {
    const int64_t id = ids[t];
    if (id == pad || id < 0 || id >= vocab) continue;  // wave-uniform; an id outside the table adds nowhere (never out of bounds)
    const float* src = gy + t * ld;
    float* dst = dW + id * ld_w;
    for (int c = lane; c < D; c += 64) atomicAdd(dst + c, src[c]);
  }
}

extern "C" int vlm_embedding_bwd(const float* gy, int ld, const int64_t* ids, int64_t n, int D, int64_t padding_idx, float* dW,
                                 int ld_w, int64_t vocab, void* stream) {
  if (n == 0 || D == 0) return VLM_OK;
  if (!gy || !ids || !dW || n < 0 || D < 0 || ld < D || ld_w < D || vocab <= 0) return VLM_ERR_ARG;
  int64_t blocks = (n + (EW_THREADS / 64) - 1) / (EW_THREADS / 64);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, gy, ld, ids, n, D,
                     padding_idx, dW, ld_w, vocab);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// DropPath row scales (timm drop_path as used at vision_transformer.py:586,:603): per-sample bernoulli(keep)/keep
// expanded to the segment-major token rows of a pass; u holds one uniform [0,1) draw per sample.
__global__ __launch_bounds__(EW_THREADS) void droppath_rows_kernel(const float* __restrict__ u, float keep, float inv_keep,
                                                                   int B, int n0, int n1, int base0, int base1,
                                                                   float* __restrict__ out) {
  const int nt = B * n0, total = nt + B * n1;
  for (int i = blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += gridDim.x * EW_THREADS) {
    int b, row;
    if (i < nt) {
      b = i / n0;
      row = base0 + i;
    } else {
      const int j = i - nt;
      b = j / n1;
      row = base1 + j;
    }
    out[row] = u[b] < keep ? inv_keep : 0.0f;
  }
}

extern "C" int vlm_droppath_rows(const float* u, float keep, int B, int n0, int n1, int base0, int base1, float* out,
                                 void* stream) {
  if (B <= 0 || n0 + n1 <= 0) return VLM_OK;
  if (!u || !out || n0 < 0 || n1 < 0 || base0 < 0 || base1 < 0 || !(keep > 0.0f) || keep > 1.0f) return VLM_ERR_ARG;
  const size_t total = (size_t)B * (n0 + n1);
  hipLaunchKernelGGL(droppath_rows_kernel, dim3(ew_grid(total)), dim3(EW_THREADS), 0, (hipStream_t)stream, u, keep,
                     1.0f / keep, B, n0, n1, base0, base1, out);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// All DropPath sites of a pass in ONE launch: out[site][row] from u[site][b] and keep[site] (blockIdx.y = site).
// u1 (optional) supplies the image segment's own draws (two unimodal passes sharing a launch keep independent masks).
__global__ __launch_bounds__(EW_THREADS) void droppath_sites_kernel(const float* __restrict__ u0, const float* __restrict__ u1,
                                                                    const float* __restrict__ keep, int B, int n0, int n1,
                                                                    int base0, int base1, int rows, float* __restrict__ out) {
  const int site = blockIdx.y;
  const float kp = keep[site], inv = 1.0f / kp;
  const float* ua = u0 + (size_t)site * B;
  const float* ub = (u1 ? u1 : u0) + (size_t)site * B;
  float* o = out + (size_t)site * rows;
  const int nt = B * n0, total = nt + B * n1;
  for (int i = blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += gridDim.x * EW_THREADS) {
    if (i < nt) {
      o[base0 + i] = ua[i / n0] < kp ? inv : 0.0f;
    } else {
      const int j = i - nt;
      o[base1 + j] = ub[j / n1] < kp ? inv : 0.0f;
    }
  }
}

extern "C" int vlm_droppath_sites(const float* u0, const float* u1, const float* keep, int n_sites, int B, int n0, int n1,
                                  int base0, int base1, int rows, float* out, void* stream) {
  if (n_sites <= 0 || B <= 0 || n0 + n1 <= 0) return VLM_OK;
  if (!u0 || !keep || !out || n0 < 0 || n1 < 0 || base0 < 0 || base1 < 0 || rows < base0 + B * n0 || rows < base1 + B * n1)
    return VLM_ERR_ARG;
  const size_t total = (size_t)B * (n0 + n1);
  int gx = (int)((total + EW_THREADS - 1) / EW_THREADS);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(droppath_sites_kernel, dim3(gx, n_sites), dim3(EW_THREADS), 0, (hipStream_t)stream, u0, u1, keep, B,
                     n0, n1, base0, base1, rows, out);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// Batched bf16 transpose (weights -> K-contiguous operands of the dgrad GEMMs): one workgroup per 64x64 tile of any of
// the listed matrices, through LDS (row stride 66 elements: the column gather walks 33 banks).  Ragged edges guarded.
__global__ __launch_bounds__(256) void transpose_tiles_kernel(const vlm_transpose_tile_t* __restrict__ tiles, int n_tiles) {
  __shared__ bf16_t tile[64][66];
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const vlm_transpose_tile_t j = tiles[t];
    const bf16_t* src = reinterpret_cast<const bf16_t*>(j.src);
    bf16_t* dst = reinterpret_cast<bf16_t*>(j.dst);
    const int r0 = j.tile_row * 64, c0 = j.tile_col * 64;
    const int lr = threadIdx.x >> 2, lc = (threadIdx.x & 3) * 16;
    // whole tiles of 16-B aligned matrices move as 16-B vectors on the global side (the LDS side stays 2-B: the
    // padded row stride of 66 elements is what makes the column walk conflict-free); ragged tiles go element-wise
    const bool fast = r0 + 64 <= j.rows && c0 + 64 <= j.cols && (j.cols & 7) == 0 && (j.rows & 7) == 0 &&
                      ((j.src | j.dst) & 15) == 0;
    if (fast) {
      const bf16x8* sp = reinterpret_cast<const bf16x8*>(src + (size_t)(r0 + lr) * j.cols + c0 + lc);
      const bf16x8 v0 = sp[0], v1 = sp[1];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        tile[lr][lc + e] = v0[e];
        tile[lr][lc + 8 + e] = v1[e];
      }
    } else {
      const int r = r0 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = c0 + lc + e;
        tile[lr][lc + e] = (r < j.rows && c < j.cols) ? src[(size_t)r * j.cols + c] : (bf16_t)0.0f;
      }
    }
    __syncthreads();
    const int c = c0 + lr;  // output row = source column
    if (fast) {
      bf16x8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o0[e] = tile[lc + e][lr];
        o1[e] = tile[lc + 8 + e][lr];
      }
      bf16x8* dp = reinterpret_cast<bf16x8*>(dst + (size_t)c * j.rows + r0 + lc);
      dp[0] = o0;
      dp[1] = o1;
    } else if (c < j.cols) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = r0 + lc + e;
        if (r < j.rows) dst[(size_t)c * j.rows + r] = tile[lc + e][lr];
      }
    }
    __syncthreads();
  }
}

extern "C" int vlm_transpose_bf16_tiles(const vlm_transpose_tile_t* tiles_dev, int n_tiles, void* stream) {
  if (n_tiles == 0) return VLM_OK;
  if (!tiles_dev || n_tiles < 0) return VLM_ERR_ARG;
  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  const int grid = n_tiles < cus * 16 ? n_tiles : cus * 16;
  hipLaunchKernelGGL(transpose_tiles_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tiles_dev, n_tiles);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_patch_im2col(const float* image, void* patches, int B, int H, int W, int P, int lead_rows,
                                void* stream) {
  if (B == 0) return VLM_OK;
  if (!image || !patches || B < 0 || P <= 0 || (P & 7) || (H % P) || (W % P) || (W & 3) || lead_rows < 0)
    return VLM_ERR_ARG;
  if (((uintptr_t)image & 15) || ((uintptr_t)patches & 15)) return VLM_ERR_ARG;
  const size_t total = (size_t)B * (lead_rows + (H / P) * (W / P)) * (3 * P * P / 8);
  hipLaunchKernelGGL(im2col_kernel, dim3(ew_grid(total)), dim3(EW_THREADS), 0, (hipStream_t)stream, image,
                     (bf16_t*)patches, B, H, W, P, lead_rows);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
