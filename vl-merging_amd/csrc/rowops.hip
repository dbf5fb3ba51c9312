// Row-wise bandwidth-bound kernels of the VLMo block: LayerNorm fwd/bwd (K5), LayerScale backward,
// bias-gradient column sums.  One wave (64 lanes) owns one token row at a time; lane l owns the float4
// column granules l*4 + 256*u, so per-column reductions over rows (dgamma/dbeta/dbias) stay in registers and
// cross-lane traffic is only the two row statistics.  All loads/stores are 16-B (fp32) or 8-B (bf16x4) vectors.
//
// Reference op sites: nn.LayerNorm(eps=1e-6) vision_transformer.py:831 / apply_ln :495-523; BertEmbeddings
// LayerNorm(1e-12) vilt_module.py:63; LayerScale `x + drop_path(gamma * branch)` :586,:603; the backward of
// each is what torch autograd derives for those ops.
#include "vlm_common.h"

#ifndef ROW_NT
#define ROW_NT true  // last-use rows of the backward row kernels are loaded non-temporally
#endif
#define ROW_THREADS 256
#define ROW_WAVES 4

// NT: the row is read for the last time (a saved activation in the backward pass): non-temporal load
template <int MAXU, bool IN_BF16, bool NT = false>
__device__ __forceinline__ void load_row(const void* base, size_t row, int ld, int D, int lane, f32x4 (&v)[MAXU]) {
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int c = lane * 4 + 256 * u;
    if (c < D) {
      if (IN_BF16) {
        const bf16x4* q = reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(base) + row * ld + c);
        const bf16x4 h = NT ? __builtin_nontemporal_load(q) : *q;
        v[u] = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      } else {
        const f32x4* q = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + row * ld + c);
        v[u] = NT ? __builtin_nontemporal_load(q) : *q;
      }
    } else {
      v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
}

template <int MAXU>
__device__ __forceinline__ void load_vec(const float* p, int D, int lane, f32x4 (&v)[MAXU], float fill) {
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int c = lane * 4 + 256 * u;
    v[u] = (c < D && p) ? *reinterpret_cast<const f32x4*>(p + c) : (f32x4){fill, fill, fill, fill};
  }
}

// ------------------------------------------------------------------------------------------- LayerNorm forward
template <int MAXU, bool OUT_F32>
__global__ __launch_bounds__(ROW_THREADS) void ln_fwd_kernel(const float* __restrict__ x, int ldx, int M, int D,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             void* __restrict__ y, int ldy,
                                                             float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 g[MAXU], b[MAXU];
  load_vec<MAXU>(gamma, D, lane, g, 1.0f);
  load_vec<MAXU>(beta, D, lane, b, 0.0f);
  const float invD = 1.0f / (float)D;
  for (size_t row = (size_t)blockIdx.x * ROW_WAVES + wave; row < (size_t)M; row += (size_t)gridDim.x * ROW_WAVES) {
    f32x4 v[MAXU];
    load_row<MAXU, false>(x, row, ldx, D, lane, v);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int c = lane * 4 + 256 * u;
      if (c < D) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = v[u][r] - mean;
          q += d * d;
        }
      }
    }
    const float var = wave_sum(q) * invD;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (stats && lane == 0) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = rstd;
    }
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int c = lane * 4 + 256 * u;
      if (c < D) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (v[u][r] - mean) * rstd * g[u][r] + b[u][r];
        if (OUT_F32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + row * ldy + c) = o;
        } else {
          bf16x4 h = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(y) + row * ldy + c) = h;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ LayerNorm backward
// dx[m,:] = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ; dx_out = dx (+ dres) ; column sums by
// one atomicAdd per column per block (fp32 grads are accumulated across passes anyway).
// SCALE (vlm_layernorm_bwd_scale): the row this kernel has just produced is the residual-stream gradient the LayerScale backward
// of the branch BELOW this LayerNorm reads next (Block.forward: x = x + gamma_1 * attn(norm1(x)); x = x + gamma_2 * mlp(norm2(x)),
// vision_transformer.py:586,:603) -- taken while it is in registers (ls_t: that branch's saved output, its gamma and row
// scale, its bf16 gradient out, its two column-sum targets) the second kernel's 4-B-per-element read of dx and its launch go.
struct ls_t {
  const bf16_t* y;
  int ldy;
  const float* gamma;
  const float* row_scale;
  bf16_t* dy;
  int lddy;
  float* dgamma;
  float* dbias;
  float* partials;
};
// SCALE == 2 (round 5): that LayerScale is folded into the branch's output projection (layerscale.hip) -- ls.y / ls.gamma are
// absent, only dy = bf16(row_scale * dx) and its column sums are taken with the row (one more accumulator set instead of two).
template <int MAXU, bool DY_BF16, int SCALE = 0>
__global__ __launch_bounds__(ROW_THREADS, MAXU <= 3 ? (SCALE ? 3 : 4) : 2) void ln_bwd_kernel(const void* __restrict__ dy, int lddy,
                                                             const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, int M, int D,
                                                             const float* __restrict__ dres, int lddres,
                                                             float* __restrict__ dx, int lddx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ partials, const ls_t ls) {
  __shared__ float red[ROW_WAVES][SCALE ? 4 : 2][MAXU * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 g[SCALE ? 1 : MAXU], ag[MAXU], ab[MAXU];
  f32x4 g2[1], ag2[SCALE == 1 ? MAXU : 1], ab2[SCALE ? MAXU : 1];
  if (SCALE) g[0] = (f32x4){1.f, 1.f, 1.f, 1.f};
  else load_vec<SCALE ? 1 : MAXU>(gamma, D, lane, g, 1.0f);
#pragma unroll
  for (int u = 0; u < MAXU; ++u) ag[u] = ab[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < (SCALE ? MAXU : 1); ++u) ab2[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < (SCALE == 1 ? MAXU : 1); ++u) ag2[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float invD = 1.0f / (float)D;
  for (size_t row = (size_t)blockIdx.x * ROW_WAVES + wave; row < (size_t)M; row += (size_t)gridDim.x * ROW_WAVES) {
    f32x4 v[MAXU], d[MAXU], rs[MAXU];
    load_row<MAXU, false, ROW_NT>(x, row, ldx, D, lane, v);
    load_row<MAXU, DY_BF16>(dy, row, lddy, D, lane, d);  // (nt here too: no gain in the step, ln_bwd alone 101 -> 105 us)
    // the residual-path gradient is fetched with the row, not after the two wave reductions (its latency used to sit
    // between the reduction and the store of every row)
    if (dres) load_row<MAXU, false, ROW_NT>(dres, row, lddres, D, lane, rs);
    f32x4 yy[SCALE == 1 ? MAXU : 1];
    float srow = 1.0f;
    if (SCALE) {
      if (SCALE == 1) load_row<SCALE == 1 ? MAXU : 1, true, ROW_NT>(ls.y, row, ls.ldy, D, lane, yy);
      if (ls.row_scale) srow = ls.row_scale[row];
    }
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int c = lane * 4 + 256 * u;
      if (c < D) {
        // (SCALE: this LayerNorm's gamma too is re-read per row -- another twelve registers off the loop-carried set)
        const f32x4 gu = (SCALE && gamma) ? *reinterpret_cast<const f32x4*>(gamma + c) : g[SCALE ? 0 : u];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xh = (v[u][r] - mean) * rstd;
          const float gg = d[u][r] * gu[r];
          s1 += gg;
          s2 += gg * xh;
          ag[u][r] += d[u][r] * xh;
          ab[u][r] += d[u][r];
          v[u][r] = xh;
          d[u][r] = gg;
        }
      }
    }
    const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int c = lane * 4 + 256 * u;
      if (c < D) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = rstd * (d[u][r] - m1 - v[u][r] * m2);
        if (dres) o += rs[u];
        *reinterpret_cast<f32x4*>(dx + row * lddx + c) = o;
        if (SCALE) {  // the same operations, in the same order, as scale_bwd_kernel on the stored row
          // (the branch's gamma is re-read per row from the cache: twelve registers held across the loop cost the third workgroup per CU)
          const int c2 = lane * 4 + 256 * u;
          g2[0] = (SCALE == 1 && ls.gamma) ? *reinterpret_cast<const f32x4*>(ls.gamma + c2) : (f32x4){1.f, 1.f, 1.f, 1.f};
          bf16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sd = srow * o[r];
            const float q = sd * g2[0][r];
            if (SCALE == 1) ag2[SCALE == 1 ? u : 0][r] += sd * yy[SCALE == 1 ? u : 0][r];
            h[r] = (bf16_t)q;
            ab2[u][r] += (float)h[r];
          }
          *reinterpret_cast<bf16x4*>(ls.dy + row * ls.lddy + c) = h;
        }
      }
    }
  }
  if (SCALE) {
#pragma unroll
    for (int u = 0; u < (SCALE ? MAXU : 1); ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wave][SCALE ? 2 : 0][u * 256 + lane * 4 + r] = SCALE == 1 ? ag2[SCALE == 1 ? u : 0][r] : 0.f;
        red[wave][SCALE ? 3 : 1][u * 256 + lane * 4 + r] = ab2[u][r];
      }
  }
  if (dgamma || dbeta || SCALE) {
#pragma unroll
    for (int u = 0; u < MAXU; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wave][0][u * 256 + lane * 4 + r] = ag[u][r];
        red[wave][1][u * 256 + lane * 4 + r] = ab[u][r];
      }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += ROW_THREADS) {
      float sg = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < ROW_WAVES; ++w) {
        sg += red[w][0][c];
        sb += red[w][1][c];
      }
      if (partials) {  // contention-free: per-block partial sums, folded by colreduce_kernel
        partials[((size_t)blockIdx.x * 2 + 0) * D + c] = sg;
        partials[((size_t)blockIdx.x * 2 + 1) * D + c] = sb;
      } else {
        if (dgamma) atomicAdd(dgamma + c, sg);
        if (dbeta) atomicAdd(dbeta + c, sb);
      }
      if (SCALE) {
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int w = 0; w < ROW_WAVES; ++w) {
          tg += red[w][SCALE ? 2 : 0][c];
          tb += red[w][SCALE ? 3 : 1][c];
        }
        if (ls.partials) {
          ls.partials[((size_t)blockIdx.x * 2 + 0) * D + c] = tg;
          ls.partials[((size_t)blockIdx.x * 2 + 1) * D + c] = tb;
        } else {
          if (ls.dgamma) atomicAdd(ls.dgamma + c, tg);
          if (ls.dbias) atomicAdd(ls.dbias + c, tb);
        }
      }
    }
  }
}

// out0[c] += sum_b partials[b][0][c] ; out1[c] += sum_b partials[b][1][c].  The per-column float atomics of ~1500
// workgroups all hit the same 6 KiB (measured ~14x below the un-contended atomic rate: 90 us of a 180 us kernel);
// plain partial stores + this fold cost ~3 us.
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ partials, int nblocks, int D,
                                                        float* __restrict__ out0, float* __restrict__ out1) {
  // grid.x covers the 2*D columns, grid.y = COLRED_SPLITS slices of the workgroup axis; 32 atomics per address
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * D) return;
  const int which = i / D, c = i - which * D;
  float* out = which ? out1 : out0;
  if (!out) return;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float s0 = 0.f, s1 = 0.f;
  int b = b0;
  for (; b + 1 < b1; b += 2) {
    s0 += partials[((size_t)(b + 0) * 2 + which) * D + c];
    s1 += partials[((size_t)(b + 1) * 2 + which) * D + c];
  }
  if (b < b1) s0 += partials[((size_t)b * 2 + which) * D + c];
  if (b0 < b1) atomicAdd(out + c, s0 + s1);
}

// Several folds in one launch (blockIdx.z = job): a transformer block's backward runs four row kernels whose partials
// can wait until the block is done, saving three ~10 us launches per block evaluation.
struct fold_jobs_t {
  vlm_fold_job_t j[VLM_MAX_FOLD_JOBS];
};
__global__ __launch_bounds__(256) void colreduce_batch_kernel(const fold_jobs_t jobs) {
  const vlm_fold_job_t job = jobs.j[blockIdx.z];
  const int D = job.D, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * D) return;
  const int which = i / D, c = i - which * D;
  float* out = which ? job.out1 : job.out0;
  if (!out) return;
  const int per = (job.nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(job.nblocks, b0 + per);
  float s0 = 0.f, s1 = 0.f;
  int b = b0;
  for (; b + 1 < b1; b += 2) {
    s0 += job.partials[((size_t)(b + 0) * 2 + which) * D + c];
    s1 += job.partials[((size_t)(b + 1) * 2 + which) * D + c];
  }
  if (b < b1) s0 += job.partials[((size_t)b * 2 + which) * D + c];
  if (b0 < b1) atomicAdd(out + c, s0 + s1);
}

extern "C" int vlm_colreduce_batch(const vlm_fold_job_t* jobs, int n_jobs, void* stream) {
  if (n_jobs == 0) return VLM_OK;
  if (!jobs || n_jobs < 0 || n_jobs > VLM_MAX_FOLD_JOBS) return VLM_ERR_ARG;
  fold_jobs_t a;
  int maxD = 0;
  for (int i = 0; i < n_jobs; ++i) {
    if (!jobs[i].partials || jobs[i].D <= 0 || jobs[i].nblocks < 0) return VLM_ERR_ARG;
    a.j[i] = jobs[i];
    if (jobs[i].D > maxD) maxD = jobs[i].D;
  }
  hipLaunchKernelGGL(colreduce_batch_kernel, dim3((2 * maxD + 255) / 256, 32, n_jobs), dim3(256), 0, (hipStream_t)stream, a);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---------------------------------------------------------------------------------------- LayerScale backward
// forward was x_new = x + rs[m]*gamma[n]*y[m,n]  (y = branch output incl. its bias, saved in bf16)
//   dy[m,n]   = rs[m]*gamma[n]*dx[m,n]          (bf16, feeds the dgrad/wgrad GEMMs)
//   dgamma[n] += sum_m rs[m]*dx[m,n]*y[m,n]  ;  dbias[n] += sum_m dy[m,n]
// CAST (y == NULL, round 5): the LayerScale is folded into the branch's output projection (layerscale.hip) -- only
// dy = bf16(rs * dx) and its column sums remain (6 instead of 8 B per element, one accumulator set).
template <int MAXU, bool CAST = false>
__global__ __launch_bounds__(ROW_THREADS) void scale_bwd_kernel(const float* __restrict__ dx, int lddx,
                                                                const bf16_t* __restrict__ y, int ldy,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ row_scale, int M, int D,
                                                                bf16_t* __restrict__ dy, int lddy,
                                                                float* __restrict__ dgamma,
                                                                float* __restrict__ dbias,
                                                                float* __restrict__ partials) {
  __shared__ float red[ROW_WAVES][2][MAXU * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 g[MAXU], ag[CAST ? 1 : MAXU], ab[MAXU];
  load_vec<MAXU>(gamma, D, lane, g, 1.0f);
#pragma unroll
  for (int u = 0; u < MAXU; ++u) ab[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < (CAST ? 1 : MAXU); ++u) ag[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (size_t row = (size_t)blockIdx.x * ROW_WAVES + wave; row < (size_t)M; row += (size_t)gridDim.x * ROW_WAVES) {
    f32x4 d[MAXU], yy[CAST ? 1 : MAXU];
    load_row<MAXU, false>(dx, row, lddx, D, lane, d);
    if (!CAST) load_row<CAST ? 1 : MAXU, true, ROW_NT>(y, row, ldy, D, lane, yy);
    const float rs = row_scale ? row_scale[row] : 1.0f;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int c = lane * 4 + 256 * u;
      if (c < D) {
        bf16x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sd = rs * d[u][r];
          const float o = sd * g[u][r];
          if (!CAST) ag[CAST ? 0 : u][r] += sd * yy[CAST ? 0 : u][r];
          h[r] = (bf16_t)o;
          ab[u][r] += (float)h[r];  // the bias gradient the GEMMs see is the rounded dy
        }
        *reinterpret_cast<bf16x4*>(dy + row * lddy + c) = h;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < MAXU; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[wave][0][u * 256 + lane * 4 + r] = CAST ? 0.f : ag[CAST ? 0 : u][r];
      red[wave][1][u * 256 + lane * 4 + r] = ab[u][r];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += ROW_THREADS) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < ROW_WAVES; ++w) {
      sg += red[w][0][c];
      sb += red[w][1][c];
    }
    if (partials) {
      partials[((size_t)blockIdx.x * 2 + 0) * D + c] = sg;
      partials[((size_t)blockIdx.x * 2 + 1) * D + c] = sb;
    } else {
      if (dgamma) atomicAdd(dgamma + c, sg);
      if (dbias) atomicAdd(dbias + c, sb);
    }
  }
}

// ------------------------------------------------------------------------------------------------ column sum
// out[n] += sum_m a[m, n]  for bf16 a; lane owns 8 columns (16-B loads), the block's 4 waves split the rows.
__global__ __launch_bounds__(ROW_THREADS) void colsum_kernel(const bf16_t* __restrict__ a, int lda, int M, int N,
                                                             float* __restrict__ out) {
  __shared__ float red[ROW_WAVES][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + lane) * 8;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  if (c0 < N) {
    for (size_t row = (size_t)blockIdx.y * ROW_WAVES + wave; row < (size_t)M; row += (size_t)gridDim.y * ROW_WAVES) {
      const bf16x8 h = *reinterpret_cast<const bf16x8*>(a + row * lda + c0);
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] += (float)h[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) red[wave][lane * 8 + r] = acc[r];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += ROW_THREADS) {
    const int n = blockIdx.x * 512 + c;
    if (n < N) atomicAdd(out + n, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

static int row_grid(int M) {
  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  int want = (M + ROW_WAVES - 1) / ROW_WAVES;
  int cap = cus * 8;
  return want < cap ? (want > 0 ? want : 1) : cap;
}

extern "C" int vlm_layernorm_fwd(const float* x, int ldx, int M, int D, const float* gamma, const float* beta,
                                 float eps, void* y, int ldy, int y_is_f32, float* stats, void* stream) {
  if (M == 0) return VLM_OK;
  if (!x || !y || M < 0 || D <= 0 || (D & 3) || (ldx & 3) || (ldy & 3)) return VLM_ERR_ARG;
  if (D > 1024) return VLM_ERR_UNSUPPORTED;
  dim3 grid(row_grid(M)), block(ROW_THREADS);
  hipStream_t s = (hipStream_t)stream;
#define LN_FWD(U, F) hipLaunchKernelGGL((ln_fwd_kernel<U, F>), grid, block, 0, s, x, ldx, M, D, gamma, beta, eps, y, ldy, stats)
  if (D <= 256) { if (y_is_f32) LN_FWD(1, true); else LN_FWD(1, false); }
  else if (D <= 768) { if (y_is_f32) LN_FWD(3, true); else LN_FWD(3, false); }  // D = 768: no dead fourth register slot
  else { if (y_is_f32) LN_FWD(4, true); else LN_FWD(4, false); }
#undef LN_FWD
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

static int layernorm_bwd_impl(const void* dy, int lddy, int dy_is_f32, const float* x, int ldx,
                              const float* stats, const float* gamma, int M, int D, const float* dres,
                              int lddres, float* dx, int lddx, float* dgamma, float* dbeta, float* workspace,
                              size_t workspace_bytes, int* deferred_blocks, void* stream, const vlm_layerscale_t* sc) {
  if (M == 0) return VLM_OK;
  if (!dy || !x || !stats || !dx || M < 0 || D <= 0 || (D & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3) ||
      (dres && (lddres & 3)))
    return VLM_ERR_ARG;
  if (D > 1024) return VLM_ERR_UNSUPPORTED;
  int g = row_grid(M);
  // one round of resident workgroups: D <= 768 runs at 4 waves/SIMD (<= 128 VGPRs) = 4 workgroups per CU
  const int resident = (D <= 768 ? 4 : 2) * (vlm_device_cus() > 0 ? vlm_device_cus() : 256);
  if (g > resident) g = resident;
  if (sc && D <= 768 && g > resident * 3 / 4) g = resident * 3 / 4;  // the fused forms hold one or two more accumulator sets: three workgroups per CU (140 / 167 registers)
  float* part = (workspace && workspace_bytes >= (size_t)g * 2 * D * sizeof(float) && (dgamma || dbeta)) ? workspace : nullptr;
  dim3 grid(g), block(ROW_THREADS);
  hipStream_t s = (hipStream_t)stream;
  ls_t ls = {};
  if (sc) {
    if (!sc->dy || (sc->y && (sc->ldy & 3)) || (sc->lddy & 3)) return VLM_ERR_ARG;
    if (!sc->y && (sc->gamma || sc->dgamma)) return VLM_ERR_ARG;  // the folded form (y == NULL) has no column scale of its own
    const bool want = sc->dgamma || sc->dbias;
    float* part2 = (sc->workspace && sc->workspace_bytes >= (size_t)g * 2 * D * sizeof(float) && want) ? sc->workspace : nullptr;
    if (deferred_blocks && want && !part2) return VLM_ERR_ARG;  // deferral needs both partial workspaces
    ls.y = reinterpret_cast<const bf16_t*>(sc->y); ls.ldy = sc->ldy; ls.gamma = sc->gamma; ls.row_scale = sc->row_scale;
    ls.dy = reinterpret_cast<bf16_t*>(sc->dy); ls.lddy = sc->lddy; ls.dgamma = sc->dgamma; ls.dbias = sc->dbias; ls.partials = part2;
  }
#define LN_BWD(U, B)                                                                                                             \
  do {                                                                                                                           \
    if (sc && sc->y) hipLaunchKernelGGL((ln_bwd_kernel<U, B, 1>), grid, block, 0, s, dy, lddy, x, ldx, stats, gamma, M, D, dres, lddres, dx, lddx, dgamma, dbeta, part, ls); \
    else if (sc) hipLaunchKernelGGL((ln_bwd_kernel<U, B, 2>), grid, block, 0, s, dy, lddy, x, ldx, stats, gamma, M, D, dres, lddres, dx, lddx, dgamma, dbeta, part, ls); \
    else hipLaunchKernelGGL((ln_bwd_kernel<U, B, 0>), grid, block, 0, s, dy, lddy, x, ldx, stats, gamma, M, D, dres, lddres, dx, lddx, dgamma, dbeta, part, ls); \
  } while (0)
  if (D <= 256) { if (dy_is_f32) LN_BWD(1, false); else LN_BWD(1, true); }
  else if (D <= 768) { if (dy_is_f32) LN_BWD(3, false); else LN_BWD(3, true); }
  else { if (dy_is_f32) LN_BWD(4, false); else LN_BWD(4, true); }
#undef LN_BWD
  VLM_CHECK_LAUNCH();
  if (deferred_blocks) {
    if (!part && (dgamma || dbeta)) return VLM_ERR_ARG;  // deferral needs the partial workspace
    *deferred_blocks = (part || ls.partials) ? g : 0;     // either set of partials: a frozen LayerNorm still leaves the LayerScale's
  } else {
    if (part) {
      hipLaunchKernelGGL(colreduce_kernel, dim3((2 * D + 255) / 256, 32), dim3(256), 0, s, part, g, D, dgamma, dbeta);
      VLM_CHECK_LAUNCH();
    }
    if (ls.partials) {
      hipLaunchKernelGGL(colreduce_kernel, dim3((2 * D + 255) / 256, 32), dim3(256), 0, s, ls.partials, g, D, ls.dgamma, ls.dbias);
      VLM_CHECK_LAUNCH();
    }
  }
  return VLM_OK;
}

extern "C" int vlm_layernorm_bwd(const void* dy, int lddy, int dy_is_f32, const float* x, int ldx,
                                 const float* stats, const float* gamma, int M, int D, const float* dres,
                                 int lddres, float* dx, int lddx, float* dgamma, float* dbeta, float* workspace,
                                 size_t workspace_bytes, int* deferred_blocks, void* stream) {
  return layernorm_bwd_impl(dy, lddy, dy_is_f32, x, ldx, stats, gamma, M, D, dres, lddres, dx, lddx, dgamma, dbeta, workspace,
                            workspace_bytes, deferred_blocks, stream, nullptr);
}

extern "C" int vlm_layernorm_bwd_scale(const void* dy, int lddy, int dy_is_f32, const float* x, int ldx,
                                       const float* stats, const float* gamma, int M, int D, const float* dres,
                                       int lddres, float* dx, int lddx, float* dgamma, float* dbeta, float* workspace,
                                       size_t workspace_bytes, const vlm_layerscale_t* scale, int* deferred_blocks, void* stream) {
  if (!scale) return VLM_ERR_ARG;
  return layernorm_bwd_impl(dy, lddy, dy_is_f32, x, ldx, stats, gamma, M, D, dres, lddres, dx, lddx, dgamma, dbeta, workspace,
                            workspace_bytes, deferred_blocks, stream, scale);
}

extern "C" int vlm_layerscale_bwd(const float* dx, int lddx, const void* y, int ldy, const float* gamma,
                                  const float* row_scale, int M, int D, void* dy, int lddy, float* dgamma,
                                  float* dbias, float* workspace, size_t workspace_bytes, int* deferred_blocks,
                                  void* stream) {
  if (M == 0) return VLM_OK;
  if (!dx || !dy || M < 0 || D <= 0 || (D & 3) || (lddx & 3) || (y && (ldy & 3)) || (lddy & 3)) return VLM_ERR_ARG;
  if (!y && dgamma) return VLM_ERR_ARG;  // y == NULL: the folded form, dy = bf16(row_scale * gamma * dx) and its column sums only
  if (D > 1024) return VLM_ERR_UNSUPPORTED;
  int g = row_grid(M);
  if (g > 1536) g = 1536;
  float* part = (workspace && workspace_bytes >= (size_t)g * 2 * D * sizeof(float) && (dgamma || dbias)) ? workspace : nullptr;
  dim3 grid(g), block(ROW_THREADS);
  hipStream_t s = (hipStream_t)stream;
#define SC_BWD(U)                                                                                                                    \
  do {                                                                                                                               \
    if (y) hipLaunchKernelGGL((scale_bwd_kernel<U, false>), grid, block, 0, s, dx, lddx, (const bf16_t*)y, ldy, gamma, row_scale, M,   \
                              D, (bf16_t*)dy, lddy, dgamma, dbias, part);                                                            \
    else hipLaunchKernelGGL((scale_bwd_kernel<U, true>), grid, block, 0, s, dx, lddx, (const bf16_t*)y, ldy, gamma, row_scale, M,      \
                            D, (bf16_t*)dy, lddy, dgamma, dbias, part);                                                              \
  } while (0)
  if (D <= 256) SC_BWD(1);
  else if (D <= 768) SC_BWD(3);
  else SC_BWD(4);
#undef SC_BWD
  VLM_CHECK_LAUNCH();
  if (deferred_blocks) {
    if (!part && (dgamma || dbias)) return VLM_ERR_ARG;
    *deferred_blocks = part ? g : 0;
  } else if (part) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((2 * D + 255) / 256, 32), dim3(256), 0, s, part, g, D, dgamma, dbias);
    VLM_CHECK_LAUNCH();
  }
  return VLM_OK;
}

extern "C" int vlm_colsum_bf16(const void* a, int lda, int M, int N, float* out, void* stream) {
  if (M == 0 || N == 0) return VLM_OK;
  if (!a || !out || M < 0 || N < 0 || (lda & 7) || (N & 7) || ((uintptr_t)a & 15)) return VLM_ERR_ARG;
  int gy = (M + ROW_WAVES * 32 - 1) / (ROW_WAVES * 32);
  if (gy > 128) gy = 128;
  if (gy < 1) gy = 1;
  dim3 grid((N + 511) / 512, gy), block(ROW_THREADS);
  hipLaunchKernelGGL(colsum_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_t*)a, lda, M, N, out);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
