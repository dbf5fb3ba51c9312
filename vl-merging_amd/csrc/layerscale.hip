// LayerScale folded into the branch's output projection (round 5).
//
// Reference: Block.forward, vision_transformer.py:489-491, :586, :603:   x = x + drop_path(gamma_1 * attn(norm1(x)))  ;
// x = x + drop_path(gamma_2 * mlp(norm2(x))), the branch ending in a Linear (attn.proj / mlp.fc2: y = a W^T + b).
// Since gamma (.) (a W^T + b) = a (diag(gamma) W)^T + gamma (.) b, the GEMMs run on the FOLDED operands
//     W' = diag(gamma) W  (bf16 shadow and its transpose),  b' = gamma (.) b  (fp32),
// the residual epilogue has no column scale and saves no copy of the branch output, and the backward pass feeds
// g = bf16(row_scale * dx) straight to dgrad (against W'^T) and wgrad.  What autograd needs of the three parameters follows from
// the RAW sums the GEMMs produce,  G = g^T a  (= dL/dW')  and  s = colsum(g)  (= dL/db'):
//     dW[n, :] = gamma[n] G[n, :] ,   db[n] = gamma[n] s[n] ,   dgamma[n] = sum_k W[n, k] G[n, k] + b[n] s[n]
// (the last one is sum_m row_scale dx[m, n] y[m, n] with y = a W^T + b written out) -- an O(N K) pass per weight and backward
// instead of an O(M N) pass per block evaluation over a saved copy of y.
//   vlm_layerscale_fold:   masters -> folded shadows (after every optimizer step / reload)
//   vlm_layerscale_finish: raw accumulators -> gradients (after the last backward that touched a block; the raw buffers are
//                          zeroed on the way, so the call composes with gradient accumulation)
#include "vlm_common.h"

struct ls_jobs_t {
  vlm_layerscale_job_t j[VLM_MAX_LAYERSCALE_JOBS];
};

// one wave per weight row: shadow[n, :] = bf16(gamma[n] * W[n, :]) ; bias_out[n] = gamma[n] * bias[n]
__global__ __launch_bounds__(256) void layerscale_fold_kernel(const ls_jobs_t jobs) {
  const vlm_layerscale_job_t& jb = jobs.j[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int n = blockIdx.x * 4 + wave; n < jb.N; n += gridDim.x * 4) {
    const float g = jb.gamma ? jb.gamma[n] : 1.0f;
    const float* w = jb.weight + (size_t)n * jb.K;
    bf16_t* o = reinterpret_cast<bf16_t*>(jb.shadow) + (size_t)n * jb.K;
    for (int k = lane * 4; k < jb.K; k += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + k);
      bf16x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = (bf16_t)(g * v[r]);
      *reinterpret_cast<bf16x4*>(o + k) = h;
    }
    if (lane == 0 && jb.bias && jb.bias_out) jb.bias_out[n] = g * jb.bias[n];
  }
}

// one wave per weight row: dot = sum_k W[n,k] G[n,k];  dW[n,:] += gamma[n] G[n,:];  G[n,:] = 0;
// dgamma[n] += dot + b[n] s[n];  db[n] += gamma[n] s[n];  s[n] = 0.
// The jobs of a launch that share one gamma vector (the modality experts of an all_moe block's proj, or of its fc2) form a GROUP:
// one wave walks the group's jobs in their fixed order and adds the row's total ONCE -- no float atomics, so the gamma
// gradients are bit-reproducible from run to run like the rest of the backward pass (round 5 added them with atomicAdd, whose
// arrival order is not fixed).
struct ls_groups_t {
  unsigned char start[VLM_MAX_LAYERSCALE_JOBS], count[VLM_MAX_LAYERSCALE_JOBS];  // per group: first job, number of jobs
};
__global__ __launch_bounds__(256) void layerscale_finish_kernel(const ls_jobs_t jobs, const ls_groups_t grp) {
  const int j0 = grp.start[blockIdx.y], nj = grp.count[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int maxN = 0;
  for (int q = 0; q < nj; ++q) maxN = jobs.j[j0 + q].N > maxN ? jobs.j[j0 + q].N : maxN;
  for (int n = blockIdx.x * 4 + wave; n < maxN; n += gridDim.x * 4) {
    float dg = 0.f;
    for (int q = 0; q < nj; ++q) {
      const vlm_layerscale_job_t& jb = jobs.j[j0 + q];
      if (n >= jb.N) continue;
      const float g = jb.gamma ? jb.gamma[n] : 1.0f;
      const float* w = jb.weight + (size_t)n * jb.K;
      float* raw = jb.raw_w + (size_t)n * jb.K;
      float* dw = jb.dweight + (size_t)n * jb.K;
      float dot = 0.f;
      for (int k = lane * 4; k < jb.K; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(raw + k);
        f32x4 d = *reinterpret_cast<const f32x4*>(dw + k);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dot += wv[r] * gv[r];
          d[r] += g * gv[r];
        }
        *reinterpret_cast<f32x4*>(dw + k) = d;
        *reinterpret_cast<f32x4*>(raw + k) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      dot = wave_sum(dot);
      if (lane == 0) {
        float s = 0.f;
        if (jb.raw_b) {
          s = jb.raw_b[n];
          jb.raw_b[n] = 0.f;
          if (jb.dbias) jb.dbias[n] += g * s;
        }
        dg += dot + (jb.bias ? jb.bias[n] * s : 0.f);
      }
    }
    const vlm_layerscale_job_t& j00 = jobs.j[j0];
    if (lane == 0 && j00.dgamma) j00.dgamma[n] += dg;  // this wave is the only writer of the group's dgamma[n] in the launch
  }
}

static int ls_check(const vlm_layerscale_job_t* jobs, int n_jobs, bool finish) {
  if (!jobs || n_jobs < 0 || n_jobs > VLM_MAX_LAYERSCALE_JOBS) return VLM_ERR_ARG;
  for (int i = 0; i < n_jobs; ++i) {
    const vlm_layerscale_job_t& j = jobs[i];
    if (!j.weight || j.N <= 0 || j.K <= 0 || (j.K & 3) || ((uintptr_t)j.weight & 15)) return VLM_ERR_ARG;
    if (finish) {
      if (!j.raw_w || !j.dweight || ((uintptr_t)j.raw_w & 15) || ((uintptr_t)j.dweight & 15)) return VLM_ERR_ARG;
    } else {
      if (!j.shadow || ((uintptr_t)j.shadow & 7)) return VLM_ERR_ARG;
    }
  }
  return VLM_OK;
}

extern "C" int vlm_layerscale_fold(const vlm_layerscale_job_t* jobs, int n_jobs, void* stream) {
  if (n_jobs == 0) return VLM_OK;
  const int rc = ls_check(jobs, n_jobs, false);
  if (rc != VLM_OK) return rc;
  ls_jobs_t a;
  int maxN = 0;
  for (int i = 0; i < n_jobs; ++i) { a.j[i] = jobs[i]; if (jobs[i].N > maxN) maxN = jobs[i].N; }
  hipLaunchKernelGGL(layerscale_fold_kernel, dim3((maxN + 3) / 4, n_jobs), dim3(256), 0, (hipStream_t)stream, a);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_layerscale_finish(const vlm_layerscale_job_t* jobs, int n_jobs, void* stream) {
  if (n_jobs == 0) return VLM_OK;
  const int rc = ls_check(jobs, n_jobs, true);
  if (rc != VLM_OK) return rc;
  // jobs that accumulate into the same gamma gradient, in the caller's order, become one group (one writer per row)
  ls_jobs_t a;
  ls_groups_t g;
  bool taken[VLM_MAX_LAYERSCALE_JOBS] = {};
  int n_groups = 0, pos = 0, maxN = 0;
  for (int i = 0; i < n_jobs; ++i) {
    if (taken[i]) continue;
    g.start[n_groups] = (unsigned char)pos;
    for (int k = i; k < n_jobs; ++k)
      if (!taken[k] && (k == i || (jobs[i].dgamma && jobs[k].dgamma == jobs[i].dgamma))) {
        taken[k] = true;
        a.j[pos++] = jobs[k];
        if (jobs[k].N > maxN) maxN = jobs[k].N;
      }
    g.count[n_groups] = (unsigned char)(pos - g.start[n_groups]);
    ++n_groups;
  }
  hipLaunchKernelGGL(layerscale_finish_kernel, dim3((maxN + 3) / 4, n_groups), dim3(256), 0, (hipStream_t)stream, a, g);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
