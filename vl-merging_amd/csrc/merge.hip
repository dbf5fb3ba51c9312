// Checkpoint-merge kernel (K12/K13/mean): one launch over every output tensor of an all_moe -> ufo merge.
// Reference: src/vilt/modules/vilt_module.py:533-638 (merge_weights), :640-746 (sum_task_vectors),
// :436-457 (regmean's bias / LayerNorm averages).
//
// Roofline: HBM-bound, 4 B written + 4*n_src B read per element, no reuse.  Work is cut into 16 KiB chunks
// (4096 floats) described by a device-resident table so that ONE grid covers all 168 tensors; each thread
// moves 16 B per access (global_load/store_dwordx4), all n_src loads of a 4-vector batch are issued before
// the first use.  Compiled with -ffp-contract=off and written with __fmul_rn/__fadd_rn so that no FMA is
// formed: the reference's CPU path rounds after the multiply and after every add (SURVEY.md 7 "hard parts").
#include "vlm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define MERGE_CHUNK 4096u  // floats per chunk: 256 threads x 4 float4
#define MERGE_THREADS 256

struct merge_chunk_t {
  uint32_t job;
  uint32_t start4;  // chunk start / 4 (float4 units)
};

struct merge_header_t {
  uint32_t n_jobs;
  uint32_t n_chunks;
  uint32_t jobs_off;    // byte offsets inside the workspace
  uint32_t chunks_off;
};

template <int MODE>
__device__ __forceinline__ float merge_scalar(const vlm_merge_job_t& j, float base, const float* w) {
  float acc;
  if (MODE == VLM_MERGE_LERP) {
    acc = 0.0f;
    for (int m = 0; m < j.n_src; ++m) acc = __fadd_rn(acc, __fmul_rn(j.ratio[m], w[m]));
  } else if (MODE == VLM_MERGE_TASKVEC) {
    acc = base;
    for (int m = 0; m < j.n_src; ++m) acc = __fadd_rn(acc, __fmul_rn(j.ratio[m], __fsub_rn(w[m], acc)));
  } else {
    acc = 0.0f;
    for (int m = 0; m < j.n_src; ++m) acc = __fadd_rn(acc, w[m]);
    acc = __fdiv_rn(acc, (float)j.n_src);
  }
  return acc;
}

// cache policy of the streams (measured per box by tools/bench_merge_variants.py, see vlm_merge_run)
template <bool NT>
__device__ __forceinline__ f32x4 merge_ld(const f32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void merge_st(f32x4 v, f32x4* p) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int MODE, int NSRC, bool NTL, bool NTS>
__device__ __forceinline__ void merge_chunk_vec(const vlm_merge_job_t& j, uint64_t start4, uint64_t n4) {
  // 4 float4 per thread per chunk, strided by the block so every wave instruction is 1 KiB contiguous
  f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(j.dst);
  const f32x4* __restrict__ base = reinterpret_cast<const f32x4*>(j.base);
  const f32x4* __restrict__ s[NSRC];
  float r[NSRC];
#pragma unroll
  for (int m = 0; m < NSRC; ++m) {
    s[m] = reinterpret_cast<const f32x4*>(j.src[m]);
    r[m] = j.ratio[m];
  }
  f32x4 v[4][NSRC];
  f32x4 b[4];
  uint64_t idx[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    idx[u] = start4 + threadIdx.x + u * MERGE_THREADS;
    if (idx[u] < n4) {
#pragma unroll
      for (int m = 0; m < NSRC; ++m) v[u][m] = merge_ld<NTL>(&s[m][idx[u]]);
      if (MODE == VLM_MERGE_TASKVEC) b[u] = merge_ld<NTL>(&base[idx[u]]);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (idx[u] < n4) {
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float acc;
        if (MODE == VLM_MERGE_LERP) {
          acc = 0.0f;
#pragma unroll
          for (int m = 0; m < NSRC; ++m)
            acc = __fadd_rn(acc, __fmul_rn(r[m], v[u][m][c]));
        } else if (MODE == VLM_MERGE_TASKVEC) {
          acc = b[u][c];
#pragma unroll
          for (int m = 0; m < NSRC; ++m)
            acc = __fadd_rn(acc, __fmul_rn(r[m], __fsub_rn(v[u][m][c], acc)));
        } else {
          acc = 0.0f;
#pragma unroll
          for (int m = 0; m < NSRC; ++m) acc = __fadd_rn(acc, v[u][m][c]);
          acc = __fdiv_rn(acc, (float)NSRC);
        }
        o[c] = acc;
      }
      merge_st<NTS>(o, &dst[idx[u]]);
    }
  }
}

template <int MODE, bool NTL, bool NTS>
__device__ __forceinline__ void merge_chunk_mode(const vlm_merge_job_t& j, uint64_t start4, uint64_t n4) {
  switch (j.n_src) {
    case 1: merge_chunk_vec<MODE, 1, NTL, NTS>(j, start4, n4); break;
    case 2: merge_chunk_vec<MODE, 2, NTL, NTS>(j, start4, n4); break;
    case 3: merge_chunk_vec<MODE, 3, NTL, NTS>(j, start4, n4); break;
    default: merge_chunk_vec<MODE, 4, NTL, NTS>(j, start4, n4); break;
  }
}

template <bool NTL, bool NTS>
__global__ __launch_bounds__(MERGE_THREADS) void vlm_merge_kernel(const unsigned char* __restrict__ ws) {
  const merge_header_t* hdr = reinterpret_cast<const merge_header_t*>(ws);
  const vlm_merge_job_t* jobs = reinterpret_cast<const vlm_merge_job_t*>(ws + hdr->jobs_off);
  const merge_chunk_t* chunks = reinterpret_cast<const merge_chunk_t*>(ws + hdr->chunks_off);
  const uint32_t n_chunks = hdr->n_chunks;
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const merge_chunk_t ck = chunks[c];
    const vlm_merge_job_t& j = jobs[ck.job];  // block-uniform => scalar loads
    const uint64_t n4 = j.n_elem >> 2;
    const uint64_t start4 = ck.start4;
    if (j.mode == VLM_MERGE_LERP) merge_chunk_mode<VLM_MERGE_LERP, NTL, NTS>(j, start4, n4);
    else if (j.mode == VLM_MERGE_TASKVEC) merge_chunk_mode<VLM_MERGE_TASKVEC, NTL, NTS>(j, start4, n4);
    else merge_chunk_mode<VLM_MERGE_MEAN, NTL, NTS>(j, start4, n4);
    // ragged tail (n_elem % 4) belongs to the chunk that holds the last float4 (or chunk 0 of a tiny job)
    const uint64_t tail0 = n4 << 2;
    const bool last = (start4 + (MERGE_CHUNK / 4) >= n4);
    if (last && threadIdx.x < (j.n_elem - tail0)) {
      const uint64_t i = tail0 + threadIdx.x;
      float w[VLM_MERGE_MAX_SRC];
      for (int m = 0; m < j.n_src; ++m) w[m] = reinterpret_cast<const float*>(j.src[m])[i];
      float b = (j.mode == VLM_MERGE_TASKVEC) ? reinterpret_cast<const float*>(j.base)[i] : 0.0f;
      float o;
      if (j.mode == VLM_MERGE_LERP) o = merge_scalar<VLM_MERGE_LERP>(j, b, w);
      else if (j.mode == VLM_MERGE_TASKVEC) o = merge_scalar<VLM_MERGE_TASKVEC>(j, b, w);
      else o = merge_scalar<VLM_MERGE_MEAN>(j, b, w);
      reinterpret_cast<float*>(j.dst)[i] = o;
    }
  }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static uint64_t merge_chunks_of(uint64_t n_elem) {
  uint64_t n4 = n_elem >> 2;
  uint64_t c = (n4 + MERGE_CHUNK / 4 - 1) / (MERGE_CHUNK / 4);
  return c == 0 ? 1 : c;  // a job shorter than 4 floats still needs its tail chunk
}

extern "C" size_t vlm_merge_plan_bytes(int n_jobs, uint64_t total_elems) {
  if (n_jobs < 0) return 0;
  // upper bound: every job may add one partial chunk
  uint64_t chunks = total_elems / MERGE_CHUNK + 2ull * (uint64_t)n_jobs + 1;
  return align_up(sizeof(merge_header_t), 256) + align_up((size_t)n_jobs * sizeof(vlm_merge_job_t), 256) +
         align_up((size_t)chunks * sizeof(merge_chunk_t), 256);
}

extern "C" int vlm_merge_plan_upload(const vlm_merge_job_t* jobs, int n_jobs, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  if (!jobs || n_jobs <= 0 || !workspace) return VLM_ERR_ARG;
  uint64_t n_chunks = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const vlm_merge_job_t& j = jobs[i];
    if (j.n_src < 1 || j.n_src > VLM_MERGE_MAX_SRC || j.mode < 0 || j.mode > VLM_MERGE_MEAN || !j.dst) return VLM_ERR_ARG;
    if (j.mode == VLM_MERGE_TASKVEC && !j.base) return VLM_ERR_ARG;
    if (((uintptr_t)j.dst & 15) || (j.base && ((uintptr_t)j.base & 15))) return VLM_ERR_ARG;
    for (int m = 0; m < j.n_src; ++m)
      if (!j.src[m] || ((uintptr_t)j.src[m] & 15)) return VLM_ERR_ARG;
    if ((j.n_elem >> 2) >= (1ull << 32)) return VLM_ERR_UNSUPPORTED;
    n_chunks += merge_chunks_of(j.n_elem);
  }
  if (n_chunks >= (1ull << 32)) return VLM_ERR_UNSUPPORTED;
  merge_header_t hdr;
  hdr.n_jobs = (uint32_t)n_jobs;
  hdr.n_chunks = (uint32_t)n_chunks;
  hdr.jobs_off = (uint32_t)align_up(sizeof(merge_header_t), 256);
  hdr.chunks_off = (uint32_t)(hdr.jobs_off + align_up((size_t)n_jobs * sizeof(vlm_merge_job_t), 256));
  size_t total = hdr.chunks_off + align_up((size_t)n_chunks * sizeof(merge_chunk_t), 256);
  if (total > workspace_bytes) return VLM_ERR_WORKSPACE;
  std::vector<unsigned char> img(total, 0);
  memcpy(img.data(), &hdr, sizeof(hdr));
  memcpy(img.data() + hdr.jobs_off, jobs, (size_t)n_jobs * sizeof(vlm_merge_job_t));
  merge_chunk_t* ck = reinterpret_cast<merge_chunk_t*>(img.data() + hdr.chunks_off);
  // interleave chunks of different jobs round-robin-free: plain job order keeps each block's stream contiguous
  uint64_t c = 0;
  for (int i = 0; i < n_jobs; ++i) {
    uint64_t nc = merge_chunks_of(jobs[i].n_elem);
    for (uint64_t k = 0; k < nc; ++k) {
      ck[c].job = (uint32_t)i;
      ck[c].start4 = (uint32_t)(k * (MERGE_CHUNK / 4));
      ++c;
    }
  }
  // pageable temporary source: the copy is waited for before `img` dies (header: this call synchronises the stream)
  if (hipMemcpyAsync(workspace, img.data(), total, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return VLM_ERR_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return VLM_ERR_LAUNCH;
  return VLM_OK;
}

extern "C" int vlm_merge_run(const void* workspace, void* stream) {
  if (!workspace) return VLM_ERR_ARG;
  // grid: G blocks per CU keep >= 2 KiB x n_src of loads in flight per SIMD; the chunk loop strides the grid.
  // VLM_MERGE_VARIANT="<blocks per CU>,<nt loads 0/1>,<nt stores 0/1>" overrides the default (measurement switch).
  struct variant_t { int blocks_per_cu, ntl, nts; };
  static const variant_t var = [] {  // parsed once, thread-safe (C++11 static initialisation), immutable afterwards
    variant_t v = {96, 1, 1};
    const char* e = getenv("VLM_MERGE_VARIANT");
    if (e) sscanf(e, "%d,%d,%d", &v.blocks_per_cu, &v.ntl, &v.nts);
    if (v.blocks_per_cu < 1 || v.blocks_per_cu > 256) v.blocks_per_cu = 96;
    return v;
  }();
  const int blocks_per_cu = var.blocks_per_cu, ntl = var.ntl, nts = var.nts;
  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  // 96 blocks per CU (8 resident at a time): 5.74 TB/s against 5.53 at 24 and 5.50 at 8 on one box, 5.89 against 5.50 on another:
  // late-finishing blocks no longer hold a whole stride of chunks back (tools/bench_merge_variants.py; nt loads + nt stores win)
  dim3 grid(cus * blocks_per_cu), block(MERGE_THREADS);
  const unsigned char* ws = (const unsigned char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  if (ntl && nts) hipLaunchKernelGGL((vlm_merge_kernel<true, true>), grid, block, 0, s, ws);
  else if (ntl) hipLaunchKernelGGL((vlm_merge_kernel<true, false>), grid, block, 0, s, ws);
  else if (nts) hipLaunchKernelGGL((vlm_merge_kernel<false, true>), grid, block, 0, s, ws);
  else hipLaunchKernelGGL((vlm_merge_kernel<false, false>), grid, block, 0, s, ws);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
