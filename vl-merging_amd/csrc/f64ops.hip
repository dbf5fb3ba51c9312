// float64 kernels of the merge side: the Gram cache's X^T X (reference
// src/cache_gram_matrices.py:246-254: `to(float64)` then matmul) and RegMean's W* = (sum_m W_m G'_m)(sum_m G'_m)^-1
// (src/vilt/modules/vilt_module.py:407-434, 459-484: `W.double() @ G`, `torch.inverse`).  All products run on
// v_mfma_f64_16x16x4_f64 (exact fp64 FMA chains); the inverse is replaced by a blocked Cholesky factorisation of the
// SPD sum of Gram matrices and two triangular solves (the driver, vl_merging_amd/regmean.py, walks the 64-wide block
// columns and launches the kernels below).
//
// MFMA f64 16x16x4 operand layout: A[16][4]: lane l holds A[l & 15][l >> 4]; B[4][16]: lane l holds B[l >> 4][l & 15];
// C/D[16][16]: register r of lane l holds row 4*r + (l >> 4) of column l & 15.
#include "vlm_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) double f64x4;

#define F64_TILE 64   // output tile per workgroup (4 waves, each 32 x 32 = 2 x 2 MFMA blocks)
#define F64_KC 16     // reduction rows staged per step

template <typename T>
__device__ __forceinline__ double f64_load(const T* p) { return (double)*p; }
template <>
__device__ __forceinline__ double f64_load<bf16_t>(const bf16_t* p) { return (double)(float)*p; }

// One 64x64 output tile's accumulation over reduction rows [k0, k1): A-side and B-side panels are staged as
// [F64_KC][64 + 1] doubles.  a(k, i) / b(k, j) are fetched through the functors (any layout / element type).
// A_KC / B_KC: the operand is stored with the REDUCTION index contiguous (a row-major A, a transposed B): consecutive lanes then
// fetch consecutive k of one row / column (16 x 8 B = one 128-B line per 16 lanes) instead of 64 different lines per wave
// instruction -- round 5: RegMean's W G products and every GEMM of the blocked Cholesky / solves read at least one operand that way.
// LDS images of a staged panel (round 5; conflict-free by the ds_read_b64 / ds_write_b64 lane-group rules, where the round-2
// [16][64 + 1] image put the two reduction rows a 32-lane group reads on the same banks: every MFMA operand read took 2 x):
//   operand stored with the OUTPUT index contiguous:    [kk][64 + 16]  (row stride 160 words = 32 mod 64: rows kk, kk + 1 in different halves)
//   operand stored with the REDUCTION index contiguous: [c][16 + 2]    (row stride 36 words: 16 consecutive c land on 16 distinct bank quads)
#define F64_LDS_DOUBLES 1280  // max(16 x 80, 64 x 18)
template <bool KC>
__device__ __forceinline__ int f64_lds_at(int kk, int c) { return KC ? c * (F64_KC + 2) + kk : kk * (F64_TILE + 16) + c; }

template <bool A_KC = false, bool B_KC = false, typename FA, typename FB>
__device__ __forceinline__ void f64_tile_mac(f64x4 (&acc)[2][2], int k0, int k1, FA a_at, FB b_at, double* sa, double* sb) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  // The next step's panel elements are requested into registers BEFORE the current step's MFMAs (round 5: the load -> LDS ->
  // barrier -> MFMA chain of one step used to run back to back: the global-load latency was exposed once per 16 reduction rows)
  constexpr int PER = (F64_KC * F64_TILE) / 256;
  double ra[PER], rb[PER];
  auto fetch = [&](int k) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u;
      const int ka = A_KC ? (e & (F64_KC - 1)) : (e >> 6), ca = A_KC ? (e / F64_KC) : (e & 63);
      const int kb = B_KC ? (e & (F64_KC - 1)) : (e >> 6), cb = B_KC ? (e / F64_KC) : (e & 63);
      ra[u] = k + ka < k1 ? a_at(k + ka, ca) : 0.0;
      rb[u] = k + kb < k1 ? b_at(k + kb, cb) : 0.0;
    }
  };
  if (k0 < k1) fetch(k0);
  for (int k = k0; k < k1; k += F64_KC) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u;
      sa[f64_lds_at<A_KC>(A_KC ? (e & (F64_KC - 1)) : (e >> 6), A_KC ? (e / F64_KC) : (e & 63))] = ra[u];
      sb[f64_lds_at<B_KC>(B_KC ? (e & (F64_KC - 1)) : (e >> 6), B_KC ? (e / F64_KC) : (e & 63))] = rb[u];
    }
    __syncthreads();
    if (k + F64_KC < k1) fetch(k + F64_KC);
#pragma unroll
    for (int k4 = 0; k4 < F64_KC; k4 += 4) {
      const int kk = k4 + (lane >> 4), c = lane & 15;
      const double a0 = sa[f64_lds_at<A_KC>(kk, wi + c)], a1 = sa[f64_lds_at<A_KC>(kk, wi + 16 + c)];
      const double b0 = sb[f64_lds_at<B_KC>(kk, wj + c)], b1 = sb[f64_lds_at<B_KC>(kk, wj + 16 + c)];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------- Gram (SYRK)
// G[D][D] += X^T X for X [M][D] (bf16 or fp32 activations, converted exactly), upper-triangular tiles only, mirrored on
// the way out; the M rows are cut into gridDim.z slices that meet in G through fp64 atomics.
template <typename T>
__global__ __launch_bounds__(256) void gram_f64_kernel(const T* __restrict__ x, int ldx, int M, int D, double* __restrict__ g) {
  __shared__ double sa[F64_LDS_DOUBLES], sb[F64_LDS_DOUBLES];
  // blockIdx.x enumerates tile pairs (ti <= tj)
  const int nt = (D + F64_TILE - 1) / F64_TILE;
  int ti = 0, rem = blockIdx.x;
  while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
  const int tj = ti + rem;
  const int i0 = ti * F64_TILE, j0 = tj * F64_TILE;
  const int per = ((M + gridDim.z - 1) / gridDim.z + F64_KC - 1) / F64_KC * F64_KC;
  const int k0 = blockIdx.z * per, k1 = k0 + per < M ? k0 + per : M;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
  if (k0 < k1)
    f64_tile_mac(acc, k0, k1,
                 [&](int k, int c) { return i0 + c < D ? f64_load(x + (size_t)k * ldx + i0 + c) : 0.0; },
                 [&](int k, int c) { return j0 + c < D ? f64_load(x + (size_t)k * ldx + j0 + c) : 0.0; }, sa, sb);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi + 16 * a + 4 * r + (lane >> 4), j = j0 + wj + 16 * b + (lane & 15);
        if (i < D && j < D) {
          const double v = acc[a][b][r];
          if (ti != tj) {
            atomicAdd(g + (size_t)i * D + j, v);
            atomicAdd(g + (size_t)j * D + i, v);
          } else {
            atomicAdd(g + (size_t)i * D + j, v);  // diagonal tile: computed in full
          }
        }
      }
}

extern "C" int vlm_gram_f64(const void* x, int ldx, int M, int D, int x_is_f32, double* gram, void* stream) {
  if (M == 0 || D == 0) return VLM_OK;
  if (!x || !gram || M < 0 || D < 0 || ldx < D) return VLM_ERR_ARG;
  const int nt = (D + F64_TILE - 1) / F64_TILE, pairs = nt * (nt + 1) / 2;
  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  // Row slices: the kernel keeps 4 workgroups per CU resident (116 registers per lane), every workgroup of a launch runs the same
  // number of steps, so the launch takes ceil(workgroups / slots) rounds of equal length.  Round 4's "ceil(slots / pairs)" put
  // D = 768 at 1092 workgroups on 1024 slots and D = 3072 at 1176: two rounds, the second nearly empty (0.53 / 0.57 of the time
  // useful -- the whole of the round-4 "0.54 of the fp64 peak").  Take the slice count whose last round is fullest, with a small
  // charge per slice for its 64 x 64 x 2 atomics on the way out.
  const int slots = 4 * cus;
  int max_slices = (M + 4 * F64_KC - 1) / (4 * F64_KC);
  if (max_slices > 32) max_slices = 32;
  int slices = 1;
  double best = -1.0;
  for (int s = 1; s <= max_slices; ++s) {
    const long wgs = (long)pairs * s, rounds = (wgs + slots - 1) / slots;
    const double score = (double)wgs / (double)(rounds * slots) - 0.004 * s;
    if (score > best) { best = score; slices = s; }
  }
  dim3 grid(pairs, 1, slices);
  if (x_is_f32)
    hipLaunchKernelGGL((gram_f64_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(x), ldx, M, D, gram);
  else
    hipLaunchKernelGGL((gram_f64_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16_t*>(x), ldx, M, D, gram);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---------------------------------------------------------------------------------------------------- GEMM
// C[M][N] = alpha * op(A) op(B) + beta * C, fp64, row-major; op(A)[i][k] = ta ? A[k][i] : A[i][k], op(B)[k][j] = tb ?
// B[j][k] : B[k][j].  A may be fp32 (a_is_f32: the fp32 checkpoint weights of RegMean enter without a host-side cast).
template <typename TA_>
__device__ __forceinline__ void gemm_f64_body(int ta, int tb, int M, int N, int K, double alpha, const TA_* __restrict__ A,
                                              int lda, const double* __restrict__ B, int ldb, double beta,
                                              double* __restrict__ C, int ldc) {
  __shared__ double sa[F64_LDS_DOUBLES], sb[F64_LDS_DOUBLES];
  const int i0 = blockIdx.y * F64_TILE, j0 = blockIdx.x * F64_TILE;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
  auto a_at = [&](int k, int c) { return i0 + c < M ? (double)(ta ? A[(size_t)k * lda + i0 + c] : A[(size_t)(i0 + c) * lda + k]) : 0.0; };
  auto b_at = [&](int k, int c) { return j0 + c < N ? (tb ? B[(size_t)(j0 + c) * ldb + k] : B[(size_t)k * ldb + j0 + c]) : 0.0; };
  // (ta, tb are launch-uniform) the reduction index is contiguous in a row-major A and in a transposed B
  if (!ta && tb) f64_tile_mac<true, true>(acc, 0, K, a_at, b_at, sa, sb);
  else if (!ta) f64_tile_mac<true, false>(acc, 0, K, a_at, b_at, sa, sb);
  else if (tb) f64_tile_mac<false, true>(acc, 0, K, a_at, b_at, sa, sb);
  else f64_tile_mac<false, false>(acc, 0, K, a_at, b_at, sa, sb);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi + 16 * a + 4 * r + (lane >> 4), j = j0 + wj + 16 * b + (lane & 15);
        if (i < M && j < N) {
          double* c = C + (size_t)i * ldc + j;
          *c = alpha * acc[a][b][r] + (beta != 0.0 ? beta * *c : 0.0);
        }
      }
}

template <typename TA_>
__global__ __launch_bounds__(256, 4) void gemm_f64_kernel(int ta, int tb, int M, int N, int K, double alpha, const TA_* __restrict__ A,
                                                       int lda, const double* __restrict__ B, int ldb, double beta,
                                                       double* __restrict__ C, int ldc, int lower) {
  if (lower && blockIdx.x > blockIdx.y) return;  // (a symmetric update: only the tiles on and below the diagonal)
  gemm_f64_body<TA_>(ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
}

// Batched forms (round 5): the 36 + 12 independent solves of a RegMean merge have two shapes; ONE launch performs the same block
// step for every matrix of a shape (blockIdx.z / .y / .x = the matrix, pointers from a kernel-argument table) -- the chain of
// ~5 000 dependent 64-wide block launches, dealt over four streams, becomes ~420 launches that fill the chip.
#define VLM_F64_MAX_BATCH 64
struct f64_tab_t {
  double* p[VLM_F64_MAX_BATCH];
};
template <typename TA_ = double>
__global__ __launch_bounds__(256, 4) void gemm_f64_batched_kernel(int ta, int tb, int M, int N, int K, double alpha, const f64_tab_t A,
                                                               size_t offA, int lda, const f64_tab_t B, size_t offB, int ldb,
                                                               double beta, const f64_tab_t C, size_t offC, int ldc, int lower) {
  if (lower && blockIdx.x > blockIdx.y) return;
  gemm_f64_body<TA_>(ta, tb, M, N, K, alpha, reinterpret_cast<const TA_*>(A.p[blockIdx.z]) + offA, lda, B.p[blockIdx.z] + offB, ldb,
                     beta, C.p[blockIdx.z] + offC, ldc);
}

// lower: C is a symmetric update (M == N) of which only the tiles on and below the diagonal are computed -- the Cholesky trailing
// update A22 -= L21 L21^T, whose upper triangle nothing reads (round 5: half the flops and half the C traffic of a K = 64 GEMM
// that is bound by reading and writing C)
static int gemm_f64_launch(int ta, int tb, int M, int N, int K, double alpha, const void* A, int lda, int a_is_f32, const double* B,
                           int ldb, double beta, double* C, int ldc, int lower, void* stream) {
  if (M == 0 || N == 0) return VLM_OK;
  if (!A || !B || !C || M < 0 || N < 0 || K < 0 || ldc < N) return VLM_ERR_ARG;
  dim3 grid((N + F64_TILE - 1) / F64_TILE, (M + F64_TILE - 1) / F64_TILE);
  if (a_is_f32)
    hipLaunchKernelGGL((gemm_f64_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, ta, tb, M, N, K, alpha,
                       reinterpret_cast<const float*>(A), lda, B, ldb, beta, C, ldc, lower);
  else
    hipLaunchKernelGGL((gemm_f64_kernel<double>), grid, dim3(256), 0, (hipStream_t)stream, ta, tb, M, N, K, alpha,
                       reinterpret_cast<const double*>(A), lda, B, ldb, beta, C, ldc, lower);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_gemm_f64(int ta, int tb, int M, int N, int K, double alpha, const void* A, int lda, int a_is_f32,
                            const double* B, int ldb, double beta, double* C, int ldc, void* stream) {
  return gemm_f64_launch(ta, tb, M, N, K, alpha, A, lda, a_is_f32, B, ldb, beta, C, ldc, 0, stream);
}

// ---------------------------------------------------------------------------------------------------- G' = a G + (1-a) diag(G), summed
// dst (+)= alpha * src + (1 - alpha) * diag(src)   (vilt_module.py:388-392 `scaling_for_non_diag`), accumulate flag
__global__ __launch_bounds__(256) void scale_gram_kernel(const double* __restrict__ src, double* __restrict__ dst, int n, double alpha,
                                                         int accumulate) {
  const size_t total = (size_t)n * n;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int i = (int)(e / n), j = (int)(e - (size_t)i * n);
    const double g = src[e];
    // the reference forms alpha * G + (1 - alpha) * diag(G) in this order (two products, one add)
    const double v = __dadd_rn(__dmul_rn(alpha, g), i == j ? __dmul_rn(1.0 - alpha, g) : 0.0);
    dst[e] = accumulate ? __dadd_rn(dst[e], v) : v;
  }
}

extern "C" int vlm_scale_gram_f64(const double* src, double* dst, int n, double alpha, int accumulate, void* stream) {
  if (n == 0) return VLM_OK;
  if (!src || !dst || n < 0) return VLM_ERR_ARG;
  size_t blocks = ((size_t)n * n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_gram_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n, alpha, accumulate);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---------------------------------------------------------------------------------------------------- Cholesky block
// In-place lower Cholesky factor of the nb x nb (nb <= 64) diagonal block at A[j0][j0] (row-major, leading dimension
// lda): one wave, the block lives in LDS.  status[0] is set to j0 + k + 1 if pivot k is not positive (not SPD).
// Thread t holds ROW t of the block in registers (every index below is a compile-time constant: a runtime-indexed local array
// would live in scratch memory); step k: the pivot and column k of the factor reach every lane by v_readlane (lane numbers are
// compile-time constants in the unrolled loops: scalar broadcasts, no LDS round trip, no barrier -- the LDS version spent 134 us
// per block on its three barriers and dependent LDS chains per step).  Unused rows / columns of a ragged block (nb < 64) are
// padded with the identity.
__device__ __forceinline__ double f64_readlane(double v, int lane) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void potrf_block_body(double* __restrict__ A, int lda, int j0, int nb, int* __restrict__ status) {
  const int t = threadIdx.x;
  double a[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) a[c] = (t < nb && c < nb) ? A[(size_t)(j0 + t) * lda + j0 + c] : (c == t ? 1.0 : 0.0);
  int bad = 0;
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    const double d = f64_readlane(a[k], k);  // a[k][k] as thread k holds it now (wave-uniform)
    if (!(d > 0.0) && !bad) bad = j0 + k + 1;
    const double rd = sqrt(d);
    const double l = t == k ? rd : a[k] / rd;  // l[t][k] for t >= k (lanes t < k: never read)
    a[k] = l;
    // row t, columns j in (k, t]: a[t][j] -= l[t][k] * l[j][k]
#pragma unroll
    for (int jj = k + 1; jj < 64; ++jj) {
      const double ljk = f64_readlane(l, jj);
      if (jj <= t) a[jj] -= l * ljk;
    }
  }
  if (bad) {  // not SPD: report the first non-positive pivot, leave the block as it was (the caller falls back on its copy)
    if (t == 0 && status) atomicCAS(status, 0, bad);
    return;
  }
  if (t < nb) {
#pragma unroll
    for (int c = 0; c < 64; ++c)
      if (c < nb) A[(size_t)(j0 + t) * lda + j0 + c] = c <= t ? a[c] : 0.0;
  }
}

__global__ __launch_bounds__(64) void potrf_block_kernel(double* __restrict__ A, int lda, int j0, int nb, int* __restrict__ status) {
  potrf_block_body(A, lda, j0, nb, status);
}
__global__ __launch_bounds__(64) void potrf_block_batched_kernel(const f64_tab_t A, int lda, int j0, int nb, int* __restrict__ status) {
  potrf_block_body(A.p[blockIdx.x], lda, j0, nb, status + blockIdx.x);
}

extern "C" int vlm_potrf_block_f64(double* A, int lda, int j0, int nb, int* status, void* stream) {
  if (nb == 0) return VLM_OK;
  if (!A || nb < 0 || nb > 64 || j0 < 0 || lda < j0 + nb) return VLM_ERR_ARG;
  hipLaunchKernelGGL(potrf_block_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, lda, j0, nb, status);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---------------------------------------------------------------------------------------------------- triangular block solves
// X (rows x nb, in place in Bm at column c0) <- X * op(L)^-1 with L the nb x nb lower-triangular block at L[l0][l0]:
//   trans = 1:  X L^T = B  (forward over the block's columns: Cholesky panel, first RegMean solve)
//   trans = 0:  X L   = B  (backward over the block's columns: second RegMean solve)
// One WAVE per row of X: lane j holds x_j (b_j on entry); step k broadcasts x_k = b_k / L[k][k] from lane k by v_readlane (lane
// numbers are compile-time constants in the unrolled loop: scalar broadcasts) and every lane still to be solved takes its
// update b_j -= x_k L[..] with its OWN element of column / row k of the triangle from LDS (stride-65 rows: conflict-free).  64 short
// steps per row and 4 rows per workgroup instead of one thread grinding through 2 016 dependent FMAs per row: rows / 4
// workgroups fill the chip, the row is read and written as one coalesced 512-B piece.  (History: a runtime-indexed `double v[64]`
// per thread lived in scratch memory, 201 us per launch; fully unrolled in registers it was instruction-fetch bound, 62 us.)
// A ragged block (nb < 64) is padded with the identity.
#define F64_TRSM_ROWS 4  // rows per wave; a workgroup (4 waves) covers 16 (8: 34 instead of 31 us per launch)
__device__ __forceinline__ void trsm_block_body(const double* __restrict__ L, int ldl, int l0, int nb, int trans,
                                                double* __restrict__ Bm, int ldb, int rows, int c0) {
  __shared__ double s[64][65];
  __shared__ double invd[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    s[r][c] = (r < nb && c < nb) ? L[(size_t)(l0 + r) * ldl + l0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  if (tid < 64) invd[tid] = 1.0 / s[tid][tid];
  __syncthreads();
  // F64_TRSM_ROWS rows per wave (round 5; one row per wave before): the block load above is shared by 16 rows instead of 4 and the four
  // rows' substitution chains are independent -- their readlane / fma latencies hide each other.  Per row the same operations in the
  // same order as ever.
  const int row0 = (blockIdx.x * 4 + wave) * F64_TRSM_ROWS;
  if (row0 >= rows) return;  // wave-uniform
  double* x = Bm + (size_t)row0 * ldb + c0;
  double b[F64_TRSM_ROWS];
#pragma unroll
  for (int i = 0; i < F64_TRSM_ROWS; ++i) b[i] = (lane < nb && row0 + i < rows) ? x[(size_t)i * ldb + lane] : 0.0;
  const double rd = invd[lane];
  if (trans) {  // X L^T = B:  x_k = (b_k - sum_{j<k} x_j L[k][j]) / L[k][k];  after x_k: b_j -= x_k L[j][k] for j > k
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      const double rdk = f64_readlane(rd, k), ljk = s[lane][k];
#pragma unroll
      for (int i = 0; i < F64_TRSM_ROWS; ++i) {
        const double xk = f64_readlane(b[i], k) * rdk;
        b[i] = lane == k ? xk : (lane > k ? __builtin_fma(-xk, ljk, b[i]) : b[i]);
      }
    }
  } else {      // X L = B:    x_k = (b_k - sum_{j>k} x_j L[j][k]) / L[k][k];  after x_k: b_j -= x_k L[k][j] for j < k
#pragma unroll
    for (int k = 63; k >= 0; --k) {
      const double rdk = f64_readlane(rd, k), lkj = s[k][lane];
#pragma unroll
      for (int i = 0; i < F64_TRSM_ROWS; ++i) {
        const double xk = f64_readlane(b[i], k) * rdk;
        b[i] = lane == k ? xk : (lane < k ? __builtin_fma(-xk, lkj, b[i]) : b[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < F64_TRSM_ROWS; ++i)
    if (lane < nb && row0 + i < rows) x[(size_t)i * ldb + lane] = b[i];
}

__global__ __launch_bounds__(256) void trsm_block_kernel(const double* __restrict__ L, int ldl, int l0, int nb, int trans,
                                                         double* __restrict__ Bm, int ldb, int rows, int c0) {
  trsm_block_body(L, ldl, l0, nb, trans, Bm, ldb, rows, c0);
}
__global__ __launch_bounds__(256) void trsm_block_batched_kernel(const f64_tab_t L, size_t offL, int ldl, int l0, int nb, int trans,
                                                                 const f64_tab_t Bm, size_t offB, int ldb, int rows, int c0) {
  trsm_block_body(L.p[blockIdx.y] + offL, ldl, l0, nb, trans, Bm.p[blockIdx.y] + offB, ldb, rows, c0);
}

extern "C" int vlm_trsm_block_f64(const double* L, int ldl, int l0, int nb, int trans, double* Bm, int ldb, int rows, int c0,
                                  void* stream) {
  if (nb == 0 || rows == 0) return VLM_OK;
  if (!L || !Bm || nb < 0 || nb > 64 || rows < 0 || l0 < 0 || c0 < 0) return VLM_ERR_ARG;
  hipLaunchKernelGGL(trsm_block_kernel, dim3((rows + 4 * F64_TRSM_ROWS - 1) / (4 * F64_TRSM_ROWS)), dim3(256), 0, (hipStream_t)stream, L, ldl, l0, nb, trans, Bm, ldb,
                     rows, c0);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---------------------------------------------------------------------------------------------------- blocked drivers
// The whole factorisation / solve as ONE call: the block-column loops run here instead of in the Python host code (a 3072^2
// factor is 48 block columns x 3 launches, the solve 2 x 48 x 2 more: issuing them through ctypes cost more host time than the
// kernels take).  Built from the block kernels above and the tile GEMM; `count` (<= VLM_F64_MAX_BATCH) matrices of ONE shape walk
// in lock step, every step ONE launch over all of them; the single-matrix entry points are the same code at count = 1.
static int f64_tab(double* const* list, int count, f64_tab_t& t) {
  if (!list || count <= 0 || count > VLM_F64_MAX_BATCH) return VLM_ERR_ARG;
  for (int i = 0; i < count; ++i) {
    if (!list[i]) return VLM_ERR_ARG;
    t.p[i] = list[i];
  }
  return VLM_OK;
}
static int gemm_f64_batched(int ta, int tb, int M, int N, int K, double alpha, const f64_tab_t& A, size_t offA, int lda,
                            const f64_tab_t& B, size_t offB, int ldb, double beta, const f64_tab_t& C, size_t offC, int ldc, int count,
                            hipStream_t s, int lower = 0) {
  if (M <= 0 || N <= 0) return VLM_OK;
  dim3 grid((N + F64_TILE - 1) / F64_TILE, (M + F64_TILE - 1) / F64_TILE, count);
  hipLaunchKernelGGL((gemm_f64_batched_kernel<double>), grid, dim3(256), 0, s, ta, tb, M, N, K, alpha, A, offA, lda, B, offB, ldb, beta, C, offC, ldc, lower);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// `count` products of ONE shape in one launch (round 5: RegMean's 96 W_m G'_m products ran one 144..576-workgroup launch each on
// 1024 slots; per shape they now fill the chip).  Per matrix bit-identical to vlm_gemm_f64.
extern "C" int vlm_gemm_f64_batched(int ta, int tb, int M, int N, int K, double alpha, const void* const* A_list, int lda, int a_is_f32,
                                    const double* const* B_list, int ldb, double beta, double* const* C_list, int ldc, int count,
                                    void* stream) {
  if (M == 0 || N == 0 || count == 0) return VLM_OK;
  if (M < 0 || N < 0 || K < 0 || ldc < N) return VLM_ERR_ARG;
  f64_tab_t A, B, C;
  int rc = f64_tab(reinterpret_cast<double* const*>(const_cast<void* const*>(A_list)), count, A);
  if (!rc) rc = f64_tab(const_cast<double* const*>(B_list), count, B);
  if (!rc) rc = f64_tab(C_list, count, C);
  if (rc) return rc;
  if (!a_is_f32) return gemm_f64_batched(ta, tb, M, N, K, alpha, A, 0, lda, B, 0, ldb, beta, C, 0, ldc, count, (hipStream_t)stream);
  dim3 grid((N + F64_TILE - 1) / F64_TILE, (M + F64_TILE - 1) / F64_TILE, count);
  hipLaunchKernelGGL((gemm_f64_batched_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, ta, tb, M, N, K, alpha, A, (size_t)0, lda,
                     B, (size_t)0, ldb, beta, C, (size_t)0, ldc, 0);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// Blocking (round 5): the factorisation and the solves walk 64-wide block columns (the block kernels' width), but the updates
// that touch everything to the right are taken once per F64_BIG = 256 columns with K = 256 -- a K = 64 update GEMM is bound by reading
// and writing its C (the trailing matrix / the right-hand side's remaining columns: 35 GB per RegMean merge at the base width, 21 ms of
// its 49); inside a 256-column block the next 64 columns first receive the block's earlier columns' contribution (a GEMM with 64
// output columns), left-looking.  Same flops, a quarter of the C traffic, the same number of launches.
static int f64_big(void) {  // VLM_F64_BIG: experiments
  static const int v = [] { const char* e = getenv("VLM_F64_BIG"); const int x = e ? atoi(e) : 256; return x >= 64 && x % 64 == 0 ? x : 256; }();
  return v;
}
#define F64_BIG f64_big()

extern "C" int vlm_cholesky_f64_batched(double* const* A_list, int count, int n, int* status, void* stream) {
  if (n == 0 || count == 0) return VLM_OK;
  if (n < 0 || !status) return VLM_ERR_ARG;
  f64_tab_t A;
  int rc = f64_tab(A_list, count, A);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  for (int J0 = 0; J0 < n; J0 += F64_BIG) {
    const int J1 = J0 + F64_BIG < n ? J0 + F64_BIG : n;
    for (int c = J0; c < J1; c += 64) {
      const int nb = n - c < 64 ? n - c : 64, c1 = c + nb, r = n - c1;
      if (c > J0) {  // A[c:, c:c1] -= L[c:, J0:c] L[c:c1, J0:c]^T
        rc = gemm_f64_batched(0, 1, n - c, nb, c - J0, -1.0, A, (size_t)c * n + J0, n, A, (size_t)c * n + J0, n, 1.0, A, (size_t)c * n + c, n, count, s);
        if (rc) return rc;
      }
      hipLaunchKernelGGL(potrf_block_batched_kernel, dim3(count), dim3(64), 0, s, A, n, c, nb, status);
      VLM_CHECK_LAUNCH();
      if (r > 0) {  // panel: L[c1:, c:c1] = A[c1:, c:c1] L[c:c1, c:c1]^-T
        hipLaunchKernelGGL(trsm_block_batched_kernel, dim3((r + 4 * F64_TRSM_ROWS - 1) / (4 * F64_TRSM_ROWS), count), dim3(256), 0, s, A, (size_t)0, n, c, nb, 1, A, (size_t)c1 * n, n, r, c);
        VLM_CHECK_LAUNCH();
      }
    }
    const int r = n - J1;
    if (r > 0) {  // trailing update A[J1:, J1:] -= L[J1:, J0:J1] L[J1:, J0:J1]^T, tiles on and below the diagonal
      rc = gemm_f64_batched(0, 1, r, r, J1 - J0, -1.0, A, (size_t)J1 * n + J0, n, A, (size_t)J1 * n + J0, n, 1.0, A, (size_t)J1 * n + J1, n, count, s, 1);
      if (rc) return rc;
    }
  }
  return VLM_OK;
}

// rhs [rows][ld] <- rhs (L L^T)^-1 in place: Y L^T = rhs forward over the block columns, then X L = Y backward.  RIGHT-looking
// across the 256-column blocks: as soon as a block of the solution is known the columns still to be solved are updated by ONE GEMM
// over all of them (rows x (n - J1) outputs) -- a left-looking sweep would compute each block with rows / 64 workgroups and a
// reduction up to n long: 12 workgroups on 256 CUs for a [768, 3072] right-hand side.
extern "C" int vlm_solve_spd_right_f64_batched(double* const* chol_list, int n, double* const* rhs_list, int ld, int rows, int count,
                                               void* stream) {
  if (n == 0 || rows == 0 || count == 0) return VLM_OK;
  if (n < 0 || rows < 0 || ld < n) return VLM_ERR_ARG;
  f64_tab_t Lt, R;
  int rc = f64_tab(chol_list, count, Lt);
  if (rc) return rc;
  rc = f64_tab(rhs_list, count, R);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  for (int J0 = 0; J0 < n; J0 += F64_BIG) {  // Y L^T = B
    const int J1 = J0 + F64_BIG < n ? J0 + F64_BIG : n;
    for (int c = J0; c < J1; c += 64) {
      const int nb = n - c < 64 ? n - c : 64;
      if (c > J0) {  // B[:, c:c1] -= Y[:, J0:c] L[c:c1, J0:c]^T
        rc = gemm_f64_batched(0, 1, rows, nb, c - J0, -1.0, R, (size_t)J0, ld, Lt, (size_t)c * n + J0, n, 1.0, R, (size_t)c, ld, count, s);
        if (rc) return rc;
      }
      hipLaunchKernelGGL(trsm_block_batched_kernel, dim3((rows + 4 * F64_TRSM_ROWS - 1) / (4 * F64_TRSM_ROWS), count), dim3(256), 0, s, Lt, (size_t)0, n, c, nb, 1, R, (size_t)0, ld, rows, c);
      VLM_CHECK_LAUNCH();
    }
    if (J1 < n) {  // B[:, J1:] -= Y[:, J0:J1] L[J1:, J0:J1]^T
      rc = gemm_f64_batched(0, 1, rows, n - J1, J1 - J0, -1.0, R, (size_t)J0, ld, Lt, (size_t)J1 * n + J0, n, 1.0, R, (size_t)J1, ld, count, s);
      if (rc) return rc;
    }
  }
  for (int J0 = ((n - 1) / F64_BIG) * F64_BIG; J0 >= 0; J0 -= F64_BIG) {  // X L = Y
    const int J1 = J0 + F64_BIG < n ? J0 + F64_BIG : n;
    for (int c = J0 + ((J1 - J0 - 1) / 64) * 64; c >= J0; c -= 64) {
      const int nb = n - c < 64 ? n - c : 64, c1 = c + nb;
      if (c1 < J1) {  // Y[:, c:c1] -= X[:, c1:J1] L[c1:J1, c:c1]
        rc = gemm_f64_batched(0, 0, rows, nb, J1 - c1, -1.0, R, (size_t)c1, ld, Lt, (size_t)c1 * n + c, n, 1.0, R, (size_t)c, ld, count, s);
        if (rc) return rc;
      }
      hipLaunchKernelGGL(trsm_block_batched_kernel, dim3((rows + 4 * F64_TRSM_ROWS - 1) / (4 * F64_TRSM_ROWS), count), dim3(256), 0, s, Lt, (size_t)0, n, c, nb, 0, R, (size_t)0, ld, rows, c);
      VLM_CHECK_LAUNCH();
    }
    if (J0 > 0) {  // Y[:, :J0] -= X[:, J0:J1] L[J0:J1, :J0]
      rc = gemm_f64_batched(0, 0, rows, J0, J1 - J0, -1.0, R, (size_t)J0, ld, Lt, (size_t)J0 * n, n, 1.0, R, (size_t)0, ld, count, s);
      if (rc) return rc;
    }
  }
  return VLM_OK;
}

// The single-matrix calls are the batched ones at count = 1 (one algorithm, one set of kernels: bit-identical by construction).
extern "C" int vlm_cholesky_f64(double* A, int n, int* status, void* stream) {
  if (n == 0) return VLM_OK;
  if (!A || n < 0 || !status) return VLM_ERR_ARG;
  double* list[1] = {A};
  return vlm_cholesky_f64_batched(list, 1, n, status, stream);
}
extern "C" int vlm_solve_spd_right_f64(const double* chol, int n, double* rhs, int ld, int rows, void* stream) {
  if (n == 0 || rows == 0) return VLM_OK;
  if (!chol || !rhs || n < 0 || rows < 0 || ld < n) return VLM_ERR_ARG;
  double* cl[1] = {const_cast<double*>(chol)};
  double* rl[1] = {rhs};
  return vlm_solve_spd_right_f64_batched(cl, n, rl, ld, rows, 1, stream);
}
