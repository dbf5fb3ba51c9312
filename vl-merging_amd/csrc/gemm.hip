// bf16 MFMA GEMM with fused epilogues for the VLMo block (K1/K6/K8/K9/K10 of SURVEY.md 2.3).
//
//   C[M,N] = epilogue( op(A)[M,K] . op(B)[K,N] )        fp32 accumulate on v_mfma_f32_16x16x32_bf16
//
// Replaces F.linear / nn.Linear at reference vision_transformer.py:335 (qkv), :360 (proj), :291-295 (fc1/fc2),
// heads.py (pooler / itm / ifm / mlm decoder) and their autograd backward GEMMs.
//   ta = 0 : A stored [M][K] (K contiguous)           ta = 1 : A stored [K][M] (M contiguous)
//   tb = 0 : B stored [N][K] (nn.Linear weight, K contiguous)   tb = 1 : B stored [K][N]
//   forward   y  = x W^T        : ta=0 tb=0
//   dgrad     dx = dy W         : ta=0 tb=1   (reduction over W's rows)
//   wgrad     dW = dy^T x       : ta=1 tb=1   (reduction over tokens; fp32 out, accumulate)
//
// Tiling (gfx950): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.
// Operands are staged global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds; the descriptor's bounds check zero-fills
// ragged M/N/K tails), double-buffered in LDS (2 x 32 KiB), next tile's DMA in flight during the current tile's MFMAs.  K-contiguous operands live in LDS as [128][64] with a 16-B-chunk XOR swizzle
// (chunk ^= row & 7) and are read with ds_read_b128; K-strided operands live as [64][128] with a 32-B-chunk XOR
// swizzle and are read transposed with ds_read_b64_tr_b16, so no operand is ever transposed in memory.
// The MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so that each lane ends
// up with 4 CONSECUTIVE output columns of one row: the epilogue then moves 16-B (fp32) / 8-B (bf16) vectors.
#include "vlm_common.h"
#include "vlm_diag.h"
#include <atomic>
#include <stdlib.h>

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_THREADS 256
#define GEMM_TILE_BYTES (GEMM_BM * GEMM_BK * 2)  // 16 KiB per operand per stage

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output's 2^-9): 1 rcp + 1 exp + 5 fma
// instead of libm erff's ~40 instructions -- the epilogue of a K=768 GEMM is as long as its main loop otherwise.
// The forms below are written in x itself (not z = x / sqrt(2)) and carry the cdf's factor 1/2 inside the polynomial: a lone wave
// pays every vector instruction of the epilogue in full, and this is 14 of them + 2 transcendentals per element
// where the textbook arrangement took 16 + 2.
//   half_erf(x) = sign(x) * (1/2 - (a1 t + ... + a5 t^5)/2 * exp(-x^2/2)),  t = 1 / (1 + p |x| / sqrt(2));  cdf = 1/2 + half_erf
// ez2 returns exp(-x*x/2), the Gaussian factor of gelu'.
__device__ __forceinline__ float gelu_cdf(float x, float& ez2) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
  ez2 = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.4426950408889634f));
  float poly = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  poly = fmaf(t, poly, 0.5f * 1.421413741f);
  poly = fmaf(t, poly, 0.5f * -0.284496736f);
  poly = fmaf(t, poly, 0.5f * 0.254829592f);
  const float e = fmaf(-(poly * t), ez2, 0.5f);
  return 0.5f + copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
  float ez2;
  return x * gelu_cdf(x, ez2);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float ez2;
  const float cdf = gelu_cdf(x, ez2);
  return fmaf(x * 0.39894228040143267794f, ez2, cdf);
}
// gelu(x) and gelu'(x) from ONE erf / exp evaluation (VLM_ACT_GELU_DERIV: the forward epilogue saves the derivative, so the
// backward epilogue is a multiplication instead of 1 rcp + 1 exp + ~14 VALU operations per element)
__device__ __forceinline__ float gelu_erf_both(float x, float& deriv) {
  float ez2;
  const float cdf = gelu_cdf(x, ez2);
  deriv = fmaf(x * 0.39894228040143267794f, ez2, cdf);
  return x * cdf;
}
// factor of the two backward activations: the saved derivative itself, or gelu' of the saved pre-activation
__device__ __forceinline__ float act_bwd_factor(int act, float saved) {
  return act == VLM_ACT_MUL_AUX ? saved : gelu_erf_grad(saved);
}

// ---- global -> register -> LDS staging (kept for K-strided operands, where it measured faster than LDS-DMA) ------
template <bool KSTRIDED>
__device__ __forceinline__ void stage_load(u32x4 (&r)[4], __amdgpu_buffer_rsrc_t rsrc, uint32_t row0, uint32_t k0,
                                           uint32_t ld, int tid) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t p = tid + 256 * u;
    uint32_t off;
    if (!KSTRIDED) {
      const uint32_t row = p >> 3, chunk = p & 7;
      off = ((row0 + row) * ld + k0 + chunk * 8) * 2;
    } else {
      const uint32_t krow = p >> 4, c16 = p & 15;
      off = ((k0 + krow) * ld + row0 + c16 * 8) * 2;
    }
    r[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
  }
}

template <bool KSTRIDED>
__device__ __forceinline__ void stage_store(const u32x4 (&r)[4], unsigned char* lds, int tid) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t p = tid + 256 * u;
    uint32_t byte;
    if (!KSTRIDED) {
      const uint32_t row = p >> 3, chunk = p & 7;
      byte = row * 128 + ((chunk ^ (row & 7)) << 4);
    } else {
      const uint32_t krow = p >> 4, c16 = p & 15;
      const uint32_t c32 = (c16 >> 1) ^ (krow & 3) ^ (((krow >> 3) & 1) << 2);
      byte = krow * 256 + c32 * 32 + (c16 & 1) * 16;
    }
    *reinterpret_cast<u32x4*>(lds + byte) = r[u];
  }
}

// ---- global -> LDS staging by LDS-DMA (buffer_load_dwordx4 ... lds) ------------------------------------------------
// ds_write_b128 moves only ~79 B/clk/CU (MI355X_MICROARCH.md, LDS table): register staging made the LDS pipe, not the
// MFMAs, the bound (measured 311 TFLOP/s).  One wave instruction writes 1 KiB of LDS linearly (wave-uniform base +
// lane*16 B) from a PER-LANE source address, so the XOR swizzles live on the source side (guide rule 21):
//   K-contiguous tile [128 rows][64 k]: instruction j covers rows 8j..8j+7; lane -> row 8j + (lane>>3), LDS slot
//     (lane&7) holds global chunk (lane&7) ^ (row&7)
//   K-strided tile [64 k][128 x]: instruction j covers k-rows 4j..4j+3; lane -> krow 4j + (lane>>4), LDS 16-B slot
//     (lane&15) holds global 32-B chunk ((lane&15)>>1) ^ (krow&3) ^ (((krow>>3)&1)<<2), same half
// The buffer descriptor's bounds check zero-fills ragged M/N/K tails in flight.
typedef __attribute__((address_space(3))) void lds_void;
template <bool KSTRIDED>
__device__ __forceinline__ void stage_dma(__amdgpu_buffer_rsrc_t rsrc, unsigned char* tile, uint32_t row0, uint32_t k0,
                                          uint32_t ld, int wave, int lane) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = wave + 4 * u;  // wave-uniform
    uint32_t off;
    if (!KSTRIDED) {
      const uint32_t row = j * 8 + (lane >> 3), chunk = (lane & 7) ^ (row & 7);
      off = ((row0 + row) * ld + k0 + chunk * 8) * 2;
    } else {
      const uint32_t krow = j * 4 + (lane >> 4), s16 = lane & 15;
      const uint32_t c32 = (s16 >> 1) ^ (krow & 3) ^ (((krow >> 3) & 1) << 2);
      off = ((k0 + krow) * ld + row0 + (c32 * 2 + (s16 & 1)) * 8) * 2;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(tile + j * 1024), 16, off, 0, 0, 0);
  }
}

// ---- LDS -> MFMA fragment: 16 rows (x) x 32 k, lane l holds row (l&15), k = 8*(l>>4) + j ---------------------
template <bool KSTRIDED>
__device__ __forceinline__ bf16x8 frag_load(const unsigned char* lds, int xblk /*16-row block in tile*/, int ksub,
                                            int lane) {
  if (!KSTRIDED) {
    const uint32_t row = xblk * 16 + (lane & 15);
    const uint32_t chunk = ksub * 4 + (lane >> 4);
    const uint32_t byte = row * 128 + ((chunk ^ (row & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(lds + byte);
  } else {
    const uint32_t g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const uint32_t row = ksub * 32 + 8 * g + q;
    const uint32_t c32 = ((uint32_t)xblk ^ q ^ ((g & 1) << 2));
    const uint32_t byte = row * 256 + c32 * 32 + 8 * p;
    // rows +0..3 -> elements 0..3, rows +4..7 -> elements 4..7 (row+4 keeps row&3 and (row>>3)&1)
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + byte));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + byte + 4 * 256));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

// One row range of a grouped call (vlm_gemm_bf16_grouped): its own weight, bias and column-sum targets; M tiles are counted
// per group (tile0 = index of its first 256-row tile in the launch's tile grid).
struct gemm_group_t {
  const void* B;
  const float* bias;
  float* col_sum;
  float* col_sum_ws;
  int ldb, row0, row_end, tile0;
};

// wgrad form (vlm_gemm_wgrad_grouped): a group is a range of TOKEN rows (the reduction) with its own K slices and its own
// block of the slice workspace; item0 = first (slice, tile) item of the group in the launch's grid.
struct gemmT_group_t {
  int row0, row_end, item0, kps, slice0;
};

struct gemm_params_t {
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  int lda, ldb, ldc;
  vlm_epilogue_t epi;
  int tiles_m, tiles_n;
  int group_m;  // raster group height in tiles (1 = row-major)
  int splits, ksteps_per_split;  // split-K (wgrad): block -> (tile, K slice), fp32 atomic accumulation
  int n_groups;                  // 256x256 kernel, GROUPED instantiations only
  gemm_group_t grp[VLM_GEMM_MAX_GROUPS];
  gemmT_group_t grpT[VLM_GEMM_MAX_GROUPS];  // wgrad kernel, GROUPED instantiation only
  VLM_DIAG_GEMM_FIELD  // empty in the product build (vlm_diag.h)
};

// v_permlane16_swap_b32: x' = [x.row0, y.row0, x.row2, y.row2], y' = [x.row1, y.row1, x.row3, y.row3] (rows = 16 lanes;
// probed on gfx950, tools/scratch/permlane.hip).  Inline asm: hipcc 7.2 folds four __builtin_amdgcn_permlane16_swap
// calls on the elements of a vector into ONE swap + broadcast (wrong results); s_nop covers the VALU-write -> swap-read
// distance the assembler cannot see.
__device__ __forceinline__ void lane16_swap(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}

// ---- fused epilogue of one wave's 64x64 sub-tile (swapped layout: lane holds row m = mw0 + 16i + (lane&15) and the
// 4 consecutive columns n = nw0 + 16j + 4*(lane>>4) + r) -----------------------------------------------------------
// Returns the number of vector-memory instructions issued AFTER the last one whose result was waited for (the stores
// of the interior fast path), or -1 when unknown (edge tiles, column sums): the persistent kernel's counted waits.
template <bool OUT_F32, bool LEAN = false>  // LEAN: no column sums, bias/gamma re-read per row block (fewer live registers)
__device__ __forceinline__ int gemm_epilogue(const gemm_params_t& p, const f32x4 (&acc)[4][4], int mw0, int nw0, int lane,
                                             float* red = nullptr, int red_off = 0) {
  const vlm_epilogue_t& e = p.epi;
  // 16-B / 8-B epilogue vectors need every leading dimension to keep 4-element alignment
  const bool vec_ok = ((p.ldc & 3) == 0) && (!e.aux || (e.ld_aux & 3) == 0) && (!e.residual || (e.ld_res & 3) == 0);
  const bool vec8_ok = vec_ok && (OUT_F32 || (p.ldc & 7) == 0) && (!e.aux || ((e.ld_aux & 7) == 0 && ((uintptr_t)e.aux & 15) == 0)) &&
                       (!e.residual || ((uintptr_t)e.residual & 15) == 0) && (!e.bias || ((uintptr_t)e.bias & 15) == 0) &&
                       (!e.col_scale || ((uintptr_t)e.col_scale & 15) == 0);
  if (vec8_ok && mw0 + 64 <= p.M && nw0 + 64 <= p.N) {
    // Interior sub-tile, two measures against the store-issue-bound tail (16 dwordx2 per lane ~ 9.4k cycles/tile):
    // (1) v_permlane16_swap_b32 between the accumulators of column blocks 2jp and 2jp+1 leaves every lane with 8
    //     consecutive columns (block 2jp+(g&1), columns 8(g>>1)..+7, g = lane>>4), so bf16 results leave as 8
    //     dwordx4 stores and every aux / residual / bias read is 16 B wide;
    // (2) the residual may alias C (in-place residual stream): all epilogue inputs of a 16-row block are fetched
    //     before anything is stored, instead of 16 dependent load->store round trips.
    const bool has_res = e.residual != nullptr, bwd = e.act == VLM_ACT_GELU_BWD || e.act == VLM_ACT_MUL_AUX, accum = OUT_F32 && e.accumulate;
    const int g = lane >> 4;
    const int nl = nw0 + (g & 1) * 16 + (g >> 1) * 8;  // + 32*jp
    float bia[2][8], gam[2][8], csum[2][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 8; ++r) csum[jp][r] = 0.f;
#define EPI_LOAD_COL_VECTORS()                                                                                          \
  _Pragma("unroll") for (int jp = 0; jp < 2; ++jp) _Pragma("unroll") for (int q = 0; q < 2; ++q) {                    \
    const f32x4 b4 = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + nl + jp * 32 + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f}; \
    const f32x4 g4 =                                                                                                    \
        e.col_scale ? *reinterpret_cast<const f32x4*>(e.col_scale + nl + jp * 32 + q * 4) : (f32x4){1.f, 1.f, 1.f, 1.f}; \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                     \
      bia[jp][q * 4 + r] = b4[r];                                                                                       \
      gam[jp][q * 4 + r] = g4[r];                                                                                       \
    }                                                                                                                   \
  }
    if (!LEAN) { EPI_LOAD_COL_VECTORS() }
#pragma unroll
    for (int hf = 0; hf < 4; ++hf) {  // one 16-row block at a time: 4-8 loads in flight, few live registers
      f32x4 rsd[1][2][2], old[1][2][2];
      bf16x8 hx[1][2];
      float rs[1];
      if (LEAN) { EPI_LOAD_COL_VECTORS() }
#pragma unroll
      for (int ii = 0; ii < 1; ++ii) {
        const size_t m = mw0 + (hf + ii) * 16 + (lane & 15);
        rs[ii] = e.row_scale ? e.row_scale[m] : 1.0f;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int n = nl + jp * 32;
          if (has_res) {
            rsd[ii][jp][0] = *reinterpret_cast<const f32x4*>(e.residual + m * e.ld_res + n);
            rsd[ii][jp][1] = *reinterpret_cast<const f32x4*>(e.residual + m * e.ld_res + n + 4);
          }
          if (bwd) hx[ii][jp] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(e.aux) + m * e.ld_aux + n);
          if (accum) {
            old[ii][jp][0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + m * p.ldc + n);
            old[ii][jp][1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + m * p.ldc + n + 4);
          }
        }
      }
#pragma unroll
      for (int ii = 0; ii < 1; ++ii) {
        const size_t m = mw0 + (hf + ii) * 16 + (lane & 15);
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int n = nl + jp * 32;
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // alpha first: a compiler-scheduled VALU op consumes the MFMA result (hazard handled by hipcc), the swap
            // then reads VALU results
            float x = acc[hf + ii][2 * jp][r] * e.alpha, y = acc[hf + ii][2 * jp + 1][r] * e.alpha;
            lane16_swap(x, y);
            v[r] = x;
            v[4 + r] = y;
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += bia[jp][r];
          if (bwd) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= act_bwd_factor(e.act, (float)hx[ii][jp][r]);
          } else if (e.act == VLM_ACT_GELU_DERIV) {
            bf16x8 h;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              float dg;
              v[r] = gelu_erf_both(v[r], dg);
              h[r] = (bf16_t)dg;
            }
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(e.aux) + m * e.ld_aux + n) = h;
          } else {
            if (e.aux) {
              bf16x8 h;
#pragma unroll
              for (int r = 0; r < 8; ++r) h[r] = (bf16_t)v[r];
              *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(e.aux) + m * e.ld_aux + n) = h;
            }
            if (e.act == VLM_ACT_GELU) {
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
            }
          }
          if (e.col_scale) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= gam[jp][r];
          }
          if (e.row_scale) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= rs[ii];
          }
          if (has_res) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += rsd[ii][jp][r >> 2][r & 3];
          }
          if (OUT_F32 && accum) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += old[ii][jp][r >> 2][r & 3];
          }
          if (!LEAN && e.col_sum) {
#pragma unroll
            for (int r = 0; r < 8; ++r) csum[jp][r] += v[r];
          }
          if (OUT_F32) {
            float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
            *reinterpret_cast<f32x4*>(c) = (f32x4){v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(c + 4) = (f32x4){v[4], v[5], v[6], v[7]};
          } else {
            bf16x8 o;
#pragma unroll
            for (int r = 0; r < 8; ++r) o[r] = (bf16_t)v[r];
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n) = o;
          }
        }
      }
    }
    if (!LEAN && e.col_sum) {
      // column sums of this wave's 64 rows: 4 row blocks summed in the lane above, the 16 rows of a block by DPP
      // rotations inside the 16-lane row; lane 0 of each row then owns 16 columns -> 64 fp32 atomics per wave
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float t = csum[jp][r];
          t += dpp_f32<0x128>(t);
          t += dpp_f32<0x124>(t);
          t += dpp_f32<0x122>(t);
          t += dpp_f32<0x121>(t);
          if ((lane & 15) == 0) {
            // red (workgroup-uniform, interior tiles): partials meet in LDS and leave as full-width 256-B atomics;
            // 4-lane atomics straight from here cost +180 us on the fc2 dgrad (memory-side atomics, contended rows)
            if (red) red[red_off + (g & 1) * 16 + (g >> 1) * 8 + jp * 32 + r] = t;
            else atomicAdd(e.col_sum + nl + jp * 32 + r, t);
          }
        }
      return -1;
    }
    // stores of the fast path: 4 row blocks x 2 column pairs x (two 16-B halves for f32) + the aux (pre-activation) copy
    return 8 * (OUT_F32 ? 2 : 1) + ((e.aux && !bwd) ? 8 : 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mw0 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float rs = e.row_scale ? e.row_scale[m] : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nw0 + j * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      const bool full = vec_ok && (n + 3 < p.N);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * e.alpha;
      if (full) {
        if (e.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += b[r];
        }
        if (e.act == VLM_ACT_GELU_BWD || e.act == VLM_ACT_MUL_AUX) {
          const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(e.aux) + (size_t)m * e.ld_aux + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= act_bwd_factor(e.act, (float)h[r]);
        } else if (e.act == VLM_ACT_GELU_DERIV) {
          bf16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float dg;
            v[r] = gelu_erf_both(v[r], dg);
            h[r] = (bf16_t)dg;
          }
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(e.aux) + (size_t)m * e.ld_aux + n) = h;
        } else {
          if (e.aux) {
            bf16x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (bf16_t)v[r];
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(e.aux) + (size_t)m * e.ld_aux + n) = h;
          }
          if (e.act == VLM_ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
          }
        }
        if (e.col_scale) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(e.col_scale + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= g[r];
        }
        if (e.row_scale) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= rs;
        }
        if (e.residual) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(e.residual + (size_t)m * e.ld_res + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += x[r];
        }
        if (OUT_F32) {
          float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
          f32x4 o = {v[0], v[1], v[2], v[3]};
          if (e.accumulate) {
            const f32x4 old = *reinterpret_cast<const f32x4*>(c);
            o += old;
          }
          if (e.col_sum) {  // edge tiles only: plain atomics
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(e.col_sum + n + r, o[r]);
          }
          *reinterpret_cast<f32x4*>(c) = o;
        } else {
          if (e.col_sum) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(e.col_sum + n + r, v[r]);
          }
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = o;
        }
      } else {  // ragged N tail: scalar path
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
          float x = v[r];
          if (e.bias) x += e.bias[n + r];
          if (e.act == VLM_ACT_GELU_BWD || e.act == VLM_ACT_MUL_AUX) {
            x *= act_bwd_factor(e.act, (float)reinterpret_cast<const bf16_t*>(e.aux)[(size_t)m * e.ld_aux + n + r]);
          } else if (e.act == VLM_ACT_GELU_DERIV) {
            float dg;
            x = gelu_erf_both(x, dg);
            reinterpret_cast<bf16_t*>(e.aux)[(size_t)m * e.ld_aux + n + r] = (bf16_t)dg;
          } else {
            if (e.aux) reinterpret_cast<bf16_t*>(e.aux)[(size_t)m * e.ld_aux + n + r] = (bf16_t)x;
            if (e.act == VLM_ACT_GELU) x = gelu_erf(x);
          }
          if (e.col_scale) x *= e.col_scale[n + r];
          if (e.row_scale) x *= rs;
          if (e.residual) x += e.residual[(size_t)m * e.ld_res + n + r];
          if (OUT_F32 && e.accumulate) x += reinterpret_cast<const float*>(p.C)[(size_t)m * p.ldc + n + r];
          if (e.col_sum) atomicAdd(e.col_sum + n + r, x);
          if (OUT_F32) {
            reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n + r] = x;
          } else {
            reinterpret_cast<bf16_t*>(p.C)[(size_t)m * p.ldc + n + r] = (bf16_t)x;
          }
        }
      }
    }
  }
  return -1;
}

// DMA_A / DMA_B: stage that operand by LDS-DMA (else through registers).  SPLITK: K is cut over gridDim.x / tiles
// slices, the MFMA is issued un-swapped so that 16 consecutive lanes hold 16 consecutive output columns, and the
// epilogue is a plain fp32 atomicAdd (C += alpha*acc): wgrad reduces over ~13.5k tokens into only 36-144 output
// tiles, a single-pass grid leaves most of the 256 CUs idle and every resident workgroup latency-bound.
template <bool TA, bool TB, bool OUT_F32, bool DMA_A, bool DMA_B, bool SPLITK>
__global__ __launch_bounds__(GEMM_THREADS, 2) void vlm_gemm_kernel(const gemm_params_t p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves, 64x64 each
  GEMM_STAMP(0)

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous
  // run of tiles with n fastest so a 128-row A panel is reused out of that XCD's L2 (bijective for any grid).
  const uint32_t ntile = p.tiles_m * p.tiles_n;
  uint32_t tile, split = 0;
  if (SPLITK) {
    tile = blockIdx.x % ntile;
    split = blockIdx.x / ntile;
  } else {
    const uint32_t nblk = gridDim.x;
    const uint32_t bid = blockIdx.x;
    const uint32_t q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  // grouped raster: the ~64 tiles an XCD runs at once form an 8 (m) x 8 (n) block, so its L2 holds 16 operand panels
  // instead of the 3.5 + tiles_n panels of a row-major sweep (qkv/fc1: 4.2-5.2 MB > the 4 MiB L2)
  uint32_t tm, tn;
  {
    const uint32_t gm = (uint32_t)p.group_m, gsz = gm * p.tiles_n, grp = tile / gsz, first = grp * gm;
    const uint32_t rows = min(gm, (uint32_t)p.tiles_m - first), in = tile - grp * gsz;
    tm = first + in % rows;
    tn = in / rows;
  }
  const uint32_t m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

  // buffer descriptors: byte extent = rows * ld * 2 so that ragged row tails read as zero
  const uint64_t a_rows = TA ? (uint64_t)p.K : (uint64_t)p.M;
  const uint64_t b_rows = TB ? (uint64_t)p.K : (uint64_t)p.N;
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)(a_rows * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)(b_rows * p.ldb * 2), 0x00020000);

  // stage s: A at smem + s*32 KiB, B 16 KiB after it
#define LDS_A(s) (smem + (s) * 2 * GEMM_TILE_BYTES)
#define LDS_B(s) (smem + (s) * 2 * GEMM_TILE_BYTES + GEMM_TILE_BYTES)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk_all = (p.K + GEMM_BK - 1) / GEMM_BK;
  const int kt0 = SPLITK ? (int)split * p.ksteps_per_split : 0;
  const int kt1 = SPLITK ? min(nk_all, kt0 + p.ksteps_per_split) : nk_all;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  u32x4 sa[4], sb[4];
  if (kt0 < kt1) {
    if (DMA_A) stage_dma<TA>(ra, LDS_A(0), m0, kt0 * GEMM_BK, p.lda, wave_u, lane);
    else { stage_load<TA>(sa, ra, m0, kt0 * GEMM_BK, p.lda, tid); stage_store<TA>(sa, LDS_A(0), tid); }
    if (DMA_B) stage_dma<TB>(rb, LDS_B(0), n0, kt0 * GEMM_BK, p.ldb, wave_u, lane);
    else { stage_load<TB>(sb, rb, n0, kt0 * GEMM_BK, p.ldb, tid); stage_store<TB>(sb, LDS_B(0), tid); }
  }
  __syncthreads();  // hipcc drains a pending LDS-DMA (vmcnt(0)) in front of the barrier
  GEMM_STAMP(1)

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < kt1) {  // next tile's loads fly during this tile's MFMAs
      if (DMA_A) stage_dma<TA>(ra, LDS_A(cur ^ 1), m0, (kt + 1) * GEMM_BK, p.lda, wave_u, lane);
      else stage_load<TA>(sa, ra, m0, (kt + 1) * GEMM_BK, p.lda, tid);
      if (DMA_B) stage_dma<TB>(rb, LDS_B(cur ^ 1), n0, (kt + 1) * GEMM_BK, p.ldb, wave_u, lane);
      else stage_load<TB>(sb, rb, n0, (kt + 1) * GEMM_BK, p.ldb, tid);
    }
    const unsigned char* la = LDS_A(cur);
    const unsigned char* lb = LDS_B(cur);
    // all 16 fragment reads of the K-step are issued before its 32 MFMAs: the compiler's own order (6 reads, wait,
    // 2 MFMAs, wait ...) left ~50 % of the wave cycles parked on lgkmcnt (SQ_WAIT_ANY), see profiles/
    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i] = frag_load<TA>(la, wm * 4 + i, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[ks][j] = frag_load<TB>(lb, wn * 4 + j, ks, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (SPLITK)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < kt1) {
      if (!DMA_A) stage_store<TA>(sa, LDS_A(cur ^ 1), tid);
      if (!DMA_B) stage_store<TB>(sb, LDS_B(cur ^ 1), tid);
    }
    __syncthreads();
  }
  GEMM_STAMP(2)

  if (SPLITK) {
    float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + i * 16 + 4 * (lane >> 4) + r;
          if (m < p.M && n < p.N) atomicAdd(C + (size_t)m * p.ldc + n, acc[i][j][r] * p.epi.alpha);
        }
      }
    return;
  }

  {
    // column sums (bias gradient): whole-tile-interior workgroups reduce their 2x2 waves through LDS (free after the
    // loop's last barrier) and issue two 64-lane contiguous atomics; edge tiles fall back to per-wave atomics
    const bool lds_sum = p.epi.col_sum && m0 + GEMM_BM <= (uint32_t)p.M && n0 + GEMM_BN <= (uint32_t)p.N && p.epi.reserved == 1;
    float* red = reinterpret_cast<float*>(smem);
    gemm_epilogue<OUT_F32>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, lds_sum ? red : nullptr, wm * 128 + wn * 64);
    if (lds_sum) {
      __syncthreads();
      if (tid < 128) {
        const float part = red[tid] + red[128 + tid];
        // workspace mode: plain store of this tile's column sums, folded later by vlm_colreduce_batch (no atomics: the
        // memory-side float atomics of 425 tile rows on the same 12 KiB cost +75 us per launch); else accumulate directly
        if (p.epi.col_sum_ws) p.epi.col_sum_ws[((size_t)tm * 2) * p.N + n0 + tid] = part;
        else atomicAdd(p.epi.col_sum + n0 + tid, part);
      }
    }
  }
  GEMM_STAMP_END()
}

template <bool TA, bool TB, bool OUT_F32, bool DMA_A, bool DMA_B, bool SPLITK>
static int launch_gemm(const gemm_params_t& p, hipStream_t stream) {
  const size_t smem = 4 * GEMM_TILE_BYTES;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vlm_gemm_kernel<TA, TB, OUT_F32, DMA_A, DMA_B, SPLITK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VLM_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n * (SPLITK ? p.splits : 1)), block(GEMM_THREADS);
  hipLaunchKernelGGL((vlm_gemm_kernel<TA, TB, OUT_F32, DMA_A, DMA_B, SPLITK>), grid, block, smem, stream, p);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ======================================================================================================================
// 256x256 macro tile, ONE workgroup per CU: 4 waves (2x2) of 128x128, both operands K-contiguous (forward, and dgrad
// against the transposed weight shadow).  Why a second kernel: at the MFMA bound a 128x128 workgroup tile with 64x64
// waves asks for 64 B/clk/CU of global->LDS traffic (the vector-memory path's peak) AND 256 B/clk/CU of LDS reads + DMA
// writes (the LDS array's peak) -- every pipe is at 100 % together, which in practice is the ~1500 cycles per 1024 MFMA
// cycles that tools/stamp_gemm.py shows.  Waves of 128x128 halve the fragment bytes per MFMA and the 256x256 tile halves
// the staged bytes per MFMA, so the matrix pipe is the only pipe near its limit.  With one wave per SIMD nothing but the
// wave's own instruction stream hides latency, so the K loop is software-pipelined by hand:
//   * K steps of 32.  Stage S is loaded global -> registers in step S-4 (two register sets, by parity), written to LDS
//     buffer S&1 in step S-2 (ds_write_b128), read as fragments in step S-1 and consumed by 64 MFMAs in step S;
//     sched_group_barrier spreads the 16 ds_read_b128, 8 ds_write_b128 and 8 buffer loads of a step between its MFMAs;
//   * one barrier per step (top of the step): the writes of stage S+1 are visible and nobody still reads the buffer that
//     stage S+2 is about to overwrite.
// Tried and dropped (tools/scratch/gemm_bench.hip; git history): an LDS-DMA version with four LDS stages and counted vmcnt
// (tie: 1128 vs 1092 TFLOP/s at 4096x4096x8192, twice the LDS); a PERSISTENT version whose next tile's first stages went
// out by inline-asm LDS-DMA under the last two K steps with a hand-counted vmcnt in front of their use (tie: 319 vs 321 us
// at 54296x3072x768 -- vmcnt is one in-order counter, so the next tile's first register-staged loads still wait for
// the epilogue's stores; with the C stores removed the same kernel takes 270 us: the store drain is what remains); a
// 256x128 tile with 128x64 waves and TWO workgroups per CU, so that one's epilogue runs under the other's K loop (same
// LDS image and looped epilogue, no fragment double buffer: 351 vs 329 us at 54296x3072x768, 335 vs 302 at K = 3072,
// 949 vs 1262 TFLOP/s at 4096^3 -- 1.5x the staged and fragment bytes per MFMA cost more than the overlap returns); all four
// prologue stages requested at once through two more register sets (one round trip instead of two: the allocator then
// keeps accumulators in VGPRs, 1675 vs 1345 cycles per step).
// LDS image of an operand stage: [256 rows][32 k] bf16, 64-B rows, 16-B slot s of row r holds chunk s ^ f((r>>2)&3),
// f = (0,2,3,1): conflict-free for ds_read_b128's lane groups (rows {0-3,12-15} x chunk c with rows {4-11} x chunk c^1).
#define BIG_BM 256
#define BIG_BN 256
#define BIG_BK 32
#define BIG_OP_BYTES (256 * BIG_BK * 2)  // 16 KiB per operand per stage

__device__ __forceinline__ uint32_t big_swz(uint32_t q) { return (0x78u >> (2 * q)) & 3u; }  // f = (0,2,3,1)

// ---- looped epilogue of a 128x128 wave tile through a wave-private LDS transpose ------------------------------------
// The straight-line epilogue above is executed once per wave and is instruction-FETCH bound (tools/scratch/gemm_bench.hip:
// with every store removed a 128x128 wave tile still took 50k cycles in it).  Here a 16-row block of accumulators is
// dropped into LDS as it stands (ds_write_b128, 528-B row pitch: conflict-free) and read back row-major: lane -> row
// 4t + (lane>>4), 8 consecutive columns (lane&15)*8 -- so one short loop body serves all 8 row blocks out of the
// instruction cache, every global access of a wave instruction is 4 rows x 512 contiguous bytes (f32) / 256 (bf16), and the
// per-element inputs (residual, GELU' argument, row scale) of the next block are in flight while this one is processed.
#ifndef EPIL_STORE_AUX
#define EPIL_STORE_AUX 2  // cache policy of the epilogue's C / aux stores: 2 = nt (54296x3072x768 plain 331 -> 299 us, GELU 421 -> 394; in situ +0.9 %: the consumer then reads more from HBM); 0 plain, 16 sc1 (no gain)
#endif
#ifndef EPIL_AUX_STORE_AUX
#define EPIL_AUX_STORE_AUX 2  // the saved pre-activation / branch copy is not read again before the backward pass: nt
#endif
#ifndef EPIL_LOAD_AUX
#define EPIL_LOAD_AUX 2  // cache policy of the GELU' argument's loads (the pre-activation saved by the forward pass, read once): 2 = nt
                         // (54296x3072x768: 419 -> 403 us).  The residual loads stay plain: nt there cost 3-40 % (in-place stream)
#endif
#define EPIL_PITCH (128 * 4 + 16)
#define EPIL_WAVE_BYTES (32 * EPIL_PITCH)
struct epil_in_t {
  f32x4 rsd[4][2];
  bf16x8 hx[4];
  float rs[4];
};

// Everything of the looped epilogue that never touches an accumulator.  The accumulators stay in the kernel body, where
// each is named statically (handing the arrays through a function or a closure left one of them in scratch).
// Every per-element access is a raw buffer operation: an input the call does not have gets a zero-length descriptor
// (loads return 0, stores are dropped, nothing reaches memory) and rows >= M fall off the end of the real ones -- so the
// loop body has NO branch around a memory operation and a fixed number of them, which is what lets the compiler wait for
// the next block's inputs with a counted vmcnt instead of vmcnt(0) (= draining this block's stores every time).
// RES: residual (+ optional row scale) inputs; AUX: 0 none, 1 the pre-activation copy is stored, 2 GELU' argument is loaded.
// What a variant does not have costs nothing (a zero-length descriptor would still cost the round trip: the dummy loads
// of an all-in-one version took 14k of its 25k cycles).
// STORE_AUX: cache policy of the C / aux stores.  WCOLS: columns of the wave tile (128; 64 served the 256x128 experiment): WCOLS/8 lanes cover a row, 512/WCOLS rows per wave
// instruction, WCOLS*4+16 bytes of LDS row pitch.
template <bool OUT_F32, bool RES, int AUX, int WCOLS = 128, int STORE_AUX = EPIL_STORE_AUX, bool ALPHA = false>
struct big_epilogue_t {
  static constexpr int LPR = WCOLS / 8, RPG = 64 / LPR, T = 16 / RPG, PITCH = WCOLS * 4 + 16;
  const gemm_params_t& p;
  const unsigned char* rd;  // this lane's read address in the wave's LDS transpose: row lane / LPR, 8 columns (lane % LPR) * 8
  __amdgpu_buffer_rsrc_t r_c, r_res, r_aux, r_rs;
  uint32_t mw0, lr, n;
  bool bwd;
  float bia[8], gam[8], csum[8];

  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, const void* fallback, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base ? base : fallback), 0, base ? (int)bytes : 0, 0x00020000);
  }

  __device__ __forceinline__ big_epilogue_t(const gemm_params_t& p_, const unsigned char* wl, uint32_t mw0_, uint32_t nw0, int lane)
      : p(p_), mw0(mw0_) {
    const vlm_epilogue_t& e = p.epi;
    bwd = AUX == 2;
    lr = lane / LPR;
    n = nw0 + (lane % LPR) * 8;
    rd = wl + lr * PITCH + (lane % LPR) * 32;
    const uint64_t c_bytes = (uint64_t)p.M * p.ldc * (OUT_F32 ? 4 : 2);
    r_c = rsrc(p.C, p.C, c_bytes);
    r_res = rsrc(e.residual, p.C, (uint64_t)p.M * e.ld_res * 4);
    r_aux = rsrc(e.aux, p.C, (uint64_t)p.M * e.ld_aux * 2);
    r_rs = rsrc(e.row_scale, p.C, (uint64_t)p.M * 4);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 b4 = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + n + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 g4 = e.col_scale ? *reinterpret_cast<const f32x4*>(e.col_scale + n + q * 4) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bia[q * 4 + r] = b4[r];
        gam[q * 4 + r] = g4[r];
        csum[q * 4 + r] = 0.f;
      }
    }
  }

  // per-element inputs of the 16-row block i (4 rows per lane): a fixed number of loads per variant, no branch
  __device__ __forceinline__ void load_inputs(epil_in_t& in, int i) const {
    const vlm_epilogue_t& e = p.epi;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const uint32_t m = mw0 + 16 * i + RPG * t + lr;
      if (RES) {
        in.rs[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, m * 4, 0, 0));
        const uint32_t ro = (m * (uint32_t)e.ld_res + n) * 4;
        in.rsd[t][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, ro, 0, 0));
        in.rsd[t][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, ro + 16, 0, 0));
      }
      if (AUX == 2) in.hx[t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_aux, (m * (uint32_t)e.ld_aux + n) * 2, 0, EPIL_LOAD_AUX));
    }
  }

  // the 16-row block i, read back from LDS at lds_off: same operation order as gemm_epilogue
  __device__ __forceinline__ void process(const epil_in_t& in, int i, int lds_off) {
    const vlm_epilogue_t& e = p.epi;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const uint32_t m = mw0 + 16 * i + RPG * t + lr;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(rd + lds_off + t * RPG * PITCH);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(rd + lds_off + t * RPG * PITCH + 16);
      // One wave per SIMD executes this alone, so every vector instruction is paid in full: the plain variant used to spend 35
      // per 8 elements, 22 of them on factors that were 1 and on sums nobody asked for (12.4k cycles per tile against a
      // 32k-cycle K loop at K = 768; 7.5k now).  So: alpha is 1 here unless ALPHA (the wgrad slices; other calls with a factor
      // go to the 128x128 kernel: launch_gemm_big_variant), the column scale exists only in the residual variants and the
      // column sums only in the GELU'-argument variant (AUX == 2).
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = ALPHA ? lo[r] * e.alpha + bia[r] : lo[r] + bia[r];
        v[4 + r] = ALPHA ? hi[r] * e.alpha + bia[4 + r] : hi[r] + bia[4 + r];
      }
      if (bwd) {
        if (e.act == VLM_ACT_MUL_AUX) {  // the forward pass saved gelu' itself (wave-uniform branch)
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] *= (float)in.hx[t][r];
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] *= gelu_erf_grad((float)in.hx[t][r]);
        }
      } else if (AUX == 1 && e.act == VLM_ACT_GELU_DERIV) {
        bf16x8 h;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float dg;
          v[r] = gelu_erf_both(v[r], dg);
          h[r] = (bf16_t)dg;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), r_aux, (m * (uint32_t)e.ld_aux + n) * 2, 0, EPIL_AUX_STORE_AUX);
      } else {
        if (AUX == 1) {
          bf16x8 h;
#pragma unroll
          for (int r = 0; r < 8; ++r) h[r] = (bf16_t)v[r];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), r_aux, (m * (uint32_t)e.ld_aux + n) * 2, 0, EPIL_AUX_STORE_AUX);
        }
        if (e.act == VLM_ACT_GELU) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
        }
      }
      if (RES) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= gam[r];  // 1.0 without a column scale: exact
        if (e.row_scale) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] *= in.rs[t];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += in.rsd[t][r >> 2][r & 3];
      }
      if (AUX == 2) {
        if (m < (uint32_t)p.M) {
#pragma unroll
          for (int r = 0; r < 8; ++r) csum[r] += v[r];
        }
      }
      if (OUT_F32) {
        const uint32_t co = (m * (uint32_t)p.ldc + n) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v[0], v[1], v[2], v[3]}), r_c, co, 0, STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v[4], v[5], v[6], v[7]}), r_c, co + 16, 0, STORE_AUX);
      } else {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = (bf16_t)v[r];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_c, (m * (uint32_t)p.ldc + n) * 2, 0, STORE_AUX);
      }
    }
  }

  // column sums of the wave tile: the 4 lane groups hold the same columns for rows = lane>>4 (mod 4)
  __device__ __forceinline__ void finish(int lane, float* ws_row) {
    const vlm_epilogue_t& e = p.epi;
    if (AUX != 2 || !e.col_sum) return;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (LPR == 8) csum[r] += __shfl_xor(csum[r], 8);
      csum[r] += __shfl_xor(csum[r], 16);
      csum[r] += __shfl_xor(csum[r], 32);
    }
    if (lane < LPR) {
      if (ws_row) {
        *reinterpret_cast<f32x4*>(ws_row + n) = (f32x4){csum[0], csum[1], csum[2], csum[3]};
        *reinterpret_cast<f32x4*>(ws_row + n + 4) = (f32x4){csum[4], csum[5], csum[6], csum[7]};
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) atomicAdd(e.col_sum + n + r, csum[r]);
      }
    }
  }
};

// The loop both 256x256 kernels run over their wave tile once `ep`, `inA` (block 0 requested) and `wl` exist: 32 rows per
// trip -- two 16-row blocks dropped into LDS, read back row-major, finished and stored, the inputs of the blocks after
// them requested meanwhile.  ONE loop body in the instruction cache for the whole wave tile; the trip counter only
// picks which (statically named) accumulators are dropped.  Every load is unconditional (the last trip re-reads block 7):
// the operation count of a trip stays fixed and the compiler's vmcnt waits stay counted.  (Four rotating input sets,
// inputs three blocks ahead: no gain for the GELU' variant -- its epilogue is VALU-bound -- spills for the residual
// variants, and 40 more VGPRs that slowed the K loop of the others from 1488 to 1828 cycles per step.)
#define EPIL_DUMP_ROW(A0, A1, II, OFF)                                                           \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
    *reinterpret_cast<f32x4*>(wr + (OFF) + 64 * j) = A0[II][j];                                  \
    *reinterpret_cast<f32x4*>(wr + (OFF) + 64 * (4 + j)) = A1[II][j];                            \
  }
// AUX == 2 without the residual stream (fc2 dgrad: the saved GELU' factor is this epilogue's only per-element input, 16 B per
// lane and row): with the factor SAVED by the forward pass the epilogue has next to no arithmetic left, and two input sets in
// flight made it wait for memory four times per wave tile.  AHEAD = 6 of the eight blocks' factors are requested at once after
// the K loop (96 registers: the fragments and staging sets are dead by then; all eight spilled five), each later one as soon as
// a set is used up: one exposed latency per tile; the loop is unrolled so that every set has a static name.
// (The fp32-residual variants with AHEAD = 4, 36 registers per set: no change, 35k cycles either way -- their epilogue runs at the
// pace of the 655 KB it moves, not of its input latency.)
#define BIG_EPILOGUE_LOOP_DEEP(AHEAD)                                                                             \
  unsigned char* wr = wl + (lane & 15) * EPIL_PITCH + (lane >> 4) * 16;                                            \
  epil_in_t in8[8];                                                                                                \
  _Pragma("unroll") for (int b8 = 0; b8 < (AHEAD); ++b8) ep.load_inputs(in8[b8], b8);                              \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                  \
    switch (q) {                                                                                                   \
      case 0: EPIL_DUMP_ROW(acc00, acc01, 0, 0) EPIL_DUMP_ROW(acc00, acc01, 1, 16 * EPIL_PITCH) break;             \
      case 1: EPIL_DUMP_ROW(acc00, acc01, 2, 0) EPIL_DUMP_ROW(acc00, acc01, 3, 16 * EPIL_PITCH) break;             \
      case 2: EPIL_DUMP_ROW(acc10, acc11, 0, 0) EPIL_DUMP_ROW(acc10, acc11, 1, 16 * EPIL_PITCH) break;             \
      default: EPIL_DUMP_ROW(acc10, acc11, 2, 0) EPIL_DUMP_ROW(acc10, acc11, 3, 16 * EPIL_PITCH) break;            \
    }                                                                                                              \
    ep.process(in8[2 * q], 2 * q, 0);                                                                              \
    if (2 * q + (AHEAD) < 8) ep.load_inputs(in8[2 * q + (AHEAD)], 2 * q + (AHEAD));                                \
    ep.process(in8[2 * q + 1], 2 * q + 1, 16 * EPIL_PITCH);                                                        \
    if (2 * q + 1 + (AHEAD) < 8) ep.load_inputs(in8[2 * q + 1 + (AHEAD)], 2 * q + 1 + (AHEAD));                    \
  }
#define BIG_EPILOGUE_LOOP()                                                                                       \
  unsigned char* wr = wl + (lane & 15) * EPIL_PITCH + (lane >> 4) * 16;                                            \
  ep.load_inputs(inB, 1);                                                                                          \
  _Pragma("unroll 1") for (int q = 0; q < 4; ++q) {                                                                \
    switch (q) {                                                                                                   \
      case 0: EPIL_DUMP_ROW(acc00, acc01, 0, 0) EPIL_DUMP_ROW(acc00, acc01, 1, 16 * EPIL_PITCH) break;             \
      case 1: EPIL_DUMP_ROW(acc00, acc01, 2, 0) EPIL_DUMP_ROW(acc00, acc01, 3, 16 * EPIL_PITCH) break;             \
      case 2: EPIL_DUMP_ROW(acc10, acc11, 0, 0) EPIL_DUMP_ROW(acc10, acc11, 1, 16 * EPIL_PITCH) break;             \
      default: EPIL_DUMP_ROW(acc10, acc11, 2, 0) EPIL_DUMP_ROW(acc10, acc11, 3, 16 * EPIL_PITCH) break;            \
    }                                                                                                              \
    ep.process(inA, 2 * q, 0);                                                                                     \
    ep.load_inputs(inA, q < 3 ? 2 * q + 2 : 7);                                                                    \
    ep.process(inB, 2 * q + 1, 16 * EPIL_PITCH);                                                                   \
    ep.load_inputs(inB, q < 3 ? 2 * q + 3 : 7);                                                                    \
  }

// GROUPED (vlm_gemm_bf16_grouped, the modality experts of an all_moe block in ONE launch, vision_transformer.py:607-681): the
// M tiles of the grid are the concatenation of every group's own 256-row tiles; a workgroup takes the weight, bias,
// column-sum targets and row bound of the group its tile belongs to (a scalar select over <= 4 groups) and is otherwise
// the same kernel -- the text expert's 14 row tiles ride in the image expert's rounds instead of a launch of their own.
template <bool OUT_F32, bool RES, int AUX, bool GROUPED = false>
__global__ __launch_bounds__(GEMM_THREADS, 1) void vlm_gemm_big_kernel(const gemm_params_t p_in) {
  __shared__ __attribute__((aligned(1024))) unsigned char sA0[BIG_OP_BYTES], sA1[BIG_OP_BYTES], sB0[BIG_OP_BYTES], sB1[BIG_OP_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char epl[4 * EPIL_WAVE_BYTES];  // wave-private epilogue transposes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const uint32_t nblk = gridDim.x, bid = blockIdx.x;
  const uint32_t q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const uint32_t tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  uint32_t tm, tn;
  {
    const uint32_t gm = (uint32_t)p_in.group_m, gsz = gm * p_in.tiles_n, grp = tile / gsz, first = grp * gm;
    const uint32_t rows = min(gm, (uint32_t)p_in.tiles_m - first), in = tile - grp * gsz;
    tm = first + in % rows;
    tn = in / rows;
  }
  gemm_params_t p_grp;  // GROUPED: this workgroup's view of the call (its group's weight, bias, column sums, row bound)
  uint32_t m0 = tm * BIG_BM, tm_ws = tm;
  if constexpr (GROUPED) {
    p_grp = p_in;
    int g = 0;
#pragma unroll
    for (int i = 1; i < VLM_GEMM_MAX_GROUPS; ++i)
      if (i < p_in.n_groups && tm >= (uint32_t)p_in.grp[i].tile0) g = i;
    const gemm_group_t G = p_in.grp[g];
    tm_ws = tm - (uint32_t)G.tile0;
    m0 = (uint32_t)G.row0 + tm_ws * BIG_BM;
    p_grp.M = G.row_end;
    p_grp.B = G.B;
    p_grp.ldb = G.ldb;
    p_grp.epi.bias = G.bias;
    p_grp.epi.col_sum = G.col_sum;
    p_grp.epi.col_sum_ws = G.col_sum_ws;
  }
  const gemm_params_t& p = GROUPED ? p_grp : p_in;
  GEMM_STAMP(0)
  const uint32_t n0 = tn * BIG_BN;
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((uint64_t)p.M * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)((uint64_t)p.N * p.ldb * 2), 0x00020000);

  // per-lane staging offsets (k0 = 0; the step adds 64 B per stage through the scalar offset): instruction j = wave + 4u
  // covers rows 16j .. 16j+15, lane -> row 16j + (lane>>2), global chunk lane&3.  (Written as "swizzled chunk, then the
  // swizzle taken out again" on purpose: with the direct form hipcc 7.2 keeps 40 accumulator registers in VGPRs and
  // shuffles them through a[52:55] inside the K loop -- 1 832 instead of 1 488 cycles per step.  tests/test_build_cpu.py
  // watches the register split of this kernel.)
  uint32_t offa[4], offb[4];
  {
    const uint32_t chunk = (lane & 3) ^ big_swz(lane >> 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t row = (wave + 4 * u) * 16 + (lane >> 2);
      offa[u] = ((m0 + row) * (uint32_t)p.lda + chunk * 8) * 2;
      offb[u] = ((n0 + row) * (uint32_t)p.ldb + chunk * 8) * 2;
    }
  }
  // fragment read addresses: 16-row block i of this wave's 128 rows: row = w*128 + 16 i + (lane&15), chunk lane>>4
  const uint32_t rd_slot = ((uint32_t)(lane >> 4) ^ big_swz((lane & 15) >> 2)) * 16 + (lane & 15) * 64;
  const uint32_t rda = wm * 128 * 64 + rd_slot, rdb = wn * 128 * 64 + rd_slot;

  // four separate 64x64 quadrants (static objects: a 4-D array indexed through unrolled loops stayed in scratch)
  f32x4 acc00[4][4], acc01[4][4], acc10[4][4], acc11[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc00[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc01[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc10[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc11[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  // "+a" pin: as soon as the code AFTER the K loop keeps many registers live (the input sets of BIG_EPILOGUE_LOOP_DEEP) hipcc's
  // allocator parks other values in accumulator registers and shuffles accumulators through VGPRs inside the K loop (96
  // v_accvgpr moves per pair of steps); an empty asm statement that wants every accumulator in an AGPR before the loop keeps
  // them there.  (A second pin in front of the epilogue made the allocator permute accumulators between the loop and the tail
  // steps through scratch.)  tests/test_build_cpu.py watches the outcome.
#define BIG_PIN_ACC()                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {         \
    asm volatile("" : "+a"(acc00[i][j]), "+a"(acc01[i][j]), "+a"(acc10[i][j]), "+a"(acc11[i][j]));      \
  }
  constexpr bool DEEP = AUX == 2 && !RES;  // BIG_EPILOGUE_LOOP_DEEP
  constexpr int AHEAD = 6;
  if constexpr (DEEP) { BIG_PIN_ACC() }

  const int nk = p.K / BIG_BK;
  const uint32_t mw0 = m0 + wm * 128, nw0 = n0 + wn * 128;
  unsigned char* const wl = epl + wave * EPIL_WAVE_BYTES;
#define EPIL_SETUP()                                       \
  big_epilogue_t<OUT_F32, RES, AUX> ep(p, wl, mw0, nw0, lane); \
  epil_in_t inA, inB;                                      \
  ep.load_inputs(inA, 0);
  bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];

#define BIG_READ(FA, FB, SA, SB)                                                         \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                        \
    FA[i] = *reinterpret_cast<const bf16x8*>((SA) + rda + i * 1024);                     \
    FB[i] = *reinterpret_cast<const bf16x8*>((SB) + rdb + i * 1024);                     \
  }
#define BIG_MFMA_Q(ACC, FA, FB, I0, J0)                                                                                \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                         \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[(J0) + j], FA[(I0) + i], ACC[i][j], 0, 0, 0);
#define BIG_MFMA(FA, FB)        \
  BIG_MFMA_Q(acc00, FA, FB, 0, 0) \
  BIG_MFMA_Q(acc01, FA, FB, 0, 4) \
  BIG_MFMA_Q(acc11, FA, FB, 4, 4) \
  BIG_MFMA_Q(acc10, FA, FB, 4, 0)
  // Register staging: stage S is loaded global -> registers in step S-4 (two register sets, by parity), written to LDS
  // buffer S&1 in step S-2, read as fragments in step S-1 and consumed in step S.  ds_write_b128: lane -> row
  // 16j + (lane>>2), global chunk lane&3 -> slot (lane&3) ^ f(lane>>4); 8 consecutive lanes cover 128 contiguous bytes.
  uint32_t wr_off[4];
  {
    const uint32_t slot = (lane & 3) ^ big_swz(lane >> 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) wr_off[u] = ((wave + 4 * u) * 16 + (lane >> 2)) * 64 + slot * 16;
  }
  {
    const uint32_t unswz = (((lane & 3) ^ big_swz(lane >> 4)) - (lane & 3)) * 16;  // bytes, may wrap: uint32 arithmetic
#pragma unroll
    for (int u = 0; u < 4; ++u) { offa[u] -= unswz; offb[u] -= unswz; }
  }
  u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
#define BIG_GLOAD(RA_, RB_, S)                                                                        \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                     \
    RA_[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, offa[u], (S) * (BIG_BK * 2), 0);               \
    RB_[u] = __builtin_amdgcn_raw_buffer_load_b128(rb, offb[u], (S) * (BIG_BK * 2), 0);               \
  }
#define BIG_LWRITE(RA_, RB_, SA, SB)                                                                  \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                     \
    *reinterpret_cast<u32x4*>((SA) + wr_off[u]) = RA_[u];                                             \
    *reinterpret_cast<u32x4*>((SB) + wr_off[u]) = RB_[u];                                             \
  }
  // step K: [barrier] write stage K+2 (register set K&1 -> LDS buffer K&1), load stage K+4 into that set, read the
  // fragments of stage K+1 (buffer (K+1)&1), 64 MFMAs on stage K.  GL / WR / RD are literals (one basic block).
#define BIG_STEP(K, GL, WR, RD, FCA, FCB, FNA, FNB, RSA, RSB, BUFA_W, BUFB_W, BUFA_R, BUFB_R)   \
  {                                                                                      \
    if (RD) {                                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
      __builtin_amdgcn_s_barrier();                                                      \
    }                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    if (WR) { BIG_LWRITE(RSA, RSB, BUFA_W, BUFB_W) }                                     \
    if (GL) { BIG_GLOAD(RSA, RSB, (K) + 4) }                                             \
    if (RD) { BIG_READ(FNA, FNB, BUFA_R, BUFB_R) }                                       \
    BIG_MFMA(FCA, FCB)                                                                   \
    _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
    }                                                                                    \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                      \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                 \
    }                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                   \
  }

  // prologue (nk >= 4, nk even: launcher): stages 0, 1 in LDS, 2, 3 in flight to registers, fragments of stage 0 read
  BIG_GLOAD(ra0, rb0, 0)
  BIG_GLOAD(ra1, rb1, 1)
  BIG_LWRITE(ra0, rb0, sA0, sB0)
  BIG_GLOAD(ra0, rb0, 2)
  BIG_LWRITE(ra1, rb1, sA1, sB1)
  BIG_GLOAD(ra1, rb1, 3)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  BIG_READ(fa0, fb0, sA0, sB0)
  GEMM_STAMP(1)

  int kt = 0;
  for (; kt < nk - 4; kt += 2) {
    BIG_STEP(kt + 0, 1, 1, 1, fa0, fb0, fa1, fb1, ra0, rb0, sA0, sB0, sA1, sB1)
    BIG_STEP(kt + 1, 1, 1, 1, fa1, fb1, fa0, fb0, ra1, rb1, sA1, sB1, sA0, sB0)
  }
#define BIG_TAIL_STEPS()                                                               \
  BIG_STEP(kt + 0, 0, 1, 1, fa0, fb0, fa1, fb1, ra0, rb0, sA0, sB0, sA1, sB1)          \
  BIG_STEP(kt + 1, 0, 1, 1, fa1, fb1, fa0, fb0, ra1, rb1, sA1, sB1, sA0, sB0)          \
  BIG_STEP(kt + 2, 0, 0, 1, fa0, fb0, fa1, fb1, ra0, rb0, sA0, sB0, sA1, sB1)          \
  BIG_STEP(kt + 3, 0, 0, 0, fa1, fb1, fa0, fb0, ra1, rb1, sA1, sB1, sA0, sB0)
  // epilogue: wave tiles are whole in N and keep the 16-B alignments (launcher), so every variant goes through the looped
  // LDS-transpose epilogue (column sums: a wave covers its 128-row half alone -- workspace slot 2 tm + wm when the half is
  // complete, else atomics)
  if constexpr (DEEP) {
    // nothing of the epilogue is set up before the last K step: its eight input sets are requested together afterwards, and
    // an early first set (below) only added to the register pressure of the tail steps (30 spilled registers)
    BIG_TAIL_STEPS()
    GEMM_STAMP(2)
    big_epilogue_t<OUT_F32, RES, AUX> ep(p, wl, mw0, nw0, lane);
    float* ws_row = (p.epi.col_sum_ws && mw0 + 128 <= (uint32_t)p.M) ? p.epi.col_sum_ws + ((size_t)(tm_ws * 2 + wm) * 2) * p.N : nullptr;
    BIG_EPILOGUE_LOOP_DEEP(AHEAD)
    ep.finish(lane, ws_row);
  } else {
    // the epilogue's first inputs (bias / scale vectors, the first 16-row block) are requested here, four K steps
    // before they are needed: nothing else loads from memory any more and the staging registers are free
    EPIL_SETUP()
    BIG_TAIL_STEPS()
    GEMM_STAMP(2)
    float* ws_row = (p.epi.col_sum_ws && mw0 + 128 <= (uint32_t)p.M) ? p.epi.col_sum_ws + ((size_t)(tm_ws * 2 + wm) * 2) * p.N : nullptr;
    BIG_EPILOGUE_LOOP()
    ep.finish(lane, ws_row);
  }
  GEMM_STAMP_END()
}

// ---- 256x256 tile for wgrad: dW[M,N] = A^T B with BOTH operands K-strided (A stored [K][M], B stored [K][N]) ---------
// Same pipeline as vlm_gemm_big_kernel (register staging, two LDS buffers, 32-deep steps, one barrier per step); the LDS
// image of an operand stage is [32 k][256 x] bf16 (512-B rows, 32-B chunks XOR-swizzled by (k&3) | ((k>>3)&1)<<2) and a
// fragment is two ds_read_b64_tr_b16.  The reduction over tokens is cut over gridDim.x / tiles slices (36 or fewer
// output tiles on 256 CUs); a slice does not add into C with atomics (1 024 256-B float atomics per workgroup would cost
// as much as 80 K steps, MI355X_MICROARCH.md "Global float atomics") but stores its fp32 tile into the caller's
// workspace [slice][M][N] through the looped epilogue, and splitk_reduce_kernel adds the slices into C.
// GROUPED (vlm_gemm_wgrad_grouped): the token rows of several experts reduce into several weight gradients in one launch;
// a workgroup's item belongs to one group, whose row range bounds its K slice (rows past the range read zeros).
template <bool GROUPED = false>
__global__ __launch_bounds__(GEMM_THREADS, 1) void vlm_gemm_bigT_kernel(const gemm_params_t p) {
  __shared__ __attribute__((aligned(1024))) unsigned char sA0[BIG_OP_BYTES], sA1[BIG_OP_BYTES], sB0[BIG_OP_BYTES], sB1[BIG_OP_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char epl[4 * EPIL_WAVE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of (slice, tile) items with the tile
  // fastest -- its ~32 resident workgroups are then (nearly) all tiles of ONE K slice, so every byte of that slice of A
  // and B enters the XCD's L2 once instead of once per tile row / column (PMC: 726 MB fetched for 417 MB of operands with
  // slice = block / tiles).
  const uint32_t ntile = p.tiles_m * p.tiles_n;
  const uint32_t q8 = gridDim.x >> 3, r8 = gridDim.x & 7, xcd = blockIdx.x & 7;
  uint32_t item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  uint32_t k_base = 0, k_end = (uint32_t)p.K, slice0 = 0;
  int nk = p.ksteps_per_split;  // even, >= 4: launcher
  if constexpr (GROUPED) {
    int g = 0;
#pragma unroll
    for (int i = 1; i < VLM_GEMM_MAX_GROUPS; ++i)
      if (i < p.n_groups && item >= (uint32_t)p.grpT[i].item0) g = i;
    const gemmT_group_t G = p.grpT[g];
    item -= (uint32_t)G.item0;
    k_base = (uint32_t)G.row0;
    k_end = (uint32_t)G.row_end;
    nk = G.kps;
    slice0 = (uint32_t)G.slice0;
  }
  const uint32_t split = item / ntile, tile = item - split * ntile;
  const uint32_t tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const uint32_t m0 = tm * BIG_BM, n0 = tn * BIG_BN;
  const uint32_t k_first = k_base + split * (uint32_t)nk * BIG_BK;
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((uint64_t)k_end * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)((uint64_t)k_end * p.ldb * 2), 0x00020000);

  // staging: wave instruction j = wave + 4u covers k-rows 2j, 2j+1; lane -> k-row 2j + (lane>>5), 16-B chunk lane&31.
  // Rows k >= K fall off the descriptor (zeros); x >= M or N reads the next row's head: those accumulators are dropped.
  uint32_t offa[4], offb[4], wr_off[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t row = 2 * (wave + 4 * u) + (lane >> 5), c16 = lane & 31;
    offa[u] = ((k_first + row) * (uint32_t)p.lda + m0 + c16 * 8) * 2;
    offb[u] = ((k_first + row) * (uint32_t)p.ldb + n0 + c16 * 8) * 2;
    wr_off[u] = row * 512 + (((c16 >> 1) ^ ((row & 3) | (((row >> 3) & 1) << 2))) << 5) + (c16 & 1) * 16;
  }
  // fragments: 16-wide x block i of this wave's 128, lane -> k-rows 8g + q (+4), g = lane>>4, q = (lane&15)>>2
  uint32_t rda[8], rdb[8];
  {
    const uint32_t g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, sw = q | ((g & 1) << 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      rda[i] = (8 * g + q) * 512 + wm * 256 + ((i ^ sw) << 5) + 8 * pp;
      rdb[i] = (8 * g + q) * 512 + wn * 256 + ((i ^ sw) << 5) + 8 * pp;
    }
  }
  f32x4 acc00[4][4], acc01[4][4], acc10[4][4], acc11[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc00[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc01[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc10[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc11[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  const uint32_t stepa = BIG_BK * (uint32_t)p.lda * 2, stepb = BIG_BK * (uint32_t)p.ldb * 2;
  bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
  u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
#define BIGT_TR(PTR) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(PTR))
#define BIGT_READ(FA, FB, SA, SB)                                                                    \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                    \
    const s16x4 al = BIGT_TR((SA) + rda[i]), ah = BIGT_TR((SA) + rda[i] + 4 * 512);                  \
    const s16x4 bl = BIGT_TR((SB) + rdb[i]), bh = BIGT_TR((SB) + rdb[i] + 4 * 512);                  \
    FA[i] = __builtin_bit_cast(bf16x8, (s16x8){al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]}); \
    FB[i] = __builtin_bit_cast(bf16x8, (s16x8){bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]}); \
  }
#define BIGT_GLOAD(RA_, RB_, S)                                                                      \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                    \
    RA_[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, offa[u], (S) * stepa, 0);                     \
    RB_[u] = __builtin_amdgcn_raw_buffer_load_b128(rb, offb[u], (S) * stepb, 0);                     \
  }
#define BIGT_LWRITE(RA_, RB_, SA, SB)                                                                \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                    \
    *reinterpret_cast<u32x4*>((SA) + wr_off[u]) = RA_[u];                                            \
    *reinterpret_cast<u32x4*>((SB) + wr_off[u]) = RB_[u];                                            \
  }
#define BIGT_STEP(K, GL, WR, RD, FCA, FCB, FNA, FNB, RSA, RSB, BUFA_W, BUFB_W, BUFA_R, BUFB_R)      \
  {                                                                                      \
    if (RD) {                                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
      __builtin_amdgcn_s_barrier();                                                      \
    }                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    if (WR) { BIGT_LWRITE(RSA, RSB, BUFA_W, BUFB_W) }                                    \
    if (GL) { BIGT_GLOAD(RSA, RSB, (K) + 4) }                                            \
    if (RD) { BIGT_READ(FNA, FNB, BUFA_R, BUFB_R) }                                      \
    BIG_MFMA(FCA, FCB)                                                                   \
    _Pragma("unroll") for (int g = 0; g < 32; ++g) {                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
    }                                                                                    \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                      \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                 \
    }                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                   \
  }
  BIGT_GLOAD(ra0, rb0, 0)
  BIGT_GLOAD(ra1, rb1, 1)
  BIGT_LWRITE(ra0, rb0, sA0, sB0)
  BIGT_GLOAD(ra0, rb0, 2)
  BIGT_LWRITE(ra1, rb1, sA1, sB1)
  BIGT_GLOAD(ra1, rb1, 3)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  BIGT_READ(fa0, fb0, sA0, sB0)
  int kt = 0;
  for (; kt < nk - 4; kt += 2) {
    BIGT_STEP(kt + 0, 1, 1, 1, fa0, fb0, fa1, fb1, ra0, rb0, sA0, sB0, sA1, sB1)
    BIGT_STEP(kt + 1, 1, 1, 1, fa1, fb1, fa0, fb0, ra1, rb1, sA1, sB1, sA0, sB0)
  }
  BIGT_STEP(kt + 0, 0, 1, 1, fa0, fb0, fa1, fb1, ra0, rb0, sA0, sB0, sA1, sB1)
  BIGT_STEP(kt + 1, 0, 1, 1, fa1, fb1, fa0, fb0, ra1, rb1, sA1, sB1, sA0, sB0)
  BIGT_STEP(kt + 2, 0, 0, 1, fa0, fb0, fa1, fb1, ra0, rb0, sA0, sB0, sA1, sB1)
  BIGT_STEP(kt + 3, 0, 0, 0, fa1, fb1, fa0, fb0, ra1, rb1, sA1, sB1, sA0, sB0)

  // this slice's fp32 tile -> workspace [split][M][N] (plain f32 epilogue, alpha applied; rows >= M dropped)
  gemm_params_t pl = p;
  pl.C = reinterpret_cast<float*>(p.C) + (size_t)(slice0 + split) * p.M * p.N;
  pl.ldc = p.N;
  const uint32_t mw0 = m0 + wm * 128, nw0 = n0 + wn * 128;
  unsigned char* const wl = epl + wave * EPIL_WAVE_BYTES;
  big_epilogue_t<true, false, 0, 128, 0, true> ep(pl, wl, mw0, nw0, lane);  // plain stores: the reduce launch reads the slices right away
  epil_in_t inA, inB;
  ep.load_inputs(inA, 0);
  BIG_EPILOGUE_LOOP()
}

// C[m][n] (+)= sum over slices of ws[s][m][n]: one f32x4 per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, float* __restrict__ C,
                                                            int ldc, int accumulate) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, per = (size_t)M * N / 4;
  if (i >= per) return;
  const size_t e = i * 4, m = e / N, n = e - m * N;
  f32x4 t = accumulate ? *reinterpret_cast<const f32x4*>(C + m * ldc + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < splits; ++s) t += *reinterpret_cast<const f32x4*>(ws + (size_t)s * M * N + e);
  *reinterpret_cast<f32x4*>(C + m * ldc + n) = t;
}

// grouped form: blockIdx.y = group; its slices start at slice0[g]
struct splitk_groups_t {
  float* C[VLM_GEMM_MAX_GROUPS];
  int slice0[VLM_GEMM_MAX_GROUPS], splits[VLM_GEMM_MAX_GROUPS], accumulate[VLM_GEMM_MAX_GROUPS];
};
__global__ __launch_bounds__(256) void splitk_reduce_grouped_kernel(const float* __restrict__ ws, const splitk_groups_t g, int M, int N, int ldc) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, per = (size_t)M * N / 4;
  if (i >= per) return;
  const int gi = blockIdx.y;
  float* const C = g.C[gi];
  const size_t e = i * 4, m = e / N, n = e - m * N;
  f32x4 t = g.accumulate[gi] ? *reinterpret_cast<const f32x4*>(C + m * ldc + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < g.splits[gi]; ++s) t += *reinterpret_cast<const f32x4*>(ws + (size_t)(g.slice0[gi] + s) * M * N + e);
  *reinterpret_cast<f32x4*>(C + m * ldc + n) = t;
}

// wgrad through the 256x256 kernel: 1 = not offered (no / too small workspace, shape), else the launch's return code
static int launch_gemm_bigT(gemm_params_t p, const vlm_epilogue_t* epi, hipStream_t stream) {
  // whole 256-column tiles only: the looped epilogue has no column bound (a 128-column tail's second wave would store into
  // the next row of the [M][N] slice and read operand columns that belong to the next k-row)
  if (!epi->splitk_ws || (p.N % BIG_BN) || (p.ldc % 4)) return 1;
  if ((uint64_t)p.M * p.N * 4 >= (1ull << 31)) return 1;
  p.tiles_m = (p.M + BIG_BM - 1) / BIG_BM;
  p.tiles_n = (p.N + BIG_BN - 1) / BIG_BN;
  const int ntile = p.tiles_m * p.tiles_n, nk = (p.K + BIG_BK - 1) / BIG_BK;
  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  int splits = cus / ntile;  // one round of workgroups, never a little more
  if (splits < 1) splits = 1;
  if (splits > nk / 16) splits = nk / 16;  // >= 16 steps per slice
  if (splits < 1) return 1;
  int kps = (nk + splits - 1) / splits;
  kps += kps & 1;
  if (kps < 4) kps = 4;
  splits = (nk + kps - 1) / kps;
  if ((size_t)splits * p.M * p.N * 4 > (size_t)epi->splitk_ws_bytes) return 1;
  p.ksteps_per_split = kps;
  p.splits = splits;
  float* const out = reinterpret_cast<float*>(p.C);
  const int ldc = p.ldc;
  p.C = epi->splitk_ws;
  hipLaunchKernelGGL(vlm_gemm_bigT_kernel<false>, dim3(ntile * splits), dim3(GEMM_THREADS), 0, stream, p);
  VLM_CHECK_LAUNCH();
  const size_t per = (size_t)p.M * p.N / 4;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const float*>(epi->splitk_ws), splits, p.M, p.N, out, ldc, epi->accumulate);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

template <bool OUT_F32, bool RES, int AUX, bool GROUPED = false>
static int launch_gemm_big(gemm_params_t p, hipStream_t stream) {
  if (!GROUPED) p.tiles_m = (p.M + BIG_BM - 1) / BIG_BM;  // GROUPED: the caller counted every group's own tiles
  p.tiles_n = (p.N + BIG_BN - 1) / BIG_BN;
  static const int group_m = [] {
    const char* e = getenv("VLM_GEMM_BIG_GROUP_M");
    int v = e ? atoi(e) : 0;
    if (v < 0) v = 0;
    return v;
  }();
  p.group_m = group_m ? group_m : (p.tiles_n >= 6 ? 4 : 1);
  GEMM_STAMP_ARM(p)
  hipLaunchKernelGGL((vlm_gemm_big_kernel<OUT_F32, RES, AUX, GROUPED>), dim3(p.tiles_m * p.tiles_n), dim3(GEMM_THREADS), 0, stream, p);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// The epilogue variants the 256x256 kernel is built for; anything else runs on the 128x128 kernel (return 1: not offered).
// GROUPED: the four variants a transformer block's forward and dgrad GEMMs use.
template <bool GROUPED = false>
static int launch_gemm_big_variant(const gemm_params_t& p, bool c_is_f32, hipStream_t s) {
  const vlm_epilogue_t& e = p.epi;
  const bool res = e.residual != nullptr;
  const int aux = e.aux ? ((e.act == VLM_ACT_GELU_BWD || e.act == VLM_ACT_MUL_AUX) ? 2 : 1) : 0;
  if (e.row_scale && !res) return 1;
  {  // what the looped epilogue leaves out (big_epilogue_t::process): a scale factor, a column scale without the residual stream,
     // column sums outside the GELU'-argument variant
    bool any_cs = e.col_sum != nullptr;
    if (GROUPED)
      for (int g = 0; g < p.n_groups; ++g) any_cs = any_cs || p.grp[g].col_sum != nullptr;
    if (e.alpha != 1.0f || (e.col_scale && !res) || (any_cs && aux != 2)) return 1;
  }
  if (!c_is_f32 && !res) {
    if (aux == 0) return launch_gemm_big<false, false, 0, GROUPED>(p, s);
    if (aux == 1) return launch_gemm_big<false, false, 1, GROUPED>(p, s);
    return launch_gemm_big<false, false, 2, GROUPED>(p, s);
  }
  if (c_is_f32 && aux != 2) {
    if (res && aux) return launch_gemm_big<true, true, 1, GROUPED>(p, s);
    if (res) return launch_gemm_big<true, true, 0, GROUPED>(p, s);  // round 5: the folded-LayerScale residual epilogue saves no aux
    if (GROUPED) return 1;
    if (aux == 0) return launch_gemm_big<true, false, 0>(p, s);
  }
  return 1;
}

// VLM_GEMM_BIG: 0 = never, 1 = by shape (default), 2 = whenever the kernel is legal (tests), 3 = by shape with the tail split; vlm_gemm_set_big_tile_mode
// overrides the environment (tests compare the two kernels in one process), -1 returns to it
static std::atomic<int> g_big_mode{-1};  // -1: follow the environment
static int gemm_big_mode() {
  const int m = g_big_mode.load(std::memory_order_relaxed);
  if (m >= 0) return m;
  static const int env_mode = [] {
    const char* e = getenv("VLM_GEMM_BIG");
    int v = e ? atoi(e) : 1;
    return v;
  }();
  return env_mode;
}
extern "C" int vlm_gemm_set_big_tile_mode(int mode) {
  if (mode < -1 || mode > 3) return VLM_ERR_ARG;
  g_big_mode.store(mode, std::memory_order_relaxed);
  return VLM_OK;
}

// ----------------------------------------------------------------------------------------------------------------------
// Tile-shape experiments (round 1, removed from the build; see DESIGN.md section 4 and git history for the code):
//  * 256x128x32, 4 waves of 128x64, three 24-KiB stages, two workgroups per CU: same ~1500 cycles per 32 MFMAs per wave
//    pair as this kernel (0.75x the operand bytes did not matter);
//  * 256x256, 8 waves, one workgroup per CU, two 64-KiB stages: 3400 cycles per 64-deep K step (MFMA bound 2048) and an
//    exposed 21k-cycle epilogue;
//  * persistent 256x256x32, four 32-KiB stages, counted vmcnt waits so the stores of tile t drain under tile t+1:
//    2088 cycles per 32-deep step (two waves of ONE workgroup on a SIMD run in lock step behind the same barrier and
//    do not overlap each other's LDS phase) and a 17.7k-cycle store-issue-bound epilogue (~7 B/clk/CU, as
//    MI355X_MICROARCH.md reports for store tails) that no partner workgroup hides.
//  * 64x128 tiles (waves of 32x64) for launches with < 1536 tiles (text pass, image pass N = 768, to fill the 512
//    workgroup slots): slower everywhere (proj fwd M = 13 574: 37 vs 32 us; 4096^3: 813 vs 1153 TFLOP/s).
//  * 128x128x32, three 16-KiB stages with counted vmcnt, THREE workgroups per CU (168 VGPRs): 328 vs 287 us (qkv
//    forward), 338 vs 271 us (K = 3072): a barrier every 16 MFMAs costs more than the third workgroup hides.
// With staging switched off (tools/stamp_gemm.py, variant _noload) this kernel's loop runs at the MFMA bound (993 of
// 1024 cycles per K step); with staging and no MFMAs it takes as long as the full loop: what remains is the vector-memory
// issue path (8 LDS-DMA instructions per wave and K step) overlapped only by the partner workgroup's MFMAs.
// staging policy: K-contiguous operands by LDS-DMA, K-strided operands through registers (VLM_GEMM_STAGE: 0 = all
// registers, 1 = all DMA, 2 = hybrid [default]); VLM_GEMM_SPLITK=0 disables split-K
static int gemm_stage_mode() {
  static const int mode = [] {
    const char* e = getenv("VLM_GEMM_STAGE");
    int v = e ? atoi(e) : 2;
    return v;
  }();
  return mode;
}
static int gemm_splitk_enabled() {
  static const int on = [] {
    const char* e = getenv("VLM_GEMM_SPLITK");
    int v = e ? atoi(e) : 1;
    return v;
  }();
  return on;
}

template <bool TA, bool TB, bool OUT_F32>
static int dispatch_stage(const gemm_params_t& p, hipStream_t s) {
  const int mode = gemm_stage_mode();
  const bool da = mode == 1 || (mode == 2 && !TA), db = mode == 1 || (mode == 2 && !TB);
  if (da && db) return launch_gemm<TA, TB, OUT_F32, true, true, false>(p, s);
  if (da) return launch_gemm<TA, TB, OUT_F32, true, false, false>(p, s);
  if (db) return launch_gemm<TA, TB, OUT_F32, false, true, false>(p, s);
  return launch_gemm<TA, TB, OUT_F32, false, false, false>(p, s);
}

static int gemm_dispatch(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                         int c_is_f32, const vlm_epilogue_t* epi, void* stream, bool allow_big);

extern "C" int vlm_gemm_bf16(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                             void* C, int ldc, int c_is_f32, const vlm_epilogue_t* epi, void* stream) {
  return gemm_dispatch(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, c_is_f32, epi, stream, true);
}

static int gemm_dispatch(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                         int c_is_f32, const vlm_epilogue_t* epi, void* stream, bool allow_big) {
  if (M < 0 || N < 0 || K < 0 || !C) return VLM_ERR_ARG;
  if (M == 0 || N == 0) return VLM_OK;
  if (!A || !B || !epi) return VLM_ERR_ARG;
  // 16-B staging granules: leading dimensions and bases must keep 8-element alignment
  if ((lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return VLM_ERR_ARG;
  // K-contiguous operands need whole 64-deep K tiles (their ragged tail would alias the next row)
  if ((!ta || !tb) && (K % GEMM_BK)) return VLM_ERR_UNSUPPORTED;
  if (((ldc & 3) == 0) && ((uintptr_t)C & 15)) return VLM_ERR_ARG;
  if (epi->accumulate && !c_is_f32) return VLM_ERR_ARG;
  if ((epi->act == VLM_ACT_GELU_BWD || epi->act == VLM_ACT_MUL_AUX || epi->act == VLM_ACT_GELU_DERIV) && !epi->aux) return VLM_ERR_ARG;
  if (epi->act < VLM_ACT_NONE || epi->act > VLM_ACT_MUL_AUX) return VLM_ERR_ARG;
  if (epi->col_sum_ws && (!epi->col_sum || (N % GEMM_BN) != 0 || ta)) return VLM_ERR_ARG;
  const uint64_t a_bytes = (uint64_t)(ta ? K : M) * lda * 2, b_bytes = (uint64_t)(tb ? K : N) * ldb * 2;
  if (a_bytes >= (1ull << 31) || b_bytes >= (1ull << 31)) return VLM_ERR_UNSUPPORTED;
  gemm_params_t p;
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.epi = *epi;
  {  // workgroup-uniform copy of the epilogue's 8-column fast-path test (gemm_epilogue: vec8_ok), for the LDS column sums
    const bool v4 = ((ldc & 3) == 0) && (!epi->aux || (epi->ld_aux & 3) == 0) && (!epi->residual || (epi->ld_res & 3) == 0);
    const bool v8 = v4 && (c_is_f32 || (ldc & 7) == 0) && (!epi->aux || ((epi->ld_aux & 7) == 0 && ((uintptr_t)epi->aux & 15) == 0)) &&
                    (!epi->residual || ((uintptr_t)epi->residual & 15) == 0) && (!epi->bias || ((uintptr_t)epi->bias & 15) == 0) &&
                    (!epi->col_scale || ((uintptr_t)epi->col_scale & 15) == 0);
    p.epi.reserved = v8 ? 1 : 0;
    if (epi->col_sum_ws && !v8) return VLM_ERR_ARG;  // workspace mode exists only on the 16-B epilogue path
  }
  p.tiles_m = (M + GEMM_BM - 1) / GEMM_BM;
  p.tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
  p.splits = 1;
  p.ksteps_per_split = 0;
  p.n_groups = 0;
  // 0 = by shape.  Fabric-side fetch per launch at M = 54 296 (rocprofv3 FETCH_SIZE x 2, tools/pmc_gemm.py), group height
  // 1 / 4 / 8 / 16 / 32:  qkv fwd (operands 87 MB) 523 / 485 / 346 / 526 / 906 MB;  fc1 fwd 879 / 623 / 431 / 663 / 1277;
  // fc2 dgrad 1462 / 626 / 460 / 676 / 1143;  fc2 fwd (N = 768, K = 3072, operands 338 MB) 534 / 620 / 851 / 1023 / 1323.
  // Times differ by < 3 %, so the choice follows the traffic: 8 for wide outputs, row-major for N = 768.
  static const int group_m = [] {
    const char* e = getenv("VLM_GEMM_GROUP_M");
    int v = e ? atoi(e) : 0;
    if (v < 0) v = 0;
    return v;
  }();
  p.group_m = group_m ? group_m : (p.tiles_n >= 12 ? 8 : 1);
  GEMM_STAMP_ARM(p)
  hipStream_t s = (hipStream_t)stream;
  // split-K: pure accumulation into fp32 (wgrad), few output tiles, long reduction
  const int ntile = p.tiles_m * p.tiles_n, nk = (K + GEMM_BK - 1) / GEMM_BK;
  const bool plain_acc = epi->accumulate && c_is_f32 && !epi->bias && !epi->col_scale && !epi->row_scale &&
                         !epi->residual && !epi->aux && !epi->col_sum && epi->act == VLM_ACT_NONE;
  // VLM_GEMM_BIGT=0: wgrad stays on the 128x128 atomic split-K kernel (A/B runs)
  static const int bigt = [] {
    const char* e = getenv("VLM_GEMM_BIGT");
    int v = e ? atoi(e) : 1;
    return v;
  }();
  if (bigt && gemm_big_mode() > 0 && ta && tb && c_is_f32 && !epi->bias && !epi->col_scale && !epi->row_scale && !epi->residual &&
      !epi->aux && !epi->col_sum && epi->act == VLM_ACT_NONE && K >= 2048) {
    const int rc = launch_gemm_bigT(p, epi, s);
    if (rc <= 0) return rc;
  }
  // (ta = 0: a dgrad whose reduction is long and whose output is small -- the MLM decoder's, [880 x 30 522] . [30 522 x 768]: 42
  // tiles on 256 CUs took 475 us)
  if (gemm_splitk_enabled() && tb && plain_acc && ntile < 512 && nk >= 32) {
    int cus = vlm_device_cus();
    if (cus <= 0) cus = 256;
    // Slices so that the launch is ONE round of the 2 x CUs resident workgroups, never a little more: measured at
    // K = 13 574 / 54 296 (tools/bench_gemm.py): 432 workgroups 97 / 332 us, 576 (1.125 rounds) 118 / 402 us, 864 113 / 337 us
    // -- fewer slices also mean fewer fp32 atomics (25 us of a 113-us launch at K = 13 574).
    static const int slots_override = [] {
    const char* e = getenv("VLM_GEMM_SPLITK_SLOTS");
    int v = e ? atoi(e) : 0;
    return v;
  }();
    const int slots = slots_override > 0 ? slots_override : 2 * cus;
    int splits = slots / ntile;
    if (splits < 1) splits = 1;
    if (splits > nk / 8) splits = nk / 8;                // keep >= 8 K-steps per slice
    if (splits > 1) {
      p.ksteps_per_split = (nk + splits - 1) / splits;
      p.splits = (nk + p.ksteps_per_split - 1) / p.ksteps_per_split;
      if (!group_m) p.group_m = p.tiles_n >= 12 ? 1 : 4;  // split-K (timed): co-resident blocks already share tiles across K slices
      if (ta) return launch_gemm<true, true, true, false, false, true>(p, s);
      return launch_gemm<false, true, true, false, false, true>(p, s);
    }
  }
  if (allow_big && !ta && !tb && (K % (4 * BIG_BK)) == 0 && gemm_big_mode() > 0) {
    const long big_tiles = (long)((M + BIG_BM - 1) / BIG_BM) * ((N + BIG_BN - 1) / BIG_BN);
    int cus = vlm_device_cus();
    if (cus <= 0) cus = 256;
    // the kernel has only the looped epilogue without a column bound: whole 256-column tiles in N, 16-B epilogue vectors
    const bool off32 = (uint64_t)M * ldc * 4 < (1ull << 31) && (!epi->residual || (uint64_t)M * epi->ld_res * 4 < (1ull << 31)) &&
                       (!epi->aux || (uint64_t)M * epi->ld_aux * 2 < (1ull << 31));  // the epilogue's buffer descriptors
    const bool ws_ok = (N % BIG_BN) == 0 && p.epi.reserved == 1 && off32 && !epi->accumulate;  // no column bound in the epilogue
    // measured (tools/bench_gemm.py, M = 13 574 and 54 296): ahead of the 128x128 kernel from half a round of tiles up
    if (ws_ok && (gemm_big_mode() >= 2 || 2 * big_tiles >= cus)) {
      // Tail split (OFF by default; VLM_GEMM_TAIL_SPLIT=1 or big-tile mode 3): one workgroup per CU means whole ROUNDS of
      // `cus` tiles; 639 tiles (M = 54 296, N = 768) are 2.5 rounds and the last half round idles half the chip for a full
      // tile time.  With the split the whole rounds run on this kernel and the remaining ROWS go to the 128x128 kernel (two
      // workgroups per CU, shorter tiles: 10 776 rows x 768 = 510 tiles = one round of it).  Standalone, plain bf16 outputs:
      // proj forward 82.7 -> 76.3 us, fc2 forward 269.5 -> 245.6 us (-8 / -9 %).  In the training step, where these calls
      // carry the fp32 residual-stream epilogue, the 128x128 part costs 41 us per call: kernel time -0.4 %, step rate
      // -0.4 % (522 more launches per 6 steps) in two A/B pairs on one box -- not adopted.
      static const int tail_split_env = [] {
        const char* e = getenv("VLM_GEMM_TAIL_SPLIT");
        return e ? atoi(e) : 0;
      }();
      const bool plain_out = !epi->residual && !c_is_f32;  // VLM_GEMM_TAIL_SPLIT=2: only calls without the fp32 residual epilogue
      const bool tail_split = ((tail_split_env == 1 || (tail_split_env == 2 && plain_out)) && gemm_big_mode() == 1) || gemm_big_mode() == 3;
      const long full = big_tiles / cus, rem = big_tiles - full * cus, tn = (N + BIG_BN - 1) / BIG_BN;
      const long rows_big = (full * cus / tn) * BIG_BM;
      if (tail_split && full >= 1 && rem > 0 && rem * 10 <= (long)cus * 6 && !epi->col_sum &&
          rows_big > 0 && rows_big < M) {
        gemm_params_t p1 = p;
        p1.M = (int)rows_big;
        const int rc = launch_gemm_big_variant<>(p1, c_is_f32 != 0, s);
        if (rc <= 0) {
          if (rc < 0) return rc;
          vlm_epilogue_t e2 = *epi;
          const size_t r0 = (size_t)rows_big;
          if (e2.residual) e2.residual += r0 * e2.ld_res;
          if (e2.aux) e2.aux = reinterpret_cast<unsigned char*>(e2.aux) + r0 * e2.ld_aux * 2;
          if (e2.row_scale) e2.row_scale += r0;
          return gemm_dispatch(0, 0, M - (int)rows_big, N, K, reinterpret_cast<const unsigned char*>(A) + r0 * lda * 2, lda, B, ldb,
                               reinterpret_cast<unsigned char*>(C) + r0 * ldc * (c_is_f32 ? 4 : 2), ldc, c_is_f32, &e2, stream,
                               /*allow_big=*/false);
        }
      }
      const int rc = launch_gemm_big_variant<>(p, c_is_f32 != 0, s);
      if (rc <= 0) return rc;
    }
  }
#ifdef GEMM_ONLY_BIG  // tools/scratch/gemm_bench.hip: skip the other instantiations (compile time)
  return VLM_ERR_UNSUPPORTED;
#endif
  const int key = (ta ? 4 : 0) | (tb ? 2 : 0) | (c_is_f32 ? 1 : 0);
  switch (key) {
    case 0: return dispatch_stage<false, false, false>(p, s);
    case 1: return dispatch_stage<false, false, true>(p, s);
    case 2: return dispatch_stage<false, true, false>(p, s);
    case 3: return dispatch_stage<false, true, true>(p, s);
    case 4: return dispatch_stage<true, false, false>(p, s);
    case 5: return dispatch_stage<true, false, true>(p, s);
    case 6: return dispatch_stage<true, true, false>(p, s);
    default: return dispatch_stage<true, true, true>(p, s);
  }
}

// Grouped call: the row ranges of `groups` (ascending, disjoint) of ONE activation matrix go through different weights
// (the modality experts of an all_moe block, vision_transformer.py:607-681) in one launch of the 256x256 kernel; shapes the
// kernel does not serve (ragged N, tiny launches, other epilogue variants) run as one plain call per group on the same stream.
extern "C" int vlm_gemm_bf16_grouped(int n_groups, const vlm_gemm_group_t* groups, int N, int K, const void* A, int lda, void* C,
                                     int ldc, int c_is_f32, const vlm_epilogue_t* epi, void* stream) {
  if (n_groups < 1 || n_groups > VLM_GEMM_MAX_GROUPS || !groups || N < 0 || K < 0 || !C || !epi) return VLM_ERR_ARG;
  if (epi->bias || epi->col_sum || epi->col_sum_ws || epi->splitk_ws) return VLM_ERR_ARG;  // per group, in vlm_gemm_group_t
  int prev_end = 0;
  for (int g = 0; g < n_groups; ++g) {
    const vlm_gemm_group_t& G = groups[g];
    if (G.row0 < prev_end || G.rows < 0 || !G.B || (G.col_sum_ws && !G.col_sum)) return VLM_ERR_ARG;
    prev_end = G.row0 + G.rows;
  }
  const int M_total = prev_end;
  if (M_total == 0 || N == 0) return VLM_OK;
  if (!A) return VLM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bool big = gemm_big_mode() > 0 && (K % (4 * BIG_BK)) == 0 && (N % BIG_BN) == 0 && !epi->accumulate && (lda & 7) == 0 &&
             ((uintptr_t)A & 15) == 0 && ((uintptr_t)C & 15) == 0;
  {  // the looped epilogue's 16-B vectors and 32-bit descriptor offsets (gemm_dispatch: v8, off32)
    const bool v4 = ((ldc & 3) == 0) && (!epi->aux || (epi->ld_aux & 3) == 0) && (!epi->residual || (epi->ld_res & 3) == 0);
    const bool v8 = v4 && (c_is_f32 || (ldc & 7) == 0) && (!epi->aux || ((epi->ld_aux & 7) == 0 && ((uintptr_t)epi->aux & 15) == 0)) &&
                    (!epi->residual || ((uintptr_t)epi->residual & 15) == 0) && (!epi->col_scale || ((uintptr_t)epi->col_scale & 15) == 0);
    const bool off32 = (uint64_t)M_total * ldc * 4 < (1ull << 31) && (!epi->residual || (uint64_t)M_total * epi->ld_res * 4 < (1ull << 31)) &&
                       (!epi->aux || (uint64_t)M_total * epi->ld_aux * 2 < (1ull << 31)) && (uint64_t)M_total * lda * 2 < (1ull << 31);
    big = big && v8 && off32;
  }
  gemm_params_t p;
  p.A = A; p.B = nullptr; p.C = C;
  p.M = M_total; p.N = N; p.K = K;
  p.lda = lda; p.ldb = 0; p.ldc = ldc;
  p.epi = *epi;
  p.epi.reserved = 1;
  p.splits = 1;
  p.ksteps_per_split = 0;
  p.n_groups = 0;
  int tiles = 0;
  for (int g = 0; g < n_groups && big; ++g) {
    const vlm_gemm_group_t& G = groups[g];
    if (G.rows == 0) continue;
    if ((G.ldb & 7) || ((uintptr_t)G.B & 15) || (uint64_t)N * G.ldb * 2 >= (1ull << 31) || (G.bias && ((uintptr_t)G.bias & 15))) big = false;
    gemm_group_t& D = p.grp[p.n_groups++];
    D.B = G.B; D.bias = G.bias; D.col_sum = G.col_sum; D.col_sum_ws = G.col_sum_ws;
    D.ldb = G.ldb; D.row0 = G.row0; D.row_end = G.row0 + G.rows; D.tile0 = tiles;
    tiles += (G.rows + BIG_BM - 1) / BIG_BM;
  }
  if (big && p.n_groups >= 1) {
    int cus = vlm_device_cus();
    if (cus <= 0) cus = 256;
    p.tiles_m = tiles;
    if (gemm_big_mode() >= 2 || 2l * tiles * (N / BIG_BN) >= cus) {
      const int rc = launch_gemm_big_variant<true>(p, c_is_f32 != 0, s);
      if (rc <= 0) return rc;
    }
  }
  for (int g = 0; g < n_groups; ++g) {  // not offered: one plain call per group
    const vlm_gemm_group_t& G = groups[g];
    if (G.rows == 0) continue;
    vlm_epilogue_t e = *epi;
    const size_t r0 = (size_t)G.row0;
    e.bias = G.bias; e.col_sum = G.col_sum; e.col_sum_ws = G.col_sum_ws;
    if (e.residual) e.residual += r0 * e.ld_res;
    if (e.aux) e.aux = reinterpret_cast<unsigned char*>(e.aux) + r0 * e.ld_aux * 2;
    if (e.row_scale) e.row_scale += r0;
    const int rc = gemm_dispatch(0, 0, G.rows, N, K, reinterpret_cast<const unsigned char*>(A) + r0 * lda * 2, lda, G.B, G.ldb,
                                 reinterpret_cast<unsigned char*>(C) + r0 * ldc * (c_is_f32 ? 4 : 2), ldc, c_is_f32, &e, stream, true);
    if (rc) return rc;
  }
  return VLM_OK;
}

// Grouped wgrad: dW_g[M,N] (+)= A[rows_g]^T B[rows_g] for every group of token rows in ONE launch of the 256x256 wgrad kernel
// plus one reduce launch (the experts of an all_moe block, vision_transformer.py:607-681: the text expert's 3 520 tokens are a
// tenth of a round on their own).  The workgroups of one round are dealt to the groups by their share of the reduction.
extern "C" int vlm_gemm_wgrad_grouped(int n_groups, const vlm_wgrad_group_t* groups, int M, int N, const void* A, int lda,
                                      const void* B, int ldb, int ldc, float* splitk_ws, uint64_t splitk_ws_bytes, void* stream) {
  if (n_groups < 1 || n_groups > VLM_GEMM_MAX_GROUPS || !groups || M < 0 || N < 0) return VLM_ERR_ARG;
  int prev_end = 0;
  for (int g = 0; g < n_groups; ++g) {
    if (groups[g].row0 < prev_end || groups[g].rows < 0 || !groups[g].C) return VLM_ERR_ARG;
    prev_end = groups[g].row0 + groups[g].rows;
  }
  if (M == 0 || N == 0) return VLM_OK;
  if (!A || !B || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return VLM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  for (int g = 0; g < n_groups; ++g)  // an empty reduction: C_g = 0 unless accumulating
    if (groups[g].rows == 0 && !groups[g].accumulate &&
        hipMemset2DAsync(groups[g].C, (size_t)ldc * 4, 0, (size_t)N * 4, M, s) != hipSuccess)
      return VLM_ERR_LAUNCH;
  const int K_total = prev_end;
  bool big = gemm_big_mode() > 0 && splitk_ws && (N % BIG_BN) == 0 && (ldc % 4) == 0 && (uint64_t)M * N * 4 < (1ull << 31) &&
             (uint64_t)K_total * lda * 2 < (1ull << 31) && (uint64_t)K_total * ldb * 2 < (1ull << 31);
  gemm_params_t p;
  p.A = A; p.B = B; p.C = splitk_ws;
  p.M = M; p.N = N; p.K = K_total;
  p.lda = lda; p.ldb = ldb; p.ldc = N;
  p.epi = vlm_epilogue_t{};
  p.epi.alpha = 1.0f;
  p.epi.reserved = 1;
  p.tiles_m = (M + BIG_BM - 1) / BIG_BM;
  p.tiles_n = (N + BIG_BN - 1) / BIG_BN;
  p.group_m = 1;
  p.splits = 1;
  p.ksteps_per_split = 0;
  p.n_groups = 0;
  splitk_groups_t rg{};
  if (big) {
    const int ntile = p.tiles_m * p.tiles_n;
    int cus = vlm_device_cus();
    if (cus <= 0) cus = 256;
    long total_steps = 0;
    int live = 0;
    for (int g = 0; g < n_groups; ++g)
      if (groups[g].rows > 0) { total_steps += (groups[g].rows + BIG_BK - 1) / BIG_BK; ++live; }
    if (live == 0) return VLM_OK;
    if ((long)live * ntile > cus) big = false;  // more tiles than one round holds: the plain path's own split rules apply
    // steps per workgroup so that the launch is one round: the smallest target whose slice count fits
    long target = (total_steps * ntile + cus - 1) / cus;
    if (target < 16) target = 16;
    for (; big; target += 2) {
      long items = 0;
      for (int g = 0; g < n_groups; ++g)
        if (groups[g].rows > 0) items += (((groups[g].rows + BIG_BK - 1) / BIG_BK + target - 1) / target) * ntile;
      if (items <= cus) break;
    }
    int item0 = 0, slice0 = 0;
    for (int g = 0; g < n_groups && big; ++g) {
      const vlm_wgrad_group_t& G = groups[g];
      if (G.rows == 0) continue;
      if ((uintptr_t)G.C & 15) { big = false; break; }
      const int nk = (G.rows + BIG_BK - 1) / BIG_BK;
      int splits = (int)((nk + target - 1) / target);
      int kps = (nk + splits - 1) / splits;
      kps += kps & 1;
      if (kps < 4) kps = 4;
      splits = (nk + kps - 1) / kps;
      const int i = p.n_groups++;
      p.grpT[i] = gemmT_group_t{G.row0, G.row0 + G.rows, item0, kps, slice0};
      rg.C[i] = G.C; rg.slice0[i] = slice0; rg.splits[i] = splits; rg.accumulate[i] = G.accumulate;
      item0 += splits * ntile;
      slice0 += splits;
    }
    if (big && (size_t)slice0 * M * N * 4 > (size_t)splitk_ws_bytes) big = false;
    if (big) {
      hipLaunchKernelGGL(vlm_gemm_bigT_kernel<true>, dim3(item0), dim3(GEMM_THREADS), 0, s, p);
      VLM_CHECK_LAUNCH();
      const size_t per = (size_t)M * N / 4;
      hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3((unsigned)((per + 255) / 256), p.n_groups), dim3(256), 0, s,
                         reinterpret_cast<const float*>(splitk_ws), rg, M, N, ldc);
      VLM_CHECK_LAUNCH();
      return VLM_OK;
    }
  }
  for (int g = 0; g < n_groups; ++g) {  // not offered: one plain wgrad call per group
    const vlm_wgrad_group_t& G = groups[g];
    if (G.rows == 0) continue;
    vlm_epilogue_t e{};
    e.alpha = 1.0f;
    e.accumulate = G.accumulate;
    e.splitk_ws = splitk_ws;
    e.splitk_ws_bytes = splitk_ws_bytes;
    const size_t r0 = (size_t)G.row0;
    const int rc = gemm_dispatch(1, 1, M, N, G.rows, reinterpret_cast<const unsigned char*>(A) + r0 * lda * 2, lda,
                                 reinterpret_cast<const unsigned char*>(B) + r0 * ldb * 2, ldb, G.C, ldc, 1, &e, stream, true);
    if (rc) return rc;
  }
  return VLM_OK;
}

