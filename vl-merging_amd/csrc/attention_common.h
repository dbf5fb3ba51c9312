// Shared pieces of the fused attention kernels (forward, backward dK/dV/dBias, backward dQ).
#pragma once
#include "vlm_common.h"

#define ATT_BQ 128
#define ATT_BK 64
#define ATT_THREADS 256
#define ATT_TILE_BYTES (64 * 128)  // 64 rows x 64 bf16
#define ATT_LOG2E 1.4426950408889634f
#define ATT_LN2 0.6931471805599453f

// raw v_exp_f32 (2^x): libm's exp2f adds a denormal-range select/scale (5 instructions per call); scores that far
// below the row maximum contribute 0 either way
#define att_exp2(x) __builtin_amdgcn_exp2f(x)
// bias-table gather: the relative-position index is stored PRE-MULTIPLIED by 4 (byte offset into the LDS column)
__device__ __forceinline__ float att_tab(const float* tab, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(tab) + byte_off);
}

typedef __attribute__((address_space(3))) s16x4 att_lds_s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// Bias of 4 consecutive positions from one 8-byte word pair: BIAS == 1 gathers the LDS table column through the int16
// byte offsets, BIAS == 2 unpacks four fp16 values of the dense bias matrix (already scaled by log2 e).
template <int BIAS>
__device__ __forceinline__ void att_bias4(const float* tab, u32x2 w, float (&bv)[4]) {
  if (BIAS == 2) {
    const f16x4 hv = __builtin_bit_cast(f16x4, w);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (float)hv[e];
  } else {
    bv[0] = att_tab(tab, w[0] & 0xffff);
    bv[1] = att_tab(tab, w[0] >> 16);
    bv[2] = att_tab(tab, w[1] & 0xffff);
    bv[3] = att_tab(tab, w[1] >> 16);
  }
}

struct attn_seq_t {
  int B, n0, n1, base0, base1, pos1;
};

struct attn_params_t {
  const bf16_t* qkv;
  int ld_qkv, H;
  bf16_t* out;
  int ld_out;
  float* lse;
  int total_rows;
  const float* bias_t;
  int R, head_row0;
  const int16_t* idx;    // [q][k]
  int ld_idx, idx_rows;
  const int16_t* idx_t;  // [k][q]
  int ld_idx_t, idx_t_rows;
  const _Float16* dense;    // [n_cols][idx_rows][ld_idx] log2e * bias, or NULL
  const _Float16* dense_t;  // [n_cols][idx_t_rows][ld_idx_t]
  const uint8_t* keep0;
  const uint8_t* keep1;
  attn_seq_t seq;
  int mode;
  float scale;
};

static inline int att_fill_params(const vlm_attn_desc_t* d, attn_params_t& p) {
  if (!d || !d->qkv || d->H <= 0 || d->B < 0 || d->n0 < 0 || d->n1 < 0) return VLM_ERR_ARG;
  if ((d->ld_qkv & 7) || ((uintptr_t)d->qkv & 15)) return VLM_ERR_ARG;
  if (d->bias_t && (!d->rel_index || d->R <= 0 || (d->ld_index & 3) || (d->pos1 & 3) || ((uintptr_t)d->rel_index & 7)))
    return VLM_ERR_ARG;
  if (d->mode != VLM_ATTN_JOINT && d->mode != VLM_ATTN_SEPARATE) return VLM_ERR_ARG;
  if (d->R > 8191) return VLM_ERR_UNSUPPORTED;  // int16 byte offsets (4*index), LDS-resident bias column
  p.qkv = reinterpret_cast<const bf16_t*>(d->qkv);
  p.ld_qkv = d->ld_qkv;
  p.H = d->H;
  p.out = nullptr;
  p.ld_out = 0;
  p.lse = nullptr;
  p.total_rows = d->total_rows;
  p.bias_t = d->bias_t;
  p.R = d->bias_t ? d->R : 0;
  p.head_row0 = d->head_row0;
  p.idx = d->rel_index;
  p.ld_idx = d->ld_index;
  p.idx_rows = d->index_rows;
  p.idx_t = d->rel_index_t;
  p.ld_idx_t = d->ld_index_t;
  p.idx_t_rows = d->index_t_rows;
  p.dense = d->bias_t ? reinterpret_cast<const _Float16*>(d->bias_dense) : nullptr;
  p.dense_t = d->bias_t ? reinterpret_cast<const _Float16*>(d->bias_dense_t) : nullptr;
  if ((p.dense && ((uintptr_t)p.dense & 7)) || (p.dense_t && ((uintptr_t)p.dense_t & 7))) return VLM_ERR_ARG;
  p.keep0 = d->keep0;
  p.keep1 = d->keep1;
  p.seq = (attn_seq_t){d->B, d->n0, d->n1, d->base0, d->base1, d->pos1};
  p.mode = d->mode;
  p.scale = d->scale;
  return VLM_OK;
}

// Bias byte-offsets (4 x relative-position index) of this lane's 32 (row, col) pairs of one 64-wide tile: 8 loads of
// 8 B (4 consecutive columns each) from row `row` of `mat`, columns col0 + 32*kb + 8*g4 + 4*hh.  They are issued one
// tile AHEAD (right after the previous tile's scores are formed, into the same 16 registers) so that their L2
// round trip hides behind the softmax / P.V work instead of stalling the start of every tile.
__device__ __forceinline__ void att_idx_tile(__amdgpu_buffer_rsrc_t mat, uint32_t row_off, uint32_t col0, int hh,
                                             u32x2 (&iw)[8]) {
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      iw[kb * 4 + g4] = __builtin_amdgcn_raw_buffer_load_b64(mat, (row_off + col0 + kb * 32 + 8 * g4 + 4 * hh) * 2, 0, 0);
}

// Four byte-offsets for this lane's FAST position and the slow positions slow0..slow0+3 of a matrix stored
// [slow][fast] (coalesced 16-bit loads; 4*hh*ld is folded into voff by the caller).
__device__ __forceinline__ void att_idx4(__amdgpu_buffer_rsrc_t mat, uint32_t voff, uint32_t slow0, uint32_t ld2,
                                         uint32_t (&o)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(mat, voff, (slow0 + e) * ld2, 0);
}

// ---- key / query ranges of one sample ---------------------------------------------------------------------------
struct att_ranges_t {
  int n[2];        // tokens in the range
  int rowbase[2];  // first activation row
  int pos[2];      // first position in relative-index coordinates
  const uint8_t* keep[2];
  int nt[2];       // 64-row tiles
};

// Ranges the rows of segment `seg` interact with: SEPARATE -> their own segment only (block-diagonal attention,
// vision_transformer.py:567-584); JOINT -> text then image.
__device__ __forceinline__ att_ranges_t att_key_ranges(const attn_seq_t& sq, int mode, int seg, int b,
                                                       const uint8_t* keep0, const uint8_t* keep1) {
  att_ranges_t kr;
  const int n[2] = {sq.n0, sq.n1};
  const int rb[2] = {sq.base0 + b * sq.n0, sq.base1 + b * sq.n1};
  const int ps[2] = {0, sq.pos1};
  const uint8_t* kp[2] = {keep0 ? keep0 + (size_t)b * sq.n0 : nullptr, keep1 ? keep1 + (size_t)b * sq.n1 : nullptr};
  if (mode == VLM_ATTN_SEPARATE) {
    kr.n[0] = n[seg]; kr.rowbase[0] = rb[seg]; kr.pos[0] = ps[seg]; kr.keep[0] = kp[seg];
    kr.n[1] = 0; kr.rowbase[1] = 0; kr.pos[1] = 0; kr.keep[1] = nullptr;
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) { kr.n[i] = n[i]; kr.rowbase[i] = rb[i]; kr.pos[i] = ps[i]; kr.keep[i] = kp[i]; }
  }
  kr.nt[0] = (kr.n[0] + 63) >> 6;
  kr.nt[1] = (kr.n[1] + 63) >> 6;
  return kr;
}

__device__ __forceinline__ void att_tile_origin(const att_ranges_t& kr, int t, int& rng, int& k0) {
  rng = t >= kr.nt[0] ? 1 : 0;
  k0 = (rng ? t - kr.nt[0] : t) << 6;
}

// ---- [64 rows][64 bf16] tiles: global -> VGPR -> LDS ---------------------------------------------------------------
// piece = tid + 256u : row = piece >> 3, 16-B chunk = piece & 7.  Rows >= n are zero-filled.
__device__ __forceinline__ void att_tile_load(u32x4 (&reg)[2], const bf16_t* base, int ld, int col, int rowbase, int k0,
                                              int n, int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int piece = tid + 256 * u, row = piece >> 3, chunk = piece & 7;
    const int k = k0 + row;
    if (k < n) {
      reg[u] = *reinterpret_cast<const u32x4*>(base + (size_t)(rowbase + k) * ld + col + chunk * 8);
    } else {
      reg[u] = (u32x4){0u, 0u, 0u, 0u};
    }
  }
}
// row-read image: 16-B chunk XOR (row & 7)  -> conflict-free ds_read_b128 of MFMA row fragments
__device__ __forceinline__ void att_tile_store_rows(const u32x4 (&reg)[2], unsigned char* lds, int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int piece = tid + 256 * u, row = piece >> 3, chunk = piece & 7;
    *reinterpret_cast<u32x4*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4)) = reg[u];
  }
}
// transposed-read image: 32-B chunk XOR ((row>>1)&1)<<1 -> conflict-free ds_read_b64_tr_b16
__device__ __forceinline__ void att_tile_store_tr(const u32x4 (&reg)[2], unsigned char* lds, int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int piece = tid + 256 * u, row = piece >> 3, chunk = piece & 7;
    const int c32 = (chunk >> 1) ^ (((row >> 1) & 1) << 1);
    *reinterpret_cast<u32x4*>(lds + row * 128 + c32 * 32 + (chunk & 1) * 16) = reg[u];
  }
}

// MFMA 32x32x16 row fragment: lane (r = lane&31, hh = lane>>5) reads tile[row][16-B chunk]
__device__ __forceinline__ bf16x8 att_k_rowfrag(const unsigned char* lds, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// MFMA 32x32x16 A/B fragment of the TRANSPOSED tile for a product whose k-slots are the rows of a 32x32
// accumulator (element j of lane-half hh <-> tile row row0 + 8*(j>>2) + 4*hh + (j&3)); the lane's own index
// (lane & 31) selects the column colblk*32 + (lane&31).
__device__ __forceinline__ bf16x8 att_tr_frag(const unsigned char* lds, int row0, int colblk, int lane) {
  const int hh = lane >> 5, g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  const int row = row0 + 4 * hh + qq;
  const int c32 = (colblk * 2 + g16) ^ (((qq >> 1) & 1) << 1);
  const unsigned char* a = lds + row * 128 + c32 * 32 + 8 * pp;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4*)(a + 8 * 128));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- forward / dQ staging: one K tile + one V tile + the additive key mask ----------------------------------------
struct att_stage_t {
  u32x4 k[2], v[2];
  float mask;
};

__device__ __forceinline__ void att_stage_load(att_stage_t& st, const bf16_t* qkv, int ld, int D, int h,
                                               const att_ranges_t& kr, int t, int tid) {
  int rng, k0;
  att_tile_origin(kr, t, rng, k0);
  att_tile_load(st.k, qkv, ld, D + h * 64, kr.rowbase[rng], k0, kr.n[rng], tid);
  att_tile_load(st.v, qkv, ld, 2 * D + h * 64, kr.rowbase[rng], k0, kr.n[rng], tid);
  st.mask = 0.f;
  if (tid < 64) {
    const int k = k0 + tid;
    const bool ok = k < kr.n[rng] && (!kr.keep[rng] || kr.keep[rng][k] != 0);
    st.mask = ok ? 0.f : -INFINITY;
  }
}

__device__ __forceinline__ void att_stage_store(const att_stage_t& st, unsigned char* ldsK, unsigned char* ldsV,
                                                float* kmask, int tid) {
  att_tile_store_rows(st.k, ldsK, tid);
  att_tile_store_tr(st.v, ldsV, tid);
  if (tid < 64) kmask[tid] = st.mask;
}
