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
// one v_max3_f32 (fmaxf(fmaxf(a, b), c) on MFMA outputs makes hipcc insert canonicalising v_max in front)
__device__ __forceinline__ float att_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// single-instruction forms (hipcc -O3 SLP-packs adjacent f32 adds into v_pk_add_f32, slow beside MFMAs, and puts a
// canonicalising v_max in front of fmaxf on MFMA outputs)
__device__ __forceinline__ float att_add(float a, float b) {
  float d;
  asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float att_max2(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// the other 32-lane half's value of x (lane i <-> lane i ^ 32) on the vector pipe: v_permlane32_swap instead of the LDS
// crossbar (ds_bpermute + lgkmcnt wait) behind __shfl_xor(x, 32)
__device__ __forceinline__ float att_other_half(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  // v_permlane32_swap vdst, src exchanges vdst[32..63] with src[0..31]: the new vdst (r[0]) carries the lower half's values
  // in its lanes 32..63, the new src (r[1]) the upper half's values in its lanes 0..31
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
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
  const _Float16* dense;    // tiled bias / scale (fp16, att_dense_layout), stationary = query; or NULL
  const _Float16* dense_t;  // stationary = key
  int dense_tiles;          // 4-KiB tiles per (layer, head) column
  const uint8_t* keep0;
  const uint8_t* keep1;
  attn_seq_t seq;
  int mode;
  float scale;
};

static inline int att_fill_params(const vlm_attn_desc_t* d, attn_params_t& p) {
  if (!d || !d->qkv || d->H <= 0 || d->B < 0 || d->n0 < 0 || d->n1 < 0) return VLM_ERR_ARG;
  if ((d->ld_qkv & 7) || ((uintptr_t)d->qkv & 15)) return VLM_ERR_ARG;
  if (d->bias_t && (!d->rel_index || d->R <= 0 || (d->ld_index & 3) || ((uintptr_t)d->rel_index & 7)))
    return VLM_ERR_ARG;
  if ((d->pos1 & 7) || d->pos1 < d->n0) return VLM_ERR_ARG;  // image positions start at a multiple of 8 (16-B bias rows)
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
  p.dense_tiles = d->dense_tiles;
  if ((p.dense && ((uintptr_t)p.dense & 15)) || (p.dense_t && ((uintptr_t)p.dense_t & 15))) return VLM_ERR_ARG;
  p.keep0 = d->keep0;
  p.keep1 = d->keep1;
  p.seq = (attn_seq_t){d->B, d->n0, d->n1, d->base0, d->base1, d->pos1};
  p.mode = d->mode;
  p.scale = d->scale;
  return VLM_OK;
}

// attention_fwd2.hip: 1 = launched, 0 = not a call for the hand-placed kernel, < 0 = error
int att_fwd2_launch(const attn_params_t& p, hipStream_t s);

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


// =====================================================================================================================
// Round-2 kernels: everything below works in POSITION space (the coordinates of the relative-position index): text
// token t sits at position t, image token i at pos1 + i (pos1 = roundup(n0, 8)), NP = pos1 + n1 positions in all.
// The additive bias enters the score accumulators through the MATRIX pipe (v_mfma_f32_32x32x16_f16: fp16 keeps 11
// significant bits of the bias, and the selection operand's 1.0 is exact): the dense table holds bias * log2(e) in fp16
// ([q][k] and [k][q] orientations, rows and columns padded with ATT_NEG_BIG so that ragged tiles mask themselves), a
// lane loads 32 contiguous bytes of its own row per 32-position block and two MFMAs against constant selection
// fragments add them to the 32x32 accumulator -- no unpack, no per-element add on the vector pipe, which is the
// bottleneck of head-dim-64 attention (exp + max + sum already cost more issue cycles than the QK^T / PV MFMAs).
#define ATT_NEG_BIG (-30000.0f)

struct att_pos_t {
  int n0, n1, pos1, NP, base0, base1, B;
};

__device__ __forceinline__ att_pos_t att_pos(const attn_seq_t& sq) {
  att_pos_t ps;
  ps.n0 = sq.n0; ps.n1 = sq.n1; ps.pos1 = sq.pos1; ps.NP = sq.pos1 + sq.n1; ps.base0 = sq.base0; ps.base1 = sq.base1;
  ps.B = sq.B;
  return ps;
}
// activation row of position p of sample b, or -1 for a gap / out-of-range position
__device__ __forceinline__ int att_row_of(const att_pos_t& ps, int b, int p) {
  if (p < ps.n0) return ps.base0 + b * ps.n0 + p;
  if (p >= ps.pos1 && p < ps.NP) return ps.base1 + b * ps.n1 + (p - ps.pos1);
  return -1;
}
// The stationary tile of a workgroup (queries in forward / dQ, keys in dK-dV) and the position range it interacts with
struct att_span_t {
  int p0;      // first position of the workgroup's 128 stationary positions
  int s_lo, s_hi;  // streamed positions [s_lo, s_hi): all of them (JOINT) or the stationary segment's own (SEPARATE)
  int part, tile_in_part;  // dense-table part (att_dense_layout) and the 128-position tile's index inside it
};
__device__ __forceinline__ att_span_t att_span(const att_pos_t& ps, int mode, int tile) {
  att_span_t sp;
  if (mode == VLM_ATTN_SEPARATE) {
    const int nt0 = (ps.n0 + ATT_BQ - 1) / ATT_BQ;
    if (tile < nt0) { sp.p0 = tile * ATT_BQ; sp.s_lo = 0; sp.s_hi = ps.n0; sp.part = 0; sp.tile_in_part = tile; }
    else { sp.p0 = ps.pos1 + (tile - nt0) * ATT_BQ; sp.s_lo = ps.pos1; sp.s_hi = ps.NP; sp.part = 1; sp.tile_in_part = tile - nt0; }
  } else {
    sp.p0 = tile * ATT_BQ; sp.s_lo = 0; sp.s_hi = ps.NP; sp.part = 0; sp.tile_in_part = tile;
  }
  return sp;
}
// ---- tiled dense bias ------------------------------------------------------------------------------------------------
// One 4-KiB tile per (stationary block of 32 positions, streamed tile of 64 positions), stored in MFMA operand order
// [blk 0..1][j 0..1][lane 0..63][8 x fp16]: lane (c = lane & 31, hh = lane >> 5) of operand (blk, j) holds the bias of
// stationary position s0 + c against streamed positions t0 + 32*blk + 16*hh + 8*j + 0..7.  A wave's four bias loads
// of a tile are therefore four fully coalesced 1-KiB reads (per-lane row reads of a row-major table cost the CU's
// address path 32 cache lines per instruction and bound the round-1 kernels).  Entries whose stationary or streamed
// position is not a valid member of the tile's segment hold ATT_NEG_BIG: ragged tiles, the gap between text and image
// positions and (SEPARATE) the foreign segment mask themselves.
// Tile order inside a (layer, head) column: JOINT: [sb][st] over all positions from 0; SEPARATE: the text segment's
// [sb][st] (origin 0) followed by the image segment's (origin pos1).
struct att_dense_layout_t {
  int nsb[2], nst[2];  // stationary blocks / streamed tiles of part 0 (JOINT: everything; SEPARATE: text) and part 1
  int org[2];          // first position of each part
  int lim[2];          // one past the last valid position of each part
  int tiles;           // per column
};
static inline __host__ __device__ att_dense_layout_t att_dense_layout(int n0, int n1, int pos1, int mode) {
  att_dense_layout_t L;
  const int NP = pos1 + n1;
  if (mode == VLM_ATTN_SEPARATE) {
    L.org[0] = 0; L.lim[0] = n0; L.nsb[0] = (n0 + ATT_BQ - 1) / ATT_BQ * 4; L.nst[0] = (n0 + ATT_BK - 1) / ATT_BK;
    L.org[1] = pos1; L.lim[1] = NP; L.nsb[1] = (n1 + ATT_BQ - 1) / ATT_BQ * 4; L.nst[1] = (n1 + ATT_BK - 1) / ATT_BK;
  } else {
    L.org[0] = 0; L.lim[0] = NP; L.nsb[0] = (NP + ATT_BQ - 1) / ATT_BQ * 4; L.nst[0] = (NP + ATT_BK - 1) / ATT_BK;
    L.org[1] = 0; L.lim[1] = 0; L.nsb[1] = 0; L.nst[1] = 0;
  }
  L.tiles = L.nsb[0] * L.nst[0] + L.nsb[1] * L.nst[1];
  return L;
}
static inline __host__ __device__ int att_num_tiles(int n0, int n1, int pos1, int mode) {
  if (mode == VLM_ATTN_SEPARATE) return (n0 + ATT_BQ - 1) / ATT_BQ + (n1 + ATT_BQ - 1) / ATT_BQ;
  return (pos1 + n1 + ATT_BQ - 1) / ATT_BQ;
}

// XCD-aware work order.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own 4 MiB
// L2): id -> (xcd = id % 8, slot = id / 8) is remapped so that every XCD walks ONE contiguous range of the logical
// order (tile fastest, then sample, then head).  The 4-5 stationary tiles of a (sample, head) then share their streamed
// K/V (or Q/dO) through one L2, and an XCD needs one head's bias slice (0.8 MB) at a time instead of all twelve.
// Returns false for the few padding ids past the end.
__device__ __forceinline__ bool att_work_item(int n_tiles, int B, int H, int& tile, int& b, int& h) {
  const int total = n_tiles * B * H;
  const int per = (total + 7) >> 3;
  const int id = blockIdx.x, logical = (id & 7) * per + (id >> 3);
  if ((id >> 3) >= per || logical >= total) return false;
  tile = logical % n_tiles;
  const int rest = logical / n_tiles;
  b = rest % B;
  h = rest / B;
  return true;
}
static inline int att_grid_size(int n_tiles, int B, int H) { return ((n_tiles * B * H + 7) / 8) * 8; }

// Constant A/B fragments of the two selection MFMAs: k-slot (hh, e) of MFMA #0 stands for position 16*hh + e of the
// 32-position block, of MFMA #1 for 16*hh + 8 + e; lane (i = lane & 31) selects itself.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ void att_select_frags(int lane, f16x8& s0, f16x8& s1) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s0[e] = (_Float16)((r == 16 * hh + e) ? 1.0f : 0.0f);
    s1[e] = (_Float16)((r == 16 * hh + 8 + e) ? 1.0f : 0.0f);
  }
}
// acc[i][c] += bias(stationary position of lane c, streamed position i) for one 32-position block
__device__ __forceinline__ f32x16 att_bias_mfma(const f16x8& s0, const f16x8& s1, const u32x4 (&w)[2], f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(s0, __builtin_bit_cast(f16x8, w[0]), acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(s1, __builtin_bit_cast(f16x8, w[1]), acc, 0, 0, 0);
}

// One wave's bias operands for a 64-position streamed tile: four coalesced 1-KiB reads of the tiled table.
// voff = byte offset of (this wave's stationary block, streamed tile 0) + lane * 16; the tile index rides in the scalar offset.
struct att_bias_t {
  u32x4 w[2][2];
};
__device__ __forceinline__ void att_bias_load(att_bias_t& bw, __amdgpu_buffer_rsrc_t mat, uint32_t voff, int st) {
  const uint32_t soff = (uint32_t)st * 4096u;
  bw.w[0][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(mat, voff, soff, 0));
  bw.w[0][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(mat, voff + 1024, soff, 0));
  bw.w[1][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(mat, voff + 2048, soff, 0));
  bw.w[1][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(mat, voff + 3072, soff, 0));
}
// the two operands of ONE 32-position block of the next tile (the forward kernel re-requests a block's rows as soon as its
// selection MFMAs have consumed them)
__device__ __forceinline__ void att_bias_load_half(att_bias_t& bw, int blk, __amdgpu_buffer_rsrc_t mat, uint32_t voff, int st) {
  const uint32_t soff = (uint32_t)st * 4096u + (uint32_t)blk * 2048u;
  bw.w[blk][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(mat, voff, soff, 0));
  bw.w[blk][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(mat, voff + 1024, soff, 0));
}
// byte offset (inside a column) of stationary block `sb_in_part` of the part the workgroup's stationary tile lies in
__device__ __forceinline__ uint32_t att_bias_voff(const att_dense_layout_t& L, int part, int sb_in_part, int lane) {
  const uint32_t first = part ? (uint32_t)(L.nsb[0] * L.nst[0]) : 0u;
  const uint32_t nst = part ? (uint32_t)L.nst[1] : (uint32_t)L.nst[0];  // (no dynamic array index: that went to scratch)
  return (first + (uint32_t)sb_in_part * nst) * 4096u + lane * 16;
}

// ---- LDS-DMA staging of [64 positions][64 bf16] tiles (8 pieces of 1 KiB, 2 per wave): the swizzles of the row image
// (16-B chunk ^ (row & 7)) and of the transposed-read image (32-B chunk ^ ((row >> 1) & 1) << 1) live on the SOURCE
// side; rows of invalid positions point past the buffer, so the hardware zero-fills them.
// Per-lane byte offsets inside a tile are computed ONCE (att_dma_t); a tile made of 64 valid image positions -- all but
// the first and the last tile of a pass -- then costs one LDS-DMA instruction per piece with the tile's first row in
// the scalar offset, no vector arithmetic at all.
typedef __attribute__((address_space(3))) void att_lds_void;
struct att_dma_t {
  uint32_t row[2];  // tile row of this lane's piece u
  uint32_t off[2];  // (row * ld + swizzled chunk * 8) * 2, without the operand's column
};
template <bool TR>
__device__ __forceinline__ att_dma_t att_dma_init(uint32_t ld, int wave, int lane) {
  att_dma_t d;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const uint32_t row = (wave * 2 + u) * 8 + (lane >> 3), c16 = lane & 7;
    uint32_t chunk;
    if (!TR) chunk = c16 ^ (row & 7);
    else chunk = (((c16 >> 1) ^ (((row >> 1) & 1) << 1)) << 1) | (c16 & 1);
    d.row[u] = row;
    d.off[u] = (row * ld + chunk * 8) * 2;
  }
  return d;
}
// all 64 positions p0 .. p0+63 are image positions of sample b below p_hi: rows are consecutive
__device__ __forceinline__ bool att_tile_plain(const att_pos_t& ps, int p0, int p_hi) {
  return p0 >= ps.pos1 && p0 + ATT_BK <= p_hi && p_hi <= ps.NP;
}
__device__ __forceinline__ void att_dma_plain(__amdgpu_buffer_rsrc_t rsrc, unsigned char* tile, const att_dma_t& d,
                                              const att_pos_t& ps, int b, int p0, uint32_t ld, uint32_t col, int wave) {
  const uint32_t soff = ((uint32_t)(ps.base1 + b * ps.n1 + (p0 - ps.pos1)) * ld + col) * 2;  // wave-uniform
#pragma unroll
  for (int u = 0; u < 2; ++u)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (att_lds_void*)(tile + (wave * 2 + u) * 1024), 16, d.off[u], soff, 0, 0);
}
// any tile: per-row position -> row mapping through selects (text rows, gap, image rows, past the end)
__device__ __forceinline__ void att_dma_any(__amdgpu_buffer_rsrc_t rsrc, unsigned char* tile, const att_dma_t& d,
                                            const att_pos_t& ps, int b, int p0, int p_hi, uint32_t ld, uint32_t col, int wave) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int p = p0 + (int)d.row[u];
    const bool txt = p < ps.n0, img = p >= ps.pos1 && p < ps.NP;
    const bool ok = (txt || img) && p < p_hi;
    const int first = txt ? ps.base0 + b * ps.n0 + p0 : ps.base1 + b * ps.n1 + (p0 - ps.pos1);  // row of the tile's row 0
    const uint32_t off = ok ? ((uint32_t)first * ld + col) * 2 + d.off[u] : 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (att_lds_void*)(tile + (wave * 2 + u) * 1024), 16, off, 0, 0, 0);
  }
}

// additive mask of streamed position p (key side): 0 keep, -inf drop
__device__ __forceinline__ float att_key_mask(const att_pos_t& ps, int b, int p, int p_hi, const uint8_t* keep0,
                                              const uint8_t* keep1) {
  const bool txt = p < ps.n0, img = p >= ps.pos1 && p < ps.NP;
  bool ok = (txt || img) && p < p_hi;
  const uint8_t* kp = txt ? keep0 : keep1;
  const size_t at = txt ? (size_t)b * ps.n0 + p : (size_t)b * ps.n1 + (p - ps.pos1);
  if (ok && kp) ok = kp[at] != 0;
  return ok ? 0.f : -INFINITY;
}

// ---- register-staged [64 positions][64 bf16] tiles for the backward kernels ----------------------------------------
// Each thread moves two 16-B pieces (piece = tid + 256u: row = piece >> 3, chunk = piece & 7) global -> VGPR -> LDS;
// the loads of tile t+1 are issued before tile t's arithmetic and written to the other stage after it.  (LDS-DMA costs
// the issuing wave ~100 cycles per 1-KiB piece next to MFMAs; the backward kernels stage up to 24 pieces per tile.)
struct att_rows2_t {
  uint32_t off[2];  // byte offset of this thread's pieces inside a plain tile (row * ld + chunk * 8) * 2
};
__device__ __forceinline__ att_rows2_t att_rows2_init(uint32_t ld, int tid) {
  att_rows2_t d;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const uint32_t piece = tid + 256 * u;
    d.off[u] = ((piece >> 3) * ld + (piece & 7) * 8) * 2;
  }
  return d;
}
// rows p0 .. p0+63 of operand column `col` (bf16 elements) of sample b; invalid positions read as zero
__device__ __forceinline__ void att_rows2_load(u32x4 (&reg)[2], __amdgpu_buffer_rsrc_t rsrc, const att_rows2_t& d,
                                               const att_pos_t& ps, int b, int p0, int p_hi, uint32_t ld, uint32_t col, int tid) {
  if (att_tile_plain(ps, p0, p_hi)) {  // workgroup-uniform: consecutive image rows
    const uint32_t soff = ((uint32_t)(ps.base1 + b * ps.n1 + (p0 - ps.pos1)) * ld + col) * 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) reg[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, d.off[u], soff, 0));
  } else {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = p0 + ((tid + 256 * u) >> 3);
      const bool txt = p < ps.n0, img = p >= ps.pos1 && p < ps.NP;
      const bool ok = (txt || img) && p < p_hi;
      const int first = txt ? ps.base0 + b * ps.n0 + p0 : ps.base1 + b * ps.n1 + (p0 - ps.pos1);
      const uint32_t off = ok ? ((uint32_t)first * ld + col) * 2 + d.off[u] : 0xFFFFFFF0u;
      reg[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  }
}
