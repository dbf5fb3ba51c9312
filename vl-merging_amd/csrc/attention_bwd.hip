// Fused attention backward: recompute-based (flash style), three launches per (layer, pass):
//   1. dQ kernel  (query-stationary, same structure as the forward; lane <-> query); its prologue also computes
//        delta[h][row] = sum_d dO*O for its rows and publishes it for the kernels below
//        S^T = K Q^T -> P^T = exp2(S2 - lse2) ; dP^T = V dO^T ; dS^T = P^T o (dP^T - delta) ;
//        dQ^T[d][q] += K^T . dS^T      (A = K^T by ds_read_b64_tr_b16, B = dS^T accumulator regs as bf16)
//   2. dK/dV kernel (key-stationary; lane <-> key)
//        S = Q K^T -> P ; dP = dO V^T ; dS = P o (dP - delta) ;
//        dV[key][d] += P^T dO ,  dK[key][d] += scale * dS^T Q      (A = accumulator regs, B = dO / Q tr-read)
//   3. dBias kernel (only when the bias table needs a gradient): batch-summed dS -> LDS histogram -> global atomics
// This is what autograd derives for reference vision_transformer.py:346-358 + F.embedding in get_rel_pos_bias
// (vilt_module.py:1061-1064); the extra MFMA products (9 instead of 5) buy a deterministic dQ without atomics and a
// bias gradient whose (slow) LDS atomics are amortised over the batch.
#include "vlm_common.h"
#include "attention_common.h"
#include "vlm_diag.h"
#include <stdlib.h>

struct attn_bwd_params_t {
  attn_params_t f;        // forward description (qkv, bias, index, ranges)
  const bf16_t* d_o;      // [rows, H*64]
  int ld_do;
  const float* lse;       // [H, rows] log2 domain
  const bf16_t* o;        // [rows, H*64] forward output
  int ld_o;
  float* delta;           // [H, rows] rowsum(dO * O): written by the dQ kernel (every row belongs to one of its tiles), read by dK/dV
  float inv_c1;           // 1 / (scale log2 e)
  bf16_t* dqkv;           // [rows, 3*H*64]
  int ld_dqkv;
  float* dbias_t;         // [n_cols, R] accumulate
  float* dbias_part;      // [items][R] per-workgroup histograms (two-stage reduction) or NULL (global atomics)
  float* nstat;           // [2][H][rows]: -lse / c1 and -delta for the 16-wave bias-gradient kernel (written by the dQ kernel) or NULL
  float* dq_colsum[2];    // per segment (0 text rows, 1 image rows): [H*64] += column sums of dQ (q_bias grad) or NULL
  float* dv_colsum[2];    // same for dV (v_bias gradient)
};

#include "attention_bwd_dkvb.h"
#include "attention_bwd_dq2.h"

// ----------------------------------------------------------------------------------------------------- dQ kernel
// Query-stationary: workgroup = 128 query positions of one (sample, head), lane <-> query, streams 64-key tiles.
// Per 32-key block (exponent units, see attention_fwd.hip):
//   E^T  = -lse + Bias^T*log2e + K (c1 Q)^T        C operand = a register tuple holding -lse (constant per lane)
//   dP^T = -delta + V dO^T                         C operand = a register tuple holding -delta
//   dS^T = exp2(E^T) * dP^T                        (one v_exp + one v_mul per score; natural-score units)
//   dQ^T[d][q] += K^T . dS^T                       (A = K^T by ds_read_b64_tr_b16, B = dS^T accumulators as bf16)
// and dQ leaves scaled by `scale` once at the end.
template <bool HAS_BIAS>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dq_kernel(const attn_bwd_params_t bp) {
  const attn_params_t& p = bp.f;
  __shared__ __attribute__((aligned(16))) unsigned char smem[6 * ATT_TILE_BYTES + 512];
  unsigned char* ldsK = smem;                          // [2] row image
  unsigned char* ldsKt = smem + 2 * ATT_TILE_BYTES;    // [2] transposed-read image
  unsigned char* ldsV = smem + 4 * ATT_TILE_BYTES;     // [2] row image
  float* kmask = reinterpret_cast<float*>(smem + 6 * ATT_TILE_BYTES);  // [2][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform BY CONSTRUCTION: tell the compiler (else waterfall loops)
  const int r = lane & 31, hh = lane >> 5;
  const att_pos_t ps = att_pos(p.seq);
  int wtile, b, h;
  if (!att_work_item(att_num_tiles(ps.n0, ps.n1, ps.pos1, p.mode), ps.B, p.H, wtile, b, h)) return;
  const att_span_t sp = att_span(ps, p.mode, wtile);
  const int D = p.H * 64;
  const int qp = sp.p0 + wave * 32 + r;
  const int qrow_raw = qp < sp.s_hi ? att_row_of(ps, b, qp) : -1;
  const bool qvalid = qrow_raw >= 0;
  const size_t qrow = qvalid ? (size_t)qrow_raw : (size_t)att_row_of(ps, b, sp.s_lo);
  const int ntiles = (sp.s_hi - sp.s_lo + ATT_BK - 1) / ATT_BK;
  const float c1 = p.scale * ATT_LOG2E;

  bf16x8 qf[4], dof[4];
  {
    const bf16_t* qptr = p.qkv + qrow * p.ld_qkv + h * 64 + 8 * hh;
    const bf16_t* dptr = bp.d_o + qrow * bp.ld_do + h * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 raw = *reinterpret_cast<const bf16x8*>(qptr + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (bf16_t)((float)raw[j] * c1);
      dof[s] = *reinterpret_cast<const bf16x8*>(dptr + 16 * s);
    }
  }
  f32x16 neglse, negdel;
  {
    // delta = rowsum(dO * O) of this lane's query, from the dO fragments already in registers and the matching 32 values of O
    // (the two lane halves hold complementary 8-column groups of the head): no separate pass over O and dO, and the value
    // is published for the dK/dV and bias-gradient kernels that follow on the stream
    const bf16_t* optr = bp.o + qrow * bp.ld_o + h * 64 + 8 * hh;
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 raw = *reinterpret_cast<const bf16x8*>(optr + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) part += (float)raw[j] * (float)dof[s][j];
    }
    part += att_other_half(part);
    const size_t at = (size_t)h * p.total_rows + qrow;
    const float lse2 = qvalid ? bp.lse[at] : INFINITY;
    if (qvalid && hh == 0) {
      bp.delta[at] = part;
      if (bp.nstat) {  // C operands of attn_bwd_dbias16_kernel: -lse / c1 (its scores stay unscaled) and -delta
        bp.nstat[at] = -lse2 * bp.inv_c1;
        bp.nstat[(size_t)p.H * p.total_rows + at] = -part;
      }
    }
    const float l2 = -lse2;                  // invalid rows: -inf, P = 0
    const float dl = qvalid ? -part : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { neglse[i] = l2; negdel[i] = dl; }
  }
  f16x8 sel0, sel1;
  att_select_frags(lane, sel0, sel1);
  const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv), 0, (uint32_t)((size_t)p.total_rows * p.ld_qkv * 2), 0x00020000);
  const att_dense_layout_t dl = att_dense_layout(ps.n0, ps.n1, ps.pos1, p.mode);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(HAS_BIAS ? p.dense + (size_t)(p.head_row0 + h) * p.dense_tiles * 2048 : nullptr), 0,
      HAS_BIAS ? (uint32_t)p.dense_tiles * 4096u : 0, 0x00020000);
  const uint32_t bvoff = att_bias_voff(dl, sp.part, sp.tile_in_part * 4 + wave, lane);

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[0][i] = o[1][i] = 0.f;

  auto tile_masked = [&](int kp0) {
    return (p.keep0 != nullptr && kp0 < ps.n0) || (p.keep1 != nullptr && kp0 + ATT_BK > ps.pos1) ||
           (!HAS_BIAS && (kp0 < ps.pos1 || kp0 + ATT_BK > sp.s_hi));
  };
  // Staging: every trip issues the SAME vector-memory operations in the same order, none under a branch (4 tile pieces,
  // 2 keep bytes, 4 bias operands) so that the compiler's vmcnt waits stay counted -- see attn_bwd_dkv_kernel.  A segment
  // without a keep mask reads through a zero-length descriptor; a trip past the last tile points every row out of range.
  const __amdgpu_buffer_rsrc_t rkeep0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.keep0 ? p.keep0 : reinterpret_cast<const uint8_t*>(p.qkv)), 0, p.keep0 ? (uint32_t)(ps.B * ps.n0) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rkeep1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.keep1 ? p.keep1 : reinterpret_cast<const uint8_t*>(p.qkv)), 0, p.keep1 ? (uint32_t)(ps.B * ps.n1) : 0u, 0x00020000);
  u32x4 rk[2], rv[2];
  uint32_t rkeep[2] = {0u, 0u};
  const uint32_t chunk2 = (uint32_t)(tid & 7) * 16;
  auto load = [&](int t) {
    const int kp0 = sp.s_lo + t * ATT_BK;
    const int lim = t < ntiles ? sp.s_hi : 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pp = kp0 + ((tid + 256 * u) >> 3);
      const bool txt = pp < ps.n0, img = pp >= ps.pos1 && pp < ps.NP;
      const bool ok = (txt || img) && pp < lim;
      const uint32_t row = (uint32_t)(txt ? ps.base0 + b * ps.n0 + pp : ps.base1 + b * ps.n1 + (pp - ps.pos1));
      const uint32_t ok_ = ok ? (row * (uint32_t)p.ld_qkv + (uint32_t)(D + h * 64)) * 2 + chunk2 : 0xFFFFFFF0u;
      const uint32_t ov_ = ok ? (row * (uint32_t)p.ld_qkv + (uint32_t)(2 * D + h * 64)) * 2 + chunk2 : 0xFFFFFFF0u;
      rk[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rkv, ok_, 0, 0));
      rv[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rkv, ov_, 0, 0));
    }
    {  // keep bytes of key position kp0 + lane (text / image segment): RAW, looked at in store()
      const int pp = kp0 + lane;
      const bool txt = pp < ps.n0 && pp < lim, img = pp >= ps.pos1 && pp < ps.NP && pp < lim;
      rkeep[0] = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rkeep0, txt ? (uint32_t)(b * ps.n0 + pp) : 0xFFFFFFF0u, 0, 0);
      rkeep[1] = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rkeep1, img ? (uint32_t)(b * ps.n1 + (pp - ps.pos1)) : 0xFFFFFFF0u, 0, 0);
    }
  };
  auto store = [&](int t, int buf) {
    att_tile_store_rows(rk, ldsK + buf * ATT_TILE_BYTES, tid);
    att_tile_store_tr(rk, ldsKt + buf * ATT_TILE_BYTES, tid);
    att_tile_store_rows(rv, ldsV + buf * ATT_TILE_BYTES, tid);
    if (wave == 0) {  // additive key mask of the tile: 0 keep, -inf drop
      const int pp = sp.s_lo + t * ATT_BK + lane;
      const bool txt = pp < ps.n0, img = pp >= ps.pos1 && pp < ps.NP;
      bool ok = (txt || img) && pp < sp.s_hi;
      if (txt && p.keep0) ok = ok && rkeep[0] != 0;
      if (img && p.keep1) ok = ok && rkeep[1] != 0;
      kmask[buf * 64 + lane] = ok ? 0.f : -INFINITY;
    }
  };
  att_bias_t bw;
  if (HAS_BIAS) att_bias_load(bw, rbias, bvoff, 0);
  load(0);
  store(0, 0);
  __syncthreads();

  ATT_STAMP_DECL()
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const int kp0 = sp.s_lo + t * ATT_BK;
    ATT_STAMP(slot++);  // trip start
    load(t + 1);
    ATT_STAMP(slot++);  // next tile requested
    const unsigned char* lk = ldsK + cur * ATT_TILE_BYTES;
    const unsigned char* lkt = ldsKt + cur * ATT_TILE_BYTES;
    const unsigned char* lv = ldsV + cur * ATT_TILE_BYTES;
    const float* km = kmask + cur * 64;
    const bool masked = tile_masked(kp0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 kfr[4], vfr[4];
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        kfr[ss] = att_k_rowfrag(lk, kb * 32 + r, 2 * ss + hh);
        vfr[ss] = att_k_rowfrag(lv, kb * 32 + r, 2 * ss + hh);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 e = neglse, dp = negdel;
      if (HAS_BIAS) e = att_bias_mfma(sel0, sel1, bw.w[kb], e);
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ss], qf[ss], e, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ss], dof[ss], dp, 0, 0, 0);
      }
      // the next tile's bias rows of this block: same registers, consumed by the selection MFMAs above -- requested HERE, a block
      // (kb = 1) to a tile and a half (kb = 0) before their use (round 5: the four loads used to be issued at the end of the trip,
      // one LDS store + barrier before the first selection MFMA of the next trip: an L2 round trip exposed per 64 keys).  Past the
      // last tile the rows are out of the descriptor's range (zeros, unused).
      if (HAS_BIAS) att_bias_load_half(bw, kb, rbias, bvoff, t + 1);
      ATT_STAMP(slot++);  // score / dP chains issued
      if (masked) {  // workgroup-uniform
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 mk = *reinterpret_cast<const f32x4*>(km + kb * 32 + 8 * g4 + 4 * hh);
#pragma unroll
          for (int i = 0; i < 4; ++i) e[4 * g4 + i] += mk[i];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] = att_exp2(e[i]) * dp[i];  // dS^T (natural units, unscaled)
      ATT_STAMP(slot++);  // dS known
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 df;
#pragma unroll
        for (int j = 0; j < 8; ++j) df[j] = (bf16_t)e[8 * s2 + j];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 kf = att_tr_frag(lkt, kb * 32 + 16 * s2, db, lane);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, df, o[db], 0, 0, 0);
        }
      }
      ATT_STAMP(slot++);  // dQ products issued
    }
    store(t + 1, cur ^ 1);
    ATT_STAMP(slot++);  // next tile in LDS
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    o[0][i] *= p.scale;
    o[1][i] *= p.scale;
  }
  if (qvalid) {
    bf16_t* op = bp.dqkv + qrow * bp.ld_dqkv + h * 64 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (bf16_t)o[db][4 * g4 + i];
        *reinterpret_cast<bf16x4*>(op + db * 32 + 8 * g4) = v;
      }
  }
  if (bp.dq_colsum[0] || bp.dq_colsum[1]) {
    // q_bias gradient = column sums of dQ per token segment: the 128x64 tile goes through LDS (row stride 68 floats:
    // conflict-free 16-B writes), every thread sums 32 rows of one column, split by the row's segment (a JOINT tile
    // can hold text and image positions) -- no second pass over dqkv
    float* red = reinterpret_cast<float*>(smem);
    const int lr = wave * 32 + r;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qvalid) v = (f32x4){o[db][4 * g4], o[db][4 * g4 + 1], o[db][4 * g4 + 2], o[db][4 * g4 + 3]};
        *reinterpret_cast<f32x4*>(red + lr * 68 + db * 32 + 8 * g4 + 4 * hh) = v;
      }
    __syncthreads();
    const int col = tid & 63, part = tid >> 6;
    float sum_t = 0.f, sum_i = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const float v = red[(part * 32 + i) * 68 + col];
      if (sp.p0 + part * 32 + i < ps.n0) sum_t += v;
      else sum_i += v;
    }
    if (bp.dq_colsum[0] && sp.p0 + part * 32 < ps.n0) atomicAdd(bp.dq_colsum[0] + h * 64 + col, sum_t);
    if (bp.dq_colsum[1] && sp.p0 + part * 32 + 32 > ps.pos1) atomicAdd(bp.dq_colsum[1] + h * 64 + col, sum_i);
  }
}

// ------------------------------------------------------------------------------------------- dK / dV kernel
// Key-stationary: workgroup = 128 key positions of one (sample, head), lane <-> key, streams 64-query tiles of Q and dO.
// Per 32-query block:
//   E[q][key]  = -lse[q] + Bias*log2e + Q (c1 K)^T     C operand = -lse of the block's 16 query rows, read from LDS
//   dP[q][key] = -delta[q] + dO V^T                    C operand = -delta, read from LDS
//   P = exp2(E) ; dS = P * dP
//   dV[key][d] += P^T dO ,  dK[key][d] += dS^T Q       (A = the accumulator registers as bf16, B = transposed reads)
// dK leaves scaled by `scale` at the end.  Dropped keys (padding tokens) add -inf to their lanes' C operands.
template <bool HAS_BIAS>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dkv_kernel(const attn_bwd_params_t bp) {
  const attn_params_t& p = bp.f;
  __shared__ __attribute__((aligned(16))) unsigned char smem[8 * ATT_TILE_BYTES + 1024];
  unsigned char* ldsQ = smem;                           // [2] row image  [64 q][64 d]
  unsigned char* ldsQt = smem + 2 * ATT_TILE_BYTES;     // [2] transposed-read image
  unsigned char* ldsO = smem + 4 * ATT_TILE_BYTES;      // [2] dO row image
  unsigned char* ldsOt = smem + 6 * ATT_TILE_BYTES;     // [2] dO transposed-read image
  float* qstat = reinterpret_cast<float*>(smem + 8 * ATT_TILE_BYTES);  // [2][-lse 64 | -delta 64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform BY CONSTRUCTION: tell the compiler (else waterfall loops)
  const int r = lane & 31, hh = lane >> 5;
  const att_pos_t ps = att_pos(p.seq);
  int wtile, b, h;
  if (!att_work_item(att_num_tiles(ps.n0, ps.n1, ps.pos1, p.mode), ps.B, p.H, wtile, b, h)) return;
  const att_span_t sp = att_span(ps, p.mode, wtile);  // stationary = keys, streamed = the queries that see them
  const int D = p.H * 64;
  const int kp = sp.p0 + wave * 32 + r;
  const int krow_raw = kp < sp.s_hi ? att_row_of(ps, b, kp) : -1;
  const bool kvalid = krow_raw >= 0;
  const size_t krow = kvalid ? (size_t)krow_raw : (size_t)att_row_of(ps, b, sp.s_lo);
  const float kmaskv = kvalid ? att_key_mask(ps, b, kp, sp.s_hi, p.keep0, p.keep1) : -INFINITY;
  // does this WAVE hold a dropped key whose scores the table does not already mask?  (padding tokens; without a bias
  // table also the gap / past-the-end positions)
  const bool wave_masked = __any(kmaskv != 0.f && (kvalid || !HAS_BIAS));
  const int ntiles = (sp.s_hi - sp.s_lo + ATT_BK - 1) / ATT_BK;
  const float c1 = p.scale * ATT_LOG2E;

  bf16x8 kf[4], vf[4];
  {
    const bf16_t* kptr = p.qkv + krow * p.ld_qkv + D + h * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 raw = *reinterpret_cast<const bf16x8*>(kptr + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[s][j] = (bf16_t)(kvalid ? (float)raw[j] * c1 : 0.f);
      vf[s] = *reinterpret_cast<const bf16x8*>(kptr + D + 16 * s);
    }
  }
  f16x8 sel0, sel1;
  att_select_frags(lane, sel0, sel1);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv), 0, (uint32_t)((size_t)p.total_rows * p.ld_qkv * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(bp.d_o), 0, (uint32_t)((size_t)p.total_rows * bp.ld_do * 2), 0x00020000);
  const att_dense_layout_t dl = att_dense_layout(ps.n0, ps.n1, ps.pos1, p.mode);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(HAS_BIAS ? p.dense_t + (size_t)(p.head_row0 + h) * p.dense_tiles * 2048 : nullptr), 0,
      HAS_BIAS ? (uint32_t)p.dense_tiles * 4096u : 0, 0x00020000);
  const uint32_t bvoff = att_bias_voff(dl, sp.part, sp.tile_in_part * 4 + wave, lane);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) dk[0][i] = dk[1][i] = dv[0][i] = dv[1][i] = 0.f;

  // ---- staging of the streamed tiles ------------------------------------------------------------------------------------
  // Every trip of the tile loop issues the SAME vector-memory operations in the same order, none of them under a branch:
  // 4 tile pieces, 1 statistic, 4 bias operands.  vmcnt is one in-order counter; where a path may or may not have issued
  // a load (round 2: a plain / ragged tile branch, the statistics of waves 0-1 only, "if there is a next tile") the compiler
  // has to assume it has not, and its wait for the OLDER bias operands at the first MFMA of a tile then also drained the
  // tile loads issued a few cycles before: 2 000 of a tile's 7 400 cycles (s_memtime stamps, tools/scratch/attn_bench.hip).
  // A trip past the last tile points every row out of range (zero fill, no traffic).
  u32x4 rq_[2], ro_[2];
  float rstat = 0.f;
  bool rstat_ok = false;
  const uint32_t chunk2 = (uint32_t)(tid & 7) * 16;
  auto row_ok = [&](int pp, int lim, uint32_t& row) {
    const bool txt = pp < ps.n0, img = pp >= ps.pos1 && pp < ps.NP;
    row = (uint32_t)(txt ? ps.base0 + b * ps.n0 + pp : ps.base1 + b * ps.n1 + (pp - ps.pos1));
    return (txt || img) && pp < lim;
  };
  // (Requesting dO after the first 32-query block and writing Q to LDS there -- half the burst behind the barrier, the two
  // register sets never live together, 222 VGPRs -- measured the same: 700 / 706 vs 706 / 698 us.  So did issuing the exponentials of
  // the score chain between the MFMAs of the dP chain: the wave's dependent chain, not a pipe, sets the time.)
  auto load_q = [&](int t) {
    const int qp0 = sp.s_lo + t * ATT_BK;
    const int lim = t < ntiles ? sp.s_hi : 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint32_t row;
      const bool ok = row_ok(qp0 + ((tid + 256 * u) >> 3), lim, row);
      rq_[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rq, ok ? (row * (uint32_t)p.ld_qkv + (uint32_t)h * 64) * 2 + chunk2 : 0xFFFFFFF0u, 0, 0));
    }
    {  // waves 0 / 2: lse, waves 1 / 3: delta of the tile's 64 query positions (waves 0 and 1 publish them).  RAW values: anything
       // computed from them here would make the compiler wait for this load, and with it for the tile loads above
      const int qq = qp0 + lane;
      const int row = qq < lim ? att_row_of(ps, b, qq) : -1;
      const float* src = (wave & 1) ? bp.delta : bp.lse;  // wave-uniform (a per-lane select loaded the POINTER through memory)
      rstat = src[(size_t)h * p.total_rows + (row >= 0 ? row : 0)];
      rstat_ok = row >= 0;
    }
  };
  auto load_o = [&](int t) {
    const int qp0 = sp.s_lo + t * ATT_BK;
    const int lim = t < ntiles ? sp.s_hi : 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint32_t row;
      const bool ok = row_ok(qp0 + ((tid + 256 * u) >> 3), lim, row);
      ro_[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rdo, ok ? (row * (uint32_t)bp.ld_do + (uint32_t)h * 64) * 2 + chunk2 : 0xFFFFFFF0u, 0, 0));
    }
  };
  auto store_q = [&](int buf) {
    att_tile_store_rows(rq_, ldsQ + buf * ATT_TILE_BYTES, tid);
    att_tile_store_tr(rq_, ldsQt + buf * ATT_TILE_BYTES, tid);
    if (wave < 2) qstat[buf * 128 + tid] = rstat_ok ? -rstat : (wave == 0 ? -INFINITY : 0.f);  // absent queries: P = exp2(-inf) = 0
  };
  auto store_o = [&](int buf) {
    att_tile_store_rows(ro_, ldsO + buf * ATT_TILE_BYTES, tid);
    att_tile_store_tr(ro_, ldsOt + buf * ATT_TILE_BYTES, tid);
  };
  auto load = [&](int t) { load_q(t); load_o(t); };
  auto store = [&](int buf) { store_q(buf); store_o(buf); };
  att_bias_t bw;
  if (HAS_BIAS) att_bias_load(bw, rbias, bvoff, 0);
  load(0);
  store(0);
  __syncthreads();

  ATT_STAMP_DECL()
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    load(t + 1);
    ATT_STAMP(slot++);  // next tile's loads issued
    const unsigned char* lq = ldsQ + cur * ATT_TILE_BYTES;
    const unsigned char* lqt = ldsQt + cur * ATT_TILE_BYTES;
    const unsigned char* lo = ldsO + cur * ATT_TILE_BYTES;
    const unsigned char* lot = ldsOt + cur * ATT_TILE_BYTES;
    const float* st = qstat + cur * 128;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 e, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(st + qb * 32 + 8 * g4 + 4 * hh);
        const f32x4 c = *reinterpret_cast<const f32x4*>(st + 64 + qb * 32 + 8 * g4 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) { e[4 * g4 + i] = a[i]; dp[4 * g4 + i] = c[i]; }
      }
      if (wave_masked) {
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] += kmaskv;
      }
      if (HAS_BIAS) e = att_bias_mfma(sel0, sel1, bw.w[qb], e);
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const bf16x8 qa = att_k_rowfrag(lq, qb * 32 + r, 2 * ss + hh), oa = att_k_rowfrag(lo, qb * 32 + r, 2 * ss + hh);
        e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ss], e, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ss], dp, 0, 0, 0);
      }
      ATT_STAMP(slot++);  // score / dP chains issued
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        e[i] = att_exp2(e[i]);  // P
        dp[i] *= e[i];          // dS (natural units)
      }
      ATT_STAMP(slot++);  // exponentials issued
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf, df;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pf[j] = (bf16_t)e[8 * s2 + j];
          df[j] = (bf16_t)dp[8 * s2 + j];
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 ob = att_tr_frag(lot, qb * 32 + 16 * s2, db, lane);
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, ob, dv[db], 0, 0, 0);
          const bf16x8 qb_ = att_tr_frag(lqt, qb * 32 + 16 * s2, db, lane);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, qb_, dk[db], 0, 0, 0);
        }
      }
      ATT_STAMP(slot++);  // dV / dK products issued
    }
    if (HAS_BIAS) att_bias_load(bw, rbias, bvoff, t + 1);  // (past the last tile: unused)
    store(cur ^ 1);
    ATT_STAMP(slot++);  // next tile stored to LDS
    __syncthreads();
    ATT_STAMP(slot++);  // barrier passed
  }

  // ---- store dK, dV: accumulator rows = keys (registers), column = d (lane & 31) ---------------------------------------
  // (2-byte stores, 64-byte segments per half wave.  Through a wave-private LDS transpose and 16-byte row stores instead:
  // no difference, 714 vs 717 us for the backward pass at 88 samples -- what the stores cost, 16 us, is their traffic.)
  {
    const int kp0w = sp.p0 + wave * 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kl = (i & 3) + 8 * (i >> 2) + 4 * hh;
      const int kpi = kp0w + kl;
      const int row = kpi < sp.s_hi ? att_row_of(ps, b, kpi) : -1;
      if (row >= 0) {
        bf16_t* dst = bp.dqkv + (size_t)row * bp.ld_dqkv + D + h * 64 + r;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dst[db * 32] = (bf16_t)(dk[db][i] * p.scale);
          dst[D + db * 32] = (bf16_t)dv[db][i];
        }
      }
    }
    if (bp.dv_colsum[0] || bp.dv_colsum[1]) {  // v_bias gradient = column sums of dV over this workgroup's keys, per segment
      float* red = reinterpret_cast<float*>(smem);  // [segment][wave][64]
      __syncthreads();
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        float sum_t = 0.f, sum_i = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kpi = kp0w + (i & 3) + 8 * (i >> 2) + 4 * hh;
          const bool ok = kpi < sp.s_hi && (kpi < ps.n0 || kpi >= ps.pos1);
          if (ok && kpi < ps.n0) sum_t += dv[db][i];
          if (ok && kpi >= ps.pos1) sum_i += dv[db][i];
        }
        sum_t += __shfl_xor(sum_t, 32, 64);
        sum_i += __shfl_xor(sum_i, 32, 64);
        if (hh == 0) {
          red[wave * 64 + db * 32 + r] = sum_t;
          red[256 + wave * 64 + db * 32 + r] = sum_i;
        }
      }
      __syncthreads();
      if (tid < 128) {
        const int sg = tid >> 6, c = tid & 63;
        const float v = red[sg * 256 + c] + red[sg * 256 + 64 + c] + red[sg * 256 + 128 + c] + red[sg * 256 + 192 + c];
        if (bp.dv_colsum[sg] && v != 0.f) atomicAdd(bp.dv_colsum[sg] + h * 64 + c, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- dBias kernel
// d(bias table column)[idx[q][key]] += sum_b dS[b,h,q,key] (autograd of F.embedding in get_rel_pos_bias,
// vilt_module.py:1061-1064).  The sum over the samples is taken FIRST, in registers: a scatter of every sample's dS would
// cost B times the atomics, so one workgroup owns a (128-key x 128-query) tile of one head, walks its share of the
// samples recomputing E and dP (2 of the 5 MFMA products of the backward, with the same matrix-pipe formulation as
// the dQ / dK-dV kernels: tiled fp16 bias through selection MFMAs, -lse / -delta as C operands) and only then feeds the
// summed dS through an LDS histogram -> global atomics.
// SIXTEEN waves (four per SIMD, <= 128 VGPRs): wave (kw = wave & 3, qw = wave >> 2) owns 32 keys x 32 queries (16 accumulator
// registers), the four operand tiles of the next sample arrive by LDS-DMA (no staging registers: four 1-KiB pieces per wave and
// sample, the row-image swizzle on the source side) and nothing is converted on the vector pipe: K stays unscaled, the C operand
// of the score product is -lse / c1 (the dQ kernel writes it), the bias enters through selection MFMAs whose "one" is 1 / c1 split
// into an fp16 head and tail (products of fp16 values are exact in the fp32 accumulator), and P = exp2(c1 * e).  The histogram is
// 64-bit FIXED POINT (ds_add_u64: integer LDS atomics run at the LDS array's rate, float ones at ~130 cycles per wave
// instruction).  [Round 2's 8-wave kernel -- two waves per SIMD at 240 registers, global -> VGPR -> LDS staging, a float
// histogram: 3.2 us per sample where its MFMAs need 0.6 -- is in the git history; DESIGN.md 4.3.]
#define ATT_DB16_THREADS 1024
__global__ __launch_bounds__(ATT_DB16_THREADS) void attn_bwd_dbias16_kernel(const attn_bwd_params_t bp, int n_groups) {
  const attn_params_t& p = bp.f;
  constexpr int STAGE = 8 * ATT_TILE_BYTES + 1024;  // K, V (128 keys), Q, dO (128 queries) row images + (-lse / c1 | -delta)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kw = wave & 3, qw = wave >> 2;
  const int r = lane & 31, hh = lane >> 5;
  const att_pos_t ps = att_pos(p.seq);
  const int D = p.H * 64;
  // ---- work item: (key tile, query tile) pair x head x sample group, XCD-aware order --------------------------------------------
  const int nkt = att_num_tiles(ps.n0, ps.n1, ps.pos1, p.mode);
  int pairs = 0;
  for (int k = 0; k < nkt; ++k) {
    const att_span_t s_ = att_span(ps, p.mode, k);
    pairs += (s_.s_hi - s_.s_lo + ATT_BQ - 1) / ATT_BQ;
  }
  const int total = pairs * p.H * n_groups, per = (total + 7) >> 3;
  const int logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || logical >= total) return;
  int item = logical % pairs;
  const int pair_id = item;
  const int grp = (logical / pairs) % n_groups;
  const int h = logical / (pairs * n_groups);
  const int part_slot = (pair_id * p.H + h) * n_groups + grp;
  int kt = 0;
  att_span_t sp = att_span(ps, p.mode, 0);
  for (;; ++kt) {
    sp = att_span(ps, p.mode, kt);
    const int nq = (sp.s_hi - sp.s_lo + ATT_BQ - 1) / ATT_BQ;
    if (item < nq) break;
    item -= nq;
    if (kt + 1 >= nkt) return;
  }
  const int qp0 = sp.s_lo + item * ATT_BQ;
  const int b_lo = (int)((long)ps.B * grp / n_groups), b_hi = (int)((long)ps.B * (grp + 1) / n_groups);
  if (b_lo >= b_hi) return;

  const int kp = sp.p0 + kw * 32 + r;  // this lane's key position
  const bool kvalid = kp < sp.s_hi && (kp < ps.n0 || kp >= ps.pos1);
  const uint8_t* keepk = kp < ps.n0 ? p.keep0 : p.keep1;
  const int keep_at = kp < ps.n0 ? kp : kp - ps.pos1, keep_n = kp < ps.n0 ? ps.n0 : ps.n1;
  const bool wave_keep = __any(kvalid && keepk != nullptr);

  // selection fragments scaled by 1 / c1 = head + tail (fp16 each)
  const float c1 = p.scale * ATT_LOG2E, inv = 1.0f / c1;
  const _Float16 inv_hi = (_Float16)inv, inv_lo = (_Float16)(inv - (float)inv_hi);
  f16x8 sh0, sh1, sl0, sl1;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool m0 = r == 16 * hh + e, m1 = r == 16 * hh + 8 + e;
    sh0[e] = m0 ? inv_hi : (_Float16)0.f;
    sl0[e] = m0 ? inv_lo : (_Float16)0.f;
    sh1[e] = m1 ? inv_hi : (_Float16)0.f;
    sl1[e] = m1 ? inv_lo : (_Float16)0.f;
  }
  const att_dense_layout_t dl = att_dense_layout(ps.n0, ps.n1, ps.pos1, p.mode);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(p.dense_t + (size_t)(p.head_row0 + h) * p.dense_tiles * 2048), 0, (uint32_t)p.dense_tiles * 4096u, 0x00020000);
  att_bias_t bw;  // this wave's 32-query block (the same for every sample): bw.w[qw & 1]
  att_bias_load_half(bw, qw & 1, rbias, att_bias_voff(dl, sp.part, sp.tile_in_part * 4 + kw, lane), (qp0 - sp.s_lo) / ATT_BK + (qw >> 1));
  // the tile's bias block is the same for every sample: the four selection MFMAs run ONCE, their result (bias / c1 in accumulator
  // layout) is added to each sample's -lse / c1 on the vector pipe (16 v_add against 4 of 12 MFMAs per sample)
  f32x16 bias_acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) bias_acc[i] = 0.f;
  {
    const u32x4 bw0 = bw.w[qw & 1][0], bw1 = bw.w[qw & 1][1];
    bias_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh0, __builtin_bit_cast(f16x8, bw0), bias_acc, 0, 0, 0);
    bias_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh1, __builtin_bit_cast(f16x8, bw1), bias_acc, 0, 0, 0);
    bias_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl0, __builtin_bit_cast(f16x8, bw0), bias_acc, 0, 0, 0);
    bias_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl1, __builtin_bit_cast(f16x8, bw1), bias_acc, 0, 0, 0);
  }

  // ---- LDS-DMA staging: wave w moves piece w (rows 8w .. 8w+7) of each of the four 128-row tiles ----------------------------
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv), 0, (uint32_t)((size_t)p.total_rows * p.ld_qkv * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(bp.d_o), 0, (uint32_t)((size_t)p.total_rows * bp.ld_do * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(bp.nstat), 0, (uint32_t)((size_t)2 * p.H * p.total_rows * 4), 0x00020000);
  const int srow = wave * 8 + (lane >> 3);
  const uint32_t schunk = (uint32_t)(((lane & 7) ^ (srow & 7)) * 16);  // source-side swizzle of the row image
  // position -> (valid, first row of its segment for sample 0, rows per sample) is per-lane constant; the sample adds b * n
  auto seg_of = [&](int pos, int lim, bool& ok, int& row0, int& stride) {
    const bool txt = pos < ps.n0, img = pos >= ps.pos1 && pos < ps.NP;
    ok = (txt || img) && pos < lim;
    row0 = txt ? ps.base0 + pos : ps.base1 + (pos - ps.pos1);
    stride = txt ? ps.n0 : ps.n1;
  };
  bool k_ok, q_ok, s_ok;
  int k_row0, k_str, q_row0, q_str, s_row0, s_str;
  seg_of(sp.p0 + srow, sp.s_hi, k_ok, k_row0, k_str);
  seg_of(qp0 + srow, sp.s_hi, q_ok, q_row0, q_str);
  seg_of(qp0 + (wave & 1) * 64 + lane, sp.s_hi, s_ok, s_row0, s_str);  // waves 0..3: one half of one statistic
  const uint32_t colK = (uint32_t)(D + h * 64) * 2, colV = (uint32_t)(2 * D + h * 64) * 2, colQ = (uint32_t)(h * 64) * 2;
  const uint32_t stat_base = (uint32_t)(((wave >> 1) * p.H + h) * p.total_rows);
  auto dma = [&](int b, int stage) {
    unsigned char* base = smem + stage * STAGE + wave * 1024;
    // (the whole offset rides in the VECTOR operand: an absent row's 0xFFFFFFF0 is then out of range whatever else is added)
    const uint32_t kb = (uint32_t)(k_row0 + b * k_str) * (uint32_t)p.ld_qkv * 2 + schunk;
    const uint32_t qb2 = (uint32_t)(q_row0 + b * q_str) * (uint32_t)p.ld_qkv * 2 + schunk;
    const uint32_t ob = (uint32_t)(q_row0 + b * q_str) * (uint32_t)bp.ld_do * 2 + schunk;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (att_lds_void*)(base), 16, k_ok ? kb + colK : 0xFFFFFFF0u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (att_lds_void*)(base + 2 * ATT_TILE_BYTES), 16, k_ok ? kb + colV : 0xFFFFFFF0u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (att_lds_void*)(base + 4 * ATT_TILE_BYTES), 16, q_ok ? qb2 + colQ : 0xFFFFFFF0u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rdo, (att_lds_void*)(base + 6 * ATT_TILE_BYTES), 16, q_ok ? ob + colQ : 0xFFFFFFF0u, 0, 0, 0);
    if (wave < 4) {  // -lse / c1 (waves 0, 1) and -delta (waves 2, 3) of the tile's 128 queries; absent queries read 0 (the table masks them)
      const uint32_t so = s_ok ? (stat_base + (uint32_t)(s_row0 + b * s_str)) * 4 : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rst, (att_lds_void*)(smem + stage * STAGE + 8 * ATT_TILE_BYTES + wave * 256), 4, so, 0, 0, 0);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // The waits are BUILTINS, not inline asm: hipcc's wait-count pass cannot see into an asm statement, believed the bias
  // loads above still pending inside the loop and put a counted vmcnt in front of their first use -- which, four or five
  // LDS-DMA instructions later, waited for the NEXT sample's first tile.  The key-padding byte of a sample is requested one
  // trip ahead and BEFORE that trip's DMA for the same reason (vmcnt counts in order: waiting for a load that is younger
  // than the DMA drains the DMA).
  auto keep_byte = [&](int b) -> uint32_t {
    return (wave_keep && kvalid && keepk && b < b_hi) ? (uint32_t)keepk[(size_t)b * keep_n + keep_at] : 1u;
  };
  uint32_t keep_cur = keep_byte(b_lo);
  dma(b_lo, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  asm volatile("s_barrier" ::: "memory");
  for (int b = b_lo; b < b_hi; ++b) {
    const int cur = (b - b_lo) & 1;
    const uint32_t keep_next = keep_byte(b + 1);
    if (b + 1 < b_hi) dma(b + 1, cur ^ 1);  // that stage was last read in the previous trip (barrier below)
    const unsigned char* ldsK = smem + cur * STAGE;
    const unsigned char* ldsV = ldsK + 2 * ATT_TILE_BYTES;
    const unsigned char* ldsQ = ldsK + 4 * ATT_TILE_BYTES;
    const unsigned char* ldsO = ldsK + 6 * ATT_TILE_BYTES;
    const float* qstat = reinterpret_cast<const float*>(ldsK + 8 * ATT_TILE_BYTES);
    const float kmaskv = (kvalid && keep_cur != 0u) ? 0.f : -INFINITY;
    const bool wave_masked = __any(kmaskv != 0.f);
    const int q0 = qw * 32;
    f32x16 e, dp;
    {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qstat + q0 + 8 * g4 + 4 * hh);
      const f32x4 c = *reinterpret_cast<const f32x4*>(qstat + 128 + q0 + 8 * g4 + 4 * hh);
#pragma unroll
      for (int i = 0; i < 4; ++i) { e[4 * g4 + i] = a[i] + bias_acc[4 * g4 + i]; dp[4 * g4 + i] = c[i]; }
    }
    if (wave_masked) {
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] += kmaskv;
    }
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const bf16x8 qa = att_k_rowfrag(ldsQ, q0 + r, 2 * ss + hh), ka = att_k_rowfrag(ldsK, kw * 32 + r, 2 * ss + hh);
      const bf16x8 oa = att_k_rowfrag(ldsO, q0 + r, 2 * ss + hh), va = att_k_rowfrag(ldsV, kw * 32 + r, 2 * ss + hh);
      e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, ka, e, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, va, dp, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(att_exp2(e[i] * c1), dp[i], acc[i]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): stage cur ^ 1 has landed (and keep_next with it)
    asm volatile("s_barrier" ::: "memory");  // ... is published, stage cur is free
    keep_cur = keep_next;
  }

  // ---- histogram of the batch-summed dS (in the LDS that held the tiles) ---------------------------------------------------
  // 64-bit FIXED-POINT bins: ds_add_f32 costs ~110-150 cycles per wave instruction on this chip whatever the address pattern
  // (47 of this kernel's 346 us at 88 samples), ds_add_u64 runs at the LDS array's rate -- and integer sums do not depend on
  // the order the waves arrive in.  The step is 2^-48 of the item's largest |sum of dS| (a bin receives at most 2^14 addends).
  unsigned long long* hist64 = reinterpret_cast<unsigned long long*>(smem);
  unsigned* smax = reinterpret_cast<unsigned*>(hist64 + p.R);
  for (int i = tid; i < p.R; i += ATT_DB16_THREADS) hist64[i] = 0ull;
  if (tid == 0) *smax = 0u;
  __syncthreads();
  {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(acc[j]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) atomicMax(smax, __float_as_uint(m));  // non-negative floats order like their bit patterns
  }
  __syncthreads();
  int ex;
  (void)frexpf(__uint_as_float(*smax), &ex);  // largest |value| < 2^ex
  if (ex < -60) ex = -60;                     // (2^(48 - ex) must stay a finite float)
  const float FIX = ldexpf(1.0f, 48 - ex), UNFIX = ldexpf(1.0f, ex - 48);
  {
    // byte offsets (4 x relative-position index) of this lane's 16 (query, key) pairs through the [query][key] orientation:
    // the 32 lanes of a half read 64 consecutive bytes of one row
    const bool kin = kp < p.ld_idx;
    const int qpos0 = qp0 + qw * 32;
    uint32_t ids[16];
    bool same = true;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int qq = qpos0 + (j & 3) + 8 * (j >> 2) + 4 * hh;
      ids[j] = (kin && qq < p.idx_rows) ? (uint32_t)(unsigned short)p.idx[(size_t)qq * p.ld_idx + kp] : 0u;
      same = same && ids[j] == ids[0];
    }
    const uint32_t first = __builtin_amdgcn_readfirstlane(ids[0]);
    same = same && ids[0] == first;
    if (__all(same)) {  // text -> image pairs share ONE table row (vilt_module.py:180-181): reduce in registers
      float tsum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) tsum += acc[j];
      tsum = wave_sum(tsum);
      if (lane == 0) atomicAdd(hist64 + (first >> 2), (unsigned long long)(long long)(tsum * FIX));
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j)  // ids are byte offsets of 4-byte entries: 2 * ids addresses the 8-byte bin
        atomicAdd(reinterpret_cast<unsigned long long*>(smem + 2 * ids[j]), (unsigned long long)(long long)(acc[j] * FIX));
    }
  }
  __syncthreads();
  if (bp.dbias_part) {
    float* g = bp.dbias_part + (size_t)part_slot * p.R;
    for (int i = tid; i < p.R; i += ATT_DB16_THREADS) g[i] = (float)(long long)hist64[i] * UNFIX;
  } else {
    float* g = bp.dbias_t + (size_t)(p.head_row0 + h) * p.R;
    for (int i = tid; i < p.R; i += ATT_DB16_THREADS) {
      const long long v = (long long)hist64[i];
      if (v != 0) atomicAdd(g + i, (float)v * UNFIX);
    }
  }
}

// dbias_t[head_row0 + h][i] += sum over the work items of head h (item = (pair * H + h) * groups + grp) of part[item][i]
__global__ __launch_bounds__(256) void attn_dbias_fold_kernel(const float* __restrict__ part, int R, int H, int groups, int pairs,
                                                              float* __restrict__ dbias_t, int head_row0) {
  const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (i >= R) return;
  float sum = 0.f;
  for (int pr = 0; pr < pairs; ++pr)
    for (int g = 0; g < groups; ++g) sum += part[((size_t)(pr * H + h) * groups + g) * R + i];
  dbias_t[(size_t)(head_row0 + h) * R + i] += sum;
}

// work items of the bias-gradient kernel for this geometry: (key tile, query tile) pairs and sample groups
static void att_dbias_items(const attn_params_t& p, int& pairs, int& groups) {
  const int NP = p.seq.pos1 + p.seq.n1;
  if (p.mode == VLM_ATTN_SEPARATE) {
    const int a = (p.seq.n0 + ATT_BQ - 1) / ATT_BQ, c = (p.seq.n1 + ATT_BQ - 1) / ATT_BQ;
    pairs = a * a + c * c;
  } else {
    const int a = (NP + ATT_BQ - 1) / ATT_BQ;
    pairs = a * a;
  }
  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  if (pairs <= 0) {
    groups = 1;
    return;
  }
  // an item costs its samples (2.2 us each) plus the histogram tail (10 us); the CUs finish within half an
  // item of each other.  (Harness sweep, 300 (pair, head) items per group: 88 samples 330 / 294 / 282 / 297 / 298 us for 1..5
  // groups, 66: 253 / 226 / 218 / 229 / 243, 22: 108 / 104 / 113 / 127 / 131 -- the model picks 3, 3, 2.)
  static const int forced = []() {  // VLM_ATT_DB_GROUPS=n overrides the model (experiments); parsed once
    const char* e = getenv("VLM_ATT_DB_GROUPS");
    return e ? atoi(e) : 0;
  }();
  if (forced > 0) {
    groups = forced < p.seq.B ? forced : p.seq.B;
    return;
  }
  groups = 1;
  float best = 1e30f;
  for (int g = 1; g <= 8 && g <= p.seq.B; ++g) {
    const float item = 2.2f * (float)p.seq.B / (float)g + 10.0f;
    const float t = ((float)pairs * p.H * g / (float)cus + 0.5f) * item;
    if (t < best) { best = t; groups = g; }
  }
}

// ---- the fused dK / dV / bias-gradient launch (attention_bwd_dkvb.h): which form, how many sample groups -----------------------
struct dkvb_plan_t {
  int W;             // waves per workgroup (8: two per SIMD; 4 when the panel leaves room for four slots only); 0: does not apply
  int panel_blocks, groups, items;
  size_t lds;
};
static dkvb_plan_t dkvb_plan(const attn_params_t& p) {
  dkvb_plan_t pl = {0, 0, 1, 0, 0};
  static const int enabled = []() {  // VLM_ATT_BWD_FUSED=0: the round-2..4 pair attn_bwd_dkv_kernel + attn_bwd_dbias16_kernel (A/B)
    const char* e = getenv("VLM_ATT_BWD_FUSED");
    return e ? atoi(e) : 1;
  }();
  if (!enabled || !p.bias_t || !p.dense_t || !p.idx) return pl;
  for (int W = 8; W >= 4; W -= 4) {
    int pb;
    const size_t lds = dkvb_lds_bytes(p, W, pb);
    if (lds) { pl.W = W; pl.panel_blocks = pb; pl.lds = lds; break; }
  }
  if (!pl.W) return pl;
  const dkvb_geom_t gm = dkvb_geom(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode);
  const int nkb = gm.nkb[0] + gm.nkb[1];
  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  // one workgroup per CU (the panel): sample groups only where heads x key blocks leave CUs idle (small models)
  int g = cus / (nkb * p.H > 0 ? nkb * p.H : 1);
  if (g < 1) g = 1;
  const int per_round = pl.W;  // a group should hold at least one full round of samples
  if (g > (p.seq.B + per_round - 1) / per_round) g = (p.seq.B + per_round - 1) / per_round;
  if (g < 1) g = 1;
  pl.groups = g;
  pl.items = nkb * p.H * g;
  return pl;
}

extern "C" size_t vlm_attention_bwd_ws_floats(const vlm_attn_desc_t* d, int with_dbias) {
  attn_params_t p;
  if (att_fill_params(d, p) != VLM_OK) return 0;
  size_t n = (size_t)p.H * p.total_rows;  // delta
  if (p.bias_t) n *= 3;                    // + (-lse / c1 | -delta) for the 16-wave bias-gradient kernel
  if (with_dbias && p.bias_t) {
    int pairs, groups;
    att_dbias_items(p, pairs, groups);
    size_t items = (size_t)pairs * p.H * groups;
    const dkvb_plan_t pl = dkvb_plan(p);
    if (pl.W && (size_t)pl.items > items) items = (size_t)pl.items;  // either form's per-item histograms fit
    n += items * p.R;
  }
  return n;
}

extern "C" int vlm_attention_bwd(const vlm_attn_desc_t* d, const void* out, int ld_out, const void* d_out,
                                 int ld_dout, const float* lse, float* delta_ws, size_t ws_floats, void* dqkv, int ld_dqkv,
                                 float* dbias_t, const vlm_attn_colsum_t* colsum, void* stream) {
  attn_bwd_params_t bp;
  int rc = att_fill_params(d, bp.f);
  if (rc != VLM_OK) return rc;
  if (!out || !d_out || !lse || !delta_ws || !dqkv) return VLM_ERR_ARG;
  bp.dbias_part = nullptr;
  bp.nstat = nullptr;
  if ((ld_out & 7) || (ld_dout & 7) || (ld_dqkv & 3) || ((uintptr_t)out & 15) || ((uintptr_t)d_out & 15))
    return VLM_ERR_ARG;
  if (bp.f.bias_t && (!bp.f.idx || (bp.f.ld_idx & 3))) return VLM_ERR_ARG;
  attn_params_t& p = bp.f;
  const int nt0 = (p.seq.n0 + ATT_BQ - 1) / ATT_BQ, nt1 = (p.seq.n1 + ATT_BQ - 1) / ATT_BQ;
  if (nt0 + nt1 == 0 || p.seq.B == 0) return VLM_OK;
  hipStream_t s = (hipStream_t)stream;
  bp.d_o = reinterpret_cast<const bf16_t*>(d_out);
  bp.ld_do = ld_dout;
  bp.lse = lse;
  bp.delta = delta_ws;
  bp.dqkv = reinterpret_cast<bf16_t*>(dqkv);
  bp.ld_dqkv = ld_dqkv;
  bp.dbias_t = dbias_t;
  for (int sgm = 0; sgm < 2; ++sgm) {
    bp.dq_colsum[sgm] = colsum ? colsum->dq[sgm] : nullptr;
    bp.dv_colsum[sgm] = colsum ? colsum->dv[sgm] : nullptr;
  }

  // every argument is validated BEFORE the first launch (a rejected call leaves the stream untouched)
  if ((size_t)p.total_rows * p.ld_qkv * 2 >= (1ull << 32) || (size_t)p.total_rows * ld_dout * 2 >= (1ull << 32)) return VLM_ERR_UNSUPPORTED;
  if (p.bias_t && (!p.dense || !p.dense_t)) return VLM_ERR_ARG;  // biased attention runs on the dense tables (vlm_bias_dense)
  if (p.dense && p.dense_tiles != att_dense_layout(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode).tiles) return VLM_ERR_ARG;
  if (ws_floats < (size_t)p.H * p.total_rows) return VLM_ERR_WORKSPACE;  // delta[h][row]
  // the bias-gradient kernel's 64-bit histogram (R bins + the item's maximum) lives in its two operand stages' LDS
  if (p.bias_t && dbias_t && ((size_t)p.R * 8 + 8 > 2 * (size_t)(8 * ATT_TILE_BYTES + 1024) || !p.idx_t)) return VLM_ERR_UNSUPPORTED;
  // ... and takes its C operands from the dQ launch: workspace = delta | nstat (-lse / c1, -delta) | per-item histograms
  const size_t hr = (size_t)p.H * p.total_rows;
  const bool want_dbias = p.bias_t && dbias_t;
  if (want_dbias && ws_floats < 3 * hr) return VLM_ERR_WORKSPACE;  // vlm_attention_bwd_ws_floats says so
  bp.nstat = want_dbias ? delta_ws + hr : nullptr;
  bp.o = reinterpret_cast<const bf16_t*>(out);
  bp.ld_o = ld_out;
  bp.inv_c1 = 1.0f / (p.scale * ATT_LOG2E);

  const int nt = att_num_tiles(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode);
  dim3 grid(att_grid_size(nt, p.seq.B, p.H)), block(ATT_THREADS);
  const dkvb_plan_t pl = want_dbias ? dkvb_plan(p) : dkvb_plan_t{0, 0, 1, 0, 0};
  if (pl.W) {
    // dQ (+ delta), then dK / dV and the bias-table gradient in ONE launch: 7 MFMA products per score instead of 9
    bp.nstat = nullptr;  // (the 16-wave kernel's C operands are not needed)
    if (!att_dq2_launch(bp, grid, s)) {  // the hand-placed stream (attention_bwd_dq2.h) takes the calls it covers
#ifdef VLM_DIAG  // harness only: VLM_DIAG_DQ_PAD_LDS=bytes of unused dynamic LDS (40960: one workgroup per CU instead of two)
      static const size_t dq_pad = [] { const char* e = getenv("VLM_DIAG_DQ_PAD_LDS"); return e ? (size_t)atoi(e) : (size_t)0; }();
      hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), grid, block, dq_pad, s, bp);
#else
      hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), grid, block, 0, s, bp);
#endif
    }
    VLM_CHECK_LAUNCH();
    const size_t need = hr + (size_t)pl.items * p.R;
    bp.dbias_part = ws_floats >= need ? delta_ws + hr : nullptr;
    static bool attr_set[2] = {false, false};  // dynamic LDS beyond 64 KB has to be asked for once per kernel
    if (pl.W == 8) {
      if (!attr_set[0]) { (void)hipFuncSetAttribute((const void*)attn_bwd_dkvb_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set[0] = true; }
      hipLaunchKernelGGL((attn_bwd_dkvb_kernel<8>), dim3((unsigned)((pl.items + 7) / 8 * 8)), dim3(512), pl.lds, s, bp, pl.groups, pl.panel_blocks);
    } else {
      if (!attr_set[1]) { (void)hipFuncSetAttribute((const void*)attn_bwd_dkvb_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set[1] = true; }
      hipLaunchKernelGGL((attn_bwd_dkvb_kernel<4>), dim3((unsigned)((pl.items + 7) / 8 * 8)), dim3(256), pl.lds, s, bp, pl.groups, pl.panel_blocks);
    }
    if (bp.dbias_part) {
      VLM_CHECK_LAUNCH();
      const dkvb_geom_t gm = dkvb_geom(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode);
      hipLaunchKernelGGL(attn_dbias_fold_kernel, dim3((p.R + 255) / 256, p.H), dim3(256), 0, s, bp.dbias_part, p.R, p.H, pl.groups,
                         gm.nkb[0] + gm.nkb[1], dbias_t, p.head_row0);
    }
  } else if (p.bias_t) {
    if (!att_dq2_launch(bp, grid, s)) hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), grid, block, 0, s, bp);
    VLM_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<true>), grid, block, 0, s, bp);
    if (dbias_t) {
      VLM_CHECK_LAUNCH();
      // work items: every (128-key tile, 128-query tile of its span) pair x heads x sample groups; the samples are cut
      // into groups until the grid has ~3 items per CU (the per-item histogram is the price of every extra group)
      int pairs, groups;
      att_dbias_items(p, pairs, groups);
      const size_t items = (size_t)pairs * p.H * groups, need = 3 * hr + items * p.R;
      bp.dbias_part = ws_floats >= need ? delta_ws + 3 * hr : nullptr;
      hipLaunchKernelGGL(attn_bwd_dbias16_kernel, dim3((unsigned)((items + 7) / 8 * 8)), dim3(ATT_DB16_THREADS), 0, s, bp, groups);
      if (bp.dbias_part) {
        VLM_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_dbias_fold_kernel, dim3((p.R + 255) / 256, p.H), dim3(256), 0, s, bp.dbias_part, p.R, p.H, groups,
                           pairs, dbias_t, p.head_row0);
      }
    }
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<false>), grid, block, 0, s, bp);
    VLM_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<false>), grid, block, 0, s, bp);
  }
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
