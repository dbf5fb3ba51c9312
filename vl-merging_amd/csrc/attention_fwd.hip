// Fused attention forward (K4 + K7 + K7b): S = scale*Q K^T + rel-pos bias, key-padding mask, softmax, P V.
// Replaces reference vision_transformer.py:346-358 (global softmax attention with additive per-head
// relative-position bias and key-padding mask) together with get_rel_pos_bias (vilt_module.py:1061-1064) and the
// block-diagonal text/image split of separate_plain_forward / moe_forward (:567-584, :619-637) via `mode`.
//
// Work decomposition: one workgroup = 128 query POSITIONS of one (sample, head); 4 waves x 32 rows; two workgroups
// per CU, so every SIMD holds two waves that fill each other's matrix / vector gaps.  Streamed tiles = 64 key
// positions; K and V arrive by LDS-DMA (buffer_load ... lds, swizzles on the source address), double buffered.
// MFMA v_mfma_f32_32x32x16_bf16, "swapped" products so that the softmax row lives on ONE lane:
//   S'^T[key][q] = Bias^T/scale (2 selection MFMAs, attention_common.h) + K . Q^T  (A = K rows, ds_read_b128)
//   O^T[d][q]   += V^T . P^T      (A = V^T via ds_read_b64_tr_b16, B = the S'^T accumulators converted to bf16 in place)
// The vector pipe is the bottleneck at head dim 64 (PMC: 68 % VALU-active against 37 % MFMA-busy with every score
// paying v_fma + v_exp + v_add + v_max3/2 + v_cvt_pk/2), so it is left with the irreducible work only -- per score
// half a v_max3, one v_exp and half a v_cvt_pk -- and everything linear rides on the matrix pipe:
//   * Q is pre-multiplied by scale*log2(e), the dense table holds bias*log2(e): the accumulators ARE exponents;
//   * the running maximum is subtracted by starting each accumulator chain from a register tuple holding -m (the MFMA
//     C operand; rewritten only when a row's maximum grows by more than 2^6, the deferred rescale);
//   * the bias add, its unpack and the ragged-tile mask: two selection MFMAs on the fp16 table (ATT_NEG_BIG padding);
//   * the row sums: one MFMA per 16 keys with an all-ones A operand (every output row = sum_k P[k][q]).
// The additive key mask is applied only on tiles that hold text positions (padding tokens).
#include "vlm_common.h"
#include "attention_common.h"
#include "vlm_diag.h"
#include <type_traits>

// Row sums ride on the matrix pipe (one MFMA per 16 keys against an all-ones operand: the sum of the bf16-ROUNDED weights that
// also enter P V, so every output row is an exact convex combination); fp32 adds on the vector pipe measured the same speed.
template <bool HAS_BIAS>
__global__ __launch_bounds__(ATT_THREADS, 3) void attn_fwd_kernel(const attn_params_t p) {
  // Two stages as SEPARATE LDS objects: hipcc orders every ds_read behind all pending LDS-DMA writes it cannot prove
  // disjoint (s_waitcnt vmcnt(0) in front of the read) -- with one array and a computed stage offset that serialised
  // the prefetch of tile t+1 with the reads of tile t.  Distinct objects + compile-time stage selection keep the DMA
  // in flight across the whole tile.
  __shared__ __attribute__((aligned(16))) unsigned char ldsK0[ATT_TILE_BYTES], ldsK1[ATT_TILE_BYTES];  // [64 keys][128 B] row image
  __shared__ __attribute__((aligned(16))) unsigned char ldsV0[ATT_TILE_BYTES], ldsV1[ATT_TILE_BYTES];  // transposed-read image
  __shared__ float kmask0[64], kmask1[64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform BY CONSTRUCTION: tell the compiler (else waterfall loops)
  const int r = lane & 31, hh = lane >> 5;
  const att_pos_t ps = att_pos(p.seq);
  int wtile, b, h;
  if (!att_work_item(att_num_tiles(ps.n0, ps.n1, ps.pos1, p.mode), ps.B, p.H, wtile, b, h)) return;
  const att_span_t sp = att_span(ps, p.mode, wtile);
  const int D = p.H * 64;

  const int qp = sp.p0 + wave * 32 + r;                 // this lane's query position
  const int qrow_raw = qp < sp.s_hi ? att_row_of(ps, b, qp) : -1;
  const bool qvalid = qrow_raw >= 0;
  const size_t qrow = qvalid ? (size_t)qrow_raw : (size_t)att_row_of(ps, b, sp.s_lo);
  const int ntiles = (sp.s_hi - sp.s_lo + ATT_BK - 1) / ATT_BK;

  // ---- Q fragments (per-wave constant) ----------------------------------------------------------------------
  bf16x8 qf[4];
  {
    const bf16_t* qptr = p.qkv + qrow * p.ld_qkv + h * 64 + 8 * hh;
    const float c1 = p.scale * ATT_LOG2E;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 raw = *reinterpret_cast<const bf16x8*>(qptr + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (bf16_t)((float)raw[j] * c1);
    }
  }
  // The running reference point m rides in a FIFTH k-step of the score product instead of a 16-register C tuple:
  // k-slots 0, 1 of the extra step hold (1, 1) on the key side (a constant operand, no LDS read) and (-m_hi, -m_lo) on
  // the query side (m is an fp16 value: bf16 high part + exact bf16 remainder), so the chain starts from C = 0 and the
  // registers of the tuple are free (3 waves per SIMD need <= 168).
  bf16x8 kone, qm;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    kone[j] = (bf16_t)((hh == 0 && j < 2) ? 1.0f : 0.0f);
    qm[j] = (bf16_t)0.0f;
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

  // m = the value subtracted from this row's exponents so far (0 until a score exceeds 2^6, then a running maximum
  // rounded to fp16 so that it is exact in every format it passes through); negm = -m in all 16 registers, the C
  // operand of each score chain; lacc = row sums (every register holds this lane's query's sum)
  float m = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[0][i] = o[1][i] = 0.f;
  f32x16 lacc;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 16; ++i) lacc[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

  // a tile needs the additive mask iff it can hold a padding token (keep masks) or -- without a bias table, whose
  // ATT_NEG_BIG entries mask every invalid position -- a gap, foreign or past-the-end position
  auto tile_masked = [&](int kp0) {
    return (p.keep0 != nullptr && kp0 < ps.n0) || (p.keep1 != nullptr && kp0 + ATT_BK > ps.pos1) ||
           (!HAS_BIAS && (kp0 < ps.pos1 || kp0 + ATT_BK > sp.s_hi));
  };
  const att_dma_t dk = att_dma_init<false>(p.ld_qkv, wave, lane), dv = att_dma_init<true>(p.ld_qkv, wave, lane);
  auto stage = [&](int t, unsigned char* dstK, unsigned char* dstV, float* dstM) {
    const int kp0 = sp.s_lo + t * ATT_BK;
    if (tile_masked(kp0) && tid < 64) dstM[tid] = att_key_mask(ps, b, kp0 + tid, sp.s_hi, p.keep0, p.keep1);
    if (att_tile_plain(ps, kp0, sp.s_hi)) {  // workgroup-uniform
      att_dma_plain(rkv, dstK, dk, ps, b, kp0, p.ld_qkv, D + h * 64, wave);
      att_dma_plain(rkv, dstV, dv, ps, b, kp0, p.ld_qkv, 2 * D + h * 64, wave);
    } else {
      att_dma_any(rkv, dstK, dk, ps, b, kp0, sp.s_hi, p.ld_qkv, D + h * 64, wave);
      att_dma_any(rkv, dstV, dv, ps, b, kp0, sp.s_hi, p.ld_qkv, 2 * D + h * 64, wave);
    }
  };
  // every wave drains its own LDS-DMA pieces (and bias rows) before the barrier that publishes the tile
#define ATT_PUBLISH()                                  \
  do {                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
    __syncthreads();                                   \
  } while (0)
// inside the tile loop of the biased kernel the four bias loads of the NEXT tile are younger than this wave's LDS-DMA pieces
// (vmcnt retires in order): the tile is published once all but those four have landed
// (one asm statement: __syncthreads() would put its own vmcnt(0) in front of the barrier -- to the compiler every pending
// vector-memory operation might be an LDS-DMA write)
#define ATT_PUBLISH_KEEP4()                                                        \
  do {                                                                             \
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
  } while (0)

  att_bias_t bw;  // ONE set: the next tile's rows are requested as soon as this tile's selection MFMAs have been issued
  if (HAS_BIAS) att_bias_load(bw, rbias, bvoff, 0);  // issued BEFORE the DMA: vmcnt retires in order
  stage(0, ldsK0, ldsV0, kmask0);
  ATT_PUBLISH();

  // one streamed tile; `bw` = this tile's bias rows (complete since the last publish), `bn` receives the next tile's
  ATT_STAMP_DECL()
  auto tile = [&](int t, const unsigned char* lk, const unsigned char* lv, const float* km, unsigned char* nk,
                  unsigned char* nv, float* nm_) {
    const int kp0 = sp.s_lo + t * ATT_BK;
    ATT_STAMP(slot++);
    if (t + 1 < ntiles) stage(t + 1, nk, nv, nm_);  // nk / nv / nm_: the other stage

    // ---- E^T = -m + Bias^T*log2e + K (c1 Q)^T : two 32-key chains of exponents ------------------------------------
    ATT_STAMP(slot++);  // after issuing the next tile's loads
    // One 32-key block at a time, start to finish: E^T = bias + (-m) + K (c1 Q)^T, its own reference-point decision, exp2,
    // row sums, O^T += V^T P^T.  Only ONE 16-register score tuple is live (two blocks in flight cost 16 more registers and
    // with them the third wave per SIMD); the other waves of the SIMD fill the chain's gaps.
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 kfr[4];
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) kfr[ss] = att_k_rowfrag(lk, kb * 32 + r, 2 * ss + hh);
      f32x16 s;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = 0.f;
      if (HAS_BIAS) s = att_bias_mfma(sel0, sel1, bw.w[kb], s);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qm, s, 0, 0, 0);  // - m
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ss], qf[ss], s, 0, 0, 0);
      ATT_STAMP(slot++);  // score chain issued
      // the next tile's rows of this block: same registers, consumed by the selection MFMAs above
      if (HAS_BIAS && t + 1 < ntiles) att_bias_load_half(bw, kb, rbias, bvoff, t + 1);
      if (tile_masked(kp0)) {  // workgroup-uniform
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 mk = *reinterpret_cast<const f32x4*>(km + kb * 32 + 8 * g4 + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[4 * g4 + e] += mk[e];
        }
      }
      float mx = att_max3(s[0], s[1], s[2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) mx = att_max3(mx, s[i], s[i + 1]);   // 3..14
      mx = att_max2(mx, s[15]);
      mx = att_max2(mx, att_other_half(mx));
      ATT_STAMP(slot++);  // maximum known
      if (__any(mx > 6.0f)) {  // some row's exponents exceed 2^6: move those rows' reference points (rare after tile 0)
        const float m_new = mx > 0.f ? (float)(_Float16)(m + mx) : m;
        const float delta = m_new - m;  // exact: both are fp16 values
        const float alpha = att_exp2(-delta);
        m = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) lacc[i] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[i] -= delta;
          o[0][i] *= alpha;
          o[1][i] *= alpha;
        }
        const bf16_t mh = (bf16_t)(-m_new);
        const bf16_t ml = (bf16_t)(-m_new - (float)mh);  // exact: an fp16 value minus its 8-bit head
        if (hh == 0) { qm[0] = mh; qm[1] = ml; }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = att_exp2(s[i]);
      ATT_STAMP(slot++);  // exponentials issued
      // row sums on the vector pipe: four independent partial sums
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[8 * s2 + j];
        lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 vf = att_tr_frag(lv, kb * 32 + 16 * s2, db, lane);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
        }
      }
      ATT_STAMP(slot++);  // P V issued
    }
    if (HAS_BIAS && t + 1 < ntiles) ATT_PUBLISH_KEEP4();
    else ATT_PUBLISH();
  };
  for (int t = 0; t < ntiles; t += 2) {  // two tiles per trip: the LDS stages are distinct objects, selected at compile time
    tile(t, ldsK0, ldsV0, kmask0, ldsK1, ldsV1, kmask1);
    if (t + 1 < ntiles) tile(t + 1, ldsK1, ldsV1, kmask1, ldsK0, ldsV0, kmask0);
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------
  const float lt = lacc[0];
  const float inv = lt > 0.f ? 1.0f / lt : 0.f;
  if (qvalid) {
    bf16_t* op = p.out + qrow * p.ld_out + h * 64 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[db][4 * g4 + e] * inv);
        *reinterpret_cast<bf16x4*>(op + db * 32 + 8 * g4) = v;
      }
    if (hh == 0 && p.lse) p.lse[(size_t)h * p.total_rows + qrow] = m + log2f(lt);
  }
}

extern "C" int vlm_attention_fwd(const vlm_attn_desc_t* d, void* out, int ld_out, float* lse, void* stream) {
  attn_params_t p;
  int rc = att_fill_params(d, p);
  if (rc != VLM_OK) return rc;
  if (!out || (ld_out & 3)) return VLM_ERR_ARG;
  if (p.bias_t && !p.dense) return VLM_ERR_ARG;  // biased attention runs on the dense table (vlm_bias_dense)
  if ((size_t)p.total_rows * p.ld_qkv * 2 >= (1ull << 32)) return VLM_ERR_UNSUPPORTED;
  p.out = reinterpret_cast<bf16_t*>(out);
  p.ld_out = ld_out;
  p.lse = lse;
  const int nt = att_num_tiles(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode);
  if (nt == 0 || p.seq.B == 0) return VLM_OK;
  if (p.dense && p.dense_tiles != att_dense_layout(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode).tiles) return VLM_ERR_ARG;
  const size_t smem = 0;
  dim3 grid(att_grid_size(nt, p.seq.B, p.H)), block(ATT_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (p.bias_t) {  // the hand-placed stream (attention_fwd2.hip) takes the calls it covers
    const int r2 = att_fwd2_launch(p, s);
    if (r2 < 0) return r2;
    if (r2 == 1) return VLM_OK;
  }
  if (p.bias_t) hipLaunchKernelGGL((attn_fwd_kernel<true>), grid, block, smem, s, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<false>), grid, block, smem, s, p);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// ---------------------------------------------------------------------------------------------- dense bias
// The reference's get_rel_pos_bias (vilt_module.py:1061-1064) for all heads and layers at once: table[index] * log2(e) in
// fp16 (exponent units: the kernels' score accumulators are base-2 exponents), in the tiled MFMA-operand order of att_dense_layout (attention_common.h), one workgroup per 4-KiB tile.
__global__ __launch_bounds__(256) void bias_dense_kernel(const float* __restrict__ bias_t, int R,
                                                         const int16_t* __restrict__ index, int ld_index, int n0, int n1,
                                                         int pos1, int mode, int k_major, _Float16* __restrict__ out) {
  const att_dense_layout_t L = att_dense_layout(n0, n1, pos1, mode);
  int tile = blockIdx.x;
  const int part = tile >= L.nsb[0] * L.nst[0] ? 1 : 0;
  if (part) tile -= L.nsb[0] * L.nst[0];
  const int sb = tile / L.nst[part], st = tile - sb * L.nst[part];
  const int lane = threadIdx.x & 63, op = threadIdx.x >> 6;  // operand (blk, j) = (op >> 1, op & 1)
  const int NP = pos1 + n1;
  const int s_pos = L.org[part] + 32 * sb + (lane & 31);
  const int t_pos0 = L.org[part] + 64 * st + 32 * (op >> 1) + 16 * (lane >> 5) + 8 * (op & 1);
  auto member = [&](int p_) { return p_ >= L.org[part] && p_ < L.lim[part] && (p_ < n0 || p_ >= pos1) && p_ < NP; };
  const float* col = bias_t + (size_t)blockIdx.y * R;
  typedef __attribute__((ext_vector_type(8))) _Float16 h8;
  h8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int t_pos = t_pos0 + e;
    float x = ATT_NEG_BIG;
    if (member(s_pos) && member(t_pos)) {
      const int q = k_major ? t_pos : s_pos, k = k_major ? s_pos : t_pos;
      x = col[((unsigned short)index[(size_t)q * ld_index + k]) >> 2] * ATT_LOG2E;
    }
    v[e] = (_Float16)x;
  }
  *reinterpret_cast<h8*>(out + ((size_t)blockIdx.y * L.tiles + blockIdx.x) * 2048 + threadIdx.x * 8) = v;
}

extern "C" size_t vlm_bias_dense_bytes(int n0, int n1, int pos1, int mode) {
  if (n0 < 0 || n1 < 0 || pos1 < n0 || (pos1 & 7) || (mode != VLM_ATTN_JOINT && mode != VLM_ATTN_SEPARATE)) return 0;
  return (size_t)att_dense_layout(n0, n1, pos1, mode).tiles * 4096;
}

extern "C" int vlm_bias_dense(const float* bias_t, int n_cols, int R, const int16_t* index, int ld_index, int n0, int n1,
                              int pos1, int mode, int k_major, void* out_f16, void* stream) {
  const size_t bytes = vlm_bias_dense_bytes(n0, n1, pos1, mode);
  if (n_cols == 0 || bytes == 0) return bytes == 0 && n0 + n1 > 0 ? VLM_ERR_ARG : VLM_OK;
  if (!bias_t || !index || !out_f16 || n_cols < 0 || R <= 0 || R > 8191 || ld_index < pos1 + n1 ||
      ((uintptr_t)out_f16 & 15))
    return VLM_ERR_ARG;
  hipLaunchKernelGGL(bias_dense_kernel, dim3((unsigned)(bytes / 4096), n_cols), dim3(256), 0, (hipStream_t)stream, bias_t, R,
                     index, ld_index, n0, n1, pos1, mode, k_major, reinterpret_cast<_Float16*>(out_f16));
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
