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
#include <type_traits>

#ifndef ATT_FWD_WAVES
#define ATT_FWD_WAVES 2
#endif
#ifdef ATT_DIAG_STAMPS  // diagnostic builds only (tools/scratch/attn_bench.hip): s_memtime inside the tile loop
__device__ unsigned long long att_stamps[8 * 64];
#define ATT_STAMP(slot) do { const int s_ = (slot); if (blockIdx.x == 8 * 100 && lane == 0 && s_ < 64) att_stamps[wave * 64 + s_] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATT_STAMP(slot) do { } while (0)
#endif
template <bool HAS_BIAS>
__global__ __launch_bounds__(ATT_THREADS, ATT_FWD_WAVES) void attn_fwd_kernel(const attn_params_t p) {
#ifdef ATT_DIAG_LDSPAD
  __shared__ unsigned char diag_pad[ATT_DIAG_LDSPAD];
  if (p.H < 0) diag_pad[threadIdx.x] = 1;
#endif
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
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;
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
  f32x16 o[2], negm, lacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[0][i] = o[1][i] = negm[i] = lacc[i] = 0.f;

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
#ifdef ATT_DIAG_NOSYNC
#define ATT_PUBLISH() do { } while (0)
#else
#define ATT_PUBLISH()                                  \
  do {                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
    __syncthreads();                                   \
  } while (0)
#endif

  att_bias_t bwA, bwB;
  if (HAS_BIAS) att_bias_load(bwA, rbias, bvoff, 0);  // issued BEFORE the DMA: vmcnt retires in order
  stage(0, ldsK0, ldsV0, kmask0);
  ATT_PUBLISH();

  // one streamed tile; `bw` = this tile's bias rows (complete since the last publish), `bn` receives the next tile's
  int slot = 0;
  auto tile = [&](int t, const att_bias_t& bw, att_bias_t& bn, const unsigned char* lk, const unsigned char* lv,
                  const float* km, unsigned char* nk, unsigned char* nv, float* nm_) {
    const int kp0 = sp.s_lo + t * ATT_BK;
    ATT_STAMP(slot++);
    if (t + 1 < ntiles) {
#ifndef ATT_DIAG_NOBIASLOAD
      if (HAS_BIAS) att_bias_load(bn, rbias, bvoff, t + 1);
#endif
#ifndef ATT_DIAG_NODMA
      stage(t + 1, nk, nv, nm_);
#endif
    }

    // ---- E^T = -m + Bias^T*log2e + K (c1 Q)^T : two 32-key chains of exponents ------------------------------------
    ATT_STAMP(slot++);  // after issuing the next tile's loads
    // all eight K row fragments are requested up front (hipcc otherwise issues each ds_read right in front of its
    // MFMA and waits out the full LDS latency eight times per tile); the bias MFMAs need no LDS data and cover it
    bf16x8 kfr[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) kfr[kb][ss] = att_k_rowfrag(lk, kb * 32 + r, 2 * ss + hh);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = negm;
      if (HAS_BIAS) s[kb] = att_bias_mfma(sel0, sel1, bw.w[kb], s[kb]);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][ss], qf[ss], s[kb], 0, 0, 0);
    ATT_STAMP(slot++);  // S chain issued
    if (tile_masked(kp0)) {  // workgroup-uniform
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 mk = *reinterpret_cast<const f32x4*>(km + kb * 32 + 8 * g4 + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[kb][4 * g4 + e] += mk[e];
        }
    }
#ifdef ATT_DIAG_NOSOFTMAX
    float mx = s[0][0];
#else
    float mx = att_max3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) mx = att_max3(mx, s[0][i], s[0][i + 1]);   // 3..14
    mx = att_max3(mx, s[0][15], s[1][0]);
#pragma unroll
    for (int i = 1; i < 15; i += 2) mx = att_max3(mx, s[1][i], s[1][i + 1]);   // 1..14
    mx = fmaxf(mx, s[1][15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#endif
    ATT_STAMP(slot++);  // max known
    if (__any(mx > 6.0f)) {  // some row's exponents exceed 2^6: move those rows' reference points (rare after tile 0)
      const float m_new = mx > 0.f ? (float)(_Float16)(m + mx) : m;
      const float delta = m_new - m;  // exact: both are fp16 values
      const float alpha = att_exp2(-delta);
      m = m_new;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[0][i] -= delta;
        s[1][i] -= delta;
        o[0][i] *= alpha;
        o[1][i] *= alpha;
        lacc[i] *= alpha;
        negm[i] = -m_new;
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#ifndef ATT_DIAG_NOSOFTMAX
        s[kb][i] = att_exp2(s[kb][i]);
#endif
      }
    ATT_STAMP(slot++);  // exps issued
    // ---- O^T += V^T P^T ;  row sums += 1^T P^T -------------------------------------------------------------------
#ifdef ATT_DIAG_NOPV
    o[0][0] += s[0][0];
    if (false)
#endif
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kb][8 * s2 + j];
        lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 vf = att_tr_frag(lv, kb * 32 + 16 * s2, db, lane);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
        }
      }
    }
    ATT_STAMP(slot++);  // PV issued
    ATT_PUBLISH();
  };
  for (int t = 0; t < ntiles; t += 2) {  // two tiles per trip: the bias registers alternate without copies
    tile(t, bwA, bwB, ldsK0, ldsV0, kmask0, ldsK1, ldsV1, kmask1);
    if (t + 1 < ntiles) tile(t + 1, bwB, bwA, ldsK1, ldsV1, kmask1, ldsK0, ldsV0, kmask0);
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
