// Fused attention forward (K4 + K7 + K7b): S = scale*Q K^T + rel-pos bias, key-padding mask, softmax, P V.
// Replaces reference vision_transformer.py:346-358 (global softmax attention with additive per-head
// relative-position bias and key-padding mask) together with get_rel_pos_bias (vilt_module.py:1061-1064, the
// [144,N,N] materialisation is never formed: the (layer, head) column of the bias table sits in LDS and is gathered
// through the int16 relative-position index) and the block-diagonal text/image split of
// separate_plain_forward / moe_forward (:567-584, :619-637) via `mode`.
//
// Work decomposition: one workgroup = 128 query rows of one (sample, head); 4 waves x 32 rows.  K/V tiles of 64
// keys are staged global -> VGPR -> LDS (double buffered, next tile's loads in flight during this tile's MFMAs).
// MFMA v_mfma_f32_32x32x16_bf16, "swapped" products so that the softmax row lives on ONE lane:
//   S^T[key][q] = K . Q^T        (A = K rows from LDS via ds_read_b128, B = Q fragment held in VGPRs)
//   O^T[d][q]  += V^T . P^T      (A = V^T via ds_read_b64_tr_b16 from the row-major V tile, B = the S^T accumulator
//                                 registers converted to bf16 in place: no LDS round trip, no cross-lane moves)
// so row max / row sum / rescale are per-lane scalars (one __shfl_xor(32) per tile joins the two half-waves).
// Softmax runs in the exp2 domain in fp32 (scale*log2e folded into one FMA with the gathered bias).
#include "vlm_common.h"
#include "attention_common.h"
#include <type_traits>

template <int BIAS>  // 0 none, 1 LDS-table gather through the int16 index, 2 dense fp16 bias
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_fwd_kernel(const attn_params_t p) {
  constexpr bool HAS_BIAS = BIAS != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsK = smem;                                    // [2][64 keys][128 B] swizzled
  unsigned char* ldsV = smem + 2 * ATT_TILE_BYTES;               // [2][64 keys][128 B] swizzled for tr reads
  float* kmask = reinterpret_cast<float*>(smem + 4 * ATT_TILE_BYTES);        // [2][64]
  float* tab = reinterpret_cast<float*>(smem + 4 * ATT_TILE_BYTES + 512);    // [R] bias column * log2e

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const attn_seq_t sq = p.seq;
  const int D = p.H * 64;

  const int nt0 = (sq.n0 + ATT_BQ - 1) / ATT_BQ;
  int qt = blockIdx.x;
  const int seg = qt >= nt0 ? 1 : 0;
  if (seg) qt -= nt0;
  const int nq = seg ? sq.n1 : sq.n0;
  const int q = qt * ATT_BQ + wave * 32 + r;
  const bool qvalid = q < nq;
  const int qc = qvalid ? q : nq - 1;
  const size_t qrow = (size_t)(seg ? sq.base1 + b * sq.n1 : sq.base0 + b * sq.n0) + qc;
  const int qpos = (seg ? sq.pos1 : 0) + qc;

  // ---- key ranges -------------------------------------------------------------------------------------------
  att_ranges_t kr = att_key_ranges(sq, p.mode, seg, b, p.keep0, p.keep1);
  const int ntiles = kr.nt[0] + kr.nt[1];

  // ---- Q fragments (per-wave constant) and bias column ------------------------------------------------------
  bf16x8 qf[4];
  {
    const bf16_t* qp = p.qkv + qrow * p.ld_qkv + h * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  if (BIAS == 1) {
    const float* col = p.bias_t + (size_t)(p.head_row0 + h) * p.R;
    for (int i = tid; i < p.R; i += ATT_THREADS) tab[i] = col[i] * ATT_LOG2E;
  }
  // 16-bit matrix the per-tile 8-byte loads walk: the shared int16 index, or this head's slice of the dense fp16 bias
  const void* mat16 = BIAS == 2 ? (const void*)(p.dense + (size_t)(p.head_row0 + h) * p.idx_rows * p.ld_idx) : (const void*)p.idx;
  const __amdgpu_buffer_rsrc_t ridx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(mat16), 0, HAS_BIAS ? p.idx_rows * p.ld_idx * 2 : 0, 0x00020000);
  const uint32_t irow = (uint32_t)qpos * p.ld_idx;
  u32x2 iw[8];
  if (HAS_BIAS) att_idx_tile(ridx, irow, (uint32_t)kr.pos[kr.nt[0] > 0 ? 0 : 1], hh, iw);

  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[0][i] = o[1][i] = 0.f;

  att_stage_t st;
  att_stage_load(st, p.qkv, p.ld_qkv, D, h, kr, 0, tid);
  att_stage_store(st, ldsK, ldsV, kmask, tid);
  __syncthreads();

  const float c1 = p.scale * ATT_LOG2E;
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) att_stage_load(st, p.qkv, p.ld_qkv, D, h, kr, t + 1, tid);
    const unsigned char* lk = ldsK + cur * ATT_TILE_BYTES;
    const unsigned char* lv = ldsV + cur * ATT_TILE_BYTES;
    const float* km = kmask + cur * 64;
    int rng, k0;
    att_tile_origin(kr, t, rng, k0);

    // ---- S^T = K Q^T : two 32-key chains -----------------------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const bf16x8 a = att_k_rowfrag(lk, kb * 32 + r, 2 * ss + hh);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ss], s[kb], 0, 0, 0);
      }
    }
    // ---- scale + bias gather (+ mask only on tiles that can hold a masked / padded key) --------------------------
    float mx = -INFINITY;
    const bool need_mask = (k0 + ATT_BK > kr.n[rng]) || (kr.keep[rng] != nullptr);  // wave-uniform
    auto score = [&](auto masked) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int kl = kb * 32 + 8 * g4 + 4 * hh;  // local key of element 0 of this group of 4
          float bv[4] = {0.f, 0.f, 0.f, 0.f};
          if (HAS_BIAS) att_bias4<BIAS>(tab, iw[kb * 4 + g4], bv);
          f32x4 mk = {0.f, 0.f, 0.f, 0.f};
          if (decltype(masked)::value) mk = *reinterpret_cast<const f32x4*>(km + kl);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = fmaf(s[kb][4 * g4 + e], c1, bv[e]);
            if (decltype(masked)::value) v += mk[e];
            s[kb][4 * g4 + e] = v;
            mx = fmaxf(mx, v);
          }
        }
      }
    };
    if (need_mask) score(std::true_type{});
    else score(std::false_type{});
    if (HAS_BIAS && t + 1 < ntiles) {  // next tile's indices: in flight during the softmax and P.V below
      int rng1, k1;
      att_tile_origin(kr, t + 1, rng1, k1);
      att_idx_tile(ridx, irow, (uint32_t)(kr.pos[rng1] + k1), hh, iw);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // deferred rescale (log2 domain): while no row of the wave grew by more than 2^6 keep the old maximum -- P then
    // ranges up to 64 instead of 1 (same relative precision in bf16 / fp32) and the 32-register O rescale is skipped
    float m_use;
    const bool grow = !(mx - m <= 6.0f);  // also true for m = -inf (first tile) and NaN
    if (__any(grow)) {
      const float m_new = fmaxf(m, mx);
      m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = att_exp2(m - m_use);  // m = -inf -> 0
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] *= alpha;
        o[1][i] *= alpha;
      }
    } else {
      m_use = m;
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = att_exp2(s[kb][i] - m_use);
        s[kb][i] = pv;
        rs += pv;
      }
    l += rs;
    // ---- O^T += V^T P^T ---------------------------------------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kb][8 * s2 + j];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 vf = att_tr_frag(lv, kb * 32 + 16 * s2, db, lane);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
        }
      }
    }
    if (t + 1 < ntiles) att_stage_store(st, ldsK + (cur ^ 1) * ATT_TILE_BYTES, ldsV + (cur ^ 1) * ATT_TILE_BYTES,
                                        kmask + (cur ^ 1) * 64, tid);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------
  const float lt = l + __shfl_xor(l, 32, 64);
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
  p.out = reinterpret_cast<bf16_t*>(out);
  p.ld_out = ld_out;
  p.lse = lse;
  const int nt0 = (p.seq.n0 + ATT_BQ - 1) / ATT_BQ, nt1 = (p.seq.n1 + ATT_BQ - 1) / ATT_BQ;
  if (nt0 + nt1 == 0 || p.seq.B == 0) return VLM_OK;
  const size_t smem = 4 * ATT_TILE_BYTES + 512 + (size_t)((p.R + 3) & ~3) * 4;
  if (smem > 160 * 1024) return VLM_ERR_UNSUPPORTED;
  dim3 grid(nt0 + nt1, p.H, p.seq.B), block(ATT_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (p.bias_t && p.dense) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VLM_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_fwd_kernel<2>), grid, block, smem, s, p);
  } else if (p.bias_t) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return VLM_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_fwd_kernel<1>), grid, block, smem, s, p);
  } else {
    hipLaunchKernelGGL((attn_fwd_kernel<0>), grid, block, smem, s, p);
  }
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}


// ---------------------------------------------------------------------------------------------- dense bias
__global__ __launch_bounds__(256) void bias_dense_kernel(const float* __restrict__ bias_t, int R,
                                                         const int16_t* __restrict__ index, int ld, int rows,
                                                         _Float16* __restrict__ out) {
  const int r = blockIdx.x, c = blockIdx.y;
  const float* col = bias_t + (size_t)c * R;
  const int16_t* irow = index + (size_t)r * ld;
  _Float16* o = out + ((size_t)c * rows + r) * ld;
  for (int k = threadIdx.x * 4; k < ld; k += 256 * 4) {  // ld % 4 == 0
    const s16x4 iv = *reinterpret_cast<const s16x4*>(irow + k);
    f16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (_Float16)(col[((unsigned short)iv[e]) >> 2] * ATT_LOG2E);
    *reinterpret_cast<f16x4*>(o + k) = v;
  }
}

extern "C" int vlm_bias_dense(const float* bias_t, int n_cols, int R, const int16_t* index, int ld_index, int index_rows,
                              void* out_f16, void* stream) {
  if (n_cols == 0 || index_rows == 0) return VLM_OK;
  if (!bias_t || !index || !out_f16 || n_cols < 0 || R <= 0 || R > 8191 || index_rows < 0 || ld_index <= 0 ||
      (ld_index & 3) || ((uintptr_t)index & 7) || ((uintptr_t)out_f16 & 7))
    return VLM_ERR_ARG;
  hipLaunchKernelGGL(bias_dense_kernel, dim3(index_rows, n_cols), dim3(256), 0, (hipStream_t)stream, bias_t, R, index,
                     ld_index, index_rows, reinterpret_cast<_Float16*>(out_f16));
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
