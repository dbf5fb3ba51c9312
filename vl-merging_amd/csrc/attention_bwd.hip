// Fused attention backward: recompute-based (flash style), four launches per (layer, pass):
//   1. delta[h][row] = sum_d dO*O
//   2. dQ kernel  (query-stationary, same structure as the forward; lane <-> query)
//        S^T = K Q^T -> P^T = exp2(S2 - lse2) ; dP^T = V dO^T ; dS^T = P^T o (dP^T - delta) ;
//        dQ^T[d][q] += K^T . dS^T      (A = K^T by ds_read_b64_tr_b16, B = dS^T accumulator regs as bf16)
//   3. dK/dV kernel (key-stationary; lane <-> key)
//        S = Q K^T -> P ; dP = dO V^T ; dS = P o (dP - delta) ;
//        dV[key][d] += P^T dO ,  dK[key][d] += scale * dS^T Q      (A = accumulator regs, B = dO / Q tr-read)
//   4. dBias kernel (only when the bias table needs a gradient): batch-summed dS -> LDS histogram -> global atomics
// This is what autograd derives for reference vision_transformer.py:346-358 + F.embedding in get_rel_pos_bias
// (vilt_module.py:1061-1064); the extra MFMA products (9 instead of 5) buy a deterministic dQ without atomics and a
// bias gradient whose (slow) LDS atomics are amortised over the batch.
#include "vlm_common.h"
#include "attention_common.h"

// ------------------------------------------------------------------------------------------------------- delta
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, int ld_o,
                                                         const bf16_t* __restrict__ d_o, int ld_do, int rows, int H,
                                                         float* __restrict__ delta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = H * 64;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    for (int c = lane * 8; c < D; c += 512) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(o + (size_t)row * ld_o + c);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(d_o + (size_t)row * ld_do + c);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)b[j];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if ((lane & 7) == 0) delta[(size_t)(c >> 6) * rows + row] = s;
    }
  }
}

struct attn_bwd_params_t {
  attn_params_t f;        // forward description (qkv, bias, index, ranges)
  const bf16_t* d_o;      // [rows, H*64]
  int ld_do;
  const float* lse;       // [H, rows] log2 domain
  const float* delta;     // [H, rows]
  bf16_t* dqkv;           // [rows, 3*H*64]
  int ld_dqkv;
  float* dbias_t;         // [n_cols, R] accumulate
  float* dq_colsum[2];    // per segment (0 text rows, 1 image rows): [H*64] += column sums of dQ (q_bias grad) or NULL
  float* dv_colsum[2];    // same for dV (v_bias gradient)
};

// ----------------------------------------------------------------------------------------------------- dQ kernel
template <int BIAS>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dq_kernel(const attn_bwd_params_t bp) {
  constexpr bool HAS_BIAS = BIAS != 0;
  const attn_params_t& p = bp.f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsK = smem;                          // [2] row image
  unsigned char* ldsKt = smem + 2 * ATT_TILE_BYTES;    // [2] tr image
  unsigned char* ldsV = smem + 4 * ATT_TILE_BYTES;     // [2] row image
  float* kmask = reinterpret_cast<float*>(smem + 6 * ATT_TILE_BYTES);
  float* tab = reinterpret_cast<float*>(smem + 6 * ATT_TILE_BYTES + 512);

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
  att_ranges_t kr = att_key_ranges(sq, p.mode, seg, b, p.keep0, p.keep1);
  const int ntiles = kr.nt[0] + kr.nt[1];

  bf16x8 qf[4], dof[4];
  {
    const bf16_t* qp = p.qkv + qrow * p.ld_qkv + h * 64 + 8 * hh;
    const bf16_t* dp = bp.d_o + qrow * bp.ld_do + h * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
      dof[s] = *reinterpret_cast<const bf16x8*>(dp + 16 * s);
    }
  }
  const float lse2 = bp.lse[(size_t)h * p.total_rows + qrow];
  const float dl = bp.delta[(size_t)h * p.total_rows + qrow];
  if (BIAS == 1) {
    const float* col = p.bias_t + (size_t)(p.head_row0 + h) * p.R;
    for (int i = tid; i < p.R; i += ATT_THREADS) tab[i] = col[i] * ATT_LOG2E;
  }
  const void* mat16 = BIAS == 2 ? (const void*)(p.dense + (size_t)(p.head_row0 + h) * p.idx_rows * p.ld_idx) : (const void*)p.idx;
  const __amdgpu_buffer_rsrc_t ridx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(mat16), 0, HAS_BIAS ? p.idx_rows * p.ld_idx * 2 : 0, 0x00020000);
  const uint32_t irow = (uint32_t)qpos * p.ld_idx;
  u32x2 iw[8];
  if (HAS_BIAS) att_idx_tile(ridx, irow, (uint32_t)kr.pos[kr.nt[0] > 0 ? 0 : 1], hh, iw);

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[0][i] = o[1][i] = 0.f;

  att_stage_t st;
  att_stage_load(st, p.qkv, p.ld_qkv, D, h, kr, 0, tid);
  att_tile_store_rows(st.k, ldsK, tid);
  att_tile_store_tr(st.k, ldsKt, tid);
  att_tile_store_rows(st.v, ldsV, tid);
  if (tid < 64) kmask[tid] = st.mask;
  __syncthreads();

  const float c1 = p.scale * ATT_LOG2E;
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) att_stage_load(st, p.qkv, p.ld_qkv, D, h, kr, t + 1, tid);
    const unsigned char* lk = ldsK + cur * ATT_TILE_BYTES;
    const unsigned char* lkt = ldsKt + cur * ATT_TILE_BYTES;
    const unsigned char* lv = ldsV + cur * ATT_TILE_BYTES;
    const float* km = kmask + cur * 64;
    int rng, k0;
    att_tile_origin(kr, t, rng, k0);
    const bool need_mask = (k0 + ATT_BK > kr.n[rng]) || (kr.keep[rng] != nullptr);

#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const bf16x8 a = att_k_rowfrag(lk, kb * 32 + r, 2 * ss + hh);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ss], s, 0, 0, 0);
        const bf16x8 va = att_k_rowfrag(lv, kb * 32 + r, 2 * ss + hh);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ss], dp, 0, 0, 0);
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int kl = kb * 32 + 8 * g4 + 4 * hh;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (HAS_BIAS) att_bias4<BIAS>(tab, iw[kb * 4 + g4], bv);
        f32x4 mk = {0.f, 0.f, 0.f, 0.f};
        if (need_mask) mk = *reinterpret_cast<const f32x4*>(km + kl);  // wave-uniform branch
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(s[4 * g4 + e], c1, bv[e]) + mk[e];
          const float pr = att_exp2(v - lse2);
          s[4 * g4 + e] = pr * (dp[4 * g4 + e] - dl) * p.scale;  // dS^T, pre-scaled
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 df;
#pragma unroll
        for (int j = 0; j < 8; ++j) df[j] = (bf16_t)s[8 * s2 + j];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 kf = att_tr_frag(lkt, kb * 32 + 16 * s2, db, lane);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, df, o[db], 0, 0, 0);
        }
      }
    }
    if (HAS_BIAS && t + 1 < ntiles) {  // next tile's indices (same registers), in flight across the barrier
      int rng1, k1;
      att_tile_origin(kr, t + 1, rng1, k1);
      att_idx_tile(ridx, irow, (uint32_t)(kr.pos[rng1] + k1), hh, iw);
    }
    if (t + 1 < ntiles) {
      att_tile_store_rows(st.k, ldsK + (cur ^ 1) * ATT_TILE_BYTES, tid);
      att_tile_store_tr(st.k, ldsKt + (cur ^ 1) * ATT_TILE_BYTES, tid);
      att_tile_store_rows(st.v, ldsV + (cur ^ 1) * ATT_TILE_BYTES, tid);
      if (tid < 64) kmask[(cur ^ 1) * 64 + tid] = st.mask;
    }
    __syncthreads();
  }
  if (qvalid) {
    bf16_t* op = bp.dqkv + qrow * bp.ld_dqkv + h * 64 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)o[db][4 * g4 + e];
        *reinterpret_cast<bf16x4*>(op + db * 32 + 8 * g4) = v;
      }
  }
  if (bp.dq_colsum[seg]) {
    // q_bias gradient = column sums of dQ: the 128x64 tile goes through LDS (row stride 68 floats: conflict-free
    // 16-B writes), every thread sums 32 rows of one column, 256 atomics per workgroup -- no second pass over dqkv
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
    float sum = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) sum += red[(part * 32 + i) * 68 + col];
    atomicAdd(bp.dq_colsum[seg] + h * 64 + col, sum);
  }
}

// ------------------------------------------------------------------------------------------- dK / dV / dBias kernel
template <int BIAS>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dkv_kernel(const attn_bwd_params_t bp) {
  constexpr bool HAS_BIAS = BIAS != 0;
  const attn_params_t& p = bp.f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsQ = smem;                           // row image  [64 q][64 d]
  unsigned char* ldsQt = smem + ATT_TILE_BYTES;         // tr image
  unsigned char* ldsO = smem + 2 * ATT_TILE_BYTES;      // dO row image
  unsigned char* ldsOt = smem + 3 * ATT_TILE_BYTES;     // dO tr image
  float* qstat = reinterpret_cast<float*>(smem + 4 * ATT_TILE_BYTES);  // [64] lse2 then [64] delta
  float* tab = qstat + 128;                                             // [R] bias column * log2e

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const attn_seq_t sq = p.seq;
  const int D = p.H * 64;
  const int nt0 = (sq.n0 + ATT_BQ - 1) / ATT_BQ;
  int kt = blockIdx.x;
  const int seg = kt >= nt0 ? 1 : 0;  // segment of this workgroup's keys
  if (seg) kt -= nt0;
  const int nk = seg ? sq.n1 : sq.n0;
  const int key = kt * ATT_BQ + wave * 32 + r;
  const bool kvalid = key < nk;
  const int kc = kvalid ? key : nk - 1;
  const size_t krow = (size_t)(seg ? sq.base1 + b * sq.n1 : sq.base0 + b * sq.n0) + kc;
  const int kpos = (seg ? sq.pos1 : 0) + kc;
  const uint8_t* keep = seg ? p.keep1 : p.keep0;
  const bool kkeep = kvalid && (!keep || keep[(size_t)b * nk + kc] != 0);
  const float kmaskv = kkeep ? 0.f : -INFINITY;
  // query ranges that see this key segment (same rule as the forward, by symmetry of the block structure)
  att_ranges_t qr = att_key_ranges(sq, p.mode, seg, b, nullptr, nullptr);
  const int ntiles = qr.nt[0] + qr.nt[1];

  bf16x8 kf[4], vf[4];
  {
    const bf16_t* kp = p.qkv + krow * p.ld_qkv + D + h * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
      vf[s] = *reinterpret_cast<const bf16x8*>(kp + D + 16 * s);
    }
  }
  if (BIAS == 1) {
    const float* col = p.bias_t + (size_t)(p.head_row0 + h) * p.R;
    for (int i = tid; i < p.R; i += ATT_THREADS) tab[i] = col[i] * ATT_LOG2E;
  }
  const void* mat16 = BIAS == 2 ? (const void*)(p.dense_t + (size_t)(p.head_row0 + h) * p.idx_t_rows * p.ld_idx_t) : (const void*)p.idx_t;
  const __amdgpu_buffer_rsrc_t ridx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(mat16), 0, HAS_BIAS ? p.idx_t_rows * p.ld_idx_t * 2 : 0, 0x00020000);
  const uint32_t irow = (uint32_t)kpos * p.ld_idx_t;
  u32x2 iw[8];
  if (HAS_BIAS) att_idx_tile(ridx, irow, (uint32_t)qr.pos[qr.nt[0] > 0 ? 0 : 1], hh, iw);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) dk[0][i] = dk[1][i] = dv[0][i] = dv[1][i] = 0.f;

  u32x4 sq_[2], so_[2];
  float s_lse = 0.f, s_dl = 0.f;
  auto stage_load = [&](int t) {
    int rng, q0;
    att_tile_origin(qr, t, rng, q0);
    att_tile_load(sq_, p.qkv, p.ld_qkv, h * 64, qr.rowbase[rng], q0, qr.n[rng], tid);
    att_tile_load(so_, bp.d_o, bp.ld_do, h * 64, qr.rowbase[rng], q0, qr.n[rng], tid);
    if (tid < 64) {
      const int qq = q0 + tid;
      const bool ok = qq < qr.n[rng];
      const size_t row = (size_t)qr.rowbase[rng] + (ok ? qq : 0);
      s_lse = ok ? bp.lse[(size_t)h * p.total_rows + row] : INFINITY;  // exp2(x - inf) = 0 for padded queries
      s_dl = ok ? bp.delta[(size_t)h * p.total_rows + row] : 0.f;
    }
  };
  auto stage_store = [&]() {
    att_tile_store_rows(sq_, ldsQ, tid);
    att_tile_store_tr(sq_, ldsQt, tid);
    att_tile_store_rows(so_, ldsO, tid);
    att_tile_store_tr(so_, ldsOt, tid);
    if (tid < 64) {
      qstat[tid] = s_lse;
      qstat[64 + tid] = s_dl;
    }
  };
  stage_load(0);
  stage_store();
  __syncthreads();

  const float c1 = p.scale * ATT_LOG2E;
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) stage_load(t + 1);
    int rng, q0;
    att_tile_origin(qr, t, rng, q0);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const bf16x8 a = att_k_rowfrag(ldsQ, qb * 32 + r, 2 * ss + hh);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[ss], s, 0, 0, 0);     // S[q][key]
        const bf16x8 oa = att_k_rowfrag(ldsO, qb * 32 + r, 2 * ss + hh);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ss], dp, 0, 0, 0);  // dP[q][key]
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int ql = qb * 32 + 8 * g4 + 4 * hh;  // local query row of element 0 of this group
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (HAS_BIAS) att_bias4<BIAS>(tab, iw[qb * 4 + g4], bv);
        const f32x4 ls = *reinterpret_cast<const f32x4*>(qstat + ql);
        const f32x4 dl = *reinterpret_cast<const f32x4*>(qstat + 64 + ql);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(s[4 * g4 + e], c1, bv[e]) + kmaskv;
          const float pr = att_exp2(v - ls[e]);
          s[4 * g4 + e] = pr;                               // P
          dp[4 * g4 + e] = pr * (dp[4 * g4 + e] - dl[e]);   // dS (natural units, w.r.t. the biased score)
        }
      }
      // dV += P^T dO ; dK += scale * dS^T Q   (A = accumulator regs as bf16, B = tr-read tiles)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf, df;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pf[j] = (bf16_t)s[8 * s2 + j];
          df[j] = (bf16_t)(dp[8 * s2 + j] * p.scale);
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 ob = att_tr_frag(ldsOt, qb * 32 + 16 * s2, db, lane);
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, ob, dv[db], 0, 0, 0);
          const bf16x8 qb_ = att_tr_frag(ldsQt, qb * 32 + 16 * s2, db, lane);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, qb_, dk[db], 0, 0, 0);
        }
      }
    }
    if (HAS_BIAS && t + 1 < ntiles) {  // next query tile's indices (same registers), in flight across the barriers
      int rng1, q1;
      att_tile_origin(qr, t + 1, rng1, q1);
      att_idx_tile(ridx, irow, (uint32_t)(qr.pos[rng1] + q1), hh, iw);
    }
    __syncthreads();
    if (t + 1 < ntiles) {
      stage_store();
      __syncthreads();
    }
  }

  // ---- store dK, dV: accumulator rows = keys (regs), column = d (lane & 31) ----------------------------------------
  {
    const int kbase = kt * ATT_BQ + wave * 32;
    const size_t row0 = (size_t)(seg ? sq.base1 + b * sq.n1 : sq.base0 + b * sq.n0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kl = (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (kbase + kl < nk) {
        bf16_t* dst = bp.dqkv + (row0 + kbase + kl) * bp.ld_dqkv + D + h * 64 + r;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dst[db * 32] = (bf16_t)dk[db][i];
          dst[D + db * 32] = (bf16_t)dv[db][i];
        }
      }
    }
    if (bp.dv_colsum[seg]) {  // v_bias gradient = column sums of dV over this workgroup's 128 keys: 64 atomics
      float* red = reinterpret_cast<float*>(smem);
      __syncthreads();
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kl = (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (kbase + kl < nk) sum += dv[db][i];
        }
        sum += __shfl_xor(sum, 32, 64);
        if (hh == 0) red[wave * 64 + db * 32 + r] = sum;
      }
      __syncthreads();
      if (tid < 64) atomicAdd(bp.dv_colsum[seg] + h * 64 + tid, red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- dBias kernel
// d(bias table column)[idx[q][key]] += sum_b dS[b,h,q,key].  LDS float atomics cost ~300 cycles per wave instruction
// on gfx950 (measured: the histogram inside the dK/dV kernel took 4x the rest of the backward), so the batch sum is
// taken FIRST, in registers: one workgroup owns a (128-key, 64-query) tile pair of one head, loops over the B
// samples recomputing S and dP (2 of the 7 MFMA products), and only then feeds the 8192 summed dS values through
// the LDS histogram (B-fold fewer atomics), with the all-indices-equal tiles (text->image pairs share ONE table
// row, vilt_module.py:180-181) reduced in registers instead.
__global__ __launch_bounds__(ATT_THREADS, 3) void attn_bwd_dbias_kernel(const attn_bwd_params_t bp) {
  // Every operand tile is fetched COOPERATIVELY (one wave instruction = 8 rows x 128 B = 8 cache lines) and the MFMA
  // fragments are read from LDS: per-lane row-fragment loads straight from global memory put 64 different cache
  // lines behind every wave instruction and left the kernel bound by the CU's address coalescer (measured 216 us;
  // the MFMA work is ~30 us).
  const attn_params_t& p = bp.f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsK = smem;                        // [128 keys][64 d] row image
  unsigned char* ldsV = smem + 2 * ATT_TILE_BYTES;   // [128 keys][64 d] row image
  unsigned char* ldsQ = smem + 4 * ATT_TILE_BYTES;   // [32 q][64 d] row image
  unsigned char* ldsO = ldsQ + 4096;                 // [32 q][64 d] dO row image
  float* qstat = reinterpret_cast<float*>(ldsO + 4096);  // lse2[32], delta[32]
  float* tab = qstat + 64;
  float* hist = reinterpret_cast<float*>(smem);  // aliases the K/V tiles: only used after the sample loop

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y;
  const attn_seq_t sq = p.seq;
  const int D = p.H * 64;
  const int nt0 = (sq.n0 + ATT_BQ - 1) / ATT_BQ;
  int kt = blockIdx.x;
  const int seg = kt >= nt0 ? 1 : 0;
  if (seg) kt -= nt0;
  const int nk = seg ? sq.n1 : sq.n0;
  att_ranges_t qr = att_key_ranges(sq, p.mode, seg, 0, nullptr, nullptr);
  const int ntq0 = (qr.n[0] + 31) >> 5, ntq1 = (qr.n[1] + 31) >> 5;  // 32-row query tiles of the interacting ranges
  if ((int)blockIdx.z >= ntq0 + ntq1) return;  // block-uniform, before any barrier
  const int rng = (int)blockIdx.z >= ntq0 ? 1 : 0;
  const int q0 = (rng ? (int)blockIdx.z - ntq0 : (int)blockIdx.z) << 5;
  const int qpos0 = qr.pos[rng] + q0;
  const int qn = qr.n[rng];
  const int qbase = qr.rowbase[rng];  // b = 0
  const int k0 = kt * ATT_BQ;
  const int key = k0 + wave * 32 + r;
  const bool kvalid = key < nk;
  const int kc = kvalid ? key : nk - 1;
  const int kpos = (seg ? sq.pos1 : 0) + kc;
  const uint8_t* keep = seg ? p.keep1 : p.keep0;
  const int kbase = seg ? sq.base1 : sq.base0;

  {
    const float* col = p.bias_t + (size_t)(p.head_row0 + h) * p.R;
    for (int i = tid; i < p.R; i += ATT_THREADS) tab[i] = col[i] * ATT_LOG2E;
  }
  const __amdgpu_buffer_rsrc_t ridx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int16_t*>(p.idx), 0, p.idx_rows * p.ld_idx * 2, 0x00020000);
  const uint32_t ld2 = (uint32_t)p.ld_idx * 2, ivoff = (uint32_t)(kpos + 4 * hh * p.ld_idx) * 2;
  // byte offsets (4 x relative-position index) of this lane's 16 (q, key) pairs: independent of the sample
  uint32_t ids[8];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    uint32_t io[4];
    att_idx4(ridx, ivoff, (uint32_t)(qpos0 + 8 * g4), ld2, io);
    ids[2 * g4] = io[0] | (io[1] << 16);
    ids[2 * g4 + 1] = io[2] | (io[3] << 16);
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  u32x4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3, rq, ro;
  float s_st = 0.f;
  const int srow = tid >> 3, schunk = tid & 7;
  // (plain macros, not lambdas: by-reference captures of register arrays were placed in scratch memory)
#define DB_LOAD_KV(U, RK, RV)                                                                              \
  {                                                                                                        \
    const int row_ = srow + 32 * (U);                                                                      \
    if (k0 + row_ < nk) {                                                                                  \
      const bf16_t* src_ = p.qkv + (size_t)(krow0_ + k0 + row_) * p.ld_qkv + D + h * 64 + schunk * 8;      \
      RK = *reinterpret_cast<const u32x4*>(src_);                                                          \
      RV = *reinterpret_cast<const u32x4*>(src_ + D);                                                      \
    } else {                                                                                               \
      RK = RV = (u32x4){0u, 0u, 0u, 0u};                                                                   \
    }                                                                                                      \
  }
#define DB_STAGE_LOAD(B_)                                                                                  \
  {                                                                                                        \
    const int krow0_ = kbase + (B_) * nk;                                                                  \
    DB_LOAD_KV(0, rk0, rv0) DB_LOAD_KV(1, rk1, rv1) DB_LOAD_KV(2, rk2, rv2) DB_LOAD_KV(3, rk3, rv3)        \
    const int rowbase_ = qbase + (B_) * qn;                                                                \
    const int qq_ = q0 + srow;                                                                             \
    if (qq_ < qn) {                                                                                        \
      rq = *reinterpret_cast<const u32x4*>(p.qkv + (size_t)(rowbase_ + qq_) * p.ld_qkv + h * 64 + schunk * 8); \
      ro = *reinterpret_cast<const u32x4*>(bp.d_o + (size_t)(rowbase_ + qq_) * bp.ld_do + h * 64 + schunk * 8); \
    } else {                                                                                               \
      rq = ro = (u32x4){0u, 0u, 0u, 0u};                                                                   \
    }                                                                                                      \
    if (tid < 64) {                                                                                        \
      const int q2_ = q0 + (tid & 31);                                                                     \
      const bool ok_ = q2_ < qn;                                                                           \
      const size_t rw_ = (size_t)rowbase_ + (ok_ ? q2_ : 0);                                               \
      const float* sp_ = (tid < 32) ? bp.lse : bp.delta;                                                   \
      s_st = ok_ ? sp_[(size_t)h * p.total_rows + rw_] : (tid < 32 ? INFINITY : 0.f);                      \
    }                                                                                                      \
  }
#define DB_STORE_KV(U, RK, RV)                                                                             \
  {                                                                                                        \
    const int row_ = srow + 32 * (U);                                                                      \
    const int byte_ = row_ * 128 + ((schunk ^ (row_ & 7)) << 4);                                           \
    *reinterpret_cast<u32x4*>(ldsK + byte_) = RK;                                                          \
    *reinterpret_cast<u32x4*>(ldsV + byte_) = RV;                                                          \
  }
#define DB_STAGE_STORE()                                                                                   \
  {                                                                                                        \
    DB_STORE_KV(0, rk0, rv0) DB_STORE_KV(1, rk1, rv1) DB_STORE_KV(2, rk2, rv2) DB_STORE_KV(3, rk3, rv3)    \
    const int byte_ = srow * 128 + ((schunk ^ (srow & 7)) << 4);                                           \
    *reinterpret_cast<u32x4*>(ldsQ + byte_) = rq;                                                          \
    *reinterpret_cast<u32x4*>(ldsO + byte_) = ro;                                                          \
    if (tid < 64) qstat[tid] = s_st;                                                                       \
  }
  DB_STAGE_LOAD(0)
  const float c1 = p.scale * ATT_LOG2E;
  for (int b = 0; b < sq.B; ++b) {
    __syncthreads();  // every wave is done with the previous sample's tiles
    DB_STAGE_STORE()
    __syncthreads();
    if (b + 1 < sq.B) DB_STAGE_LOAD(b + 1)  // flies during this sample's MFMAs
    const bool kkeep = kvalid && (!keep || keep[(size_t)b * nk + kc] != 0);
    const float kmaskv = kkeep ? 0.f : -INFINITY;
    f32x16 s, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const bf16x8 a = att_k_rowfrag(ldsQ, r, 2 * ss + hh);
      const bf16x8 kf = att_k_rowfrag(ldsK, wave * 32 + r, 2 * ss + hh);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf, s, 0, 0, 0);
      const bf16x8 oa = att_k_rowfrag(ldsO, r, 2 * ss + hh);
      const bf16x8 vf = att_k_rowfrag(ldsV, wave * 32 + r, 2 * ss + hh);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf, dp, 0, 0, 0);
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int ql = 8 * g4 + 4 * hh;
      const uint32_t w0 = ids[2 * g4], w1 = ids[2 * g4 + 1];
      const float bv[4] = {att_tab(tab, w0 & 0xffff), att_tab(tab, w0 >> 16), att_tab(tab, w1 & 0xffff), att_tab(tab, w1 >> 16)};
      const f32x4 ls = *reinterpret_cast<const f32x4*>(qstat + ql);
      const f32x4 dl = *reinterpret_cast<const f32x4*>(qstat + 32 + ql);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = fmaf(s[4 * g4 + e], c1, bv[e]) + kmaskv;
        const float pr = att_exp2(v - ls[e]);
        acc[4 * g4 + e] += pr * (dp[4 * g4 + e] - dl[e]);
      }
    }
  }
  // ---- histogram of the batch-summed dS (in the LDS that held the K/V tiles) ---------------------------------------
  __syncthreads();
  for (int i = tid; i < p.R; i += ATT_THREADS) hist[i] = 0.f;
  __syncthreads();
  {
    bool same = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) same = same && (ids[i] == ids[0]) && ((ids[i] >> 16) == (ids[i] & 0xffff));
    const uint32_t first = __builtin_amdgcn_readfirstlane(ids[0]);
    same = same && (ids[0] == first);
    if (__all(same)) {
      float tsum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) tsum += acc[i];
      tsum = wave_sum(tsum);
      if (lane == 0) atomicAdd(reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(hist) + (first & 0xffff)), tsum);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t w = ids[i >> 1];
        atomicAdd(reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(hist) + ((i & 1) ? (w >> 16) : (w & 0xffff))), acc[i]);
      }
    }
  }
  __syncthreads();
  float* g = bp.dbias_t + (size_t)(p.head_row0 + h) * p.R;
  for (int i = tid; i < p.R; i += ATT_THREADS) {
    const float v = hist[i];
    if (v != 0.f) atomicAdd(g + i, v);
  }
}

extern "C" int vlm_attention_bwd(const vlm_attn_desc_t* d, const void* out, int ld_out, const void* d_out,
                                 int ld_dout, const float* lse, float* delta_ws, void* dqkv, int ld_dqkv,
                                 float* dbias_t, const vlm_attn_colsum_t* colsum, void* stream) {
  attn_bwd_params_t bp;
  int rc = att_fill_params(d, bp.f);
  if (rc != VLM_OK) return rc;
  if (!out || !d_out || !lse || !delta_ws || !dqkv) return VLM_ERR_ARG;
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

  int cus = vlm_device_cus();
  if (cus <= 0) cus = 256;
  int dg = (p.total_rows + 3) / 4;
  if (dg > cus * 8) dg = cus * 8;
  hipLaunchKernelGGL(attn_delta_kernel, dim3(dg), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(out), ld_out, bp.d_o,
                     ld_dout, p.total_rows, p.H, delta_ws);
  VLM_CHECK_LAUNCH();

  dim3 grid(nt0 + nt1, p.H, p.seq.B), block(ATT_THREADS);
  const size_t Rp = (size_t)((p.R + 3) & ~3);
  const size_t smem_dq = 6 * ATT_TILE_BYTES + 512 + Rp * 4;
  const size_t smem_dkv = 4 * ATT_TILE_BYTES + 512 + Rp * 4;
  const size_t smem_db = 4 * ATT_TILE_BYTES + 2 * 4096 + 256 + Rp * 4;  // histogram aliases the K/V tiles (R*4 <= 32 KiB)
  if (smem_dq > 160 * 1024 || smem_dkv > 160 * 1024) return VLM_ERR_UNSUPPORTED;
  if (p.bias_t) {
    const bool dense = false;  // the round-1 backward kernels gather through the index (dense tables changed format)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dq) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dkv) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dq) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dkv) != hipSuccess)
      return VLM_ERR_LAUNCH;
    if (dense) hipLaunchKernelGGL((attn_bwd_dq_kernel<2>), grid, block, smem_dq, s, bp);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<1>), grid, block, smem_dq, s, bp);
    VLM_CHECK_LAUNCH();
    if (dense) hipLaunchKernelGGL((attn_bwd_dkv_kernel<2>), grid, block, smem_dkv, s, bp);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<1>), grid, block, smem_dkv, s, bp);
    if (dbias_t) {
      VLM_CHECK_LAUNCH();
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dbias_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_db) != hipSuccess)
        return VLM_ERR_LAUNCH;
      const int ntq = (p.seq.n0 + 31) / 32 + (p.seq.n1 + 31) / 32;  // upper bound of 32-row query tiles per key tile
      hipLaunchKernelGGL(attn_bwd_dbias_kernel, dim3(nt0 + nt1, p.H, ntq), block, smem_db, s, bp);
    }
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<0>), grid, block, smem_dq, s, bp);
    VLM_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<0>), grid, block, smem_dkv, s, bp);
  }
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
