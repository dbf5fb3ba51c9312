// dK / dV AND the relative-position-bias gradient in ONE launch (round 5).  Included by attention_bwd.hip.
//
// What it replaces: attn_bwd_dkv_kernel + attn_bwd_dbias16_kernel recomputed S and dP twice (9 MFMA products per score for the
// 5 the algorithm has); the bias-table gradient is a histogram of G[q][k] = sum over the samples of dS, and dS exists, per
// sample, in the dK/dV kernel's registers.  Here a workgroup owns (head h, a block of 32 KEY positions) for ALL samples of its
// sample group, and keeps G for its 32 keys x every query position as an fp32 panel in LDS (617 x 32 x 4 B = 79 KB at 384^2,
// 120 KB at 480^2): the sum over the samples is taken in place, the histogram once at the end, from the panel.
// Reference: autograd of vision_transformer.py:346-358 (dK, dV) and of F.embedding in get_rel_pos_bias, vilt_module.py:1061-1064.
//
// Decomposition (key on the lane, as in attn_bwd_dkv_kernel): a WAVE owns one SAMPLE at a time -- its K / V fragments of the 32
// keys in registers, dK / dV complete in its own accumulators (no cross-wave sum), its own 8-KB LDS slot for the 32-query block of
// Q and dO it is working on (ONE image serves the row reads of S / dP and the transposed reads of dV / dK: chunk ^ f(row),
// f = row bits (3, 2 | 1), conflict-free for the coalesced 16-B stores, ds_read_b128 and ds_read_b64_tr_b16 by the rules of
// MI355X_MICROARCH.md -- tools/lds_layout_check.py).  The W waves of a workgroup work on W different samples at once and walk the
// query blocks SKEWED: wave w is at block (t + w S) mod nblk in step t, S = nblk / W, so within any S consecutive steps no two
// waves touch the same 4-KB block of the panel and the read-modify-write of G needs no atomics, only a workgroup barrier every S
// steps.  Nothing else is shared: no staging barrier, no statistics in LDS.
// Everything linear rides on the matrix pipe: -lse and -delta enter as a 5th k-step (the row's value split into two bf16 terms
// against ones on the key side -- 16 significant bits in the fp32 accumulator), the key-padding mask as one more k-slot of that
// step, the bias through two selection MFMAs on the tiled fp16 table (attention_common.h).
#pragma once

#define ATT_KB 32  // key positions per workgroup = query positions per block

// byte offset of 16-B chunk `ch` of row `row` inside a [32 rows][64 bf16] block image
__device__ __forceinline__ uint32_t dkvb_off(uint32_t row, uint32_t ch) {
  const uint32_t f = ((row >> 2) & 3u) | (((row >> 1) & 1u) << 2);
  return row * 128u + ((ch ^ f) << 4);
}

struct dkvb_geom_t {
  int nkb[2];  // key blocks of part 0 (JOINT: every position from 0; SEPARATE: the text segment) and part 1 (SEPARATE: image)
};
static inline __host__ __device__ dkvb_geom_t dkvb_geom(int n0, int n1, int pos1, int mode) {
  dkvb_geom_t g;
  if (mode == VLM_ATTN_SEPARATE) { g.nkb[0] = (n0 + ATT_KB - 1) / ATT_KB; g.nkb[1] = (n1 + ATT_KB - 1) / ATT_KB; }
  else { g.nkb[0] = (pos1 + n1 + ATT_KB - 1) / ATT_KB; g.nkb[1] = 0; }
  return g;
}
// query blocks the keys of a part see (the panel's height in 4-KB blocks)
static inline __host__ __device__ int dkvb_nblk(int n0, int n1, int pos1, int mode, int part) {
  if (mode == VLM_ATTN_SEPARATE) return ((part ? n1 : n0) + ATT_KB - 1) / ATT_KB;
  return (pos1 + n1 + ATT_KB - 1) / ATT_KB;
}

typedef __attribute__((address_space(3))) s16x4 dkvb_lds_s16x4;

typedef __attribute__((ext_vector_type(8))) float dkvb_f32x8;
// eight fp32 -> eight bf16 (round to nearest even) as FOUR v_cvt_pk_bf16_f32: the element-wise (bf16_t) casts compiled to one
// conversion per value plus v_perm re-packing.  (Not inline asm: hipcc does not pad the VALU-write -> MFMA-operand hazard of an
// asm statement -- cdna_hip_programming.md 5.7 item 2 -- and the first asm version of this returned wrong dK on some waves.)
__device__ __forceinline__ bf16x8 dkvb_pack8(const f32x16& v, int first) {
  dkvb_f32x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = v[first + i];
  return __builtin_convertvector(t, bf16x8);
}

template <int W>
__global__ __launch_bounds__(W * 64, 1) void attn_bwd_dkvb_kernel(const attn_bwd_params_t bp, int n_groups, int panel_blocks) {
  const attn_params_t& p = bp.f;
  extern __shared__ __attribute__((aligned(1024))) unsigned char dkvb_smem[];
  float* panel = reinterpret_cast<float*>(dkvb_smem);             // [nblk][4 g4][64 lanes][4]: G in accumulator order
  unsigned char* slots = dkvb_smem + (size_t)panel_blocks * 4096;  // [W][Q block 4 KB | dO block 4 KB]; the histogram at the end

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const att_pos_t ps = att_pos(p.seq);
  const int D = p.H * 64;
  // ---- work item: (head, key block, sample group); XCD-aware: the key blocks of one head share their Q / dO through one L2 ----
  const dkvb_geom_t gm = dkvb_geom(ps.n0, ps.n1, ps.pos1, p.mode);
  const int NKB = gm.nkb[0] + gm.nkb[1];
  const int total = NKB * p.H * n_groups, per = (total + 7) >> 3;
  const int logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || logical >= total) return;
  const int kbi = logical % NKB;
  const int grp = (logical / NKB) % n_groups;
  const int h = logical / (NKB * n_groups);
  const int part = kbi >= gm.nkb[0] ? 1 : 0;
  const int sb = part ? kbi - gm.nkb[0] : kbi;  // 32-position stationary block inside its part (att_dense_layout)
  int s_lo, s_hi;                                // streamed (query) positions
  if (p.mode == VLM_ATTN_SEPARATE) { s_lo = part ? ps.pos1 : 0; s_hi = part ? ps.NP : ps.n0; }
  else { s_lo = 0; s_hi = ps.NP; }
  const int kp0 = s_lo + sb * ATT_KB;
  const int nblk = (s_hi - s_lo + ATT_KB - 1) / ATT_KB;
  const int b_lo = (int)((long)ps.B * grp / n_groups), b_hi = (int)((long)ps.B * (grp + 1) / n_groups);
  const int wact = nblk < W ? nblk : W;  // waves that take samples (a short query range cannot keep W skewed waves apart)
  const int S = nblk / wact;             // >= 1
  const float c1 = p.scale * ATT_LOG2E;

  // ---- per-lane constants -------------------------------------------------------------------------------------------------
  const int kp = kp0 + r;  // this lane's key position
  const bool kvalid = kp < s_hi && (kp < ps.n0 || kp >= ps.pos1);
  const bool ktext = kp < ps.n0;
  const uint8_t* keepk = ktext ? p.keep0 : p.keep1;
  const int keep_at = ktext ? kp : kp - ps.pos1, keep_n = ktext ? ps.n0 : ps.n1;
  const int krow0 = ktext ? ps.base0 + kp : ps.base1 + (kp - ps.pos1), kstr = ktext ? ps.n0 : ps.n1;  // row of sample b: krow0 + b * kstr
  f16x8 sel0, sel1;
  att_select_frags(lane, sel0, sel1);
  const att_dense_layout_t dl = att_dense_layout(ps.n0, ps.n1, ps.pos1, p.mode);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(p.dense_t + (size_t)(p.head_row0 + h) * p.dense_tiles * 2048), 0, (uint32_t)p.dense_tiles * 4096u, 0x00020000);
  const uint32_t bvoff = att_bias_voff(dl, part, sb, lane);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv), 0, (uint32_t)((size_t)p.total_rows * p.ld_qkv * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(bp.d_o), 0, (uint32_t)((size_t)p.total_rows * bp.ld_do * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rlse = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(bp.lse + (size_t)h * p.total_rows), 0, (uint32_t)p.total_rows * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdel = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(bp.delta + (size_t)h * p.total_rows), 0, (uint32_t)p.total_rows * 4u, 0x00020000);

  // per-lane parts of the Q / dO load addresses (16-B chunk of the head's 128-B row segment; row (lane >> 3) of an 8-row group)
  const uint32_t c16 = (uint32_t)(lane & 7) * 16u + (uint32_t)h * 128u;
  const uint32_t voff_q = (uint32_t)(lane >> 3) * (uint32_t)(p.ld_qkv * 2) + c16, voff_o = (uint32_t)(lane >> 3) * (uint32_t)(bp.ld_do * 2) + c16;
  // LDS addresses of this lane inside a block image (the Q image; dO's is 4096 further)
  unsigned char* slot = slots + wave * 8192;
  // (row bits 4 and up do not enter the swizzle: 16 rows further = + 2048 bytes; 8 rows further flips chunk bit 1; chunk bit 2 /
  // the pair of chunks (1, 2) move under a plain XOR of the address)
  const uint32_t wr_off0 = dkvb_off(lane >> 3, lane & 7), wr_off1 = dkvb_off(8 + (lane >> 3), lane & 7);  // store rows 8u + (lane >> 3)
  const uint32_t row_off0 = dkvb_off(r, hh);  // row fragment of k-step ss: chunk 2 ss + hh -> row_off0 ^ (ss << 5)
  uint32_t tr_off[2][2];                      // [db][lo / hi] transposed fragment of rows 16 s2 + 8 hi + 4 hh + qq: + s2 * 2048
  {
    const int g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int hi = 0; hi < 2; ++hi)
        tr_off[db][hi] = dkvb_off(8 * hi + 4 * hh + qq, 2 * (db * 2 + g16) + (pp >> 1)) + 8 * (pp & 1);
  }

  // ---- zero the panel ---------------------------------------------------------------------------------------------------------
  for (int i = tid; i < nblk * 256; i += W * 64) reinterpret_cast<f32x4*>(panel)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // v_bias gradient: column sums of dV over this wave's keys, per token segment, kept across the wave's samples
  float cs_t[2] = {0.f, 0.f}, cs_i[2] = {0.f, 0.f};

  const int rounds = (b_hi - b_lo + wact - 1) / wact;
  for (int rd = 0; rd < rounds; ++rd) {
    const int b = b_lo + rd * wact + wave;
    const bool active = wave < wact && b < b_hi;  // wave-uniform

    // ---- this wave's sample: K (scaled by scale * log2 e), V fragments of the 32 keys; the key side of the statistics k-step ----
    bf16x8 kf[4], vf[4], bstat;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) dk[0][i] = dk[1][i] = dv[0][i] = dv[1][i] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) bstat[j] = (bf16_t)0.f;
    // next block's operands, requested one block ahead
    u32x4 nq[4], no_[4], nbw[2];
    float nlse = 0.f, ndel = 0.f;
    // Every lane ALWAYS loads: a position outside the streamed range (ragged last block, the text / image gap) is clamped to a
    // row that exists, its scores are switched off by the statistics k-step (-30000) and by the table's own padding, so whatever
    // finite Q / dO values arrive there are multiplied by P = 0.  (A per-lane `ok ? offset : out-of-range` made hipcc put every
    // load under an exec-mask branch of its own: ~120 scalar instructions per block and no counted vmcnt.)
    const uint32_t row_txt = (uint32_t)(ps.base0 + b * ps.n0), row_img = (uint32_t)(ps.base1 + b * ps.n1 - ps.pos1);
    auto row_of = [&](int qp) -> uint32_t {
      qp = qp < s_hi - 1 ? qp : s_hi - 1;
      return (uint32_t)qp + (qp < ps.n0 ? row_txt : row_img);
    };
    auto request = [&](int j) {
      const int q0 = s_lo + j * ATT_KB;
      // 17 of the 20 blocks of a 384^2 pass are 32 consecutive image rows: their addresses are a scalar base + per-lane
      // constants (no vector arithmetic at all); both paths issue the same loads in the same order
      if (q0 >= ps.pos1 && q0 + ATT_KB <= s_hi) {  // (wave-uniform)
        const uint32_t row0 = row_img + (uint32_t)q0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          nq[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, voff_q, (row0 + 8u * u) * (uint32_t)(p.ld_qkv * 2), 0));
          no_[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, voff_o, (row0 + 8u * u) * (uint32_t)(bp.ld_do * 2), 0));
        }
        nlse = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rlse, (uint32_t)r * 4u, row0 * 4u, 0));
        ndel = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdel, (uint32_t)r * 4u, row0 * 4u, 0));
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t row = row_of(q0 + 8 * u + (lane >> 3));
          nq[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, row * (uint32_t)(p.ld_qkv * 2) + c16, 0, 0));
          no_[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, row * (uint32_t)(bp.ld_do * 2) + c16, 0, 0));
        }
        const uint32_t row = row_of(q0 + r);  // both lane halves ask for the same 32 rows (the upper half's copy is not used)
        nlse = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rlse, row * 4u, 0, 0));
        ndel = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdel, row * 4u, 0, 0));
      }
      {
        const uint32_t soff = (uint32_t)(j >> 1) * 4096u + (uint32_t)(j & 1) * 2048u;
        nbw[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff, soff, 0));
        nbw[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff + 1024, soff, 0));
      }
    };
    if (active) {
      const size_t krow = (size_t)(krow0 + b * kstr);
      const bf16_t* kptr = p.qkv + (kvalid ? krow : (size_t)0) * p.ld_qkv + D + h * 64 + 8 * hh;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(kptr + 16 * s);
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[s][j] = (bf16_t)(kvalid ? (float)raw[j] * c1 : 0.f);
        vf[s] = *reinterpret_cast<const bf16x8*>(kptr + D + 16 * s);
      }
      const bool kept = kvalid && (!keepk || keepk[(size_t)b * keep_n + keep_at] != 0);
      if (hh == 0) {
        bstat[0] = bstat[1] = bstat[2] = (bf16_t)1.0f;
        bstat[3] = (bf16_t)(kept ? 0.f : ATT_NEG_BIG);  // dropped key: every score of its column goes to -30000, P = 0
      }
      request(wave * S);
    }

    if (!active) {  // (wave-uniform) an idle wave of this round only keeps the barriers company
      for (int t = 0, ph = 0; t < nblk; ++t) {
        if (++ph == S || t + 1 == nblk) { ph = 0; __syncthreads(); }
      }
      continue;
    }
    for (int t = 0, ph = 0, j = wave * S; t < nblk; ++t) {
      // ---- the block requested one step ago: Q / dO -> this wave's LDS slot (one image for row reads and transposed reads) ------
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t wo = (u & 1 ? wr_off1 : wr_off0) + (u >> 1) * 2048u;
        *reinterpret_cast<u32x4*>(slot + wo) = nq[u];
        *reinterpret_cast<u32x4*>(slot + 4096 + wo) = no_[u];
      }
      // ---- query side of the statistics k-step: (-lse, -delta) of row q0 + r as three bf16 terms each ------------------------
      bf16x8 astat_e, astat_d;
      {
        const int qp = s_lo + j * ATT_KB + r;
        const bool qok = qp < s_hi && (qp < ps.n0 || qp >= ps.pos1);
        const float ve = qok ? -nlse : ATT_NEG_BIG, vd = qok ? -ndel : 0.f;
        // two bf16 terms each: 16 significant bits -- an error of 2^-17 |lse| < 2e-4 in the exponent, 1e-4 relative in P,
        // against the 2^-9 of P's own rounding to bf16
        const bf16_t e0 = (bf16_t)ve, d0 = (bf16_t)vd;
        const bf16_t e1 = (bf16_t)(ve - (float)e0), d1 = (bf16_t)(vd - (float)d0);
        const bf16_t z = (bf16_t)0.f;
        const bool lo = hh == 0;
        astat_e = (bf16x8){lo ? e0 : z, lo ? e1 : z, z, lo ? (bf16_t)1.0f : z, z, z, z, z};
        astat_d = (bf16x8){lo ? d0 : z, lo ? d1 : z, z, z, z, z, z, z};
      }
      // ---- E = -lse + mask + Bias log2e + Q (c1 K)^T ;  dP = -delta + dO V^T --------------------------------------------------
      f32x16 e, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] = dp[i] = 0.f;
      e = att_bias_mfma(sel0, sel1, nbw, e);
      {  // the requested operands are consumed: the next block's requests go out into the same registers
        int jn = j + 1;
        if (jn >= nblk) jn -= nblk;
        request(t + 1 < nblk ? jn : j);  // (the last step re-requests its own block: same operations every step, result unused)
      }
      e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(astat_e, bstat, e, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(astat_d, bstat, dp, 0, 0, 0);
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(slot + (row_off0 ^ (uint32_t)(ss << 5)));
        const bf16x8 oa = *reinterpret_cast<const bf16x8*>(slot + 4096 + (row_off0 ^ (uint32_t)(ss << 5)));
        e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ss], e, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ss], dp, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        e[i] = att_exp2(e[i]);  // P
        dp[i] *= e[i];          // dS (natural units)
      }
      // ---- G += dS: this wave is the only one in this 4-KB block of the panel until the next barrier ----------------------------
      {
        float* g = panel + (size_t)j * 1024 + lane * 4;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 v = *reinterpret_cast<const f32x4*>(g + g4 * 256);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += dp[4 * g4 + i];
          *reinterpret_cast<f32x4*>(g + g4 * 256) = v;
        }
      }
      // ---- dV += P^T dO ,  dK += dS^T Q -------------------------------------------------------------------------------------------
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = dkvb_pack8(e, 8 * s2), df = dkvb_pack8(dp, 8 * s2);  // P and dS of rows 8 s2 .. 8 s2 + 7
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const s16x4 olo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dkvb_lds_s16x4*)(slot + 4096 + s2 * 2048 + tr_off[db][0]));
          const s16x4 ohi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dkvb_lds_s16x4*)(slot + 4096 + s2 * 2048 + tr_off[db][1]));
          const s16x8 ov = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, __builtin_bit_cast(bf16x8, ov), dv[db], 0, 0, 0);
          const s16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dkvb_lds_s16x4*)(slot + s2 * 2048 + tr_off[db][0]));
          const s16x4 qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dkvb_lds_s16x4*)(slot + s2 * 2048 + tr_off[db][1]));
          const s16x8 qv = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, __builtin_bit_cast(bf16x8, qv), dk[db], 0, 0, 0);
        }
      }
      if (++j >= nblk) j = 0;
      if (++ph == S || t + 1 == nblk) { ph = 0; __syncthreads(); }  // the waves move on to panel blocks another wave has just left
    }

    // ---- this sample's dK, dV: accumulator rows = keys (registers), column = d (lane & 31) -------------------------------------
    if (active) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kpi = kp0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const int row = kpi < s_hi ? att_row_of(ps, b, kpi) : -1;
        if (row >= 0) {
          bf16_t* dst = bp.dqkv + (size_t)row * bp.ld_dqkv + D + h * 64 + r;
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dst[db * 32] = (bf16_t)(dk[db][i] * p.scale);
            dst[D + db * 32] = (bf16_t)dv[db][i];
            if (kpi < ps.n0) cs_t[db] += dv[db][i];
            else cs_i[db] += dv[db][i];
          }
        }
      }
    }
  }

  // ---- v_bias gradient ----------------------------------------------------------------------------------------------------------
  if (bp.dv_colsum[0] || bp.dv_colsum[1]) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const float st = cs_t[db] + att_other_half(cs_t[db]), si = cs_i[db] + att_other_half(cs_i[db]);
      if (hh == 0) {
        if (bp.dv_colsum[0] && st != 0.f) atomicAdd(bp.dv_colsum[0] + h * 64 + db * 32 + r, st);
        if (bp.dv_colsum[1] && si != 0.f) atomicAdd(bp.dv_colsum[1] + h * 64 + db * 32 + r, si);
      }
    }
  }
  if (!bp.dbias_t) return;  // (workgroup-uniform)

  // ---- histogram of the panel through the index -> the table column's gradient ------------------------------------------------------
  // 64-bit FIXED-POINT bins as in attn_bwd_dbias16_kernel (integer LDS atomics run at the LDS array's rate; the sum does not
  // depend on the order the waves arrive in); the bins live in the slots' LDS, free by now.
  unsigned long long* hist64 = reinterpret_cast<unsigned long long*>(slots);
  unsigned* smax = reinterpret_cast<unsigned*>(hist64 + p.R);
  for (int i = tid; i < p.R; i += W * 64) hist64[i] = 0ull;
  if (tid == 0) *smax = 0u;
  __syncthreads();
  {
    float m = 0.f;
    for (int i = tid; i < nblk * 256; i += W * 64) {
      const f32x4 v = reinterpret_cast<const f32x4*>(panel)[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    if (lane == 0) atomicMax(smax, __float_as_uint(m));  // non-negative floats order like their bit patterns
  }
  __syncthreads();
  int ex;
  (void)frexpf(__uint_as_float(*smax), &ex);  // largest |value| < 2^ex
  if (ex < -60) ex = -60;
  const float FIX = ldexpf(1.0f, 46 - ex), UNFIX = ldexpf(1.0f, ex - 46);  // a bin receives at most 2^15 addends (1 024 queries x 32 keys)
  {
    const bool kin = kp < p.ld_idx;
    for (int n = wave; n < nblk * 4; n += W) {  // one (block, g4) group of 64 lanes x 4 queries per trip
      const int j = n >> 2, g4 = n & 3;
      const f32x4 v = *reinterpret_cast<const f32x4*>(panel + (size_t)j * 1024 + g4 * 256 + lane * 4);
      const int qq0 = s_lo + j * ATT_KB + 8 * g4 + 4 * hh;
      uint32_t ids[4];
      bool same = true;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qq = qq0 + e;
        ids[e] = (kin && qq < p.idx_rows) ? (uint32_t)(unsigned short)p.idx[(size_t)qq * p.ld_idx + kp] : 0u;
        same = same && ids[e] == ids[0];
      }
      const uint32_t first = __builtin_amdgcn_readfirstlane(ids[0]);
      same = same && ids[0] == first;
      if (__all(same)) {  // text <-> image pairs share ONE table row (vilt_module.py:180-181): reduce in registers
        float tsum = wave_sum(v[0] + v[1] + v[2] + v[3]);
        if (lane == 0) atomicAdd(hist64 + (first >> 2), (unsigned long long)(long long)(tsum * FIX));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)  // ids are byte offsets of 4-byte entries: 2 * ids addresses the 8-byte bin
          atomicAdd(reinterpret_cast<unsigned long long*>(slots + 2 * ids[e]), (unsigned long long)(long long)(v[e] * FIX));
      }
    }
  }
  __syncthreads();
  if (bp.dbias_part) {
    float* g = bp.dbias_part + (size_t)((kbi * p.H + h) * n_groups + grp) * p.R;
    for (int i = tid; i < p.R; i += W * 64) g[i] = (float)(long long)hist64[i] * UNFIX;
  } else {
    float* g = bp.dbias_t + (size_t)(p.head_row0 + h) * p.R;
    for (int i = tid; i < p.R; i += W * 64) {
      const long long v = (long long)hist64[i];
      if (v != 0) atomicAdd(g + i, (float)v * UNFIX);
    }
  }
}

// LDS bytes of a launch, or 0 when the panel does not fit beside W slots (and the R-bin histogram that later takes their place)
static inline size_t dkvb_lds_bytes(const attn_params_t& p, int W, int& panel_blocks) {
  const int a = dkvb_nblk(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode, 0), b = dkvb_nblk(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode, 1);
  panel_blocks = a > b ? a : b;
  size_t tail = (size_t)W * 8192, hist = (size_t)p.R * 8 + 16;
  if (hist > tail) tail = hist;
  const size_t need = (size_t)panel_blocks * 4096 + tail;
  return need <= 160 * 1024 ? need : 0;
}
