// dQ of the fused attention backward as a hand-placed instruction stream (round 6; included by attention_bwd.hip after
// attn_bwd_params_t).  Same operator and outputs as attn_bwd_dq_kernel (reference: autograd of vision_transformer.py:346-358): dQ,
// delta = rowsum(dO * O) and the C operands of the bias-gradient kernel from its prologue, the q_bias column sums from its
// epilogue.  It takes the calls that have a dense bias table and a geometry the stream covers (att_dq2_eligible).
//
// Structure (gen/attn_dq2_gen.py holds the register map and the stream): a wave = 32 query positions of one (sample, head), a
// workgroup = 128 positions, TWO workgroups per CU (252 registers per wave).  Keys stream in 32-key blocks through a ring of four
// 8-KiB LDS stages [K | V] in the image of attention_bwd_dkvb.h (dkvb_off: one K image for the row reads of the score chain and
// the transposed reads of the dQ products, conflict-free for both and for the LDS-DMA writes; the round-2 row image is 2-way on
// every ds_read_b128), filled by LDS-DMA two blocks ahead behind counted vmcnt waits.
// Per block a wave alternates an MFMA phase (the E and dP chains of the block, interleaved, + the previous block's dQ products: 16
// MFMAs with the LDS-DMA and bias requests in their gaps) and a vector phase (exp2, multiply, convert: 40 instructions, with the
// next fragments' LDS reads in flight), one s_barrier between them: the two waves of a SIMD can sit in opposite phases.
// -lse and -delta enter through a statistics k-step each (three bf16 terms: 24 bits of the fp32 value; the key mask rides in the
// E step's k-slot 2), so no 16-register C tuples are held: the round-2 kernel's 242 registers were what kept its waves from
// running ahead of each other.
#pragma once

#define DQ2_STAGE 8192
#define DQ2_KM (4 * DQ2_STAGE)
#define DQ2_LDS (DQ2_KM + 1024)

typedef __attribute__((ext_vector_type(16))) unsigned dq2_u32x16;
typedef __attribute__((ext_vector_type(8))) unsigned dq2_u32x8;

// v88..v175 (E, dP, dS, fragments), v184..v187 (statistics key side), v196..v201 (temporaries) belong to the stream
#define DQ2_CLOBBER_V \
  "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", \
  "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",       \
  "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134",       \
  "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149",       \
  "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v184", "v185", "v186", "v187",       \
  "v196", "v197", "v198", "v199", "v200", "v201"

// x as three bf16 terms (hi, mid, lo): hi + mid + lo == x to 24 bits
__device__ __forceinline__ void dq2_split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const bf16_t h = (bf16_t)x;
  const float r1 = x - (float)h;
  const bf16_t m = (bf16_t)r1;
  const bf16_t l = (bf16_t)(r1 - (float)m);
  hi = (uint32_t)__builtin_bit_cast(unsigned short, h);
  mid = (uint32_t)__builtin_bit_cast(unsigned short, m);
  lo = (uint32_t)__builtin_bit_cast(unsigned short, l);
}

__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dq2_kernel(const attn_bwd_params_t bp) {
  const attn_params_t& p = bp.f;
  __shared__ __attribute__((aligned(16))) unsigned char lds[DQ2_LDS > 128 * 68 * 4 ? DQ2_LDS : 128 * 68 * 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const att_pos_t ps = att_pos(p.seq);
  int wtile, b, h;
  if (!att_work_item(att_num_tiles(ps.n0, ps.n1, ps.pos1, p.mode), ps.B, p.H, wtile, b, h)) return;
  const att_span_t sp = att_span(ps, p.mode, wtile);
  ATT_STAMP(0);
  const int D = p.H * 64;
  const int ntiles = (sp.s_hi - sp.s_lo + ATT_BK - 1) / ATT_BK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(att_lds_void*)lds;
  const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv), 0, (uint32_t)((size_t)p.total_rows * p.ld_qkv * 2), 0x00020000);
  const att_dense_layout_t dl = att_dense_layout(ps.n0, ps.n1, ps.pos1, p.mode);
  const uint32_t bvoff = att_bias_voff(dl, sp.part, sp.tile_in_part * 4 + wave, lane);
  const uint32_t rb_bytes = (uint32_t)p.dense_tiles * 4096u;
  const _Float16* bias_col = p.dense + (size_t)(p.head_row0 + h) * p.dense_tiles * 2048;
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(bias_col), 0, rb_bytes, 0x00020000);

  // ---- prologue (1): EVERY load first -- plain loads (bias operands of blocks 0 and 1, Q, dO, O, lse, the keep byte) ahead of
  // the LDS-DMA pieces: hipcc waits vmcnt(0) at the first use of a plain load's result while an LDS-DMA is in flight, so a load
  // issued after the pieces costs its own round trip behind theirs (12.7k cycles of prologue measured that way, three trips)
  u32x4 b0w[2], b1w[2];
  b0w[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff, 0, 0));
  b0w[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff + 1024, 0, 0));
  b1w[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff + 2048, 0, 0));
  b1w[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff + 3072, 0, 0));
  const int qp = sp.p0 + wave * 32 + r;
  const int qrow_raw = qp < sp.s_hi ? att_row_of(ps, b, qp) : -1;
  const bool qvalid = qrow_raw >= 0;
  const size_t qrow = qvalid ? (size_t)qrow_raw : (size_t)att_row_of(ps, b, sp.s_lo);
  bf16x8 rawq[4], dof[4], rawo[4];
  {
    const bf16_t* qptr = p.qkv + qrow * p.ld_qkv + h * 64 + 8 * hh;
    const bf16_t* dptr = bp.d_o + qrow * bp.ld_do + h * 64 + 8 * hh;
    const bf16_t* optr = bp.o + qrow * bp.ld_o + h * 64 + 8 * hh;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      rawq[ss] = *reinterpret_cast<const bf16x8*>(qptr + 16 * ss);
      dof[ss] = *reinterpret_cast<const bf16x8*>(dptr + 16 * ss);
      rawo[ss] = *reinterpret_cast<const bf16x8*>(optr + 16 * ss);
    }
  }
  const size_t stat_at = (size_t)h * p.total_rows + qrow;
  const float lse_raw = bp.lse[stat_at];
  // key position of this thread's mask word (tiles 0 and 1) and its keep byte
  const int mpos = sp.s_lo + (tid >> 6) * ATT_BK + (tid & 63);
  const bool mtxt = mpos < ps.n0, mimg = mpos >= ps.pos1 && mpos < ps.NP;
  const bool mok = tid < 128 && (mtxt || mimg) && mpos < sp.s_hi;
  // (through a descriptor, out of range where there is nothing to read: a load under a branch would be waited for inside it)
  const __amdgpu_buffer_rsrc_t rkeep0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.keep0 ? p.keep0 : reinterpret_cast<const uint8_t*>(p.qkv)), 0, p.keep0 ? (uint32_t)(ps.B * ps.n0) : 0u, 0x00020000);
  const uint32_t keep_raw = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rkeep0, (mok && mtxt) ? (uint32_t)(b * ps.n0 + mpos) : 0xFFFFFFF0u, 0, 0);

  // ---- prologue (2): blocks 0..2 by LDS-DMA (this wave's 8 rows of each of the three images) -----------------------------
  const uint32_t drow = (uint32_t)wave * 8u + (uint32_t)(lane >> 3), c16 = lane & 7;
  // LDS-DMA writes linearly (lane i -> 16 B at base + 16 i): the image's swizzle goes on the SOURCE chunk
  const uint32_t fsw = ((drow >> 2) & 3u) | (((drow >> 1) & 1u) << 2);
  const uint32_t voff = (drow * p.ld_qkv + ((c16 ^ fsw) << 3)) * 2u;
  auto stage_block = [&](int blk) {
    const int kp0 = sp.s_lo + blk * 32, pq = kp0 + (int)drow;
    const bool txt = pq < ps.n0, img = pq >= ps.pos1 && pq < ps.NP;
    const bool ok = (txt || img) && pq < sp.s_hi;
    const int first = txt ? ps.base0 + b * ps.n0 + kp0 : ps.base1 + b * ps.n1 + (kp0 - ps.pos1);
    const uint32_t rowoff = (uint32_t)first * (uint32_t)p.ld_qkv * 2u;
    const uint32_t kcol = (uint32_t)(D + h * 64) * 2u, vcol = (uint32_t)(2 * D + h * 64) * 2u;
    unsigned char* dst = lds + (blk & 3) * DQ2_STAGE + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rkv, (att_lds_void*)dst, 16, ok ? rowoff + kcol + voff : 0xFFFFFFF0u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rkv, (att_lds_void*)(dst + 4096), 16, ok ? rowoff + vcol + voff : 0xFFFFFFF0u, 0, 0, 0);
  };
  stage_block(0);
  stage_block(1);
  stage_block(2);

  // ---- prologue (3): consume -- mask words, Q / dO fragments, delta, the statistics operands ---------------------------------
  {
    // words of the statistics step's k-slots 2, 3 on the key side: (mask, 1): tiles 0 and 1 from the keep mask / the segment
    // limits, then 64 "no mask" words, then 64 zero words (the upper half-wave's k-slots are not used)
    uint32_t w = 0u;
    if (tid < 128) w = ((mok && !(mtxt && p.keep0 && keep_raw == 0)) ? 0u : 0xC6EAu) | 0x3F800000u;
    else if (tid < 192) w = 0x3F800000u;
    *reinterpret_cast<uint32_t*>(lds + DQ2_KM + tid * 4) = w;
  }
  const float c1 = p.scale * ATT_LOG2E;
  dq2_u32x16 qv, dov;
#pragma unroll
  for (int ss = 0; ss < 4; ++ss) {
    bf16x8 sc;
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = (bf16_t)((float)rawq[ss][j] * c1);
    const u32x4 wq = __builtin_bit_cast(u32x4, sc), wd = __builtin_bit_cast(u32x4, dof[ss]);
#pragma unroll
    for (int e = 0; e < 4; ++e) { qv[4 * ss + e] = wq[e]; dov[4 * ss + e] = wd[e]; }
  }
  dq2_u32x8 stats;
  {
    float part = 0.f;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss)
#pragma unroll
      for (int j = 0; j < 8; ++j) part += (float)rawo[ss][j] * (float)dof[ss][j];
    part += att_other_half(part);
    const size_t at = stat_at;
    const float lse2 = qvalid ? lse_raw : 30000.0f;  // invalid rows: E = -30 000, P = 0
    if (qvalid && hh == 0) {
      bp.delta[at] = part;
      if (bp.nstat) {
        bp.nstat[at] = -lse2 * bp.inv_c1;
        bp.nstat[(size_t)p.H * p.total_rows + at] = -part;
      }
    }
    uint32_t lh, lm, ll, dh, dm, dlw;
    dq2_split3(-lse2, lh, lm, ll);
    dq2_split3(qvalid ? -part : 0.f, dh, dm, dlw);
    const bool lo = hh == 0;
    stats[0] = lo ? (lh | (lm << 16)) : 0u;      // query side of E's statistics step: k-slots (-lse hi, -lse mid),
    stats[1] = lo ? (0x3F80u | (ll << 16)) : 0u; //   (1 [times the key's mask], -lse lo)
    stats[2] = 0u; stats[3] = 0u;
    stats[4] = lo ? (dh | (dm << 16)) : 0u;      // dP's: (-delta hi, -delta mid), (0, -delta lo)
    stats[5] = lo ? (dlw << 16) : 0u;
    stats[6] = 0u; stats[7] = 0u;
  }
  dq2_u32x16 ad;
  dq2_u32x8 cs;
#pragma unroll
  for (int ss = 0; ss < 4; ++ss) ad[ss] = lds0 + dkvb_off(r, 2 * ss + hh);   // row fragment of k-step ss (V image: + 4096)
  {
    const int g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int hi = 0; hi < 2; ++hi)  // transposed fragment of rows 16 s2 + 8 hi + 4 hh + qq (+ 2048 s2), 16-column group 2 db + g16
        ad[4 + 2 * db + hi] = lds0 + dkvb_off(8 * hi + 4 * hh + qq, 2 * (db * 2 + g16) + (pp >> 1)) + 8 * (pp & 1);
  }
  const uint32_t zword = lds0 + DQ2_KM + 768 + r * 4;
  ad[8] = hh == 0 ? lds0 + DQ2_KM + r * 4 : zword;          // mask words of tile 0
  ad[9] = hh == 0 ? lds0 + DQ2_KM + 512 + r * 4 : zword;    // "no mask"
  ad[10] = voff; ad[11] = drow; ad[12] = bvoff;
  ad[13] = hh == 0 ? lds0 + DQ2_KM + 256 + r * 4 : zword;   // mask words of tile 1
  ad[14] = 0xFFFFFFF0u;
  ad[15] = 0u;
  {
    f16x8 sel0, sel1;
    att_select_frags(lane, sel0, sel1);
    const u32x4 a0 = __builtin_bit_cast(u32x4, sel0), a1 = __builtin_bit_cast(u32x4, sel1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { cs[e] = a0[e]; cs[4 + e] = a1[e]; }
  }
  const uint64_t qa = (uint64_t)(uintptr_t)p.qkv;
  const u32x4 rkv4 = {(uint32_t)qa, (uint32_t)(qa >> 32) & 0xffffu, (uint32_t)((size_t)p.total_rows * p.ld_qkv * 2), 0x00020000u};
  const uint64_t ba = (uint64_t)(uintptr_t)bias_col;
  const u32x4 rb4 = {(uint32_t)ba, (uint32_t)(ba >> 32) & 0xffffu, rb_bytes, 0x00020000u};
  dq2_u32x8 bwv, bw1v;
#pragma unroll
  for (int e = 0; e < 4; ++e) { bwv[e] = b0w[0][e]; bwv[4 + e] = b0w[1][e]; bw1v[e] = b1w[0][e]; bw1v[4 + e] = b1w[1][e]; }
  dq2_u32x8 sc;
  {
    const int kp3 = sp.s_lo + 96;  // the stream's first own block
    const uint32_t first = (uint32_t)(ps.base1 + b * ps.n1 + (kp3 - ps.pos1));
    sc[0] = (first * (uint32_t)p.ld_qkv + (uint32_t)(D + h * 64)) * 2u;
    sc[1] = (first * (uint32_t)p.ld_qkv + (uint32_t)(2 * D + h * 64)) * 2u;
    sc[2] = (uint32_t)(2 * ntiles);
    sc[3] = (uint32_t)(sp.s_hi - kp3);
    sc[4] = lds0 + (uint32_t)wave * 1024u;
    sc[5] = (uint32_t)p.ld_qkv * 64u;
    sc[6] = 0u; sc[7] = 0u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  ATT_STAMP(20);
  f32x16 o0, o1;
  dq2_u32x16 ad_out;
  dq2_u32x8 sc_out, bw_out, bw1_out;
  int tid2 = tid;  // every per-lane value the epilogue needs is recomputed from the copy that comes out of the statement
  asm volatile(
#include "attention_dq2_body.inc"
      : "={a[0:15]}"(o0), "={a[16:31]}"(o1), "={v[176:183]}"(bw_out), "={s[48:55]}"(sc_out), "={v[32:47]}"(ad_out), "+v"(tid2),
        "={v[160:167]}"(bw1_out)
      : "{v[48:55]}"(cs), "{v[56:71]}"(qv), "{v[72:87]}"(dov), "{v[188:195]}"(stats), "2"(bwv), "3"(sc), "4"(ad), "6"(bw1v),
        "{s[40:43]}"(rkv4), "{s[44:47]}"(rb4)
      : DQ2_CLOBBER_V, "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "vcc", "scc", "memory");

  ATT_STAMP(50);
  // ---- epilogue: dQ = scale * acc; the q_bias column sums through LDS (as attn_bwd_dq_kernel) ------------------------------
  const int lane2 = tid2 & 63, wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6), r2 = lane2 & 31, hh2 = lane2 >> 5;
  const int qp2 = sp.p0 + wave2 * 32 + r2;
  const int raw2 = qp2 < sp.s_hi ? att_row_of(ps, b, qp2) : -1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    o0[i] *= p.scale;
    o1[i] *= p.scale;
  }
  if (raw2 >= 0) {
    bf16_t* op = bp.dqkv + (size_t)raw2 * bp.ld_dqkv + h * 64 + 4 * hh2;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (bf16_t)(db ? o1 : o0)[4 * g4 + i];
        *reinterpret_cast<bf16x4*>(op + db * 32 + 8 * g4) = v;
      }
  }
  if (bp.dq_colsum[0] || bp.dq_colsum[1]) {
    __syncthreads();  // every wave is done with the ring
    float* red = reinterpret_cast<float*>(lds);
    const int lr = wave2 * 32 + r2;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (raw2 >= 0) v = (f32x4){(db ? o1 : o0)[4 * g4], (db ? o1 : o0)[4 * g4 + 1], (db ? o1 : o0)[4 * g4 + 2], (db ? o1 : o0)[4 * g4 + 3]};
        *reinterpret_cast<f32x4*>(red + lr * 68 + db * 32 + 8 * g4 + 4 * hh2) = v;
      }
    __syncthreads();
    const int col = tid2 & 63, part = tid2 >> 6;
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
  ATT_STAMP(51);
}

// a dense bias table, no image keep mask (mask words exist for tiles 0 and 1; the stream's own loads start at position 96 with
// image rows: the text segment and the gap must end inside tile 0), 32-bit buffer offsets
static bool att_dq2_eligible(const attn_params_t& p) {
  if (!p.dense || p.keep1) return false;
  if (p.seq.pos1 > ATT_BK) return false;
  return true;
}

// 1 = launched, 0 = not a call for this kernel
static int att_dq2_launch(const attn_bwd_params_t& bp, dim3 grid, hipStream_t s) {
  static const int enabled = [] { const char* e = getenv("VLM_ATT_DQ2"); return e ? atoi(e) : 1; }();
  if (!enabled || !att_dq2_eligible(bp.f)) return 0;
  // diagnostic: VLM_ATT_DQ2_LDS_PAD bytes of unused dynamic LDS leave room for fewer workgroups per CU (docs/experiments.md, round 6)
  static const int lds_pad = [] { const char* e = getenv("VLM_ATT_DQ2_LDS_PAD"); return e ? atoi(e) : 0; }();
  hipLaunchKernelGGL(attn_bwd_dq2_kernel, grid, dim3(ATT_THREADS), (size_t)lds_pad, s, bp);
  return 1;
}
