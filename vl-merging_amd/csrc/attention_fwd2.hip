// Fused attention forward, hand-placed instruction stream (round 6).  Same operator as attention_fwd.hip (reference
// vision_transformer.py:346-358 + get_rel_pos_bias vilt_module.py:1061-1064), same inputs and outputs; this is the kernel
// vlm_attention_fwd launches whenever the call has a dense bias table and a geometry it covers (att_fwd2_eligible), the
// round-2 kernel stays for everything else.
//
// Why another kernel.  The round-2/3 kernel (three 32-row waves per SIMD, compiler-scheduled) spends ~1 200 SIMD cycles per
// (32 key x 32 query) block for 416 cycles of MFMA: its waves take turns instead of overlapping (SQ_VALU_MFMA_COEXEC ~ 23 % of
// the busy cycles), and per score it moves 4 B into the CU (2 B of K/V at 128 queries per workgroup, 2 B of fp16 bias).
// Here ONE wave per SIMD owns the whole 512-entry register file and an explicit stream (gen/attn_fwd2_gen.py):
//   * a wave = 32 query positions of TWO samples of the same (head, 128-position tile): the bias operands of a (query block,
//     key block) pair are loaded once and serve both samples' selection MFMAs (1 B of bias per score instead of 2), and the two
//     samples' units alternate, so the exponentials of one sample issue in the gaps of the other sample's score chain;
//   * K / V of both samples arrive by LDS-DMA into a 4-deep ring, two tiles ahead of their use, behind counted vmcnt waits;
//     one s_barrier per 64-key tile;
//   * every MFMA gap carries a fixed set of vector / LDS / memory instructions placed by the generator, which also counts the
//     s_waitcnt values and pads the hazards hipcc does not pad inside an asm statement;
//   * the key-padding mask rides in the statistics k-step (k-slot 2: mask word on the key side against 1 on the query side).
//
// LDS (one workgroup per CU): 4 stages x 32 KiB [K sample 0 | K sample 1 | V sample 0 | V sample 1] (the row image and the
// transposed-read image of attention_common.h), then the mask words of tiles 0 and 1 [tile][sample][64 x u32] and 1 KiB of zeros.
// Registers: gen/attn_fwd2_gen.py (the map is repeated in the operand list of the asm statement below).
#include "vlm_common.h"
#include "attention_common.h"
#include "vlm_diag.h"
#include <stdlib.h>

#define F2_STAGE 16384
#define F2_NSTAGE 4
#define F2_KM (F2_NSTAGE * F2_STAGE)
#define F2_LDS (F2_KM + 2048)

typedef __attribute__((ext_vector_type(16))) unsigned u32x16;
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;

// v88..v159 (S, P, K / V fragments), v168..v179 (statistics operands) and v186..v191 (temporaries) belong to the stream;
// v0..v23 stay with the compiler (v24..v31: the second bias operand set, an operand of the statement)
#define F2_CLOBBER_V \
  "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", \
  "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", \
  "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", \
  "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", \
  "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", \
  "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v186", "v187", \
  "v188", "v189", "v190", "v191"

__global__ __launch_bounds__(ATT_THREADS, 2) void attn_fwd2_kernel(const attn_params_t p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[F2_LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const att_pos_t ps = att_pos(p.seq);
  const int npairs = (ps.B + 1) >> 1;
  int wtile, bp, h;
  if (!att_work_item(att_num_tiles(ps.n0, ps.n1, ps.pos1, p.mode), npairs, p.H, wtile, bp, h)) return;
  const att_span_t sp = att_span(ps, p.mode, wtile);
  ATT_STAMP(0);
  const int D = p.H * 64;
  const int bs[2] = {2 * bp, 2 * bp + 1 < ps.B ? 2 * bp + 1 : 2 * bp};  // an odd batch's last pair computes sample 0 twice
  const bool has1 = 2 * bp + 1 < ps.B;
  const int ntiles = (sp.s_hi - sp.s_lo + ATT_BK - 1) / ATT_BK;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(att_lds_void*)lds;
  const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv), 0, (uint32_t)((size_t)p.total_rows * p.ld_qkv * 2), 0x00020000);
  const att_dense_layout_t dl = att_dense_layout(ps.n0, ps.n1, ps.pos1, p.mode);
  const uint32_t bvoff = att_bias_voff(dl, sp.part, sp.tile_in_part * 4 + wave, lane);
  const uint32_t rb_bytes = (uint32_t)p.dense_tiles * 4096u;
  const _Float16* bias_col = p.dense + (size_t)(p.head_row0 + h) * p.dense_tiles * 2048;
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(bias_col), 0, rb_bytes, 0x00020000);

  // ---- prologue (1): everything that goes to memory first -- bias rows of tile 0, blocks 0 and 1 of both samples, sample 0's
  // pieces of block 2 (one LDS-DMA piece = this wave's 8 rows of a 32-key block), the mask words
  u32x4 b0w[2], b1w[2];  // key blocks 0 and 1's operands (two sets; the stream re-requests a set for the NEXT trip's block)
  b0w[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff, 0, 0));
  b0w[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff + 1024, 0, 0));
  b1w[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff + 2048, 0, 0));
  b1w[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, bvoff + 3072, 0, 0));
  const uint32_t drow = (uint32_t)wave * 8u + (uint32_t)(lane >> 3), c16 = lane & 7;
  // K: the image of attention_bwd_dkvb.h (chunk ^ f(row): conflict-free for the LDS-DMA writes AND ds_read_b128; the round-2 row
  // image chunk ^ (row & 7) is 2-way on every ds_read_b128, tools/lds_layout_check.py); LDS-DMA writes linearly, so the swizzle
  // goes on the SOURCE chunk
  const uint32_t fswK = ((drow >> 2) & 3u) | (((drow >> 1) & 1u) << 2);
  const uint32_t voffK = (drow * p.ld_qkv + ((c16 ^ fswK) << 3)) * 2u;
  const uint32_t voffV = (drow * p.ld_qkv + (((((c16 >> 1) ^ (((drow >> 1) & 1) << 1)) << 1) | (c16 & 1)) << 3)) * 2u;
  auto stage_block = [&](int blk, int s) {
    const int kp0 = sp.s_lo + blk * 32, pq = kp0 + (int)drow;
    const bool txt = pq < ps.n0, img = pq >= ps.pos1 && pq < ps.NP;
    const bool ok = (txt || img) && pq < sp.s_hi;
    const int first = txt ? ps.base0 + bs[s] * ps.n0 + kp0 : ps.base1 + bs[s] * ps.n1 + (kp0 - ps.pos1);
    const uint32_t rowoff = (uint32_t)first * (uint32_t)p.ld_qkv * 2u;
    unsigned char* dst = lds + (blk & 3) * F2_STAGE + s * 4096 + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rkv, (att_lds_void*)dst, 16, ok ? rowoff + (uint32_t)(D + h * 64) * 2u + voffK : 0xFFFFFFF0u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rkv, (att_lds_void*)(dst + 8192), 16, ok ? rowoff + (uint32_t)(2 * D + h * 64) * 2u + voffV : 0xFFFFFFF0u, 0, 0, 0);
  };
  // plain loads and their uses BEFORE the LDS-DMA pieces: hipcc waits vmcnt(0) at the first use of a plain load's result while an
  // LDS-DMA is in flight (a plain load issued after the pieces costs its own round trip behind theirs)
  // this thread's mask word: key position (tile tid >> 7, sample (tid >> 6) & 1, key tid & 63); its keep byte comes through a
  // descriptor, out of range where there is nothing to read (a load under a branch would be waited for inside it)
  const int ms = (tid >> 6) & 1, mpos = sp.s_lo + (tid >> 7) * ATT_BK + (tid & 63);
  const bool mtxt = mpos < ps.n0, mimg = mpos >= ps.pos1 && mpos < ps.NP;
  const bool mok = (mtxt || mimg) && mpos < sp.s_hi;
  const __amdgpu_buffer_rsrc_t rkeep0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.keep0 ? p.keep0 : reinterpret_cast<const uint8_t*>(p.qkv)), 0, p.keep0 ? (uint32_t)(ps.B * ps.n0) : 0u, 0x00020000);
  const uint32_t keep_raw = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(
      rkeep0, (mok && mtxt) ? (uint32_t)((ms ? bs[1] : bs[0]) * ps.n0 + mpos) : 0xFFFFFFF0u, 0, 0);
  const int qp = sp.p0 + wave * 32 + r;
  u32x16 qv[2];
  const float c1 = p.scale * ATT_LOG2E;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int raw = qp < sp.s_hi ? att_row_of(ps, bs[s], qp) : -1;
    const size_t qrow = raw >= 0 ? (size_t)raw : (size_t)att_row_of(ps, bs[s], sp.s_lo);
    const bf16_t* qptr = p.qkv + qrow * p.ld_qkv + h * 64 + 8 * hh;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const bf16x8 rawq = *reinterpret_cast<const bf16x8*>(qptr + 16 * ss);
      bf16x8 sc;
#pragma unroll
      for (int j = 0; j < 8; ++j) sc[j] = (bf16_t)((float)rawq[j] * c1);
      const u32x4 w = __builtin_bit_cast(u32x4, sc);
#pragma unroll
      for (int e = 0; e < 4; ++e) qv[s][4 * ss + e] = w[e];
    }
  }
  stage_block(0, 0);
  stage_block(0, 1);
  stage_block(1, 0);
  stage_block(1, 1);
  stage_block(2, 0);
  *reinterpret_cast<uint32_t*>(lds + F2_KM + tid * 4) = (mok && !(mtxt && p.keep0 && keep_raw == 0)) ? 0u : 0xC6EAu;  // bf16(-30 000) in k-slot 2
  *reinterpret_cast<uint32_t*>(lds + F2_KM + 1024 + tid * 4) = 0u;

  u32x16 ad;
  u32x8 cs;
#pragma unroll
  for (int ss = 0; ss < 4; ++ss) {
    const uint32_t fr = (((uint32_t)r >> 2) & 3u) | ((((uint32_t)r >> 1) & 1u) << 2);
    ad[ss] = lds0 + r * 128 + ((((uint32_t)(2 * ss + hh)) ^ fr) << 4);
  }
  {
    const int g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int c32 = (db * 2 + g16) ^ (((qq >> 1) & 1) << 1);
      ad[4 + db] = lds0 + (4 * hh + qq) * 128 + c32 * 32 + 8 * pp;
    }
  }
  const uint32_t zword = lds0 + F2_KM + 1024 + r * 4;
  ad[6] = hh == 0 ? lds0 + F2_KM + r * 4 : zword;
  ad[7] = zword;
  ad[8] = voffK; ad[9] = voffV; ad[10] = drow; ad[11] = bvoff;
  ad[12] = hh == 0 ? lds0 + F2_KM + 512 + r * 4 : zword;
  ad[13] = 0xFFFFFFF0u;
  ad[14] = 0u; ad[15] = 0x3F803F80u;  // (1, 1) in bf16: the row sums' v_dot2c operand
  {
    f16x8 sel0, sel1;
    att_select_frags(lane, sel0, sel1);
    const u32x4 a0 = __builtin_bit_cast(u32x4, sel0), a1 = __builtin_bit_cast(u32x4, sel1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { cs[e] = a0[e]; cs[4 + e] = a1[e]; }
  }
  // descriptor words for the stream's own buffer operations: uniform by construction, and SAID to be (readfirstlane) -- the
  // compiler may hold the base pointer in vector registers for the plain loads above and has no vector -> scalar copy
  const uint64_t qa = (uint64_t)(uintptr_t)p.qkv;
  const u32x4 rkv4 = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)qa),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(qa >> 32) & 0xffffu)),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((size_t)p.total_rows * p.ld_qkv * 2)), 0x00020000u};
  const uint64_t ba = (uint64_t)(uintptr_t)bias_col;
  const u32x4 rb4 = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ba),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(ba >> 32) & 0xffffu)),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)rb_bytes), 0x00020000u};
  u32x8 bwv, bw1v;
#pragma unroll
  for (int e = 0; e < 4; ++e) { bwv[e] = b0w[0][e]; bwv[4 + e] = b0w[1][e]; bw1v[e] = b1w[0][e]; bw1v[4 + e] = b1w[1][e]; }
  // the stream's scalar state: the DMA block is block 2 (its sample-1 pieces are the stream's first two)
  u32x8 sc;
  {
    const int kp2 = sp.s_lo + 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t first = (uint32_t)(ps.base1 + bs[s] * ps.n1 + (kp2 - ps.pos1));
      sc[s] = (first * (uint32_t)p.ld_qkv + (uint32_t)(D + h * 64)) * 2u;
      sc[2 + s] = (first * (uint32_t)p.ld_qkv + (uint32_t)(2 * D + h * 64)) * 2u;
    }
    sc[4] = (uint32_t)ntiles;
    sc[5] = (uint32_t)(sp.s_hi - kp2);
    sc[6] = lds0 + (uint32_t)wave * 1024u;
    sc[7] = (uint32_t)p.ld_qkv * 64u;
#pragma unroll
    for (int e = 0; e < 8; ++e) sc[e] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc[e]);  // uniform, and said to be
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  ATT_STAMP(20);
  f32x16 o00, o01, o10, o11;
  u32x16 ad_out;
  u32x8 sc_out, bw_out, bw1_out;
  float m_out[2], l_out[4];
  // the thread index goes THROUGH the statement: whatever the epilogue needs per lane (its rows, its validity) is recomputed
  // from the copy that comes out, so that no per-lane value has to stay in a register across the stream (the stream owns
  // v32..v191 and a0..a63 of the 256 registers a wave may have at two waves per SIMD)
  int tid2 = tid;
  asm volatile(
#include "attention_fwd2_body.inc"
      : "={a[0:15]}"(o00), "={a[16:31]}"(o01), "={a[32:47]}"(o10), "={a[48:63]}"(o11), "={v180}"(m_out[0]), "={v181}"(m_out[1]),
        "={v182}"(l_out[0]), "={v183}"(l_out[1]), "={v184}"(l_out[2]), "={v185}"(l_out[3]), "={v[160:167]}"(bw_out),
        "={s[48:55]}"(sc_out), "={v[32:47]}"(ad_out), "={v[24:31]}"(bw1_out), "+v"(tid2)
      : "{v[48:55]}"(cs), "{v[56:71]}"(qv[0]), "{v[72:87]}"(qv[1]), "10"(bwv), "11"(sc), "12"(ad), "13"(bw1v), "{s[40:43]}"(rkv4),
        "{s[44:47]}"(rb4)
      : F2_CLOBBER_V, "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "vcc", "scc", "memory");

  ATT_STAMP(50);
  // ---- epilogue -----------------------------------------------------------------------------------------------------------
  const int lane2 = tid2 & 63, wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6), hh2 = lane2 >> 5;
  const int qp2 = sp.p0 + wave2 * 32 + (lane2 & 31);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const f32x16& oa = s ? o10 : o00;
    const f32x16& ob = s ? o11 : o01;
    // a lane's 16 score registers are the keys of ITS half-wave (rows (i & 3) + 8 (i >> 2) + 4 hh of a block): the two halves'
    // partial sums meet here
    const float lh = l_out[2 * s] + l_out[2 * s + 1];
    const float lt = lh + __shfl_xor(lh, 32, 64);
    const float m = m_out[s];
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    const int raw = qp2 < sp.s_hi ? att_row_of(ps, bs[s], qp2) : -1;
    if (raw >= 0 && (s == 0 || has1)) {
      bf16_t* op = p.out + (size_t)raw * p.ld_out + h * 64 + 4 * hh2;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (bf16_t)((db ? ob : oa)[4 * g4 + e] * inv);
          *reinterpret_cast<bf16x4*>(op + db * 32 + 8 * g4) = v;
        }
      if (hh2 == 0 && p.lse) p.lse[(size_t)h * p.total_rows + raw] = m + log2f(lt);
    }
  }
  ATT_STAMP(51);
}

// The geometries the stream covers: a dense bias table, no image keep mask (its mask words exist for tiles 0 and 1 only, and its
// own loads start at position 64 with image rows: the text segment and the gap must end inside tile 0), 32-bit buffer offsets.
static bool att_fwd2_eligible(const attn_params_t& p) {
  if (!p.dense || p.keep1) return false;
  if (p.seq.pos1 > ATT_BK) return false;
  if ((size_t)p.total_rows * p.ld_qkv * 2 >= (1ull << 32)) return false;
  return true;
}

// returns 1 when it has launched the call, 0 when the call is not for this kernel, < 0 on error
int att_fwd2_launch(const attn_params_t& p, hipStream_t s) {
  static const int enabled = [] { const char* e = getenv("VLM_ATT_FWD2"); return e ? atoi(e) : 1; }();
  if (!enabled || !att_fwd2_eligible(p)) return 0;
  const int nt = att_num_tiles(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode);
  dim3 grid(att_grid_size(nt, (p.seq.B + 1) / 2, p.H)), block(ATT_THREADS);
  // diagnostic (docs/experiments.md, round 6): VLM_ATT_FWD2_LDS_PAD bytes of unused dynamic LDS leave room for ONE workgroup per CU
  static const int lds_pad = [] { const char* e = getenv("VLM_ATT_FWD2_LDS_PAD"); return e ? atoi(e) : 0; }();
  hipLaunchKernelGGL(attn_fwd2_kernel, grid, block, (size_t)lds_pad, s, p);
  if (hipGetLastError() != hipSuccess) return VLM_ERR_LAUNCH;
  return 1;
}
