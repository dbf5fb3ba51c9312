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

#define F2_STAGE 32768
#define F2_NSTAGE 4
#define F2_KM (F2_NSTAGE * F2_STAGE)
#define F2_LDS (F2_KM + 2048)

typedef __attribute__((ext_vector_type(16))) unsigned u32x16;
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;

// v96..v175 (S, P, K / V fragments) and v206..v223 (temporaries, current addresses) belong to the stream
#define F2_CLOBBER_V                                                                                                            \
  "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",   \
  "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126",       \
  "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141",       \
  "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156",       \
  "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171",       \
  "v172", "v173", "v174", "v175", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216",       \
  "v217", "v218", "v219", "v220", "v221", "v222", "v223"

__global__ __launch_bounds__(ATT_THREADS, 1) void attn_fwd2_kernel(const attn_params_t p) {
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

  const int qp = sp.p0 + wave * 32 + r;
  const int ntiles = (sp.s_hi - sp.s_lo + ATT_BK - 1) / ATT_BK;
  bool qvalid[2];
  size_t qrow[2];
  u32x16 qv[2];
  const float c1 = p.scale * ATT_LOG2E;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int raw = qp < sp.s_hi ? att_row_of(ps, bs[s], qp) : -1;
    qvalid[s] = raw >= 0;
    qrow[s] = qvalid[s] ? (size_t)raw : (size_t)att_row_of(ps, bs[s], sp.s_lo);
    const bf16_t* qptr = p.qkv + qrow[s] * p.ld_qkv + h * 64 + 8 * hh;
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

  const uint32_t lds0 = (uint32_t)(uintptr_t)(att_lds_void*)lds;
  const att_dma_t dk = att_dma_init<false>(p.ld_qkv, wave, lane), dv = att_dma_init<true>(p.ld_qkv, wave, lane);
  const att_dense_layout_t dl = att_dense_layout(ps.n0, ps.n1, ps.pos1, p.mode);
  const uint32_t bvoff = att_bias_voff(dl, sp.part, sp.tile_in_part * 4 + wave, lane);

  // ---- per-lane constants of the stream ---------------------------------------------------------------------------------
  u32x16 ad, cs;
#pragma unroll
  for (int ss = 0; ss < 4; ++ss) ad[ss] = lds0 + r * 128 + (((2 * ss + hh) ^ (r & 7)) << 4);
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
  ad[15] = hh == 0 ? lds0 + F2_KM + 512 + r * 4 : zword;
  ad[8] = dk.off[0]; ad[9] = dk.off[1]; ad[10] = dv.off[0]; ad[11] = dv.off[1];
  ad[12] = dk.row[0]; ad[13] = dk.row[1];
  ad[14] = bvoff;
  {
    f16x8 sel0, sel1;
    att_select_frags(lane, sel0, sel1);
    const u32x4 a0 = __builtin_bit_cast(u32x4, sel0), a1 = __builtin_bit_cast(u32x4, sel1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { cs[e] = a0[e]; cs[4 + e] = a1[e]; cs[8 + e] = 0x3F803F80u; cs[12 + e] = 0u; }
    cs[12] = 0xFFFFFFF0u;
  }
  const uint64_t qa = (uint64_t)(uintptr_t)p.qkv;
  const u32x4 rkv4 = {(uint32_t)qa, (uint32_t)(qa >> 32) & 0xffffu, (uint32_t)((size_t)p.total_rows * p.ld_qkv * 2), 0x00020000u};
  const uint64_t ba = (uint64_t)(uintptr_t)(p.dense + (size_t)(p.head_row0 + h) * p.dense_tiles * 2048);
  const u32x4 rb4 = {(uint32_t)ba, (uint32_t)(ba >> 32) & 0xffffu, (uint32_t)p.dense_tiles * 4096u, 0x00020000u};
  const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, rkv4[2], 0x00020000);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(p.dense + (size_t)(p.head_row0 + h) * p.dense_tiles * 2048), 0, rb4[2], 0x00020000);

  // ---- prologue: bias rows of tile 0, tiles 0 and 1 of both samples, sample 0's K pieces of tile 2, mask words -----------
  att_bias_t b0;
  att_bias_load(b0, rbias, bvoff, 0);
  u32x16 bwv;
#pragma unroll
  for (int e = 0; e < 4; ++e) { bwv[e] = b0.w[0][0][e]; bwv[4 + e] = b0.w[0][1][e]; bwv[8 + e] = b0.w[1][0][e]; bwv[12 + e] = b0.w[1][1][e]; }
  {
    const int t = tid >> 7, s = (tid >> 6) & 1, k = tid & 63;
    const float mk = att_key_mask(ps, bs[s], sp.s_lo + t * ATT_BK + k, sp.s_hi, p.keep0, p.keep1);
    *reinterpret_cast<uint32_t*>(lds + F2_KM + tid * 4) = mk < 0.f ? 0xC6EAu : 0u;  // bf16(-30 000) in k-slot 2
    *reinterpret_cast<uint32_t*>(lds + F2_KM + 1024 + tid * 4) = 0u;
  }
  auto stage_tile = [&](int t, int s, bool k_only) {
    const int kp0 = sp.s_lo + t * ATT_BK;
    unsigned char* dstK = lds + t * F2_STAGE + s * 8192;
    unsigned char* dstV = dstK + 16384;
    if (att_tile_plain(ps, kp0, sp.s_hi)) {
      att_dma_plain(rkv, dstK, dk, ps, bs[s], kp0, p.ld_qkv, D + h * 64, wave);
      if (!k_only) att_dma_plain(rkv, dstV, dv, ps, bs[s], kp0, p.ld_qkv, 2 * D + h * 64, wave);
    } else {
      att_dma_any(rkv, dstK, dk, ps, bs[s], kp0, sp.s_hi, p.ld_qkv, D + h * 64, wave);
      if (!k_only) att_dma_any(rkv, dstV, dv, ps, bs[s], kp0, sp.s_hi, p.ld_qkv, 2 * D + h * 64, wave);
    }
  };
  stage_tile(0, 0, false);
  stage_tile(0, 1, false);
  if (ntiles > 1) { stage_tile(1, 0, false); stage_tile(1, 1, false); }
  if (ntiles > 2) stage_tile(2, 0, true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- the stream's scalar state: the DMA tile is tile 2 ------------------------------------------------------------------
  u32x8 sc;
  {
    const int kp2 = sp.s_lo + 2 * ATT_BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t first = (uint32_t)(ps.base1 + bs[s] * ps.n1 + (kp2 - ps.pos1));
      sc[s] = (first * (uint32_t)p.ld_qkv + (uint32_t)(D + h * 64)) * 2u;
      sc[2 + s] = (first * (uint32_t)p.ld_qkv + (uint32_t)(2 * D + h * 64)) * 2u;
    }
    sc[4] = (uint32_t)ntiles;
    sc[5] = (uint32_t)(sp.s_hi - kp2);
    sc[6] = lds0 + 2 * F2_STAGE + (uint32_t)wave * 2048u;
    sc[7] = (uint32_t)p.ld_qkv * 128u;
  }

  ATT_STAMP(1);
  f32x16 o00, o01, o10, o11, l0, l1;
  u32x16 bw_out;
  float m_out[2];
  u32x8 sc_out;
  asm volatile(
#include "attention_fwd2_body.inc"
      : "={a[0:15]}"(o00), "={a[16:31]}"(o01), "={a[32:47]}"(o10), "={a[48:63]}"(o11), "={a[64:79]}"(l0), "={a[80:95]}"(l1),
        "={v204}"(m_out[0]), "={v205}"(m_out[1]), "={v[176:191]}"(bw_out), "={s[48:55]}"(sc_out)
      : "{v[32:47]}"(ad), "{v[48:63]}"(cs), "{v[64:79]}"(qv[0]), "{v[80:95]}"(qv[1]), "8"(bwv), "9"(sc),
        "{s[40:43]}"(rkv4), "{s[44:47]}"(rb4), "{s64}"(lds0)
      : F2_CLOBBER_V, "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s65", "vcc", "scc", "memory");

  ATT_STAMP(2);
  // ---- epilogue -----------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const f32x16& oa = s ? o10 : o00;
    const f32x16& ob = s ? o11 : o01;
    const float lt = s ? l1[0] : l0[0];
    const float m = m_out[s];
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    if (qvalid[s] && (s == 0 || has1)) {
      bf16_t* op = p.out + qrow[s] * p.ld_out + h * 64 + 4 * hh;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (bf16_t)((db ? ob : oa)[4 * g4 + e] * inv);
          *reinterpret_cast<bf16x4*>(op + db * 32 + 8 * g4) = v;
        }
      if (hh == 0 && p.lse) p.lse[(size_t)h * p.total_rows + qrow[s]] = m + log2f(lt);
    }
  }
  ATT_STAMP(3);
}

// The geometries the stream covers: a dense bias table, no image keep mask (its mask words exist for tiles 0 and 1 only: the
// text segment and the gap must end inside them), everything the 32-bit buffer offsets reach.
static bool att_fwd2_eligible(const attn_params_t& p) {
  if (!p.dense || p.keep1) return false;
  if (p.seq.pos1 > 2 * ATT_BK) return false;
  if ((size_t)p.total_rows * p.ld_qkv * 2 >= (1ull << 32)) return false;
  return true;
}

// returns 1 when it has launched the call, 0 when the call is not for this kernel, < 0 on error
int att_fwd2_launch(const attn_params_t& p, hipStream_t s) {
  static const int enabled = [] { const char* e = getenv("VLM_ATT_FWD2"); return e ? atoi(e) : 0; }();
  if (!enabled || !att_fwd2_eligible(p)) return 0;
  const int nt = att_num_tiles(p.seq.n0, p.seq.n1, p.seq.pos1, p.mode);
  dim3 grid(att_grid_size(nt, (p.seq.B + 1) / 2, p.H)), block(ATT_THREADS);
  hipLaunchKernelGGL(attn_fwd2_kernel, grid, block, 0, s, p);
  if (hipGetLastError() != hipSuccess) return VLM_ERR_LAUNCH;
  return 1;
}
