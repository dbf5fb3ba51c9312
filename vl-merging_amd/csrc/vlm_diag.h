// vlm_diag.h -- everything diagnostic the kernel sources contain, behind ONE switch the product build never sets.
//
// -DVLM_DIAG is passed only by the standalone harnesses (tools/scratch/gemm_bench.hip, tools/scratch/attn_bench.hip,
// tools/stamp_gemm.py): in-kernel s_memtime / s_memrealtime stamps that say where a workgroup's cycles go.  Without it every
// macro below is empty, gemm_params_t carries no stamp pointer and no kernel executes a stamp.  The knock-out builds of
// rounds 1-3 (a kernel with one of its parts compiled out, results wrong) are gone from the sources: what they measured is
// recorded in DESIGN.md 4.1-4.3, and the code is in the git history of those rounds.
#pragma once

#ifdef VLM_DIAG

// ---- GEMM: 8 u64 per workgroup: [0..3] s_memtime at start / after the prologue / after the K loop / after the epilogue's stores
// were acknowledged, [4..5] s_memrealtime at start / end ------------------------------------------------------------------------
#define VLM_DIAG_GEMM_FIELD unsigned long long* stamps;
#define GEMM_STAMP(k)                                                                                                  \
  if (p.stamps && threadIdx.x == 0) {                                                                                  \
    p.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();                                             \
    if ((k) == 0 || (k) == 3) p.stamps[(size_t)blockIdx.x * 8 + 4 + (k) / 3] = __builtin_amdgcn_s_memrealtime();      \
  }
#define GEMM_STAMP_END()                                                   \
  __builtin_amdgcn_s_waitcnt(0); /* stores issued AND acknowledged */      \
  GEMM_STAMP(3)
static unsigned long long* g_vlm_diag_stamp_buffer = nullptr;
extern "C" void vlm_debug_set_stamp_buffer(void* ptr) { g_vlm_diag_stamp_buffer = (unsigned long long*)ptr; }
#define GEMM_STAMP_ARM(p) (p).stamps = g_vlm_diag_stamp_buffer;

// ---- attention: s_memtime of ONE workgroup's waves at up to 60 points of the tile loop -----------------------------------------
#ifndef ATT_STAMP_BLOCK
#define ATT_STAMP_BLOCK (8 * 100)
#endif
// slots 0..59: s_memtime (shader clock); slots 60 / 61: s_memrealtime (100 MHz) taken together with slots 20 / 50 -> the clock
__device__ unsigned long long att_stamps[16 * 64];
#define ATT_STAMP_DECL() [[maybe_unused]] int slot = 0;
#define ATT_STAMP(slot_expr)                                                                         \
  do {                                                                                               \
    const int s_ = (slot_expr);                                                                      \
    if (blockIdx.x == ATT_STAMP_BLOCK && lane == 0 && s_ < 60) {                                     \
      att_stamps[wave * 64 + s_] = __builtin_amdgcn_s_memtime();                                     \
      if (s_ == 20) att_stamps[wave * 64 + 60] = __builtin_amdgcn_s_memrealtime();                   \
      if (s_ == 50) att_stamps[wave * 64 + 61] = __builtin_amdgcn_s_memrealtime();                   \
    }                                                                                                \
  } while (0)

#else

#define VLM_DIAG_GEMM_FIELD
#define GEMM_STAMP(k)
#define GEMM_STAMP_END()
#define GEMM_STAMP_ARM(p)
#define ATT_STAMP_DECL()
#define ATT_STAMP(slot_expr) do { } while (0)

#endif
