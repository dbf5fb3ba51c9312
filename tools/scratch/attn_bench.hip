// Standalone timing harness for the attention kernels (diagnostic, never part of the product library).
// Build (CPU container):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DVLM_DIAG: in-kernel stamps] \
//                           -I include tools/scratch/attn_bench.hip -o tools/scratch/attn_bench[_variant]
// Run (GPU box):          tools/scratch/attn_bench [B] [mode 0 joint / 1 separate] [bias 0/1]
#include "../../vl-merging_amd/csrc/attention_fwd.hip"
#include "../../vl-merging_amd/csrc/attention_bwd.hip"
#include "../../vl-merging_amd/csrc/attention_fwd2.hip"
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <vector>

extern "C" int vlm_device_cus(void) { return 256; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 88, mode = argc > 2 ? atoi(argv[2]) : 0, with_bias = argc > 3 ? atoi(argv[3]) : 1;
  const int what = argc > 4 ? atoi(argv[4]) : 0;  // 0 fwd, 1 bwd
  const int with_dbias = argc > 5 ? atoi(argv[5]) : 1;  // 0: no bias-table gradient (dQ + dK/dV kernels only)
  const int n0 = 40, n1 = 577, H = 12, D = H * 64, pos1 = 40, NP = pos1 + n1, R = 2294, ncols = 144;
  const int rows = B * (n0 + n1);
  std::vector<uint16_t> hq((size_t)rows * 3 * D);
  uint32_t s = 12345;
  for (auto& v : hq) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.0f * 3.0f - 1.5f; uint32_t u; memcpy(&u, &f, 4); v = u >> 16; }
  void *qkv, *out, *dout, *dqkv; float *lse, *delta, *bias_t, *dbias; int16_t *idx, *idx_t; void *dense, *dense_t;
  CK(hipMalloc(&qkv, hq.size() * 2)); CK(hipMemcpy(qkv, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)rows * D * 2)); CK(hipMalloc(&dout, (size_t)rows * D * 2)); CK(hipMalloc(&dqkv, (size_t)rows * 3 * D * 2));
  CK(hipMemcpy(dout, hq.data(), (size_t)rows * D * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&lse, (size_t)H * rows * 4)); CK(hipMalloc(&delta, (size_t)H * rows * 4 + ((size_t)64 << 20)));
  std::vector<float> hb((size_t)ncols * R);
  for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  CK(hipMalloc(&bias_t, hb.size() * 4)); CK(hipMemcpy(bias_t, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dbias, hb.size() * 4)); CK(hipMemset(dbias, 0, hb.size() * 4));
  const int ld = (NP + 3) / 4 * 4;
  std::vector<int16_t> hi((size_t)NP * ld), hit((size_t)NP * ld);
  for (int q = 0; q < NP; ++q) for (int k = 0; k < NP; ++k) { int v = ((q * 131 + k * 7) % R) * 4; hi[(size_t)q * ld + k] = v; hit[(size_t)k * ld + q] = v; }
  CK(hipMalloc(&idx, hi.size() * 2)); CK(hipMemcpy(idx, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&idx_t, hi.size() * 2)); CK(hipMemcpy(idx_t, hit.data(), hi.size() * 2, hipMemcpyHostToDevice));
  const size_t cb = vlm_bias_dense_bytes(n0, n1, pos1, mode);
  CK(hipMalloc(&dense, cb * ncols)); CK(hipMalloc(&dense_t, cb * ncols));
  if (vlm_bias_dense(bias_t, ncols, R, idx, ld, n0, n1, pos1, mode, 0, dense, 0) || vlm_bias_dense(bias_t, ncols, R, idx, ld, n0, n1, pos1, mode, 1, dense_t, 0)) { printf("bias_dense failed\n"); return 1; }
  vlm_attn_desc_t d = {};
  d.qkv = qkv; d.ld_qkv = 3 * D; d.H = H; d.total_rows = rows; d.R = R; d.bias_t = with_bias ? bias_t : nullptr;
  d.rel_index = idx; d.rel_index_t = idx_t; d.ld_index = ld; d.index_rows = NP; d.ld_index_t = ld; d.index_t_rows = NP;
  d.head_row0 = 12; d.mode = mode; d.B = B; d.n0 = n0; d.n1 = n1; d.base0 = 0; d.base1 = B * n0; d.pos1 = pos1; d.scale = 0.125f;
  d.bias_dense = dense; d.bias_dense_t = dense_t; d.dense_tiles = (int)(cb / 4096);
  auto run = [&]() {
    int rc = what == 0 ? vlm_attention_fwd(&d, out, D, lse, 0)
                       : vlm_attention_bwd(&d, out, D, dout, D, lse, delta, vlm_attention_bwd_ws_floats(&d, 1), dqkv, 3 * D, with_bias && with_dbias ? dbias : nullptr, nullptr, 0);
    if (rc) { printf("launch failed rc=%d\n", rc); exit(1); }
  };
  if (what == 1) { int rc = vlm_attention_fwd(&d, out, D, lse, 0); if (rc) { printf("fwd rc=%d\n", rc); return 1; } }
  {
    int nb0 = -1, nb1 = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, attn_fwd_kernel<false>, 256, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, attn_fwd_kernel<true>, 256, 0);
    printf("occupancy (blocks/CU): nobias %d bias %d\n", nb0, nb1);
  }
  for (int i = 0; i < 3; ++i) run();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < n; ++i) run();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / n;
  const double nn = mode ? (double)n0 * n0 + (double)n1 * n1 : (double)(n0 + n1) * (n0 + n1);
  const double fl = 4.0 * B * H * 64 * nn * (what ? 2.5 : 1.0);
  printf("%s B=%d mode=%d bias=%d: %.1f us  %.0f TFLOP/s\n", what ? "bwd" : "fwd", B, mode, with_bias, us, fl / us / 1e6);
  if (what == 1 && with_bias && with_dbias) {  // checksum of one pass's bias-table gradient (compare builds)
    CK(hipMemset(dbias, 0, hb.size() * 4));
    run();
    CK(hipDeviceSynchronize());
    std::vector<float> g(hb.size());
    CK(hipMemcpy(g.data(), dbias, g.size() * 4, hipMemcpyDeviceToHost));
    double sum = 0, sabs = 0, w = 0;
    for (size_t i = 0; i < g.size(); ++i) { sum += g[i]; sabs += fabs(g[i]); w += g[i] * (double)((i * 2654435761u) % 1000); }
    printf("dbias checksum: sum %.6e abs %.6e weighted %.6e  [%g %g %g %g]\n", sum, sabs, w, g[12 * R + 5], g[13 * R + 700], g[20 * R + 1500], g[23 * R + 2200]);
  }
#ifdef VLM_DIAG
  {
    unsigned long long st[8 * 64];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(att_stamps), sizeof(st)));
    for (int w = 0; w < 8; w += 4) {
      printf("wave %d stamps (slot:delta to the previous stamped slot):", w);
      unsigned long long prev = 0;
      for (int i = 0; i < 60; ++i)
        if (st[w * 64 + i]) { if (prev) printf(" %d:%llu", i, st[w * 64 + i] - prev); prev = st[w * 64 + i]; }
      printf("\n");
      if (st[w * 64 + 61] > st[w * 64 + 60])
        printf("wave %d clock: %.3f GHz (slots 20..50: %llu shader cycles in %llu x 10 ns)\n", w,
               (double)(st[w * 64 + 50] - st[w * 64 + 20]) / ((double)(st[w * 64 + 61] - st[w * 64 + 60]) * 10.0),
               st[w * 64 + 50] - st[w * 64 + 20], st[w * 64 + 61] - st[w * 64 + 60]);
    }
  }
#endif
  return 0;
}
