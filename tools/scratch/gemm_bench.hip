// Standalone timing harness for the 256x256 GEMM kernel (diagnostic, never part of the product library).
// Build (CPU container):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGEMM_ONLY_BIG [-DVLM_DIAG: in-kernel stamps] -I include \
//                           tools/scratch/gemm_bench.hip -o tools/scratch/gemm_bench[_variant]
// Run (GPU box):          tools/scratch/gemm_bench M N K [epilogue variant 0..4]
#include "../../vl-merging_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

extern "C" int vlm_device_cus(void) { return 256; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  // variant: 0 plain bf16, 1 bias + GELU + pre-activation copy (fc1 fwd), 2 GELU' + column sums (fc2 dgrad),
  //          5 GELU' factor saved by the forward pass (MUL_AUX) + column sums (fc2 dgrad as the engine runs it),
  //          3 f32 residual stream: bias, gamma, row scale, residual in place, branch copy (proj / fc2 fwd), 4 plain f32
  const int variant = argc > 4 ? atoi(argv[4]) : 0;
  const int f32 = variant == 3 || variant == 4;
  std::vector<uint16_t> h((size_t)(M > N ? M : N) * K);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; uint32_t u; memcpy(&u, &f, 4); v = u >> 16; }
  void *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
  vlm_epilogue_t e = {};
  e.alpha = 1.0f;
  void *aux = nullptr; float *vec = nullptr, *ws = nullptr;
  CK(hipMalloc(&aux, (size_t)M * N * 2)); CK(hipMemset(aux, 0, (size_t)M * N * 2));
  CK(hipMalloc(&vec, (size_t)(4 * N + M) * 4)); CK(hipMemset(vec, 0, (size_t)(4 * N + M) * 4));
  CK(hipMalloc(&ws, (size_t)(M / 128 + 2) * 2 * N * 4));
  if (variant == 1) { e.bias = vec; e.act = VLM_ACT_GELU; e.aux = aux; e.ld_aux = N; }
  if (variant == 2) { e.act = VLM_ACT_GELU_BWD; e.aux = aux; e.ld_aux = N; e.col_sum = vec + N; e.col_sum_ws = ws; }
  if (variant == 5) { e.act = VLM_ACT_MUL_AUX; e.aux = aux; e.ld_aux = N; e.col_sum = vec + N; e.col_sum_ws = ws; }  // fc2 dgrad with the saved GELU' factor
  if (variant == 3) { e.bias = vec; e.col_scale = vec + 2 * N; e.row_scale = vec + 4 * N; e.residual = (float*)nullptr; e.aux = aux; e.ld_aux = N; }
  if (variant == 3) { e.residual = (const float*)C; e.ld_res = N; }
  setenv("VLM_GEMM_BIG", "2", 1);
  auto run = [&]() {
    int rc = vlm_gemm_bf16(0, 0, M, N, K, A, K, B, K, C, N, f32, &e, 0);
    if (rc) { printf("launch failed rc=%d\n", rc); exit(1); }
  };
  for (int i = 0; i < 3; ++i) run();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < n; ++i) run();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / n, fl = 2.0 * M * N * K;
  const double tiles = (double)((M + 255) / 256) * ((N + 255) / 256), rounds = tiles / 256.0;
  printf("M=%d N=%d K=%d  %.1f us  %.1f TFLOP/s   tiles %.0f (%.2f rounds)  -> %.0f cycles @2.4GHz per 32-deep step per tile-round\n", M, N, K, us,
         fl / us / 1e6, tiles, rounds, us * 2400.0 / (K / 32.0) / (rounds < 1 ? 1 : rounds));
#ifdef VLM_DIAG
  {
    const int wgs = (int)tiles;
    unsigned long long* st; CK(hipMalloc(&st, (size_t)wgs * 64)); CK(hipMemset(st, 0, (size_t)wgs * 64));
    vlm_debug_set_stamp_buffer(st);
    run(); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs((size_t)wgs * 8);
    CK(hipMemcpy(hs.data(), st, (size_t)wgs * 64, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0, tot = 0, real = 0;
    int live = 0;
    for (int w = 0; w < wgs; ++w) live += hs[(size_t)w * 8 + 3] != 0;  // a persistent launch has fewer workgroups than tiles
    for (int w = 0; w < live; ++w) {
      const unsigned long long* q = &hs[(size_t)w * 8];
      pro += (double)(q[1] - q[0]); loop += (double)(q[2] - q[1]); epi += (double)(q[3] - q[2]); tot += (double)(q[3] - q[0]);
      real += (double)(q[5] - q[4]);
    }
    printf("stamps (avg per workgroup, s_memtime ticks): prologue %.0f  loop %.0f (%.0f per 32-deep step)  epilogue %.0f  total %.0f ; s_memrealtime ticks %.0f -> memtime/realtime = %.3f\n",
           pro / live, loop / live, loop / live / (K / 32.0), epi / live, tot / live, real / live, tot / real);
  }
#endif
  return 0;
}
