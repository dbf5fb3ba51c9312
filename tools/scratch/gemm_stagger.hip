// Stagger experiment for the 256x256 GEMM kernel (diagnostic, never part of the product library): the first round's
// workgroups start in P phases, `ticks` 100-MHz ticks apart, so that the CUs' epilogues (HBM-bound when every CU is in
// its epilogue at once) fall under other CUs' K loops.  All arms are timed interleaved in ONE process.
// Build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGEMM_ONLY_BIG -ffp-contract=off -I include tools/scratch/gemm_stagger.hip -o tools/scratch/gemm_stagger
#include "../../vl-merging_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>

extern "C" int vlm_device_cus(void) { return 256; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 54296, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  const int variant = argc > 4 ? atoi(argv[4]) : 0;
  const int f32 = variant >= 3;
  std::vector<uint16_t> h((size_t)(M > N ? M : N) * K);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; uint32_t u; memcpy(&u, &f, 4); v = u >> 16; }
  void *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
  CK(hipMemset(C, 0, (size_t)M * N * 4));
  vlm_epilogue_t e = {};
  e.alpha = 1.0f;
  void* aux = nullptr; float *vec = nullptr, *ws = nullptr;
  CK(hipMalloc(&aux, (size_t)M * N * 2)); CK(hipMemset(aux, 0, (size_t)M * N * 2));
  CK(hipMalloc(&vec, (size_t)(4 * N + M) * 4)); CK(hipMemset(vec, 0, (size_t)(4 * N + M) * 4));
  CK(hipMalloc(&ws, (size_t)(M / 128 + 2) * 2 * N * 4));
  if (variant == 1) { e.bias = vec; e.act = VLM_ACT_GELU_DERIV; e.aux = aux; e.ld_aux = N; }
  if (variant == 2) { e.act = VLM_ACT_MUL_AUX; e.aux = aux; e.ld_aux = N; e.col_sum = vec + N; e.col_sum_ws = ws; }
  if (variant == 3) { e.bias = vec; e.col_scale = vec + 2 * N; e.row_scale = vec + 4 * N; e.aux = aux; e.ld_aux = N; e.residual = (const float*)C; e.ld_res = N; }
  setenv("VLM_GEMM_BIG", "2", 1);
  auto run = [&]() {
    int rc = vlm_gemm_bf16(0, 0, M, N, K, A, K, B, K, C, N, f32, &e, 0);
    if (rc) { printf("launch failed rc=%d\n", rc); exit(1); }
  };
  struct arm_t { int phases, ticks; std::vector<float> us; };
  std::vector<arm_t> arms;
  arms.push_back({1, 0, {}});
  const int tick_list[] = {300, 600, 900, 1200, 1600, 2000, 2600, 3400};
  for (int t : tick_list) arms.push_back({2, t, {}});
  for (int t : {300, 500, 800, 1100}) arms.push_back({4, t, {}});
  for (int t : {150, 300, 500}) arms.push_back({8, t, {}});
  for (int i = 0; i < 5; ++i) run();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 7, n = 6;
  for (int r = 0; r < rounds; ++r)
    for (auto& a : arms) {
      g_stagger_phases = a.phases; g_stagger_ticks = a.ticks;
      run();
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < n; ++i) run();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      a.us.push_back(ms * 1e3f / n);
    }
  const double fl = 2.0 * M * N * K;
  const double tiles = (double)((M + 255) / 256) * ((N + 255) / 256);
  printf("M=%d N=%d K=%d variant %d  tiles %.0f (%.2f rounds)\n", M, N, K, variant, tiles, tiles / 256.0);
  for (auto& a : arms) {
    std::sort(a.us.begin(), a.us.end());
    const float med = a.us[a.us.size() / 2];
    printf("  phases %d  step %5.1f us : median %7.1f us  min %7.1f  %7.1f TFLOP/s  (vs base %+.1f %%)\n", a.phases, a.ticks / 100.0, med, a.us[0],
           fl / med / 1e6, 100.0 * (med / arms[0].us[arms[0].us.size() / 2] - 1.0));
  }
  return 0;
}
