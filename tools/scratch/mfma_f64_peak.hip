// What a loop of nothing but independent v_mfma_f64_16x16x4_f64 sustains on this chip (fp64 roofline check, round 5).
// hipcc -O3 --offload-arch=gfx950 tools/scratch/mfma_f64_peak.hip -o tools/scratch/mfma_f64_peak ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) double f64x4;
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD / 1) void k(double* out, int iters, double a0, double b0) {
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f64x4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* out; hipMalloc(&out, 1024 * 1024 * 8);
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int threads = 256 * wps, blocks = 256 * 4, iters = 4000;
    auto launch = [&]() {
      if (wps == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
      else if (wps == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
      else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
    };
    launch(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 5.0 * blocks * (threads / 64) * (double)iters * 8 * (16 * 16 * 4 * 2);
    printf("waves per SIMD %d: %.1f fp64 TFLOP/s\n", wps, flop / (ms * 1e-3) / 1e12);
  }
  return 0;
}
