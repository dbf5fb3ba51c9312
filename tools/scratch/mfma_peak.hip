// What a loop of nothing but independent MFMAs reaches on this chip, per instruction shape (diagnostic).
// hipcc -O3 --offload-arch=gfx950 tools/scratch/mfma_peak.hip -o tools/scratch/mfma_peak ; run: mfma_peak [waves per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 3 + j); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 3 + j); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same loop with 8 rotating operand pairs of random bits (normal-range bf16 values): what the DATA costs
template <int NACC>
__global__ __launch_bounds__(256) void k16r(float* out, int iters, const unsigned* rnd) {
  bf16x8 a[8], b[8];
  for (int q = 0; q < 8; ++q) {
    unsigned w[8];
    for (int j = 0; j < 8; ++j) w[j] = rnd[(threadIdx.x * 8 + q) * 8 + j];
    for (int j = 0; j < 8; ++j) {
      unsigned short ha = (unsigned short)((w[j] & 0x807f) | 0x3f00), hb = (unsigned short)(((w[j] >> 16) & 0x807f) | 0x3f00);
      a[q][j] = __builtin_bit_cast(__bf16, ha);
      b[q][j] = __builtin_bit_cast(__bf16, hb);
    }
  }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[(i >> 3) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k32r(float* out, int iters, const unsigned* rnd) {
  bf16x8 a[4], b[4];
  for (int q = 0; q < 4; ++q) {
    unsigned w[8];
    for (int j = 0; j < 8; ++j) w[j] = rnd[(threadIdx.x * 8 + q) * 8 + j];
    for (int j = 0; j < 8; ++j) {
      unsigned short ha = (unsigned short)((w[j] & 0x807f) | 0x3f00), hb = (unsigned short)(((w[j] >> 16) & 0x807f) | 0x3f00);
      a[q][j] = __builtin_bit_cast(__bf16, ha);
      b[q][j] = __builtin_bit_cast(__bf16, hb);
    }
  }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double run(F launch, double flop) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return flop * 5 / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? (atoi(argv[1]) > 4 ? 4 : atoi(argv[1])) : 4;
  float* out; CK(hipMalloc(&out, 256 * 512 * 4 * 4));
  const int iters = 20000, grid = 256;
  {
    const double fl = (double)grid * waves * iters * 32 * (2.0 * 16 * 16 * 32);
    printf("16x16x32, 32 accumulators (128 regs), %d waves/CU: %.0f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k16<32>, dim3(grid), dim3(64 * waves), 0, 0, out, iters); }, fl));
  }
  {
    const double fl = (double)grid * waves * iters * 8 * (2.0 * 32 * 32 * 16);
    printf("32x32x16,  8 accumulators (128 regs), %d waves/CU: %.0f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k32<8>, dim3(grid), dim3(64 * waves), 0, 0, out, iters); }, fl));
  }
  if (waves <= 4) {
    unsigned* rnd; CK(hipMalloc(&rnd, 256 * 64 * 4));
    unsigned hr[256 * 64]; unsigned st = 12345; for (auto& v : hr) { st = st * 1664525u + 1013904223u; v = st; }
    CK(hipMemcpy(rnd, hr, sizeof(hr), hipMemcpyHostToDevice));
    const double flr = (double)grid * waves * iters * 64 * (2.0 * 16 * 16 * 32);
    const double flq = (double)grid * waves * iters * 16 * (2.0 * 32 * 32 * 16);
    printf("32x32x16, 16 accumulators, 4x4 rotating RANDOM operands, %d waves/CU: %.0f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k32r<16>, dim3(grid), dim3(64 * waves), 0, 0, out, iters, rnd); }, flq));
    const double fl48 = (double)grid * waves * iters * 48 * (2.0 * 16 * 16 * 32);
    printf("16x16x32, 48 accumulators, rotating RANDOM operands, %d waves/CU: %.0f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k16r<48>, dim3(grid), dim3(64 * waves), 0, 0, out, iters, rnd); }, fl48));
    const double fl32 = (double)grid * waves * iters * 32 * (2.0 * 16 * 16 * 32);
    printf("16x16x32, 32 accumulators, rotating RANDOM operands, %d waves/CU: %.0f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k16r<32>, dim3(grid), dim3(64 * waves), 0, 0, out, iters, rnd); }, fl32));
    printf("16x16x32, 64 accumulators, 8x8 rotating RANDOM operands, %d waves/CU: %.0f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k16r<64>, dim3(grid), dim3(64 * waves), 0, 0, out, iters, rnd); }, flr));
    const double fl = (double)grid * waves * iters * 64 * (2.0 * 16 * 16 * 32);
    printf("16x16x32, 64 accumulators (256 regs), %d waves/CU: %.0f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k16<64>, dim3(grid), dim3(64 * waves), 0, 0, out, iters); }, fl));
    const double fl2 = (double)grid * waves * iters * 16 * (2.0 * 32 * 32 * 16);
    printf("32x32x16, 16 accumulators (256 regs), %d waves/CU: %.0f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k32<16>, dim3(grid), dim3(64 * waves), 0, 0, out, iters); }, fl2));
  }
  return 0;
}
