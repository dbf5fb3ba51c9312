// semantics probe for v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950 (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  unsigned x = threadIdx.x, y = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 4) printf(" %u", h[a * 64 + i]); printf("\n"); }
  return 0;
}
