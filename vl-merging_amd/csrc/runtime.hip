// Library identity + device queries (host only).
#include "vlm_common.h"
#include <atomic>
#include <stdlib.h>

extern "C" int vlm_abi_version(void) { return VLM_ABI_VERSION; }

// CU budget: what grid sizing and split-K slice counts take for "the CUs of this device".  By default the device's own
// count; VLM_GEMM_CUS=n in the environment (read once) or vlm_set_cu_budget(n) lowers it -- in a data-parallel job RCCL's
// kernels take CUs away from grids sized for all 256 (a wgrad launch of exactly one round of workgroups then runs 1 + epsilon
// rounds), so the first multi-GPU measurements can leave them room without a rebuild.  0 / negative: back to the device's count.
static std::atomic<int> g_cu_budget{0};

extern "C" int vlm_set_cu_budget(int cus) {
  g_cu_budget.store(cus > 0 ? cus : 0, std::memory_order_relaxed);
  return VLM_OK;
}

static int device_cus_raw(void) {
  static int cached = 0;  // benign race: every thread computes the same value
  if (cached > 0) return cached;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return VLM_ERR_LAUNCH;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return VLM_ERR_LAUNCH;
  cached = cus;
  return cus;
}

extern "C" int vlm_device_cus(void) {
  const int raw = device_cus_raw();
  if (raw <= 0) return raw;
  static const int env_budget = [] {
    const char* e = getenv("VLM_GEMM_CUS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 0;
  }();
  int b = g_cu_budget.load(std::memory_order_relaxed);
  if (b <= 0) b = env_budget;
  return (b > 0 && b < raw) ? b : raw;
}

// ---- contention stand-in (measurement tool): `workgroups` workgroups that hold their CU's resources (`lds_bytes` of
// LDS: with >= 48 KB a 256x256 GEMM workgroup no longer fits beside one) and spin for `microseconds`, as a single-GPU stand-in for
// the CUs a collective's kernels occupy during backward.  Launched by ddp.FlatGradReducer(standin=...) on the communication
// stream next to a copy of the bucket's bytes; never part of a training step otherwise.
__global__ void vlm_occupy_kernel(unsigned long long ticks) {
  extern __shared__ unsigned char occupy_lds[];
  if (threadIdx.x == 0 && ticks == ~0ull) occupy_lds[0] = 1;  // (keeps the allocation alive)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int vlm_debug_occupy(int workgroups, int threads, int lds_bytes, int microseconds, void* stream) {
  if (workgroups <= 0 || microseconds <= 0) return VLM_OK;
  if (threads <= 0 || threads > 1024 || (threads & 63) || lds_bytes < 0 || lds_bytes > 160 * 1024) return VLM_ERR_ARG;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)vlm_occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  hipLaunchKernelGGL(vlm_occupy_kernel, dim3(workgroups), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream,
                     (unsigned long long)microseconds * 100ull);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
