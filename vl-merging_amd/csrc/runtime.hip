// Library identity + device queries (host only).
#include "vlm_common.h"

extern "C" int vlm_abi_version(void) { return VLM_ABI_VERSION; }

extern "C" int vlm_device_cus(void) {
  static int cached = 0;  // benign race: every thread computes the same value
  if (cached > 0) return cached;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return VLM_ERR_LAUNCH;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return VLM_ERR_LAUNCH;
  cached = cus;
  return cus;
}
