// sn_launch.h -- host-side launch hygiene shared by the launchers: device properties and per-kernel attributes are queried /
// set ONCE per (device, kernel) instead of on every launch (hipDeviceGetAttribute + hipFuncSetAttribute cost a few
// microseconds each -- invisible next to a 170 ms frame, not next to the ~1 ms launches of a mixed-precision training step),
// and nothing here synchronises or allocates, so every launcher stays capturable in a HIP graph.
#pragma once
#include <hip/hip_runtime.h>

namespace snh {

constexpr int MAX_DEVICES = 64;

inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
  return dev;
}

// compute units of the current device (persistent kernels launch one workgroup per CU)
inline int cu_count() {
  static int cached[MAX_DEVICES];              // 0 = not queried yet (benign race: every thread writes the same value)
  const int dev = current_device();
  if (dev < 0) return 256;
  int v = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (v == 0) {
    v = 256;
    (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    __atomic_store_n(&cached[dev], v, __ATOMIC_RELAXED);
  }
  return v;
}

}  // namespace snh

// Raise the dynamic-LDS limit of kernel KFN_ to LDS_ bytes once per device.  Expands to a block with its own static
// high-water marks, so every call site (= every kernel instantiation) is tracked separately.  `return`s the hipError_t
// as int from the enclosing launcher on failure.
#define SN_ENSURE_DYN_LDS(KFN_, LDS_)                                                                               \
  do {                                                                                                              \
    static int sn_lds_set_[snh::MAX_DEVICES];                                                                       \
    const int sn_dev_ = snh::current_device();                                                                      \
    if (sn_dev_ < 0 || __atomic_load_n(&sn_lds_set_[sn_dev_], __ATOMIC_RELAXED) < (int)(LDS_)) {                    \
      hipError_t sn_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(KFN_),                                   \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_));              \
      if (sn_e_ != hipSuccess) return (int)sn_e_;                                                                   \
      if (sn_dev_ >= 0) __atomic_store_n(&sn_lds_set_[sn_dev_], (int)(LDS_), __ATOMIC_RELAXED);                     \
    }                                                                                                               \
  } while (0)
