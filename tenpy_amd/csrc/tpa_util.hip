// Library/device info entry points.
#include "tpa_common.h"
#include <string.h>

extern "C" int tpa_device_info(char *name, int name_len, int *n_cu, int64_t *hbm_bytes) {
    int dev = 0;
    TPA_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    TPA_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return 0;
}
