// Library identification (include/genpose_hip.h).
#include <string.h>

#include "gp_common.h"

#ifdef GP_TIMING
__device__ unsigned long long gp_dbg_ts[8 * 32];
__device__ unsigned long long gp_dbg_wg[1024 * 4];
#endif

extern "C" {

/* tuning builds only: copy the 4x32 phase timestamps of the last instrumented kernel to the host */
int gp_debug_timestamps(unsigned long long *out) {
#ifdef GP_TIMING
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gp_dbg_ts), sizeof(unsigned long long) * 256) == hipSuccess ? GP_OK : GP_ELAUNCH;
#else
    (void)out;
    return GP_EINVAL;
#endif
}

/* tuning builds only: per-workgroup (HW_ID, XCC_ID, start, end) of the last instrumented kernel */
int gp_debug_wg_stamps(unsigned long long *out) {
#ifdef GP_TIMING
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gp_dbg_wg), sizeof(unsigned long long) * 4096) == hipSuccess ? GP_OK : GP_ELAUNCH;
#else
    (void)out;
    return GP_EINVAL;
#endif
}

int gp_version(void) { return 1; }

int gp_device_arch(char *buf, int buflen) {
    if (!buf || buflen <= 0) return GP_EINVAL;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GP_ELAUNCH;
    strncpy(buf, prop.gcnArchName, (size_t)buflen - 1);
    buf[buflen - 1] = 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? GP_OK : GP_EARCH;
}

}  // extern "C"
