// Library identification (include/genpose_hip.h).
#include <string.h>

#include "gp_common.h"

extern "C" {

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
