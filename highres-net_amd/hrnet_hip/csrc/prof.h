// Built-in kernel timing: when enabled (hrn_profile_enable), every launcher brackets its kernel with a hipEvent pair
// on the launch stream and files it under a kernel-family name together with the launch's algorithmic FLOPs and
// bytes.  bench.py reads the totals back (hrn_profile_get) to state achieved TFLOP/s / GB/s per family against the
// gfx950 roofline.  Disabled (the default) it costs one predictable branch per launch.
#pragma once
#include <hip/hip_runtime.h>

struct HrnProfScope {
    int rec;
    hipStream_t stream;
    HrnProfScope(const char* family, double flops, double bytes, hipStream_t s);
    ~HrnProfScope();
};
