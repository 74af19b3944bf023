// Shared host-side helpers of libsbev_hip.so (error slot, launch checks).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/sbev_hip.h"

namespace sbev {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SBEV_ELAUNCH;
    }
    return SBEV_OK;
}

#define SBEV_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::sbev::set_error(__VA_ARGS__);     \
            return SBEV_EINVAL;                 \
        }                                       \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront

// optional HIP-event bracket around sampler launches (decoder.hip; switched by sbev_profile_sampler)
bool profile_begin(hipStream_t s, hipEvent_t* e0, hipEvent_t* e1);
void profile_end(hipStream_t s, hipEvent_t e0, hipEvent_t e1);

}  // namespace sbev
