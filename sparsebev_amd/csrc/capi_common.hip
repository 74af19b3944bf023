// Error slot + device query of the C ABI (include/sbev_hip.h).
#include "sbev_common.hpp"
#include <atomic>
#include <cstring>

namespace sbev {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sbev

namespace sbev {
static std::atomic<int> g_convention{0};
int box_convention() { return g_convention.load(std::memory_order_relaxed); }
}  // namespace sbev

extern "C" int sbev_set_box_convention(int convention) {
    SBEV_REQUIRE(convention == SBEV_BOX_V1_0_0 || convention == SBEV_BOX_V0_17_1, "sbev_set_box_convention: unknown convention %d", convention);
    sbev::g_convention.store(convention, std::memory_order_relaxed);
    return SBEV_OK;
}

extern "C" int sbev_get_box_convention(void) { return sbev::box_convention(); }

extern "C" int sbev_abi_version(void) { return SBEV_ABI_VERSION; }

extern "C" const char* sbev_last_error(void) { return sbev::g_err; }

extern "C" int sbev_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}
