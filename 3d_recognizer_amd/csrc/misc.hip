// Error text, version and small host helpers of the C ABI (include/rl_randlanet.h).
#include "rl_common.h"
#include <atomic>

static thread_local char g_err[512] = "";

void rl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local const char* g_kernel = "";
void rl_note_kernel(const char* name) { g_kernel = name; }

static std::atomic<long long> g_launches{0};
void rl_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" int64_t rl_launch_count(void) { return (int64_t)g_launches.load(std::memory_order_relaxed); }

// A measurement aid: ONE workgroup of one wavefront that occupies its stream for `us` microseconds (wall clock: the 100 MHz
// constant counter) and touches nothing - a stand-in of known length for a collective when a schedule around it is timed on one
// GPU (tools/allreduce_standin.py).  Bounded (<= 20 ms), so the wavefront always ends.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int rl_spin_us(int us, void* stream) {
    RL_REQUIRE(us >= 0 && us <= 20000, RL_ERR_ARGS, "rl_spin_us: 0 .. 20000 us");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)us * 100);
    RL_LAUNCH_CHECK("rl_spin_us");
    return RL_OK;
}

extern "C" const char* rl_last_error(void) { return g_err; }
extern "C" const char* rl_last_kernel(void) { return g_kernel; }
extern "C" int rl_version(void) { return RL_VERSION; }
extern "C" int rl_row_blocks(int64_t rows, int rows_per_tile) {
    return rl_row_blocks_host(rows, rows_per_tile);
}
