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

extern "C" const char* rl_last_error(void) { return g_err; }
extern "C" const char* rl_last_kernel(void) { return g_kernel; }
extern "C" int rl_version(void) { return RL_VERSION; }
extern "C" int rl_row_blocks(int64_t rows, int rows_per_tile) {
    return rl_row_blocks_host(rows, rows_per_tile);
}
