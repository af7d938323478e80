// Cost of a hand-rolled grid barrier on gfx950 (one workgroup of 1024 threads per CU): (a) arrival counter + generation word,
// (b) one flag per workgroup, polled by the first G lanes of every workgroup.  hipcc --offload-arch=gfx950 -O3 -o gb grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ unsigned long long g_bar[2];
__device__ unsigned g_flags[1024];

__device__ __forceinline__ void barrier_counter() {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long gen = __hip_atomic_load(&g_bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long a = __hip_atomic_fetch_add(&g_bar[0], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (a == gridDim.x - 1) {
            __hip_atomic_store(&g_bar[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&g_bar[1], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&g_bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

// epoch: a value no flag holds yet (the caller counts barriers)
__device__ __forceinline__ void barrier_flags(unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&g_flags[blockIdx.x], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < gridDim.x) {
        while ((int)(__hip_atomic_load(&g_flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) __builtin_amdgcn_s_sleep(1);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_counter(int n, float* out) {
    float v = 0.f;
    for (int i = 0; i < n; ++i) { barrier_counter(); v += 1.f; }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__global__ __launch_bounds__(1024) void k_flags(int n, unsigned base, float* out) {
    float v = 0.f;
    for (int i = 0; i < n; ++i) { barrier_flags(base + i + 1); v += 1.f; }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__global__ __launch_bounds__(1024) void k_empty(float* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.f; }

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int G = pr.multiProcessorCount;
    printf("CUs %d\n", G);
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned base = 0;
    for (int n : {0, 1, 2, 10, 100}) {
        for (int variant = 0; variant < 2; ++variant) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                for (int j = 0; j < 20; ++j) {
                    if (variant == 0) k_counter<<<G, 1024>>>(n, out);
                    else { k_flags<<<G, 1024>>>(n, base, out); base += n; }
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%s n=%3d: %.2f us per launch\n", variant ? "flags  " : "counter", n, best * 1000.f / 20);
        }
    }
    float h[4];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("check %g %s\n", h[0], hipGetErrorString(hipGetLastError()));
    return 0;
}
