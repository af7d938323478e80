// What a DEPENDENT kernel launch costs on the GPU's timeline: issued into a stream one by one vs replayed from a hipGraph.
// hipcc --offload-arch=gfx950 -O3 -o lf launch_floor_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ __launch_bounds__(256) void k_sweep(const float* __restrict__ a, float* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i] * 1.0001f + 1.f;
}

int main() {
    float *p, *a, *b;
    const long n = 1 << 20;      // 4 MB: a small sweep
    CK(hipMalloc(&p, 4096)); CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(p, 0, 4096)); CK(hipMemset(a, 0, n * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 200;
    for (int variant = 0; variant < 2; ++variant) {
        auto issue = [&]() {
            for (int j = 0; j < N; ++j) {
                if (variant == 0) k_tiny<<<1, 64, 0, st>>>(p);
                else k_sweep<<<1024, 256, 0, st>>>(j & 1 ? b : a, j & 1 ? a : b, n);
            }
        };
        // stream
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); issue(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%s: stream  %.2f us per launch\n", variant ? "4 MB sweep" : "tiny kernel", best * 1000.f / N);
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); issue(); CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%s: graph   %.2f us per node\n", variant ? "4 MB sweep" : "tiny kernel", best * 1000.f / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
