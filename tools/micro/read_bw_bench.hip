// What a streaming kernel reaches on this GPU, reads alone and reads + writes, operands cold from HBM (1.3 GB swept, nothing
// survives in the 256 MB Infinity Cache) and resident (84 MB re-read):  hipcc --offload-arch=gfx950 -O3 read_bw_bench.hip -o rb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ a, long n4, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a + i + u * stride)); v[u] = make_float4(t[0], t[1], t[2], t[3]); } else v[u] = a[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { const float4 v = a[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[0] = 1.f;
}
// contiguous band per workgroup (the layout of the row-sweep kernels: a workgroup owns 128-row tiles)
template <int U>
__global__ __launch_bounds__(256) void read_band_kernel(const float4* __restrict__ a, long n4, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long band = 256 * U;                       // float4 per workgroup and trip
    for (long b = (long)blockIdx.x * band; b + band <= n4; b += (long)gridDim.x * band) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[b + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[0] = 1.f;
}
template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, long n4) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) b[i + u * stride] = v[u];
    }
}

// two streams at the same offsets (G and Y of a BatchNorm-backward sweep), contiguous band per workgroup
template <int U>
__global__ __launch_bounds__(256) void read2_band_kernel(const float4* __restrict__ a, const float4* __restrict__ b, long n4, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long band = 256 * U;
    for (long t = (long)blockIdx.x * band; t + band <= n4; t += (long)gridDim.x * band) {
        float4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = a[t + u * 256 + threadIdx.x]; w[u] = b[t + u * 256 + threadIdx.x]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x * w[u].x; acc.y += v[u].y * w[u].y; acc.z += v[u].z * w[u].z; acc.w += v[u].w * w[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[0] = 1.f;
}
// the row-sweep kernels' ownership: a workgroup owns TILE-byte tiles (blockIdx.x, + gridDim.x, ...), U x 4 KB per trip inside a tile
template <int U, int TILE4>
__global__ __launch_bounds__(256) void read2_tile_kernel(const float4* __restrict__ a, const float4* __restrict__ b, long n4, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long ntiles = n4 / TILE4;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (long t = tile * TILE4; t < (tile + 1) * TILE4; t += 256 * U) {
            float4 v[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { v[u] = a[t + u * 256 + threadIdx.x]; w[u] = b[t + u * 256 + threadIdx.x]; }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += v[u].x * w[u].x; acc.y += v[u].y * w[u].y; acc.z += v[u].z * w[u].z; acc.w += v[u].w * w[u].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[0] = 1.f;
}

__global__ void fill_kernel(float* a, long n, unsigned seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        a[i] = (float)(x & 0xFFFFFF) * (1.f / 8388608.f) - 1.f;
    }
}

template <typename F>
static float timed(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    const long big = 1344l << 20, small = 84l << 20;          // bytes
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 0, big)); CK(hipMemset(b, 0, big));
    if (getenv("RB_RANDOM")) {      // random payload instead of zeros
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (float*)a, big / 4, 1u);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (float*)b, big / 4, 2u);
        CK(hipDeviceSynchronize());
        printf("random payload\n");
    }
    const int grids[] = {512, 1024, 2048, 4096};
    for (int gi = 0; gi < 4; ++gi) {
        const int g = grids[gi];
        for (int which = 0; which < 2; ++which) {
            const long bytes = which ? small : big;
            const long n4 = bytes / 16;
            const char* nm = which ? "resident 84 MB" : "cold 1.3 GB   ";
            float t;
            t = timed([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(g), dim3(256), 0, 0, a, n4, out); }, which ? 50 : 5);
            printf("grid %4d %s read  x4      %8.1f us %6.2f TB/s\n", g, nm, t, bytes / t * 1e-6);
            t = timed([&] { hipLaunchKernelGGL((read_kernel<8, false>), dim3(g), dim3(256), 0, 0, a, n4, out); }, which ? 50 : 5);
            printf("grid %4d %s read  x8      %8.1f us %6.2f TB/s\n", g, nm, t, bytes / t * 1e-6);
            t = timed([&] { hipLaunchKernelGGL((read_kernel<8, true>), dim3(g), dim3(256), 0, 0, a, n4, out); }, which ? 50 : 5);
            printf("grid %4d %s read  x8 nt   %8.1f us %6.2f TB/s\n", g, nm, t, bytes / t * 1e-6);
            t = timed([&] { hipLaunchKernelGGL((read_band_kernel<8>), dim3(g), dim3(256), 0, 0, a, n4, out); }, which ? 50 : 5);
            printf("grid %4d %s read  band x8 %8.1f us %6.2f TB/s\n", g, nm, t, bytes / t * 1e-6);
            t = timed([&] { hipLaunchKernelGGL((copy_kernel<4>), dim3(g), dim3(256), 0, 0, a, b, n4); }, which ? 50 : 5);
            printf("grid %4d %s copy  x4      %8.1f us %6.2f TB/s (read + write)\n", g, nm, t, 2.0 * bytes / t * 1e-6);
        }
    }
    // launches of the size of one BatchNorm-backward sweep (2 x 84 MB), each on the next slice of the 1.3 GB buffers: cold every time
    {
        const long slice4 = (84l << 20) / 16;
        int turn = 0;
        for (int g : {512, 1024, 2048}) {
            float t = timed([&] { const long o = (long)(turn++ % 16) * slice4; hipLaunchKernelGGL((read2_band_kernel<4>), dim3(g), dim3(256), 0, 0, a + o, b + o, slice4, out); }, 32);
            printf("grid %4d 2 x 84 MB cold slices, band x4 (2 streams)   %8.1f us %6.2f TB/s\n", g, t, 2.0 * (84l << 20) / t * 1e-6);
            t = timed([&] { const long o = (long)(turn++ % 16) * slice4; hipLaunchKernelGGL((read2_band_kernel<8>), dim3(g), dim3(256), 0, 0, a + o, b + o, slice4, out); }, 32);
            printf("grid %4d 2 x 84 MB cold slices, band x8 (2 streams)   %8.1f us %6.2f TB/s\n", g, t, 2.0 * (84l << 20) / t * 1e-6);
            t = timed([&] { const long o = (long)(turn++ % 16) * slice4; hipLaunchKernelGGL((read2_tile_kernel<2, 2048>), dim3(g), dim3(256), 0, 0, a + o, b + o, slice4, out); }, 32);
            printf("grid %4d 2 x 84 MB cold slices, 32 KB tiles, 8 KB trips %8.1f us %6.2f TB/s\n", g, t, 2.0 * (84l << 20) / t * 1e-6);
            t = timed([&] { const long o = (long)(turn++ % 16) * slice4; hipLaunchKernelGGL((read2_tile_kernel<4, 2048>), dim3(g), dim3(256), 0, 0, a + o, b + o, slice4, out); }, 32);
            printf("grid %4d 2 x 84 MB cold slices, 32 KB tiles, 16 KB trips %8.1f us %6.2f TB/s\n", g, t, 2.0 * (84l << 20) / t * 1e-6);
            t = timed([&] { const long o = (long)(turn++ % 16) * slice4; hipLaunchKernelGGL((read_band_kernel<8>), dim3(g), dim3(256), 0, 0, a + o, slice4, out); }, 32);
            printf("grid %4d 1 x 84 MB cold slices, band x8 (1 stream)    %8.1f us %6.2f TB/s\n", g, t, 1.0 * (84l << 20) / t * 1e-6);
        }
    }
    return 0;
}
