// Input pipeline on the device: the per-cloud work of the reference's
// PointCloudPreprocessor.preprocess (randlanet/utils/dataset.py:61-97) - sub-sample, optional
// normalisation, augmentation (randlanet/utils/augmentation.py:38-167: jitter, scale, rotate, shift,
// each around the centre / scaled by the mean radius of the cloud AS IT IS AT THAT STAGE) and the
// collation into (B, n, 3+F) float32 / (B, n) int64 - for clouds that live in HBM.
//
// Random numbers are NOT drawn here: the caller supplies the sample indices, the per-point jitter
// noise and the per-cloud scale / rotation / shift draws (the host mirror draws them from numpy's
// global stream in the reference's order, or on the device in its fast mode), so the result is a
// deterministic function of its inputs and can be checked against the reference.
//
// One workgroup of 1024 lanes per cloud; the cloud stays in an fp64 scratch (the reference computes in
// float64 once the jitter noise is added) and every stage is one sweep over it with a block-wide
// fp64 reduction where the next stage needs the centre or the mean radius.  A 40960-point cloud is
// 1 MB of scratch: it never leaves L2.
#include "rl_common.h"

namespace {

struct V3 {
    double x, y, z;
};

// ---- sums over a CLOUD that several workgroups share -----------------------------------------------------------------------------
// One workgroup per cloud (round 2) left a batch of four clouds on four CUs for 0.33 ms of dependent fp64 sweeps - a tenth of the
// training step it feeds.  Now `gw` workgroups split a cloud's points; a stage that needs the centre or the mean radius of the
// whole cloud reduces in three steps: the workgroup's own sum (fixed butterfly + wavefronts in order), one record per workgroup
// in `part`, a barrier over the cloud's workgroups, and every workgroup adds the gw records in index order (deterministic, the
// same value in all of them).  The barrier is an arrival counter per cloud: gw device-scope atomics at ~35 ns each (64 of them
// on one address cost as much as a launch - tools/micro/grid_barrier_bench.hip - 16 do not); the gw * B workgroups of a launch
// fit the device at once (sized from the occupancy calculator, per device), other streams' kernels only delay them, and the spin
// is bounded (cloud_barrier).  Records and counters live in the caller's scratch, zeroed by a kernel of this call.
struct CloudSync {
    double* part;                  // [gw][4] records of this cloud, reused stage after stage
    unsigned* arrive;              // this cloud's arrival counter
    unsigned passed;               // barriers this workgroup has been through
    int gw, w;
    int* gave_up;                  // (LDS) set when a barrier timed out: see below
};
// Every wavefront reaches the end of the kernel whatever its peers do: the spin is BOUNDED (~2^22 polls of ~0.25 us: about a
// second, against ~1 us for a barrier that works).  A workgroup whose peers never arrive - the launch was not co-resident: a CU
// mask, a partition mode - stops waiting for good, raises the cloud's ERROR WORD in the call's scratch
// (rl_batch_assemble_flag_u32: the loader reads the words back asynchronously and raises on the host) and writes the
// sampled coordinates as they were read - finite, so that the neighbour search behind the loader, whose key network assumes
// finite inputs, never sees a NaN (round 5 wrote NaN: a rare hang had become a rare out-of-range index) - instead of
// hanging the GPU.
constexpr unsigned ASM_SPIN_LIMIT = 1u << 22;
__device__ __forceinline__ void cloud_barrier(CloudSync& cs) {
    __syncthreads();
    if (cs.gw > 1) {
        if (threadIdx.x == 0 && !*cs.gave_up) {
            __threadfence();
            const unsigned target = (cs.passed + 1u) * (unsigned)cs.gw;
            atomicAdd(cs.arrive, 1u);
            unsigned spins = 0;
            while (__hip_atomic_load(cs.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > ASM_SPIN_LIMIT) { *cs.gave_up = 1; break; }
            }
            __threadfence();
        }
        __syncthreads();
    }
    cs.passed += 1u;
}
__device__ __forceinline__ double wg_sum(double v, double* red) {
    v = rl_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                 // red may still be read from the previous reduction
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];   // fixed order: deterministic
    return s;
}
__device__ __forceinline__ double wg_max(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = red[0];
    const int nw = blockDim.x >> 6;
    for (int w = 1; w < nw; ++w) s = fmax(s, red[w]);
    return s;
}
// up to four sums (or maxima) of the cloud at once: v[] = this lane's terms; every lane of every workgroup gets the totals
template <int NV, bool MAX = false>
__device__ __forceinline__ void cloud_reduce(double (&v)[NV], CloudSync& cs, double* red) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = MAX ? wg_max(v[k], red) : wg_sum(v[k], red);
    if (cs.gw == 1) return;
    // ONE barrier per reduction: the records alternate between two sets, and set s is written again two reductions later - behind
    // the next reduction's barrier, which nobody passes before everybody has read this one's records
    double* part = cs.part + (cs.passed & 1u) * 64;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) part[cs.w * 4 + k] = v[k];
    }
    cloud_barrier(cs);
    // (plain loads, all in flight at once: the barrier's acquire fence has emptied this CU's vector cache; an agent-scope atomic load
    // per record would be gw dependent round trips)
    double rec[16][NV];
#pragma unroll
    for (int w = 0; w < 16; ++w)
#pragma unroll
        for (int k = 0; k < NV; ++k) rec[w][k] = w < cs.gw ? part[w * 4 + k] : 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = rec[0][k];
#pragma unroll
        for (int w = 1; w < 16; ++w)
            if (w < cs.gw) s = MAX ? fmax(s, rec[w][k]) : s + rec[w][k];
        v[k] = s;
    }
}

// this workgroup's points: [i0, i1)
__device__ __forceinline__ V3 cloud_mean(const double* X, int i0, int i1, int n, CloudSync& cs, double* red) {
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        v[0] += X[3 * i + 0];
        v[1] += X[3 * i + 1];
        v[2] += X[3 * i + 2];
    }
    cloud_reduce<3>(v, cs, red);
    V3 c;
    c.x = v[0] / n; c.y = v[1] / n; c.z = v[2] / n;
    return c;
}
// mean distance to c (augmentation.py:26-35)
__device__ __forceinline__ double cloud_mean_radius(const double* X, int i0, int i1, int n, V3 c, CloudSync& cs, double* red) {
    double v[1] = {0.0};
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const double dx = X[3 * i + 0] - c.x, dy = X[3 * i + 1] - c.y, dz = X[3 * i + 2] - c.z;
        v[0] += sqrt(dx * dx + dy * dy + dz * dz);
    }
    cloud_reduce<1>(v, cs, red);
    return v[0] / n;
}

__global__ __launch_bounds__(1024) void batch_assemble_kernel(const rl_cloud_job* __restrict__ jobs,
                                                              const int64_t* __restrict__ indices,
                                                              const double* __restrict__ noise, int n, int F,
                                                              double* __restrict__ scratch, float* __restrict__ out_input,
                                                              int64_t* __restrict__ out_labels, double* __restrict__ sync_part,
                                                              unsigned* __restrict__ sync_count, int sync_stride) {
    __shared__ double red[16];
    __shared__ int gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    __syncthreads();
    const int b = blockIdx.y, gw = gridDim.x;
    CloudSync cs;
    cs.gw = gw; cs.w = blockIdx.x; cs.passed = 0u; cs.gave_up = &gave_up;
    cs.part = sync_part + (long)b * sync_stride;          // two sets of 16 records of 4 doubles (sync_stride doubles per cloud)
    cs.arrive = sync_count + 2l * b * sync_stride;         // (unsigned units: two per double)
    const rl_cloud_job job = jobs[b];
    const int64_t* idx = indices + (long)b * n;
    double* X = scratch + (long)b * n * 3;
    float* out = out_input + (long)b * n * (3 + F);
    int64_t* lab = out_labels + (long)b * n;
    const int C = 3 + F;
    const int per = (n + gw - 1) / gw;
    const int i0 = min(n, cs.w * per), i1 = min(n, i0 + per);       // this workgroup's points: it alone reads and writes them

    // sub-sample (dataset.py:76-81): coordinates to the fp64 scratch, features and labels straight out
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const long j = idx[i];
        if (job.xyz_f64) {
            const double* s = (const double*)job.xyz + 3 * j;
            X[3 * i + 0] = s[0]; X[3 * i + 1] = s[1]; X[3 * i + 2] = s[2];
        } else {
            const float* s = (const float*)job.xyz + 3 * j;
            X[3 * i + 0] = (double)s[0]; X[3 * i + 1] = (double)s[1]; X[3 * i + 2] = (double)s[2];
        }
        for (int f = 0; f < F; ++f) out[(long)i * C + 3 + f] = job.features[j * F + f];
        lab[i] = job.labels[j];
    }
    __syncthreads();

    if (job.normalization != 0) {
        // dataset.py:82-93: centre, then divide by the mean / max / std of the distances to it
        const V3 c = cloud_mean(X, i0, i1, n, cs, red);
        double sm[1] = {0.0};
        double mx[1] = {0.0};
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            X[3 * i + 0] -= c.x; X[3 * i + 1] -= c.y; X[3 * i + 2] -= c.z;
            const double r = sqrt(X[3 * i + 0] * X[3 * i + 0] + X[3 * i + 1] * X[3 * i + 1] + X[3 * i + 2] * X[3 * i + 2]);
            sm[0] += r;
            mx[0] = fmax(mx[0], r);
        }
        double radius = 1.0;
        if (job.normalization == 1) {
            cloud_reduce<1>(sm, cs, red);
            radius = sm[0] / n;
        } else if (job.normalization == 2) {
            cloud_reduce<1, true>(mx, cs, red);
            radius = mx[0];
        } else if (job.normalization == 3) {
            cloud_reduce<1>(sm, cs, red);
            const double mean = sm[0] / n;
            double v[1] = {0.0};
            for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
                const double r = sqrt(X[3 * i + 0] * X[3 * i + 0] + X[3 * i + 1] * X[3 * i + 1] + X[3 * i + 2] * X[3 * i + 2]);
                v[0] += (r - mean) * (r - mean);
            }
            cloud_reduce<1>(v, cs, red);
            radius = sqrt(v[0] / n);   // np.std: population standard deviation
        }
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            X[3 * i + 0] /= radius; X[3 * i + 1] /= radius; X[3 * i + 2] /= radius;
        }
        __syncthreads();
    }

    if (job.augment) {
        // jitter (augmentation.py:38-58): clip(radius * variance * noise, +-limit) added to every point
        V3 c = cloud_mean(X, i0, i1, n, cs, red);
        double radius = cloud_mean_radius(X, i0, i1, n, c, cs, red);
        const double* nz = noise + (long)b * n * 3;
        const double amp = radius * job.jitter_variance;
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double d = noise ? amp * nz[3 * i + a] : 0.0;   // no noise given: jitter is skipped
                d = fmin(fmax(d, -job.jitter_limit), job.jitter_limit);
                X[3 * i + a] = d + X[3 * i + a];
            }
        }
        __syncthreads();
        // scale about the centre (augmentation.py:61-80)
        c = cloud_mean(X, i0, i1, n, cs, red);
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            X[3 * i + 0] = (X[3 * i + 0] - c.x) * job.scale + c.x;
            X[3 * i + 1] = (X[3 * i + 1] - c.y) * job.scale + c.y;
            X[3 * i + 2] = (X[3 * i + 2] - c.z) * job.scale + c.z;
        }
        __syncthreads();
        // rotate about the centre: (x - c) . R^T + c   (augmentation.py:83-128), R row-major
        c = cloud_mean(X, i0, i1, n, cs, red);
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            const double x = X[3 * i + 0] - c.x, y = X[3 * i + 1] - c.y, z = X[3 * i + 2] - c.z;
            X[3 * i + 0] = ((x * job.R[0] + y * job.R[1]) + z * job.R[2]) + c.x;
            X[3 * i + 1] = ((x * job.R[3] + y * job.R[4]) + z * job.R[5]) + c.y;
            X[3 * i + 2] = ((x * job.R[6] + y * job.R[7]) + z * job.R[8]) + c.z;
        }
        __syncthreads();
        // shift by mean radius * uniform draws (augmentation.py:131-144)
        c = cloud_mean(X, i0, i1, n, cs, red);
        radius = cloud_mean_radius(X, i0, i1, n, c, cs, red);
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            X[3 * i + 0] += radius * job.shift[0];
            X[3 * i + 1] += radius * job.shift[1];
            X[3 * i + 2] += radius * job.shift[2];
        }
        __syncthreads();
    }
    // torch.from_numpy(xyz).float() (dataset.py:51): round to float32, coordinates first (dataset.py:53)
    __syncthreads();
    // a cloud-wide sum never completed: nothing of this share is trustworthy - say so in the cloud's error word (the values
    // written are the finite ones at hand: whatever stage the share reached)
    if (gave_up != 0 && threadIdx.x == 0) atomicOr(cs.arrive + 2, 1u);
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        out[(long)i * C + 0] = (float)X[3 * i + 0];
        out[(long)i * C + 1] = (float)X[3 * i + 1];
        out[(long)i * C + 2] = (float)X[3 * i + 2];
    }
}

// ---- the draws of the device-rng loader in ONE launch -----------------------------------------------------------------------
// The fast mode of the device loader used to draw a batch with torch: a randperm (= a radix sort, ~6 launches) and a randn per
// cloud - ~40 launches and 0.27 ms of sort kernels for a batch of four, a third of what the loop loses against the bare step.
// Here: thread i of cloud b writes sample index i and the three jitter draws of point i, both pure functions of (seed, b, i):
//   * sampling without replacement = the first n values of a keyed PERMUTATION of [0, n_points): a six-round unbalanced Feistel
//     network on ceil(log2(n_points)) bits, walked until it lands inside the range (cycle walking: < 2 steps on average) -
//     no sort, no scratch; past n_points (n > n_points: preprocessing.sample_points pads with replacement) uniform draws;
//   * normals = Box-Muller on 53-bit uniforms from Philox4x32-10 (the generator of the dropout kernels).
// Not the reference's numpy stream - that is the loader's "numpy" mode - but draws of the same distributions.
__device__ __forceinline__ uint4 draw_philox(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * ctr.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * ctr.z;
        ctr = make_uint4((unsigned)(p1 >> 32) ^ ctr.y ^ key.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ ctr.w ^ key.y, (unsigned)p0);
        key.x += 0x9E3779B9u;
        key.y += 0xBB67AE85u;
    }
    return ctr;
}
__device__ __forceinline__ unsigned draw_mix(unsigned x) {      // (murmur3's finalizer)
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned feistel(unsigned x, int lbits, int rbits, unsigned k0, unsigned k1) {
    const unsigned lmask = (1u << lbits) - 1u, rmask = (1u << rbits) - 1u;
    unsigned L = x >> rbits, R = x & rmask;
#pragma unroll
    for (int r = 0; r < 6; r += 2) {
        L ^= draw_mix(R ^ k0 ^ (0x9E3779B9u * (r + 1))) & lmask;
        R ^= draw_mix(L ^ k1 ^ (0x7F4A7C15u * (r + 2))) & rmask;
    }
    return (L << rbits) | R;
}
__device__ __forceinline__ double draw_u53(unsigned hi, unsigned lo) {       // (0, 1]
    const unsigned long long m = (((unsigned long long)hi << 32) | lo) >> 11;
    return ((double)m + 1.0) * (1.0 / 9007199254740992.0);
}

__global__ __launch_bounds__(256) void batch_draw_kernel(const rl_cloud_job* __restrict__ jobs, int n, unsigned long long seed,
                                                         int64_t* __restrict__ indices, double* __restrict__ noise) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    if (indices) {
        const long n_src = jobs[b].n_points;
        long j;
        if (i < n_src) {
            int bits = 2;
            while ((1l << bits) < n_src) ++bits;
            const int rbits = bits >> 1, lbits = bits - rbits;
            const uint4 k = draw_philox(make_uint4(0xFFFFFFFFu, (unsigned)b, 0u, 0x5A17u), key);     // this cloud's permutation keys
            unsigned x = (unsigned)i;
            do x = feistel(x, lbits, rbits, k.x, k.y); while ((long)x >= n_src);
            j = (long)x;
        } else {
            const uint4 r = draw_philox(make_uint4((unsigned)i, (unsigned)b, 1u, 0x5A17u), key);
            j = (long)((((unsigned long long)r.x << 32) | r.y) % (unsigned long long)n_src);
        }
        indices[(long)b * n + i] = j;
    }
    if (noise) {
        const uint4 r0 = draw_philox(make_uint4((unsigned)i, (unsigned)b, 2u, 0x5A17u), key);
        const uint4 r1 = draw_philox(make_uint4((unsigned)i, (unsigned)b, 3u, 0x5A17u), key);
        const double m0 = sqrt(-2.0 * log(draw_u53(r0.x, r0.y))), a0 = 6.283185307179586 * draw_u53(r0.z, r0.w);
        const double m1 = sqrt(-2.0 * log(draw_u53(r1.x, r1.y))), a1 = 6.283185307179586 * draw_u53(r1.z, r1.w);
        double* o = noise + ((long)b * n + i) * 3;
        o[0] = m0 * cos(a0);
        o[1] = m0 * sin(a0);
        o[2] = m1 * cos(a1);
    }
}

}  // namespace

extern "C" int rl_batch_draw(const rl_cloud_job* jobs_dev, int B, int n, uint64_t seed, int64_t* indices, double* noise, void* stream) {
    RL_REQUIRE(jobs_dev && B > 0 && n > 0 && (indices || noise), RL_ERR_ARGS, "rl_batch_draw: bad arguments");
    hipLaunchKernelGGL(batch_draw_kernel, dim3(rl_cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, jobs_dev, n,
                       (unsigned long long)seed, indices, noise);
    rl_note_kernel("batch_draw_kernel");
    RL_LAUNCH_CHECK("rl_batch_draw");
    return RL_OK;
}

// Workgroups per cloud a launch may use: all gw * B of them must be resident at once (cloud_barrier spins).  Per DEVICE and from
// the occupancy calculator (resident 1024-lane workgroups per CU x CUs), asked once per device under a lock.
#include <mutex>
static int assemble_capacity() {
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (cached[dev] == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, batch_assemble_kernel, 1024, 0) != hipSuccess || per_cu <= 0) per_cu = 0;
        (void)hipGetLastError();
        cached[dev] = cus > 0 && per_cu > 0 ? cus : -1;        // one workgroup per CU is all this kernel asks for
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}

// the cloud-wide sums' records and arrival counters live in the CALLER's scratch, behind the coordinates: per cloud two sets of
// 16 records of 4 doubles, then two counters - zeroed by a kernel of this call (no state in the library: two loaders on two
// streams do not meet, and a launch that was aborted leaves nothing behind)
constexpr int ASM_SYNC_DOUBLES = 130;
__global__ void batch_assemble_reset_kernel(unsigned* __restrict__ count, int stride_u32, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { count[(long)b * stride_u32] = 0u; count[(long)b * stride_u32 + 1] = 0u; count[(long)b * stride_u32 + 2] = 0u; }
}

// where cloud b's error word sits, counted in 32-bit words from the start of `scratch` (0 = fine; 1 = a cloud-wide rendezvous
// timed out and that cloud's output is not what the reference would produce)
extern "C" int64_t rl_batch_assemble_flag_u32(int B, int n, int b) {
    if (B <= 0 || n <= 0 || b < 0 || b >= B) return -1;
    return 2 * ((int64_t)B * n * 3 + (int64_t)b * ASM_SYNC_DOUBLES + 128) + 2;
}

extern "C" int64_t rl_batch_assemble_scratch_doubles(int B, int n) {
    if (B <= 0 || n <= 0) return 0;
    return (int64_t)B * n * 3 + (int64_t)B * ASM_SYNC_DOUBLES;
}

extern "C" int rl_batch_assemble(const rl_cloud_job* jobs_dev, int B, int n, int F, const int64_t* indices,
                                 const double* noise, double* scratch, float* out_input, int64_t* out_labels,
                                 void* stream) {
    RL_REQUIRE(jobs_dev && indices && scratch && out_input && out_labels, RL_ERR_ARGS, "rl_batch_assemble: null pointer");
    RL_REQUIRE(B > 0 && n > 0 && F >= 0, RL_ERR_ARGS, "rl_batch_assemble: bad sizes (B %d, n %d, F %d)", B, n, F);
    // workgroups per cloud: up to 16, all gw * B of a launch resident at once (one of 1024 lanes per CU: 256 on an MI355X)
    const int cap = assemble_capacity();
    int gw = cap / B;
    gw = gw > 16 ? 16 : (gw < 1 ? 1 : gw);
    while (gw > 1 && (long)(gw - 1) * 1024 >= n) --gw;        // (no workgroup without points)
    if (getenv("RL_ASSEMBLE_ONE_WG")) gw = 1;
    double* part = scratch + (long)B * n * 3;                 // [B][ASM_SYNC_DOUBLES]: 128 doubles of records, then the counters
    unsigned* count = reinterpret_cast<unsigned*>(part + 128);
    hipStream_t st = (hipStream_t)stream;
    const int sync_stride = ASM_SYNC_DOUBLES;
    // (always: the error words are read back by the caller whatever gw was)
    hipLaunchKernelGGL(batch_assemble_reset_kernel, dim3(rl_cdiv(B, 64)), dim3(64), 0, st, count, 2 * ASM_SYNC_DOUBLES, B);
    RL_LAUNCH_CHECK("rl_batch_assemble(reset)");
    // (A COOPERATIVE launch - the runtime's own co-residency check - was built and measured in round 5: the loader's launch then
    // serialises with the training stream's graph replays, Model.train 883 -> 333 clouds/s.  Kept: the capacity from the occupancy
    // calculator per device, records / counters per call, the bounded spin.)
    hipLaunchKernelGGL(batch_assemble_kernel, dim3(gw, B), dim3(1024), 0, st, jobs_dev, indices, noise, n,
                       F, scratch, out_input, out_labels, part, count, sync_stride);
    rl_note_kernel("batch_assemble_kernel");
    RL_LAUNCH_CHECK("rl_batch_assemble");
    return RL_OK;
}
