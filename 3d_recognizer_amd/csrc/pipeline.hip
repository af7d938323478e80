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

__device__ __forceinline__ double block_sum(double v, double* red) {
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
__device__ __forceinline__ double block_max(double v, double* red) {
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

__device__ __forceinline__ V3 block_mean(const double* X, int n, double* red) {
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        sx += X[3 * i + 0];
        sy += X[3 * i + 1];
        sz += X[3 * i + 2];
    }
    V3 c;
    c.x = block_sum(sx, red) / n;
    c.y = block_sum(sy, red) / n;
    c.z = block_sum(sz, red) / n;
    return c;
}
// mean distance to c (augmentation.py:26-35)
__device__ __forceinline__ double block_mean_radius(const double* X, int n, V3 c, double* red) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double dx = X[3 * i + 0] - c.x, dy = X[3 * i + 1] - c.y, dz = X[3 * i + 2] - c.z;
        s += sqrt(dx * dx + dy * dy + dz * dz);
    }
    return block_sum(s, red) / n;
}

__global__ __launch_bounds__(1024) void batch_assemble_kernel(const rl_cloud_job* __restrict__ jobs,
                                                              const int64_t* __restrict__ indices,
                                                              const double* __restrict__ noise, int n, int F,
                                                              double* __restrict__ scratch, float* __restrict__ out_input,
                                                              int64_t* __restrict__ out_labels) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    const rl_cloud_job job = jobs[b];
    const int64_t* idx = indices + (long)b * n;
    double* X = scratch + (long)b * n * 3;
    float* out = out_input + (long)b * n * (3 + F);
    int64_t* lab = out_labels + (long)b * n;
    const int C = 3 + F;

    // sub-sample (dataset.py:76-81): coordinates to the fp64 scratch, features and labels straight out
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
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
        const V3 c = block_mean(X, n, red);
        double s = 0.0, mx = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            X[3 * i + 0] -= c.x; X[3 * i + 1] -= c.y; X[3 * i + 2] -= c.z;
            const double r = sqrt(X[3 * i + 0] * X[3 * i + 0] + X[3 * i + 1] * X[3 * i + 1] + X[3 * i + 2] * X[3 * i + 2]);
            s += r;
            mx = fmax(mx, r);
        }
        double radius = 1.0;
        if (job.normalization == 1) radius = block_sum(s, red) / n;
        else if (job.normalization == 2) radius = block_max(mx, red);
        else if (job.normalization == 3) {
            const double mean = block_sum(s, red) / n;
            double v = 0.0;
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const double r = sqrt(X[3 * i + 0] * X[3 * i + 0] + X[3 * i + 1] * X[3 * i + 1] + X[3 * i + 2] * X[3 * i + 2]);
                v += (r - mean) * (r - mean);
            }
            radius = sqrt(block_sum(v, red) / n);   // np.std: population standard deviation
        }
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            X[3 * i + 0] /= radius; X[3 * i + 1] /= radius; X[3 * i + 2] /= radius;
        }
        __syncthreads();
    }

    if (job.augment) {
        // jitter (augmentation.py:38-58): clip(radius * variance * noise, +-limit) added to every point
        V3 c = block_mean(X, n, red);
        double radius = block_mean_radius(X, n, c, red);
        const double* nz = noise + (long)b * n * 3;
        const double amp = radius * job.jitter_variance;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double d = noise ? amp * nz[3 * i + a] : 0.0;   // no noise given: jitter is skipped
                d = fmin(fmax(d, -job.jitter_limit), job.jitter_limit);
                X[3 * i + a] = d + X[3 * i + a];
            }
        }
        __syncthreads();
        // scale about the centre (augmentation.py:61-80)
        c = block_mean(X, n, red);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            X[3 * i + 0] = (X[3 * i + 0] - c.x) * job.scale + c.x;
            X[3 * i + 1] = (X[3 * i + 1] - c.y) * job.scale + c.y;
            X[3 * i + 2] = (X[3 * i + 2] - c.z) * job.scale + c.z;
        }
        __syncthreads();
        // rotate about the centre: (x - c) . R^T + c   (augmentation.py:83-128), R row-major
        c = block_mean(X, n, red);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double x = X[3 * i + 0] - c.x, y = X[3 * i + 1] - c.y, z = X[3 * i + 2] - c.z;
            X[3 * i + 0] = ((x * job.R[0] + y * job.R[1]) + z * job.R[2]) + c.x;
            X[3 * i + 1] = ((x * job.R[3] + y * job.R[4]) + z * job.R[5]) + c.y;
            X[3 * i + 2] = ((x * job.R[6] + y * job.R[7]) + z * job.R[8]) + c.z;
        }
        __syncthreads();
        // shift by mean radius * uniform draws (augmentation.py:131-144)
        c = block_mean(X, n, red);
        radius = block_mean_radius(X, n, c, red);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            X[3 * i + 0] += radius * job.shift[0];
            X[3 * i + 1] += radius * job.shift[1];
            X[3 * i + 2] += radius * job.shift[2];
        }
        __syncthreads();
    }
    // torch.from_numpy(xyz).float() (dataset.py:51): round to float32, coordinates first (dataset.py:53)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        out[(long)i * C + 0] = (float)X[3 * i + 0];
        out[(long)i * C + 1] = (float)X[3 * i + 1];
        out[(long)i * C + 2] = (float)X[3 * i + 2];
    }
}

}  // namespace

extern "C" int rl_batch_assemble(const rl_cloud_job* jobs_dev, int B, int n, int F, const int64_t* indices,
                                 const double* noise, double* scratch, float* out_input, int64_t* out_labels,
                                 void* stream) {
    RL_REQUIRE(jobs_dev && indices && scratch && out_input && out_labels, RL_ERR_ARGS, "rl_batch_assemble: null pointer");
    RL_REQUIRE(B > 0 && n > 0 && F >= 0, RL_ERR_ARGS, "rl_batch_assemble: bad sizes (B %d, n %d, F %d)", B, n, F);
    hipLaunchKernelGGL(batch_assemble_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, jobs_dev, indices, noise, n,
                       F, scratch, out_input, out_labels);
    rl_note_kernel("batch_assemble_kernel");
    RL_LAUNCH_CHECK("rl_batch_assemble");
    return RL_OK;
}
