// BatchNorm2d(eps 1e-6, momentum 0.99) of the reference's SharedMLP / bn_start
// (randlanet/utils/modules.py:85-89, 496-499) folded into per-channel (scale, shift) that the
// consumers apply while loading ("lazy" operands), and its backward.
//
// Forward: producers leave one partial (sum, sum of squares) per workgroup and channel, in
// double; rl_bn_finalize sums them in slot order (deterministic), forms mean / biased variance,
// updates the running statistics the way torch does and emits scale = gamma*invstd,
// shift = beta - mean*scale.
// Backward: G holds d(loss)/d(activated output).  With g = G*act'(z) and xhat = (Y-mean)*invstd,
//   dbeta = sum g, dgamma = sum g*xhat, dY = scale*(g - mean(g) - xhat*mean(g*xhat)).
#include "rl_common.h"

namespace {

// Sum of the per-workgroup partials of channel c over all slots by a whole workgroup (256 lanes: at most four slots per
// lane, i.e. ONE round of memory latency for up to 1024 slots - these kernels are pure latency), in a fixed order:
// lane t adds slots t, t+256, ... ; the lanes are then combined by a fixed butterfly + a fixed sum over the wavefronts.
__device__ __forceinline__ void slot_sums_wg(const double* __restrict__ stats, int nslots, int C, int c,
                                             double (*red)[2], double& s_out, double& q_out) {
    const int t = threadIdx.x;
    double s = 0.0, q = 0.0;
    const double* base = stats + c;
    for (int i = t; i < nslots; i += 256) {
        s += base[((long)i * 2 + 0) * C];
        q += base[((long)i * 2 + 1) * C];
    }
    s = rl_wave_sum(s);
    q = rl_wave_sum(q);
    if ((t & 63) == 0) { red[t >> 6][0] = s; red[t >> 6][1] = q; }
    __syncthreads();
    s_out = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    q_out = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
}

// one wavefront's worth of the same sum (kept for rl_bn_reduce_slots)
__device__ __forceinline__ void slot_sums(const double* __restrict__ stats, int nslots, int C, int c, int lane,
                                          double& s_out, double& q_out) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    const double* base = stats + c;
    int i = lane;
    for (; i + 192 < nslots; i += 256) {
        s0 += base[((long)i * 2 + 0) * C];          q0 += base[((long)i * 2 + 1) * C];
        s1 += base[((long)(i + 64) * 2 + 0) * C];   q1 += base[((long)(i + 64) * 2 + 1) * C];
        s2 += base[((long)(i + 128) * 2 + 0) * C];  q2 += base[((long)(i + 128) * 2 + 1) * C];
        s3 += base[((long)(i + 192) * 2 + 0) * C];  q3 += base[((long)(i + 192) * 2 + 1) * C];
    }
    for (; i < nslots; i += 64) {
        s0 += base[((long)i * 2 + 0) * C];
        q0 += base[((long)i * 2 + 1) * C];
    }
    s_out = rl_wave_sum((s0 + s1) + (s2 + s3));
    q_out = rl_wave_sum((q0 + q1) + (q2 + q3));
}

// one workgroup per channel
__device__ __forceinline__ void bn_finalize_body(
    const double* __restrict__ stats, int nslots, double count, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
    int64_t* nbt, float momentum, float eps, int training, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    const float* __restrict__ folded_bias, const int pivoted, float* __restrict__ pivot_io, const int c, double (*red)[2]) {
    // lane 0's per-channel constants are requested BEFORE the slot sums: behind them they were a second, dependent memory round
    // trip of a kernel that is nothing but latency (only this lane reads the running mean - and rewrites it at the end)
    float fb = 0.f, g = 1.f, b = 0.f, rm = 0.f, rv = 0.f, pv = 0.f;
    if (threadIdx.x == 0) {
        if (pivot_io) pv = pivot_io[c];
        // folded_bias: the producer left the layer's bias OUT of the tensor (it cancels in y - mean): the statistics are
        // those of y - bias, and so is the tensor the (scale, shift) pair will be applied to; only the RUNNING mean is that of y
        if (folded_bias) fb = folded_bias[c];
        if (gamma) g = gamma[c];
        if (beta) b = beta[c];
        if (rmean) rm = rmean[c];
        if (rvar) rv = rvar[c];
    }
    double s = 0.0, q = 0.0;
    if (training) slot_sums_wg(stats, nslots, C, c, red, s, q);
    if (threadIdx.x != 0) return;
    double mean, var;
    if (training) {
        // pivoted: the sums are those of (t - pivot), (t - pivot)^2 around pivot = running mean - folded bias (the producer
        // subtracted the same fp32 value per element before squaring): a channel whose spread is tiny against its mean keeps
        // its variance, which E[t^2] - E[t]^2 on fp32 partial sums loses (the reference's ATen BatchNorm is two-pass)
        // (round 6) pivot_io: the caller's own pivot vector - the PREVIOUS batch's mean of y, left there by this kernel a step ago
        // (zero for a fresh or freshly loaded model) - instead of the running mean, which may sit anywhere after a
        // load_state_dict: a pivot ten standard deviations off the batch mean costs var = Q/n - (S/n)^2 two digits
        const double pivot = pivoted ? (double)((pivot_io ? pv : rm) - fb) : 0.0;
        const double ms = s / count;
        mean = pivot + ms;
        var = q / count - ms * ms;
        if (var < 0.0) var = 0.0;
    } else {
        mean = (double)rm - (double)fb;
        var = (double)rv;
    }
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = invstd;
    if (training && rmean && rvar) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rm + momentum * ((float)mean + fb);
        rvar[c] = (1.f - momentum) * rv + momentum * (float)unbiased;
        if (c == 0 && nbt) nbt[0] += 1;
    }
    if (training && pivot_io) pivot_io[c] = (float)mean + fb;        // the next step's pivot: this batch's mean of y
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const double* __restrict__ stats, int nslots, double count, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
    int64_t* nbt, float momentum, float eps, int training, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    const float* __restrict__ folded_bias, int pivoted, float* __restrict__ pivot_io) {
    __shared__ double red[4][2];
    bn_finalize_body(stats, nslots, count, C, gamma, beta, rmean, rvar, nbt, momentum, eps, training, scale, shift, save_mean,
                     save_invstd, folded_bias, pivoted, pivot_io, (int)blockIdx.x, red);
}
// Several independent layers' folds in ONE launch (blockIdx.y = layer, blockIdx.x = channel): a fold is a 5 us launch that
// does 0.5 us of work, and the folds of layers at one dependency depth (mlp1 / shortcut / mlp_rpe1 of an encoder level,
// pool1.mlp / mlp_rpe2) are all wanted at the same moment.
constexpr int BNF_MAX = 24;      // (24 x 120 B of kernel arguments; an eval forward folds its ~45 layers in two launches)
struct BnFoldBatch {
    rl_bn_finalize_item it[BNF_MAX];
};
__global__ __launch_bounds__(256) void bn_finalize_batch_kernel(const BnFoldBatch b) {
    __shared__ double red[4][2];
    const rl_bn_finalize_item& t = b.it[blockIdx.y];
    if ((int)blockIdx.x >= t.C) return;
    bn_finalize_body(t.stats, t.nslots, (double)t.count, t.C, t.gamma, t.beta, t.running_mean, t.running_var, t.num_batches_tracked,
                     t.momentum, t.eps, t.training, t.scale, t.shift, t.save_mean, t.save_invstd, t.folded_bias, t.pivoted, t.pivot,
                     (int)blockIdx.x, red);
}

// (nslots, 2, C) partials -> (2, C) totals, one wavefront per channel, fixed order: the piece a data-parallel caller
// all-reduces between ranks in the SyncBN / equivalence mode before rl_bn_finalize / rl_bn_bwd_finalize (nslots = 1)
__global__ __launch_bounds__(256) void bn_reduce_slots_kernel(const double* __restrict__ stats, int nslots, int C,
                                                              double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s, q;
    slot_sums(stats, nslots, C, c, lane, s, q);
    if (lane == 0) {
        out[c] = s;
        out[C + c] = q;
    }
}

struct BwdParams {
    float* G;
    const float* Y;
    long ld, bstride;
    int n, C;
    long M;
    int contig;
    int act;
    float slope;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* invstd;
    double* stats;
    const float* coef;
    int tile;
};

__device__ __forceinline__ long row_off(const BwdParams& p, long R) {
    if (p.contig) return R * p.ld;
    const int b = (int)((unsigned)R / (unsigned)p.n);   // B * n < 2^31 (checked on the host)
    const int i = (int)(R - (long)b * p.n);
    return ((long)b * p.bstride + i) * p.ld;
}

// rows per tile of the backward sweeps: small tensors get small tiles so that they still spread over
// the chip (a 2560 x 512 tensor in 256-row tiles would run on 10 of 256 CUs)
static inline int bn_tile(long M) {
    int t = 16;
    while (t < 256 && (long)t * 4096 < M) t <<= 1;   // >= 4 tiles per workgroup slot until tiles reach 256 rows
    return t;
}

// thread layout shared by reduce and apply: tpr threads sweep one row, 256/tpr rows in flight
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BwdParams p) {
    __shared__ float red[2][256];
    const int C = p.C;
    const int tpr = C < 256 ? C : 256;
    const int rpar = 256 / tpr;
    const int col = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const bool active = rsub < rpar;
    const long ntiles = (p.M + p.tile - 1) / p.tile;
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const long rend = min(p.M, (tile + 1) * p.tile);
            for (long R = tile * p.tile + rsub; R < rend; R += rpar) {
                const long off = row_off(p, R);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int c = col + it * 256;
                    if (c < C) {
                        const float y = p.Y[off + c];
                        const float sc = p.scale ? p.scale[c] : 1.f, sh = p.shift ? p.shift[c] : 0.f;
                        const float g = p.G[off + c] * rl_act_grad(y * sc + sh, p.act, p.slope);
                        const float xh = (y - p.mean[c]) * p.invstd[c];
                        sg[it] += g;
                        sx[it] += g * xh;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (it * 256 >= C) break;
        __syncthreads();
        red[0][threadIdx.x] = active ? sg[it] : 0.f;
        red[1][threadIdx.x] = active ? sx[it] : 0.f;
        __syncthreads();
        if (threadIdx.x < tpr) {
            const int c = threadIdx.x + it * 256;
            if (c < C) {
                double s = 0.0, q = 0.0;
                for (int j = 0; j < rpar; ++j) {
                    s += (double)red[0][j * tpr + threadIdx.x];
                    q += (double)red[1][j * tpr + threadIdx.x];
                }
                p.stats[((long)blockIdx.x * 2 + 0) * C + c] = s;
                p.stats[((long)blockIdx.x * 2 + 1) * C + c] = q;
            }
        }
    }
}

__device__ __forceinline__ void bn_bwd_finalize_body(const double* __restrict__ stats, int nslots, double count, int C,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ coef, const int c, double (*red)[2]) {
    double s, q;
    slot_sums_wg(stats, nslots, C, c, red, s, q);
    if (threadIdx.x != 0) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    coef[c] = (float)(s / count);
    coef[C + c] = (float)(q / count);
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ stats, int nslots,
                                                              double count, int C, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double red[4][2];
    bn_bwd_finalize_body(stats, nslots, count, C, dgamma, dbeta, coef, (int)blockIdx.x, red);     // one workgroup per channel
}
// the two BatchNorms behind a residual junction (mlp2 and shortcut) in one launch: blockIdx.y = layer
__global__ __launch_bounds__(256) void bn_bwd_finalize_pair_kernel(const double* __restrict__ stats0, const double* __restrict__ stats1,
                                                                   int nslots, double count, int C, float* dgamma0, float* dbeta0,
                                                                   float* coef0, float* dgamma1, float* dbeta1, float* coef1) {
    __shared__ double red[4][2];
    if (blockIdx.y == 0) bn_bwd_finalize_body(stats0, nslots, count, C, dgamma0, dbeta0, coef0, (int)blockIdx.x, red);
    else bn_bwd_finalize_body(stats1, nslots, count, C, dgamma1, dbeta1, coef1, (int)blockIdx.x, red);
}

// several layers' backward finalizes in ONE launch (blockIdx.y = layer, blockIdx.x = channel): the virtual rpe stage whose sums its
// pooling kernel left and the per-point layer whose reduce sweep ran right behind it are wanted at the same moment (round 6)
constexpr int BNBF_MAX = 8;
struct BnBwdFinBatch {
    rl_bn_bwd_finalize_item it[BNBF_MAX];
};
__global__ __launch_bounds__(256) void bn_bwd_finalize_batch_kernel(const BnBwdFinBatch b) {
    __shared__ double red[4][2];
    const rl_bn_bwd_finalize_item& t = b.it[blockIdx.y];
    if ((int)blockIdx.x >= t.C) return;
    bn_bwd_finalize_body(t.stats, t.nslots, (double)t.count, t.C, t.dgamma, t.dbeta, t.coef, (int)blockIdx.x, red);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BwdParams p) {
    const int C = p.C;
    const int tpr = C < 256 ? C : 256;
    const int rpar = 256 / tpr;
    const int col = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    if (rsub >= rpar) return;
    const long ntiles = (p.M + p.tile - 1) / p.tile;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long rend = min(p.M, (tile + 1) * p.tile);
        for (long R = tile * p.tile + rsub; R < rend; R += rpar) {
            const long off = row_off(p, R);
            for (int c = col; c < C; c += 256) {
                const float y = p.Y[off + c];
                const float sc = p.scale ? p.scale[c] : 1.f, sh = p.shift ? p.shift[c] : 0.f;
                float g = p.G[off + c] * rl_act_grad(y * sc + sh, p.act, p.slope);
                if (p.coef) {
                    const float xh = (y - p.mean[c]) * p.invstd[c];
                    g = g - p.coef[c] - xh * p.coef[C + c];
                }
                p.G[off + c] = sc * g;
            }
        }
    }
}

// ---- float4 variants: C % 4 == 0, C/4 a power of two <= 256, 16-byte aligned rows ----------
// tpr = min(C/4, 256) lanes sweep one row (16 B each), 256/tpr rows in flight per workgroup
typedef float bnf4 __attribute__((ext_vector_type(4)));
// Per-thread constants of its channel quad, loaded ONCE (the row loops used to re-load them per row - G is written through
// a plain pointer, so the compiler could not hoist them - and to pass results through pointers to locals, which put them
// in scratch): BatchNorm affine, saved mean / invstd, and the activation derivative as "z > 0 ? 1 : neg".
struct BnQuad {
    bnf4 sc, sh, mu, is;
    float neg;
    __device__ __forceinline__ void load(const BwdParams& p, int c, bool want_xhat) {
        sc = p.scale ? *reinterpret_cast<const bnf4*>(p.scale + c) : (bnf4){1.f, 1.f, 1.f, 1.f};
        sh = p.shift ? *reinterpret_cast<const bnf4*>(p.shift + c) : (bnf4){0.f, 0.f, 0.f, 0.f};
        mu = (bnf4){0.f, 0.f, 0.f, 0.f};
        is = mu;
        if (want_xhat) {
            mu = *reinterpret_cast<const bnf4*>(p.mean + c);
            is = *reinterpret_cast<const bnf4*>(p.invstd + c);
        }
        neg = p.act == RL_ACT_RELU ? 0.f : (p.act == RL_ACT_LRELU ? p.slope : 1.f);     // rl_act_grad: z > 0 ? 1 : neg
    }
    // g = gin * act'(y*sc + sh)      xhat = (y - mean) * invstd
    __device__ __forceinline__ bnf4 grad(const bnf4 y, const bnf4 gin) const {
        const bnf4 z = y * sc + sh;
        bnf4 d;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = z[j] > 0.f ? 1.f : neg;
        return gin * d;
    }
    __device__ __forceinline__ bnf4 xhat(const bnf4 y) const { return (y - mu) * is; }
};

__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const BwdParams p) {
    __shared__ float red[256][9];
    const int C = p.C, c4 = C >> 2;
    const int tpr = c4 < 256 ? c4 : 256;
    const int rpar = 256 / tpr;
    const int q = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int c = q * 4;
    const long ntiles = (p.M + p.tile - 1) / p.tile;
    BnQuad k;
    k.load(p, c, true);
    bnf4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long rend = min(p.M, (tile + 1) * p.tile);
        long R = tile * p.tile + rsub;
        for (; R + rpar < rend; R += 2 * rpar) {          // two rows per trip: four loads in flight per lane
            const long o0 = row_off(p, R) + c, o1 = row_off(p, R + rpar) + c;
            const bnf4 y0 = *reinterpret_cast<const bnf4*>(p.Y + o0), g0 = *reinterpret_cast<const bnf4*>(p.G + o0);
            const bnf4 y1 = *reinterpret_cast<const bnf4*>(p.Y + o1), g1 = *reinterpret_cast<const bnf4*>(p.G + o1);
            const bnf4 d0 = k.grad(y0, g0), d1 = k.grad(y1, g1);
            a0 += d0; a1 += d0 * k.xhat(y0);
            a0 += d1; a1 += d1 * k.xhat(y1);
        }
        if (R < rend) {
            const long o0 = row_off(p, R) + c;
            const bnf4 y0 = *reinterpret_cast<const bnf4*>(p.Y + o0), g0 = *reinterpret_cast<const bnf4*>(p.G + o0);
            const bnf4 d0 = k.grad(y0, g0);
            a0 += d0; a1 += d0 * k.xhat(y0);
        }
    }
    float acc[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    // lanes of one wavefront that share a channel quad (tpr < 64) combine by butterfly
    for (int o = 32; o >= tpr && o >= 1; o >>= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < tpr) {
        // remaining copies of this quad: one per wavefront (tpr < 64) or one per row lane (tpr >= 64)
        const int step = tpr < 64 ? 64 : tpr;
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = threadIdx.x; t < 256; t += step)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += (double)red[t][j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p.stats[((long)blockIdx.x * 2 + 0) * C + c + j] = s[j];
            p.stats[((long)blockIdx.x * 2 + 1) * C + c + j] = s[4 + j];
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const BwdParams p) {
    const int C = p.C, c4 = C >> 2;
    const int tpr = c4 < 256 ? c4 : 256;
    const int rpar = 256 / tpr;
    const int q = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int c = q * 4;
    const bool full = p.coef != nullptr;       // BatchNorm backward (else only the activation derivative and the scale)
    BnQuad k;
    k.load(p, c, full);
    bnf4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0;
    if (full) {
        k0 = *reinterpret_cast<const bnf4*>(p.coef + c);
        k1 = *reinterpret_cast<const bnf4*>(p.coef + C + c);
    }
    const bnf4 osc = p.scale ? k.sc : (bnf4){1.f, 1.f, 1.f, 1.f};
    auto row = [&](const bnf4 y, const bnf4 gi) {
        bnf4 g = k.grad(y, gi);
        if (full) g = g - k0 - k.xhat(y) * k1;
        return g * osc;
    };
    const long ntiles = (p.M + p.tile - 1) / p.tile;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long rend = min(p.M, (tile + 1) * p.tile);
        long R = tile * p.tile + rsub;
        for (; R + rpar < rend; R += 2 * rpar) {          // two rows per trip: four loads in flight per lane
            const long o0 = row_off(p, R) + c, o1 = row_off(p, R + rpar) + c;
            const bnf4 y0 = *reinterpret_cast<const bnf4*>(p.Y + o0), g0 = *reinterpret_cast<const bnf4*>(p.G + o0);
            const bnf4 y1 = *reinterpret_cast<const bnf4*>(p.Y + o1), g1 = *reinterpret_cast<const bnf4*>(p.G + o1);
            *reinterpret_cast<bnf4*>(p.G + o0) = row(y0, g0);
            *reinterpret_cast<bnf4*>(p.G + o1) = row(y1, g1);
        }
        if (R < rend) {
            const long o0 = row_off(p, R) + c;
            const bnf4 y0 = *reinterpret_cast<const bnf4*>(p.Y + o0), g0 = *reinterpret_cast<const bnf4*>(p.G + o0);
            *reinterpret_cast<bnf4*>(p.G + o0) = row(y0, g0);
        }
    }
}

// ---- residual junction: one gradient, two BatchNorms (see rl_resid_bn_bwd_desc) ------------------------------
struct ResidParams {
    float* G; float* G2; const float* O; float slope; long M; int C;
    const float* Y1; const float* sc1; const float* mu1; const float* is1;
    const float* Y2; const float* sc2; const float* mu2; const float* is2;
    double* stats1; double* stats2; const float* coef1; const float* coef2;
    int tile;
};

__global__ __launch_bounds__(256) void resid_bn_bwd_reduce_kernel(const ResidParams p) {
    __shared__ float red[256][13];
    const int C = p.C, c4 = C >> 2;
    const int tpr = c4 < 256 ? c4 : 256;
    const int rpar = 256 / tpr;
    const int q = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int c = q * 4;
    const float4 mu1 = *reinterpret_cast<const float4*>(p.mu1 + c), is1 = *reinterpret_cast<const float4*>(p.is1 + c);
    const float4 mu2 = *reinterpret_cast<const float4*>(p.mu2 + c), is2 = *reinterpret_cast<const float4*>(p.is2 + c);
    const long ntiles = (p.M + p.tile - 1) / p.tile;
    float acc[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long rend = min(p.M, (tile + 1) * p.tile);
        for (long R = tile * p.tile + rsub; R < rend; R += rpar) {
            const long off = R * C + c;
            const float4 gi = *reinterpret_cast<const float4*>(p.G + off);
            const float4 o = *reinterpret_cast<const float4*>(p.O + off);
            const float4 y1 = *reinterpret_cast<const float4*>(p.Y1 + off);
            const float4 y2 = *reinterpret_cast<const float4*>(p.Y2 + off);
            const float g[4] = {o.x > 0.f ? gi.x : gi.x * p.slope, o.y > 0.f ? gi.y : gi.y * p.slope,
                                o.z > 0.f ? gi.z : gi.z * p.slope, o.w > 0.f ? gi.w : gi.w * p.slope};
            const float x1[4] = {(y1.x - mu1.x) * is1.x, (y1.y - mu1.y) * is1.y, (y1.z - mu1.z) * is1.z, (y1.w - mu1.w) * is1.w};
            const float x2[4] = {(y2.x - mu2.x) * is2.x, (y2.y - mu2.y) * is2.y, (y2.z - mu2.z) * is2.z, (y2.w - mu2.w) * is2.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] += g[j];
                acc[4 + j] += g[j] * x1[j];
                acc[8 + j] += g[j] * x2[j];
            }
        }
    }
    for (int o = 32; o >= tpr && o >= 1; o >>= 1) {
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < tpr) {
        const int step = tpr < 64 ? 64 : tpr;
        double s[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = threadIdx.x; t < 256; t += step)
#pragma unroll
            for (int j = 0; j < 12; ++j) s[j] += (double)red[t][j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p.stats1[((long)blockIdx.x * 2 + 0) * C + c + j] = s[j];
            p.stats1[((long)blockIdx.x * 2 + 1) * C + c + j] = s[4 + j];
            p.stats2[((long)blockIdx.x * 2 + 0) * C + c + j] = s[j];
            p.stats2[((long)blockIdx.x * 2 + 1) * C + c + j] = s[8 + j];
        }
    }
}

__global__ __launch_bounds__(256) void resid_bn_bwd_apply_kernel(const ResidParams p) {
    const int C = p.C, c4 = C >> 2;
    const int tpr = c4 < 256 ? c4 : 256;
    const int rpar = 256 / tpr;
    const int q = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int c = q * 4;
    const float4 mu1 = *reinterpret_cast<const float4*>(p.mu1 + c), is1 = *reinterpret_cast<const float4*>(p.is1 + c);
    const float4 mu2 = *reinterpret_cast<const float4*>(p.mu2 + c), is2 = *reinterpret_cast<const float4*>(p.is2 + c);
    const float4 s1 = *reinterpret_cast<const float4*>(p.sc1 + c), s2 = *reinterpret_cast<const float4*>(p.sc2 + c);
    const float4 a0 = *reinterpret_cast<const float4*>(p.coef1 + c), a1 = *reinterpret_cast<const float4*>(p.coef1 + C + c);
    const float4 b0 = *reinterpret_cast<const float4*>(p.coef2 + c), b1 = *reinterpret_cast<const float4*>(p.coef2 + C + c);
    const long ntiles = (p.M + p.tile - 1) / p.tile;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long rend = min(p.M, (tile + 1) * p.tile);
        for (long R = tile * p.tile + rsub; R < rend; R += rpar) {
            const long off = R * C + c;
            const float4 gi = *reinterpret_cast<const float4*>(p.G + off);
            const float4 o = *reinterpret_cast<const float4*>(p.O + off);
            const float4 y1 = *reinterpret_cast<const float4*>(p.Y1 + off);
            const float4 y2 = *reinterpret_cast<const float4*>(p.Y2 + off);
            float4 g;
            g.x = o.x > 0.f ? gi.x : gi.x * p.slope; g.y = o.y > 0.f ? gi.y : gi.y * p.slope;
            g.z = o.z > 0.f ? gi.z : gi.z * p.slope; g.w = o.w > 0.f ? gi.w : gi.w * p.slope;
            float4 r1, r2;
            r1.x = (g.x - a0.x - (y1.x - mu1.x) * is1.x * a1.x) * s1.x; r1.y = (g.y - a0.y - (y1.y - mu1.y) * is1.y * a1.y) * s1.y;
            r1.z = (g.z - a0.z - (y1.z - mu1.z) * is1.z * a1.z) * s1.z; r1.w = (g.w - a0.w - (y1.w - mu1.w) * is1.w * a1.w) * s1.w;
            r2.x = (g.x - b0.x - (y2.x - mu2.x) * is2.x * b1.x) * s2.x; r2.y = (g.y - b0.y - (y2.y - mu2.y) * is2.y * b1.y) * s2.y;
            r2.z = (g.z - b0.z - (y2.z - mu2.z) * is2.z * b1.z) * s2.z; r2.w = (g.w - b0.w - (y2.w - mu2.w) * is2.w * b1.w) * s2.w;
            *reinterpret_cast<float4*>(p.G + off) = r1;
            *reinterpret_cast<float4*>(p.G2 + off) = r2;
        }
    }
}

// ---- small tensors: the whole BatchNorm backward in ONE launch ------------------------------------------------------------
// reduce -> finalize -> apply is three dependent launches of ~5 us each on a tensor of a few thousand rows (level 3, the
// summit, the first decoder stage): 15 us for a microsecond of work.  With SM_R (1 or 2) rows per lane a workgroup of 1024
// lanes can own ONE channel quad and ALL its rows: the rows stay in registers, the batch sums never leave the workgroup
// (a fixed butterfly per wavefront, then the 16 wavefronts in a fixed order, in double), and the result is written by the
// lanes that loaded the operands.  Same expressions per element as the three kernels; the sums group differently (another
// fixed order of the same terms).
constexpr int SM_T = 1024;
// Workgroup -> channel quad.  A workgroup uses 16 bytes of every 128-byte line it touches; the other quads of the line belong
// to other workgroups - which must sit on the SAME XCD, or every XCD's L2 fetches the line for itself (measured: 33 us for
// 5120 x 256, eight times the tensor over the fabric).  Workgroup ids go to the XCDs round-robin (id % 8): XCD x takes the
// contiguous quads [x * per, (x + 1) * per), per = ceil(quads / 8).
__device__ __forceinline__ int small_quad(int C) {
    const int nq = C >> 2, per = (nq + 7) >> 3;
    const int q = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    return ((int)(blockIdx.x >> 3) < per && q < nq) ? q : -1;
}
static inline unsigned small_grid(int C) { return (unsigned)(8 * (((C >> 2) + 7) >> 3)); }
template <int SM_R>
__global__ __launch_bounds__(SM_T) void bn_bwd_small_kernel(const BwdParams p, const double count, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ coef_out) {
    __shared__ float red[SM_T / 64][8];
    const int C = p.C, c = small_quad(C) * 4;
    if (c < 0) return;
    BnQuad k;
    k.load(p, c, true);
    bnf4 d[SM_R], xh[SM_R];
    long off[SM_R];
    bnf4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    {
        bnf4 y[SM_R], g[SM_R];
#pragma unroll
        for (int j = 0; j < SM_R; ++j) {
            const long R = (long)j * SM_T + threadIdx.x;
            const bool in = R < p.M;
            off[j] = in ? row_off(p, R) + c : -1;
            y[j] = in ? *reinterpret_cast<const bnf4*>(p.Y + off[j]) : (bnf4){0.f, 0.f, 0.f, 0.f};
            g[j] = in ? *reinterpret_cast<const bnf4*>(p.G + off[j]) : (bnf4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < SM_R; ++j) {
            d[j] = k.grad(y[j], g[j]);
            xh[j] = k.xhat(y[j]);
            a0 += d[j];
            a1 += d[j] * xh[j];
        }
    }
    float acc[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = rl_wave_sum(acc[j]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x >> 6][j] = acc[j];
    }
    __syncthreads();
    // eight lanes finish the eight sums (the wavefronts in a fixed order, in double - and ONE fp64 division each, not 8192)
    __shared__ __attribute__((aligned(16))) float kk[8];
    if (threadIdx.x < 8) {
        const int j = threadIdx.x;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SM_T / 64; ++w) s += (double)red[w][j];
        const float m = (float)(s / count);
        kk[j] = m;
        float* tot = j < 4 ? dbeta : dgamma;
        if (tot) tot[c + (j & 3)] = (float)s;
        if (coef_out) coef_out[(j < 4 ? 0 : C) + c + (j & 3)] = m;
    }
    __syncthreads();
    const bnf4 k0 = *reinterpret_cast<const bnf4*>(kk), k1 = *reinterpret_cast<const bnf4*>(kk + 4);
    const bnf4 osc = p.scale ? k.sc : (bnf4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < SM_R; ++j)
        if (off[j] >= 0) *reinterpret_cast<bnf4*>(p.G + off[j]) = (d[j] - k0 - xh[j] * k1) * osc;
}

// the same for the residual junction (one gradient, two BatchNorms: resid_bn_bwd_reduce / _apply)
template <int SM_R>
__global__ __launch_bounds__(SM_T) void resid_bn_bwd_small_kernel(const ResidParams p, const double count, float* __restrict__ dgamma1,
                                                                  float* __restrict__ dbeta1, float* __restrict__ dgamma2,
                                                                  float* __restrict__ dbeta2) {
    __shared__ float red[SM_T / 64][12];
    const int C = p.C, c = small_quad(C) * 4;
    if (c < 0) return;
    const bnf4 mu1 = *reinterpret_cast<const bnf4*>(p.mu1 + c), is1 = *reinterpret_cast<const bnf4*>(p.is1 + c);
    const bnf4 mu2 = *reinterpret_cast<const bnf4*>(p.mu2 + c), is2 = *reinterpret_cast<const bnf4*>(p.is2 + c);
    bnf4 g[SM_R], x1[SM_R], x2[SM_R];
    bnf4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
    {
        bnf4 gi[SM_R], o[SM_R], y1[SM_R], y2[SM_R];
#pragma unroll
        for (int j = 0; j < SM_R; ++j) {
            const long R = (long)j * SM_T + threadIdx.x;
            const bool in = R < p.M;
            const long off = in ? R * C + c : 0;
            const bnf4 z = {0.f, 0.f, 0.f, 0.f};
            gi[j] = in ? *reinterpret_cast<const bnf4*>(p.G + off) : z;
            o[j] = in ? *reinterpret_cast<const bnf4*>(p.O + off) : z;
            y1[j] = in ? *reinterpret_cast<const bnf4*>(p.Y1 + off) : z;
            y2[j] = in ? *reinterpret_cast<const bnf4*>(p.Y2 + off) : z;
        }
#pragma unroll
        for (int j = 0; j < SM_R; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[j][e] = o[j][e] > 0.f ? gi[j][e] : gi[j][e] * p.slope;
            x1[j] = (y1[j] - mu1) * is1;
            x2[j] = (y2[j] - mu2) * is2;
            a0 += g[j];
            a1 += g[j] * x1[j];
            a2 += g[j] * x2[j];
        }
    }
    float acc[12] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3]};
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = rl_wave_sum(acc[j]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 12; ++j) red[threadIdx.x >> 6][j] = acc[j];
    }
    __syncthreads();
    __shared__ __attribute__((aligned(16))) float kk[12];      // mean g, mean g*xhat1, mean g*xhat2
    if (threadIdx.x < 12) {
        const int j = threadIdx.x, e = j & 3;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SM_T / 64; ++w) s += (double)red[w][j];
        kk[j] = (float)(s / count);
        if (j < 4) {
            if (dbeta1) dbeta1[c + e] = (float)s;
            if (dbeta2) dbeta2[c + e] = (float)s;
        } else if (j < 8) {
            if (dgamma1) dgamma1[c + e] = (float)s;
        } else if (dgamma2) dgamma2[c + e] = (float)s;
    }
    __syncthreads();
    const bnf4 k0 = *reinterpret_cast<const bnf4*>(kk), k1 = *reinterpret_cast<const bnf4*>(kk + 4), k2 = *reinterpret_cast<const bnf4*>(kk + 8);
    const bnf4 s1 = *reinterpret_cast<const bnf4*>(p.sc1 + c), s2 = *reinterpret_cast<const bnf4*>(p.sc2 + c);
#pragma unroll
    for (int j = 0; j < SM_R; ++j) {
        const long R = (long)j * SM_T + threadIdx.x;
        if (R < p.M) {
            const long off = R * C + c;
            *reinterpret_cast<bnf4*>(p.G + off) = (g[j] - k0 - x1[j] * k1) * s1;
            *reinterpret_cast<bnf4*>(p.G2 + off) = (g[j] - k0 - x2[j] * k2) * s2;
        }
    }
}
// Where it pays: each lane-row costs a 64-byte sector for 16 bytes used, so L2 -> L1 moves four times the tensor; measured inside
// the benchmark's step (5 plain tensors of 1280 - 5120 rows, one junction of 5120 x 512) the one-launch form took 25 us / 61 us
// against 15 us / 35 us for the three launches with 5 rows per lane - it is used up to TWO rows per lane (2048 rows: the
// summit at any batch size, levels 2 - 3 at small per-GPU batches), where the launch floor is all there is to save.
constexpr int SM_ROWS = 2 * SM_T;
constexpr int SM_ROWS_RESID = 2 * SM_T;

int resid_fill(ResidParams* p, const rl_resid_bn_bwd_desc* d, const char* who) {
    RL_REQUIRE(d && d->G && d->O && d->Y1 && d->Y2 && d->rows > 0 && d->C > 0, RL_ERR_ARGS, "%s: bad descriptor", who);
    RL_REQUIRE(d->scale1 && d->mean1 && d->invstd1 && d->scale2 && d->mean2 && d->invstd2, RL_ERR_ARGS, "%s: needs both BatchNorms' scale / mean / invstd", who);
    RL_REQUIRE(rl_resid_bn_bwd_supported(d->rows, d->C), RL_ERR_UNSUPPORTED, "%s: C = %d is not a multiple of 4 with C/4 a power of two <= 256", who, d->C);
    const uintptr_t al = (uintptr_t)d->G | (uintptr_t)d->O | (uintptr_t)d->Y1 | (uintptr_t)d->Y2 | (uintptr_t)d->scale1 | (uintptr_t)d->scale2 |
                         (uintptr_t)d->mean1 | (uintptr_t)d->mean2 | (uintptr_t)d->invstd1 | (uintptr_t)d->invstd2 | (uintptr_t)d->G2 |
                         (uintptr_t)d->coef1 | (uintptr_t)d->coef2;
    RL_REQUIRE((al & 15) == 0, RL_ERR_ARGS, "%s: tensors must be 16-byte aligned", who);
    p->G = d->G; p->G2 = d->G2; p->O = d->O; p->slope = d->slope; p->M = d->rows; p->C = d->C;
    p->Y1 = d->Y1; p->sc1 = d->scale1; p->mu1 = d->mean1; p->is1 = d->invstd1;
    p->Y2 = d->Y2; p->sc2 = d->scale2; p->mu2 = d->mean2; p->is2 = d->invstd2;
    p->stats1 = d->stats1; p->stats2 = d->stats2; p->coef1 = d->coef1; p->coef2 = d->coef2;
    p->tile = bn_tile(d->rows);
    return RL_OK;
}

bool vec_ok(const BwdParams& p) {
    const int c4 = p.C >> 2;
    return (p.C % 4 == 0) && c4 <= 256 && (c4 & (c4 - 1)) == 0 && (p.ld % 4 == 0) &&
           (((uintptr_t)p.G & 15) == 0) && (((uintptr_t)p.Y & 15) == 0);
}

int fill(BwdParams* p, const rl_bn_bwd_desc* d, const char* who) {
    RL_REQUIRE(d && d->G && d->Y && d->B > 0 && d->n > 0 && d->C > 0 && d->C <= 1024, RL_ERR_ARGS, "%s: bad descriptor", who);
    RL_REQUIRE(d->ld >= d->C && d->bstride >= d->n, RL_ERR_ARGS, "%s: bad strides", who);
    RL_REQUIRE((int64_t)d->B * d->n < (1l << 31), RL_ERR_ARGS, "%s: too many rows", who);
    p->G = d->G; p->Y = d->Y; p->ld = d->ld; p->bstride = d->bstride; p->n = d->n; p->C = d->C;
    p->M = (long)d->B * d->n; p->contig = d->bstride == d->n;
    p->act = d->act; p->slope = d->slope; p->scale = d->scale; p->shift = d->shift;
    p->mean = d->mean; p->invstd = d->invstd; p->stats = d->stats; p->coef = d->coef;
    p->tile = bn_tile(p->M);
    return RL_OK;
}

}  // namespace

extern "C" int rl_bn_finalize(const double* stats, int nslots, int64_t count, int C, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                              float momentum, float eps, int training, float* scale, float* shift,
                              float* save_mean, float* save_invstd, const float* folded_bias, int pivoted, float* pivot,
                              void* stream) {
    RL_REQUIRE(C > 0 && scale && shift, RL_ERR_ARGS, "rl_bn_finalize: bad arguments");
    RL_REQUIRE(!pivoted || (training && (running_mean || pivot)), RL_ERR_ARGS,
               "rl_bn_finalize: pivoted statistics need training mode and the pivot (or the running mean)");
    if (training) RL_REQUIRE(stats && nslots > 0 && count > 0, RL_ERR_ARGS, "rl_bn_finalize: training needs partial statistics");
    else RL_REQUIRE(running_mean && running_var, RL_ERR_ARGS, "rl_bn_finalize: eval needs running statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, nslots,
                       (double)count, C, gamma, beta, running_mean, running_var, nbt, momentum, eps, training,
                       scale, shift, save_mean, save_invstd, folded_bias, pivoted, pivot);
    RL_LAUNCH_CHECK("rl_bn_finalize");
    return RL_OK;
}

extern "C" int rl_bn_finalize_batch(const rl_bn_finalize_item* items, int count, void* stream) {
    RL_REQUIRE(items != nullptr && count >= 0, RL_ERR_ARGS, "rl_bn_finalize_batch: bad arguments");
    for (int base = 0; base < count; base += BNF_MAX) {
        BnFoldBatch b;
        const int n = count - base < BNF_MAX ? count - base : BNF_MAX;
        int maxc = 1;
        for (int i = 0; i < n; ++i) {
            const rl_bn_finalize_item& t = items[base + i];
            RL_REQUIRE(t.C > 0 && t.scale && t.shift, RL_ERR_ARGS, "rl_bn_finalize_batch: item %d: bad arguments", base + i);
            if (t.training) RL_REQUIRE(t.stats && t.nslots > 0 && t.count > 0, RL_ERR_ARGS, "rl_bn_finalize_batch: item %d: training needs partial statistics", base + i);
            else RL_REQUIRE(t.running_mean && t.running_var, RL_ERR_ARGS, "rl_bn_finalize_batch: item %d: eval needs running statistics", base + i);
            RL_REQUIRE(!t.pivoted || (t.training && (t.running_mean || t.pivot)), RL_ERR_ARGS,
                       "rl_bn_finalize_batch: item %d: pivoted statistics need training mode and the pivot (or the running mean)", base + i);
            b.it[i] = t;
            maxc = t.C > maxc ? t.C : maxc;
        }
        hipLaunchKernelGGL(bn_finalize_batch_kernel, dim3(maxc, n), dim3(256), 0, (hipStream_t)stream, b);
        RL_LAUNCH_CHECK("rl_bn_finalize_batch");
    }
    return RL_OK;
}

extern "C" int rl_bn_bwd_finalize_batch(const rl_bn_bwd_finalize_item* items, int count, void* stream) {
    RL_REQUIRE(items != nullptr && count > 0 && count <= BNBF_MAX, RL_ERR_ARGS, "rl_bn_bwd_finalize_batch: 1 .. %d items", BNBF_MAX);
    BnBwdFinBatch b;
    int maxc = 1;
    for (int i = 0; i < count; ++i) {
        const rl_bn_bwd_finalize_item& t = items[i];
        RL_REQUIRE(t.stats && t.nslots > 0 && t.count > 0 && t.C > 0 && t.coef, RL_ERR_ARGS, "rl_bn_bwd_finalize_batch: item %d: bad arguments", i);
        b.it[i] = t;
        if (t.C > maxc) maxc = t.C;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_batch_kernel, dim3(maxc, count), dim3(256), 0, (hipStream_t)stream, b);
    RL_LAUNCH_CHECK("rl_bn_bwd_finalize_batch");
    return RL_OK;
}

extern "C" int rl_bn_bwd_finalize_pair(const double* stats0, const double* stats1, int nslots, int64_t count, int C, float* dgamma0,
                                       float* dbeta0, float* coef0, float* dgamma1, float* dbeta1, float* coef1, void* stream) {
    RL_REQUIRE(stats0 && stats1 && nslots > 0 && count > 0 && C > 0 && coef0 && coef1, RL_ERR_ARGS, "rl_bn_bwd_finalize_pair: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_pair_kernel, dim3(C, 2), dim3(256), 0, (hipStream_t)stream, stats0, stats1, nslots, (double)count,
                       C, dgamma0, dbeta0, coef0, dgamma1, dbeta1, coef1);
    RL_LAUNCH_CHECK("rl_bn_bwd_finalize_pair");
    return RL_OK;
}

extern "C" int rl_bn_reduce_slots(const double* stats, int nslots, int C, double* out, void* stream) {
    RL_REQUIRE(stats && out && nslots > 0 && C > 0, RL_ERR_ARGS, "rl_bn_reduce_slots: bad arguments");
    hipLaunchKernelGGL(bn_reduce_slots_kernel, dim3(rl_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, stats, nslots, C, out);
    RL_LAUNCH_CHECK("rl_bn_reduce_slots");
    return RL_OK;
}

extern "C" int rl_bn_bwd_slots(int64_t rows) { return rl_row_blocks_host(rows, bn_tile(rows)); }

extern "C" int rl_bn_bwd_reduce(const rl_bn_bwd_desc* d, void* stream) {
    BwdParams p;
    int rc = fill(&p, d, "rl_bn_bwd_reduce");
    if (rc) return rc;
    RL_REQUIRE(p.stats && p.mean && p.invstd, RL_ERR_ARGS, "rl_bn_bwd_reduce: needs stats/mean/invstd");
    if (vec_ok(p))
        hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel, dim3(rl_row_blocks_host(p.M, p.tile)), dim3(256), 0,
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(rl_row_blocks_host(p.M, p.tile)), dim3(256), 0,
                           (hipStream_t)stream, p);
    rl_note_kernel(vec_ok(p) ? "bn_bwd_reduce_vec_kernel" : "bn_bwd_reduce_kernel");
    RL_LAUNCH_CHECK("rl_bn_bwd_reduce");
    return RL_OK;
}

extern "C" int rl_bn_bwd_finalize(const double* stats, int nslots, int64_t count, int C, float* dgamma,
                                  float* dbeta, float* coef, void* stream) {
    RL_REQUIRE(stats && nslots > 0 && count > 0 && C > 0 && coef, RL_ERR_ARGS, "rl_bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats,
                       nslots, (double)count, C, dgamma, dbeta, coef);
    RL_LAUNCH_CHECK("rl_bn_bwd_finalize");
    return RL_OK;
}

extern "C" int rl_bn_bwd_apply(const rl_bn_bwd_desc* d, void* stream) {
    BwdParams p;
    int rc = fill(&p, d, "rl_bn_bwd_apply");
    if (rc) return rc;
    if (p.coef) RL_REQUIRE(p.mean && p.invstd, RL_ERR_ARGS, "rl_bn_bwd_apply: coef needs mean/invstd");
    if (vec_ok(p))
        hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(rl_row_blocks_host(p.M, p.tile)), dim3(256), 0,
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(rl_row_blocks_host(p.M, p.tile)), dim3(256), 0,
                           (hipStream_t)stream, p);
    rl_note_kernel(vec_ok(p) ? "bn_bwd_apply_vec_kernel" : "bn_bwd_apply_kernel");
    RL_LAUNCH_CHECK("rl_bn_bwd_apply");
    return RL_OK;
}

extern "C" int rl_bn_bwd_fused_supported(int64_t rows, int C, int64_t ld) {
    return (rows > 0 && rows <= SM_ROWS && C > 0 && C % 4 == 0 && ld % 4 == 0 && getenv("RL_NO_BN_SMALL") == nullptr) ? 1 : 0;
}

extern "C" int rl_bn_bwd_fused(const rl_bn_bwd_desc* d, int64_t count, float* dgamma, float* dbeta, float* coef_out, void* stream) {
    BwdParams p;
    int rc = fill(&p, d, "rl_bn_bwd_fused");
    if (rc) return rc;
    RL_REQUIRE(p.mean && p.invstd && count > 0, RL_ERR_ARGS, "rl_bn_bwd_fused: needs mean / invstd and the row count");
    RL_REQUIRE(rl_bn_bwd_fused_supported(p.M, p.C, p.ld) && (((uintptr_t)p.G | (uintptr_t)p.Y) & 15) == 0, RL_ERR_UNSUPPORTED,
               "rl_bn_bwd_fused: %ld rows x %d channels (ld %ld) is not a small 16-byte-aligned tensor - use reduce / finalize / apply", p.M, p.C, p.ld);
    const uintptr_t al = (uintptr_t)p.scale | (uintptr_t)p.shift | (uintptr_t)p.mean | (uintptr_t)p.invstd;
    RL_REQUIRE((al & 15) == 0, RL_ERR_ARGS, "rl_bn_bwd_fused: per-channel vectors must be 16-byte aligned");
    const dim3 grid(small_grid(p.C));
    if (p.M <= SM_T) hipLaunchKernelGGL(bn_bwd_small_kernel<1>, grid, dim3(SM_T), 0, (hipStream_t)stream, p, (double)count, dgamma, dbeta, coef_out);
    else hipLaunchKernelGGL(bn_bwd_small_kernel<2>, grid, dim3(SM_T), 0, (hipStream_t)stream, p, (double)count, dgamma, dbeta, coef_out);
    rl_note_kernel("bn_bwd_small_kernel");
    RL_LAUNCH_CHECK("rl_bn_bwd_fused");
    return RL_OK;
}

extern "C" int rl_resid_bn_bwd_fused_supported(int64_t rows, int C) {
    return (rl_resid_bn_bwd_supported(rows, C) && rows <= SM_ROWS_RESID && getenv("RL_NO_BN_SMALL") == nullptr) ? 1 : 0;
}

extern "C" int rl_resid_bn_bwd_fused(const rl_resid_bn_bwd_desc* d, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, void* stream) {
    ResidParams p;
    int rc = resid_fill(&p, d, "rl_resid_bn_bwd_fused");
    if (rc) return rc;
    RL_REQUIRE(p.G2, RL_ERR_ARGS, "rl_resid_bn_bwd_fused: needs G2");
    RL_REQUIRE(rl_resid_bn_bwd_fused_supported(p.M, p.C), RL_ERR_UNSUPPORTED, "rl_resid_bn_bwd_fused: %ld rows is not a small tensor", p.M);
    const dim3 grid(small_grid(p.C));
    if (p.M <= SM_T) hipLaunchKernelGGL(resid_bn_bwd_small_kernel<1>, grid, dim3(SM_T), 0, (hipStream_t)stream, p, (double)p.M, dgamma1, dbeta1, dgamma2, dbeta2);
    else hipLaunchKernelGGL(resid_bn_bwd_small_kernel<2>, grid, dim3(SM_T), 0, (hipStream_t)stream, p, (double)p.M, dgamma1, dbeta1, dgamma2, dbeta2);
    rl_note_kernel("resid_bn_bwd_small_kernel");
    RL_LAUNCH_CHECK("rl_resid_bn_bwd_fused");
    return RL_OK;
}

extern "C" int rl_resid_bn_bwd_supported(int64_t rows, int C) {
    const int c4 = C >> 2;
    return (rows > 0 && rows < (1l << 31) && C % 4 == 0 && c4 >= 1 && c4 <= 256 && (c4 & (c4 - 1)) == 0) ? 1 : 0;
}

extern "C" int rl_resid_bn_bwd_reduce(const rl_resid_bn_bwd_desc* d, void* stream) {
    ResidParams p;
    int rc = resid_fill(&p, d, "rl_resid_bn_bwd_reduce");
    if (rc) return rc;
    RL_REQUIRE(p.stats1 && p.stats2, RL_ERR_ARGS, "rl_resid_bn_bwd_reduce: needs stats1 and stats2");
    hipLaunchKernelGGL(resid_bn_bwd_reduce_kernel, dim3(rl_row_blocks_host(p.M, p.tile)), dim3(256), 0, (hipStream_t)stream, p);
    rl_note_kernel("resid_bn_bwd_reduce_kernel");
    RL_LAUNCH_CHECK("rl_resid_bn_bwd_reduce");
    return RL_OK;
}

extern "C" int rl_resid_bn_bwd_apply(const rl_resid_bn_bwd_desc* d, void* stream) {
    ResidParams p;
    int rc = resid_fill(&p, d, "rl_resid_bn_bwd_apply");
    if (rc) return rc;
    RL_REQUIRE(p.G2 && p.coef1 && p.coef2, RL_ERR_ARGS, "rl_resid_bn_bwd_apply: needs G2, coef1 and coef2");
    hipLaunchKernelGGL(resid_bn_bwd_apply_kernel, dim3(rl_row_blocks_host(p.M, p.tile)), dim3(256), 0, (hipStream_t)stream, p);
    rl_note_kernel("resid_bn_bwd_apply_kernel");
    RL_LAUNCH_CHECK("rl_resid_bn_bwd_apply");
    return RL_OK;
}
