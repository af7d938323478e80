// Loss, metrics and the optimiser step of the training loop on gfx950:
//   rl_loss_forward / rl_loss_backward : FocalTverskyLoss (dice / tversky / focal_tversky),
//       FocalLoss and cross entropy of the reference (randlanet/utils/losses.py:17-87,
//       trainer.py:244-269) fused with the counts behind accuracy / iou (utils/metrics.py:8-59):
//       one pass over the logits, one packed result -> one host read-back per step instead of
//       the reference's 2C+2 .item() synchronisations (trainer.py:121-131).
//   rl_adam_step : torch.optim.Adam defaults over one flat buffer (trainer.py:78-80, 119).
// logits are (B,C,N) fp32 (the reference's layout, N contiguous -> coalesced per class).
#include "rl_common.h"

namespace {

constexpr int LS_MAXC = 32;
constexpr int LS_ROWS = 256;
#define LS_EPS 1e-7f  // losses.py:4

// per-slot record: [0] tp  [1] sum p  [2] inter(pred==c&label==c)  [3] label count  [4] pred count, each C wide,
// then [5*C] the sum of point-wise losses (cross entropy / focal)
__device__ __forceinline__ int rec_size(int C) { return 5 * C + 1; }

__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       int B, int C, int N, int kind, float gamma,
                                                       double* __restrict__ work) {
    __shared__ double accw[4][5 * LS_MAXC + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 4 * (5 * LS_MAXC + 1); e += 256) (&accw[0][0])[e] = 0.0;
    __syncthreads();
    const long total = (long)B * N;
    const long ntiles = (total + LS_ROWS - 1) / LS_ROWS;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long e = tile * LS_ROWS + tid;
        const bool valid = e < total;
        const long b = valid ? e / N : 0;
        const long i = valid ? e - b * N : 0;
        const float* z = logits + (b * C) * (long)N + i;
        const int lab = valid ? (int)labels[e] : -1;
        float m = -INFINITY;
        int pred = 0;
        for (int c = 0; c < C; ++c) {
            const float v = valid ? z[(long)c * N] : 0.f;
            if (v > m) { m = v; pred = c; }
        }
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf((valid ? z[(long)c * N] : 0.f) - m);
        const float inv = 1.f / den;
        float pl = 0.f;  // point-wise loss
        for (int c = 0; c < C; ++c) {
            const float zc = valid ? z[(long)c * N] : 0.f;
            const float pc = expf(zc - m) * inv;
            const float yc = (lab == c) ? 1.f : 0.f;
            float v0 = valid ? yc * pc : 0.f;
            float v1 = valid ? pc : 0.f;
            float v2 = (valid && pred == c && lab == c) ? 1.f : 0.f;
            float v3 = (valid && lab == c) ? 1.f : 0.f;
            float v4 = (valid && pred == c) ? 1.f : 0.f;
            if (valid) {
                if (kind == 0) {
                    if (lab == c) pl += (logf(den) + m) - zc;
                } else if (kind == 1) {
                    const float yy = fminf(fmaxf(yc, LS_EPS), 1.f - LS_EPS);
                    const float pp = fminf(fmaxf(pc, LS_EPS), 1.f - LS_EPS);
                    pl += -yy * logf(pp) * powf(1.f - pp, gamma);
                }
            }
            v0 = rl_wave_sum(v0); v1 = rl_wave_sum(v1); v2 = rl_wave_sum(v2);
            v3 = rl_wave_sum(v3); v4 = rl_wave_sum(v4);
            if (lane == 0) {
                accw[wave][0 * C + c] += (double)v0;
                accw[wave][1 * C + c] += (double)v1;
                accw[wave][2 * C + c] += (double)v2;
                accw[wave][3 * C + c] += (double)v3;
                accw[wave][4 * C + c] += (double)v4;
            }
        }
        pl = rl_wave_sum(pl);
        if (lane == 0) accw[wave][5 * C] += (double)pl;
    }
    __syncthreads();
    const int rs = rec_size(C);
    for (int e = tid; e < rs; e += 256)
        work[(long)blockIdx.x * rs + e] = accw[0][e] + accw[1][e] + accw[2][e] + accw[3][e];
}

// mode bit 0: sum the slots into the totals record; bit 1: form the loss / counts from the totals record (between the
// two a data-parallel caller may all-reduce the totals record: the "global batch" loss of the equivalence mode)
__global__ __launch_bounds__(256) void loss_finalize_kernel(double* __restrict__ work, int nslots, double points, int C,
                                                            int kind, float alpha, float gamma, int neglect,
                                                            double* __restrict__ out, int mode) {
    __shared__ double tot[5 * LS_MAXC + 1];
    __shared__ double part[4][5 * LS_MAXC + 1];
    const int rs = rec_size(C);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (mode & 1) {
        // every thread owns the slots t, t + 256, ... (all their loads are independent: one memory round trip for the kernel, where
        // a wavefront per entry walked three entries one after the other); fixed order -> deterministic
        // (eight entries at a time: their loads are issued together - entry by entry the wavefront sum of one entry stood between
        // the loads of the next, eleven dependent round trips for two classes)
        for (int e0 = 0; e0 < rs; e0 += 8) {
            double s[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = 0.0;
            for (int i = threadIdx.x; i < nslots; i += 256) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < rs) s[u] += work[(long)i * rs + e0 + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double t = rl_wave_sum(s[u]);
                if (lane == 0 && e0 + u < rs) part[wave][e0 + u] = t;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < rs; e += 256) {
            const double s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
            tot[e] = s;
            work[(long)RL_MAX_SLOTS * rs + e] = s;      // totals record, read by the backward kernel
        }
    } else {
        for (int e = threadIdx.x; e < rs; e += 256) tot[e] = work[(long)RL_MAX_SLOTS * rs + e];
    }
    __syncthreads();
    if (!(mode & 2)) return;
    if (threadIdx.x == 0) {
        double loss;
        if (kind == 2) {
            const int c0 = neglect ? 1 : 0;
            double acc = 0.0;
            for (int c = c0; c < C; ++c) {
                const double tp = tot[c], sp = tot[C + c], sy = tot[3 * C + c];
                const double ti = (tp + (double)LS_EPS) /
                                  (tp + (double)alpha * (sy - tp) + (1.0 - (double)alpha) * (sp - tp) + (double)LS_EPS);
                acc += pow(1.0 - ti, (double)gamma);
            }
            loss = acc / (double)(C - c0);
        } else {
            loss = tot[5 * C] / points;
        }
        out[0] = loss;
        for (int c = 0; c < C; ++c) {
            out[1 + 0 * C + c] = tot[2 * C + c];
            out[1 + 1 * C + c] = tot[3 * C + c];
            out[1 + 2 * C + c] = tot[4 * C + c];
            out[1 + 3 * C + c] = tot[1 * C + c];
        }
    }
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       int B, int C, int N, int kind, float alpha, float gamma, int neglect,
                                                       const double* __restrict__ totals, float grad_scale,
                                                       double norm_points, float* __restrict__ dlogits) {
    __shared__ float cu[LS_MAXC], cw[LS_MAXC];  // dL/dp_c[n] = cu[c]*y_c[n] + cw[c]
    if (threadIdx.x < LS_MAXC) {
        float u = 0.f, w = 0.f;
        const int c = threadIdx.x;
        const int c0 = neglect ? 1 : 0;
        if (kind == 2 && c < C && c >= c0) {
            const double tp = totals[c], sp = totals[C + c], sy = totals[3 * C + c];
            const double D = tp + (double)alpha * (sy - tp) + (1.0 - (double)alpha) * (sp - tp) + (double)LS_EPS;
            const double ti = (tp + (double)LS_EPS) / D;
            const double base = 1.0 - ti;
            const double dl = -((double)gamma / (double)(C - c0)) *
                              ((gamma == 1.f) ? 1.0 : pow(base > 0.0 ? base : 0.0, (double)gamma - 1.0));
            u = (float)(dl / D);
            w = (float)(-dl * (tp + (double)LS_EPS) * (1.0 - (double)alpha) / (D * D));
        }
        cu[c] = u;
        cw[c] = w;
    }
    __syncthreads();
    const long total = (long)B * N;
    const float invn = 1.f / (float)norm_points;          // the mean is over the GLOBAL batch in the equivalence mode
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / N;
        const long i = e - b * N;
        const float* z = logits + (b * C) * (long)N + i;
        float* dz = dlogits + (b * C) * (long)N + i;
        const int lab = (int)labels[e];
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, z[(long)c * N]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(z[(long)c * N] - m);
        const float inv = 1.f / den;
        if (kind == 0) {
            for (int c = 0; c < C; ++c) {
                const float pc = expf(z[(long)c * N] - m) * inv;
                dz[(long)c * N] = (pc - (lab == c ? 1.f : 0.f)) * invn * grad_scale;
            }
            continue;
        }
        // dL/dp_c, then softmax backward dz_c = p_c (dp_c - sum_j p_j dp_j)
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float pc = expf(z[(long)c * N] - m) * inv;
            const float yc = (lab == c) ? 1.f : 0.f;
            float dp;
            if (kind == 2) dp = cu[c] * yc + cw[c];
            else {
                const float yy = fminf(fmaxf(yc, LS_EPS), 1.f - LS_EPS);
                dp = 0.f;
                if (pc >= LS_EPS && pc <= 1.f - LS_EPS)
                    dp = -yy * (powf(1.f - pc, gamma) / pc - gamma * logf(pc) * powf(1.f - pc, gamma - 1.f)) * invn;
            }
            dot += pc * dp;
        }
        for (int c = 0; c < C; ++c) {
            const float pc = expf(z[(long)c * N] - m) * inv;
            const float yc = (lab == c) ? 1.f : 0.f;
            float dp;
            if (kind == 2) dp = cu[c] * yc + cw[c];
            else {
                const float yy = fminf(fmaxf(yc, LS_EPS), 1.f - LS_EPS);
                dp = 0.f;
                if (pc >= LS_EPS && pc <= 1.f - LS_EPS)
                    dp = -yy * (powf(1.f - pc, gamma) / pc - gamma * logf(pc) * powf(1.f - pc, gamma - 1.f)) * invn;
            }
            dz[(long)c * N] = pc * (dp - dot) * grad_scale;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The HEAD of the network as one kernel each way (round 5): Dropout -> fc_end.3 (32 -> C, no BatchNorm) -> un-permute -> loss +
// metric counts  (reference modules.py:525-530, 608-611; losses.py:17-87; metrics.py:8-59), for the fused training step only
// (`_train.TrainStep`): seven launches over a (rows x 32) tensor and (B, C, N) logits - dropout, GEMM, un-permute, loss partials;
// loss backward, permute, input-gradient GEMM, dropout backward, the BatchNorm-backward sums of fc_end.1, fc_end.3's share of the
// narrow weight gradients - become two.  The logits of the step are never stored.
//   rows are in PERMUTED order (row r of cloud b is point perm[r] of that cloud: its label is labels[b][perm[r]]);
//   eight lanes share a row (16 bytes each): the 32 -> C product is 4 FMAs per class and lane + a 3-step butterfly over the octet;
//   the Dropout mask is dropout_kernel's: Philox4x32-10 on (quad index, key | seed) - same bits as the un-fused path;
//   forward: one (5C + 1)-double record per workgroup in loss_fwd_kernel's layout, finished by loss_finalize_kernel;
//   backward: dz from the finished totals (loss_bwd_kernel's formulas), G = (dz . W) * mask -> the gradient w.r.t. fc_end.1's
//   ACTIVATED output, the two BatchNorm-backward sums of fc_end.1 (rl_bn_bwd_reduce's slots) and one partial slab of fc_end.3's
//   weight / bias gradient per workgroup (rl_wgrad_reduce_batch's layout: dW[C][32], then db[C]).
constexpr int HD_REGC = 8;          // classes head_fwd / head_bwd carry in registers; 9 .. HD_MAXC: headw_fwd / headw_bwd (MFMA + LDS)
constexpr int HD_MAXC = 32;         // classes the fused head carries in registers (the metric: 2); more -> the un-fused path
constexpr int HD_K = 32;            // input channels of fc_end.3 (modules.py:529)
struct HeadParams {
    const float* X;                 // (rows, 32) raw output of fc_end.1
    RlLazy lazy;                    // its folded BatchNorm + activation
    const float* mean; const float* invstd;     // backward: saved batch statistics of fc_end.1
    const float* W; const float* bias;          // fc_end.3: [C][32], [C]
    const int64_t* perm; const int64_t* labels;
    long perm_bs;                   // 0: one permutation for all clouds; N: cloud b's own (rl_band_sort)
    int B, N, C;
    int kind; float alpha, gamma; int neglect;
    const int64_t* key; unsigned long long seed; unsigned threshold; float dscale; unsigned long long first_quad; int drop;
    double* work;
    float* G; double* bstats; float* slab; float grad_scale; double norm_points;
    unsigned* mask;                 // (rows) the Dropout keep bits of a row as one word: written by the forward, read by the backward (or null)
};

// this lane's four activated values of row R (lane l of the row's octet holds channels 4l .. 4l+3): x raw, z activated
__device__ __forceinline__ void head_load(const HeadParams& p, long R, int l, float4& x, float4& z) {
    x = *reinterpret_cast<const float4*>(p.X + R * HD_K + 4 * l);
    z.x = rl_lazy(p.lazy, x.x, 4 * l + 0); z.y = rl_lazy(p.lazy, x.y, 4 * l + 1);
    z.z = rl_lazy(p.lazy, x.z, 4 * l + 2); z.w = rl_lazy(p.lazy, x.w, 4 * l + 3);
}
// the keep bits of this lane's quad (bit j: element 4l + j is kept): Philox4x32-10 exactly as rows.hip's dropout_kernel
__device__ __forceinline__ unsigned head_keep_bits(const HeadParams& p, unsigned long long k, long R, int l) {
    const unsigned long long gq = p.first_quad + (unsigned long long)(R * (HD_K / 4) + l);
    uint4 c = make_uint4((unsigned)gq, (unsigned)(gq >> 32), (unsigned)k, (unsigned)(k >> 32));
    uint2 kk = make_uint2((unsigned)p.seed, (unsigned)(p.seed >> 32));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c.z;
        c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ kk.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ kk.y, (unsigned)p0);
        kk.x += 0x9E3779B9u;
        kk.y += 0xBB67AE85u;
    }
    return (c.x >= p.threshold ? 1u : 0u) | (c.y >= p.threshold ? 2u : 0u) | (c.z >= p.threshold ? 4u : 0u) | (c.w >= p.threshold ? 8u : 0u);
}
// Dropout of the quad: v * scale or 0 (like dropout_kernel: not v * keep, which would turn an infinite v into NaN where it is dropped)
__device__ __forceinline__ float4 head_drop(const float4 v, unsigned bits, float scale) {
    return make_float4((bits & 1u) ? v.x * scale : 0.f, (bits & 2u) ? v.y * scale : 0.f, (bits & 4u) ? v.z * scale : 0.f,
                       (bits & 8u) ? v.w * scale : 0.f);
}
// logit c of the row: every lane of the octet gets it (w: this lane's quad of the class' weight row, b: the class' bias)
__device__ __forceinline__ float head_logit(const float4 d, const float4 w, float b) {
    float v = ((d.x * w.x + d.y * w.y) + (d.z * w.z + d.w * w.w));
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v + b;
}
// What a trip needs that does not change between trips lives in registers: this lane's quad of the folded BatchNorm, of every
// class' weight row, the biases, the Dropout key.  Left as loads inside the trip loop the compiler cannot hoist them (the loop stores)
// and each is a dependent round trip behind an s_waitcnt vmcnt(0) - which also waits for the next trip's prefetch.
template <int MC>
struct HeadConst {
    float4 sc, sh;          // lazy scale / shift of channels 4l .. 4l+3 (1 / 0 when the input is already activated)
    bool lazy;
    float4 w[MC];
    float b[MC];
    unsigned long long key;
    __device__ __forceinline__ void load(const HeadParams& p, int l) {
        lazy = p.lazy.scale != nullptr;
        sc = make_float4(1.f, 1.f, 1.f, 1.f); sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lazy) {
            sc = *reinterpret_cast<const float4*>(p.lazy.scale + 4 * l);
            sh = *reinterpret_cast<const float4*>(p.lazy.shift + 4 * l);
        }
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            w[c] = c < p.C ? *reinterpret_cast<const float4*>(p.W + c * HD_K + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[c] = c < p.C ? p.bias[c] : 0.f;
        }
        key = p.drop ? (unsigned long long)p.key[0] : 0ull;
    }
    __device__ __forceinline__ float4 act(const HeadParams& p, const float4 x) const {
        if (!lazy) return x;
        return make_float4(rl_act(x.x * sc.x + sh.x, p.lazy.act, p.lazy.slope), rl_act(x.y * sc.y + sh.y, p.lazy.act, p.lazy.slope),
                           rl_act(x.z * sc.z + sh.z, p.lazy.act, p.lazy.slope), rl_act(x.w * sc.w + sh.w, p.lazy.act, p.lazy.slope));
    }
};

// Work split inside a trip of 32 rows: the eight lanes of a row's octet do what is per CHANNEL (load, BatchNorm + activation,
// the Dropout bits, the 32 -> C product); the logits then cross LDS and ONE lane per row (lanes 0 - 31 of the workgroup) does what
// is per ROW (softmax, loss terms, counts) - in the octet layout that part cost eight times its work, and the per-trip wavefront
// reductions of its sums as much again: the sums stay in that lane's registers (a lane sees ~10 rows) until the end.
// An LDS-only rendezvous: __syncthreads() also waits for every global load in flight (vmcnt(0)) - the loads of the NEXT trip that
// these kernels issue at the top of a trip would have to land before its first barrier.
__device__ __forceinline__ void head_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// The label of row Rr (permuted order) is labels[b][perm[r]]: two DEPENDENT loads.  Issued together they put a full memory round
// trip into the middle of every trip (the row lanes wait for the index before they can ask for the label, the other wavefronts wait
// for the row lanes at the barrier): the index is fetched TWO trips ahead, the label one trip ahead with the index of the trip before.
struct HeadLab {
    unsigned off;       // b * N of the row whose index is in flight / at hand
    long idx;           // perm[r]
};
__device__ __forceinline__ HeadLab head_perm(const HeadParams& p, long Rr, long total) {
    const unsigned Rc = (unsigned)(Rr < total ? Rr : total - 1);        // rows < 2^27 (host check): 32-bit division (a 64-bit one is ~100 instructions)
    const unsigned b = Rc / (unsigned)p.N;
    HeadLab h;
    h.off = b * (unsigned)p.N;
    h.idx = p.perm[(long)b * p.perm_bs + (Rc - h.off)];
    return h;
}
__device__ __forceinline__ int head_label(const HeadParams& p, const HeadLab& h) { return (int)p.labels[(long)h.off + h.idx]; }

template <int MC>      // classes carried in registers: 2, 4 or 8 (the smallest that holds C)
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadParams p) {
    __shared__ float lgs[2][32][MC];
    // a row lane's fp32 sums are moved into doubles every HD_FLUSH trips (the grid is capped at RL_MAX_SLOTS workgroups, so a
    // lane's share grows with the batch: rows / 32768 - thousands at tens of millions of rows; the unfused loss_fwd_kernel
    // promotes per 256-row tile).  Counts stay exact, probability / loss sums keep fp32 rounding over at most 64 terms.
    constexpr int HD_FLUSH = 64;
    __shared__ double accd[32][5 * MC + 1];
    const int tid = threadIdx.x, l = tid & 7;
    const int C = p.C;
    float acc[5 * MC + 1];
#pragma unroll
    for (int e = 0; e < 5 * MC + 1; ++e) acc[e] = 0.f;
    if (tid < 32) {
#pragma unroll
        for (int e = 0; e < 5 * MC + 1; ++e) accd[tid][e] = 0.0;
    }
    int trips = 0;
    const long total = (long)p.B * p.N;
    const long niter = (total + 31) / 32;
    int buf = 0;
    HeadConst<MC> hc;
    hc.load(p, l);
    // one trip ahead: the row's 16 bytes of this lane and (row lanes) the row's label
    long it = blockIdx.x;
    float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
    int labn = 0;
    HeadLab hn = {0u, 0};
    if (it < niter) {
        const long R0 = it * 32 + (tid >> 3);
        xn = *reinterpret_cast<const float4*>(p.X + (R0 < total ? R0 : total - 1) * HD_K + 4 * l);
        if (tid < 32) {
            labn = head_label(p, head_perm(p, it * 32 + tid, total));
            hn = head_perm(p, (it + gridDim.x < niter ? it + gridDim.x : it) * 32 + tid, total);
        }
    }
    for (; it < niter; it += gridDim.x, buf ^= 1) {
        const long R = it * 32 + (tid >> 3);
        const bool valid = R < total;
        const long Rc = valid ? R : total - 1;
        const float4 x = xn;
        const int lab = labn;
        {       // the next trip's loads (past the last trip: this one's again, never used - no branch around a load)
            const long itn = it + gridDim.x < niter ? it + gridDim.x : it;
            const long Rn = itn * 32 + (tid >> 3);
            xn = *reinterpret_cast<const float4*>(p.X + (Rn < total ? Rn : total - 1) * HD_K + 4 * l);
            if (tid < 32) {
                labn = head_label(p, hn);                      // the next trip's label: its index arrived a trip ago
                const long it2 = it + 2 * gridDim.x < niter ? it + 2 * gridDim.x : itn;
                hn = head_perm(p, it2 * 32 + tid, total);      // the index of the trip after next
            }
        }
        const float4 z = hc.act(p, x);
        float4 d = z;
        if (p.drop) {
            unsigned bits = head_keep_bits(p, hc.key, Rc, l);
            d = head_drop(z, bits, p.dscale);
            if (p.mask) {       // the row's 32 keep bits as one word for the backward (bit 4l + j)
                unsigned wbits = bits << (4 * l);
                wbits |= __shfl_xor(wbits, 1, 64); wbits |= __shfl_xor(wbits, 2, 64); wbits |= __shfl_xor(wbits, 4, 64);
                if (l == 0 && valid) p.mask[R] = wbits;
            }
        }
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            if (c >= C) break;
            const float v = head_logit(d, hc.w[c], hc.b[c]);
            if (l == 0) lgs[buf][tid >> 3][c] = v;
        }
        head_lds_barrier();        // (one barrier per trip: the next trip writes the other buffer)
        if (tid < 32) {
            const long Rr = it * 32 + tid;
            if (Rr < total) {
                float lg[MC];
#pragma unroll
                for (int c = 0; c < MC; ++c) lg[c] = c < C ? lgs[buf][tid][c] : -INFINITY;
                float m = -INFINITY;
                int pred = 0;
#pragma unroll
                for (int c = 0; c < MC; ++c)
                    if (c < C && lg[c] > m) { m = lg[c]; pred = c; }
                float den = 0.f;
#pragma unroll
                for (int c = 0; c < MC; ++c)
                    if (c < C) den += expf(lg[c] - m);
                const float inv = 1.f / den;
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    if (c >= C) break;
                    const float zc = lg[c];
                    const float pc = expf(zc - m) * inv;
                    const float yc = (lab == c) ? 1.f : 0.f;
                    acc[0 * MC + c] += yc * pc;
                    acc[1 * MC + c] += pc;
                    acc[2 * MC + c] += (pred == c && lab == c) ? 1.f : 0.f;
                    acc[3 * MC + c] += (lab == c) ? 1.f : 0.f;
                    acc[4 * MC + c] += (pred == c) ? 1.f : 0.f;
                    if (p.kind == 0) {
                        if (lab == c) acc[5 * MC] += (logf(den) + m) - zc;
                    } else if (p.kind == 1) {
                        const float yy = fminf(fmaxf(yc, LS_EPS), 1.f - LS_EPS);
                        const float pp = fminf(fmaxf(pc, LS_EPS), 1.f - LS_EPS);
                        acc[5 * MC] += -yy * logf(pp) * powf(1.f - pp, p.gamma);
                    }
                }
            }
            if (++trips == HD_FLUSH) {        // (the row lanes' own LDS row: no other lane touches it)
                trips = 0;
#pragma unroll
                for (int e = 0; e < 5 * MC + 1; ++e) { accd[tid][e] += (double)acc[e]; acc[e] = 0.f; }
            }
        }
    }
    // the 32 row lanes (half of wavefront 0) hold the workgroup's sums: doubles, a fixed butterfly over those 32 lanes
    if (tid < 64) {
        const int rs = rec_size(C);
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                if (c >= C) break;
                double v = tid < 32 ? accd[tid][k * MC + c] + (double)acc[k * MC + c] : 0.0;
                v = rl_wave_sum(v);
                if (tid == 0) p.work[(long)blockIdx.x * rs + k * C + c] = v;
            }
        double v = tid < 32 ? accd[tid][5 * MC] + (double)acc[5 * MC] : 0.0;
        v = rl_wave_sum(v);
        if (tid == 0) p.work[(long)blockIdx.x * rs + 5 * C] = v;
    }
}

template <int MC>
__global__ __launch_bounds__(256) void head_bwd_kernel(const HeadParams p) {
    __shared__ float cu[MC], cw[MC];  // dL/dp_c[n] = cu[c]*y_c[n] + cw[c]  (loss_bwd_kernel)
    __shared__ float lgs[32][MC], dzs[32][MC];
    __shared__ float red[4][8][MC * 4 + MC + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l = tid & 7;
    const int C = p.C;
    const int rs = rec_size(C);
    const double* totals = p.work + (long)RL_MAX_SLOTS * rs;
    if (tid < MC) {
        float u = 0.f, w = 0.f;
        const int c = tid;
        const int c0 = p.neglect ? 1 : 0;
        if (p.kind == 2 && c < C && c >= c0) {
            const double tp = totals[c], sp = totals[C + c], sy = totals[3 * C + c];
            const double D = tp + (double)p.alpha * (sy - tp) + (1.0 - (double)p.alpha) * (sp - tp) + (double)LS_EPS;
            const double ti = (tp + (double)LS_EPS) / D;
            const double base = 1.0 - ti;
            const double dl = -((double)p.gamma / (double)(C - c0)) *
                              ((p.gamma == 1.f) ? 1.0 : pow(base > 0.0 ? base : 0.0, (double)p.gamma - 1.0));
            u = (float)(dl / D);
            w = (float)(-dl * (tp + (double)LS_EPS) * (1.0 - (double)p.alpha) / (D * D));
        }
        cu[c] = u;
        cw[c] = w;
    }
    __syncthreads();
    const float invn = 1.f / (float)p.norm_points;
    // per-lane partial sums over the rows this lane's octet position sees: dW[c][4l .. 4l+3], db[c] (lane l == 0 only), and the
    // BatchNorm-backward sums of fc_end.1's channels 4l .. 4l+3
    float4 aw[MC];
    float ab[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) { aw[c] = make_float4(0.f, 0.f, 0.f, 0.f); ab[c] = 0.f; }
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sx = sg;
    float4 mu = sg, is = sg;
    if (p.bstats) {
        mu = *reinterpret_cast<const float4*>(p.mean + 4 * l);
        is = *reinterpret_cast<const float4*>(p.invstd + 4 * l);
    }
    HeadConst<MC> hc;
    hc.load(p, l);
    const float4 bsc = hc.sc, bsh = hc.sh;
    const long total = (long)p.B * p.N;
    const long niter = (total + 31) / 32;
    // one trip ahead: this lane's 16 bytes of the row, the row's Dropout bits and (row lanes) its label
    long it = blockIdx.x;
    float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned mwn = 0xFFFFFFFFu;
    int labn = 0;
    HeadLab hn = {0u, 0};
    const bool use_mask = p.drop && p.mask;
    if (it < niter) {
        const long R0 = it * 32 + (tid >> 3), Rq = R0 < total ? R0 : total - 1;
        xn = *reinterpret_cast<const float4*>(p.X + Rq * HD_K + 4 * l);
        if (use_mask) mwn = p.mask[Rq];
        if (tid < 32) {
            labn = head_label(p, head_perm(p, it * 32 + tid, total));
            hn = head_perm(p, (it + gridDim.x < niter ? it + gridDim.x : it) * 32 + tid, total);
        }
    }
    for (; it < niter; it += gridDim.x) {
        const long R = it * 32 + (tid >> 3);
        const bool valid = R < total;
        const long Rc = valid ? R : total - 1;
        const float4 x = xn;
        const unsigned mw = mwn;
        const int lab = labn;
        {       // the next trip's loads (past the last trip: this one's again, never used)
            const long itn = it + gridDim.x < niter ? it + gridDim.x : it;
            const long Rn = itn * 32 + (tid >> 3), Rq = Rn < total ? Rn : total - 1;
            xn = *reinterpret_cast<const float4*>(p.X + Rq * HD_K + 4 * l);
            if (use_mask) mwn = p.mask[Rq];
            if (tid < 32) {
                labn = head_label(p, hn);
                const long it2 = it + 2 * gridDim.x < niter ? it + 2 * gridDim.x : itn;
                hn = head_perm(p, it2 * 32 + tid, total);
            }
        }
        const float4 z = hc.act(p, x);
        unsigned bits = 15u;
        if (p.drop) bits = use_mask ? (mw >> (4 * l)) & 15u : head_keep_bits(p, hc.key, Rc, l);
        const float4 d = p.drop ? head_drop(z, bits, p.dscale) : z;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            if (c >= C) break;
            const float v = head_logit(d, hc.w[c], hc.b[c]);
            if (l == 0) lgs[tid >> 3][c] = v;
        }
        head_lds_barrier();
        if (tid < 32) {         // per ROW: softmax and the loss derivative (loss_bwd_kernel's formulas)
            const long Rr = it * 32 + tid;
            float dz[MC];
#pragma unroll
            for (int c = 0; c < MC; ++c) dz[c] = 0.f;
            if (Rr < total) {
                float lg[MC];
#pragma unroll
                for (int c = 0; c < MC; ++c) lg[c] = c < C ? lgs[tid][c] : -INFINITY;
                float m = -INFINITY;
#pragma unroll
                for (int c = 0; c < MC; ++c)
                    if (c < C) m = fmaxf(m, lg[c]);
                float den = 0.f;
#pragma unroll
                for (int c = 0; c < MC; ++c)
                    if (c < C) den += expf(lg[c] - m);
                const float inv = 1.f / den;
                if (p.kind == 0) {
#pragma unroll
                    for (int c = 0; c < MC; ++c)
                        if (c < C) dz[c] = (expf(lg[c] - m) * inv - (lab == c ? 1.f : 0.f)) * invn * p.grad_scale;
                } else {
                    float dot = 0.f;
                    float dpv[MC], pcv[MC];
#pragma unroll
                    for (int c = 0; c < MC; ++c) {
                        pcv[c] = 0.f; dpv[c] = 0.f;
                        if (c < C) {
                            const float pc = expf(lg[c] - m) * inv;
                            const float yc = (lab == c) ? 1.f : 0.f;
                            float dp;
                            if (p.kind == 2) dp = cu[c] * yc + cw[c];
                            else {
                                const float yy = fminf(fmaxf(yc, LS_EPS), 1.f - LS_EPS);
                                dp = 0.f;
                                if (pc >= LS_EPS && pc <= 1.f - LS_EPS)
                                    dp = -yy * (powf(1.f - pc, p.gamma) / pc - p.gamma * logf(pc) * powf(1.f - pc, p.gamma - 1.f)) * invn;
                            }
                            pcv[c] = pc; dpv[c] = dp;
                            dot += pc * dp;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < MC; ++c)
                        if (c < C) dz[c] = pcv[c] * (dpv[c] - dot) * p.grad_scale;
                }
            }
#pragma unroll
            for (int c = 0; c < MC; ++c) dzs[tid][c] = dz[c];
        }
        head_lds_barrier();
        // per CHANNEL again: fc_end.3's dW[c][k] += dz[c] * d[k], db[c] += dz[c]; dD[k] = sum_c dz[c] * W[c][k]
        float4 dd = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            if (c >= C) break;
            const float dzc = dzs[tid >> 3][c];         // (0 for a row past the end)
            aw[c].x += dzc * d.x; aw[c].y += dzc * d.y; aw[c].z += dzc * d.z; aw[c].w += dzc * d.w;
            if (l == 0) ab[c] += dzc;
            dd.x += dzc * hc.w[c].x; dd.y += dzc * hc.w[c].y; dd.z += dzc * hc.w[c].z; dd.w += dzc * hc.w[c].w;
        }
        // Dropout backward: the gradient w.r.t. fc_end.1's activated output
        const float4 g = p.drop ? head_drop(dd, bits, p.dscale) : dd;
        if (valid) {
            *reinterpret_cast<float4*>(p.G + R * HD_K + 4 * l) = g;
            if (p.bstats) {
                // rl_bn_bwd_reduce's sums for fc_end.1: g' = g * act'(x * scale + shift), xhat = (x - mean) * invstd
                const float g0 = g.x * rl_act_grad(x.x * bsc.x + bsh.x, p.lazy.act, p.lazy.slope);
                const float g1 = g.y * rl_act_grad(x.y * bsc.y + bsh.y, p.lazy.act, p.lazy.slope);
                const float g2 = g.z * rl_act_grad(x.z * bsc.z + bsh.z, p.lazy.act, p.lazy.slope);
                const float g3 = g.w * rl_act_grad(x.w * bsc.w + bsh.w, p.lazy.act, p.lazy.slope);
                sg.x += g0; sg.y += g1; sg.z += g2; sg.w += g3;
                sx.x += g0 * ((x.x - mu.x) * is.x); sx.y += g1 * ((x.y - mu.y) * is.y);
                sx.z += g2 * ((x.z - mu.z) * is.z); sx.w += g3 * ((x.w - mu.w) * is.w);
            }
        }
    }
    // combine: over the eight octets of a wavefront (lanes l, l + 8, ...), then over the four wavefronts in a fixed order
    constexpr int NV = MC * 4 + MC + 8;
    float v[NV];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        v[4 * c + 0] = aw[c].x; v[4 * c + 1] = aw[c].y; v[4 * c + 2] = aw[c].z; v[4 * c + 3] = aw[c].w;
        v[MC * 4 + c] = ab[c];
    }
    v[MC * 5 + 0] = sg.x; v[MC * 5 + 1] = sg.y; v[MC * 5 + 2] = sg.z; v[MC * 5 + 3] = sg.w;
    v[MC * 5 + 4] = sx.x; v[MC * 5 + 5] = sx.y; v[MC * 5 + 6] = sx.z; v[MC * 5 + 7] = sx.w;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float t = v[j];
        t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
        if (lane < 8) red[wave][lane][j] = t;
    }
    __syncthreads();
    if (tid < 8) {
        const int ll = tid;
        float* slab = p.slab + (long)blockIdx.x * (C * HD_K + C);
        for (int c = 0; c < C; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                slab[c * HD_K + 4 * ll + j] = (red[0][ll][4 * c + j] + red[1][ll][4 * c + j]) + (red[2][ll][4 * c + j] + red[3][ll][4 * c + j]);
            if (ll == 0)
                slab[C * HD_K + c] = (red[0][0][MC * 4 + c] + red[1][0][MC * 4 + c]) + (red[2][0][MC * 4 + c] + red[3][0][MC * 4 + c]);
        }
        if (p.bstats) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double a0 = ((double)red[0][ll][MC * 5 + j] + (double)red[1][ll][MC * 5 + j]) +
                                  ((double)red[2][ll][MC * 5 + j] + (double)red[3][ll][MC * 5 + j]);
                const double a1 = ((double)red[0][ll][MC * 5 + 4 + j] + (double)red[1][ll][MC * 5 + 4 + j]) +
                                  ((double)red[2][ll][MC * 5 + 4 + j] + (double)red[3][ll][MC * 5 + 4 + j]);
                p.bstats[((long)blockIdx.x * 2 + 0) * HD_K + 4 * ll + j] = a0;
                p.bstats[((long)blockIdx.x * 2 + 1) * HD_K + 4 * ll + j] = a1;
            }
        }
    }
}

// ---- the fused head for 9 .. 32 classes (round 6: BASELINE configs S / Kt, 13 and 20 classes) -------------------------------------
// head_fwd / head_bwd carry a class' weight row and its partial sums in registers: at 16 / 32 classes that is 256 VGPRs and slower
// than the nine launches it replaces (measured: S 699 -> 678, Kt 351 -> 314 clouds/s).  Here the three small products of a trip of
// 32 rows - logits = D.W^T (32 x 32 x MC), dD = dz.W (32 x MC x 32), dW += dz^T.D (MC x 32 x 32) - run on the matrix pipe in exact
// fp32 (v_mfma_f32_16x16x4_f32, one 16 x 16 tile per wavefront), their operands staged in LDS: D (the row's Dropout output), W,
// the logits, dz, dD.  What is per ROW (softmax, loss terms, counts, dz) stays with one lane per row, reading its row of logits
// from LDS and adding its loss terms straight into its LDS row of doubles.  Same Dropout bits, same label fetch two / one trips
// ahead, same records and slabs as the register kernels: the caller cannot tell which pair ran.
//   MFMA 16x16x4 layouts (as in gemm.hip): A lane l -> A[row l & 15][k l >> 4], B lane l -> B[k l >> 4][col l & 15],
//   D lane l, reg r -> row (l >> 4) * 4 + r, col l & 15.
typedef float hw_f32x4 __attribute__((ext_vector_type(4)));
constexpr int HW_LD = 36;       // LDS row stride of the 32-float staging rows (16-byte aligned, conflict-free column reads)
struct HeadWideConst {
    float4 sc, sh;
    bool lazy;
    unsigned long long key;
    __device__ __forceinline__ void load(const HeadParams& p, int l) {
        lazy = p.lazy.scale != nullptr;
        sc = make_float4(1.f, 1.f, 1.f, 1.f); sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lazy) {
            sc = *reinterpret_cast<const float4*>(p.lazy.scale + 4 * l);
            sh = *reinterpret_cast<const float4*>(p.lazy.shift + 4 * l);
        }
        key = p.drop ? (unsigned long long)p.key[0] : 0ull;
    }
    __device__ __forceinline__ float4 act(const HeadParams& p, const float4 x) const {
        if (!lazy) return x;
        return make_float4(rl_act(x.x * sc.x + sh.x, p.lazy.act, p.lazy.slope), rl_act(x.y * sc.y + sh.y, p.lazy.act, p.lazy.slope),
                           rl_act(x.z * sc.z + sh.z, p.lazy.act, p.lazy.slope), rl_act(x.w * sc.w + sh.w, p.lazy.act, p.lazy.slope));
    }
};
// logits of the trip's 32 rows: the wavefronts below NT = 2 * MC / 16 own one 16 x 16 tile each (rows rb, classes cb)
template <int MC>
__device__ __forceinline__ void headw_logits(const float (*Dl)[HW_LD], const float (&bw)[8], float bias, float (*lgs)[MC + 1],
                                             int wave, int lane) {
    constexpr int NT = 2 * (MC / 16);
    if (wave < NT) {
        const int rb = wave & 1, cb = wave >> 1;
        hw_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Dl[rb * 16 + (lane & 15)][4 * s + (lane >> 4)], bw[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) lgs[rb * 16 + (lane >> 4) * 4 + r][cb * 16 + (lane & 15)] = acc[r] + bias;
    }
}

template <int MC>       // 16 or 32: the smallest that holds C
__global__ __launch_bounds__(256) void headw_fwd_kernel(const HeadParams p) {
    __shared__ __attribute__((aligned(16))) float Dl[32][HW_LD];
    __shared__ float Wl[MC][HW_LD];
    __shared__ float lgs[32][MC + 1];
    __shared__ int labl[32];
    constexpr int J = MC / 8;               // classes per lane of a row's octet: lane l owns classes l, l + 8, ...
    __shared__ double part[4][8][5 * J + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 7;
    const int C = p.C;
    for (int e = tid; e < MC * 32; e += 256) Wl[e >> 5][e & 31] = (e >> 5) < C ? p.W[e] : 0.f;
    __syncthreads();
    // what is per ROW runs on the row's OCTET (lane l: classes l, l + 8, ...; maximum / sums by a 3-step butterfly) - with one lane
    // per row, 32 lanes did all of it while 224 waited.  Sums in fp32 per lane, moved into doubles every HW_FLUSH trips.
    constexpr int HW_FLUSH = 64;
    float acc[5 * J + 1];
    double accd[5 * J + 1];
#pragma unroll
    for (int e = 0; e < 5 * J + 1; ++e) { acc[e] = 0.f; accd[e] = 0.0; }
    int trips = 0;
    constexpr int NT = 2 * (MC / 16);
    float bw[8], bias = 0.f;
    {
        const int cb = wave < NT ? wave >> 1 : 0, cc = cb * 16 + (lane & 15);
#pragma unroll
        for (int s = 0; s < 8; ++s) bw[s] = Wl[cc][4 * s + (lane >> 4)];        // B[k][col] = W[col][k]
        bias = cc < C ? p.bias[cc] : 0.f;
    }
    const long total = (long)p.B * p.N;
    const long niter = (total + 31) / 32;
    HeadWideConst hc;
    hc.load(p, l);
    long it = blockIdx.x;
    float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
    int labn = 0;
    HeadLab hn = {0u, 0};
    if (it < niter) {
        const long R0 = it * 32 + (tid >> 3);
        xn = *reinterpret_cast<const float4*>(p.X + (R0 < total ? R0 : total - 1) * HD_K + 4 * l);
        if (tid < 32) {
            labn = head_label(p, head_perm(p, it * 32 + tid, total));
            hn = head_perm(p, (it + gridDim.x < niter ? it + gridDim.x : it) * 32 + tid, total);
        }
    }
    for (; it < niter; it += gridDim.x) {
        const long R = it * 32 + (tid >> 3);
        const bool valid = R < total;
        const long Rc = valid ? R : total - 1;
        const float4 x = xn;
        const int lab = labn;
        {
            const long itn = it + gridDim.x < niter ? it + gridDim.x : it;
            const long Rn = itn * 32 + (tid >> 3);
            xn = *reinterpret_cast<const float4*>(p.X + (Rn < total ? Rn : total - 1) * HD_K + 4 * l);
            if (tid < 32) {
                labn = head_label(p, hn);
                const long it2 = it + 2 * gridDim.x < niter ? it + 2 * gridDim.x : itn;
                hn = head_perm(p, it2 * 32 + tid, total);
            }
        }
        const float4 z = hc.act(p, x);
        float4 d = z;
        if (p.drop) {
            unsigned bits = head_keep_bits(p, hc.key, Rc, l);
            d = head_drop(z, bits, p.dscale);
            if (p.mask) {
                unsigned wbits = bits << (4 * l);
                wbits |= __shfl_xor(wbits, 1, 64); wbits |= __shfl_xor(wbits, 2, 64); wbits |= __shfl_xor(wbits, 4, 64);
                if (l == 0 && valid) p.mask[R] = wbits;
            }
        }
        *reinterpret_cast<float4*>(&Dl[tid >> 3][4 * l]) = d;
        if (tid < 32) labl[tid] = lab;
        head_lds_barrier();
        headw_logits<MC>(Dl, bw, bias, lgs, wave, lane);
        head_lds_barrier();
        {
            const int row = tid >> 3;
            const int labr = labl[row];
            float lg[J];
            float m = -INFINITY;
            int pred = 0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = l + 8 * j;
                lg[j] = c < C ? lgs[row][c] : -INFINITY;
                if (lg[j] > m) { m = lg[j]; pred = c; }           // (ascending classes: the lowest index among equals, like the row loop)
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                const float mo = __shfl_xor(m, o, 64);
                const int po = __shfl_xor(pred, o, 64);
                if (mo > m || (mo == m && po < pred)) { m = mo; pred = po; }
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (l + 8 * j < C) den += expf(lg[j] - m);
            den += __shfl_xor(den, 1, 64); den += __shfl_xor(den, 2, 64); den += __shfl_xor(den, 4, 64);
            const float inv = 1.f / den;
            if (valid) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int c = l + 8 * j;
                    if (c < C) {
                        const float pc = expf(lg[j] - m) * inv;
                        const float yc = (labr == c) ? 1.f : 0.f;
                        acc[0 * J + j] += yc * pc;
                        acc[1 * J + j] += pc;
                        acc[2 * J + j] += (pred == c && labr == c) ? 1.f : 0.f;
                        acc[3 * J + j] += (labr == c) ? 1.f : 0.f;
                        acc[4 * J + j] += (pred == c) ? 1.f : 0.f;
                        if (p.kind == 0) {
                            if (labr == c) acc[5 * J] += (logf(den) + m) - lg[j];
                        } else if (p.kind == 1) {
                            const float yy = fminf(fmaxf(yc, LS_EPS), 1.f - LS_EPS);
                            const float pp = fminf(fmaxf(pc, LS_EPS), 1.f - LS_EPS);
                            acc[5 * J] += -yy * logf(pp) * powf(1.f - pp, p.gamma);
                        }
                    }
                }
            }
            if (++trips == HW_FLUSH) {
                trips = 0;
#pragma unroll
                for (int e = 0; e < 5 * J + 1; ++e) { accd[e] += (double)acc[e]; acc[e] = 0.f; }
            }
        }
        // (the next trip's Dl / labl stores come behind the two barriers above; its first barrier orders them behind these LDS reads)
    }
    // the octets of a wavefront (lanes l, l + 8, ...) by a fixed butterfly, the four wavefronts through LDS in order; then lane l
    // writes its classes' entries of the workgroup's record, the loss term summed over the eight lanes
#pragma unroll
    for (int e = 0; e < 5 * J + 1; ++e) {
        double v = accd[e] + (double)acc[e];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lane < 8) part[wave][lane][e] = v;
    }
    __syncthreads();
    if (tid < 8) {
        const int rs = rec_size(C);
        double* w = p.work + (long)blockIdx.x * rs;
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = tid + 8 * j;
                if (c < C) w[k * C + c] = (part[0][tid][k * J + j] + part[1][tid][k * J + j]) + (part[2][tid][k * J + j] + part[3][tid][k * J + j]);
            }
        double ls = (part[0][tid][5 * J] + part[1][tid][5 * J]) + (part[2][tid][5 * J] + part[3][tid][5 * J]);
        ls += __shfl_xor(ls, 1, 64); ls += __shfl_xor(ls, 2, 64); ls += __shfl_xor(ls, 4, 64);
        if (tid == 0) w[5 * C] = ls;
    }
}

template <int MC>
__global__ __launch_bounds__(256) void headw_bwd_kernel(const HeadParams p) {
    __shared__ __attribute__((aligned(16))) float Dl[32][HW_LD];
    __shared__ __attribute__((aligned(16))) float dDl[32][HW_LD];
    __shared__ float Wl[MC][HW_LD];
    __shared__ float lgs[32][MC + 1], dzs[32][MC + 1];
    __shared__ float cu[MC], cw[MC];
    __shared__ float red[4][8][8];
    __shared__ int labl[32];
    constexpr int J = MC / 8;               // classes per lane of a row's octet (see headw_fwd_kernel)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 7;
    const int C = p.C;
    const int rs = rec_size(C);
    const double* totals = p.work + (long)RL_MAX_SLOTS * rs;
    for (int e = tid; e < MC * 32; e += 256) Wl[e >> 5][e & 31] = (e >> 5) < C ? p.W[e] : 0.f;
    if (tid < MC) {
        float u = 0.f, w = 0.f;
        const int c = tid;
        const int c0 = p.neglect ? 1 : 0;
        if (p.kind == 2 && c < C && c >= c0) {
            const double tp = totals[c], sp = totals[C + c], sy = totals[3 * C + c];
            const double D = tp + (double)p.alpha * (sy - tp) + (1.0 - (double)p.alpha) * (sp - tp) + (double)LS_EPS;
            const double ti = (tp + (double)LS_EPS) / D;
            const double base = 1.0 - ti;
            const double dl = -((double)p.gamma / (double)(C - c0)) *
                              ((p.gamma == 1.f) ? 1.0 : pow(base > 0.0 ? base : 0.0, (double)p.gamma - 1.0));
            u = (float)(dl / D);
            w = (float)(-dl * (tp + (double)LS_EPS) * (1.0 - (double)p.alpha) / (D * D));
        }
        cu[c] = u;
        cw[c] = w;
    }
    __syncthreads();
    const float invn = 1.f / (float)p.norm_points;
    constexpr int NT = 2 * (MC / 16);       // logits tiles, and dW tiles (MC / 16 class blocks x 2 k blocks)
    // loop-invariant MFMA operands of this wavefront's tiles, in registers
    float bw[8], bias = 0.f;                // logits: B[k][class] = W[class][k]
    float bd[MC / 4];                       // dD:     B[class][k]  = W[class][k]      (tile rows rb2, k block kb2)
    {
        const int cb = wave < NT ? wave >> 1 : 0, cc = cb * 16 + (lane & 15);
#pragma unroll
        for (int s = 0; s < 8; ++s) bw[s] = Wl[cc][4 * s + (lane >> 4)];
        bias = cc < C ? p.bias[cc] : 0.f;
        const int kb2 = wave >> 1;
#pragma unroll
        for (int s = 0; s < MC / 4; ++s) bd[s] = Wl[4 * s + (lane >> 4)][kb2 * 16 + (lane & 15)];
    }
    hw_f32x4 accw = {0.f, 0.f, 0.f, 0.f};   // this wavefront's 16 x 16 tile of dW (class block wave & (MC / 16 - 1), k block wave / (MC / 16))
    float ab = 0.f;                         // db[tid] (tid < MC)
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sx = sg;
    float4 mu = sg, is = sg;
    if (p.bstats) {
        mu = *reinterpret_cast<const float4*>(p.mean + 4 * l);
        is = *reinterpret_cast<const float4*>(p.invstd + 4 * l);
    }
    HeadWideConst hc;
    hc.load(p, l);
    const float4 bsc = hc.sc, bsh = hc.sh;
    const long total = (long)p.B * p.N;
    const long niter = (total + 31) / 32;
    long it = blockIdx.x;
    float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned mwn = 0xFFFFFFFFu;
    int labn = 0;
    HeadLab hn = {0u, 0};
    const bool use_mask = p.drop && p.mask;
    if (it < niter) {
        const long R0 = it * 32 + (tid >> 3), Rq = R0 < total ? R0 : total - 1;
        xn = *reinterpret_cast<const float4*>(p.X + Rq * HD_K + 4 * l);
        if (use_mask) mwn = p.mask[Rq];
        if (tid < 32) {
            labn = head_label(p, head_perm(p, it * 32 + tid, total));
            hn = head_perm(p, (it + gridDim.x < niter ? it + gridDim.x : it) * 32 + tid, total);
        }
    }
    for (; it < niter; it += gridDim.x) {
        const long R = it * 32 + (tid >> 3);
        const bool valid = R < total;
        const long Rc = valid ? R : total - 1;
        const float4 x = xn;
        const unsigned mw = mwn;
        const int lab = labn;
        {
            const long itn = it + gridDim.x < niter ? it + gridDim.x : it;
            const long Rn = itn * 32 + (tid >> 3), Rq = Rn < total ? Rn : total - 1;
            xn = *reinterpret_cast<const float4*>(p.X + Rq * HD_K + 4 * l);
            if (use_mask) mwn = p.mask[Rq];
            if (tid < 32) {
                labn = head_label(p, hn);
                const long it2 = it + 2 * gridDim.x < niter ? it + 2 * gridDim.x : itn;
                hn = head_perm(p, it2 * 32 + tid, total);
            }
        }
        const float4 z = hc.act(p, x);
        unsigned bits = 15u;
        if (p.drop) bits = use_mask ? (mw >> (4 * l)) & 15u : head_keep_bits(p, hc.key, Rc, l);
        const float4 d = p.drop ? head_drop(z, bits, p.dscale) : z;
        *reinterpret_cast<float4*>(&Dl[tid >> 3][4 * l]) = d;
        if (tid < 32) labl[tid] = lab;
        head_lds_barrier();
        headw_logits<MC>(Dl, bw, bias, lgs, wave, lane);
        head_lds_barrier();
        {       // per ROW, on the row's octet: softmax and the loss derivative (loss_bwd_kernel's formulas), dz of every class to LDS
            const int row = tid >> 3;
            const int labr = labl[row];
            float lg[J], pc[J], dp[J];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                lg[j] = l + 8 * j < C ? lgs[row][l + 8 * j] : -INFINITY;
                m = fmaxf(m, lg[j]);
            }
            m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                pc[j] = l + 8 * j < C ? expf(lg[j] - m) : 0.f;
                den += pc[j];
            }
            den += __shfl_xor(den, 1, 64); den += __shfl_xor(den, 2, 64); den += __shfl_xor(den, 4, 64);
            const float inv = 1.f / den;
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = l + 8 * j;
                pc[j] *= inv;
                dp[j] = 0.f;
                if (c < C) {
                    const float yc = (labr == c) ? 1.f : 0.f;
                    if (p.kind == 0) dp[j] = (pc[j] - yc) * invn;          // (dz itself for cross entropy)
                    else if (p.kind == 2) dp[j] = cu[c] * yc + cw[c];
                    else {
                        const float yy = fminf(fmaxf(yc, LS_EPS), 1.f - LS_EPS);
                        if (pc[j] >= LS_EPS && pc[j] <= 1.f - LS_EPS)
                            dp[j] = -yy * (powf(1.f - pc[j], p.gamma) / pc[j] - p.gamma * logf(pc[j]) * powf(1.f - pc[j], p.gamma - 1.f)) * invn;
                    }
                    dot += pc[j] * dp[j];
                }
            }
            dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = l + 8 * j;
                float dz = 0.f;
                if (valid && c < C) dz = (p.kind == 0 ? dp[j] : pc[j] * (dp[j] - dot)) * p.grad_scale;
                dzs[row][c] = dz;           // (c < MC always: the classes past C and the rows past the end hold zeros)
            }
        }
        head_lds_barrier();
        {
            // dD[row][k] = sum_c dz[row][c] W[c][k]: four 16 x 16 tiles (rows wave & 1, k block wave >> 1), MC / 4 steps
            const int rb2 = wave & 1, kb2 = wave >> 1;
            hw_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < MC / 4; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dzs[rb2 * 16 + (lane & 15)][4 * s + (lane >> 4)], bd[s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) dDl[rb2 * 16 + (lane >> 4) * 4 + r][kb2 * 16 + (lane & 15)] = acc[r];
            // dW[c][k] += sum_rows dz[row][c] D[row][k]: NT tiles, 8 steps over the trip's 32 rows
            if (wave < NT) {
                const int cbw = wave & (MC / 16 - 1), kbw = wave / (MC / 16);
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    accw = __builtin_amdgcn_mfma_f32_16x16x4f32(dzs[4 * s + (lane >> 4)][cbw * 16 + (lane & 15)],
                                                               Dl[4 * s + (lane >> 4)][kbw * 16 + (lane & 15)], accw, 0, 0, 0);
            }
            if (tid < MC) {
                float t = 0.f;
#pragma unroll 8
                for (int r = 0; r < 32; ++r) t += dzs[r][tid];
                ab += t;
            }
        }
        head_lds_barrier();
        const float4 dd = *reinterpret_cast<const float4*>(&dDl[tid >> 3][4 * l]);
        const float4 g = p.drop ? head_drop(dd, bits, p.dscale) : dd;
        if (valid) {
            *reinterpret_cast<float4*>(p.G + R * HD_K + 4 * l) = g;
            if (p.bstats) {
                const float g0 = g.x * rl_act_grad(x.x * bsc.x + bsh.x, p.lazy.act, p.lazy.slope);
                const float g1 = g.y * rl_act_grad(x.y * bsc.y + bsh.y, p.lazy.act, p.lazy.slope);
                const float g2 = g.z * rl_act_grad(x.z * bsc.z + bsh.z, p.lazy.act, p.lazy.slope);
                const float g3 = g.w * rl_act_grad(x.w * bsc.w + bsh.w, p.lazy.act, p.lazy.slope);
                sg.x += g0; sg.y += g1; sg.z += g2; sg.w += g3;
                sx.x += g0 * ((x.x - mu.x) * is.x); sx.y += g1 * ((x.y - mu.y) * is.y);
                sx.z += g2 * ((x.z - mu.z) * is.z); sx.w += g3 * ((x.w - mu.w) * is.w);
            }
        }
        // (the next trip's first barrier orders its Dl / dzs / dDl stores behind the LDS reads above)
    }
    // this workgroup's slab: dW[C][32] from the tile accumulators, db[C]; the BatchNorm-backward sums as in head_bwd_kernel
    float* slab = p.slab + (long)blockIdx.x * (C * HD_K + C);
    if (wave < NT) {
        const int cbw = wave & (MC / 16 - 1), kbw = wave / (MC / 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = cbw * 16 + (lane >> 4) * 4 + r;
            if (c < C) slab[c * HD_K + kbw * 16 + (lane & 15)] = accw[r];
        }
    }
    if (tid < C) slab[C * HD_K + tid] = ab;
    if (p.bstats) {
        float v[8] = {sg.x, sg.y, sg.z, sg.w, sx.x, sx.y, sx.z, sx.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = v[j];
            t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            if (lane < 8) red[wave][lane][j] = t;
        }
        __syncthreads();
        if (tid < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double a0 = ((double)red[0][tid][j] + (double)red[1][tid][j]) + ((double)red[2][tid][j] + (double)red[3][tid][j]);
                const double a1 = ((double)red[0][tid][4 + j] + (double)red[1][tid][4 + j]) + ((double)red[2][tid][4 + j] + (double)red[3][tid][4 + j]);
                p.bstats[((long)blockIdx.x * 2 + 0) * HD_K + 4 * tid + j] = a0;
                p.bstats[((long)blockIdx.x * 2 + 1) * HD_K + 4 * tid + j] = a1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ lr, float b1,
                                                   float b2, float eps, float gscale, const int64_t* __restrict__ step) {
    // step[0] was incremented by adam_tick_kernel, earlier on the same stream
    const double t = (double)step[0];
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    const float step_size = lr[0] / bc1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float gr = g[e] * gscale;
        const float mm = b1 * m[e] + (1.f - b1) * gr;
        const float vv = b2 * v[e] + (1.f - b2) * gr * gr;
        m[e] = mm;
        v[e] = vv;
        p[e] -= step_size * (mm / (sqrtf(vv) / bc2s + eps));
    }
}
// confidences = softmax over the class axis of (B,C,N) logits (Model.upsample / Model.predict, model.py:137, 229)
__global__ __launch_bounds__(256) void softmax_cf_kernel(const float* __restrict__ logits, int B, int C, int N,
                                                         float* __restrict__ out) {
    const long total = (long)B * N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / N;
        const long i = e - b * N;
        const float* z = logits + (b * C) * (long)N + i;
        float* o = out + (b * C) * (long)N + i;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, z[(long)c * N]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(z[(long)c * N] - m);
        for (int c = 0; c < C; ++c) o[(long)c * N] = expf(z[(long)c * N] - m) / den;
    }
}

__global__ void adam_tick_kernel(int64_t* step) { step[0] += 1; }

}  // namespace

extern "C" int64_t rl_loss_work_doubles(int64_t points, int C) {
    (void)points;
    return (int64_t)(RL_MAX_SLOTS + 1) * (5 * C + 1);
}

static int loss_check(const char* who, const void* logits, const void* labels, int B, int C, int N, int kind) {
    RL_REQUIRE(logits && labels && B > 0 && N > 0 && C > 0, RL_ERR_ARGS, "%s: bad arguments", who);
    RL_REQUIRE(C <= LS_MAXC, RL_ERR_UNSUPPORTED, "%s: C=%d exceeds %d classes", who, C, LS_MAXC);
    RL_REQUIRE(kind >= 0 && kind <= 2, RL_ERR_ARGS, "%s: unknown loss kind %d", who, kind);
    return RL_OK;
}

static int loss_forward_impl(const float* logits, const int64_t* labels, int B, int C, int N, int kind, float alpha,
                             float gamma, int neglect_background, double* work, double* out, int mode, double points,
                             void* stream, const char* who) {
    RL_REQUIRE(work, RL_ERR_ARGS, "%s: null work", who);
    RL_REQUIRE(C > 0 && C <= LS_MAXC, RL_ERR_UNSUPPORTED, "%s: C=%d outside 1..%d classes", who, C, LS_MAXC);
    RL_REQUIRE(kind >= 0 && kind <= 2, RL_ERR_ARGS, "%s: unknown loss kind %d", who, kind);
    RL_REQUIRE(!(kind == 2 && neglect_background && C < 2), RL_ERR_ARGS, "%s: needs a foreground class", who);
    hipStream_t st = (hipStream_t)stream;
    int nslots = 0;
    if (mode & 1) {
        int rc = loss_check(who, logits, labels, B, C, N, kind);
        if (rc) return rc;
        nslots = rl_row_blocks_host((long)B * N, LS_ROWS);
        hipLaunchKernelGGL(loss_fwd_kernel, dim3(nslots), dim3(256), 0, st, logits, labels, B, C, N, kind, gamma, work);
        rl_note_kernel("loss_fwd_kernel");
        RL_LAUNCH_CHECK(who);
    }
    if (mode & 2) RL_REQUIRE(out && points > 0, RL_ERR_ARGS, "%s: null out / bad point count", who);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, work, nslots, points, C, kind, alpha, gamma,
                       neglect_background, out, mode);
    RL_LAUNCH_CHECK(who);
    return RL_OK;
}

extern "C" int rl_loss_forward(const float* logits, const int64_t* labels, int B, int C, int N, int kind, float alpha,
                               float gamma, int neglect_background, double* work, double* out, void* stream) {
    return loss_forward_impl(logits, labels, B, C, N, kind, alpha, gamma, neglect_background, work, out, 3,
                             (double)B * (double)N, stream, "rl_loss_forward");
}

extern "C" int rl_loss_partials(const float* logits, const int64_t* labels, int B, int C, int N, int kind, float gamma,
                                double* work, void* stream) {
    return loss_forward_impl(logits, labels, B, C, N, kind, 0.5f, gamma, 0, work, nullptr, 1, 1.0, stream, "rl_loss_partials");
}

extern "C" int64_t rl_loss_totals_offset(int C) { return (int64_t)RL_MAX_SLOTS * (5 * C + 1); }

extern "C" int rl_loss_from_totals(int64_t points_total, int C, int kind, float alpha, float gamma, int neglect_background,
                                   double* work, double* out, void* stream) {
    return loss_forward_impl(nullptr, nullptr, 0, C, 0, kind, alpha, gamma, neglect_background, work, out, 2,
                             (double)points_total, stream, "rl_loss_from_totals");
}

static int loss_backward_impl(const float* logits, const int64_t* labels, int B, int C, int N, int kind, float alpha,
                              float gamma, int neglect_background, const double* work, float grad_scale,
                              double norm_points, float* dlogits, void* stream) {
    int rc = loss_check("rl_loss_backward", logits, labels, B, C, N, kind);
    if (rc) return rc;
    RL_REQUIRE(work && dlogits && norm_points > 0, RL_ERR_ARGS, "rl_loss_backward: null work/dlogits");
    const long total = (long)B * N;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    const double* totals = work + (long)RL_MAX_SLOTS * (5 * C + 1);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, logits, labels, B, C, N, kind,
                       alpha, gamma, neglect_background, totals, grad_scale, norm_points, dlogits);
    rl_note_kernel("loss_bwd_kernel");
    RL_LAUNCH_CHECK("rl_loss_backward");
    return RL_OK;
}

extern "C" int rl_loss_backward(const float* logits, const int64_t* labels, int B, int C, int N, int kind, float alpha,
                                float gamma, int neglect_background, const double* work, float grad_scale,
                                float* dlogits, void* stream) {
    return loss_backward_impl(logits, labels, B, C, N, kind, alpha, gamma, neglect_background, work, grad_scale,
                              (double)B * (double)N, dlogits, stream);
}

extern "C" int rl_loss_backward_global(const float* logits, const int64_t* labels, int B, int C, int N, int kind, float alpha,
                                       float gamma, int neglect_background, const double* work, float grad_scale,
                                       int64_t points_total, float* dlogits, void* stream) {
    return loss_backward_impl(logits, labels, B, C, N, kind, alpha, gamma, neglect_background, work, grad_scale,
                              (double)points_total, dlogits, stream);
}

// ---- the fused head (Dropout -> fc_end.3 -> un-permute -> loss / counts), see head_fwd_kernel --------------------------------
static int head_grid(long rows) {
    long g = (rows + 31) / 32;            // 32 rows per workgroup and trip
    g = (g + 7) / 8;                      // >= 8 trips per workgroup where there is that much work ...
    if (g > RL_MAX_SLOTS) g = RL_MAX_SLOTS;      // ... and never more slots than rl_bn_bwd_finalize / the loss finalize read
    return (int)(g < 1 ? 1 : g);
}
extern "C" int rl_head_supported(int C, int K) { return (C >= 1 && C <= HD_MAXC && K == HD_K) ? 1 : 0; }
extern "C" int rl_head_grid(int64_t rows) { return head_grid(rows); }

static int head_fill(HeadParams* p, const rl_head_desc* d, const char* who, bool backward) {
    RL_REQUIRE(d && d->X && d->W && d->bias && d->perm && d->labels && d->work, RL_ERR_ARGS, "%s: null pointer", who);
    RL_REQUIRE(d->B > 0 && d->N > 0 && (int64_t)d->B * d->N * HD_K < (1l << 32), RL_ERR_ARGS, "%s: bad sizes", who);
    RL_REQUIRE(rl_head_supported(d->C, HD_K), RL_ERR_UNSUPPORTED, "%s: 1 .. %d classes (got %d)", who, HD_MAXC, d->C);
    RL_REQUIRE(d->loss_kind >= 0 && d->loss_kind <= 2, RL_ERR_ARGS, "%s: unknown loss kind %d", who, d->loss_kind);
    RL_REQUIRE(!(d->loss_kind == 2 && d->neglect_background && d->C < 2), RL_ERR_ARGS, "%s: needs a foreground class", who);
    RL_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), RL_ERR_ARGS, "%s: scale / shift must come together", who);
    RL_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f && (d->drop_p == 0.f || d->drop_key), RL_ERR_ARGS, "%s: bad Dropout arguments", who);
    RL_REQUIRE((((uintptr_t)d->X | (uintptr_t)d->W | (uintptr_t)d->G) & 15) == 0, RL_ERR_ARGS, "%s: X / W / G must be 16-byte aligned", who);
    p->X = d->X;
    p->lazy.scale = d->scale; p->lazy.shift = d->shift; p->lazy.act = d->act; p->lazy.slope = d->slope;
    p->mean = d->mean; p->invstd = d->invstd;
    p->W = d->W; p->bias = d->bias; p->perm = d->perm; p->labels = d->labels;
    RL_REQUIRE(d->perm_bstride == 0 || d->perm_bstride >= d->N, RL_ERR_ARGS, "%s: perm_bstride must be 0 or >= N", who);
    p->perm_bs = (long)d->perm_bstride;
    p->B = d->B; p->N = d->N; p->C = d->C;
    p->kind = d->loss_kind; p->alpha = d->alpha; p->gamma = d->gamma; p->neglect = d->neglect_background;
    p->drop = d->drop_p > 0.f;
    p->key = d->drop_key; p->seed = d->drop_seed;
    const double t = (double)d->drop_p * 4294967296.0;
    p->threshold = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    p->dscale = 1.0f / (1.0f - d->drop_p);
    p->first_quad = (unsigned long long)d->drop_first_row * (HD_K / 4);
    p->work = d->work;
    p->G = d->G; p->bstats = d->bn_bwd_stats; p->slab = d->slab;
    p->grad_scale = d->grad_scale; p->norm_points = (double)d->B * (double)d->N;
    p->mask = reinterpret_cast<unsigned*>(d->drop_mask);
    if (backward) {
        RL_REQUIRE(d->G && d->slab && d->slab_floats >= (int64_t)head_grid((long)d->B * d->N) * (d->C * HD_K + d->C), RL_ERR_ARGS,
                   "%s: needs G and a slab of rl_head_grid(rows) * (C*32 + C) floats", who);
        RL_REQUIRE(!d->bn_bwd_stats || (d->scale && d->mean && d->invstd), RL_ERR_ARGS,
                   "%s: bn_bwd_stats needs the folded BatchNorm and its saved mean / invstd", who);
    }
    return RL_OK;
}

extern "C" int rl_head_fwd(const rl_head_desc* d, double* out, void* stream) {
    HeadParams p;
    int rc = head_fill(&p, d, "rl_head_fwd", false);
    if (rc) return rc;
    RL_REQUIRE(out, RL_ERR_ARGS, "rl_head_fwd: null out");
    hipStream_t st = (hipStream_t)stream;
    const int g = head_grid((long)d->B * d->N);
    // (the MFMA kernels at 2 classes, measured: 6.353 -> 6.374 ms per step - up to 8 classes the register kernels stay)
    if (d->C <= 2) hipLaunchKernelGGL(head_fwd_kernel<2>, dim3(g), dim3(256), 0, st, p);
    else if (d->C <= 4) hipLaunchKernelGGL(head_fwd_kernel<4>, dim3(g), dim3(256), 0, st, p);
    else if (d->C <= 8) hipLaunchKernelGGL(head_fwd_kernel<8>, dim3(g), dim3(256), 0, st, p);
    else if (d->C <= 16) hipLaunchKernelGGL(headw_fwd_kernel<16>, dim3(g), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(headw_fwd_kernel<32>, dim3(g), dim3(256), 0, st, p);
    rl_note_kernel("head_fwd_kernel");
    RL_LAUNCH_CHECK("rl_head_fwd");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, d->work, g, (double)d->B * (double)d->N, d->C, d->loss_kind,
                       d->alpha, d->gamma, d->neglect_background, out, 3);
    RL_LAUNCH_CHECK("rl_head_fwd(finalize)");
    return RL_OK;
}

extern "C" int rl_head_bwd(const rl_head_desc* d, void* stream) {
    HeadParams p;
    int rc = head_fill(&p, d, "rl_head_bwd", true);
    if (rc) return rc;
    const int g = head_grid((long)d->B * d->N);
    if (d->C <= 2) hipLaunchKernelGGL(head_bwd_kernel<2>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    else if (d->C <= 4) hipLaunchKernelGGL(head_bwd_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    else if (d->C <= 8) hipLaunchKernelGGL(head_bwd_kernel<8>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    else if (d->C <= 16) hipLaunchKernelGGL(headw_bwd_kernel<16>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(headw_bwd_kernel<32>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    rl_note_kernel("head_bwd_kernel");
    RL_LAUNCH_CHECK("rl_head_bwd");
    return RL_OK;
}

extern "C" int rl_softmax_cf(const float* logits, int B, int C, int N, float* out, void* stream) {
    RL_REQUIRE(logits && out && B > 0 && C > 0 && N > 0, RL_ERR_ARGS, "rl_softmax_cf: bad arguments");
    long g = ((long)B * N + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(softmax_cf_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, logits, B, C, N, out);
    RL_LAUNCH_CHECK("rl_softmax_cf");
    return RL_OK;
}

extern "C" int rl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                            const float* lr, float beta1, float beta2, float eps, float grad_scale, int64_t* step,
                            void* stream) {
    RL_REQUIRE(param && grad && exp_avg && exp_avg_sq && lr && step && n >= 0, RL_ERR_ARGS, "rl_adam_step: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step);
    RL_LAUNCH_CHECK("rl_adam_tick");
    if (n == 0) return RL_OK;
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((int)g), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, (long)n, lr,
                       beta1, beta2, eps, grad_scale, step);
    rl_note_kernel("adam_kernel");
    RL_LAUNCH_CHECK("rl_adam_step");
    return RL_OK;
}
