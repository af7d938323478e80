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
    const int rs = rec_size(C);
    // one wavefront per record entry, lanes stride over the slots (fixed order -> deterministic)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = wave; e < rs; e += 4) {
        double s = 0.0;
        if (mode & 1) {
            for (int i = lane; i < nslots; i += 64) s += work[(long)i * rs + e];
            s = rl_wave_sum(s);
        } else {
            s = work[(long)RL_MAX_SLOTS * rs + e];
        }
        if (lane == 0) {
            tot[e] = s;
            if (mode & 1) work[(long)RL_MAX_SLOTS * rs + e] = s;  // totals record, read by the backward kernel
        }
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
