// Row movement and the neighbourhood-pooling element-wise kernels of the RandLA-Net path:
//   rl_copy_rows / rl_scatter_add_rows : permutation + prefix slicing (reference
//       randlanet/utils/modules.py:571-573, 608), PointFeatureAugmentation gather + concat
//       (modules.py:213-221), nearest-neighbour interpolation gather + skip concat
//       (modules.py:359-364, 600-602), and the transposed (gradient) movements
//   rl_attpool_fwd / _bwd : softmax over the K neighbours + weighted sum (modules.py:246-253)
//   rl_add_act_fwd / _bwd : residual sum of two BatchNorm outputs + LeakyReLU (modules.py:325)
//   rl_scale_mask : Dropout(0.5) of fc_end (modules.py:528) with a caller-drawn keep mask
//   rl_logits_unpermute / rl_logits_permute_grad : inverse permutation + (B,C,N) layout
//       (modules.py:608-611)
// All are HBM-bound: channel-last rows make every access a contiguous C-float segment.
#include "rl_common.h"

namespace {

struct RowsParams {
    const float* src;
    long lds, src_bstride;
    float* dst;
    long ldd;
    long rows, rows_per_batch;
    int C;
    const int32_t* i32;
    const int64_t* i64;
    int shared;
    int accumulate;
    RlLazy lazy;
};

// I = uint32_t when the element count fits 32 bits (64-bit integer division is emulated with
// ~100 instructions on CDNA; two of them per 16 bytes moved made these kernels ALU-bound), else int64_t
template <typename I>
__device__ __forceinline__ long src_row(const RowsParams& p, I r) {
    const I b = r / (I)p.rows_per_batch;
    const I i = r - b * (I)p.rows_per_batch;
    long j = i;
    if (p.i32) j = p.shared ? p.i32[i] : p.i32[r];
    else if (p.i64) j = p.shared ? p.i64[i] : p.i64[r];
    return (long)b * p.src_bstride + j;
}

// one thread per (row, channel quad) when C % 4 == 0 and pointers allow, else per element.
// The quad path handles FOUR chunks per trip: their row indices are requested together, then their rows, then (when
// accumulating) the old destinations - four independent chains in flight per lane instead of one index -> row -> store
// chain at a time; the lazy scale / shift come as two 16-byte loads (they were eight scalar ones) and the activation is
// max(z, z*e) (e = 1 none, 0 ReLU, slope LeakyReLU) instead of a switch per element.
template <typename I>
__device__ __forceinline__ void copy_rows_quads(const RowsParams& p) {
    const int cpr = p.C / 4;    // chunks per row
    const long total = p.rows * cpr;
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const bool lazy = p.lazy.scale != nullptr;
        const float es = (!lazy || p.lazy.act == RL_ACT_NONE) ? 1.f : (p.lazy.act == RL_ACT_RELU ? 0.f : p.lazy.slope);
        const long stride = (long)gridDim.x * 256;
        for (long e0 = (long)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += 4 * stride) {
            long so[4], dofs[4];
            int cc[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long e = e0 + u * stride;
                ok[u] = e < total;
                const I ee = (I)(ok[u] ? e : e0);
                const I r = ee / (I)cpr;
                cc[u] = (int)(ee - r * (I)cpr) * 4;
                so[u] = src_row<I>(p, r) * p.lds + cc[u];
                dofs[u] = (long)r * p.ldd + cc[u];
            }
            v4f v[4], o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v4f*>(p.src + so[u]);
            if (p.accumulate) {
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = *reinterpret_cast<const v4f*>(p.dst + dofs[u]);
            }
            if (lazy) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const v4f sc = *reinterpret_cast<const v4f*>(p.lazy.scale + cc[u]);
                    const v4f sh = *reinterpret_cast<const v4f*>(p.lazy.shift + cc[u]);
                    const v4f z = v[u] * sc + sh;
                    v[u] = __builtin_elementwise_max(z, z * es);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p.accumulate) v[u] += o[u];
                if (ok[u]) *reinterpret_cast<v4f*>(p.dst + dofs[u]) = v[u];
            }
        }
    }
}

template <int VEC, typename I>
__global__ __launch_bounds__(256) void copy_rows_kernel(const RowsParams p) {
    const int cpr = p.C / VEC;  // chunks per row
    const long total = p.rows * cpr;
    if constexpr (VEC == 4) {
        copy_rows_quads<I>(p);
    } else {
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const I r = (I)e / (I)cpr;
            const int c = (int)((I)e - r * (I)cpr) * VEC;
            const long so = src_row<I>(p, r) * p.lds + c;
            const long dofs = (long)r * p.ldd + c;
            float v = rl_lazy(p.lazy, p.src[so], c);
            if (p.accumulate) v += p.dst[dofs];
            p.dst[dofs] = v;
        }
    }
}

// Two independent row copies in ONE launch (the two halves of a concat: blockIdx.y picks the copy) - a dependent launch
// costs ~4.6 us in a replayed graph whatever it does, and neither half fills the chip for long.
struct RowsPair {
    RowsParams p[2];
};
__global__ __launch_bounds__(256) void copy_rows_pair_kernel(const RowsPair q) {
    const RowsParams& p = q.p[blockIdx.y];
    copy_rows_quads<uint32_t>(p);
}

// dst[(b*dst_bstride + index[r])*ldd + c] += src[r*lds + c]
template <typename I>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const RowsParams p) {
    const long total = p.rows * p.C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const I r = (I)e / (I)p.C;
        const int c = (int)((I)e - r * (I)p.C);
        const long drow = src_row<I>(p, r);  // the index addresses the destination here
        atomicAdd(p.dst + drow * p.ldd + c, p.src[(long)r * p.lds + c]);
    }
}

int fill(RowsParams* p, const rl_rows_desc* d, const char* who) {
    RL_REQUIRE(d && d->src && d->dst && d->rows >= 0 && d->rows_per_batch > 0 && d->C > 0, RL_ERR_ARGS, "%s: bad descriptor", who);
    RL_REQUIRE(!(d->index32 && d->index64), RL_ERR_ARGS, "%s: give index32 or index64, not both", who);
    RL_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), RL_ERR_ARGS, "%s: scale/shift must come together", who);
    p->src = d->src; p->lds = d->lds; p->src_bstride = d->src_bstride; p->dst = d->dst; p->ldd = d->ldd;
    p->rows = d->rows; p->rows_per_batch = d->rows_per_batch; p->C = d->C; p->i32 = d->index32;
    p->i64 = d->index64; p->shared = d->index_shared; p->accumulate = d->accumulate;
    p->lazy.scale = d->scale; p->lazy.shift = d->shift; p->lazy.act = d->act; p->lazy.slope = d->slope;
    return RL_OK;
}

inline bool fits32(long elements) { return elements < (1l << 32); }

int grid_for(long work) {
    long g = (work + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g < 4096 ? g : 4096);
}

// ---------------------------------------------------------------------- attentive pooling
// thread = (point, channel); the K rows of a point are K*C floats apart by C
// K = 16 (the network's neighbourhood size): the sixteen scores and rows of a (point, channel) column are requested at
// once and kept in registers - one pass over S instead of two / three, one exp per element instead of two, sixteen
// loads in flight per lane.  Same arithmetic in the same order as the generic kernels below (bitwise equal results).
template <typename I>
__global__ __launch_bounds__(256) void attpool_fwd16_kernel(const float* __restrict__ X, const float* __restrict__ S,
                                                            long P, int C, float* __restrict__ Pout) {
    const long total = P * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long pt = (long)((I)e / (I)C);
        const int c = (int)(e - pt * C);
        const float* s = S + pt * 16 * C + c;
        const float* x = X + pt * 16 * C + c;
        float sv[16], xv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sv[k] = s[(long)k * C];
#pragma unroll
        for (int k = 0; k < 16; ++k) xv[k] = x[(long)k * C];
        float m = sv[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) m = fmaxf(m, sv[k]);
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float w = expf(sv[k] - m);
            den += w;
            num += w * xv[k];
        }
        Pout[e] = num / den;
    }
}
template <typename I>
__global__ __launch_bounds__(256) void attpool_bwd16_kernel(const float* __restrict__ X, const float* __restrict__ S,
                                                            const float* __restrict__ Pout, const float* __restrict__ dP,
                                                            long P, int C, float* __restrict__ dS, float* __restrict__ dXa) {
    const long total = P * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long pt = (long)((I)e / (I)C);
        const int c = (int)(e - pt * C);
        const long base = pt * 16 * C + c;
        float sv[16], xv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sv[k] = S[base + (long)k * C];
#pragma unroll
        for (int k = 0; k < 16; ++k) xv[k] = X[base + (long)k * C];
        const float g = dP[e], po = Pout[e];
        float m = sv[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) m = fmaxf(m, sv[k]);
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            sv[k] = expf(sv[k] - m);
            den += sv[k];
        }
        const float inv = 1.f / den;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long o = base + (long)k * C;
            const float a = sv[k] * inv;
            dXa[o] = g * a;
            dS[o] = a * g * (xv[k] - po);
        }
    }
}

template <typename I>
__global__ __launch_bounds__(256) void attpool_fwd_kernel(const float* __restrict__ X, const float* __restrict__ S,
                                                          long P, int K, int C, float* __restrict__ Pout) {
    const long total = P * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long pt = (long)((I)e / (I)C);
        const int c = (int)(e - pt * C);
        const float* s = S + pt * K * C + c;
        const float* x = X + pt * K * C + c;
        float m = s[0];
        for (int k = 1; k < K; ++k) m = fmaxf(m, s[(long)k * C]);
        float den = 0.f, num = 0.f;
        for (int k = 0; k < K; ++k) {
            const float w = expf(s[(long)k * C] - m);
            den += w;
            num += w * x[(long)k * C];
        }
        Pout[e] = num / den;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void attpool_bwd_kernel(const float* __restrict__ X, const float* __restrict__ S,
                                                          const float* __restrict__ Pout, const float* __restrict__ dP,
                                                          long P, int K, int C, float* __restrict__ dS,
                                                          float* __restrict__ dXa) {
    const long total = P * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long pt = (long)((I)e / (I)C);
        const int c = (int)(e - pt * C);
        const long base = pt * K * C + c;
        float m = S[base];
        for (int k = 1; k < K; ++k) m = fmaxf(m, S[base + (long)k * C]);
        float den = 0.f;
        for (int k = 0; k < K; ++k) den += expf(S[base + (long)k * C] - m);
        const float inv = 1.f / den, g = dP[e], po = Pout[e];
        for (int k = 0; k < K; ++k) {
            const long o = base + (long)k * C;
            const float a = expf(S[o] - m) * inv;
            dXa[o] = g * a;
            dS[o] = a * g * (X[o] - po);
        }
    }
}

// four channels per lane, two quads per trip (C % 4 == 0, 16-byte aligned operands): the scalar kernel below moved 4 bytes
// per lane and load
__global__ __launch_bounds__(256) void add_act_fwd_vec_kernel(const float* __restrict__ Y1, const float* __restrict__ s1,
                                                              const float* __restrict__ b1, const float* __restrict__ Y2,
                                                              const float* __restrict__ s2, const float* __restrict__ b2,
                                                              long quads, int C, float slope, float* __restrict__ O) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const unsigned cq = (unsigned)C >> 2;
    const long stride = (long)gridDim.x * 256;
    auto one = [&](long q) {
        const int c = (int)((unsigned long)q % cq) * 4;
        const v4f y1 = *reinterpret_cast<const v4f*>(Y1 + q * 4), y2 = *reinterpret_cast<const v4f*>(Y2 + q * 4);
        const v4f a1 = *reinterpret_cast<const v4f*>(s1 + c), c1 = *reinterpret_cast<const v4f*>(b1 + c);
        const v4f a2 = *reinterpret_cast<const v4f*>(s2 + c), c2 = *reinterpret_cast<const v4f*>(b2 + c);
        const v4f z = (y1 * a1 + c1) + (y2 * a2 + c2);
        v4f o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = z[j] > 0.f ? z[j] : z[j] * slope;
        *reinterpret_cast<v4f*>(O + q * 4) = o;
    };
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    for (; q + stride < quads; q += 2 * stride) { one(q); one(q + stride); }
    if (q < quads) one(q);
}

template <typename I>
__global__ __launch_bounds__(256) void add_act_fwd_kernel(const float* __restrict__ Y1, const float* __restrict__ s1,
                                                          const float* __restrict__ b1, const float* __restrict__ Y2,
                                                          const float* __restrict__ s2, const float* __restrict__ b2,
                                                          long total, int C, float slope, float* __restrict__ O) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)((I)e % (I)C);
        const float z = (Y1[e] * s1[c] + b1[c]) + (Y2[e] * s2[c] + b2[c]);
        O[e] = z > 0.f ? z : z * slope;
    }
}

__global__ __launch_bounds__(256) void add_act_bwd_kernel(float* __restrict__ G, const float* __restrict__ O, long total,
                                                          float slope) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256)
        G[e] = O[e] > 0.f ? G[e] : G[e] * slope;
}

// one lane per neighbourhood row: 2 gathers of 12 B from the (L2-resident) coordinates, 48 B written
__global__ __launch_bounds__(256) void rpe_build_kernel(const float* __restrict__ xyz, long xyz_bstride,
                                                        const int32_t* __restrict__ idx, const float* __restrict__ d2,
                                                        unsigned n, unsigned k, long rows, float* __restrict__ out,
                                                        int is_distance) {
    for (long R = (long)blockIdx.x * 256 + threadIdx.x; R < rows; R += (long)gridDim.x * 256) {
        const unsigned pt = (unsigned)R / k;          // rows < 2^31 (checked on the host)
        const unsigned b = pt / n, i = pt - b * n;
        const float* xb = xyz + (long)b * xyz_bstride * 3;
        const int j = idx[R];
        const float xi = xb[(long)i * 3 + 0], yi = xb[(long)i * 3 + 1], zi = xb[(long)i * 3 + 2];
        const float xj = xb[(long)j * 3 + 0], yj = xb[(long)j * 3 + 1], zj = xb[(long)j * 3 + 2];
        float4* o = reinterpret_cast<float4*>(out + R * 12);
        o[0] = make_float4(xi, yi, zi, xj);
        o[1] = make_float4(yj, zj, xi - xj, yi - yj);
        o[2] = make_float4(zi - zj, is_distance ? d2[R] : __fsqrt_rn(d2[R]), 0.f, 0.f);
    }
}

__global__ __launch_bounds__(256) void scale_mask_kernel(float* __restrict__ x, const uint8_t* __restrict__ mask, float scale,
                                                         long total) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256)
        x[e] = mask[e] ? x[e] * scale : 0.f;
}

// ---- Dropout(0.5) of fc_end (modules.py:528) with a counter-based generator -----------------------------------
// Philox4x32-10 (Salmon et al., the generator behind torch's CUDA dropout): the keep decision of element e is a pure
// function of (seed, key, e), so the backward regenerates the mask instead of storing it and a captured graph draws
// fresh masks on every replay (the key lives in device memory and is bumped by rl_dropout_tick).
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
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

__global__ void dropout_tick_kernel(int64_t* counter, int64_t* key_out) {
    const int64_t v = counter[0] + 1;
    counter[0] = v;
    key_out[0] = v;
}

// one thread per 4 consecutive elements; FWD: dst = keep ? lazy(src)*scale : 0;  !FWD: G = keep ? G*scale : 0 (in place)
template <bool FWD>
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ src, RlLazy lazy, int C, float* __restrict__ dst,
                                                      long quads, const int64_t* __restrict__ key, unsigned long long seed,
                                                      unsigned threshold, float scale, unsigned long long first_quad) {
    const unsigned long long k = (unsigned long long)key[0];
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
        const unsigned long long gq = first_quad + (unsigned long long)q;     // index in the WHOLE batch's tensor (shards of one batch)
        const uint4 r = philox4x32_10(make_uint4((unsigned)gq, (unsigned)(gq >> 32), (unsigned)k, (unsigned)(k >> 32)),
                                      make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
        float4 v = *reinterpret_cast<const float4*>(src + q * 4);
        if (FWD && lazy.scale) {
            const int c = (int)(((unsigned)q * 4u) % (unsigned)C);      // elements < 2^32 (host check); C % 4 == 0: a quad stays inside one row
            v.x = rl_lazy(lazy, v.x, c + 0); v.y = rl_lazy(lazy, v.y, c + 1);
            v.z = rl_lazy(lazy, v.z, c + 2); v.w = rl_lazy(lazy, v.w, c + 3);
        }
        v.x = r.x >= threshold ? v.x * scale : 0.f;
        v.y = r.y >= threshold ? v.y * scale : 0.f;
        v.z = r.z >= threshold ? v.z * scale : 0.f;
        v.w = r.w >= threshold ? v.w * scale : 0.f;
        *reinterpret_cast<float4*>(dst + q * 4) = v;
    }
}

// UpSampler (modules.py:343-414) on channel-first features: out[b][f][q] = sum_j w_j feat[b][f][idx[b][q][j]],
// w_j = (1+eps)/(dist_j^power + eps) normalised over j (power 0 -> plain nearest-neighbour copy of j = 0)
__global__ __launch_bounds__(256) void upsample_cf_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ d2, int B, int F, int N1, int N2, int k,
                                                          int power, float* __restrict__ out) {
    const long total = (long)B * N2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / N2;
        const long q = e - b * N2;
        const int32_t* id = idx + e * k;
        const float* fb = feat + b * (long)F * N1;
        float* ob = out + b * (long)F * N2 + q;
        if (power == 0) {
            const int j = id[0];
            for (int f = 0; f < F; ++f) ob[(long)f * N2] = fb[(long)f * N1 + j];
            continue;
        }
        float wsum = 0.f;
        for (int j = 0; j < k; ++j) {
            const float dist = __fsqrt_rn(d2[e * k + j]);
            const float dp = power == 2 ? dist * dist : dist;
            wsum += (1.0f + 1e-7f) / (dp + 1e-7f);
        }
        for (int f = 0; f < F; ++f) {
            float acc = 0.f;
            for (int j = 0; j < k; ++j) {
                const float dist = __fsqrt_rn(d2[e * k + j]);
                const float dp = power == 2 ? dist * dist : dist;
                const float w = ((1.0f + 1e-7f) / (dp + 1e-7f)) / wsum;
                acc += w * fb[(long)f * N1 + id[j]];
            }
            ob[(long)f * N2] = acc;
        }
    }
}

// out[b][c][perm[i]] = in[b][i][c]
__global__ __launch_bounds__(256) void logits_unpermute_kernel(const float* __restrict__ in, const int64_t* __restrict__ perm,
                                                               long perm_bs, int B, int N, int C, float* __restrict__ out) {
    const long total = (long)B * N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / N;
        const long i = e - b * N;
        const long j = perm[b * perm_bs + i];
        for (int c = 0; c < C; ++c) out[(b * C + c) * N + j] = in[e * C + c];
    }
}
__global__ __launch_bounds__(256) void logits_permute_grad_kernel(const float* __restrict__ dout, const int64_t* __restrict__ perm,
                                                                  long perm_bs, int B, int N, int C, float* __restrict__ din) {
    const long total = (long)B * N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / N;
        const long i = e - b * N;
        const long j = perm[b * perm_bs + i];
        for (int c = 0; c < C; ++c) din[e * C + c] = dout[(b * C + c) * N + j];
    }
}

}  // namespace

extern "C" int rl_copy_rows(const rl_rows_desc* d, void* stream) {
    RowsParams p;
    int rc = fill(&p, d, "rl_copy_rows");
    if (rc) return rc;
    if (p.rows == 0) return RL_OK;
    const bool v4 = (p.C % 4 == 0) && (p.lds % 4 == 0) && (p.ldd % 4 == 0) &&
                    (((uintptr_t)p.src & 15) == 0) && (((uintptr_t)p.dst & 15) == 0) &&
                    (!p.lazy.scale || ((((uintptr_t)p.lazy.scale) | ((uintptr_t)p.lazy.shift)) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    const bool small = fits32(p.rows * p.C) && fits32(p.rows_per_batch);
    const long quads4 = (p.rows * (p.C / 4) + 3) / 4;      // the quad path moves four chunks per lane and trip
    if (v4 && small)  hipLaunchKernelGGL((copy_rows_kernel<4, uint32_t>), dim3(grid_for(quads4)), dim3(256), 0, st, p);
    else if (v4)      hipLaunchKernelGGL((copy_rows_kernel<4, int64_t>), dim3(grid_for(quads4)), dim3(256), 0, st, p);
    else if (small)   hipLaunchKernelGGL((copy_rows_kernel<1, uint32_t>), dim3(grid_for(p.rows * p.C)), dim3(256), 0, st, p);
    else              hipLaunchKernelGGL((copy_rows_kernel<1, int64_t>), dim3(grid_for(p.rows * p.C)), dim3(256), 0, st, p);
    rl_note_kernel("copy_rows_kernel");
    RL_LAUNCH_CHECK("rl_copy_rows");
    return RL_OK;
}

extern "C" int rl_copy_rows_pair(const rl_rows_desc* d0, const rl_rows_desc* d1, void* stream) {
    RowsPair q;
    int rc = fill(&q.p[0], d0, "rl_copy_rows_pair");
    if (rc) return rc;
    rc = fill(&q.p[1], d1, "rl_copy_rows_pair");
    if (rc) return rc;
    bool ok = true;
    long quads = 0;
    for (const RowsParams& p : q.p) {
        ok = ok && p.rows > 0 && (p.C % 4 == 0) && (p.lds % 4 == 0) && (p.ldd % 4 == 0) && (((uintptr_t)p.src & 15) == 0) &&
             (((uintptr_t)p.dst & 15) == 0) && (!p.lazy.scale || ((((uintptr_t)p.lazy.scale) | ((uintptr_t)p.lazy.shift)) & 15) == 0) &&
             fits32(p.rows * p.C) && fits32(p.rows_per_batch);
        const long q4 = (p.rows * (p.C / 4) + 3) / 4;
        quads = q4 > quads ? q4 : quads;
    }
    if (!ok) {      // shapes the 16-byte path does not take: two ordinary launches
        rc = rl_copy_rows(d0, stream);
        return rc ? rc : rl_copy_rows(d1, stream);
    }
    hipLaunchKernelGGL(copy_rows_pair_kernel, dim3(grid_for(quads), 2), dim3(256), 0, (hipStream_t)stream, q);
    rl_note_kernel("copy_rows_pair_kernel");
    RL_LAUNCH_CHECK("rl_copy_rows_pair");
    return RL_OK;
}

extern "C" int rl_scatter_add_rows(const rl_rows_desc* d, void* stream) {
    RowsParams p;
    int rc = fill(&p, d, "rl_scatter_add_rows");
    if (rc) return rc;
    RL_REQUIRE(p.i32 || p.i64, RL_ERR_ARGS, "rl_scatter_add_rows: needs an index");
    RL_REQUIRE(rl_float_atomics_allowed(), RL_ERR_UNSUPPORTED,
               "rl_scatter_add_rows: fp32 atomics are order-dependent; the deterministic path is rl_csr_build + rl_segment_sum_rows "
               "(set RL_ALLOW_FLOAT_ATOMICS=1 to use this entry point anyway)");
    RL_REQUIRE(p.lazy.scale == nullptr, RL_ERR_ARGS, "rl_scatter_add_rows: no lazy transform here");
    if (p.rows == 0) return RL_OK;
    if (fits32(p.rows * p.C) && fits32(p.rows_per_batch))
        hipLaunchKernelGGL(scatter_add_rows_kernel<uint32_t>, dim3(grid_for(p.rows * p.C)), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(scatter_add_rows_kernel<int64_t>, dim3(grid_for(p.rows * p.C)), dim3(256), 0, (hipStream_t)stream, p);
    rl_note_kernel("scatter_add_rows_kernel");
    RL_LAUNCH_CHECK("rl_scatter_add_rows");
    return RL_OK;
}

extern "C" int rl_attpool_fwd(const float* X, const float* S, int64_t P, int K, int C, float* Pout, void* stream) {
    RL_REQUIRE(X && S && Pout && P >= 0 && K > 0 && C > 0, RL_ERR_ARGS, "rl_attpool_fwd: bad arguments");
    if (P == 0) return RL_OK;
    if (fits32(P * C)) {
        if (K == 16) hipLaunchKernelGGL(attpool_fwd16_kernel<uint32_t>, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, X, S, (long)P, C, Pout);
        else hipLaunchKernelGGL(attpool_fwd_kernel<uint32_t>, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, X, S, (long)P, K, C, Pout);
    } else {
        hipLaunchKernelGGL(attpool_fwd_kernel<int64_t>, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, X, S, (long)P, K, C, Pout);
    }
    rl_note_kernel("attpool_fwd_kernel");
    RL_LAUNCH_CHECK("rl_attpool_fwd");
    return RL_OK;
}

extern "C" int rl_attpool_bwd(const float* X, const float* S, const float* Pout, const float* dP, int64_t P, int K,
                              int C, float* dS, float* dXa, void* stream) {
    RL_REQUIRE(X && S && Pout && dP && dS && dXa && P >= 0 && K > 0 && C > 0, RL_ERR_ARGS, "rl_attpool_bwd: bad arguments");
    if (P == 0) return RL_OK;
    if (fits32(P * C) && K == 16)
        hipLaunchKernelGGL(attpool_bwd16_kernel<uint32_t>, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, X, S, Pout, dP,
                           (long)P, C, dS, dXa);
    else if (fits32(P * C))
        hipLaunchKernelGGL(attpool_bwd_kernel<uint32_t>, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, X, S, Pout, dP,
                           (long)P, K, C, dS, dXa);
    else
        hipLaunchKernelGGL(attpool_bwd_kernel<int64_t>, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, X, S, Pout, dP,
                           (long)P, K, C, dS, dXa);
    rl_note_kernel("attpool_bwd_kernel");
    RL_LAUNCH_CHECK("rl_attpool_bwd");
    return RL_OK;
}

extern "C" int rl_add_act_fwd(const float* Y1, const float* s1, const float* b1, const float* Y2, const float* s2,
                              const float* b2, int64_t rows, int C, float slope, float* O, void* stream) {
    RL_REQUIRE(Y1 && s1 && b1 && Y2 && s2 && b2 && O && rows >= 0 && C > 0, RL_ERR_ARGS, "rl_add_act_fwd: bad arguments");
    if (rows == 0) return RL_OK;
    const bool vec = C % 4 == 0 && ((((uintptr_t)Y1) | ((uintptr_t)Y2) | ((uintptr_t)O) | ((uintptr_t)s1) | ((uintptr_t)b1) |
                                      ((uintptr_t)s2) | ((uintptr_t)b2)) & 15) == 0;
    if (vec) {
        const long quads = (long)rows * (C / 4);
        hipLaunchKernelGGL(add_act_fwd_vec_kernel, dim3(grid_for((quads + 1) / 2)), dim3(256), 0, (hipStream_t)stream, Y1, s1, b1, Y2,
                           s2, b2, quads, C, slope, O);
    } else if (fits32(rows * C))
        hipLaunchKernelGGL(add_act_fwd_kernel<uint32_t>, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, Y1, s1, b1, Y2,
                           s2, b2, (long)rows * C, C, slope, O);
    else
        hipLaunchKernelGGL(add_act_fwd_kernel<int64_t>, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, Y1, s1, b1, Y2,
                           s2, b2, (long)rows * C, C, slope, O);
    rl_note_kernel("add_act_fwd_kernel");
    RL_LAUNCH_CHECK("rl_add_act_fwd");
    return RL_OK;
}

extern "C" int rl_add_act_bwd(float* G, const float* O, int64_t rows, int C, float slope, void* stream) {
    RL_REQUIRE(G && O && rows >= 0 && C > 0, RL_ERR_ARGS, "rl_add_act_bwd: bad arguments");
    if (rows == 0) return RL_OK;
    hipLaunchKernelGGL(add_act_bwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, G, O,
                       (long)rows * C, slope);
    rl_note_kernel("add_act_bwd_kernel");
    RL_LAUNCH_CHECK("rl_add_act_bwd");
    return RL_OK;
}

static int rpe_build_impl(const float* xyz, int64_t xyz_bstride, const int32_t* nbr_idx, const float* nbr_d2, int B,
                          int n, int k, float* out, int is_distance, void* stream) {
    RL_REQUIRE(xyz && nbr_idx && nbr_d2 && out && B > 0 && n > 0 && k > 0 && xyz_bstride >= n, RL_ERR_ARGS,
               "rl_rpe_build: bad arguments");
    RL_REQUIRE((((uintptr_t)out) & 15) == 0, RL_ERR_ARGS, "rl_rpe_build: out must be 16-byte aligned");
    const long rows = (long)B * n * k;
    RL_REQUIRE(rows < (1l << 31), RL_ERR_ARGS, "rl_rpe_build: too many rows");
    hipLaunchKernelGGL(rpe_build_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, xyz, (long)xyz_bstride,
                       nbr_idx, nbr_d2, (unsigned)n, (unsigned)k, rows, out, is_distance);
    rl_note_kernel("rpe_build_kernel");
    RL_LAUNCH_CHECK("rl_rpe_build");
    return RL_OK;
}

extern "C" int rl_rpe_build(const float* xyz, int64_t xyz_bstride, const int32_t* nbr_idx, const float* nbr_d2, int B,
                            int n, int k, float* out, void* stream) {
    return rpe_build_impl(xyz, xyz_bstride, nbr_idx, nbr_d2, B, n, k, out, 0, stream);
}

extern "C" int rl_rpe_build_dist(const float* xyz, int64_t xyz_bstride, const int32_t* nbr_idx, const float* nbr_dist, int B,
                                 int n, int k, float* out, void* stream) {
    return rpe_build_impl(xyz, xyz_bstride, nbr_idx, nbr_dist, B, n, k, out, 1, stream);
}

extern "C" int rl_scale_mask(float* x, const uint8_t* mask, float scale, int64_t count, void* stream) {
    RL_REQUIRE(x && mask && count >= 0, RL_ERR_ARGS, "rl_scale_mask: bad arguments");
    if (count == 0) return RL_OK;
    hipLaunchKernelGGL(scale_mask_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, mask, scale, (long)count);
    RL_LAUNCH_CHECK("rl_scale_mask");
    return RL_OK;
}

extern "C" int rl_dropout_tick(int64_t* counter, int64_t* key_out, void* stream) {
    RL_REQUIRE(counter && key_out, RL_ERR_ARGS, "rl_dropout_tick: bad arguments");
    hipLaunchKernelGGL(dropout_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, key_out);
    RL_LAUNCH_CHECK("rl_dropout_tick");
    return RL_OK;
}

static int dropout_args(const float* src, float* dst, int64_t rows, int64_t first_row, int C, const int64_t* key, float p,
                        const char* who, unsigned* threshold) {
    RL_REQUIRE(src && dst && key && rows >= 0 && first_row >= 0 && C > 0 && C % 4 == 0, RL_ERR_ARGS,
               "%s: bad arguments (C must be a multiple of 4)", who);
    RL_REQUIRE(p >= 0.f && p < 1.f, RL_ERR_ARGS, "%s: p must be in [0, 1)", who);
    RL_REQUIRE(rows * C < (1l << 32), RL_ERR_ARGS, "%s: too many elements", who);
    RL_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, RL_ERR_ARGS, "%s: tensors must be 16-byte aligned", who);
    const double t = (double)p * 4294967296.0;
    *threshold = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    return RL_OK;
}

extern "C" int rl_dropout_fwd(const float* src, const float* scale, const float* shift, int act, float slope, float* dst,
                              int64_t rows, int64_t first_row, int C, const int64_t* key, uint64_t seed, float p, void* stream) {
    unsigned thr;
    int rc = dropout_args(src, dst, rows, first_row, C, key, p, "rl_dropout_fwd", &thr);
    if (rc) return rc;
    RL_REQUIRE((scale == nullptr) == (shift == nullptr), RL_ERR_ARGS, "rl_dropout_fwd: scale/shift must come together");
    if (rows == 0) return RL_OK;
    RlLazy lz; lz.scale = scale; lz.shift = shift; lz.act = act; lz.slope = slope;
    const long quads = (long)rows * C / 4;
    hipLaunchKernelGGL(dropout_kernel<true>, dim3(grid_for(quads)), dim3(256), 0, (hipStream_t)stream, src, lz, C, dst, quads, key,
                       (unsigned long long)seed, thr, 1.0f / (1.0f - p), (unsigned long long)first_row * (unsigned)C / 4u);
    rl_note_kernel("dropout_kernel");
    RL_LAUNCH_CHECK("rl_dropout_fwd");
    return RL_OK;
}

extern "C" int rl_dropout_bwd(float* G, int64_t rows, int64_t first_row, int C, const int64_t* key, uint64_t seed, float p,
                              void* stream) {
    unsigned thr;
    int rc = dropout_args(G, G, rows, first_row, C, key, p, "rl_dropout_bwd", &thr);
    if (rc) return rc;
    if (rows == 0) return RL_OK;
    RlLazy lz; lz.scale = nullptr; lz.shift = nullptr; lz.act = 0; lz.slope = 0.f;
    const long quads = (long)rows * C / 4;
    hipLaunchKernelGGL(dropout_kernel<false>, dim3(grid_for(quads)), dim3(256), 0, (hipStream_t)stream, G, lz, C, G, quads, key,
                       (unsigned long long)seed, thr, 1.0f / (1.0f - p), (unsigned long long)first_row * (unsigned)C / 4u);
    rl_note_kernel("dropout_kernel");
    RL_LAUNCH_CHECK("rl_dropout_bwd");
    return RL_OK;
}

extern "C" int rl_upsample_cf(const float* feat, const int32_t* idx, const float* d2, int B, int F, int N1, int N2, int k,
                              int power, float* out, void* stream) {
    RL_REQUIRE(feat && idx && out && B > 0 && F > 0 && N1 > 0 && N2 >= 0 && k > 0, RL_ERR_ARGS, "rl_upsample_cf: bad arguments");
    RL_REQUIRE(power >= 0 && power <= 2 && (power == 0 || d2), RL_ERR_ARGS, "rl_upsample_cf: power must be 0, 1 or 2");
    if (N2 == 0) return RL_OK;
    hipLaunchKernelGGL(upsample_cf_kernel, dim3(grid_for((long)B * N2)), dim3(256), 0, (hipStream_t)stream, feat, idx, d2, B,
                       F, N1, N2, k, power, out);
    RL_LAUNCH_CHECK("rl_upsample_cf");
    return RL_OK;
}

extern "C" int rl_logits_unpermute_b(const float* in, const int64_t* perm, int64_t perm_bstride, int B, int N, int C, float* out,
                                     void* stream) {
    RL_REQUIRE(in && perm && out && B > 0 && N > 0 && C > 0 && (perm_bstride == 0 || perm_bstride >= N), RL_ERR_ARGS,
               "rl_logits_unpermute: bad arguments");
    hipLaunchKernelGGL(logits_unpermute_kernel, dim3(grid_for((long)B * N)), dim3(256), 0, (hipStream_t)stream, in, perm,
                       (long)perm_bstride, B, N, C, out);
    RL_LAUNCH_CHECK("rl_logits_unpermute");
    return RL_OK;
}
extern "C" int rl_logits_unpermute(const float* in, const int64_t* perm, int B, int N, int C, float* out, void* stream) {
    return rl_logits_unpermute_b(in, perm, 0, B, N, C, out, stream);
}

extern "C" int rl_logits_permute_grad_b(const float* dout, const int64_t* perm, int64_t perm_bstride, int B, int N, int C,
                                        float* din, void* stream) {
    RL_REQUIRE(dout && perm && din && B > 0 && N > 0 && C > 0 && (perm_bstride == 0 || perm_bstride >= N), RL_ERR_ARGS,
               "rl_logits_permute_grad: bad arguments");
    hipLaunchKernelGGL(logits_permute_grad_kernel, dim3(grid_for((long)B * N)), dim3(256), 0, (hipStream_t)stream, dout,
                       perm, (long)perm_bstride, B, N, C, din);
    RL_LAUNCH_CHECK("rl_logits_permute_grad");
    return RL_OK;
}
extern "C" int rl_logits_permute_grad(const float* dout, const int64_t* perm, int B, int N, int C, float* din,
                                      void* stream) {
    return rl_logits_permute_grad_b(dout, perm, 0, B, N, C, din, stream);
}
