// Per-point linear layers on gfx950: the shared MLPs (1x1 conv / conv-transpose), fc_start and
// the attentive-pooling score Linear of the reference (randlanet/utils/modules.py:82-104, 235,
// 494) as one row-streaming MFMA kernel, plus the matching weight-gradient kernel.
//
// fp32 in / fp32 accumulate on v_mfma_f32_16x16x4_f32: the result is an exact fp32 FMA chain
// over k, so this path meets the 1e-3 logit bound against the reference's CPU forward.
//   A fragment: lane l holds A[row l&15][k l>>4];  B fragment: B[k l>>4][col l&15]
//   C/D       : lane l, reg r -> row (l>>4)*4 + r, col l&15
//
// Forward tile: 128 rows x (16*NT) columns per workgroup (4 wavefronts x 32 rows), K streamed in
// chunks of 16 through LDS.  Workgroups stride over row tiles, so BatchNorm's batch statistics
// (column sums of Y and Y^2) are kept in registers across tiles and leave the workgroup once,
// as one deterministic partial per workgroup (no atomics).  The A operand is read with the
// producer's BatchNorm+activation applied on the fly ("lazy" operand), or synthesised from
// xyz / neighbour indices (relative position encoding, modules.py:173-186) without ever being
// materialised.
#include "rl_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GM_BM = 128;
constexpr int GM_BK = 16;
constexpr int GM_AS = 20;  // LDS row stride of the A tile (floats): 16-B aligned rows, 2-way at worst

struct AOperand {
    const float* A;
    long lda, a_bstride;
    int a_mode;
    RlLazy lazy;
    const float* xyz;
    long xyz_bstride;
    const int32_t* nbr_idx;
    const float* nbr_d2;
    int nbr_k;
    int n;        // rows per cloud (mode 0) / points per cloud (mode 1)
    int K;
    long M;       // total rows
    int contig;   // mode 0: a_bstride == n
    int vec4;     // mode 0: 16-byte loads allowed (K % 4 == 0)
    int vec4p;    // mode 0: 16-byte loads allowed up to K rounded up to 4 (the row padding exists: lda >= that)
};

__device__ __forceinline__ long a_row_offset(const AOperand& a, long R) {
    if (a.contig) return R * a.lda;
    const int b = (int)((unsigned)R / (unsigned)a.n);   // R < 2^31 (fill_a)
    const int i = (int)(R - (long)b * a.n);
    return ((long)b * a.a_bstride + i) * a.lda;
}

// Stage rows [row0, row0+nrows) x columns [k0, k0+kw) of the A operand into LDS (row stride
// `stride`), zero-filling beyond row_limit / K.  kw is a multiple of 16.
__device__ __forceinline__ void stage_a(const AOperand& a, long row0, int nrows, long row_limit,
                                        int k0, int kw, float* As, int stride) {
    const int tid = threadIdx.x;
    if (a.a_mode == 1) {
        // relative position encoding: one lane builds one 10-channel row
        for (int r = tid; r < nrows; r += 256) {
            const long R = row0 + r;
            float v[10];
#pragma unroll
            for (int c = 0; c < 10; ++c) v[c] = 0.f;
            if (R < row_limit) {
                const long p = (unsigned)R / (unsigned)a.nbr_k;
                const int b = (int)((unsigned)p / (unsigned)a.n);
                const int i = (int)(p - (long)b * a.n);
                const int j = a.nbr_idx[R];
                const float* xb = a.xyz + (long)b * a.xyz_bstride * 3;
                const float xi = xb[(long)i * 3 + 0], yi = xb[(long)i * 3 + 1], zi = xb[(long)i * 3 + 2];
                const float xj = xb[(long)j * 3 + 0], yj = xb[(long)j * 3 + 1], zj = xb[(long)j * 3 + 2];
                v[0] = xi; v[1] = yi; v[2] = zi;
                v[3] = xj; v[4] = yj; v[5] = zj;
                v[6] = xi - xj; v[7] = yi - yj; v[8] = zi - zj;
                v[9] = __fsqrt_rn(a.nbr_d2[R]);
            }
            // K == 10 fits one chunk, so k0 == 0 here (checked on the host): static indices only
            float* dst = As + r * stride;
#pragma unroll
            for (int c = 0; c < 10; ++c) dst[c] = v[c];
            for (int c = 10; c < kw; ++c) dst[c] = 0.f;
        }
        return;
    }
    if (a.vec4) {
        const int qpr = kw >> 2;  // float4 per row
        for (int e = tid; e < nrows * qpr; e += 256) {
            const int r = e / qpr, q = e - r * qpr;
            const long R = row0 + r;
            const int kc = k0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (R < row_limit && kc < a.K) {
                v = *reinterpret_cast<const float4*>(a.A + a_row_offset(a, R) + kc);
                if (a.lazy.scale) {
                    v.x = rl_lazy(a.lazy, v.x, kc + 0);
                    v.y = rl_lazy(a.lazy, v.y, kc + 1);
                    v.z = rl_lazy(a.lazy, v.z, kc + 2);
                    v.w = rl_lazy(a.lazy, v.w, kc + 3);
                }
            }
            *reinterpret_cast<float4*>(As + r * stride + q * 4) = v;
        }
        return;
    }
    for (int e = tid; e < nrows * kw; e += 256) {
        const int r = e / kw, c = e - r * kw;
        const long R = row0 + r;
        const int kc = k0 + c;
        float v = 0.f;
        if (R < row_limit && kc < a.K) v = rl_lazy(a.lazy, a.A[a_row_offset(a, R) + kc], kc);
        As[r * stride + c] = v;
    }
}

struct GemmParams {
    AOperand a;
    int N;
    const float* W;
    long w_ks, w_ns;
    const float* bias;
    float* Y;
    long ldy, y_bstride;
    int rows_per_batch;  // rows of Y per cloud
    int y_contig;
    int accumulate;
    double* stats;
    // shifted statistics (round 5): the sums are those of (y - pivot) and (y - pivot)^2, pivot[c] = piv_mean[c] - piv_bias[c]
    // (the layer's running mean, minus the conv bias this GEMM leaves out): a channel whose spread is small against its mean
    // keeps its variance in fp32 partial sums.  NULL = pivot 0.  rl_bn_finalize(pivoted) adds the pivot back.
    const float* piv_mean;
    const float* piv_bias;
    int ksplit;        // > 1: blockIdx.z owns a K range and writes raw partial tiles to kslab
    int kchunk;        // K range per split (multiple of 32)
    float* kslab;      // [ksplit][M][N]
    int stat_slots;    // slots the finalize kernel reads; a smaller grid zero-fills the rest
    int gx, ny;                // pgemm: logical grid (row-tile workgroups, column tiles); see the index mapping there
    const float* addend;       // split-scatter epilogue (pgemm only), see rl_gemm_desc
    float* out2;
    const int32_t* out2_index;
    long out2_bstride;
    int split_col;
    const __bf16* wsplit;      // wgemm: head plane [N][K] (k contiguous), tail plane follows at + N*K
    // BatchNorm-BACKWARD sums as a by-product (sgemm_kernel only, round 6): this product IS the gradient G w.r.t. the activated output of
    // a layer whose raw output Yl (same rows, same layout as Y) and folded BatchNorm are given - the epilogue leaves
    // sum g' and sum g' xhat per column in `stats` (g' = G act'(Yl scale + shift), xhat = (Yl - mean) invstd): rl_bn_bwd_reduce's slots
    const float* bnb_Y;
    const float* bnb_scale;
    const float* bnb_shift;
    const float* bnb_mean;
    const float* bnb_invstd;
    float bnb_neg;             // act'(z) for z <= 0: 1 (none) / 0 (ReLU) / slope
    // rl_gemm_pair (wgemm2_kernel only): a SECOND product over the same A' in the same launch - the column blocks from pair_ny1
    // on belong to it: its own weight planes, columns, dense output, statistics and pivot
    int pair_ny1;              // 0: no second product
    int N2;
    float* Y2;
    double* stats2;
    const float* piv_mean2;
    const float* piv_bias2;
    const __bf16* wsplit2;
};

__device__ __forceinline__ float stat_pivot(const GemmParams& p, int c) {
    if (!p.stats || !p.piv_mean || c >= p.N) return 0.f;
    return p.piv_mean[c] - (p.piv_bias ? p.piv_bias[c] : 0.f);
}

template <int NT>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
    constexpr int BN = 16 * NT;
    constexpr int WS = BN + ((BN % 32 == 0) ? 16 : 0);  // stride % 32 == 16: conflict-free B reads
    __shared__ __attribute__((aligned(16))) float As[GM_BM * GM_AS];
    __shared__ float Ws[GM_BK * WS];
    __shared__ double red[4][2][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int lr = lane & 15, lq = lane >> 4;
    const int col0 = blockIdx.y * BN;
    const int K = p.a.K, N = p.N;
    const long M = p.a.M;
    const long ntiles = (M + GM_BM - 1) / GM_BM;

    float ssum[NT], ssq[NT], piv[NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        ssum[nb] = ssq[nb] = 0.f;
        piv[nb] = stat_pivot(p, col0 + nb * 16 + lr);
    }

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * GM_BM;
        f32x4 acc[2][NT];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) acc[rb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int k0 = 0; k0 < K; k0 += GM_BK) {
            __syncthreads();
            stage_a(p.a, row0, GM_BM, M, k0, GM_BK, As, GM_AS);
            for (int e = tid; e < GM_BK * BN; e += 256) {
                int kk, c;
                if (p.w_ns == 1) { kk = e / BN; c = e - kk * BN; }
                else             { c = e / GM_BK; kk = e - c * GM_BK; }
                float v = 0.f;
                if (k0 + kk < K && col0 + c < N) v = p.W[(long)(k0 + kk) * p.w_ks + (long)(col0 + c) * p.w_ns];
                Ws[kk * WS + c] = v;
            }
            __syncthreads();
            const int ksteps = min(4, (K - k0 + 3) >> 2);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < ksteps) {
                    const int kc = ks * 4 + lq;
                    const float a0 = As[(wave * 32 + lr) * GM_AS + kc];
                    const float a1 = As[(wave * 32 + 16 + lr) * GM_AS + kc];
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        const float bv = Ws[kc * WS + nb * 16 + lr];
                        acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][nb], 0, 0, 0);
                        acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][nb], 0, 0, 0);
                    }
                }
            }
        }
        // epilogue: bias, store, BatchNorm partial statistics
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long R = row0 + wave * 32 + rb * 16 + lq * 4 + r;
                if (R < M) {
                    long yoff;
                    if (p.y_contig) yoff = R * p.ldy;
                    else {
                        const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                        const int i = (int)(R - (long)b * p.rows_per_batch);
                        yoff = ((long)b * p.y_bstride + i) * p.ldy;
                    }
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        const int c = col0 + nb * 16 + lr;
                        if (c < N) {
                            float v = acc[rb][nb][r];
                            if (p.bias) v += p.bias[c];
                            if (p.accumulate) v += p.Y[yoff + c];
                            p.Y[yoff + c] = v;
                            const float dv = v - piv[nb];
                            ssum[nb] += dv;
                            ssq[nb] += dv * dv;
                        }
                    }
                }
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            float s = ssum[nb], q = ssq[nb];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) {
                red[wave][0][nb * 16 + lane] = (double)s;
                red[wave][1][nb * 16 + lane] = (double)q;
            }
        }
        __syncthreads();
        if (tid < BN && col0 + tid < N) {
            const double s = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
            const double q = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
            p.stats[((long)blockIdx.x * 2 + 0) * N + col0 + tid] = s;
            p.stats[((long)blockIdx.x * 2 + 1) * N + col0 + tid] = q;
            for (long slot = blockIdx.x + gridDim.x; slot < p.stat_slots; slot += gridDim.x) {
                p.stats[(slot * 2 + 0) * N + col0 + tid] = 0.0;
                p.stats[(slot * 2 + 1) * N + col0 + tid] = 0.0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ wgrad
constexpr int WG_RB = 32;   // rows per LDS step
constexpr int WG_T = 64;    // tile of dW: 64 (n) x 64 (k) per workgroup
constexpr int WG_S = 80;    // LDS row stride (floats): % 32 == 16 -> conflict-free fragment reads

struct WgradParams {
    AOperand a;
    int N;
    const float* dY;
    long lddy, dy_bstride;
    int rows_per_batch;
    int dy_contig;
    float* slab;
    long rows_per_block;
    int has_bias;
};

__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float dYs[WG_RB * WG_S];
    __shared__ __attribute__((aligned(16))) float As[WG_RB * WG_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int lr = lane & 15, lq = lane >> 4;
    const int N = p.N, K = p.a.K;
    const int n0 = blockIdx.y * WG_T, k0 = blockIdx.z * WG_T;
    const int nvalid = min(WG_T, N - n0), kvalid = min(WG_T, K - k0);
    const int nkb = (kvalid + 15) >> 4;
    const bool wave_active = wave * 16 < nvalid;
    const long r_begin = (long)blockIdx.x * p.rows_per_block;
    const long r_end = min(p.a.M, r_begin + p.rows_per_block);

    f32x4 acc[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) acc[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    for (long r0 = r_begin; r0 < r_end; r0 += WG_RB) {
        __syncthreads();
        for (int e = tid; e < WG_RB * WG_T; e += 256) {
            const int r = e >> 6, c = e & 63;
            const long R = r0 + r;
            float v = 0.f;
            if (R < r_end && c < nvalid) {
                long off;
                if (p.dy_contig) off = R * p.lddy;
                else {
                    const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                    const int i = (int)(R - (long)b * p.rows_per_batch);
                    off = ((long)b * p.dy_bstride + i) * p.lddy;
                }
                v = p.dY[off + n0 + c];
            }
            dYs[r * WG_S + c] = v;
        }
        stage_a(p.a, r0, WG_RB, r_end, k0, nkb * 16, As, WG_S);
        __syncthreads();
        if (p.has_bias && blockIdx.z == 0 && tid < WG_T) {
#pragma unroll 8
            for (int r = 0; r < WG_RB; ++r) bsum += dYs[r * WG_S + tid];
        }
        if (wave_active) {
#pragma unroll
            for (int rs = 0; rs < WG_RB / 4; ++rs) {
                const int rr = rs * 4 + lq;
                const float av = dYs[rr * WG_S + wave * 16 + lr];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb < nkb) {
                        const float bv = As[rr * WG_S + kb * 16 + lr];
                        acc[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[kb], 0, 0, 0);
                    }
                }
            }
        }
    }
    float* out = p.slab + (long)blockIdx.x * ((long)N * K + N);
    if (wave_active) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb < nkb) {
                const int k = k0 + kb * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wave * 16 + lq * 4 + r;
                    if (n < N && k < K) out[(long)n * K + k] = acc[kb][r];
                }
            }
        }
    }
    if (p.has_bias && blockIdx.z == 0 && tid < nvalid) out[(long)N * K + n0 + tid] = bsum;
}

// 64 consecutive output elements (256 contiguous bytes per slab row) x 4 slab lanes per workgroup; fixed
// summation order (lane-strided partial sums in 4 independent chains, then a fixed LDS tree): deterministic.
constexpr int WR_E = 64;
__device__ __forceinline__ void wgrad_reduce_tile(const float* __restrict__ slab, int nsplit, int N, int K,
                                                  float* __restrict__ dW, long w_ks, long w_ns,
                                                  float* __restrict__ dbias, long tile, float (*red)[WR_E + 1]) {
    const long per = (long)N * K + N;
    const int ex = threadIdx.x & (WR_E - 1), sy = threadIdx.x / WR_E;
    const long e = tile * WR_E + ex;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < per) {
        const float* src = slab + e;
        int i = sy;
        for (; i + 12 < nsplit; i += 16) {
            s0 += src[(long)i * per];
            s1 += src[(long)(i + 4) * per];
            s2 += src[(long)(i + 8) * per];
            s3 += src[(long)(i + 12) * per];
        }
        for (; i < nsplit; i += 4) s0 += src[(long)i * per];
    }
    red[sy][ex] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sy == 0 && e < per) {
        const float t = (red[0][ex] + red[1][ex]) + (red[2][ex] + red[3][ex]);
        if (e < (long)N * K) {
            const int n = (int)((unsigned)e / (unsigned)K), k = (int)(e - (long)n * K);
            dW[(long)k * w_ks + (long)n * w_ns] = t;
        } else if (dbias) {
            dbias[e - (long)N * K] = t;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit,
                                                           int N, int K, float* __restrict__ dW,
                                                           long w_ks, long w_ns,
                                                           float* __restrict__ dbias) {
    __shared__ float red[4][WR_E + 1];
    wgrad_reduce_tile(slab, nsplit, N, K, dW, w_ks, w_ns, dbias, blockIdx.x, red);
}

// the same reduction for up to RB_MAX layers in one launch: workgroup -> (item, tile) through the prefix
// table in the kernel arguments
constexpr int RB_MAX = 48;
struct ReduceBatch {
    rl_wgrad_reduce_item item[RB_MAX];
    int first_tile[RB_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch b) {
    __shared__ float red[4][WR_E + 1];
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.first_tile[i + 1]) ++i;
    const rl_wgrad_reduce_item& it = b.item[i];
    wgrad_reduce_tile(it.slab, it.nsplit, it.N, it.K, it.dW, (long)it.w_ks, (long)it.w_ns, it.dbias,
                      (long)blockIdx.x - b.first_tile[i], red);
}

// 128 x 128 tiles of dW as soon as one side reaches 128: a partly empty tile on the transposed-staging kernel is
// 2-3x faster than full 64 x 64 tiles on the older one (40960x256x32: 91 -> 34 us)
inline int wgrad_tile(int N, int K) { return (N >= 128 || K >= 128) ? 128 : WG_T; }

void wgrad_split(long M, int N, int K, int* nsplit, long* rows_per_block) {
    const int T = wgrad_tile(N, K);
    const int ny = rl_cdiv(N, T), nz = rl_cdiv(K, T);
    long want = 1024 / ((long)ny * nz);
    if (want < 1) want = 1;
    // row ranges of >= 256 rows keep the partial slabs small; shorter ones (down to 64 rows) only where
    // the launch would otherwise leave most of the 256 CUs idle
    long maxs = (M + 255) / 256;
    const long fill = 256 / ((long)ny * nz) + 1;
    if (maxs < fill) maxs = fill < (M + 63) / 64 ? fill : (M + 63) / 64;
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    if (T == 128) {
        // the 128x128 kernel holds two workgroups per CU: one resident round (512 workgroups) instead of 1.25,
        // and a quarter less partial-slab traffic
        const long round = 512 / ((long)ny * nz);
        if (round >= 1 && want > round) want = round;
    }
    // the wide layers of a backward pass run as ONE grouped launch (rl_wgrad_batch), so a layer need not fill the chip alone:
    // half the row splits = half the partial-slab traffic (measured: wgrad_reduce_batch 0.102 -> 0.080 ms, step 7.83 -> 7.78 ms;
    // a quarter: no further gain, an eighth: the grouped launch itself slows down).  RL_WGRAD_SHARE overrides (diagnostics).
    static const long share = getenv("RL_WGRAD_SHARE") ? atol(getenv("RL_WGRAD_SHARE")) : 2;
    if (share > 1 && T == 128) { want /= share; if (want < 1) want = 1; }
    long rpb = (M + want - 1) / want;
    rpb = ((rpb + 63) / 64) * 64;   // multiple of both kernels' row chunks (32 / 64)
    *rows_per_block = rpb;
    *nsplit = (int)((M + rpb - 1) / rpb);
    if (*nsplit < 1) *nsplit = 1;
}

int fill_a(AOperand* a, const char* who, const float* A, long lda, long a_bstride, int a_mode,
           int in_act, float in_slope, const float* in_scale, const float* in_shift,
           const float* xyz, long xyz_bstride, const int32_t* nbr_idx, const float* nbr_d2, int nbr_k,
           int B, int n, int K) {
    RL_REQUIRE(B > 0 && n > 0 && K > 0, RL_ERR_ARGS, "%s: bad sizes B=%d n=%d K=%d", who, B, n, K);
    a->A = A; a->lda = lda; a->a_bstride = a_bstride; a->a_mode = a_mode;
    a->lazy.scale = in_scale; a->lazy.shift = in_shift; a->lazy.act = in_act; a->lazy.slope = in_slope;
    a->xyz = xyz; a->xyz_bstride = xyz_bstride; a->nbr_idx = nbr_idx; a->nbr_d2 = nbr_d2; a->nbr_k = nbr_k;
    a->n = n; a->K = K;
    if (a_mode == 1) {
        RL_REQUIRE(K == 10, RL_ERR_ARGS, "%s: relative position encoding source needs K == 10 (got %d)", who, K);
        RL_REQUIRE(xyz && nbr_idx && nbr_d2 && nbr_k > 0 && xyz_bstride >= n, RL_ERR_ARGS, "%s: incomplete RPE source", who);
        a->M = (long)B * n * nbr_k;
        a->contig = 1; a->vec4 = 0; a->vec4p = 0;
    } else {
        RL_REQUIRE(a_mode == 0, RL_ERR_ARGS, "%s: unknown a_mode %d", who, a_mode);
        RL_REQUIRE(A && lda >= K && a_bstride >= n, RL_ERR_ARGS, "%s: bad A operand", who);
        RL_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), RL_ERR_ARGS, "%s: in_scale/in_shift must come together", who);
        a->M = (long)B * n;
        a->contig = (a_bstride == n);
        a->vec4 = (lda % 4 == 0) && (K % 4 == 0) && (((uintptr_t)A & 15) == 0);
        a->vec4p = (lda % 4 == 0) && ((K + 3) / 4 * 4 <= lda) && (((uintptr_t)A & 15) == 0);
    }
    RL_REQUIRE(a->M < (1l << 31), RL_ERR_ARGS, "%s: too many rows", who);
    return RL_OK;
}


// ===========================================================================================
// Streaming variants for the narrow layers (K <= 64 and N <= 64: levels 0/1 of the encoder,
// the decoder tail and fc_end), where a layer is pure HBM traffic: no LDS, no barriers.
//
// Forward: a wavefront owns 16-row blocks.  Lane (i = l&15, j = l>>4) loads 16 B of row i at
// column 16c + 4j (one dwordx4 per 16-column chunk c; the wavefront reads 16 rows x 64 B, fully
// coalesced when rows are contiguous).  MFMA step (c, s) uses component s of that load, i.e.
// k = 16c + 4j + s: any k order is valid as long as A and B fragments agree, and the weight
// fragments - which live in registers for the whole kernel - are loaded in exactly that order.
//
// Weight gradient: lane (c = l&15, r = l>>4) loads dY[row0 + r][16nb + c] and A'[row0 + r][16kb + c]
// (4 rows x 64 B per load instruction); these are the MFMA operands as they stand, with the row
// as the reduction index.  The four wavefronts of a workgroup take interleaved row groups and
// are combined through LDS in a fixed order; every workgroup leaves one partial slab.
// ===========================================================================================

template <int KC, int NT, bool BNB = false>
__global__ __launch_bounds__(256) void sgemm_kernel(const GemmParams p) {
    __shared__ double red[4][2][16 * NT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int li = lane & 15, lj = lane >> 4;
    const int K = p.a.K, N = p.N;
    const long M = p.a.M;

    float wf[KC][4][NT];
    float sc[KC][4], sh[KC][4];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * c + 4 * lj + s;
            sc[c][s] = (p.a.lazy.scale && k < K) ? p.a.lazy.scale[k] : 1.f;
            sh[c][s] = (p.a.lazy.scale && k < K) ? p.a.lazy.shift[k] : 0.f;
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                const int n = nb * 16 + li;
                wf[c][s][nb] = (k < K && n < N) ? p.W[(long)k * p.w_ks + (long)n * p.w_ns] : 0.f;
            }
        }
    // (Round 4 tried the product TRANSPOSED - weights as the first MFMA operand, a lane then owns four consecutive columns of its row
    // and the block leaves as 16-byte stores.  Y came out bit-identical and the step time unchanged, but the per-lane grouping of
    // the statistics changed with it, and two tests that sit on nearly degenerate BatchNorm channels - the 1029-point ragged
    // configuration, the two-rank equivalence step - moved from 1e-4 to 0.5 - 3 % gradient differences: mean / variance come out of
    // sum and sum of squares by subtraction.  Reverted: nothing was gained.)
    float bias[NT], ssum[NT], ssq[NT], piv[NT];
    float bsc[BNB ? NT : 1], bsh[BNB ? NT : 1], bmu[BNB ? NT : 1], bis[BNB ? NT : 1];      // BNB: the layer's folded BatchNorm per column
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        const int n = nb * 16 + li;
        bias[nb] = (p.bias && n < N) ? p.bias[n] : 0.f;
        ssum[nb] = ssq[nb] = 0.f;
        piv[nb] = BNB ? 0.f : stat_pivot(p, n);
        if constexpr (BNB) {
            bsc[nb] = n < N ? p.bnb_scale[n] : 1.f; bsh[nb] = n < N ? p.bnb_shift[n] : 0.f;
            bmu[nb] = n < N ? p.bnb_mean[n] : 0.f;  bis[nb] = n < N ? p.bnb_invstd[n] : 0.f;
        }
    }
    const bool lazy = p.a.lazy.scale != nullptr;
    const float es = (!lazy || p.a.lazy.act == RL_ACT_NONE) ? 1.f : (p.a.lazy.act == RL_ACT_RELU ? 0.f : p.a.lazy.slope);
    const bool full = N == 16 * NT;

    const long nblk = (M + 15) >> 4;
    const long bstep = (long)gridDim.x * 4;
    // the rows of the wavefront's next 16-row block are requested before the MFMAs of the current one
    auto fetch = [&](long blk, float4 (&a)[KC]) {
        const long row = blk * 16 + li;
        const bool rvalid = row < M;
        const long aoff = rvalid ? a_row_offset(p.a, row) : 0;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rvalid && 16 * c + 4 * lj < K) a[c] = *reinterpret_cast<const float4*>(p.a.A + aoff + 16 * c + 4 * lj);
        }
    };
    float4 a[KC], an[KC];
    long blk = (long)blockIdx.x * 4 + wave;
    if (blk < nblk) fetch(blk, a);
    for (; blk < nblk; blk += bstep) {
        const bool rvalid = blk * 16 + li < M;
        if (blk + bstep < nblk) fetch(blk + bstep, an);
        f32x4 acc[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            float av[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float v = av[s];
                if (lazy) {      // lazy BatchNorm + activation without a switch: act(z) = max(z, z*e), e = 1 (none) / 0 (ReLU) / slope
                    const float z = v * sc[c][s] + sh[c][s];
                    v = fmaxf(z, z * es);
                }
                if (!rvalid || 16 * c + 4 * lj + s >= K) v = 0.f;
#pragma unroll
                for (int nb = 0; nb < NT; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, wf[c][s][nb], acc[nb], 0, 0, 0);
            }
        }
        // the next block's rows (requested above) are waited for HERE, before this block's stores: vmcnt is in-order, so the
        // wait the compiler would place at the top of the next iteration would also sit out these stores' round trip
        asm volatile("" ::: "memory");          // (pins the loads above / stores below at the IR level; see pool.hip loads_landed)
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long R = blk * 16 + lj * 4 + r;
            if (R < M) {
                long yoff;
                if (p.y_contig) yoff = R * p.ldy;
                else {
                    const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                    const int i = (int)(R - (long)b * p.rows_per_batch);
                    yoff = ((long)b * p.y_bstride + i) * p.ldy;
                }
                // (the options are tested once per row and a row's loads are issued together: see tile_rows_epilogue)
                float* y = p.Y + yoff + li;
                float v[NT];
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) v[nb] = acc[nb][r] + bias[nb];
                if (p.accumulate) {
                    float o[NT];
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) o[nb] = (full || nb * 16 + li < N) ? y[nb * 16] : 0.f;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) v[nb] += o[nb];
                }
                if constexpr (BNB) {
                    // this row of the layer's raw output (same offset as the gradient row): the terms rl_bn_bwd_reduce would sum
                    const float* yl = p.bnb_Y + yoff + li;
                    float t[NT];
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) t[nb] = (full || nb * 16 + li < N) ? yl[nb * 16] : 0.f;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        if (full || nb * 16 + li < N) {
                            y[nb * 16] = v[nb];
                            const float z = t[nb] * bsc[nb] + bsh[nb];
                            const float gp = v[nb] * (z > 0.f ? 1.f : p.bnb_neg);
                            ssum[nb] += gp;
                            ssq[nb] += gp * ((t[nb] - bmu[nb]) * bis[nb]);
                        }
                    }
                } else {
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) {
                    if (full || nb * 16 + li < N) {
                        y[nb * 16] = v[nb];
                        const float dv = v[nb] - piv[nb];
                        ssum[nb] += dv;
                        ssq[nb] += dv * dv;
                    }
                }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) a[c] = an[c];
    }
    if (p.stats) {
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            float s = ssum[nb], q = ssq[nb];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) {
                red[wave][0][nb * 16 + lane] = (double)s;
                red[wave][1][nb * 16 + lane] = (double)q;
            }
        }
        __syncthreads();
        if (tid < 16 * NT && tid < N) {
            p.stats[((long)blockIdx.x * 2 + 0) * N + tid] = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
            p.stats[((long)blockIdx.x * 2 + 1) * N + tid] = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
            for (long slot = blockIdx.x + gridDim.x; slot < p.stat_slots; slot += gridDim.x) {
                p.stats[(slot * 2 + 0) * N + tid] = 0.0;
                p.stats[(slot * 2 + 1) * N + tid] = 0.0;
            }
        }
    }
}

constexpr int SW_U = 4;  // row groups (of 4 rows) in flight per wavefront

// (bx: the workgroup's slab / row block; red: KT * NT * 4 * 64 + NT * 64 floats of LDS)
template <int KT, int NT>
__device__ __forceinline__ void swgrad_body(const WgradParams& p, const int bx, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int lc = lane & 15, lr = lane >> 4;
    const int N = p.N, K = p.a.K;
    const AOperand& a = p.a;
    const long r_begin = (long)bx * p.rows_per_block;
    const long r_end = min(a.M, r_begin + p.rows_per_block);

    float sc[KT], sh[KT];
#pragma unroll
    for (int kb = 0; kb < KT; ++kb) {
        const int k = kb * 16 + lc;
        sc[kb] = (a.lazy.scale && k < K) ? a.lazy.scale[k] : 1.f;
        sh[kb] = (a.lazy.scale && k < K) ? a.lazy.shift[k] : 0.f;
    }
    const bool lazy = a.lazy.scale != nullptr;
    const float es = (!lazy || a.lazy.act == RL_ACT_NONE) ? 1.f : (a.lazy.act == RL_ACT_RELU ? 0.f : a.lazy.slope);
    f32x4 acc[NT][KT];
    float bsum[NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        bsum[nb] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KT; ++kb) acc[nb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // one wavefront-iteration = 4*SW_U rows; the loads of iteration i+1 are issued before the MFMAs of
    // iteration i (register double buffer), so each wavefront keeps two iterations of HBM requests in flight
    auto fetch = [&](long base, float (&dy)[SW_U][NT], float (&av)[SW_U][KT]) {
#pragma unroll
        for (int u = 0; u < SW_U; ++u) {
            const long R = base + u * 4 + lr;
            const bool valid = R < r_end;
            long doff = 0;
            if (valid) {
                if (p.dy_contig) doff = R * p.lddy;
                else {
                    const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                    const int i = (int)(R - (long)b * p.rows_per_batch);
                    doff = ((long)b * p.dy_bstride + i) * p.lddy;
                }
            }
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                const int n = nb * 16 + lc;
                dy[u][nb] = (valid && n < N) ? p.dY[doff + n] : 0.f;
            }
            if (a.a_mode == 1) {
                // relative position encoding, channel lc of row R (modules.py:173-186)
                float v = 0.f;
                if (valid && lc < 10) {
                    if (lc == 9) v = __fsqrt_rn(a.nbr_d2[R]);
                    else {
                        const long pt = (unsigned)R / (unsigned)a.nbr_k;
                        const int b = (int)((unsigned)pt / (unsigned)a.n);
                        const int i = (int)(pt - (long)b * a.n);
                        const int ax = lc % 3;
                        const float* xb = a.xyz + (long)b * a.xyz_bstride * 3;
                        const float xi = xb[(long)i * 3 + ax];
                        if (lc < 3) v = xi;
                        else {
                            const float xj = xb[(long)a.nbr_idx[R] * 3 + ax];
                            v = lc < 6 ? xj : xi - xj;
                        }
                    }
                }
                av[u][0] = v;
#pragma unroll
                for (int kb = 1; kb < KT; ++kb) av[u][kb] = 0.f;
            } else {
                const long aoff = valid ? a_row_offset(a, R) : 0;
#pragma unroll
                for (int kb = 0; kb < KT; ++kb) {
                    const int k = kb * 16 + lc;
                    av[u][kb] = (valid && k < K) ? a.A[aoff + k] : 0.f;
                }
            }
        }
    };
    auto consume = [&](long base, float (&dy)[SW_U][NT], float (&av)[SW_U][KT]) {
        if (a.a_mode != 1 && lazy) {
#pragma unroll
            for (int u = 0; u < SW_U; ++u) {
                const bool valid = base + u * 4 + lr < r_end;
#pragma unroll
                for (int kb = 0; kb < KT; ++kb)
                    if (valid && kb * 16 + lc < K) {
                        const float z = av[u][kb] * sc[kb] + sh[kb];
                        av[u][kb] = fmaxf(z, z * es);       // act(z) = max(z, z*e): no switch per element
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < SW_U; ++u)
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                bsum[nb] += dy[u][nb];
#pragma unroll
                for (int kb = 0; kb < KT; ++kb)
                    acc[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[u][nb], av[u][kb], acc[nb][kb], 0, 0, 0);
            }
    };
    {
        float dy0[SW_U][NT], av0[SW_U][KT], dy1[SW_U][NT], av1[SW_U][KT];
        long base = r_begin + (long)wave * 4 * SW_U;
        if (base < r_end) fetch(base, dy0, av0);
        while (base < r_end) {
            const long nb1 = base + 16 * SW_U;
            if (nb1 < r_end) fetch(nb1, dy1, av1);
            consume(base, dy0, av0);
            if (nb1 >= r_end) break;
            const long nb2 = nb1 + 16 * SW_U;
            if (nb2 < r_end) fetch(nb2, dy0, av0);
            consume(nb1, dy1, av1);
            base = nb2;
        }
    }
    // bias partials: lanes sharing a column (same lc) combine first
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        bsum[nb] += __shfl_xor(bsum[nb], 16, 64);
        bsum[nb] += __shfl_xor(bsum[nb], 32, 64);
    }
    // wavefronts 1..3 hand their tiles to wavefront 0 through LDS, in order
    float* rb = red + KT * NT * 4 * 64;
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
#pragma unroll
                for (int kb = 0; kb < KT; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((nb * KT + kb) * 4 + r) * 64 + lane] = acc[nb][kb][r];
                rb[nb * 64 + lane] = bsum[nb];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
#pragma unroll
                for (int kb = 0; kb < KT; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[nb][kb][r] += red[((nb * KT + kb) * 4 + r) * 64 + lane];
                bsum[nb] += rb[nb * 64 + lane];
            }
        }
    }
    if (wave == 0) {
        float* out = p.slab + (long)bx * ((long)N * K + N);
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
#pragma unroll
            for (int kb = 0; kb < KT; ++kb) {
                const int k = kb * 16 + lc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb * 16 + lr * 4 + r;
                    if (n < N && k < K) out[(long)n * K + k] = acc[nb][kb][r];
                }
            }
            const int n = nb * 16 + lc;
            if (p.has_bias && lr == 0 && n < N) out[(long)N * K + n] = bsum[nb];
        }
    }
}

template <int KT, int NT>
__global__ __launch_bounds__(256) void swgrad_kernel(const WgradParams p) {
    __shared__ float red[KT * NT * 4 * 64 + NT * 64];
    swgrad_body<KT, NT>(p, (int)blockIdx.x, red);
}

// slab split for the streaming wgrad: one slab per workgroup
void swgrad_split(long M, int* nsplit, long* rows_per_block) {
    static const long rows_wg = getenv("RL_SWGRAD_ROWS") ? atol(getenv("RL_SWGRAD_ROWS")) : 256;
    long want = (M + rows_wg - 1) / rows_wg;        // 256 rows per workgroup (512: 387 -> 367 us per step over all launches; 1024+: 480)
    const long fill = (M + 127) / 128 < 256 ? (M + 127) / 128 : 256;   // one workgroup per CU where the rows allow
    if (want < fill) want = fill;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    long rpb = (M + want - 1) / want;
    rpb = ((rpb + 63) / 64) * 64;
    *rows_per_block = rpb;
    *nsplit = (int)((M + rpb - 1) / rpb);
    if (*nsplit < 1) *nsplit = 1;
}

// (K <= 16 also with up to 128 output columns: mlp_rpe1 of the 128-wide level, 10 -> 128)
inline bool stream_wgrad_ok(int N, int K) { return (N <= 64 && K <= 64) || (K <= 16 && N <= 128); }

template <int KC>
void launch_sgemm(int N, int gx, hipStream_t st, const GemmParams& p) {
    if (p.bnb_Y) {       // with the BatchNorm-backward sums of the layer this product is the gradient of (N <= 64)
        if (N <= 16)      hipLaunchKernelGGL((sgemm_kernel<KC, 1, true>), dim3(gx), dim3(256), 0, st, p);
        else if (N <= 32) hipLaunchKernelGGL((sgemm_kernel<KC, 2, true>), dim3(gx), dim3(256), 0, st, p);
        else              hipLaunchKernelGGL((sgemm_kernel<KC, 4, true>), dim3(gx), dim3(256), 0, st, p);
        return;
    }
    if (N <= 16)      hipLaunchKernelGGL((sgemm_kernel<KC, 1>), dim3(gx), dim3(256), 0, st, p);
    else if (N <= 32) hipLaunchKernelGGL((sgemm_kernel<KC, 2>), dim3(gx), dim3(256), 0, st, p);
    else if (N <= 64) hipLaunchKernelGGL((sgemm_kernel<KC, 4>), dim3(gx), dim3(256), 0, st, p);
    else if constexpr (KC == 1) hipLaunchKernelGGL((sgemm_kernel<1, 8>), dim3(gx), dim3(256), 0, st, p);   // K <= 16 only
}
template <int KT>
void launch_swgrad(int N, dim3 grid, hipStream_t st, const WgradParams& p) {
    if (N <= 16)      hipLaunchKernelGGL((swgrad_kernel<KT, 1>), grid, dim3(256), 0, st, p);
    else if (N <= 32) hipLaunchKernelGGL((swgrad_kernel<KT, 2>), grid, dim3(256), 0, st, p);
    else if (N <= 64) hipLaunchKernelGGL((swgrad_kernel<KT, 4>), grid, dim3(256), 0, st, p);
    else if constexpr (KT == 1) hipLaunchKernelGGL((swgrad_kernel<1, 8>), grid, dim3(256), 0, st, p);       // K <= 16 only
}

// ===========================================================================================
// Pipelined LDS kernel for the wide layers (K > 64 or N > 64): 128 x (16*NT) tile, K in chunks
// of 32, the next chunk's global loads (16 B per lane, A and W alike) are issued into registers
// before the MFMA loop of the current chunk and written to LDS after it, so HBM/L2 latency hides
// under 128 MFMAs per wavefront instead of stalling every 16 columns of K.
// LDS strides: A rows 34 floats (fragment reads of 16 rows x 2 k hit 32 distinct banks),
// W rows BN+16 (stride % 32 == 16: the two k rows of a half-wave read disjoint bank halves).
// ===========================================================================================
constexpr int PG_BK = 32;
constexpr int PG_AS = 36;   // LDS row stride (floats): 16-byte aligned rows, 16 rows x b128 hit 64 distinct banks

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// Epilogue of the LDS-tiled GEMMs (pgemm_kernel / wgemm_kernel / wgemm2_kernel) for one 16-row block of a wavefront: rows
// Rb .. Rb + 15, acc[nb][r] = element (row lq * 4 + r, column col0 + nb * 16 + lr).  The run-time options (bias, accumulate,
// addend, split store) are uniform per launch and a block lies on one side of split_col unless a tile straddles it: the
// common cases are a straight-line path (the options tested once per row, all loads of a row issued together); written
// as one loop with the tests inside, the unrolled epilogue was ~200 scalar branches per tile and cost as much as the
// tile's K loop (tools/wgemm_ab.py).
template <int NT, bool STATS>
__device__ __forceinline__ void tile_rows_epilogue(const GemmParams& p, const f32x4 (&acc)[NT], long Rb, int col0, int lr, int lq,
                                                   float (&ssum)[NT], float (&ssq)[NT]) {
    constexpr int BN = 16 * NT;
    const int N = p.N;
    const long M = p.a.M;
    // where this block's columns go: all to Y, all to the dense out2, or (a tile that straddles split_col, indexed out2) mixed
    const bool to_y = !p.out2 || col0 + BN <= p.split_col;
    const bool to_out2 = p.out2 && !p.out2_index && col0 >= p.split_col;
    if (to_y || to_out2) {
        const bool full = col0 + BN <= N;
        const bool has_bias = p.bias != nullptr;
        const bool accumulate = to_y && p.accumulate;
        float bv[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            const int c = col0 + nb * 16 + lr;
            bv[nb] = (has_bias && c < N) ? p.bias[c] : 0.f;
        }
        float pv[STATS ? NT : 1];             // shifted statistics: sums of (y - pivot), see GemmParams
        if constexpr (STATS) {
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) pv[nb] = stat_pivot(p, col0 + nb * 16 + lr);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long R = Rb + lq * 4 + r;
            if (R < M) {
                float* y;
                if (to_y) {
                    long yoff;
                    if (p.y_contig) yoff = R * p.ldy;
                    else {
                        const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                        const int i = (int)(R - (long)b * p.rows_per_batch);
                        yoff = ((long)b * p.y_bstride + i) * p.ldy;
                    }
                    y = p.Y + yoff + col0 + lr;
                } else {
                    y = p.out2 + R * (N - p.split_col) - p.split_col + col0 + lr;
                }
                float v[NT];
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) v[nb] = acc[nb][r];
                if (has_bias) {
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) v[nb] += bv[nb];
                }
                if (p.addend) {
                    const float* ad = p.addend + R * N + col0 + lr;
                    float o[NT];
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) o[nb] = (full || col0 + nb * 16 + lr < N) ? ad[nb * 16] : 0.f;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) v[nb] += o[nb];
                }
                if (accumulate) {
                    float o[NT];
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) o[nb] = (full || col0 + nb * 16 + lr < N) ? y[nb * 16] : 0.f;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) v[nb] += o[nb];
                }
                if (full) {
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        y[nb * 16] = v[nb];
                        if constexpr (STATS) {
                            const float dv = v[nb] - pv[nb];
                            ssum[nb] += dv;
                            ssq[nb] += dv * dv;
                        }
                    }
                } else {
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        if (col0 + nb * 16 + lr < N) {
                            y[nb * 16] = v[nb];
                            if constexpr (STATS) {
                                const float dv = v[nb] - pv[nb];
                                ssum[nb] += dv;
                                ssq[nb] += dv * dv;
                            }
                        }
                    }
                }
            }
        }
        return;
    }
    // a tile that straddles split_col, or the indexed (atomic) out2: element by element
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long R = Rb + lq * 4 + r;
        if (R < M) {
            long yoff;
            if (p.y_contig) yoff = R * p.ldy;
            else {
                const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                const int i = (int)(R - (long)b * p.rows_per_batch);
                yoff = ((long)b * p.y_bstride + i) * p.ldy;
            }
            long o2 = 0;
            if (p.out2) {
                if (p.out2_index) {
                    const int b2 = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                    o2 = ((long)b2 * p.out2_bstride + p.out2_index[R]) * (N - p.split_col) - p.split_col;
                } else {
                    o2 = R * (N - p.split_col) - p.split_col;      // dense: one row per source row
                }
            }
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                const int c = col0 + nb * 16 + lr;
                if (c < N) {
                    float v = acc[nb][r];
                    if (p.bias) v += p.bias[c];
                    if (p.addend) v += p.addend[R * N + c];
                    if (p.out2 && c >= p.split_col) {
                        if (p.out2_index) atomicAdd(p.out2 + o2 + c, v);
                        else p.out2[o2 + c] = v;
                    } else {
                        if (p.accumulate) v += p.Y[yoff + c];
                        p.Y[yoff + c] = v;
                        if constexpr (STATS) {
                            const float dv = v - stat_pivot(p, c);
                            ssum[nb] += dv;
                            ssq[nb] += dv * dv;
                        }
                    }
                }
            }
        }
    }
}

// fp32 value -> bf16 head + bf16 tail (v = hi + lo up to 2^-17 relative)
__device__ __forceinline__ void split_bf16(const float4 v, bf16x4& hi, bf16x4& lo) {
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// TERMS = 0: fp32 MFMA (v_mfma_f32_16x16x4_f32), the exact-product reference mode.
// TERMS = 3: every fp32 operand is split into a bf16 head and tail on its way into LDS and the product is
//            a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): products carry
//            ~2^-16 relative error instead of 2^-24, three MFMAs of 16 cycles replace eight of 32.
// TERMS = 1: heads only (plain bf16 operands, fp32 accumulate) - the throughput mode.
template <int NT, int TERMS>
__global__ __launch_bounds__(256, 2) void pgemm_kernel(const GemmParams p) {
    constexpr int BN = 16 * NT;
    constexpr int BS = 40;   // bf16 LDS row stride (80 B): 16 rows x ds_read_b128 fall on 16 distinct bank quads
    constexpr int NSPL = TERMS == 3 ? 2 : 1;
    constexpr int WQ = (BN * 8 + 255) / 256;              // float4 of W per lane and chunk
    // both operands sit in LDS with k contiguous ([row][k] and [n][k]); MFMA step s of a chunk takes
    // k = 8*(lane>>4) + s from either, so a lane's eight values per operand tile are two ds_read_b128
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[TERMS == 0 ? GM_BM * PG_AS * 4 : GM_BM * BS * 2 * NSPL];
    __shared__ __attribute__((aligned(16))) unsigned char lds_w[TERMS == 0 ? BN * PG_AS * 4 : BN * BS * 2 * NSPL];
    __shared__ double red[4][2][BN];
    float* As = reinterpret_cast<float*>(lds_a);
    float* Wt = reinterpret_cast<float*>(lds_w);
    __bf16* Ah = reinterpret_cast<__bf16*>(lds_a);
    __bf16* Al = Ah + GM_BM * BS;       // only touched when TERMS == 3
    __bf16* Wh = reinterpret_cast<__bf16*>(lds_w);
    __bf16* Wl = Wh + BN * BS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    // Workgroup -> (row-tile slot bx, column tile by).  With several column tiles the tiles of one row tile share the
    // A operand: they get consecutive slots on the SAME XCD (workgroup ids go round-robin over the 8 XCDs, each with
    // its own L2), so the second one finds A in that L2 instead of fetching it from HBM again.
    int bx = blockIdx.x, by = 0;
    if (p.ny > 1) {
        const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
        by = s % p.ny;
        bx = (s / p.ny) * 8 + xcd;
        if (bx >= p.gx) return;      // padding of the last group of eight
    }
    const int col0 = by * BN;
    const int K = p.a.K, N = p.N;
    const long M = p.a.M;
    const long ntiles = (M + GM_BM - 1) / GM_BM;
    const bool lazy = p.a.lazy.scale != nullptr;
    const bool relu = p.a.lazy.act == RL_ACT_RELU;
    const float nslope = p.a.lazy.act == RL_ACT_NONE ? 1.f : p.a.lazy.slope;
    const bool w_ncontig = (p.w_ns == 1);
    const int aq = tid & 7;            // this lane's k-quad inside a chunk (same for its 4 rows)

    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) ssum[nb] = ssq[nb] = 0.f;

    for (long tile = bx; tile < ntiles; tile += p.gx) {
        const long row0 = tile * GM_BM;
        long aoff[4];
        bool aval[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long R = row0 + (tid >> 3) + 32 * i;
            aval[i] = R < M;
            aoff[i] = aval[i] ? a_row_offset(p.a, R) : 0;
        }
        f32x4 acc[2][NT];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) acc[rb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        float4 ra[4], rw[WQ];
        auto fetch = [&](int k0) {
            const int ka = k0 + aq * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (aval[i] && ka < K) ra[i] = *reinterpret_cast<const float4*>(p.a.A + aoff[i] + ka);
            }
            if (w_ncontig) {
                // W[k][n], n contiguous: unit u = 16 columns x 16 k; lane (q = l&3, kl = l>>2) loads 4 columns of
                // one k (16 B, four lanes cover 64 contiguous bytes) and writes them transposed (see commit)
#pragma unroll
                for (int i = 0; i < WQ; ++i) {
                    const int u = i * 4 + wave;
                    const int n4 = (u % NT) * 16 + (lane & 3) * 4, kk = k0 + (u / NT) * 16 + (lane >> 2);
                    rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (u < 2 * NT && col0 + n4 < N && kk < K)
                        rw[i] = *reinterpret_cast<const float4*>(p.W + (long)kk * p.w_ks + col0 + n4);
                }
            } else {
#pragma unroll
                for (int i = 0; i < WQ; ++i) {
                    const int e = tid + i * 256;
                    const int n = e >> 3, q = e & 7;
                    rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < BN * 8 && col0 + n < N && k0 + q * 4 < K)
                        rw[i] = *reinterpret_cast<const float4*>(p.W + (long)(col0 + n) * p.w_ns + k0 + q * 4);
                }
            }
        };
        auto actf = [&](float z) {
            const float neg = relu ? 0.f : z * nslope;
            return z > 0.f ? z : neg;
        };
        auto commit = [&](int k0) {
            const int ka = k0 + aq * 4;
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lazy && ka < K) {
                sc = *reinterpret_cast<const float4*>(p.a.lazy.scale + ka);
                sh = *reinterpret_cast<const float4*>(p.a.lazy.shift + ka);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 v = ra[i];
                if (lazy && aval[i] && ka < K) {
                    v.x = actf(v.x * sc.x + sh.x);
                    v.y = actf(v.y * sc.y + sh.y);
                    v.z = actf(v.z * sc.z + sh.z);
                    v.w = actf(v.w * sc.w + sh.w);
                }
                if constexpr (TERMS == 0) {
                    *reinterpret_cast<float4*>(As + ((tid >> 3) + 32 * i) * PG_AS + aq * 4) = v;
                } else {
                    bf16x4 hi, lo;
                    split_bf16(v, hi, lo);
                    *reinterpret_cast<bf16x4*>(Ah + ((tid >> 3) + 32 * i) * BS + aq * 4) = hi;
                    if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(Al + ((tid >> 3) + 32 * i) * BS + aq * 4) = lo;
                }
            }
            if (w_ncontig) {
                // transposed scalar writes: the 64 lanes of a store hit banks 16*(l&3) + (l>>2) + const, all distinct
#pragma unroll
                for (int i = 0; i < WQ; ++i) {
                    const int u = i * 4 + wave;
                    if (u < 2 * NT) {
                        if constexpr (TERMS == 0) {
                            float* dst = Wt + ((u % NT) * 16 + (lane & 3) * 4) * PG_AS + (u / NT) * 16 + (lane >> 2);
                            dst[0] = rw[i].x; dst[PG_AS] = rw[i].y; dst[2 * PG_AS] = rw[i].z; dst[3 * PG_AS] = rw[i].w;
                        } else {
                            bf16x4 hi, lo;
                            split_bf16(rw[i], hi, lo);
                            const int o = ((u % NT) * 16 + (lane & 3) * 4) * BS + (u / NT) * 16 + (lane >> 2);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                Wh[o + j * BS] = hi[j];
                                if constexpr (TERMS == 3) Wl[o + j * BS] = lo[j];
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < WQ; ++i) {
                    const int e = tid + i * 256;
                    const int n = e >> 3, q = e & 7;
                    if (e < BN * 8) {
                        if constexpr (TERMS == 0) {
                            *reinterpret_cast<float4*>(Wt + n * PG_AS + q * 4) = rw[i];
                        } else {
                            bf16x4 hi, lo;
                            split_bf16(rw[i], hi, lo);
                            *reinterpret_cast<bf16x4*>(Wh + n * BS + q * 4) = hi;
                            if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(Wl + n * BS + q * 4) = lo;
                        }
                    }
                }
            }
        };

        const int k_begin = (p.ksplit > 1) ? blockIdx.z * p.kchunk : 0;
        const int k_end = (p.ksplit > 1) ? min(K, k_begin + p.kchunk) : K;
        const float* a_frag = As + (wave * 32 + lr) * PG_AS + lq * 8;
        const float* w_frag = Wt + lr * PG_AS + lq * 8;
        fetch(k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += PG_BK) {
            __syncthreads();
            commit(k0);
            __syncthreads();
            if (k0 + PG_BK < k_end) fetch(k0 + PG_BK);
            if constexpr (TERMS == 0) {
                // A fragments of the whole chunk up front (4 x b128); W fragments in groups of GS column blocks,
                // group g+1 requested before the 8*GS MFMAs of group g: LDS latency never waits on an idle pipe
                constexpr int GS = NT < 2 ? NT : 2;
                constexpr int GH = NT / GS;          // groups per half chunk
                float4 af[2][2], bf[2][GS];
    #pragma unroll
                for (int h = 0; h < 2; ++h) {
                    af[h][0] = *reinterpret_cast<const float4*>(a_frag + h * 4);
                    af[h][1] = *reinterpret_cast<const float4*>(a_frag + 16 * PG_AS + h * 4);
                }
    #pragma unroll
                for (int j = 0; j < GS; ++j) bf[0][j] = *reinterpret_cast<const float4*>(w_frag + j * 16 * PG_AS);
    #pragma unroll
                for (int g = 0; g < 2 * GH; ++g) {
                    const int h = g / GH, nb0 = (g % GH) * GS;
                    if (g + 1 < 2 * GH) {
                        const int h1 = (g + 1) / GH, nb1 = ((g + 1) % GH) * GS;
    #pragma unroll
                        for (int j = 0; j < GS; ++j)
                            bf[(g + 1) & 1][j] = *reinterpret_cast<const float4*>(w_frag + (nb1 + j) * 16 * PG_AS + h1 * 4);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler would sink them)
                    const float a0[4] = {af[h][0].x, af[h][0].y, af[h][0].z, af[h][0].w};
                    const float a1[4] = {af[h][1].x, af[h][1].y, af[h][1].z, af[h][1].w};
    #pragma unroll
                    for (int s = 0; s < 4; ++s) {
    #pragma unroll
                        for (int j = 0; j < GS; ++j) {
                            const float4 b4 = bf[g & 1][j];
                            const float bv = s == 0 ? b4.x : s == 1 ? b4.y : s == 2 ? b4.z : b4.w;
                            acc[0][nb0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bv, acc[0][nb0 + j], 0, 0, 0);
                            acc[1][nb0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bv, acc[1][nb0 + j], 0, 0, 0);
                        }
                    }
                }
            } else {
                // bf16 operands: one ds_read_b128 = the lane's 8 k-values of a 16x16x32 MFMA; per column-block pair
                // the next pair's fragments are requested before the 4*TERMS MFMAs of the current one
                const __bf16* ah_frag = Ah + (wave * 32 + lr) * BS + lq * 8;
                const __bf16* wh_frag = Wh + lr * BS + lq * 8;
                constexpr int LO_A = GM_BM * BS, LO_W = BN * BS;    // offsets of the tail arrays
                constexpr int GS = NT < 2 ? NT : 2;
                constexpr int NG = NT / GS;
                bf16x8 a_h[2], a_l[2], b_h[2][GS], b_l[2][GS];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    a_h[rb] = *reinterpret_cast<const bf16x8*>(ah_frag + rb * 16 * BS);
                    if constexpr (TERMS == 3) a_l[rb] = *reinterpret_cast<const bf16x8*>(ah_frag + LO_A + rb * 16 * BS);
                }
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    b_h[0][j] = *reinterpret_cast<const bf16x8*>(wh_frag + j * 16 * BS);
                    if constexpr (TERMS == 3) b_l[0][j] = *reinterpret_cast<const bf16x8*>(wh_frag + LO_W + j * 16 * BS);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int nb0 = g * GS;
                    if (g + 1 < NG) {
#pragma unroll
                        for (int j = 0; j < GS; ++j) {
                            b_h[(g + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(wh_frag + (nb0 + GS + j) * 16 * BS);
                            if constexpr (TERMS == 3)
                                b_l[(g + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(wh_frag + LO_W + (nb0 + GS + j) * 16 * BS);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // term by term over the group's accumulators: dependent MFMAs are 2*GS issues apart
#pragma unroll
                    for (int j = 0; j < GS; ++j)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
                            acc[rb][nb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h[rb], b_h[g & 1][j], acc[rb][nb0 + j], 0, 0, 0);
                    if constexpr (TERMS == 3) {
#pragma unroll
                        for (int j = 0; j < GS; ++j)
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb)
                                acc[rb][nb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h[rb], b_l[g & 1][j], acc[rb][nb0 + j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < GS; ++j)
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb)
                                acc[rb][nb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l[rb], b_h[g & 1][j], acc[rb][nb0 + j], 0, 0, 0);
                    }
                }
            }
        }
        if (p.ksplit > 1) {
            // split-K: raw partial tile to this split's slab; bias / accumulate / statistics happen in the reducer
            float* slab = p.kslab + (long)blockIdx.z * M * N;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long R = row0 + wave * 32 + rb * 16 + lq * 4 + r;
                    if (R < M) {
#pragma unroll
                        for (int nb = 0; nb < NT; ++nb) {
                            const int c = col0 + nb * 16 + lr;
                            if (c < N) slab[R * N + c] = acc[rb][nb][r];
                        }
                    }
                }
            continue;
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) tile_rows_epilogue<NT, true>(p, acc[rb], row0 + wave * 32 + rb * 16, col0, lr, lq, ssum, ssq);
    }
    if (p.stats && p.ksplit <= 1) {
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            float s = ssum[nb], q = ssq[nb];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) {
                red[wave][0][nb * 16 + lane] = (double)s;
                red[wave][1][nb * 16 + lane] = (double)q;
            }
        }
        __syncthreads();
        if (tid < BN && col0 + tid < N) {
            p.stats[((long)bx * 2 + 0) * N + col0 + tid] = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
            p.stats[((long)bx * 2 + 1) * N + col0 + tid] = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
            for (long slot = bx + p.gx; slot < p.stat_slots; slot += p.gx) {
                p.stats[(slot * 2 + 0) * N + col0 + tid] = 0.0;
                p.stats[(slot * 2 + 1) * N + col0 + tid] = 0.0;
            }
        }
    }
}


// ===========================================================================================
// wgemm_kernel: the wide GEMM again for the bf16 arithmetic modes, re-cut for memory latency.
// pgemm_kernel<8,3> runs two 4-wavefront workgroups per CU at 256 VGPRs: two wavefronts per SIMD, one 32-deep chunk
// of prefetch each - rocprofv3 shows its wavefronts parked on memory 58 % of their life (SQ_WAIT_ANY) while the matrix
// pipe is 15 % busy.  Here:
//   * the weight arrives already split into bf16 head / tail planes, [n][k] k-contiguous (rl_split_weights, once per
//     step for all wide layers): its staging is a 16-byte copy, no conversion, no transposed scalar LDS writes for the
//     dgrad orientation, and half the staging registers;
//   * eight wavefronts of 16 rows x 128 columns share a 128 x 128 tile (32 accumulator registers each instead of 64),
//     which fits 128 VGPRs: two workgroups per CU = FOUR wavefronts per SIMD, twice the loads in flight.
// Everything else (lazy BatchNorm on the A operand, statistics epilogue, split-K, split epilogue, XCD-aware tile order)
// is pgemm_kernel's.
// ===========================================================================================
template <int TERMS, bool STATS>   // TERMS 3: bf16x3, 1: bf16; STATS: BatchNorm partial statistics in the epilogue
__global__ __launch_bounds__(512, 4) void wgemm_kernel(const GemmParams p) {
    constexpr int BN = 128, BS = 40, NT = 8;
    constexpr int NSPL = TERMS == 3 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) __bf16 lds_a[GM_BM * BS * NSPL];
    __shared__ __attribute__((aligned(16))) __bf16 lds_w[BN * BS * NSPL];
    __bf16* Ah = lds_a;
    __bf16* Al = lds_a + GM_BM * BS;       // only touched when TERMS == 3
    __bf16* Wh = lds_w;
    __bf16* Wl = lds_w + BN * BS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int lr = lane & 15, lq = lane >> 4;
    int bx = blockIdx.x, by = 0;
    if (p.ny > 1) {
        const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
        by = s % p.ny;
        bx = (s / p.ny) * 8 + xcd;
        if (bx >= p.gx) return;      // padding of the last group of eight
    }
    const int col0 = by * BN;
    const int K = p.a.K, N = p.N;
    const long M = p.a.M;
    const long ntiles = (M + GM_BM - 1) / GM_BM;
    const bool lazy = p.a.lazy.scale != nullptr;
    const bool relu = p.a.lazy.act == RL_ACT_RELU;
    const float nslope = p.a.lazy.act == RL_ACT_NONE ? 1.f : p.a.lazy.slope;
    const int aq = tid & 7;            // this lane's k-quad inside a chunk (same for its 2 rows)
    const int arow = tid >> 3;         // 0..63, +64 for the second row
    // weight staging: per plane 128 rows x 4 pieces of 16 bytes; 256 lanes per plane, two pieces each
    const int wplane = tid >> 8;       // 0 head, 1 tail
    const int wpiece = (tid & 255) * 2;
    const __bf16* wsrc = p.wsplit + (wplane ? (long)N * K : 0);
    const bool w_active = wplane < NSPL;

    // BatchNorm partial statistics: a tile's column sums exist only during its epilogue (16 registers that would
    // otherwise be alive through the K loop and push the kernel over its 128-VGPR budget into scratch); the workgroup's
    // running totals are two doubles in the first 128 lanes.
    double tot_s = 0.0, tot_q = 0.0;

    for (long tile = bx; tile < ntiles; tile += p.gx) {
        const long row0 = tile * GM_BM;
        long aoff[2];
        bool aval[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long R = row0 + arow + 64 * i;
            aval[i] = R < M;
            aoff[i] = aval[i] ? a_row_offset(p.a, R) : 0;
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        float4 ra[2];
        bf16x8 rw[2];
        auto fetch = [&](int k0) {
            const int ka = k0 + aq * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (aval[i] && ka < K) ra[i] = *reinterpret_cast<const float4*>(p.a.A + aoff[i] + ka);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int piece = wpiece + j, n = piece >> 2, q = piece & 3;
#pragma unroll
                for (int e = 0; e < 8; ++e) rw[j][e] = (__bf16)0.f;
                if (w_active && col0 + n < N && k0 + q * 8 < K)
                    rw[j] = *reinterpret_cast<const bf16x8*>(wsrc + (long)(col0 + n) * K + k0 + q * 8);
            }
        };
        auto actf = [&](float z) {
            const float neg = relu ? 0.f : z * nslope;
            return z > 0.f ? z : neg;
        };
        auto commit = [&](int k0) {
            const int ka = k0 + aq * 4;
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lazy && ka < K) {
                sc = *reinterpret_cast<const float4*>(p.a.lazy.scale + ka);
                sh = *reinterpret_cast<const float4*>(p.a.lazy.shift + ka);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float4 v = ra[i];
                if (lazy && aval[i] && ka < K) {
                    v.x = actf(v.x * sc.x + sh.x);
                    v.y = actf(v.y * sc.y + sh.y);
                    v.z = actf(v.z * sc.z + sh.z);
                    v.w = actf(v.w * sc.w + sh.w);
                }
                bf16x4 hi, lo;
                split_bf16(v, hi, lo);
                *reinterpret_cast<bf16x4*>(Ah + (arow + 64 * i) * BS + aq * 4) = hi;
                if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(Al + (arow + 64 * i) * BS + aq * 4) = lo;
            }
            if (w_active) {
                __bf16* dstp = wplane ? Wl : Wh;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int piece = wpiece + j, n = piece >> 2, q = piece & 3;
                    *reinterpret_cast<bf16x8*>(dstp + n * BS + q * 8) = rw[j];
                }
            }
        };

        const int k_begin = (p.ksplit > 1) ? blockIdx.z * p.kchunk : 0;
        const int k_end = (p.ksplit > 1) ? min(K, k_begin + p.kchunk) : K;
        const __bf16* ah_frag = Ah + (wave * 16 + lr) * BS + lq * 8;
        const __bf16* wh_frag = Wh + lr * BS + lq * 8;
        constexpr int LO_A = GM_BM * BS, LO_W = BN * BS;
        fetch(k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += PG_BK) {
            __syncthreads();
            commit(k0);
            __syncthreads();
            if (k0 + PG_BK < k_end) fetch(k0 + PG_BK);
            // four wavefronts per SIMD hide the LDS latency between them: the B fragments are not double-buffered
            // (16 registers instead of 32 keep the kernel inside its 128-VGPR budget)
            bf16x8 a_h, a_l;
            a_h = *reinterpret_cast<const bf16x8*>(ah_frag);
            if constexpr (TERMS == 3) a_l = *reinterpret_cast<const bf16x8*>(ah_frag + LO_A);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb0 = g * 2;
                bf16x8 b_h[2], b_l[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b_h[j] = *reinterpret_cast<const bf16x8*>(wh_frag + (nb0 + j) * 16 * BS);
                    if constexpr (TERMS == 3) b_l[j] = *reinterpret_cast<const bf16x8*>(wh_frag + LO_W + (nb0 + j) * 16 * BS);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[nb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h[j], acc[nb0 + j], 0, 0, 0);
                if constexpr (TERMS == 3) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[nb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l[j], acc[nb0 + j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[nb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h[j], acc[nb0 + j], 0, 0, 0);
                }
            }
        }
        if (p.ksplit > 1) {
            float* slab = p.kslab + (long)blockIdx.z * M * N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long R = row0 + wave * 16 + lq * 4 + r;
                if (R < M) {
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        const int c = col0 + nb * 16 + lr;
                        if (c < N) slab[R * N + c] = acc[nb][r];
                    }
                }
            }
            continue;
        }
        float ssum[NT], ssq[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) ssum[nb] = ssq[nb] = 0.f;
        tile_rows_epilogue<NT, STATS>(p, acc, row0 + wave * 16, col0, lr, lq, ssum, ssq);
        if constexpr (STATS) if (p.stats && p.ksplit <= 1) {
            __syncthreads();                                        // the operand tiles are free: reuse them for the reduction
            double* red = reinterpret_cast<double*>(lds_a);         // [8][2][128] doubles = 16 KB <= 20 KB (10 KB in bf16 mode: use both)
            static_assert(sizeof(lds_a) + sizeof(lds_w) >= 8 * 2 * BN * sizeof(double), "statistics scratch");
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                float sv = ssum[nb], qv = ssq[nb];
                sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
                qv += __shfl_xor(qv, 16, 64); qv += __shfl_xor(qv, 32, 64);
                if (lane < 16) {
                    red[(wave * 2 + 0) * BN + nb * 16 + lane] = (double)sv;
                    red[(wave * 2 + 1) * BN + nb * 16 + lane] = (double)qv;
                }
            }
            __syncthreads();
            if (tid < BN) {
                for (int w = 0; w < 8; ++w) {
                    tot_s += red[(w * 2 + 0) * BN + tid];
                    tot_q += red[(w * 2 + 1) * BN + tid];
                }
            }
            // (the next tile's first staging write is behind the barrier at the top of its K loop)
        }
    }
    if constexpr (STATS) if (p.stats && p.ksplit <= 1) {
        if (tid < BN && col0 + tid < N) {
            p.stats[((long)bx * 2 + 0) * N + col0 + tid] = tot_s;
            p.stats[((long)bx * 2 + 1) * N + col0 + tid] = tot_q;
            for (long slot = bx + p.gx; slot < p.stat_slots; slot += p.gx) {
                p.stats[(slot * 2 + 0) * N + col0 + tid] = 0.0;
                p.stats[(slot * 2 + 1) * N + col0 + tid] = 0.0;
            }
        }
    }
}

// ===========================================================================================
// The same epilogue for a TRANSPOSED accumulator (the MFMA operands exchanged, round 4): acc[nb][j] = element (row lr, column
// col0 + nb * 16 + lq * 4 + j) - a lane owns four consecutive columns of ITS row, so a block leaves (and its addend / old Y
// arrive) in 16-byte pieces: 8 stores per lane and 128-column block instead of 32.  No statistics here (a launch that wants them
// keeps the column-per-lane layout, whose column sums are two shuffles).  `vec`: N, ldy and the pointers allow 16-byte accesses.
template <int NT>
__device__ __forceinline__ void tile_rows_epilogue_t(const GemmParams& p, const f32x4 (&acc)[NT], long Rb, int col0, int lr, int lq, bool vec) {
    constexpr int BN = 16 * NT;
    const int N = p.N;
    const long R = Rb + lr;
    if (R >= p.a.M) return;
    const bool to_y = !p.out2 || col0 + BN <= p.split_col;
    const bool to_out2 = p.out2 && !p.out2_index && col0 >= p.split_col;
    long yoff;
    if (p.y_contig) yoff = R * p.ldy;
    else {
        const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
        const int i = (int)(R - (long)b * p.rows_per_batch);
        yoff = ((long)b * p.y_bstride + i) * p.ldy;
    }
    const int c0 = col0 + 4 * lq;                   // this lane's first column of block 0
    if (vec && (to_y || to_out2)) {
        float* y = to_y ? p.Y + yoff + c0 : p.out2 + R * (N - p.split_col) - p.split_col + c0;
        const bool full = col0 + BN <= N;
        const bool accumulate = to_y && p.accumulate;
        f32x4 v[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) v[nb] = acc[nb];
        if (p.bias) {
#pragma unroll
            for (int nb = 0; nb < NT; ++nb)
                if (full || c0 + nb * 16 < N) v[nb] += *reinterpret_cast<const f32x4*>(p.bias + c0 + nb * 16);
        }
        if (p.addend) {
            const float* ad = p.addend + R * N + c0;
            f32x4 o[NT];
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) o[nb] = (full || c0 + nb * 16 < N) ? *reinterpret_cast<const f32x4*>(ad + nb * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) v[nb] += o[nb];
        }
        if (accumulate) {
            f32x4 o[NT];
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) o[nb] = (full || c0 + nb * 16 < N) ? *reinterpret_cast<const f32x4*>(y + nb * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) v[nb] += o[nb];
        }
#pragma unroll
        for (int nb = 0; nb < NT; ++nb)
            if (full || c0 + nb * 16 < N) *reinterpret_cast<f32x4*>(y + nb * 16) = v[nb];
        return;
    }
    // element by element: a tile that straddles split_col, the indexed (atomic) out2, or operands that rule out 16-byte accesses
    long o2 = 0;
    if (p.out2) {
        if (p.out2_index) {
            const int b2 = (int)((unsigned)R / (unsigned)p.rows_per_batch);
            o2 = ((long)b2 * p.out2_bstride + p.out2_index[R]) * (N - p.split_col) - p.split_col;
        } else o2 = R * (N - p.split_col) - p.split_col;
    }
#pragma unroll
    for (int nb = 0; nb < NT; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + nb * 16 + j;
            if (c < N) {
                float t = acc[nb][j];
                if (p.bias) t += p.bias[c];
                if (p.addend) t += p.addend[R * N + c];
                if (p.out2 && c >= p.split_col) {
                    if (p.out2_index) atomicAdd(p.out2 + o2 + c, t);
                    else p.out2[o2 + c] = t;
                } else {
                    if (p.accumulate) t += p.Y[yoff + c];
                    p.Y[yoff + c] = t;
                }
            }
        }
}

// wgemm2_kernel: wgemm_kernel's tile and arithmetic as a PERSISTENT workgroup per CU whose operands arrive by LDS-DMA
// (global_load_lds_dwordx4) from dedicated loader wavefronts.
// wgemm_kernel keeps one 32-deep chunk of loads in flight per workgroup, in registers, behind two barriers per chunk; its
// wavefronts wait 63 % of their life (profiles/r03_pmc_sq_step.md), the under-filled launches of the deep levels pay one
// full memory latency per chunk and every tile starts with an empty pipeline.  Here
//   * twelve wavefronts: 0-7 compute (16 rows x 128 columns each, as before), 8-9 issue the A pieces, 10-11 the W pieces.
//     vmcnt retires in order PER WAVEFRONT: the loaders' queues hold nothing but their own DMA (no epilogue stores, and
//     the deep A queue does not drain for the shallow W queue), so a counted s_waitcnt says exactly which chunk landed;
//   * the fp32 A tile goes to LDS RAW, AS chunks deep (no staging registers: the depth is an LDS budget); the lazy
//     BatchNorm + activation and the bf16 head / tail split happen on the fragment, after the ds_read - every element is
//     still converted exactly once, by the wavefront that owns its 16 rows; the pre-split weight planes follow in a ring of their own, WS chunks deep
//     (two put the weight's L2 latency on every step's critical path);
//   * the chunk stream runs ACROSS the workgroup's tiles: while the compute wavefronts store tile t, the first chunks of
//     tile t+1 are already in flight;
//   * one raw s_barrier per chunk.  Loaders: wait (counted vmcnt) -> barrier -> refill the stage chunk q-1 freed.
//     Compute: barrier -> fragments -> 24 MFMAs (type-major: dependent MFMAs eight apart);
//   * LDS image of an A chunk: [128 rows][8 pieces of 16 B], piece p of row r stored at p ^ ((r >> 1) & 7) - the DMA writes
//     lane-linear (1 KB per wavefront instruction = 8 rows), so the swizzle is applied to the SOURCE address and again
//     on the fragment read: the 64 lanes of a ds_read_b128 fall on 16 distinct 16-byte slots, 4 lanes each.
// LDS: AS * 16 (A) + WS * 16 (W) + 8 (scale / shift) + 16 (statistics) KB = 152 KB at AS = WS = 4; 3 wavefronts per SIMD (124 of
// 168 VGPRs).
// Same products in the same order per accumulator as wgemm_kernel: Y is bitwise the same; the BatchNorm partial sums are
// grouped per wavefront over the workgroup's tiles (another, equally fixed, order of the same doubles).
// Needs K % 32 == 0 and K <= 1024.
// ===========================================================================================
// The DMA is issued from inline asm ON PURPOSE: hipcc (ROCm 7.2) follows a __builtin_amdgcn_global_load_lds with
// s_waitcnt vmcnt(0) in front of the next ds_read of the same __shared__ object - every chunk would wait for the loads it
// has just issued.  Hidden in asm, the only waits are the counted ones below.  l: LDS byte address (wavefront-uniform),
// the wavefront's 64 lanes land at l + 16 * lane.
typedef __attribute__((address_space(3))) unsigned char rl_lds_byte;
__device__ __forceinline__ unsigned lds_address(const void* q) {
    return (unsigned)(uintptr_t)(const rl_lds_byte*)q;
}
__device__ __forceinline__ void glds16x4(const void* g0, const void* g1, const void* g2, const void* g3, unsigned l0, unsigned l1,
                                         unsigned l2, unsigned l3) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(l0), "s"(l1), "s"(l2), "s"(l3) : "memory");
}
__device__ __forceinline__ void glds16x2(const void* g0, const void* g1, unsigned l0, unsigned l1) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g0), "v"(g1), "s"(l0), "s"(l1) : "memory");
}
template <int PER>       // s_waitcnt vmcnt(PER * n): PER instructions per chunk, n = 0 .. 4 chunks may stay in flight (wavefront-uniform)
__device__ __forceinline__ void wait_vm_chunks(int n) {
    static_assert(PER == 2 || PER == 4 || PER == 8, "pieces per loader wavefront and chunk");
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) { if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else if (n == 2) { if (PER == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else if (PER == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else if (n == 3) { if (PER == 8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else if (PER == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else { if (PER == 8) asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); else if (PER == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
}
__device__ __forceinline__ void raw_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");          // (compiler-only: no LDS access may move over the barrier)
}

#ifndef W2_ABLATE
#define W2_ABLATE 0      // diagnostics builds only: 1 = a quarter of the B fragment reads, 2 = no Y stores, 3 = no MFMAs
#endif
constexpr int W2_KMAX = 1024;
constexpr int W2_THREADS = 768;
// Round 6: the output tile is a template parameter, BM x BN = 128 x 128 (above), 64 x 128 or 64 x 64 - chosen by the host
// (wide_plan) so that the launches of the deep levels (M = 1280 ... 20480 rows at batch 8: 40 - 160 tiles of 128 x 128 on 256 CUs,
// most of them behind a K split and its reducer launch) put >= 2x the workgroups on the chip in ONE pass.  The eight compute
// wavefronts then form BM / 16 row blocks x CG column groups (64-row tiles: 4 x 2), a wavefront owns 16 rows x BN / CG columns;
// the rings hold BM-row A chunks and BN-column weight chunks (same depth, less LDS), the loaders issue BM / 16 and
// BN / 16 (x 2 planes) pieces per chunk.  Per accumulator the same products in the same order as the 128 x 128 tile: Y is
// bitwise the same whatever the tile; only the grouping of the BatchNorm partial sums (slots per 64 rows) differs.
template <int TERMS, bool STATS, int AS, int WS, int BM = 128, int BN = 128>
__global__ __launch_bounds__(W2_THREADS) void wgemm2_kernel(const GemmParams pk) {
    static_assert(AS >= 3 && AS <= 6 && WS >= 2 && WS <= 6, "ring depths vs the wait table");
    static_assert((BM == 128 || BM == 64) && (BN == 128 || BN == 64), "tile shapes");
    constexpr int RB = BM / 16;                 // 16-row blocks of a tile
    constexpr int CG = 8 / RB;                  // column groups the eight compute wavefronts form
    constexpr int NT = BN / 16 / CG;            // 16-column blocks per wavefront
    constexpr bool TR = !STATS;                 // transposed accumulator (16-byte epilogue) where no column statistics are wanted
    constexpr int A_STAGE = BM * 128;           // bytes: BM rows x 32 fp32
    constexpr int W_PLANE = BN * 64;            // bytes: BN columns x 32 bf16
    constexpr int W_STAGE = 2 * W_PLANE;
    constexpr int OFF_W = AS * A_STAGE, OFF_SC = OFF_W + WS * W_STAGE, OFF_RED = OFF_SC + 2 * W2_KMAX * 4;
    constexpr int TOTAL = OFF_RED + (STATS ? 8 * 2 * BN * 8 : 0);
    __shared__ __attribute__((aligned(1024))) unsigned char lds[TOTAL];       // ONE object
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = 0;
    if (pk.ny > 1) {
        const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
        by = s % pk.ny;
        bx = (s / pk.ny) * 8 + xcd;
        if (bx >= pk.gx) return;
    }
    // a pair launch (rl_gemm_pair: mlp1 + shortcut of an encoder level behind ONE read of their common input): the column blocks
    // from pair_ny1 on compute the second product - wavefront-uniform, decided once: everything below reads `p`
    GemmParams p = pk;
    if (pk.pair_ny1 > 0 && by >= pk.pair_ny1) {
        by -= pk.pair_ny1;
        p.N = pk.N2; p.Y = pk.Y2; p.ldy = pk.N2; p.stats = pk.stats2;
        p.piv_mean = pk.piv_mean2; p.piv_bias = pk.piv_bias2; p.wsplit = pk.wsplit2;
        p.split_col = pk.N2;
    }
    const int col0 = by * BN;
    const int K = p.a.K, N = p.N;
    const long M = p.a.M;
    const long ntiles = (M + BM - 1) / BM;
    const bool lazy = p.a.lazy.scale != nullptr;
    float* lsc = reinterpret_cast<float*>(lds + OFF_SC);
    float* lsh = lsc + W2_KMAX;
    double* red = reinterpret_cast<double*>(lds + OFF_RED);        // [8 wavefronts][2][BN]
    const unsigned lds0 = lds_address(lds);
    if (lazy) {
        for (int k = tid; k < K; k += W2_THREADS) {
            lsc[k] = p.a.lazy.scale[k];
            lsh[k] = p.a.lazy.shift[k];
        }
    }
    if constexpr (STATS) {
        for (int i = tid; i < 8 * 2 * BN; i += W2_THREADS) red[i] = 0.0;
    }
    __syncthreads();

    const int k_begin = (p.ksplit > 1) ? blockIdx.z * p.kchunk : 0;
    const int k_end = (p.ksplit > 1) ? min(K, k_begin + p.kchunk) : K;
    const int nch = (k_end - k_begin) / PG_BK;
    // 16-byte accesses of the transposed epilogue: every row start and column group 16-byte aligned in all tensors it touches
    const bool vec_t = (N % 4 == 0) && (p.ldy % 4 == 0) && (p.split_col % 4 == 0) &&
                       (((reinterpret_cast<uintptr_t>(p.Y) | reinterpret_cast<uintptr_t>(p.out2) | reinterpret_cast<uintptr_t>(p.addend) |
                          reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.kslab)) & 15) == 0);
    const int my_tiles = bx < ntiles ? (int)((ntiles - 1 - bx) / p.gx) + 1 : 0;
    const int T = my_tiles * nch;                                  // chunk steps of this workgroup = barriers every wavefront passes

    if (wave >= 8) __builtin_amdgcn_s_setprio(3);      // the loaders' few instructions go first: 1 - 2 % on the mid-size shapes
    if (wave >= 10) {
        // ---- W loaders: BN / 16 pieces of 1 KB per plane and chunk (16 at 128 columns in bf16x3), WS - 1 chunks ahead --------
        constexpr int PPP = BN / 16;                               // pieces per plane
        constexpr int WI = (TERMS == 3 ? 2 : 1) * PPP / 2;         // per loader wavefront: 8 / 4 / 2
        const int lw = wave - 10;
        const __bf16* wsrc[WI];
        unsigned wdst[WI];
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int idx = lw * WI + j, plane = idx / PPP, rg = idx % PPP;
            int col = col0 + rg * 16 + (lane >> 2);
            col = col < N ? col : N - 1;        // columns past N are never stored
            wsrc[j] = p.wsplit + (long)plane * N * K + (long)col * K + (lane & 3) * 8 + k_begin;
            wdst[j] = lds0 + OFF_W + plane * W_PLANE + rg * 1024;
        }
        auto issue_w = [&](int q) {
            const int k = (q % nch) * PG_BK;
            const unsigned o = __builtin_amdgcn_readfirstlane((q % WS) * W_STAGE);
            if constexpr (WI % 4 == 0) {
#pragma unroll
                for (int j = 0; j < WI; j += 4)
                    glds16x4(wsrc[j] + k, wsrc[j + 1] + k, wsrc[j + 2] + k, wsrc[j + 3] + k, wdst[j] + o, wdst[j + 1] + o, wdst[j + 2] + o,
                             wdst[j + 3] + o);
            } else {
                glds16x2(wsrc[0] + k, wsrc[1] + k, wdst[0] + o, wdst[1] + o);
            }
        };
        for (int q = 0; q < WS - 1 && q < T; ++q) issue_w(q);
        for (int q = 0; q < T; ++q) {
            wait_vm_chunks<WI>(min(WS - 2, T - 1 - q));
            raw_barrier();
            if (q + WS - 1 < T) issue_w(q + WS - 1);
        }
    } else if (wave >= 8) {
        // ---- A loaders: BM / 8 pieces of 1 KB (8 rows x 128 B) per chunk, AS - 1 chunks ahead, across the tiles --------------
        constexpr int AI = BM / 16;             // pieces per loader wavefront and chunk
        const int lw = wave - 8;
        const float* asrc[AI];
        auto tile_sources = [&](int j) {        // j-th tile of this workgroup
            const long row0 = ((long)bx + (long)j * p.gx) * BM;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int r = (lw * AI + i) * 8 + (lane >> 3);
                long R = row0 + r;
                R = R < M ? R : M - 1;          // rows past M are never stored
                asrc[i] = p.a.A + a_row_offset(p.a, R) + (((lane & 7) ^ ((r >> 1) & 7)) << 2) + k_begin;
            }
        };
        int iq = 0, ic = 0;                     // next chunk step to issue, its chunk inside its tile
        auto issue_a = [&]() {
            if (ic == 0) tile_sources(iq / nch);
            const int k = ic * PG_BK;
            const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + (iq % AS) * A_STAGE + lw * AI * 1024);
            glds16x4(asrc[0] + k, asrc[1] + k, asrc[2] + k, asrc[3] + k, l, l + 1024, l + 2048, l + 3072);
            if constexpr (AI == 8) glds16x4(asrc[4] + k, asrc[5] + k, asrc[6] + k, asrc[7] + k, l + 4096, l + 5120, l + 6144, l + 7168);
            ++iq;
            ic = ic + 1 == nch ? 0 : ic + 1;
        };
        while (iq < AS - 1 && iq < T) issue_a();
        for (int q = 0; q < T; ++q) {
            wait_vm_chunks<AI>(min(AS - 2, T - 1 - q));
            raw_barrier();
            if (iq < T) issue_a();
        }
    } else {
        // ---- compute ---------------------------------------------------------------------------------------------------------
        const int lr = lane & 15, lq = lane >> 4;
        const bool relu = p.a.lazy.act == RL_ACT_RELU;
        const float nslope = p.a.lazy.act == RL_ACT_NONE ? 1.f : p.a.lazy.slope;
        const int rb = wave % RB, cg = wave / RB;                    // this wavefront's row block and column group of the tile
        const int cg0 = cg * NT * 16;                                // its first column inside the tile
        const int frow = rb * 16 + lr;                               // this lane's fragment row in the tile
        const int fsw = (frow >> 1) & 7;
        const int a_frag0 = frow * 128 + (((2 * lq) ^ fsw) << 4), a_frag1 = frow * 128 + (((2 * lq + 1) ^ fsw) << 4);
        const int w_frag = lr * 64 + lq * 16 + cg0 * 64;
        auto actf = [&](float z) {
            const float neg = relu ? 0.f : z * nslope;
            return z > 0.f ? z : neg;
        };
        f32x4 acc[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int c = 0;
        long tile = bx;
        for (int q = 0; q < T; ++q) {
            raw_barrier();
            const unsigned char* Ab = lds + (q % AS) * A_STAGE;
            const unsigned char* Wb = lds + OFF_W + (q % WS) * W_STAGE;
            float4 v0 = *reinterpret_cast<const float4*>(Ab + a_frag0);
            float4 v1 = *reinterpret_cast<const float4*>(Ab + a_frag1);
            if (lazy) {
                const int kk = k_begin + c * PG_BK + lq * 8;
                const float4 sc0 = *reinterpret_cast<const float4*>(lsc + kk), sc1 = *reinterpret_cast<const float4*>(lsc + kk + 4);
                const float4 sh0 = *reinterpret_cast<const float4*>(lsh + kk), sh1 = *reinterpret_cast<const float4*>(lsh + kk + 4);
                v0.x = actf(v0.x * sc0.x + sh0.x); v0.y = actf(v0.y * sc0.y + sh0.y);
                v0.z = actf(v0.z * sc0.z + sh0.z); v0.w = actf(v0.w * sc0.w + sh0.w);
                v1.x = actf(v1.x * sc1.x + sh1.x); v1.y = actf(v1.y * sc1.y + sh1.y);
                v1.z = actf(v1.z * sc1.z + sh1.z); v1.w = actf(v1.w * sc1.w + sh1.w);
            }
            bf16x4 h0, l0, h1, l1;
            split_bf16(v0, h0, l0);
            split_bf16(v1, h1, l1);
            const bf16x8 a_h = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 a_l = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            bf16x8 b_h[NT], b_l[NT];
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
#if W2_ABLATE == 1
                const int nbs = nb & 1;
#else
                const int nbs = nb;
#endif
                b_h[nb] = *reinterpret_cast<const bf16x8*>(Wb + w_frag + nbs * 1024);
                if constexpr (TERMS == 3) b_l[nb] = *reinterpret_cast<const bf16x8*>(Wb + W_PLANE + w_frag + nbs * 1024);
            }
#if W2_ABLATE == 3
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) { acc[nb][0] += (float)b_h[nb][0] * (float)a_h[0]; acc[nb][1] += (float)b_l[nb][1] * (float)a_l[1]; }
#else
            // TR (launches without statistics): operands exchanged - the tile comes out transposed, a lane owns four consecutive
            // columns of its row (tile_rows_epilogue_t); the same products in the same order, the same bits
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) acc[nb] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_h[nb], a_h, acc[nb], 0, 0, 0)
                                                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h[nb], acc[nb], 0, 0, 0);
            if constexpr (TERMS == 3) {
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) acc[nb] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_l[nb], a_h, acc[nb], 0, 0, 0)
                                                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l[nb], acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) acc[nb] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_h[nb], a_l, acc[nb], 0, 0, 0)
                                                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h[nb], acc[nb], 0, 0, 0);
            }
#endif
            if (++c < nch) continue;
            // ---- the tile is complete: epilogue (the loaders are already AS - 1 chunks into the next tile) -------------------
            c = 0;
            const long row0 = tile * BM;
            tile += p.gx;
#if W2_ABLATE == 4
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) asm volatile("" :: "v"(acc[nb]));
            if (p.ksplit > 1000) {
#else
            if (p.ksplit > 1) {
#endif
                float* slab = p.kslab + (long)blockIdx.z * M * N;
                if constexpr (TR) {
                    const long R = row0 + rb * 16 + lr;
                    if (R < M) {
#pragma unroll
                        for (int nb = 0; nb < NT; ++nb) {
                            const int cc = col0 + cg0 + nb * 16 + lq * 4;
                            if (vec_t) {
                                if (cc < N) *reinterpret_cast<f32x4*>(slab + R * N + cc) = acc[nb];
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (cc + j < N) slab[R * N + cc + j] = acc[nb][j];
                            }
                        }
                    }
                } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long R = row0 + rb * 16 + lq * 4 + r;
                    if (R < M) {
#pragma unroll
                        for (int nb = 0; nb < NT; ++nb) {
                            const int cc = col0 + cg0 + nb * 16 + lr;
                            if (cc < N) slab[R * N + cc] = acc[nb][r];
                        }
                    }
                }
                }
#if W2_ABLATE == 4
            } else if (p.ksplit > 1000) {
#else
            } else {
#endif
                float ssum[NT], ssq[NT];
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) ssum[nb] = ssq[nb] = 0.f;
                if constexpr (TR) tile_rows_epilogue_t<NT>(p, acc, row0 + rb * 16, col0 + cg0, lr, lq, vec_t);
                else tile_rows_epilogue<NT, STATS>(p, acc, row0 + rb * 16, col0 + cg0, lr, lq, ssum, ssq);
                if constexpr (STATS) if (p.stats) {
                    // this wavefront's own slice of the scratch: no other wavefront touches it before the final barrier
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        float sv = ssum[nb], qv = ssq[nb];
                        sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
                        qv += __shfl_xor(qv, 16, 64); qv += __shfl_xor(qv, 32, 64);
                        if (lane < 16) {
                            red[(wave * 2 + 0) * BN + cg0 + nb * 16 + lane] += (double)sv;
                            red[(wave * 2 + 1) * BN + cg0 + nb * 16 + lane] += (double)qv;
                        }
                    }
                }
            }
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    if constexpr (STATS) if (p.stats && p.ksplit <= 1) {
        __syncthreads();
        if (tid < BN && col0 + tid < N) {
            double tot_s = 0.0, tot_q = 0.0;
            for (int w = 0; w < 8; ++w) {
                tot_s += red[(w * 2 + 0) * BN + tid];
                tot_q += red[(w * 2 + 1) * BN + tid];
            }
            p.stats[((long)bx * 2 + 0) * N + col0 + tid] = tot_s;
            p.stats[((long)bx * 2 + 1) * N + col0 + tid] = tot_q;
            for (long slot = bx + p.gx; slot < p.stat_slots; slot += p.gx) {
                p.stats[(slot * 2 + 0) * N + col0 + tid] = 0.0;
                p.stats[(slot * 2 + 1) * N + col0 + tid] = 0.0;
            }
        }
    }
}

// bf16 head / tail planes of a weight in the orientation a GEMM reads it: out[n*K + k] = head(W[k*w_ks + n*w_ns]),
// tails N*K elements further.  All wide layers of a step in one launch.
constexpr int WS_MAX = 64;
struct SplitBatch {
    rl_wsplit_item item[WS_MAX];
    int first_block[WS_MAX + 1];
    int count;
};
// (round 6) a lane owns EIGHT consecutive k of one column n: one 16-byte store per plane instead of eight 2-byte ones, 32-bit
// index arithmetic (the element index went through an emulated 64-bit division), and the lanes of a wavefront run along
// whichever index is contiguous in the weight as stored - k for the forward orientation (two float4 loads per lane), n for the
// input-gradient orientation (eight loads, each coalesced across the lanes): 28 -> 9 us for the 37 weight uses of a step.
__global__ __launch_bounds__(256) void split_weights_kernel(const SplitBatch b) {
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.first_block[i + 1]) ++i;
    const rl_wsplit_item& it = b.item[i];
    const unsigned K = (unsigned)it.K, N = (unsigned)it.N, kg = K >> 3;            // K % 8 == 0 (host check)
    const unsigned e = ((unsigned)blockIdx.x - (unsigned)b.first_block[i]) * 256u + threadIdx.x;
    if (e >= N * kg) return;
    unsigned n, g;
    if (it.w_ns == 1) { g = e / N; n = e - g * N; }        // W stored n-contiguous: consecutive lanes = consecutive n
    else { n = e / kg; g = e - n * kg; }                    // k-contiguous (or any other strides): consecutive lanes = consecutive k groups
    const unsigned k0 = g * 8u;
    const float* src = it.W + (long)k0 * it.w_ks + (long)n * it.w_ns;
    float w[8];
    if (it.w_ks == 1 && ((((uintptr_t)src) & 15) == 0)) {
        const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = c.x; w[5] = c.y; w[6] = c.z; w[7] = c.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = src[(long)j * it.w_ks];
    }
    bf16x8 h, t;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = (__bf16)w[j];
        t[j] = (__bf16)(w[j] - (float)h[j]);
    }
    __bf16* out = reinterpret_cast<__bf16*>(it.out);
    const unsigned o = n * K + k0;
    *reinterpret_cast<bf16x8*>(out + o) = h;
    *reinterpret_cast<bf16x8*>(out + (unsigned long)N * K + o) = t;
}

// Split-K reducer: Y = sum over splits (fixed order) + bias (+ Y when accumulating), plus the BatchNorm
// partial statistics, one slot per workgroup exactly like the single-pass kernels (rl_row_blocks(M,128)).
// Needs N % 4 == 0.
template <int CB>   // columns per workgroup: 64, or 16 when there are too few row tiles to fill the chip otherwise
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmParams p) {
    // workgroup (x, y): row tiles x, x + gridDim.x, ... (x = the statistics slot) and columns [CB*y, CB*y+CB)
    constexpr int QN = CB / 4, RP = 256 / QN;   // column quads, rows in flight
    __shared__ float red[256][9];
    const int N = p.N;
    const long M = p.a.M;
    const int q = threadIdx.x & (QN - 1), rsub = threadIdx.x / QN;
    const int c = blockIdx.y * CB + q * 4;
    const bool cvalid = c < N;
    const long ntiles = (M + GM_BM - 1) / GM_BM;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && cvalid) bias = make_float4(p.bias[c], p.bias[c + 1], p.bias[c + 2], p.bias[c + 3]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 piv = make_float4(0.f, 0.f, 0.f, 0.f);        // shifted statistics, see GemmParams
    if (cvalid) piv = make_float4(stat_pivot(p, c), stat_pivot(p, c + 1), stat_pivot(p, c + 2), stat_pivot(p, c + 3));
    if (cvalid) {
        for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const long rend = min(M, (tile + 1) * GM_BM);
            for (long R = tile * GM_BM + rsub; R < rend; R += RP) {
                float4 v = bias;
                for (int s = 0; s < p.ksplit; ++s) {
                    const float4 t = *reinterpret_cast<const float4*>(p.kslab + ((long)s * M + R) * N + c);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                long yoff;
                if (p.y_contig) yoff = R * p.ldy;
                else {
                    const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                    const int i = (int)(R - (long)b * p.rows_per_batch);
                    yoff = ((long)b * p.y_bstride + i) * p.ldy;
                }
                float* y = p.Y + yoff + c;
                if (p.accumulate) { v.x += y[0]; v.y += y[1]; v.z += y[2]; v.w += y[3]; }
                y[0] = v.x; y[1] = v.y; y[2] = v.z; y[3] = v.w;
                v.x -= piv.x; v.y -= piv.y; v.z -= piv.z; v.w -= piv.w;
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                acc[4] += v.x * v.x; acc[5] += v.y * v.y; acc[6] += v.z * v.z; acc[7] += v.w * v.w;
            }
        }
    }
    if (!p.stats) return;
    // lanes with equal (l & (QN-1)) share a column quad inside a wavefront; then the four wavefronts through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int o = 32; o >= QN; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < QN && cvalid) {
        double sm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int j = 0; j < 8; ++j) sm[j] += (double)red[w * 64 + threadIdx.x][j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p.stats[((long)blockIdx.x * 2 + 0) * N + c + j] = sm[j];
            p.stats[((long)blockIdx.x * 2 + 1) * N + c + j] = sm[4 + j];
            for (long slot = blockIdx.x + gridDim.x; slot < p.stat_slots; slot += gridDim.x) {
                p.stats[(slot * 2 + 0) * N + c + j] = 0.0;
                p.stats[(slot * 2 + 1) * N + c + j] = 0.0;
            }
        }
    }
}

// compute units of the current device (256 on an MI355X; fewer in a partitioned mode), asked once per device
inline int cu_count() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n : 256;
    }
    return cached[dev];
}

// Output tile of the LDS-DMA wide GEMM (round 6): 128 x 128, or - when that leaves at most half / a quarter of the CUs with a tile -
// 64 x 128 / 64 x 64, so that the deep levels' launches put twice / four times the workgroups on the chip in one pass (most of
// them lose their K split and its reducer launch that way).  RL_WGEMM_TILE=128 keeps the one tile (A/B switch, bitwise the same Y).
int g_wgemm_small = -1;
inline bool wgemm_small_tiles() {
    if (g_wgemm_small < 0) {
        const char* e = getenv("RL_WGEMM_TILE");
        g_wgemm_small = (e && !strcmp(e, "128")) ? 0 : 1;
    }
    return g_wgemm_small == 1;
}
int g_wgemm_force = 0;          // rl_set_wgemm_tile("64x128" / "64x64"): that tile for every launch of the LDS-DMA kernel (measurements)
int g_gemm_no_ksplit = 0;       // rl_set_gemm_ksplit(0): tests compare kernels / tiles bit for bit on ONE summation order
struct WidePlan { int bm, bn, ksplit; };
int g_wgemm_narrow = -1;
inline bool wgemm_narrow_dma() {
    if (g_wgemm_narrow < 0) {
        const char* e = getenv("RL_WGEMM_NARROW");
        g_wgemm_narrow = (e && !strcmp(e, "0")) ? 0 : 1;
    }
    return g_wgemm_narrow == 1 && wgemm_small_tiles();
}
// dma: the launch will run wgemm2_kernel (the only kernel with the small tiles)
inline WidePlan wide_plan(long M, int N, int K, bool dma) {
    WidePlan w{128, 128, 1};
    long tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int cus = cu_count();
    if (dma && g_wgemm_force && N > 64) {           // diagnostics: one small tile everywhere (tools/wgemm_tile_bench.py)
        w.bm = 64;
        w.bn = g_wgemm_force == 2 ? 64 : 128;
        tiles = ((M + 63) / 64) * ((N + w.bn - 1) / w.bn);
    } else if (dma && wgemm_small_tiles() && N > 64) {
        if (2 * tiles <= cus) {
            w.bm = 64;
            tiles = ((M + 63) / 64) * ((N + 127) / 128);
            if (2 * tiles <= cus) {
                w.bn = 64;
                tiles = ((M + 63) / 64) * ((N + 63) / 64);
            }
        }
    }
    // K splits of a launch with few output tiles (1 = none).  Never for the LDS-DMA kernel (round 6): its rings keep a workgroup
    // fed through a long K loop, and a split costs a second launch (the reducer) plus the slab round trip - measured on every
    // wide shape of config A at 1 / 2 / 4 / 8 clouds (tools/wgemm_tile_bench.py): the single pass wins or ties everywhere, by
    // up to 2x on the deep levels (e.g. 2560 x 256 x 128: 14.8 -> 9.5 us forward, 13.7 -> 7.1 us input gradient)
    // (RL_WGEMM_TILE=128 restores round 5 as a whole - one tile shape AND its K splits - for A/B runs)
    if ((dma && wgemm_small_tiles()) || g_gemm_no_ksplit || N % 4 || tiles >= 128 || K < 256) return w;
    long s = 256 / tiles;
    if (s > K / 64) s = K / 64;      // at least two 32-deep chunks per split
    if (s > 16) s = 16;
    w.ksplit = s < 2 ? 1 : (int)s;
    return w;
}
inline int gemm_ksplit(long M, int N, int K) { return wide_plan(M, N, K, false).ksplit; }      // (the upper bound over both plans)

// wide-GEMM arithmetic (see pgemm_kernel): 0 = fp32 MFMA, 3 = bf16x3 (default), 1 = bf16.  Initial value from
// RL_WIDE_GEMM = fp32 | bf16x3 | bf16; rl_set_wide_gemm() changes it at run time.
int g_wide_terms = -1;
inline int parse_wide(const char* e) {
    if (!e || !strcmp(e, "bf16x3")) return 3;
    if (!strcmp(e, "fp32")) return 0;
    if (!strcmp(e, "bf16")) return 1;
    return -1;
}
inline int wide_gemm_terms() {
    if (g_wide_terms < 0) {
        const int t = parse_wide(getenv("RL_WIDE_GEMM"));
        g_wide_terms = t < 0 ? 3 : t;
    }
    return g_wide_terms;
}
template <int NT>
void launch_pgemm(dim3 logical, hipStream_t st, GemmParams p) {
    p.gx = (int)logical.x; p.ny = (int)logical.y;
    const dim3 grid(p.ny > 1 ? (unsigned)(8 * rl_cdiv(p.gx, 8) * p.ny) : logical.x, 1, logical.z);
    const int t = wide_gemm_terms();
    if (t == 0)      hipLaunchKernelGGL((pgemm_kernel<NT, 0>), grid, dim3(256), 0, st, p);
    else if (t == 1) hipLaunchKernelGGL((pgemm_kernel<NT, 1>), grid, dim3(256), 0, st, p);
    else             hipLaunchKernelGGL((pgemm_kernel<NT, 3>), grid, dim3(256), 0, st, p);
}

// the 8-wavefront kernel: bf16 arithmetic modes, pre-split weight, K a multiple of 8 (16-byte pieces of a weight row)
inline bool wgemm_ok(const GemmParams& p) {
    return p.wsplit != nullptr && wide_gemm_terms() != 0 && (p.a.K % 8 == 0) && (((uintptr_t)p.wsplit & 15) == 0);
}
constexpr int W2_AS = 4, W2_WS = 4;
// 0: register-staged wgemm_kernel, 1 (default): LDS-DMA wgemm2_kernel - the same Y bitwise.  RL_WGEMM_STAGING / rl_set_wgemm_staging.
int g_wgemm_staging = -1;
inline int wgemm_staging() {
    if (g_wgemm_staging < 0) {
        const char* e = getenv("RL_WGEMM_STAGING");
        g_wgemm_staging = (e && !strcmp(e, "registers")) ? 0 : 1;
    }
    return g_wgemm_staging;
}
// diagnostics (rl_set_sgemm_grid_div): the streaming GEMM on 1/div of its workgroups - Y is bitwise the same, only the
// grouping of the per-lane BatchNorm partial sums changes (the regression knob of tests/test_net_gpu.py)
int g_sgemm_grid_div = 1;
// the LDS-DMA kernel takes this launch (else the register-staged wgemm_kernel)
inline bool wgemm2_usable(const GemmParams& p) {
    return wgemm_staging() == 1 && p.a.K % PG_BK == 0 && p.a.K <= W2_KMAX && (((uintptr_t)p.a.A | (uintptr_t)(p.a.lda * 4)) & 15) == 0;
}
template <int TERMS, bool STATS>
void launch_wgemm2_tile(const WidePlan& w, dim3 g2, hipStream_t st, const GemmParams& p) {
    if (w.bm == 128)     hipLaunchKernelGGL((wgemm2_kernel<TERMS, STATS, W2_AS, W2_WS, 128, 128>), g2, dim3(W2_THREADS), 0, st, p);
    else if (w.bn == 128) hipLaunchKernelGGL((wgemm2_kernel<TERMS, STATS, W2_AS, W2_WS, 64, 128>), g2, dim3(W2_THREADS), 0, st, p);
    else                 hipLaunchKernelGGL((wgemm2_kernel<TERMS, STATS, W2_AS, W2_WS, 64, 64>), g2, dim3(W2_THREADS), 0, st, p);
}
// returns the kernel function it dispatched to.  `w`: the output tile (wide_plan) when the LDS-DMA kernel runs the launch
const char* launch_wgemm(dim3 logical, hipStream_t st, GemmParams p, bool splitk = false, WidePlan w = WidePlan{128, 128, 1}) {
    p.gx = (int)logical.x; p.ny = (int)logical.y;
    const dim3 grid(p.ny > 1 ? (unsigned)(8 * rl_cdiv(p.gx, 8) * p.ny) : logical.x, 1, logical.z);
    const bool stats = p.stats != nullptr && p.ksplit <= 1;
    if (wgemm2_usable(p)) {
        // one persistent workgroup per CU: the row tiles are dealt round-robin to gx = CUs / ny workgroup rows
        p.gx = rl_cdiv(p.a.M, w.bm);
        p.ny = rl_cdiv(p.N, w.bn);
        int cap = cu_count() / (p.ny > 0 ? p.ny : 1) / 8 * 8;
        if (cap < 8) cap = 8;
        if (p.gx > cap) p.gx = cap;
        const dim3 g2(p.ny > 1 ? (unsigned)(8 * rl_cdiv(p.gx, 8) * p.ny) : (unsigned)p.gx, 1, logical.z);
        if (wide_gemm_terms() == 1) {
            if (stats) launch_wgemm2_tile<1, true>(w, g2, st, p);
            else       launch_wgemm2_tile<1, false>(w, g2, st, p);
        } else {
            if (stats) launch_wgemm2_tile<3, true>(w, g2, st, p);
            else       launch_wgemm2_tile<3, false>(w, g2, st, p);
        }
        return splitk ? "wgemm2_kernel+splitk" : "wgemm2_kernel";
    }
    if (wide_gemm_terms() == 1) {
        if (stats) hipLaunchKernelGGL((wgemm_kernel<1, true>), grid, dim3(512), 0, st, p);
        else       hipLaunchKernelGGL((wgemm_kernel<1, false>), grid, dim3(512), 0, st, p);
    } else {
        if (stats) hipLaunchKernelGGL((wgemm_kernel<3, true>), grid, dim3(512), 0, st, p);
        else       hipLaunchKernelGGL((wgemm_kernel<3, false>), grid, dim3(512), 0, st, p);
    }
    return splitk ? "wgemm_kernel+splitk" : "wgemm_kernel";
}

inline bool pgemm_ok(const GemmParams& p) {
    if (p.a.a_mode != 0 || !p.a.vec4) return false;
    if (((uintptr_t)p.W & 15) != 0) return false;
    if (p.a.lazy.scale && ((((uintptr_t)p.a.lazy.scale) | ((uintptr_t)p.a.lazy.shift)) & 15)) return false;
    if (p.w_ns == 1) return (p.N % 4 == 0) && (p.w_ks % 4 == 0);
    if (p.w_ks == 1) return (p.a.K % 4 == 0) && (p.w_ns % 4 == 0);
    return false;
}

// Pipelined wide weight-gradient: 64 (n) x 64 (k) tile of dW per workgroup, rows streamed in chunks
// of 64 through LDS with the next chunk's dwordx4 loads in flight during the 64 MFMAs per wavefront
// of the current one (same structure as pgemm_kernel; rows are the reduction index).
constexpr int PW_RB = 64;

__global__ __launch_bounds__(256) void pwgrad_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float dYs[PW_RB * WG_S];
    __shared__ __attribute__((aligned(16))) float As[PW_RB * WG_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int lr = lane & 15, lq = lane >> 4;
    const int N = p.N, K = p.a.K;
    const int n0 = blockIdx.y * WG_T, k0 = blockIdx.z * WG_T;
    const int nvalid = min(WG_T, N - n0), kvalid = min(WG_T, K - k0);
    const int nkb = (kvalid + 15) >> 4;
    const bool wave_active = wave * 16 < nvalid;
    const long r_begin = (long)blockIdx.x * p.rows_per_block;
    const long r_end = min(p.a.M, r_begin + p.rows_per_block);
    const int q4 = (tid & 15) * 4;          // this lane's column quad (same for its 4 rows)
    const bool lazy = p.a.lazy.scale != nullptr;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lazy && q4 < kvalid) {
        sc = *reinterpret_cast<const float4*>(p.a.lazy.scale + k0 + q4);
        sh = *reinterpret_cast<const float4*>(p.a.lazy.shift + k0 + q4);
    }

    f32x4 acc[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) acc[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    float4 rd[4], ra[4];

    auto fetch = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long R = r0 + (tid >> 4) + 16 * i;
            rd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (R < r_end) {
                if (q4 < nvalid) {
                    long off;
                    if (p.dy_contig) off = R * p.lddy;
                    else {
                        const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                        const int ii = (int)(R - (long)b * p.rows_per_batch);
                        off = ((long)b * p.dy_bstride + ii) * p.lddy;
                    }
                    rd[i] = *reinterpret_cast<const float4*>(p.dY + off + n0 + q4);
                }
                if (q4 < kvalid) {
                    float4 v = *reinterpret_cast<const float4*>(p.a.A + a_row_offset(p.a, R) + k0 + q4);
                    if (lazy) {
                        v.x = rl_act(v.x * sc.x + sh.x, p.a.lazy.act, p.a.lazy.slope);
                        v.y = rl_act(v.y * sc.y + sh.y, p.a.lazy.act, p.a.lazy.slope);
                        v.z = rl_act(v.z * sc.z + sh.z, p.a.lazy.act, p.a.lazy.slope);
                        v.w = rl_act(v.w * sc.w + sh.w, p.a.lazy.act, p.a.lazy.slope);
                    }
                    ra[i] = v;
                }
            }
        }
    };

    if (r_begin < r_end) fetch(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += PW_RB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i;
            *reinterpret_cast<float4*>(dYs + r * WG_S + q4) = rd[i];
            *reinterpret_cast<float4*>(As + r * WG_S + q4) = ra[i];
        }
        __syncthreads();
        if (r0 + PW_RB < r_end) fetch(r0 + PW_RB);
        if (p.has_bias && blockIdx.z == 0 && tid < WG_T) {
#pragma unroll 16
            for (int r = 0; r < PW_RB; ++r) bsum += dYs[r * WG_S + tid];
        }
        if (wave_active) {
#pragma unroll
            for (int rs = 0; rs < PW_RB / 4; ++rs) {
                const int rr = rs * 4 + lq;
                const float av = dYs[rr * WG_S + wave * 16 + lr];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb < nkb) {
                        const float bv = As[rr * WG_S + kb * 16 + lr];
                        acc[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[kb], 0, 0, 0);
                    }
                }
            }
        }
    }
    float* out = p.slab + (long)blockIdx.x * ((long)N * K + N);
    if (wave_active) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb < nkb) {
                const int k = k0 + kb * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wave * 16 + lq * 4 + r;
                    if (n < N && k < K) out[(long)n * K + k] = acc[kb][r];
                }
            }
        }
    }
    if (p.has_bias && blockIdx.z == 0 && tid < nvalid) out[(long)N * K + n0 + tid] = bsum;
}

// Wide weight gradient with a 128 (n) x 128 (k) tile of dW per workgroup (both N and K >= 128).  It is the
// pgemm_kernel loop with the row as the reduction index: dW[n][k] = sum_r dY[r][n] * A'[r][k].  Chunks of 32
// rows are staged TRANSPOSED in LDS ([n][row] and [k][row], stride 36), so that a lane's eight reduction
// values per operand tile (rows 8*(lane>>4) .. +7) are two ds_read_b128; wavefront w owns n-blocks 2w, 2w+1.
// Staging: lane (q = l&3, rl = l>>2) of unit u (16 columns x 16 rows) loads 4 columns of one row (four lanes
// cover 64 contiguous bytes) and writes them as 4 scalars whose banks 16*q + rl + const are all distinct.
constexpr int PW2_RB = 32;
constexpr int PW2_T = 128;
constexpr int PW2_S = 36;

template <int TERMS>   // 0: fp32 MFMA; 3: bf16 head+tail operands, three bf16 MFMAs per product (see pgemm_kernel); 1: bf16
__global__ __launch_bounds__(256, 2) void pwgrad128_kernel(const WgradParams p) {
    constexpr int BS = 40;   // bf16 row stride: 32 reduction rows + 8 pad (80 B)
    constexpr int NSPL = TERMS == 3 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char lds_d[TERMS == 0 ? PW2_T * PW2_S * 4 : PW2_T * BS * 2 * NSPL];
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[TERMS == 0 ? PW2_T * PW2_S * 4 : PW2_T * BS * 2 * NSPL];
    float* dYt = reinterpret_cast<float*>(lds_d);
    float* At = reinterpret_cast<float*>(lds_a);
    __bf16* Dh = reinterpret_cast<__bf16*>(lds_d);
    __bf16* Dl = Dh + PW2_T * BS;
    __bf16* Xh = reinterpret_cast<__bf16*>(lds_a);
    __bf16* Xl = Xh + PW2_T * BS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int N = p.N, K = p.a.K;
    const int n0 = blockIdx.y * PW2_T, k0 = blockIdx.z * PW2_T;
    const int nvalid = min(PW2_T, N - n0), kvalid = min(PW2_T, K - k0);
    const long r_begin = (long)blockIdx.x * p.rows_per_block;
    const long r_end = min(p.a.M, r_begin + p.rows_per_block);
    const bool lazy = p.a.lazy.scale != nullptr;
    const bool relu = p.a.lazy.act == RL_ACT_RELU;
    const float nslope = p.a.lazy.act == RL_ACT_NONE ? 1.f : p.a.lazy.slope;
    auto actf = [&](float z) {
        const float neg = relu ? 0.f : z * nslope;
        return z > 0.f ? z : neg;
    };
    // staging units of this lane: u = i*4 + wave -> columns (u%8)*16 + 4*(l&3), rows (u/8)*16 + (l>>2)
    int ucol[4], urow[4];
    float4 sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = i * 4 + wave;
        ucol[i] = (u & 7) * 16 + (lane & 3) * 4;
        urow[i] = (u >> 3) * 16 + (lane >> 2);
        sc[i] = make_float4(1.f, 1.f, 1.f, 1.f);
        sh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lazy && ucol[i] < kvalid) {
            sc[i] = *reinterpret_cast<const float4*>(p.a.lazy.scale + k0 + ucol[i]);
            sh[i] = *reinterpret_cast<const float4*>(p.a.lazy.shift + k0 + ucol[i]);
        }
    }
    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) acc[i][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    float4 rd[4], ra[4];

    auto fetch = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long R = r0 + urow[i];
            rd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (R < r_end) {
                if (ucol[i] < nvalid) {
                    long off;
                    if (p.dy_contig) off = R * p.lddy;
                    else {
                        const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                        const int ii = (int)(R - (long)b * p.rows_per_batch);
                        off = ((long)b * p.dy_bstride + ii) * p.lddy;
                    }
                    rd[i] = *reinterpret_cast<const float4*>(p.dY + off + n0 + ucol[i]);
                }
                if (ucol[i] < kvalid) ra[i] = *reinterpret_cast<const float4*>(p.a.A + a_row_offset(p.a, R) + k0 + ucol[i]);
            }
        }
    };
    auto commit = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = ra[i];
            if (lazy && r0 + urow[i] < r_end && ucol[i] < kvalid) {
                v.x = actf(v.x * sc[i].x + sh[i].x);
                v.y = actf(v.y * sc[i].y + sh[i].y);
                v.z = actf(v.z * sc[i].z + sh[i].z);
                v.w = actf(v.w * sc[i].w + sh[i].w);
            }
            if constexpr (TERMS == 0) {
                float* dd = dYt + ucol[i] * PW2_S + urow[i];
                dd[0] = rd[i].x; dd[PW2_S] = rd[i].y; dd[2 * PW2_S] = rd[i].z; dd[3 * PW2_S] = rd[i].w;
                float* da = At + ucol[i] * PW2_S + urow[i];
                da[0] = v.x; da[PW2_S] = v.y; da[2 * PW2_S] = v.z; da[3 * PW2_S] = v.w;
            } else {
                bf16x4 dh, dl, xh, xl;
                split_bf16(rd[i], dh, dl);
                split_bf16(v, xh, xl);
                const int o = ucol[i] * BS + urow[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Dh[o + j * BS] = dh[j];
                    Xh[o + j * BS] = xh[j];
                    if constexpr (TERMS == 3) {
                        Dl[o + j * BS] = dl[j];
                        Xl[o + j * BS] = xl[j];
                    }
                }
            }
        }
    };

    const float* n_frag = dYt + (wave * 32 + lr) * PW2_S + lq * 8;
    const float* k_frag = At + lr * PW2_S + lq * 8;
    if (r_begin < r_end) fetch(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += PW2_RB) {
        __syncthreads();
        commit(r0);
        __syncthreads();
        if (r0 + PW2_RB < r_end) fetch(r0 + PW2_RB);
        if (p.has_bias && blockIdx.z == 0 && tid < PW2_T) {
            if constexpr (TERMS == 0) {
#pragma unroll
                for (int j = 0; j < PW2_RB / 4; ++j) {
                    const float4 t = *reinterpret_cast<const float4*>(dYt + tid * PW2_S + j * 4);
                    bsum += (t.x + t.y) + (t.z + t.w);
                }
            } else {
#pragma unroll
                for (int j = 0; j < PW2_RB / 8; ++j) {
                    const bf16x8 h = *reinterpret_cast<const bf16x8*>(Dh + tid * BS + j * 8);
                    float t = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t += (float)h[e];
                    if constexpr (TERMS == 3) {
                        const bf16x8 l = *reinterpret_cast<const bf16x8*>(Dl + tid * BS + j * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) t += (float)l[e];
                    }
                    bsum += t;
                }
            }
        }
        if constexpr (TERMS == 0) {
            float4 af[2][2], bf[2][2];
    #pragma unroll
            for (int h = 0; h < 2; ++h) {
                af[h][0] = *reinterpret_cast<const float4*>(n_frag + h * 4);
                af[h][1] = *reinterpret_cast<const float4*>(n_frag + 16 * PW2_S + h * 4);
            }
    #pragma unroll
            for (int j = 0; j < 2; ++j) bf[0][j] = *reinterpret_cast<const float4*>(k_frag + j * 16 * PW2_S);
    #pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int h = g / 4, kb0 = (g % 4) * 2;
                if (g + 1 < 8) {
                    const int h1 = (g + 1) / 4, kb1 = ((g + 1) % 4) * 2;
    #pragma unroll
                    for (int j = 0; j < 2; ++j)
                        bf[(g + 1) & 1][j] = *reinterpret_cast<const float4*>(k_frag + (kb1 + j) * 16 * PW2_S + h1 * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float a0[4] = {af[h][0].x, af[h][0].y, af[h][0].z, af[h][0].w};
                const float a1[4] = {af[h][1].x, af[h][1].y, af[h][1].z, af[h][1].w};
    #pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
    #pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 b4 = bf[g & 1][j];
                        const float bv = s2 == 0 ? b4.x : s2 == 1 ? b4.y : s2 == 2 ? b4.z : b4.w;
                        acc[0][kb0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s2], bv, acc[0][kb0 + j], 0, 0, 0);
                        acc[1][kb0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s2], bv, acc[1][kb0 + j], 0, 0, 0);
                    }
                }
            }
        } else {
            // one 16x16x32 MFMA covers the chunk's 32 rows; fragments: a lane's rows 8*(lane>>4) .. +7, one ds_read_b128
            const __bf16* dh_frag = Dh + (wave * 32 + lr) * BS + lq * 8;
            const __bf16* xh_frag = Xh + lr * BS + lq * 8;
            constexpr int LO = PW2_T * BS;
            bf16x8 a_h[2], a_l[2], b_h[2][2], b_l[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a_h[i] = *reinterpret_cast<const bf16x8*>(dh_frag + i * 16 * BS);
                if constexpr (TERMS == 3) a_l[i] = *reinterpret_cast<const bf16x8*>(dh_frag + LO + i * 16 * BS);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b_h[0][j] = *reinterpret_cast<const bf16x8*>(xh_frag + j * 16 * BS);
                if constexpr (TERMS == 3) b_l[0][j] = *reinterpret_cast<const bf16x8*>(xh_frag + LO + j * 16 * BS);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kb0 = g * 2;
                if (g + 1 < 4) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        b_h[(g + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(xh_frag + (kb0 + 2 + j) * 16 * BS);
                        if constexpr (TERMS == 3)
                            b_l[(g + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(xh_frag + LO + (kb0 + 2 + j) * 16 * BS);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][kb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h[i], b_h[g & 1][j], acc[i][kb0 + j], 0, 0, 0);
                if constexpr (TERMS == 3) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[i][kb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h[i], b_l[g & 1][j], acc[i][kb0 + j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[i][kb0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l[i], b_h[g & 1][j], acc[i][kb0 + j], 0, 0, 0);
                }
            }
        }
    }
    float* out = p.slab + (long)blockIdx.x * ((long)N * K + N);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const int k = k0 + kb * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + (2 * wave + i) * 16 + lq * 4 + r;
                if (n < N && k < K) out[(long)n * K + k] = acc[i][kb][r];
            }
        }
    }
    if (p.has_bias && blockIdx.z == 0 && tid < nvalid) out[(long)N * K + n0 + tid] = bsum;
}

// The same weight-gradient tile cut for memory latency like wgemm_kernel: eight wavefronts of 16 (n) x 128 (k) share the
// 128 x 128 tile of dW (32 accumulator registers each instead of 64), which fits 128 VGPRs: two workgroups per CU = four
// wavefronts per SIMD, twice the loads in flight (pwgrad128_kernel<3>: 240 VGPRs, two per SIMD, 39 % of its wavefront
// cycles parked on memory).  bf16 arithmetic modes only; same operands and MFMA sequence per accumulator -> same bits.
// RB: A and dY are stored as bf16 rows (bf16-storage mode; then TERMS = 1 is exact - the tails would be zero)
typedef __attribute__((ext_vector_type(4))) short s16x4g;
template <int TERMS, bool RB = false>   // 3: bf16x3; 1: bf16
__device__ __forceinline__ void pwgrad128w_body(const WgradParams& p, const int bx, const int by, const int bz) {
    // bf16 planes ROW-MAJOR: [32 rows][RP], RP = 128 + 16 (288 B: the four rows of a transposing read fall on disjoint bank
    // groups).  A lane's four converted columns of a row leave as ONE ds_write_b64 per plane - 8 LDS writes per lane and chunk
    // where the column-major image took 32 ds_write_b16 - and the MFMA fragments (8 consecutive ROWS of one column per lane)
    // come back through ds_read_b64_tr_b16, four rows per read.
    constexpr int RP = 144;
    constexpr int NSPL = TERMS == 3 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) __bf16 lds_d[PW2_RB * RP * NSPL];
    __shared__ __attribute__((aligned(16))) __bf16 lds_a[PW2_RB * RP * NSPL];
    __bf16* Dh = lds_d;
    __bf16* Dl = Dh + PW2_RB * RP;
    __bf16* Xh = lds_a;
    __bf16* Xl = Xh + PW2_RB * RP;
    // the lazy BatchNorm scale / shift of this workgroup's 128 k-columns: 1 KB of LDS read in the staging step instead of
    // 16 registers per lane (the kernel sat 2 dwords over its 128-VGPR budget, i.e. in scratch)
    __shared__ __attribute__((aligned(16))) float lz[2][PW2_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int N = p.N, K = p.a.K;
    const int n0 = by * PW2_T, k0 = bz * PW2_T;
    const int nvalid = min(PW2_T, N - n0), kvalid = min(PW2_T, K - k0);
    const long r_begin = (long)bx * p.rows_per_block;
    const long r_end = min(p.a.M, r_begin + p.rows_per_block);
    const bool lazy = p.a.lazy.scale != nullptr;
    const bool relu = p.a.lazy.act == RL_ACT_RELU;
    const float nslope = p.a.lazy.act == RL_ACT_NONE ? 1.f : p.a.lazy.slope;
    auto actf = [&](float z) {
        const float neg = relu ? 0.f : z * nslope;
        return z > 0.f ? z : neg;
    };
    // staging units of this lane: u = i*8 + wave -> columns (u%8)*16 + 4*(l&3), rows (u/8)*16 + (l>>2)
    int ucol[2], urow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = i * 8 + wave;
        ucol[i] = (u & 7) * 16 + (lane & 3) * 4;
        urow[i] = (u >> 3) * 16 + (lane >> 2);
    }
    if (tid < PW2_T) {
        const bool in = lazy && tid < kvalid;
        lz[0][tid] = in ? p.a.lazy.scale[k0 + tid] : 1.f;
        lz[1][tid] = in ? p.a.lazy.shift[k0 + tid] : 0.f;
    }
    // (the first barrier of the row loop makes lz visible before its first use in commit)
    // wavefront (nh, kw) owns n columns nh*64 .. +63 x k columns kw*32 .. +31 of the dW tile: 12 fragment reads per chunk (8 of dY,
    // 4 of A) instead of the 18 of a 16 x 128 strip (2 + 16) - the LDS pipe is this kernel's bound (profiles/r04_pmc_sq_step.md:
    // 5.8 % of the wavefront cycles x 16 wavefronts per CU; halving its L2 misses changed nothing)
    const int nh = wave >> 2, kw = wave & 3;
    f32x4 acc[4][2];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc[nb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // db = column sums of dY: every lane adds the fp32 values it stages (its four columns, its rows of every chunk) - 8 additions per
    // chunk; the 16 lanes of a wavefront that share the columns are combined once, at the end.  (The sums used to be rebuilt from
    // the bf16 heads and tails of the planes: 32 conversions + 36 additions + 16 cross-lane reads per chunk in two of the eight
    // wavefronts - the two every barrier then waited for.  SQ counters: 188 vector instructions per wavefront and chunk.)
    const bool want_bias = p.has_bias && bz == 0;
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    float4 rd[2], ra[2];

    auto fetch = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long R = r0 + urow[i];
            rd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (R < r_end) {
                if (ucol[i] < nvalid) {
                    long off;
                    if (p.dy_contig) off = R * p.lddy;
                    else {
                        const int b = (int)((unsigned)R / (unsigned)p.rows_per_batch);
                        const int ii = (int)(R - (long)b * p.rows_per_batch);
                        off = ((long)b * p.dy_bstride + ii) * p.lddy;
                    }
                    rd[i] = rl_ldx4<RB>(p.dY, off + n0 + ucol[i]);
                }
                if (ucol[i] < kvalid) ra[i] = rl_ldx4<RB>(p.a.A, a_row_offset(p.a, R) + k0 + ucol[i]);
            }
        }
    };
    auto commit = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = ra[i];
            if (lazy && r0 + urow[i] < r_end && ucol[i] < kvalid) {
                const float4 sc = *reinterpret_cast<const float4*>(&lz[0][ucol[i]]);
                const float4 sh = *reinterpret_cast<const float4*>(&lz[1][ucol[i]]);
                v.x = actf(v.x * sc.x + sh.x);
                v.y = actf(v.y * sc.y + sh.y);
                v.z = actf(v.z * sc.z + sh.z);
                v.w = actf(v.w * sc.w + sh.w);
            }
            bf16x4 dh, dl, xh, xl;
            if (want_bias) bs += (f32x4){rd[i].x, rd[i].y, rd[i].z, rd[i].w};
            split_bf16(rd[i], dh, dl);
            split_bf16(v, xh, xl);
            const int o = urow[i] * RP + ucol[i];
            *reinterpret_cast<bf16x4*>(Dh + o) = dh;
            *reinterpret_cast<bf16x4*>(Xh + o) = xh;
            if constexpr (TERMS == 3) {
                *reinterpret_cast<bf16x4*>(Dl + o) = dl;
                *reinterpret_cast<bf16x4*>(Xl + o) = xl;
            }
        }
    };

    // transposing read: lane 4q + p of a 16-lane group gives the address of row q, columns 4p .. 4p+3 of a 4-row x 16-column
    // block and receives column (its index in the group) of the four rows
    const int tr_off = (lq * 8 + (lr >> 2)) * RP + 4 * (lr & 3);
    const __bf16* dh_frag = Dh + tr_off + nh * 64;
    const __bf16* xh_frag = Xh + tr_off + kw * 32;
    constexpr int LO = PW2_RB * RP;
    auto frag = [&](const __bf16* q) {      // rows 8*lq .. 8*lq+7 of column (block start + lr)
        const bf16x4 r0 = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g*)q));
        const bf16x4 r1 = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g*)(q + 4 * RP)));
        return __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    if (r_begin < r_end) fetch(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += PW2_RB) {
        __syncthreads();
        commit(r0);
        __syncthreads();
        if (r0 + PW2_RB < r_end) fetch(r0 + PW2_RB);
        bf16x8 b_h[2], b_l[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            b_h[kb] = frag(xh_frag + kb * 16);
            if constexpr (TERMS == 3) b_l[kb] = frag(xh_frag + LO + kb * 16);
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            bf16x8 a_h, a_l;
            a_h = frag(dh_frag + nb * 16);
            if constexpr (TERMS == 3) a_l = frag(dh_frag + LO + nb * 16);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                acc[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h[kb], acc[nb][kb], 0, 0, 0);
            if constexpr (TERMS == 3) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    acc[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l[kb], acc[nb][kb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    acc[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h[kb], acc[nb][kb], 0, 0, 0);
            }
        }
    }
    float* out = p.slab + (long)bx * ((long)N * K + N);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int k = k0 + kw * 32 + kb * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + nh * 64 + nb * 16 + lq * 4 + r;
                if (n < N && k < K) out[(long)n * K + k] = acc[nb][kb][r];
            }
        }
    }
    if (want_bias) {
        // lanes l, l + 4, ... , l + 60 staged the same four columns (rows l >> 2 and 16 + (l >> 2) of every chunk)
#pragma unroll
        for (int o = 4; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) bs[j] += __shfl_xor(bs[j], o, 64);
        if (lane < 4 && ucol[0] < nvalid) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(long)N * K + n0 + ucol[0] + j] = bs[j];
        }
    }
}

template <int TERMS, bool RB = false>
__global__ __launch_bounds__(512, 4) void pwgrad128w_kernel(const WgradParams p) {
    pwgrad128w_body<TERMS, RB>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- the same kernel for SEVERAL layers in one launch (rl_wgrad_batch) -----------------------------------------------
// The wide weight gradients of a backward pass are independent of each other and of the dY -> dX chain (only the optimiser
// needs them), and most of them are small: 80 - 216 workgroups on a chip that holds 512 of these.  Launched one by one each
// leaves most CUs idle for 10 - 30 us; launched together (workgroup -> (layer, tile) through the prefix table in the
// kernel arguments) they fill the chip.  Same per-workgroup arithmetic, same slabs: bitwise the results of rl_wgrad.
struct WBItem {
    const float* A; const float* dY; const float* sc; const float* sh; float* slab;
    long lda, a_bstride, lddy, dy_bstride, rows_per_block, M;
    int N, K, n, rows_per_batch, act;
    float slope;
    int first_block;           // of this layer in the launch
    int nsplit;                // its row blocks (slabs)
    unsigned short gy, gz;     // its column-tile / k-tile counts
    unsigned char a_contig, dy_contig, has_bias, pad;
};
constexpr int WB_MAX = 24;     // 24 x 136 B + 8 B of header: well inside the 4 KB of kernel arguments
struct WgradBatch {
    WBItem item[WB_MAX];
    int count, pad;
};
template <int TERMS>
__global__ __launch_bounds__(512, 4) void pwgrad128w_batch_kernel(const WgradBatch b) {
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.item[i + 1].first_block) ++i;
    const WBItem& it = b.item[i];
    const int local = (int)blockIdx.x - it.first_block;
    const int per = (int)it.gy * (int)it.gz;
    // Workgroup -> (row block, tile).  A layer wider than one 128 x 128 tile reads each row block once per tile of the other
    // operand's direction (81920 x 256 x 256: both operands twice); workgroups go to the eight XCDs round-robin (id % 8), each
    // with an L2 of its own, so the tiles of ONE row block are given ids 8 apart: same XCD, dispatched back to back, streaming
    // the same rows at the same time - the second reader of a row finds it in that L2.  (Layer starts are multiples of 8;
    // ids past the layer's last row block have nothing to do.)
    int bx, t = 0;
    if (per > 1) {
        const int grp = local / (8 * per), rem = local - grp * 8 * per;
        t = rem >> 3;
        bx = grp * 8 + (rem & 7);
    } else bx = local;
    if (bx >= it.nsplit) return;
    const int by = t / (int)it.gz, bz = t - by * (int)it.gz;
    WgradParams p;
    p.a.A = it.A; p.a.lda = it.lda; p.a.a_bstride = it.a_bstride; p.a.a_mode = 0;
    p.a.lazy.scale = it.sc; p.a.lazy.shift = it.sh; p.a.lazy.act = it.act; p.a.lazy.slope = it.slope;
    p.a.xyz = nullptr; p.a.xyz_bstride = 0; p.a.nbr_idx = nullptr; p.a.nbr_d2 = nullptr; p.a.nbr_k = 0;
    p.a.n = it.n; p.a.K = it.K; p.a.M = it.M; p.a.contig = it.a_contig; p.a.vec4 = 1; p.a.vec4p = 1;
    p.N = it.N; p.dY = it.dY; p.lddy = it.lddy; p.dy_bstride = it.dy_bstride; p.rows_per_batch = it.rows_per_batch;
    p.dy_contig = it.dy_contig; p.slab = it.slab; p.rows_per_block = it.rows_per_block; p.has_bias = it.has_bias;
    pwgrad128w_body<TERMS, false>(p, bx, by, bz);
}

// The narrow (streaming) weight gradients of a backward pass in one launch as well: sixteen launches of 5 - 46 us, most
// of them one or two workgroups per CU, become one.  gy = 4 * log2(KT) + log2(NT) names the layer's (KT, NT) body; a
// workgroup runs exactly the arithmetic of its swgrad_kernel<KT, NT> launch: bitwise the same slabs.
__global__ __launch_bounds__(256) void swgrad_batch_kernel(const WgradBatch b) {
    __shared__ float red[4 * 4 * 4 * 64 + 8 * 64];      // the largest body's (KT * NT <= 16, NT <= 8)
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.item[i + 1].first_block) ++i;
    const WBItem& it = b.item[i];
    WgradParams p;
    p.a.A = it.A; p.a.lda = it.lda; p.a.a_bstride = it.a_bstride; p.a.a_mode = 0;
    p.a.lazy.scale = it.sc; p.a.lazy.shift = it.sh; p.a.lazy.act = it.act; p.a.lazy.slope = it.slope;
    p.a.xyz = nullptr; p.a.xyz_bstride = 0; p.a.nbr_idx = nullptr; p.a.nbr_d2 = nullptr; p.a.nbr_k = 0;
    p.a.n = it.n; p.a.K = it.K; p.a.M = it.M; p.a.contig = it.a_contig; p.a.vec4 = 0; p.a.vec4p = 0;
    p.N = it.N; p.dY = it.dY; p.lddy = it.lddy; p.dy_bstride = it.dy_bstride; p.rows_per_batch = it.rows_per_batch;
    p.dy_contig = it.dy_contig; p.slab = it.slab; p.rows_per_block = it.rows_per_block; p.has_bias = it.has_bias;
    const int bx = (int)blockIdx.x - it.first_block;
    if (bx >= it.nsplit) return;
    switch (it.gy) {
        case 0: swgrad_body<1, 1>(p, bx, red); break;
        case 1: swgrad_body<1, 2>(p, bx, red); break;
        case 2: swgrad_body<1, 4>(p, bx, red); break;
        case 3: swgrad_body<1, 8>(p, bx, red); break;
        case 4: swgrad_body<2, 1>(p, bx, red); break;
        case 5: swgrad_body<2, 2>(p, bx, red); break;
        case 6: swgrad_body<2, 4>(p, bx, red); break;
        case 8: swgrad_body<4, 1>(p, bx, red); break;
        case 9: swgrad_body<4, 2>(p, bx, red); break;
        default: swgrad_body<4, 4>(p, bx, red); break;
    }
}

inline bool pwgrad_ok(const WgradParams& p) {
    if (p.a.a_mode != 0 || !p.a.vec4) return false;
    if ((p.N % 4) || (p.lddy % 4) || (((uintptr_t)p.dY) & 15)) return false;
    if (p.a.lazy.scale && ((((uintptr_t)p.a.lazy.scale) | ((uintptr_t)p.a.lazy.shift)) & 15)) return false;
    return true;
}

}  // namespace

int rl_wide_terms() { return wide_gemm_terms(); }

extern "C" int rl_set_wide_gemm(const char* mode) {
    const int t = parse_wide(mode);
    RL_REQUIRE(mode != nullptr && t >= 0, RL_ERR_ARGS, "rl_set_wide_gemm: mode must be \"fp32\", \"bf16x3\" or \"bf16\"");
    g_wide_terms = t;
    return RL_OK;
}
extern "C" int rl_set_wgemm_staging(const char* how) {
    RL_REQUIRE(how != nullptr && (!strcmp(how, "registers") || !strcmp(how, "dma")), RL_ERR_ARGS,
               "rl_set_wgemm_staging: \"registers\" or \"dma\"");
    g_wgemm_staging = !strcmp(how, "dma");
    return RL_OK;
}
extern "C" int rl_set_wgemm_tile(const char* how) {
    RL_REQUIRE(how && (!strcmp(how, "auto") || !strcmp(how, "128") || !strcmp(how, "64x128") || !strcmp(how, "64x64")), RL_ERR_ARGS,
               "rl_set_wgemm_tile: expected auto | 128 | 64x128 | 64x64");
    g_wgemm_force = !strcmp(how, "64x128") ? 1 : !strcmp(how, "64x64") ? 2 : 0;
    g_wgemm_small = strcmp(how, "128") ? 1 : 0;
    return RL_OK;
}
extern "C" int rl_set_gemm_ksplit(int enable) {
    g_gemm_no_ksplit = enable ? 0 : 1;
    return RL_OK;
}
extern "C" int rl_set_sgemm_grid_div(int div) {
    RL_REQUIRE(div >= 1 && div <= 64, RL_ERR_ARGS, "rl_set_sgemm_grid_div: 1 .. 64");
    g_sgemm_grid_div = div;
    return RL_OK;
}
extern "C" const char* rl_get_wide_gemm(void) {
    const int t = wide_gemm_terms();
    return t == 0 ? "fp32" : t == 1 ? "bf16" : "bf16x3";
}

extern "C" int rl_gemm(const rl_gemm_desc* d, void* stream) {
    RL_REQUIRE(d != nullptr, RL_ERR_ARGS, "rl_gemm: null descriptor");
    GemmParams p{};
    int rc = fill_a(&p.a, "rl_gemm", d->A, d->lda, d->a_bstride, d->a_mode, d->in_act, d->in_slope,
                    d->in_scale, d->in_shift, d->xyz, d->xyz_bstride, d->nbr_idx, d->nbr_d2, d->nbr_k,
                    d->B, d->n, d->K);
    if (rc) return rc;
    RL_REQUIRE(d->N > 0 && d->W && d->Y && d->ldy > 0, RL_ERR_ARGS, "rl_gemm: bad W/Y");
    p.N = d->N; p.W = d->W; p.w_ks = d->w_ks; p.w_ns = d->w_ns; p.bias = d->bias;
    p.Y = d->Y; p.ldy = d->ldy; p.y_bstride = d->y_bstride;
    p.rows_per_batch = (d->a_mode == 1) ? d->n * d->nbr_k : d->n;
    RL_REQUIRE(d->y_bstride >= p.rows_per_batch, RL_ERR_ARGS, "rl_gemm: y_bstride smaller than rows per cloud");
    p.y_contig = (d->y_bstride == p.rows_per_batch);
    p.accumulate = d->accumulate; p.stats = d->stats;
    p.piv_mean = d->stats ? d->stats_pivot_mean : nullptr; p.piv_bias = d->stats ? d->stats_pivot_bias : nullptr;
    p.addend = d->addend; p.out2 = d->out2; p.out2_index = d->out2_index; p.out2_bstride = d->out2_bstride;
    RL_REQUIRE(!d->out2_index || rl_float_atomics_allowed(), RL_ERR_UNSUPPORTED,
               "rl_gemm: out2_index scatters with fp32 atomics (order-dependent); the deterministic path is a dense out2 + "
               "rl_segment_sum_rows (set RL_ALLOW_FLOAT_ATOMICS=1 to use it anyway)");
    p.split_col = d->out2 ? d->split_col : d->N;
    p.wsplit = reinterpret_cast<const __bf16*>(d->W_split);
    const bool split = d->addend != nullptr || d->out2 != nullptr;
    if (split) {
        RL_REQUIRE(!d->out2 || (d->split_col > 0 && d->split_col < d->N && (!d->out2_index || d->out2_bstride > 0)), RL_ERR_ARGS,
                   "rl_gemm: bad split-scatter fields");
        RL_REQUIRE(d->ldy >= (d->out2 ? d->split_col : d->N), RL_ERR_ARGS, "rl_gemm: ldy smaller than the columns stored to Y");
        RL_REQUIRE(d->stats == nullptr, RL_ERR_UNSUPPORTED, "rl_gemm: split-scatter epilogue has no statistics");
    } else {
        RL_REQUIRE(d->ldy >= d->N, RL_ERR_ARGS, "rl_gemm: bad W/Y");
    }
    const int gx = rl_row_blocks_host(p.a.M, GM_BM);
    hipStream_t st = (hipStream_t)stream;
    // the slots the caller's finalize reads (rl_gemm_stat_slots): every kernel below zero-fills those beyond its own grid
    p.stat_slots = (int)rl_gemm_stat_slots(p.a.M, d->N, d->K);
    if (split) {
        RL_REQUIRE((d->K > 64 || d->N > 64), RL_ERR_UNSUPPORTED, "rl_gemm: split-scatter epilogue needs K or N > 64");
        p.ksplit = 1; p.kchunk = 0; p.kslab = nullptr;
        RL_REQUIRE(pgemm_ok(p), RL_ERR_UNSUPPORTED,
                   "rl_gemm: split-scatter epilogue needs the LDS-tiled kernel (aligned operands, K % 4 == 0)");
        if (d->N <= 16)      launch_pgemm<1>(dim3(gx, 1), st, p);
        else if (d->N <= 32) launch_pgemm<2>(dim3(gx, 1), st, p);
        else if (d->N <= 64) launch_pgemm<4>(dim3(gx, 1), st, p);
        const char* wide = nullptr;
        if (d->N <= 64) {}
        else if (wgemm_ok(p)) {
            WidePlan w = wide_plan(p.a.M, d->N, d->K, wgemm2_usable(p));
            w.ksplit = 1;
            wide = launch_wgemm(dim3(gx, rl_cdiv(d->N, 128)), st, p, false, w);
        }
        else                 launch_pgemm<8>(dim3(gx, rl_cdiv(d->N, 128)), st, p);
        rl_note_kernel(wide ? wide : d->N <= 64 ? "pgemm_kernel" : "pgemm_kernel<8>");
        RL_LAUNCH_CHECK("rl_gemm(split-scatter)");
        return RL_OK;
    }
    const bool streams = d->a_mode == 0 && p.a.vec4p && ((d->K <= 64 && d->N <= 64) || (d->K <= 16 && d->N <= 128));
    if (d->bnb_Y) {
        RL_REQUIRE(streams && d->N <= 64 && !split, RL_ERR_UNSUPPORTED,
                   "rl_gemm: the BatchNorm-backward sums are a by-product of the streaming kernel only (K, N <= 64; ask rl_gemm_streams)");
        RL_REQUIRE(d->stats && d->bnb_scale && d->bnb_shift && d->bnb_mean && d->bnb_invstd && !d->stats_pivot_mean, RL_ERR_ARGS,
                   "rl_gemm: bnb_Y needs stats, the layer's scale / shift / mean / invstd and no pivot");
        p.bnb_Y = d->bnb_Y; p.bnb_scale = d->bnb_scale; p.bnb_shift = d->bnb_shift; p.bnb_mean = d->bnb_mean; p.bnb_invstd = d->bnb_invstd;
        p.bnb_neg = d->bnb_act == RL_ACT_RELU ? 0.f : (d->bnb_act == RL_ACT_LRELU ? d->bnb_slope : 1.f);
        p.piv_mean = p.piv_bias = nullptr;
    }
    if (streams) {
        // every wavefront first loads the whole weight matrix into registers: with >= 2048 weights per
        // wavefront, fewer and longer-lived workgroups (two per CU) beat one 128-row tile per workgroup
        int sg = gx;
        if ((long)d->K * d->N >= 2048) sg = gx > 512 ? 512 : (gx > 256 ? 256 : gx);
        if (g_sgemm_grid_div > 1) sg = sg / g_sgemm_grid_div > 0 ? sg / g_sgemm_grid_div : 1;
        if (d->K <= 16)      launch_sgemm<1>(d->N, sg, st, p);
        else if (d->K <= 32) launch_sgemm<2>(d->N, sg, st, p);
        else                 launch_sgemm<4>(d->N, sg, st, p);
        rl_note_kernel("sgemm_kernel");
        RL_LAUNCH_CHECK("rl_gemm(stream)");
        return RL_OK;
    }
    p.ksplit = 1; p.kchunk = 0; p.kslab = nullptr;
    WidePlan plan = wide_plan(p.a.M, d->N, d->K, pgemm_ok(p) && d->N > 64 && wgemm_ok(p) && wgemm2_usable(p));
    if (pgemm_ok(p) && d->N > 64 && d->kslab != nullptr) {
        const int ks = plan.ksplit;
        if (ks > 1 && d->kslab_floats >= (int64_t)ks * p.a.M * d->N) {
            p.ksplit = ks;
            p.kchunk = ((d->K + ks - 1) / ks + 31) / 32 * 32;
            p.ksplit = (d->K + p.kchunk - 1) / p.kchunk;
            p.kslab = d->kslab;
            const char* wide = nullptr;
            if (wgemm_ok(p)) wide = launch_wgemm(dim3(gx, rl_cdiv(d->N, 128), p.ksplit), st, p, true, plan);
            else launch_pgemm<8>(dim3(gx, rl_cdiv(d->N, 128), p.ksplit), st, p);
            rl_note_kernel(wide ? wide : "pgemm_kernel<8>+splitk");
            RL_LAUNCH_CHECK("rl_gemm(split-K)");
            if ((long)gx * rl_cdiv(d->N, 64) >= 256)
                hipLaunchKernelGGL(gemm_splitk_reduce_kernel<64>, dim3(gx, rl_cdiv(d->N, 64)), dim3(256), 0, st, p);
            else
                hipLaunchKernelGGL(gemm_splitk_reduce_kernel<16>, dim3(gx, rl_cdiv(d->N, 16)), dim3(256), 0, st, p);
            RL_LAUNCH_CHECK("rl_gemm(split-K reduce)");
            return RL_OK;
        }
    }
    // (round 6) K > 64 with 16 < N <= 64 (mlp1 of the deep levels, input gradients into narrow tensors): the LDS-DMA kernel on
    // 64 x 64 tiles instead of the 4-wavefront register-staged one (RL_WGEMM_NARROW=0 keeps that); bitwise the same Y
    if (pgemm_ok(p) && d->N > 16 && d->N <= 64 && wgemm_narrow_dma() && wgemm_ok(p) && wgemm2_usable(p)) {
        const WidePlan w{64, 64, 1};
        const char* wide = launch_wgemm(dim3(gx, 1), st, p, false, w);
        rl_note_kernel(wide);
        RL_LAUNCH_CHECK("rl_gemm(dma, narrow)");
        return RL_OK;
    }
    if (pgemm_ok(p)) {
        if (d->N <= 16)      launch_pgemm<1>(dim3(gx, 1), st, p);
        else if (d->N <= 32) launch_pgemm<2>(dim3(gx, 1), st, p);
        else if (d->N <= 64) launch_pgemm<4>(dim3(gx, 1), st, p);
        const char* wide = nullptr;
        if (d->N <= 64) {}
        else if (wgemm_ok(p)) { plan.ksplit = 1; wide = launch_wgemm(dim3(gx, rl_cdiv(d->N, 128)), st, p, false, plan); }
        else                 launch_pgemm<8>(dim3(gx, rl_cdiv(d->N, 128)), st, p);
        rl_note_kernel(d->N <= 16 ? "pgemm_kernel<1>" : d->N <= 32 ? "pgemm_kernel<2>" : d->N <= 64 ? "pgemm_kernel<4>" : wide ? wide : "pgemm_kernel<8>");
        RL_LAUNCH_CHECK("rl_gemm(pipelined)");
        return RL_OK;
    }
    if (d->N <= 16)      hipLaunchKernelGGL((gemm_kernel<1>), dim3(gx, 1), dim3(256), 0, st, p);
    else if (d->N <= 32) hipLaunchKernelGGL((gemm_kernel<2>), dim3(gx, 1), dim3(256), 0, st, p);
    else if (d->N <= 64) hipLaunchKernelGGL((gemm_kernel<4>), dim3(gx, 1), dim3(256), 0, st, p);
    else                 hipLaunchKernelGGL((gemm_kernel<8>), dim3(gx, rl_cdiv(d->N, 128)), dim3(256), 0, st, p);
    rl_note_kernel("gemm_kernel");
    RL_LAUNCH_CHECK("rl_gemm");
    return RL_OK;
}

// 1 if rl_gemm runs this product on the streaming kernel (the one that can leave BatchNorm-backward sums: rl_gemm_desc.bnb_Y)
extern "C" int rl_gemm_streams(const rl_gemm_desc* d) {
    if (!d || d->a_mode != 0 || d->addend || d->out2) return 0;
    const bool vec4p = (d->lda % 4 == 0) && ((d->K + 3) / 4 * 4 <= d->lda) && (((uintptr_t)d->A & 15) == 0);
    return (vec4p && d->K <= 64 && d->N <= 64) ? 1 : 0;
}

// ---- two products over ONE A' in one launch -------------------------------------------------------------------------------------
// mlp1 and shortcut of an encoder level (modules.py:314, 325) both read the level's input: as two launches the rows are fetched
// twice and the narrow one (d/2 columns) leaves most of the chip idle.  Here the column blocks of both weights are dealt to the
// persistent workgroups of ONE wgemm2 launch (128 x 128 tiles; per accumulator the products of a single launch in the same
// order: Y1 / Y2 are bitwise what rl_gemm gives with the same planes).
static bool gemm_pair_fits(const rl_gemm_desc* a, const rl_gemm_desc* b) {
    if (!a || !b) return false;
    if (a->A != b->A || a->lda != b->lda || a->a_bstride != b->a_bstride || a->a_mode != 0 || b->a_mode != 0) return false;
    if (a->B != b->B || a->n != b->n || a->K != b->K) return false;
    if (a->in_scale != b->in_scale || a->in_shift != b->in_shift || a->in_act != b->in_act || a->in_slope != b->in_slope) return false;
    if (!a->W_split || !b->W_split || wide_gemm_terms() == 0 || wgemm_staging() != 1) return false;
    if (a->K % PG_BK || a->K > W2_KMAX || a->N <= 0 || b->N <= 0) return false;
    // a product with K <= 64 and N <= 64 runs on the streaming kernel in EXACT fp32 products (rl_gemm): pairing it would change its
    // arithmetic to bf16x3 - not done (measured: the 1029-point ragged configuration's gradients moved 4 % over their bound)
    if ((a->K <= 64 && a->N <= 64) || (b->K <= 64 && b->N <= 64)) return false;
    if (a->bias || b->bias || a->accumulate || b->accumulate || a->addend || b->addend || a->out2 || b->out2) return false;
    if ((a->stats == nullptr) != (b->stats == nullptr)) return false;
    if (a->ldy != a->N || b->ldy != b->N || a->y_bstride != a->n || b->y_bstride != b->n || !a->Y || !b->Y) return false;
    if ((((uintptr_t)a->A | (uintptr_t)(a->lda * 4) | (uintptr_t)a->W_split | (uintptr_t)b->W_split | (uintptr_t)a->Y | (uintptr_t)b->Y) & 15) != 0)
        return false;
    if (a->N % 4 || b->N % 4) return false;
    return true;
}
extern "C" int rl_gemm_pair_supported(const rl_gemm_desc* a, const rl_gemm_desc* b) { return gemm_pair_fits(a, b) ? 1 : 0; }
extern "C" int rl_gemm_pair(const rl_gemm_desc* a, const rl_gemm_desc* b, void* stream) {
    RL_REQUIRE(gemm_pair_fits(a, b), RL_ERR_UNSUPPORTED,
               "rl_gemm_pair: needs one plain A' (K %% 32 == 0, K <= 1024) for both, pre-split weights in a bf16 arithmetic mode, the "
               "dense outputs, statistics for both or none, no bias / accumulate / split epilogue");
    GemmParams p{};
    int rc = fill_a(&p.a, "rl_gemm_pair", a->A, a->lda, a->a_bstride, 0, a->in_act, a->in_slope, a->in_scale, a->in_shift,
                    nullptr, 0, nullptr, nullptr, 0, a->B, a->n, a->K);
    if (rc) return rc;
    p.N = a->N; p.W = a->W; p.w_ks = a->w_ks; p.w_ns = a->w_ns; p.bias = nullptr;
    p.Y = a->Y; p.ldy = a->ldy; p.y_bstride = a->y_bstride; p.rows_per_batch = a->n; p.y_contig = 1;
    p.accumulate = 0; p.stats = a->stats;
    p.piv_mean = a->stats ? a->stats_pivot_mean : nullptr; p.piv_bias = a->stats ? a->stats_pivot_bias : nullptr;
    p.split_col = a->N; p.wsplit = reinterpret_cast<const __bf16*>(a->W_split);
    p.ksplit = 1; p.kchunk = 0; p.kslab = nullptr;
    p.stat_slots = rl_row_blocks_host(p.a.M, GM_BM);          // (128-row tiles: the slot count of both products)
    p.N2 = b->N; p.Y2 = b->Y; p.stats2 = b->stats; p.wsplit2 = reinterpret_cast<const __bf16*>(b->W_split);
    p.piv_mean2 = b->stats ? b->stats_pivot_mean : nullptr; p.piv_bias2 = b->stats ? b->stats_pivot_bias : nullptr;
    p.pair_ny1 = rl_cdiv(a->N, 128);
    RL_REQUIRE(wgemm2_usable(p), RL_ERR_UNSUPPORTED, "rl_gemm_pair: the LDS-DMA kernel does not take this operand");
    // the grid of launch_wgemm for ny = the column blocks of both products
    p.ny = p.pair_ny1 + rl_cdiv(b->N, 128);
    p.gx = rl_cdiv(p.a.M, 128);
    int cap = cu_count() / p.ny / 8 * 8;
    if (cap < 8) cap = 8;
    if (p.gx > cap) p.gx = cap;
    const dim3 g2((unsigned)(8 * rl_cdiv(p.gx, 8) * p.ny), 1, 1);
    const bool stats = p.stats != nullptr;
    const WidePlan w{128, 128, 1};
    hipStream_t st = (hipStream_t)stream;
    if (wide_gemm_terms() == 1) {
        if (stats) launch_wgemm2_tile<1, true>(w, g2, st, p);
        else       launch_wgemm2_tile<1, false>(w, g2, st, p);
    } else {
        if (stats) launch_wgemm2_tile<3, true>(w, g2, st, p);
        else       launch_wgemm2_tile<3, false>(w, g2, st, p);
    }
    rl_note_kernel("wgemm2_kernel");
    RL_LAUNCH_CHECK("rl_gemm_pair");
    return RL_OK;
}

extern "C" int rl_split_weights(const rl_wsplit_item* items, int count, void* stream) {
    RL_REQUIRE(items != nullptr && count >= 0, RL_ERR_ARGS, "rl_split_weights: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < count; base += WS_MAX) {
        SplitBatch b;
        b.count = count - base < WS_MAX ? count - base : WS_MAX;
        long blocks = 0;
        for (int i = 0; i < b.count; ++i) {
            const rl_wsplit_item& it = items[base + i];
            RL_REQUIRE(it.W && it.out && it.K > 0 && it.N > 0, RL_ERR_ARGS, "rl_split_weights: bad item %d", base + i);
            RL_REQUIRE(((uintptr_t)it.out & 15) == 0, RL_ERR_ARGS, "rl_split_weights: item %d: out must be 16-byte aligned", base + i);
            RL_REQUIRE(it.K % 8 == 0 && (long)it.N * it.K < (1l << 31), RL_ERR_ARGS, "rl_split_weights: item %d: K %% 8 == 0, N * K < 2^31", base + i);
            b.item[i] = it;
            b.first_block[i] = (int)blocks;
            blocks += rl_cdiv((long)it.N * (it.K / 8), 256);
        }
        b.first_block[b.count] = (int)blocks;
        if (blocks > 0) hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
        RL_LAUNCH_CHECK("rl_split_weights");
    }
    rl_note_kernel("split_weights_kernel");
    return RL_OK;
}

// BatchNorm partial-statistics slots rl_gemm fills for an (M, N, K) product: one per 128-row block, or per 64-row block where
// the wide GEMM may run on 64-row tiles (wide_plan) - whichever kernel takes the launch zero-fills the slots it does not use.
extern "C" int64_t rl_gemm_stat_slots(int64_t M, int N, int K) {
    if (N > 64 && K % PG_BK == 0 && K <= W2_KMAX && wide_gemm_terms() != 0 && wgemm_staging() == 1) {
        const WidePlan w = wide_plan(M, N, K, true);
        if (w.bm == 64 && w.ksplit == 1) return rl_row_blocks_host(M, 64);
    }
    // (the narrow products the LDS-DMA kernel takes on 64 x 64 tiles, see rl_gemm)
    if (N > 16 && N <= 64 && K > 64 && K % PG_BK == 0 && K <= W2_KMAX && wide_gemm_terms() != 0 && wgemm_staging() == 1 && wgemm_narrow_dma())
        return rl_row_blocks_host(M, 64);
    return rl_row_blocks_host(M, GM_BM);
}

extern "C" int64_t rl_gemm_kslab_floats(int64_t M, int N, int K) {
    const int ks = gemm_ksplit(M, N, K);
    return ks > 1 ? (int64_t)ks * M * N : 0;
}

extern "C" int64_t rl_wgrad_slab_floats(int64_t M, int N, int K) {
    int nsplit; long rpb;
    if (stream_wgrad_ok(N, K)) swgrad_split(M, &nsplit, &rpb);
    else wgrad_split(M, N, K, &nsplit, &rpb);
    return (int64_t)nsplit * ((int64_t)N * K + N);
}

// descriptor -> kernel parameters + slab split (shared by rl_wgrad and rl_wgrad_batch)
static int wgrad_fill(const rl_wgrad_desc* d, WgradParams* pp, int* nsplit_out, bool* streaming_out) {
    RL_REQUIRE(d != nullptr, RL_ERR_ARGS, "rl_wgrad: null descriptor");
    WgradParams& p = *pp;
    int rc = fill_a(&p.a, "rl_wgrad", d->A, d->lda, d->a_bstride, d->a_mode, d->in_act, d->in_slope,
                    d->in_scale, d->in_shift, d->xyz, d->xyz_bstride, d->nbr_idx, d->nbr_d2, d->nbr_k,
                    d->B, d->n, d->K);
    if (rc) return rc;
    RL_REQUIRE(d->N > 0 && d->dY && d->dW && d->slab && d->lddy >= d->N, RL_ERR_ARGS, "rl_wgrad: bad dY/dW/slab");
    p.N = d->N; p.dY = d->dY; p.lddy = d->lddy; p.dy_bstride = d->dy_bstride;
    p.rows_per_batch = (d->a_mode == 1) ? d->n * d->nbr_k : d->n;
    RL_REQUIRE(d->dy_bstride >= p.rows_per_batch, RL_ERR_ARGS, "rl_wgrad: dy_bstride smaller than rows per cloud");
    p.dy_contig = (d->dy_bstride == p.rows_per_batch);
    p.slab = d->slab; p.has_bias = d->dbias != nullptr;
    int nsplit; long rpb;
    const bool streaming = stream_wgrad_ok(d->N, d->K);
    RL_REQUIRE(!(d->rows_bf16 && streaming), RL_ERR_UNSUPPORTED, "rl_wgrad: bf16 rows are supported by the wide weight-gradient kernel only");
    if (streaming) swgrad_split(p.a.M, &nsplit, &rpb);
    else wgrad_split(p.a.M, d->N, d->K, &nsplit, &rpb);
    RL_REQUIRE(d->slab_floats >= (int64_t)nsplit * ((int64_t)d->N * d->K + d->N), RL_ERR_ARGS,
               "rl_wgrad: slab too small (%ld floats)", (long)d->slab_floats);
    p.rows_per_block = rpb;
    *nsplit_out = nsplit; *streaming_out = streaming;
    return RL_OK;
}

// can this layer join a grouped launch (rl_wgrad_batch)?  The wide 128 x 128-tile kernel in a bf16 arithmetic mode, fp32 rows
static bool swgrad_batchable(const rl_wgrad_desc* d, bool streaming) {
    static const bool off = getenv("RL_NO_SWGRAD_BATCH") != nullptr;       // diagnostics: one launch per narrow layer
    return streaming && !off && d->a_mode == 0 && !d->rows_bf16;
}
static bool wgrad_batchable(const rl_wgrad_desc* d, const WgradParams& p, bool streaming) {
    if (swgrad_batchable(d, streaming)) return true;
    return !streaming && !d->rows_bf16 && pwgrad_ok(p) && wgrad_tile(d->N, d->K) == 128 && wide_gemm_terms() != 0 &&
           getenv("RL_WGRAD_4WAVE") == nullptr && rl_cdiv(d->N, 128) < 65536 && rl_cdiv(d->K, 128) < 65536;
}

extern "C" int rl_wgrad_batchable(const rl_wgrad_desc* d) {
    WgradParams p;
    int nsplit; bool streaming;
    if (wgrad_fill(d, &p, &nsplit, &streaming) != RL_OK) return 0;
    return wgrad_batchable(d, p, streaming) ? (streaming ? 2 : 1) : 0;      // 1: the wide tile kernel, 2: the streaming kernel
}

extern "C" int rl_wgrad_batch(const rl_wgrad_desc* descs, int count, void* stream) {
    RL_REQUIRE(descs != nullptr && count >= 0, RL_ERR_ARGS, "rl_wgrad_batch: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int terms = wide_gemm_terms();
    // the queue holds wide layers (128 x 128-tile kernel) and narrow ones (streaming kernel): one grouped launch per kind
    // and per WB_MAX layers
    bool any[2] = {false, false};
    for (int kind = 0; kind < 2; ++kind) {
        WgradBatch b;
        b.count = 0; b.pad = 0;
        long blocks = 0;
        auto flush = [&]() -> int {
            if (b.count == 0 || blocks == 0) { b.count = 0; blocks = 0; return RL_OK; }
            if (kind == 1)       hipLaunchKernelGGL(swgrad_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
            else if (terms == 1) hipLaunchKernelGGL(pwgrad128w_batch_kernel<1>, dim3((unsigned)blocks), dim3(512), 0, st, b);
            else                 hipLaunchKernelGGL(pwgrad128w_batch_kernel<3>, dim3((unsigned)blocks), dim3(512), 0, st, b);
            RL_LAUNCH_CHECK("rl_wgrad_batch");
            b.count = 0; blocks = 0;
            return RL_OK;
        };
        for (int i = 0; i < count; ++i) {
            const rl_wgrad_desc* d = descs + i;
            WgradParams p;
            int nsplit; bool streaming;
            int rc = wgrad_fill(d, &p, &nsplit, &streaming);
            if (rc) return rc;
            RL_REQUIRE(wgrad_batchable(d, p, streaming), RL_ERR_UNSUPPORTED,
                       "rl_wgrad_batch: layer %d (N = %d, K = %d) cannot join a grouped launch - check rl_wgrad_batchable first", i, d->N, d->K);
            RL_REQUIRE(d->defer_reduce, RL_ERR_ARGS, "rl_wgrad_batch: the layers' slabs are summed by rl_wgrad_reduce_batch (defer_reduce must be set)");
            if ((streaming ? 1 : 0) != kind) continue;
            any[kind] = true;
            WBItem& it = b.item[b.count];
            it.A = p.a.A; it.dY = p.dY; it.sc = p.a.lazy.scale; it.sh = p.a.lazy.shift; it.slab = p.slab;
            it.lda = p.a.lda; it.a_bstride = p.a.a_bstride; it.lddy = p.lddy; it.dy_bstride = p.dy_bstride;
            it.rows_per_block = p.rows_per_block; it.M = p.a.M;
            it.N = p.N; it.K = p.a.K; it.n = p.a.n; it.rows_per_batch = p.rows_per_batch; it.act = p.a.lazy.act; it.slope = p.a.lazy.slope;
            it.nsplit = nsplit;
            if (streaming) {
                const int kt = d->K <= 16 ? 0 : d->K <= 32 ? 1 : 2, nt = d->N <= 16 ? 0 : d->N <= 32 ? 1 : d->N <= 64 ? 2 : 3;
                it.gy = (unsigned short)(4 * kt + nt); it.gz = 1;
                it.first_block = (int)blocks;
                blocks += nsplit;
            } else {
                it.gy = (unsigned short)rl_cdiv(d->N, 128); it.gz = (unsigned short)rl_cdiv(d->K, 128);
                const int per = (int)it.gy * (int)it.gz;
                if (per > 1) blocks = (blocks + 7) / 8 * 8;          // (see the kernel: tiles of a row block share an XCD)
                it.first_block = (int)blocks;
                blocks += per > 1 ? (long)((nsplit + 7) / 8 * 8) * per : (long)nsplit;
            }
            it.a_contig = (unsigned char)p.a.contig; it.dy_contig = (unsigned char)p.dy_contig; it.has_bias = (unsigned char)p.has_bias; it.pad = 0;
            RL_REQUIRE(blocks < (1l << 30), RL_ERR_ARGS, "rl_wgrad_batch: too many workgroups");
            if (++b.count == WB_MAX) {
                rc = flush();
                if (rc) return rc;
            }
        }
        const int rc = flush();
        if (rc) return rc;
    }
    rl_note_kernel(any[1] && !any[0] ? "swgrad_batch_kernel" : "pwgrad128w_batch_kernel");
    return RL_OK;
}

extern "C" int rl_wgrad(const rl_wgrad_desc* d, void* stream) {
    WgradParams p;
    int nsplit; bool streaming;
    int rc = wgrad_fill(d, &p, &nsplit, &streaming);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (streaming) {
        if (d->K <= 16)      launch_swgrad<1>(d->N, dim3(nsplit), st, p);
        else if (d->K <= 32) launch_swgrad<2>(d->N, dim3(nsplit), st, p);
        else                 launch_swgrad<4>(d->N, dim3(nsplit), st, p);
        rl_note_kernel("swgrad_kernel");
    } else {
        const bool pipelined = pwgrad_ok(p);
        const int T = pipelined ? wgrad_tile(d->N, d->K) : WG_T;
        dim3 grid(nsplit, rl_cdiv(d->N, T), rl_cdiv(d->K, T));
        RL_REQUIRE(!d->rows_bf16 || (pipelined && T == 128 && wide_gemm_terms() != 0 && d->a_mode == 0), RL_ERR_UNSUPPORTED,
                   "rl_wgrad: bf16 rows are supported by the wide (128 x 128 tile) weight-gradient kernel only");
        if (d->rows_bf16) {
            RL_REQUIRE(d->lda % 4 == 0 && d->lddy % 4 == 0, RL_ERR_ARGS, "rl_wgrad: bf16 rows need leading dimensions that are multiples of 4");
            hipLaunchKernelGGL((pwgrad128w_kernel<1, true>), grid, dim3(512), 0, st, p);
        } else if (pipelined && T == 128) {
            const int t = wide_gemm_terms();
            static const bool narrow_wg = getenv("RL_WGRAD_4WAVE") != nullptr;      // diagnostics: the 4-wavefront kernel
            if (t == 0)      hipLaunchKernelGGL(pwgrad128_kernel<0>, grid, dim3(256), 0, st, p);
            else if (narrow_wg && t == 1) hipLaunchKernelGGL(pwgrad128_kernel<1>, grid, dim3(256), 0, st, p);
            else if (narrow_wg)           hipLaunchKernelGGL(pwgrad128_kernel<3>, grid, dim3(256), 0, st, p);
            else if (t == 1) hipLaunchKernelGGL(pwgrad128w_kernel<1>, grid, dim3(512), 0, st, p);
            else             hipLaunchKernelGGL(pwgrad128w_kernel<3>, grid, dim3(512), 0, st, p);
        }
        else if (pipelined) hipLaunchKernelGGL(pwgrad_kernel, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, st, p);
        rl_note_kernel(pipelined && T == 128 ? (wide_gemm_terms() != 0 && getenv("RL_WGRAD_4WAVE") == nullptr ? "pwgrad128w_kernel" : "pwgrad128_kernel")
                                             : pipelined ? "pwgrad_kernel" : "wgrad_kernel");
    }
    RL_LAUNCH_CHECK("rl_wgrad");
    if (d->defer_reduce) return RL_OK;
    const long per = (long)d->N * d->K + d->N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rl_cdiv(per, WR_E)), dim3(256), 0, st, d->slab, nsplit,
                       d->N, d->K, d->dW, (long)d->w_ks, (long)d->w_ns, d->dbias);
    RL_LAUNCH_CHECK("rl_wgrad_reduce");
    return RL_OK;
}

extern "C" int rl_wgrad_nsplit(int64_t M, int N, int K) {
    int nsplit; long rpb;
    if (stream_wgrad_ok(N, K)) swgrad_split(M, &nsplit, &rpb);
    else wgrad_split(M, N, K, &nsplit, &rpb);
    return nsplit;
}

extern "C" int rl_wgrad_reduce_batch(const rl_wgrad_reduce_item* items, int count, void* stream) {
    RL_REQUIRE(items != nullptr && count >= 0, RL_ERR_ARGS, "rl_wgrad_reduce_batch: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < count; base += RB_MAX) {
        ReduceBatch b;
        b.count = count - base < RB_MAX ? count - base : RB_MAX;
        long tiles = 0;
        for (int i = 0; i < b.count; ++i) {
            const rl_wgrad_reduce_item& it = items[base + i];
            RL_REQUIRE(it.slab && it.dW && it.nsplit > 0 && it.N > 0 && it.K > 0, RL_ERR_ARGS,
                       "rl_wgrad_reduce_batch: bad item %d", base + i);
            b.item[i] = it;
            b.first_tile[i] = (int)tiles;
            tiles += rl_cdiv((long)it.N * it.K + it.N, WR_E);
            RL_REQUIRE(tiles < (1l << 30), RL_ERR_ARGS, "rl_wgrad_reduce_batch: too many tiles");
        }
        b.first_tile[b.count] = (int)tiles;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)tiles), dim3(256), 0, st, b);
        RL_LAUNCH_CHECK("rl_wgrad_reduce_batch");
    }
    rl_note_kernel("wgrad_reduce_batch_kernel");
    return RL_OK;
}
