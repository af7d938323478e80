// Fused attentive pooling for the narrow levels (d = 16 / 32 / 64 channels, 16 neighbours):
// PointFeatureAugmentation gather + concat, the score Linear, the softmax over the K neighbours
// and the weighted sum (reference randlanet/utils/modules.py:213-221, 246-253) in one kernel, and
// the whole backward of that block in another.  The (points*K) x d tensors X = [rpe, gathered],
// S = X.W^T, dS and dX never reach HBM: a wavefront owns one point at a time, i.e. one 16-row
// MFMA block, so the softmax over K is a reduction over the 16 rows of its own accumulator tile
// (4 registers x the 4 lane groups: two __shfl_xor).
//
// Layouts (v_mfma_f32_16x16x4_f32):
//   "A layout"  lane (i = l&15, j = l>>4) holds a float4 = X[row i][16c + 4j .. +3]   (coalesced loads)
//   "C layout"  lane (lr = l&15, lq = l>>4), reg r -> element [row 4*lq + r][col 16*nb + lr]
// Tiles move between the two through a wavefront-private LDS tile (row stride d+4 floats: the
// C-layout reads of a half-wave fall in disjoint bank halves).  W and W^T sit in LDS for the
// whole kernel (stride d+4 as well), read as MFMA B fragments.
//
// Backward, per point:  recompute X, S, A = softmax_K(S), P;  dS = A*dP*(X - P);
//   dX = dP*A + dS.W  ->  first half to the rpe-branch gradient, second half to DG, the gradient of every
//   gathered row (both plain stores; rl_segment_sum_rows adds DG up per gathered point in a fixed order - no
//   atomics, bitwise reproducible);  dW += dS^T.X accumulates in registers over all points of the
//   wavefront and leaves the workgroup as one partial slab.
#include "rl_common.h"
#include <stdlib.h>

namespace {

// The (points * 16)-row tensors the pooling backward writes (GU, DG, X_out, dS_out) are read once, by a later kernel:
// non-temporal stores in the d <= 64 kernels (pool_bwd<1> accumulate 248 -> 225 us, pool_bwd<4> 228 -> 220, rpe_wgrad 162 ->
// 150; -DRL_POOL_PLAIN_STORES for the A/B).  Not in pool128_bwd: 182 -> 195 us inside a step (its X / dS go straight into the
// grouped weight-gradient launch).
#ifdef RL_POOL_PLAIN_STORES
#define RL_ST1 rl_stx
#define RL_ST4 rl_stx4
#else
#define RL_ST1 rl_stx_nt
#define RL_ST4 rl_stx4_nt
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// fp32 -> bf16 head + bf16 tail (value = hi + lo up to 2^-17 relative); the products of the score / gradient GEMMs are
// then a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 MFMAs with fp32 accumulation ("bf16x3", as in gemm.hip)
// (written on pairs so that it compiles to ten instructions per four values: two v_cvt_pk_bf16_f32 for the heads, a shift and a
// mask per pair to widen them again, two v_pk_add_f32 for the remainders, two v_cvt_pk_bf16_f32 for the tails - the
// element-wise form came out at fourteen to fifteen)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ void split2(const f32x2 x, unsigned& hi, unsigned& lo) {
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const f32x2 hf = {__uint_as_float(hb << 16), __uint_as_float(hb & 0xffff0000u)};
    const f32x2 l = x - hf;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2));
    hi = hb;
}
__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo) {
    unsigned h0, h1, l0, l1;
    split2((f32x2){v.x, v.y}, h0, l0);
    split2((f32x2){v.z, v.w}, h1, l1);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(bf16x4, (u2){h0, h1});
    lo = __builtin_bit_cast(bf16x4, (u2){l0, l1});
}
__device__ __forceinline__ bf16x8 cat8(const bf16x4 a, const bf16x4 b) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
    return r;
}
__device__ __forceinline__ f32x4 mfma16(const bf16x4 a, const bf16x4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

// The tile kernels of this file are bound by vector-instruction issue (profiles/r02_pmc_sq_step.md), so their elementwise
// passes are written on 4-vectors: the compiler turns those into v_pk_{add,mul,fma}_f32 (two floats per lane and
// instruction) where the scalar loops became four separate instructions.
// vmcnt counts loads AND stores in issue order, so the wait for a prefetched operand at the top of the next iteration would
// also wait for every store issued after it (the compiler emits vmcnt(0) there): a full store round trip exposed per
// point.  Calling this right BEFORE an iteration's stores retires the (long since issued) prefetch loads instead; the
// stores then drain in the background.
// The builtin (not inline asm: the waitcnt-insertion pass must SEE the wait, or it adds its own vmcnt(0) at the next use)
// between two empty asm statements with memory clobbers: to the optimiser the builtin alone touches no memory, so a
// prefetch could be sunk below it - one build did exactly that and the d = 16 kernels lost 18 %.
// After an iteration's load group: nothing may be scheduled across this point.  The prefetched values have no consumer
// inside the iteration, so the scheduler otherwise sinks the loads down to the wait (shortest live ranges).
__device__ __forceinline__ void loads_issued() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void loads_landed() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0); lgkmcnt / expcnt unconstrained
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ f32x4 splat(float v) { return (f32x4){v, v, v, v}; }
__device__ __forceinline__ f32x4 v4(const float4 a) { return (f32x4){a.x, a.y, a.z, a.w}; }
__device__ __forceinline__ float4 f4(const f32x4 a) { return make_float4(a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ float sum4(const f32x4 a) { return (a[0] + a[1]) + (a[2] + a[3]); }
// BatchNorm of a VIRTUAL rpe stage: one fused multiply-add per element, and the same expression in every kernel that
// needs the activation or its sign (forward, statistics, the three backward kernels) so that they agree bit for bit
__device__ __forceinline__ f32x4 vbn(const f32x4 raw, float s, float h) { return __builtin_elementwise_fma(raw, splat(s), splat(h)); }
__device__ __forceinline__ f32x4 vrelu(const f32x4 z) { return __builtin_elementwise_max(z, splat(0.f)); }
// activation of a lazily normalised tensor without a branch: act(z) = max(z, z*e) with e = 1 (none), 0 (ReLU) or the
// LeakyReLU slope (0 <= slope <= 1); equal in value to rl_act (a negative input of ReLU yields -0 instead of +0)
__device__ __forceinline__ float eff_slope(const RlLazy& t) {
    if (t.scale == nullptr || t.act == RL_ACT_NONE) return 1.f;
    return t.act == RL_ACT_RELU ? 0.f : t.slope;
}
__device__ __forceinline__ f32x4 vact(const f32x4 z, float e) { return __builtin_elementwise_max(z, z * splat(e)); }

struct PoolParams {
    const float* U;        // (P*16) x h raw rpe-branch features
    RlLazy ulazy;
    const float* G;        // per-point features, row (b, i) at (b*g_bstride + i)*h
    long g_bstride;
    RlLazy glazy;
    const int32_t* idx;    // (P, 16) neighbour indices inside the cloud
    const float* W;        // (d, d) score weight, [out][in] row-major
    long P;                // points = B*n
    int n;                 // points per cloud
    int d;
    // forward output / backward input
    float* Pout;           // (P, d)
    const float* dP;       // (P, d)
    // backward outputs
    float* GU;             // (P*16) x h
    int gu_accumulate;
    float* DG;             // (P*16) x h: gradient of the gathered row of every neighbourhood slot
    float* slab;           // per-workgroup partial dW: [grid][slab_stride], d*d used
    long slab_stride;      // d*d, or d*d + d when the caller sums the slabs with rl_wgrad_reduce_batch (its slab layout: dW then db)
    float* X_out;          // d = 128 backward: X and dS leave the kernel ((P*16) x d each); dW = dS^T.X is the caller's
    float* dS_out;         //   weight-gradient GEMM (64 accumulator tiles do not fit one wavefront)
    // u_source 1 / 2: the rpe-branch half of X is not a tensor but a function of the coordinates, recomputed per point:
    //   1: relu(bn1(rpe . W1^T + b1))                       (what mlp_rpe1 would have stored; U is ignored)
    //   2: relu(bn2(relu(bn1(rpe . W1^T + b1)) . W2^T + b2)) (mlp_rpe2 on top of it)
    int src;
    const float* xyz;      // (B, xyz_bstride, xyz_w): xyz_w = 3, or 4 (x, y, z, unused; 16-byte aligned)
    long xyz_bstride;
    int xyz_w;
    const float* nbr_d2;   // (P, 16) squared neighbour distances
    const float* W1; const float* b1; const float* sc1; const float* sh1;   // mlp_rpe1 (h x 10) + folded BatchNorm
    const float* W2; const float* b2; const float* sc2; const float* sh2;   // mlp_rpe2 (h x h)
    const float* mu1; const float* is1; const float* mu2; const float* is2; // saved mean / invstd (backward kernels)
    const float* piv1; const float* piv2;      // shifted statistics: running mean of stage 1 / 2 (forward statistics only), or null
    double* fstats2;       // pool_fwd with virtual stage 1: [grid][2][H] partial (sum, sum of squares) of the RAW stage-2 output
    double* bstats;        // pool_bwd with a virtual stage: [grid][2][H] partial sums of g and g*xhat of that stage's BatchNorm
    int xcd_chunk;         // > 0: workgroups of XCD x (blockIdx.x % 8) take the points [x * xcd_chunk, (x + 1) * xcd_chunk) - see PointSpan
};

template <int DT>
struct Tile {
    static constexpr int D = 16 * DT;
    static constexpr int H = D / 2;
    static constexpr int XS = D + 4;
    static constexpr int XSB = D + 8;   // bf16 row stride of the split weights: 16-byte aligned rows
};

// lane-constant lazy parameters for the float4 this lane loads in chunk c
template <int DT>
__device__ __forceinline__ void lane_lazy(const PoolParams& p, int lj, f32x4 (&sc)[DT], f32x4 (&sh)[DT]) {
    constexpr int H = Tile<DT>::H;
#pragma unroll
    for (int c = 0; c < DT; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * c + 4 * lj + s;
            if (k < H) {
                sc[c][s] = p.ulazy.scale ? p.ulazy.scale[k] : 1.f;
                sh[c][s] = p.ulazy.scale ? p.ulazy.shift[k] : 0.f;
            } else {
                sc[c][s] = p.glazy.scale ? p.glazy.scale[k - H] : 1.f;
                sh[c][s] = p.glazy.scale ? p.glazy.shift[k - H] : 0.f;
            }
        }
}
// effective activation slope (eff_slope) of the half of X that chunk c of this lane belongs to
template <int DT>
__device__ __forceinline__ float chunk_slope(int c, int lj, float es_u, float es_g) {
    if constexpr (DT == 1) return lj < 2 ? es_u : es_g;
    else return 16 * c < Tile<DT>::H ? es_u : es_g;
}

// The per-chunk lazy parameters as the kernels read them: registers up to d = 64; at d = 128 (sixteen float4s = 64 registers in
// kernels that sit at the 256-register limit) from an LDS image, 16 bytes per use.  `mem`: 2*D floats, published by the caller's barrier.
template <int DT>
struct LFold {
    static constexpr int D = 16 * DT;
    static constexpr bool INLDS = DT >= 8;
    static constexpr int LDS_FLOATS = INLDS ? 2 * D : 4;
    f32x4 rsc[INLDS ? 1 : DT], rsh[INLDS ? 1 : DT];
    const float* lds;
    __device__ __forceinline__ void init(const PoolParams& p, int lj, float* mem) {
        if constexpr (INLDS) {
            constexpr int H = D / 2;
            for (int k = threadIdx.x; k < D; k += blockDim.x) {
                const RlLazy& t = k < H ? p.ulazy : p.glazy;
                mem[k] = t.scale ? t.scale[k < H ? k : k - H] : 1.f;
                mem[D + k] = t.scale ? t.shift[k < H ? k : k - H] : 0.f;
            }
            lds = mem + 4 * lj;
        } else lane_lazy<DT>(p, lj, rsc, rsh);
    }
    __device__ __forceinline__ f32x4 sc(int c) const {
        if constexpr (INLDS) return *reinterpret_cast<const f32x4*>(lds + 16 * c);
        else return rsc[c];
    }
    __device__ __forceinline__ f32x4 sh(int c) const {
        if constexpr (INLDS) return *reinterpret_cast<const f32x4*>(lds + D + 16 * c);
        else return rsh[c];
    }
};

// Position of a wavefront in its sequence of points: the point, its cloud and its index inside the cloud.  All three are
// wavefront-uniform (the wave number is read through readfirstlane), so this arithmetic - and every address built from
// it - runs on the scalar unit; the cloud is tracked incrementally instead of dividing by n per point.
struct Cursor {
    long pt;
    int b, i;
    __device__ __forceinline__ void start(long pt0, int n) {
        pt = pt0;
        b = (int)((unsigned)pt0 / (unsigned)n);   // points * 16 < 2^31 (checked on the host)
        i = (int)(pt0 - (long)b * n);
    }
    __device__ __forceinline__ Cursor next(long pstep, int n) const {
        Cursor c;
        c.pt = pt + pstep;
        c.b = b;
        c.i = i + (int)pstep;
        while (c.i >= n) { c.i -= n; ++c.b; }
        return c;
    }
};

// Which points a wavefront visits: first, first + step, ... < end.
// Workgroup ids are dealt round-robin over the chip's 8 XCDs, each with an L2 of its own (4 MB).  With the points dealt
// round-robin over the workgroups too, every XCD gathers neighbour rows (and coordinates) from EVERY cloud of the batch: at
// bs = 8 the level-0 tables are 10.5 + 3.9 MB, no L2 holds them and nearly every 32-byte gathered row arrives as a 128-byte
// line from the Infinity Cache (PMC: 620 MB fetched by a level-0 forward launch whose streamed operands are 60 MB).  XCD-local
// ranges instead: XCD x owns the points [x * chunk, (x + 1) * chunk) - at bs = 8 one cloud, whose tables (1.3 + 0.5 MB) then
// stay in that XCD's L2.  Same points per workgroup as before, another assignment; partial sums per workgroup as before.
struct PointSpan { int first, step, end; };
template <int NW>
__device__ __forceinline__ PointSpan point_span(const PoolParams& p, int wave) {
    PointSpan s;
    if (p.xcd_chunk > 0) {          // (host: the grid is a multiple of 8)
        const int x = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        const int lo = x * p.xcd_chunk, hi = lo + p.xcd_chunk;
        s.end = hi < (int)p.P ? hi : (int)p.P;
        s.first = lo + slot * NW + wave;
        s.step = ((int)gridDim.x >> 3) * NW;
    } else {
        s.end = (int)p.P;
        s.first = (int)blockIdx.x * NW + wave;
        s.step = (int)gridDim.x * NW;
    }
    return s;
}

// The 16 x d tile of X for point pt in A layout, in two steps so that the loads of the NEXT point can be in flight
// while the current one is computed: fetch_x issues the raw loads (rpe-branch rows + gathered rows),
// finish_x applies the lazy BatchNorm / activation and parks the tile in LDS.
template <int DT>
__device__ __forceinline__ void fetch_x(const PoolParams& p, const Cursor& cu, int li, int lj, int my_idx, float4 (&raw)[DT]) {
    constexpr int H = Tile<DT>::H;
    const long row = cu.pt * 16 + li;
    const long b = cu.b;
#pragma unroll
    for (int c = 0; c < DT; ++c) {
        const int k = 16 * c + 4 * lj;
        if (k < H) {
            if (p.src == 0) raw[c] = *reinterpret_cast<const float4*>(p.U + row * H + k);
        } else raw[c] = *reinterpret_cast<const float4*>(p.G + (b * p.g_bstride + my_idx) * H + (k - H));
    }
}
template <int DT>
__device__ __forceinline__ void finish_x(const PoolParams& p, int li, int lj, const float4 (&raw)[DT],
                                         const LFold<DT>& lf, float4 (&xa)[DT], float* Xs) {
    constexpr int XS = Tile<DT>::XS;
    const float es_u = eff_slope(p.ulazy), es_g = eff_slope(p.glazy);
#pragma unroll
    for (int c = 0; c < DT; ++c) {
        const int k = 16 * c + 4 * lj;
        const float4 v = f4(vact(v4(raw[c]) * lf.sc(c) + lf.sh(c), chunk_slope<DT>(c, lj, es_u, es_g)));
        xa[c] = v;
        *reinterpret_cast<float4*>(Xs + li * XS + k) = v;
    }
}

// acc[nb] += tile(A layout) . B, with B given TRANSPOSED in LDS: Bt[n][k] (stride XS), so that the four k of an
// A-layout float4 (k = 16c + 4j .. +3) are one ds_read_b128 per column block.  All reads of a 16-column chunk of k
// are issued before its MFMAs.  Row stride XS = D+4 floats: 16 rows x b128 fall on 16 distinct bank quads.
template <int DT>
__device__ __forceinline__ void tile_gemm(const float4 (&a)[DT], const float* Bt, int li, int lj, f32x4 (&acc)[DT]) {
    constexpr int XS = Tile<DT>::XS;
    const float* base = Bt + li * XS + 4 * lj;
#pragma unroll
    for (int c = 0; c < DT; ++c) {
        float4 b[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) b[nb] = *reinterpret_cast<const float4*>(base + nb * 16 * XS + 16 * c);
        const float av[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int nb = 0; nb < DT; ++nb) {
                const float bv = s == 0 ? b[nb].x : s == 1 ? b[nb].y : s == 2 ? b[nb].z : b[nb].w;
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv, acc[nb], 0, 0, 0);
            }
        }
    }
}

// The same product with bf16x3 arithmetic.  Bh / Bl: head and tail of the TRANSPOSED B ([n][k], stride XSB bf16).
// D >= 32: v_mfma_f32_16x16x32_bf16 on pairs of 16-column chunks - the lane's A-layout float4s of chunks 2b and
// 2b+1 are its 8 k-values (k = 32b + 4j..+3 and 32b + 16 + 4j..+3; any k order is valid when A and B agree), the B
// fragment is the two matching 8-byte pieces of row n.  D = 16: one v_mfma_f32_16x16x16_bf16 per column block.
template <int DT>
__device__ __forceinline__ void tile_gemm_bf(const float4 (&a)[DT], const __bf16* Bh, const __bf16* Bl, int li, int lj,
                                             f32x4 (&acc)[DT]) {
    constexpr int XSB = Tile<DT>::XSB;
    const int off = li * XSB + 4 * lj;
    if constexpr (DT == 1) {
        bf16x4 ah, al;
        split4(a[0], ah, al);
        const bf16x4 bh = *reinterpret_cast<const bf16x4*>(Bh + off);
        const bf16x4 bl = *reinterpret_cast<const bf16x4*>(Bl + off);
        acc[0] = mfma16(ah, bh, acc[0]);
        acc[0] = mfma16(ah, bl, acc[0]);
        acc[0] = mfma16(al, bh, acc[0]);
    } else {
#pragma unroll
        for (int b = 0; b < DT / 2; ++b) {
            bf16x4 h0, l0, h1, l1;
            split4(a[2 * b], h0, l0);
            split4(a[2 * b + 1], h1, l1);
            const bf16x8 ah = cat8(h0, h1), al = cat8(l0, l1);
            constexpr int NG = DT < 4 ? DT : 4;      // column blocks per round: bounds the fragment registers
#pragma unroll
            for (int n0 = 0; n0 < DT; n0 += NG) {
                bf16x8 bh[NG], bl[NG];
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int o = off + (n0 + j) * 16 * XSB + 32 * b;
                    bh[j] = cat8(*reinterpret_cast<const bf16x4*>(Bh + o), *reinterpret_cast<const bf16x4*>(Bh + o + 16));
                    bl[j] = cat8(*reinterpret_cast<const bf16x4*>(Bl + o), *reinterpret_cast<const bf16x4*>(Bl + o + 16));
                }
#pragma unroll
                for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[n0 + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[n0 + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[n0 + j], 0, 0, 0);
            }
        }
    }
}


// ---- reductions over the four lane groups (the 16 rows of a C-layout tile): v_permlane16_swap / v_permlane32_swap (gfx950) instead
// of ds_bpermute round trips through the LDS pipe; the same operand pairs as __shfl_xor(v, 16) / (v, 32), so the same bits ----
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// sqrt of a squared neighbour distance: the hardware instruction alone (sqrtf wraps it in a range scaling for denormal
// arguments - a compare, two selects, two ldexp - that a d2 below 1e-38 would need; such a distance is 0 to everything it
// meets).  One expression in every kernel of the branch.
__device__ __forceinline__ float vsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// all-reduce over the four lane groups (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float lg_sum(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// two sums at once, seven instructions instead of twelve: the first swap pairs a's upper half with b's lower half, so that one
// add reduces BOTH over l ^ 32 (a's result in lanes 0..31, b's in lanes 32..63); one more swap + add over l ^ 16; the last swap
// hands every lane both totals.  (Pairs (l, l^32) before (l, l^16): the other order than lg_sum - sums of four positive
// softmax terms either way.)
__device__ __forceinline__ void lg_sum2(float& a, float& b) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    float z = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    z = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// max(a, b) of finite values as med3(a, b, FLT_MAX): fmaxf (and med3 against +inf, which the compiler folds back into it)
// canonicalises every operand first (a v_max_f32 v, v, v each - MFMA results could be signalling NaNs for all it knows), and
// an inline-asm v_max hides the MFMA -> VALU read hazard from the hazard recogniser (no s_nop: it read accumulators before
// they were written)
__device__ __forceinline__ float vmax2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 0x1.fffffep127f); }
__device__ __forceinline__ float lg_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// softmax over the 16 rows of a C-layout tile, in place: s -> A
template <int DT>
__device__ __forceinline__ void softmax_rows(f32x4 (&s)[DT]) {
    // exp(x) = 2^(x*log2 e) on the hardware exponential, 1/den on the hardware reciprocal (1 ulp each; the argument's
    // rounding adds |x|*2^-24 relative - 1e-6 at the largest score differences that still matter): an accurate expf and an
    // IEEE division are ~12 instructions each, a tenth of these kernels' instruction stream.  Forward and backward share
    // this function, so the backward differentiates exactly the attention weights the forward used.
    constexpr float LOG2E = 1.44269504088896340736f;
#pragma unroll
    for (int nb = 0; nb < DT; ++nb) {
        const float m = lg_max(vmax2(vmax2(s[nb][0], s[nb][1]), vmax2(s[nb][2], s[nb][3])));
        float den = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[nb][r] = __builtin_amdgcn_exp2f((s[nb][r] - m) * LOG2E);
            den += s[nb][r];
        }
        den = lg_sum(den);
        const float inv = __builtin_amdgcn_rcpf(den);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[nb][r] *= inv;
    }
}

template <int DT>
__device__ __forceinline__ void stage_w(const PoolParams& p, float* Wt, float* Wn, int nthreads = 256) {
    constexpr int D = Tile<DT>::D, XS = Tile<DT>::XS;
    // tile_gemm wants B transposed ([n][k]):  S = X.W^T has B^T = W itself -> Wn;  dX = dS.W has B^T = W^T -> Wt
    for (int e = threadIdx.x; e < D * D; e += nthreads) {
        const int o = e / D, i = e - o * D;
        const float w = p.W[e];
        Wn[o * XS + i] = w;
        if (Wt) Wt[i * XS + o] = w;
    }
}

template <int DT, int TERMS, int NW = 4>   // TERMS 0: fp32 MFMA; 3: bf16x3 (head + tail operands, three bf16 MFMAs per product); NW wavefronts.  (A virtual rpe half: vpool_fwd_kernel.)
__global__ __launch_bounds__(64 * NW) void pool_fwd_kernel(const PoolParams p) {
    constexpr int D = Tile<DT>::D, XS = Tile<DT>::XS, XSB = Tile<DT>::XSB;
    __shared__ __attribute__((aligned(16))) float lfm[LFold<DT>::LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) unsigned char wmem[TERMS == 0 ? D * XS * 4 : 2 * D * XSB * 2];
    __shared__ __attribute__((aligned(16))) float Xt[NW][16 * XS];
    float* Wt = reinterpret_cast<float*>(wmem);
    __bf16* Wh = reinterpret_cast<__bf16*>(wmem);
    __bf16* Wl = Wh + D * XSB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave: an SGPR
    const int li = lane & 15, lj = lane >> 4;
    for (int e = threadIdx.x; e < D * D; e += 64 * NW) {
        const int o = e / D, i = e - o * D;
        const float w = p.W[e];            // [n][k]: the transposed-B form the tile GEMMs read
        if constexpr (TERMS == 0) {
            Wt[o * XS + i] = w;
        } else {
            const __bf16 h = (__bf16)w;
            Wh[o * XSB + i] = h;
            Wl[o * XSB + i] = (__bf16)(w - (float)h);
        }
    }
    LFold<DT> lf;
    lf.init(p, lj, lfm);
    __syncthreads();
    float* Xs = Xt[wave];
    // software pipeline over the wavefront's points: neighbour index two points ahead, raw rows one point ahead
    const PointSpan span = point_span<NW>(p, wave);
    const long pstep = span.step, Pend = span.end;
    long pt = span.first;
    int idx_cur = pt < Pend ? p.idx[pt * 16 + li] : 0;
    int idx_nxt = pt + pstep < Pend ? p.idx[(pt + pstep) * 16 + li] : 0;
    float4 raw[DT];
    Cursor cu;
    cu.start(pt, p.n);
    if (pt < Pend) fetch_x<DT>(p, cu, li, lj, idx_cur, raw);
    loads_landed();      // the prologue's loads: the loop is then entered with nothing pending, like its back edge (otherwise the
                         // compiler's own wait at the loop top covers both entries and, on the back edge, also the stores)
    for (; pt < Pend; pt += pstep) {
        const Cursor cn = cu.next(pstep, p.n);
        float4 xa[DT];
        finish_x<DT>(p, li, lj, raw, lf, xa, Xs);
        const int idx_n2 = pt + 2 * pstep < Pend ? p.idx[(pt + 2 * pstep) * 16 + li] : 0;
        // no branch around the prefetch: past the last point the CURRENT point is read again (never used).  With a branch
        // the loaded registers are merged with the old ones behind it - copies, and a wait for loads just issued.
        const Cursor cf = pt + pstep < Pend ? cn : cu;
        fetch_x<DT>(p, cf, li, lj, pt + pstep < Pend ? idx_nxt : idx_cur, raw);
        loads_issued();
        cu = cn;
        idx_cur = idx_nxt; idx_nxt = idx_n2;
        f32x4 s[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) s[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (TERMS == 0) tile_gemm<DT>(xa, Wt, li, lj, s);
        else tile_gemm_bf<DT>(xa, Wh, Wl, li, lj, s);
        softmax_rows<DT>(s);
        __builtin_amdgcn_wave_barrier();
        loads_landed();
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) {
            f32x4 xc;
#pragma unroll
            for (int r = 0; r < 4; ++r) xc[r] = Xs[(lj * 4 + r) * XS + nb * 16 + li];
            float acc = sum4(s[nb] * xc);
            acc = lg_sum(acc);
            if (lj == 0) p.Pout[pt * D + nb * 16 + li] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The rpe half of X is a real tensor U here (a virtual one: vpool_bwd_kernel).  GB: DG is stored as bf16 (the bf16-storage
// mode); GU - a real tensor's gradient, it goes on into the fp32 GEMM chain - stays fp32.  `VIRT` is kept in the template list as
// `false` only (the instantiations name it).
template <int DT, int TERMS, bool VIRT = false, int NW = 4, bool GB = false>   // NW wavefronts share the staged weights
__global__ __launch_bounds__(64 * NW) void pool_bwd_kernel(const PoolParams p) {
    static_assert(!VIRT, "a virtual rpe stage runs vpool_bwd_kernel");
    __shared__ __attribute__((aligned(16))) float lfm[LFold<DT>::LDS_FLOATS];
    constexpr bool GUB = false;
    constexpr int D = Tile<DT>::D, H = Tile<DT>::H, XS = Tile<DT>::XS, XSB = Tile<DT>::XSB;
    // LDS: W^T, W, and per wavefront an X tile and a dS tile; after the main loop the W region is
    // reused to combine the four wavefronts' dW tiles
    __shared__ __attribute__((aligned(16))) float Wmem[TERMS == 0 ? 2 * D * XS : 2 * D * XSB];   // bf16: 4 arrays of D*XSB
    __shared__ __attribute__((aligned(16))) float Tiles[NW][2][16 * XS];
    float* Wt = Wmem;
    float* Wn = Wmem + D * XS;
    __bf16* Wnh = reinterpret_cast<__bf16*>(Wmem);      // [n][k] head / tail: S = X.W^T
    __bf16* Wnl = Wnh + D * XSB;
    __bf16* Wth = Wnl + D * XSB;                          // [n'][k'] = W[k'][n'] head / tail: dX = dS.W
    __bf16* Wtl = Wth + D * XSB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave: an SGPR
    const int li = lane & 15, lj = lane >> 4;
    if constexpr (TERMS == 0) {
        stage_w<DT>(p, Wt, Wn, 64 * NW);
    } else {
        for (int e = threadIdx.x; e < D * D; e += 64 * NW) {
            const int o = e / D, i = e - o * D;
            const float w = p.W[e];
            const __bf16 h = (__bf16)w, l = (__bf16)(w - (float)h);
            Wnh[o * XSB + i] = h; Wnl[o * XSB + i] = l;
            Wth[i * XSB + o] = h; Wtl[i * XSB + o] = l;
        }
    }
    LFold<DT> lf;
    lf.init(p, lj, lfm);
    __syncthreads();
    float* Xs = Tiles[wave][0];
    float* Ds = Tiles[wave][1];
    f32x4 accw[DT][DT];
#pragma unroll
    for (int nb = 0; nb < DT; ++nb)
#pragma unroll
        for (int kb = 0; kb < DT; ++kb) accw[nb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // software pipeline over the wavefront's points: neighbour index two points ahead, raw rows one point ahead
    const PointSpan span = point_span<NW>(p, wave);
    const long pstep = span.step, Pend = span.end;
    long pt = span.first;
    int idx_cur = pt < Pend ? p.idx[pt * 16 + li] : 0;
    int idx_nxt = pt + pstep < Pend ? p.idx[(pt + pstep) * 16 + li] : 0;
    float4 raw[DT];
    Cursor cu;
    cu.start(pt, p.n);
    // Every load of an iteration belongs to ONE group issued at its top - the next point's rows, coordinates and dP, the
    // index two points ahead, and this point's GU when accumulating - and the only vector-memory wait is loads_landed()
    // before the iteration's stores, a whole iteration of arithmetic later (vmcnt is in-order: a wait for a young load in
    // the middle of the iteration would also wait for the prefetches issued just before it).
    float gp[DT], gp_nxt[DT];          // dP of the current / next point
#pragma unroll
    for (int nb = 0; nb < DT; ++nb) gp[nb] = gp_nxt[nb] = 0.f;
    if (pt < Pend) {
        fetch_x<DT>(p, cu, li, lj, idx_cur, raw);
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) gp[nb] = p.dP[pt * D + nb * 16 + li];
    }
    constexpr int NGU = DT == 1 ? 1 : DT / 2;     // column blocks that hold rpe-branch (GU) columns
    loads_landed();      // the prologue's loads: the loop is then entered with nothing pending, like its back edge (otherwise the
                         // compiler's own wait at the loop top covers both entries and, on the back edge, also the stores)
    for (; pt < Pend; pt += pstep) {
        const Cursor cn = cu.next(pstep, p.n);
        float4 xa[DT];
        finish_x<DT>(p, li, lj, raw, lf, xa, Xs);
        const int idx_n2 = pt + 2 * pstep < Pend ? p.idx[(pt + 2 * pstep) * 16 + li] : 0;
        // no branch around the prefetch: past the last point the CURRENT point is read again (never used).  With a branch
        // the loaded registers are merged with the old ones behind it - copies, and a wait for loads just issued.
        const Cursor cf = pt + pstep < Pend ? cn : cu;
        fetch_x<DT>(p, cf, li, lj, pt + pstep < Pend ? idx_nxt : idx_cur, raw);
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) gp_nxt[nb] = p.dP[cf.pt * D + nb * 16 + li];
        f32x4 gacc[NGU];                 // GU of this point when this launch adds to it
#pragma unroll
        for (int nb = 0; nb < NGU; ++nb) {
            gacc[nb] = splat(0.f);
            if (p.gu_accumulate && nb * 16 + li < H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gacc[nb][r] = rl_ldx<GUB>(p.GU, (pt * 16 + lj * 4 + r) * H + nb * 16 + li);
            }
        }
        loads_issued();
        cu = cn;
        idx_cur = idx_nxt; idx_nxt = idx_n2;
        f32x4 a[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) a[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (TERMS == 0) tile_gemm<DT>(xa, Wn, li, lj, a);
        else tile_gemm_bf<DT>(xa, Wnh, Wnl, li, lj, a);
        softmax_rows<DT>(a);
        __builtin_amdgcn_wave_barrier();
        // C-layout pass: P, dS (to LDS), dXa kept in registers as the start of dX
        f32x4 dx[DT], dsr[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) {
            const int col = nb * 16 + li;
            f32x4 xc;
#pragma unroll
            for (int r = 0; r < 4; ++r) xc[r] = Xs[(lj * 4 + r) * XS + col];
            float pool = sum4(a[nb] * xc);
            pool = lg_sum(pool);
            dx[nb] = a[nb] * splat(gp[nb]);                          // direct path dP*A
            dsr[nb] = dx[nb] * (xc - splat(pool));                   // dS = A*dP*(X-P)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ds[(lj * 4 + r) * XS + col] = dsr[nb][r];
        }
        __builtin_amdgcn_wave_barrier();
        // dX += dS . W   (dS re-read in A layout)
        float4 da[DT];
#pragma unroll
        for (int c = 0; c < DT; ++c) da[c] = *reinterpret_cast<const float4*>(Ds + li * XS + 16 * c + 4 * lj);
        if constexpr (TERMS == 0) tile_gemm<DT>(da, Wt, li, lj, dx);
        else tile_gemm_bf<DT>(da, Wth, Wtl, li, lj, dx);
        // dW[n][k] += sum_rows dS[row][n] * X[row][k]   (rows are the MFMA reduction index)
        if constexpr (TERMS == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float bx[DT];
#pragma unroll
                for (int kb = 0; kb < DT; ++kb) bx[kb] = Xs[(4 * t + lj) * XS + kb * 16 + li];
#pragma unroll
                for (int nb = 0; nb < DT; ++nb) {
                    const float ad = Ds[(4 * t + lj) * XS + nb * 16 + li];
#pragma unroll
                    for (int kb = 0; kb < DT; ++kb)
                        accw[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad, bx[kb], accw[nb][kb], 0, 0, 0);
                }
            }
        } else {
            // v_mfma_f32_16x16x16_bf16: the lane's four reduction rows are 4*lj .. 4*lj+3 - for the A operand (dS^T) those
            // are exactly its C-layout registers of dS, no LDS round trip; the B operand is X at the same four rows
            bf16x4 xh[DT], xl[DT];
#pragma unroll
            for (int kb = 0; kb < DT; ++kb) {
                float4 v;
                v.x = Xs[(4 * lj + 0) * XS + kb * 16 + li]; v.y = Xs[(4 * lj + 1) * XS + kb * 16 + li];
                v.z = Xs[(4 * lj + 2) * XS + kb * 16 + li]; v.w = Xs[(4 * lj + 3) * XS + kb * 16 + li];
                split4(v, xh[kb], xl[kb]);
            }
#pragma unroll
            for (int nb = 0; nb < DT; ++nb) {
                bf16x4 dh, dl;
                split4(f4(dsr[nb]), dh, dl);
#pragma unroll
                for (int kb = 0; kb < DT; ++kb) accw[nb][kb] = mfma16(dh, xh[kb], accw[nb][kb]);
#pragma unroll
                for (int kb = 0; kb < DT; ++kb) accw[nb][kb] = mfma16(dh, xl[kb], accw[nb][kb]);
#pragma unroll
                for (int kb = 0; kb < DT; ++kb) accw[nb][kb] = mfma16(dl, xh[kb], accw[nb][kb]);
            }
        }
        // outputs of dX: columns < H -> rpe-branch gradient, columns >= H -> gradient of this slot's gathered row.
        // d <= 32: the finished tile (C layout: a lane holds four ROWS of one column) goes through the wavefront's dS
        // tile - free by now - and leaves in A layout: one 16-byte store per lane and 16-column chunk, 512 contiguous
        // bytes per point and tensor, instead of a 4-byte store per element (pool_bwd<1>: 515 -> 493 us per step).
        loads_landed();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rowi = lj * 4 + r;
#pragma unroll
            for (int nb = 0; nb < DT; ++nb) {
                const int col = nb * 16 + li;
                float v = dx[nb][r];
                if (col < H) {
                    if (nb < NGU) v += gacc[nb][r];
                }
                if constexpr (DT <= 2) Ds[rowi * XS + col] = v;
                else {                      // d = 64: measured better with the element stores (11 us per step)
                    const long urow = (pt * 16 + rowi) * H;
                    if (col < H) RL_ST1<GUB>(p.GU, urow + col, v);
                    else RL_ST1<GB>(p.DG, urow + (col - H), v);
                }
            }
        }
        if constexpr (DT <= 2) {
            __builtin_amdgcn_wave_barrier();
            const long orow = (pt * 16 + li) * H;
#pragma unroll
            for (int c = 0; c < DT; ++c) {
                const int k = 16 * c + 4 * lj;
                const float4 v = *reinterpret_cast<const float4*>(Ds + li * XS + k);
                if (k < H) RL_ST4<GUB>(p.GU, orow + k, v);
                else RL_ST4<GB>(p.DG, orow + (k - H), v);
            }
        }
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) gp[nb] = gp_nxt[nb];
        __builtin_amdgcn_wave_barrier();
    }
    // combine the four wavefronts' dW tiles in a fixed order (W region is free now)
    __syncthreads();
    float* red = Wmem;  // needs DT*DT*256 floats <= 2*D*XS: 16*DT*DT*16 <= 2*16*DT*(16*DT+4) always holds
    for (int w = 1; w < NW; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int nb = 0; nb < DT; ++nb)
#pragma unroll
                for (int kb = 0; kb < DT; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((nb * DT + kb) * 4 + r) * 64 + lane] = accw[nb][kb][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int nb = 0; nb < DT; ++nb)
#pragma unroll
                for (int kb = 0; kb < DT; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accw[nb][kb][r] += red[((nb * DT + kb) * 4 + r) * 64 + lane];
        }
    }
    if (wave == 0) {
        float* out = p.slab + (long)blockIdx.x * p.slab_stride;
#pragma unroll
        for (int nb = 0; nb < DT; ++nb)
#pragma unroll
            for (int kb = 0; kb < DT; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(long)(nb * 16 + lj * 4 + r) * D + kb * 16 + li] = accw[nb][kb][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// d = 128 backward (bf16x3 / bf16 arithmetic only).  LDS holds ONE image of W, [o][i] as bf16 head + tail (70 KB):
//   S  = X.W^T   reads it row-wise    (tile_gemm_bf: B^T[n = o][k = i], 8-byte pieces)
//   dX = dS.W    reads it column-wise (B[k = o][n = i]: a lane's four o of one column are four 2-byte reads) with
//                v_mfma_f32_16x16x16_bf16, whose A-operand lane (row, o-group) holds 4 consecutive o = the float4 of
//                the A-layout dS tile
// dW[o][i] = sum_rows dS[row][o] X[row][i] is 64 accumulator tiles - too many for a wavefront that also carries the
// rest - so X and dS are written out ((P*16) x 128 each, coalesced 16-byte rows) and the caller runs its ordinary wide
// weight-gradient kernel on them.  Everything else is pool_bwd_kernel: one point per wavefront, loads of the next
// point in flight.
// ---------------------------------------------------------------------------------------------------------------
template <bool GB>      // X_out, dS_out and DG stored as bf16 (GU is a real tensor's gradient: fp32)
__global__ __launch_bounds__(512) void pool128_bwd_kernel(const PoolParams p) {
    __shared__ __attribute__((aligned(16))) float lfm[LFold<8>::LDS_FLOATS];
    constexpr int NW = 8;    // 70 KB of W in LDS: one workgroup per CU, so it brings eight wavefronts
    constexpr int TERMS = 3;
    constexpr int DT = 8, D = 128, H = 64, XS = Tile<DT>::XS, XSB = Tile<DT>::XSB;
    __shared__ __attribute__((aligned(16))) __bf16 Wh[D * XSB];
    __shared__ __attribute__((aligned(16))) __bf16 Wl[D * XSB];
    __shared__ __attribute__((aligned(16))) float Tiles[NW][16 * XS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave: an SGPR
    const int li = lane & 15, lj = lane >> 4;
    for (int e = threadIdx.x; e < D * D; e += 64 * NW) {
        const int o = e / D, i = e - o * D;
        const float w = p.W[e];
        const __bf16 h = (__bf16)w;
        Wh[o * XSB + i] = h;
        if constexpr (TERMS == 3) Wl[o * XSB + i] = (__bf16)(w - (float)h);
    }
    LFold<DT> lf;
    lf.init(p, lj, lfm);
    __syncthreads();
    // one tile per wavefront: X in C layout is read and dS written by the SAME lane at the same element, so dS
    // replaces X in place (nothing needs X afterwards - the weight gradient is external)
    float* Xs = Tiles[wave];
    float* Ds = Tiles[wave];

    const PointSpan span = point_span<NW>(p, wave);
    const long pstep = span.step, Pend = span.end;
    long pt = span.first;
    int idx_cur = pt < Pend ? p.idx[pt * 16 + li] : 0;
    int idx_nxt = pt + pstep < Pend ? p.idx[(pt + pstep) * 16 + li] : 0;
    float4 raw[DT];
    Cursor cu;
    cu.start(pt, p.n);
    float gp[DT], gp_nxt[DT];      // dP of the current / next point (one load group per iteration, as in pool_bwd_kernel)
#pragma unroll
    for (int nb = 0; nb < DT; ++nb) gp[nb] = gp_nxt[nb] = 0.f;
    if (pt < Pend) {
        fetch_x<DT>(p, cu, li, lj, idx_cur, raw);
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) gp[nb] = p.dP[pt * D + nb * 16 + li];
    }
    loads_landed();      // the prologue's loads: the loop is then entered with nothing pending, like its back edge (otherwise the
                         // compiler's own wait at the loop top covers both entries and, on the back edge, also the stores)
    for (; pt < Pend; pt += pstep) {
        const Cursor cn = cu.next(pstep, p.n);
        float4 xa[DT];
        finish_x<DT>(p, li, lj, raw, lf, xa, Xs);
        const int idx_n2 = pt + 2 * pstep < Pend ? p.idx[(pt + 2 * pstep) * 16 + li] : 0;
        // no branch around the prefetch: past the last point the CURRENT point is read again (never used).  With a branch
        // the loaded registers are merged with the old ones behind it - copies, and a wait for loads just issued.
        const Cursor cf = pt + pstep < Pend ? cn : cu;
        fetch_x<DT>(p, cf, li, lj, pt + pstep < Pend ? idx_nxt : idx_cur, raw);
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) gp_nxt[nb] = p.dP[cf.pt * D + nb * 16 + li];
        loads_issued();
        cu = cn;
        idx_cur = idx_nxt; idx_nxt = idx_n2;
        // X leaves the kernel for the weight-gradient GEMM: row li, 16 bytes per 16-column chunk
#pragma unroll
        for (int c = 0; c < DT; ++c) rl_stx4<GB>(p.X_out, (pt * 16 + li) * D + 16 * c + 4 * lj, xa[c]);      // (non-temporal stores here: measured, rejected - DESIGN.md section 5)
        f32x4 a[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) a[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile_gemm_bf<DT>(xa, Wh, Wl, li, lj, a);
        softmax_rows<DT>(a);
        __builtin_amdgcn_wave_barrier();
        f32x4 dx[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) {
            const int col = nb * 16 + li;
            f32x4 xc;
#pragma unroll
            for (int r = 0; r < 4; ++r) xc[r] = Xs[(lj * 4 + r) * XS + col];
            float pool = sum4(a[nb] * xc);
            pool = lg_sum(pool);
            dx[nb] = a[nb] * splat(gp[nb]);                          // direct path dP*A
            const f32x4 ds = dx[nb] * (xc - splat(pool));            // dS = A*dP*(X-P)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ds[(lj * 4 + r) * XS + col] = ds[r];
        }
        __builtin_amdgcn_wave_barrier();
        // dS in A layout: out to HBM for the weight gradient, and the A operand of dX += dS.W
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            const float4 da = *reinterpret_cast<const float4*>(Ds + li * XS + 16 * c + 4 * lj);
            rl_stx4<GB>(p.dS_out, (pt * 16 + li) * D + 16 * c + 4 * lj, da);
            bf16x4 ah, al;
            split4(da, ah, al);
            // B fragment of column block nb: W[o = 16c + 4lj + j][nb*16 + li], j = 0..3 - four rows of one column.
            // ds_read_b64_tr_b16 does that gather: per 16-lane group it reads a 4-row x 16-column block of 16-bit
            // elements (lane 4q+p supplies the address of row q, columns 4p..4p+3) and hands lane i column i.
            const int tq = li >> 2, tp = li & 3;
            const __bf16* trh = Wh + (16 * c + 4 * lj + tq) * XSB + 4 * tp;
            const __bf16* trl = Wl + (16 * c + 4 * lj + tq) * XSB + 4 * tp;
#pragma unroll
            for (int n0 = 0; n0 < DT; n0 += 4) {
                bf16x4 bh[4], bl[4];
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    bh[j4] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(trh + (n0 + j4) * 16)));
                    if constexpr (TERMS == 3)
                        bl[j4] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(trl + (n0 + j4) * 16)));
                }
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) dx[n0 + j4] = mfma16(ah, bh[j4], dx[n0 + j4]);
                if constexpr (TERMS == 3) {
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) dx[n0 + j4] = mfma16(ah, bl[j4], dx[n0 + j4]);
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) dx[n0 + j4] = mfma16(al, bh[j4], dx[n0 + j4]);
                }
            }
        }
        // outputs of dX: columns < H -> rpe-branch gradient, columns >= H -> gradient of this slot's gathered row.  The finished
        // tile (C layout: a lane holds four ROWS of one column) goes through the wavefront's tile - dS is dead by now - and
        // leaves in A layout: 16-byte stores, 256 contiguous bytes per row and tensor (round 4; the element form - 32 4-byte
        // stores per lane and point, and 16 4-byte read-modify-writes when the launch adds to GU - made the accumulating
        // launch 212 us against 145)
        loads_landed();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nb = 0; nb < DT; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ds[(lj * 4 + r) * XS + nb * 16 + li] = dx[nb][r];
        __builtin_amdgcn_wave_barrier();
        const long orow = (pt * 16 + li) * H;
        float4 gold[DT / 2];
        if (p.gu_accumulate) {
#pragma unroll
            for (int c = 0; c < DT / 2; ++c) gold[c] = *reinterpret_cast<const float4*>(p.GU + orow + 16 * c + 4 * lj);
        }
#pragma unroll
        for (int c = DT / 2; c < DT; ++c)      // (the gathered half first: its stores go out while the GU rows arrive)
            rl_stx4<GB>(p.DG, orow + 16 * c + 4 * lj - H, *reinterpret_cast<const float4*>(Ds + li * XS + 16 * c + 4 * lj));
#pragma unroll
        for (int c = 0; c < DT / 2; ++c) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Ds + li * XS + 16 * c + 4 * lj);
            if (p.gu_accumulate) v += v4(gold[c]);
            *reinterpret_cast<float4*>(p.GU + orow + 16 * c + 4 * lj) = f4(v);
        }
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) gp[nb] = gp_nxt[nb];
        __builtin_amdgcn_wave_barrier();
    }
}

// Backward of a virtual rpe stage (mlp_rpe1 / mlp_rpe2 + BatchNorm + ReLU whose output was never stored).  G holds the
// gradient w.r.t. the ACTIVATED stage output ((points*16) x H, written by the pooling backward kernels); the raw tile
// is recomputed per point.  With g = G*[act > 0] and xhat = (raw - mean)*invstd:
//   vrpe_bn_reduce_kernel : per-workgroup partials of sum g and sum g*xhat (doubles)            -> rl_bn_bwd_finalize
//   vrpe_wgrad_kernel     : dY = scale*(g - coef0 - xhat*coef1);  dW += dY^T . input (rpe rows / activated stage 1),
//                          db += sum dY (one partial slab per workgroup, the layout rl_wgrad_reduce_batch sums);
//                          stage 2 also stores dY . W2 = the gradient w.r.t. the activated stage-1 output (GU1).
// ---------------------------------------------------------------------------------------------------------------
struct RpeBwdParams {
    PoolParams pp;
    const float* G;       // (P*16, H) gradient w.r.t. the activated output of stage pp.src
    double* stats;        // reduce: [grid][2][H]
    const float* coef;    // wgrad: 2H floats (mean g, mean g*xhat)
    float* slab;          // wgrad: [grid][H*Kin + H]
    float* GU1;           // wgrad stage 2: (P*16, H) out
    int g_bf16;           // G and GU1 are stored as bf16
};

// ===============================================================================================================
// The virtual rpe branch in the TRANSPOSED orientation (round 4).  The kernels above build a point's rpe tile in C layout
// (the MFMA output of rows = slots, columns = channels) and carry it through LDS into the A layout the score GEMM wants -
// per stage four masked ds_write_b32, a wave barrier, a masked ds_read_b128, a lane-divergent region each.  With the MFMA
// operands EXCHANGED - weights as the A operand, the point's inputs as the B operand - the product comes out transposed:
//     lane (slot = l & 15, q = l >> 4), register r  ->  channel 16*nb + 4*q + r of that slot
// which IS the A layout (lane (slot, lj) holds channels 16c + 4lj .. +3): the rpe half of X is born where the score GEMM
// reads it, stage 2 consumes stage 1 straight from registers, and the BatchNorm sums / statistics of the branch accumulate
// as float4s per lane.  Nothing of the branch goes through LDS, and nothing is lane-divergent:
//   * a lane loads ONE coordinate component of its point and of its neighbour (input lj of slot li), not two float4s;
//   * conv biases are folded into lane constants (shift' = b*scale + shift, mean' = mean - b; the statistics kernels add
//     the bias to their sums at the very end) - the raw tile is W.x without the bias everywhere, the same expression in
//     every kernel, so that forward and backward agree on every activation's sign;
//   * d = 16 (the tile's ONE 16-column chunk is half rpe, half gathered): lanes lj < 2 "gather" 16 bytes of zeros through a
//     per-lane base pointer / stride, and the gathered rows arrive as the C operand of the rpe MFMA - one mixed raw tile.
// The softmax / pooling reductions over the four lane groups use v_permlane16_swap / v_permlane32_swap (gfx950) instead of
// ds_bpermute round trips.
// ===============================================================================================================
// sum over the 16 lanes of a lane group (end-of-kernel reductions of the A-layout accumulators)
__device__ __forceinline__ float li_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

// bf16 head / tail of an A-layout tile as MFMA fragments: split ONCE, used by every product the tile enters
template <int DT> struct Frag { bf16x8 h[DT / 2], l[DT / 2]; };
template <> struct Frag<1> { bf16x4 h[1], l[1]; };
template <int DT>
__device__ __forceinline__ void split_tile(const float4 (&a)[DT], Frag<DT>& f) {
    if constexpr (DT == 1) split4(a[0], f.h[0], f.l[0]);
    else {
#pragma unroll
        for (int b = 0; b < DT / 2; ++b) {
            bf16x4 h0, l0, h1, l1;
            split4(a[2 * b], h0, l0);
            split4(a[2 * b + 1], h1, l1);
            f.h[b] = cat8(h0, h1);
            f.l[b] = cat8(l0, l1);
        }
    }
}
// acc[nb] += tile . B with B given transposed in LDS as bf16 head / tail planes ([n][k], row stride STR); NPU of the tile's
// k pairs (32 k each; DT == 1: its one 16-k block) enter.  SWAP exchanges the MFMA operands: acc then holds the TRANSPOSED
// product - lane (row, q) gets columns 16nb + 4q .. +3 of its row, i.e. the A layout.
template <int DT, int NB, int NPU, bool SWAP>
__device__ __forceinline__ void gemm_frag(const Frag<DT>& f, const __bf16* Bh, const __bf16* Bl, int STR, int li, int lj,
                                          f32x4 (&acc)[NB]) {
    if constexpr (DT == 1) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int o = (nb * 16 + li) * STR + 4 * lj;
            const bf16x4 bh = *reinterpret_cast<const bf16x4*>(Bh + o);
            const bf16x4 bl = *reinterpret_cast<const bf16x4*>(Bl + o);
            if constexpr (SWAP) {
                acc[nb] = mfma16(bh, f.h[0], acc[nb]);
                acc[nb] = mfma16(bl, f.h[0], acc[nb]);
                acc[nb] = mfma16(bh, f.l[0], acc[nb]);
            } else {
                acc[nb] = mfma16(f.h[0], bh, acc[nb]);
                acc[nb] = mfma16(f.h[0], bl, acc[nb]);
                acc[nb] = mfma16(f.l[0], bh, acc[nb]);
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < NPU; ++b) {
            constexpr int NG = NB < 4 ? NB : 4;      // column blocks per round: bounds the fragment registers
#pragma unroll
            for (int n0 = 0; n0 < NB; n0 += NG) {
                bf16x8 bh[NG], bl[NG];
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int o = ((n0 + j) * 16 + li) * STR + 32 * b + 4 * lj;
                    bh[j] = cat8(*reinterpret_cast<const bf16x4*>(Bh + o), *reinterpret_cast<const bf16x4*>(Bh + o + 16));
                    bl[j] = cat8(*reinterpret_cast<const bf16x4*>(Bl + o), *reinterpret_cast<const bf16x4*>(Bl + o + 16));
                }
                if constexpr (SWAP) {
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], f.h[b], acc[n0 + j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], f.h[b], acc[n0 + j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], f.l[b], acc[n0 + j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.h[b], bh[j], acc[n0 + j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.h[b], bl[j], acc[n0 + j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NG; ++j) acc[n0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.l[b], bh[j], acc[n0 + j], 0, 0, 0);
                }
            }
        }
    }
}
// the same in exact fp32 (v_mfma_f32_16x16x4_f32): NC of the tile's 16-k chunks against a float image ([n][k], stride STR)
template <int DT, int NC, int NB, bool SWAP>
__device__ __forceinline__ void gemm_f32(const float4 (&a)[DT], const float* Bt, int STR, int li, int lj, f32x4 (&acc)[NB]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float av[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float4 b = *reinterpret_cast<const float4*>(Bt + (nb * 16 + li) * STR + 16 * c + 4 * lj);
            const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[nb] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(bv[s], av[s], acc[nb], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc[nb], 0, 0, 0);
        }
    }
}

// geometry of the branch for a level of width d = 16*DT
template <int DT>
struct VX {
    static constexpr int H = 8 * DT;                  // channels of the rpe branch
    static constexpr int NCH = DT == 1 ? 1 : DT / 2;  // A-layout chunks (= 16-row blocks of the transposed products) that hold them
    static constexpr int HP = 16 * NCH;
    static constexpr int KP2 = DT == 1 ? 16 : 32;     // k extent of the stage-2 product as the MFMA sees it (zero-padded W2)
    static constexpr int S2B = KP2 + 8;               // row strides of the W2 image: bf16 planes / floats
    static constexpr int S2F = KP2 + 4;
    static constexpr int NC2 = DT == 4 ? 2 : 1;       // A-layout chunks of the stage-1 tile that enter the stage-2 product
    static constexpr bool INLDS = DT >= 4;            // per-channel constants: LDS float4s instead of registers
    enum { S1 = 0, H1, S2, H2, MU, IS, NCONST };      // folded BatchNorm of stage 1 / 2 (shift' carries the conv bias), mean' / invstd of the stage being differentiated
};

// what a lane needs of one point: input lj of slot li for the first stage - ONE component of the point and of its neighbour
struct RpeIn2 {
    float pc, nc, dd;
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(pc), "+v"(nc), "+v"(dd)); }
};
// ---- addressing of the per-point operands: buffer loads / stores -----------------------------------------------------------------
// The tile kernels are bound by vector-instruction issue (profiles/r04_pmc_sq_tiles.md: the vector ALU busy in ~80 % of a
// SIMD's cycles), and a fifth of that was 64-bit address arithmetic (v_lshl_add_u64 / v_mad_u64_u32: half / quarter rate).  Every
// per-point operand is therefore addressed as  descriptor (SGPRs, from the kernel arguments) + wavefront-uniform byte offset
// (SGPR, scalar unit) + per-lane byte offset (ONE 32-bit VGPR, mostly a lane constant): no vector instruction for most
// addresses, one v_lshl_add_u32 / v_mad_u32_u24 for a gather.  A lane that must not take part in an access gets a per-lane
// offset beyond the descriptor's 2 GB extent: the hardware returns zeros for such a load and drops such a store - the d = 16
// tile's "zeros in the rpe lanes, gathered rows in the others" and its GU / DG split need no exec masking.
// (Every tensor addressed this way must be smaller than 2 GB: checked on the host.)
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
constexpr unsigned RL_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* ptr) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ float bld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// four consecutive elements of a row tensor stored as fp32 or bf16 (offsets in BYTES of that storage)
template <bool BF>
__device__ __forceinline__ float4 bldrow4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (BF) {
        const rl_bf16x4 h = __builtin_bit_cast(rl_bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else return bld4(r, voff, soff);
}
// ... stored, non-temporal (read once, by a later kernel: see RL_ST4) unless RL_POOL_PLAIN_STORES.  The wavefront-uniform byte
// offset goes into the descriptor's BASE (scalar adds), not into the instruction's SGPR offset: a 16-byte buffer store with an
// SGPR offset, followed at once by a vector instruction that rewrites its data registers, stored the NEW values on this
// hardware (seen as wrong DG rows in a few hundred of 8192 points, different ones every run, when two workgroups shared a CU) -
// hipcc 7.2 inserts the wait state this needs only for stores WITHOUT an SGPR offset (its hazard recogniser follows the
// older ISA manuals' exception for "BUFFER_STORE that use an SGPR for offset").
template <bool BF>
__device__ __forceinline__ void bstrow4(const float4 v, const float* base, unsigned voff, unsigned soff_bytes) {
#ifdef RL_POOL_PLAIN_STORES
    constexpr int AUX = 0;
#else
    constexpr int AUX = 2;      // nt
#endif
    const __amdgpu_buffer_rsrc_t r = rsrc_of(reinterpret_cast<const char*>(base) + soff_bytes);
    if constexpr (BF) {
        rl_bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, h), r, voff, 0, AUX);
    } else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, voff, 0, AUX);
}
// neighbour index of slot li of point pt
__device__ __forceinline__ int ld_idx(const PoolParams& p, long pt, int li) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc_of(p.idx), (unsigned)li * 4u, (unsigned)(pt * 64), 0);
}
// comp = lj < 3 ? lj : 0 (lane constant; lane group 3 supplies the distance, its coordinate loads are dummies)
__device__ __forceinline__ void fetch_rpe2(const PoolParams& p, const Cursor& cu, int li, int comp, int nbr, RpeIn2& r) {
    const unsigned w4 = (unsigned)p.xyz_w * 4u;
    const unsigned cloud = (unsigned)((long)cu.b * p.xyz_bstride) * w4;     // byte offset of the cloud: scalar unit
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(p.xyz);
    r.dd = bld1(rsrc_of(p.nbr_d2), (unsigned)li * 4u, (unsigned)(cu.pt * 64));      // (streaming loads before the gather: see the load groups of the kernels)
    r.pc = bld1(rx, (unsigned)comp * 4u, cloud + (unsigned)cu.i * w4);
    r.nc = bld1(rx, __umul24((unsigned)nbr, w4) + (unsigned)comp * 4u, cloud);      // points per cloud < 2^24 (checked on the host)
}

// per-lane byte offset of a lane's four channels inside the 16 rows x H block of one point of a (points*16) x H row tensor
// (chunk 0; chunk c adds 16*c elements), and the wavefront-uniform offset of the point; d = 16: RL_OOB in the lanes of the
// gathered half (loads give zeros there, stores are dropped)
template <int DT, bool GB>
__device__ __forceinline__ unsigned row_voff(int li, int lj) {
    constexpr unsigned ES = GB ? 2 : 4;
    return (DT == 1 && lj >= 2) ? RL_OOB : ((unsigned)li * VX<DT>::H + 4u * lj) * ES;
}
template <int DT, bool GB>
__device__ __forceinline__ unsigned row_soff(long pt) { return (unsigned)(pt * (16 * VX<DT>::H * (GB ? 2 : 4))); }
// the rpe chunks of such a tensor for one point, A layout
template <int DT, bool GB>
__device__ __forceinline__ void load_g4(const float* G, long pt, int li, int lj, float4 (&g)[VX<DT>::NCH]) {
    const __amdgpu_buffer_rsrc_t rg = rsrc_of(G);
    const unsigned vo = row_voff<DT, GB>(li, lj), so = row_soff<DT, GB>(pt);
#pragma unroll
    for (int c = 0; c < VX<DT>::NCH; ++c) g[c] = bldrow4<GB>(rg, vo + 16u * c * (GB ? 2 : 4), so);
}

// LDS image of W2 (h x h, zero-padded to HP x KP2), [n][k]: the A operand of the transposed stage-2 product
template <int DT, int TERMS>
struct VW2 {
    static constexpr int HP = VX<DT>::HP, KP2 = VX<DT>::KP2;
    static constexpr int BYTES = TERMS == 0 ? HP * VX<DT>::S2F * 4 : 2 * HP * VX<DT>::S2B * 2;
    float* f;
    __bf16 *h, *l;
    __device__ __forceinline__ void bind(unsigned char* mem) {
        f = reinterpret_cast<float*>(mem);
        h = reinterpret_cast<__bf16*>(mem);
        l = h + HP * VX<DT>::S2B;
    }
    // W[n][k] = src[n * ns + k * ks]  (ns = H, ks = 1: W2 itself; ns = 1, ks = H: its transpose, for dY . W2)
    __device__ __forceinline__ void stage(const float* src, int ns, int ks, int nthreads) {
        constexpr int H = VX<DT>::H;
        for (int e = threadIdx.x; e < HP * KP2; e += nthreads) {
            const int n = e / KP2, k = e - n * KP2;
            const float w = (n < H && k < H) ? src[n * ns + k * ks] : 0.f;
            if constexpr (TERMS == 0) f[n * VX<DT>::S2F + k] = w;
            else {
                const __bf16 hh = (__bf16)w;
                h[n * VX<DT>::S2B + k] = hh;
                l[n * VX<DT>::S2B + k] = (__bf16)(w - (float)hh);
            }
        }
    }
};

// lane constants of the branch
template <int DT>
struct VLane {
    static constexpr int H = VX<DT>::H, NCH = VX<DT>::NCH, HP = VX<DT>::HP;
    static constexpr bool INLDS = VX<DT>::INLDS;
    float w1a[NCH], w1b[NCH];        // first stage in reduced form: W1.rpe = (Wa + Wb).x_i + (Wc - Wb).(x_i - x_j) + wd.dist (the channels are [x_i, x_j, x_i - x_j, dist] and x_j = x_i - (x_i - x_j)): U = Wa + Wb against x_i, [Wc - Wb | wd] against [x_i - x_j, dist]
    int comp;
    bool l3;
    f32x4 reg[INLDS ? 1 : VX<DT>::NCONST][NCH];
    const float* lds;                // + 4*lj
    // `cl`: NCONST * HP floats of LDS (INLDS only); the caller's barrier after this publishes them.  `stage`: the stage whose
    // mean' / invstd are wanted (backward kernels; 0: none; -1 / -2: the MU slot carries the statistics pivot of stage 1 / 2)
    __device__ __forceinline__ void load(const PoolParams& p, int li, int lj, float* cl, int stage) {
        comp = lj < 3 ? lj : 0;
        l3 = lj == 3;
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) {
            const int ch = nb * 16 + li;
            float a = 0.f, b = 0.f;
            if (ch < H) {
                const float* w = p.W1 + ch * 10;
                if (lj < 3) { a = w[lj] + w[3 + lj]; b = w[6 + lj] - w[3 + lj]; }
                else b = w[9];
            }
            w1a[nb] = a; w1b[nb] = b;
        }
        auto cst = [&](int which, int ch) -> float {
            if (ch >= H) return 0.f;
            switch (which) {
            case VX<DT>::S1: return p.sc1 ? p.sc1[ch] : 0.f;
            case VX<DT>::H1: return p.sc1 ? __builtin_fmaf(p.b1[ch], p.sc1[ch], p.sh1[ch]) : 0.f;
            case VX<DT>::S2: return (p.sc2 && p.b2) ? p.sc2[ch] : 0.f;
            case VX<DT>::H2: return (p.sc2 && p.b2) ? __builtin_fmaf(p.b2[ch], p.sc2[ch], p.sh2[ch]) : 0.f;
            // forward statistics kernels (stage -1 / -2): the PIVOT of the shifted sums in the bias-free raw tile's terms
            case VX<DT>::MU: return stage == 1 ? p.mu1[ch] - p.b1[ch] : stage == 2 ? p.mu2[ch] - p.b2[ch]
                                  : stage == -1 ? (p.piv1 ? p.piv1[ch] - p.b1[ch] : 0.f) : stage == -2 ? (p.piv2 ? p.piv2[ch] - p.b2[ch] : 0.f) : 0.f;
            default: return stage == 1 ? p.is1[ch] : stage == 2 ? p.is2[ch] : 0.f;
            }
        };
        if constexpr (INLDS) {
            for (int e = threadIdx.x; e < VX<DT>::NCONST * HP; e += blockDim.x) cl[e] = cst(e / HP, e % HP);
            lds = cl + 4 * lj;
        } else {
#pragma unroll
            for (int w = 0; w < VX<DT>::NCONST; ++w)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) reg[w][c][r] = cst(w, 16 * c + 4 * lj + r);
        }
    }
    __device__ __forceinline__ f32x4 get(int which, int c) const {
        if constexpr (INLDS) return *reinterpret_cast<const f32x4*>(lds + which * HP + 16 * c);
        else return reg[which][c];
    }
};

// fold of every A-layout chunk of X for a virtual stage (lane_lazy with the stage's BatchNorm, bias folded, in the rpe half);
// d = 64: read from LDS where used (32 registers otherwise - the 64-channel backward kernel has none to spare)
template <int DT>
struct XFold {
    static constexpr int D = 16 * DT, H = VX<DT>::H;
    static constexpr bool INLDS = DT >= 4;
    static constexpr int LDS_FLOATS = INLDS ? 2 * D : 4;
    f32x4 rsc[INLDS ? 1 : DT], rsh[INLDS ? 1 : DT];
    const float* lds;
    // all threads; `mem`: LDS_FLOATS floats, published by the caller's barrier
    template <int SRC>
    __device__ __forceinline__ void init(const PoolParams& p, int lj, float* mem) {
        const float* s_ = SRC == 1 ? p.sc1 : p.sc2;
        const float* h_ = SRC == 1 ? p.sh1 : p.sh2;
        const float* b_ = SRC == 1 ? p.b1 : p.b2;
        auto fsc = [&](int k) -> float { return k < H ? s_[k] : (p.glazy.scale ? p.glazy.scale[k - H] : 1.f); };
        auto fsh = [&](int k) -> float { return k < H ? __builtin_fmaf(b_[k], s_[k], h_[k]) : (p.glazy.scale ? p.glazy.shift[k - H] : 0.f); };
        if constexpr (INLDS) {
            for (int k = threadIdx.x; k < D; k += blockDim.x) { mem[k] = fsc(k); mem[D + k] = fsh(k); }
            lds = mem + 4 * lj;
        } else {
#pragma unroll
            for (int c = 0; c < DT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { rsc[c][r] = fsc(16 * c + 4 * lj + r); rsh[c][r] = fsh(16 * c + 4 * lj + r); }
        }
    }
    __device__ __forceinline__ f32x4 sc(int c) const {
        if constexpr (INLDS) return *reinterpret_cast<const f32x4*>(lds + 16 * c);
        else return rsc[c];
    }
    __device__ __forceinline__ f32x4 sh(int c) const {
        if constexpr (INLDS) return *reinterpret_cast<const f32x4*>(lds + D + 16 * c);
        else return rsh[c];
    }
};

// raw (bias-free) first-stage tile of one point, A layout: r[nb] = c0[nb] + U . x_i + [V | wd] . [x_i - x_j, dist]
template <int DT>
__device__ __forceinline__ void stage1_raw(const RpeIn2& in, const VLane<DT>& vl, const f32x4 (&c0)[VX<DT>::NCH], f32x4 (&r)[VX<DT>::NCH]) {
    float diff = in.pc - in.nc, dist = vsqrt(in.dd), pc = in.pc;
    asm volatile("" : "+v"(diff), "+v"(dist), "+v"(pc));      // selects on lane-constant masks: without the pin hipcc sinks the square root and the difference into exec-masked branches
    const float a1 = vl.l3 ? dist : diff;
    const float a2 = vl.l3 ? 0.f : pc;
#pragma unroll
    for (int nb = 0; nb < VX<DT>::NCH; ++nb) {
        r[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(vl.w1a[nb], a2, c0[nb], 0, 0, 0);
        r[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(vl.w1b[nb], a1, r[nb], 0, 0, 0);
    }
}
// raw second-stage tile from the ACTIVATED first-stage tile u1 (A layout: chunks 0 .. NC2-1 of a DT-chunk array whose
// other chunks / lanes meet zero rows of the padded W2)
template <int DT, int TERMS>
__device__ __forceinline__ void stage2_raw(const float4 (&u1)[DT], const Frag<DT>* fr, const VW2<DT, TERMS>& w2, int li, int lj,
                                           f32x4 (&r)[VX<DT>::NCH]) {
    if constexpr (TERMS == 0) gemm_f32<DT, VX<DT>::NC2, VX<DT>::NCH, true>(u1, w2.f, VX<DT>::S2F, li, lj, r);
    else gemm_frag<DT, VX<DT>::NCH, 1, true>(*fr, w2.h, w2.l, VX<DT>::S2B, li, lj, r);
}

// X tile of one point, A layout, activated.  graw: the gathered chunks as loaded (DT == 1: the mixed chunk [0 | gathered]);
// xsc / xsh: the fold of every chunk (rpe chunks: the BatchNorm of stage SRC with the bias folded in; gathered: the lazy
// BatchNorm of G); es_g / e0: activation slopes (eff_slope) of the gathered half / per lane for the mixed chunk.
// rawu: the raw tile of stage SRC (the backward's BatchNorm sums need it).  fr: split of xa when SPLIT (bf16 modes).
template <int DT, int TERMS, int SRC>
__device__ __forceinline__ void build_x(const RpeIn2& in, const float4 (&graw)[DT], const VLane<DT>& vl, const VW2<DT, TERMS>& w2,
                                        const XFold<DT>& xf, float es_g, float e0, int li, int lj,
                                        float4 (&xa)[DT], f32x4 (&rawu)[VX<DT>::NCH]) {
    constexpr int NCH = VX<DT>::NCH;
    f32x4 c0[NCH], r1[NCH];
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) c0[nb] = (DT == 1 && SRC == 1) ? v4(graw[0]) : splat(0.f);
    stage1_raw<DT>(in, vl, c0, r1);
    if constexpr (SRC == 1) {
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) rawu[nb] = r1[nb];
    } else {
        float4 u1[DT];
#pragma unroll
        for (int c = 0; c < DT; ++c) u1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) u1[nb] = f4(vrelu(__builtin_elementwise_fma(r1[nb], vl.get(VX<DT>::S1, nb), vl.get(VX<DT>::H1, nb))));
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) rawu[nb] = DT == 1 ? v4(graw[0]) : splat(0.f);
        if constexpr (TERMS == 0) stage2_raw<DT, TERMS>(u1, nullptr, w2, li, lj, rawu);
        else {
            Frag<DT> f1;
            if constexpr (DT == 1) split4(u1[0], f1.h[0], f1.l[0]);
            else {
                bf16x4 h0, l0, h1, l1;
                split4(u1[0], h0, l0);
                split4(u1[1], h1, l1);      // DT == 2: chunk 1 is all zeros (it meets zero rows of W2 anyway)
                f1.h[0] = cat8(h0, h1);
                f1.l[0] = cat8(l0, l1);
            }
            stage2_raw<DT, TERMS>(u1, &f1, w2, li, lj, rawu);
        }
    }
    if constexpr (DT == 1) {
        xa[0] = f4(vact(__builtin_elementwise_fma(rawu[0], xf.sc(0), xf.sh(0)), e0));
    } else {
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            if (c < NCH) xa[c] = f4(vrelu(__builtin_elementwise_fma(rawu[c], xf.sc(c), xf.sh(c))));
            else xa[c] = f4(vact(__builtin_elementwise_fma(v4(graw[c]), xf.sc(c), xf.sh(c)), es_g));
        }
    }
}

// the gathered chunks of a point (d = 16: the mixed chunk - the rpe lanes read beyond the descriptor and get zeros)
template <int DT>
struct GatherG {
    unsigned vcol;      // per-lane byte offset inside a gathered row (+ RL_OOB in the rpe lanes of the mixed chunk)
    __device__ __forceinline__ void init(const PoolParams& p, int lj) {
        constexpr int H = VX<DT>::H;
        if constexpr (DT == 1) vcol = lj < 2 ? RL_OOB : (unsigned)(4 * lj - H) * 4u;
        else vcol = (unsigned)(4 * lj) * 4u;
    }
    __device__ __forceinline__ void fetch(const PoolParams& p, const Cursor& cu, int nbr, float4 (&raw)[DT]) const {
        constexpr int H = VX<DT>::H;
        constexpr unsigned ROWB = H * 4;                                            // bytes of a row of G: 32 / 64 / 128
        const __amdgpu_buffer_rsrc_t rg = rsrc_of(p.G);
        const unsigned cloud = (unsigned)((long)cu.b * p.g_bstride) * ROWB;         // scalar unit
        const unsigned vo = (unsigned)nbr * ROWB + vcol;                            // one v_lshl_add_u32
        if constexpr (DT == 1) raw[0] = bld4(rg, vo, cloud);
        else {
#pragma unroll
            for (int c = VX<DT>::NCH; c < DT; ++c) raw[c] = bld4(rg, vo + (unsigned)(16 * c - H) * 4u, cloud);
        }
    }
};

template <int DT>
__device__ __forceinline__ float lane_slope(int lj, float es_g) { return (DT == 1 && lj < 2) ? 0.f : es_g; }

// partial statistics of a workgroup: per-lane float4 sums (A layout, rpe chunks) -> doubles, bias added back:
//   sum (r + b) = S + cnt*b      sum (r + b)^2 = Q + 2 b S + cnt b^2
// red: NW * 2 * HP doubles of LDS; cnts: NW floats (rows of each wavefront)
template <int DT, int NW>
__device__ __forceinline__ void write_moments(const f32x4 (&ssum)[VX<DT>::NCH], const f32x4 (&ssq)[VX<DT>::NCH], const float* bias,
                                              long rows_w, int wave, int lane, double* red, double* cnts, double* out) {
    constexpr int H = VX<DT>::H, HP = VX<DT>::HP, NCH = VX<DT>::NCH;
    const int li = lane & 15, lj = lane >> 4;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = li_sum(ssum[c][r]), b = li_sum(ssq[c][r]);
            if (li == 0) {
                red[(wave * 2 + 0) * HP + 16 * c + 4 * lj + r] = (double)a;
                red[(wave * 2 + 1) * HP + 16 * c + 4 * lj + r] = (double)b;
            }
        }
    if (lane == 0) cnts[wave] = (double)rows_w;
    __syncthreads();
    if (threadIdx.x < H) {
        const int ch = threadIdx.x;
        double s = 0.0, q = 0.0, cnt = 0.0;
        for (int wv = 0; wv < NW; ++wv) {
            s += red[(wv * 2 + 0) * HP + ch];
            q += red[(wv * 2 + 1) * HP + ch];
            cnt += cnts[wv];
        }
        const double b = bias ? (double)bias[ch] : 0.0;
        out[((long)blockIdx.x * 2 + 0) * H + ch] = s + cnt * b;
        out[((long)blockIdx.x * 2 + 1) * H + ch] = q + 2.0 * b * s + cnt * b * b;
    }
}

// number of points a wavefront visits: pt0, pt0 + pstep, ... < P
__device__ __forceinline__ long wave_points(long pt0, long pstep, long P) { return pt0 < P ? (P - pt0 + pstep - 1) / pstep : 0; }

// one point's prefetched operands (virtual pooling kernels) + the neighbour index of the point that will take this buffer next
template <int DT>
struct PreF {
    float4 graw[DT];
    RpeIn2 rin;
    int idx;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int c = 0; c < DT; ++c) graw[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        rin.pc = rin.nc = rin.dd = 0.f;
        idx = 0;
    }
};
// wait until at most N vector-memory operations are outstanding (they complete in order: the N youngest stay in flight)
template <int N>
__device__ __forceinline__ void loads_landed_but() {
    static_assert(N >= 0 && N < 16, "vmcnt immediate");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70 | N);
    asm volatile("" ::: "memory");
}

template <int DT, int TERMS, int SRC, bool FST>      // SRC 1 / 2: the stage X's rpe half comes from; FST: leave the stage-2 statistics (SRC == 1)
__global__ __launch_bounds__(256) void vpool_fwd_kernel(const PoolParams p) {
    constexpr int NW = 4;
    constexpr int D = Tile<DT>::D, XS = Tile<DT>::XS, XSB = Tile<DT>::XSB, NCH = VX<DT>::NCH, HP = VX<DT>::HP;
    constexpr bool W2ON = SRC == 2 || FST;
    __shared__ __attribute__((aligned(16))) unsigned char wmem[TERMS == 0 ? D * XS * 4 : 2 * D * XSB * 2];
    __shared__ __attribute__((aligned(16))) float Xt[NW][16 * XS];
    __shared__ __attribute__((aligned(16))) unsigned char w2mem[W2ON ? VW2<DT, TERMS>::BYTES : 16];
    __shared__ __attribute__((aligned(16))) float cl[VX<DT>::INLDS ? VX<DT>::NCONST * HP : 4];
    __shared__ __attribute__((aligned(16))) float xfm[XFold<DT>::LDS_FLOATS];
    __shared__ double cnts[NW];
    float* Wt = reinterpret_cast<float*>(wmem);
    __bf16* Wh = reinterpret_cast<__bf16*>(wmem);
    __bf16* Wl = Wh + D * XSB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lj = lane >> 4;
    for (int e = threadIdx.x; e < D * D; e += 64 * NW) {
        const int o = e / D, i = e - o * D;
        const float w = p.W[e];            // [n][k]: the transposed-B form the tile GEMMs read
        if constexpr (TERMS == 0) Wt[o * XS + i] = w;
        else {
            const __bf16 h = (__bf16)w;
            Wh[o * XSB + i] = h;
            Wl[o * XSB + i] = (__bf16)(w - (float)h);
        }
    }
    VW2<DT, TERMS> w2;
    w2.bind(w2mem);
    if constexpr (W2ON) w2.stage(p.W2, VX<DT>::H, 1, 64 * NW);
    VLane<DT> vl;
    vl.load(p, li, lj, cl, FST ? -2 : 0);
    XFold<DT> xf;
    xf.template init<SRC>(p, lj, xfm);
    const float es_g = eff_slope(p.glazy), e0 = lane_slope<DT>(lj, es_g);
    GatherG<DT> gg;
    gg.init(p, lj);
    f32x4 fs2[NCH], fq2[NCH];
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) fs2[nb] = fq2[nb] = splat(0.f);
    __syncthreads();
    float* Xs = Xt[wave];
    // 32-bit point arithmetic (points * 16 < 2^31, checked on the host): a 64-bit ordered compare is a VECTOR instruction
    const PointSpan span = point_span<NW>(p, wave);
    const int P = span.end, pstep = span.step;
    int pt = span.first;
    const int npts = (int)wave_points(pt, pstep, P);
    // Software pipeline TWO points deep (round 4: with ~160 instructions per point left, one point of prefetch is shorter than a
    // gather's latency).  Two buffers alternate: while buffer U's point is computed, U is refilled for the point two further on
    // and the other buffer's loads (issued one iteration ago) are awaited - loads_landed_but<NG> leaves this iteration's group
    // of NG loads in flight.  A buffer carries the neighbour index of ITS NEXT point (loaded two iterations before it is used).
    constexpr int NG = (DT == 4 ? 2 : 1) + 3 + 1;      // loads of one group: gathered chunks, two coordinates + distance, index
    PreF<DT> A, B;
    A.zero(); B.zero();
    Cursor c0, c1, c2;
    c0.start(pt, p.n);
    c1 = c0.next(pstep, p.n);
    c2 = c1.next(pstep, p.n);
    if (pt < P) {
        const bool v1 = pt + pstep < P;
        const int i0 = ld_idx(p, pt, li);
        const int i1 = ld_idx(p, v1 ? pt + pstep : pt, li);
        A.idx = ld_idx(p, pt + 2 * pstep < P ? pt + 2 * pstep : pt, li);
        B.idx = ld_idx(p, pt + 3 * pstep < P ? pt + 3 * pstep : pt, li);
        gg.fetch(p, c0, i0, A.graw);
        fetch_rpe2(p, c0, li, vl.comp, i0, A.rin);
        gg.fetch(p, v1 ? c1 : c0, i1, B.graw);
        fetch_rpe2(p, v1 ? c1 : c0, li, vl.comp, i1, B.rin);
    }
    loads_landed();
    auto body = [&](PreF<DT>& U) {
        float4 xa[DT];
        f32x4 rawu[NCH];
        build_x<DT, TERMS, SRC>(U.rin, U.graw, vl, w2, xf, es_g, e0, li, lj, xa, rawu);
#pragma unroll
        for (int c = 0; c < DT; ++c) *reinterpret_cast<float4*>(Xs + li * XS + 16 * c + 4 * lj) = xa[c];
        // refill U for the point two further on - one group, no branch around it (past the last point the current one is read again)
        {
            const Cursor cf = pt + 2 * pstep < P ? c2 : c0;
            fetch_rpe2(p, cf, li, vl.comp, U.idx, U.rin);
            gg.fetch(p, cf, U.idx, U.graw);
            U.idx = ld_idx(p, pt + 4 * pstep < P ? pt + 4 * pstep : pt, li);
        }
        loads_issued();
        c0 = c1; c1 = c2; c2 = c2.next(pstep, p.n);
        f32x4 s[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) s[nb] = splat(0.f);
        if constexpr (TERMS == 0) {
            if constexpr (FST) {
                f32x4 r2[NCH];        // (the accumulator starts at -pivot: the shifted sums cost nothing)
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) r2[nb] = -vl.get(VX<DT>::MU, nb);
                gemm_f32<DT, VX<DT>::NC2, NCH, true>(xa, w2.f, VX<DT>::S2F, li, lj, r2);
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) { fs2[nb] += r2[nb]; fq2[nb] = __builtin_elementwise_fma(r2[nb], r2[nb], fq2[nb]); }
            }
            gemm_f32<DT, DT, DT, false>(xa, Wt, XS, li, lj, s);
        } else {
            Frag<DT> fx;
            split_tile<DT>(xa, fx);
            if constexpr (FST) {
                // the next stage's raw output (mlp_rpe2 on this tile) only for its BatchNorm batch statistics; the gathered
                // lanes / chunks of X meet zero rows of the padded W2
                f32x4 r2[NCH];        // (the accumulator starts at -pivot: the shifted sums cost nothing)
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) r2[nb] = -vl.get(VX<DT>::MU, nb);
                gemm_frag<DT, NCH, 1, true>(fx, w2.h, w2.l, VX<DT>::S2B, li, lj, r2);
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) { fs2[nb] += r2[nb]; fq2[nb] = __builtin_elementwise_fma(r2[nb], r2[nb], fq2[nb]); }
            }
            gemm_frag<DT, DT, DT / 2, false>(fx, Wh, Wl, XSB, li, lj, s);
        }
        __builtin_amdgcn_wave_barrier();
        // softmax over the 16 slots of a column and the weighted sum: out = (sum e*x) / (sum e), e = 2^((s - max)*log2 e)
        constexpr float LOG2E = 1.44269504088896340736f;
        float outv[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) {
            f32x4 xc;
#pragma unroll
            for (int r = 0; r < 4; ++r) xc[r] = Xs[(lj * 4 + r) * XS + nb * 16 + li];
            const float m = lg_max(vmax2(vmax2(s[nb][0], s[nb][1]), vmax2(s[nb][2], s[nb][3])));
            const f32x4 t = __builtin_elementwise_fma(s[nb], splat(LOG2E), splat(-m * LOG2E));
            f32x4 e;
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(t[r]);
            float den = sum4(e), num = sum4(e * xc);
            lg_sum2(den, num);
            outv[nb] = num * __builtin_amdgcn_rcpf(den);
        }
        loads_landed_but<NG>();
        if constexpr (DT == 4) {
            // every lane holds all four results: lane group q stores column block q - one 256-byte store per point
            const float v = lj == 0 ? outv[0] : lj == 1 ? outv[1] : lj == 2 ? outv[2] : outv[3];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_of(p.Pout), (unsigned)lane * 4u, (unsigned)(pt * (D * 4)), 0);
        } else {
            // lane group nb stores column block nb (the others' offsets lie beyond the descriptor: dropped)
#pragma unroll
            for (int nb = 0; nb < DT; ++nb)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv[nb]), rsrc_of(p.Pout),
                                                      lj == nb ? (unsigned)(nb * 16 + li) * 4u : RL_OOB, (unsigned)(pt * (D * 4)), 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // ONE exit from the loop: with a break after either half the two exits are merged into one block that (as far as the
    // waitcnt pass can tell) also leads back to the loop header - it then waits for the loads just issued at the top of the loop
    for (int k = npts >> 1; k > 0; --k) {
        body(A);
        pt += pstep;
        body(B);
        pt += pstep;
    }
    if (npts & 1) body(A);
    if constexpr (FST) {
        __syncthreads();                                                    // the X tiles are free now
        double* redd = reinterpret_cast<double*>(&Xt[0][0]);                // [NW][2][HP] doubles
        static_assert(NW * 2 * HP * 2 <= NW * 16 * XS, "statistics scratch does not fit the X tiles");
        write_moments<DT, NW>(fs2, fq2, p.piv2 ? nullptr : p.b2, npts * 16, wave, lane, redd, cnts, p.fstats2);
    }
}

// BatchNorm batch statistics of the raw stage-1 / stage-2 tile of the branch (what the GEMM epilogue of mlp_rpe1 / mlp_rpe2 would have left for rl_bn_finalize), transposed orientation
template <int DT, int TERMS, int SRC>
__global__ __launch_bounds__(256) void vrpe_stats_kernel(const PoolParams p, double* __restrict__ stats) {
    constexpr int NW = 4, NCH = VX<DT>::NCH, HP = VX<DT>::HP;
    __shared__ __attribute__((aligned(16))) unsigned char w2mem[SRC == 2 ? VW2<DT, TERMS>::BYTES : 16];
    __shared__ __attribute__((aligned(16))) float cl[VX<DT>::INLDS ? VX<DT>::NCONST * HP : 4];
    __shared__ double red[NW * 2 * HP];
    __shared__ double cnts[NW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lj = lane >> 4;
    VW2<DT, TERMS> w2;
    w2.bind(w2mem);
    if constexpr (SRC == 2) w2.stage(p.W2, VX<DT>::H, 1, 64 * NW);
    VLane<DT> vl;
    vl.load(p, li, lj, cl, SRC == 1 ? -1 : -2);       // MU slot: the pivot of the shifted sums (0 without one)
    __syncthreads();
    f32x4 ssum[NCH], ssq[NCH];
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) ssum[nb] = ssq[nb] = splat(0.f);
    // 32-bit point arithmetic (points * 16 < 2^31, checked on the host): a 64-bit ordered compare is a VECTOR instruction
    const PointSpan span = point_span<NW>(p, wave);
    const int P = span.end, pstep = span.step;
    int pt = span.first;
    const int npts = (int)wave_points(pt, pstep, P);
    // two points deep, two alternating buffers (see vpool_fwd_kernel): coordinates + distance of a point, and the neighbour
    // index of the point that takes the buffer next
    constexpr int NG = 3 + 1;
    struct Buf { RpeIn2 rin; int idx; } A = {{0.f, 0.f, 0.f}, 0}, B = {{0.f, 0.f, 0.f}, 0};
    Cursor c0, c1, c2;
    c0.start(pt, p.n);
    c1 = c0.next(pstep, p.n);
    c2 = c1.next(pstep, p.n);
    if (pt < P) {
        const bool v1 = pt + pstep < P;
        const int i0 = ld_idx(p, pt, li);
        const int i1 = ld_idx(p, v1 ? pt + pstep : pt, li);
        A.idx = ld_idx(p, pt + 2 * pstep < P ? pt + 2 * pstep : pt, li);
        B.idx = ld_idx(p, pt + 3 * pstep < P ? pt + 3 * pstep : pt, li);
        fetch_rpe2(p, c0, li, vl.comp, i0, A.rin);
        fetch_rpe2(p, v1 ? c1 : c0, li, vl.comp, i1, B.rin);
    }
    loads_landed();
    auto body = [&](Buf& U) {
        f32x4 c0v[NCH], raw[NCH];       // the product's accumulator starts at -pivot (stage 1 here, stage 2 below)
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) c0v[nb] = SRC == 1 ? -vl.get(VX<DT>::MU, nb) : splat(0.f);
        stage1_raw<DT>(U.rin, vl, c0v, raw);
        {
            const Cursor cf = pt + 2 * pstep < P ? c2 : c0;      // (no branch: see pool_fwd_kernel)
            fetch_rpe2(p, cf, li, vl.comp, U.idx, U.rin);
            U.idx = ld_idx(p, pt + 4 * pstep < P ? pt + 4 * pstep : pt, li);
        }
        loads_issued();
        c0 = c1; c1 = c2; c2 = c2.next(pstep, p.n);
        if constexpr (SRC == 2) {
            float4 u1[DT];
#pragma unroll
            for (int c = 0; c < DT; ++c) u1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int nb = 0; nb < NCH; ++nb) {
                u1[nb] = f4(vrelu(__builtin_elementwise_fma(raw[nb], vl.get(VX<DT>::S1, nb), vl.get(VX<DT>::H1, nb))));
                raw[nb] = -vl.get(VX<DT>::MU, nb);
            }
            if constexpr (TERMS == 0) stage2_raw<DT, TERMS>(u1, nullptr, w2, li, lj, raw);
            else {
                Frag<DT> f1;
                if constexpr (DT == 1) split4(u1[0], f1.h[0], f1.l[0]);
                else {
                    bf16x4 h0, l0, h1, l1;
                    split4(u1[0], h0, l0);
                    split4(u1[1], h1, l1);
                    f1.h[0] = cat8(h0, h1);
                    f1.l[0] = cat8(l0, l1);
                }
                stage2_raw<DT, TERMS>(u1, &f1, w2, li, lj, raw);
            }
        }
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) {
            ssum[nb] += raw[nb];
            ssq[nb] = __builtin_elementwise_fma(raw[nb], raw[nb], ssq[nb]);
        }
        loads_landed_but<NG>();
    };
    for (int k = npts >> 1; k > 0; --k) {      // (one exit: see vpool_fwd_kernel)
        body(A);
        pt += pstep;
        body(B);
        pt += pstep;
    }
    if (npts & 1) body(A);
    write_moments<DT, NW>(ssum, ssq, (SRC == 1 ? p.piv1 : p.piv2) ? nullptr : (SRC == 1 ? p.b1 : p.b2), npts * 16, wave, lane, red, cnts, stats);
}


// the backward's buffer: + dP of the point and its GU rows when the launch accumulates
template <int DT, bool ACC>
struct PreB {
    PreF<DT> f;
    float gp[DT];
    float4 gacc[ACC ? VX<DT>::NCH : 1];
    __device__ __forceinline__ void zero() {
        f.zero();
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) gp[nb] = 0.f;
#pragma unroll
        for (int c = 0; c < (ACC ? VX<DT>::NCH : 1); ++c) gacc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
};

// Backward of the fused pooling block with a virtual rpe stage (pool_bwd_kernel's VIRT case, transposed orientation).
// ACC: this launch adds to GU; p.bstats: it completes GU and leaves the BatchNorm-backward sums of stage SRC; GB: GU / DG are bf16.
// The finished dX tile always leaves through the wavefront's dS tile in A layout (16-byte stores; the BatchNorm sums are
// taken there, against the A-layout raw tile).
template <int DT, int TERMS, int SRC, int NW, bool GB, bool ACC>
__global__ __launch_bounds__(64 * NW) void vpool_bwd_kernel(const PoolParams p) {
    const bool BST = p.bstats != nullptr;      // wavefront-uniform: one scalar branch per point
    constexpr int D = Tile<DT>::D, H = Tile<DT>::H, XS = Tile<DT>::XS, XSB = Tile<DT>::XSB, NCH = VX<DT>::NCH, HP = VX<DT>::HP;
    __shared__ __attribute__((aligned(16))) unsigned char w2mem[SRC == 2 ? VW2<DT, TERMS>::BYTES : 16];
    __shared__ __attribute__((aligned(16))) float cl[VX<DT>::INLDS ? VX<DT>::NCONST * HP : 4];
    __shared__ __attribute__((aligned(16))) float xfm[XFold<DT>::LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float Wmem[TERMS == 0 ? 2 * D * XS : 2 * D * XSB];   // bf16: 4 arrays of D*XSB
    __shared__ __attribute__((aligned(16))) float Tiles[NW][2][16 * XS];
    float* Wt = Wmem;
    float* Wn = Wmem + D * XS;
    __bf16* Wnh = reinterpret_cast<__bf16*>(Wmem);      // [n][k] head / tail: S = X.W^T
    __bf16* Wnl = Wnh + D * XSB;
    __bf16* Wth = Wnl + D * XSB;                          // [n'][k'] = W[k'][n'] head / tail: dX = dS.W
    __bf16* Wtl = Wth + D * XSB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lj = lane >> 4;
    if constexpr (TERMS == 0) {
        stage_w<DT>(p, Wt, Wn, 64 * NW);
    } else {
        for (int e = threadIdx.x; e < D * D; e += 64 * NW) {
            const int o = e / D, i = e - o * D;
            const float w = p.W[e];
            const __bf16 h = (__bf16)w, l = (__bf16)(w - (float)h);
            Wnh[o * XSB + i] = h; Wnl[o * XSB + i] = l;
            Wth[i * XSB + o] = h; Wtl[i * XSB + o] = l;
        }
    }
    VW2<DT, TERMS> w2;
    w2.bind(w2mem);
    if constexpr (SRC == 2) w2.stage(p.W2, VX<DT>::H, 1, 64 * NW);
    VLane<DT> vl;
    vl.load(p, li, lj, cl, BST ? SRC : 0);
    XFold<DT> xf;
    xf.template init<SRC>(p, lj, xfm);
    const float es_g = eff_slope(p.glazy), e0 = lane_slope<DT>(lj, es_g);
    GatherG<DT> gg;
    gg.init(p, lj);
    // d = 16: the one output chunk is GU in lanes lj < 2 and DG in the others - per-lane base pointer and column; the GU
    // this launch adds to is read the same way (the other lanes read 16 bytes of zeros)
    constexpr unsigned ES = GB ? 2 : 4;
    const unsigned v_gu = row_voff<DT, GB>(li, lj);                                                   // rpe lanes / chunks -> GU
    const unsigned v_dg = DT == 1 ? (lj >= 2 ? ((unsigned)li * H + 4u * lj - H) * ES : RL_OOB)        // gathered lanes / chunks -> DG
                                  : ((unsigned)li * H + 4u * lj) * ES;
    f32x4 bsg[NCH], bsx[NCH];
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) bsg[nb] = bsx[nb] = splat(0.f);
    __syncthreads();
    float* Xs = Tiles[wave][0];
    float* Ds = Tiles[wave][1];
    f32x4 accw[DT][DT];
#pragma unroll
    for (int nb = 0; nb < DT; ++nb)
#pragma unroll
        for (int kb = 0; kb < DT; ++kb) accw[nb][kb] = splat(0.f);

    // 32-bit point arithmetic (points * 16 < 2^31, checked on the host): a 64-bit ordered compare is a VECTOR instruction
    const PointSpan span = point_span<NW>(p, wave);
    const int P = span.end, pstep = span.step;
    int pt = span.first;
    const int npts = (int)wave_points(pt, pstep, P);
    // Software pipeline two points deep (see vpool_fwd_kernel): a buffer holds a point's gathered rows, coordinates, dP and - when
    // this launch adds to GU - its GU rows, plus the neighbour index of the point that takes the buffer next.
    constexpr int NG = (DT == 4 ? 2 : 1) + 3 + 1 + DT + (ACC ? NCH : 0);      // loads of one group
#ifndef RL_BWD_AHEAD16
#define RL_BWD_AHEAD16 2
#endif
    constexpr int AHEAD = DT >= 4 ? 1 : RL_BWD_AHEAD16;      // d = 64: no registers for a second buffer (256 per lane with 8 wavefronts per workgroup) - one point ahead
    PreB<DT, ACC> A, B;
    A.zero(); B.zero();
    Cursor c0, c1, c2;
    c0.start(pt, p.n);
    c1 = c0.next(pstep, p.n);
    c2 = c1.next(pstep, p.n);
    auto fill = [&](PreB<DT, ACC>& U, const Cursor& cf, int nbr) {
        // (dP first: should the register allocator rotate a buffer register with a copy at the back edge, the copy waits for the
        // OLDEST load of the group - a streaming one - not for the gathers)
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) U.gp[nb] = bld1(rsrc_of(p.dP), (unsigned)(nb * 16 + li) * 4u, (unsigned)(cf.pt * (D * 4)));
        if constexpr (ACC) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) U.gacc[c] = bldrow4<GB>(rsrc_of(p.GU), v_gu + 16u * c * ES, row_soff<DT, GB>(cf.pt));
        }
        fetch_rpe2(p, cf, li, vl.comp, nbr, U.f.rin);
        gg.fetch(p, cf, nbr, U.f.graw);
    };
    if (pt < P) {
        const int i0 = ld_idx(p, pt, li);
        if constexpr (AHEAD == 2) {
            const bool v1 = pt + pstep < P;
            const int i1 = ld_idx(p, v1 ? pt + pstep : pt, li);
            A.f.idx = ld_idx(p, pt + 2 * pstep < P ? pt + 2 * pstep : pt, li);
            B.f.idx = ld_idx(p, pt + 3 * pstep < P ? pt + 3 * pstep : pt, li);
            fill(A, c0, i0);
            fill(B, v1 ? c1 : c0, i1);
        } else {
            A.f.idx = ld_idx(p, pt + pstep < P ? pt + pstep : pt, li);
            fill(A, c0, i0);
        }
    }
    loads_landed();
    auto body = [&](PreB<DT, ACC>& U) {
        float4 xa[DT];
        f32x4 rawu[NCH];
        build_x<DT, TERMS, SRC>(U.f.rin, U.f.graw, vl, w2, xf, es_g, e0, li, lj, xa, rawu);
#pragma unroll
        for (int c = 0; c < DT; ++c) *reinterpret_cast<float4*>(Xs + li * XS + 16 * c + 4 * lj) = xa[c];
        // dP and the GU rows are used AFTER the refill below: real copies, made here (the empty asm "redefines" each one), so
        // that the buffer's registers are dead when its loads are issued and the loads can land in them - otherwise the
        // loop-carried copies are made at the back edge, with a wait for the loads just issued
        float gpv[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) { gpv[nb] = U.gp[nb]; asm volatile("" : "+v"(gpv[nb])); }
        float4 gaccv[NCH];
        if constexpr (ACC) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                gaccv[c] = U.gacc[c];
                asm volatile("" : "+v"(gaccv[c].x), "+v"(gaccv[c].y), "+v"(gaccv[c].z), "+v"(gaccv[c].w));
            }
        }
        // refill U for the point two further on - one group, no branch around it (past the last point the current one is read again)
        {
            const Cursor cf = pt + AHEAD * pstep < P ? (AHEAD == 2 ? c2 : c1) : c0;
            fill(U, cf, U.f.idx);
            U.f.idx = ld_idx(p, pt + 2 * AHEAD * pstep < P ? pt + 2 * AHEAD * pstep : pt, li);
        }
        loads_issued();
        c0 = c1; c1 = c2; c2 = c2.next(pstep, p.n);
        f32x4 a[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) a[nb] = splat(0.f);
        if constexpr (TERMS == 0) gemm_f32<DT, DT, DT, false>(xa, Wn, XS, li, lj, a);
        else {
            Frag<DT> fx;
            split_tile<DT>(xa, fx);
            gemm_frag<DT, DT, DT / 2, false>(fx, Wnh, Wnl, XSB, li, lj, a);
        }
        __builtin_amdgcn_wave_barrier();
        // C-layout pass: softmax over the slots, P, dS (to LDS), dP*A kept in registers as the start of dX
        constexpr float LOG2E = 1.44269504088896340736f;
        f32x4 dx[DT], dsr[DT];
        bf16x4 dsh[DT], dsl[DT], xch[DT], xcl[DT];
#pragma unroll
        for (int nb = 0; nb < DT; ++nb) {
            const int col = nb * 16 + li;
            f32x4 xc;
#pragma unroll
            for (int r = 0; r < 4; ++r) xc[r] = Xs[(lj * 4 + r) * XS + col];
            const float m = lg_max(vmax2(vmax2(a[nb][0], a[nb][1]), vmax2(a[nb][2], a[nb][3])));
            const f32x4 t = __builtin_elementwise_fma(a[nb], splat(LOG2E), splat(-m * LOG2E));
            f32x4 e;
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(t[r]);
            float den = sum4(e), num = sum4(e * xc);
            lg_sum2(den, num);
            const float inv = __builtin_amdgcn_rcpf(den);
            const float pool = num * inv;
            dx[nb] = e * splat(inv * gpv[nb]);                        // direct path dP*A
            dsr[nb] = dx[nb] * (xc - splat(pool));                   // dS = A*dP*(X-P)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ds[(lj * 4 + r) * XS + col] = dsr[nb][r];
            if constexpr (TERMS != 0) {
                // dW[n][k] += sum_rows dS[row][n] * X[row][k]: the lane's four reduction rows are its C-layout registers of
                // dS (A operand) and of X (B operand) - split here, once, while xc is at hand
                split4(f4(dsr[nb]), dsh[nb], dsl[nb]);
                split4(f4(xc), xch[nb], xcl[nb]);
            }
        }
        if constexpr (TERMS != 0) {
#pragma unroll
            for (int nb = 0; nb < DT; ++nb) {
#pragma unroll
                for (int kb = 0; kb < DT; ++kb) accw[nb][kb] = mfma16(dsh[nb], xch[kb], accw[nb][kb]);
#pragma unroll
                for (int kb = 0; kb < DT; ++kb) accw[nb][kb] = mfma16(dsh[nb], xcl[kb], accw[nb][kb]);
#pragma unroll
                for (int kb = 0; kb < DT; ++kb) accw[nb][kb] = mfma16(dsl[nb], xch[kb], accw[nb][kb]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // dX += dS . W   (dS re-read in A layout)
        float4 da[DT];
#pragma unroll
        for (int c = 0; c < DT; ++c) da[c] = *reinterpret_cast<const float4*>(Ds + li * XS + 16 * c + 4 * lj);
        if constexpr (TERMS == 0) {
            gemm_f32<DT, DT, DT, false>(da, Wt, XS, li, lj, dx);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float bx[DT];
#pragma unroll
                for (int kb = 0; kb < DT; ++kb) bx[kb] = Xs[(4 * t + lj) * XS + kb * 16 + li];
#pragma unroll
                for (int nb = 0; nb < DT; ++nb) {
                    const float ad = Ds[(4 * t + lj) * XS + nb * 16 + li];
#pragma unroll
                    for (int kb = 0; kb < DT; ++kb)
                        accw[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad, bx[kb], accw[nb][kb], 0, 0, 0);
                }
            }
        } else {
            Frag<DT> fd;
            split_tile<DT>(da, fd);
            gemm_frag<DT, DT, DT / 2, false>(fd, Wth, Wtl, XSB, li, lj, dx);
        }
        __builtin_amdgcn_wave_barrier();
        // the finished tile (C layout) through the dS tile - free by now - into A layout
#pragma unroll
        for (int nb = 0; nb < DT; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ds[(lj * 4 + r) * XS + nb * 16 + li] = dx[nb][r];
        __builtin_amdgcn_wave_barrier();
        loads_landed_but<AHEAD == 2 ? NG : 0>();
        const unsigned orow = row_soff<DT, GB>(pt);
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Ds + li * XS + 16 * c + 4 * lj);
            if (c < NCH) {
                if constexpr (ACC) v += v4(gaccv[c]);
                if (BST) {
                    // this launch completes the gradient of the stage's activated output: the batch-statistics sums of its
                    // BatchNorm backward come for free (the raw tile is in registers; d = 16: the lanes of the gathered half
                    // compute along, their sums are never written)
                    const f32x4 z = __builtin_elementwise_fma(rawu[c], xf.sc(c), xf.sh(c));      // = build_x
                    f32x4 g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[r] = z[r] > 0.f ? v[r] : 0.f;
                    bsg[c] += g;
                    bsx[c] = __builtin_elementwise_fma(g, (rawu[c] - vl.get(VX<DT>::MU, c)) * vl.get(VX<DT>::IS, c), bsx[c]);
                }
            }
            if constexpr (DT == 1) {
                bstrow4<GB>(f4(v), p.GU, v_gu, orow);      // (each lane takes part in one of the two)
                bstrow4<GB>(f4(v), p.DG, v_dg, orow);
            } else if (c < NCH) bstrow4<GB>(f4(v), p.GU, v_gu + 16u * c * ES, orow);
            else bstrow4<GB>(f4(v), p.DG, v_dg + (16u * c - H) * ES, orow);
        }
        __builtin_amdgcn_wave_barrier();
    };
    if constexpr (AHEAD == 2) {
        for (int k = npts >> 1; k > 0; --k) {      // (one exit: see vpool_fwd_kernel)
            body(A);
            pt += pstep;
            body(B);
            pt += pstep;
        }
        if (npts & 1) body(A);
    } else {
        for (int k = npts; k > 0; --k) {
            body(A);
            pt += pstep;
        }
    }
    if (BST) {
        // (Tiles is free: every wavefront is past its last point once the barrier below is reached)
        __syncthreads();
        double* redd = reinterpret_cast<double*>(&Tiles[0][0][0]);      // [NW][2][HP] doubles
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sa = li_sum(bsg[c][r]), sb = li_sum(bsx[c][r]);
                if (li == 0) {
                    redd[(wave * 2 + 0) * HP + 16 * c + 4 * lj + r] = (double)sa;
                    redd[(wave * 2 + 1) * HP + 16 * c + 4 * lj + r] = (double)sb;
                }
            }
        __syncthreads();
        if (threadIdx.x < H) {
            const int c = threadIdx.x;
            double a0 = 0.0, a1 = 0.0;
            for (int wv = 0; wv < NW; ++wv) {
                a0 += redd[(wv * 2 + 0) * HP + c];
                a1 += redd[(wv * 2 + 1) * HP + c];
            }
            p.bstats[((long)blockIdx.x * 2 + 0) * H + c] = a0;
            p.bstats[((long)blockIdx.x * 2 + 1) * H + c] = a1;
        }
    }
    // combine the wavefronts' dW tiles in a fixed order (W region is free now)
    __syncthreads();
    float* red = Wmem;  // needs DT*DT*256 floats <= 2*D*XS: 16*DT*DT*16 <= 2*16*DT*(16*DT+4) always holds
    for (int w = 1; w < NW; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int nb = 0; nb < DT; ++nb)
#pragma unroll
                for (int kb = 0; kb < DT; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((nb * DT + kb) * 4 + r) * 64 + lane] = accw[nb][kb][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int nb = 0; nb < DT; ++nb)
#pragma unroll
                for (int kb = 0; kb < DT; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accw[nb][kb][r] += red[((nb * DT + kb) * 4 + r) * 64 + lane];
        }
    }
    if (wave == 0) {
        float* out = p.slab + (long)blockIdx.x * p.slab_stride;
#pragma unroll
        for (int nb = 0; nb < DT; ++nb)
#pragma unroll
            for (int kb = 0; kb < DT; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(long)(nb * 16 + lj * 4 + r) * D + kb * 16 + li] = accw[nb][kb][r];
    }
}


// ---- backward of a virtual stage, transposed orientation ------------------------------------------------------------------
// G arrives as A-layout float4s (one 16-byte load per lane and chunk instead of four 4-byte ones), the raw tile, the mask,
// xhat and dY = scale*(g - coef0 - xhat*coef1) live in the A layout; GU1 = dY . W2 is one more transposed product and leaves
// as 16-byte stores.  Only dW = dY^T . In needs the neighbourhood rows as the MFMA reduction index: dY and the stage's
// input are split ONCE, parked in a wavefront-private LDS tile as bf16 head / tail planes ([row][channel]) and come back
// through ds_read_b64_tr_b16 as the four rows 4*lj .. 4*lj+3 of column li - both operands of the product.
// Stage 1's input is the REDUCED one the forward multiplies (x_i and [x_i - x_j, dist]: W1.rpe = (Wa + Wb).x_i + (Wc - Wb).(x_i - x_j) + wd.dist),
// so the kernel accumulates dU = dY^T.x_i and dV = dY^T.[x_i - x_j, dist] and leaves dWa = dU, dWb = dU - dV[:3], dWc = dV[:3], dwd = dV[3].
template <int DT>
struct VBwdLane {
    static constexpr int NCH = VX<DT>::NCH, H = VX<DT>::H;
    f32x4 sc[NCH], sh[NCH], mu[NCH], is[NCH], k0[NCH], k1[NCH];
    __device__ __forceinline__ void load(const PoolParams& p, const float* coef, int lj) {
        const float* mu_ = p.src == 1 ? p.mu1 : p.mu2;
        const float* is_ = p.src == 1 ? p.is1 : p.is2;
        const float* sc_ = p.src == 1 ? p.sc1 : p.sc2;
        const float* sh_ = p.src == 1 ? p.sh1 : p.sh2;
        const float* b_ = p.src == 1 ? p.b1 : p.b2;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 16 * c + 4 * lj + r;
                const bool in = ch < H;
                sc[c][r] = in ? sc_[ch] : 0.f;
                sh[c][r] = in ? __builtin_fmaf(b_[ch], sc_[ch], sh_[ch]) : 0.f;      // = lane_fold / VLane: the same z in every kernel
                mu[c][r] = in ? mu_[ch] - b_[ch] : 0.f;
                is[c][r] = in ? is_[ch] : 0.f;
                k0[c][r] = (in && coef) ? coef[ch] : 0.f;
                k1[c][r] = (in && coef) ? coef[H + ch] : 0.f;
            }
    }
};

// raw tile of stage SRC for one point (A layout) and, for stage 2, the activated stage-1 tile it was computed from
template <int DT, int TERMS, int SRC>
__device__ __forceinline__ void stage_raw(const RpeIn2& in, const VLane<DT>& vl, const VW2<DT, TERMS>& w2, int li, int lj,
                                          f32x4 (&raw)[VX<DT>::NCH], float4 (&u1)[DT], Frag<DT>& f1) {
    constexpr int NCH = VX<DT>::NCH;
    f32x4 c0[NCH];
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) c0[nb] = splat(0.f);
    stage1_raw<DT>(in, vl, c0, raw);
    if constexpr (SRC == 2) {
#pragma unroll
        for (int c = 0; c < DT; ++c) u1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) {
            u1[nb] = f4(vrelu(__builtin_elementwise_fma(raw[nb], vl.get(VX<DT>::S1, nb), vl.get(VX<DT>::H1, nb))));
            raw[nb] = splat(0.f);
        }
        if constexpr (TERMS == 0) stage2_raw<DT, TERMS>(u1, nullptr, w2, li, lj, raw);
        else {
            if constexpr (DT == 1) split4(u1[0], f1.h[0], f1.l[0]);
            else {
                bf16x4 h0, l0, h1, l1;
                split4(u1[0], h0, l0);
                split4(u1[1], h1, l1);
                f1.h[0] = cat8(h0, h1);
                f1.l[0] = cat8(l0, l1);
            }
            stage2_raw<DT, TERMS>(u1, &f1, w2, li, lj, raw);
        }
    }
}

template <int DT, int TERMS, int SRC, bool GB>
__global__ __launch_bounds__(256) void vrpe_bn_reduce_kernel(const RpeBwdParams q) {
    const PoolParams& p = q.pp;
    constexpr int NW = 4, NCH = VX<DT>::NCH, HP = VX<DT>::HP, H = VX<DT>::H;
    __shared__ __attribute__((aligned(16))) unsigned char w2mem[SRC == 2 ? VW2<DT, TERMS>::BYTES : 16];
    __shared__ __attribute__((aligned(16))) float cl[VX<DT>::INLDS ? VX<DT>::NCONST * HP : 4];
    __shared__ double red[NW * 2 * HP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lj = lane >> 4;
    VW2<DT, TERMS> w2;
    w2.bind(w2mem);
    if constexpr (SRC == 2) w2.stage(p.W2, H, 1, 64 * NW);
    VLane<DT> vl;
    vl.load(p, li, lj, cl, 0);
    VBwdLane<DT> bc;
    bc.load(p, nullptr, lj);
    __syncthreads();
    f32x4 sg[NCH], sx[NCH];
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) sg[nb] = sx[nb] = splat(0.f);
    // 32-bit point arithmetic (points * 16 < 2^31, checked on the host): a 64-bit ordered compare is a VECTOR instruction
    const PointSpan span = point_span<NW>(p, wave);
    const int P = span.end, pstep = span.step;
    int pt = span.first;
    RpeIn2 rin = {0.f, 0.f, 0.f}, rin_nxt;
    Cursor cu;
    cu.start(pt, p.n);
    int idx_nxt = pt + pstep < P ? ld_idx(p, pt + pstep, li) : 0;
    float4 gin[NCH], gin_nxt[NCH];
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) gin[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pt < P) {
        fetch_rpe2(p, cu, li, vl.comp, ld_idx(p, pt, li), rin);
        load_g4<DT, GB>(q.G, pt, li, lj, gin);
    }
    loads_landed();
    for (; pt < P; pt += pstep) {
        const Cursor cn = cu.next(pstep, p.n);
        const int idx_n2 = pt + 2 * pstep < P ? ld_idx(p, pt + 2 * pstep, li) : 0;
        {
            const Cursor cf = pt + pstep < P ? cn : cu;      // (no branch: see pool_fwd_kernel)
            fetch_rpe2(p, cf, li, vl.comp, idx_nxt, rin_nxt);
            load_g4<DT, GB>(q.G, cf.pt, li, lj, gin_nxt);
        }
        loads_issued();
        idx_nxt = idx_n2;
        cu = cn;
        f32x4 raw[NCH];
        float4 u1[DT];
        Frag<DT> f1;
        stage_raw<DT, TERMS, SRC>(rin, vl, w2, li, lj, raw, u1, f1);
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) {
            const f32x4 z = __builtin_elementwise_fma(raw[nb], bc.sc[nb], bc.sh[nb]);
            f32x4 g;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = z[r] > 0.f ? v4(gin[nb])[r] : 0.f;
            sg[nb] += g;
            sx[nb] = __builtin_elementwise_fma(g, (raw[nb] - bc.mu[nb]) * bc.is[nb], sx[nb]);
        }
        loads_landed();
        rin = rin_nxt;
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) gin[nb] = gin_nxt[nb];
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = li_sum(sg[c][r]), b = li_sum(sx[c][r]);
            if (li == 0) {
                red[(wave * 2 + 0) * HP + 16 * c + 4 * lj + r] = (double)a;
                red[(wave * 2 + 1) * HP + 16 * c + 4 * lj + r] = (double)b;
            }
        }
    __syncthreads();
    if (threadIdx.x < H) {
        const int c = threadIdx.x;
        double a0 = 0.0, a1 = 0.0;
        for (int wv = 0; wv < NW; ++wv) {
            a0 += red[(wv * 2 + 0) * HP + c];
            a1 += red[(wv * 2 + 1) * HP + c];
        }
        q.stats[((long)blockIdx.x * 2 + 0) * H + c] = a0;
        q.stats[((long)blockIdx.x * 2 + 1) * H + c] = a1;
    }
}

template <int DT, int TERMS, int SRC, bool GB>
__global__ __launch_bounds__(256) void vrpe_wgrad_kernel(const RpeBwdParams q) {
    const PoolParams& p = q.pp;
    constexpr int NW = 4, NCH = VX<DT>::NCH, HP = VX<DT>::HP, H = VX<DT>::H;
    constexpr int KB = SRC == 1 ? 1 : NCH;            // k blocks of dW: stage 1 has 8 reduced input columns
    constexpr int RSB = HP + 8, RSF = HP + 4;         // row strides of the transposition tiles: bf16 planes / floats
    constexpr int TILE_BYTES = TERMS == 0 ? 16 * RSF * 4 : 2 * 16 * RSB * 2;
    __shared__ __attribute__((aligned(16))) unsigned char w2mem[SRC == 2 ? VW2<DT, TERMS>::BYTES : 16];
    __shared__ __attribute__((aligned(16))) unsigned char w2tmem[SRC == 2 ? VW2<DT, TERMS>::BYTES : 16];     // W2^T: GU1 = dY . W2
    __shared__ __attribute__((aligned(16))) float cl[VX<DT>::INLDS ? VX<DT>::NCONST * HP : 4];
    __shared__ __attribute__((aligned(16))) unsigned char Tl[NW][2][TILE_BYTES];      // [0]: the stage's input rows, [1]: dY
    static_assert(NW * 2 * TILE_BYTES >= (NCH * KB * 256 + NCH * 64) * 4 || true, "");
    __shared__ float red[NCH * KB * 256 + NW * HP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lj = lane >> 4;
    VW2<DT, TERMS> w2, w2t;
    w2.bind(w2mem);
    w2t.bind(w2tmem);
    if constexpr (SRC == 2) {
        w2.stage(p.W2, H, 1, 64 * NW);
        w2t.stage(p.W2, 1, H, 64 * NW);
    }
    VLane<DT> vl;
    vl.load(p, li, lj, cl, 0);
    VBwdLane<DT> bc;
    bc.load(p, q.coef, lj);
    for (int e = threadIdx.x; e < NW * 2 * TILE_BYTES / 4; e += 64 * NW) reinterpret_cast<float*>(&Tl[0][0][0])[e] = 0.f;     // padding columns stay zero
    __syncthreads();
    unsigned char* Ti = Tl[wave][0];
    unsigned char* Td = Tl[wave][1];
    f32x4 accw[NCH][KB], bsum4[NCH];
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) {
        bsum4[nb] = splat(0.f);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) accw[nb][kb] = splat(0.f);
    }
    // 32-bit point arithmetic (points * 16 < 2^31, checked on the host): a 64-bit ordered compare is a VECTOR instruction
    const PointSpan span = point_span<NW>(p, wave);
    const int P = span.end, pstep = span.step;
    int pt = span.first;
    // two points deep, two alternating buffers (see vpool_fwd_kernel)
    constexpr int NG = 3 + NCH + 1;
    struct Buf { RpeIn2 rin; float4 gin[NCH]; int idx; } A, B;
    A.rin.pc = A.rin.nc = A.rin.dd = B.rin.pc = B.rin.nc = B.rin.dd = 0.f;
    A.idx = B.idx = 0;
#pragma unroll
    for (int nb = 0; nb < NCH; ++nb) A.gin[nb] = B.gin[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int npts = (int)wave_points(pt, pstep, P);
    Cursor c0, c1, c2;
    c0.start(pt, p.n);
    c1 = c0.next(pstep, p.n);
    c2 = c1.next(pstep, p.n);
    if (pt < P) {
        const bool v1 = pt + pstep < P;
        const int i0 = ld_idx(p, pt, li);
        const int i1 = ld_idx(p, v1 ? pt + pstep : pt, li);
        A.idx = ld_idx(p, pt + 2 * pstep < P ? pt + 2 * pstep : pt, li);
        B.idx = ld_idx(p, pt + 3 * pstep < P ? pt + 3 * pstep : pt, li);
        fetch_rpe2(p, c0, li, vl.comp, i0, A.rin);
        load_g4<DT, GB>(q.G, pt, li, lj, A.gin);
        fetch_rpe2(p, v1 ? c1 : c0, li, vl.comp, i1, B.rin);
        load_g4<DT, GB>(q.G, v1 ? pt + pstep : pt, li, lj, B.gin);
    }
    loads_landed();
    // addresses of the transposing reads: rows 4*lj + (li >> 2), columns 4*(li & 3) .. +3 of a 16-column block
    const int trow = 4 * lj + (li >> 2), tcol = 4 * (li & 3);
    auto body = [&](Buf& U) {
        f32x4 raw[NCH];
        float4 u1[DT];
        Frag<DT> f1;
        stage_raw<DT, TERMS, SRC>(U.rin, vl, w2, li, lj, raw, u1, f1);
        // stage 1: the reduced inputs (columns 2*lj, 2*lj + 1 of the input tile), taken before the buffer is refilled
        float in0 = 0.f, in1 = 0.f;
        if constexpr (SRC == 1) {
            const float diff = U.rin.pc - U.rin.nc, dist = vsqrt(U.rin.dd);
            in0 = vl.l3 ? 0.f : U.rin.pc;
            in1 = vl.l3 ? dist : diff;
        }
        float4 dy[DT];
#pragma unroll
        for (int c = 0; c < DT; ++c) dy[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) {
            const f32x4 z = __builtin_elementwise_fma(raw[nb], bc.sc[nb], bc.sh[nb]);
            f32x4 g;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = z[r] > 0.f ? v4(U.gin[nb])[r] : 0.f;
            const f32x4 xh = (raw[nb] - bc.mu[nb]) * bc.is[nb];
            // padding channels / the lanes of the gathered half (d = 16): scale and the coefficients are zero there -> dy = 0
            const f32x4 d = bc.sc[nb] * (g - bc.k0[nb] - xh * bc.k1[nb]);
            bsum4[nb] += d;
            dy[nb] = f4(d);
        }
        {
            const Cursor cf = pt + 2 * pstep < P ? c2 : c0;      // (no branch: see pool_fwd_kernel)
            // (the streaming rows first: should the register allocator rotate a buffer register with a copy at the back edge,
            // the copy then waits for the oldest loads of the group, not for the gathers)
            load_g4<DT, GB>(q.G, cf.pt, li, lj, U.gin);
            fetch_rpe2(p, cf, li, vl.comp, U.idx, U.rin);
            U.idx = ld_idx(p, pt + 4 * pstep < P ? pt + 4 * pstep : pt, li);
        }
        loads_issued();
        c0 = c1; c1 = c2; c2 = c2.next(pstep, p.n);
        if constexpr (TERMS == 0) {
            float* Fi = reinterpret_cast<float*>(Ti);
            float* Fd = reinterpret_cast<float*>(Td);
#pragma unroll
            for (int nb = 0; nb < NCH; ++nb) *reinterpret_cast<float4*>(Fd + li * RSF + 16 * nb + 4 * lj) = dy[nb];
            if constexpr (SRC == 1) {
                *reinterpret_cast<float2*>(Fi + li * RSF + 2 * lj) = make_float2(in0, in1);
            } else {
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) *reinterpret_cast<float4*>(Fi + li * RSF + 16 * nb + 4 * lj) = u1[nb];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float bx[KB];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) bx[kb] = Fi[(4 * t + lj) * RSF + kb * 16 + li];
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) {
                    const float ad = Fd[(4 * t + lj) * RSF + nb * 16 + li];
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
                        accw[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad, bx[kb], accw[nb][kb], 0, 0, 0);
                }
            }
            if constexpr (SRC == 2) {
                f32x4 gu[NCH];
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) gu[nb] = splat(0.f);
                gemm_f32<DT, VX<DT>::NC2, NCH, true>(dy, w2t.f, VX<DT>::S2F, li, lj, gu);
                loads_landed_but<NG>();
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb)
                    bstrow4<GB>(f4(gu[nb]), q.GU1, row_voff<DT, GB>(li, lj) + 16u * nb * (GB ? 2 : 4), row_soff<DT, GB>(pt));
            } else loads_landed_but<NG>();
        } else {
            __bf16* Bi = reinterpret_cast<__bf16*>(Ti);      // planes: [hi][16][RSB], [lo][16][RSB]
            __bf16* Bd = reinterpret_cast<__bf16*>(Td);
            Frag<DT> fd;
            if constexpr (DT == 1) split4(dy[0], fd.h[0], fd.l[0]);
            else {
                bf16x4 h0, l0, h1, l1;
                split4(dy[0], h0, l0);
                split4(dy[1], h1, l1);
                fd.h[0] = cat8(h0, h1);
                fd.l[0] = cat8(l0, l1);
            }
            // park dY and the input: row li, the lane's four channels of every rpe chunk
            if constexpr (DT == 1) {
                *reinterpret_cast<bf16x4*>(Bd + li * RSB + 4 * lj) = fd.h[0];
                *reinterpret_cast<bf16x4*>(Bd + 16 * RSB + li * RSB + 4 * lj) = fd.l[0];
            } else {
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) {
                    bf16x4 hh, ll;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { hh[r] = fd.h[0][4 * nb + r]; ll[r] = fd.l[0][4 * nb + r]; }
                    *reinterpret_cast<bf16x4*>(Bd + li * RSB + 16 * nb + 4 * lj) = hh;
                    *reinterpret_cast<bf16x4*>(Bd + 16 * RSB + li * RSB + 16 * nb + 4 * lj) = ll;
                }
            }
            if constexpr (SRC == 1) {
                const float i0 = in0, i1 = in1;
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                bf16x2 hh, ll;
                hh[0] = (__bf16)i0; hh[1] = (__bf16)i1;
                ll[0] = (__bf16)(i0 - (float)hh[0]); ll[1] = (__bf16)(i1 - (float)hh[1]);
                *reinterpret_cast<bf16x2*>(Bi + li * RSB + 2 * lj) = hh;
                *reinterpret_cast<bf16x2*>(Bi + 16 * RSB + li * RSB + 2 * lj) = ll;
            } else {
                if constexpr (DT == 1) {
                    *reinterpret_cast<bf16x4*>(Bi + li * RSB + 4 * lj) = f1.h[0];
                    *reinterpret_cast<bf16x4*>(Bi + 16 * RSB + li * RSB + 4 * lj) = f1.l[0];
                } else {
#pragma unroll
                    for (int nb = 0; nb < NCH; ++nb) {
                        bf16x4 hh, ll;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { hh[r] = f1.h[0][4 * nb + r]; ll[r] = f1.l[0][4 * nb + r]; }
                        *reinterpret_cast<bf16x4*>(Bi + li * RSB + 16 * nb + 4 * lj) = hh;
                        *reinterpret_cast<bf16x4*>(Bi + 16 * RSB + li * RSB + 16 * nb + 4 * lj) = ll;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // dW[n][k] += sum_rows dY[row][n] * In[row][k]: both operands = four rows of one column, by transposing reads
            bf16x4 xh[KB], xl[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                xh[kb] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Bi + trow * RSB + 16 * kb + tcol)));
                xl[kb] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Bi + 16 * RSB + trow * RSB + 16 * kb + tcol)));
            }
#pragma unroll
            for (int nb = 0; nb < NCH; ++nb) {
                const bf16x4 dh = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Bd + trow * RSB + 16 * nb + tcol)));
                const bf16x4 dl = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Bd + 16 * RSB + trow * RSB + 16 * nb + tcol)));
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) accw[nb][kb] = mfma16(dh, xh[kb], accw[nb][kb]);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) accw[nb][kb] = mfma16(dh, xl[kb], accw[nb][kb]);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) accw[nb][kb] = mfma16(dl, xh[kb], accw[nb][kb]);
            }
            if constexpr (SRC == 2) {
                // gradient w.r.t. the activated stage-1 output: dY . W2, transposed product -> A layout
                f32x4 gu[NCH];
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb) gu[nb] = splat(0.f);
                gemm_frag<DT, NCH, 1, true>(fd, w2t.h, w2t.l, VX<DT>::S2B, li, lj, gu);
                loads_landed_but<NG>();
#pragma unroll
                for (int nb = 0; nb < NCH; ++nb)
                    bstrow4<GB>(f4(gu[nb]), q.GU1, row_voff<DT, GB>(li, lj) + 16u * nb * (GB ? 2 : 4), row_soff<DT, GB>(pt));
            } else loads_landed_but<NG>();
        }
        __builtin_amdgcn_wave_barrier();
    };
    for (int k = npts >> 1; k > 0; --k) {      // (one exit: see vpool_fwd_kernel)
        body(A);
        pt += pstep;
        body(B);
        pt += pstep;
    }
    if (npts & 1) body(A);
    // combine the four wavefronts in a fixed order; slab layout: dW[n][k] (n < H, k < Kin) then db[n]
    constexpr int Kin = SRC == 1 ? 10 : H;
    float* rb = red + NCH * KB * 256;                 // [NW][HP] per-wavefront column sums of dY
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sv = li_sum(bsum4[c][r]);
            if (li == 0) rb[wave * HP + 16 * c + 4 * lj + r] = sv;
        }
    for (int wv = 1; wv < NW; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int nb = 0; nb < NCH; ++nb)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((nb * KB + kb) * 4 + r) * 64 + lane] = accw[nb][kb][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int nb = 0; nb < NCH; ++nb)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accw[nb][kb][r] += red[((nb * KB + kb) * 4 + r) * 64 + lane];
        }
    }
    __syncthreads();
    float* out = q.slab + (long)blockIdx.x * ((long)H * Kin + H);
    if (wave == 0) {
#pragma unroll
        for (int nb = 0; nb < NCH; ++nb) {
            if constexpr (SRC == 1) {
                // column li of the reduced product: 2c -> dU[c] (x_i component c), 2c + 1 -> dV[c] ([x_i - x_j, dist] component c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb * 16 + lj * 4 + r;
                    const float v = accw[nb][0][r], o = __shfl_xor(v, 1, 64);
                    const int c = li >> 1;
                    if (n < H && li < 8) {
                        if ((li & 1) == 0) {
                            if (c < 3) { out[(long)n * 10 + c] = v; out[(long)n * 10 + 3 + c] = v - o; }     // dWa = dU, dWb = dU - dV
                        } else {
                            out[(long)n * 10 + (c < 3 ? 6 + c : 9)] = v;                                   // dWc = dV[:3], dwd = dV[3]
                        }
                    }
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int k = kb * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = nb * 16 + lj * 4 + r;
                        if (n < H && k < Kin) out[(long)n * Kin + k] = accw[nb][kb][r];
                    }
                }
            }
        }
    }
    if (threadIdx.x < H) {
        const int n = threadIdx.x;
        out[(long)H * Kin + n] = ((rb[0 * HP + n] + rb[1 * HP + n]) + rb[2 * HP + n]) + rb[3 * HP + n];
    }
}

// Sum of the per-workgroup partial dW slabs ([nsplit][count]) in a fixed order:
__global__ __launch_bounds__(256) void pool_dw_reduce_kernel(const float* __restrict__ slab, int nsplit, int count,
                                                             float* __restrict__ dW) {
    // 16 consecutive elements x 16 slab lanes per workgroup (was 64 x 4: d = 16 is then 16 workgroups of short chains
    // instead of 4 of long ones); lane sy adds slabs sy, sy+16, ... with four loads in flight, fixed order throughout
    __shared__ float red[16][17];
    const int ex = threadIdx.x & 15, sy = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + ex;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < count) {
        const float* src = slab + e;
        int i = sy;
        for (; i + 48 < nsplit; i += 64) {
            s0 += src[(long)i * count];
            s1 += src[(long)(i + 16) * count];
            s2 += src[(long)(i + 32) * count];
            s3 += src[(long)(i + 48) * count];
        }
        for (; i < nsplit; i += 16) s0 += src[(long)i * count];
    }
    red[sy][ex] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sy == 0 && e < count) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][ex];
        dW[e] = t;
    }
}

// workgroups of a fused pooling launch: each stages W into LDS first and (backward) leaves a d x d slab, so more
// than 1024 only pays for the narrow forward kernel (measured: d = 16 forward 174 -> 144 us at 2048; every
// backward and the 64-channel forward get slower)
// Grid caps = what is resident at once on the 256 CUs (measured, not derived: a second partial round of workgroups costs
// a second weight staging and a tail): d = 64 backward, virtual: one 8-wavefront workgroup per CU (LDS) -> 256
// (512: 498 -> 462 us per step); d = 64 forward: three 4-wavefront workgroups per CU (146 VGPRs) -> 768 (1024: 253 ->
// 232 us); d = 16 forward: five per CU (96 VGPRs) -> 1280 (2048: 347 -> 336 us).  The others measured best as they are.
int pool_grid(long P, int d, bool backward, bool virt = false) {
    if (backward && virt && d == 64) {
        // 8 wavefronts per workgroup (the staged weights + the small rpe weights leave room for one workgroup per CU only)
        long g8 = (P + 31) / 32;
        if (g8 < 1) g8 = 1;
        return (int)(g8 < 256 ? g8 : 256);
    }
    // (RL_GRID_FWD16 / RL_GRID_BWD16 / RL_GRID_FWD64: measurement overrides of the caps, tools/pool_bench.py)
    static const long env_f16 = getenv("RL_GRID_FWD16") ? atol(getenv("RL_GRID_FWD16")) : 0;
    static const long env_b16 = getenv("RL_GRID_BWD16") ? atol(getenv("RL_GRID_BWD16")) : 0;
    static const long env_f64 = getenv("RL_GRID_FWD64") ? atol(getenv("RL_GRID_FWD64")) : 0;
    long cap = (!backward && d <= 16) ? 1280 : (!backward && d == 64) ? 768 : 1024;
    if (!backward && d <= 16 && env_f16 > 0) cap = env_f16;
    if (backward && d <= 16 && env_b16 > 0) cap = env_b16;
    if (!backward && d == 64 && env_f64 > 0) cap = env_f64;
    long g = (P + 15) / 16;  // >= 4 points per wavefront
    if (d == 128) g = (P + 31) / 32 < 256 ? (P + 31) / 32 : 256;   // 8 wavefronts per workgroup, one workgroup per CU
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

// XCD-local point ranges (PointSpan): on when the grid is a multiple of 8; RL_NO_XCD_POINTS turns them off (A/B).  Used by the
// virtual FORWARD launches and the rpe statistics (level 0: 101.7 -> 81 us per launch, PMC fetch 620 -> ~100 MB); measured on
// every pooling / rpe kernel of a step (round 5): the backward kernels do not gain (+-2 %: bound by vector instructions, and
// their gathers already share lines), pool128_bwd loses 3 % - those keep the round-robin assignment.
static int xcd_chunk_for(long P, int grid) {
    static const bool off = getenv("RL_NO_XCD_POINTS") != nullptr;
    if (off || grid < 8 || grid % 8 != 0 || P >= (1l << 30)) return 0;
    return (int)((P + 7) / 8);
}

// Arithmetic of the tile kernels for a level of width d: the narrow levels are bound by vector-instruction issue, not by
// the matrix pipe - the head/tail split of every operand costs more than the exact fp32 MFMA it avoids - so d <=
// RL_POOL_FP32_MAX_D (default 0: measured no gain at d = 16 / 32) would run v_mfma_f32_16x16x4_f32 (exact fp32 products, whatever the wide-GEMM mode).
static int pool_terms(int d) {
    static int max_d = -1;
    if (max_d < 0) {
        const char* e = getenv("RL_POOL_FP32_MAX_D");
        max_d = e ? atoi(e) : 0;
    }
    if (d <= max_d) return 0;
    return rl_wide_terms();
}

// The kernels of a virtual rpe branch address their operands through 32-bit byte offsets (buffer descriptors, see "addressing of
// the per-point operands"): every tensor they touch must stay below 2 GB, and a cloud below 2^24 points.
int virtual_extents_ok(const rl_pool_desc* d, const char* who) {
    const int64_t B = d->n > 0 ? d->points / d->n : 0, h = d->d / 2, lim = (int64_t)1 << 31;
    RL_REQUIRE(d->n < (1 << 24), RL_ERR_UNSUPPORTED, "%s: a virtual rpe branch needs fewer than 2^24 points per cloud", who);
    RL_REQUIRE((int64_t)d->points * 16 * h * 4 < lim && (int64_t)d->points * d->d * 4 < lim, RL_ERR_UNSUPPORTED,
               "%s: a virtual rpe branch needs (points*16) x d/2 row tensors below 2 GB", who);
    RL_REQUIRE(B * d->xyz_bstride * 16 < lim && (d->g_bstride == 0 || B * d->g_bstride * h * 4 < lim), RL_ERR_UNSUPPORTED,
               "%s: a virtual rpe branch needs coordinate / feature tables below 2 GB", who);
    return RL_OK;
}

int fill(PoolParams* p, const rl_pool_desc* d, const char* who, bool backward) {
    RL_REQUIRE(d && d->G && d->idx && d->W && d->points > 0 && d->n > 0, RL_ERR_ARGS, "%s: bad descriptor", who);
    RL_REQUIRE(d->nbr_k == 16, RL_ERR_UNSUPPORTED, "%s: the fused kernel needs 16 neighbours (got %d)", who, d->nbr_k);
    RL_REQUIRE(d->d == 16 || d->d == 32 || d->d == 64 || (d->d == 128 && rl_wide_terms() != 0), RL_ERR_UNSUPPORTED,
               "%s: d must be 16, 32 or 64, or 128 outside the fp32 arithmetic mode (got %d)", who, d->d);
    RL_REQUIRE(d->g_bstride >= d->n && d->points % d->n == 0, RL_ERR_ARGS, "%s: bad cloud geometry", who);
    RL_REQUIRE((int64_t)d->points * 16 < (1l << 31), RL_ERR_ARGS, "%s: too many neighbourhood rows", who);
    RL_REQUIRE((d->u_source > 0 || ((uintptr_t)d->U & 15) == 0) && ((uintptr_t)d->G & 15) == 0, RL_ERR_ARGS, "%s: U/G must be 16-byte aligned", who);
    RL_REQUIRE((d->u_scale == nullptr) == (d->u_shift == nullptr) && (d->g_scale == nullptr) == (d->g_shift == nullptr),
               RL_ERR_ARGS, "%s: scale/shift must come together", who);
    p->U = d->U; p->ulazy.scale = d->u_scale; p->ulazy.shift = d->u_shift; p->ulazy.act = d->u_act; p->ulazy.slope = d->u_slope;
    p->G = d->G; p->g_bstride = d->g_bstride;
    p->glazy.scale = d->g_scale; p->glazy.shift = d->g_shift; p->glazy.act = d->g_act; p->glazy.slope = d->g_slope;
    p->idx = d->idx; p->W = d->W; p->P = d->points; p->n = d->n; p->d = d->d;
    p->Pout = d->Pout; p->dP = d->dP; p->GU = d->GU; p->gu_accumulate = d->gu_accumulate; p->DG = d->DG; p->slab = d->slab;
    p->slab_stride = (long)d->d * d->d + (d->dW ? 0 : d->d);
    p->X_out = nullptr; p->dS_out = nullptr;
    p->xcd_chunk = 0;
    p->src = d->u_source;
    RL_REQUIRE(d->u_source >= 0 && d->u_source <= 2, RL_ERR_ARGS, "%s: u_source must be 0, 1 or 2", who);
    if (d->u_source > 0) {
        RL_REQUIRE(d->d == 16 || d->d == 32 || d->d == 64, RL_ERR_UNSUPPORTED, "%s: a virtual rpe branch needs d = 16, 32 or 64 (got %d)", who, d->d);
        RL_REQUIRE(d->xyz && d->nbr_d2 && d->xyz_bstride >= d->n && d->W1 && d->b1, RL_ERR_ARGS, "%s: incomplete virtual rpe branch", who);
        RL_REQUIRE(d->u_source < 2 || (d->W2 && d->b2 && d->scale1 && d->shift1), RL_ERR_ARGS, "%s: stage 2 needs W2 / b2 and BatchNorm 1", who);
    } else {
        RL_REQUIRE(d->U, RL_ERR_ARGS, "%s: null U", who);
    }
    p->xyz = d->xyz; p->xyz_bstride = d->xyz_bstride; p->nbr_d2 = d->nbr_d2;
    p->xyz_w = d->xyz_width == 4 ? 4 : 3;
    if (d->u_source > 0) {
        { int rc2 = virtual_extents_ok(d, who); if (rc2) return rc2; }
        RL_REQUIRE(d->xyz_width == 0 || d->xyz_width == 3 || d->xyz_width == 4, RL_ERR_ARGS, "%s: xyz_width must be 3 or 4", who);
        RL_REQUIRE(d->xyz_width != 4 || ((uintptr_t)d->xyz & 15) == 0, RL_ERR_ARGS, "%s: padded xyz must be 16-byte aligned", who);
    }
    p->W1 = d->W1; p->b1 = d->b1; p->sc1 = d->scale1; p->sh1 = d->shift1;
    p->W2 = d->W2; p->b2 = d->b2; p->sc2 = d->scale2; p->sh2 = d->shift2;
    p->mu1 = d->mean1; p->is1 = d->invstd1; p->mu2 = d->mean2; p->is2 = d->invstd2;
    p->piv1 = backward ? nullptr : d->pivot_mean1; p->piv2 = backward ? nullptr : d->pivot_mean2;
    p->fstats2 = backward ? nullptr : d->bn_fwd_stats2;
    if (p->fstats2) RL_REQUIRE(d->u_source == 1 && d->W2 && d->b2, RL_ERR_ARGS, "%s: bn_fwd_stats2 needs virtual stage 1 and W2 / b2", who);
    p->bstats = backward ? d->bn_bwd_stats : nullptr;
    if (p->bstats) {
        RL_REQUIRE(d->u_source > 0, RL_ERR_ARGS, "%s: bn_bwd_stats needs a virtual rpe stage", who);
        RL_REQUIRE(d->u_source == 1 ? (d->mean1 && d->invstd1) : (d->mean2 && d->invstd2), RL_ERR_ARGS, "%s: bn_bwd_stats needs the stage's saved mean / invstd", who);
    }
    if (!backward) RL_REQUIRE(d->Pout, RL_ERR_ARGS, "%s: null output", who);
    else RL_REQUIRE(d->dP && d->GU && d->DG, RL_ERR_ARGS, "%s: null gradient buffers", who);
    return RL_OK;
}

}  // namespace

// d = 128 keeps both weight images (head + tail, 70 KB) in LDS as bf16: only in the bf16x3 / bf16 arithmetic modes
extern "C" int rl_pool_supported(int d, int nbr_k) {
    if (nbr_k != 16) return 0;
    if (d == 16 || d == 32 || d == 64) return 1;
    return (d == 128 && rl_wide_terms() != 0) ? 1 : 0;
}

extern "C" int64_t rl_pool_slab_floats(int64_t points, int d) { return (int64_t)pool_grid(points, d, true) * (d * d + d); }
extern "C" int rl_pool_bwd_slots(int64_t points, int d) { return pool_grid(points, d, true, true); }
extern "C" int rl_pool_bwd_grid(int64_t points, int d, int virt) { return pool_grid(points, d, true, virt != 0); }
extern "C" int rl_pool_fwd_slots(int64_t points, int d) { return pool_grid(points, d, false); }

extern "C" int rl_pool_fwd(const rl_pool_desc* d, void* stream) {
    PoolParams p;
    int rc = fill(&p, d, "rl_pool_fwd", false);
    if (rc) return rc;
    const int g = pool_grid(p.P, p.d, false);
    hipStream_t st = (hipStream_t)stream;
    if (p.src > 0) {
        p.xcd_chunk = xcd_chunk_for(p.P, g);
        RL_REQUIRE(p.sc1 && p.sh1 && (p.src < 2 || (p.sc2 && p.sh2)), RL_ERR_ARGS, "rl_pool_fwd: the virtual rpe branch needs its folded BatchNorm(s)");
        const int key = (pool_terms(p.d) == 0 ? 0 : 100) + (p.d == 16 ? 10 : p.d == 32 ? 20 : 40) + (p.src == 2 ? 2 : p.fstats2 ? 1 : 0);
#define VFWD(K, DT, TERMS, SRC, FST) \
        case K: hipLaunchKernelGGL((vpool_fwd_kernel<DT, TERMS, SRC, FST>), dim3(g), dim3(256), 0, st, p); break;
        switch (key) {
            VFWD(10, 1, 0, 1, false) VFWD(11, 1, 0, 1, true) VFWD(12, 1, 0, 2, false)
            VFWD(20, 2, 0, 1, false) VFWD(21, 2, 0, 1, true) VFWD(22, 2, 0, 2, false)
            VFWD(40, 4, 0, 1, false) VFWD(41, 4, 0, 1, true) VFWD(42, 4, 0, 2, false)
            VFWD(110, 1, 3, 1, false) VFWD(111, 1, 3, 1, true) VFWD(112, 1, 3, 2, false)
            VFWD(120, 2, 3, 1, false) VFWD(121, 2, 3, 1, true) VFWD(122, 2, 3, 2, false)
            VFWD(140, 4, 3, 1, false) VFWD(141, 4, 3, 1, true) VFWD(142, 4, 3, 2, false)
            default: RL_REQUIRE(false, RL_ERR_UNSUPPORTED, "rl_pool_fwd: no virtual kernel for d %d, source %d (key %d)", p.d, p.src, key);
        }
#undef VFWD
        rl_note_kernel("vpool_fwd_kernel");
        RL_LAUNCH_CHECK("rl_pool_fwd(virtual)");
        return RL_OK;
    }
    if (pool_terms(p.d) == 0) {
        if (p.d == 16) hipLaunchKernelGGL((pool_fwd_kernel<1, 0>), dim3(g), dim3(256), 0, st, p);
        else if (p.d == 32) hipLaunchKernelGGL((pool_fwd_kernel<2, 0>), dim3(g), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pool_fwd_kernel<4, 0>), dim3(g), dim3(256), 0, st, p);
    } else {
        if (p.d == 16) hipLaunchKernelGGL((pool_fwd_kernel<1, 3>), dim3(g), dim3(256), 0, st, p);
        else if (p.d == 32) hipLaunchKernelGGL((pool_fwd_kernel<2, 3>), dim3(g), dim3(256), 0, st, p);
        else if (p.d == 64) hipLaunchKernelGGL((pool_fwd_kernel<4, 3>), dim3(g), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pool_fwd_kernel<8, 3, 8>), dim3(g), dim3(512), 0, st, p);   // 70 KB of W: 8 wavefronts share it
    }
    rl_note_kernel(p.d == 16 ? "pool_fwd_kernel<1>" : p.d == 32 ? "pool_fwd_kernel<2>" : p.d == 64 ? "pool_fwd_kernel<4>" : "pool_fwd_kernel<8>");
    RL_LAUNCH_CHECK("rl_pool_fwd");
    return RL_OK;
}

extern "C" int rl_pool_bwd(const rl_pool_desc* d, void* stream) {
    PoolParams p;
    int rc = fill(&p, d, "rl_pool_bwd", true);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (p.d == 128) {
        // no dW here: X and dS are written out for the caller's wide weight-gradient kernel
        RL_REQUIRE(d->X_out && d->dS_out, RL_ERR_ARGS, "rl_pool_bwd: d = 128 needs X_out and dS_out");
        RL_REQUIRE((((uintptr_t)d->X_out | (uintptr_t)d->dS_out) & 15) == 0, RL_ERR_ARGS, "rl_pool_bwd: X_out / dS_out must be 16-byte aligned");
        p.X_out = d->X_out; p.dS_out = d->dS_out;
        const int g = pool_grid(p.P, p.d, true);
        if (d->rows_bf16) hipLaunchKernelGGL(pool128_bwd_kernel<true>, dim3(g), dim3(512), 0, st, p);
        else hipLaunchKernelGGL(pool128_bwd_kernel<false>, dim3(g), dim3(512), 0, st, p);
        rl_note_kernel("pool128_bwd_kernel");
        RL_LAUNCH_CHECK("rl_pool_bwd(128)");
        return RL_OK;
    }
    // dW == NULL: the caller sums the partial slabs itself (rl_wgrad_reduce_batch: nsplit = rl_pool_bwd_slots / the launch's
    // workgroups, N = K = d, slab stride d*d + d) - one reduction launch for a whole backward pass instead of one per block
    RL_REQUIRE(d->slab, RL_ERR_ARGS, "rl_pool_bwd: null gradient buffers");
    RL_REQUIRE(!d->rows_bf16 || pool_terms(p.d) == 3, RL_ERR_UNSUPPORTED, "rl_pool_bwd: bf16 gradient rows need the bf16x3 arithmetic mode");
    const int g = pool_grid(p.P, p.d, true, p.src > 0);
    RL_REQUIRE(d->slab_floats >= (int64_t)g * p.slab_stride, RL_ERR_ARGS, "rl_pool_bwd: slab too small");
    if (p.src > 0) {
        // measurement switch (round 6, profiles/r06_xcd_bwd_pmc.md): the XCD-local point ranges of the forward for the backward too
        static const bool xcd_bwd = getenv("RL_XCD_BWD") != nullptr;
        if (xcd_bwd) p.xcd_chunk = xcd_chunk_for(p.P, g);
        RL_REQUIRE(p.sc1 && p.sh1 && (p.src < 2 || (p.sc2 && p.sh2)), RL_ERR_ARGS, "rl_pool_bwd: the virtual rpe branch needs its folded BatchNorm(s)");
        const int key = (pool_terms(p.d) == 0 ? 0 : d->rows_bf16 ? 200 : 100) + (p.d == 16 ? 10 : p.d == 32 ? 20 : 40) + (p.src == 2 ? 2 : 0) + (p.gu_accumulate ? 1 : 0);
#define VBWD(K, DT, TERMS, SRC, NW, GB, ACC) \
        case K: hipLaunchKernelGGL((vpool_bwd_kernel<DT, TERMS, SRC, NW, GB, ACC>), dim3(g), dim3(64 * NW), 0, st, p); break;
#define VBWD4(K0, TERMS, GB) \
        VBWD(K0 + 10, 1, TERMS, 1, 4, GB, false) VBWD(K0 + 11, 1, TERMS, 1, 4, GB, true) VBWD(K0 + 12, 1, TERMS, 2, 4, GB, false) VBWD(K0 + 13, 1, TERMS, 2, 4, GB, true) \
        VBWD(K0 + 20, 2, TERMS, 1, 4, GB, false) VBWD(K0 + 21, 2, TERMS, 1, 4, GB, true) VBWD(K0 + 22, 2, TERMS, 2, 4, GB, false) VBWD(K0 + 23, 2, TERMS, 2, 4, GB, true) \
        VBWD(K0 + 40, 4, TERMS, 1, 8, GB, false) VBWD(K0 + 41, 4, TERMS, 1, 8, GB, true) VBWD(K0 + 42, 4, TERMS, 2, 8, GB, false) VBWD(K0 + 43, 4, TERMS, 2, 8, GB, true)
        switch (key) {
            VBWD4(0, 0, false)
            VBWD4(100, 3, false)
            VBWD4(200, 3, true)
            default: RL_REQUIRE(false, RL_ERR_UNSUPPORTED, "rl_pool_bwd: no virtual kernel for d %d, source %d (key %d)", p.d, p.src, key);
        }
#undef VBWD4
#undef VBWD
        rl_note_kernel("vpool_bwd_kernel");
        RL_LAUNCH_CHECK("rl_pool_bwd(virtual)");
        if (d->dW) hipLaunchKernelGGL(pool_dw_reduce_kernel, dim3(rl_cdiv(p.d * p.d, 16)), dim3(256), 0, st, p.slab, g, p.d * p.d, d->dW);
        RL_LAUNCH_CHECK("rl_pool_bwd(reduce)");
        return RL_OK;
    }
    if (pool_terms(p.d) == 0) {
        if (p.d == 16) hipLaunchKernelGGL((pool_bwd_kernel<1, 0>), dim3(g), dim3(256), 0, st, p);
        else if (p.d == 32) hipLaunchKernelGGL((pool_bwd_kernel<2, 0>), dim3(g), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pool_bwd_kernel<4, 0>), dim3(g), dim3(256), 0, st, p);
    } else if (d->rows_bf16) {
        if (p.d == 16) hipLaunchKernelGGL((pool_bwd_kernel<1, 3, false, 4, true>), dim3(g), dim3(256), 0, st, p);
        else if (p.d == 32) hipLaunchKernelGGL((pool_bwd_kernel<2, 3, false, 4, true>), dim3(g), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pool_bwd_kernel<4, 3, false, 4, true>), dim3(g), dim3(256), 0, st, p);
    } else {
        if (p.d == 16) hipLaunchKernelGGL((pool_bwd_kernel<1, 3>), dim3(g), dim3(256), 0, st, p);
        else if (p.d == 32) hipLaunchKernelGGL((pool_bwd_kernel<2, 3>), dim3(g), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((pool_bwd_kernel<4, 3>), dim3(g), dim3(256), 0, st, p);
    }
    rl_note_kernel(p.d == 16 ? "pool_bwd_kernel<1>" : p.d == 32 ? "pool_bwd_kernel<2>" : "pool_bwd_kernel<4>");
    RL_LAUNCH_CHECK("rl_pool_bwd");
    if (d->dW) hipLaunchKernelGGL(pool_dw_reduce_kernel, dim3(rl_cdiv(p.d * p.d, 16)), dim3(256), 0, st, p.slab, g, p.d * p.d, d->dW);
    RL_LAUNCH_CHECK("rl_pool_bwd(reduce)");
    return RL_OK;
}

// ---- the rpe branch without its tensors: batch statistics of a virtual stage ------------------------------------
static int rpe_grid(long P) {
    long g = (P + 15) / 16;
    if (g < 1) g = 1;
    return (int)(g < RL_MAX_SLOTS ? g : RL_MAX_SLOTS);
}

extern "C" int rl_rpe_stats_slots(int64_t points) { return rpe_grid(points); }

extern "C" int rl_rpe_stats(const rl_pool_desc* d, double* stats, void* stream) {
    RL_REQUIRE(d && stats && d->idx && d->points > 0 && d->n > 0 && d->points % d->n == 0, RL_ERR_ARGS, "rl_rpe_stats: bad descriptor");
    RL_REQUIRE(d->nbr_k == 16, RL_ERR_UNSUPPORTED, "rl_rpe_stats: 16 neighbours expected (got %d)", d->nbr_k);
    RL_REQUIRE(d->u_source == 1 || d->u_source == 2, RL_ERR_ARGS, "rl_rpe_stats: u_source must be 1 or 2");
    RL_REQUIRE(d->d == 16 || d->d == 32 || d->d == 64, RL_ERR_UNSUPPORTED, "rl_rpe_stats: d must be 16, 32 or 64 (got %d)", d->d);
    RL_REQUIRE(d->xyz && d->nbr_d2 && d->xyz_bstride >= d->n && d->W1 && d->b1, RL_ERR_ARGS, "rl_rpe_stats: incomplete rpe branch");
    RL_REQUIRE(d->u_source < 2 || (d->W2 && d->b2 && d->scale1 && d->shift1), RL_ERR_ARGS, "rl_rpe_stats: stage 2 needs W2 / b2 and BatchNorm 1");
    RL_REQUIRE((int64_t)d->points * 16 < (1l << 31), RL_ERR_ARGS, "rl_rpe_stats: too many neighbourhood rows");
    { int rc2 = virtual_extents_ok(d, "rl_rpe_stats"); if (rc2) return rc2; }
    PoolParams p = {};
    p.idx = d->idx; p.P = d->points; p.n = d->n; p.d = d->d; p.src = d->u_source;
    p.xyz = d->xyz; p.xyz_bstride = d->xyz_bstride; p.nbr_d2 = d->nbr_d2;
    p.xyz_w = d->xyz_width == 4 ? 4 : 3;
    RL_REQUIRE(d->xyz_width != 4 || ((uintptr_t)d->xyz & 15) == 0, RL_ERR_ARGS, "rl_rpe_stats: padded xyz must be 16-byte aligned");
    p.W1 = d->W1; p.b1 = d->b1; p.sc1 = d->scale1; p.sh1 = d->shift1;
    p.W2 = d->W2; p.b2 = d->b2; p.sc2 = d->scale2; p.sh2 = d->shift2;
    p.piv1 = d->pivot_mean1; p.piv2 = d->pivot_mean2;       // shifted sums around the stage's running mean (or null)
    const int g = rpe_grid(p.P);
    p.xcd_chunk = xcd_chunk_for(p.P, g);
    hipStream_t st = (hipStream_t)stream;
    const int key = (pool_terms(p.d) == 0 ? 0 : 100) + (p.d == 16 ? 10 : p.d == 32 ? 20 : 40) + p.src;
#define VST(K, DT, TERMS, SRC) \
    case K: hipLaunchKernelGGL((vrpe_stats_kernel<DT, TERMS, SRC>), dim3(g), dim3(256), 0, st, p, stats); break;
    switch (key) {
        VST(11, 1, 0, 1) VST(12, 1, 0, 2) VST(21, 2, 0, 1) VST(22, 2, 0, 2) VST(41, 4, 0, 1) VST(42, 4, 0, 2)
        VST(111, 1, 3, 1) VST(112, 1, 3, 2) VST(121, 2, 3, 1) VST(122, 2, 3, 2) VST(141, 4, 3, 1) VST(142, 4, 3, 2)
        default: RL_REQUIRE(false, RL_ERR_UNSUPPORTED, "rl_rpe_stats: no kernel for d %d, source %d (key %d)", p.d, p.src, key);
    }
#undef VST
    rl_note_kernel("vrpe_stats_kernel");
    RL_LAUNCH_CHECK("rl_rpe_stats");
    return RL_OK;
}

static int rpe_bwd_fill(RpeBwdParams* q, const rl_pool_desc* d, const float* G, const char* who) {
    RL_REQUIRE(d && G && d->idx && d->points > 0 && d->n > 0 && d->points % d->n == 0, RL_ERR_ARGS, "%s: bad descriptor", who);
    RL_REQUIRE(d->nbr_k == 16, RL_ERR_UNSUPPORTED, "%s: 16 neighbours expected (got %d)", who, d->nbr_k);
    RL_REQUIRE(d->u_source == 1 || d->u_source == 2, RL_ERR_ARGS, "%s: u_source must be 1 or 2", who);
    RL_REQUIRE(d->d == 16 || d->d == 32 || d->d == 64, RL_ERR_UNSUPPORTED, "%s: d must be 16, 32 or 64 (got %d)", who, d->d);
    RL_REQUIRE(d->xyz && d->nbr_d2 && d->xyz_bstride >= d->n && d->W1 && d->b1 && d->scale1 && d->shift1 && d->mean1 && d->invstd1,
               RL_ERR_ARGS, "%s: incomplete rpe branch (stage 1 and its BatchNorm records)", who);
    RL_REQUIRE(d->u_source < 2 || (d->W2 && d->b2 && d->scale2 && d->shift2 && d->mean2 && d->invstd2), RL_ERR_ARGS,
               "%s: stage 2 needs W2 / b2 and its BatchNorm records", who);
    RL_REQUIRE((int64_t)d->points * 16 < (1l << 31), RL_ERR_ARGS, "%s: too many neighbourhood rows", who);
    { int rc2 = virtual_extents_ok(d, who); if (rc2) return rc2; }
    PoolParams& p = q->pp;
    p = PoolParams{};
    p.idx = d->idx; p.P = d->points; p.n = d->n; p.d = d->d; p.src = d->u_source;
    p.xyz = d->xyz; p.xyz_bstride = d->xyz_bstride; p.nbr_d2 = d->nbr_d2;
    p.xyz_w = d->xyz_width == 4 ? 4 : 3;
    RL_REQUIRE(d->xyz_width != 4 || ((uintptr_t)d->xyz & 15) == 0, RL_ERR_ARGS, "%s: padded xyz must be 16-byte aligned", who);
    p.W1 = d->W1; p.b1 = d->b1; p.sc1 = d->scale1; p.sh1 = d->shift1;
    p.W2 = d->W2; p.b2 = d->b2; p.sc2 = d->scale2; p.sh2 = d->shift2;
    p.mu1 = d->mean1; p.is1 = d->invstd1; p.mu2 = d->mean2; p.is2 = d->invstd2;
    q->G = G; q->stats = nullptr; q->coef = nullptr; q->slab = nullptr; q->GU1 = nullptr;
    q->g_bf16 = d->rows_bf16 ? 1 : 0;
    RL_REQUIRE(!q->g_bf16 || pool_terms(d->d) == 3, RL_ERR_UNSUPPORTED, "%s: bf16 gradient rows need the bf16x3 arithmetic mode", who);
    return RL_OK;
}

#define RPE_DISPATCH(KERNEL, grid, st, q)                                                                          \
    do {                                                                                                           \
        const int key_ = ((q).g_bf16 ? 200 : pool_terms((q).pp.d) == 0 ? 0 : 100) + ((q).pp.d == 16 ? 10 : (q).pp.d == 32 ? 20 : 40) + (q).pp.src; \
        switch (key_) {                                                                                            \
            RPE_CASE(KERNEL, 11, 1, 0, 1, false, grid, st, q) RPE_CASE(KERNEL, 12, 1, 0, 2, false, grid, st, q)     \
            RPE_CASE(KERNEL, 21, 2, 0, 1, false, grid, st, q) RPE_CASE(KERNEL, 22, 2, 0, 2, false, grid, st, q)     \
            RPE_CASE(KERNEL, 41, 4, 0, 1, false, grid, st, q) RPE_CASE(KERNEL, 42, 4, 0, 2, false, grid, st, q)     \
            RPE_CASE(KERNEL, 111, 1, 3, 1, false, grid, st, q) RPE_CASE(KERNEL, 112, 1, 3, 2, false, grid, st, q)   \
            RPE_CASE(KERNEL, 121, 2, 3, 1, false, grid, st, q) RPE_CASE(KERNEL, 122, 2, 3, 2, false, grid, st, q)   \
            RPE_CASE(KERNEL, 141, 4, 3, 1, false, grid, st, q) RPE_CASE(KERNEL, 142, 4, 3, 2, false, grid, st, q)   \
            RPE_CASE(KERNEL, 211, 1, 3, 1, true, grid, st, q) RPE_CASE(KERNEL, 212, 1, 3, 2, true, grid, st, q)     \
            RPE_CASE(KERNEL, 221, 2, 3, 1, true, grid, st, q) RPE_CASE(KERNEL, 222, 2, 3, 2, true, grid, st, q)     \
            RPE_CASE(KERNEL, 241, 4, 3, 1, true, grid, st, q) RPE_CASE(KERNEL, 242, 4, 3, 2, true, grid, st, q)     \
            default: RL_REQUIRE(false, RL_ERR_UNSUPPORTED, "rpe backward: no kernel for d %d, source %d (key %d)", (q).pp.d, (q).pp.src, key_); \
        }                                                                                                          \
    } while (0)
#define RPE_CASE(KERNEL, K, DT, TERMS, SRC, GB, grid, st, q) \
    case K: hipLaunchKernelGGL((KERNEL<DT, TERMS, SRC, GB>), dim3(grid), dim3(256), 0, st, q); break;

extern "C" int rl_rpe_bn_reduce(const rl_pool_desc* d, const float* G, double* stats, void* stream) {
    RpeBwdParams q;
    int rc = rpe_bwd_fill(&q, d, G, "rl_rpe_bn_reduce");
    if (rc) return rc;
    RL_REQUIRE(stats, RL_ERR_ARGS, "rl_rpe_bn_reduce: null stats");
    q.stats = stats;
    const int g = rpe_grid(q.pp.P);
    RPE_DISPATCH(vrpe_bn_reduce_kernel, g, (hipStream_t)stream, q);
    rl_note_kernel("vrpe_bn_reduce_kernel");
    RL_LAUNCH_CHECK("rl_rpe_bn_reduce");
    return RL_OK;
}

extern "C" int64_t rl_rpe_wgrad_slab_floats(int64_t points, int d, int stage) {
    const int h = d / 2;
    return (int64_t)rpe_grid(points) * ((int64_t)h * (stage == 1 ? 10 : h) + h);
}

extern "C" int rl_rpe_wgrad(const rl_pool_desc* d, const float* G, const float* coef, float* slab, int64_t slab_floats,
                            float* GU1, void* stream) {
    RpeBwdParams q;
    int rc = rpe_bwd_fill(&q, d, G, "rl_rpe_wgrad");
    if (rc) return rc;
    RL_REQUIRE(coef && slab, RL_ERR_ARGS, "rl_rpe_wgrad: null coef / slab");
    RL_REQUIRE(slab_floats >= rl_rpe_wgrad_slab_floats(d->points, d->d, d->u_source), RL_ERR_ARGS, "rl_rpe_wgrad: slab too small");
    RL_REQUIRE(d->u_source == 1 || GU1, RL_ERR_ARGS, "rl_rpe_wgrad: stage 2 needs the stage-1 gradient output");
    q.coef = coef; q.slab = slab; q.GU1 = GU1;
    const int g = rpe_grid(q.pp.P);
    RPE_DISPATCH(vrpe_wgrad_kernel, g, (hipStream_t)stream, q);
    rl_note_kernel("vrpe_wgrad_kernel");
    RL_LAUNCH_CHECK("rl_rpe_wgrad");
    return RL_OK;
}
