// Shared host/device helpers for the gfx950 kernels behind include/rl_randlanet.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rl_randlanet.h"

// ---- error plumbing ---------------------------------------------------------------------
void rl_set_error(const char* fmt, ...);
// remembers the (main) kernel function an entry point dispatched to; read back by rl_last_kernel()
void rl_note_kernel(const char* name);

#define RL_REQUIRE(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            rl_set_error(__VA_ARGS__);   \
            return (code);               \
        }                                \
    } while (0)

#define RL_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            rl_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return RL_ERR_LAUNCH;                                                    \
        }                                                                            \
    } while (0)

// every kernel launch of the library goes through hipLaunchKernelGGL: count them (rl_launch_count: what a rocprofv3 kernel
// trace would count for the same calls - bench.py reports launches per step from it)
void rl_count_launch();
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                         \
    do {                                                                                                         \
        rl_count_launch();                                                                                       \
        kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                       \
    } while (0)

// arithmetic mode of the MFMA-heavy kernels (gemm.hip owns it, rl_set_wide_gemm): 0 fp32, 3 bf16x3, 1 bf16
int rl_wide_terms();

static inline int rl_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// The two entry points that add with fp32 atomics (rl_scatter_add_rows, rl_gemm's out2_index) are NOT part of the network's
// schedule - nothing on the path is order-dependent - and refuse to run unless the caller opts in.
#include <stdlib.h>
static inline bool rl_float_atomics_allowed() {
    const char* e = getenv("RL_ALLOW_FLOAT_ATOMICS");
    return e && e[0] == '1';
}

// Number of row blocks a row-streaming kernel uses for M rows (one partial-statistics slot per
// block); shared by the producers (gemm / reduce kernels) and the finalize kernels.
static inline int rl_row_blocks_host(long rows, int rows_per_tile) {
    long tiles = (rows + rows_per_tile - 1) / rows_per_tile;
    if (tiles < 1) tiles = 1;
    return (int)(tiles < RL_MAX_SLOTS ? tiles : RL_MAX_SLOTS);
}

// ---- device helpers ---------------------------------------------------------------------
#define RL_ACT_NONE 0
#define RL_ACT_RELU 1
#define RL_ACT_LRELU 2

__device__ __forceinline__ float rl_act(float z, int act, float slope) {
    if (act == RL_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == RL_ACT_LRELU) return z > 0.f ? z : z * slope;
    return z;
}
// derivative of the activation w.r.t. its input z (torch: relu/leaky_relu backward use z > 0)
__device__ __forceinline__ float rl_act_grad(float z, int act, float slope) {
    if (act == RL_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == RL_ACT_LRELU) return z > 0.f ? 1.f : slope;
    return 1.f;
}

// lazily-normalised operand: value = act(raw * scale[c] + shift[c]); scale == nullptr -> raw
struct RlLazy {
    const float* scale;
    const float* shift;
    int act;
    float slope;
};
__device__ __forceinline__ float rl_lazy(const RlLazy& t, float raw, int c) {
    if (t.scale == nullptr) return raw;
    return rl_act(raw * t.scale[c] + t.shift[c], t.act, t.slope);
}

// ---- storage type of a tensor: fp32, or bf16 in the bf16-storage throughput mode (rl_randlanet.h, "storage") ----------
// Pointers stay typed `float*` in the parameter blocks; BF says what the bytes are.  Indices count ELEMENTS.
typedef __bf16 rl_bf16x4 __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ float rl_ldx(const float* p, long i) {
    if constexpr (BF) return (float)reinterpret_cast<const __bf16*>(p)[i];
    else return p[i];
}
template <bool BF>
__device__ __forceinline__ void rl_stx(float* p, long i, float v) {
    if constexpr (BF) reinterpret_cast<__bf16*>(p)[i] = (__bf16)v;      // round to nearest even (v_cvt_pk_bf16_f32)
    else p[i] = v;
}
// four consecutive elements, i % 4 == 0 and the row start 16-byte (fp32) / 8-byte (bf16) aligned
template <bool BF>
__device__ __forceinline__ float4 rl_ldx4(const float* p, long i) {
    if constexpr (BF) {
        const rl_bf16x4 h = *reinterpret_cast<const rl_bf16x4*>(reinterpret_cast<const __bf16*>(p) + i);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else return *reinterpret_cast<const float4*>(p + i);
}
template <bool BF>
__device__ __forceinline__ void rl_stx4(float* p, long i, const float4 v) {
    if constexpr (BF) {
        rl_bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        *reinterpret_cast<rl_bf16x4*>(reinterpret_cast<__bf16*>(p) + i) = h;
    } else *reinterpret_cast<float4*>(p + i) = v;
}
// streaming stores: rows written once and read once by a LATER kernel (GU / DG / X_out / dS_out of the pooling backward) bypass
// the L2 allocation, which the same kernel needs for the rows it gathers 16 times over
template <bool BF>
__device__ __forceinline__ void rl_stx_nt(float* p, long i, float v) {
    if constexpr (BF) __builtin_nontemporal_store((__bf16)v, reinterpret_cast<__bf16*>(p) + i);
    else __builtin_nontemporal_store(v, p + i);
}
template <bool BF>
__device__ __forceinline__ void rl_stx4_nt(float* p, long i, const float4 v) {
    typedef float rl_f32x4 __attribute__((ext_vector_type(4)));
    if constexpr (BF) {
        rl_bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        __builtin_nontemporal_store(h, reinterpret_cast<rl_bf16x4*>(reinterpret_cast<__bf16*>(p) + i));
    } else {
        const rl_f32x4 w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<rl_f32x4*>(p + i));
    }
}
// the same with the type known at run time only (a wavefront-uniform flag: one scalar branch)
__device__ __forceinline__ float4 rl_ld4(const float* p, long i, int bf) { return bf ? rl_ldx4<true>(p, i) : rl_ldx4<false>(p, i); }
__device__ __forceinline__ float rl_ld1(const float* p, long i, int bf) { return bf ? rl_ldx<true>(p, i) : rl_ldx<false>(p, i); }
__device__ __forceinline__ void rl_st4(float* p, long i, const float4 v, int bf) {
    if (bf) rl_stx4<true>(p, i, v);
    else rl_stx4<false>(p, i, v);
}
__device__ __forceinline__ void rl_st1(float* p, long i, float v, int bf) {
    if (bf) rl_stx<true>(p, i, v);
    else rl_stx<false>(p, i, v);
}

__device__ __forceinline__ double rl_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float rl_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
