// Exact K nearest neighbours on gfx950 - replaces knn_tpk.knn (reference
// randlanet/utils/src/knn.cpp:43-61, nanoflann kd-tree) and the python searches of
// randlanet/utils/knn.py.
//
// Arithmetic contract (bit-exact with the reference C++): d2 = ((dx*dx)+(dy*dy))+(dz*dz),
// dx = q - s, IEEE fp32, no FMA (nanoflann.hpp:488-497) -> VALU kernel, __fmul_rn/__fadd_rn.
// Rows ascend by the 64-bit key (bits(d2) << 32 | index): d2 >= +0 so its bit pattern is
// monotone, and exact-distance ties resolve to the lowest index, independent of scan order.
//
// Two searches with identical results: the tiled scan below (small supports, or no workspace) and
// the uniform-grid search of knn_grid.hip (supports >= 512 points when a workspace is given).
//
// Tiled brute-force search: a workgroup owns 256 queries (one per lane, 4 wavefronts); support
// points stream through LDS in tiles of 1024 (x,y,z,pad) read back as wave-wide broadcasts
// (one ds_read_b128, conflict-free); each lane keeps its K best keys sorted in registers and
// the insertion (a min/max ripple through the list) is entered only by wavefronts in which
// some lane beats its current K-th key.
#include "rl_common.h"

namespace {

constexpr int KNN_BLOCK = 256;
constexpr int KNN_TILE = 1024;

template <int KMAX>
__global__ __launch_bounds__(KNN_BLOCK) void knn_brute_kernel(
    const float* __restrict__ S, long s_bs, const float* __restrict__ Q, long q_bs, int Ns,
    int Nq, int k, int32_t* __restrict__ idx32, int64_t* __restrict__ idx64,
    float* __restrict__ d2out) {
    __shared__ float4 tile[KNN_TILE];
    const int b = blockIdx.y;
    const int qi = blockIdx.x * KNN_BLOCK + threadIdx.x;
    const float* Sb = S + (size_t)b * s_bs * 3;
    const float* Qb = Q + (size_t)b * q_bs * 3;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (qi < Nq) {
        qx = Qb[(size_t)qi * 3 + 0];
        qy = Qb[(size_t)qi * 3 + 1];
        qz = Qb[(size_t)qi * 3 + 2];
    }
    unsigned long long best[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) best[s] = ~0ull;

    for (int t0 = 0; t0 < Ns; t0 += KNN_TILE) {
        const int cnt = min(KNN_TILE, Ns - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += KNN_BLOCK) {
            const float* p = Sb + (size_t)(t0 + j) * 3;
            tile[j] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float4 c = tile[j];
            const float dx = __fsub_rn(qx, c.x), dy = __fsub_rn(qy, c.y), dz = __fsub_rn(qz, c.z);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            unsigned long long key =
                ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(t0 + j);
            if (key < best[KMAX - 1]) {
#pragma unroll
                for (int s = 0; s < KMAX; ++s) {
                    const bool lt = key < best[s];
                    const unsigned long long lo = lt ? key : best[s];
                    const unsigned long long hi = lt ? best[s] : key;
                    best[s] = lo;
                    key = hi;
                }
            }
        }
    }
    if (qi < Nq) {
        const size_t o = ((size_t)b * Nq + qi) * k;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            if (s < k) {
                const unsigned id = (unsigned)(best[s] & 0xffffffffull);
                if (idx32) idx32[o + s] = (int32_t)id;
                if (idx64) idx64[o + s] = (int64_t)id;
                d2out[o + s] = __uint_as_float((unsigned)(best[s] >> 32));
            }
        }
    }
}

template <int KMAX>
int launch(const float* S, long s_bs, const float* Q, long q_bs, int B, int Ns, int Nq, int k,
           int32_t* i32, int64_t* i64, float* d2, hipStream_t st) {
    dim3 grid(rl_cdiv(Nq, KNN_BLOCK), B);
    hipLaunchKernelGGL((knn_brute_kernel<KMAX>), grid, dim3(KNN_BLOCK), 0, st, S, s_bs, Q, q_bs, Ns,
                       Nq, k, i32, i64, d2);
    rl_note_kernel("knn_brute_kernel");
    RL_LAUNCH_CHECK("rl_knn");
    return RL_OK;
}

}  // namespace

int64_t rl_knn_grid_workspace_bytes(int B, int Ns, int k);
int rl_knn_grid_run(const float* S, long s_bs, const float* Q, long q_bs, int B, int Ns, int Nq, int k, int32_t* i32,
                    int64_t* i64, float* d2, void* workspace, int64_t workspace_bytes, hipStream_t st);

namespace {

constexpr int KNN_GRID_MIN_SUPPORT = 512;  // below this the tiled scan is cheaper than building a grid

int knn_dispatch(const float* S, long s_bs, const float* Q, long q_bs, int B, int Ns, int Nq, int k,
                 int32_t* i32, int64_t* i64, float* d2, void* workspace, int64_t workspace_bytes, void* stream) {
    RL_REQUIRE(B >= 0 && Ns >= 0 && Nq >= 0 && k > 0, RL_ERR_ARGS, "rl_knn: bad sizes B=%d Ns=%d Nq=%d k=%d", B, Ns, Nq, k);
    RL_REQUIRE(Ns >= k, RL_ERR_FEW_SUPPORT, "Not enough points in support to find %d neighboors", k);
    RL_REQUIRE(k <= RL_KNN_MAX_K, RL_ERR_UNSUPPORTED, "rl_knn: k=%d exceeds RL_KNN_MAX_K=%d", k, RL_KNN_MAX_K);
    if (B == 0 || Nq == 0) return RL_OK;
    RL_REQUIRE(S && Q && d2 && (i32 || i64), RL_ERR_ARGS, "rl_knn: null pointer");
    RL_REQUIRE(B <= 65535, RL_ERR_ARGS, "rl_knn: B=%d too large", B);
    hipStream_t st = (hipStream_t)stream;
    if (workspace != nullptr && Ns >= KNN_GRID_MIN_SUPPORT)
        return rl_knn_grid_run(S, s_bs, Q, q_bs, B, Ns, Nq, k, i32, i64, d2, workspace, workspace_bytes, st);
    if (k == 1) return launch<1>(S, s_bs, Q, q_bs, B, Ns, Nq, k, i32, i64, d2, st);
    if (k <= 4) return launch<4>(S, s_bs, Q, q_bs, B, Ns, Nq, k, i32, i64, d2, st);
    if (k <= 8) return launch<8>(S, s_bs, Q, q_bs, B, Ns, Nq, k, i32, i64, d2, st);
    if (k <= 16) return launch<16>(S, s_bs, Q, q_bs, B, Ns, Nq, k, i32, i64, d2, st);
    if (k <= 32) return launch<32>(S, s_bs, Q, q_bs, B, Ns, Nq, k, i32, i64, d2, st);
    return launch<64>(S, s_bs, Q, q_bs, B, Ns, Nq, k, i32, i64, d2, st);
}

}  // namespace

extern "C" int64_t rl_knn_workspace_bytes(int B, int Ns, int Nq, int k) {
    (void)Nq;
    if (B <= 0 || Ns < KNN_GRID_MIN_SUPPORT || k <= 0) return 0;
    return rl_knn_grid_workspace_bytes(B, Ns, k);
}

extern "C" int rl_knn_f32(const float* support, const float* query, int B, int Ns, int Nq, int k,
                          int64_t* idx_out, float* d2_out, void* workspace, int64_t workspace_bytes, void* stream) {
    return knn_dispatch(support, Ns, query, Nq, B, Ns, Nq, k, nullptr, idx_out, d2_out, workspace, workspace_bytes, stream);
}

extern "C" int rl_knn_i32(const float* support, int64_t support_bstride, const float* query,
                          int64_t query_bstride, int B, int Ns, int Nq, int k, int32_t* idx_out,
                          float* d2_out, void* workspace, int64_t workspace_bytes, void* stream) {
    RL_REQUIRE(support_bstride >= Ns && query_bstride >= Nq, RL_ERR_ARGS, "rl_knn_i32: batch stride smaller than the cloud");
    return knn_dispatch(support, support_bstride, query, query_bstride, B, Ns, Nq, k, idx_out, nullptr, d2_out, workspace,
                        workspace_bytes, stream);
}
