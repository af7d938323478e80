// rl_band_sort: a spatial order INSIDE the sampling bands of the step's random permutation.
//
// The reference draws ONE permutation of the N points per forward (modules.py:571) and sub-samples by taking prefixes of it
// (modules.py:587-598: level l keeps the first N / dec^l permuted points).  Which points a level keeps is therefore decided by the
// BAND a point falls in - [0, N/dec^L), [N/dec^L, N/dec^(L-1)), ..., [N/dec, N) - and by nothing else: the order of the points
// inside a band is free (every operator of the network is a set function of a level's points: K-NN, gathers, BatchNorm sums).
// The kernels, however, care a great deal: with a random order the 16 neighbours of a point are 16 random rows of a cloud's
// table (a 128-byte line fetched for every 32-byte row at level 0), with the points of a band in cell order they are rows the
// neighbouring points have just touched.  tools/locality_probe.py: 6.60 -> 6.43 ms per step at bs = 8, and no further gain
// beyond 4 bits per axis - so this is a STABLE counting sort by (band, 4096-cell Morton code), per cloud:
//   perm_out[b][band_start + rank] = perm[r]     for the r of that band, ordered by (cell of cloud b's point perm[r], r)
// Stable = deterministic: the result is a pure function of the coordinates and the permutation (no atomics on global memory, no
// arrival order anywhere), so training stays bitwise reproducible.  The sampled sets, and with them the function the network
// computes, are those of the reference's permutation; what changes is the order of floating-point sums (and which of several
// EXACTLY equidistant candidates a K-NN search keeps).
//
// Four launches: bounding box per cloud; per tile of 1024 band positions the cell codes + a histogram row; per (cloud, band) the
// exclusive scan (cell-major, tile-minor); per tile the ranks (wavefront match by ballots, in position order) and the scatter.
#include "rl_common.h"

namespace {

constexpr int BS_TILE = 1024;        // band positions per tile
constexpr int BS_CELLS = 4096;       // 4 bits per axis
constexpr int BS_MAXB = 8;           // bands (encoder levels + 1)
constexpr int BS_BBW = 32;           // workgroups per cloud of the bounding-box pass

struct BandPlan {
    int B, N, nbands;
    int edge[BS_MAXB + 1];           // band k = positions [edge[k], edge[k + 1])
    int tile0[BS_MAXB + 1];          // first tile of band k inside a cloud; tile0[nbands] = tiles per cloud
};

struct BandParams {
    BandPlan pl;
    const float* rows; long row_stride;      // point (b, i) at rows + (b * N + i) * row_stride: x, y, z first
    const int64_t* perm;                     // (N) shared permutation
    float* bbox;                             // [B][BS_BBW][6] partial boxes: min xyz, max xyz
    unsigned short* cell;                    // [B][N] cell code of position r
    int* hist;                               // [B * tiles per cloud][BS_CELLS] counts, then offsets inside the band
    int64_t* out;                            // [B][N]
};

// BS_BBW workgroups per cloud leave partial boxes (a point per lane and trip: one workgroup per cloud walked 40 dependent trips)
__global__ __launch_bounds__(256) void bs_bbox_kernel(const BandParams p) {
    __shared__ float red[6][4];
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const float* base = p.rows + (long)b * p.pl.N * p.row_stride;
    for (int i = blockIdx.x * 256 + t; i < p.pl.N; i += BS_BBW * 256) {
        const float* q = base + (long)i * p.row_stride;
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], q[a]); hi[a] = fmaxf(hi[a], q[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    if (t < 6) {
        float v = red[t][0];
        for (int w = 1; w < 4; ++w) v = t < 3 ? fminf(v, red[t][w]) : fmaxf(v, red[t][w]);
        p.bbox[((long)b * BS_BBW + blockIdx.x) * 6 + t] = v;
    }
}
// the cloud's box from its partial boxes (min / max: any order gives the same bits)
__device__ __forceinline__ void cloud_box(const BandParams& p, int b, float (&lo)[3], float (&sc)[3]) {
    const float* bb = p.bbox + (long)b * BS_BBW * 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = bb[a], h = bb[3 + a];
        for (int w = 1; w < BS_BBW; ++w) { l = fminf(l, bb[w * 6 + a]); h = fmaxf(h, bb[w * 6 + 3 + a]); }
        lo[a] = l;
        const float ext = h - l;
        sc[a] = ext > 0.f ? 16.f / ext : 0.f;
    }
}

// tile id inside a cloud -> (band, first position, positions in the tile)
__device__ __forceinline__ void tile_span(const BandPlan& pl, int tc, int& band, int& r0, int& len) {
    band = 0;
    while (band + 1 < pl.nbands && tc >= pl.tile0[band + 1]) ++band;
    r0 = pl.edge[band] + (tc - pl.tile0[band]) * BS_TILE;
    const int end = pl.edge[band + 1];
    len = end - r0 < BS_TILE ? end - r0 : BS_TILE;
}

__device__ __forceinline__ unsigned spread4(unsigned v) {       // 4 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

__global__ __launch_bounds__(256) void bs_hist_kernel(const BandParams p) {
    __shared__ int hist[BS_CELLS];
    const int tiles = p.pl.tile0[p.pl.nbands];
    const int b = blockIdx.x / tiles, tc = blockIdx.x % tiles, t = threadIdx.x;
    int band, r0, len;
    tile_span(p.pl, tc, band, r0, len);
    for (int i = t; i < BS_CELLS; i += 256) hist[i] = 0;
    __syncthreads();
    float lo[3], sc[3];
    cloud_box(p, b, lo, sc);
    const float* base = p.rows + (long)b * p.pl.N * p.row_stride;
    for (int e = t; e < len; e += 256) {
        const int r = r0 + e;
        const float* q = base + p.perm[r] * p.row_stride;
        unsigned code = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int c = (int)((q[a] - lo[a]) * sc[a]);
            c = c < 0 ? 0 : (c > 15 ? 15 : c);
            code |= spread4((unsigned)c) << a;
        }
        p.cell[(long)b * p.pl.N + r] = (unsigned short)code;
        atomicAdd(&hist[code], 1);              // LDS, counts only: the order of the adds does not matter
    }
    __syncthreads();
    int* row = p.hist + (long)blockIdx.x * BS_CELLS;
    for (int i = t; i < BS_CELLS; i += 256) row[i] = hist[i];
}

// one workgroup per (cloud, band): counts -> offsets inside the band, cell-major / tile-minor (= stable over the tiles).
// A thread owns four consecutive cells (one int4 per tile row); the rows of eight tiles are requested together - walked row by row
// the big band's thirty rows were thirty dependent round trips (75 us).
__global__ __launch_bounds__(1024) void bs_scan_kernel(const BandParams p) {
    __shared__ int part[1024];
    const int tiles = p.pl.tile0[p.pl.nbands];
    const int b = blockIdx.x / p.pl.nbands, band = blockIdx.x % p.pl.nbands, t = threadIdx.x;
    const int t0 = p.pl.tile0[band], nt = p.pl.tile0[band + 1] - t0;
    int4* H = reinterpret_cast<int4*>(p.hist + ((long)b * tiles + t0) * BS_CELLS) + t;      // row k at H + k * (BS_CELLS / 4)
    constexpr int RS = BS_CELLS / 4, U = 8;
    int4 tot = make_int4(0, 0, 0, 0);
    for (int k0 = 0; k0 < nt; k0 += U) {
        int4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = k0 + u < nt ? H[(long)(k0 + u) * RS] : make_int4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) { tot.x += v[u].x; tot.y += v[u].y; tot.z += v[u].z; tot.w += v[u].w; }
    }
    const int mine = tot.x + tot.y + tot.z + tot.w;
    part[t] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int u = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += u;
        __syncthreads();
    }
    // running offsets of this thread's four cells: cell c starts where cell c - 1 ends; inside a cell tile by tile
    int4 run;
    run.x = part[t] - mine; run.y = run.x + tot.x; run.z = run.y + tot.y; run.w = run.z + tot.z;
    for (int k0 = 0; k0 < nt; k0 += U) {
        int4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = k0 + u < nt ? H[(long)(k0 + u) * RS] : make_int4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < nt) H[(long)(k0 + u) * RS] = run;
            run.x += v[u].x; run.y += v[u].y; run.z += v[u].z; run.w += v[u].w;
        }
    }
}

// one wavefront per tile: positions in order, 64 at a time; equal cells inside a group are ranked by lane (ballot match)
__global__ __launch_bounds__(64) void bs_scatter_kernel(const BandParams p) {
    __shared__ int cur[BS_CELLS];
    const int tiles = p.pl.tile0[p.pl.nbands];
    const int b = blockIdx.x / tiles, tc = blockIdx.x % tiles, lane = threadIdx.x;
    int band, r0, len;
    tile_span(p.pl, tc, band, r0, len);
    const int* row = p.hist + (long)blockIdx.x * BS_CELLS;
    for (int i = lane; i < BS_CELLS; i += 64) cur[i] = row[i];
    __syncthreads();                             // (one wavefront: a cheap barrier, and the compiler's licence to see other lanes' writes)
    const unsigned short* cells = p.cell + (long)b * p.pl.N;
    int64_t* out = p.out + (long)b * p.pl.N + p.pl.edge[band];
    // the tile's codes and permutation entries are requested up front (16 + 16 loads per lane): the 16 groups below are a
    // dependent chain through the LDS cursors, and with a global round trip inside every link it was 17 us long
    constexpr int G = BS_TILE / 64;
    unsigned codes[G];
    int64_t pv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int e = g * 64 + lane;
        codes[g] = e < len ? cells[r0 + e] : 0xFFFFu;          // (0xFFFF: no live cell - 12 bits)
        pv[g] = e < len ? p.perm[r0 + e] : 0;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g * 64 >= len) break;                               // wavefront-uniform
        const bool live = g * 64 + lane < len;
        const unsigned code = codes[g];
        // lanes of this group with the same code
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 12; ++bit) {
            const unsigned long long m = __ballot((code >> bit) & 1u);
            same &= ((code >> bit) & 1u) ? m : ~m;
        }
        if (live) {
            const unsigned long long below = same & ((1ull << lane) - 1ull);
            const int pos = cur[code] + __popcll(below);
            out[pos] = pv[g];
            if (below == 0ull) cur[code] += __popcll(same);      // the first lane of the match advances the cell's cursor
        }
        __syncthreads();
    }
}

int make_plan(BandPlan* pl, int B, int N, const int* edges, int nbands, const char* who) {
    RL_REQUIRE(B > 0 && B <= 65535 && N > 0 && nbands >= 1 && nbands <= BS_MAXB && edges, RL_ERR_ARGS,
               "%s: bad sizes (1 .. 65535 clouds, 1 .. %d bands)", who, BS_MAXB);
    RL_REQUIRE(edges[0] == 0 && edges[nbands] == N, RL_ERR_ARGS, "%s: the bands must cover [0, N)", who);
    pl->B = B; pl->N = N; pl->nbands = nbands;
    int tiles = 0;
    for (int k = 0; k < nbands; ++k) {
        RL_REQUIRE(edges[k + 1] > edges[k], RL_ERR_ARGS, "%s: empty or unordered band %d", who, k);
        pl->edge[k] = edges[k];
        pl->tile0[k] = tiles;
        tiles += (edges[k + 1] - edges[k] + BS_TILE - 1) / BS_TILE;
    }
    pl->edge[nbands] = N;
    pl->tile0[nbands] = tiles;
    for (int k = nbands + 1; k <= BS_MAXB; ++k) { pl->edge[k] = N; pl->tile0[k] = tiles; }
    RL_REQUIRE((int64_t)B * tiles < (1 << 30), RL_ERR_ARGS, "%s: too many tiles", who);
    return RL_OK;
}

inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

}  // namespace

// bytes of scratch rl_band_sort needs for B clouds of N points in `nbands` bands with these edges (nbands + 1 ints, 0 ... N)
extern "C" int64_t rl_band_sort_workspace_bytes(int B, int N, const int* edges, int nbands) {
    BandPlan pl;
    if (make_plan(&pl, B, N, edges, nbands, "rl_band_sort_workspace_bytes") != RL_OK) return -1;
    const int64_t tiles = (int64_t)B * pl.tile0[nbands];
    return align256((int64_t)B * BS_BBW * 6 * 4) + align256((int64_t)B * N * 2) + align256(tiles * BS_CELLS * 4);
}

extern "C" int rl_band_sort(const float* rows, int64_t row_stride, const int64_t* perm, int B, int N, const int* edges, int nbands,
                            int64_t* perm_out, void* workspace, int64_t workspace_bytes, void* stream) {
    RL_REQUIRE(rows && perm && perm_out && workspace && row_stride >= 3, RL_ERR_ARGS, "rl_band_sort: bad arguments");
    BandParams p;
    int rc = make_plan(&p.pl, B, N, edges, nbands, "rl_band_sort");
    if (rc) return rc;
    const int64_t need = rl_band_sort_workspace_bytes(B, N, edges, nbands);
    RL_REQUIRE(workspace_bytes >= need, RL_ERR_ARGS, "rl_band_sort: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)need);
    RL_REQUIRE(((uintptr_t)workspace & 255) == 0, RL_ERR_ARGS, "rl_band_sort: workspace must be 256-byte aligned");
    char* ws = (char*)workspace;
    p.rows = rows; p.row_stride = row_stride; p.perm = perm; p.out = perm_out;
    p.bbox = (float*)ws; ws += align256((int64_t)B * BS_BBW * 6 * 4);
    p.cell = (unsigned short*)ws; ws += align256((int64_t)B * N * 2);
    p.hist = (int*)ws;
    const int tiles = B * p.pl.tile0[nbands];
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bs_bbox_kernel, dim3(BS_BBW, B), dim3(256), 0, st, p);
    hipLaunchKernelGGL(bs_hist_kernel, dim3(tiles), dim3(256), 0, st, p);
    hipLaunchKernelGGL(bs_scan_kernel, dim3(B * nbands), dim3(1024), 0, st, p);
    hipLaunchKernelGGL(bs_scatter_kernel, dim3(tiles), dim3(64), 0, st, p);
    rl_note_kernel("bs_scatter_kernel");
    RL_LAUNCH_CHECK("rl_band_sort");
    return RL_OK;
}
