// Exact K nearest neighbours through a uniform grid - the large-cloud path of rl_knn_*.
//
// Same contract as the brute-force kernel in knn.hip (reference knn_tpk.knn, knn.cpp:43-61):
// d2 = ((dx*dx)+(dy*dy))+(dz*dz) in un-fused fp32, rows ascending by (d2, index).  The result is
// defined by the 64-bit key (bits(d2) << 32 | index), so it does not depend on the order in which
// candidates are visited: the grid only prunes, it never approximates.
//
// Per call and cloud:  bounding box -> grid geometry (~max(2, k/2) points per cell) -> counting
// sort of the support points by cell (histogram, scan, scatter into (x,y,z,index) records of
// 16 B, read back with one dwordx4 load per candidate) -> one lane per query walks the cells in
// Chebyshev rings around its own cell, keeping its K best keys in registers, and stops when the
// K-th distance lies strictly inside the nearest face of the visited block that still has
// unvisited cells behind it (ties beyond that face could carry a lower index, so the test is
// strict and carries a rounding slack).  For a self-search the lanes walk the queries in cell
// order, so a wavefront's lanes share cells and cache lines.
#include "rl_common.h"
#include <stdlib.h>

namespace {

// candidates per round trip of a lane (8: 170 VGPRs, 6: spills at the four-wavefront cap, 4: 16 B of scratch, 3: none; 2 is slower)
constexpr int KNN_SB = 3;


struct GridGeom {
    float lo[3];
    float inv[3];
    float h[3];
    int n[3];
    int ncell;
    float slack;
};

__device__ __forceinline__ unsigned enc(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// One search = one TASK (support prefix, query prefix, k); a launch set handles up to KNN_MAX_TASKS tasks
// of B clouds each: blockIdx.y = task * B + cloud.  A forward pass needs eight searches that depend on
// nothing but the coordinates (encoder K-NN on four levels, decoder 1-NN on four): batching them makes
// the small levels - a few wavefronts each, latency-bound - run beside the big one instead of after it.
constexpr int KNN_MAX_TASKS = 8;

struct KTask {
    const float* S;
    long s_bs;
    const float* Q;
    long q_bs;
    int Ns, Nq, k, self_mode;
    int32_t* idx32;
    int64_t* idx64;
    float* d2;
    GridGeom* geom;   // [B]
    unsigned* bbox;   // [B][6]
    int* count;       // [B][cstride]
    int* start;
    int* cursor;
    float4* sorted;   // [B][Ns]
    int cstride, maxcells;
    float per_cell;
    int lanes;        // lanes that share a query in the ring walk: 1, 2 or 4 (k <= 16 only)
};

struct KMulti {
    int ntasks, B;
    int xcd;               // grid_query_kernel: workgroups of an XCD serve the clouds b = XCD mod B (see there)
    KTask t[KNN_MAX_TASKS];
};

// Workgroup ids (x + gridDim.x * y) go round-robin over the 8 XCDs, each with its own L2.  With y = task * B + cloud every XCD
// worked on a slice of EVERY cloud.  The (x, cloud) pairs of a task are re-dealt so that the cloud is the id modulo B: for
// B | 8 the workgroups of one XCD then serve ONE cloud (B = 8) or two, whose cell table and sorted points stay in that L2.
// Used by the query kernel (242 -> 212 us per step at bs = 8); the grid build's count / scatter passes measured neutral with it
// (count 27.5 -> 30.9 us: its same-cell atomics then meet in one L2; scatter 35.4 -> 30.4) and keep the plain order.
__device__ __forceinline__ void knn_deal(const KMulti& m, int& b, int& bx) {
    if (m.xcd) {
        const int lin = (int)blockIdx.x + (int)gridDim.x * b;
        b = lin % m.B;
        bx = lin / m.B;
    }
}

__device__ __forceinline__ int kmax_of(int k) { return k == 1 ? 1 : k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64; }

// resets the bounding boxes and the cell histograms (a kernel rather than hipMemsetAsync: the whole
// search must replay from a captured hipGraph)
__global__ __launch_bounds__(256) void grid_init_kernel(const KMulti m) {
    const int task = blockIdx.y / m.B, b = blockIdx.y % m.B;
    const KTask& T = m.t[task];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 6) T.bbox[b * 6 + i] = (i < 3) ? 0xffffffffu : 0u;
    int* cnt = T.count + (size_t)b * T.cstride;
    for (int e = i; e < T.cstride; e += gridDim.x * 256) cnt[e] = 0;
}

__global__ __launch_bounds__(256) void grid_bbox_kernel(const KMulti m) {
    const int task = blockIdx.y / m.B, b = blockIdx.y % m.B;
    const KTask& T = m.t[task];
    if (blockIdx.x * 256 >= T.Ns) return;
    const float* Sb = T.S + (size_t)b * T.s_bs * 3;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = blockIdx.x * 256 + threadIdx.x; j < T.Ns; j += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = Sb[(size_t)j * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
    }
    // one set of six atomics per workgroup: hundreds of wavefronts hitting the same six words serialise in L2
    __shared__ float red[4][6];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[threadIdx.x >> 6][a] = mn[a];
            red[threadIdx.x >> 6][3 + a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float lo = fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]));
        const float hi = fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]));
        atomicMin(&T.bbox[b * 6 + a], enc(lo));
        atomicMax(&T.bbox[b * 6 + 3 + a], enc(hi));
    }
}

__global__ void grid_geom_kernel(const KMulti m) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m.ntasks * m.B) return;
    const int task = c / m.B, b = c % m.B;
    const KTask& T = m.t[task];
    GridGeom g;
    float ext[3];
    double vol = 1.0;
    int flat = 0;
    float extent = 0.f;
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = dec(T.bbox[b * 6 + a]);
        const float hi = dec(T.bbox[b * 6 + 3 + a]);
        ext[a] = hi - g.lo[a];
        if (!(ext[a] > 0.f)) { ext[a] = 0.f; flat++; } else vol *= (double)ext[a];
        extent = fmaxf(extent, ext[a]);
    }
    double want = (double)T.Ns / (double)T.per_cell;
    if (want < 1.0) want = 1.0;
    const double side = (flat == 3) ? 1.0 : pow(vol / want, 1.0 / (double)(3 - flat));
    long total = 1;
    for (int a = 0; a < 3; ++a) {
        int n = (ext[a] > 0.f && side > 0.0) ? (int)ceil((double)ext[a] / side) : 1;
        if (n < 1) n = 1;
        if (n > 1024) n = 1024;
        g.n[a] = n;
        total *= n;
    }
    while (total > T.maxcells) {  // ceil() overshoot on thin clouds: halve the longest axis
        int a = 0;
        if (g.n[1] > g.n[a]) a = 1;
        if (g.n[2] > g.n[a]) a = 2;
        g.n[a] = (g.n[a] + 1) / 2;
        total = (long)g.n[0] * g.n[1] * g.n[2];
    }
    for (int a = 0; a < 3; ++a) {
        g.h[a] = (ext[a] > 0.f) ? ext[a] / (float)g.n[a] : 1.f;
        g.inv[a] = 1.f / g.h[a];
    }
    g.ncell = (int)total;
    g.slack = 1e-5f * extent + 1e-30f;
    T.geom[b] = g;
}

__device__ __forceinline__ void cell_coords(const GridGeom& g, float x, float y, float z, int c[3]) {
    const float p[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int ci = (int)floorf((p[a] - g.lo[a]) * g.inv[a]);
        ci = ci < 0 ? 0 : ci;
        ci = ci >= g.n[a] ? g.n[a] - 1 : ci;
        c[a] = ci;
    }
}
__device__ __forceinline__ int cell_id(const GridGeom& g, const int c[3]) {
    return (c[2] * g.n[1] + c[1]) * g.n[0] + c[0];
}

__global__ __launch_bounds__(256) void grid_count_kernel(const KMulti m) {
    const int task = blockIdx.y / m.B, b = blockIdx.y % m.B;
    const KTask& T = m.t[task];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= T.Ns) return;
    const GridGeom g = T.geom[b];
    const float* p = T.S + ((size_t)b * T.s_bs + j) * 3;
    int c[3];
    cell_coords(g, p[0], p[1], p[2], c);
    atomicAdd(&T.count[(size_t)b * T.cstride + cell_id(g, c)], 1);
}

// exclusive scan of count[0..ncell) -> start[0..ncell], cursor copy; one workgroup per cloud
__global__ __launch_bounds__(1024) void grid_scan_kernel(const KMulti m) {
    __shared__ int part[1024];
    const int task = blockIdx.x / m.B, b = blockIdx.x % m.B, t = threadIdx.x;
    const KTask& T = m.t[task];
    const int ncell = T.geom[b].ncell;
    const int* cnt = T.count + (size_t)b * T.cstride;
    int* st = T.start + (size_t)b * T.cstride;
    int* cu = T.cursor + (size_t)b * T.cstride;
    const int chunk = (ncell + 1023) / 1024;
    const int lo = t * chunk, hi = min(ncell, lo + chunk);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;  // exclusive prefix of this chunk
    for (int i = lo; i < hi; ++i) {
        st[i] = run;
        cu[i] = run;
        run += cnt[i];
    }
    if (t == 1023) st[ncell] = part[1023];
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const KMulti m) {
    const int task = blockIdx.y / m.B, b = blockIdx.y % m.B;
    const KTask& T = m.t[task];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= T.Ns) return;
    const GridGeom g = T.geom[b];
    const float* p = T.S + ((size_t)b * T.s_bs + j) * 3;
    int c[3];
    cell_coords(g, p[0], p[1], p[2], c);
    const int pos = atomicAdd(&T.cursor[(size_t)b * T.cstride + cell_id(g, c)], 1);
    T.sorted[(size_t)b * T.Ns + pos] = make_float4(p[0], p[1], p[2], __int_as_float(j));
}

// Scan candidates [begin, end).  Loads are issued eight at a time before any of them is used: with
// ~8 points per cell one cell costs one memory round trip instead of eight (at small clouds there
// is a single wavefront per SIMD and nothing else hides that latency).
// sorted insertion of one key (ascending (d2, index) order; a key that is not smaller than the last entry falls out)
// A key is (float bits of d2) << 32 | index.  d2 >= 0 is finite, so the high word is below 0x7f800000 and the 64 bits,
// read as an IEEE double, are a non-negative finite number (or a denormal / zero) whose floating-point order IS the
// unsigned integer order of the keys.  One step of the network is then v_min_f64 + v_max_f64 (full rate, two
// instructions) instead of a 64-bit compare and four selects - the network is most of this kernel's instruction stream.
// Empty slots hold +infinity as a double (above every key in both orders; the all-ones pattern would be a NaN, which
// min / max drop).  Inline asm: the builtins canonicalise their inputs first (a third instruction per step); these
// inputs are never NaNs and fp64 denormals are always preserved on this target, so the raw instructions return one
// of their operands bit for bit.
constexpr unsigned long long KEY_EMPTY = 0x7FF0000000000000ull;
template <int KMAX>
__device__ __forceinline__ void insert_key(unsigned long long key, unsigned long long (&best)[KMAX]) {
    double kd = __builtin_bit_cast(double, key);
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        const double bs = __builtin_bit_cast(double, best[s]);
        double lo, hi;
        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(kd), "v"(bs));
        asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(kd), "v"(bs));
        best[s] = __builtin_bit_cast(unsigned long long, lo);
        kd = hi;
    }
}

// One batch of SB candidates: keys for all of them first, then insertions only for those that beat the lane's current
// K-th entry.  The insertion network is ~100 instructions and a wavefront runs it whenever ANY lane needs it: done
// per candidate slot that is practically always; done per "next passing candidate of each lane" the trip count is the
// largest number of passing candidates of any lane in the batch (2-3 of 8 instead of 8 of 8).
template <int KMAX, int SB>
__device__ __forceinline__ void take_batch(const float4 (&cand)[SB], unsigned valid, float qx, float qy, float qz,
                                           unsigned long long (&best)[KMAX]) {
    unsigned long long key[SB];
    unsigned pass = 0;
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const float4 c = cand[i];
        const float dx = __fsub_rn(qx, c.x), dy = __fsub_rn(qy, c.y), dz = __fsub_rn(qz, c.z);
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        key[i] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(c.w);
        if (((valid >> i) & 1u) && key[i] < best[KMAX - 1]) pass |= 1u << i;
    }
    while (pass) {
        const int i = __ffs(pass) - 1;
        pass &= pass - 1;
        unsigned long long kk = key[0];
#pragma unroll
        for (int j = 1; j < SB; ++j)
            if (j == i) kk = key[j];
        insert_key<KMAX>(kk, best);
    }
}

template <int KMAX>
__device__ __forceinline__ void scan_range(const float4* __restrict__ pts, int begin, int end, float qx, float qy,
                                           float qz, unsigned long long (&best)[KMAX]) {
    constexpr int SB = KNN_SB;
    for (int t = begin; t < end; t += SB) {
        float4 cand[SB];
        unsigned valid = 0;
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            cand[i] = pts[min(t + i, end - 1)];
            if (t + i < end) valid |= 1u << i;
        }
        take_batch<KMAX, SB>(cand, valid, qx, qy, qz, best);
    }
}

// Centre-out enumeration of the 2r+1 offsets of a ring axis: 0, -1, +1, -2, +2, ...  Rows of cells near the query are
// scanned first, so the k-th distance falls early and later candidates enter the insertion network less often (the order
// does not change the result: every cell the stopping rule requires is still visited).
__device__ __forceinline__ int centre_out(int t) { return (t & 1) ? -((t + 1) >> 1) : (t >> 1); }

// distance from coordinate q to the slab of cells `ci` along axis a (0 inside), shrunk by the
// rounding slack so that it never over-estimates the distance to a point binned into that cell
__device__ __forceinline__ float axis_gap(const GridGeom& g, int a, int ci, float q) {
    const float lo = g.lo[a] + (float)ci * g.h[a];
    const float hi = lo + g.h[a];
    float gap = fmaxf(lo - q, q - hi);   // support points never lie outside the grid (it is their bbox)
    gap -= g.slack;
    return gap > 0.f ? gap : 0.f;
}

template <int KMAX>
__device__ __forceinline__ void query_one_lane(const KTask& T, int b, int bx) {
    const int k = T.k, Nq = T.Nq, Ns = T.Ns;
    const int t = bx * 256 + threadIdx.x;
    if (t >= Nq) return;
    const int self_mode = T.self_mode;
    const float* Q = T.Q;
    const long q_bs = T.q_bs;
    int32_t* idx32 = T.idx32;
    int64_t* idx64 = T.idx64;
    float* d2out = T.d2;
    const GridGeom g = T.geom[b];
    const float4* pts = T.sorted + (size_t)b * Ns;
    const int* st = T.start + (size_t)b * T.cstride;
    float qx, qy, qz;
    int qi;
    if (self_mode) {  // queries are the support points themselves: walk them in cell order
        const float4 me = pts[t];
        qx = me.x; qy = me.y; qz = me.z;
        qi = __float_as_int(me.w);
    } else {
        const float* p = Q + ((size_t)b * q_bs + t) * 3;
        qx = p[0]; qy = p[1]; qz = p[2];
        qi = t;
    }
    int c[3];
    cell_coords(g, qx, qy, qz, c);
    unsigned long long best[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) best[s] = KEY_EMPTY;

    const float q3[3] = {qx, qy, qz};
    for (int r = 0;; ++r) {
        const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.n[2] - 1);
        const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.n[1] - 1);
        const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.n[0] - 1);
        for (int tz = 0; tz <= 2 * r; ++tz) {
            const int z = c[2] + centre_out(tz);
            if (z < z0 || z > z1) continue;
            const bool zedge = (z == c[2] - r) || (z == c[2] + r);
            const float dz = axis_gap(g, 2, z, qz);
            for (int ty = 0; ty <= 2 * r; ++ty) {
                const int y = c[1] + centre_out(ty);
                if (y < y0 || y > y1) continue;
                const bool full_row = zedge || y == c[1] - r || y == c[1] + r;
                const float dy = axis_gap(g, 1, y, qy);
                const float dyz = dy * dy + dz * dz;
                const int row = (z * g.n[1] + y) * g.n[0];
                // K-th distance so far (one look per row: cells of a row are scanned as one run below)
                unsigned long long kth = best[KMAX - 1];
                if (k != KMAX) {
#pragma unroll
                    for (int s = 0; s < KMAX; ++s)
                        if (s == k - 1) kth = best[s];
                }
                // strict test with a rounding margin: ties must be looked at
                const bool have = kth != KEY_EMPTY;
                const float lim = have ? __uint_as_float((unsigned)(kth >> 32)) * 1.00001f + 1e-30f : INFINITY;
                if (dyz > lim) continue;            // nothing in this row of cells can be closer than the K-th
                if (full_row) {
                    // the whole x-run lies on ring r.  Its cells are consecutive in the sorted array, so the run is ONE
                    // candidate range: two dependent loads per row instead of two per cell.  Cells whose lower
                    // bound already exceeds the K-th distance are cut from both ends first (arithmetic only).
                    int xa = x0, xb = x1;
                    while (xa <= xb) {
                        const float gx = axis_gap(g, 0, xa, qx);
                        if (gx * gx + dyz > lim) ++xa; else break;
                    }
                    while (xb >= xa) {
                        const float gx = axis_gap(g, 0, xb, qx);
                        if (gx * gx + dyz > lim) --xb; else break;
                    }
                    if (xa <= xb) scan_range<KMAX>(pts, st[row + xa], st[row + xb + 1], qx, qy, qz, best);
                } else {
                    // only the two end cells of the run are on ring r
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int x = e == 0 ? c[0] - r : c[0] + r;
                        if (x < x0 || x > x1 || (e == 1 && r == 0)) continue;
                        const float dxg = axis_gap(g, 0, x, qx);
                        if (dxg * dxg + dyz > lim) continue;
                        scan_range<KMAX>(pts, st[row + x], st[row + x + 1], qx, qy, qz, best);
                    }
                }
            }
        }
        const bool covers = (c[0] - r <= 0) && (c[0] + r >= g.n[0] - 1) && (c[1] - r <= 0) && (c[1] + r >= g.n[1] - 1) &&
                            (c[2] - r <= 0) && (c[2] + r >= g.n[2] - 1);
        if (covers) break;
        unsigned long long kth = best[KMAX - 1];
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
            if (s == k - 1) kth = best[s];
        if (kth != KEY_EMPTY) {
            float bound = INFINITY;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (c[a] - r > 0) bound = fminf(bound, q3[a] - (g.lo[a] + (float)(c[a] - r) * g.h[a]));
                if (c[a] + r < g.n[a] - 1) bound = fminf(bound, (g.lo[a] + (float)(c[a] + r + 1) * g.h[a]) - q3[a]);
            }
            bound -= g.slack;
            const float kd = __uint_as_float((unsigned)(kth >> 32));
            if (bound > 0.f && kd < bound * bound * 0.999999f) break;
        }
    }
    const size_t o = ((size_t)b * Nq + qi) * k;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s < k) {
            const bool empty = best[s] == KEY_EMPTY;        // fewer than k support points: index -1, distance NaN
            const unsigned id = empty ? 0xffffffffu : (unsigned)(best[s] & 0xffffffffull);
            if (idx32) idx32[o + s] = (int32_t)id;
            if (idx64) idx64[o + s] = (int64_t)id;
            d2out[o + s] = __uint_as_float(empty ? 0xffffffffu : (unsigned)(best[s] >> 32));
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// Four lanes per query (k <= 16).  With one lane per query the 163840-query level of config A is 2.5 wavefronts
// per SIMD walking long chains of dependent loads: latency-bound with nothing to hide it.  Here lanes 4q..4q+3 of a
// wavefront share query q: they walk the same rings / rows (uniform control flow inside the quad), each scans
// every fourth candidate of a run into its OWN sorted top-K, and the quad agrees on the pruning bound
//     B = min( min_j kth_j ,  max_j best_j[ceil(k/4) - 1] )
// (the union's k-th best is <= every lane's own k-th, and <= the largest of the four ceil(k/4)-th entries because
// the four lists then hold >= k entries below it; candidates are interleaved, so the second bound is tight).
// B only ever over-estimates the true k-th distance: the quad visits a superset of the cells the exact bound would,
// results stay exact.  At the end the four disjoint sorted lists are merged by RANK through LDS: the position of a
// key in the union = its index in its own list + the number of smaller keys in the other three (binary search).
// -------------------------------------------------------------------------------------------------------------
template <int LANES>
__device__ __forceinline__ unsigned long long group_min(unsigned long long v) {
#pragma unroll
    for (int o = 1; o < LANES; o <<= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}
template <int LANES>
__device__ __forceinline__ unsigned long long group_max(unsigned long long v) {
#pragma unroll
    for (int o = 1; o < LANES; o <<= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w > v ? w : v;
    }
    return v;
}

template <int KMAX, int LANES>
__device__ __forceinline__ unsigned long long group_bound(const unsigned long long (&best)[KMAX], int k) {
    const int q = (k + LANES - 1) / LANES - 1;
    unsigned long long own_k = best[KMAX - 1], own_q = best[0];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s == k - 1) own_k = best[s];
        if (s == q) own_q = best[s];
    }
    const unsigned long long a = group_min<LANES>(own_k), b = group_max<LANES>(own_q);
    return a < b ? a : b;
}

// this lane's share of the candidates [begin, end): begin + sub, begin + sub + LANES, ...
template <int KMAX, int LANES>
__device__ __forceinline__ void scan_range_n(const float4* __restrict__ pts, int begin, int end, int sub, float qx, float qy,
                                            float qz, unsigned long long (&best)[KMAX]) {
    constexpr int SB = KNN_SB;
    for (int t = begin + sub; t < end; t += LANES * SB) {
        float4 cand[SB];
        unsigned valid = 0;
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            cand[i] = pts[min(t + LANES * i, end - 1)];
            if (t + LANES * i < end) valid |= 1u << i;
        }
        take_batch<KMAX, SB>(cand, valid, qx, qy, qz, best);
    }
}

template <int KMAX, int LANES>
__device__ __forceinline__ void query_n_lanes(const KTask& T, int b, int bx, unsigned long long* lds) {
    constexpr int QPB = 256 / LANES;           // queries per workgroup
    unsigned long long (*lists)[LANES][KMAX] = reinterpret_cast<unsigned long long (*)[LANES][KMAX]>(lds);
    const int k = T.k, Nq = T.Nq, Ns = T.Ns;
    const int qslot = threadIdx.x / LANES, sub = threadIdx.x % LANES;
    const int t = bx * QPB + qslot;
    if (bx * QPB >= Nq) return;                // whole workgroup
    const bool active = t < Nq;                // same for the four lanes of a query; every lane reaches the barrier
    unsigned long long best[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) best[s] = KEY_EMPTY;
    int qi = 0;
    if (active) {
        const GridGeom g = T.geom[b];
        const float4* pts = T.sorted + (size_t)b * Ns;
        const int* st = T.start + (size_t)b * T.cstride;
        float qx, qy, qz;
        if (T.self_mode) {  // queries are the support points themselves: walk them in cell order
            const float4 me = pts[t];
            qx = me.x; qy = me.y; qz = me.z;
            qi = __float_as_int(me.w);
        } else {
            const float* p = T.Q + ((size_t)b * T.q_bs + t) * 3;
            qx = p[0]; qy = p[1]; qz = p[2];
            qi = t;
        }
        int c[3];
        cell_coords(g, qx, qy, qz, c);
        const float q3[3] = {qx, qy, qz};
        for (int r = 0;; ++r) {
            const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.n[2] - 1);
            const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.n[1] - 1);
            const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.n[0] - 1);
            for (int tz = 0; tz <= 2 * r; ++tz) {
                const int z = c[2] + centre_out(tz);
                if (z < z0 || z > z1) continue;
                const bool zedge = (z == c[2] - r) || (z == c[2] + r);
                const float dz = axis_gap(g, 2, z, qz);
                for (int ty = 0; ty <= 2 * r; ++ty) {
                    const int y = c[1] + centre_out(ty);
                    if (y < y0 || y > y1) continue;
                    const bool full_row = zedge || y == c[1] - r || y == c[1] + r;
                    const float dy = axis_gap(g, 1, y, qy);
                    const float dyz = dy * dy + dz * dz;
                    const int row = (z * g.n[1] + y) * g.n[0];
                    const unsigned long long kth = group_bound<KMAX, LANES>(best, k);
                    const float lim = kth != KEY_EMPTY ? __uint_as_float((unsigned)(kth >> 32)) * 1.00001f + 1e-30f : INFINITY;
                    if (dyz > lim) continue;
                    if (full_row) {
                        int xa = x0, xb = x1;
                        while (xa <= xb) {
                            const float gx = axis_gap(g, 0, xa, qx);
                            if (gx * gx + dyz > lim) ++xa; else break;
                        }
                        while (xb >= xa) {
                            const float gx = axis_gap(g, 0, xb, qx);
                            if (gx * gx + dyz > lim) --xb; else break;
                        }
                        if (xa <= xb) scan_range_n<KMAX, LANES>(pts, st[row + xa], st[row + xb + 1], sub, qx, qy, qz, best);
                    } else {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int x = e == 0 ? c[0] - r : c[0] + r;
                            if (x < x0 || x > x1 || (e == 1 && r == 0)) continue;
                            const float dxg = axis_gap(g, 0, x, qx);
                            if (dxg * dxg + dyz > lim) continue;
                            scan_range_n<KMAX, LANES>(pts, st[row + x], st[row + x + 1], sub, qx, qy, qz, best);
                        }
                    }
                }
            }
            const bool covers = (c[0] - r <= 0) && (c[0] + r >= g.n[0] - 1) && (c[1] - r <= 0) && (c[1] + r >= g.n[1] - 1) &&
                                (c[2] - r <= 0) && (c[2] + r >= g.n[2] - 1);
            if (covers) break;
            const unsigned long long kth = group_bound<KMAX, LANES>(best, k);
            if (kth != KEY_EMPTY) {
                float bound = INFINITY;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (c[a] - r > 0) bound = fminf(bound, q3[a] - (g.lo[a] + (float)(c[a] - r) * g.h[a]));
                    if (c[a] + r < g.n[a] - 1) bound = fminf(bound, (g.lo[a] + (float)(c[a] + r + 1) * g.h[a]) - q3[a]);
                }
                bound -= g.slack;
                const float kd = __uint_as_float((unsigned)(kth >> 32));
                if (bound > 0.f && kd < bound * bound * 0.999999f) break;
            }
        }
    }
    // merge the four disjoint sorted lists of a query by rank
#pragma unroll
    for (int s = 0; s < KMAX; ++s) lists[qslot][sub][s] = best[s];
    __syncthreads();
    if (active) {
        const size_t o = ((size_t)b * Nq + qi) * k;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            const unsigned long long key = best[s];
            if (key == KEY_EMPTY) continue;
            int rank = s;
#pragma unroll
            for (int j = 1; j < LANES; ++j) {
                const unsigned long long* L = lists[qslot][(sub + j) % LANES];
                int lo = 0;                      // number of keys in L smaller than `key` (L is sorted, sentinels last)
#pragma unroll
                for (int w = KMAX / 2; w >= 1; w >>= 1)
                    if (L[lo + w - 1] < key) lo += w;
                if (lo < KMAX && L[lo] < key) ++lo;   // KMAX a power of two: the halving above covers KMAX - 1 entries
                rank += lo;
            }
            if (rank < k) {
                const unsigned id = (unsigned)(key & 0xffffffffull);
                if (T.idx32) T.idx32[o + rank] = (int32_t)id;
                if (T.idx64) T.idx64[o + rank] = (int64_t)id;
                T.d2[o + rank] = __uint_as_float((unsigned)(key >> 32));
            }
        }
    }
}

// one launch serves every task of a k bucket, whatever its lane count: the tasks run side by side
template <int KMAX>
// Occupancy is what this kernel runs on (it waits on chains of dependent loads: cell start -> points -> next cell): with
// 8 candidates per round trip it needed 170 VGPRs = TWO wavefronts per SIMD; three candidates per trip fit 126 = FOUR
// (all searches of a bs = 8 step: 535 us -> 476 at 168 VGPRs / three wavefronts -> 418).  The K = 32 / 64 bodies would
// spill under the cap and keep their registers.
__global__ __launch_bounds__(256, KMAX <= 16 ? 4 : 1) void grid_query_kernel(const KMulti m) {
    __shared__ unsigned long long lds[KMAX <= 16 ? 256 * KMAX : 1];
    const int task = blockIdx.y / m.B;
    int b = blockIdx.y % m.B, bx = blockIdx.x;
    knn_deal(m, b, bx);
    const KTask& T = m.t[task];
    if (kmax_of(T.k) != KMAX) return;            // another instantiation of this kernel serves that task
    if constexpr (KMAX <= 16) {
        if (T.lanes == 4) { query_n_lanes<KMAX, 4>(T, b, bx, lds); return; }
        if (T.lanes == 2) { query_n_lanes<KMAX, 2>(T, b, bx, lds); return; }
    }
    query_one_lane<KMAX>(T, b, bx);
}

struct Plan {
    int maxcells, cstride;
    float per_cell;
    size_t off_geom, off_bbox, off_count, off_start, off_cursor, off_sorted, bytes;
};

Plan make_plan(int B, int Ns, int k) {
    Plan p;
    static const float occ = getenv("RL_KNN_OCC") ? (float)atof(getenv("RL_KNN_OCC")) : 0.5f;   // tuning knob (points per cell / k)
    p.per_cell = fmaxf(2.f, occ * (float)k);
    const long want = (long)((double)Ns / p.per_cell) + 1;
    p.maxcells = (int)(4 * want + 64);
    p.cstride = p.maxcells + 1;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 255) & ~(size_t)255; return at; };
    p.off_geom = take(sizeof(GridGeom) * B);
    p.off_bbox = take(sizeof(unsigned) * 6 * B);
    p.off_count = take(sizeof(int) * (size_t)p.cstride * B);
    p.off_start = take(sizeof(int) * (size_t)p.cstride * B);
    p.off_cursor = take(sizeof(int) * (size_t)p.cstride * B);
    p.off_sorted = take(sizeof(float4) * (size_t)Ns * B);
    p.bytes = o;
    return p;
}

void bind(KTask* t, const Plan& p, char* w) {
    t->geom = (GridGeom*)(w + p.off_geom);
    t->bbox = (unsigned*)(w + p.off_bbox);
    t->count = (int*)(w + p.off_count);
    t->start = (int*)(w + p.off_start);
    t->cursor = (int*)(w + p.off_cursor);
    t->sorted = (float4*)(w + p.off_sorted);
    t->cstride = p.cstride;
    t->maxcells = p.maxcells;
    t->per_cell = p.per_cell;
}

// lanes per query: the more queries a search has, the less it gains from splitting one (the split costs a weaker
// pruning bound and a merge).  At four wavefronts per SIMD one lane per query wins from 16384 queries on (bs = 8 step, all
// searches: threshold 131072 418 us, 65536 379, 16384 376, 4096 390); RL_KNN_LANES = 1 | 2 | 4 forces one choice,
// RL_KNN_LANE_THR moves the threshold
inline int lanes_for(long queries, int k) {
    static const int force = getenv("RL_KNN_LANES") ? atoi(getenv("RL_KNN_LANES")) : 0;
    if (k > 16) return 1;
    if (force == 1 || force == 2 || force == 4) return force;
    static const long thr = getenv("RL_KNN_LANE_THR") ? atol(getenv("RL_KNN_LANE_THR")) : 16384;
    return queries >= thr ? 1 : 4;
}

int run_multi(const KMulti& m_in, hipStream_t st) {
    KMulti m = m_in;
    static const bool no_xcd = getenv("RL_NO_XCD_POINTS") != nullptr;      // A/B switch shared with the pooling kernels
    m.xcd = (!no_xcd && m.B > 1 && 8 % m.B == 0) ? 1 : 0;
    for (int i = 0; i < m.ntasks; ++i) m.t[i].lanes = lanes_for((long)m.B * m.t[i].Nq, m.t[i].k);
    int maxNs = 1, maxNq = 1, maxc = 1;
    bool need[6] = {false, false, false, false, false, false};
    for (int i = 0; i < m.ntasks; ++i) {
        maxNs = max(maxNs, m.t[i].Ns);
        maxNq = max(maxNq, m.t[i].Nq);
        maxc = max(maxc, m.t[i].cstride);
        const int k = m.t[i].k;
        need[k == 1 ? 0 : k <= 4 ? 1 : k <= 8 ? 2 : k <= 16 ? 3 : k <= 32 ? 4 : 5] = true;
    }
    const int clouds = m.ntasks * m.B;
    int gi = rl_cdiv(maxc, 256);
    if (gi > 64) gi = 64;
    hipLaunchKernelGGL(grid_init_kernel, dim3(gi, clouds), dim3(256), 0, st, m);
    int gb = rl_cdiv(maxNs, 2048);   // >= 8 points per lane, at most 16 workgroups (= atomics per word) per cloud
    if (gb > 16) gb = 16;
    hipLaunchKernelGGL(grid_bbox_kernel, dim3(gb, clouds), dim3(256), 0, st, m);
    hipLaunchKernelGGL(grid_geom_kernel, dim3(rl_cdiv(clouds, 64)), dim3(64), 0, st, m);
    hipLaunchKernelGGL(grid_count_kernel, dim3(rl_cdiv(maxNs, 256), clouds), dim3(256), 0, st, m);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(clouds), dim3(1024), 0, st, m);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(rl_cdiv(maxNs, 256), clouds), dim3(256), 0, st, m);
    int gxq = 1;
    for (int i = 0; i < m.ntasks; ++i) gxq = max(gxq, rl_cdiv((long)m.t[i].Nq * m.t[i].lanes, 256));
    dim3 grid(gxq, clouds);
    if (need[0]) hipLaunchKernelGGL((grid_query_kernel<1>), grid, dim3(256), 0, st, m);
    if (need[1]) hipLaunchKernelGGL((grid_query_kernel<4>), grid, dim3(256), 0, st, m);
    if (need[2]) hipLaunchKernelGGL((grid_query_kernel<8>), grid, dim3(256), 0, st, m);
    if (need[3]) hipLaunchKernelGGL((grid_query_kernel<16>), grid, dim3(256), 0, st, m);
    if (need[4]) hipLaunchKernelGGL((grid_query_kernel<32>), grid, dim3(256), 0, st, m);
    if (need[5]) hipLaunchKernelGGL((grid_query_kernel<64>), grid, dim3(256), 0, st, m);
    rl_note_kernel("grid_query_kernel");
    RL_LAUNCH_CHECK("rl_knn(grid)");
    return RL_OK;
}

}  // namespace

int64_t rl_knn_grid_workspace_bytes(int B, int Ns, int k) { return (int64_t)make_plan(B, Ns, k).bytes; }

// Called by knn.hip's dispatcher once sizes were validated.
int rl_knn_grid_run(const float* S, long s_bs, const float* Q, long q_bs, int B, int Ns, int Nq, int k, int32_t* i32,
                    int64_t* i64, float* d2, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    const Plan p = make_plan(B, Ns, k);
    RL_REQUIRE(workspace && workspace_bytes >= (int64_t)p.bytes, RL_ERR_ARGS, "rl_knn: workspace too small (%ld < %ld bytes)",
               (long)workspace_bytes, (long)p.bytes);
    RL_REQUIRE(((uintptr_t)workspace & 255) == 0, RL_ERR_ARGS, "rl_knn: workspace must be 256-byte aligned");
    KMulti m;
    m.ntasks = 1;
    m.B = B;
    KTask& t = m.t[0];
    t.S = S; t.s_bs = s_bs; t.Q = Q; t.q_bs = q_bs; t.Ns = Ns; t.Nq = Nq; t.k = k;
    t.self_mode = (S == Q && s_bs == q_bs && Ns == Nq) ? 1 : 0;
    t.idx32 = i32; t.idx64 = i64; t.d2 = d2;
    bind(&t, p, (char*)workspace);
    return run_multi(m, st);
}

extern "C" int64_t rl_knn_multi_workspace_bytes(const rl_knn_task* tasks, int ntasks, int B) {
    if (!tasks || ntasks <= 0 || B <= 0) return 0;
    int64_t total = 0;
    for (int i = 0; i < ntasks; ++i) total += (int64_t)make_plan(B, tasks[i].Ns, tasks[i].k).bytes;
    return total;
}

extern "C" int rl_knn_multi(const rl_knn_task* tasks, int ntasks, int B, void* workspace, int64_t workspace_bytes,
                            void* stream) {
    RL_REQUIRE(tasks && ntasks > 0 && ntasks <= KNN_MAX_TASKS && B > 0 && B * ntasks <= 65535, RL_ERR_ARGS,
               "rl_knn_multi: 1..%d tasks expected", KNN_MAX_TASKS);
    RL_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, RL_ERR_ARGS, "rl_knn_multi: workspace must be 256-byte aligned");
    KMulti m;
    m.ntasks = ntasks;
    m.B = B;
    size_t off = 0;
    for (int i = 0; i < ntasks; ++i) {
        const rl_knn_task& u = tasks[i];
        RL_REQUIRE(u.support && u.query && u.idx_out && u.d2_out && u.Nq > 0 && u.k > 0, RL_ERR_ARGS, "rl_knn_multi: bad task %d", i);
        RL_REQUIRE(u.Ns >= u.k, RL_ERR_FEW_SUPPORT, "Not enough points in support to find %d neighboors", u.k);
        RL_REQUIRE(u.k <= RL_KNN_MAX_K, RL_ERR_UNSUPPORTED, "rl_knn_multi: k=%d exceeds RL_KNN_MAX_K=%d", u.k, RL_KNN_MAX_K);
        RL_REQUIRE(u.support_bstride >= u.Ns && u.query_bstride >= u.Nq, RL_ERR_ARGS, "rl_knn_multi: batch stride smaller than the cloud");
        const Plan p = make_plan(B, u.Ns, u.k);
        RL_REQUIRE((int64_t)(off + p.bytes) <= workspace_bytes, RL_ERR_ARGS, "rl_knn_multi: workspace too small");
        KTask& t = m.t[i];
        t.S = u.support; t.s_bs = u.support_bstride; t.Q = u.query; t.q_bs = u.query_bstride;
        t.Ns = u.Ns; t.Nq = u.Nq; t.k = u.k;
        t.self_mode = (u.support == u.query && u.support_bstride == u.query_bstride && u.Ns == u.Nq) ? 1 : 0;
        t.idx32 = u.idx_out; t.idx64 = nullptr; t.d2 = u.d2_out;
        bind(&t, p, (char*)workspace + off);
        off += p.bytes;
    }
    return run_multi(m, (hipStream_t)stream);
}
