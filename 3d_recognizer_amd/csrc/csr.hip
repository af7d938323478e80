// Deterministic backward of the neighbour gathers of the RandLA-Net path.
//
// The forward gathers rows through a neighbour graph: PointFeatureAugmentation reads
// features[idx[b][i][k]] (reference randlanet/utils/modules.py:213-221) and the decoder's nearest-
// neighbour interpolation reads features[nn[b][i]] (modules.py:359-364).  The reference's backward of
// both is torch's scatter_add_ over the same index, whose order of summation is undefined; here the
// graph is transposed once per step ("who gathered from me") and every destination row sums its
// contributions in ascending source-row order - a fixed order, so training is bitwise reproducible.
//
//   rl_csr_build        : for up to 8 graphs of the same B clouds, destination (b, j) -> the list of local
//                         source rows r = i*k + kk with idx[b][i][kk] == j, ascending.  A two-level counting
//                         sort whose counters live in LDS (one global atomic per (tile, bucket) instead of one
//                         per entry: device-scope atomics run at ~10 G/s, 5 M of them cost 0.9 ms):
//                           1. tiles of 4096 entries histogram their destinations' buckets (256 destinations
//                              each) in LDS and add the totals to the cloud's bucket sizes;
//                           2. bucket sizes are scanned; every tile reserves room in each bucket with one
//                              atomic and moves its entries there, packed as (destination in bucket, row);
//                           3. one workgroup per bucket counts / scans / places its entries with LDS counters
//                              - that is `offsets` and the segments in arrival order - and every segment is
//                              then sorted by rank: short ones (<= CAP entries) in lane-private LDS, long ones
//                              (duplicate-heavy clouds) by a whole workgroup afterwards.
//                         The order entries arrive in never matters: the final per-segment sort is canonical.
//                         Graphs too large for the packing fall back to per-entry global atomics.
//   rl_segment_sum_rows : dst[(b, j), :] (=|+=) sum over the segment of src[(b, r), col0 : col0 + C].
#include "rl_common.h"

namespace {

constexpr int CSR_MAX_TASKS = 8;

struct CsrTask {
    const int32_t* idx;   // (B, n_src, k) destination index inside the cloud
    int n_src, k, n_dst;
    int32_t* offsets;     // (B, n_dst + 1) local to the cloud's entries
    int32_t* entries;     // (B, n_src*k) local source rows, segment by segment, ascending
    int32_t* tmp;         // (B, n_src*k) entries in arrival order
    int32_t* cursor;      // (B, n_dst) counts, then fill cursors
    int32_t* long_list;   // segments longer than cap: b*n_dst + j
    int32_t* long_count;
    int long_cap;
    int cap;              // 32 / 64 / 128: longest segment the lane-private LDS sort takes
    // two-level path
    int nbuckets;         // ceil(n_dst / 256)
    int ntiles;           // ceil(n_src*k / 4096)
    int32_t* bsize;       // (B, nbuckets) entries per bucket, then reserve cursors
    int32_t* bstart;      // (B, nbuckets + 1) exclusive scan
};
struct CsrMulti {
    CsrTask t[CSR_MAX_TASKS];
    int ntasks;
    int B;
    int xcd;        // the (tile / bucket, cloud) pairs of a task are re-dealt so that the cloud is the workgroup id modulo B (csr_deal)
};

// Workgroup ids (x + gridDim.x * y) go round-robin over the 8 XCDs, each with its own L2.  With y = task * B + cloud every XCD
// worked on a slice of EVERY cloud; dealt this way (B | 8) an XCD's workgroups stay with one cloud (B = 8) or two, whose index /
// bucket arrays (2.6 MB at level 0) then live in that L2: the scattered 4-byte writes of pass 2 combine there.
__device__ __forceinline__ void csr_deal(const CsrMulti& m, int& b, int& bx) {
    if (m.xcd) {
        const int lin = (int)blockIdx.x + (int)gridDim.x * b;
        b = lin % m.B;
        bx = lin / m.B;
    }
}

__global__ __launch_bounds__(256) void csr_zero_kernel(const CsrMulti m) {
    const CsrTask& T = m.t[blockIdx.y];
    const long total = (long)m.B * T.n_dst;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) T.cursor[e] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) T.long_count[0] = 0;
}

__global__ __launch_bounds__(256) void csr_count_kernel(const CsrMulti m) {
    const CsrTask& T = m.t[blockIdx.y];
    const unsigned per = (unsigned)T.n_src * (unsigned)T.k;
    const long total = (long)m.B * per;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const unsigned b = (unsigned)e / per;      // total < 2^31 (checked on the host)
        const int j = T.idx[e];
        if ((unsigned)j < (unsigned)T.n_dst) atomicAdd(&T.cursor[(long)b * T.n_dst + j], 1);
    }
}

// exclusive scan of one cloud's counts -> offsets[0..n_dst], cursor := segment start; one workgroup per (task, cloud)
__global__ __launch_bounds__(1024) void csr_scan_kernel(const CsrMulti m) {
    __shared__ int part[1024];
    const int task = blockIdx.x / m.B, b = blockIdx.x % m.B, t = threadIdx.x;
    const CsrTask& T = m.t[task];
    int* cu = T.cursor + (long)b * T.n_dst;
    int* off = T.offsets + (long)b * (T.n_dst + 1);
    const int chunk = (T.n_dst + 1023) / 1024;
    const int lo = min(T.n_dst, t * chunk), hi = min(T.n_dst, lo + chunk);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cu[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) {
        const int c = cu[i];
        off[i] = run;
        cu[i] = run;
        run += c;
    }
    if (t == 1023) off[T.n_dst] = part[1023];
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const CsrMulti m) {
    const CsrTask& T = m.t[blockIdx.y];
    const unsigned per = (unsigned)T.n_src * (unsigned)T.k;
    const long total = (long)m.B * per;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const unsigned b = (unsigned)e / per;
        const int j = T.idx[e];
        if ((unsigned)j >= (unsigned)T.n_dst) continue;
        const int pos = atomicAdd(&T.cursor[(long)b * T.n_dst + j], 1);
        T.tmp[(long)b * per + pos] = (int)((unsigned)e - b * per);
    }
}

// ---- two-level path -----------------------------------------------------------------------------------------
constexpr int CSR_TILE = 4096;       // entries per tile (256 threads x 16)
constexpr int CSR_BUCKET = 256;      // destinations per bucket
constexpr int CSR_MAX_BUCKETS = 1024;

__global__ __launch_bounds__(256) void csr2_zero_kernel(const CsrMulti m) {
    const CsrTask& T = m.t[blockIdx.y];
    const long total = (long)m.B * T.nbuckets;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) T.bsize[e] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) T.long_count[0] = 0;
}

// pass 1: bucket sizes.  grid (max tiles, ntasks * B)
__global__ __launch_bounds__(256) void csr2_hist_kernel(const CsrMulti m) {
    __shared__ int hist[CSR_MAX_BUCKETS];
    const int task = blockIdx.y / m.B;
    int b = blockIdx.y % m.B, bx = blockIdx.x;
    csr_deal(m, b, bx);
    const CsrTask& T = m.t[task];
    if (bx >= T.ntiles) return;
    const int per = T.n_src * T.k;
    for (int i = threadIdx.x; i < T.nbuckets; i += 256) hist[i] = 0;
    __syncthreads();
    const int32_t* idx = T.idx + (long)b * per;
    const int e0 = bx * CSR_TILE;
#pragma unroll 4
    for (int i = 0; i < CSR_TILE / 256; ++i) {
        const int e = e0 + i * 256 + threadIdx.x;
        if (e < per) {
            const int j = idx[e];
            if ((unsigned)j < (unsigned)T.n_dst) atomicAdd(&hist[j >> 8], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T.nbuckets; i += 256)
        if (hist[i] > 0) atomicAdd(&T.bsize[(long)b * T.nbuckets + i], hist[i]);
}

// exclusive scan of a cloud's bucket sizes; bsize becomes the reserve cursor (0).  grid (ntasks * B), 1024 threads
__global__ __launch_bounds__(1024) void csr2_scan_kernel(const CsrMulti m) {
    __shared__ int part[1024];
    const int task = blockIdx.x / m.B, b = blockIdx.x % m.B, t = threadIdx.x;
    const CsrTask& T = m.t[task];
    int* bs = T.bsize + (long)b * T.nbuckets;
    const int v = t < T.nbuckets ? bs[t] : 0;
    part[t] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int u = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += u;
        __syncthreads();
    }
    int* st = T.bstart + (long)b * (T.nbuckets + 1);
    if (t < T.nbuckets) {
        st[t] = part[t] - v;
        bs[t] = 0;
    }
    if (t == T.nbuckets - 1) st[T.nbuckets] = part[t];
}

// pass 2: entries move to their bucket, packed (destination inside the bucket << 24 | local source row)
__global__ __launch_bounds__(256) void csr2_scatter_kernel(const CsrMulti m) {
    __shared__ int hist[CSR_MAX_BUCKETS];
    const int task = blockIdx.y / m.B;
    int b = blockIdx.y % m.B, bx = blockIdx.x;
    csr_deal(m, b, bx);
    const CsrTask& T = m.t[task];
    if (bx >= T.ntiles) return;
    const int per = T.n_src * T.k;
    for (int i = threadIdx.x; i < T.nbuckets; i += 256) hist[i] = 0;
    __syncthreads();
    const int32_t* idx = T.idx + (long)b * per;
    const int e0 = bx * CSR_TILE;
    int jv[CSR_TILE / 256], rk[CSR_TILE / 256];
#pragma unroll
    for (int i = 0; i < CSR_TILE / 256; ++i) {
        const int e = e0 + i * 256 + threadIdx.x;
        jv[i] = -1;
        if (e < per) {
            const int j = idx[e];
            if ((unsigned)j < (unsigned)T.n_dst) {
                jv[i] = j;
                rk[i] = atomicAdd(&hist[j >> 8], 1);     // rank inside (tile, bucket): LDS
            }
        }
    }
    __syncthreads();
    // one reservation per non-empty (tile, bucket): hist becomes the tile's base inside the bucket
    for (int i = threadIdx.x; i < T.nbuckets; i += 256) {
        const int c = hist[i];
        hist[i] = c > 0 ? atomicAdd(&T.bsize[(long)b * T.nbuckets + i], c) : 0;
    }
    __syncthreads();
    const int* st = T.bstart + (long)b * (T.nbuckets + 1);
    int32_t* mid = T.tmp + (long)b * per;
#pragma unroll
    for (int i = 0; i < CSR_TILE / 256; ++i) {
        if (jv[i] >= 0) {
            const int h = jv[i] >> 8;
            mid[st[h] + hist[h] + rk[i]] = ((jv[i] & 255) << 24) | (e0 + i * 256 + threadIdx.x);
        }
    }
}

// pass 3: one workgroup per (cloud, bucket): offsets of its 256 destinations, then every entry goes to its
// destination's segment - short segments (<= CAP) are assembled in a per-destination LDS row, sorted by rank by the
// destination's lane and written out in order; longer ones are written in arrival order and queued for
// csr2_sort_long_kernel.
template <int CAP>
__global__ __launch_bounds__(256) void csr2_bucket_kernel(const CsrMulti m) {
    __shared__ int cnt[CSR_BUCKET], start[CSR_BUCKET], cur[CSR_BUCKET];
    __shared__ int seg[CSR_BUCKET][CAP + 1];
    const int task = blockIdx.y / m.B;
    int b = blockIdx.y % m.B, bx = blockIdx.x;
    csr_deal(m, b, bx);
    const CsrTask& T = m.t[task];
    if (T.cap != CAP || bx >= T.nbuckets) return;
    const int per = T.n_src * T.k, h = bx, t = threadIdx.x;
    const int* st = T.bstart + (long)b * (T.nbuckets + 1);
    const int s0 = st[h], mcount = st[h + 1] - s0;
    const int32_t* mid = T.tmp + (long)b * per + s0;
    int32_t* ent = T.entries + (long)b * per + s0;
    cnt[t] = 0;
    cur[t] = 0;
    __syncthreads();
    for (int e = t; e < mcount; e += 256) atomicAdd(&cnt[(unsigned)mid[e] >> 24], 1);
    __syncthreads();
    // exclusive scan of the 256 counters (Hillis-Steele in LDS)
    const int c = cnt[t];
    start[t] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int u = (t >= o) ? start[t - o] : 0;
        __syncthreads();
        start[t] += u;
        __syncthreads();
    }
    const int my_start = start[t] - c;
    __syncthreads();
    start[t] = my_start;
    const int j = h * CSR_BUCKET + t;
    int* off = T.offsets + (long)b * (T.n_dst + 1);
    if (j < T.n_dst) off[j] = s0 + my_start;
    if (j == T.n_dst - 1) off[T.n_dst] = st[T.nbuckets];
    __syncthreads();
    for (int e = t; e < mcount; e += 256) {
        const int p = mid[e];
        const int lo = (unsigned)p >> 24, r = p & 0xffffff;
        const int pos = atomicAdd(&cur[lo], 1);
        if (cnt[lo] <= CAP) seg[lo][pos] = r;
        else ent[start[lo] + pos] = r;          // long segment: arrival order now, sorted by the next kernel
    }
    __syncthreads();
    if (c == 0) return;
    if (c > CAP) {
        const int slot = atomicAdd(T.long_count, 1);     // which slot a segment gets does not matter: each is sorted alone
        if (slot < T.long_cap) T.long_list[slot] = b * T.n_dst + j;
        return;
    }
    const int* mine = seg[t];
    int* out = ent + my_start;
    for (int a = 0; a < c; ++a) {
        const int v = mine[a];
        int rank = 0;
        for (int q = 0; q < c; ++q) rank += (mine[q] < v) ? 1 : 0;
        out[rank] = v;
    }
}

// long segments of the two-level path: copied to the scratch array (its bucket data is no longer needed), then ranked back
// into place.  The copy is read back by OTHER lanes of the workgroup: agent-scope loads, not the (non-coherent) L1.
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void csr2_sort_long_kernel(const CsrMulti m) {
    __shared__ int chunk[1024];
    const CsrTask& T = m.t[blockIdx.y];
    const int nlong = min(T.long_count[0], T.long_cap);
    const unsigned per = (unsigned)T.n_src * (unsigned)T.k;
    for (int it = blockIdx.x; it < nlong; it += gridDim.x) {
        const unsigned d = (unsigned)T.long_list[it];
        const unsigned b = d / (unsigned)T.n_dst;
        const int j = (int)(d - b * (unsigned)T.n_dst);
        const int* off = T.offsets + (long)b * (T.n_dst + 1);
        const int s0 = off[j], len = off[j + 1] - s0;
        int* src = T.tmp + (long)b * per + s0;          // same position in the scratch array: segments do not overlap
        int* dst = T.entries + (long)b * per + s0;
        for (int a = threadIdx.x; a < len; a += 256) src[a] = dst[a];
        __threadfence();
        __syncthreads();
        for (int a0 = 0; a0 < len; a0 += 256) {
            const int a = a0 + threadIdx.x;
            const int v = a < len ? ld_agent(src + a) : 0;
            int rank = 0;
            for (int c0 = 0; c0 < len; c0 += 1024) {
                __syncthreads();
                for (int c = threadIdx.x; c < 1024; c += 256) chunk[c] = c0 + c < len ? ld_agent(src + c0 + c) : 0x7fffffff;
                __syncthreads();
                const int cn = min(1024, len - c0);
                for (int c = 0; c < cn; ++c) rank += (chunk[c] < v) ? 1 : 0;
            }
            if (a < len) dst[rank] = v;
        }
        __syncthreads();
    }
}

// One lane per destination: its segment goes to a lane-private LDS row (stride CAP + 1 words: the 64 lanes of a
// wavefront read distinct banks), every entry's rank = how many entries of the segment are smaller (they are distinct),
// entry -> entries[start + rank].  All LDS reads are independent, nothing waits on a chain.
template <int CAP, int THREADS>
__global__ __launch_bounds__(THREADS) void csr_sort_kernel(const CsrMulti m) {
    __shared__ int seg[THREADS][CAP + 1];
    const CsrTask& T = m.t[blockIdx.y];
    if (T.cap != CAP) return;
    const long total = (long)m.B * T.n_dst;
    const unsigned per = (unsigned)T.n_src * (unsigned)T.k;
    for (long d = (long)blockIdx.x * THREADS + threadIdx.x; d < total; d += (long)gridDim.x * THREADS) {
        const unsigned b = (unsigned)d / (unsigned)T.n_dst;
        const int j = (int)((unsigned)d - b * (unsigned)T.n_dst);
        const int* off = T.offsets + (long)b * (T.n_dst + 1);
        const int s0 = off[j], len = off[j + 1] - s0;
        if (len <= 0) continue;
        const int* src = T.tmp + (long)b * per + s0;
        int* dst = T.entries + (long)b * per + s0;
        if (len == 1) { dst[0] = src[0]; continue; }
        if (len > CAP) {
            const int slot = atomicAdd(T.long_count, 1);     // which slot a segment gets does not matter: each is sorted alone
            if (slot < T.long_cap) T.long_list[slot] = (int)d;
            continue;
        }
        int* mine = seg[threadIdx.x];
        for (int a = 0; a < len; ++a) mine[a] = src[a];
        for (int a = 0; a < len; ++a) {
            const int v = mine[a];
            int rank = 0;
            for (int c = 0; c < len; ++c) rank += (mine[c] < v) ? 1 : 0;
            dst[rank] = v;
        }
    }
}

// long segments (more than CAP gatherers of one point: duplicated coordinates): one workgroup each, rank by comparison
// against LDS-staged chunks of the segment
__global__ __launch_bounds__(256) void csr_sort_long_kernel(const CsrMulti m) {
    __shared__ int chunk[1024];
    const CsrTask& T = m.t[blockIdx.y];
    const int nlong = min(T.long_count[0], T.long_cap);
    const unsigned per = (unsigned)T.n_src * (unsigned)T.k;
    for (int it = blockIdx.x; it < nlong; it += gridDim.x) {
        const unsigned d = (unsigned)T.long_list[it];
        const unsigned b = d / (unsigned)T.n_dst;
        const int j = (int)(d - b * (unsigned)T.n_dst);
        const int* off = T.offsets + (long)b * (T.n_dst + 1);
        const int s0 = off[j], len = off[j + 1] - s0;
        const int* src = T.tmp + (long)b * per + s0;
        int* dst = T.entries + (long)b * per + s0;
        for (int a0 = 0; a0 < len; a0 += 256) {
            const int a = a0 + threadIdx.x;
            const int v = a < len ? src[a] : 0;
            int rank = 0;
            for (int c0 = 0; c0 < len; c0 += 1024) {
                __syncthreads();
                for (int c = threadIdx.x; c < 1024; c += 256) chunk[c] = c0 + c < len ? src[c0 + c] : 0x7fffffff;
                __syncthreads();
                const int cn = min(1024, len - c0);
                for (int c = 0; c < cn; ++c) rank += (chunk[c] < v) ? 1 : 0;
            }
            if (a < len) dst[rank] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------ segment sum
struct SegParams {
    const float* src;
    long lds, src_bstride;   // source row (b, r) at src + (b*src_bstride + r)*lds
    float* dst;
    long ldd, dst_bstride;
    const int32_t* offsets;
    const int32_t* entries;
    int n_dst, C;
    long per;                // entries per cloud
    long total;              // B * n_dst
    int accumulate;
    long xcd_chunk;          // > 0: the workgroups of XCD x (blockIdx.x % 8) sum the destinations [x * chunk, (x + 1) * chunk)
};

// tpr = C/4 lanes (a power of two <= 64... or 128/256 for wider rows) share one destination row, 16 bytes each
template <int TPR, bool SB = false>     // SB: the source rows are stored as bf16 (bf16-storage mode), sums and dst stay fp32
__global__ __launch_bounds__(256) void segment_sum_vec_kernel(const SegParams p) {
    constexpr int RPW = 256 / TPR;      // destination rows in flight per workgroup
    const int q = threadIdx.x % TPR, rsub = threadIdx.x / TPR;
    // XCD-local destination ranges (as the pooling forward's PointSpan) where a source row is narrower than a 128-byte line: the
    // four 32-byte rows of a line belong to ONE source point and go to four of its neighbours - with the points in cell order
    // (rl_band_sort) those are destinations a few positions apart, and on one XCD the line is fetched into that L2 once
    // (<2>: 77.1 -> 70.3 us per launch; before the cell order 97.3 -> 94.5; rows of a full line or more: 35.9 -> 37.5, not used).
    long d = (long)blockIdx.x * RPW + rsub, dstep = (long)gridDim.x * RPW, dend = p.total;
    if (p.xcd_chunk > 0) {
        const long lo = (long)(blockIdx.x & 7) * p.xcd_chunk;
        dend = lo + p.xcd_chunk < p.total ? lo + p.xcd_chunk : p.total;
        d = lo + (long)(blockIdx.x >> 3) * RPW + rsub;
        dstep = (long)(gridDim.x >> 3) * RPW;
    }
    for (; d < dend; d += dstep) {
        const unsigned b = (unsigned)d / (unsigned)p.n_dst;
        const int j = (int)((unsigned)d - b * (unsigned)p.n_dst);
        const int* off = p.offsets + (long)b * (p.n_dst + 1);
        const int s0 = off[j], s1 = off[j + 1];
        const int* ent = p.entries + (long)b * p.per;
        const long sb = (long)b * p.src_bstride * p.lds;       // element offset of the cloud's first source row
        for (int c = q * 4; c < p.C; c += TPR * 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int e = s0;
            // Eight rows per trip, their indices requested one trip ahead: a segment of 16 entries is 1 + 2 dependent
            // round trips instead of 8 (index -> row, four times over).  The adds keep the entry order.
            if (e + 8 <= s1) {
                int r[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) r[t] = ent[e + t];
                for (; e + 8 <= s1; e += 8) {
                    float4 v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = rl_ldx4<SB>(p.src, sb + (long)r[t] * p.lds + c);
                    if (e + 16 <= s1) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) r[t] = ent[e + 8 + t];
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) { acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w; }
                }
            }
            for (; e + 4 <= s1; e += 4) {      // four rows requested before the first is added (order of the adds is fixed)
                const int r0 = ent[e], r1 = ent[e + 1], r2 = ent[e + 2], r3 = ent[e + 3];
                const float4 v0 = rl_ldx4<SB>(p.src, sb + (long)r0 * p.lds + c);
                const float4 v1 = rl_ldx4<SB>(p.src, sb + (long)r1 * p.lds + c);
                const float4 v2 = rl_ldx4<SB>(p.src, sb + (long)r2 * p.lds + c);
                const float4 v3 = rl_ldx4<SB>(p.src, sb + (long)r3 * p.lds + c);
                acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
                acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
                acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
                acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
            }
            for (; e < s1; ++e) {
                const float4 v = rl_ldx4<SB>(p.src, sb + (long)ent[e] * p.lds + c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            float4* o = reinterpret_cast<float4*>(p.dst + ((long)b * p.dst_bstride + j) * p.ldd + c);
            if (p.accumulate) {
                const float4 old = *o;
                acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
            }
            *o = acc;
        }
    }
}

__global__ __launch_bounds__(256) void segment_sum_kernel(const SegParams p) {
    const long elems = p.total * p.C;
    for (long x = (long)blockIdx.x * 256 + threadIdx.x; x < elems; x += (long)gridDim.x * 256) {
        const long d = x / p.C;
        const int c = (int)(x - d * p.C);
        const long b = d / p.n_dst;
        const int j = (int)(d - b * p.n_dst);
        const int* off = p.offsets + b * (p.n_dst + 1);
        const int* ent = p.entries + b * p.per;
        const float* sb = p.src + b * p.src_bstride * p.lds;
        float acc = 0.f;
        for (int e = off[j]; e < off[j + 1]; ++e) acc += sb[(long)ent[e] * p.lds + c];
        float* o = p.dst + (b * p.dst_bstride + j) * p.ldd + c;
        *o = p.accumulate ? *o + acc : acc;
    }
}

struct Plan {
    size_t tmp, cursor, long_list, long_count, bsize, bstart, bytes;
    int long_cap, cap, nbuckets, ntiles;
    bool two_level;
};
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
inline bool two_level_ok(const rl_csr_task& t) {
    return (t.n_dst + CSR_BUCKET - 1) / CSR_BUCKET <= CSR_MAX_BUCKETS && (int64_t)t.n_src * t.k < (1 << 24);
}
Plan make_plan(int B, const rl_csr_task& t, bool two_level) {
    Plan p;
    const size_t ent = (size_t)B * t.n_src * t.k;
    // lane-private sort capacity: at least twice the mean in-degree where the LDS allows
    const double mean = (double)t.n_src * t.k / (double)(t.n_dst > 0 ? t.n_dst : 1);
    p.two_level = two_level;
    if (two_level) p.cap = mean * 2 <= 32 ? 32 : 56;
    else p.cap = mean * 2 <= 32 ? 32 : (mean * 2 <= 64 ? 64 : 128);
    p.long_cap = (int)(ent / (size_t)p.cap) + 1;
    p.nbuckets = (t.n_dst + CSR_BUCKET - 1) / CSR_BUCKET;
    p.ntiles = (int)(((size_t)t.n_src * t.k + CSR_TILE - 1) / CSR_TILE);
    size_t o = 0;
    p.tmp = o;        o += up256(ent * 4);
    p.cursor = o;     o += two_level ? 0 : up256((size_t)B * t.n_dst * 4);
    p.long_list = o;  o += up256((size_t)p.long_cap * 4);
    p.long_count = o; o += 256;
    p.bsize = o;      o += up256((size_t)B * p.nbuckets * 4);
    p.bstart = o;     o += up256((size_t)B * (p.nbuckets + 1) * 4);
    p.bytes = o;
    return p;
}
inline bool all_two_level(const rl_csr_task* tasks, int ntasks) {
    for (int i = 0; i < ntasks; ++i)
        if (!two_level_ok(tasks[i])) return false;
    return true;
}

}  // namespace

extern "C" int64_t rl_csr_workspace_bytes(const rl_csr_task* tasks, int ntasks, int B) {
    if (!tasks || ntasks <= 0 || B <= 0) return 0;
    int64_t total = 0;
    for (int i = 0; i < ntasks; ++i) {
        if (tasks[i].n_src <= 0 || tasks[i].k <= 0 || tasks[i].n_dst <= 0) return 0;
        total += (int64_t)make_plan(B, tasks[i], all_two_level(tasks, ntasks)).bytes;
    }
    return total;
}

extern "C" int rl_csr_build(const rl_csr_task* tasks, int ntasks, int B, void* workspace, int64_t workspace_bytes,
                            void* stream) {
    RL_REQUIRE(tasks && ntasks > 0 && ntasks <= CSR_MAX_TASKS && B > 0 && (long)B * ntasks <= 65535, RL_ERR_ARGS,
               "rl_csr_build: 1..%d tasks expected", CSR_MAX_TASKS);
    RL_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, RL_ERR_ARGS, "rl_csr_build: workspace must be 256-byte aligned");
    CsrMulti m;
    m.ntasks = ntasks;
    m.B = B;
    static const bool no_xcd = getenv("RL_NO_XCD_POINTS") != nullptr;       // A/B switch shared with the pooling / KNN kernels
    m.xcd = (!no_xcd && B > 1 && 8 % B == 0) ? 1 : 0;
    char* ws = (char*)workspace;
    int64_t used = 0;
    long max_ent = 1, max_dst = 1;
    int max_tiles = 1, max_buckets = 1;
    bool need[3] = {false, false, false};
    for (int i = 0; i < ntasks; ++i)
        RL_REQUIRE(tasks[i].idx && tasks[i].offsets && tasks[i].entries && tasks[i].n_src > 0 && tasks[i].k > 0 && tasks[i].n_dst > 0,
                   RL_ERR_ARGS, "rl_csr_build: bad task %d", i);
    const bool two = all_two_level(tasks, ntasks);
    for (int i = 0; i < ntasks; ++i) {
        const rl_csr_task& t = tasks[i];
        RL_REQUIRE((int64_t)B * t.n_src * t.k < (1l << 31) && (int64_t)B * t.n_dst < (1l << 31), RL_ERR_ARGS,
                   "rl_csr_build: task %d is too large", i);
        const Plan p = make_plan(B, t, two);
        RL_REQUIRE(used + (int64_t)p.bytes <= workspace_bytes, RL_ERR_ARGS, "rl_csr_build: workspace too small");
        CsrTask& T = m.t[i];
        T.idx = t.idx; T.n_src = t.n_src; T.k = t.k; T.n_dst = t.n_dst; T.offsets = t.offsets; T.entries = t.entries;
        T.tmp = (int32_t*)(ws + used + p.tmp);
        T.cursor = (int32_t*)(ws + used + p.cursor);
        T.long_list = (int32_t*)(ws + used + p.long_list);
        T.long_count = (int32_t*)(ws + used + p.long_count);
        T.bsize = (int32_t*)(ws + used + p.bsize);
        T.bstart = (int32_t*)(ws + used + p.bstart);
        T.long_cap = p.long_cap; T.cap = p.cap; T.nbuckets = p.nbuckets; T.ntiles = p.ntiles;
        need[p.cap == 32 ? 0 : (p.cap == 64 || p.cap == 56) ? 1 : 2] = true;
        used += (int64_t)p.bytes;
        max_ent = max_ent > (long)B * t.n_src * t.k ? max_ent : (long)B * t.n_src * t.k;
        max_dst = max_dst > (long)B * t.n_dst ? max_dst : (long)B * t.n_dst;
        max_tiles = max_tiles > p.ntiles ? max_tiles : p.ntiles;
        max_buckets = max_buckets > p.nbuckets ? max_buckets : p.nbuckets;
    }
    hipStream_t st = (hipStream_t)stream;
    auto gridx = [](long work, int per) { long g = (work + per - 1) / per; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); };
    if (two) {
        hipLaunchKernelGGL(csr2_zero_kernel, dim3(gridx((long)B * max_buckets, 256), ntasks), dim3(256), 0, st, m);
        hipLaunchKernelGGL(csr2_hist_kernel, dim3(max_tiles, ntasks * B), dim3(256), 0, st, m);
        hipLaunchKernelGGL(csr2_scan_kernel, dim3(ntasks * B), dim3(1024), 0, st, m);
        hipLaunchKernelGGL(csr2_scatter_kernel, dim3(max_tiles, ntasks * B), dim3(256), 0, st, m);
        if (need[0]) hipLaunchKernelGGL(csr2_bucket_kernel<32>, dim3(max_buckets, ntasks * B), dim3(256), 0, st, m);
        if (need[1]) hipLaunchKernelGGL(csr2_bucket_kernel<56>, dim3(max_buckets, ntasks * B), dim3(256), 0, st, m);
        hipLaunchKernelGGL(csr2_sort_long_kernel, dim3(128, ntasks), dim3(256), 0, st, m);
        rl_note_kernel("csr2_bucket_kernel");
        RL_LAUNCH_CHECK("rl_csr_build");
        return RL_OK;
    }
    // graphs beyond the packing limits (> 262144 destinations or >= 2^24 entries per cloud): per-entry global atomics
    hipLaunchKernelGGL(csr_zero_kernel, dim3(gridx(max_dst, 256), ntasks), dim3(256), 0, st, m);
    hipLaunchKernelGGL(csr_count_kernel, dim3(gridx(max_ent, 256), ntasks), dim3(256), 0, st, m);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(ntasks * B), dim3(1024), 0, st, m);
    hipLaunchKernelGGL(csr_fill_kernel, dim3(gridx(max_ent, 256), ntasks), dim3(256), 0, st, m);
    if (need[0]) hipLaunchKernelGGL((csr_sort_kernel<32, 256>), dim3(gridx(max_dst, 256), ntasks), dim3(256), 0, st, m);
    if (need[1]) hipLaunchKernelGGL((csr_sort_kernel<64, 128>), dim3(gridx(max_dst, 128), ntasks), dim3(128), 0, st, m);
    if (need[2]) hipLaunchKernelGGL((csr_sort_kernel<128, 64>), dim3(gridx(max_dst, 64), ntasks), dim3(64), 0, st, m);
    hipLaunchKernelGGL(csr_sort_long_kernel, dim3(128, ntasks), dim3(256), 0, st, m);
    rl_note_kernel("csr_sort_kernel");
    RL_LAUNCH_CHECK("rl_csr_build");
    return RL_OK;
}

extern "C" int rl_segment_sum_rows(const rl_segsum_desc* d, void* stream) {
    RL_REQUIRE(d && d->src && d->dst && d->offsets && d->entries && d->B > 0 && d->n_dst > 0 && d->C > 0 && d->entries_per_cloud >= 0,
               RL_ERR_ARGS, "rl_segment_sum_rows: bad descriptor");
    RL_REQUIRE(d->lds >= d->C && d->ldd >= d->C && d->dst_bstride >= d->n_dst && d->src_bstride > 0, RL_ERR_ARGS,
               "rl_segment_sum_rows: bad strides");
    RL_REQUIRE((int64_t)d->B * d->n_dst < (1l << 31), RL_ERR_ARGS, "rl_segment_sum_rows: too many rows");
    SegParams p;
    p.src = d->src; p.lds = d->lds; p.src_bstride = d->src_bstride; p.dst = d->dst; p.ldd = d->ldd;
    p.dst_bstride = d->dst_bstride; p.offsets = d->offsets; p.entries = d->entries; p.n_dst = d->n_dst; p.C = d->C;
    p.per = d->entries_per_cloud; p.total = (long)d->B * d->n_dst; p.accumulate = d->accumulate;
    p.xcd_chunk = 0;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (d->C % 4 == 0) && (d->lds % 4 == 0) && (d->ldd % 4 == 0) && ((uintptr_t)d->dst & 15) == 0 &&
                     ((uintptr_t)d->src & (d->src_bf16 ? 7 : 15)) == 0;
    RL_REQUIRE(!d->src_bf16 || vec, RL_ERR_UNSUPPORTED, "rl_segment_sum_rows: bf16 source rows need C, lds, ldd multiples of 4 and aligned tensors");
    if (vec) {
        const int c4 = d->C / 4;
        int tpr = 1;
        while (tpr < c4 && tpr < 256) tpr <<= 1;      // power of two >= C/4 (lanes beyond the row idle), at most 256
        const long rpw = 256 / tpr;
        long g = (p.total + rpw - 1) / rpw;
        if (g > 8192) g = 8192;
        if (g < 1) g = 1;
        // XCD-local destination ranges for rows narrower than a line (C <= 16): see segment_sum_vec_kernel
        static const bool no_xcd = getenv("RL_NO_XCD_POINTS") != nullptr;       // A/B switch shared with the pooling / KNN kernels
        p.xcd_chunk = (!no_xcd && d->C <= 16 && g >= 8 && g % 8 == 0) ? (p.total + 7) / 8 : 0;
#define SEG_CASE(T)                                                                                          \
    case T:                                                                                                  \
        if (d->src_bf16) hipLaunchKernelGGL((segment_sum_vec_kernel<T, true>), dim3(g), dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((segment_sum_vec_kernel<T, false>), dim3(g), dim3(256), 0, st, p);            \
        break;
        switch (tpr) {
            SEG_CASE(1) SEG_CASE(2) SEG_CASE(4) SEG_CASE(8) SEG_CASE(16) SEG_CASE(32) SEG_CASE(64) SEG_CASE(128)
            default:
                if (d->src_bf16) hipLaunchKernelGGL((segment_sum_vec_kernel<256, true>), dim3(g), dim3(256), 0, st, p);
                else hipLaunchKernelGGL((segment_sum_vec_kernel<256, false>), dim3(g), dim3(256), 0, st, p);
                break;
        }
#undef SEG_CASE
        rl_note_kernel("segment_sum_vec_kernel");
    } else {
        long g = (p.total * p.C + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(segment_sum_kernel, dim3(g < 1 ? 1 : g), dim3(256), 0, st, p);
        rl_note_kernel("segment_sum_kernel");
    }
    RL_LAUNCH_CHECK("rl_segment_sum_rows");
    return RL_OK;
}
