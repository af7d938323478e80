// Host twin of the exact K-nearest-neighbour search: rl_knn_f32_cpu has rl_knn_f32's contract but runs
// on the calling host with plain host pointers.  It backs the package's CPU device (reference randlanet/model.py:38-40
// picks the CPU when no GPU is present; config P = predict.py on a GPU-less box) and replaces, like the device kernel,
// the reference's compiled knn_tpk.knn (randlanet/utils/src/bindings.cpp:5-7, knn.cpp:11-61) - same distance expression
// ((dx*dx)+(dy*dy))+(dz*dz) in IEEE fp32 without FMA (nanoflann.hpp:488-497), rows ascending by (d2, index).
//
// Algorithm: per cloud a uniform grid (~8 supports per cell, counting sort), per query a walk over cubic shells of
// cells around the query's cell; the search stops once the K-th best squared distance is not larger than the squared
// distance from the query to the nearest face of the next shell (every unvisited support is at least that far).
// Queries are split over hardware threads.  No GPU, no HIP call.
#include "rl_common.h"

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace {

struct HostGrid {
    float lo[3], inv_cell, cell;
    int dim[3];
    std::vector<int> start;      // ncell + 1
    std::vector<int> order;      // support indices sorted by cell (ascending index inside a cell)
};

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void build_grid(const float* S, int Ns, HostGrid* g) {
    float lo[3] = {S[0], S[1], S[2]}, hi[3] = {S[0], S[1], S[2]};
    for (int i = 1; i < Ns; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], S[3 * i + a]);
            hi[a] = std::max(hi[a], S[3 * i + a]);
        }
    double ext[3], vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = std::max((double)hi[a] - (double)lo[a], 1e-12);
        vol *= ext[a];
    }
    double cell = std::cbrt(vol * 8.0 / (double)Ns);
    const double longest = std::max(ext[0], std::max(ext[1], ext[2]));
    cell = std::max(cell, longest / 256.0);
    if (!(cell > 0.0) || !std::isfinite(cell)) cell = 1.0;
    g->cell = (float)cell;
    g->inv_cell = (float)(1.0 / cell);
    long ncell = 1;
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = lo[a];
        g->dim[a] = std::max(1, std::min(256, (int)(ext[a] / cell) + 1));
        ncell *= g->dim[a];
    }
    g->start.assign(ncell + 1, 0);
    std::vector<int> cell_of(Ns);
    for (int i = 0; i < Ns; ++i) {
        int c[3];
        for (int a = 0; a < 3; ++a) c[a] = clampi((int)((S[3 * i + a] - g->lo[a]) * g->inv_cell), 0, g->dim[a] - 1);
        cell_of[i] = (c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0];
        g->start[cell_of[i] + 1]++;
    }
    for (long c = 0; c < ncell; ++c) g->start[c + 1] += g->start[c];
    g->order.resize(Ns);
    std::vector<int> cur(g->start.begin(), g->start.end() - 1);
    for (int i = 0; i < Ns; ++i) g->order[cur[cell_of[i]]++] = i;
}

struct Cand {
    float d2;
    int idx;
};
inline bool before(const Cand& a, const Cand& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx); }

void search_range(const float* S, const HostGrid& g, const float* Q, int q0, int q1, int k, int64_t* idx_out, float* d2_out) {
    std::vector<Cand> best(k);
    for (int q = q0; q < q1; ++q) {
        const float qx = Q[3 * q], qy = Q[3 * q + 1], qz = Q[3 * q + 2];
        const float qp[3] = {qx, qy, qz};
        int c0[3];
        for (int a = 0; a < 3; ++a) c0[a] = clampi((int)((qp[a] - g.lo[a]) * g.inv_cell), 0, g.dim[a] - 1);
        int have = 0;
        const int rmax = std::max(g.dim[0], std::max(g.dim[1], g.dim[2]));
        for (int r = 0; r <= rmax; ++r) {
            if (have == k && r > 0) {
                // distance from the query to the nearest face of shell r (cells at Chebyshev distance r from c0)
                float reach = INFINITY;
                for (int a = 0; a < 3; ++a) {
                    const float lo_face = qp[a] - (g.lo[a] + (float)(c0[a] - r + 1) * g.cell);   // to the low side
                    const float hi_face = (g.lo[a] + (float)(c0[a] + r) * g.cell) - qp[a];       // to the high side
                    if (c0[a] - r >= 0) reach = std::min(reach, lo_face);
                    if (c0[a] + r < g.dim[a]) reach = std::min(reach, hi_face);
                }
                if (reach == INFINITY) break;                     // the grid is exhausted
                // cells are assigned with float arithmetic: keep a margin of 1e-4 cell widths on the bound
                const float safe = reach - 1e-4f * g.cell;
                if (safe > 0.f && safe * safe > best[k - 1].d2) break;
            }
            for (int cz = c0[2] - r; cz <= c0[2] + r; ++cz) {
                if (cz < 0 || cz >= g.dim[2]) continue;
                for (int cy = c0[1] - r; cy <= c0[1] + r; ++cy) {
                    if (cy < 0 || cy >= g.dim[1]) continue;
                    const bool face = (cz == c0[2] - r || cz == c0[2] + r || cy == c0[1] - r || cy == c0[1] + r);
                    const int step = face ? 1 : 2 * r;            // inner rows of the shell: only its two x-ends
                    for (int cx = c0[0] - r; cx <= c0[0] + r; cx += (step > 0 ? step : 1)) {
                        if (cx < 0 || cx >= g.dim[0]) continue;
                        const long cell = ((long)cz * g.dim[1] + cy) * g.dim[0] + cx;
                        for (int e = g.start[cell]; e < g.start[cell + 1]; ++e) {
                            const int j = g.order[e];
                            const float dx = qx - S[3 * j], dy = qy - S[3 * j + 1], dz = qz - S[3 * j + 2];
                            Cand c;
                            c.d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
                            c.idx = j;
                            if (have < k) {
                                int p = have++;
                                while (p > 0 && before(c, best[p - 1])) { best[p] = best[p - 1]; --p; }
                                best[p] = c;
                            } else if (before(c, best[k - 1])) {
                                int p = k - 1;
                                while (p > 0 && before(c, best[p - 1])) { best[p] = best[p - 1]; --p; }
                                best[p] = c;
                            }
                        }
                    }
                }
            }
        }
        for (int s = 0; s < k; ++s) {
            idx_out[(long)q * k + s] = s < have ? best[s].idx : -1;
            d2_out[(long)q * k + s] = s < have ? best[s].d2 : 0.f;
        }
    }
}

}  // namespace

extern "C" int rl_knn_f32_cpu(const float* support, const float* query, int B, int Ns, int Nq, int k, int64_t* idx_out,
                              float* d2_out) {
    RL_REQUIRE(B > 0 && Ns >= 0 && Nq >= 0 && k > 0, RL_ERR_ARGS, "rl_knn_f32_cpu: bad sizes");
    RL_REQUIRE(Ns >= k, RL_ERR_FEW_SUPPORT, "Not enough points in support to find %d neighboors", k);   // knn.cpp:15-17
    RL_REQUIRE(support && query && idx_out && d2_out, RL_ERR_ARGS, "rl_knn_f32_cpu: null pointer");
    if (Nq == 0) return RL_OK;
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::min<unsigned>(hw ? hw : 1, 16);
    if ((long)Nq * Ns < (1l << 22)) nthreads = 1;
    for (int b = 0; b < B; ++b) {
        const float* S = support + (long)b * Ns * 3;
        const float* Q = query + (long)b * Nq * 3;
        HostGrid g;
        build_grid(S, Ns, &g);
        int64_t* io = idx_out + (long)b * Nq * k;
        float* dob = d2_out + (long)b * Nq * k;
        if (nthreads == 1) {
            search_range(S, g, Q, 0, Nq, k, io, dob);
        } else {
            std::vector<std::thread> pool;
            const int chunk = (Nq + nthreads - 1) / nthreads;
            for (int t = 0; t < nthreads; ++t) {
                const int q0 = t * chunk, q1 = std::min(Nq, q0 + chunk);
                if (q0 < q1) pool.emplace_back(search_range, S, std::cref(g), Q, q0, q1, k, io, dob);
            }
            for (auto& th : pool) th.join();
        }
    }
    return RL_OK;
}
