/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (3d_recognizer_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker / CPU baseline.
 *
 * CPU restatement of the reference's exact K-nearest-neighbour operator
 *   knn_tpk.knn(support, querry, k)      /root/reference/randlanet/utils/src/knn.cpp:43-61
 *   _single_batch_knn                    knn.cpp:11-41   (Ns >= k check :15-17, -1 fill :18-19)
 *   nanoflann_knn_neighbors<T>           neighbors.h:281-322
 *   L2_Simple_Adaptor::evalMetric        nanoflann.hpp:488-497  -> d2 = ((dx*dx)+(dy*dy))+(dz*dz), dx = q - s
 *   KNNResultSet::addPoint               nanoflann.hpp:193-226  -> ascending d2
 *
 * The answer is defined, not the traversal: the reference walks a kd-tree (leaf 15,
 * neighbors.h:295) and breaks exact-distance ties by traversal order
 * (nanoflann.hpp:205, strict '>', first seen wins), which no other search can
 * reproduce.  This restatement (and the HIP kernel it checks) orders by
 * (d2, lowest index).  Pinned against the compiled reference (oracle/_ref/knn_tpk.so)
 * by tests/test_oracle_knn.py and the committed fixtures tests/golden/knn_*.npz:
 * d2 bit-equal in every slot; idx equal wherever d2 is unique in its row and differs
 * from the (k+1)-th d2.
 *
 * Two searches with identical results:
 *   knn_oracle_brute : O(Nq*Ns) scan, the definition.
 *   knn_oracle_grid  : exact uniform-grid search (ring expansion with a conservative
 *                      stopping bound); used where the brute scan would take minutes,
 *                      and as the single-threaded CPU-baseline KNN (the reference's
 *                      kd-tree search is single-threaded too, knn.cpp:52-56).
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared  (no -ffast-math: d2 must be plain IEEE fp32).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KNN_ERR_ARGS (-1)
#define KNN_ERR_FEW_SUPPORT (-2) /* knn.cpp:15-17 "Not enough points in support" */
#define KNN_ERR_NOMEM (-3)

static inline float d2_ref(const float *q, const float *s) {
    /* nanoflann.hpp:488-497: result = 0; result += diff*diff for x, y, z in order. */
    float dx = q[0] - s[0], dy = q[1] - s[1], dz = q[2] - s[2];
    float r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

/* sorted insertion by (d2, idx); list holds *cnt <= k entries */
static inline void push(float *bd, int64_t *bi, int *cnt, int k, float d, int64_t j) {
    int n = *cnt;
    if (n == k) {
        if (d > bd[k - 1] || (d == bd[k - 1] && j > bi[k - 1])) return;
        n = k - 1;
    }
    int p = n;
    while (p > 0 && (bd[p - 1] > d || (bd[p - 1] == d && bi[p - 1] > j))) {
        bd[p] = bd[p - 1];
        bi[p] = bi[p - 1];
        --p;
    }
    bd[p] = d;
    bi[p] = j;
    if (*cnt < k) (*cnt)++;
}

int knn_oracle_brute(const float *support, const float *query, int B, int Ns, int Nq, int k,
                     int64_t *idx_out, float *d2_out) {
    if (B < 0 || Ns < 0 || Nq < 0 || k <= 0) return KNN_ERR_ARGS;
    if (Ns < k) return KNN_ERR_FEW_SUPPORT;
    for (int b = 0; b < B; ++b) {
        const float *S = support + (size_t)b * Ns * 3;
        const float *Q = query + (size_t)b * Nq * 3;
        for (int i = 0; i < Nq; ++i) {
            float *bd = d2_out + ((size_t)b * Nq + i) * k;
            int64_t *bi = idx_out + ((size_t)b * Nq + i) * k;
            int cnt = 0;
            for (int j = 0; j < Ns; ++j) push(bd, bi, &cnt, k, d2_ref(Q + 3 * i, S + 3 * j), j);
        }
    }
    return 0;
}

typedef struct {
    int n[3];
    double lo[3], h[3], inv[3];
    int *start; /* ncell + 1 */
    int *order; /* Ns: support indices sorted by cell */
} grid_t;

static int cell_of(const grid_t *g, const float *p, int c[3]) {
    for (int a = 0; a < 3; ++a) {
        double t = ((double)p[a] - g->lo[a]) * g->inv[a];
        int ci = (int)floor(t);
        if (ci < 0) ci = 0;
        if (ci >= g->n[a]) ci = g->n[a] - 1;
        c[a] = ci;
    }
    return (c[2] * g->n[1] + c[1]) * g->n[0] + c[0];
}

static int grid_build(grid_t *g, const float *S, int Ns, double per_cell) {
    double hi[3];
    for (int a = 0; a < 3; ++a) { g->lo[a] = S[a]; hi[a] = S[a]; }
    for (int j = 1; j < Ns; ++j)
        for (int a = 0; a < 3; ++a) {
            double v = S[3 * j + a];
            if (v < g->lo[a]) g->lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    double ext[3], vol = 1.0;
    int flat = 0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = hi[a] - g->lo[a];
        if (ext[a] <= 0) { ext[a] = 0; flat++; } else vol *= ext[a];
    }
    double ncell = (double)Ns / per_cell;
    if (ncell < 1) ncell = 1;
    double side = (flat == 3) ? 1.0 : pow(vol / ncell, 1.0 / (3 - flat));
    long total = 1;
    for (int a = 0; a < 3; ++a) {
        int n = (ext[a] > 0 && side > 0) ? (int)ceil(ext[a] / side) : 1;
        if (n < 1) n = 1;
        if (n > 256) n = 256;
        g->n[a] = n;
        g->h[a] = (ext[a] > 0) ? ext[a] / n : 1.0;
        g->inv[a] = 1.0 / g->h[a];
        total *= n;
    }
    g->start = (int *)calloc((size_t)total + 1, sizeof(int));
    g->order = (int *)malloc((size_t)(Ns > 0 ? Ns : 1) * sizeof(int));
    int *cid = (int *)malloc((size_t)(Ns > 0 ? Ns : 1) * sizeof(int));
    if (!g->start || !g->order || !cid) { free(cid); return KNN_ERR_NOMEM; }
    int c[3];
    for (int j = 0; j < Ns; ++j) { cid[j] = cell_of(g, S + 3 * j, c); g->start[cid[j] + 1]++; }
    for (long t = 0; t < total; ++t) g->start[t + 1] += g->start[t];
    int *fill = (int *)malloc((size_t)total * sizeof(int));
    if (!fill) { free(cid); return KNN_ERR_NOMEM; }
    memcpy(fill, g->start, (size_t)total * sizeof(int));
    for (int j = 0; j < Ns; ++j) g->order[fill[cid[j]]++] = j; /* ascending j inside a cell */
    free(fill);
    free(cid);
    return 0;
}

int knn_oracle_grid(const float *support, const float *query, int B, int Ns, int Nq, int k,
                    int64_t *idx_out, float *d2_out) {
    if (B < 0 || Ns < 0 || Nq < 0 || k <= 0) return KNN_ERR_ARGS;
    if (Ns < k) return KNN_ERR_FEW_SUPPORT;
    for (int b = 0; b < B; ++b) {
        const float *S = support + (size_t)b * Ns * 3;
        const float *Q = query + (size_t)b * Nq * 3;
        grid_t g;
        int rc = grid_build(&g, S, Ns, fmax(2.0, 0.5 * k));
        if (rc) return rc;
        double extent = 0;
        for (int a = 0; a < 3; ++a) extent = fmax(extent, g.h[a] * g.n[a]);
        const double slack = 1e-5 * extent + 1e-30;
        for (int i = 0; i < Nq; ++i) {
            const float *q = Q + 3 * i;
            float *bd = d2_out + ((size_t)b * Nq + i) * k;
            int64_t *bi = idx_out + ((size_t)b * Nq + i) * k;
            int cnt = 0, c[3];
            cell_of(&g, q, c);
            for (int r = 0;; ++r) {
                int lo[3], hi[3], covers = 1;
                for (int a = 0; a < 3; ++a) {
                    lo[a] = c[a] - r; hi[a] = c[a] + r;
                    if (lo[a] > 0 || hi[a] < g.n[a] - 1) covers = 0;
                    if (lo[a] < 0) lo[a] = 0;
                    if (hi[a] > g.n[a] - 1) hi[a] = g.n[a] - 1;
                }
                for (int z = lo[2]; z <= hi[2]; ++z)
                    for (int y = lo[1]; y <= hi[1]; ++y)
                        for (int x = lo[0]; x <= hi[0]; ++x) {
                            int dzr = abs(z - c[2]), dyr = abs(y - c[1]), dxr = abs(x - c[0]);
                            int ring = dzr > dyr ? dzr : dyr;
                            if (dxr > ring) ring = dxr;
                            if (ring != r) continue; /* inner cells were visited in earlier rings */
                            int cell = (z * g.n[1] + y) * g.n[0] + x;
                            for (int t = g.start[cell]; t < g.start[cell + 1]; ++t) {
                                int j = g.order[t];
                                push(bd, bi, &cnt, k, d2_ref(q, S + 3 * j), j);
                            }
                        }
                if (covers) break;
                if (cnt == k) {
                    /* every unvisited point lies beyond a face of the (2r+1)^3 block that is
                       still inside the grid; stop only when the k-th distance is strictly
                       inside the nearest such face (ties beyond it could carry a lower index) */
                    double bound = INFINITY;
                    for (int a = 0; a < 3; ++a) {
                        if (c[a] - r > 0) bound = fmin(bound, (double)q[a] - (g.lo[a] + (c[a] - r) * g.h[a]));
                        if (c[a] + r < g.n[a] - 1) bound = fmin(bound, (g.lo[a] + (c[a] + r + 1) * g.h[a]) - (double)q[a]);
                    }
                    bound -= slack;
                    if (bound > 0 && (double)bd[k - 1] < bound * bound) break;
                }
            }
        }
        free(g.start);
        free(g.order);
    }
    return 0;
}
