/*
 * TEST INFRASTRUCTURE -- CPU oracle (see oracle.h).
 *
 * neigh.c: neighbour sets.  The reference finds candidates with an R*-tree (rstar 0.9.3,
 * Cargo.lock:1963 -- not vendored) and then re-pins the SET itself:
 *   N(i) = { j : |x_i - x_j|^2 < s_ij^2 },  s_ij = ((h_i + h_j) * 0.5) * k,   self included
 * (strict filter neighborhood_search.rs:138-147, symmetrisation :157-185, brute-force definition
 * :214-237 and simulation.rs:1810-1863).  Candidate enumeration here uses the reference's OWN
 * uniform-grid scheme (neighborhood_search.rs:243-321, CellGrid :355-410), one grid per size class
 * (orc_build_neighbors), which is a superset generator for that predicate; orc_check_neighborhood is the
 * O(N^2) definition it is tested against.  List ORDER is unpinned in the reference (R*-tree traversal
 * order); the oracle uses ascending j.
 */
#include "oracle.h"
#include "sphmath.h"

#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float cs;
    int minx, miny, sx, sy;
} grid_t;

/* neighborhood_search.rs:253-255, 261-275 */
static grid_t make_grid(const float* pos, uint64_t n, float cs)
{
    grid_t g;
    float mnx = pos[0], mny = pos[1], mxx = pos[0], mxy = pos[1];
    for (uint64_t i = 0; i < n; i++) {
        mnx = fminf(mnx, pos[2 * i]);
        mxx = fmaxf(mxx, pos[2 * i]);
        mny = fminf(mny, pos[2 * i + 1]);
        mxy = fmaxf(mxy, pos[2 * i + 1]);
    }
    g.cs = cs;
    g.minx = (int)floorf(mnx / cs) - 1;
    g.miny = (int)floorf(mny / cs) - 1;
    int maxx = (int)floorf(mxx / cs) + 2;
    int maxy = (int)floorf(mxy / cs) + 2;
    g.sx = maxx - g.minx;
    g.sy = maxy - g.miny;
    return g;
}

static inline void cell_of(const grid_t* g, float x, float y, int* cx, int* cy)
{
    *cx = (int)floorf(x / g->cs) - g->minx;
    *cy = (int)floorf(y / g->cs) - g->miny;
}

static float max_h(const oracle_ctx* c)
{
    float hm = 0.f;
    for (uint64_t i = 0; i < c->n; i++) hm = fmaxf(hm, c->h2[i]);
    return hm;
}

/* linear CellGrid index, x fastest (neighborhood_search.rs:383-395) with cell = 2*h_max */
void orc_cell_indices(oracle_ctx* c)
{
    if (c->n == 0) return;
    float cs = max_h(c) * 2.f;
    grid_t g = make_grid(c->pos, c->n, cs);
    c->grid.cell_size = cs;
    c->grid.cells_min_x = g.minx;
    c->grid.cells_min_y = g.miny;
    c->grid.size_x = g.sx;
    c->grid.size_y = g.sy;
    for (uint64_t i = 0; i < c->n; i++) {
        int cx, cy;
        cell_of(&g, c->pos[2 * i], c->pos[2 * i + 1], &cx, &cy);
        c->cell_index[i] = (uint32_t)cx + (uint32_t)cy * (uint32_t)g.sx;
    }
}

static int cmp_u32(const void* a, const void* b)
{
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : (x > y);
}

/* ascending j: insertion sort for the short lists of a lattice, qsort beyond */
static void sort_list(uint32_t* a, uint32_t n)
{
    if (n > 48) {
        qsort(a, n, sizeof(uint32_t), cmp_u32);
        return;
    }
    for (uint32_t i = 1; i < n; i++) {
        uint32_t v = a[i], j = i;
        for (; j > 0 && a[j - 1] > v; j--) a[j] = a[j - 1];
        a[j] = v;
    }
}

/* Candidate enumeration.  The SET is fixed by the predicate alone (header comment); any superset generator gives the same
 * lists.  With ONE grid of cell = k * h_max (the reference's CellGrid scheme) a fine particle of a 50:1 scene tests the
 * ~200 000 fine particles of 3 x 3 coarse cells, which makes BASELINE configs[4] (4M particles) unusable as a check.  So the
 * particles are binned by size class L = floor(log2(h / h_min)) and every class gets its own CellGrid with
 * cell = k * (upper h bound of the class); a query particle i visits, per class, the cells within
 * ((h_i + hmax_L) / 2) * k of its position.  Uniform scenes have one class and reproduce the single-grid scheme exactly. */
#define ORC_MAX_CLASSES 24
typedef struct {
    grid_t g;
    float hmax;        /* upper bound of h in the class */
    uint32_t* cstart;  /* [ncell + 1] */
    uint32_t* order;   /* particle indices of the class, cell by cell */
    uint32_t count;
} size_class_grid;

static inline uint32_t visit_candidates(const size_class_grid* G, int ncls, const float* pos, const float* h, float k, uint64_t i, uint32_t* out)
{
    const float xi = pos[2 * i], yi = pos[2 * i + 1], hi = h[i];
    uint32_t cnt = 0;
    for (int L = 0; L < ncls; L++) {
        const size_class_grid* q = &G[L];
        if (!q->count) continue;
        const float reach = orc_hij(hi, q->hmax) * k;
        const grid_t* g = &q->g;
        int x0 = (int)floorf((xi - reach) / g->cs) - g->minx, x1 = (int)floorf((xi + reach) / g->cs) - g->minx;
        int y0 = (int)floorf((yi - reach) / g->cs) - g->miny, y1 = (int)floorf((yi + reach) / g->cs) - g->miny;
        if (x0 < 0) x0 = 0;
        if (y0 < 0) y0 = 0;
        if (x1 >= g->sx) x1 = g->sx - 1;
        if (y1 >= g->sy) y1 = g->sy - 1;
        for (int yy = y0; yy <= y1; yy++) {
            const size_t row = (size_t)yy * (size_t)g->sx;
            for (uint32_t s = q->cstart[row + (size_t)x0]; s < q->cstart[row + (size_t)x1 + 1]; s++) {
                const uint32_t j = q->order[s];
                const float dx = xi - pos[2 * j], dy = yi - pos[2 * j + 1];
                const float sr = orc_hij(hi, h[j]) * k;
                if (orc_norm_sq(dx, dy) < sr * sr) {
                    if (out) out[cnt] = j;
                    cnt++;
                }
            }
        }
    }
    return cnt;
}

int orc_build_neighbors(oracle_ctx* c, float k)
{
    const uint64_t n = c->n;
    if (n == 0) return SPH_OK;
    const float* pos = c->pos;
    const float* h = c->h2;
    float hmin = h[0], hmax = h[0];
    for (uint64_t i = 0; i < n; i++) {
        hmin = fminf(hmin, h[i]);
        hmax = fmaxf(hmax, h[i]);
    }
    if (!(hmin > 0.f) || !isfinite(hmax)) return orc_fail(c, SPH_ERR_POSITION_NOT_FINITE, "oracle: smoothing lengths are not positive and finite");
    int ncls = 1;
    while (ncls < ORC_MAX_CLASSES && hmin * (float)(1u << ncls) <= hmax) ncls++;
    uint8_t* cls = (uint8_t*)malloc(n);
    size_class_grid G[ORC_MAX_CLASSES];
    memset(G, 0, sizeof G);
    if (!cls) return orc_fail(c, SPH_ERR_DEVICE, "oracle: out of memory");
    for (uint64_t i = 0; i < n; i++) {
        int L = 0;
        while (L + 1 < ncls && h[i] >= hmin * (float)(1u << (L + 1))) L++;
        cls[i] = (uint8_t)L;
        G[L].count++;
    }
    int rc = SPH_OK;
    for (int L = 0; L < ncls && !rc; L++) {
        size_class_grid* q = &G[L];
        if (!q->count) continue;
        q->hmax = (L + 1 == ncls) ? hmax : fminf(hmax, hmin * (float)(1u << (L + 1)));
        q->g = make_grid(pos, n, q->hmax * k);   /* box of ALL particles: every query position falls inside */
        const size_t ncell = (size_t)q->g.sx * (size_t)q->g.sy;
        q->cstart = (uint32_t*)calloc(ncell + 1, sizeof(uint32_t));
        q->order = (uint32_t*)malloc((size_t)q->count * sizeof(uint32_t));
        uint32_t* fill = (uint32_t*)malloc(ncell * sizeof(uint32_t));
        if (!q->cstart || !q->order || !fill) {
            free(fill);
            rc = orc_fail(c, SPH_ERR_DEVICE, "oracle: out of memory");
            break;
        }
        for (uint64_t i = 0; i < n; i++) {
            if (cls[i] != L) continue;
            int cx, cy;
            cell_of(&q->g, pos[2 * i], pos[2 * i + 1], &cx, &cy);
            q->cstart[(size_t)cx + (size_t)cy * (size_t)q->g.sx + 1]++;
        }
        for (size_t t = 0; t < ncell; t++) q->cstart[t + 1] += q->cstart[t];
        memcpy(fill, q->cstart, ncell * sizeof(uint32_t));
        for (uint64_t i = 0; i < n; i++) {
            if (cls[i] != L) continue;
            int cx, cy;
            cell_of(&q->g, pos[2 * i], pos[2 * i + 1], &cx, &cy);
            q->order[fill[(size_t)cx + (size_t)cy * (size_t)q->g.sx]++] = (uint32_t)i;
        }
        free(fill);
    }
    /* pass 1: counts; pass 2: fill (lists sorted ascending).  One size class: every particle costs the same, and a static
     * schedule touches the list memory from the threads that read it in the sweeps (first touch on a multi-socket host);
     * several classes: a coarse particle among fine ones tests thousands of candidates, so the work is handed out dynamically */
    int too_many = 0;
    omp_set_schedule(ncls == 1 ? omp_sched_static : omp_sched_dynamic, ncls == 1 ? 0 : 256);
    if (!rc) {
#pragma omp parallel for schedule(runtime) reduction(| : too_many)
        for (int64_t ii = 0; ii < (int64_t)n; ii++) {
            const uint32_t cnt = visit_candidates(G, ncls, pos, h, k, (uint64_t)ii, NULL);
            if (cnt > ORC_MAX_NEIGHBOR_COUNT) too_many = 1;
            c->neighbor_count[ii] = cnt;
        }
        if (too_many)
            rc = orc_fail(c, SPH_ERR_TOO_MANY_NEIGHBORS, "exceeded maximum allowed number of %d neighbors", ORC_MAX_NEIGHBOR_COUNT);
    }
    if (!rc) {
        c->nb_off[0] = 0;
        for (uint64_t i = 0; i < n; i++) c->nb_off[i + 1] = c->nb_off[i] + c->neighbor_count[i];
        if (c->nb_off[n] > c->nb_cap) {
            free(c->nb_idx);
            c->nb_cap = c->nb_off[n] + c->nb_off[n] / 4 + 1024;
            c->nb_idx = (uint32_t*)malloc(c->nb_cap * sizeof(uint32_t));
            if (!c->nb_idx) rc = orc_fail(c, SPH_ERR_DEVICE, "oracle: out of memory");
        }
    }
    if (!rc) {
#pragma omp parallel for schedule(runtime)
        for (int64_t ii = 0; ii < (int64_t)n; ii++) {
            uint32_t* out = c->nb_idx + c->nb_off[ii];
            const uint32_t cnt = visit_candidates(G, ncls, pos, h, k, (uint64_t)ii, out);
            sort_list(out, cnt);
        }
    }
    for (int L = 0; L < ncls; L++) {
        free(G[L].cstart);
        free(G[L].order);
    }
    free(cls);
    return rc;
}

/* neighborhood_search.rs:56-70: order-preserving retain with the same predicate at radius k */
void orc_filter_down(oracle_ctx* c, float k)
{
    const uint64_t n = c->n;
    const float* pos = c->pos;
    const float* h = c->h2;
    if (n == 0) return;
    /* retain() per list, order kept: survivors are counted, offsets re-summed, then written to a second buffer */
    /* the second index buffer is kept across steps (fresh pages every step cost more than the filter itself) */
    if (c->nb_cap2 < c->nb_cap) {
        free(c->nb_idx2);
        c->nb_idx2 = (uint32_t*)malloc((c->nb_cap ? c->nb_cap : 1) * sizeof(uint32_t));
        c->nb_cap2 = c->nb_idx2 ? c->nb_cap : 0;
    }
    uint64_t* keep = (uint64_t*)malloc((n + 1) * sizeof(uint64_t));
    uint32_t* out = c->nb_idx2;
    if (!keep || !out) {   /* out of memory: the sequential in-place form */
        free(keep);
        uint64_t w = 0;
        for (uint64_t i = 0; i < n; i++) {
            uint64_t b = c->nb_off[i], e = c->nb_off[i + 1];
            c->nb_off[i] = w;
            for (uint64_t q = b; q < e; q++) {
                uint32_t j = c->nb_idx[q];
                float dx = pos[2 * i] - pos[2 * j], dy = pos[2 * i + 1] - pos[2 * j + 1];
                float s = orc_hij(h[i], h[j]) * k;
                if (orc_norm_sq(dx, dy) < s * s) c->nb_idx[w++] = j;
            }
        }
        c->nb_off[n] = w;
        return;
    }
#pragma omp parallel for schedule(static)
    for (int64_t ii = 0; ii < (int64_t)n; ii++) {
        const uint64_t i = (uint64_t)ii;
        uint64_t cnt = 0;
        for (uint64_t q = c->nb_off[i]; q < c->nb_off[i + 1]; q++) {
            uint32_t j = c->nb_idx[q];
            float dx = pos[2 * i] - pos[2 * j], dy = pos[2 * i + 1] - pos[2 * j + 1];
            float s = orc_hij(h[i], h[j]) * k;
            cnt += orc_norm_sq(dx, dy) < s * s;
        }
        keep[i + 1] = cnt;
    }
    keep[0] = 0;
    for (uint64_t i = 0; i < n; i++) keep[i + 1] += keep[i];
#pragma omp parallel for schedule(static)
    for (int64_t ii = 0; ii < (int64_t)n; ii++) {
        const uint64_t i = (uint64_t)ii;
        uint64_t w = keep[i];
        for (uint64_t q = c->nb_off[i]; q < c->nb_off[i + 1]; q++) {
            uint32_t j = c->nb_idx[q];
            float dx = pos[2 * i] - pos[2 * j], dy = pos[2 * i + 1] - pos[2 * j + 1];
            float s = orc_hij(h[i], h[j]) * k;
            if (orc_norm_sq(dx, dy) < s * s) out[w++] = j;
        }
    }
    memcpy(c->nb_off, keep, (n + 1) * sizeof(uint64_t));
    c->nb_idx2 = c->nb_idx;   /* ping-pong: both buffers hold nb_cap entries */
    c->nb_cap2 = c->nb_cap;
    c->nb_idx = out;
    free(keep);
}

/* simulation.rs:1810-1863 for every i (O(N^2)); k = 2 */
int orc_check_neighborhood(oracle_ctx* c)
{
    const uint64_t n = c->n;
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(| : bad)
    for (int64_t ii = 0; ii < (int64_t)n; ii++) {
        uint64_t i = (uint64_t)ii;
        uint64_t q = c->nb_off[i], e = c->nb_off[i + 1];
        for (uint64_t j = 0; j < n; j++) {
            float dx = c->pos[2 * i] - c->pos[2 * j], dy = c->pos[2 * i + 1] - c->pos[2 * j + 1];
            float sr = orc_hij(c->h2[i], c->h2[j]) * 2.f;
            int want = orc_norm_sq(dx, dy) < sr * sr;
            int have = (q < e && c->nb_idx[q] == j);
            if (have) q++;
            if (want != have) bad = 1;
        }
    }
    if (bad) return orc_fail(c, SPH_ERR_CHECK_NEIGHBORHOOD, "neighbour list differs from brute force");
    return SPH_OK;
}
