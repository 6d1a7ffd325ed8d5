/*
 * TEST INFRASTRUCTURE -- CPU oracle (see oracle.h).
 *
 * neigh.c: neighbour sets.  The reference finds candidates with an R*-tree (rstar 0.9.3,
 * Cargo.lock:1963 -- not vendored) and then re-pins the SET itself:
 *   N(i) = { j : |x_i - x_j|^2 < s_ij^2 },  s_ij = ((h_i + h_j) * 0.5) * k,   self included
 * (strict filter neighborhood_search.rs:138-147, symmetrisation :157-185, brute-force definition
 * :214-237 and simulation.rs:1810-1863).  Candidate enumeration here uses the reference's OWN
 * uniform-grid scheme (neighborhood_search.rs:243-321, CellGrid :355-410) with cell size
 * = k * h_max, which is a superset generator for that predicate.  List ORDER is unpinned in the
 * reference (R*-tree traversal order); the oracle uses ascending j.
 */
#include "oracle.h"
#include "sphmath.h"

#include <math.h>
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

int orc_build_neighbors(oracle_ctx* c, float k)
{
    const uint64_t n = c->n;
    if (n == 0) return SPH_OK;
    const float* pos = c->pos;
    const float* h = c->h2;
    const float cs = max_h(c) * k;
    grid_t g = make_grid(pos, n, cs);
    const size_t ncell = (size_t)g.sx * (size_t)g.sy;

    uint32_t* cstart = (uint32_t*)calloc(ncell + 1, sizeof(uint32_t));
    uint32_t* cellid = (uint32_t*)malloc(n * sizeof(uint32_t));
    uint32_t* order = (uint32_t*)malloc(n * sizeof(uint32_t));
    if (!cstart || !cellid || !order) return orc_fail(c, SPH_ERR_DEVICE, "oracle: out of memory");
    for (uint64_t i = 0; i < n; i++) {
        int cx, cy;
        cell_of(&g, pos[2 * i], pos[2 * i + 1], &cx, &cy);
        cellid[i] = (uint32_t)cx + (uint32_t)cy * (uint32_t)g.sx;
        cstart[cellid[i] + 1]++;
    }
    for (size_t q = 0; q < ncell; q++) cstart[q + 1] += cstart[q];
    {
        uint32_t* fill = (uint32_t*)malloc(ncell * sizeof(uint32_t));
        memcpy(fill, cstart, ncell * sizeof(uint32_t));
        for (uint64_t i = 0; i < n; i++) order[fill[cellid[i]]++] = (uint32_t)i;
        free(fill);
    }

    /* pass 1: counts; pass 2: fill (lists sorted ascending) */
    int too_many = 0;
#pragma omp parallel for schedule(static) reduction(| : too_many)
    for (int64_t ii = 0; ii < (int64_t)n; ii++) {
        uint64_t i = (uint64_t)ii;
        const float xi = pos[2 * i], yi = pos[2 * i + 1], hi = h[i];
        int cx = (int)(cellid[i] % (uint32_t)g.sx), cy = (int)(cellid[i] / (uint32_t)g.sx);
        uint32_t cnt = 0;
        for (int oy = -1; oy <= 1; oy++) {
            int yy = cy + oy;
            if (yy < 0 || yy >= g.sy) continue;
            for (int ox = -1; ox <= 1; ox++) {
                int xx = cx + ox;
                if (xx < 0 || xx >= g.sx) continue;
                size_t cc = (size_t)xx + (size_t)yy * (size_t)g.sx;
                for (uint32_t q = cstart[cc]; q < cstart[cc + 1]; q++) {
                    uint32_t j = order[q];
                    float dx = xi - pos[2 * j], dy = yi - pos[2 * j + 1];
                    float s = orc_hij(hi, h[j]) * k;
                    if (orc_norm_sq(dx, dy) < s * s) cnt++;
                }
            }
        }
        if (cnt > ORC_MAX_NEIGHBOR_COUNT) too_many = 1;
        c->neighbor_count[i] = cnt;
    }
    if (too_many) {
        free(cstart); free(cellid); free(order);
        return orc_fail(c, SPH_ERR_TOO_MANY_NEIGHBORS, "exceeded maximum allowed number of %d neighbors",
                        ORC_MAX_NEIGHBOR_COUNT);
    }
    c->nb_off[0] = 0;
    for (uint64_t i = 0; i < n; i++) c->nb_off[i + 1] = c->nb_off[i] + c->neighbor_count[i];
    if (c->nb_off[n] > c->nb_cap) {
        free(c->nb_idx);
        c->nb_cap = c->nb_off[n] + c->nb_off[n] / 4 + 1024;
        c->nb_idx = (uint32_t*)malloc(c->nb_cap * sizeof(uint32_t));
        if (!c->nb_idx) return orc_fail(c, SPH_ERR_DEVICE, "oracle: out of memory");
    }
#pragma omp parallel for schedule(static)
    for (int64_t ii = 0; ii < (int64_t)n; ii++) {
        uint64_t i = (uint64_t)ii;
        const float xi = pos[2 * i], yi = pos[2 * i + 1], hi = h[i];
        int cx = (int)(cellid[i] % (uint32_t)g.sx), cy = (int)(cellid[i] / (uint32_t)g.sx);
        uint32_t* out = c->nb_idx + c->nb_off[i];
        uint32_t cnt = 0;
        for (int oy = -1; oy <= 1; oy++) {
            int yy = cy + oy;
            if (yy < 0 || yy >= g.sy) continue;
            for (int ox = -1; ox <= 1; ox++) {
                int xx = cx + ox;
                if (xx < 0 || xx >= g.sx) continue;
                size_t cc = (size_t)xx + (size_t)yy * (size_t)g.sx;
                for (uint32_t q = cstart[cc]; q < cstart[cc + 1]; q++) {
                    uint32_t j = order[q];
                    float dx = xi - pos[2 * j], dy = yi - pos[2 * j + 1];
                    float s = orc_hij(hi, h[j]) * k;
                    if (orc_norm_sq(dx, dy) < s * s) out[cnt++] = j;
                }
            }
        }
        qsort(out, cnt, sizeof(uint32_t), cmp_u32);
    }
    free(cstart); free(cellid); free(order);
    return SPH_OK;
}

/* neighborhood_search.rs:56-70: order-preserving retain with the same predicate at radius k */
void orc_filter_down(oracle_ctx* c, float k)
{
    const uint64_t n = c->n;
    const float* pos = c->pos;
    const float* h = c->h2;
    /* compact in place: new offsets <= old offsets, process sequentially */
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
