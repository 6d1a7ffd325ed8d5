/*
 * TEST INFRASTRUCTURE -- CPU oracle (see oracle.h).
 *
 * adapt.c: the apply half of single_step_adaptivity, statement by statement:
 *   share_particles   adaptivity/particle_sharing.rs:152-253
 *   merge_particles   adaptivity/particle_merging.rs:270-385  (incl. the sequential swap-with-the-last deletion loop)
 *   split_particles   adaptivity/splitting.rs:19-82
 * ParticleVec::swap / truncate / extend (simulation.rs:248-271) touch EVERY field of the vector and the same calls reach
 * neighs and boundary_handler; the fields kept here are the ones the step reads again.
 */
#include "oracle.h"
#include "sphmath.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MERGE_PARTNER_AVAILABLE 0xFFFFFFFFu /* adaptivity/mod.rs:29 */
#define MERGE_PARTNER_DELETE 0xFFFFFFFEu    /* adaptivity/mod.rs:30 */

/* LevelEstimationState::target_mass (simulation.rs:213-237) */
static float target_mass(const oracle_ctx* c, const sph_params* p, uint64_t i)
{
    float level = fmaxf(c->level[i], -p->maximum_surface_distance);
    float interp = level / -p->maximum_surface_distance;
    float mass_fine = orc_radius_to_sphere_volume(p->particle_radius_fine) * p->rest_density;
    float mass_base = orc_radius_to_sphere_volume(p->particle_radius_base) * p->rest_density;
    if (p->sizing_function == SPH_SIZING_MASS) return mass_fine * (1.f - interp) + mass_base * interp;
    if (p->sizing_function == SPH_SIZING_RADIUS) {
        float r = p->particle_radius_fine * (1.f - interp) + p->particle_radius_base * interp;
        return orc_radius_to_sphere_volume(r) * p->rest_density;
    }
    float e = 1.f / 2.f;
    float r = p->particle_radius_fine * (1.f - powf(interp, e)) + p->particle_radius_base * powf(interp, e);
    return orc_radius_to_sphere_volume(r) * p->rest_density;
}

/* dropped_mass_sharing (particle_sharing.rs:242-253) */
static float dropped_mass_sharing(const oracle_ctx* c, const sph_params* p, const sph_adapt_params* ap, uint64_t i)
{
    float target = target_mass(c, p, i);
    return fminf(c->mass[i] - target, target * ap->max_mass_transfer_sharing * ap->dt);
}

/* the receiving half shared by share_particles (:165-215) and merge_particles (:277-325) */
static void receive(oracle_ctx* c, const sph_params* p, uint64_t i, uint64_t j, float dropped_mass_j, uint16_t counter_j)
{
    float mass_i = c->mass[i];
    float mass_n = dropped_mass_j / (float)counter_j;
    float mass = mass_i + mass_n;
    for (int d = 0; d < 2; d++) {
        c->vel[2 * i + d] = (mass_i * c->vel[2 * i + d] + mass_n * c->vel[2 * j + d]) / mass;
        c->pos[2 * i + d] = (mass_i * c->pos[2 * i + d] + mass_n * c->pos[2 * j + d]) / mass;
    }
    c->mass[i] = mass;
    c->h2_next[i] = orc_h_from_mass(mass, p->rest_density);
}

static void invalidate_lists(oracle_ctx* c) { memset(c->nb_off, 0, (c->n + 1) * sizeof(uint64_t)); }

int oracle_share_particles(oracle_ctx* c, const sph_params* p, const sph_adapt_params* ap, const uint32_t* partner, const uint16_t* counter)
{
    if (!c || !p || !ap || (c->n && (!partner || !counter))) return SPH_ERR_INVALID_ARGUMENT;
    /* receivers read their donor's values of BEFORE the call (the donor is written by the second loop only) */
    for (uint64_t i = 0; i < c->n; i++) {
        uint32_t j = partner[i];
        if (j == MERGE_PARTNER_AVAILABLE || j == MERGE_PARTNER_DELETE) continue;
        if (j >= c->n) return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "merge_partner holds an index outside the particle vector (particle i=%llu)", (unsigned long long)i);
        if (counter[j] < ap->minimum_share_partners) continue;
        receive(c, p, i, j, dropped_mass_sharing(c, p, ap, j), counter[j]);
    }
    for (uint64_t i = 0; i < c->n; i++) {
        if (partner[i] != MERGE_PARTNER_DELETE) continue;
        if (counter[i] < ap->minimum_share_partners) continue;
        float dropped = dropped_mass_sharing(c, p, ap, i);
        c->mass[i] -= dropped;
        c->h2_next[i] = orc_h_from_mass(c->mass[i], p->rest_density);
    }
    invalidate_lists(c);
    return SPH_OK;
}

static void swapf(float* a, uint64_t i, uint64_t j, int w)
{
    for (int d = 0; d < w; d++) { float t = a[w * i + d]; a[w * i + d] = a[w * j + d]; a[w * j + d] = t; }
}
static void swap8(uint8_t* a, uint64_t i, uint64_t j) { uint8_t t = a[i]; a[i] = a[j]; a[j] = t; }

/* ParticleVec::swap + neighs.swap + boundary_handler.swap for the persistent fields */
static void swap_particles(oracle_ctx* c, uint64_t i, uint64_t j)
{
    swapf(c->mass, i, j, 1); swapf(c->pos, i, j, 2); swapf(c->vel, i, j, 2); swapf(c->h2, i, j, 1); swapf(c->h2_next, i, j, 1);
    swapf(c->level, i, j, 1); swapf(c->level_old, i, j, 1);
    swapf(c->lam, i, j, ORC_MAX_PLANES); swapf(c->lam_gx, i, j, ORC_MAX_PLANES); swapf(c->lam_gy, i, j, ORC_MAX_PLANES);
    swap8(c->lam_n, i, j); swap8(c->size_class, i, j);
}

int oracle_merge_particles(oracle_ctx* c, const sph_params* p, const sph_adapt_params* ap, const uint32_t* partner_in, const uint16_t* counter_in)
{
    if (!c || !p || !ap || (c->n && (!partner_in || !counter_in))) return SPH_ERR_INVALID_ARGUMENT;
    if (c->n == 0) return SPH_OK;
    /* merge_partner / merge_counter are ParticleVec fields: they are swapped along with everything else in the loop below */
    uint32_t* partner = (uint32_t*)malloc(c->n * sizeof(uint32_t));
    uint16_t* counter = (uint16_t*)malloc(c->n * sizeof(uint16_t));
    memcpy(partner, partner_in, c->n * sizeof(uint32_t));
    memcpy(counter, counter_in, c->n * sizeof(uint16_t));
    for (uint64_t i = 0; i < c->n; i++) {
        uint32_t j = partner[i];
        if (j == MERGE_PARTNER_AVAILABLE || j == MERGE_PARTNER_DELETE) continue;
        if (j >= c->n) {
            free(partner); free(counter);
            return orc_fail(c, SPH_ERR_INVALID_ARGUMENT, "merge_partner holds an index outside the particle vector (particle i=%llu)", (unsigned long long)i);
        }
        if (counter[j] < ap->minimum_merge_partners) continue;
        receive(c, p, i, j, c->mass[j] /* dropped_mass_merging: the whole mass (:372-385) */, counter[j]);
    }
    /* delete particles by swapping them to the end of the array (:337-365) */
    uint64_t last_particle_id = c->n - 1;
    uint64_t i = 0;
    int emptied = 0;
    for (;;) {
        if (i > last_particle_id) break;
        if (partner[i] == MERGE_PARTNER_DELETE && counter[i] >= ap->minimum_merge_partners) {
            float dropped = c->mass[i];
            c->mass[i] -= dropped;
            if (c->mass[i] < 0.000001f) {
                swap_particles(c, i, last_particle_id);
                { uint32_t t = partner[i]; partner[i] = partner[last_particle_id]; partner[last_particle_id] = t; }
                { uint16_t t = counter[i]; counter[i] = counter[last_particle_id]; counter[last_particle_id] = t; }
                if (last_particle_id == 0) { emptied = 1; break; } /* usize underflow in the reference: it panics when everything is deleted */
                last_particle_id -= 1;
                continue;
            }
        }
        i += 1;
    }
    c->n = emptied ? 0 : last_particle_id + 1; /* truncate */
    free(partner); free(counter);
    invalidate_lists(c);
    return SPH_OK;
}

/* SplitPatterns (splitting.rs:84-120): pattern k has k + 2 children; positions concatenated */
static float* g_patterns = NULL; /* process-wide is enough for the tests: one table (split-patterns.yaml) */
static uint32_t g_n_patterns = 0;

int oracle_set_split_patterns(oracle_ctx* c, uint32_t n_patterns, const float* pos_s_xy)
{
    if (!c || (n_patterns && !pos_s_xy)) return SPH_ERR_INVALID_ARGUMENT;
    size_t n_pos = 0;
    for (uint32_t k = 0; k < n_patterns; k++) n_pos += (size_t)k + 2;
    free(g_patterns);
    g_patterns = (float*)malloc((n_pos ? n_pos : 1) * 2 * sizeof(float));
    memcpy(g_patterns, pos_s_xy, n_pos * 2 * sizeof(float));
    g_n_patterns = n_patterns;
    return SPH_OK;
}

int oracle_split_particles(oracle_ctx* c, const sph_params* p, const sph_adapt_params* ap)
{
    if (!c || !p || !ap) return SPH_ERR_INVALID_ARGUMENT;
    const uint64_t num_particles = c->n;
    uint64_t new_particle_id = num_particles;
    const uint64_t max_children = (uint64_t)g_n_patterns + 1;
    for (uint64_t i = 0; i < num_particles; i++) {
        if (c->size_class[i] != 4 /* TooLarge */) continue;
        if (isnan(c->level[i])) return orc_fail(c, SPH_ERR_NO_SPLIT_PATTERN, "internal error: entered unreachable code (particle i=%llu)", (unsigned long long)i);
        float target = target_mass(c, p, i);
        float r = roundf(c->mass[i] / target);
        uint64_t num_children = !(r > 0.f) ? 0 : (uint64_t)r; /* `as usize` saturates, NaN -> 0 */
        if (num_children > max_children) {
            if (ap->fail_on_missing_split_pattern)
                return orc_fail(c, SPH_ERR_NO_SPLIT_PATTERN, "no split pattern for a 1-to-%llu split (particle i=%llu)", (unsigned long long)num_children, (unsigned long long)i);
            num_children = max_children;
        }
        if (!(num_children > 1)) return orc_fail(c, SPH_ERR_NO_SPLIT_PATTERN, "assertion failed: num_children > 1 (particle i=%llu)", (unsigned long long)i);
        const float* pattern = g_patterns + 2 * ((num_children - 1) * num_children / 2 - 1);
        float particle_radius = orc_sphere_volume_to_radius(c->mass[i] / 1.f /* INIT_REST_DENSITY */);
        float child_mass = c->mass[i] / (float)num_children;
        float child_h_next = orc_h_from_mass(child_mass, p->rest_density);
        float ovx = c->vel[2 * i], ovy = c->vel[2 * i + 1], opx = c->pos[2 * i], opy = c->pos[2 * i + 1];
        float olevel = c->level[i], olevel_old = c->level_old[i];
        float scale = particle_radius;
        /* particles.extend / neighs.extend / boundary_handler.extend (num_children - 1) */
        if (c->n + num_children - 1 > c->cap) return orc_fail(c, SPH_ERR_CAPACITY, "splitting exceeds the capacity");
        for (uint64_t q = c->n; q < c->n + num_children - 1; q++) {
            c->mass[q] = 0.f; c->pos[2 * q] = c->pos[2 * q + 1] = 0.f; c->vel[2 * q] = c->vel[2 * q + 1] = 0.f;
            c->h2[q] = 0.f; c->h2_next[q] = 0.f; c->level[q] = NAN; c->level_old[q] = 0.f; c->lam_n[q] = 0; c->size_class[q] = 2;
        }
        c->n += num_children - 1;
        for (uint64_t child_id = 0; child_id < num_children; child_id++) {
            if (child_id == 0) {
                c->mass[i] = child_mass;
                c->vel[2 * i] = ovx; c->vel[2 * i + 1] = ovy;
                c->pos[2 * i] = opx + pattern[0] * scale; c->pos[2 * i + 1] = opy + pattern[1] * scale;
                c->h2[i] = child_h_next;
                c->h2_next[i] = child_h_next;
                c->level[i] = olevel;
                c->level_old[i] = olevel_old;
            } else {
                uint64_t q = new_particle_id;
                c->mass[q] = child_mass;
                c->vel[2 * q] = ovx; c->vel[2 * q + 1] = ovy;
                c->pos[2 * q] = opx + pattern[2 * child_id] * scale; c->pos[2 * q + 1] = opy + pattern[2 * child_id + 1] * scale;
                c->h2[i] = child_h_next;         /* (sic) the parent's slot, splitting.rs:73 */
                c->h2_next[q] = child_h_next;
                c->level[q] = olevel;
                c->level_old[i] = olevel_old;    /* (sic) splitting.rs:76 */
                new_particle_id += 1;
            }
        }
    }
    invalidate_lists(c);
    return SPH_OK;
}
