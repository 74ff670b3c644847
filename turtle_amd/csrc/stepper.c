/*
 * stepper.c -- the layered-topography stepper handle [ref src/turtle/
 * stepper.c:379-931, stepper.h:45-110].
 *
 * The reference keeps linked lists of layers -> (data, offset) metas -> data
 * sources and walks them per sample on the CPU.  Here the same description is
 * kept in small host arrays and FLATTENED into POD tables in HBM
 * (internal.h); every sample, step and trace is then computed by the kernels
 * of device.hip.  Scalar entry points are the batch ones with n = 1.
 */
#include "host.h"

#include <time.h>

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* what the stepper holds in HBM, on the device it was last used on */
static void stepper_release_device(struct turtle_stepper * s)
{
        int drained = 1;
        if ((s->d_tables != NULL) || (s->d_stats != NULL) || (s->d_parked != NULL))
                drained = (tamd_dev_sync_device(s->device) == 0);
        if (drained) { /* else: leaked, rather than freed under a launch that may still read it */
                tamd_dev_free_on(s->device, s->d_tables);
                tamd_dev_free_on(s->device, s->d_stats);
                tamd_dev_free_on(s->device, s->d_parked);
        }
        s->d_tables = NULL, s->d_stats = NULL, s->d_parked = NULL, s->d_scratch_ds = NULL;
        s->d_tables_size = 0, s->parked_capacity = 0, s->epoch = 0;
}

/* The stepper's HBM (tables, counters, scratch) is on the device of the thread
 * that uses it: what it held elsewhere goes when that thread has moved on.
 * Before anything of it is allocated or used. */
static int stepper_bind_device(struct turtle_stepper * s)
{
        if (tamd_dev_init()) return 1;
        if (s->device != tamd_dev_current()) {
                if (s->device >= 0) stepper_release_device(s);
                s->device = tamd_dev_current();
        }
        return 0;
}

/* ---- construction [ref stepper.c:547-600] -------------------------------- */

enum turtle_return turtle_stepper_create(struct turtle_stepper ** stepper)
{
        TAMD_ERROR_INIT(&turtle_stepper_create);
        struct turtle_stepper * s = calloc(1, sizeof(*s));
        if (s == NULL)
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        s->device = -1;
        s->local_range = 1.; /* defaults [ref stepper.c:558-560] */
        s->slope_factor = 0.4;
        s->resolution_factor = 1E-02;
        *stepper = s;
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_stepper_destroy(struct turtle_stepper ** stepper)
{
        if ((stepper == NULL) || (*stepper == NULL)) return TURTLE_RETURN_SUCCESS;
        struct turtle_stepper * s = *stepper;
        int i;
        for (i = 0; i < s->n_data; i++) /* owned clients [ref stepper.c:578-586] */
                if (s->data[i].client != NULL) turtle_client_destroy(&s->data[i].client);
        for (i = 0; i < s->n_layers; i++) free(s->layers[i].meta);
        stepper_release_device(s);
        free(s->data);
        free(s->layers);
        free(s);
        *stepper = NULL;
        return TURTLE_RETURN_SUCCESS;
}

/* [ref stepper.c:364-388]: an empty top layer is reused */
static int push_layer(struct turtle_stepper * s)
{
        if ((s->n_layers > 0) && (s->layers[s->n_layers - 1].size == 0)) return 0;
        if (s->n_layers == s->cap_layers) {
                const int cap = s->cap_layers ? 2 * s->cap_layers : 4;
                struct tamd_layer * l = realloc(s->layers, cap * sizeof(*l));
                if (l == NULL) return 1;
                s->layers = l, s->cap_layers = cap;
        }
        memset(&s->layers[s->n_layers++], 0, sizeof(*s->layers));
        tamd_geometry_changed();
        return 0;
}

enum turtle_return turtle_stepper_add_layer(struct turtle_stepper * stepper)
{
        TAMD_ERROR_INIT(&turtle_stepper_add_layer);
        if (push_layer(stepper))
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        return TURTLE_RETURN_SUCCESS;
}

static int push_data(struct turtle_stepper * s, const struct tamd_data * d)
{
        if (s->n_data == s->cap_data) {
                const int cap = s->cap_data ? 2 * s->cap_data : 4;
                struct tamd_data * p = realloc(s->data, cap * sizeof(*p));
                if (p == NULL) return -1;
                s->data = p, s->cap_data = cap;
        }
        s->data[s->n_data] = *d;
        return s->n_data++;
}

/* [ref stepper.c:390-409]: the first data creates layer 0; metas append to
 * the top layer */
static int push_meta(struct turtle_stepper * s, int data, double offset)
{
        if ((s->n_layers == 0) && push_layer(s)) return 1;
        struct tamd_layer * l = &s->layers[s->n_layers - 1];
        if (l->size == l->capacity) {
                const int cap = l->capacity ? 2 * l->capacity : 4;
                struct tamd_layer_meta * m = realloc(l->meta, cap * sizeof(*m));
                if (m == NULL) return 1;
                l->meta = m, l->capacity = cap;
        }
        l->meta[l->size].data = data;
        l->meta[l->size].offset = offset;
        l->size++;
        s->last.valid = 0; /* (the host's cached sample: scalar.c) */
        tamd_geometry_changed();
        return 0;
}

/* [ref stepper.c:411-470]: one data per distinct stack; a locked stack gets a
 * client owned by the stepper */
enum turtle_return turtle_stepper_add_stack(
    struct turtle_stepper * stepper, struct turtle_stack * stack, double offset)
{
        TAMD_ERROR_INIT(&turtle_stepper_add_stack);
        int i, data = -1;
        for (i = 0; i < stepper->n_data; i++)
                if ((stepper->data[i].kind == TAMD_STACK) && (stepper->data[i].stack == stack))
                        data = i;
        if (data < 0) {
                struct tamd_data d = { TAMD_STACK, NULL, stack, NULL };
                if (stack->lock != NULL) {
                        const enum turtle_return rc = turtle_client_create(&d.client, stack);
                        if (rc != TURTLE_RETURN_SUCCESS) return rc;
                }
                data = push_data(stepper, &d);
                if (data < 0) {
                        turtle_client_destroy(&d.client);
                        return TAMD_RAISE(
                            TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
                }
        }
        if (push_meta(stepper, data, offset))
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        return TURTLE_RETURN_SUCCESS;
}

/* [ref stepper.c:472-509] */
enum turtle_return turtle_stepper_add_map(
    struct turtle_stepper * stepper, struct turtle_map * map, double offset)
{
        TAMD_ERROR_INIT(&turtle_stepper_add_map);
        int i, data = -1;
        for (i = 0; i < stepper->n_data; i++)
                if ((stepper->data[i].kind == TAMD_MAP) && (stepper->data[i].map == map))
                        data = i;
        if (data < 0) {
                const struct tamd_data d = { TAMD_MAP, map, NULL, NULL };
                data = push_data(stepper, &d);
        }
        if ((data < 0) || push_meta(stepper, data, offset))
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        return TURTLE_RETURN_SUCCESS;
}

/* [ref stepper.c:511-545] */
enum turtle_return turtle_stepper_add_flat(struct turtle_stepper * stepper, double offset)
{
        TAMD_ERROR_INIT(&turtle_stepper_add_flat);
        int i, data = -1;
        for (i = 0; i < stepper->n_data; i++)
                if (stepper->data[i].kind == TAMD_FLAT) data = i;
        if (data < 0) {
                const struct tamd_data d = { TAMD_FLAT, NULL, NULL, NULL };
                data = push_data(stepper, &d);
        }
        if ((data < 0) || push_meta(stepper, data, offset))
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        return TURTLE_RETURN_SUCCESS;
}

/* A second stepper over the same geometry: the layers and their data in the order they
 * were added, the geoid and the settings (turtle_amd.h: one stepper is one stream of
 * calls; a batch more in flight takes a stepper more).  Built with the public calls, so
 * that it is exactly what the caller's own sequence of them would have made. */
enum turtle_return turtle_amd_stepper_clone(
    const struct turtle_stepper * stepper, struct turtle_stepper ** clone)
{
        TAMD_ERROR_INIT(&turtle_amd_stepper_clone);
        if ((stepper == NULL) || (clone == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "a stepper and where to put its clone");
        *clone = NULL;
        struct turtle_stepper * c = NULL;
        enum turtle_return rc = turtle_stepper_create(&c);
        int i, j;
        for (i = 0; (rc == TURTLE_RETURN_SUCCESS) && (i < stepper->n_layers); i++) {
                /* (an empty layer is one more add_layer: push_layer reuses an empty top layer) */
                rc = turtle_stepper_add_layer(c);
                for (j = 0; (rc == TURTLE_RETURN_SUCCESS) && (j < stepper->layers[i].size); j++) {
                        const struct tamd_layer_meta * m = &stepper->layers[i].meta[j];
                        const struct tamd_data * d = &stepper->data[m->data];
                        if (d->kind == TAMD_MAP)
                                rc = turtle_stepper_add_map(c, d->map, m->offset);
                        else if (d->kind == TAMD_STACK)
                                rc = turtle_stepper_add_stack(c, d->stack, m->offset);
                        else
                                rc = turtle_stepper_add_flat(c, m->offset);
                }
        }
        if (rc != TURTLE_RETURN_SUCCESS) {
                turtle_stepper_destroy(&c);
                return rc;
        }
        c->geoid = stepper->geoid;
        c->local_range = stepper->local_range;
        c->slope_factor = stepper->slope_factor;
        c->resolution_factor = stepper->resolution_factor;
        *clone = c;
        return TURTLE_RETURN_SUCCESS;
}

/* ---- setters/getters [ref stepper.c:617-672] ----------------------------- */

void turtle_stepper_geoid_set(struct turtle_stepper * stepper, struct turtle_map * geoid)
{
        stepper->geoid = geoid;
        stepper->last.valid = 0;
        tamd_geometry_changed();
}

struct turtle_map * turtle_stepper_geoid_get(const struct turtle_stepper * stepper)
{
        return stepper->geoid;
}

double turtle_stepper_range_get(const struct turtle_stepper * stepper)
{
        return stepper->local_range;
}

void turtle_stepper_range_set(struct turtle_stepper * stepper, double range)
{
        stepper->local_range = range; /* recorded only: see turtle_amd.h */
}

/* nothing is cached between calls, so there is no history to reset */
/* [ref stepper.c:662-672]: forgets the cached sample (the scalar calls on the host keep one) */
void turtle_stepper_reset(struct turtle_stepper * stepper) { stepper->last.valid = 0; }

double turtle_stepper_slope_get(const struct turtle_stepper * stepper)
{
        return stepper->slope_factor;
}

void turtle_stepper_slope_set(struct turtle_stepper * stepper, double slope)
{
        stepper->slope_factor = slope;
}

double turtle_stepper_resolution_get(const struct turtle_stepper * stepper)
{
        return stepper->resolution_factor;
}

void turtle_stepper_resolution_set(struct turtle_stepper * stepper, double resolution)
{
        stepper->resolution_factor = resolution;
}

/* ---- flattening ---------------------------------------------------------- */

struct grid_list {
        struct turtle_map ** map;
        int n, cap;
};

static int grid_index(struct grid_list * g, struct turtle_map * m)
{
        int i;
        for (i = 0; i < g->n; i++)
                if (g->map[i] == m) return i;
        if (g->n == g->cap) {
                const int cap = g->cap ? 2 * g->cap : 16;
                struct turtle_map ** p = realloc(g->map, cap * sizeof(*p));
                if (p == NULL) return -1;
                g->map = p, g->cap = cap;
        }
        g->map[g->n] = m;
        return g->n++;
}

static int stepper_flatten_locked(struct turtle_stepper * s, char * message, size_t size);

/* Returns 0, -1 for a device error (tamd_dev_error has the text), or a
 * positive enum turtle_return with `message` filled in. */
int tamd_stepper_flatten(struct turtle_stepper * s, char * message, size_t size)
{
        if (stepper_bind_device(s)) return -1;
        if (s->d_stats == NULL) {
                const size_t words = 4 + TAMD_TRACE_COUNTERS;
                if (tamd_dev_malloc((void **)&s->d_stats, words * sizeof(*s->d_stats))) return -1;
                if (tamd_dev_zero(s->d_stats, words * sizeof(*s->d_stats))) return -1;
        }
        s->view.slope = s->slope_factor;
        s->view.resolution = s->resolution_factor;
        if ((s->epoch == tamd_geometry_epoch_get()) && (s->d_tables != NULL)) return 0;
        /* the lists of tiles are read, and maps uploaded, under the lock */
        tamd_geometry_lock();
        const int rc_ = stepper_flatten_locked(s, message, size);
        tamd_geometry_unlock();
        return rc_;
}

static int stepper_flatten_locked(struct turtle_stepper * s, char * message, size_t size)
{
        const unsigned long epoch_now = tamd_geometry_epoch_get();

        /* the tiles that are in memory go into the tables; the others read
         * TAMD_TILE_PAGED there and come in when a batch wants them (paging.c) */
        int i, j, n_stacks = 0, n_tiles = 0, n_metas = 0;
        for (i = 0; i < s->n_data; i++) {
                if (s->data[i].kind != TAMD_STACK) continue;
                n_stacks++;
                n_tiles += s->data[i].stack->latitude_n * s->data[i].stack->longitude_n;
        }
        s->n_table = n_tiles;
        for (i = 0; i < s->n_layers; i++) n_metas += s->layers[i].size;

        /* unique grids: maps, tiles, geoid */
        struct grid_list gl = { NULL, 0, 0 };
        int rc = 0;
        int * data_src = calloc(s->n_data + 1, sizeof(*data_src));
        int * tiles = calloc(n_tiles + 1, sizeof(*tiles));
        struct tamd_stack * stacks = calloc(n_stacks + 1, sizeof(*stacks));
        if ((data_src == NULL) || (tiles == NULL) || (stacks == NULL)) rc = 1;
        int stack_count = 0, tile_count = 0;
        for (i = 0; (rc == 0) && (i < s->n_data); i++) {
                struct tamd_data * d = &s->data[i];
                if (d->kind == TAMD_MAP) {
                        data_src[i] = grid_index(&gl, d->map);
                        if (data_src[i] < 0) rc = 1;
                } else if (d->kind == TAMD_STACK) {
                        struct turtle_stack * st = d->stack;
                        struct tamd_stack * t = &stacks[stack_count];
                        t->lat0 = st->latitude_0, t->lon0 = st->longitude_0;
                        t->dlat = st->latitude_delta, t->dlon = st->longitude_delta;
                        t->inv_dlat = 1. / t->dlat, t->inv_dlon = 1. / t->dlon;
                        t->nlat = st->latitude_n, t->nlon = st->longitude_n;
                        t->tile_first = tile_count;
                        const int slots = st->latitude_n * st->longitude_n;
                        for (j = 0; (rc == 0) && (j < slots); j++) {
                                int g = (st->path[j] != NULL) ? TAMD_TILE_PAGED : TAMD_TILE_NONE;
                                if (st->tile[j] != NULL) {
                                        g = grid_index(&gl, st->tile[j]);
                                        if (g < 0) rc = 1;
                                }
                                tiles[tile_count++] = g;
                        }
                        data_src[i] = stack_count++;
                }
        }
        int geoid = -1;
        if ((rc == 0) && (s->geoid != NULL)) {
                geoid = grid_index(&gl, s->geoid);
                if (geoid < 0) rc = 1;
        }
        if (rc != 0) {
                free(gl.map), free(data_src), free(tiles), free(stacks);
                snprintf(message, size, "could not allocate memory");
                return TURTLE_RETURN_MEMORY_ERROR;
        }

        /* one blob: grids | stacks | metas | layer_first | tiles | slot_nodes */
        const size_t o_grids = 0;
        const size_t o_stacks = o_grids + (size_t)(gl.n + 1) * sizeof(struct tamd_grid);
        const size_t o_metas = o_stacks + (size_t)(n_stacks + 1) * sizeof(struct tamd_stack);
        const size_t o_first = o_metas + (size_t)(n_metas + 1) * sizeof(struct tamd_meta);
        const size_t o_tiles = o_first + (size_t)(s->n_layers + 2) * sizeof(int);
        const size_t o_nodes =
            (o_tiles + (size_t)(n_tiles + 1) * sizeof(int) + 15) & ~(size_t)15;
        const size_t bytes = o_nodes + (size_t)(n_tiles + 1) * sizeof(void *);
        char * host = calloc(1, bytes);
        if (host == NULL) {
                free(gl.map), free(data_src), free(tiles), free(stacks);
                snprintf(message, size, "could not allocate memory");
                return TURTLE_RETURN_MEMORY_ERROR;
        }
        struct tamd_grid * h_grids = (struct tamd_grid *)(host + o_grids);
        struct tamd_meta * h_metas = (struct tamd_meta *)(host + o_metas);
        int * h_first = (int *)(host + o_first);
        int dev_fail = 0;
        for (i = 0; i < gl.n; i++)
                if (tamd_map_sync(gl.map[i], &h_grids[i])) dev_fail = 1;
        /* regular stacks: shared tile shape + one node pointer per slot */
        const uint16_t ** h_nodes = (const uint16_t **)(host + o_nodes);
        int fast_ok = 1;
        for (i = 0; i < gl.n; i++)
                if ((h_grids[i].nx < 2) || (h_grids[i].ny < 2)) fast_ok = 0;
        for (i = 0; i < n_stacks; i++) {
                struct tamd_stack * t = &stacks[i];
                const int slots = t->nlat * t->nlon;
                const struct tamd_grid * proto = NULL;
                /* slot and cell share a 32-bit id in the lanes' caches (slot << 24 |
                 * cell, ~0u: empty): at most 254 slots of at most 2^24 cells */
                int regular = (slots > 0) && (slots <= 254);
                t->nodes_first = t->tile_first;
                for (j = 0; j < slots; j++) {
                        const int g = tiles[t->tile_first + j];
                        h_nodes[t->nodes_first + j] = (g >= 0) ? h_grids[g].nodes : NULL;
                        if (g < 0) continue;
                        const struct tamd_grid * q = &h_grids[g];
                        if (proto == NULL) proto = q;
                        if (((size_t)q->nx * (size_t)q->ny > ((size_t)1 << 24)) || (q->nx != proto->nx) || (q->ny != proto->ny) ||
                            (q->dx != proto->dx) || (q->dy != proto->dy) ||
                            (q->z0 != proto->z0) || (q->dz != proto->dz) ||
                            (q->is_signed != proto->is_signed) ||
                            (q->x0 != t->lon0 + (j % t->nlon) * t->dlon) ||
                            (q->y0 != t->lat0 + (j / t->nlon) * t->dlat) ||
                            /* ... and spans its lattice cell, no more */
                            (fabs((q->nx - 1) * q->dx - t->dlon) > 1E-10 * fabs(t->dlon)) ||
                            (fabs((q->ny - 1) * q->dy - t->dlat) > 1E-10 * fabs(t->dlat)))
                                regular = 0;
                }
                t->regular = regular && (proto != NULL);
                if (t->regular) t->proto = *proto;
        }
        memcpy(host + o_stacks, stacks, (size_t)n_stacks * sizeof(*stacks));
        memcpy(host + o_tiles, tiles, (size_t)n_tiles * sizeof(*tiles));
        int m = 0;
        for (i = 0; i < s->n_layers; i++) {
                h_first[i] = m;
                /* last added first [ref stepper.c:722-724] */
                for (j = s->layers[i].size - 1; j >= 0; j--, m++) {
                        const struct tamd_layer_meta * lm = &s->layers[i].meta[j];
                        h_metas[m].kind = s->data[lm->data].kind;
                        h_metas[m].src = data_src[lm->data];
                        h_metas[m].offset = lm->offset;
                }
        }
        h_first[s->n_layers] = m;

        if (!dev_fail && (bytes > s->d_tables_size)) {
                tamd_dev_sync();
                tamd_dev_free(s->d_tables);
                s->d_tables = NULL, s->d_tables_size = 0;
                if (tamd_dev_malloc(&s->d_tables, bytes))
                        dev_fail = 1;
                else
                        s->d_tables_size = bytes;
        }
        if (!dev_fail && tamd_dev_h2d(s->d_tables, host, bytes)) dev_fail = 1;

        if (!dev_fail) {
                char * d = s->d_tables;
                s->view.grids = (const struct tamd_grid *)(d + o_grids);
                s->view.stacks = (const struct tamd_stack *)(d + o_stacks);
                s->view.metas = (const struct tamd_meta *)(d + o_metas);
                s->view.layer_first = (const int *)(d + o_first);
                s->view.tiles = (const int *)(d + o_tiles);
                s->view.slot_nodes = (const uint16_t * const *)(d + o_nodes);
                s->view.fast_ok = fast_ok;
                s->view.n_layers = s->n_layers;
                s->view.geoid = geoid;
                s->view.mode = TAMD_MODE_GENERIC;
                if ((s->n_layers == 1) && (n_metas == 1) && (geoid < 0)) {
                        if ((h_metas[0].kind == TAMD_MAP) &&
                            (h_grids[h_metas[0].src].proj.type < 0))
                                s->view.mode = TAMD_MODE_ONE_MAP;
                        if (h_metas[0].kind == TAMD_STACK) s->view.mode = TAMD_MODE_ONE_STACK;
                }
                s->epoch = epoch_now;
        }
        free(host), free(gl.map), free(data_src), free(tiles), free(stacks);
        return dev_fail ? -1 : 0;
}

/* ---- batch calls over paged stacks (paging.c) --------------------------- */

static int stepper_is_paged(const struct turtle_stepper * s)
{
        int i;
        for (i = 0; i < s->n_data; i++)
                if ((s->data[i].kind == TAMD_STACK) && tamd_stack_is_paged(s->data[i].stack)) return 1;
        return 0;
}

/* the tiles a round wanted, stack by stack (the tile table lists the stacks in
 * the order of s->data): 0 if none could come in, -1 with `code` set on error */
static int stepper_page_in(struct turtle_stepper * s, const unsigned * wanted,
    const unsigned * wanted_first, int few, int * code, char * message, size_t size)
{
        int i, first = 0, loaded = 0;
        for (i = 0; i < s->n_data; i++) {
                if (s->data[i].kind != TAMD_STACK) continue;
                struct turtle_stack * st = s->data[i].stack;
                const int got = tamd_stack_page_in(st, wanted, wanted_first, first, few, message, size);
                if (got < 0) {
                        *code = -got;
                        return -1;
                }
                loaded += got;
                first += st->latitude_n * st->longitude_n;
        }
        return loaded;
}

/* Runs `launch` (the kernels of one round) until nothing is listed any more.
 * Returns an enum turtle_return; TURTLE_RETURN_LIBRARY_ERROR: device failure
 * (message empty) */
typedef int stepper_round_t(struct turtle_stepper * stepper, struct tamd_paging pg, int round,
    void * args);

static int stepper_rounds(struct turtle_stepper * stepper, long n, stepper_round_t * launch,
    void * args, char * message, size_t size)
{
        struct tamd_pager pager;
        memset(&pager, 0, sizeof(pager));
        message[0] = 0;
        int paged = 0; /* did this call bring tiles in? */
        int rc = tamd_stepper_flatten(stepper, message, size);
        if (rc != 0) return (rc < 0) ? TURTLE_RETURN_LIBRARY_ERROR : rc;
        if (stepper_is_paged(stepper) && tamd_pager_begin(&pager, n, stepper->n_table))
                return TURTLE_RETURN_LIBRARY_ERROR;
        static int trace_rounds = -1; /* TURTLE_AMD_PAGING_TRACE=1: the time of each round, to stderr */
        if (trace_rounds < 0) trace_rounds = (getenv("TURTLE_AMD_PAGING_TRACE") != NULL);
        for (;;) {
                struct tamd_paging pg;
                struct timespec t0, t1, t2, t3;
                if (trace_rounds) clock_gettime(CLOCK_MONOTONIC, &t0);
                /* tables and launches of a round: nothing they point at may go meanwhile */
                tamd_geometry_use_begin();
                rc = tamd_stepper_flatten(stepper, message, size);
                if (rc != 0)
                        rc = (rc < 0) ? TURTLE_RETURN_LIBRARY_ERROR : rc;
                else if (tamd_pager_round(&pager, &pg) || launch(stepper, pg, pager.rounds, args))
                        rc = TURTLE_RETURN_LIBRARY_ERROR;
                tamd_geometry_use_end();
                if (rc != 0) break;
                if (trace_rounds) clock_gettime(CLOCK_MONOTONIC, &t1);
                unsigned long long faulted = 0;
                if (tamd_pager_collect(&pager, &faulted)) {
                        rc = TURTLE_RETURN_LIBRARY_ERROR;
                        break;
                }
                if (trace_rounds) clock_gettime(CLOCK_MONOTONIC, &t2);
                if (faulted == 0) {
                        if (trace_rounds)
                                fprintf(stderr, "[paging] round %d: tables+launch %.2f ms, kernels %.2f ms, done\n", pager.rounds,
                                    1e3 * (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_nsec - t0.tv_nsec),
                                    1e3 * (t2.tv_sec - t1.tv_sec) + 1e-6 * (t2.tv_nsec - t1.tv_nsec));
                        break;
                }
                int code = 0;
                paged = 1;
                const int got = stepper_page_in(stepper, pager.wanted, pager.pinned,
                    faulted <= TAMD_PAGING_FEW, &code, message, size);
                if (trace_rounds) {
                        clock_gettime(CLOCK_MONOTONIC, &t3);
                        fprintf(stderr, "[paging] round %d: tables+launch %.2f ms, kernels %.2f ms, %llu items wait, %d tiles in %.2f ms (%lu from their buffers so far)\n",
                            pager.rounds, 1e3 * (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_nsec - t0.tv_nsec),
                            1e3 * (t2.tv_sec - t1.tv_sec) + 1e-6 * (t2.tv_nsec - t1.tv_nsec), faulted, got,
                            1e3 * (t3.tv_sec - t2.tv_sec) + 1e-6 * (t3.tv_nsec - t2.tv_nsec), tamd_stack_buffer_hits);
                }
                if (got < 0) {
                        rc = code;
                        break;
                }
                if (pager.rounds > TAMD_PAGING_ROUNDS) { /* cannot be: a round serves an item */
                        snprintf(message, size,
                            "the stacks of the stepper are too small (stack_size) for this batch");
                        rc = TURTLE_RETURN_MEMORY_ERROR;
                        break;
                }
        }
        tamd_pager_end(&pager);
        stepper->last_rounds = (pager.rounds > 0) ? pager.rounds : 1;
        if (paged) {
                /* whatever the call came to: the tiles this thread had brought in for a
                 * round that will not run are anybody's again, and the stacks go back to
                 * their sizes (a pin never outlives the call that set it) */
                int i;
                for (i = 0; i < stepper->n_data; i++)
                        if (stepper->data[i].kind == TAMD_STACK) tamd_stack_trim(stepper->data[i].stack);
        }
        return rc;
}

#define FLATTEN_OR_RETURN(stepper)                                             \
        do {                                                                   \
                char message_[4200];                                           \
                const int rc_ = tamd_stepper_flatten((stepper), message_, sizeof(message_)); \
                if (rc_ < 0) return TAMD_RAISE_DEVICE();                       \
                if (rc_ > 0) return TAMD_RAISE((enum turtle_return)rc_, "%s", message_); \
        } while (0)

/* ---- batch entry points --------------------------------------------------- */

struct position_args {
        long n;
        void *lat, *lon, *height;
        int layer;
        void *pos, *index;
};

static int position_round(struct turtle_stepper * stepper, struct tamd_paging pg, int round, void * p)
{
        struct position_args * a = p;
        (void)round;
        return tamd_k_position(stepper->view, a->n, a->lat, a->lon, a->height, a->layer, a->pos,
            a->index, pg);
}

enum turtle_return turtle_stepper_position_n(struct turtle_stepper * stepper, long n,
    const double * latitude, const double * longitude, const double * height,
    int layer_index, double * position, int * data_index, int space)
{
        TAMD_ERROR_INIT(&turtle_stepper_position_n);
        if ((layer_index < 0) || (layer_index >= stepper->n_layers))
                return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "no valid data");
        if ((position == NULL) || (data_index == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        struct tamd_stage st;
        struct position_args args = { n, NULL, NULL, NULL, layer_index, NULL, NULL };
        const size_t nb = (size_t)n * sizeof(double);
        if (tamd_stage_begin(&st, space, 6 * nb + n * sizeof(int)) ||
            tamd_stage_in(&st, latitude, nb, &args.lat) ||
            tamd_stage_in(&st, longitude, nb, &args.lon) ||
            tamd_stage_in(&st, height, nb, &args.height) ||
            tamd_stage_in(&st, position, 3 * nb, &args.pos) || /* untouched rows keep their value */
            tamd_stage_out(&st, data_index, n * sizeof(int), &args.index))
                return TAMD_RAISE_DEVICE();
        char message[4200];
        const int rc = stepper_rounds(stepper, n, &position_round, &args, message, sizeof(message));
        if (rc == TURTLE_RETURN_LIBRARY_ERROR) return TAMD_RAISE_DEVICE();
        if (rc != 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        if (tamd_stage_fetch(&st, position, 3 * nb, args.pos) ||
            tamd_stage_fetch(&st, data_index, n * sizeof(int), args.index) || tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

/* Grow-only scratch of the batch calls: 4 n ints (the rays a trace hands from
 * pass to pass, the rays whose step crossed a boundary and what they found
 * there / the rays a batch of steps defers to its bisection pass; the step
 * counts of a trace whose caller wants none) followed by 4 n doubles (the step a
 * ray that waits for a tile was about to take; the crossing steps; the path
 * lengths of a trace whose caller wants none; the step lengths a trace sorts its
 * hand-over by).  0 if it holds n entries; 1 if n
 * is beyond an int (the caller does without); parked_capacity < 0 after a device
 * failure. */
static int tamd_stepper_scratch(struct turtle_stepper * stepper, long n)
{
        if (n >= 2147483647L) return 1;
        if (stepper_bind_device(stepper)) {
                stepper->parked_capacity = -1;
                return 1;
        }
        if (n <= stepper->parked_capacity) return 0;
        if (stepper->d_parked != NULL) {
                tamd_dev_sync();
                tamd_dev_free(stepper->d_parked);
                stepper->d_parked = NULL;
        }
        /* (the lists of the passes, and room to order the hand-over: internal.h) */
        /* (+ 256: the sort's room begins at the next multiple of 256 bytes behind the ints) */
        const size_t ints = ((((size_t)n * TAMD_TRACE_SORT_INTS * sizeof(int) + 256 + TAMD_TRACE_SORT_TEMP +
                                  (size_t)n * TAMD_TRACE_COPY_BYTES) + 255) / 256) * 256;
        stepper->parked_capacity = 0;
        if (tamd_dev_malloc((void **)&stepper->d_parked, ints + (size_t)n * 4 * sizeof(double))) {
                stepper->parked_capacity = -1;
                return 1;
        }
        stepper->d_scratch_ds = (double *)((char *)stepper->d_parked + ints);
        stepper->parked_capacity = n;
        return 0;
}

struct step_args {
        long n;
        void *pos, *dir, *lat, *lon, *alt, *elev, *step, *index;
        int flags, listed;
};

static int step_round(struct turtle_stepper * stepper, struct tamd_paging pg, int round, void * p)
{
        struct step_args * a = p;
        (void)round;
        if (a->dir == NULL)
                return tamd_k_step(stepper->view, a->n, a->pos, a->dir, a->lat, a->lon, a->alt,
                    a->elev, a->step, a->index, a->flags, pg);
        return tamd_k_step_dir(stepper->view, a->n, a->pos, a->dir, a->lat, a->lon, a->alt, a->elev,
            a->step, a->index, a->flags, a->listed ? stepper->d_parked : NULL,
            a->listed ? stepper->d_scratch_ds : NULL, pg, stepper->d_stats, stepper->d_stats + 4);
}

static enum turtle_return step_n(struct tamd_error * error, struct turtle_stepper * stepper,
    long n, double * position, const double * direction, double * latitude,
    double * longitude, double * altitude, double * elevation, double * step, int * index,
    int flags, int space)
{
        struct tamd_error error_ = *error;
        if ((position == NULL) || (index == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        if ((flags & TURTLE_AMD_STEP_RESUME) && !(flags & TAMD_STEP_COMPACT) &&
            ((altitude == NULL) || (elevation == NULL)))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS,
                    "TURTLE_AMD_STEP_RESUME needs altitude, elevation and index");
        struct tamd_stage st;
        void *dp, *dd, *dla, *dlo, *dal, *del, *dst, *dix;
        const size_t nb = (size_t)n * sizeof(double);
        const int resume = (flags & TURTLE_AMD_STEP_RESUME) != 0;
        if (tamd_stage_begin(&st, space, 12 * nb + 2 * n * sizeof(int)) ||
            tamd_stage_in(&st, position, 3 * nb, &dp) ||
            tamd_stage_in(&st, direction, 3 * nb, &dd))
                return TAMD_RAISE_DEVICE();
        int bad = 0;
        if (resume) {
                bad |= tamd_stage_in(&st, latitude, nb, &dla);
                bad |= tamd_stage_in(&st, longitude, nb, &dlo);
                bad |= tamd_stage_in(&st, altitude, nb, &dal);
                bad |= tamd_stage_in(&st, elevation, 2 * nb, &del);
                bad |= tamd_stage_in(&st, index, 2 * n * sizeof(int), &dix);
        } else {
                bad |= tamd_stage_out(&st, latitude, nb, &dla);
                bad |= tamd_stage_out(&st, longitude, nb, &dlo);
                bad |= tamd_stage_out(&st, altitude, nb, &dal);
                bad |= tamd_stage_out(&st, elevation, 2 * nb, &del);
                bad |= tamd_stage_out(&st, index, 2 * n * sizeof(int), &dix);
        }
        bad |= tamd_stage_out(&st, step, nb, &dst);
        /* with a direction: two passes, the second for the rays that crossed a
         * boundary (a single step bisects in place: nothing to pack) */
        const int listed = (direction != NULL) && (n > 1) && (tamd_stepper_scratch(stepper, n) == 0);
        if ((direction != NULL) && (n > 1) && !listed && (stepper->parked_capacity < 0)) bad = 1;
        if (bad) return TAMD_RAISE_DEVICE();
        struct step_args args = { n, dp, dd, dla, dlo, dal, del, dst, dix, flags, listed };
        char message[4200];
        const int rc = stepper_rounds(stepper, n, &step_round, &args, message, sizeof(message));
        if (rc == TURTLE_RETURN_LIBRARY_ERROR) return TAMD_RAISE_DEVICE();
        if (rc != 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        if (((direction != NULL) && tamd_stage_fetch(&st, position, 3 * nb, dp)) ||
            tamd_stage_fetch(&st, latitude, nb, dla) ||
            tamd_stage_fetch(&st, longitude, nb, dlo) ||
            tamd_stage_fetch(&st, altitude, nb, dal) ||
            tamd_stage_fetch(&st, elevation, 2 * nb, del) ||
            tamd_stage_fetch(&st, step, nb, dst) ||
            tamd_stage_fetch(&st, index, 2 * n * sizeof(int), dix) || tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

/* ---- scattering walk ------------------------------------------------------ */

struct walk_args {
        long n;
        void *pos, *alt, *elev, *index, *length, *steps;
        unsigned long long seed, stream;
        long first;
};

static int walk_round(struct turtle_stepper * stepper, struct tamd_paging pg, int round, void * p)
{
        struct walk_args * a = p;
        (void)round;
        return tamd_k_step_walk(stepper->view, a->n, a->pos, a->alt, a->elev, a->index, a->seed,
            a->stream, a->first, a->length, a->steps, stepper->d_parked, stepper->d_scratch_ds, pg,
            stepper->d_stats, stepper->d_stats + 4);
}

static int walk_start_round(struct turtle_stepper * stepper, struct tamd_paging pg, int round, void * p)
{
        struct walk_args * a = p;
        (void)round;
        return tamd_k_step(stepper->view, a->n, a->pos, NULL, NULL, NULL, a->alt, a->elev, NULL,
            a->index, 0, pg);
}

enum turtle_return turtle_stepper_scatter_n(struct turtle_stepper * stepper, long n,
    double * position, unsigned long long seed, long first_ray, int first_step, int n_steps,
    double * altitude, double * elevation, int * index, double * length, int * steps, int flags,
    int space)
{
        TAMD_ERROR_INIT(&turtle_stepper_scatter_n);
        if ((position == NULL) || (altitude == NULL) || (elevation == NULL) || (index == NULL) ||
            (length == NULL) || (steps == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        if ((n_steps < 0) || (first_step < 0))
                return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "invalid input parameter(s)");
        if (n <= 0) return TURTLE_RETURN_SUCCESS;
        if (tamd_stepper_scratch(stepper, n) != 0) {
                if (stepper->parked_capacity < 0) return TAMD_RAISE_DEVICE();
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "batch too large");
        }
        struct tamd_stage st;
        struct walk_args a = { n, NULL, NULL, NULL, NULL, NULL, NULL, seed, 0, first_ray };
        const size_t nb = (size_t)n * sizeof(double);
        const int start = (flags & TURTLE_AMD_SCATTER_START) != 0;
        int bad = tamd_stage_begin(&st, space, 7 * nb + 3 * n * sizeof(int)) ||
            tamd_stage_in(&st, position, 3 * nb, &a.pos);
        if (!bad && start)
                bad = tamd_stage_out(&st, altitude, nb, &a.alt) ||
                    tamd_stage_out(&st, elevation, 2 * nb, &a.elev) ||
                    tamd_stage_out(&st, index, 2 * n * sizeof(int), &a.index) ||
                    tamd_stage_out(&st, length, nb, &a.length) ||
                    tamd_stage_out(&st, steps, n * sizeof(int), &a.steps);
        else if (!bad)
                bad = tamd_stage_in(&st, altitude, nb, &a.alt) ||
                    tamd_stage_in(&st, elevation, 2 * nb, &a.elev) ||
                    tamd_stage_in(&st, index, 2 * n * sizeof(int), &a.index) ||
                    tamd_stage_in(&st, length, nb, &a.length) ||
                    tamd_stage_in(&st, steps, n * sizeof(int), &a.steps);
        if (bad) return TAMD_RAISE_DEVICE();
        char message[4200];
        int rc = 0, k;
        if (start) {
                rc = stepper_rounds(stepper, n, &walk_start_round, &a, message, sizeof(message));
                if ((rc == 0) && (tamd_dev_zero(a.length, nb) || tamd_dev_zero(a.steps, n * sizeof(int))))
                        rc = TURTLE_RETURN_LIBRARY_ERROR;
        } else if (stepper->d_stats == NULL) {
                rc = tamd_stepper_flatten(stepper, message, sizeof(message));
                if (rc < 0) rc = TURTLE_RETURN_LIBRARY_ERROR;
        }
        /* the counters are those of THIS call (turtle_stepper_trace_stats), whatever the
         * stepper's last batch left there */
        if ((rc == 0) && tamd_dev_zero(stepper->d_stats, 4 * sizeof(*stepper->d_stats)))
                rc = TURTLE_RETURN_LIBRARY_ERROR;
        /* every tile resident: the whole walk in one launch, the rays' state in
         * registers (k_walk); else generation by generation, each in rounds over the
         * rays that wait for a tile (TURTLE_AMD_WALK=steps: that form always) */
        static int by_steps = -1;
        if (by_steps < 0) {
                const char * env = getenv("TURTLE_AMD_WALK");
                by_steps = ((env != NULL) && (strcmp(env, "steps") == 0)) ? 1 : 0;
        }
        if ((rc == 0) && !by_steps && !stepper_is_paged(stepper) && (n_steps > 0)) {
                tamd_geometry_use_begin();
                rc = tamd_stepper_flatten(stepper, message, sizeof(message));
                if (rc < 0) rc = TURTLE_RETURN_LIBRARY_ERROR;
                if ((rc == 0) && tamd_k_walk(stepper->view, n, a.pos, a.alt, a.elev, a.index, seed,
                        first_ray, first_step, n_steps, a.length, a.steps, stepper->d_stats,
                        stepper->d_stats + 4))
                        rc = TURTLE_RETURN_LIBRARY_ERROR;
                tamd_geometry_use_end();
                n_steps = 0;
        }
        for (k = 0; (rc == 0) && (k < n_steps); k++) {
                a.stream = (unsigned long long)(first_step + k);
                rc = stepper_rounds(stepper, n, &walk_round, &a, message, sizeof(message));
        }
        if (rc == TURTLE_RETURN_LIBRARY_ERROR) return TAMD_RAISE_DEVICE();
        if (rc != 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        if (tamd_stage_fetch(&st, position, 3 * nb, a.pos) || tamd_stage_fetch(&st, altitude, nb, a.alt) ||
            tamd_stage_fetch(&st, elevation, 2 * nb, a.elev) ||
            tamd_stage_fetch(&st, index, 2 * n * sizeof(int), a.index) ||
            tamd_stage_fetch(&st, length, nb, a.length) ||
            tamd_stage_fetch(&st, steps, n * sizeof(int), a.steps) || tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

/* A walk, step by step, with the least state between the calls: `next` (the tentative length of
 * the next step) and `index` are all a step resumes from [ref stepper.c:708-710, :799-813]. */
enum turtle_return turtle_stepper_walk_n(struct turtle_stepper * stepper, long n, double * position,
    const double * direction, double * next, double * step, int * index, int space)
{
        TAMD_ERROR_INIT(&turtle_stepper_walk_n);
        if (next == NULL) return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        return step_n(&error_, stepper, n, position, direction, NULL, NULL, next, NULL, step, index,
            TAMD_STEP_COMPACT | ((direction != NULL) ? TURTLE_AMD_STEP_RESUME : 0), space);
}

enum turtle_return turtle_stepper_step_n(struct turtle_stepper * stepper, long n,
    double * position, const double * direction, double * latitude, double * longitude,
    double * altitude, double * elevation, double * step, int * index, int flags,
    int space)
{
        TAMD_ERROR_INIT(&turtle_stepper_step_n);
        return step_n(&error_, stepper, n, position, direction, latitude, longitude,
            altitude, elevation, step, index, flags, space);
}

struct trace_args {
        long n;
        void *pos, *dir, *index, *length, *n_steps;
        int max_steps, flags, scratch;
};

static int trace_round(struct turtle_stepper * stepper, struct tamd_paging pg, int round, void * p)
{
        struct trace_args * a = p;
        (void)round;
        pg.tentative = a->scratch ? stepper->d_scratch_ds : NULL;
        /* the lists of the passes; the packed media of the crossings hold 16 bits each */
        const int listed = a->scratch && (stepper->n_layers < 65000) && (stepper->n_data < 65000);
        return tamd_k_trace(stepper->view, a->n, a->pos, a->dir, a->max_steps, a->index, a->length,
            a->n_steps, a->flags | (listed ? TAMD_TRACE_SORT_ROOM : 0), listed ? stepper->d_parked : NULL,
            listed ? stepper->d_scratch_ds + a->n : NULL, pg, stepper->d_stats, stepper->d_stats + 4);
}

enum turtle_return turtle_stepper_trace_n(struct turtle_stepper * stepper, long n,
    double * position, const double * direction, int max_steps, int * index,
    double * length, int * n_steps, int flags, int space)
{
        TAMD_ERROR_INIT(&turtle_stepper_trace_n);
        if ((position == NULL) || (direction == NULL) || (index == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        struct trace_args args = { n, NULL, NULL, NULL, NULL, NULL, max_steps, flags, 0 };
        args.scratch = (tamd_stepper_scratch(stepper, n) == 0);
        if (!args.scratch && (stepper->parked_capacity < 0)) return TAMD_RAISE_DEVICE();
        struct tamd_stage st;
        const size_t nb = (size_t)n * sizeof(double);
        if (tamd_stage_begin(&st, space, 7 * nb + 3 * n * sizeof(int)) ||
            tamd_stage_in(&st, position, 3 * nb, &args.pos) ||
            tamd_stage_in(&st, direction, 3 * nb, &args.dir) ||
            ((flags & TURTLE_AMD_TRACE_RESUME) ?
                    tamd_stage_in(&st, index, 2 * n * sizeof(int), &args.index) :
                    tamd_stage_out(&st, index, 2 * n * sizeof(int), &args.index)) ||
            tamd_stage_out(&st, length, nb, &args.length) ||
            tamd_stage_out(&st, n_steps, n * sizeof(int), &args.n_steps))
                return TAMD_RAISE_DEVICE();
        /* a ray that waits for a tile keeps its path length and step count in
         * these arrays, and so does a ray handed from pass to pass: they exist
         * even if the caller has none -- which outputs a caller asks for changes
         * neither the kernels that run nor a bit of the others */
        if (stepper_is_paged(stepper) && !args.scratch)
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR,
                    "a batch this large cannot run over stacks with tiles left to page in");
        if (args.scratch && (n > 0)) {
                if (args.length == NULL) args.length = stepper->d_scratch_ds + 2 * n;
                if (args.n_steps == NULL) args.n_steps = stepper->d_parked + 3 * n;
        }
        char message[4200];
        const int rc = stepper_rounds(stepper, n, &trace_round, &args, message, sizeof(message));
        if (rc == TURTLE_RETURN_LIBRARY_ERROR) return TAMD_RAISE_DEVICE();
        if (rc != 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        if (tamd_stage_fetch(&st, position, 3 * nb, args.pos) ||
            tamd_stage_fetch(&st, index, 2 * n * sizeof(int), args.index) ||
            tamd_stage_fetch(&st, length, nb, args.length) ||
            tamd_stage_fetch(&st, n_steps, n * sizeof(int), args.n_steps) || tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

int turtle_amd_stepper_rounds(const struct turtle_stepper * stepper) { return stepper->last_rounds; }

enum turtle_return turtle_stepper_trace_stats(
    struct turtle_stepper * stepper, unsigned long long stats[4])
{
        TAMD_ERROR_INIT(&turtle_stepper_trace_stats);
        memset(stats, 0, 4 * sizeof(*stats));
        if (stepper->d_stats == NULL) return TURTLE_RETURN_SUCCESS;
        if (tamd_dev_d2h(stats, stepper->d_stats, 4 * sizeof(*stats)))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

/* ---- scalar entry points: the batch ones with n = 1 ------------------------ */

/* [ref stepper.c:780-875].  Outside every data with index == NULL is the one
 * case the reference raises [ref stepper.c:751-754, :870-873]. */
enum turtle_return turtle_stepper_step(struct turtle_stepper * stepper, double * position,
    const double * direction, double * latitude, double * longitude, double * altitude,
    double * elevation, double * step, int * index)
{
        TAMD_ERROR_INIT(&turtle_stepper_step);
        if (tamd_scalar_on_host() && tamd_h_stepper_takes(stepper)) { /* (the caller's option: scalar.c) */
                char message[4200];
                const int rc = tamd_h_stepper_step(stepper, position, direction, latitude, longitude,
                    altitude, elevation, step, index, message, sizeof(message));
                if (rc == TURTLE_RETURN_DOMAIN_ERROR) return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "no valid data");
                if (rc != 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
                return TURTLE_RETURN_SUCCESS;
        }
        int idx[2] = { -1, -1 };
        const enum turtle_return rc = step_n(&error_, stepper, 1, position, direction,
            latitude, longitude, altitude, elevation, step, idx, 0, TURTLE_AMD_HOST);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        if (index != NULL) {
                index[0] = idx[0], index[1] = idx[1];
        } else if (idx[0] < 0)
                return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "no valid data");
        return TURTLE_RETURN_SUCCESS;
}

/* [ref stepper.c:877-931] */
enum turtle_return turtle_stepper_position(struct turtle_stepper * stepper, double latitude,
    double longitude, double height, int layer_index, double * position, int * data_index)
{
        TAMD_ERROR_INIT(&turtle_stepper_position);
        if ((layer_index < 0) || (layer_index >= stepper->n_layers))
                return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "no valid data");
        int di = -1;
        if (tamd_scalar_on_host() && tamd_h_stepper_takes(stepper)) { /* (the caller's option: scalar.c) */
                char message[4200];
                const int rc = tamd_h_stepper_position(stepper, latitude, longitude, height, layer_index,
                    position, &di, message, sizeof(message));
                if (rc != 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
                if (data_index != NULL)
                        *data_index = di;
                else if (di < 0)
                        return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "no valid data");
                return TURTLE_RETURN_SUCCESS;
        }
        const enum turtle_return rc = turtle_stepper_position_n(stepper, 1, &latitude,
            &longitude, &height, layer_index, position, &di, TURTLE_AMD_HOST);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        if (data_index != NULL)
                *data_index = di;
        else if (di < 0)
                return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "no valid data");
        return TURTLE_RETURN_SUCCESS;
}
