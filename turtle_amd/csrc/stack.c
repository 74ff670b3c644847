/*
 * stack.c -- a directory of equally sized tiles covering a lat/lon lattice
 * [ref src/turtle/stack.c:46-450].
 *
 * Host side: scan the directory, build the slot -> file table, load tiles.
 * Device side: the table becomes a flat tile directory (struct tamd_stack) and
 * every loaded tile stays resident in HBM, so a lookup is O(1) instead of the
 * reference's MRU list walk [ref stack.c:300-335] and never touches the disk
 * while stepping.  `stack_size` (max_size) is kept for API compatibility; with
 * 288 GB of HBM there is nothing to evict for the configurations in scope.
 */
#include "host.h"

#include <dirent.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

static int is_tile_file(const char * name)
{
        int (*probe)(const char *, struct turtle_map *);
        int (*read)(const char *, struct turtle_map *);
        return tamd_codec_for(name, &probe, &read);
}

static int tile_probe(const char * path, struct turtle_map * meta)
{
        int (*probe)(const char *, struct turtle_map *);
        int (*read)(const char *, struct turtle_map *);
        if (!tamd_codec_for(path, &probe, &read)) return TURTLE_RETURN_BAD_EXTENSION;
        return probe(path, meta);
}


/* [ref stack.c:46-213] */
enum turtle_return turtle_stack_create(struct turtle_stack ** stack,
    const char * path, int size, turtle_stack_locker_t * lock,
    turtle_stack_locker_t * unlock)
{
        TAMD_ERROR_INIT(&turtle_stack_create);
        *stack = NULL;
        if (((lock == NULL) && (unlock != NULL)) || ((unlock == NULL) && (lock != NULL)))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "inconsistent lock & unlock");

        DIR * dir = opendir(path);
        if (dir == NULL)
                return TAMD_RAISE(TURTLE_RETURN_PATH_ERROR, "could not access %s", path);

        /* first pass: lattice extents and tile span [ref stack.c:60-124] */
        double lat_min = DBL_MAX, long_min = DBL_MAX;
        double lat_max = -DBL_MAX, long_max = -DBL_MAX;
        double lat_delta = 0., long_delta = 0.;
        struct dirent * entry;
        char file[4096];
        while ((entry = readdir(dir)) != NULL) {
                snprintf(file, sizeof(file), "%s/%s", path, entry->d_name);
                struct stat sb;
                if ((stat(file, &sb) != 0) || S_ISDIR(sb.st_mode)) continue;
                if (!is_tile_file(entry->d_name)) continue; /* unknown format */
                struct turtle_map meta;
                const int rc = tile_probe(file, &meta);
                if (rc != TURTLE_RETURN_SUCCESS) {
                        closedir(dir);
                        return TAMD_RAISE((enum turtle_return)rc,
                            (rc == TURTLE_RETURN_PATH_ERROR) ?
                                "could not open file `%s'" :
                                "invalid file name or layout for `%s'",
                            file);
                }
                const double dx = meta.dx * (meta.nx - 1);
                const double dy = meta.dy * (meta.ny - 1);
                if (long_delta == 0.)
                        long_delta = dx;
                else if (long_delta != dx) {
                        closedir(dir);
                        return TAMD_RAISE(
                            TURTLE_RETURN_BAD_FORMAT, "inconsistent longitude span");
                }
                if (lat_delta == 0.)
                        lat_delta = dy;
                else if (lat_delta != dy) {
                        closedir(dir);
                        return TAMD_RAISE(
                            TURTLE_RETURN_BAD_FORMAT, "inconsistent latitude span");
                }
                if (meta.x0 < long_min) long_min = meta.x0;
                if (meta.y0 < lat_min) lat_min = meta.y0;
                if (meta.x0 + dx > long_max) long_max = meta.x0 + dx;
                if (meta.y0 + dy > lat_max) lat_max = meta.y0 + dy;
        }

        int lat_n = 0, long_n = 0; /* [ref stack.c:127-140] */
        if ((lat_delta > 0.) && (long_delta > 0.)) {
                const double dx = (long_max - long_min) / long_delta;
                long_n = (int)(dx + FLT_EPSILON);
                const double dy = (lat_max - lat_min) / lat_delta;
                lat_n = (int)(dy + FLT_EPSILON);
                if ((fabs(long_n - dx) > FLT_EPSILON) || (fabs(lat_n - dy) > FLT_EPSILON)) {
                        closedir(dir);
                        return TAMD_RAISE(TURTLE_RETURN_BAD_FORMAT,
                            (fabs(long_n - dx) > FLT_EPSILON) ? "invalid longitude grid" :
                                                                "invalid latitude grid");
                }
        }

        struct turtle_stack * s = calloc(1, sizeof(*s));
        const size_t slots = (size_t)lat_n * long_n;
        if (s != NULL) {
                s->root = strdup(path);
                s->path = calloc(slots ? slots : 1, sizeof(*s->path));
                s->tile = calloc(slots ? slots : 1, sizeof(*s->tile));
                s->stamp = calloc(slots ? slots : 1, sizeof(*s->stamp));
                s->owner = calloc(slots ? slots : 1, sizeof(*s->owner));
                int k;
                for (k = 0; k < TAMD_STAGE_SLOTS; k++) s->stage_device[k] = -1, s->stage_tile[k] = -1; /* free, empty */
        }
        if ((s == NULL) || (s->root == NULL) || (s->path == NULL) || (s->tile == NULL) ||
            (s->stamp == NULL) || (s->owner == NULL)) {
                closedir(dir);
                if (s != NULL) {
                        free(s->root), free(s->path), free(s->tile), free(s->stamp), free(s->owner);
                        free(s);
                }
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        }
        s->max_size = (size > 0) ? size : INT_MAX;
        s->lock = lock, s->unlock = unlock;
        s->latitude_0 = lat_min, s->longitude_0 = long_min;
        s->latitude_delta = lat_delta, s->longitude_delta = long_delta;
        s->latitude_n = lat_n, s->longitude_n = long_n;

        /* second pass: slot -> file [ref stack.c:167-198] */
        rewinddir(dir);
        while ((slots > 0) && ((entry = readdir(dir)) != NULL)) {
                snprintf(file, sizeof(file), "%s/%s", path, entry->d_name);
                struct stat sb;
                if ((stat(file, &sb) != 0) || S_ISDIR(sb.st_mode)) continue;
                if (!is_tile_file(entry->d_name)) continue;
                struct turtle_map meta;
                if (tile_probe(file, &meta) != TURTLE_RETURN_SUCCESS) continue;
                const int ix = (int)((meta.x0 - long_min) / long_delta);
                const int iy = (int)((meta.y0 - lat_min) / lat_delta);
                const size_t i = (size_t)iy * long_n + ix;
                if (s->path[i] == NULL) s->n_files++;
                free(s->path[i]);
                s->path[i] = strdup(file);
        }
        closedir(dir);
        *stack = s;
        return TURTLE_RETURN_SUCCESS;
}

/* ---- staging buffers and spare HBM buffers (host.h) ------------------------ */

unsigned long tamd_stack_buffer_hits = 0;

/* size and time stamp of a file (what a staging buffer's contents are good for) */
static int file_identity(const char * path, long long * size, long long * time)
{
        struct stat st;
        if (stat(path, &st) != 0) return -1;
        *size = (long long)st.st_size;
        *time = (long long)st.st_mtim.tv_sec * 1000000000ll + (long long)st.st_mtim.tv_nsec;
        return 0;
}

/* A staging buffer of at least `bytes` for tile `tile` of the directory: its slot, or -1
 * (the tile then takes the slow way: laid out and copied when it is first needed).
 * *cached: the buffer holds that tile's nodes already (host.h, stage_tile).  Else a
 * free buffer: one that holds nothing, or the one whose contents are oldest. */
static int stage_acquire(struct turtle_stack * s, size_t bytes, int tile, int * cached)
{
        int k, slot = -1;
        *cached = 0;
        if ((s->stage_bytes != 0) && (bytes > s->stage_bytes)) return -1; /* (one shape a stack) */
        long long size = -1, time = -1;
        const int known = (tile >= 0) && (file_identity(s->path[tile], &size, &time) == 0);
        for (k = 0; known && (k < TAMD_STAGE_SLOTS); k++) {
                if ((s->stage_tile[k] != tile) || (s->stage[k] == NULL) || (s->stage_device[k] == -2)) continue;
                if ((s->stage_file_size[k] != size) || (s->stage_file_time[k] != time)) {
                        s->stage_tile[k] = -1; /* the file has changed: read it */
                        continue;
                }
                if ((s->stage_device[k] >= 0) && tamd_dev_sync_device(s->stage_device[k])) return -1;
                s->stage_device[k] = -2;
                s->stage_stamp[k] = ++s->clock;
                *cached = 1;
                __atomic_add_fetch(&tamd_stack_buffer_hits, 1, __ATOMIC_RELAXED);
                return k;
        }
        for (k = 0; k < TAMD_STAGE_SLOTS; k++) {
                if (s->stage_device[k] != -1) continue;
                if ((slot < 0) || ((s->stage_tile[k] < 0) && (s->stage_tile[slot] >= 0)) ||
                    (((s->stage_tile[k] < 0) == (s->stage_tile[slot] < 0)) && (s->stage_stamp[k] < s->stage_stamp[slot])))
                        slot = k;
        }
        for (k = 0; (k < TAMD_STAGE_SLOTS) && (slot < 0); k++) {
                if (s->stage_device[k] < 0) continue;
                /* copied from, on that device: free once it has drained */
                if (tamd_dev_sync_device(s->stage_device[k])) return -1;
                int j;
                const int device = s->stage_device[k];
                for (j = 0; j < TAMD_STAGE_SLOTS; j++)
                        if (s->stage_device[j] == device) s->stage_device[j] = -1;
                slot = k;
        }
        if (slot < 0) return -1;
        if (s->stage[slot] == NULL) {
                void * p = NULL;
                if (tamd_dev_host_alloc(&p, bytes)) return -1;
                s->stage[slot] = p;
                s->stage_bytes = bytes;
        }
        s->stage_device[slot] = -2;
        s->stage_tile[slot] = known ? tile : -1;
        s->stage_file_size[slot] = size, s->stage_file_time[slot] = time;
        s->stage_stamp[slot] = ++s->clock;
        return slot;
}

static void stage_release_all(struct turtle_stack * s)
{
        int k;
        for (k = 0; k < TAMD_STAGE_SLOTS; k++) {
                if (s->stage_device[k] >= 0) (void)tamd_dev_sync_device(s->stage_device[k]);
                tamd_dev_host_free(s->stage[k]);
                s->stage[k] = NULL, s->stage_device[k] = -1, s->stage_tile[k] = -1;
        }
        s->stage_bytes = 0;
}

/* the HBM buffers of a tile that goes are kept for the next to come (its device has
 * drained: the caller holds the geometry for writing and has waited) */
static void spare_keep(struct turtle_stack * s, struct turtle_map * m)
{
        int d;
        const size_t bytes = tamd_blocked_bytes(m->nx, m->ny);
        if (s->spare_bytes == 0) s->spare_bytes = bytes;
        for (d = 0; d < TAMD_MAX_DEVICES; d++) {
                if ((m->d_nodes[d] == NULL) || (bytes != s->spare_bytes) ||
                    (s->n_spare[d] >= TAMD_SPARE_HBM))
                        continue;
                if (tamd_dev_sync_device(d)) continue; /* (it is freed the usual way) */
                s->spare[d][s->n_spare[d]++] = m->d_nodes[d];
                m->d_nodes[d] = NULL;
                m->d_fresh &= ~(1u << d);
        }
}

void * tamd_stack_spare_take(struct turtle_stack * s, int device, size_t bytes)
{
        if ((s == NULL) || (bytes != s->spare_bytes) || (s->n_spare[device] == 0)) return NULL;
        return s->spare[device][--s->n_spare[device]];
}

static void spare_release_all(struct turtle_stack * s)
{
        int d, k;
        for (d = 0; d < TAMD_MAX_DEVICES; d++) {
                for (k = 0; k < s->n_spare[d]; k++) tamd_dev_free_on(d, s->spare[d][k]);
                s->n_spare[d] = 0;
        }
        s->spare_bytes = 0;
}

/* a tile's staging buffer has been copied from (tamd_map_sync), or the tile goes */
void tamd_stack_staged_done(struct turtle_map * m, int device)
{
        if ((m->stack != NULL) && (m->staged_slot >= 0) && (m->staged_slot < TAMD_STAGE_SLOTS))
                m->stack->stage_device[m->staged_slot] = (device >= 0) ? device : -1;
        m->staged = NULL, m->staged_slot = -1;
}

static void stack_release_tiles(struct turtle_stack * s)
{
        const int n = s->latitude_n * s->longitude_n;
        int i;
        tamd_geometry_write_begin();
        for (i = 0; i < n; i++) {
                if (s->tile[i] == NULL) continue;
                struct turtle_map * m = s->tile[i];
                tamd_stack_staged_done(m, -1);
                m->stack = NULL; /* do not walk back into the table */
                turtle_map_destroy(&m);
                s->tile[i] = NULL, s->owner[i] = NULL;
        }
        s->n_loaded = 0;
        stage_release_all(s);
        spare_release_all(s);
        tamd_geometry_changed();
        tamd_geometry_write_end();
}

/* [ref stack.c:228-237] */
void turtle_stack_destroy(struct turtle_stack ** stack)
{
        if ((stack == NULL) || (*stack == NULL)) return;
        struct turtle_stack * s = *stack;
        stack_release_tiles(s);
        const int n = s->latitude_n * s->longitude_n;
        int i;
        for (i = 0; i < n; i++) free(s->path[i]);
        free(s->path), free(s->tile), free(s->stamp), free(s->owner), free(s->root);
        free(s);
        *stack = NULL;
}

/* [ref stack.c:240-254] */
enum turtle_return turtle_stack_clear(struct turtle_stack * stack)
{
        TAMD_ERROR_INIT(&turtle_stack_clear);
        if ((stack->lock != NULL) && (stack->lock() != 0))
                return TAMD_RAISE(TURTLE_RETURN_LOCK_ERROR, "could not acquire the lock");
        stack_release_tiles(stack);
        if ((stack->unlock != NULL) && (stack->unlock() != 0))
                return TAMD_RAISE(TURTLE_RETURN_UNLOCK_ERROR, "could not release the lock");
        return TURTLE_RETURN_SUCCESS;
}

/* Tiles the stack keeps in memory between calls [ref stack.c:150]: what the
 * caller asked for.  While a batch runs the tiles its first waiting item needs
 * stay whatever their number (a lookup near a seam consults the boxes of the
 * 3 x 3 tiles around it, a bisection can need two such neighbourhoods: up to 16)
 * -- as the reference's stack exceeds its size by the tiles its clients have
 * pinned [ref stack.c:433-442: only unpinned tiles go] -- and the stack is
 * trimmed back when the call ends (tamd_stack_trim). */
int tamd_stack_budget(const struct turtle_stack * s)
{
        return (s->max_size <= 0) ? INT_MAX : s->max_size;
}

/* what names the calling thread in `owner` */
static __thread char t_me;

/* the least recently wanted tiles go until the stack is within its size (the call of
 * this thread is over: what it had paged in for its next round is anybody's) */
void tamd_stack_trim(struct turtle_stack * s)
{
        const int n = s->latitude_n * s->longitude_n, budget = tamd_stack_budget(s);
        int i;
        if (s->lock != NULL) (void)s->lock();
        tamd_geometry_write_begin();
        for (i = 0; i < n; i++)
                if (s->owner[i] == &t_me) s->owner[i] = NULL;
        while (s->n_loaded > budget) {
                int out = -1;
                for (i = 0; i < n; i++) {
                        if ((s->tile[i] == NULL) || (s->owner[i] != NULL)) continue;
                        if ((out < 0) || (s->stamp[i] < s->stamp[out])) out = i;
                }
                if (out < 0) break;
                struct turtle_map * m = s->tile[out];
                tamd_stack_staged_done(m, -1);
                spare_keep(s, m);
                m->stack = NULL; /* do not walk back into the table */
                turtle_map_destroy(&m);
                s->tile[out] = NULL;
                s->n_loaded--;
                tamd_geometry_changed();
        }
        tamd_geometry_write_end();
        if (s->unlock != NULL) (void)s->unlock();
}

int turtle_amd_stack_resident(const struct turtle_stack * stack) { return stack->n_loaded; }

/* Can a lookup meet a tile that has a file and is not in memory?  Now -- or, with
 * other threads at work on the stack, by the time its kernel runs: a stack that
 * may not keep all its files can lose a tile to another thread's call at any
 * moment, so its batches always run with the paging bookkeeping in place. */
int tamd_stack_is_paged(const struct turtle_stack * s)
{
        return (s->n_loaded < s->n_files) || (tamd_stack_budget(s) < s->n_files);
}

/* `count` tiles from their files into memory, side by side (tiles.c), each laid out
 * in a staging buffer where one is free; they go on to HBM at the next device call.
 * All of them or none: an enum turtle_return. */
static int stack_load_tiles(struct turtle_stack * s, const int * which, int count, char * message,
    size_t size)
{
        if (count <= 0) return TURTLE_RETURN_SUCCESS;
        struct tamd_tile_job * jobs = calloc((size_t)count, sizeof(*jobs));
        int * slot = malloc((size_t)count * sizeof(*slot));
        if ((jobs == NULL) || (slot == NULL)) {
                free(jobs), free(slot);
                snprintf(message, size, "could not allocate memory");
                return TURTLE_RETURN_MEMORY_ERROR;
        }
        int k, rc = TURTLE_RETURN_SUCCESS;
        struct turtle_map meta;
        size_t bytes = s->stage_bytes;
        if ((bytes == 0) && (tile_probe(s->path[which[0]], &meta) == TURTLE_RETURN_SUCCESS))
                bytes = tamd_blocked_bytes(meta.nx, meta.ny);
        for (k = 0; k < count; k++) {
                jobs[k].path = s->path[which[k]];
                slot[k] = (bytes > 0) ? stage_acquire(s, bytes, which[k], &jobs[k].cached) : -1;
                jobs[k].staged = (slot[k] >= 0) ? s->stage[slot[k]] : NULL;
                jobs[k].staged_bytes = s->stage_bytes;
        }
        tamd_tiles_decode(jobs, count);
        for (k = 0; k < count; k++) {
                if ((jobs[k].rc != TURTLE_RETURN_SUCCESS) && (rc == TURTLE_RETURN_SUCCESS)) {
                        rc = jobs[k].rc;
                        snprintf(message, size, "could not load tile `%s'", jobs[k].path);
                }
        }
        for (k = 0; k < count; k++) {
                struct turtle_map * m = jobs[k].map;
                if ((rc != TURTLE_RETURN_SUCCESS) || (m == NULL) || (m->staged == NULL)) {
                        if (slot[k] >= 0) s->stage_device[slot[k]] = -1, s->stage_tile[slot[k]] = -1;
                        if (m != NULL) m->staged = NULL, m->staged_slot = -1;
                } else
                        m->staged_slot = slot[k];
                if (m == NULL) continue;
                if (rc != TURTLE_RETURN_SUCCESS) {
                        free(m->nodes), free(m);
                        continue;
                }
                m->stack = s;
                s->tile[which[k]] = m;
                s->stamp[which[k]] = ++s->clock;
                s->n_loaded++;
        }
        if (rc == TURTLE_RETURN_SUCCESS) tamd_geometry_changed();
        free(jobs), free(slot);
        return rc;
}

static void stack_drop_tile(struct turtle_stack * s, int i)
{
        struct turtle_map * m = s->tile[i];
        s->owner[i] = NULL;
        tamd_stack_staged_done(m, -1);
        spare_keep(s, m);
        m->stack = NULL; /* do not walk back into the table */
        turtle_map_destroy(&m);
        s->tile[i] = NULL;
        s->n_loaded--;
        tamd_geometry_changed();
}

int tamd_stack_preload(struct turtle_stack * s, char * message, size_t size)
{
        const int n = s->latitude_n * s->longitude_n, budget = tamd_stack_budget(s);
        int i, rc = TURTLE_RETURN_SUCCESS, count = 0;
        int * which = malloc((size_t)(n ? n : 1) * sizeof(*which));
        if (which == NULL) return TURTLE_RETURN_MEMORY_ERROR;
        tamd_geometry_lock();
        for (i = 0; (i < n) && (s->n_loaded + count < budget); i++)
                if ((s->path[i] != NULL) && (s->tile[i] == NULL)) which[count++] = i;
        /* (a batch of tiles at a time: each holds a staging buffer until it is uploaded) */
        for (i = 0; (i < count) && (rc == TURTLE_RETURN_SUCCESS); i += TAMD_STAGE_SLOTS)
                rc = stack_load_tiles(s, which + i, (count - i < TAMD_STAGE_SLOTS) ? count - i : TAMD_STAGE_SLOTS,
                    message, size);
        tamd_geometry_unlock();
        free(which);
        return rc;
}

static int stack_page_in(struct turtle_stack * s, const unsigned * wanted,
    const unsigned * wanted_first, int first_bit, int few, char * message, size_t size);

/* Tiles come and go under the stack's lock, when it has one [ref client.c:126-
 * 188: the reference's clients take it around every change of tile] */
int tamd_stack_page_in(struct turtle_stack * s, const unsigned * wanted,
    const unsigned * wanted_first, int first_bit, int few, char * message, size_t size)
{
        if ((s->lock != NULL) && (s->lock() != 0)) {
                snprintf(message, size, "could not acquire the lock");
                return -TURTLE_RETURN_LOCK_ERROR;
        }
        tamd_geometry_write_begin();
        int rc = stack_page_in(s, wanted, wanted_first, first_bit, few, message, size);
        tamd_geometry_write_end();
        if ((s->unlock != NULL) && (s->unlock() != 0) && (rc >= 0)) {
                snprintf(message, size, "could not release the lock");
                rc = -TURTLE_RETURN_UNLOCK_ERROR;
        }
        return rc;
}

static int stack_page_in(struct turtle_stack * s, const unsigned * wanted,
    const unsigned * wanted_first, int first_bit, int few, char * message, size_t size)
{
        const int n = s->latitude_n * s->longitude_n;
        /* The last few items of a batch (rays that go from tile to tile along the seams: a
         * few hundred of C3's ten million) would each take a round per tile, and a round costs
         * its tiles whatever the number of items it serves.  They are served as the reference
         * serves its clients, who each keep the tile they are on beyond the stack's size [ref
         * stack.c:433-442: only unpinned tiles go]: their tiles come in beyond the size -- by
         * at most that size again, or TAMD_PAGING_SLACK tiles -- and the stack is trimmed back
         * when the call ends (tamd_stack_trim), as it is for the tiles of the first item. */
        int budget = tamd_stack_budget(s);
        if (few && (budget < INT_MAX - TAMD_PAGING_SLACK))
                budget += (budget < TAMD_PAGING_SLACK) ? budget : TAMD_PAGING_SLACK;
        int i, loaded = 0;
#define FIRST(i) ((wanted_first[((i) + first_bit) >> 5] >> (((i) + first_bit) & 31)) & 1u)
#define DEMAND(i) (wanted[(size_t)((i) + first_bit) * TAMD_DEMAND_STRIDE])
        /* the resident tiles this round wanted are the most recently used; the ones this
         * thread brought in last time have had their round */
        for (i = 0; i < n; i++) {
                if (DEMAND(i) && (s->tile[i] != NULL)) s->stamp[i] = ++s->clock;
                if (s->owner[i] == &t_me) s->owner[i] = NULL;
        }
        /* First WHO comes and who goes, decided one tile after the other as the reference
         * would meet them -- then the tiles that go, go, and those that come are read side
         * by side (stack_load_tiles). */
        char * here = malloc((size_t)(n ? n : 1));  /* resident, as the plan proceeds */
        int * come = malloc((size_t)(n ? n : 1) * sizeof(*come));
        if ((here == NULL) || (come == NULL)) {
                free(here), free(come);
                snprintf(message, size, "could not allocate memory");
                return -TURTLE_RETURN_MEMORY_ERROR;
        }
        int resident = s->n_loaded;
        for (i = 0; i < n; i++) here[i] = (s->tile[i] != NULL);
        for (;;) {
                /* the next tile to bring in: one of the first item's, else the one
                 * in most demand */
                int want = -1, first = 0;
                for (i = 0; i < n; i++) {
                        if (here[i] || (s->path[i] == NULL) || !DEMAND(i)) continue;
                        if (FIRST(i)) {
                                want = i, first = 1;
                                break;
                        }
                        if ((want < 0) || (DEMAND(i) > DEMAND(want))) want = i;
                }
                if (want < 0) break;
                if (resident >= budget) {
                        /* who goes: the tile in least demand, the least recently
                         * wanted of those; never one of the first item's, nor one
                         * that this round brings in */
                        int out = -1;
                        for (i = 0; i < n; i++) {
                                if (!here[i] || (s->tile[i] == NULL) || FIRST(i) || (s->owner[i] != NULL)) continue;
                                if ((out < 0) || (DEMAND(i) < DEMAND(out)) ||
                                    ((DEMAND(i) == DEMAND(out)) && (s->stamp[i] < s->stamp[out])))
                                        out = i;
                        }
                        /* ... and only for a tile in more demand; the first item's come in
                         * whatever has to go -- or nothing, if all that is in memory is
                         * its own (the stack is trimmed when the call ends) */
                        if (!first && ((out < 0) || (DEMAND(out) >= DEMAND(want)))) break;
                        if (out >= 0) here[out] = 0, resident--;
                }
                here[want] = 1, resident++;
                come[loaded++] = want;
        }
        for (i = 0; i < n; i++)
                if (!here[i] && (s->tile[i] != NULL)) stack_drop_tile(s, i);
        int rc = TURTLE_RETURN_SUCCESS;
        for (i = 0; (i < loaded) && (rc == TURTLE_RETURN_SUCCESS); i += TAMD_STAGE_SLOTS)
                rc = stack_load_tiles(s, come + i, (loaded - i < TAMD_STAGE_SLOTS) ? loaded - i : TAMD_STAGE_SLOTS,
                    message, size);
        for (i = 0; (rc == TURTLE_RETURN_SUCCESS) && (i < loaded); i++) s->owner[come[i]] = &t_me;
        free(here), free(come);
#undef FIRST
#undef DEMAND
        return (rc == TURTLE_RETURN_SUCCESS) ? loaded : -rc;
}

/* One tile for the host's scalar path (scalar.c), as the reference loads one [ref
 * stack.c:428-446]: beyond the stack's size the least recently used tiles go first
 * (none that a thread has just paged in for its round) -- and the point is
 * interpolated in it BEFORE the exclusive hold on the geometry is given up: another
 * thread's load may take the tile away the moment it is. */
int tamd_stack_host_fetch(struct turtle_stack * s, int slot, double latitude, double longitude,
    double * z, int * inside, char * message, size_t size)
{
        const int n = s->latitude_n * s->longitude_n, budget = tamd_stack_budget(s);
        if ((s->lock != NULL) && (s->lock() != 0)) {
                snprintf(message, size, "could not acquire the lock");
                return TURTLE_RETURN_LOCK_ERROR;
        }
        tamd_geometry_write_begin();
        int rc = TURTLE_RETURN_SUCCESS;
        if (s->tile[slot] == NULL) {
                while (s->n_loaded >= budget) {
                        int i, out = -1;
                        for (i = 0; i < n; i++) {
                                if ((s->tile[i] == NULL) || (s->owner[i] != NULL)) continue;
                                if ((out < 0) || (s->stamp[i] < s->stamp[out])) out = i;
                        }
                        if (out < 0) break;
                        stack_drop_tile(s, out);
                }
                rc = stack_load_tiles(s, &slot, 1, message, size);
        }
        if (rc == TURTLE_RETURN_SUCCESS) {
                s->stamp[slot] = ++s->clock; /* [ref stack.c:391-396] */
                double elevation;
                if (tamd_h_map_elevation(s->tile[slot], longitude, latitude, &elevation))
                        *z = elevation, *inside = 1;
        }
        tamd_geometry_write_end();
        if ((s->unlock != NULL) && (s->unlock() != 0) && (rc == TURTLE_RETURN_SUCCESS)) {
                snprintf(message, size, "could not release the lock");
                rc = TURTLE_RETURN_UNLOCK_ERROR;
        }
        return rc;
}

/* [ref stack.c:257-297]: bring tiles into memory, in directory order, until the
 * stack is full (they go on to HBM at the next device call) */
enum turtle_return turtle_stack_load(struct turtle_stack * stack)
{
        TAMD_ERROR_INIT(&turtle_stack_load);
        if ((stack->latitude_n == 0) || (stack->longitude_n == 0))
                return TURTLE_RETURN_SUCCESS;
        if ((stack->lock != NULL) && (stack->lock() != 0))
                return TAMD_RAISE(TURTLE_RETURN_LOCK_ERROR, "could not acquire the lock");
        char message[4200];
        const int rc = tamd_stack_preload(stack, message, sizeof(message));
        if ((stack->unlock != NULL) && (stack->unlock() != 0))
                return TAMD_RAISE(TURTLE_RETURN_UNLOCK_ERROR, "could not release the lock");
        if (rc != TURTLE_RETURN_SUCCESS)
                return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        return TURTLE_RETURN_SUCCESS;
}

/* One-stack view for the elevation / gradient kernels, in a device block of the
 * calling THREAD (rebuilt when tiles came or went, or the thread last built
 * another stack's).  Returns 0, -1 on a device error, or a positive enum
 * turtle_return with `message` set. */
static __thread struct {
        const struct turtle_stack * stack;
        unsigned long epoch;
        int device;
        struct tamd_view view;
} t_view = { NULL, 0, -1, { 0 } };

static int stack_view(struct turtle_stack * s, struct tamd_view * view, char * message,
    size_t size)
{
        if (tamd_dev_init()) return -1;
        const int slots = s->latitude_n * s->longitude_n;
        const size_t bytes = sizeof(struct tamd_stack) + sizeof(struct tamd_meta) +
            (size_t)(slots + 1) * (sizeof(int) + sizeof(struct tamd_grid));
        void * block;
        int grown = 0;
        if (tamd_dev_block(1, &block, bytes, &grown)) return -1;
        tamd_geometry_lock();
        if (!grown && (t_view.stack == s) && (t_view.epoch == tamd_geometry_epoch_get()) &&
            (t_view.device == tamd_dev_current())) {
                *view = t_view.view;
                tamd_geometry_unlock();
                return 0;
        }
        t_view.stack = NULL;
        char * host = calloc(1, bytes);
        if (host == NULL) {
                tamd_geometry_unlock();
                snprintf(message, size, "could not allocate memory");
                return TURTLE_RETURN_MEMORY_ERROR;
        }
        struct tamd_grid * grids = (struct tamd_grid *)host;
        struct tamd_stack * st = (struct tamd_stack *)(grids + slots + 1);
        struct tamd_meta * meta = (struct tamd_meta *)(st + 1);
        int * tiles = (int *)(meta + 1);
        int i, n_grids = 0;
        for (i = 0; i < slots; i++) {
                tiles[i] = (s->path[i] != NULL) ? TAMD_TILE_PAGED : TAMD_TILE_NONE;
                if (s->tile[i] == NULL) continue;
                if (tamd_map_sync(s->tile[i], &grids[n_grids])) {
                        tamd_geometry_unlock();
                        free(host);
                        return -1;
                }
                tiles[i] = n_grids++;
        }
        const unsigned long epoch = tamd_geometry_epoch_get();
        tamd_geometry_unlock();
        st->lat0 = s->latitude_0, st->lon0 = s->longitude_0;
        st->dlat = s->latitude_delta, st->dlon = s->longitude_delta;
        st->inv_dlat = 1. / st->dlat, st->inv_dlon = 1. / st->dlon;
        st->nlat = s->latitude_n, st->nlon = s->longitude_n;
        st->tile_first = 0;
        meta->kind = TAMD_STACK;
        if (tamd_dev_h2d(block, host, bytes)) {
                free(host);
                return -1;
        }
        char * dev = block;
        memset(&t_view.view, 0, sizeof(t_view.view));
        t_view.view.grids = (const struct tamd_grid *)dev;
        t_view.view.stacks = (const struct tamd_stack *)(dev + ((char *)st - host));
        t_view.view.metas = (const struct tamd_meta *)(dev + ((char *)meta - host));
        t_view.view.tiles = (const int *)(dev + ((char *)tiles - host));
        t_view.view.n_layers = 1;
        t_view.view.geoid = -1;
        t_view.stack = s, t_view.epoch = epoch, t_view.device = tamd_dev_current();
        *view = t_view.view;
        free(host);
        return 0;
}

/* The rounds of a batch call on one stack (paging.c): `launch` runs the kernel
 * of a round.  Returns 0, -1 (device) or a positive enum turtle_return. */
struct stack_call {
        struct turtle_stack * stack;
        long n;
        void *a, *b, *c, *d, *e;
        int gradient;
};

static int stack_rounds(struct stack_call * call, char * message, size_t size)
{
        struct turtle_stack * s = call->stack;
        struct tamd_pager pager;
        memset(&pager, 0, sizeof(pager));
        if (tamd_stack_is_paged(s) &&
            tamd_pager_begin(&pager, call->n, s->latitude_n * s->longitude_n))
                return -1;
        int rc = 0;
        for (;;) {
                struct tamd_view view;
                struct tamd_paging pg;
                tamd_geometry_use_begin();
                if ((rc = stack_view(s, &view, message, size)) == 0) {
                        if (tamd_pager_round(&pager, &pg) ||
                            (call->gradient ?
                                    tamd_k_gradient(view, call->n, call->a, call->b, call->c, call->d, call->e, pg) :
                                    tamd_k_elevation(view, call->n, call->a, call->b, call->c, call->e, pg)))
                                rc = -1;
                }
                tamd_geometry_use_end();
                if (rc != 0) break;
                unsigned long long faulted = 0;
                if (tamd_pager_collect(&pager, &faulted)) {
                        rc = -1;
                        break;
                }
                if (faulted == 0) break;
                const int got = tamd_stack_page_in(s, pager.wanted, pager.pinned, 0,
                    faulted <= TAMD_PAGING_FEW, message, size);
                if (got < 0) {
                        rc = -got;
                        break;
                }
                if (pager.rounds > TAMD_PAGING_ROUNDS) { /* cannot be: a round serves an item */
                        snprintf(message, size, "stack of %d tiles is too small for this query (%s)",
                            tamd_stack_budget(s), s->root);
                        rc = TURTLE_RETURN_MEMORY_ERROR;
                        break;
                }
        }
        tamd_pager_end(&pager);
        tamd_stack_trim(s);
        return rc;
}

static int stack_elevation_n(struct turtle_stack * stack, long n,
    const double * latitude, const double * longitude, double * elevation,
    int * inside, int space, char * message, size_t size)
{
        struct tamd_stage st;
        void *da, *db, *dz, *di;
        const size_t nb = (size_t)n * sizeof(double);
        if (tamd_stage_begin(&st, space, 3 * nb + n * sizeof(int))) return -1;
        if (tamd_stage_in(&st, latitude, nb, &da) || tamd_stage_in(&st, longitude, nb, &db) ||
            tamd_stage_out(&st, elevation, nb, &dz) ||
            tamd_stage_out(&st, inside, n * sizeof(int), &di))
                return -1;
        struct stack_call call = { stack, n, da, db, dz, NULL, di, 0 };
        const int rc = stack_rounds(&call, message, size);
        if (rc != 0) return rc;
        if (tamd_stage_fetch(&st, elevation, nb, dz) ||
            tamd_stage_fetch(&st, inside, n * sizeof(int), di))
                return -1;
        return tamd_stage_end(&st) ? -1 : 0;
}

enum turtle_return turtle_stack_elevation_n(struct turtle_stack * stack, long n,
    const double * latitude, const double * longitude, double * elevation,
    int * inside, int space)
{
        TAMD_ERROR_INIT(&turtle_stack_elevation_n);
        if ((stack == NULL) || (inside == NULL) || (elevation == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        char message[4200];
        const int rc = stack_elevation_n(stack, n, latitude, longitude, elevation, inside,
            space, message, sizeof(message));
        if (rc < 0) return TAMD_RAISE_DEVICE();
        if (rc > 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        return TURTLE_RETURN_SUCCESS;
}

/* Shared by the stack and client scalar calls [ref stack.c:338-361] */
enum turtle_return tamd_stack_elevation_scalar(struct turtle_stack * stack,
    turtle_function_t * caller, double latitude, double longitude, double * elevation,
    int * inside)
{
        struct tamd_error error_ = { TURTLE_RETURN_SUCCESS, caller };
        if (inside != NULL) *inside = 0;
        double z = 0.;
        int in = 0;
        char message[4200];
        const int rc = tamd_scalar_on_host() ? /* (the caller's option: scalar.c) */
            tamd_h_stack_elevation(stack, latitude, longitude, &z, &in, message, sizeof(message)) :
            stack_elevation_n(stack, 1, &latitude, &longitude, &z, &in, TURTLE_AMD_HOST, message,
                sizeof(message));
        if (rc < 0) return TAMD_RAISE_DEVICE();
        if (rc > 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        *elevation = in ? z : 0.;
        if (inside != NULL)
                *inside = in;
        else if (!in) /* [ref stack.c:403-411, error.h:86-91] */
                return TAMD_RAISE(TURTLE_RETURN_PATH_ERROR,
                    "missing elevation data in `%s'", stack->root);
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_stack_elevation(struct turtle_stack * stack,
    double latitude, double longitude, double * elevation, int * inside)
{
        return tamd_stack_elevation_scalar(stack,
            (turtle_function_t *)&turtle_stack_elevation, latitude, longitude, elevation,
            inside);
}

/* ---- gradient [ref stack.c:364-388] ----------------------------------------- */

static int stack_gradient_n(struct turtle_stack * stack, long n, const double * latitude,
    const double * longitude, double * glat, double * glon, int * inside, int space,
    char * message, size_t size)
{
        struct tamd_stage st;
        void *da, *db, *dga, *dgb, *di;
        const size_t nb = (size_t)n * sizeof(double);
        if (tamd_stage_begin(&st, space, 4 * nb + n * sizeof(int))) return -1;
        if (tamd_stage_in(&st, latitude, nb, &da) || tamd_stage_in(&st, longitude, nb, &db) ||
            tamd_stage_in(&st, glat, nb, &dga) || tamd_stage_in(&st, glon, nb, &dgb) ||
            tamd_stage_out(&st, inside, n * sizeof(int), &di))
                return -1;
        struct stack_call call = { stack, n, da, db, dga, dgb, di, 1 };
        const int rc = stack_rounds(&call, message, size);
        if (rc != 0) return rc;
        if (tamd_stage_fetch(&st, glat, nb, dga) || tamd_stage_fetch(&st, glon, nb, dgb) ||
            tamd_stage_fetch(&st, inside, n * sizeof(int), di))
                return -1;
        return tamd_stage_end(&st) ? -1 : 0;
}

enum turtle_return turtle_stack_gradient_n(struct turtle_stack * stack, long n,
    const double * latitude, const double * longitude, double * glat, double * glon,
    int * inside, int space)
{
        TAMD_ERROR_INIT(&turtle_stack_gradient_n);
        if ((stack == NULL) || (inside == NULL) || (glat == NULL) || (glon == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        char message[4200];
        const int rc = stack_gradient_n(stack, n, latitude, longitude, glat, glon, inside,
            space, message, sizeof(message));
        if (rc < 0) return TAMD_RAISE_DEVICE();
        if (rc > 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_stack_gradient(struct turtle_stack * stack, double latitude,
    double longitude, double * glat, double * glon, int * inside)
{
        TAMD_ERROR_INIT(&turtle_stack_gradient);
        if (inside != NULL) *inside = 0;
        int in = 0;
        char message[4200];
        const int rc = stack_gradient_n(stack, 1, &latitude, &longitude, glat, glon, &in,
            TURTLE_AMD_HOST, message, sizeof(message));
        if (rc < 0) return TAMD_RAISE_DEVICE();
        if (rc > 0) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        if (inside != NULL)
                *inside = in;
        else if (!in)
                return TAMD_RAISE(TURTLE_RETURN_PATH_ERROR,
                    "missing elevation data in `%s'", stack->root);
        return TURTLE_RETURN_SUCCESS;
}
