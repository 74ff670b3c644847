/*
 * tiles.c -- the tiles of a stack, from their files to the layout HBM holds them
 * in, several at a time.  A round of a batch call over a paged stack (paging.c)
 * waits for the tiles it brings in [ref stack.c:399-450: the reference loads a
 * tile the moment a query needs it]; reading a 3601^2 tile, turning its byte
 * order and rows and laying its nodes out in blocks takes ~15 ms on one core.
 * So the tiles of a round are cut in BANDS of rows (formats whose rows sit at
 * known offsets: .hgt, uncompressed GeoTIFF) and a crew of worker threads takes
 * the bands, each read, decoded and laid out straight into a page-locked staging
 * buffer from which the upload is one copy.  Workers touch files and host memory
 * only: no device call.
 */
#define _GNU_SOURCE
#include "host.h"

#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

size_t tamd_blocked_bytes(int nx, int ny)
{
        const size_t nbx = ((size_t)nx + TAMD_BLOCK - 1) / TAMD_BLOCK;
        const size_t nby = ((size_t)ny + TAMD_BLOCK - 1) / TAMD_BLOCK;
        return nbx * nby * TAMD_BLOCK * TAMD_BLOCK * sizeof(uint16_t);
}

/* rows iy0 .. iy1 - 1 of nx nodes -> their places in blocks of TAMD_BLOCK x TAMD_BLOCK
 * nodes (internal.h); iy0 a multiple of TAMD_BLOCK, iy1 too or ny: the padding of
 * the block rows concerned is zeroed */
void tamd_blocked_fill_rows(const struct turtle_map * map, uint16_t * blocked, int iy0, int iy1)
{
        const size_t nbx = ((size_t)map->nx + TAMD_BLOCK - 1) / TAMD_BLOCK;
        const size_t cell = TAMD_BLOCK * TAMD_BLOCK;
        const int whole = map->nx / TAMD_BLOCK, rest = map->nx % TAMD_BLOCK;
        int ix, iy;
        for (iy = iy0; iy < iy1; iy++) {
                const uint16_t * row = map->nodes + (size_t)iy * map->nx;
                uint16_t * to = blocked + ((size_t)(iy / TAMD_BLOCK) * nbx) * cell +
                    (size_t)(iy % TAMD_BLOCK) * TAMD_BLOCK;
                int b;
                for (b = 0; b < whole; b++) /* 16 bytes at a time */
                        memcpy(to + (size_t)b * cell, row + (size_t)b * TAMD_BLOCK,
                            TAMD_BLOCK * sizeof(*row));
                if (rest) {
                        uint16_t * last = to + (size_t)whole * cell;
                        for (ix = 0; ix < TAMD_BLOCK; ix++)
                                last[ix] = (ix < rest) ? row[(size_t)whole * TAMD_BLOCK + ix] : 0;
                }
        }
        if ((iy1 == map->ny) && (map->ny % TAMD_BLOCK)) { /* the rows below the last block row's end */
                for (iy = map->ny; iy % TAMD_BLOCK; iy++) {
                        uint16_t * to = blocked + ((size_t)(iy / TAMD_BLOCK) * nbx) * cell +
                            (size_t)(iy % TAMD_BLOCK) * TAMD_BLOCK;
                        size_t b;
                        for (b = 0; b < nbx; b++) memset(to + b * cell, 0, TAMD_BLOCK * sizeof(*to));
                }
        }
}

void tamd_blocked_fill(const struct turtle_map * map, uint16_t * blocked)
{
        tamd_blocked_fill_rows(map, blocked, 0, map->ny);
}

typedef int read_rows_t(const char *, struct turtle_map *, int, int);

static read_rows_t * rows_reader(const char * path)
{
        const char * ext = strrchr(path, '.');
        if (ext == NULL) return NULL;
        if (strcmp(ext + 1, "hgt") == 0) return &tamd_hgt_read_rows;
        if (strcmp(ext + 1, "tif") == 0) return &tamd_tiff_read_rows;
        return NULL;
}

/* a band of a tile (or, a format without a rows reader: the whole tile) */
struct band {
        struct tamd_tile_job * job;
        int iy0, iy1;
        int rc;
};

static void band_run(struct band * b)
{
        struct turtle_map * m = b->job->map;
        read_rows_t * rows = rows_reader(b->job->path);
        if (b->job->cached) { /* (no band is cut for such a tile: nothing to read) */
                b->rc = TURTLE_RETURN_SUCCESS;
                return;
        }
        if (rows != NULL)
                b->rc = rows(b->job->path, m, b->iy0, b->iy1);
        else {
                int (*probe)(const char *, struct turtle_map *);
                int (*read)(const char *, struct turtle_map *);
                b->rc = tamd_codec_for(b->job->path, &probe, &read) ? read(b->job->path, m) :
                                                                      TURTLE_RETURN_BAD_EXTENSION;
        }
        if ((b->rc == TURTLE_RETURN_SUCCESS) && (m->staged != NULL))
                tamd_blocked_fill_rows(m, m->staged, b->iy0, b->iy1);
}

struct crew {
        struct band * bands;
        int n, next;
        pthread_mutex_t lock;
};

static void * crew_run(void * arg)
{
        struct crew * c = arg;
        for (;;) {
                pthread_mutex_lock(&c->lock);
                const int i = c->next++;
                pthread_mutex_unlock(&c->lock);
                if (i >= c->n) return NULL;
                band_run(&c->bands[i]);
        }
}

void tamd_tiles_decode(struct tamd_tile_job * jobs, int n)
{
        if (n <= 0) return;
        int threads = 16, k;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
                const int cores = CPU_COUNT(&set);
                if (threads > cores) threads = cores;
        }
        if (threads < 1) threads = 1;
        /* the tiles: header, memory */
        int bands_per_tile = (threads + n - 1) / n;
        if (bands_per_tile > 32) bands_per_tile = 32;
        for (k = 0; k < n; k++) {
                struct tamd_tile_job * job = &jobs[k];
                int (*probe)(const char *, struct turtle_map *);
                int (*read)(const char *, struct turtle_map *);
                job->map = NULL;
                if (!tamd_codec_for(job->path, &probe, &read)) {
                        job->rc = TURTLE_RETURN_BAD_EXTENSION;
                        continue;
                }
                struct turtle_map * m = calloc(1, sizeof(*m));
                int rc = (m == NULL) ? TURTLE_RETURN_MEMORY_ERROR : probe(job->path, m);
                /* (a tile whose nodes are in its staging buffer already comes without a host
                 * copy: 26 MB of fresh pages a tile are what a round would then wait for) */
                const int cached = job->cached && (job->staged != NULL) && (rc == TURTLE_RETURN_SUCCESS) &&
                    (tamd_blocked_bytes(m->nx, m->ny) <= job->staged_bytes);
                job->cached = cached;
                if ((rc == TURTLE_RETURN_SUCCESS) && !cached) {
                        m->nodes = malloc((size_t)m->nx * m->ny * sizeof(*m->nodes));
                        if (m->nodes == NULL) rc = TURTLE_RETURN_MEMORY_ERROR;
                }
                if (cached) m->lazy_path = job->path;
                if (rc != TURTLE_RETURN_SUCCESS) {
                        if (m != NULL) free(m->nodes);
                        free(m);
                        job->rc = (rc > N_TURTLE_RETURNS) ? TURTLE_RETURN_BAD_FORMAT : rc;
                        continue;
                }
                m->staged = NULL, m->staged_slot = -1;
                if ((job->staged != NULL) && (tamd_blocked_bytes(m->nx, m->ny) <= job->staged_bytes))
                        m->staged = job->staged;
                job->map = m;
                job->rc = TURTLE_RETURN_SUCCESS;
        }
        /* their bands */
        struct band * bands = calloc((size_t)n * bands_per_tile, sizeof(*bands));
        int n_bands = 0;
        if (bands == NULL) {
                for (k = 0; k < n; k++) {
                        if (jobs[k].map == NULL) continue;
                        free(jobs[k].map->nodes), free(jobs[k].map);
                        jobs[k].map = NULL, jobs[k].rc = TURTLE_RETURN_MEMORY_ERROR;
                }
                return;
        }
        for (k = 0; k < n; k++) {
                if ((jobs[k].map == NULL) || jobs[k].cached) continue;
                const int ny = jobs[k].map->ny;
                int count = (rows_reader(jobs[k].path) != NULL) ? bands_per_tile : 1;
                /* whole block rows to a band */
                int height = ((ny + count - 1) / count + TAMD_BLOCK - 1) / TAMD_BLOCK * TAMD_BLOCK;
                if (height < TAMD_BLOCK) height = TAMD_BLOCK;
                int iy;
                for (iy = 0; iy < ny; iy += height) {
                        bands[n_bands].job = &jobs[k];
                        bands[n_bands].iy0 = iy;
                        bands[n_bands].iy1 = (iy + height < ny) ? iy + height : ny;
                        n_bands++;
                }
        }
        if (threads > n_bands) threads = n_bands;
        struct crew c = { bands, n_bands, 0, PTHREAD_MUTEX_INITIALIZER };
        pthread_t tid[16];
        int t, started = 0;
        for (t = 1; t < threads; t++) /* the caller is the first worker */
                if (pthread_create(&tid[started], NULL, crew_run, &c) == 0) started++;
        crew_run(&c);
        for (t = 0; t < started; t++) pthread_join(tid[t], NULL);
        pthread_mutex_destroy(&c.lock);
        for (k = 0; k < n_bands; k++) {
                struct tamd_tile_job * job = bands[k].job;
                if ((bands[k].rc != TURTLE_RETURN_SUCCESS) && (job->rc == TURTLE_RETURN_SUCCESS))
                        job->rc = (bands[k].rc > N_TURTLE_RETURNS) ? TURTLE_RETURN_BAD_FORMAT : bands[k].rc;
        }
        for (k = 0; k < n; k++) {
                if ((jobs[k].map == NULL) || (jobs[k].rc == TURTLE_RETURN_SUCCESS)) continue;
                free(jobs[k].map->nodes), free(jobs[k].map);
                jobs[k].map = NULL;
        }
        free(bands);
}
