/*
 * map.c -- a single DEM grid: host handle, HBM residency, scalar and batch
 * elevation [ref src/turtle/map.c:54-421].  The bilinear arithmetic itself is
 * in device.hip (d_grid_elevation); nothing here interpolates.
 */
#include "host.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- what threads share (host.h) ------------------------------------------ */
static pthread_mutex_t g_lock;
static pthread_rwlock_t g_use = PTHREAD_RWLOCK_INITIALIZER;
static pthread_once_t g_lock_once = PTHREAD_ONCE_INIT;
static unsigned long g_epoch = 1;

static void lock_init(void)
{
        pthread_mutexattr_t attr;
        pthread_mutexattr_init(&attr);
        pthread_mutexattr_settype(&attr, PTHREAD_MUTEX_RECURSIVE);
        pthread_mutex_init(&g_lock, &attr);
        pthread_mutexattr_destroy(&attr);
}

void tamd_geometry_lock(void)
{
        pthread_once(&g_lock_once, lock_init);
        pthread_mutex_lock(&g_lock);
}

void tamd_geometry_unlock(void) { pthread_mutex_unlock(&g_lock); }
unsigned long tamd_geometry_epoch_get(void) { return __atomic_load_n(&g_epoch, __ATOMIC_ACQUIRE); }
void tamd_geometry_changed(void) { __atomic_add_fetch(&g_epoch, 1, __ATOMIC_ACQ_REL); }

void tamd_geometry_use_begin(void) { pthread_rwlock_rdlock(&g_use); }
void tamd_geometry_use_end(void) { pthread_rwlock_unlock(&g_use); }

/* Whoever may FREE HBM copies holds the geometry exclusively: first against its
 * users (no thread between building its tables and queueing its launches), then
 * the lock (always in that order; nested in one thread: the outermost counts). */
static __thread int t_write_depth = 0;

void tamd_geometry_write_begin(void)
{
        if (t_write_depth++ == 0) pthread_rwlock_wrlock(&g_use);
        tamd_geometry_lock();
}

void tamd_geometry_write_end(void)
{
        tamd_geometry_unlock();
        if (--t_write_depth == 0) pthread_rwlock_unlock(&g_use);
}

/* The HBM copies of a map go (inside tamd_geometry_write_begin / _end) once what
 * is queued on any stream of their devices has run. */
void tamd_map_release(struct turtle_map * m)
{
        int d;
        for (d = 0; d < TAMD_MAX_DEVICES; d++) {
                if (m->d_nodes[d] == NULL) continue;
                /* (a device that cannot be waited for: the copy is leaked, not freed
                 * under a launch that may still read it) */
                if (tamd_dev_sync_device(d) == 0) tamd_dev_free_on(d, m->d_nodes[d]);
                m->d_nodes[d] = NULL;
        }
        m->d_fresh = 0;
}

/* [ref map.c:54-99] */
enum turtle_return turtle_map_create(struct turtle_map ** map,
    const struct turtle_map_info * info, const char * projection)
{
        TAMD_ERROR_INIT(&turtle_map_create);
        *map = NULL;
        if ((info->nx <= 0) || (info->ny <= 0) || (info->z[0] == info->z[1]))
                return TAMD_RAISE(
                    TURTLE_RETURN_DOMAIN_ERROR, "invalid input parameter(s)");
        struct turtle_projection proj;
        char message[256];
        const int prc = tamd_projection_configure(&proj, projection, message, sizeof(message));
        if (prc != TURTLE_RETURN_SUCCESS)
                return TAMD_RAISE((enum turtle_return)prc, "%s", message);

        struct turtle_map * m = calloc(1, sizeof(*m));
        if (m != NULL) m->nodes = calloc((size_t)info->nx * info->ny, sizeof(*m->nodes));
        if ((m == NULL) || (m->nodes == NULL)) {
                free(m);
                return TAMD_RAISE(
                    TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        }
        m->nx = info->nx;
        m->ny = info->ny;
        m->x0 = info->x[0];
        m->y0 = info->y[0];
        m->z0 = info->z[0];
        m->dx = (info->nx > 1) ? (info->x[1] - info->x[0]) / (info->nx - 1) : 0.;
        m->dy = (info->ny > 1) ? (info->y[1] - info->y[0]) / (info->ny - 1) : 0.;
        m->dz = (info->z[1] - info->z[0]) / 65535;
        strcpy(m->encoding, "none");
        m->projection = proj;
        *map = m;
        return TURTLE_RETURN_SUCCESS;
}

/* [ref map.c:102-113] */
void turtle_map_destroy(struct turtle_map ** map)
{
        if ((map == NULL) || (*map == NULL)) return;
        struct turtle_map * m = *map;
        tamd_geometry_write_begin();
        if (m->stack != NULL) { /* a tile leaves its stack */
                struct turtle_stack * s = m->stack;
                const int n = s->latitude_n * s->longitude_n;
                int i;
                for (i = 0; i < n; i++) {
                        if (s->tile[i] == m) {
                                s->tile[i] = NULL;
                                s->n_loaded--;
                        }
                }
        }
        tamd_map_release(m);
        tamd_geometry_changed();
        tamd_geometry_write_end();
        free(m->nodes);
        free(m);
        *map = NULL;
}

/* Extension dispatch [ref src/turtle/io.c:60-104]: the reference's five
 * formats -- hgt, GeoTIFF-16 (uncompressed strips), PNG-16 maps, grd, asc. */
int tamd_codec_for(const char * path, int (**probe)(const char *, struct turtle_map *),
    int (**read)(const char *, struct turtle_map *))
{
        const char * ext = strrchr(path, '.');
        if (ext == NULL) return 0;
        if (strcmp(ext + 1, "hgt") == 0) {
                *probe = &tamd_hgt_probe, *read = &tamd_hgt_read;
                return 1;
        }
        if (strcmp(ext + 1, "tif") == 0) {
                *probe = &tamd_tiff_probe, *read = &tamd_tiff_read;
                return 1;
        }
        if (strcmp(ext + 1, "png") == 0) {
                *probe = &tamd_png_probe, *read = &tamd_png_read;
                return 1;
        }
        if (strcmp(ext + 1, "grd") == 0) {
                *probe = &tamd_grd_probe, *read = &tamd_grd_read;
                return 1;
        }
        if (strcmp(ext + 1, "asc") == 0) {
                *probe = &tamd_asc_probe, *read = &tamd_asc_read;
                return 1;
        }
        return 0;
}

enum turtle_return tamd_map_load_(struct turtle_map ** map, const char * path,
    struct tamd_error * error, const char * file, int line)
{
        *map = NULL;
        const char * ext = strrchr(path, '.');
        if (ext == NULL)
                return tamd_raise_(error, TURTLE_RETURN_BAD_EXTENSION, file, line,
                    "missing file extension");
        int (*probe)(const char *, struct turtle_map *);
        int (*read)(const char *, struct turtle_map *);
        if (!tamd_codec_for(path, &probe, &read))
                return tamd_raise_(error, TURTLE_RETURN_BAD_EXTENSION, file, line,
                    "unsuported file format `%s'", ext + 1);

        struct turtle_map * m = calloc(1, sizeof(*m));
        if (m == NULL)
                return tamd_raise_(error, TURTLE_RETURN_MEMORY_ERROR, file, line,
                    "could not allocate memory for map `%s'", path);
        int rc = probe(path, m);
        if (rc == TURTLE_RETURN_SUCCESS) {
                m->nodes = malloc((size_t)m->nx * m->ny * sizeof(*m->nodes));
                rc = (m->nodes == NULL) ? TURTLE_RETURN_MEMORY_ERROR : read(path, m);
        }
        if (rc != TURTLE_RETURN_SUCCESS) {
                free(m->nodes);
                free(m);
                if (rc == TURTLE_RETURN_BAD_FORMAT + 101) {
                        return tamd_raise_(error, TURTLE_RETURN_BAD_FORMAT, file, line,
                            "inconsistent data in file `%s'", path);
                }
                if (rc == TURTLE_RETURN_BAD_FORMAT + 102)
                        return tamd_raise_(error, TURTLE_RETURN_BAD_FORMAT, file, line,
                            "could not read the header of file `%s'", path);
                const char * text = (rc == TURTLE_RETURN_PATH_ERROR) ?
                    "could not open file `%s'" :
                    ((rc == TURTLE_RETURN_MEMORY_ERROR) ?
                            "could not allocate memory for map `%s'" :
                            ((rc == TURTLE_RETURN_BAD_FORMAT + 100) ?
                                    "missing data when reading file `%s'" :
                                    ((strcmp(ext + 1, "hgt") == 0) ?
                                            "invalid hgt filename for `%s'" :
                                            "not an uncompressed 16-bit strip TIFF: `%s'")));
                if (rc == TURTLE_RETURN_BAD_FORMAT + 100) rc = TURTLE_RETURN_BAD_FORMAT;
                return tamd_raise_(error, (enum turtle_return)rc, file, line, text, path);
        }
        *map = m;
        return TURTLE_RETURN_SUCCESS;
}

/* [ref map.c:157-162] */
enum turtle_return turtle_map_load(struct turtle_map ** map, const char * path)
{
        TAMD_ERROR_INIT(&turtle_map_load);
        return tamd_map_load_(map, path, &error_, __FILE__, __LINE__);
}

/* [ref map.c:183-205]; the 16-bit code is written as map.c:47-51 (default
 * encoding) or as a plain int16 (hgt; the reference's own setter,
 * io/hgt.c:133-137, byte-swaps after converting the double, which we do not
 * reproduce: the stored value here is the elevation itself). */
enum turtle_return turtle_map_fill(
    struct turtle_map * map, int ix, int iy, double elevation)
{
        TAMD_ERROR_INIT(&turtle_map_fill);
        if (map == NULL)
                return TAMD_RAISE(
                    TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        if ((ix < 0) || (ix >= map->nx) || (iy < 0) || (iy >= map->ny))
                return TAMD_RAISE(
                    TURTLE_RETURN_DOMAIN_ERROR, "point is outside of map");
        if ((map->dz <= 0.) && (elevation != map->z0))
                return TAMD_RAISE(
                    TURTLE_RETURN_DOMAIN_ERROR, "inconsistent elevation value");
        if ((elevation < map->z0) || (elevation > map->z0 + 65535 * map->dz))
                return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR,
                    "elevation is outside of map span");
        uint16_t code;
        if (map->is_signed)
                code = (uint16_t)(int16_t)elevation;
        else
                code = (uint16_t)round((elevation - map->z0) / map->dz);
        tamd_geometry_lock();
        map->nodes[(size_t)iy * map->nx + ix] = code;
        map->d_fresh = 0; /* every HBM copy is stale */
        tamd_geometry_changed();
        tamd_geometry_unlock();
        return TURTLE_RETURN_SUCCESS;
}

/* [ref map.c:208-226]: node coordinates and its stored value (a decode of one
 * 16-bit code, not an interpolation) */
enum turtle_return turtle_map_node(const struct turtle_map * map, int ix, int iy,
    double * x, double * y, double * elevation)
{
        TAMD_ERROR_INIT(&turtle_map_node);
        if (map == NULL)
                return TAMD_RAISE(
                    TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        if ((ix < 0) || (ix >= map->nx) || (iy < 0) || (iy >= map->ny))
                return TAMD_RAISE(
                    TURTLE_RETURN_DOMAIN_ERROR, "point is outside of map");
        if (x != NULL) *x = map->x0 + ix * map->dx;
        if (y != NULL) *y = map->y0 + iy * map->dy;
        if (elevation != NULL) {
                const uint16_t code = map->nodes[(size_t)iy * map->nx + ix];
                *elevation = map->is_signed ? (double)(int16_t)code :
                                              map->z0 + code * map->dz;
        }
        return TURTLE_RETURN_SUCCESS;
}

/* [ref map.c:394-400] */
const struct turtle_projection * turtle_map_projection(const struct turtle_map * map)
{
        if ((map == NULL) || (map->projection.type < 0)) return NULL;
        return &map->projection;
}

/* [ref map.c:403-421] */
void turtle_map_meta(const struct turtle_map * map, struct turtle_map_info * info,
    const char ** projection)
{
        if (info != NULL) {
                info->nx = map->nx;
                info->ny = map->ny;
                info->x[0] = map->x0;
                info->x[1] = map->x0 + (map->nx - 1) * map->dx;
                info->y[0] = map->y0;
                info->y[1] = map->y0 + (map->ny - 1) * map->dy;
                info->z[0] = map->z0;
                info->z[1] = map->z0 + 65535 * map->dz;
                info->encoding = map->encoding;
        }
        if (projection != NULL) *projection = turtle_projection_name(&map->projection);
}

/* A tile that came back from a staging buffer has no host copy of its nodes (host.h): the
 * first reader on the host -- the scalar path, an upload to a second device -- reads the file,
 * under the geometry lock (readers on the host hold the geometry in use: nobody frees the tile) */
int tamd_map_host_nodes(struct turtle_map * map)
{
        if (__atomic_load_n(&map->nodes, __ATOMIC_ACQUIRE) != NULL) return TURTLE_RETURN_SUCCESS;
        int rc = TURTLE_RETURN_SUCCESS;
        tamd_geometry_lock();
        if (map->nodes == NULL) {
                int (*probe)(const char *, struct turtle_map *);
                int (*read)(const char *, struct turtle_map *);
                struct turtle_map copy = *map;
                copy.nodes = NULL;
                if ((map->lazy_path == NULL) || !tamd_codec_for(map->lazy_path, &probe, &read))
                        rc = TURTLE_RETURN_PATH_ERROR;
                else {
                        copy.nodes = malloc((size_t)map->nx * map->ny * sizeof(*copy.nodes));
                        rc = (copy.nodes == NULL) ? TURTLE_RETURN_MEMORY_ERROR : read(map->lazy_path, &copy);
                        if (rc > N_TURTLE_RETURNS) rc = TURTLE_RETURN_BAD_FORMAT;
                }
                if (rc == TURTLE_RETURN_SUCCESS)
                        __atomic_store_n(&map->nodes, copy.nodes, __ATOMIC_RELEASE);
                else
                        free(copy.nodes);
        }
        tamd_geometry_unlock();
        return rc;
}

int tamd_map_sync(struct turtle_map * map, struct tamd_grid * grid)
{
        /* HBM layout: blocks of TAMD_BLOCK x TAMD_BLOCK nodes (internal.h); one copy
         * per device, made when a thread on that device first needs it */
        if (tamd_dev_init()) return 1;
        const int device = tamd_dev_current();
        if ((device < 0) || (device >= TAMD_MAX_DEVICES)) return 1;
        const size_t nbx = ((size_t)map->nx + TAMD_BLOCK - 1) / TAMD_BLOCK;
        const size_t nby = ((size_t)map->ny + TAMD_BLOCK - 1) / TAMD_BLOCK;
        const size_t bytes = nbx * nby * TAMD_BLOCK * TAMD_BLOCK * sizeof(*map->nodes);
        tamd_geometry_lock();
        if (map->d_nodes[device] == NULL) {
                /* a tile: the buffer of one that went, if its stack kept any */
                map->d_nodes[device] = tamd_stack_spare_take(map->stack, device, bytes);
                if ((map->d_nodes[device] == NULL) && tamd_dev_malloc(&map->d_nodes[device], bytes)) {
                        tamd_geometry_unlock();
                        return 1;
                }
                map->d_fresh &= ~(1u << device);
        }
        if (!(map->d_fresh & (1u << device))) {
                int failed;
                if (map->staged != NULL) {
                        /* a tile just read: laid out already, in page-locked memory
                         * (tiles.c) */
                        /* ... and waited for: the copy is on THIS thread's stream, and once
                         * the tile reads "current" another thread's launches, on another
                         * stream, may read it (26 MB from page-locked memory: half a
                         * millisecond) */
                        const int queued =
                            (tamd_dev_copy_async(map->d_nodes[device], map->staged, bytes, 1) == 0);
                        failed = !queued || tamd_dev_sync();
                        /* a copy that was queued and could not be waited for may still be
                         * reading the buffer: the slot is free once that device has drained
                         * (stage_acquire), not now */
                        tamd_stack_staged_done(map, (queued && failed) ? device : -1);
                } else {
                        uint16_t * blocked = (tamd_map_host_nodes(map) == TURTLE_RETURN_SUCCESS) ? malloc(bytes) : NULL;
                        if (blocked == NULL) {
                                tamd_geometry_unlock();
                                return 1;
                        }
                        tamd_blocked_fill(map, blocked);
                        /* (a copy being rewritten while launches of other threads read it:
                         * turtle_map_fill on a map in use is the caller's race, as in the
                         * reference) */
                        failed = tamd_dev_h2d(map->d_nodes[device], blocked, bytes);
                        free(blocked);
                }
                if (failed) {
                        tamd_geometry_unlock();
                        return 1;
                }
                map->d_fresh |= 1u << device;
        }
        tamd_geometry_unlock();
        if (grid != NULL) {
                grid->nodes = map->d_nodes[device];
                grid->nx = map->nx, grid->ny = map->ny;
                grid->x0 = map->x0, grid->y0 = map->y0;
                grid->dx = map->dx, grid->dy = map->dy;
                grid->inv_dx = 1. / map->dx, grid->inv_dy = 1. / map->dy;
                /* int16 codecs return the code itself [ref io/hgt.c:127-131] */
                grid->z0 = map->is_signed ? 0. : map->z0;
                grid->dz = map->is_signed ? 1. : map->dz;
                grid->is_signed = map->is_signed;
                grid->nbx = (int)nbx;
                tamd_projection_desc(&map->projection, &grid->proj);
        }
        return 0;
}

/* A one-grid view for the elevation kernel: tables live in the scratch arena */
static int map_view(struct turtle_map * map, struct tamd_view * view)
{
        struct {
                struct tamd_grid grid;
                struct tamd_meta meta;
        } tables;
        memset(&tables, 0, sizeof(tables));
        if (tamd_map_sync(map, &tables.grid)) return 1;
        tables.meta.kind = TAMD_MAP;
        void * dev;
        if (tamd_scratch_get(&dev, sizeof(tables))) return 1;
        if (tamd_dev_h2d(dev, &tables, sizeof(tables))) return 1;
        memset(view, 0, sizeof(*view));
        view->grids = (const struct tamd_grid *)dev;
        view->metas = (const struct tamd_meta *)((char *)dev + sizeof(struct tamd_grid));
        view->n_layers = 1;
        view->geoid = -1;
        return 0;
}

static int map_elevation_n(struct turtle_map * map, long n, const double * x,
    const double * y, double * elevation, int * inside, int space)
{
        struct tamd_stage st;
        struct tamd_view view;
        void *dx, *dy, *dz, *di;
        const size_t nb = (size_t)n * sizeof(double);
        if (tamd_stage_begin(&st, space, 3 * nb + n * sizeof(int) + 4096)) return 1;
        if (space == TURTLE_AMD_DEVICE) tamd_scratch_reset();
        if (map_view(map, &view)) return 1;
        if (tamd_stage_in(&st, x, nb, &dx) || tamd_stage_in(&st, y, nb, &dy) ||
            tamd_stage_out(&st, elevation, nb, &dz) ||
            tamd_stage_out(&st, inside, n * sizeof(int), &di))
                return 1;
        {
                const struct tamd_paging none = { NULL, NULL, NULL, NULL, NULL, NULL, NULL, -1 };
                if (tamd_k_elevation(view, n, dx, dy, dz, di, none)) return 1;
        }
        if (tamd_stage_fetch(&st, elevation, nb, dz) ||
            tamd_stage_fetch(&st, inside, n * sizeof(int), di))
                return 1;
        /* the one-grid tables sit in the scratch arena: finish before reuse */
        if (tamd_stage_end(&st)) return 1;
        return tamd_dev_sync();
}

enum turtle_return turtle_map_elevation_n(const struct turtle_map * map, long n,
    const double * x, const double * y, double * elevation, int * inside, int space)
{
        TAMD_ERROR_INIT(&turtle_map_elevation_n);
        if ((map == NULL) || (inside == NULL) || (elevation == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        if (map_elevation_n((struct turtle_map *)map, n, x, y, elevation, inside, space))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

/* [ref map.c:380-385 -> :229-277] */
enum turtle_return turtle_map_elevation(const struct turtle_map * map, double x,
    double y, double * elevation, int * inside)
{
        TAMD_ERROR_INIT(&turtle_map_elevation);
        double z = 0.;
        int in = 0;
        if (tamd_scalar_on_host()) /* (the caller's option: scalar.c) */
                in = tamd_h_map_elevation(map, x, y, &z);
        else if (map_elevation_n((struct turtle_map *)map, 1, &x, &y, &z, &in, TURTLE_AMD_HOST))
                return TAMD_RAISE_DEVICE();
        if (in) *elevation = z; /* an outside point leaves *elevation untouched */
        if (inside != NULL)
                *inside = in;
        else if (!in)
                return TAMD_RAISE(
                    TURTLE_RETURN_DOMAIN_ERROR, "point is outside of map");
        return TURTLE_RETURN_SUCCESS;
}

/* ---- gradient [ref map.c:280-392] ------------------------------------------ */

static int map_gradient_n(struct turtle_map * map, long n, const double * x,
    const double * y, double * gx, double * gy, int * inside, int space)
{
        struct tamd_stage st;
        struct tamd_view view;
        void *dx, *dy, *dgx, *dgy, *di;
        const size_t nb = (size_t)n * sizeof(double);
        if (tamd_stage_begin(&st, space, 4 * nb + n * sizeof(int) + 4096)) return 1;
        if (space == TURTLE_AMD_DEVICE) tamd_scratch_reset();
        if (map_view(map, &view)) return 1;
        /* gx, gy are in-out: a point outside the map leaves them untouched */
        if (tamd_stage_in(&st, x, nb, &dx) || tamd_stage_in(&st, y, nb, &dy) ||
            tamd_stage_in(&st, gx, nb, &dgx) || tamd_stage_in(&st, gy, nb, &dgy) ||
            tamd_stage_out(&st, inside, n * sizeof(int), &di))
                return 1;
        {
                const struct tamd_paging none = { NULL, NULL, NULL, NULL, NULL, NULL, NULL, -1 };
                if (tamd_k_gradient(view, n, dx, dy, dgx, dgy, di, none)) return 1;
        }
        if (tamd_stage_fetch(&st, gx, nb, dgx) || tamd_stage_fetch(&st, gy, nb, dgy) ||
            tamd_stage_fetch(&st, inside, n * sizeof(int), di))
                return 1;
        if (tamd_stage_end(&st)) return 1;
        return tamd_dev_sync();
}

enum turtle_return turtle_map_gradient_n(const struct turtle_map * map, long n,
    const double * x, const double * y, double * gx, double * gy, int * inside, int space)
{
        TAMD_ERROR_INIT(&turtle_map_gradient_n);
        if ((map == NULL) || (inside == NULL) || (gx == NULL) || (gy == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        if (map_gradient_n((struct turtle_map *)map, n, x, y, gx, gy, inside, space))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

/* [ref map.c:387-392] */
enum turtle_return turtle_map_gradient(const struct turtle_map * map, double x, double y,
    double * gx, double * gy, int * inside)
{
        TAMD_ERROR_INIT(&turtle_map_gradient);
        int in = 0;
        if (map_gradient_n((struct turtle_map *)map, 1, &x, &y, gx, gy, &in, TURTLE_AMD_HOST))
                return TAMD_RAISE_DEVICE();
        if (inside != NULL)
                *inside = in;
        else if (!in)
                return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "point is outside of map");
        return TURTLE_RETURN_SUCCESS;
}
