/*
 * internal.h -- declarations shared by the host C objects and the HIP device
 * layer of libturtle_amd.  Not installed.
 *
 * Split of responsibilities:
 *   host (C99: error.c map.c hgt.c stack.c client.c stepper.c ecef.c batch.c)
 *        owns the opaque handles of the public API, file ingest, the error
 *        convention, and flattening a stepper into the POD tables below;
 *   device (device.hip) owns HBM, the stream, and every kernel.  All
 *        arithmetic of the path happens there.
 */
#ifndef TURTLE_AMD_INTERNAL_H
#define TURTLE_AMD_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "turtle_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* POD tables read by the kernels (layout shared by host and device code)   */
/* ------------------------------------------------------------------------ */

/* One DEM grid resident in HBM: 16-bit nodes, native little-endian, in BLOCKS
 * of 8 x 8 nodes (128 bytes, one cache line; rows south->north inside a block
 * and from block to block, nbx blocks per block row, the grid padded to whole
 * blocks): node (ix, iy) is at ((iy / 8) * nbx + ix / 8) * 64 + (iy % 8) * 8 +
 * ix % 8.  A sample reads the 2 x 2 nodes of a cell; in rows of nx nodes those
 * are two lines 2 nx bytes apart, in blocks one line three times out of four,
 * and the next cells of the ray -- whichever way it heads -- are in it too.
 * Decoding the file format (byte order, row flip, sign) happens ONCE at upload
 * instead of per node access as the reference's get_z callbacks do [ref
 * src/turtle/map.h:47-49, io/hgt.c:127-131, map.c:41-44]; integers are exact,
 * so parity is unaffected.  z = z0 + v * dz with v read as int16 if is_signed
 * else uint16 (signed codecs use z0 = 0, dz = 1, which reproduces "(int16)v"
 * exactly). */
#define TAMD_BLOCK 8
/* A map projection [ref src/turtle/projection.h:29-46]; type < 0: geodetic */
enum tamd_proj_type { TAMD_PROJ_NONE = -1, TAMD_PROJ_LAMBERT = 0, TAMD_PROJ_UTM = 1 };

struct tamd_proj {
        int type;           /* enum tamd_proj_type */
        int lambert_tag;    /* 0..5: I, II, IIe, III, IV, 93 */
        double longitude_0; /* UTM central meridian, degrees */
        int hemisphere;     /* UTM: +1 north, -1 south */
        int pad_;
};

struct tamd_grid {
        const uint16_t * nodes;
        int nx, ny;
        double x0, y0, dx, dy;
        double z0, dz;
        double inv_dx, inv_dy; /* 1/dx, 1/dy: the fast-math kernels multiply */
        int is_signed;
        int nbx; /* blocks of TAMD_BLOCK x TAMD_BLOCK nodes per block row */
        struct tamd_proj proj; /* x, y of a projected map; the stepper projects
                                * (latitude, longitude) first [ref stepper.c:243-248] */
};

/* Tile directory of a stack [ref src/turtle/stack.h:32-49]: O(1) lookup
 * replaces the reference's MRU list scan [ref stack.c:300-335]. */
struct tamd_stack {
        double lat0, lon0, dlat, dlon;
        double inv_dlat, inv_dlon; /* the fast-math lookup multiplies (seams: exact) */
        int nlat, nlon;
        int tile_first; /* offset into the tiles[] table: grid index or -1 */
        /* `regular`: every tile present has the same shape and encoding (nx,
         * ny, dx, dy, z0, dz, sign) and sits exactly on the lattice (x0 ==
         * lon0 + ix*dlon, y0 == lat0 + iy*dlat) whose cell it spans ((nx-1) dx
         * == dlon up to rounding), as SRTM/ASTER tiles do.  The
         * fast-math kernels then need one pointer per tile (slot_nodes[
         * nodes_first + slot], NULL for a missing tile) instead of a whole
         * per-lane grid descriptor; `proto` holds the shared shape. */
        int regular;
        int nodes_first, pad_;
        struct tamd_grid proto;
};

enum tamd_kind { TAMD_FLAT = 0, TAMD_MAP = 1, TAMD_STACK = 2 };

/* values of the tile table (tamd_view.tiles) besides a grid index */
#define TAMD_TILE_NONE (-1)  /* no file for this slot */
#define TAMD_TILE_PAGED (-2) /* a file, not resident: see "Paging" in device.hip */

/* One (data, offset) entry of a layer [ref src/turtle/stepper.h:80-85], stored
 * in the reference's iteration order: last added first [ref stepper.c:722-724] */
struct tamd_meta {
        int kind; /* enum tamd_kind */
        int src;  /* grid index (MAP) or stack index (STACK) */
        double offset;
};

enum tamd_mode {
        TAMD_MODE_GENERIC = 0,   /* any layers / data / geoid */
        TAMD_MODE_ONE_MAP = 1,   /* one layer, one geodetic map, no geoid */
        TAMD_MODE_ONE_STACK = 2  /* one layer, one stack, no geoid */
};

/* Everything a kernel needs about a stepper, passed BY VALUE as a kernel
 * argument so that it sits in scalar registers. */
struct tamd_view {
        const struct tamd_grid * grids;
        const struct tamd_stack * stacks;
        const int * tiles;
        const uint16_t * const * slot_nodes; /* see tamd_stack.regular */
        const struct tamd_meta * metas;
        const int * layer_first; /* n_layers + 1 offsets into metas */
        int n_layers;
        int geoid; /* grid index or -1 */
        double slope, resolution;
        int mode; /* enum tamd_mode */
        int fast_ok; /* every grid has nx, ny >= 2: the clamped fast lookup applies */
};

/* ------------------------------------------------------------------------ */
/* Device layer (device.hip).  Every function returns 0 on success or a     */
/* non-zero value after recording a message readable with tamd_dev_error(). */
/* ------------------------------------------------------------------------ */

/* The device, the stream, the arithmetic mode, the scratch arena and the blocks
 * below belong to the calling THREAD (device.hip: struct Ctx). */
const char * tamd_dev_error(void);
int tamd_dev_init(void);   /* idempotent; selects the thread's device, makes its stream */
void tamd_dev_release(void); /* frees what the calling thread holds on its device */
int tamd_dev_sync_device(int device); /* every stream of that device */
void tamd_dev_free_on(int device, void * ptr);
/* a grow-only block of the calling thread (0: pager, 1: a stack's own tables) */
int tamd_dev_block(int which, void ** ptr, size_t bytes, int * grown);
#define TAMD_MAX_DEVICES 16
int tamd_dev_count(void);
int tamd_dev_select(int device);
int tamd_dev_current(void);
int tamd_dev_cus(void);
int tamd_dev_stream_set(void * stream);
int tamd_dev_sync(void);
void tamd_dev_math_set(int strict); /* 1: reference-order arithmetic in k_trace */
int tamd_dev_math_get(void);
void tamd_dev_in_flight_set(int batches); /* the batches the thread keeps in flight (a hint: device.hip) */
int tamd_dev_in_flight_get(void);

int tamd_dev_malloc(void ** ptr, size_t bytes);
void tamd_dev_free(void * ptr);
int tamd_dev_h2d(void * dst, const void * src, size_t bytes); /* stream-ordered, then synced */
int tamd_dev_d2h(void * dst, const void * src, size_t bytes);
int tamd_dev_zero(void * dst, size_t bytes);                   /* stream-ordered */
/* a pinned host buffer of the calling thread (grow-only), and copies between it and
 * HBM that are queued on the thread's stream and not waited for */
int tamd_dev_pinned(void ** ptr, size_t bytes);
int tamd_dev_copy_async(void * dst, const void * src, size_t bytes, int to_device);
/* page-locked host memory that outlives the call (tile staging) */
int tamd_dev_host_alloc(void ** ptr, size_t bytes);
void tamd_dev_host_free(void * ptr);

/* Grow-only scratch arena for HOST-space calls: reset at the start of each
 * API call, handed out in 256-byte aligned pieces. */
void tamd_scratch_reset(void);
int tamd_scratch_get(void ** ptr, size_t bytes);

/* One round of a batch call over a geometry with paged tiles: which items to
 * run (ids / n_in, NULL for all of 0 .. n-1) and where to list the ones that
 * met a tile that is not resident (faulted / n_faulted) and, over the tile
 * table, how much each tile is wanted.  All NULL: nothing is paged. */
/* The demand counters of the tiles are a line apart: a batch whose rays start over tiles
 * that are not resident adds to a handful of them millions of times in one pass, and in
 * one line those additions queue in ONE channel of the L2 -- behind them, the loads of
 * every other wave that go through that channel. */
#define TAMD_DEMAND_STRIDE 32
struct tamd_paging {
        const int * ids;
        const unsigned long long * n_in;
        int * faulted;
        unsigned long long * n_faulted;
        unsigned * wanted;       /* per entry of the tile table: how many listed items want it
                                  * (entry t at wanted[t * TAMD_DEMAND_STRIDE]: a cache line each) */
        unsigned * wanted_first; /* bitmap: wanted by the first item of the list (served without fail) */
        double * tentative;      /* traces: per ray, the step a waiting ray was about to take */
        int first_id;            /* the item whose wants go to wanted_first (-1: the first listed) */
};

/* Kernel launchers.  All pointers are DEVICE pointers; NULL output pointers
 * are allowed where the public API allows them. */
int tamd_k_ecef_from_geodetic(long n, const double * lat, const double * lon,
    const double * elev, double * ecef);
int tamd_k_ecef_to_geodetic(long n, const double * ecef, double * lat,
    double * lon, double * alt);
int tamd_k_ecef_from_horizontal(long n, const double * lat, const double * lon,
    const double * az, const double * el, double * dir);
int tamd_k_ecef_to_horizontal(long n, const double * lat, const double * lon,
    const double * dir, double * az, double * el);
/* elevation of n points on metas[0] of the view (a MAP or a STACK entry) */
int tamd_k_elevation(struct tamd_view view, long n, const double * a,
    const double * b, double * z, int * inside, struct tamd_paging pg);
/* gradient of n points on metas[0]: MAP (x, y) -> (gx, gy); STACK (lat, lon)
 * -> (glat, glon); ga/gb are in-out */
int tamd_k_gradient(struct tamd_view view, long n, const double * a,
    const double * b, double * ga, double * gb, int * inside, struct tamd_paging pg);
int tamd_k_position(struct tamd_view view, long n, const double * lat,
    const double * lon, const double * height, int layer, double * pos,
    int * data_index, struct tamd_paging pg);
int tamd_k_step(struct tamd_view view, long n, double * pos,
    const double * dir, double * lat, double * lon, double * alt,
    double * elev, double * step, int * index, int flags, struct tamd_paging pg);
/* stats: 4 x uint64 on the device (rays, steps, samples, capped); queue:
 * TAMD_TRACE_COUNTERS x uint64 (work queues and list lengths of the passes of a
 * fast trace: see run_trace in device.hip); both zeroed by the launcher.
 * parked: int[3 n] and cross_ds: double[n], scratch for the lists of rays handed
 * from pass to pass (the long rays; the rays that crossed a boundary, for
 * k_cross), or NULL for a single-pass launch that bisects in place; with them,
 * length and n_steps must not be NULL.  With TAMD_TRACE_SORT_ROOM in `flags`,
 * `parked` has room for TAMD_TRACE_SORT_INTS x n ints and TAMD_TRACE_SORT_TEMP
 * bytes more behind them: the hand-over list is then ORDERED before the lined
 * pass reads it (run_trace in device.hip). */
#define TAMD_TRACE_COUNTERS 96
#define TAMD_TRACE_SORT_ROOM 0x100
/* a flag of the step kernels beside enum turtle_amd_step_flags: `alt` holds the tentative length
 * of the next step instead of the altitude, `elev` is not used (turtle_stepper_walk_n) */
#define TAMD_STEP_COMPACT 0x200
#define TAMD_TRACE_SORT_INTS 7
#define TAMD_TRACE_SORT_TEMP ((size_t)32 << 20)
/* ... and, behind those, TAMD_TRACE_COPY_BYTES x n bytes for the rays themselves in the order
 * the trace takes them (position, direction, index, path length, step count) */
#define TAMD_TRACE_COPY_BYTES 72
int tamd_k_trace(struct tamd_view view, long n, double * pos,
    const double * dir, int max_steps, int * index, double * length,
    int * n_steps, int flags, int * parked, double * cross_ds, struct tamd_paging pg,
    unsigned long long * stats, unsigned long long * queue);
/* n single steps with a direction: the step kernel lists the rays that crossed
 * a boundary (cross_ray / cross_ds: scratch for n entries each, or NULL to
 * bisect in place) and a second kernel bisects them, packed; `flags` are enum
 * turtle_amd_step_flags */
int tamd_k_step_dir(struct tamd_view view, long n, double * pos,
    const double * dir, double * lat, double * lon, double * alt,
    double * elev, double * step, int * index, int flags, int * cross_ray,
    double * cross_ds, struct tamd_paging pg, unsigned long long * stats,
    unsigned long long * queue);
/* one generation of turtle_stepper_scatter_n: single steps resumed from the
 * sample in alt / elev / index, directions drawn in the kernels from Philox(first
 * + ray, stream; seed), the step added to length[] and steps[]; cross_ray /
 * cross_ds as for tamd_k_step_dir (not NULL); stats are NOT zeroed */
int tamd_k_step_walk(struct tamd_view view, long n, double * pos, double * alt,
    double * elev, int * index, unsigned long long seed, unsigned long long stream, long first,
    double * length, int * steps, int * cross_ray, double * cross_ds, struct tamd_paging pg,
    unsigned long long * stats, unsigned long long * queue);
/* the same walk, all its generations in one launch, the rays' state in registers
 * (every tile resident); stats are NOT zeroed */
int tamd_k_walk(struct tamd_view view, long n, double * pos, double * alt, double * elev,
    int * index, unsigned long long seed, long first, int first_step, int n_steps, double * length,
    int * steps, unsigned long long * stats, unsigned long long * queue);
int tamd_k_philox(long n, unsigned long long seed, unsigned long long stream,
    long first, unsigned * out);
int tamd_k_isotropic(long n, unsigned long long seed, unsigned long long stream,
    long first, double * dir);
/* forward (inverse == 0: lat, lon -> x, y) or inverse projection of n points */
int tamd_k_project(struct tamd_proj proj, int inverse, long n, const double * a,
    const double * b, double * c, double * d);
int tamd_k_tally(long n, const int * index, const double * length,
    int n_media, unsigned long long * hits, int n_bins, double length_max,
    unsigned long long * histogram);

#ifdef __cplusplus
}
#endif
#endif
