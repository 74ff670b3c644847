/*
 * host.h -- host-side objects behind the opaque handles of turtle_amd.h.
 * Plain C99; the structures here never reach the device (internal.h has the
 * POD tables that do).
 */
#ifndef TURTLE_AMD_HOST_H
#define TURTLE_AMD_HOST_H

#include "internal.h"

/* ---- error context [ref src/turtle/error.h:31-47] --------------------- */
struct tamd_error {
        enum turtle_return code;
        turtle_function_t * function;
};

#define TAMD_ERROR_INIT(fn)                                                    \
        struct tamd_error error_ = { TURTLE_RETURN_SUCCESS,                    \
                (turtle_function_t *)(fn) }

/* Record + raise in one go: formats "{ fn [#code], file:line } text" and calls
 * the installed handler once [ref src/turtle/error.c:108-138]. */
enum turtle_return tamd_raise_(struct tamd_error * error, enum turtle_return rc,
    const char * file, int line, const char * format, ...);
#define TAMD_RAISE(rc, ...) tamd_raise_(&error_, (rc), __FILE__, __LINE__, __VA_ARGS__)
/* device-layer failure => LIBRARY_ERROR carrying the HIP message */
#define TAMD_RAISE_DEVICE()                                                    \
        TAMD_RAISE(TURTLE_RETURN_LIBRARY_ERROR, "device error: %s", tamd_dev_error())

/* ---- projections [ref src/turtle/projection.h:38-46] ---------------------- */
struct turtle_projection {
        int type; /* enum tamd_proj_type */
        int lambert_tag;
        double longitude_0;
        int hemisphere;
        char tag[64];
};
int tamd_projection_configure(struct turtle_projection * projection, const char * name,
    char * message, size_t size);
void tamd_projection_desc(const struct turtle_projection * projection, struct tamd_proj * desc);

/* ---- maps ---------------------------------------------------------------- */
struct turtle_map {
        /* meta data [ref src/turtle/map.h:41-56] */
        int nx, ny;
        double x0, y0, z0;
        double dx, dy, dz;
        char encoding[8];
        int is_signed;        /* int16 codecs (hgt): z = (int16)v */
        struct turtle_projection projection; /* type < 0: geodetic */
        struct turtle_stack * stack; /* owner, or NULL */

        uint16_t * nodes;     /* host copy: native endian, rows south->north -- or NULL for a
                               * tile that came back from a staging buffer (stage_tile): the
                               * kernels read the HBM copy; the host reads `lazy_path` when it
                               * first needs the nodes (tamd_map_host_nodes) */
        const char * lazy_path; /* the tile's file (the stack's string), or NULL */
        void * d_nodes[TAMD_MAX_DEVICES]; /* HBM copies (in blocks: internal.h), one per device */
        unsigned d_fresh;     /* bit d: the copy on device d is current */
        /* a tile of a stack, just read: its nodes in the HBM layout, in one of the stack's
         * page-locked staging buffers (slot `staged_slot`), waiting for their first upload */
        uint16_t * staged;
        int staged_slot;
};

/* What threads share -- the HBM copies of a map, the tiles of a stack, the epoch
 * -- changes under this (recursive) lock.  Launches do not take it: a thread reads
 * the geometry through tables of its own (its stepper's, or its block 1), built
 * under the lock from tiles that stay until every stream of the device has
 * drained (tamd_map_release). */
void tamd_geometry_lock(void);
void tamd_geometry_unlock(void);
unsigned long tamd_geometry_epoch_get(void);
void tamd_geometry_changed(void);
/* A thread holds the geometry IN USE from the moment it builds its tables until
 * the launches that read them are queued (shared with other users; exclusive
 * against tamd_map_release).  Not held across anything that may free tiles. */
void tamd_geometry_use_begin(void);
void tamd_geometry_use_end(void);
/* ... and whoever may free tiles or maps holds it exclusively, plus the lock */
void tamd_geometry_write_begin(void);
void tamd_geometry_write_end(void);
/* frees the HBM copies of a map (inside write_begin / _end) once what is queued on
 * their devices has run */
void tamd_map_release(struct turtle_map * map);

/* the host copy of the nodes is there (a tile that came without: read now); 0, or an
 * enum turtle_return */
int tamd_map_host_nodes(struct turtle_map * map);
/* fills the decode parameters of `grid` and makes the HBM copy current */
int tamd_map_sync(struct turtle_map * map, struct tamd_grid * grid);
enum turtle_return tamd_map_load_(struct turtle_map ** map, const char * path,
    struct tamd_error * error, const char * file, int line);
/* Codecs (hgt.c, tiff.c): a header-only probe that fills the meta data, and a
 * full read into map->nodes (native endian, rows south->north).  Both return
 * an enum turtle_return; BAD_FORMAT + 100 stands for "missing data", + 101 for
 * "inconsistent data", + 102 for "could not read the header". */
int tamd_hgt_probe(const char * path, struct turtle_map * meta);
int tamd_hgt_read(const char * path, struct turtle_map * map);
int tamd_hgt_read_rows(const char * path, struct turtle_map * map, int iy0, int iy1);
int tamd_tiff_probe(const char * path, struct turtle_map * meta);
int tamd_tiff_read(const char * path, struct turtle_map * map);
int tamd_tiff_read_rows(const char * path, struct turtle_map * map, int iy0, int iy1);
int tamd_png_probe(const char * path, struct turtle_map * meta);
int tamd_png_read(const char * path, struct turtle_map * map);
int tamd_grd_probe(const char * path, struct turtle_map * meta);
int tamd_grd_read(const char * path, struct turtle_map * map);
int tamd_asc_probe(const char * path, struct turtle_map * meta);
int tamd_asc_read(const char * path, struct turtle_map * map);
/* Extension dispatch [ref src/turtle/io.c:60-104]: 0 if no codec handles it */
int tamd_codec_for(const char * path, int (**probe)(const char *, struct turtle_map *),
    int (**read)(const char *, struct turtle_map *));

/* ---- stacks -------------------------------------------------------------- */
struct turtle_stack {
        int max_size;
        turtle_stack_locker_t * lock, * unlock;
        double latitude_0, latitude_delta;
        double longitude_0, longitude_delta;
        int latitude_n, longitude_n;
        char * root;
        char ** path;             /* [lat_n * long_n] file of each slot or NULL */
        struct turtle_map ** tile; /* [lat_n * long_n] loaded tile or NULL */
        unsigned long * stamp;    /* [lat_n * long_n] when the tile was last wanted */
        const void ** owner;      /* [lat_n * long_n] the thread that has just paged the tile in and
                                   * has not run its round over it yet (or NULL): nobody else's
                                   * page-in or trim takes it away meanwhile, as the reference's
                                   * clients pin the tile they use [ref stack.c:433-442] */
        unsigned long clock;
        int n_loaded, n_files;
        /* Tiles come in through page-locked staging buffers (tiles.c): worker threads read and
         * decode the files of a round side by side, straight into the HBM layout; the upload is
         * a queued copy.  A slot is free again once the device it was copied to has drained
         * (`stage_device`: -1 free, -2 holds a tile not uploaded yet, d >= 0 a copy to
         * device d was queued and could not be waited for: free once d has drained).
         * What a stack keeps for its lifetime: up to 16 page-locked buffers and, per
         * device, up to 16 HBM buffers of one tile each -- 26 MB apiece for 3601 x 3601
         * tiles, 0.4 GB of either at most; turtle_stack_clear / _destroy give them back. */
#define TAMD_STAGE_SLOTS 16
        uint16_t * stage[TAMD_STAGE_SLOTS];
        int stage_device[TAMD_STAGE_SLOTS];
        size_t stage_bytes;
        /* A buffer still holds a tile's nodes, laid out for HBM, after its upload -- and
         * after the tile has left the stack: a tile that COMES BACK while they are there
         * (a batch over a stack smaller than its ground goes to and fro between two sets
         * of tiles, round after round) is not read from its file again: its nodes are
         * copied back from the buffer and uploaded.  `stage_tile`: the directory slot of
         * that tile (-1: none), with the size and time stamp its file had when it was read
         * (a file that has changed is read again); `stage_stamp`: when the buffer was last
         * filled or found -- the least recent one is overwritten first.  No memory beyond
         * the buffers themselves. */
        int stage_tile[TAMD_STAGE_SLOTS];
        long long stage_file_size[TAMD_STAGE_SLOTS], stage_file_time[TAMD_STAGE_SLOTS];
        unsigned long stage_stamp[TAMD_STAGE_SLOTS];
        /* HBM buffers of tiles that went, kept for the tiles that come (one size: the tiles of
         * a stack have one shape): hipMalloc / hipFree wait for the device */
#define TAMD_SPARE_HBM 32
        void * spare[TAMD_MAX_DEVICES][TAMD_SPARE_HBM];
        int n_spare[TAMD_MAX_DEVICES];
        size_t spare_bytes;
};

/* tiles.c: the files of `n` tiles read and decoded by up to `threads` workers */
struct tamd_tile_job {
        const char * path;
        uint16_t * staged;        /* in: a staging buffer of `staged_bytes`, or NULL */
        size_t staged_bytes;
        int cached;               /* in: `staged` holds this tile's nodes already (see stage_tile):
                                   * the file's header is read, its nodes are not */
        struct turtle_map * map;  /* out: the tile (calloc'ed; nodes malloc'ed), or NULL */
        int rc;                   /* out: an enum turtle_return */
};
void tamd_tiles_decode(struct tamd_tile_job * jobs, int n);
/* stack.c: a spare HBM buffer of that size on that device (or NULL); a tile's staging buffer
 * has been copied from on `device` (-1: it was not, and is free) */
void * tamd_stack_spare_take(struct turtle_stack * stack, int device, size_t bytes);
void tamd_stack_staged_done(struct turtle_map * tile, int device);
size_t tamd_blocked_bytes(int nx, int ny);
void tamd_blocked_fill(const struct turtle_map * map, uint16_t * blocked);
void tamd_blocked_fill_rows(const struct turtle_map * map, uint16_t * blocked, int iy0, int iy1);

/* Tiles the stack keeps in memory between calls [ref stack.c:150]: max_size, or
 * no limit; tamd_stack_trim brings it back there when a batch call ends (while
 * it runs, the tiles its first waiting item needs stay, whatever their number) */
int tamd_stack_budget(const struct turtle_stack * stack);
void tamd_stack_trim(struct turtle_stack * stack);
/* Are there tiles with a file that are not in memory? */
int tamd_stack_is_paged(const struct turtle_stack * stack);
/* [ref stack.c:257-297] tiles in directory order until the budget is reached;
 * 0 on success, else an enum turtle_return with a message in `message` */
int tamd_stack_preload(struct turtle_stack * stack, char * message, size_t size);
/* Bring in tiles that a round wanted: wanted[first_bit + slot] = how many of
 * its listed items want the tile, resident or not.  Beyond the budget a tile in
 * less demand makes room -- the least recently wanted among those nobody wants
 * [ref stack.c:433-443].  The tiles of the bitmap `wanted_first` (the first
 * item of the list: at most a 3 x 3 neighbourhood and one more) come in without
 * fail, whatever has to go.  Returns the number of tiles loaded, or minus an
 * enum turtle_return with `message` set. */
/* `few`: the round left only a handful of items waiting (TAMD_PAGING_FEW): their tiles come in
 * beyond the stack's size, by up to that size again or TAMD_PAGING_SLACK tiles, whichever is less */
/* (diagnostics, TURTLE_AMD_PAGING_TRACE: tiles that came from a staging buffer, in all) */
extern unsigned long tamd_stack_buffer_hits;
#define TAMD_PAGING_FEW 4096
#define TAMD_PAGING_SLACK 8
int tamd_stack_page_in(struct turtle_stack * stack, const unsigned * wanted,
    const unsigned * wanted_first, int first_bit, int few, char * message, size_t size);

struct turtle_client {
        struct turtle_stack * stack;
};

/* ---- stepper ------------------------------------------------------------- */
struct tamd_data {
        int kind;                    /* enum tamd_kind */
        struct turtle_map * map;
        struct turtle_stack * stack;
        struct turtle_client * client; /* owned, when the stack has a lock */
};

struct tamd_layer_meta {
        int data; /* index into stepper->data */
        double offset;
};

struct tamd_layer {
        struct tamd_layer_meta * meta; /* in the order they were added */
        int size, capacity;
};

/* the reference's `last` sample [ref src/turtle/stepper.h:93-98], for the scalar calls
 * answered on the host (scalar.c) */
struct tamd_host_sample {
        int valid;
        double position[3];
        double latitude, longitude, altitude;
        double elevation[2];
        int index[2];
};

struct turtle_stepper {
        struct tamd_host_sample last;
        struct tamd_data * data;
        int n_data, cap_data;
        struct tamd_layer * layers;
        int n_layers, cap_layers;
        struct turtle_map * geoid;
        double local_range, slope_factor, resolution_factor;

        /* device tables (rebuilt when the global geometry epoch moves, or the
         * stepper's thread has moved to another device) */
        unsigned long epoch;
        int device;                   /* where the tables and the scratch below are (-1: nowhere) */
        void * d_tables;
        size_t d_tables_size;
        struct tamd_view view;
        int n_table;                  /* entries of the tile table (all stacks) */
        unsigned long long * d_stats; /* 4 stats + the TAMD_TRACE_COUNTERS of a trace */
        int * d_parked;               /* scratch of the batch calls: ray ids ... */
        double * d_scratch_ds;        /* ... and one double each (same block) */
        long parked_capacity;
        int last_rounds;              /* rounds the last batch call took (1: nothing was paged in) */
};

/* Any change to what kernels may read (map nodes, tiles, layers) bumps the epoch
 * (tamd_geometry_changed). */

/* Builds/refreshes stepper->view; returns an enum turtle_return and a message */
int tamd_stepper_flatten(struct turtle_stepper * stepper, char * message, size_t size);

/* ---- rounds of a batch call over paged stacks (paging.c) ------------------- */
struct tamd_pager {
        int active, rounds;
        int * d_list[2];
        unsigned long long * d_count;
        unsigned * d_wanted;
        unsigned * wanted; /* host copies after tamd_pager_collect: demand per tile ... */
        unsigned * wanted_first; /* ... and the bitmap of the first item's tiles */
        unsigned * pinned; /* every tile the item served without fail has asked for so far */
        int first_id;      /* that item (-1: the first of the next list) */
        size_t words, first_offset, bitmap_words;
};
int tamd_pager_begin(struct tamd_pager * pager, long n, int table_entries);
int tamd_pager_round(struct tamd_pager * pager, struct tamd_paging * pg);
int tamd_pager_collect(struct tamd_pager * pager, unsigned long long * n_faulted);
void tamd_pager_end(struct tamd_pager * pager);
#define TAMD_PAGING_ROUNDS 100000 /* a batch needs about one round per tile it touches */

/* ---- HOST/DEVICE array staging for the batch calls ----------------------- */
#define TAMD_STAGE_PENDING 12
struct tamd_stage {
        int space;
        int packed;          /* small HOST call: through the thread's pinned buffer (stage.c) */
        int n_pending;
        char * pinned;
        size_t pinned_used;
        struct {
                void * user;
                const void * dev;
                size_t bytes;
        } pending[TAMD_STAGE_PENDING]; /* outputs to bring back: tamd_stage_end does, in one copy */
};
int tamd_stage_begin(struct tamd_stage * st, int space, size_t total_bytes);
/* returns the device address to use for a user array (NULL stays NULL) */
int tamd_stage_in(struct tamd_stage * st, const void * user, size_t bytes, void ** dev);
int tamd_stage_out(struct tamd_stage * st, void * user, size_t bytes, void ** dev);
int tamd_stage_fetch(struct tamd_stage * st, void * user, size_t bytes, const void * dev);
int tamd_stage_end(struct tamd_stage * st);

/* ---- the scalar entry points on the host (scalar.c; opt-in) ------------------- */
int tamd_scalar_on_host(void);
void tamd_h_from_geodetic(double latitude, double longitude, double elevation, double ecef[3]);
void tamd_h_to_geodetic(const double ecef[3], double * latitude, double * longitude, double * altitude);
void tamd_h_from_horizontal(double latitude, double longitude, double azimuth, double elevation,
    double direction[3]);
void tamd_h_to_horizontal(double latitude, double longitude, const double direction[3], double * azimuth,
    double * elevation);
int tamd_h_map_elevation(const struct turtle_map * map, double x, double y, double * z);
int tamd_h_stack_elevation(struct turtle_stack * stack, double latitude, double longitude, double * z,
    int * inside, char * message, size_t size);
int tamd_h_stepper_takes(const struct turtle_stepper * stepper);
int tamd_h_stepper_step(struct turtle_stepper * stepper, double * position, const double * direction,
    double * latitude, double * longitude, double * altitude, double * elevation, double * step_length,
    int * index, char * message, size_t size);
int tamd_h_stepper_position(struct turtle_stepper * stepper, double latitude, double longitude,
    double height, int layer_index, double * position, int * data_index, char * message, size_t size);
/* stack.c: one tile into memory for the host path, the least recently used going beyond the
 * stack's size [ref stack.c:399-450]; an enum turtle_return */
int tamd_stack_host_fetch(struct turtle_stack * stack, int slot, double latitude, double longitude,
    double * z, int * inside, char * message, size_t size);

#endif
