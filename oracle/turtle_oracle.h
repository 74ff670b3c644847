/*
 * turtle_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (C99, libm only) restatement of the reference's ray/terrain stepper
 * hot path, written from scratch for use as (1) the parity checker of the HIP
 * path and (2) the timed CPU baseline on the GPU box, where the reference
 * itself (/root/reference) does not exist.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under turtle_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below against tests/golden/ fixtures, which tests/golden/generate.py produced by
 * driving the real reference (oracle/_ref/libturtle_ref.so, compiled from
 * /root/reference by oracle/Makefile) on the same inputs; when oracle/_ref is
 * present the same tests also compare call-by-call against it.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef TURTLE_ORACLE_H
#define TURTLE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* How a grid's 16-bit payload is to be read (the reference defers decoding to
 * a per-codec get_z callback, src/turtle/map.h:47-49). */
enum orc_layout {
        /* turtle_map_create grids: row iy as stored (south->north), native
         * uint16, z = z0 + v*dz            (src/turtle/map.c:41-44) */
        ORC_LAYOUT_DEFAULT = 0,
        /* raw .hgt payload: big-endian int16, file rows north->south,
         * z = (int16)ntohs(v)              (src/turtle/io/hgt.c:127-131) */
        ORC_LAYOUT_HGT = 1
};

/* src/turtle/projection.h:29-46 */
enum orc_projection { ORC_PROJ_NONE = -1, ORC_PROJ_LAMBERT = 0, ORC_PROJ_UTM = 1 };

struct orc_proj {
        int type;           /* enum orc_projection */
        int lambert_tag;    /* 0..5: I, II, IIe, III, IV, 93 */
        double longitude_0; /* UTM central meridian */
        int hemisphere;     /* UTM: +1 north, -1 south */
};

struct orc_grid {
        int nx, ny;
        double x0, y0, dx, dy, z0, dz;
        int layout;            /* enum orc_layout */
        const uint16_t * data; /* nx*ny raw nodes, as the reference stores them */
        struct orc_proj proj;  /* type < 0: a geodetic grid (x = lon, y = lat) */
};

/* A tile directory with every tile resident (src/turtle/stack.h:32-49). */
struct orc_stack {
        double lat0, lon0, dlat, dlon;
        int nlat, nlon;
        const int * tile; /* [nlat*nlon] -> grid index, or -1 if no file */
};

enum orc_kind { ORC_FLAT = 0, ORC_MAP = 1, ORC_STACK = 2 };

struct orc_meta {
        int kind;      /* enum orc_kind */
        int src;       /* grid index (ORC_MAP) or stack index (ORC_STACK) */
        double offset; /* src/turtle/stepper.h:80-85 */
};

/* Flattened geometry: layers bottom->top, and inside a layer the metas in
 * the reference's *iteration* order, i.e. last added first
 * (src/turtle/stepper.c:719-724). */
struct orc_geometry {
        int n_layers;
        const int * layer_first;      /* n_layers+1 offsets into metas */
        const struct orc_meta * metas;
        const struct orc_grid * grids;
        const struct orc_stack * stacks;
        int geoid;                    /* grid index or -1 (stepper.c:42-50) */
};

/* src/turtle/stepper.h:93-98 */
struct orc_sample {
        double position[3];
        double geographic[3]; /* lat, lon, alt */
        double elevation[2];
        int index[2];
};

/* src/turtle/stepper.h:45-58 (the single "geodetic" transform) and :101-110 */
struct orc_stepper {
        const struct orc_geometry * geometry;
        double local_range, slope_factor, resolution_factor;
        struct orc_sample last;
        /* Local linear approximation state (stepper.c:85-171) */
        double reference_ecef[3];
        double reference_geographic[3];
        double jacobian[3][3];
        /* counters (not in the reference): exact transforms and samples */
        long n_transforms, n_samples;
};

/* ecef.c:41-55, :63-130, :160-176, :178-207 */
void orc_ecef_from_geodetic(double latitude, double longitude,
    double elevation, double ecef[3]);
void orc_ecef_to_geodetic(const double ecef[3], double * latitude,
    double * longitude, double * altitude);
void orc_ecef_from_horizontal(double latitude, double longitude,
    double azimuth, double elevation, double direction[3]);
void orc_ecef_to_horizontal(double latitude, double longitude,
    const double direction[3], double * azimuth, double * elevation);

/* projection.c:192-230, :286-295, :377-408 (forward), :304-318, :417-448 (inverse) */
void orc_project(const struct orc_proj * proj, double latitude, double longitude,
    double * x, double * y);
void orc_unproject(const struct orc_proj * proj, double x, double y,
    double * latitude, double * longitude);
void orc_project_n(const struct orc_proj * proj, long n, const double * latitude,
    const double * longitude, double * x, double * y);
void orc_unproject_n(const struct orc_proj * proj, long n, const double * x,
    const double * y, double * latitude, double * longitude);

/* map.c:229-277; returns inside (0/1) */
int orc_grid_elevation(const struct orc_grid * grid, double x, double y,
    double * z);
/* map.c:280-378 (as is, slip at :353 included); gx/gy are in-out */
int orc_grid_gradient(const struct orc_grid * grid, double x, double y,
    double * gx, double * gy);
void orc_grid_gradient_n(const struct orc_grid * grid, long n,
    const double * x, const double * y, double * gx, double * gy, int * inside);
/* map.c:208-226 (node value only) */
double orc_grid_node(const struct orc_grid * grid, int ix, int iy);
/* stack.c:300-361 + :399-450 with all tiles resident; returns inside */
int orc_stack_elevation(const struct orc_geometry * geometry,
    const struct orc_stack * stack, double latitude, double longitude,
    double * z);

/* stepper.c:547-570 defaults; :617-672 setters are plain field writes */
void orc_stepper_init(
    struct orc_stepper * stepper, const struct orc_geometry * geometry);
void orc_stepper_reset(struct orc_stepper * stepper);
/* stepper.c:780-875; every output pointer may be NULL; returns 0 or
 * TURTLE_RETURN_DOMAIN_ERROR (6) exactly where the reference does */
int orc_stepper_step(struct orc_stepper * stepper, double * position,
    const double * direction, double * latitude, double * longitude,
    double * altitude, double * elevation, double * step_length, int * index);
/* stepper.c:877-931 */
int orc_stepper_position(struct orc_stepper * stepper, double latitude,
    double longitude, double height, int layer_index, double * position,
    int * data_index);

/* Batch drivers (the harness loop of examples/example-stepper.c:128-140 and
 * SURVEY 8d): for each ray, sample at the origin, then step until index[0]
 * differs from its initial value or max_steps is reached.  `range` is the
 * local_range to use (0 = exact transform every sample).  Runs on `threads`
 * pthreads, one stepper per thread, rays block-partitioned.  Outputs are
 * per ray; any may be NULL except none.  Returns total steps. */
long orc_trace_n(const struct orc_geometry * geometry, double slope,
    double resolution, double range, long n, double * position /*[n][3] io*/,
    const double * direction /*[n][3]*/, int max_steps, int * index /*[n][2]*/,
    double * length /*[n]*/, int * n_steps /*[n]*/, int threads,
    long * n_samples /* optional: total samples */,
    long * n_transforms /* optional: total exact ECEF->geodetic transforms */);

/* One turtle_stepper_step per ray with a fresh stepper history each
 * (direction may be NULL => sample only).  Arrays are per ray; NULL skips. */
void orc_step_n(const struct orc_geometry * geometry, double slope,
    double resolution, long n, double * position, const double * direction,
    double * latitude, double * longitude, double * altitude,
    double * elevation /*[n][2]*/, double * step_length, int * index /*[n][2]*/);

void orc_position_n(const struct orc_geometry * geometry, long n,
    const double * latitude, const double * longitude, const double * height,
    int layer_index, double * position /*[n][3]*/, int * data_index);

void orc_ecef_to_geodetic_n(long n, const double * ecef, double * latitude,
    double * longitude, double * altitude);
void orc_ecef_from_geodetic_n(long n, const double * latitude,
    const double * longitude, const double * elevation, double * ecef);
void orc_ecef_from_horizontal_n(long n, const double * latitude,
    const double * longitude, const double * azimuth, const double * elevation,
    double * direction);
void orc_ecef_to_horizontal_n(long n, const double * latitude,
    const double * longitude, const double * direction, double * azimuth,
    double * elevation);
void orc_grid_elevation_n(const struct orc_grid * grid, long n,
    const double * x, const double * y, double * z, int * inside);
void orc_stack_elevation_n(const struct orc_geometry * geometry, int stack,
    long n, const double * latitude, const double * longitude, double * z,
    int * inside);

#ifdef __cplusplus
}
#endif
#endif
