/*
 * turtle_amd.h -- C ABI of libturtle_amd.so, an MI355X-native implementation
 * of TURTLE's optimistic ray/terrain stepper path.
 *
 * Two groups of entry points:
 *
 *  (1) DROP-IN: the reference's own public functions for this path, with the
 *      same names, argument meaning, ownership and error convention, so that a
 *      caller written against the reference's `turtle.h` links unchanged
 *      (`#include "turtle.h"` from this directory forwards here).  Each
 *      declaration cites the reference declaration it replaces as
 *      [ref include/turtle.h:LINE] and the implementation it restates as
 *      [impl src/turtle/FILE.c:LINE] (paths under the reference tree).
 *      These are scalar calls: each one runs the same device kernels as the
 *      batch form with n = 1 and therefore costs one kernel launch -- unless
 *      the caller asks for them to be answered on the host
 *      (turtle_amd_scalar_set: a restatement of the reference's one-point
 *      functions for callers that keep its per-ray loop; off by default).
 *      Without a usable gfx950 device every computing entry point fails with
 *      TURTLE_RETURN_LIBRARY_ERROR through the error handler, whatever that
 *      option says: nothing in this library stands in for a missing GPU.
 *
 *  (2) BATCH EXTENSION (suffix _n, prefix turtle_amd_): the same operations on
 *      n independent rays/points per call, which is what a GPU is for.  The
 *      reference has no such entry points; INTEGRATION.md shows the few lines
 *      a maintainer adds to bind them.  Arrays are plain C arrays, one element
 *      (or one double[3] / double[2] / int[2] group) per ray, in HOST or DEVICE
 *      memory according to `space`.
 *
 * Units and conventions are the reference's: degrees, metres, WGS84 ECEF.
 */
#ifndef TURTLE_AMD_H
#define TURTLE_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef TURTLE_API
#define TURTLE_API
#endif

/* ---- return codes and handles [ref include/turtle.h:35-62, :64-106] ---- */
enum turtle_return {
        TURTLE_RETURN_SUCCESS = 0,
        TURTLE_RETURN_BAD_ADDRESS,
        TURTLE_RETURN_BAD_EXTENSION,
        TURTLE_RETURN_BAD_FORMAT,
        TURTLE_RETURN_BAD_PROJECTION,
        TURTLE_RETURN_BAD_JSON,
        TURTLE_RETURN_DOMAIN_ERROR,
        TURTLE_RETURN_LIBRARY_ERROR,
        TURTLE_RETURN_LOCK_ERROR,
        TURTLE_RETURN_MEMORY_ERROR,
        TURTLE_RETURN_PATH_ERROR,
        TURTLE_RETURN_UNLOCK_ERROR,
        N_TURTLE_RETURNS
};

struct turtle_projection;
struct turtle_map;
struct turtle_stack;
struct turtle_client;
struct turtle_stepper;

/* [ref include/turtle.h:93-106] */
struct turtle_map_info {
        int nx, ny;           /* nodes along x and y */
        double x[2];          /* x range (longitude for geodetic maps) */
        double y[2];          /* y range (latitude) */
        double z[2];          /* elevation range spanned by the 16-bit code */
        const char * encoding;
};

/* ---- error handling [ref include/turtle.h:114-194; impl error.c:28-198] ----
 * Every enum-returning function returns TURTLE_RETURN_SUCCESS or a code and,
 * unless the handler is NULL, calls the handler once with a message shaped
 * "{ <function> [#<code>], <file>:<line> } <text>".  The default handler prints
 * the message to stderr and exit(EXIT_FAILURE)s, as the reference's does. */
typedef void turtle_function_t(void);
typedef void turtle_error_handler_t(enum turtle_return code,
    turtle_function_t * function, const char * message);
typedef int turtle_stack_locker_t(void); /* [ref include/turtle.h:149] */

TURTLE_API const char * turtle_error_function(turtle_function_t * function);
TURTLE_API turtle_error_handler_t * turtle_error_handler_get(void);
TURTLE_API void turtle_error_handler_set(turtle_error_handler_t * handler);

/* ---- ECEF transforms [ref include/turtle.h:556-602; impl ecef.c:41-207] ---- */
TURTLE_API void turtle_ecef_from_geodetic(
    double latitude, double longitude, double elevation, double ecef[3]);
TURTLE_API void turtle_ecef_to_geodetic(const double ecef[3], double * latitude,
    double * longitude, double * altitude);
TURTLE_API void turtle_ecef_from_horizontal(double latitude, double longitude,
    double azimuth, double elevation, double direction[3]);
TURTLE_API void turtle_ecef_to_horizontal(double latitude, double longitude,
    const double direction[3], double * azimuth, double * elevation);

/* ---- projections [ref include/turtle.h:235-333; impl projection.c:53-468] ----
 * Names as in the reference: "Lambert I|II|IIe|III|IV|93", "UTM <zone>N|S",
 * "UTM <central meridian>.<fraction>N|S". */
TURTLE_API enum turtle_return turtle_projection_create(
    struct turtle_projection ** projection, const char * name);
TURTLE_API void turtle_projection_destroy(
    struct turtle_projection ** projection);
TURTLE_API enum turtle_return turtle_projection_configure(
    struct turtle_projection * projection, const char * name);
TURTLE_API const char * turtle_projection_name(
    const struct turtle_projection * projection);
TURTLE_API enum turtle_return turtle_projection_project(
    const struct turtle_projection * projection, double latitude,
    double longitude, double * x, double * y);
TURTLE_API enum turtle_return turtle_projection_unproject(
    const struct turtle_projection * projection, double x, double y,
    double * latitude, double * longitude);

/* ---- maps [ref include/turtle.h:362-543; impl map.c:54-421] ----
 * `projection` is NULL for a geodetic grid (x = longitude, y = latitude) or a
 * projection name; under a stepper a projected map is looked up at the
 * projected coordinates [impl stepper.c:65-83, :243-248].
 * turtle_map_load reads .hgt tiles [impl io/hgt.c:45-151] and uncompressed,
 * stripped GeoTIFF-16 .tif files [impl io/geotiff16.c:165-258] with a native
 * reader (no libtiff), and the reference's own .png map format (16-bit
 * greyscale + JSON "topography" header, incl. its projection) [impl
 * io/png16.c:183-448] with a native reader (zlib's inflate only), and the
 * text formats .grd / .asc [impl io/grd.c, io/asc.c]; other
 * extensions return TURTLE_RETURN_BAD_EXTENSION, compressed or tiled TIFFs and
 * non-16-bit or interlaced PNGs TURTLE_RETURN_BAD_FORMAT. */
TURTLE_API enum turtle_return turtle_map_create(struct turtle_map ** map,
    const struct turtle_map_info * info, const char * projection);
TURTLE_API void turtle_map_destroy(struct turtle_map ** map);
TURTLE_API enum turtle_return turtle_map_load(
    struct turtle_map ** map, const char * path);
/* [ref include/turtle.h:401-424; impl map.c:165-180] Writes the map as .png
 * (the reference's map format, projection included) or as .tif (GeoTIFF-16:
 * maps with z scale [-32767, 32768] and no projection), without libpng /
 * libtiff; .hgt / .grd / .asc return TURTLE_RETURN_BAD_FORMAT, anything else
 * TURTLE_RETURN_BAD_EXTENSION, as in the reference. */
TURTLE_API enum turtle_return turtle_map_dump(
    const struct turtle_map * map, const char * path);
TURTLE_API enum turtle_return turtle_map_fill(
    struct turtle_map * map, int ix, int iy, double elevation);
TURTLE_API enum turtle_return turtle_map_node(const struct turtle_map * map,
    int ix, int iy, double * x, double * y, double * elevation);
TURTLE_API enum turtle_return turtle_map_elevation(
    const struct turtle_map * map, double x, double y, double * elevation,
    int * inside);
/* [ref include/turtle.h:515-517; impl map.c:280-392].  gx, gy are in-out: the
 * reference leaves them untouched outside the map and, by a slip at
 * map.c:352-353 that is reproduced, stores the y-gradient in gx and leaves gy
 * untouched for a point in the grid's first half-row. */
TURTLE_API enum turtle_return turtle_map_gradient(
    const struct turtle_map * map, double x, double y, double * gx, double * gy,
    int * inside);
TURTLE_API const struct turtle_projection * turtle_map_projection(
    const struct turtle_map * map);
TURTLE_API void turtle_map_meta(const struct turtle_map * map,
    struct turtle_map_info * info, const char ** projection);

/* ---- tile stacks [ref include/turtle.h:637-719; impl stack.c:46-450] ----
 * As in the reference a stack keeps at most `stack_size` tiles in memory (no
 * limit if <= 0), loads a tile when a query first needs it and drops the least
 * recently used one to make room [ref stack.c:150, :399-450].  Here "memory" is
 * HBM and a query is a batch: the kernels list the rays / points that met a
 * tile that is not resident, the host pages those tiles in and the list runs
 * again, until it is empty -- results are those of a stack with every tile
 * loaded.  The effective limit is never below 16: a lookup decides a seam
 * against the boxes of the neighbouring tiles, and the bisection of a ray's
 * crossing can need two 3 x 3 neighbourhoods, side by side, at once.
 * turtle_stack_load brings tiles in up to the limit (all of them without
 * one): after it a batch over the loaded area runs in a single round.  Lookup
 * semantics: half-open tile boxes, exclusive outer upper edge, missing tile =>
 * inside = 0. */
TURTLE_API enum turtle_return turtle_stack_create(struct turtle_stack ** stack,
    const char * path, int stack_size, turtle_stack_locker_t * lock,
    turtle_stack_locker_t * unlock);
TURTLE_API void turtle_stack_destroy(struct turtle_stack ** stack);
TURTLE_API enum turtle_return turtle_stack_clear(struct turtle_stack * stack);
TURTLE_API enum turtle_return turtle_stack_load(struct turtle_stack * stack);
/* extension: the number of tiles in memory right now [ref stack.h: tiles.size] */
TURTLE_API int turtle_amd_stack_resident(const struct turtle_stack * stack);
TURTLE_API enum turtle_return turtle_stack_elevation(
    struct turtle_stack * stack, double latitude, double longitude,
    double * elevation, int * inside);

/* [ref include/turtle.h:748-750; impl stack.c:364-388] */
TURTLE_API enum turtle_return turtle_stack_gradient(
    struct turtle_stack * stack, double latitude, double longitude,
    double * glat, double * glon, int * inside);

/* ---- clients [ref include/turtle.h:773-842; impl client.c:41-188] ----
 * Same answers as the stack; the reference's per-thread tile pinning has no
 * device analogue and is not reproduced. */
TURTLE_API enum turtle_return turtle_client_create(
    struct turtle_client ** client, struct turtle_stack * stack);
TURTLE_API enum turtle_return turtle_client_destroy(
    struct turtle_client ** client);
TURTLE_API enum turtle_return turtle_client_clear(
    struct turtle_client * client);
TURTLE_API enum turtle_return turtle_client_elevation(
    struct turtle_client * client, double latitude, double longitude,
    double * elevation, int * inside);

/* ---- stepper [ref include/turtle.h:859-1155; impl stepper.c:379-931] ---- */
TURTLE_API enum turtle_return turtle_stepper_create(
    struct turtle_stepper ** stepper);
TURTLE_API enum turtle_return turtle_stepper_destroy(
    struct turtle_stepper ** stepper);
TURTLE_API void turtle_stepper_geoid_set(
    struct turtle_stepper * stepper, struct turtle_map * geoid);
TURTLE_API struct turtle_map * turtle_stepper_geoid_get(
    const struct turtle_stepper * stepper);
TURTLE_API void turtle_stepper_reset(struct turtle_stepper * stepper);
/* The local-linear-approximation range [impl stepper.c:85-171] is stored and
 * returned for compatibility; the device always uses the exact transform,
 * i.e. behaves as the reference does at range 0 (difference at the default
 * range 1: 1.5e-9 relative on path length). */
TURTLE_API void turtle_stepper_range_set(
    struct turtle_stepper * stepper, double range);
TURTLE_API double turtle_stepper_range_get(
    const struct turtle_stepper * stepper);
TURTLE_API double turtle_stepper_slope_get(
    const struct turtle_stepper * stepper);
TURTLE_API void turtle_stepper_slope_set(
    struct turtle_stepper * stepper, double slope);
TURTLE_API double turtle_stepper_resolution_get(
    const struct turtle_stepper * stepper);
TURTLE_API void turtle_stepper_resolution_set(
    struct turtle_stepper * stepper, double resolution);
TURTLE_API enum turtle_return turtle_stepper_add_layer(
    struct turtle_stepper * stepper);
TURTLE_API enum turtle_return turtle_stepper_add_stack(
    struct turtle_stepper * stepper, struct turtle_stack * stack,
    double offset);
TURTLE_API enum turtle_return turtle_stepper_add_map(
    struct turtle_stepper * stepper, struct turtle_map * map, double offset);
TURTLE_API enum turtle_return turtle_stepper_add_flat(
    struct turtle_stepper * stepper, double ground_level);
/* [ref include/turtle.h:1126-1129; impl stepper.c:780-875] */
TURTLE_API enum turtle_return turtle_stepper_step(
    struct turtle_stepper * stepper, double * position,
    const double * direction, double * latitude, double * longitude,
    double * altitude, double * elevation, double * step, int * index);
/* [ref include/turtle.h:1153-1155; impl stepper.c:877-931] */
TURTLE_API enum turtle_return turtle_stepper_position(
    struct turtle_stepper * stepper, double latitude, double longitude,
    double height, int layer_index, double * position, int * data_index);

/* ======================================================================== */
/*                    BATCH EXTENSION (not in the reference)                */
/* ======================================================================== */

/* Where the arrays of a batch call live. */
enum turtle_amd_space {
        TURTLE_AMD_HOST = 0,  /* host pointers: copied in/out, call is synchronous */
        TURTLE_AMD_DEVICE = 1 /* device pointers on the selected GPU: the call
                                 only enqueues work on the library stream */
};

/* Device / stream management, per calling THREAD: the device a thread's calls
 * run on ($LOCAL_RANK if set, else 0, until it calls turtle_amd_device_set), its
 * stream, its arithmetic mode and the scratch memory its calls use are its own.
 * The reference's threading rule carries over [ref include/turtle.h:129-132,
 * :620-626, examples/example-pthread.c]: a stepper (and its client) belongs to
 * one thread at a time; maps and stacks may be shared -- a stack whose tiles can
 * change (stack_size below its number of files) with lock / unlock callbacks, as
 * in the reference.  One process may drive one GPU (one process per GPU: what
 * bench.py does) or several, a thread each: maps and tiles get a copy on every
 * device that uses them.  A worker thread calls turtle_amd_thread_release before
 * it ends (its stream and scratch memory are freed then, never behind its back). */
TURTLE_API int turtle_amd_device_count(void);
TURTLE_API enum turtle_return turtle_amd_device_set(int device);
TURTLE_API int turtle_amd_device_get(void);
TURTLE_API void turtle_amd_thread_release(void);
/* Use a caller-owned hipStream_t (e.g. torch's current stream) for every
 * subsequent launch of the calling thread; NULL restores the library's own stream.
 * A batch call on device arrays returns when its kernels are queued.  One thread may
 * keep several batches IN FLIGHT: a stepper and a stream per batch (turtle_amd_stream_set
 * before each call; the steppers may share their maps and stacks).  A trace ends with
 * a few rays of thousands of steps in an all but empty GPU, which another stream's
 * batch fills: C2 batches take 2.6 ms each with two in flight, 2.2-2.3 with three, 3.6 ms
 * one at a time, the same bits (bench.py `in_flight`; tests/test_gpu_properties.py). */
TURTLE_API enum turtle_return turtle_amd_stream_set(void * hip_stream);
/* A second stepper over the same geometry -- the layers and their data in the order
 * they were added, the geoid, range, slope and resolution -- for a batch more in flight.
 * It borrows the same maps and stacks; destroy it with turtle_stepper_destroy. */
TURTLE_API enum turtle_return turtle_amd_stepper_clone(
    const struct turtle_stepper * stepper, struct turtle_stepper ** clone);
/* A hint, per thread: how many batches the caller keeps in flight (default 1).  With two or
 * more a trace kernel takes a smaller share of every compute unit, which leaves room for a
 * wave of another batch's kernel beside its own: better for the batches together, worse for
 * one alone.  Results do not depend on it. */
TURTLE_API void turtle_amd_in_flight_set(int batches);
TURTLE_API int turtle_amd_in_flight_get(void);
TURTLE_API enum turtle_return turtle_amd_synchronize(void);
/* Number of compute units of the selected device (0 if none). */
TURTLE_API int turtle_amd_compute_units(void);

/* Arithmetic of the trace kernel (turtle_stepper_trace_n) and of
 * turtle_ecef_to_geodetic(_n), which exposes the same transform for checking.
 *   STRICT  the reference's expressions in the reference's operand order, no
 *           FMA contraction, OCML asin/acos/atan2: differs from the x86
 *           reference only by the last ulp of those three functions.
 *   FAST    (default) the same algorithm with shared reciprocals, rsqrt-based
 *           roots, one polynomial arctangent and FMAs: ~3x fewer instructions
 *           per sample; coordinates differ from STRICT by a few ulp (<= 3e-9 m
 *           in altitude).  In turtle_stepper_trace_n a long ray samples a cubic
 *           Taylor line of the transform along its path (truncation <= 1e-9 m
 *           near a boundary), advances its position with one fused operation a
 *           step, and every crossing of the batch is located afterwards, in a
 *           kernel of its own, inside the reference's bracket by false position
 *           rather than by halving (the end point agrees to 1e-8 m).
 *           WHAT IS GUARANTEED (and tested, with no allowance, against the
 *           reference's golden vectors and the CPU restatement at full size):
 *           the same medium index; the path length within 1e-6 relative
 *           (measured: <= 5e-8 on every test, 1.6e-8 on C2's million rays); the
 *           step count equal, or off by one on a ray that grazes a surface
 *           within 1e-9 m (measured: 4 rays of C2's million).  The one ray kind
 *           outside the length bar is a ray that reaches no boundary before
 *           max_steps: its "length" is a sum of clearance-sized steps, not a
 *           distance (tests/test_gpu_parity.py names the one the suite has).
 * Both are checked against the reference's golden vectors at the 1e-6 bar.
 * Every other kernel (elevation, position, step, the other ecef transforms)
 * is always STRICT. */
/* WHERE the scalar (one point a call) drop-in functions compute.  DEVICE (default):
 * in the kernels, with n = 1 -- every call a launch and two copies, 20-35 us.  HOST:
 * turtle_ecef_*, turtle_map_elevation, turtle_stack_elevation, turtle_client_elevation,
 * turtle_stepper_step and turtle_stepper_position are answered by a host restatement of
 * the same reference functions (turtle_amd/csrc/scalar.c; the reference's arithmetic
 * with the exact transform, its `last`-sample cache, its tile list) on the host copies
 * of the maps and tiles -- ~0.1 us a call, for callers that keep the reference's per-ray
 * loop.  The batch calls (`_n`) run on the GPU whatever this says; a geometry with a
 * projected map stays with the kernels; and a usable device is required either way:
 * this is an option of a GPU library, not a fallback for machines without one.
 * Process-wide; set it before the stepping starts.  Without a call, the environment decides
 * at the first scalar call: TURTLE_AMD_SCALAR=host (a caller relinked against this library,
 * its source unchanged: examples/reference_loop.c); a call wins over the variable.  A stack
 * that threads share has lock / unlock callbacks, as in the reference; a host lookup in it
 * holds its tile against the other threads' loads while it reads it. */
enum turtle_amd_scalar { TURTLE_AMD_SCALAR_DEVICE = 0, TURTLE_AMD_SCALAR_HOST = 1 };
TURTLE_API void turtle_amd_scalar_set(int mode);
TURTLE_API int turtle_amd_scalar_get(void);

enum turtle_amd_math { TURTLE_AMD_MATH_FAST = 0, TURTLE_AMD_MATH_STRICT = 1 };
TURTLE_API void turtle_amd_math_set(int mode);
TURTLE_API int turtle_amd_math_get(void);

/* n independent ECEF transforms; same arithmetic as the scalar forms. */
TURTLE_API enum turtle_return turtle_ecef_from_geodetic_n(long n,
    const double * latitude, const double * longitude,
    const double * elevation, double * ecef /* [n][3] */, int space);
TURTLE_API enum turtle_return turtle_ecef_to_geodetic_n(long n,
    const double * ecef /* [n][3] */, double * latitude, double * longitude,
    double * altitude, int space);
TURTLE_API enum turtle_return turtle_ecef_from_horizontal_n(long n,
    const double * latitude, const double * longitude, const double * azimuth,
    const double * elevation, double * direction /* [n][3] */, int space);
TURTLE_API enum turtle_return turtle_ecef_to_horizontal_n(long n,
    const double * latitude, const double * longitude,
    const double * direction /* [n][3] */, double * azimuth,
    double * elevation, int space);

/* n projections / inverse projections */
TURTLE_API enum turtle_return turtle_projection_project_n(
    const struct turtle_projection * projection, long n,
    const double * latitude, const double * longitude, double * x, double * y,
    int space);
TURTLE_API enum turtle_return turtle_projection_unproject_n(
    const struct turtle_projection * projection, long n, const double * x,
    const double * y, double * latitude, double * longitude, int space);

/* n bilinear lookups.  `inside` is mandatory (a point outside the data is
 * reported there, never raised). */
TURTLE_API enum turtle_return turtle_map_elevation_n(
    const struct turtle_map * map, long n, const double * x, const double * y,
    double * elevation, int * inside, int space);
TURTLE_API enum turtle_return turtle_stack_elevation_n(
    struct turtle_stack * stack, long n, const double * latitude,
    const double * longitude, double * elevation, int * inside, int space);

/* n gradients (the surface normal a Monte-Carlo needs at a hit point).  The
 * output arrays are in-out, as in the scalar calls. */
TURTLE_API enum turtle_return turtle_map_gradient_n(
    const struct turtle_map * map, long n, const double * x, const double * y,
    double * gx, double * gy, int * inside, int space);
TURTLE_API enum turtle_return turtle_stack_gradient_n(
    struct turtle_stack * stack, long n, const double * latitude,
    const double * longitude, double * glat, double * glon, int * inside,
    int space);

/* n calls of turtle_stepper_position; `position` rows of rays with no data
 * are left untouched and their data_index is -1 (data_index is mandatory). */
TURTLE_API enum turtle_return turtle_stepper_position_n(
    struct turtle_stepper * stepper, long n, const double * latitude,
    const double * longitude, const double * height, int layer_index,
    double * position /* [n][3] */, int * data_index, int space);

/* Flags of turtle_stepper_step_n. */
enum turtle_amd_step_flags {
        /* On entry altitude[], elevation[][2] and index[][2] hold the values a
         * previous call returned for the SAME positions: skip the sample at
         * the start point, exactly as the reference's `last` cache does when
         * the position is unchanged [impl stepper.c:708-710, :745-748]. */
        TURTLE_AMD_STEP_RESUME = 1
};

/* n independent turtle_stepper_step calls.  `direction` may be NULL (sample
 * only; step[] is the tentative length).  Every output array may be NULL
 * except index, which is mandatory: leaving the data is reported as
 * index[r][0] = -1, never raised.  position is updated in place. */
TURTLE_API enum turtle_return turtle_stepper_step_n(
    struct turtle_stepper * stepper, long n, double * position /* [n][3] */,
    const double * direction /* [n][3] or NULL */, double * latitude,
    double * longitude, double * altitude, double * elevation /* [n][2] */,
    double * step, int * index /* [n][2] */, int flags, int space);

/* The same steps for a walk that keeps the LEAST state between its calls.  What a step resumes
 * from is the medium the ray is in and the tentative length its last sample gave it [impl
 * stepper.c:799-813: all the reference reads its cached sample for]: `next` holds that length --
 * one double in, one out, where turtle_stepper_step_n with TURTLE_AMD_STEP_RESUME moves altitude and
 * two elevations each way (a third of what a step streams; a batch of single steps is bound by its
 * memory traffic) -- and `index` the medium.  direction == NULL begins a walk: the positions are
 * sampled, `next` and `index` filled (step[], if given, is that tentative length too); with a
 * direction every ray with index[r][0] >= 0 takes the step turtle_stepper_step would: position
 * advanced in place, step[r] its length, index[r] and next[r] for the one after.  A ray that has
 * left the data (index[r][0] = -1) takes no step.  The same arithmetic on the same values as the
 * RESUME form: the same bits. */
TURTLE_API enum turtle_return turtle_stepper_walk_n(
    struct turtle_stepper * stepper, long n, double * position /* [n][3] */,
    const double * direction /* [n][3], or NULL to begin */, double * next /* [n], in / out */,
    double * step /* [n] or NULL */, int * index /* [n][2], in / out */, int space);

/* Flags of turtle_stepper_scatter_n. */
enum turtle_amd_scatter_flags {
        /* Begin a walk: sample the positions first (as turtle_stepper_step_n with
         * a NULL direction does) and zero length[] and steps[].  Without it the
         * call continues the walk whose state the arrays hold. */
        TURTLE_AMD_SCATTER_START = 1
};

/* A scattering walk: generations first_step .. first_step + n_steps - 1 of
 *     for every ray still inside the data (index[r][0] >= 0):
 *         turtle_stepper_step(position[r], direction = isotropic unit vector of
 *                             Philox(first_ray + r, generation; seed))
 *         length[r] += the step's length;  steps[r] += 1
 * i.e. the loop of examples/example-pthread.c:66-99 with a new direction at
 * every step (turtle_amd_isotropic_n gives the same vectors), each step resumed
 * from the sample the last one returned (TURTLE_AMD_STEP_RESUME: one sample per
 * step, as the reference's `last` cache has it [impl stepper.c:708-710]).  The
 * directions are drawn inside the step kernels and the sums kept there, so a
 * generation moves 80 bytes of ray state in and 64 out and nothing else.
 * altitude, elevation and index are the sample state between generations (in /
 * out; filled by the call itself with TURTLE_AMD_SCATTER_START); a ray that
 * leaves the data keeps index[r][0] = -1 and takes no further step.
 * turtle_stepper_trace_stats reports the totals of the call. */
TURTLE_API enum turtle_return turtle_stepper_scatter_n(
    struct turtle_stepper * stepper, long n, double * position /* [n][3] */,
    unsigned long long seed, long first_ray, int first_step, int n_steps,
    double * altitude, double * elevation /* [n][2] */, int * index /* [n][2] */,
    double * length, int * steps, int flags, int space);

/* Flags of turtle_stepper_trace_n. */
enum turtle_amd_trace_flags {
        /* On entry index[r][0] holds the medium the ray is in, as returned by
         * the previous trace/step that left it at this position.  A ray that
         * has just been put ON a boundary by the bisection sits within 1e-8 m
         * of it, where re-deriving the medium from a fresh sample is
         * ill-conditioned; the reference never re-derives it either (its next
         * step starts from the cached `last` sample [impl stepper.c:708-710,
         * :826]).  Use this flag to continue rays through successive media. */
        TURTLE_AMD_TRACE_RESUME = 1
};

/* The per-ray loop of a Monte-Carlo harness, moved into one kernel: for each
 * ray sample its start point, then step until index[0] differs from its
 * initial value (a boundary was located) or max_steps steps were taken
 * [shape of examples/example-stepper.c:128-140].  Outputs per ray: final
 * index pair, path length = sum of the step lengths, number of steps; the
 * position is advanced in place.  Rays that start outside the data take 0
 * steps and report index[0] = -1.  length/n_steps may be NULL: which outputs a
 * caller asks for changes neither the kernels that run nor a bit of the others. */
TURTLE_API enum turtle_return turtle_stepper_trace_n(
    struct turtle_stepper * stepper, long n, double * position /* [n][3] */,
    const double * direction /* [n][3] */, int max_steps,
    int * index /* [n][2] */, double * length, int * n_steps, int flags,
    int space);

/* Totals of the LAST trace_n or scatter_n call on this stepper, accumulated on the
 * device: stats[0] rays, [1] steps, [2] samples (transform + layer lookup; the
 * same from run to run), [3] rays that stopped at max_steps.  Synchronises the
 * stream. */
TURTLE_API enum turtle_return turtle_stepper_trace_stats(
    struct turtle_stepper * stepper, unsigned long long stats[4]);

/* Rounds the last batch call on this stepper took: 1 when every tile it needed was
 * in memory, one more each time tiles had to come in for rays that waited (paged
 * stacks: stack_size below the tiles a batch touches). */
TURTLE_API int turtle_amd_stepper_rounds(const struct turtle_stepper * stepper);

/* Reduce trace results for a multi-GPU tally: hits[m + 1] counts
 * rays whose final index[0] == m, for m in [-1, n_media); histogram[b] counts
 * path lengths in bin b of n_bins linear bins over [0, length_max), the last
 * extra bin (histogram[n_bins]) collecting overflows.  Both are uint64 and are
 * ADDED to, so ranks can accumulate and then all-reduce them. */
TURTLE_API enum turtle_return turtle_amd_tally_n(long n,
    const int * index /* [n][2] */, const double * length, int n_media,
    unsigned long long * hits /* [n_media + 1] */, int n_bins,
    double length_max, unsigned long long * histogram /* [n_bins + 1] */,
    int space);

/* Counter-based random directions for scattering harnesses (BASELINE config
 * C5: a new isotropic direction after every step).  Philox-4x32-10 with counter
 * = (first_ray + r, stream) and key = seed; `stream` is typically the step
 * number, first_ray the global index of this shard's first ray.  The same
 * (ray, stream, seed) gives the same words on any rank.  turtle_amd_philox_n
 * exposes the raw 4 x 32-bit blocks (for known-answer tests). */
TURTLE_API enum turtle_return turtle_amd_isotropic_n(long n,
    unsigned long long seed, unsigned long long stream, long first_ray,
    double * direction /* [n][3] */, int space);
TURTLE_API enum turtle_return turtle_amd_philox_n(long n,
    unsigned long long seed, unsigned long long stream, long first_ray,
    unsigned int * words /* [n][4] */, int space);

#ifdef __cplusplus
}
#endif
#endif
