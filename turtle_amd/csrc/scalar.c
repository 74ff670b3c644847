/*
 * scalar.c -- the reference's ONE-POINT entry points on the host, for callers that
 * keep its per-ray loop (turtle_stepper_step in a while(), one ray and one step
 * at a call: [ref examples/example-stepper.c:128-140]).
 *
 * OPT-IN: turtle_amd_scalar_set(TURTLE_AMD_SCALAR_HOST).  By default every
 * computing call of the library runs in a kernel, the scalar ones with n = 1
 * (20-35 us of launch and copies a call, where the reference takes 0.1 us);
 * with the option, the scalar drop-in calls below are answered here -- a host
 * restatement of the same reference functions, operand for operand, on the host
 * copies the library keeps of its maps and tiles -- and nothing else changes:
 * every batch call (`_n`) runs on the GPU whatever the option says, and so does
 * a scalar call over a geometry this file does not take (a projected map).  It is
 * not a fallback: without a usable device the calls still fail, like all others
 * (the option only says WHERE one point is computed, on a machine that has the
 * GPU the library is for).
 *
 * What is restated (citations: paths under the reference tree):
 *   turtle_ecef_*                    [ref src/turtle/ecef.c:41-207]
 *   turtle_map_elevation             [ref src/turtle/map.c:229-277]
 *   turtle_stack_elevation, its tile list and its loads
 *                                    [ref src/turtle/stack.c:300-361, :391-450]
 *   turtle_stepper_step, _position   [ref src/turtle/stepper.c:37-51, :199-264,
 *                                     :687-875, :877-931]
 * always with the exact transform (the reference at local range 0), as in the
 * kernels.  A stepper keeps the reference's `last` sample [ref stepper.c:708-710]:
 * a step from the point the last one returned costs one sample.
 */
#include "host.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* -1: not asked yet -- the environment decides at first use (TURTLE_AMD_SCALAR=host|device:
 * a caller relinked against this library, source unchanged, gets the host's scalar calls by
 * setting a variable); turtle_amd_scalar_set() wins over it, before or after */
static int g_scalar_mode = -1;

static int scalar_mode(void)
{
        int mode = __atomic_load_n(&g_scalar_mode, __ATOMIC_RELAXED);
        if (mode < 0) {
                const char * e = getenv("TURTLE_AMD_SCALAR");
                mode = ((e != NULL) && ((strcmp(e, "host") == 0) || (strcmp(e, "HOST") == 0))) ?
                    TURTLE_AMD_SCALAR_HOST : TURTLE_AMD_SCALAR_DEVICE;
                int unset = -1; /* (a set() that came in between stands) */
                if (!__atomic_compare_exchange_n(&g_scalar_mode, &unset, mode, 0, __ATOMIC_RELAXED,
                        __ATOMIC_RELAXED))
                        mode = unset;
        }
        return mode;
}

void turtle_amd_scalar_set(int mode)
{
        __atomic_store_n(&g_scalar_mode,
            (mode == TURTLE_AMD_SCALAR_HOST) ? TURTLE_AMD_SCALAR_HOST : TURTLE_AMD_SCALAR_DEVICE,
            __ATOMIC_RELAXED);
}
int turtle_amd_scalar_get(void) { return scalar_mode(); }

/* do the scalar calls run here?  (and is there a device: see the header of this file) */
int tamd_scalar_on_host(void) { return (scalar_mode() == TURTLE_AMD_SCALAR_HOST) && (tamd_dev_init() == 0); }

/* ---- WGS84 [ref ecef.c:30-38] ------------------------------------------------ */
#define WGS_A 6378137
#define WGS_B 6356752.3142
#define WGS_E 0.081819190842622
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* [ref ecef.c:41-55] */
void tamd_h_from_geodetic(double latitude, double longitude, double elevation, double ecef[3])
{
        const double a = WGS_A, e = WGS_E;
        const double s = sin(latitude * M_PI / 180.);
        const double c = cos(latitude * M_PI / 180.);
        const double R = a / sqrt(1. - e * e * s * s);
        ecef[0] = (R + elevation) * c * cos(longitude * M_PI / 180.);
        ecef[1] = (R + elevation) * c * sin(longitude * M_PI / 180.);
        ecef[2] = (R * (1. - e * e) + elevation) * s;
}

/* [ref ecef.c:63-130] B. R. Bowring / Olson's closed form; all three outputs */
void tamd_h_to_geodetic(const double ecef[3], double * latitude, double * longitude, double * altitude)
{
        const double a = WGS_A;
        const double e2 = WGS_E * WGS_E;
        const double a1 = a * e2, a2 = a1 * a1, a3 = 0.5 * a1 * e2, a4 = 2.5 * a2, a5 = a1 + a3, a6 = 1. - e2;
        const double x = ecef[0], y = ecef[1], z = ecef[2];
        if ((x == 0.) && (y == 0.)) { /* [ref ecef.c:77-84] */
                *latitude = (z >= 0.) ? 90. : -90.;
                *longitude = 0.;
                *altitude = fabs(z) - WGS_B;
                return;
        }
        *longitude = atan2(y, x) * 180. / M_PI;
        const double zp = fabs(z);
        const double w2 = x * x + y * y, w = sqrt(w2);
        const double z2 = z * z, r2 = w2 + z2, r = sqrt(r2);
        const double s2 = z2 / r2, c2 = w2 / r2;
        const double u0 = a2 / r, v0 = a3 - a4 / r;
        double c, s, ss, la;
        if (c2 > 0.3) { /* [ref ecef.c:101-107] */
                s = (zp / r) * (1. + c2 * (a1 + u0 + s2 * v0) / r);
                la = asin(s);
                ss = s * s;
                c = sqrt(1. - ss);
        } else { /* [ref ecef.c:108-115] */
                c = (w / r) * (1. - s2 * (a5 - u0 - c2 * v0) / r);
                la = acos(c);
                ss = 1. - c * c;
                s = sqrt(ss);
        }
        const double g = 1. - e2 * ss; /* [ref ecef.c:117-129] */
        const double rg = a / sqrt(g), rf = a6 * rg;
        const double u = w - rg * c, v = zp - rf * s;
        const double f = c * u + s * v, m = c * v - s * u;
        const double p = m / (rf / g + f);
        la += p;
        if (z < 0.) la = -la;
        *latitude = la * 180. / M_PI;
        *altitude = f + 0.5 * m * p;
}

/* the local East, North, Up [ref ecef.c:136-154] */
static void h_enu(double latitude, double longitude, double e[3], double n[3], double u[3])
{
        const double lambda = longitude * M_PI / 180., phi = latitude * M_PI / 180.;
        const double sl = sin(lambda), cl = cos(lambda), sp = sin(phi), cp = cos(phi);
        e[0] = -sl, e[1] = cl, e[2] = 0.;
        n[0] = -cl * sp, n[1] = -sl * sp, n[2] = cp;
        u[0] = cl * cp, u[1] = sl * cp, u[2] = sp;
}

/* [ref ecef.c:160-176] */
void tamd_h_from_horizontal(double latitude, double longitude, double azimuth, double elevation,
    double direction[3])
{
        double e[3], n[3], u[3];
        h_enu(latitude, longitude, e, n, u);
        const double az = azimuth * M_PI / 180., el = elevation * M_PI / 180.;
        const double ce = cos(el);
        const double r[3] = { ce * sin(az), ce * cos(az), sin(el) };
        int i;
        for (i = 0; i < 3; i++) direction[i] = r[0] * e[i] + r[1] * n[i] + r[2] * u[i];
}

/* [ref ecef.c:178-207]; a null direction leaves the outputs untouched */
void tamd_h_to_horizontal(double latitude, double longitude, const double direction[3],
    double * azimuth, double * elevation)
{
        double e[3], n[3], u[3];
        h_enu(latitude, longitude, e, n, u);
        const double x = e[0] * direction[0] + e[1] * direction[1] + e[2] * direction[2];
        const double y = n[0] * direction[0] + n[1] * direction[1] + n[2] * direction[2];
        const double z = u[0] * direction[0] + u[1] * direction[1] + u[2] * direction[2];
        double r = direction[0] * direction[0] + direction[1] * direction[1] + direction[2] * direction[2];
        if (r <= FLT_EPSILON) return;
        r = sqrt(r);
        if (azimuth != NULL) *azimuth = atan2(x, y) * 180. / M_PI;
        if (elevation != NULL) {
                const double arg = z / r; /* [ref ecef.c:197-205]: rounding may take it past 1 */
                *elevation = (arg > 1.) ? 90. : ((arg < -1.) ? -90. : asin(arg) * 180. / M_PI);
        }
}

/* ---- one grid [ref map.c:41-44, :229-277] -------------------------------------- */
static double h_node(const struct turtle_map * m, int ix, int iy)
{
        const uint16_t raw = m->nodes[(size_t)iy * m->nx + ix];
        /* int16 codecs give the code itself [ref io/hgt.c:127-131, geotiff16.c:230-233] */
        return m->is_signed ? (double)(int16_t)raw : m->z0 + (double)raw * m->dz;
}

/* 1 inside (*z set), 0 outside (*z untouched) */
int tamd_h_map_elevation(const struct turtle_map * m, double x, double y, double * z)
{
        if (isnan(x) || isnan(y)) return 0; /* [ref map.c:233-240] */
        /* (a tile that came back from a staging buffer: its nodes are read now) */
        if ((m->nodes == NULL) && (tamd_map_host_nodes((struct turtle_map *)m) != TURTLE_RETURN_SUCCESS)) return 0;
        double hx = (x - m->x0) / m->dx;
        double hy = (y - m->y0) / m->dy;
        if ((hx > m->nx - 1) || (hx < 0) || (hy > m->ny - 1) || (hy < 0)) return 0; /* [ref map.c:247-255] */
        int ix = (int)hx, iy = (int)hy;
        if (ix == m->nx - 1) /* [ref map.c:256-265] */
                ix--, hx = 1.;
        else
                hx -= ix;
        if (iy == m->ny - 1)
                iy--, hy = 1.;
        else
                hy -= iy;
        const double z00 = h_node(m, ix, iy), z10 = h_node(m, ix + 1, iy);
        const double z01 = h_node(m, ix, iy + 1), z11 = h_node(m, ix + 1, iy + 1);
        *z = z00 * (1. - hx) * (1. - hy) + z01 * (1. - hx) * hy + z10 * hx * (1. - hy) +
            z11 * hx * hy; /* [ref map.c:272-273] */
        return 1;
}

/* ---- a stack: the tile list, the loads [ref stack.c:300-361, :391-450] ------------ */

/* half-open box of a tile [ref stack.c:307-311, :320-321] */
static int h_tile_holds(const struct turtle_map * m, double latitude, double longitude)
{
        const double hx = (longitude - m->x0) / m->dx, hy = (latitude - m->y0) / m->dy;
        return (hx >= 0.) && (hx < m->nx - 1) && (hy >= 0.) && (hy < m->ny - 1);
}

/* 0: *inside and *z set (z = 0 outside); else an enum turtle_return with `message`.
 *
 * A stack with lock / unlock callbacks is the one threads may share [ref
 * include/turtle.h:620-626, examples/example-pthread.c:66-125], and its tiles can go
 * at any moment: to another thread's load here, to a batch call's page-in or trim.
 * The reference keeps a client's tile by a count under the lock [ref stack.c:433-442];
 * here the lookup and the interpolation hold the geometry IN USE (shared), which
 * whoever frees a tile holds exclusively (tamd_geometry_write_begin) -- and a tile
 * that has to be read is interpolated before that hold is given up
 * (tamd_stack_host_fetch), so that a stack of size 1 under two threads still gets
 * every thread its answer.  A stack without callbacks is one thread's, as in the
 * reference, and pays for none of this. */
int tamd_h_stack_elevation(struct turtle_stack * s, double latitude, double longitude, double * z,
    int * inside, char * message, size_t size)
{
        const int n = s->latitude_n * s->longitude_n, shared = (s->lock != NULL);
        int i, hit = -1;
        *inside = 0, *z = 0.;
        /* (a NaN passes the reference's test of its head tile, which is written the other way
         * round [ref stack.c:310-311], and is then outside that tile [ref map.c:233-240]; with
         * no tile in memory it indexes the directory with (int)NaN) */
        if (isnan(latitude) || isnan(longitude)) return 0;
        if (shared) tamd_geometry_use_begin();
        /* the tile most recently used whose box holds the point [ref stack.c:300-335: the
         * head, then down the list, a hit moving to its head]: stamps order the list here */
        for (i = 0; i < n; i++) {
                if ((s->tile[i] == NULL) || !h_tile_holds(s->tile[i], latitude, longitude)) continue;
                if ((hit < 0) || (s->stamp[i] > s->stamp[hit])) hit = i;
        }
        int rc = 0;
        if (hit < 0) { /* [ref stack.c:399-450] the directory names the file; it is loaded */
                if ((longitude >= s->longitude_0) && (latitude >= s->latitude_0)) {
                        const int ix = (int)((longitude - s->longitude_0) / s->longitude_delta);
                        const int iy = (int)((latitude - s->latitude_0) / s->latitude_delta);
                        if ((ix < s->longitude_n) && (iy < s->latitude_n) &&
                            (s->path[iy * s->longitude_n + ix] != NULL))
                                hit = iy * s->longitude_n + ix;
                }
                if ((hit >= 0) && (s->tile[hit] == NULL)) {
                        if (shared) tamd_geometry_use_end();
                        return tamd_stack_host_fetch(s, hit, latitude, longitude, z, inside, message, size);
                }
        }
        if (hit >= 0) {
                /* [ref stack.c:391-396] */
                __atomic_store_n(&s->stamp[hit], __atomic_add_fetch(&s->clock, 1, __ATOMIC_RELAXED),
                    __ATOMIC_RELAXED);
                double elevation;
                if (tamd_h_map_elevation(s->tile[hit], longitude, latitude, &elevation))
                        *z = elevation, *inside = 1;
        }
        if (shared) tamd_geometry_use_end();
        return rc;
}

/* ---- the stepper [ref stepper.c:687-931] ------------------------------------------- */

/* can this stepper's geometry be answered here?  (projected maps are not: their
 * projection runs in the kernels only) */
int tamd_h_stepper_takes(const struct turtle_stepper * st)
{
        int i;
        for (i = 0; i < st->n_data; i++)
                if ((st->data[i].kind == TAMD_MAP) && (st->data[i].map->projection.type >= 0)) return 0;
        return 1;
}

/* one data at geodetic coordinates: 0 with *inside, *z; else an error code */
static int h_data_elevation(struct turtle_stepper * st, const struct tamd_data * d, double latitude,
    double longitude, double * z, int * inside, char * message, size_t size)
{
        *inside = 0;
        if (d->kind == TAMD_FLAT) { /* [ref stepper.c:252-264] */
                *z = 0., *inside = 1;
                return 0;
        }
        if (d->kind == TAMD_MAP) { /* [ref stepper.c:240-241]: x = longitude, y = latitude */
                *inside = tamd_h_map_elevation(d->map, longitude, latitude, z);
                return 0;
        }
        (void)st;
        return tamd_h_stack_elevation(d->stack, latitude, longitude, z, inside, message, size);
}

/* [ref stepper.c:37-51, :703-756]: geodetic coordinates (less the geoid's undulation),
 * then the layers bottom to top, the data of a layer last added first */
static int h_sample(struct turtle_stepper * st, const double position[3], struct tamd_host_sample * s,
    char * message, size_t size)
{
        if (st->last.valid && (position[0] == st->last.position[0]) && (position[1] == st->last.position[1]) &&
            (position[2] == st->last.position[2])) { /* [ref stepper.c:708-710, :745-748] */
                if (s != &st->last) *s = st->last;
                return 0;
        }
        s->valid = 0;
        memcpy(s->position, position, sizeof(s->position));
        tamd_h_to_geodetic(position, &s->latitude, &s->longitude, &s->altitude);
        if (st->geoid != NULL) {
                double undulation;
                const double lo = (s->longitude >= 0) ? s->longitude : s->longitude + 360.;
                if (tamd_h_map_elevation(st->geoid, lo, s->latitude, &undulation)) s->altitude -= undulation;
        }
        s->index[0] = s->index[1] = -1;
        s->elevation[0] = -DBL_MAX, s->elevation[1] = DBL_MAX; /* [ref stepper.c:713-716] */
        int layer, done = 0;
        for (layer = 0; (layer < st->n_layers) && !done; layer++) {
                const struct tamd_layer * l = &st->layers[layer];
                int j, data_index = 0;
                for (j = l->size - 1; j >= 0; j--, data_index++) {
                        double elevation;
                        int inside;
                        const int rc = h_data_elevation(st, &st->data[l->meta[j].data], s->latitude,
                            s->longitude, &elevation, &inside, message, size);
                        if (rc != 0) return rc;
                        if (!inside) continue;
                        elevation += l->meta[j].offset; /* [ref stepper.c:737] */
                        s->index[1] = data_index;
                        if (elevation >= s->altitude) { /* [ref stepper.c:690-694] */
                                s->index[0] = layer;
                                s->elevation[1] = elevation;
                                done = 1;
                        } else { /* [ref stepper.c:695-699] */
                                s->index[0] = layer + 1;
                                s->elevation[0] = elevation;
                        }
                        break;
                }
        }
        s->valid = 1;
        return 0;
}

static void h_publish(const struct tamd_host_sample * s, double * latitude, double * longitude,
    double * altitude, double * elevation, int * index)
{ /* [ref stepper.c:758-778] */
        if (latitude != NULL) *latitude = s->latitude;
        if (longitude != NULL) *longitude = s->longitude;
        if (altitude != NULL) *altitude = s->altitude;
        if (elevation != NULL) {
                elevation[0] = (s->index[0] >= 0) ? s->elevation[0] : 0.;
                elevation[1] = (s->index[0] >= 0) ? s->elevation[1] : 0.;
        }
        if (index != NULL) index[0] = s->index[0], index[1] = s->index[1];
}

/* [ref stepper.c:780-875].  0, or an enum turtle_return (`message` set, but for
 * DOMAIN_ERROR "no valid data": the caller's text) */
int tamd_h_stepper_step(struct turtle_stepper * st, double * position, const double * direction,
    double * latitude, double * longitude, double * altitude, double * elevation, double * step_length,
    int * index, char * message, size_t size)
{
        int rc = h_sample(st, position, &st->last, message, size);
        if (rc != 0) return rc;
        if (st->last.index[0] < 0) { /* [ref stepper.c:751-755, :791-796] */
                if (index == NULL) return TURTLE_RETURN_DOMAIN_ERROR;
                h_publish(&st->last, latitude, longitude, altitude, elevation, index);
                if (step_length != NULL) *step_length = 0.;
                return 0;
        }
        double ds = 0.; /* [ref stepper.c:799-813] */
        int i;
        for (i = 0; i < 2; i++) {
                if ((st->last.index[0] == 0) && (i == 0)) continue;
                if ((st->last.index[0] == st->n_layers) && (i == 1)) break;
                const double dsi = fabs(st->last.altitude - st->last.elevation[i]);
                if ((dsi < ds) || (ds <= 0.)) ds = dsi;
        }
        ds *= st->slope_factor;
        if (ds < st->resolution_factor) ds = st->resolution_factor;
        if (direction == NULL) { /* [ref stepper.c:816-821] */
                h_publish(&st->last, latitude, longitude, altitude, elevation, index);
                if (step_length != NULL) *step_length = ds;
                return 0;
        }
        for (i = 0; i < 3; i++) position[i] += direction[i] * ds; /* [ref stepper.c:824] */
        const int medium0 = st->last.index[0];
        rc = h_sample(st, position, &st->last, message, size);
        if (rc != 0) return rc;
        if (st->last.index[0] != medium0) { /* [ref stepper.c:832-864] */
                double ds0 = -ds, ds1 = 0.;
                while (ds1 - ds0 > 1E-08) {
                        const double ds2 = 0.5 * (ds0 + ds1);
                        const double q[3] = { position[0] + direction[0] * ds2,
                                position[1] + direction[1] * ds2, position[2] + direction[2] * ds2 };
                        struct tamd_host_sample s2;
                        s2.valid = 0;
                        rc = h_sample(st, q, &s2, message, size);
                        if (rc != 0) return rc;
                        if (s2.index[0] == medium0)
                                ds0 = ds2;
                        else {
                                ds1 = ds2;
                                st->last = s2; /* (its position is q: a step from there is cached) */
                        }
                }
                ds += ds1;
                for (i = 0; i < 3; i++) position[i] += direction[i] * ds1;
        }
        h_publish(&st->last, latitude, longitude, altitude, elevation, index);
        if (step_length != NULL) *step_length = ds;
        if ((st->last.index[0] < 0) && (index == NULL)) return TURTLE_RETURN_DOMAIN_ERROR; /* [ref :870-873] */
        return 0;
}

/* [ref stepper.c:877-931]: *data_index = -1 and the position untouched when no data of the
 * layer holds the point */
int tamd_h_stepper_position(struct turtle_stepper * st, double latitude, double longitude, double height,
    int layer_index, double * position, int * data_index, char * message, size_t size)
{
        const struct tamd_layer * l = &st->layers[layer_index];
        int j, index = 0;
        *data_index = -1;
        for (j = l->size - 1; j >= 0; j--, index++) {
                double elevation = 0.;
                int inside;
                const int rc = h_data_elevation(st, &st->data[l->meta[j].data], latitude, longitude,
                    &elevation, &inside, message, size);
                if (rc != 0) return rc;
                if (!inside) continue;
                elevation += l->meta[j].offset;
                if (st->geoid != NULL) {
                        double undulation;
                        const double lo = (longitude >= 0) ? longitude : longitude + 360.;
                        if (tamd_h_map_elevation(st->geoid, lo, latitude, &undulation)) elevation += undulation;
                }
                tamd_h_from_geodetic(latitude, longitude, elevation + height, position);
                *data_index = index;
                return 0;
        }
        return 0;
}
