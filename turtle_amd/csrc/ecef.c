/*
 * ecef.c -- host entry points of the WGS84 transforms [ref src/turtle/
 * ecef.c:41-207].  The arithmetic is in device.hip; the scalar calls launch
 * it with n = 1.  They return void, as in the reference, so a device failure
 * can only be reported through the error handler (which by default exits).
 */
#include "host.h"

#include <stddef.h>

static int run4(int (*kernel)(long, const double *, const double *, const double *,
                    const double *, double *),
    long n, const double * a, const double * b, const double * c, const double * d,
    size_t out_doubles, double * out, int space)
{
        struct tamd_stage st;
        void *da, *db, *dc, *dd, *dout;
        const size_t nb = (size_t)n * sizeof(double);
        return tamd_stage_begin(&st, space, (4 + out_doubles) * nb) ||
            tamd_stage_in(&st, a, nb, &da) || tamd_stage_in(&st, b, nb, &db) ||
            tamd_stage_in(&st, c, nb, &dc) || tamd_stage_in(&st, d, nb, &dd) ||
            tamd_stage_out(&st, out, out_doubles * nb, &dout) ||
            kernel(n, da, db, dc, dd, dout) ||
            tamd_stage_fetch(&st, out, out_doubles * nb, dout) || tamd_stage_end(&st);
}

static int k_from_geodetic(long n, const double * lat, const double * lon,
    const double * elev, const double * unused, double * ecef)
{
        (void)unused;
        return tamd_k_ecef_from_geodetic(n, lat, lon, elev, ecef);
}

enum turtle_return turtle_ecef_from_geodetic_n(long n, const double * latitude,
    const double * longitude, const double * elevation, double * ecef, int space)
{
        TAMD_ERROR_INIT(&turtle_ecef_from_geodetic_n);
        if (run4(&k_from_geodetic, n, latitude, longitude, elevation, NULL, 3, ecef, space))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_ecef_from_horizontal_n(long n, const double * latitude,
    const double * longitude, const double * azimuth, const double * elevation,
    double * direction, int space)
{
        TAMD_ERROR_INIT(&turtle_ecef_from_horizontal_n);
        if (run4(&tamd_k_ecef_from_horizontal, n, latitude, longitude, azimuth, elevation, 3,
                direction, space))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_ecef_to_geodetic_n(long n, const double * ecef,
    double * latitude, double * longitude, double * altitude, int space)
{
        TAMD_ERROR_INIT(&turtle_ecef_to_geodetic_n);
        struct tamd_stage st;
        void *de, *dla, *dlo, *dal;
        const size_t nb = (size_t)n * sizeof(double);
        if (tamd_stage_begin(&st, space, 6 * nb) || tamd_stage_in(&st, ecef, 3 * nb, &de) ||
            tamd_stage_out(&st, latitude, nb, &dla) ||
            tamd_stage_out(&st, longitude, nb, &dlo) ||
            tamd_stage_out(&st, altitude, nb, &dal) ||
            tamd_k_ecef_to_geodetic(n, de, dla, dlo, dal) ||
            tamd_stage_fetch(&st, latitude, nb, dla) ||
            tamd_stage_fetch(&st, longitude, nb, dlo) ||
            tamd_stage_fetch(&st, altitude, nb, dal) || tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_ecef_to_horizontal_n(long n, const double * latitude,
    const double * longitude, const double * direction, double * azimuth,
    double * elevation, int space)
{
        TAMD_ERROR_INIT(&turtle_ecef_to_horizontal_n);
        struct tamd_stage st;
        void *dla, *dlo, *dd, *daz, *del;
        const size_t nb = (size_t)n * sizeof(double);
        /* outputs are read-modify-write: a null direction leaves them untouched
         * [ref ecef.c:194] */
        if (tamd_stage_begin(&st, space, 7 * nb) || tamd_stage_in(&st, latitude, nb, &dla) ||
            tamd_stage_in(&st, longitude, nb, &dlo) ||
            tamd_stage_in(&st, direction, 3 * nb, &dd) ||
            tamd_stage_in(&st, azimuth, nb, &daz) || tamd_stage_in(&st, elevation, nb, &del) ||
            tamd_k_ecef_to_horizontal(n, dla, dlo, dd, daz, del) ||
            tamd_stage_fetch(&st, azimuth, nb, daz) ||
            tamd_stage_fetch(&st, elevation, nb, del) || tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

/* ---- scalar forms ---------------------------------------------------------- */

static void raise_void(turtle_function_t * caller)
{
        struct tamd_error error_ = { TURTLE_RETURN_SUCCESS, caller };
        TAMD_RAISE_DEVICE();
}

void turtle_ecef_from_geodetic(
    double latitude, double longitude, double elevation, double ecef[3])
{
        if (tamd_scalar_on_host()) { /* (the caller's option: scalar.c) */
                tamd_h_from_geodetic(latitude, longitude, elevation, ecef);
                return;
        }
        if (run4(&k_from_geodetic, 1, &latitude, &longitude, &elevation, NULL, 3, ecef,
                TURTLE_AMD_HOST))
                raise_void((turtle_function_t *)&turtle_ecef_from_geodetic);
}

void turtle_ecef_from_horizontal(double latitude, double longitude, double azimuth,
    double elevation, double direction[3])
{
        if (tamd_scalar_on_host()) {
                tamd_h_from_horizontal(latitude, longitude, azimuth, elevation, direction);
                return;
        }
        if (run4(&tamd_k_ecef_from_horizontal, 1, &latitude, &longitude, &azimuth,
                &elevation, 3, direction, TURTLE_AMD_HOST))
                raise_void((turtle_function_t *)&turtle_ecef_from_horizontal);
}

void turtle_ecef_to_geodetic(
    const double ecef[3], double * latitude, double * longitude, double * altitude)
{
        if (tamd_scalar_on_host()) {
                double la, lo, al;
                tamd_h_to_geodetic(ecef, &la, &lo, &al);
                if (latitude != NULL) *latitude = la;
                if (longitude != NULL) *longitude = lo;
                if (altitude != NULL) *altitude = al;
                return;
        }
        struct tamd_stage st;
        void *de, *dla, *dlo, *dal;
        if (tamd_stage_begin(&st, TURTLE_AMD_HOST, 6 * sizeof(double)) ||
            tamd_stage_in(&st, ecef, 3 * sizeof(double), &de) ||
            tamd_stage_out(&st, latitude, sizeof(double), &dla) ||
            tamd_stage_out(&st, longitude, sizeof(double), &dlo) ||
            tamd_stage_out(&st, altitude, sizeof(double), &dal) ||
            tamd_k_ecef_to_geodetic(1, de, dla, dlo, dal) ||
            tamd_stage_fetch(&st, latitude, sizeof(double), dla) ||
            tamd_stage_fetch(&st, longitude, sizeof(double), dlo) ||
            tamd_stage_fetch(&st, altitude, sizeof(double), dal) || tamd_stage_end(&st))
                raise_void((turtle_function_t *)&turtle_ecef_to_geodetic);
}

void turtle_ecef_to_horizontal(double latitude, double longitude,
    const double direction[3], double * azimuth, double * elevation)
{
        if (tamd_scalar_on_host()) {
                tamd_h_to_horizontal(latitude, longitude, direction, azimuth, elevation);
                return;
        }
        struct tamd_stage st;
        void *dla, *dlo, *dd, *daz, *del;
        if (tamd_stage_begin(&st, TURTLE_AMD_HOST, 7 * sizeof(double)) ||
            tamd_stage_in(&st, &latitude, sizeof(double), &dla) ||
            tamd_stage_in(&st, &longitude, sizeof(double), &dlo) ||
            tamd_stage_in(&st, direction, 3 * sizeof(double), &dd) ||
            tamd_stage_in(&st, azimuth, sizeof(double), &daz) ||
            tamd_stage_in(&st, elevation, sizeof(double), &del) ||
            tamd_k_ecef_to_horizontal(1, dla, dlo, dd, daz, del) ||
            tamd_stage_fetch(&st, azimuth, sizeof(double), daz) ||
            tamd_stage_fetch(&st, elevation, sizeof(double), del) || tamd_stage_end(&st))
                raise_void((turtle_function_t *)&turtle_ecef_to_horizontal);
}
