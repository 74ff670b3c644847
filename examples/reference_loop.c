/*
 * reference_loop.c -- a caller written for the REFERENCE and nothing else.
 *
 * It includes "turtle.h", uses only functions that header of niess/turtle declares
 * and keeps the reference's per-ray loop of scalar turtle_stepper_step calls (the
 * shape of the reference's examples/example-stepper.c:116-140).  Relinked against
 * libturtle_amd, source unchanged, it runs as it stands: by default every scalar
 * call is a kernel launch (tens of microseconds); with
 *
 *     TURTLE_AMD_SCALAR=host ./reference_loop
 *
 * the same calls are answered on the host (csrc/scalar.c) at the reference's own
 * cost, a few hundred nanoseconds a call.  Prints one line per ray and a timing line.
 *
 *   cc -std=c99 -Iinclude examples/reference_loop.c -Lturtle_amd -lturtle_amd -lm \
 *      -Wl,-rpath,$PWD/turtle_amd -o reference_loop
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "turtle.h"

int main(int argc, char * argv[])
{
        const int n_rays = (argc > 1) ? atoi(argv[1]) : 16;

        /* a 101 x 101 map over 1 x 1 degree, a ridge running north-south */
        struct turtle_map * map;
        const struct turtle_map_info info = { 101, 101, { 3., 4. }, { 45., 46. }, { 0., 2000. }, NULL };
        turtle_map_create(&map, &info, NULL);
        int ix, iy;
        for (iy = 0; iy < 101; iy++)
                for (ix = 0; ix < 101; ix++)
                        turtle_map_fill(map, ix, iy, 600. + 500. * exp(-0.002 * (ix - 50) * (ix - 50)));

        struct turtle_stepper * stepper;
        turtle_stepper_create(&stepper);
        turtle_stepper_add_map(stepper, map, 0.);

        struct timespec t0, t1;
        long calls = 0;
        int r;
        /* (ray -1 is ray 0 once more, untimed: the first call of a process brings the device up) */
        for (r = -1; r < n_rays; r++) {
                if (r == 0) clock_gettime(CLOCK_MONOTONIC, &t0);
                const int q = (r < 0) ? 0 : r;
                const double latitude = 45.5, longitude = 3.2 + 0.6 * q / n_rays;
                const double azimuth = 360. * q / n_rays, elevation = -3. - 5. * (q % 7) / 7.;
                double position[3], direction[3];
                int layer;
                turtle_stepper_position(stepper, latitude, longitude, 300., 0, position, &layer);
                turtle_ecef_from_horizontal(latitude, longitude, azimuth, elevation, direction);

                /* [ref examples/example-stepper.c:128-140] */
                double total = 0., step;
                int index[2], medium, n = 0;
                turtle_stepper_step(stepper, position, NULL, NULL, NULL, NULL, NULL, NULL, index);
                medium = index[0];
                if (r >= 0) calls += 3;
                while ((index[0] == medium) && (n < 100000)) {
                        turtle_stepper_step(stepper, position, direction, NULL, NULL, NULL, NULL, &step, index);
                        total += step;
                        n++;
                        if (r >= 0) calls++;
                }
                if (r >= 0) printf("ray %d: medium %d -> %d after %d steps, %.12e m\n", r, medium, index[0], n, total);
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("%ld scalar calls, %.0f ns a call\n", calls,
            (1e9 * (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec)) / calls);

        turtle_stepper_destroy(&stepper);
        turtle_map_destroy(&map);
        return 0;
}
