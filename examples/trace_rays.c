/*
 * trace_rays.c -- a plain C caller of libturtle_amd.
 *
 * Builds a small geodetic map with the reference's own API, then traces rays
 * to their first boundary three ways: with the reference's per-ray loop of scalar
 * turtle_stepper_step calls (the shape of the reference's
 * examples/example-stepper.c:116-140) answered by the kernels, a launch a call
 * (the default; a few rays only); the same loop with the scalar calls answered
 * on the host (turtle_amd_scalar_set: for callers that keep that loop; all the
 * rays, timed); and with one turtle_stepper_trace_n call.  Prints them and exits
 * non-zero if they disagree.
 *
 *   cc -Iinclude examples/trace_rays.c -Lturtle_amd -lturtle_amd -lm \
 *      -Wl,-rpath,$PWD/turtle_amd -o trace_rays
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "turtle.h" /* the drop-in name; forwards to turtle_amd.h */

#define N_RAYS 256
#define N_CHECK 8

static void on_error(enum turtle_return code, turtle_function_t * function, const char * message)
{
        (void)function;
        fprintf(stderr, "turtle error %d: %s\n", (int)code, message);
        exit(EXIT_FAILURE);
}

int main(void)
{
        turtle_error_handler_set(&on_error);

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

        /* rays: a fan of azimuths from points along the 45.5 N parallel */
        static double lat[N_RAYS], lon[N_RAYS], height[N_RAYS], az[N_RAYS], el[N_RAYS];
        static double position[N_RAYS][3], direction[N_RAYS][3], length[N_RAYS];
        static int data_index[N_RAYS], index[N_RAYS][2], n_steps[N_RAYS];
        int r;
        for (r = 0; r < N_RAYS; r++) {
                lat[r] = 45.5, lon[r] = 3.2 + 0.6 * r / N_RAYS;
                height[r] = 300., az[r] = 360. * r / N_RAYS, el[r] = -3. - 5. * (r % 7) / 7.;
        }
        turtle_stepper_position_n(stepper, N_RAYS, lat, lon, height, 0, &position[0][0],
            data_index, TURTLE_AMD_HOST);
        turtle_ecef_from_horizontal_n(N_RAYS, lat, lon, az, el, &direction[0][0], TURTLE_AMD_HOST);

        /* (1) the reference's way, for the first few rays */
        double scalar_length[N_CHECK];
        int scalar_medium[N_CHECK], scalar_steps[N_CHECK];
        for (r = 0; r < N_CHECK; r++) {
                double p[3] = { position[r][0], position[r][1], position[r][2] };
                int idx[2];
                turtle_stepper_step(stepper, p, NULL, NULL, NULL, NULL, NULL, NULL, idx);
                const int medium = idx[0];
                double total = 0., ds;
                int n = 0;
                do {
                        turtle_stepper_step(stepper, p, direction[r], NULL, NULL, NULL, NULL, &ds, idx);
                        total += ds, n++;
                } while ((idx[0] == medium) && (n < 100000));
                scalar_length[r] = total, scalar_medium[r] = idx[0], scalar_steps[r] = n;
        }

        /* (1b) the same loop, the scalar calls answered on the host: every ray */
        static double host_length[N_RAYS];
        static int host_medium[N_RAYS], host_steps[N_RAYS];
        turtle_amd_scalar_set(TURTLE_AMD_SCALAR_HOST);
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        long host_calls = 0;
        for (r = 0; r < N_RAYS; r++) {
                double p[3] = { position[r][0], position[r][1], position[r][2] };
                int idx[2];
                turtle_stepper_step(stepper, p, NULL, NULL, NULL, NULL, NULL, NULL, idx);
                const int medium = idx[0];
                double total = 0., ds;
                int n = 0;
                do {
                        turtle_stepper_step(stepper, p, direction[r], NULL, NULL, NULL, NULL, &ds, idx);
                        total += ds, n++;
                } while ((idx[0] == medium) && (n < 100000));
                host_length[r] = total, host_medium[r] = idx[0], host_steps[r] = n;
                host_calls += n + 1;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        turtle_amd_scalar_set(TURTLE_AMD_SCALAR_DEVICE);
        const double host_ns = 1e9 * (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_nsec - t0.tv_nsec);
        printf("scalar loop on the host: %ld calls of turtle_stepper_step, %.0f ns a call\n", host_calls,
            host_ns / (double)host_calls);

        /* (2) one batch call for all rays */
        turtle_stepper_trace_n(stepper, N_RAYS, &position[0][0], &direction[0][0], 100000,
            &index[0][0], length, n_steps, 0, TURTLE_AMD_HOST);

        int bad = 0, hits = 0;
        long steps = 0;
        for (r = 0; r < N_RAYS; r++) hits += (index[r][0] == 0), steps += n_steps[r];
        for (r = 0; r < N_CHECK; r++) {
                const double rel = fabs(length[r] - scalar_length[r]) / scalar_length[r];
                printf("ray %d: scalar loop %.6f m in %d steps -> medium %d | trace_n %.6f m in %d "
                       "steps -> medium %d\n",
                    r, scalar_length[r], scalar_steps[r], scalar_medium[r], length[r], n_steps[r],
                    index[r][0]);
                if ((rel > 1e-9) || (scalar_medium[r] != index[r][0]) || (scalar_steps[r] != n_steps[r]))
                        bad++;
        }
        for (r = 0; r < N_RAYS; r++) { /* the host's loop against the kernels, every ray */
                const double rel = fabs(length[r] - host_length[r]) / host_length[r];
                if ((rel > 1e-9) || (host_medium[r] != index[r][0]) || (host_steps[r] != n_steps[r])) bad++;
        }
        printf("%d rays, %ld steps, %d hit the ground; %d disagreements\n", N_RAYS, steps, hits, bad);

        turtle_stepper_destroy(&stepper);
        turtle_map_destroy(&map);
        return bad ? EXIT_FAILURE : EXIT_SUCCESS;
}
