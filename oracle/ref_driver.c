/*
 * ref_driver.c -- TEST INFRASTRUCTURE.  A harness around the REAL reference
 * (oracle/_ref/libturtle_ref.so, compiled by oracle/Makefile from the sources
 * under /root/reference), so that bench.py can time the reference's own CPU
 * path beside the GPU's (cpu_baseline.kind = "reference").
 *
 * It is the loop of the reference's example harness [ref
 * examples/example-stepper.c:116-140] -- sample the start point, then
 * turtle_stepper_step until the medium changes -- over n rays, on `threads`
 * pthreads with one stepper each (a stepper is not re-entrant [ref
 * src/turtle/stepper.h:101-110]) sharing one read-only map -- or, for the
 * stack configurations (C3, C5), the pattern of the reference's threaded example
 * [ref examples/example-pthread.c:66-125]: ONE turtle_stack shared by every
 * thread, created with lock / unlock callbacks (a mutex here, a semaphore
 * there), and one client per worker -- the one a stepper makes for itself when
 * it is given a locked stack [ref src/turtle/stepper.c:411-470].  The walk of C5
 * (a new direction at every step) takes its directions from the caller.  Only
 * the reference's PUBLIC API is used; this file contains none of its code.
 */
#define _POSIX_C_SOURCE 200809L
#include "turtle.h" /* the reference's header: -I$(REF)/include, build container only */

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct job {
        struct turtle_stack * stack; /* or NULL: the map */
        int walk_steps;              /* > 0: a scattering walk of that many steps; direction[k][r][3] */
        long n;
        struct turtle_map * map;
        double range, slope, resolution;
        long begin, end;
        double * position;
        const double * direction;
        int max_steps;
        int * index;
        double * length;
        int * n_steps;
        long steps;
        int failed;
};

static void * worker(void * arg)
{
        struct job * job = arg;
        struct turtle_stepper * stepper = NULL;
        if ((turtle_stepper_create(&stepper) != TURTLE_RETURN_SUCCESS) ||
            (((job->stack != NULL) ? turtle_stepper_add_stack(stepper, job->stack, 0.) :
                                     turtle_stepper_add_map(stepper, job->map, 0.)) !=
                TURTLE_RETURN_SUCCESS)) {
                job->failed = 1;
                return NULL;
        }
        turtle_stepper_range_set(stepper, job->range);
        turtle_stepper_slope_set(stepper, job->slope);
        turtle_stepper_resolution_set(stepper, job->resolution);
        long r;
        for (r = job->begin; r < job->end; r++) {
                double * pos = job->position + 3 * r;
                const double * dir = job->direction + 3 * r;
                int idx[2];
                double total = 0.;
                int n = 0;
                turtle_stepper_step(stepper, pos, NULL, NULL, NULL, NULL, NULL, NULL, idx);
                const int medium = idx[0];
                if (job->walk_steps > 0) {
                        /* C5: a new direction at every step; a ray that has left the
                         * data takes no further step */
                        int k;
                        for (k = 0; (k < job->walk_steps) && (idx[0] >= 0); k++) {
                                double ds;
                                turtle_stepper_step(stepper, pos, job->direction + 3 * ((long)k * job->n + r),
                                    NULL, NULL, NULL, NULL, &ds, idx);
                                total += ds;
                                n++;
                        }
                } else if (medium >= 0) {
                        while (n < job->max_steps) {
                                double ds;
                                turtle_stepper_step(
                                    stepper, pos, dir, NULL, NULL, NULL, NULL, &ds, idx);
                                total += ds;
                                n++;
                                if (idx[0] != medium) break;
                        }
                }
                if (job->index != NULL) job->index[2 * r] = idx[0], job->index[2 * r + 1] = idx[1];
                if (job->length != NULL) job->length[r] = total;
                if (job->n_steps != NULL) job->n_steps[r] = n;
                job->steps += n;
        }
        turtle_stepper_destroy(&stepper);
        return NULL;
}

/* Traces n rays through the map at `map_path` (any format the reference loads
 * without external libraries: .hgt).  Returns the total number of steps, or -1. */
long ref_trace_map_n(const char * map_path, double range, double slope, double resolution,
    long n, double * position, const double * direction, int max_steps, int * index,
    double * length, int * n_steps, int threads, double * seconds /* of the stepping alone */)
{
        turtle_error_handler_set(NULL); /* return codes, no exit() */
        struct turtle_map * map = NULL;
        if (turtle_map_load(&map, map_path) != TURTLE_RETURN_SUCCESS) return -1;
        if (threads < 1) threads = 1;
        struct job * jobs = calloc((size_t)threads, sizeof(*jobs));
        pthread_t * tid = calloc((size_t)threads, sizeof(*tid));
        long steps = -1;
        if ((jobs != NULL) && (tid != NULL)) {
                int t;
                for (t = 0; t < threads; t++) {
                        struct job * j = &jobs[t];
                        j->map = map, j->range = range, j->slope = slope, j->resolution = resolution;
                        j->begin = n * t / threads, j->end = n * (t + 1) / threads;
                        j->position = position, j->direction = direction;
                        j->max_steps = max_steps;
                        j->index = index, j->length = length, j->n_steps = n_steps;
                }
                struct timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                if (threads == 1)
                        worker(&jobs[0]);
                else {
                        for (t = 0; t < threads; t++) pthread_create(&tid[t], NULL, worker, &jobs[t]);
                        for (t = 0; t < threads; t++) pthread_join(tid[t], NULL);
                }
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (seconds != NULL)
                        *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
                steps = 0;
                for (t = 0; t < threads; t++) {
                        steps += jobs[t].steps;
                        if (jobs[t].failed) steps = -1;
                }
        }
        free(jobs);
        free(tid);
        turtle_map_destroy(&map);
        return steps;
}

static pthread_mutex_t stack_mutex = PTHREAD_MUTEX_INITIALIZER;
static int stack_lock(void) { return pthread_mutex_lock(&stack_mutex); }
static int stack_unlock(void) { return pthread_mutex_unlock(&stack_mutex); }

/* The same through a turtle_stack over the tiles in `stack_path` (.hgt), shared by
 * `threads` workers with a client each; walk_steps > 0: the scattering walk of C5,
 * direction[walk_steps][n][3], else the trace to the first boundary, direction[n][3].
 * The tiles are loaded before the clock starts (stack_size 0: all of them). */
static long stack_n(int locked, const char * stack_path, int stack_size, double range, double slope,
    double resolution, long n, double * position, const double * direction, int max_steps,
    int walk_steps, int * index, double * length, int * n_steps, int threads, double * seconds)
{
        turtle_error_handler_set(NULL);
        struct turtle_stack * stack = NULL;
        if (!locked) threads = 1; /* a stack without lock / unlock is one thread's [ref include/turtle.h:620-626] */
        if (turtle_stack_create(&stack, stack_path, stack_size, locked ? &stack_lock : NULL,
                locked ? &stack_unlock : NULL) != TURTLE_RETURN_SUCCESS)
                return -1;
        if (turtle_stack_load(stack) != TURTLE_RETURN_SUCCESS) {
                turtle_stack_destroy(&stack);
                return -1;
        }
        if (threads < 1) threads = 1;
        struct job * jobs = calloc((size_t)threads, sizeof(*jobs));
        pthread_t * tid = calloc((size_t)threads, sizeof(*tid));
        long steps = -1;
        if ((jobs != NULL) && (tid != NULL)) {
                int t;
                for (t = 0; t < threads; t++) {
                        struct job * j = &jobs[t];
                        j->stack = stack, j->walk_steps = walk_steps, j->n = n;
                        j->range = range, j->slope = slope, j->resolution = resolution;
                        j->begin = n * t / threads, j->end = n * (t + 1) / threads;
                        j->position = position, j->direction = direction;
                        j->max_steps = max_steps;
                        j->index = index, j->length = length, j->n_steps = n_steps;
                }
                struct timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                for (t = 0; t < threads; t++) pthread_create(&tid[t], NULL, worker, &jobs[t]);
                for (t = 0; t < threads; t++) pthread_join(tid[t], NULL);
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (seconds != NULL)
                        *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
                steps = 0;
                for (t = 0; t < threads; t++) {
                        steps += jobs[t].steps;
                        if (jobs[t].failed) steps = -1;
                }
        }
        free(jobs);
        free(tid);
        turtle_stack_destroy(&stack);
        return steps;
}

long ref_stack_n(const char * stack_path, int stack_size, double range, double slope,
    double resolution, long n, double * position, const double * direction, int max_steps,
    int walk_steps, int * index, double * length, int * n_steps, int threads, double * seconds)
{
        return stack_n(1, stack_path, stack_size, range, slope, resolution, n, position, direction, max_steps,
            walk_steps, index, length, n_steps, threads, seconds);
}

/* The same through a stack WITHOUT lock / unlock, one thread: the stepper then looks the stack up
 * directly (turtle_stack_elevation), not through a client of its own -- whose memo of "no data at
 * this integer (latitude, longitude)" [ref client.c:117-124, :157-160] truncates toward zero, so that
 * a point at longitude -0.3 and a point at +0.3 share a memo cell: a ray that leaves a mosaic
 * through its rim at longitude (or latitude) 0 is then located a degree too early (found in round 4
 * on C5: docs/lab_notebook_r4.md). */
long ref_stack_unlocked_n(const char * stack_path, int stack_size, double range, double slope,
    double resolution, long n, double * position, const double * direction, int max_steps,
    int walk_steps, int * index, double * length, int * n_steps, double * seconds)
{
        return stack_n(0, stack_path, stack_size, range, slope, resolution, n, position, direction, max_steps,
            walk_steps, index, length, n_steps, 1, seconds);
}
