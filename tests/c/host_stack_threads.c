/*
 * host_stack_threads.c -- the host's scalar stack lookups (csrc/scalar.c) under threads.
 *
 * The reference's threaded pattern [ref examples/example-pthread.c:66-125]: ONE stack with
 * lock / unlock callbacks, shared; every thread looks points up through it while the stack
 * may keep only `stack_size` tiles, so that one thread's load takes away the tile another
 * has just found.  Every answer must equal the one a stack that keeps everything gives to a
 * single thread, bit for bit.  Built by tests/test_host_threads.py with -fsanitize=thread
 * (host objects instrumented; CPU only), which is what finds a lookup that reads a tile
 * while another thread frees it.
 *
 *   usage: host_stack_threads <directory of .hgt tiles> <stack_size> <threads> <lookups>
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "turtle.h"

/* csrc/scalar.c (not part of the public header: the entry points want a device) */
int tamd_h_stack_elevation(struct turtle_stack * stack, double latitude, double longitude, double * z,
    int * inside, char * message, size_t size);

static pthread_mutex_t g_mutex = PTHREAD_MUTEX_INITIALIZER;
static int lock(void) { return pthread_mutex_lock(&g_mutex); }
static int unlock(void) { return pthread_mutex_unlock(&g_mutex); }

static struct turtle_stack * g_stack;
static int g_lookups;
static double * g_expected; /* [lookups] per thread-independent point */

/* points hopping between the four tiles N45-46 x E003-004, the same for every thread but
 * out of step with each other */
static void point(int k, double * latitude, double * longitude)
{
        unsigned h = (unsigned)k * 2654435761u;
        *latitude = 45.05 + 1.9 * ((h >> 8) & 0xffff) / 65536.;
        *longitude = 3.05 + 1.9 * ((h >> 12) & 0xffff) / 65536.;
}

struct job {
        int id, failed;
};

static void * work(void * arg)
{
        struct job * job = arg;
        char message[4200];
        int i;
        for (i = 0; i < g_lookups; i++) {
                const int k = (i + 37 * job->id) % g_lookups;
                double latitude, longitude, z;
                int inside;
                point(k, &latitude, &longitude);
                const int rc = tamd_h_stack_elevation(g_stack, latitude, longitude, &z, &inside, message,
                    sizeof(message));
                if ((rc != 0) || !inside || (z != g_expected[k])) {
                        fprintf(stderr, "thread %d, point %d: rc %d inside %d z %.17g expected %.17g\n", job->id,
                            k, rc, inside, z, g_expected[k]);
                        job->failed++;
                }
        }
        return NULL;
}

int main(int argc, char * argv[])
{
        if (argc < 5) return 2;
        const int size = atoi(argv[2]), threads = atoi(argv[3]);
        g_lookups = atoi(argv[4]);
        turtle_error_handler_set(NULL);
        g_expected = malloc(g_lookups * sizeof(*g_expected));

        /* what a single thread gets from a stack that keeps every tile */
        struct turtle_stack * all;
        if (turtle_stack_create(&all, argv[1], 0, NULL, NULL) != TURTLE_RETURN_SUCCESS) return 3;
        char message[4200];
        int k;
        for (k = 0; k < g_lookups; k++) {
                double latitude, longitude;
                int inside;
                point(k, &latitude, &longitude);
                if (tamd_h_stack_elevation(all, latitude, longitude, &g_expected[k], &inside, message,
                        sizeof(message)) || !inside)
                        return 4;
        }
        turtle_stack_destroy(&all);

        if (turtle_stack_create(&g_stack, argv[1], size, &lock, &unlock) != TURTLE_RETURN_SUCCESS) return 3;
        pthread_t * thread = malloc(threads * sizeof(*thread));
        struct job * job = calloc(threads, sizeof(*job));
        int failed = 0;
        for (k = 0; k < threads; k++) {
                job[k].id = k;
                pthread_create(&thread[k], NULL, &work, &job[k]);
        }
        for (k = 0; k < threads; k++) {
                pthread_join(thread[k], NULL);
                failed += job[k].failed;
        }
        turtle_stack_destroy(&g_stack);
        printf("%d threads x %d lookups over a stack of size %d: %d wrong answers\n", threads, g_lookups, size,
            failed);
        free(thread), free(job), free(g_expected);
        return failed ? 1 : 0;
}
