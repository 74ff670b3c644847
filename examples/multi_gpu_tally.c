/*
 * multi_gpu_tally.c -- a C host driving several GPUs from one process: a thread
 * per GPU (the shape of the reference's examples/example-pthread.c:66-125: one
 * stepper per thread over shared terrain), rays block-partitioned, and the
 * only exchange of the whole job -- hit counts and a path-length histogram,
 * uint64 -- summed with ncclAllReduce(ncclUint64, ncclSum) over RCCL.
 *
 *   gcc -std=gnu99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include \
 *       examples/multi_gpu_tally.c -Lturtle_amd -lturtle_amd -L/opt/rocm/lib \
 *       -lrccl -lamdhip64 -lpthread -lm -Wl,-rpath,$PWD/turtle_amd -o multi_gpu_tally
 *   ./multi_gpu_tally [n_gpus [rays_per_gpu [share]]]
 *
 * `share` (any third argument): the ranks share the visible GPUs round robin -- two ranks on
 * ONE device, to rehearse a communicator of more than one where there is one GPU.  RCCL (2.x
 * of ROCm 7) refuses that at ncclCommInitAll ("invalid usage": a duplicate device), so this
 * mode only shows that it does; the threads themselves run (tests/test_gpu_paging.py shares a
 * device between four of them).
 *
 * Every rank ends with the same sums; rank 0 prints them, and checks them
 * against one more trace of ALL the rays on its own GPU (integer sums: equal
 * to the bit).  Exit code 0 if they agree.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "turtle.h"

#define N_MEDIA 2
#define N_BINS 1024
#define LENGTH_MAX 65536.
#define TALLY_WORDS ((N_MEDIA + 1) + (N_BINS + 1))

static void on_error(enum turtle_return code, turtle_function_t * function, const char * message)
{
        (void)function;
        fprintf(stderr, "turtle error %d: %s\n", (int)code, message);
        exit(EXIT_FAILURE);
}

#define CHECK(call)                                                                            \
        do {                                                                                   \
                const int rc_ = (int)(call);                                                   \
                if (rc_ != 0) {                                                                \
                        fprintf(stderr, "%s failed (%d) at line %d\n", #call, rc_, __LINE__);  \
                        exit(EXIT_FAILURE);                                                    \
                }                                                                              \
        } while (0)

struct job {
        int rank, world;
        long n; /* rays of this rank */
        struct turtle_map * map; /* shared: read-only while stepping */
        ncclComm_t comm;
        unsigned long long tally[TALLY_WORDS]; /* the reduced sums, as this rank received them */
};

/* rays of the global array [first, first + n): a fan over the map, by index */
static void make_rays(long first, long n, double * lat, double * lon, double * height,
    double * az, double * el)
{
        long r;
        for (r = 0; r < n; r++) {
                const double u = fmod(0.6180339887498949 * (double)(first + r), 1.);
                const double w = fmod(0.7548776662466927 * (double)(first + r), 1.);
                lat[r] = 45.1 + 0.8 * u, lon[r] = 3.1 + 0.8 * w;
                height[r] = 300., az[r] = 360. * fmod(0.5698402909980532 * (double)(first + r), 1.);
                el[r] = -1. - 9. * fmod(0.3247179572447460 * (double)(first + r), 1.);
        }
}

/* trace rays [first, first + n) on the calling thread's GPU and tally them into
 * the device array d_tally (hits, then histogram) */
static void trace_and_tally(struct turtle_map * map, long first, long n, unsigned long long * d_tally)
{
        struct turtle_stepper * stepper;
        turtle_stepper_create(&stepper);
        turtle_stepper_add_map(stepper, map, 0.);
        double *lat = malloc(5 * n * sizeof(double)), *lon = lat + n, *height = lon + n, *az = height + n,
               *el = az + n;
        make_rays(first, n, lat, lon, height, az, el);
        /* the rays live in HBM from here on: positions and directions are made there */
        double *d_in, *d_pos, *d_dir, *d_len;
        int *d_idx, *d_di;
        CHECK(hipMalloc((void **)&d_in, 5 * n * sizeof(double)));
        CHECK(hipMalloc((void **)&d_pos, 3 * n * sizeof(double)));
        CHECK(hipMalloc((void **)&d_dir, 3 * n * sizeof(double)));
        CHECK(hipMalloc((void **)&d_len, n * sizeof(double)));
        CHECK(hipMalloc((void **)&d_idx, 2 * n * sizeof(int)));
        CHECK(hipMalloc((void **)&d_di, n * sizeof(int)));
        CHECK(hipMemcpy(d_in, lat, 5 * n * sizeof(double), hipMemcpyHostToDevice));
        turtle_stepper_position_n(stepper, n, d_in, d_in + n, d_in + 2 * n, 0, d_pos, d_di, TURTLE_AMD_DEVICE);
        turtle_ecef_from_horizontal_n(n, d_in, d_in + n, d_in + 3 * n, d_in + 4 * n, d_dir, TURTLE_AMD_DEVICE);
        turtle_stepper_trace_n(stepper, n, d_pos, d_dir, 100000, d_idx, d_len, NULL, 0, TURTLE_AMD_DEVICE);
        turtle_amd_tally_n(n, d_idx, d_len, N_MEDIA, d_tally, N_BINS, LENGTH_MAX, d_tally + N_MEDIA + 1,
            TURTLE_AMD_DEVICE);
        turtle_amd_synchronize();
        CHECK(hipFree(d_in));
        CHECK(hipFree(d_pos));
        CHECK(hipFree(d_dir));
        CHECK(hipFree(d_len));
        CHECK(hipFree(d_idx));
        CHECK(hipFree(d_di));
        free(lat);
        turtle_stepper_destroy(&stepper);
}

static void * worker(void * arg)
{
        struct job * job = arg;
        turtle_amd_device_set(job->rank % turtle_amd_device_count()); /* this THREAD's GPU from here on */
        unsigned long long * d_tally;
        CHECK(hipMalloc((void **)&d_tally, sizeof(job->tally)));
        CHECK(hipMemset(d_tally, 0, sizeof(job->tally)));
        trace_and_tally(job->map, job->rank * job->n, job->n, d_tally);
        /* the only exchange: ~8 KB of integer sums */
        hipStream_t stream;
        CHECK(hipStreamCreate(&stream));
        CHECK(ncclAllReduce(d_tally, d_tally, TALLY_WORDS, ncclUint64, ncclSum, job->comm, stream));
        CHECK(hipStreamSynchronize(stream));
        CHECK(hipMemcpy(job->tally, d_tally, sizeof(job->tally), hipMemcpyDeviceToHost));
        if (job->rank == 0) { /* all the rays again, on one GPU: the same sums */
                unsigned long long one[TALLY_WORDS];
                CHECK(hipMemset(d_tally, 0, sizeof(one)));
                trace_and_tally(job->map, 0, job->world * job->n, d_tally);
                CHECK(hipMemcpy(one, d_tally, sizeof(one), hipMemcpyDeviceToHost));
                if (memcmp(one, job->tally, sizeof(one)) != 0) {
                        fprintf(stderr, "reduced tally differs from the one-GPU tally\n");
                        exit(EXIT_FAILURE);
                }
        }
        CHECK(hipStreamDestroy(stream));
        CHECK(hipFree(d_tally));
        turtle_amd_thread_release();
        return NULL;
}

int main(int argc, char * argv[])
{
        turtle_error_handler_set(&on_error);
        int world = (argc > 1) ? atoi(argv[1]) : turtle_amd_device_count();
        const long n = (argc > 2) ? atol(argv[2]) : 200000;
        if (world < 1) world = 1;
        const int share = (argc > 3);
        if (!share && (world > turtle_amd_device_count())) {
                fprintf(stderr, "%d GPUs asked for, %d visible\n", world, turtle_amd_device_count());
                return EXIT_FAILURE;
        }
        /* terrain: a 1001 x 1001 map with two ridges, filled through the reference's API */
        struct turtle_map * map;
        const struct turtle_map_info info = { 1001, 1001, { 3., 4. }, { 45., 46. }, { 0., 3000. }, NULL };
        turtle_map_create(&map, &info, NULL);
        int ix, iy;
        for (iy = 0; iy < 1001; iy++)
                for (ix = 0; ix < 1001; ix++)
                        turtle_map_fill(map, ix, iy,
                            500. + 400. * sin(0.013 * ix) * cos(0.017 * iy) +
                                300. * exp(-2e-5 * (ix - 500.) * (ix - 500.)));

        int devices[16];
        ncclComm_t comms[16];
        int g;
        for (g = 0; g < world; g++) devices[g] = share ? g % turtle_amd_device_count() : g;
        const int init = (int)ncclCommInitAll(comms, world, devices);
        if (init != 0) {
                fprintf(stderr, "ncclCommInitAll over devices");
                for (g = 0; g < world; g++) fprintf(stderr, " %d", devices[g]);
                fprintf(stderr, " failed: %s\n", ncclGetErrorString((ncclResult_t)init));
                return EXIT_FAILURE;
        }
        struct job jobs[16];
        pthread_t threads[16];
        for (g = 0; g < world; g++) {
                jobs[g].rank = g, jobs[g].world = world, jobs[g].n = n, jobs[g].map = map, jobs[g].comm = comms[g];
                pthread_create(&threads[g], NULL, &worker, &jobs[g]);
        }
        for (g = 0; g < world; g++) pthread_join(threads[g], NULL);
        for (g = 1; g < world; g++)
                if (memcmp(jobs[g].tally, jobs[0].tally, sizeof(jobs[0].tally)) != 0) {
                        fprintf(stderr, "rank %d received other sums than rank 0\n", g);
                        return EXIT_FAILURE;
                }
        unsigned long long hits = 0, inside = 0;
        int b;
        for (b = 0; b <= N_MEDIA; b++) hits += jobs[0].tally[b];
        for (b = 0; b < N_BINS; b++) inside += jobs[0].tally[N_MEDIA + 1 + b];
        printf("%d GPU(s) x %ld rays: exits %llu, ground %llu, still in the air %llu; %llu path lengths below "
               "%.0f m; reduced tally == one-GPU tally\n", world, n, jobs[0].tally[0], jobs[0].tally[1],
            jobs[0].tally[2], inside, LENGTH_MAX);
        for (g = 0; g < world; g++) ncclCommDestroy(comms[g]);
        turtle_map_destroy(&map);
        return (hits == (unsigned long long)world * n) ? EXIT_SUCCESS : EXIT_FAILURE;
}
