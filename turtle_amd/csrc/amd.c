/*
 * amd.c -- device/stream management and the tally reduction of the batch
 * extension (turtle_amd_* in turtle_amd.h).
 */
#include "host.h"

int turtle_amd_device_count(void) { return tamd_dev_count(); }

enum turtle_return turtle_amd_device_set(int device)
{
        TAMD_ERROR_INIT(&turtle_amd_device_set);
        /* the calling THREAD's device from here on: its steppers rebuild their tables
         * there at their next call, maps and tiles get a copy there when first needed */
        if (tamd_dev_select(device)) return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

int turtle_amd_device_get(void) { return tamd_dev_current(); }

void turtle_amd_thread_release(void) { tamd_dev_release(); }

enum turtle_return turtle_amd_stream_set(void * hip_stream)
{
        TAMD_ERROR_INIT(&turtle_amd_stream_set);
        if (tamd_dev_stream_set(hip_stream)) return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_amd_synchronize(void)
{
        TAMD_ERROR_INIT(&turtle_amd_synchronize);
        if (tamd_dev_sync()) return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

int turtle_amd_compute_units(void) { return tamd_dev_cus(); }

void turtle_amd_in_flight_set(int batches) { tamd_dev_in_flight_set(batches); }

int turtle_amd_in_flight_get(void) { return tamd_dev_in_flight_get(); }

void turtle_amd_math_set(int mode) { tamd_dev_math_set(mode == TURTLE_AMD_MATH_STRICT); }

int turtle_amd_math_get(void)
{
        return tamd_dev_math_get() ? TURTLE_AMD_MATH_STRICT : TURTLE_AMD_MATH_FAST;
}

enum turtle_return turtle_amd_tally_n(long n, const int * index, const double * length,
    int n_media, unsigned long long * hits, int n_bins, double length_max,
    unsigned long long * histogram, int space)
{
        TAMD_ERROR_INIT(&turtle_amd_tally_n);
        if ((index == NULL) || (length == NULL) || (hits == NULL) || (histogram == NULL))
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        if ((n_media < 0) || (n_bins < 1) || !(length_max > 0.))
                return TAMD_RAISE(TURTLE_RETURN_DOMAIN_ERROR, "invalid input parameter(s)");
        struct tamd_stage st;
        void *dix, *dlen, *dh, *dg;
        const size_t hb = (size_t)(n_media + 1) * sizeof(*hits);
        const size_t gb = (size_t)(n_bins + 1) * sizeof(*histogram);
        if (tamd_stage_begin(&st, space,
                (size_t)n * (sizeof(double) + 2 * sizeof(int)) + hb + gb) ||
            tamd_stage_in(&st, index, 2 * (size_t)n * sizeof(int), &dix) ||
            tamd_stage_in(&st, length, (size_t)n * sizeof(double), &dlen) ||
            tamd_stage_in(&st, hits, hb, &dh) || tamd_stage_in(&st, histogram, gb, &dg) ||
            tamd_k_tally(n, dix, dlen, n_media, dh, n_bins, length_max, dg) ||
            tamd_stage_fetch(&st, hits, hb, dh) || tamd_stage_fetch(&st, histogram, gb, dg) ||
            tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_amd_philox_n(long n, unsigned long long seed,
    unsigned long long stream, long first_ray, unsigned int * words, int space)
{
        TAMD_ERROR_INIT(&turtle_amd_philox_n);
        struct tamd_stage st;
        void * dw;
        const size_t bytes = (size_t)n * 4 * sizeof(unsigned int);
        if (words == NULL) return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        if (tamd_stage_begin(&st, space, bytes) || tamd_stage_out(&st, words, bytes, &dw) ||
            tamd_k_philox(n, seed, stream, first_ray, dw) ||
            tamd_stage_fetch(&st, words, bytes, dw) || tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_amd_isotropic_n(long n, unsigned long long seed,
    unsigned long long stream, long first_ray, double * direction, int space)
{
        TAMD_ERROR_INIT(&turtle_amd_isotropic_n);
        struct tamd_stage st;
        void * dd;
        const size_t bytes = (size_t)n * 3 * sizeof(double);
        if (direction == NULL)
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null argument");
        if (tamd_stage_begin(&st, space, bytes) || tamd_stage_out(&st, direction, bytes, &dd) ||
            tamd_k_isotropic(n, seed, stream, first_ray, dd) ||
            tamd_stage_fetch(&st, direction, bytes, dd) || tamd_stage_end(&st))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}
