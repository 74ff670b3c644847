// calib_fetch.hip -- what rocprofv3's FETCH_SIZE counts for the access pattern of
// the stepper's kernels: 4-byte gathers from random cache lines of a table far
// larger than the Infinity Cache (so that every distinct line is fetched from
// HBM exactly once per kernel), against a wide streaming read as the control
// (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of those).
//   hipcc -O3 --offload-arch=gfx950 scripts/calib_fetch.hip -o calib_fetch
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// thread i reads 4 bytes of line perm(i): `lines` is a power of two, the
// multiplier odd, so perm is a bijection: every line is touched exactly once
__global__ void gather_one_per_line(const unsigned * table, unsigned lines, unsigned words_per_line, unsigned * out)
{
        const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
        const unsigned line = (i * 2654435761u) & (lines - 1u);
        const unsigned v = table[(size_t)line * words_per_line + (i & 7u)];
        if (v == 0xdeadbeefu) out[0] = v; // never: keeps the load
}

// two 4-byte reads 64 bytes apart in the same 128-byte line (the two node rows of a cell)
__global__ void gather_two_per_line(const unsigned * table, unsigned lines, unsigned * out)
{
        const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
        const unsigned line = (i * 2654435761u) & (lines - 1u);
        const unsigned v = table[(size_t)line * 32u + (i & 7u)] ^ table[(size_t)line * 32u + 16u + (i & 7u)];
        if (v == 0xdeadbeefu) out[0] = v;
}

// the control: 16 bytes per lane, coalesced
__global__ void stream_read(const uint4 * table, size_t n, unsigned * out)
{
        unsigned acc = 0;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
                const uint4 v = table[i];
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
        if (acc == 0xdeadbeefu) out[0] = acc;
}

int main()
{
        const size_t bytes = (size_t)4 << 30; // 4 GiB table
        unsigned *table, *out;
        CHECK(hipMalloc((void **)&table, bytes));
        CHECK(hipMalloc((void **)&out, 256));
        CHECK(hipMemset(table, 1, bytes));
        const unsigned lines128 = (unsigned)(bytes / 128), lines64 = (unsigned)(bytes / 64);
        // every 4th line of each kind: the table is never read whole (no cache can help)
        const unsigned n128 = lines128 / 4, n64 = lines64 / 4;
        for (int rep = 0; rep < 3; rep++) {
                hipLaunchKernelGGL(gather_one_per_line, dim3(n128 / 256), dim3(256), 0, 0, table, lines128, 32u, out);
                hipLaunchKernelGGL(gather_one_per_line, dim3(n64 / 256), dim3(256), 0, 0, table, lines64, 16u, out);
                hipLaunchKernelGGL(gather_two_per_line, dim3(n128 / 256), dim3(256), 0, 0, table, lines128, out);
                hipLaunchKernelGGL(stream_read, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)table, bytes / 16 / 4, out);
        }
        CHECK(hipDeviceSynchronize());
        printf("expected bytes: one_per_128B_line %zu distinct 128-B lines = %zu B (or %zu B at 64 B each); "
               "one_per_64B_line %zu distinct 64-B lines = %zu B; two_per_line %zu lines = %zu B; "
               "stream %zu B\n", (size_t)n128, (size_t)n128 * 128, (size_t)n128 * 64, (size_t)n64, (size_t)n64 * 64,
            (size_t)n128, (size_t)n128 * 128, bytes / 4);
        return 0;
}
