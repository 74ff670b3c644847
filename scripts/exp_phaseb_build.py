#!/usr/bin/env python3
"""Diagnostic build of the library for exp_phaseb_counts.py: a copy of device.hip with counters in
k_trace<MODEL> (what the waves of the lined pass do, and why lanes leave the lean loop), compiled
into scratch/prof3/libturtle_amd.so.  The product source is not touched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "turtle_amd", "csrc")
OUT = os.path.join(ROOT, "scratch", "prof3")
d = open(os.path.join(CSRC, "device.hip")).read()


def sub(old, new, count=1):
    global d
    assert d.count(old) == count, (d.count(old), old)
    d = d.replace(old, new)


d = d.replace("namespace {\n\nconstexpr double kPi", "namespace {\n\n__device__ unsigned long long g_cnt[32];\n__device__ unsigned long long g_span[4096][4];\n\nconstexpr double kPi", 1)
old = "        int creep_wait = 0;"
i = d.index(old)
d = d[:i] + "        unsigned long long c_[24] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};\n        const unsigned long long span0_ = __builtin_amdgcn_s_memtime();\n        unsigned long long dry_ = 0;\n" + d[i:]
sub("                                const bool stopped = (ray >= 0) & !going;\n",
    "                                c_[sparse ? 2 : 3] += 1;\n                                const bool stopped = (ray >= 0) & !going;\n")
sub("                                                count++;\n                                                /* d_step_length for one surface",
    "                                                count++;\n                                                c_[sparse ? 4 : 5] += 1;\n                                                /* d_step_length for one surface")
sub("                                        if (it == 0) creep_wait = kCreepBackoff;",
    "                                        if (it == 0) creep_wait = kCreepBackoff, c_[7] += 1;")
sub("                        const int count_in = count;\n                        for (int it = 0; it < 4096; it++) {",
    "                        const int count_in = count;\n                        if (!sparse) c_[6] += 1;\n"
    "                        const unsigned long long lt0_ = __builtin_amdgcn_s_memtime();\n"
    "                        for (int it = 0; it < 4096; it++) {")
sub("                const bool drain = !MODEL && (ph.park_after > 0) && exhausted && (ray >= 0) &&",
    "                const unsigned long long gt0_ = __builtin_amdgcn_s_memtime();\n"
    "                if (MODEL) c_[0] += 1, c_[1] += (ray >= 0) ? 1 : 0, c_[11] += (live > ph.creep_lanes) ? 1 : 0,\n"
    "                        c_[20] += ((ray >= 0) && !lined_) ? 1 : 0, c_[21] += ((ray >= 0) && (state == ST_BISECT)) ? 1 : 0,\n"
    "                        c_[22] += ((ray >= 0) && (state == ST_INIT)) ? 1 : 0;\n"
    "                const bool drain = !MODEL && (ph.park_after > 0) && exhausted && (ray >= 0) &&")
sub("                                break;\n                        }\n                        my_samples += (ull)(count - count_in);",
    "                                break;\n                        }\n                        c_[9] += __builtin_amdgcn_s_memtime() - lt0_;\n                        my_samples += (ull)(count - count_in);")
sub("                /* ---- park over-long rays (phase A; whole wave takes part) ---- */",
    "                if (MODEL) c_[10] += __builtin_amdgcn_s_memtime() - gt0_;\n"
    "                /* ---- park over-long rays (phase A; whole wave takes part) ---- */")
sub("                                        f_line_relay<MODE>(v, ctx, qx, qy, qz, dx, dy, dz, line, s, cache);\n                                        line.s = -t;",
    "                                        f_line_relay<MODE>(v, ctx, qx, qy, qz, dx, dy, dz, line, s, cache);\n                                        line.s = -t, c_[8] += 1;")
sub("                                        relay_wait = ((n_need == 0) | now) ? 0 : waited + 1;",
    "                                        relay_wait = ((n_need == 0) | now) ? 0 : waited + 1;\n                                        c_[19] += (relay & now) ? 1 : 0, c_[23] += (relay & !now) ? 1 : 0;")
sub("                                        double fx = hx - cx, fy = hy - cy;\n                                        if (going & (max(",
    """                                        if ((u == 0) && (ray >= 0)) {
                                                c_[12] += 1;
                                                c_[13] += (state != ST_STEP) ? 1 : 0;
                                                c_[14] += ((state == ST_STEP) && !(lined_ & line.valid)) ? 1 : 0;
                                                c_[15] += ((state == ST_STEP) && lined_ && line.valid && !going) ? 1 : 0;
                                        }
                                        double fx = hx - cx, fy = hy - cy;
                                        if (going & (max(""")
sub("                                                const double tx = __builtin_trunc(hx), ty = __builtin_trunc(hy);\n                                                const int ix = (int)tx, iy = (int)ty;",
    "                                                if (u == 0) c_[16] += 1;\n                                                const double tx = __builtin_trunc(hx), ty = __builtin_trunc(hy);\n                                                const int ix = (int)tx, iy = (int)ty;")
sub("                                        const double s2 = sl * sl;\n                                        going = going &",
    """                                        const double s2 = sl * sl;
                                        if ((u == 0) && going) {
                                                c_[17] += !f_line_serves(line, sl, clearance) ? 1 : 0;
                                                c_[18] += !(__builtin_fma(sgn, t, 0.) > 0.) ? 1 : 0;
                                        }
                                        going = going &""")
sub("        block_tally(stats, my_rays, my_steps, my_samples, my_capped);\n}\n\n/* The least waves a SIMD the kernel must fit",
    """        if (MODEL && ((threadIdx.x & 63) == 0)) {
                const unsigned w_ = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 4095u;
                g_span[w_][0] = span0_, g_span[w_][1] = __builtin_amdgcn_s_memtime(), g_span[w_][2] = dry_, g_span[w_][3] = c_[0];
        }
        if (MODEL) {
                /* wave-level counters: lane 0's copy; lane counters: all lanes */
                const bool l0 = ((threadIdx.x & 63) == 0);
                for (int k_ = 0; k_ < 24; k_++) {
                        const bool per_lane = (k_ == 1) || (k_ == 4) || (k_ == 5) || (k_ == 8) || (k_ >= 12);
                        if (per_lane || l0) atomicAdd(&g_cnt[k_], c_[k_]);
                }
        }
        block_tally(stats, my_rays, my_steps, my_samples, my_capped);
}

/* The least waves a SIMD the kernel must fit""")
sub('extern "C" int tamd_dev_select(', '''extern "C" int tamd_dev_span_read(unsigned long long * out)
{
        HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(g_span)));
        return 0;
}
extern "C" int tamd_dev_cnt_read(unsigned long long * out, int reset)
{
        HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cnt), sizeof(g_cnt)));
        if (reset) { static unsigned long long zero[32]; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), zero, sizeof(zero))); }
        return 0;
}
extern "C" int tamd_dev_select(''')
d = d.replace("                                if ((long)base >= n) {\n                                        exhausted = true;", "                                if ((long)base >= n) {\n                                        exhausted = true;\n                                        if (dry_ == 0) dry_ = __builtin_amdgcn_s_memtime();", 1)
os.makedirs(OUT, exist_ok=True)
open(os.path.join(OUT, "device.hip"), "w").write(d)
subprocess.check_call(["make", "-C", CSRC], stdout=subprocess.DEVNULL)
obj = os.path.join(CSRC, "build", "device_prof3.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
                       "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(OUT, "device.hip"), "-o", obj])
others = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build")))
          if f.endswith(".o") and not f.startswith("device")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", os.path.join(OUT, "libturtle_amd.so")]
                      + others + [obj, "-lm", "-lz"])
print(os.path.join(OUT, "libturtle_amd.so"))
