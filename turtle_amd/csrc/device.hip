/*
 * device.hip -- the gfx950 device layer of libturtle_amd: HBM management, the
 * stream, and every kernel of the stepper path.  Written for CDNA4 (wave64);
 * no other target is supported.
 *
 * Arithmetic contract: the kernels evaluate the reference's expressions in
 * the reference's operand order in IEEE fp64 (this file is compiled with
 * -ffp-contract=off, so no FMA is formed across the reference's roundings).
 * +, -, *, / and sqrt are correctly rounded on gfx950, so they agree bit for
 * bit with the x86 reference; sin/cos/asin/acos/atan2 come from ROCm's OCML
 * and may differ from glibc in the last ulp, which is the only source of
 * GPU/CPU differences (<= 1e-9 relative on a path length; the parity bar is
 * 1e-6).  Citations [ref FILE:LINE] are paths under the reference tree.
 *
 * The trace kernel also has a FAST arithmetic (default), a two-phase launch and
 * a cubic Taylor line along each long ray: see the comments at f_to_geodetic,
 * RayLine, PhaseIO and k_trace, and DESIGN.md 3.1.
 *
 * Kernels (one thread = one ray/point; all are fp64 VALU work with a 4-node
 * 16-bit gather per sample, see DESIGN.md for the roofline of each):
 *   k_ecef_*        batch ECEF transforms              [ref ecef.c:41-207]
 *   k_elevation     batch bilinear lookup, map/stack   [ref map.c:229-277, stack.c:300-361]
 *   k_position      batch turtle_stepper_position      [ref stepper.c:877-931]
 *   k_step          batch turtle_stepper_step [ref stepper.c:780-875]: the sample and
 *   k_step_fast     the tentative step; rays that crossed a boundary are listed
 *                   (k_step_fast: the fast-math body of the one-map / one-stack
 *                   modes held to 128 registers)
 *   k_bisect        ... and bisected here, packed [ref stepper.c:836-864]
 *   k_gradient, k_project   batch gradients and map projections
 *   k_trace         persistent-wave trace-to-boundary loop (the hot kernel)
 *   k_isotropic     Philox-4x32-10 isotropic directions (scattering harness)
 *   k_tally         hit counts + path-length histogram (uint64, exact)
 */
#include <hip/hip_runtime.h>
#include <hipcub/device/device_radix_sort.hpp>

#include <sys/syscall.h>
#include <unistd.h>

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "internal.h"

typedef unsigned long long ull;

/* Node arrays live in HBM.  Pointers that reach a kernel inside a descriptor
 * table are generic as far as the compiler can tell, and a generic load is a
 * flat_load (slower, and it ties up both memory counters): say "global". */
typedef const __attribute__((address_space(1))) uint16_t * global_nodes_t;
#define GLOBAL_NODES(p) ((global_nodes_t)(p))

/* ======================================================================== */
/*                               device math                                */
/* ======================================================================== */

namespace {

constexpr double kPi = 3.14159265358979323846; /* [ref ecef.c:30-33] */
constexpr double kA = 6378137;                 /* [ref ecef.c:36-38] */
constexpr double kB = 6356752.3142;
constexpr double kE = 0.081819190842622;

/* [ref ecef.c:41-55] */
__device__ __forceinline__ void d_from_geodetic(
    double latitude, double longitude, double elevation, double & x, double & y, double & z)
{
        const double a = kA, e = kE;
        const double s = sin(latitude * kPi / 180.);
        const double c = cos(latitude * kPi / 180.);
        const double R = a / sqrt(1. - e * e * s * s);
        x = (R + elevation) * c * cos(longitude * kPi / 180.);
        y = (R + elevation) * c * sin(longitude * kPi / 180.);
        z = (R * (1. - e * e) + elevation) * s;
}

/* [ref ecef.c:63-130] Olson (1996) closed form.  All three outputs are always
 * produced (the pointer-null shortcuts of the scalar API live on the host). */
__device__ __forceinline__ void d_to_geodetic(
    double x, double y, double z, double & latitude, double & longitude, double & altitude)
{
        const double a = kA;
        const double e2 = kE * kE;
        const double a1 = a * e2;
        const double a2 = a1 * a1;
        const double a3 = 0.5 * a1 * e2;
        const double a4 = 2.5 * a2;
        const double a5 = a1 + a3;
        const double a6 = 1. - e2;

        if ((x == 0.) && (y == 0.)) { /* [ref ecef.c:77-84] */
                latitude = (z >= 0.) ? 90. : -90.;
                longitude = 0.;
                altitude = fabs(z) - kB;
                return;
        }

        longitude = atan2(y, x) * 180. / kPi;

        const double zp = fabs(z);
        const double w2 = x * x + y * y;
        const double w = sqrt(w2);
        const double z2 = z * z;
        const double r2 = w2 + z2;
        const double r = sqrt(r2);
        const double s2 = z2 / r2;
        const double c2 = w2 / r2;

        double c, s, ss, la;
        const double u0 = a2 / r;
        const double v0 = a3 - a4 / r;
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
        const double rg = a / sqrt(g);
        const double rf = a6 * rg;
        const double u = w - rg * c;
        const double v = zp - rf * s;
        const double f = c * u + s * v;
        const double m = c * v - s * u;
        const double p = m / (rf / g + f);

        la += p;
        if (z < 0.) la = -la;
        latitude = la * 180. / kPi;
        altitude = f + 0.5 * m * p;
}

/* [ref ecef.c:136-154] */
__device__ __forceinline__ void d_enu(
    double latitude, double longitude, double e[3], double n[3], double u[3])
{
        const double lambda = longitude * kPi / 180.;
        const double phi = latitude * kPi / 180.;
        const double sl = sin(lambda), cl = cos(lambda);
        const double sp = sin(phi), cp = cos(phi);
        e[0] = -sl, e[1] = cl, e[2] = 0.;
        n[0] = -cl * sp, n[1] = -sl * sp, n[2] = cp;
        u[0] = cl * cp, u[1] = sl * cp, u[2] = sp;
}

/* ---- fast-math variant of the transform ----------------------------------
 *
 * Same algorithm (Olson 1996, [ref ecef.c:63-130]), leaner arithmetic: the ten
 * divisions become reciprocals shared between terms, sqrt/rsqrt pairs come
 * from one v_rsq_f64 seed with two Goldschmidt steps, asin/acos/atan2 become
 * one first-octant arctangent (3 sectors of half-width pi/16, one division,
 * degree-8 minimax polynomial in t^2, error < 1e-19), and FMAs are used freely.
 * Each primitive is good to ~1 ulp, so latitude/longitude/altitude differ from
 * the strict evaluation by a few ulp (<= 3e-9 m in altitude, <= 1e-13 deg):
 * the same order as the OCML-vs-glibc differences of the strict path and four
 * orders of magnitude inside the 1e-6 parity bar.  tests/test_gpu_parity.py
 * checks both variants against the reference's golden vectors.
 *
 * Why it exists: the trace kernel's run time on the 1 M-ray workload is the
 * latency of its longest ray (11 327 sequential samples), i.e. proportional to
 * the instruction count of ONE sample, and its throughput on larger batches is
 * fp64-VALU bound.  This variant needs ~3x fewer instructions per sample. */

__device__ __forceinline__ double f_rcp(double a)
{
        double y = __builtin_amdgcn_rcp(a);
        double e = __builtin_fma(-a, y, 1.);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-a, y, 1.);
        return __builtin_fma(y, e, y);
}

/* sqrt(a) and 1/sqrt(a) for a normal, positive a (no denormal scaling) */
__device__ __forceinline__ void f_sqrt_rsqrt(double a, double & root, double & inverse)
{
        const double y = __builtin_amdgcn_rsq(a);
        double g = a * y, h = 0.5 * y;
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        /* one correction of the root: g += (a - g*g) * h */
        const double d = __builtin_fma(-g, g, a);
        root = __builtin_fma(d, h, g);
        inverse = h + h;
}

/* atan(y / x) for 0 <= y <= x, x > 0: result in [0, pi/4] */
__device__ __forceinline__ double f_atan_octant(double y, double x)
{
        const bool s1 = y > x * 0.198912367379658;  /* tan(pi/16) */
        const bool s2 = y > x * 0.6681786379192989; /* tan(3pi/16) */
        const double tk = s2 ? 1. : (s1 ? 0.41421356237309503 : 0.);
        const double th = s2 ? 0.7853981633974483 : (s1 ? 0.39269908169872414 : 0.);
        const double num = __builtin_fma(-x, tk, y);
        const double den = __builtin_fma(y, tk, x);
        const double rd = f_rcp(den);
        double t = num * rd;
        t = __builtin_fma(__builtin_fma(-den, t, num), rd, t);
        const double u = t * t;
        double q = 0.050273062752334695;
        q = __builtin_fma(q, u, -0.0660516727229625);
        q = __builtin_fma(q, u, 0.0768988435768423);
        q = __builtin_fma(q, u, -0.0909085307967067);
        q = __builtin_fma(q, u, 0.11111110348139375);
        q = __builtin_fma(q, u, -0.14285714279864764);
        q = __builtin_fma(q, u, 0.19999999999977505);
        q = __builtin_fma(q, u, -0.333333333333333);
        return th + __builtin_fma(t * u, q, t);
}

/* atan2 for s >= 0, c >= 0 (not both 0): result in [0, pi/2] */
__device__ __forceinline__ double f_atan2_q1(double s, double c)
{
        const bool swap = s > c;
        const double a = f_atan_octant(swap ? c : s, swap ? s : c);
        return swap ? 1.5707963267948966 - a : a;
}

__device__ __forceinline__ double f_atan2(double y, double x)
{
        const double ax = fabs(x), ay = fabs(y);
        double a = f_atan2_q1(ay, ax);
        if (x < 0.) a = 3.141592653589793 - a;
        return copysign(a, y);
}

/* ---- a ray's geodetic coordinates as cubics in its path length -----------
 *
 * Along a straight ray q(s) = O + d s the latitude, longitude and altitude are
 * smooth functions of the scalar s, and their Taylor series at O follow from
 * the transform itself.  With (E, N, U) the components of the (constant)
 * direction in the East-North-Up frame of the moving point, M and N' the
 * meridional and prime-vertical radii,
 *     lat' = N / (M + h)      lon' = E / ((N' + h) cos lat)      h' = U
 *     E' = lon' (N sin lat - U cos lat)
 *     N' = -lat' U - lon' E sin lat        U' = lat' N + lon' E cos lat
 * and differentiating twice more gives the second and third derivatives in
 * closed form (~130 flops, no transcendental: the sines and cosines are at
 * hand in the closed-form transform of O).  The neglected term is c4 s^4 with
 * c4 <= 1e-21 (1 + tan^3 lat) m^-3 -- measured against a 40-digit evaluation
 * of the transform, latitudes to 89.5 degrees, any direction, h <= 9 km; at
 * latitude 45: 2.5e-13 m at 100 m, 1.6e-10 m at 500 m, 2.5e-9 m at 1 km.  For
 * comparison the reference's own local approximation (first order,
 * finite-difference Jacobian, 1 m range, [ref stepper.c:85-171]) is off by
 * 8e-8 m.
 *
 * A sample on the line costs 9 FMAs instead of the ~230 instructions of the
 * closed form; in phase B (rays skimming the ground with ~0.5 m steps) a line
 * serves ~1000 samples.  Whether a sample comes from the line depends on the
 * ray alone (its own line and path parameter), never on its wave. */
constexpr double kLineRange = 4000.; /* m, either side of the origin: hard limit */
/* The truncation a line is allowed near a boundary: a third of the closed form's own
 * rounding noise there (3e-9 m).  Rounds 1 and 2 allowed 2e-10 m; at 1e-9 m a line
 * reaches 1.5 x as far (5^(1/4): 760 m at latitude 45) and a ray takes a third fewer
 * closed forms (round 3, measured: C2 4.17 -> 4.05 ms, C4 30.2 -> 28.8; 3e-9 and 1e-8
 * bring no more, the million rays of C2 against the CPU restatement the same 0 / 0,
 * worst path length 2.1e-8 against 1.6e-8). */
#ifndef LINE_TOLERANCE
#define LINE_TOLERANCE 1e-9
#endif
#ifndef LINE_TAU0
#define LINE_TAU0 2e-9
#endif
constexpr double kLineTolerance = LINE_TOLERANCE; /* see f_line_serves */
/* what a lean step counts a clearance as, at most: with k4 <= kLineTolerance / 1.8e-21 (the
 * equator) a sample that passes its reach test is then inside kLineRange as well */
constexpr double kLeanClearance = 400.;
static_assert(kLineTolerance / 1.8e-21 * kLeanClearance < kLineRange * kLineRange * kLineRange * kLineRange,
    "a lean step's reach test must imply the line's hard limit");
/* A ray's position is ACCUMULATED step by step, B += d * ds with the reference's
 * roundings [ref stepper.c:824, :862-863], in every phase: each step leaves B
 * up to half an ulp of 6.4e6 m per coordinate (8e-10 m) off the straight line,
 * mostly the same way from step to step (a skimming ray adds the same increment
 * thousands of times: 2.7e-6 m measured over the 11 326 steps of C2's longest
 * ray) -- and the reference decides on ITS positions.  The line is a function of
 * the path length alone, so a sample taken from it answers for the ideal point
 * O + d * s, which is off the reference's by that drift.  That is harmless where
 * it only sizes the next step, and decisive where the medium is decided within
 * the drift of the boundary (a ray tangent to the ground: one step more or
 * less is 1e-2 m of path; with positions kept ON the line, as round 1 had them,
 * 1 ray of C2's million ended a step early, 1.6e-6 of its path, and the others
 * were within 2e-7 instead of 5e-9).  So the line keeps count of what its
 * truncation and the drift since it was laid can amount to (tau: kLineTau0 at
 * the origin + kLineDrift per accepted step), and a sample whose clearance is
 * not above it is taken again by the closed form AT THE ACCUMULATED POSITION, as
 * phase A would -- which lays a new line there, whose drift starts from nothing
 * (the samples of a bisection all leave from one accumulated position: the line
 * laid at the first of them that is too close to call serves the others -- ALL
 * the others (`bracketed`): it has no drift, and the last halvings of every ray
 * are within 1e-9 m of the ground, where its truncation (kLineTolerance) is below
 * the closed form's own noise.) */
constexpr double kLineTau0 = LINE_TAU0;  /* m: the truncation allowed near a boundary (1e-9 m) and
                                     * the rounding of latitude and longitude (8e-10 m on the
                                     * ground, a third of that in elevation) */
constexpr double kLineDrift = 1e-9; /* m per step: (3 x (2^-31)^2)^0.5 = 8.1e-10 rounded up */

struct RayLine {
        double s;                /* path parameter of the ray's position B */
        double lat[4], lon[4], alt[4]; /* degrees, degrees, metres; [k]: s^k */
        double k4;               /* kLineTolerance / c4, c4 s^4 metres bounding the neglected term */
        double tau;
        bool valid;
};

/* Is the line good enough for a sample at parameter s that came out at
 * `clearance` metres from the nearest boundary?  The truncation error must be
 * below kLineTolerance = 1e-9 m (a third of the closed form's own rounding noise)
 * -- or, far from any boundary, below 1e-9 OF the clearance: all such a sample
 * decides is the length of the next step, to the same relative accuracy.  At
 * latitude 45 this lets a line serve 760 m near the ground and ~3 km in free flight. */
__device__ __forceinline__ bool f_line_serves(const RayLine & L, double s, double clearance,
    bool bracketed = false)
{
        const double s2 = s * s;
        return (s2 * s2 <= L.k4 * fmax(clearance, 1.)) &
            ((bracketed & (L.tau <= kLineTau0)) | (clearance > L.tau));
}

__device__ __forceinline__ void f_line_eval(const RayLine & L, double s, double & latitude,
    double & longitude, double & altitude)
{
        latitude = __builtin_fma(
            s, __builtin_fma(s, __builtin_fma(s, L.lat[3], L.lat[2]), L.lat[1]), L.lat[0]);
        longitude = __builtin_fma(
            s, __builtin_fma(s, __builtin_fma(s, L.lon[3], L.lon[2]), L.lon[1]), L.lon[0]);
        altitude = __builtin_fma(
            s, __builtin_fma(s, __builtin_fma(s, L.alt[3], L.alt[2]), L.alt[1]), L.alt[0]);
}

/* The series at a point whose transform is known: S, C = sin, cos of its
 * latitude; sl, cl of its longitude; rn = N', rm = M; iw2 = 1 / (1 - e2 S^2). */
__device__ __forceinline__ void f_line_build(RayLine & L, double latitude, double longitude,
    double h, double S, double C, double sl, double cl, double rn, double rm, double iw2,
    double dx, double dy, double dz)
{
        constexpr double kRad2Deg = 57.29577951308232;
        const double e2 = kE * kE;
        /* the direction in the local frame */
        const double a = __builtin_fma(cl, dx, sl * dy);
        const double E = __builtin_fma(cl, dy, -(sl * dx));
        const double N = __builtin_fma(C, dz, -(S * a));
        const double U = __builtin_fma(C, a, S * dz);
        /* the radii and their first two derivatives in latitude */
        const double SC = S * C;
        const double k = e2 * SC * iw2;
        const double k1 = e2 * iw2 * ((C * C - S * S) + 2. * e2 * SC * SC * iw2);
        const double rn1 = rn * k, rm1 = 3. * rm * k;
        const double rn2 = __builtin_fma(rn1, k, rn * k1);
        const double rm2 = 3. * __builtin_fma(rm1, k, rm * k1);
        const double re = rn + h;
        const double rho = f_rcp(rm + h), nu = f_rcp(re * C);
        /* first derivatives */
        const double p1 = N * rho, l1 = E * nu, h1 = U;
        const double SNCU = __builtin_fma(S, N, -(C * U));
        const double E1 = l1 * SNCU;
        const double N1 = -(p1 * U) - l1 * S * E;
        const double U1 = __builtin_fma(p1, N, l1 * C * E);
        /* second */
        const double f = __builtin_fma(rm1, p1, h1);       /* (M + h)' */
        const double rho1 = -(rho * rho) * f;
        const double gq = __builtin_fma(rn1, p1, h1);
        const double g = gq * C - re * S * p1;             /* ((N' + h) cos lat)' */
        const double nu1 = -(nu * nu) * g;
        const double p2 = __builtin_fma(N1, rho, N * rho1);
        const double l2 = __builtin_fma(E1, nu, E * nu1);
        const double h2 = U1;
        const double E2 = l2 * SNCU + l1 * (C * p1 * N + S * N1 + S * p1 * U - C * U1);
        const double N2 = -(p2 * U) - p1 * U1 - l2 * S * E - l1 * C * p1 * E - l1 * S * E1;
        const double U2 = p2 * N + p1 * N1 + l2 * C * E - l1 * S * p1 * E + l1 * C * E1;
        /* third */
        const double f1 = rm2 * p1 * p1 + rm1 * p2 + h2;
        const double rho2 = 2. * rho * rho * rho * f * f - rho * rho * f1;
        const double g1 = (rn2 * p1 * p1 + rn1 * p2 + h2) * C - 2. * gq * S * p1 -
            re * C * p1 * p1 - re * S * p2;
        const double nu2 = 2. * nu * nu * nu * g * g - nu * nu * g1;
        const double p3 = N2 * rho + 2. * N1 * rho1 + N * rho2;
        const double l3 = E2 * nu + 2. * E1 * nu1 + E * nu2;
        const double h3 = U2;

        L.lat[0] = latitude, L.lat[1] = kRad2Deg * p1;
        L.lat[2] = (0.5 * kRad2Deg) * p2, L.lat[3] = (kRad2Deg / 6.) * p3;
        L.lon[0] = longitude, L.lon[1] = kRad2Deg * l1;
        L.lon[2] = (0.5 * kRad2Deg) * l2, L.lon[3] = (kRad2Deg / 6.) * l3;
        L.alt[0] = h, L.alt[1] = h1, L.alt[2] = 0.5 * h2, L.alt[3] = h3 * (1. / 6.);
        L.s = 0.;
        L.tau = kLineTau0;
        /* measured (40-digit reference, any direction, h <= 9 km): the
         * fourth-order term is within 1e-21 (1 + tan^3 lat) s^4 metres */
        const double tl = fabs(S) * nu * re;
        L.k4 = kLineTolerance / (1.2e-21 * __builtin_fma(tl * tl, tl, 1.5));
        /* not near a pole (1 / cos lat), nor where the longitude wraps; and only
         * where the bound above was measured: s is a LENGTH (a unit direction;
         * the reference steps along any vector, and so does the closed form that
         * a ray without a valid line keeps using), not deep inside the Earth
         * (1 / (M + h)).  NaNs fail every test. */
        const double dd = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz));
        L.valid = (C > 1e-3) & (fabs(longitude) < 179.9) & (fabs(dd - 1.) < 1e-6) & (h > -1e5);
}

__device__ __forceinline__ void f_to_geodetic(double x, double y, double z,
    double & latitude, double & longitude, double & altitude, RayLine * build = nullptr,
    double dx = 0., double dy = 0., double dz = 0.)
{
        constexpr double kRad2Deg = 57.29577951308232;
        const double a = kA;
        const double e2 = kE * kE;
        const double a1 = a * e2;
        const double a2 = a1 * a1;
        const double a3 = 0.5 * a1 * e2;
        const double a4 = 2.5 * a2;
        const double a5 = a1 + a3;
        const double a6 = 1. - e2;

        if ((x == 0.) && (y == 0.)) { /* [ref ecef.c:77-84] */
                latitude = (z >= 0.) ? 90. : -90.;
                longitude = 0.;
                altitude = fabs(z) - kB;
                if (build != nullptr) build->valid = false;
                return;
        }

        longitude = f_atan2(y, x) * kRad2Deg;

        const double zp = fabs(z);
        const double w2 = __builtin_fma(x, x, y * y);
        const double z2 = z * z;
        const double r2 = w2 + z2;
        double r, ir, w, iw;
        f_sqrt_rsqrt(r2, r, ir);
        f_sqrt_rsqrt(w2, w, iw);
        if (w2 == 0.) w = 0.; /* x*x + y*y underflowed: on the axis to within 1e-162 m */
        const double ir2 = ir * ir;
        const double s2 = z2 * ir2;
        const double c2 = w2 * ir2;
        const double u0 = a2 * ir;
        const double v0 = __builtin_fma(-a4, ir, a3);

        /* [ref ecef.c:101-115] both seeds are cheap; selecting instead of
         * branching keeps the wave converged whatever the latitudes */
        const double s_seed = (zp * ir) * __builtin_fma(c2 * (a1 + u0 + s2 * v0), ir, 1.);
        const double c_seed = (w * ir) * __builtin_fma(-s2 * (a5 - u0 - c2 * v0), ir, 1.);
        const bool low = c2 > 0.3; /* |latitude| below ~56.8 deg: seed the sine */
        const double seed = low ? s_seed : c_seed;
        const double ss = low ? seed * seed : __builtin_fma(-seed, seed, 1.);
        double other, unused;
        f_sqrt_rsqrt(low ? 1. - ss : ss, other, unused);
        const double s = low ? seed : other;
        const double c = low ? other : seed;
        double la = f_atan2_q1(s, c);

        const double g = __builtin_fma(-e2, ss, 1.); /* [ref ecef.c:117-129] */
        double sg, isg;
        f_sqrt_rsqrt(g, sg, isg);
        const double rg = a * isg;
        const double rf = a6 * rg;
        const double u = __builtin_fma(-rg, c, w);
        const double v = __builtin_fma(-rf, s, zp);
        const double f = __builtin_fma(c, u, s * v);
        const double m = __builtin_fma(c, v, -(s * u));
        const double p = m * f_rcp(__builtin_fma(rf * isg, isg, f));
        (void)sg;

        la += p;
        if (z < 0.) la = -la;
        latitude = la * kRad2Deg;
        altitude = __builtin_fma(0.5 * m, p, f);

        if (build != nullptr) { /* everything it needs is at hand */
                /* sine and cosine of the corrected latitude, and the radii there:
                 * p is ~4e-8 rad, which the radii of the seed would turn into
                 * 4e-10 of the distance along the line (2e-7 m at 500 m) */
                const double sf = __builtin_fma(c, p, s), cf = __builtin_fma(-s, p, c);
                double sw, isw;
                f_sqrt_rsqrt(__builtin_fma(-e2 * sf, sf, 1.), sw, isw);
                (void)sw;
                const double isw2 = isw * isw, rn = a * isw;
                f_line_build(*build, latitude, longitude, altitude, (z < 0.) ? -sf : sf, cf,
                    y * iw, x * iw, rn, a6 * rn * isw2, isw2, dx, dy, dz);
                build->valid = build->valid & (w2 != 0.);
        }
}

/* ---- map projections ------------------------------------------------------ */

/* [ref projection.c:329-349]: e, n, c, lambda_c, xs, ys of Lambert I, II, IIe,
 * III, IV (NTG_71) and Lambert 93 (RGF93) */
__constant__ double kLambert[6][6] = {
        { 0.08248325676, 0.7604059656, 11603796.98, 0.04079234433, 600000.0, 5657616.674 },
        { 0.08248325676, 0.7289686274, 11745793.39, 0.04079234433, 600000.0, 6199695.768 },
        { 0.08248325676, 0.7289686274, 11745793.39, 0.04079234433, 600000.0, 8199695.768 },
        { 0.08248325676, 0.6959127966, 11947992.52, 0.04079234433, 600000.0, 6791905.085 },
        { 0.08248325676, 0.6712679322, 12136281.99, 0.04079234433, 234.358, 7239161.542 },
        { 0.08181919112, 0.7253743710, 11755528.70, 0.05235987756, 700000.0, 12657560.145 }
};

/* [ref projection.c:192-210, :238-244, :286-295 (Lambert), :377-408 (UTM)] */
__device__ __noinline__ void d_project(
    const tamd_proj & pr, double latitude, double longitude, double & x, double & y)
{
        if (pr.type == TAMD_PROJ_LAMBERT) {
                const double * P = kLambert[pr.lambert_tag];
                const double e = P[0];
                const double phi = latitude * kPi / 180.;
                const double s = sin(phi);
                const double L = log(tan(0.25 * kPi + 0.5 * phi) *
                    pow((1. - e * s) / (1. + e * s), 0.5 * e));
                const double cenL = P[2] * exp(-P[1] * L);
                const double lambda = longitude / 180. * kPi;
                const double theta = P[1] * (lambda - P[3]);
                x = P[4] + cenL * sin(theta);
                y = P[5] - cenL * cos(theta);
                return;
        }
        const double a = 6378.137E+03;
        const double f = 1. / 298.257223563;
        const double E0 = 5E+05;
        const double N0 = (pr.hemisphere > 0) ? 0. : 1E+07;
        const double k0 = 0.9996;
        const double n = f / (2. - f);
        const double A = a / (1. + n) * (1. + n * n * (0.25 + 0.0625 * n * n));
        const double alpha[3] = { n * (0.5 + n * (-2. / 3. + 5. / 16. * n)),
                n * n * (13. / 48. - 3. / 5. * n), 61. / 240. * n * n * n };
        const double c = 2. * sqrt(n) / (1. + n);
        const double s = sin(latitude * kPi / 180.);
        const double t = sinh(atanh(s) - c * atanh(c * s));
        const double dl = (longitude - pr.longitude_0) * kPi / 180.;
        const double zeta = atan2(t, cos(dl));
        const double eta = atanh(sin(dl) / sqrt(1. + t * t));
        double xs = 0., ys = 0.;
        for (int i = 0; i < 3; i++) {
                xs += alpha[i] * cos(2. * (i + 1) * zeta) * sinh(2. * (i + 1) * eta);
                ys += alpha[i] * sin(2. * (i + 1) * zeta) * cosh(2. * (i + 1) * eta);
        }
        x = E0 + k0 * A * (eta + xs);
        y = N0 + k0 * A * (zeta + ys);
}

/* [ref projection.c:213-230, :253-268, :304-318 (Lambert), :417-448 (UTM)] */
__device__ __noinline__ void d_unproject(
    const tamd_proj & pr, double x, double y, double & latitude, double & longitude)
{
        if (pr.type == TAMD_PROJ_LAMBERT) {
                const double * P = kLambert[pr.lambert_tag];
                const double e = P[0];
                const double dx = x - P[4];
                const double dy = y - P[5];
                const double R = sqrt(dx * dx + dy * dy);
                const double gamma = atan2(dx, -dy);
                longitude = (P[3] + gamma / P[1]) * 180. / kPi;
                const double L = -log(R / P[2]) / P[1];
                const double eL = exp(L);
                double phi0 = 2. * atan(eL) - 0.5 * kPi;
                for (int it = 0; it < 64; it++) { /* converges in 3-4 rounds */
                        const double s = sin(phi0);
                        const double phi1 =
                            2. * atan(pow((1. + e * s) / (1. - e * s), 0.5 * e) * eL) - 0.5 * kPi;
                        const bool stop = fabs(phi1 - phi0) <= (double)FLT_EPSILON;
                        phi0 = phi1;
                        if (stop) break;
                }
                latitude = phi0 / kPi * 180.;
                return;
        }
        const double a = 6378.137E+03;
        const double f = 1. / 298.257223563;
        const double E0 = 5E+05;
        const double N0 = (pr.hemisphere > 0) ? 0. : 1E+07;
        const double k0 = 0.9996;
        const double n = f / (2. - f);
        const double A = a / (1. + n) * (1. + n * n * (0.25 + 0.0625 * n * n));
        const double beta[3] = { n * (0.5 + n * (-2. / 3. + 37. / 96. * n)),
                n * n * (1. / 48. + 1. / 15. * n), 17. / 480. * n * n * n };
        const double delta[3] = { n * (2. + n * (-2. / 3. - 2. * n)),
                n * n * (7. / 3. - 8. / 5. * n), 56. / 15. * n * n * n };
        const double zeta0 = (y - N0) / (k0 * A);
        const double eta0 = (x - E0) / (k0 * A);
        double zeta = zeta0, eta = eta0;
        for (int i = 0; i < 3; i++) {
                zeta -= beta[i] * sin(2. * (i + 1) * zeta0) * cosh(2. * (i + 1) * eta0);
                eta -= beta[i] * cos(2. * (i + 1) * zeta0) * sinh(2. * (i + 1) * eta0);
        }
        const double chi = asin(sin(zeta) / cosh(eta));
        double s = 0.;
        for (int i = 0; i < 3; i++) s += delta[i] * sin(2. * (i + 1) * chi);
        latitude = (chi + s) * 180. / kPi;
        longitude = pr.longitude_0 + atan2(sinh(eta), cos(zeta)) * 180. / kPi;
}

/* ---- one grid --------------------------------------------------------- */

/* Where node (ix, iy) sits in HBM: the grid is stored in blocks of 8 x 8 nodes
 * (128 bytes = one cache line: internal.h), so that the four nodes of a cell --
 * and the cells a ray visits next, whichever way it heads -- share a line far
 * more often than in rows of 7 KB. */
__device__ __forceinline__ unsigned d_node_index(int nbx, int ix, int iy)
{
        return (((unsigned)iy >> 3) * (unsigned)nbx + ((unsigned)ix >> 3)) * 64u +
            (((unsigned)iy & 7u) << 3) + ((unsigned)ix & 7u);
}

/* the four raw nodes of cell (ix, iy): lo = z00 | z10 << 16, hi = z01 | z11 << 16.
 * One index computation; the upper row is +8 inside a block, or a jump to the
 * next block row.  The two nodes of a row are neighbours in memory except in a
 * block's last column: one 32-bit load (2-byte aligned) fetches both, and only
 * the lanes in a last column (1 in 8) go back for the node of the next block.
 * Per wave that is ~144 line look-ups in the vector L1 instead of 256 -- the
 * gathers are most of what a batch of single steps asks of it.  (The pair load
 * never overruns the array: a cell's left-hand nodes have ix <= nx - 2, which
 * is never the last node of the last block.) */
typedef unsigned __attribute__((aligned(2))) u32_a2;
typedef const __attribute__((address_space(1))) u32_a2 * global_pair_t;

__device__ __forceinline__ void d_cell_fetch(
    const uint16_t * nodes, int nbx, int ix, int iy, unsigned & lo, unsigned & hi)
{
        global_nodes_t p = GLOBAL_NODES(nodes) + d_node_index(nbx, ix, iy);
        const unsigned up = (((unsigned)iy & 7u) == 7u) ? (unsigned)nbx * 64u - 56u : 8u;
        lo = *(global_pair_t)p, hi = *(global_pair_t)(p + up);
        if (((unsigned)ix & 7u) == 7u) { /* 64 - 7: the next block's first column */
                const unsigned z10 = p[57], z11 = p[up + 57];
                lo = (lo & 0xffffu) | (z10 << 16), hi = (hi & 0xffffu) | (z11 << 16);
        }
}

__device__ __forceinline__ double d_node(const tamd_grid & g, int ix, int iy)
{
        const uint16_t raw = GLOBAL_NODES(g.nodes)[d_node_index(g.nbx, ix, iy)];
        const double v = g.is_signed ? (double)(int16_t)raw : (double)raw;
        return g.z0 + v * g.dz; /* [ref map.c:41-44]; exact for z0=0, dz=1 */
}

/* [ref map.c:229-277]: inclusive upper edge, truncation toward zero, the
 * four-term sum in the reference's operand order. */
/* Last cell a lane looked up: its id and its four raw nodes.  A ray that
 * creeps along the surface (the long rays that set the run time of a launch)
 * stays in one 20-30 m cell for tens of steps; re-using the nodes takes the
 * gather out of its critical path. */
struct CellCache {
        unsigned id;     /* iy * nx + ix, or ~0u when empty */
        unsigned lo, hi; /* (z00 | z10 << 16), (z01 | z11 << 16), raw codes */
        /* regular stacks: the tile the last lookup fell in, and its nodes (saves the
         * dependent pointer load of every sample that stays in the tile) */
        int slot;
        const uint16_t * tile;
};

/* [ref map.c:229-277], fast-math form.  Differences from the strict form, none
 * of which changes an elevation by more than ~1e-12 m: (x - x0) is multiplied
 * by 1/dx instead of divided (except within 1e-6 cell of the rim, where the
 * exact quotient decides inside/outside as in the reference); the cell index is
 * clamped instead of special-cased (hx == nx-1 gives ix = nx-2, fx = 1 either
 * way); the two nodes of a row come from one unaligned 32-bit load. */
struct CellAt {
        double hx, hy; /* node coordinates of the point */
        int ix, iy;    /* its cell, clamped into the grid */
        unsigned id;   /* iy * nx + ix */
        bool inside;   /* [ref map.c:247-255], NaN => false */
        bool rim;      /* within 1e-6 cell of the rim: exact quotients were used */
};

__device__ __forceinline__ CellAt f_grid_locate(const tamd_grid & g, double x, double y)
{
        CellAt c;
        c.hx = (x - g.x0) * g.inv_dx;
        c.hy = (y - g.y0) * g.inv_dy;
        const double mx = (double)(g.nx - 1), my = (double)(g.ny - 1);
        c.rim = !((c.hx > 1e-6) && (c.hx < mx - 1e-6) && (c.hy > 1e-6) && (c.hy < my - 1e-6));
        if (__builtin_expect(c.rim, 0)) {
                c.hx = (x - g.x0) / g.dx;
                c.hy = (y - g.y0) / g.dy;
        }
        c.inside = (c.hx >= 0.) && (c.hx <= mx) && (c.hy >= 0.) && (c.hy <= my);
        c.ix = min(max((int)c.hx, 0), g.nx - 2);
        c.iy = min(max((int)c.hy, 0), g.ny - 2);
        c.id = (unsigned)c.iy * (unsigned)g.nx + (unsigned)c.ix;
        return c;
}

__device__ __forceinline__ unsigned d_upper_word(double x)
{
        return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32);
}

/* The bilinear patch over a cell, fast-math form: z00 + fx b + fy (c + fx d) with
 * b = z10 - z00, c = z01 - z00, d = (z11 - z10) - c -- three fused operations
 * where the reference's four-term sum [ref map.c:270-276] takes thirteen, within
 * an ulp or two of it (1e-13 m).  The lean steps of the lined pass keep b, c, d
 * per cell. */
__device__ __forceinline__ double f_patch(double z00, double b, double c, double d, double fx, double fy)
{
        return __builtin_fma(fy, __builtin_fma(fx, d, c), __builtin_fma(fx, b, z00));
}

/* the bilinear blend of a cell's four raw nodes (lo = z00 | z10 << 16, hi =
 * z01 | z11 << 16); one function so that every caller rounds identically */
__device__ __forceinline__ double f_grid_blend(
    const tamd_grid & g, const CellAt & c, unsigned lo, unsigned hi)
{
        const double fx = c.hx - (double)c.ix, fy = c.hy - (double)c.iy;
        double z00, z10, z01, z11;
        if (g.is_signed) {
                z00 = (double)(int16_t)(lo & 0xffffu), z10 = (double)((int)lo >> 16);
                z01 = (double)(int16_t)(hi & 0xffffu), z11 = (double)((int)hi >> 16);
        } else {
                z00 = (double)(lo & 0xffffu), z10 = (double)(lo >> 16);
                z01 = (double)(hi & 0xffffu), z11 = (double)(hi >> 16);
        }
        z00 = __builtin_fma(z00, g.dz, g.z0), z10 = __builtin_fma(z10, g.dz, g.z0);
        z01 = __builtin_fma(z01, g.dz, g.z0), z11 = __builtin_fma(z11, g.dz, g.z0);
        return f_patch(z00, z10 - z00, z01 - z00, (z11 - z10) - (z01 - z00), fx, fy);
}

__device__ __forceinline__ bool f_grid_elevation(
    const tamd_grid & g, double x, double y, double & z, CellCache * cache = nullptr)
{
        const CellAt c = f_grid_locate(g, x, y);
        unsigned lo, hi;
        if ((cache != nullptr) && (cache->id == c.id)) {
                lo = cache->lo, hi = cache->hi;
        } else {
                d_cell_fetch(g.nodes, g.nbx, c.ix, c.iy, lo, hi);
                if (cache != nullptr) cache->id = c.id, cache->lo = lo, cache->hi = hi;
        }
        z = f_grid_blend(g, c, lo, hi);
        return c.inside;
}

/* [ref map.c:229-277]: inclusive upper edge, truncation toward zero, the
 * four-term sum in the reference's operand order. */
template <bool FAST = false>
__device__ __forceinline__ bool d_grid_elevation(
    const tamd_grid & g, double x, double y, double & z)
{
        if (FAST) return f_grid_elevation(g, x, y, z);
        if (isnan(x) || isnan(y)) return false; /* [ref map.c:233-240] */
        double hx = (x - g.x0) / g.dx;
        double hy = (y - g.y0) / g.dy;
        if ((hx > g.nx - 1) || (hx < 0) || (hy > g.ny - 1) || (hy < 0))
                return false; /* [ref map.c:247-255] */
        int ix = (int)hx;
        int iy = (int)hy;
        if (ix == g.nx - 1) { /* [ref map.c:256-265] */
                ix--;
                hx = 1.;
        } else
                hx -= ix;
        if (iy == g.ny - 1) {
                iy--;
                hy = 1.;
        } else
                hy -= iy;
        const double z00 = d_node(g, ix, iy);
        const double z10 = d_node(g, ix + 1, iy);
        const double z01 = d_node(g, ix, iy + 1);
        const double z11 = d_node(g, ix + 1, iy + 1);
        z = z00 * (1. - hx) * (1. - hy) + z01 * (1. - hx) * hy +
            z10 * hx * (1. - hy) + z11 * hx * hy; /* [ref map.c:272-273] */
        return true;
}

/* [ref map.c:280-378], as it is -- including the slip at map.c:352-353: for a
 * point in the grid's first half-row (iy == 0, hy <= 0.5) the y-gradient lands
 * in gx and gy is left untouched.  gx, gy are therefore in-out. */
__device__ __forceinline__ bool d_grid_gradient(
    const tamd_grid & g, double x, double y, double & gx, double & gy)
{
        if (isnan(x) || isnan(y)) return false;
        double hx = (x - g.x0) / g.dx;
        double hy = (y - g.y0) / g.dy;
        if ((hx > g.nx - 1) || (hx < 0) || (hy > g.ny - 1) || (hy < 0)) return false;
        int ix = (int)hx;
        int iy = (int)hy;
        if (ix == g.nx - 1) {
                ix--;
                hx = 1.;
        } else
                hx -= ix;
        if (iy == g.ny - 1) {
                iy--;
                hy = 1.;
        } else
                hy -= iy;
        const double z00 = d_node(g, ix, iy), z10 = d_node(g, ix + 1, iy);
        const double z01 = d_node(g, ix, iy + 1), z11 = d_node(g, ix + 1, iy + 1);

        if (hx <= 0.5) { /* [ref map.c:324-335] */
                const double gx1 = (z10 - z00) * (1. - hy) + (z11 - z01) * hy;
                if (ix == 0) {
                        gx = gx1 / g.dx;
                } else {
                        const double z_10 = d_node(g, ix - 1, iy), z_11 = d_node(g, ix - 1, iy + 1);
                        const double gx0 = (z00 - z_10) * (1. - hy) + (z01 - z_11) * hy;
                        const double ax = hx + 0.5;
                        gx = (gx0 * (1. - ax) + gx1 * ax) / g.dx;
                }
        } else { /* [ref map.c:336-348] */
                const double gx0 = (z10 - z00) * (1. - hy) + (z11 - z01) * hy;
                if (ix == g.nx - 2) {
                        gx = gx0 / g.dx;
                } else {
                        const double z20 = d_node(g, ix + 2, iy), z21 = d_node(g, ix + 2, iy + 1);
                        const double gx1 = (z20 - z10) * (1. - hy) + (z21 - z11) * hy;
                        const double ax = hx - 0.5;
                        gx = (gx0 * (1. - ax) + gx1 * ax) / g.dx;
                }
        }
        if (hy <= 0.5) { /* [ref map.c:350-361] */
                const double gy1 = (z01 - z00) * (1. - hx) + (z11 - z10) * hx;
                if (iy == 0) {
                        gx = gy1 / g.dy; /* sic [ref map.c:353] */
                } else {
                        const double z0_1 = d_node(g, ix, iy - 1), z1_1 = d_node(g, ix + 1, iy - 1);
                        const double gy0 = (z00 - z0_1) * (1. - hx) + (z10 - z1_1) * hx;
                        const double ay = hy + 0.5;
                        gy = (gy0 * (1. - ay) + gy1 * ay) / g.dy;
                }
        } else { /* [ref map.c:362-374] */
                const double gy0 = (z01 - z00) * (1. - hx) + (z11 - z10) * hx;
                if (iy == g.ny - 2) {
                        gy = gy0 / g.dy;
                } else {
                        const double z02 = d_node(g, ix, iy + 2), z12 = d_node(g, ix + 1, iy + 2);
                        const double gy1 = (z02 - z01) * (1. - hx) + (z12 - z11) * hx;
                        const double ay = hy - 0.5;
                        gy = (gy0 * (1. - ay) + gy1 * ay) / g.dy;
                }
        }
        return true;
}

/* ---- tile directory --------------------------------------------------- */

/* half-open box of a resident tile [ref stack.c:307-311, :320-321] */
__device__ __forceinline__ bool d_tile_holds(
    const tamd_grid & g, double latitude, double longitude)
{
        const double hx = (longitude - g.x0) / g.dx;
        const double hy = (latitude - g.y0) / g.dy;
        return (hx >= 0.) && (hx < g.nx - 1) && (hy >= 0.) && (hy < g.ny - 1);
}

/* [ref stack.c:338-361] with every tile resident.  The reference scans its
 * tile list for the one whose half-open box holds the point and only then
 * falls back on the directory formula of turtle_stack_load_ [ref
 * stack.c:413-424], applying the inclusive bilinear test to that tile.  Here
 * the directory formula proposes the tile first (O(1)); its neighbours are
 * consulted only when rounding at a seam makes the box test disagree, which
 * reproduces the list scan's answer without the list. */
/* Paging.  A stack may hold more tiles than it keeps in HBM (its stack_size,
 * [ref stack.c:150, :434-443]).  In the tile table a tile that has a file but is
 * not resident reads TAMD_TILE_PAGED; a lookup that needs such a tile -- to
 * answer, or to decide a seam -- returns a FAULT code instead of a tile:
 * tile_fault(table index), any value below -1.  The kernels list the rays /
 * points that met one (page_fault), with the tiles they want; the host brings
 * those in (evicting the least recently wanted) and runs the list again. */
/* What a faulting lookup wants: tiles of the 3 x 3 neighbourhood of the
 * directory slot `centre` (an index into the tile table; -1: no fault); bit
 * 3 (j + 1) + (i + 1) of `mask` stands for the tile at centre + j * stride + i.
 * A point inside a tile wants that tile only; a point on a seam, where the
 * boxes of the neighbours decide, wants every neighbour that has a file: they
 * all come in together, and none is dropped to make room for another. */
struct TileFault {
        int centre, mask, stride;
};
constexpr double kRimGuard = 1e-9; /* of a directory cell: ~1e4 x the rounding of fx, fy */
/* Of a tile cell: how close to a seam the fast lookup trusts hx = (x - x0) * (1/dx)
 * to land on the same side as the reference's quotient.  The two differ by
 * < 4 ulp of hx (2e-11 cell for the largest grid, 65 535 nodes a side); 1e-9
 * leaves a factor 50, and is narrow enough (3e-8 m of a 30 m cell) that the
 * bisection of an exit through the mosaic's rim -- which converges ON the seam,
 * to 1e-8 m -- takes the exact way, with its dependent loads, for its last
 * couple of samples only and not for a dozen. */
constexpr double kSeamGuard = 1e-9;
constexpr int kTileFault = TAMD_TILE_PAGED; /* d_stack_tile's return value then */

/* The tile of a stack that answers for a point, -1 for none, or kTileFault
 * with `f` filled in [ref stack.c:300-335, :413-424]: see d_stack_elevation. */
__device__ __forceinline__ int d_stack_tile(const tamd_view & v, const tamd_stack & st,
    double latitude, double longitude, TileFault & f)
{
        const double fx = (longitude - st.lon0) / st.dlon;
        const double fy = (latitude - st.lat0) / st.dlat;
        /* No tile box reaches further than one cell from the directory -- and in
         * a regular stack (tiles exactly on the lattice, each spanning its cell up
         * to rounding) none reaches beyond its rim: out there the directory
         * formula [ref stack.c:413-424] finds no tile either.  This early answer
         * is what a ray that has left the mosaic gets at every later step of a
         * batch, and half the samples of the bisection of its exit: without it
         * each of them costs its whole wave the neighbourhood scan below (four
         * dependent loads, a dozen divisions). */
        const double reach = st.regular ? kRimGuard : 1.5;
        if (!((fx > -reach) && (fx < st.nlon + reach) && (fy > -reach) &&
                (fy < st.nlat + reach)))
                return -1;
        const int cx = min(max((int)fx, 0), st.nlon - 1);
        const int cy = min(max((int)fy, 0), st.nlat - 1);
        const int * tiles = v.tiles + st.tile_first;
        int tile = tiles[cy * st.nlon + cx];
        f.centre = st.tile_first + cy * st.nlon + cx, f.stride = st.nlon, f.mask = 1 << 4;
        /* Outside the directory's range the formula [ref stack.c:413-424] names no
         * tile, and the reference loads none: only tiles that ARE in memory can
         * answer (its list scan); a tile that is not is wanted only for a point in
         * the range, or within rounding of its rim. */
        const bool in_range = (fx > -kRimGuard) && (fx < st.nlon + kRimGuard) && (fy > -kRimGuard) &&
            (fy < st.nlat + kRimGuard);
        if (tile == TAMD_TILE_PAGED) {
                if (in_range) return kTileFault;
                tile = TAMD_TILE_NONE;
        }
        if ((tile < 0) || !d_tile_holds(v.grids[tile], latitude, longitude)) {
                /* rare: a seam, the rim, or a hole in the mosaic.  A neighbour
                 * that is not resident may be the one whose box holds the point:
                 * the whole neighbourhood has to be in memory to tell */
                int mask = 0, paged = 0;
                for (int j = -1; j <= 1; j++) {
                        for (int i = -1; i <= 1; i++) {
                                const int ix = cx + i, iy = cy + j;
                                if ((ix < 0) || (ix >= st.nlon) || (iy < 0) || (iy >= st.nlat)) continue;
                                const int t = tiles[iy * st.nlon + ix];
                                if (t == TAMD_TILE_NONE) continue;
                                mask |= 1 << (3 * (j + 1) + (i + 1));
                                if ((t == TAMD_TILE_PAGED) && in_range) paged = 1;
                        }
                }
                if (paged) {
                        f.mask = mask;
                        return kTileFault;
                }
                tile = -1;
                for (int j = -1; (j <= 1) && (tile < 0); j++) {
                        for (int i = -1; (i <= 1) && (tile < 0); i++) {
                                const int ix = cx + i, iy = cy + j;
                                if (((i == 0) && (j == 0)) || (ix < 0) || (ix >= st.nlon) ||
                                    (iy < 0) || (iy >= st.nlat))
                                        continue;
                                const int t = tiles[iy * st.nlon + ix];
                                if ((t >= 0) && d_tile_holds(v.grids[t], latitude, longitude))
                                        tile = t;
                        }
                }
                if (tile < 0) { /* [ref stack.c:413-424] */
                        if ((longitude < st.lon0) || (latitude < st.lat0)) return -1;
                        if (!(fx < st.nlon) || !(fy < st.nlat)) return -1;
                        tile = tiles[(int)fy * st.nlon + (int)fx]; /* == the centre: resident or none */
                }
        }
        return tile;
}

/* 1: inside (z set), 0: outside (z = 0), -1: a fault (f filled in) */
template <bool FAST = false>
__device__ __forceinline__ int d_stack_elevation(const tamd_view & v,
    const tamd_stack & st, double latitude, double longitude, double & z, TileFault & f)
{
        z = 0.;
        const int tile = d_stack_tile(v, st, latitude, longitude, f);
        if (tile < 0) return (tile == kTileFault) ? -1 : 0;
        const bool inside = d_grid_elevation<FAST>(v.grids[tile], longitude, latitude, z);
        if (!inside) z = 0.;
        return inside ? 1 : 0;
}

/* Fast-math lookup in a `regular` stack (see struct tamd_stack): interior
 * points take the tile by the directory formula and read its nodes through one
 * pointer; anything within kSeamGuard of a cell of a tile seam or of the directory's rim,
 * and any irregular stack, goes through the general routine above, which
 * decides seams and edges exactly as the reference does. */
/* `slots`: the stack's node pointers, one per directory slot, copied to LDS by
 * the kernel (see d_load_ctx) when TABLE.  The per-lane copy of the last tile's
 * pointer is then not needed (8 registers fewer in the trace kernel), and a
 * sample in a new tile waits for HBM once (the nodes), not twice in a row. */
typedef const uint16_t * node_ptr_t;
typedef const __attribute__((address_space(3))) node_ptr_t * lds_slots_t;

template <bool TABLE = false> /* TABLE: `slots` is there whenever the stack is regular */
__device__ __forceinline__ int f_stack_elevation(const tamd_view & v,
    const tamd_stack & st, double latitude, double longitude, double & z,
    CellCache * cache, TileFault & f, lds_slots_t slots = nullptr)
{
        if (st.regular) {
                const tamd_grid & p = st.proto;
                /* 1/dlon, 1/dlat: a last-ulp difference from the quotient can only
                 * pick the neighbouring tile for a point ON a seam, which is then
                 * not `interior` below and goes the exact way */
                const double fx = (longitude - st.lon0) * st.inv_dlon;
                const double fy = (latitude - st.lat0) * st.inv_dlat;
                const bool in_dir =
                    (fx > 0.) && (fx < (double)st.nlon) && (fy > 0.) && (fy < (double)st.nlat);
                const int tx = in_dir ? (int)fx : 0, ty = in_dir ? (int)fy : 0;
                const double x0 = st.lon0 + tx * st.dlon, y0 = st.lat0 + ty * st.dlat;
                const double hx = (longitude - x0) * p.inv_dx;
                const double hy = (latitude - y0) * p.inv_dy;
                const double mx = (double)(p.nx - 1) - kSeamGuard, my = (double)(p.ny - 1) - kSeamGuard;
                const bool interior =
                    in_dir && (hx > kSeamGuard) && (hx < mx) && (hy > kSeamGuard) && (hy < my);
                const int slot = ty * st.nlon + tx;
                const uint16_t * nodes = nullptr;
                if (TABLE) {
                        if (interior) nodes = slots[slot];
                } else if (interior) {
                        if ((cache != nullptr) && (cache->slot == slot))
                                nodes = cache->tile;
                        else {
                                nodes = v.slot_nodes[st.nodes_first + slot];
                                if (cache != nullptr) cache->slot = slot, cache->tile = nodes;
                        }
                }
                if (nodes != nullptr) {
                        const int ix = (int)hx, iy = (int)hy;
                        const double fxc = hx - (double)ix, fyc = hy - (double)iy;
                        const unsigned cell = (unsigned)iy * (unsigned)p.nx + (unsigned)ix;
                        const unsigned id = ((unsigned)slot << 24) | cell;
                        unsigned lo, hi;
                        if ((cache != nullptr) && (cache->id == id)) {
                                lo = cache->lo, hi = cache->hi;
                        } else {
                                d_cell_fetch(nodes, p.nbx, ix, iy, lo, hi);
                                if (cache != nullptr)
                                        cache->id = id, cache->lo = lo, cache->hi = hi;
                        }
                        double z00, z10, z01, z11;
                        if (p.is_signed) {
                                z00 = (double)(int16_t)(lo & 0xffffu), z10 = (double)((int)lo >> 16);
                                z01 = (double)(int16_t)(hi & 0xffffu), z11 = (double)((int)hi >> 16);
                        } else {
                                z00 = (double)(lo & 0xffffu), z10 = (double)(lo >> 16);
                                z01 = (double)(hi & 0xffffu), z11 = (double)(hi >> 16);
                        }
                        z00 = __builtin_fma(z00, p.dz, p.z0), z10 = __builtin_fma(z10, p.dz, p.z0);
                        z01 = __builtin_fma(z01, p.dz, p.z0), z11 = __builtin_fma(z11, p.dz, p.z0);
                        z = f_patch(z00, z10 - z00, z01 - z00, (z11 - z10) - (z01 - z00), fxc, fyc);
                        return 1;
                }
        }
        return d_stack_elevation<true>(v, st, latitude, longitude, z, f);
}

/* ---- layered sample ---------------------------------------------------- */

struct Sample {
        double lat, lon, alt;
        double e0, e1; /* bounding elevations [ref stepper.h:93-98] */
        int m, k;      /* index[0] = medium/layer, index[1] = data */
        TileFault fault; /* .centre >= 0: tiles have to be paged in (nothing else of
                          * the sample is then meaningful) */
        int slot;        /* tile-table index of the stack tile that answered (the first
                          * stack consulted), or -1: a ray that later waits for another
                          * tile wants this one kept too -- it is where it resumes */
};

/* 1: inside, 0: outside, -1: a fault, f filled in (stacks only) */
template <bool FAST = false>
__device__ __forceinline__ int d_source_elevation(const tamd_view & v,
    const tamd_meta & mt, double latitude, double longitude, double & z, TileFault & f)
{
        if (mt.kind == TAMD_FLAT) { /* [ref stepper.c:252-264] */
                z = 0.;
                return 1;
        } else if (mt.kind == TAMD_MAP) {
                const tamd_grid & g = v.grids[mt.src];
                if (g.proj.type >= 0) { /* [ref stepper.c:243-248, :304-311] */
                        double x, y;
                        d_project(g.proj, latitude, longitude, x, y);
                        return d_grid_elevation<FAST>(g, x, y, z);
                }
                /* [ref stepper.c:240-241] geodetic grid: x = lon, y = lat */
                return d_grid_elevation<FAST>(g, longitude, latitude, z);
        }
        return d_stack_elevation<FAST>(v, v.stacks[mt.src], latitude, longitude, z, f);
}

/* [ref stepper.c:703-756] + check_layer [ref stepper.c:687-701], always with
 * the exact transform (the reference at local_range = 0) and, when a geoid is
 * set, its undulation removed from the altitude [ref stepper.c:37-51]. */
/* Descriptors of the single data source of the one-map / one-stack modes,
 * read ONCE per kernel (they are wave-uniform: SGPRs) instead of per sample. */
struct OneCtx {
        tamd_grid grid;
        tamd_stack stack;
        double offset;
        lds_slots_t slots; /* one-stack mode, fast math, regular stack: see f_stack_elevation */
};

/* Called by every thread of the block, at the top of the kernel (it holds a
 * barrier when it fills the LDS table: blocks are 256 threads, and a regular
 * stack has at most 255 slots). */
template <int MODE, bool FAST = false>
__device__ __forceinline__ void d_load_ctx(const tamd_view & v, OneCtx & c)
{
        c.slots = nullptr;
        if (MODE == TAMD_MODE_GENERIC) return;
        const tamd_meta mt = v.metas[0];
        c.offset = mt.offset;
        if (MODE == TAMD_MODE_ONE_MAP) c.grid = v.grids[mt.src];
        if (MODE == TAMD_MODE_ONE_STACK) c.stack = v.stacks[mt.src];
        if ((MODE == TAMD_MODE_ONE_STACK) && FAST) {
                __shared__ node_ptr_t table[256];
                if (c.stack.regular) {
                        if ((int)threadIdx.x < c.stack.nlat * c.stack.nlon)
                                table[threadIdx.x] = v.slot_nodes[c.stack.nodes_first + threadIdx.x];
                        __syncthreads();
                        c.slots = (lds_slots_t)table;
                }
        }
}

/* The layers at geodetic coordinates already in s.lat, s.lon, s.alt */
template <int MODE, bool FAST = false>
__device__ __forceinline__ void d_classify(
    const tamd_view & v, const OneCtx & ctx, Sample & s, CellCache * cache = nullptr)
{
        s.m = -1, s.k = -1, s.fault.centre = -1, s.slot = -1;
        s.e0 = -DBL_MAX, s.e1 = DBL_MAX; /* [ref stepper.c:713-716] */

        if (MODE != TAMD_MODE_GENERIC) {
                /* one layer holding one data: no loops, no geoid */
                double elevation;
                int inside;
                if (MODE == TAMD_MODE_ONE_MAP)
                        inside = FAST ?
                            f_grid_elevation(ctx.grid, s.lon, s.lat, elevation, cache) :
                            d_grid_elevation<false>(ctx.grid, s.lon, s.lat, elevation);
                else
                        inside = FAST ?
                            f_stack_elevation<true>(v, ctx.stack, s.lat, s.lon, elevation, cache, s.fault, ctx.slots) :
                            d_stack_elevation<false>(v, ctx.stack, s.lat, s.lon, elevation, s.fault);
                if ((MODE == TAMD_MODE_ONE_STACK) && (inside >= 0)) s.slot = s.fault.centre;
                if ((MODE != TAMD_MODE_ONE_STACK) || (inside >= 0)) s.fault.centre = -1;
                if (inside > 0) {
                        elevation += ctx.offset;
                        s.k = 0;
                        if (elevation >= s.alt) {
                                s.m = 0;
                                s.e1 = elevation;
                        } else {
                                s.m = 1;
                                s.e0 = elevation;
                        }
                }
                return;
        }

        if (v.geoid >= 0) {
                double undulation;
                const double lo = (s.lon >= 0) ? s.lon : s.lon + 360.;
                if (d_grid_elevation<FAST>(v.grids[v.geoid], lo, s.lat, undulation))
                        s.alt -= undulation;
        }
        for (int layer = 0; layer < v.n_layers; layer++) {
                const int end = v.layer_first[layer + 1];
                int data_index = 0;
                for (int j = v.layer_first[layer]; j < end; j++, data_index++) {
                        const tamd_meta mt = v.metas[j];
                        double elevation;
                        TileFault f = { -1, 0, 0 };
                        const int inside = d_source_elevation<FAST>(v, mt, s.lat, s.lon, elevation, f);
                        if (inside < 0) { /* the layers cannot be told without that tile */
                                s.fault = f;
                                s.m = -1, s.k = -1;
                                return;
                        }
                        if (s.slot < 0) s.slot = f.centre;
                        if (inside == 0) continue;
                        elevation += mt.offset; /* [ref stepper.c:737] */
                        s.k = data_index;
                        if (elevation >= s.alt) { /* [ref stepper.c:690-694] */
                                s.m = layer;
                                s.e1 = elevation;
                                return;
                        }
                        s.m = layer + 1; /* [ref stepper.c:695-699] */
                        s.e0 = elevation;
                        break;
                }
        }
}

template <int MODE, bool FAST = false>
__device__ __forceinline__ void d_sample(const tamd_view & v, const OneCtx & ctx, double x,
    double y, double z, Sample & s, CellCache * cache = nullptr)
{
        if (FAST)
                f_to_geodetic(x, y, z, s.lat, s.lon, s.alt);
        else
                d_to_geodetic(x, y, z, s.lat, s.lon, s.alt);
        d_classify<MODE, FAST>(v, ctx, s, cache);
}

/* A sample of a ray that carries a line: at (x, y, z), which is parameter sl
 * of the line.  Taken from the line if it serves; else by the closed form,
 * which lays a new line through the point (origin there: the caller re-bases
 * its path parameter).  Returns true in that case.  Which of the two happens
 * depends on the ray's own line and sample only. */
template <int MODE>
__device__ __forceinline__ bool f_line_try(const tamd_view & v, const OneCtx & ctx,
    const RayLine & line, double sl, Sample & s, CellCache * cache, bool bracketed = false)
{
        bool serves = line.valid && (fabs(sl) <= kLineRange);
        if (serves) {
                f_line_eval(line, sl, s.lat, s.lon, s.alt);
                d_classify<MODE, true>(v, ctx, s, cache);
                serves = f_line_serves(line, sl, fmin(fabs(s.alt - s.e0), fabs(s.alt - s.e1)), bracketed);
        }
        return serves;
}

template <int MODE>
__device__ __forceinline__ void f_line_relay(const tamd_view & v, const OneCtx & ctx,
    double x, double y, double z, double dx, double dy, double dz, RayLine & line, Sample & s,
    CellCache * cache)
{
        f_to_geodetic(x, y, z, s.lat, s.lon, s.alt, &line, dx, dy, dz);
        d_classify<MODE, true>(v, ctx, s, cache);
}

template <int MODE>
__device__ __forceinline__ bool f_sample_on_line(const tamd_view & v, const OneCtx & ctx,
    double x, double y, double z, double dx, double dy, double dz, RayLine & line, double sl,
    Sample & s, CellCache * cache)
{
        const bool serves = f_line_try<MODE>(v, ctx, line, sl, s, cache);
        if (!serves) f_line_relay<MODE>(v, ctx, x, y, z, dx, dy, dz, line, s, cache);
        return !serves;
}

/* Where a fast trace samples next inside the bracket [ds0, ds1] of a crossing
 * [ref stepper.c:840-860 halves it, 27 times from a metre to 1e-8 m].  c0, c1:
 * the clearances (distance to the nearest boundary, >= 0) of the samples at its
 * two ends -- over a bracket of a metre or less both measure the same boundary:
 * the crossing is where they interpolate to zero (false position, with the
 * Illinois rule: the clearance of an end that has stayed put while the other moved
 * twice is halved, or a bent surface would keep every sample on one side).  The
 * sample is taken 0.4e-8 m to the side of that estimate whose end is the farther
 * one, so that once the estimate is good the two ends close in from both sides:
 * two such samples and the bracket is 0.8e-8 m wide, where the reference's test
 * ends it too, around the same crossing (both brackets hold it, both are below
 * 1e-8 m: the end points agree to that).  A bracket stays a bracket whatever the
 * estimate is worth (a cell's edge, another layer nearby, a first clearance that
 * was only guessed); after kBracketPatience samples the midpoint takes over. */
constexpr int kBracketPatience = 24;
__device__ __forceinline__ double f_bracket_point(double ds0, double ds1, double c0, double c1,
    int taken)
{
        const double w = ds1 - ds0;
        const double sum = c0 + c1;
        double t = 0.5 * (ds0 + ds1);
        if ((taken < kBracketPatience) && (sum > 0.) && (sum < 1e30) && (w > 2.5e-8)) {
                const double r = ds0 + w * (c0 / sum);
                const double aim = ((r - ds0) >= (ds1 - r)) ? r - 0.4e-8 : r + 0.4e-8;
                t = fmin(fmax(aim, ds0 + 0.25e-8), ds1 - 0.25e-8);
        }
        return t;
}

/* b + d t, a ray's next position [ref stepper.c:824, :862-863].  The reference rounds
 * the product and then the sum; the fast arithmetic of a trace fuses them.  The
 * sum's rounding (an ulp of 6.4e6 m: 9e-10 m) is what accumulates into the drift
 * that kLineDrift prices, the same either way; the product's (1e-16 of the step)
 * changes which way the sum rounds once in 1e7 steps. */
template <bool FAST>
__device__ __forceinline__ double d_along(double b, double d, double t)
{
        return FAST ? __builtin_fma(d, t, b) : b + d * t;
}

/* [ref stepper.c:799-813] tentative step length from the last sample */
__device__ __forceinline__ double d_step_length(
    const tamd_view & v, double alt, double e0, double e1, int m)
{
        double ds = 0.;
        if (m != 0) {
                const double dsi = fabs(alt - e0);
                if ((dsi < ds) || (ds <= 0.)) ds = dsi;
        }
        if (m != v.n_layers) {
                const double dsi = fabs(alt - e1);
                if ((dsi < ds) || (ds <= 0.)) ds = dsi;
        }
        ds *= v.slope;
        if (ds < v.resolution) ds = v.resolution;
        return ds;
}

/* ======================================================================== */
/*                                 kernels                                  */
/* ======================================================================== */

__global__ void k_ecef_from_geodetic(long n, const double * __restrict__ lat,
    const double * __restrict__ lon, const double * __restrict__ elev,
    double * __restrict__ ecef)
{
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n;
             r += (long)gridDim.x * blockDim.x) {
                double x, y, z;
                d_from_geodetic(lat[r], lon[r], elev[r], x, y, z);
                ecef[3 * r] = x, ecef[3 * r + 1] = y, ecef[3 * r + 2] = z;
        }
}

template <bool FAST>
__global__ void k_ecef_to_geodetic(long n, const double * __restrict__ ecef,
    double * __restrict__ lat, double * __restrict__ lon, double * __restrict__ alt)
{
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n;
             r += (long)gridDim.x * blockDim.x) {
                double la, lo, al;
                if (FAST)
                        f_to_geodetic(ecef[3 * r], ecef[3 * r + 1], ecef[3 * r + 2], la, lo, al);
                else
                        d_to_geodetic(ecef[3 * r], ecef[3 * r + 1], ecef[3 * r + 2], la, lo, al);
                if (lat) lat[r] = la;
                if (lon) lon[r] = lo;
                if (alt) alt[r] = al;
        }
}

/* [ref ecef.c:160-176] */
__global__ void k_ecef_from_horizontal(long n, const double * __restrict__ lat,
    const double * __restrict__ lon, const double * __restrict__ az_,
    const double * __restrict__ el_, double * __restrict__ dir)
{
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n;
             r += (long)gridDim.x * blockDim.x) {
                double e[3], nn[3], u[3];
                d_enu(lat[r], lon[r], e, nn, u);
                const double az = az_[r] * kPi / 180.;
                const double el = el_[r] * kPi / 180.;
                const double ce = cos(el);
                const double q0 = ce * sin(az), q1 = ce * cos(az), q2 = sin(el);
                for (int i = 0; i < 3; i++)
                        dir[3 * r + i] = q0 * e[i] + q1 * nn[i] + q2 * u[i];
        }
}

/* [ref ecef.c:178-207] */
__global__ void k_ecef_to_horizontal(long n, const double * __restrict__ lat,
    const double * __restrict__ lon, const double * __restrict__ dir,
    double * __restrict__ az, double * __restrict__ el)
{
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n;
             r += (long)gridDim.x * blockDim.x) {
                double e[3], nn[3], u[3];
                d_enu(lat[r], lon[r], e, nn, u);
                const double d0 = dir[3 * r], d1 = dir[3 * r + 1], d2 = dir[3 * r + 2];
                const double x = e[0] * d0 + e[1] * d1 + e[2] * d2;
                const double y = nn[0] * d0 + nn[1] * d1 + nn[2] * d2;
                const double z = u[0] * d0 + u[1] * d1 + u[2] * d2;
                double rr = d0 * d0 + d1 * d1 + d2 * d2;
                if (rr <= FLT_EPSILON) continue; /* outputs untouched [ref ecef.c:194] */
                rr = sqrt(rr);
                if (az) az[r] = atan2(x, y) * 180. / kPi;
                if (el) {
                        const double arg = z / rr;
                        el[r] = (arg > 1.) ? 90. :
                                             ((arg < -1.) ? -90. : asin(arg) * 180. / kPi);
                }
        }
}

__global__ void k_project(tamd_proj pr, int inverse, long n, const double * __restrict__ a,
    const double * __restrict__ b, double * __restrict__ c, double * __restrict__ d)
{
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n;
             r += (long)gridDim.x * blockDim.x) {
                double u, w;
                if (inverse)
                        d_unproject(pr, a[r], b[r], u, w);
                else
                        d_project(pr, a[r], b[r], u, w);
                c[r] = u, d[r] = w;
        }
}

/* One round of a batch call over a geometry with paged tiles (see tile_fault).
 * ids / n_in: the rays or points of this round (NULL: all of 0 .. n-1, the first
 * round); faulted / n_faulted / wanted: where to list those that need a tile
 * that is not resident, and a bitmap, over the tile table, of the tiles they
 * need.  All NULL for a geometry with every tile resident. */
typedef struct tamd_paging Paging;

/* the whole wave calls: lists the lanes with a fault (one atomic per wave), and
 * counts, tile by tile, how many listed items want it (the host keeps the tiles
 * in demand) -- the tiles of the very first item of the list go into the
 * bitmap `wanted_first` too: the host serves that one without fail, so that
 * every round completes at least one item whatever the others compete for */
__device__ __forceinline__ void page_fault(const Paging & pg, const TileFault & f, long r,
    int also = -1 /* one more tile the item wants kept: where it resumes from */)
{
        const bool fault = f.centre >= 0;
        const ull mask = __ballot(fault);
        if (mask == 0) return;
        const int leader = __builtin_ctzll(mask);
        ull base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(pg.n_faulted, (ull)__popcll(mask));
        base = __shfl(base, leader, 64);
        if (fault) {
                const int rank = __builtin_amdgcn_mbcnt_hi(
                    (unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                pg.faulted[base + rank] = (int)r;
                /* the item the host serves without fail: the one it named, else the
                 * first of this list (which it will name from the next round on) */
                const bool first = (pg.first_id >= 0) ? (r == (long)pg.first_id) : ((base + rank) == 0);
                for (int b = 0; b < 9; b++) {
                        if (!((f.mask >> b) & 1)) continue;
                        const int t = f.centre + (b / 3 - 1) * f.stride + (b % 3 - 1);
                        if (b != 4) atomicAdd(&pg.wanted[(size_t)t * TAMD_DEMAND_STRIDE], 1u); /* (the centre: below) */
                        if (first) atomicOr(&pg.wanted_first[t >> 5], 1u << (t & 31));
                }
                if (also >= 0) {
                        atomicAdd(&pg.wanted[(size_t)also * TAMD_DEMAND_STRIDE], 1u);
                        if (first) atomicOr(&pg.wanted_first[also >> 5], 1u << (also & 31));
                }
        }
        /* the tile an item is IN: one addition per wave and tile (the rays of a batch that start
         * over tiles not resident fault together, a wave at a time, on a handful of tiles) */
        const bool centre = fault && (((f.mask >> 4) & 1) != 0);
        ull left = __ballot(centre);
        while (left != 0) {
                const int lead = __builtin_ctzll(left);
                const int t0 = __shfl(f.centre, lead, 64);
                const ull same = __ballot(centre && (f.centre == t0));
                if ((int)(threadIdx.x & 63) == lead)
                        atomicAdd(&pg.wanted[(size_t)t0 * TAMD_DEMAND_STRIDE], (unsigned)__popcll(same));
                left &= ~same;
        }
}

/* Loop of the one-thread-per-item kernels: whole waves go round (page_fault is
 * a wave-wide call); `r` is the item of this lane, or -1 */
#define PAGED_ITEMS(pg, n, i0, r)                                                              \
        const long n_items_ = ((pg).n_in != nullptr) ? (long)*(pg).n_in : (n);                 \
        for (long i0 = blockIdx.x * (long)blockDim.x; i0 < n_items_;                           \
             i0 += (long)gridDim.x * blockDim.x)                                               \
                for (long i_ = i0 + threadIdx.x, r = (i_ < n_items_) ?                         \
                             (((pg).ids != nullptr) ? (long)(pg).ids[i_] : i_) : -1, once_ = 1; \
                     once_; once_ = 0)

/* Elevation of n points on the view's first meta.  For a MAP the arguments
 * are (x, y) [ref map.c:380-385]; for a STACK (latitude, longitude)
 * [ref stack.c:338-361]. */
__global__ void k_elevation(tamd_view v, long n, const double * __restrict__ a,
    const double * __restrict__ b, double * __restrict__ z, int * __restrict__ inside, Paging pg)
{
        const tamd_meta mt = v.metas[0];
        PAGED_ITEMS(pg, n, i0, r)
        {
                int in = 0;
                TileFault f = { -1, 0, 0 };
                if (r >= 0) {
                        double zz = 0.;
                        if (mt.kind == TAMD_MAP)
                                in = d_grid_elevation(v.grids[mt.src], a[r], b[r], zz) ? 1 : 0;
                        else
                                in = d_stack_elevation(v, v.stacks[mt.src], a[r], b[r], zz, f);
                        if (in >= 0) f.centre = -1;
                        if (in >= 0) {
                                /* an outside point leaves a MAP's z untouched in the
                                 * reference and zeroes a STACK's; report 0 for both */
                                z[r] = in ? zz : 0.;
                                inside[r] = in;
                        }
                }
                if (pg.faulted != nullptr) page_fault(pg, f, r);
        }
}

/* Gradient of n points on the view's first meta: a MAP takes (x, y) and
 * returns (gx, gy) [ref map.c:387-392]; a STACK takes (latitude, longitude)
 * and returns (glat, glon) [ref stack.c:364-388].  Outputs are in-out. */
__global__ void k_gradient(tamd_view v, long n, const double * __restrict__ a,
    const double * __restrict__ b, double * __restrict__ ga, double * __restrict__ gb,
    int * __restrict__ inside, Paging pg)
{
        const tamd_meta mt = v.metas[0];
        PAGED_ITEMS(pg, n, i0, r)
        {
                int tile = 0;
                TileFault f = { -1, 0, 0 };
                if (r >= 0) {
                        bool in = false;
                        if (mt.kind == TAMD_MAP) {
                                double gx = ga[r], gy = gb[r];
                                in = d_grid_gradient(v.grids[mt.src], a[r], b[r], gx, gy);
                                ga[r] = gx, gb[r] = gy;
                        } else {
                                tile = d_stack_tile(v, v.stacks[mt.src], a[r], b[r], f);
                                if (tile != kTileFault) f.centre = -1;
                                if (tile == -1) {
                                        ga[r] = gb[r] = 0.; /* [ref stack.c:378-382] */
                                } else if (tile >= 0) {
                                        double glat = ga[r], glon = gb[r];
                                        /* x = longitude, y = latitude; gx -> glon, gy -> glat */
                                        in = d_grid_gradient(v.grids[tile], b[r], a[r], glon, glat);
                                        ga[r] = glat, gb[r] = glon;
                                }
                        }
                        if (tile != kTileFault) inside[r] = in ? 1 : 0;
                }
                if (pg.faulted != nullptr) page_fault(pg, f, r);
        }
}

/* [ref stepper.c:877-931] */
__global__ void k_position(tamd_view v, long n, const double * __restrict__ lat,
    const double * __restrict__ lon, const double * __restrict__ height, int layer,
    double * __restrict__ pos, int * __restrict__ data_index, Paging pg)
{
        PAGED_ITEMS(pg, n, i0, r)
        {
                TileFault fault = { -1, 0, 0 };
                if (r >= 0) {
                        const double la = lat[r], lo = lon[r];
                        int found = -1, di = 0;
                        double elevation = 0.;
                        const int end = v.layer_first[layer + 1];
                        for (int j = v.layer_first[layer]; j < end; j++, di++) {
                                const tamd_meta mt = v.metas[j];
                                TileFault f;
                                const int in = d_source_elevation(v, mt, la, lo, elevation, f);
                                if (in < 0) {
                                        fault = f;
                                        break;
                                }
                                if (in == 0) continue;
                                elevation += mt.offset;
                                if (v.geoid >= 0) { /* [ref stepper.c:905-914] */
                                        double undulation;
                                        const double l360 = (lo >= 0) ? lo : lo + 360.;
                                        if (d_grid_elevation(
                                                v.grids[v.geoid], l360, la, undulation))
                                                elevation += undulation;
                                }
                                found = di;
                                break;
                        }
                        if (fault.centre < 0) {
                                data_index[r] = found;
                                if (found >= 0) {
                                        double x, y, z;
                                        d_from_geodetic(la, lo, elevation + height[r], x, y, z);
                                        pos[3 * r] = x, pos[3 * r + 1] = y, pos[3 * r + 2] = z;
                                }
                        }
                }
                if (pg.faulted != nullptr) page_fault(pg, fault, r);
        }
}

__device__ __forceinline__ ull wave_sum(ull v)
{
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        return v;
}

/* Add a block's four counters to the global ones: waves -> LDS -> 4 atomics per
 * block.  (One set of atomics per WAVE is what a kernel can least afford: the
 * counters share a cache line, same-line atomics complete at ~5 ns apiece, and
 * the step kernel's 8 192 waves spent 4x their run time queueing there.) */
__device__ __forceinline__ void block_tally(ull * __restrict__ stats, ull a, ull b, ull c, ull d)
{
        __shared__ ull part[4][4]; /* [wave][counter]: blocks are 256 threads */
        a = wave_sum(a), b = wave_sum(b), c = wave_sum(c), d = wave_sum(d);
        const int wave = (int)(threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0) part[wave][0] = a, part[wave][1] = b, part[wave][2] = c, part[wave][3] = d;
        __syncthreads();
        if (threadIdx.x < 4) {
                const ull sum = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] +
                    part[3][threadIdx.x];
                if (sum != 0) atomicAdd(&stats[threadIdx.x], sum);
        }
}

/* ---- counter-based random directions (scattering harness, config C5) ------
 * Philox-4x32-10 (Salmon et al., SC'11): counter = (ray id, stream), key =
 * seed.  One block of four 32-bit words gives two 53-bit uniforms, mapped to
 * an isotropic unit vector.  Any (ray, stream) pair can be regenerated
 * anywhere, so shards need no shared RNG state. */
__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1)
{
        for (int round = 0; round < 10; round++) {
                const unsigned long long p0 = 0xD2511F53ull * c[0];
                const unsigned long long p1 = 0xCD9E8D57ull * c[2];
                const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0;
                const unsigned n1 = (unsigned)p1;
                const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
                const unsigned n3 = (unsigned)p0;
                c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
                k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
        }
}

/* sin and cos of 2 pi u, 0 <= u < 1, to the last ulp or so: the quadrant comes off
 * u exactly (k = round(4 u), r = u - k / 4 in [-1/8, 1/8]: no rounding), the rest is
 * the classic pair of polynomials on [-pi/4, pi/4] (fdlibm's k_sin / k_cos
 * coefficients) and a swap / sign by quadrant.  ~30 instructions where OCML's
 * sin + cos take ~250 with their large-argument paths: the walk kernel draws a
 * direction per step and is bound by the rate its instructions issue at. */
__device__ __forceinline__ void d_sincos_2pi(double u, double & s, double & c)
{
        const double k = __builtin_rint(4. * u); /* 0 .. 4 */
        const double t = 6.283185307179586 * __builtin_fma(-0.25, k, u);
        const double z = t * t;
        double ps = 1.58969099521155010221e-10;
        ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
        ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
        ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
        ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
        ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
        const double sn = __builtin_fma(t * z, ps, t);
        double pc = -1.13596475577881948265e-11;
        pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
        pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
        pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
        pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
        pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
        const double cs = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.));
        const int q = (int)k & 3;
        /* 2 pi u = q pi / 2 + t */
        s = (q == 0) ? sn : ((q == 1) ? cs : ((q == 2) ? -sn : -cs));
        c = (q == 0) ? cs : ((q == 1) ? -sn : ((q == 2) ? -cs : sn));
}

/* the isotropic unit vector of (ray id, stream; seed): see k_isotropic */
__device__ __forceinline__ void d_isotropic(ull id, ull stream, ull seed, double & x, double & y, double & z)
{
        unsigned c[4] = { (unsigned)id, (unsigned)(id >> 32), (unsigned)stream, (unsigned)(stream >> 32) };
        philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
        const double scale = 1. / 9007199254740992.; /* 2^-53 */
        const double u1 = (double)(((ull)(c[0] >> 5) << 26) | (c[1] >> 6)) * scale;
        const double u2 = (double)(((ull)(c[2] >> 5) << 26) | (c[3] >> 6)) * scale;
        const double ct = 2. * u1 - 1.;
        const double st = sqrt(1. - ct * ct);
        double sp, cp;
        d_sincos_2pi(u2, sp, cp);
        x = st * cp, y = st * sp, z = ct;
}

/* Where a batch of single steps takes its directions from: an array, or -- a
 * scattering walk (turtle_stepper_scatter_n) -- Philox(first + ray, stream; seed)
 * drawn in the kernel (the 48 bytes a ray's direction costs to write and read
 * back are a third of what a step moves); and what it adds up per ray. */
struct StepWalk {
        int on;
        ull seed, stream;
        long first;
        double * length; /* += the length of the step */
        int * steps;     /* += 1 */
};

/* What a batch of single steps defers to its second pass: the rays that
 * crossed a boundary (~2-5 % of them), and the tentative length of each. */
struct CrossList {
        int * ray;      /* NULL: the step kernel bisects in place */
        double * ds;
        ull * count;
        int * other;    /* traces: the medium and data index of the sample that crossed,
                         * packed (cross_pack); batches of single steps: NULL */
};

/* (medium, data index) of a sample, each -1 .. 65 534, in one int */
__device__ __forceinline__ int cross_pack(int m, int k) { return (m + 1) | ((k + 1) << 16); }
__device__ __forceinline__ void cross_unpack(int packed, int & m, int & k)
{
        m = (packed & 0xffff) - 1, k = (int)((unsigned)packed >> 16) - 1;
}

/* One turtle_stepper_step per thread [ref stepper.c:780-875].  With a direction
 * a step is one sample at the tentative position -- and, for the few rays whose
 * medium changed there, a ~27-sample bisection.  Run inside this kernel that
 * loop would idle the other lanes of every wave that holds such a ray (nearly
 * all of them do), so the kernel only LISTS those rays (position moved to the
 * tentative point, medium unchanged) and k_bisect finishes them, packed: a batch
 * of n single steps costs n + 27 x (rays that crossed) samples, all at full
 * lane occupancy, and exposes n-way parallelism to the gathers (a scattering
 * Monte-Carlo over a 2.6 GB mosaic is bound by their latency).
 * stats (or NULL): rays, steps, samples, steps that did not cross. */
template <int MODE, bool FAST>
__device__ __forceinline__ void step_items(const tamd_view & v, long n,
    double * __restrict__ pos, const double * __restrict__ dir,
    double * __restrict__ lat, double * __restrict__ lon, double * __restrict__ alt,
    double * __restrict__ elev, double * __restrict__ step, int * __restrict__ index,
    int flags, CrossList cross, Paging pg, ull * __restrict__ stats, StepWalk walk)
{
        OneCtx ctx;
        d_load_ctx<MODE, FAST>(v, ctx);
        const bool directed = (dir != nullptr) || walk.on;
        ull my_rays = 0, my_steps = 0, my_samples = 0, my_plain = 0;
        PAGED_ITEMS(pg, n, i0, r) /* whole waves go round: see the listings */
        {
                bool listed = false;
                double listed_ds = 0.;
                TileFault fault = { -1, 0, 0 }; /* tiles to page in: the ray is left as it is, and listed */
                int home = -1;
                /* TAMD_STEP_COMPACT (turtle_stepper_walk_n): all a step resumes from, besides the
                 * medium, is the tentative length the last sample gave it [ref stepper.c:799-813:
                 * what the cached sample is read for] -- `alt` holds THAT, one double in and one
                 * out, and the altitude and the two elevations (24 B in, 24 B out) stay in
                 * registers.  The same function of the same values, evaluated when the sample is
                 * taken instead of when the next step begins: the same bits. */
                const bool compact = (flags & TAMD_STEP_COMPACT) != 0;
                double ds_given = -1.;
                /* a walk: a ray that has left the data takes no further step */
                if (walk.on && (r >= 0) && (index[2 * r] < 0)) r = -1;
                if (r >= 0) {
                double px = pos[3 * r], py = pos[3 * r + 1], pz = pos[3 * r + 2];
                Sample s;
                if ((flags & TURTLE_AMD_STEP_RESUME) && directed && (index[2 * r] >= 0)) {
                        /* the caller hands back the sample of this position
                         * [ref stepper.c:708-710, :745-748]; its latitude and
                         * longitude are not read: whatever becomes of the step,
                         * the ones published are the new sample's */
                        s.lat = 0., s.lon = 0.;
                        if (compact) {
                                ds_given = alt[r];
                                s.alt = 0., s.e0 = -DBL_MAX, s.e1 = DBL_MAX;
                        } else {
                                s.alt = alt[r];
                                s.e0 = elev[2 * r], s.e1 = elev[2 * r + 1];
                        }
                        s.m = index[2 * r], s.k = index[2 * r + 1];
                } else {
                        d_sample<MODE, FAST>(v, ctx, px, py, pz, s);
                        my_samples++;
                        fault = s.fault;
                        home = s.slot;
                }

                double ds = 0.;
                if ((s.m >= 0) && (fault.centre < 0)) {
                        ds = (ds_given >= 0.) ? ds_given : d_step_length(v, s.alt, s.e0, s.e1, s.m);
                        if (directed) {
                                double dx, dy, dz;
                                if (walk.on)
                                        d_isotropic((ull)(walk.first + r), walk.stream, walk.seed, dx, dy, dz);
                                else
                                        dx = dir[3 * r], dy = dir[3 * r + 1], dz = dir[3 * r + 2];
                                px += dx * ds, py += dy * ds, pz += dz * ds;
                                const int medium0 = s.m, data0 = s.k;
                                Sample s1;
                                d_sample<MODE, FAST>(v, ctx, px, py, pz, s1);
                                my_samples++, my_steps++;
                                fault = s1.fault;
                                if (fault.centre >= 0) {
                                        /* nothing */
                                } else if (s1.m == medium0) {
                                        s = s1;
                                        my_plain++;
                                } else if (cross.ray != nullptr) {
                                        /* [ref stepper.c:832-838] the second pass
                                         * starts from here: position moved, the
                                         * medium and data index still the old ones */
                                        listed = true, listed_ds = ds;
                                        s.m = medium0, s.k = data0;
                                } else { /* [ref stepper.c:832-864] */
                                        double ds0 = -ds, ds1 = 0.;
                                        s = s1;
                                        while (ds1 - ds0 > 1E-08) {
                                                const double ds2 = 0.5 * (ds0 + ds1);
                                                Sample s2;
                                                d_sample<MODE, FAST>(v, ctx, px + dx * ds2,
                                                    py + dy * ds2, pz + dz * ds2, s2);
                                                my_samples++;
                                                if (s2.fault.centre >= 0) {
                                                        fault = s2.fault;
                                                        break;
                                                }
                                                if (s2.m == medium0)
                                                        ds0 = ds2;
                                                else {
                                                        ds1 = ds2;
                                                        s = s2;
                                                }
                                        }
                                        ds += ds1;
                                        px += dx * ds1, py += dy * ds1, pz += dz * ds1;
                                }
                                if (fault.centre < 0) pos[3 * r] = px, pos[3 * r + 1] = py, pos[3 * r + 2] = pz;
                                /* (a listed ray: k_bisect's, with the length it ends up with) */
                                if (walk.on && !listed && (fault.centre < 0)) walk.length[r] += ds, walk.steps[r] += 1;
                        }
                }
                if (fault.centre >= 0) listed = false;
                if (!listed && (fault.centre < 0)) my_rays++;
                /* sample_publish [ref stepper.c:758-778] (a listed ray: k_bisect's) */
                if (!listed && (fault.centre < 0)) {
                        if (lat) lat[r] = s.lat;
                        if (lon) lon[r] = s.lon;
                        if (alt) alt[r] = !compact ? s.alt : ((s.m >= 0) ? d_step_length(v, s.alt, s.e0, s.e1, s.m) : 0.);
                        if (elev && !compact) {
                                elev[2 * r] = (s.m >= 0) ? s.e0 : 0.;
                                elev[2 * r + 1] = (s.m >= 0) ? s.e1 : 0.;
                        }
                        if (step) step[r] = ds;
                }
                if (fault.centre < 0) index[2 * r] = s.m, index[2 * r + 1] = s.k;
                }
                if (pg.faulted != nullptr) page_fault(pg, fault, r, home);
                /* list the rays that crossed: one atomic per wave */
                if (cross.ray != nullptr) {
                        const ull mask = __ballot(listed);
                        if (mask != 0) {
                                const int leader = __builtin_ctzll(mask);
                                ull base = 0;
                                if ((int)(threadIdx.x & 63) == leader)
                                        base = atomicAdd(cross.count, (ull)__popcll(mask));
                                base = __shfl(base, leader, 64);
                                if (listed) {
                                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                            __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                                        cross.ray[base + rank] = (int)r;
                                        cross.ds[base + rank] = listed_ds;
                                }
                        }
                }
        }
        if (stats != nullptr) block_tally(stats, my_rays, my_steps, my_samples, my_plain);
}

/* The kernel proper, once per arithmetic.  A batch of single steps is bound by
 * the latency of its dependent loads (ray state -> nodes), so waves in flight
 * count: the fast-math body of the specialised modes fits 128 registers (4 waves
 * per SIMD instead of 3: -7 %, measured on C5) when told to; the strict one would
 * spill 230 bytes a lane for it (+23 %), and so would the fast body of the
 * generic mode: they are left alone. */
#define STEP_ARGS                                                                              \
        tamd_view v, long n, double * __restrict__ pos, const double * __restrict__ dir,       \
            double * __restrict__ lat, double * __restrict__ lon, double * __restrict__ alt,   \
            double * __restrict__ elev, double * __restrict__ step, int * __restrict__ index,  \
            int flags, CrossList cross, Paging pg, ull * __restrict__ stats, StepWalk walk
#define STEP_PASS v, n, pos, dir, lat, lon, alt, elev, step, index, flags, cross, pg, stats, walk
template <int MODE, bool FAST>
__global__ void __launch_bounds__(256) k_step(STEP_ARGS)
{
        step_items<MODE, FAST>(STEP_PASS);
}
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_step_fast(STEP_ARGS)
{
        step_items<MODE, true>(STEP_PASS);
}
#undef STEP_ARGS
#undef STEP_PASS

/* The second pass of a batch of single steps: the bisection of the listed rays
 * [ref stepper.c:836-864], every lane busy.  The ray's position is the tentative
 * point q, its index the medium it left; the first sample (at q again: the
 * same arithmetic on the same point as in the first pass) is the first
 * candidate for the medium it entered. */
template <int MODE, bool FAST>
__global__ void __launch_bounds__(256) k_bisect(tamd_view v, double * __restrict__ pos,
    const double * __restrict__ dir, double * __restrict__ lat, double * __restrict__ lon,
    double * __restrict__ alt, double * __restrict__ elev, double * __restrict__ step,
    int * __restrict__ index, int flags, CrossList cross, Paging pg, ull * __restrict__ stats, StepWalk walk)
{
        OneCtx ctx;
        d_load_ctx<MODE, FAST>(v, ctx);
        const bool compact = (flags & TAMD_STEP_COMPACT) != 0; /* see step_items */
        const long n = (long)*cross.count;
        ull my_rays = 0, my_samples = 0;
        for (long i0 = blockIdx.x * (long)blockDim.x; i0 < n; i0 += (long)gridDim.x * blockDim.x) {
                const long i = i0 + threadIdx.x; /* whole waves go round (page_fault) */
                TileFault fault = { -1, 0, 0 };
                int home = -1;
                long r = -1;
                if (i < n) {
                r = cross.ray[i];
                double ds = cross.ds[i];
                double px = pos[3 * r], py = pos[3 * r + 1], pz = pos[3 * r + 2];
                double dx, dy, dz;
                if (walk.on)
                        d_isotropic((ull)(walk.first + r), walk.stream, walk.seed, dx, dy, dz);
                else
                        dx = dir[3 * r], dy = dir[3 * r + 1], dz = dir[3 * r + 2];
                const int medium0 = index[2 * r];
                CellCache cell = { ~0u, 0u, 0u, -1, nullptr };
                CellCache * cache = (FAST && (MODE != TAMD_MODE_GENERIC)) ? &cell : nullptr;
                Sample s;
                /* fast math: the bracket is a segment of the ray behind q, so the
                 * ~27 samples come from the line laid at q (see RayLine) */
                RayLine line;
                double at = 0.; /* q's parameter on the line */
                if (FAST) {
                        f_to_geodetic(px, py, pz, s.lat, s.lon, s.alt, &line, dx, dy, dz);
                        d_classify<MODE, true>(v, ctx, s, cache);
                } else
                        d_sample<MODE, false>(v, ctx, px, py, pz, s, cache);
                fault = s.fault;
                home = s.slot;
                double ds0 = -ds, ds1 = 0.;
                int halvings = 0;
                while ((fault.centre < 0) && (ds1 - ds0 > 1E-08) && (halvings++ <= 1200)) {
                        const double ds2 = 0.5 * (ds0 + ds1);
                        Sample s2;
                        if (FAST) {
                                if (f_sample_on_line<MODE>(v, ctx, px + dx * ds2, py + dy * ds2,
                                        pz + dz * ds2, dx, dy, dz, line, at + ds2, s2, cache))
                                        at = -ds2; /* a new line, laid at this sample */
                        } else
                                d_sample<MODE, false>(v, ctx, px + dx * ds2, py + dy * ds2,
                                    pz + dz * ds2, s2, cache);
                        my_samples++;
                        if (s2.fault.centre >= 0) {
                                fault = s2.fault;
                        } else if (s2.m == medium0)
                                ds0 = ds2;
                        else {
                                ds1 = ds2;
                                s = s2;
                        }
                }
                if (fault.centre >= 0) {
                        /* a tile to page in (the bracket straddles a third tile): the
                         * ray goes back before its step and is listed for the next
                         * round, which takes it from k_step again */
                        pos[3 * r] = px - dx * ds, pos[3 * r + 1] = py - dy * ds, pos[3 * r + 2] = pz - dz * ds;
                } else {
                ds += ds1;
                pos[3 * r] = px + dx * ds1, pos[3 * r + 1] = py + dy * ds1, pos[3 * r + 2] = pz + dz * ds1;
                if (lat) lat[r] = s.lat;
                if (lon) lon[r] = s.lon;
                if (alt) alt[r] = !compact ? s.alt : ((s.m >= 0) ? d_step_length(v, s.alt, s.e0, s.e1, s.m) : 0.);
                if (elev && !compact) {
                        elev[2 * r] = (s.m >= 0) ? s.e0 : 0.;
                        elev[2 * r + 1] = (s.m >= 0) ? s.e1 : 0.;
                }
                if (step) step[r] = ds;
                if (walk.on) walk.length[r] += ds, walk.steps[r] += 1;
                index[2 * r] = s.m, index[2 * r + 1] = s.k;
                my_rays++;
                }
                }
                if (pg.faulted != nullptr) page_fault(pg, fault, r, home);
        }
        if (stats != nullptr) block_tally(stats, my_rays, 0, my_samples, 0);
}

/* ---- the hot kernel ---------------------------------------------------- */

constexpr int kChunk = 64; /* rays a wave draws from the global queue at once */
#ifndef PAGED_REFILL
#define PAGED_REFILL 24
#endif
constexpr int kPagedRefill = PAGED_REFILL; /* over paged tiles: the free lanes a wave waits for before it takes new rays */
constexpr int kTailChunk = 8; /* ... in the last phase of a fast trace: few, and long */
constexpr int kCreepLanes = 8; /* the creep loop engages at or below this many live lanes */
#ifndef CREEP_UNROLL
#define CREEP_UNROLL 32
#endif
constexpr int kCreepUnroll = CREEP_UNROLL; /* steps per trip of the one-map creep loop */
#ifndef CREEP_BACKOFF
#define CREEP_BACKOFF 8
#endif
constexpr int kCreepBackoff = CREEP_BACKOFF; /* general iterations a busy wave waits after a useless group */
#ifndef TRACE_RELAY_BATCH
#define TRACE_RELAY_BATCH 12
#endif
#ifndef TRACE_RELAY_PATIENCE
#define TRACE_RELAY_PATIENCE 3
#endif
constexpr int kRelayBatch = TRACE_RELAY_BATCH;       /* lanes of a busy wave that a closed form waits for ... */
constexpr int kRelayPatience = TRACE_RELAY_PATIENCE; /* ... at most this many general iterations */

enum { ST_INIT = 0, ST_STEP = 1, ST_BISECT = 2 };

/* Persistent waves; one ray per lane; ONE sample per lane per iteration.
 *
 * A ray alternates between optimistic steps (one sample each) and, once, a
 * ~20-30 sample bisection that locates the boundary it crossed.  Run as the
 * reference writes it (a while loop inside the step) the bisection would
 * idle the other 63 lanes of the wave.  Instead each lane carries a small
 * state machine (INIT -> STEP -> BISECT -> done) and every trip round the loop
 * evaluates exactly one sample for every live lane, whatever its state: the
 * expensive part (ECEF->geodetic + layer lookup) is always executed with a
 * full exec mask, and only the cheap bookkeeping diverges.
 *
 * Every state samples at  q = B + d * t :
 *   INIT    t = 0                      B = the ray's origin
 *   STEP    t = ds (tentative length)  B = last accepted position
 *   BISECT  t = (ds0 + ds1) / 2        B = the tentative position that crossed
 * and a STEP sample always moves B to q (accepted, or the bisection's origin),
 * so the per-lane state is B, d, three step scalars, the path length, the
 * step count and three small integers.  The arithmetic per ray is exactly
 * that of calling turtle_stepper_step in a loop [ref stepper.c:780-875].
 *
 * Finished lanes are refilled from a global ray queue: lanes that need a ray
 * are ranked with ballot/mbcnt, and the wave draws kChunk ray ids at a time
 * with one atomic (wave-aggregated), so the queue sees n / 64 atomics.
 *
 * Results do not depend on which lane runs a ray (rays are independent), so
 * the output is deterministic. */
/* TRACE_CARRY_MEDIUM: the caller knows which medium each ray is in (a trace
 * resumed after a boundary, or after parking) */
enum { TRACE_CARRY_MEDIUM = 1 };

/* ---- a workgroup's pool of rays in LDS (round 4) --------------------------
 *
 * The lined pass alternates two kinds of work that want different lanes: LEAN
 * steps (a ray on its line over one grid: ~45 vector instructions a step, for
 * the lanes whose line and cell still serve) and GENERAL iterations (a closed
 * form that lays a new line, a new ray's first sample, a crossing to list:
 * ~1 000 instructions for the whole wave, whoever needs them).  With a ray
 * fixed to its lane, a group of lean steps ran for the lanes still on their
 * lines while the others waited, and a general iteration for the lanes that
 * needed one while the others took a single step or none: measured on C2
 * (round 3) 45 % of the lanes occupied and 5.6 x the wave-instructions the lean
 * steps alone would take.
 *
 * So the four waves of a block exchange rays through LDS.  Before each piece of
 * work a wave looks at what its lanes hold and at the pool and takes a ROLE:
 *   LEAN     its rays that need a general iteration go to the pool's `service`
 *            list, rays that are `ready` to step come out of the pool into the
 *            free lanes, and the wave runs a group of lean steps, all lanes going;
 *   SERVICE  its ready rays go to the pool's `ready` list, rays that wait for a
 *            general iteration come out of the pool (then new rays from the
 *            batch's queue), and the wave runs one general iteration for a full
 *            wave of rays that need it;
 *   AS IS    (the end of a pass: the batch's queue is dry and this wave and the
 *            pool hold less than a wave of rays) nothing goes to the pool, rays
 *            left there come out, and the wave does what it did before round 4.
 * A ray is a record of 27 words (position, direction, path length, step, its
 * line, its cell's nodes, its counters); records move under one lock a block
 * (held for the copy: a few hundred cycles, four waves).  Which lane or wave
 * takes a step of a ray never changed a bit of it (the same functions on the
 * same values), and does not now: `test_ray_pool_changes_no_bit`.
 *
 * A kernel must always end: a wave leaves when the batch's queue is dry, its
 * lanes are empty and so is the pool -- a ray in the pool was put there by a
 * wave that is still running and looks at the pool again before it leaves. */
#ifndef TRACE_POOL
#define TRACE_POOL 1
#endif
#ifndef TRACE_POOL_SLOTS
#define TRACE_POOL_SLOTS 192
#endif
#ifndef TRACE_POOL_LEAN_MIN
#define TRACE_POOL_LEAN_MIN 48
#endif
#ifndef TRACE_POOL_REFILL_FREE
#define TRACE_POOL_REFILL_FREE 64
#endif
/* -DTRACE_POOL_STATS: what the waves of a pooled pass do, summed over the launch (a diagnostic
 * build: scripts/exp_pool_stats.py reads the counters) */
#ifdef TRACE_POOL_STATS
__device__ unsigned long long g_pool_stats[32];
#define PSTAT(i, v) (pstat_[i] += (unsigned long long)(v))
#define PSTAT_CLOCK() __builtin_amdgcn_s_memtime()
#else
#define PSTAT(i, v) ((void)0)
#define PSTAT_CLOCK() 0ull
#endif
constexpr int kPoolSlots = TRACE_POOL_SLOTS;
constexpr int kPoolLeanMin = TRACE_POOL_LEAN_MIN;       /* ready rays for which a wave turns LEAN */
constexpr int kPoolRefillFree = TRACE_POOL_REFILL_FREE; /* free slots below which no new ray is drawn */
constexpr int kPoolDoubles = 23, kPoolInts = 8; /* a ray's record: 216 bytes */
enum { POOL_READY = 0, POOL_SERVICE = 1 };
enum { ROLE_AS_IS = 0, ROLE_LEAN = 1, ROLE_SERVICE = 2 };
struct RayPool {
        double d[kPoolDoubles][kPoolSlots]; /* field by field: a wave's lanes move 64 slots at once */
        int i[kPoolInts][kPoolSlots];
        unsigned short free_slot[kPoolSlots];    /* a stack */
        unsigned short list[2][kPoolSlots];      /* two rings: first in, first out */
        int lock;
        int n_free;
        int head[2], count[2];
};

__device__ __forceinline__ int lane_rank(ull mask)
{
        return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

/* Two-phase launches.  Steps per ray are heavy-tailed (C2: median 163, max
 * 11 327) and a ray's samples are sequential, so a launch lasts as long as its
 * longest ray.  Phase A therefore PARKS any ray that reaches `park_after` steps
 * (its state goes back to the ray arrays, its id to a list) and phase B resumes
 * the parked rays with each ray's line (MODEL; see RayLine), which makes a sample
 * a tenth of the closed form's instructions.
 * Which arithmetic a sample uses depends only on the ray's own step count and
 * positions, never on scheduling: results stay deterministic. */
/* Where a ray's FINAL results go when the passes work on the rays in an order of their own
 * (run_trace, k_ray_cells): the caller's arrays, at the ray's place there.  order_of == NULL: the
 * arrays the passes work on, same place. */
struct RayOut {
        const int * order_of;
        double * pos;
        int * index;
        double * length;
        int * n_steps;
};

struct PhaseIO {
        const int * ids;     /* phase B: the parked ray ids (else NULL: slot == ray) */
        const ull * n_dev;   /* phase B: their number, on the device */
        int * parked;        /* phase A: where to list parked rays (or NULL) */
        ull * n_parked;
        int park_after;      /* hand a ray over to the next phase at this step count (<= 0: never) */
        int accumulate;      /* 1 (phase B): length / n_steps continue from the arrays; 2 (a
                              * later round of a paged geometry): the tentative step too */
        Paging pg;           /* where to list the rays that need a tile paged in (or NULLs) */
        int drain_lanes;     /* phase A: hand over when the queue is dry and the wave is down
                              * to this many rays */
        int line_after;      /* phases B, C: the step count from which a ray steps on its
                              * line (see LINED) */
        int chunk;           /* rays a wave draws from the queue at once */
        int creep_lanes;     /* the creep loop engages at or below this many live lanes */
        int dense_go;        /* ... and above, while at least this many lanes step on (0: never) */
        CrossList cross;     /* CROSS: where to list the rays whose step crossed a boundary */
        /* the sorted hand-over (phase A -> B): the list is filled from both ends -- the
         * rays expected to go on for long from the front, the others from the back --
         * and read front first (see where phase A parks) */
        ull * n_parked_back;    /* A: the number of rays listed from the back (or NULL: one end) */
        const ull * n_dev_back; /* B: the same, to read the list */
        double * ds_mark;       /* A: a ray's step length at step mark_at */
        int mark_at;
        float long_if;          /* A: to the front, if expected to take more further steps than this */
        int pool;               /* B: the waves of a block exchange rays through LDS (RayPool) */
        RayOut out;             /* where a ray that ENDS in this pass leaves its results */
        /* A, rays in an order of the library's own: a new ray comes from the caller's arrays
         * (`out.order_of` says from which place) and leaves its direction in dir_copy for the passes
         * that follow */
        const double * pos_in, * dir_in;
        const int * index_in;
        double * dir_copy;
        unsigned char * sort_key; /* A: the key the hand-over list is ordered by before B reads it, per
                                 * place on the list (or NULL: B reads it as it was filled) */
};

/* CROSS: a ray whose step crossed a boundary is not bisected here (ST_BISECT
 * does not exist then: ~11-27 further samples of ONE lane, each a whole general
 * iteration of its wave -- measured on C2's lined pass: 65 % of the wave-cycles
 * for 15 % of the samples) but handed over as it stands at the tentative point --
 * position, the medium it left, path length and step count in the ray arrays;
 * its id, the tentative length and the medium the sample found in ph.cross -- and
 * k_cross locates every crossing of the batch afterwards, in full waves.  The
 * lane takes a new ray at once. */
template <int MODE, bool FAST, bool MODEL, bool PAGED, bool CROSS, bool POOL = false>
__device__ __forceinline__ void trace_body(const tamd_view & v, long n,
    double * __restrict__ pos, const double * __restrict__ dir, int max_steps,
    int * __restrict__ index, double * __restrict__ length, int * __restrict__ n_steps,
    int flags, const PhaseIO & ph, ull * __restrict__ stats, ull * __restrict__ queue)
{
        const long capacity = n; /* of the lists ph.ids, ph.parked */
        long n_front = n;
        if (ph.n_dev != nullptr) {
                n_front = (long)*ph.n_dev;
                n = n_front + ((ph.n_dev_back != nullptr) ? (long)*ph.n_dev_back : 0);
        }
        /* MODEL: besides its accumulated position B (bx, by, bz: the reference's
         * roundings, in every phase: see kLineTau0) a ray on its line carries
         * line.s, the path length from the point where the line was laid to B */
        RayLine line;
        line.valid = false, line.s = 0., line.tau = kLineTau0;
        /* Which arithmetic a sample uses depends on the ray's step count alone:
         * below ph.line_after the closed form at the accumulated position, as in
         * phase A; from there on the ray's line.  Phase A can then hand a ray over
         * at ANY step (it does, when the queue runs dry: see `drain`) without
         * changing a bit of the result.  LINED: this lane's ray is on its line. */
        bool lined_ = false;
#define LINED (MODEL && lined_)
        /* tiles to page in: stacks only, and only where some are not resident (the
         * bookkeeping costs a wave per SIMD in the one-stack kernel) */
        constexpr bool CAN_FAULT = PAGED && (MODE != TAMD_MODE_ONE_MAP);
        long pool_next = 0, pool_end = 0; /* wave-uniform */
        int creep_wait = 0;                /* wave-uniform: general iterations before a busy wave tries lean steps again */
        int relay_wait = 0;                /* general iterations that lanes of a busy wave have waited for a closed form */
        bool exhausted = false;            /* wave-uniform */
        OneCtx ctx;
        d_load_ctx<MODE, FAST>(v, ctx);
        CellCache cell = { ~0u, 0u, 0u, -1, nullptr };

        long ray = -1;
        bool dead = false;
        int state = ST_INIT, count = 0, count0 = 0;
        double bx = 0, by = 0, bz = 0, dx = 0, dy = 0, dz = 0, len = 0;
        double ds = 0, ds0 = 0, ds1 = 0;
        double c0 = 0, c1 = 0; /* FAST: the clearances at the two ends of the bracket (f_bracket_point) */
        int m = -1, k = -1, bm = -1, bk = -1, halvings = 0;
        int home = -1; /* CAN_FAULT: the tile of the ray's last sample (see Sample.slot) */
        ull my_rays = 0, my_steps = 0, my_samples = 0, my_capped = 0;

        /* ---- the block's pool of rays (see RayPool) ---- */
        constexpr bool POOLED = POOL && (TRACE_POOL != 0) && FAST && MODEL && CROSS && !PAGED &&
            ((MODE == TAMD_MODE_ONE_MAP) || (MODE == TAMD_MODE_ONE_STACK));
        typedef __attribute__((address_space(3))) RayPool * lds_pool_t;
        lds_pool_t P = nullptr;
        bool pooled = false;   /* block-uniform */
        bool stopped_ = false; /* this lane's ray left a group of lean steps: it needs a general iteration */
        int role = ROLE_AS_IS; /* wave-uniform */
        bool pool_empty = true; /* wave-uniform: nothing was left in the pool at the last look */
        int pool_free = kPoolSlots;
        int idle_trips = 0;
#ifdef TRACE_POOL_STATS
        unsigned long long pstat_[24] = { 0 };
#endif
        if constexpr (POOLED) {
                __shared__ RayPool pool_;
                P = (lds_pool_t)&pool_;
                pooled = (ph.pool > 0) && ((MODE == TAMD_MODE_ONE_MAP) || ctx.stack.regular);
                if (pooled) {
                        for (int t = threadIdx.x; t < kPoolSlots; t += 256) P->free_slot[t] = (unsigned short)t;
                        if (threadIdx.x == 0) {
                                P->lock = 0, P->n_free = kPoolSlots;
                                P->head[0] = P->head[1] = 0, P->count[0] = P->count[1] = 0;
                        }
                        __syncthreads();
                }
        }
        /* One look at the pool: the wave takes its role and rays change places.  Whole wave. */
        auto pool_exchange = [&]() {
                if constexpr (POOLED) {
                        const bool has = (ray >= 0);
                        const bool ready = has & !stopped_ & (state == ST_STEP) & lined_ & line.valid &
                            (cell.id != ~0u) & (count + kCreepUnroll < max_steps);
                        const bool waits = has & !ready;
                        const int r_own = __popcll(__ballot(ready)), s_own = __popcll(__ballot(waits));
                        const int e_own = 64 - r_own - s_own;
                        const unsigned long long t_in = PSTAT_CLOCK();
                        (void)t_in;
                        /* the lock: one lane asks, the wave waits */
                        if ((threadIdx.x & 63) == 0) {
                                /* (bounded: a kernel must always end, whatever went wrong) */
                                int expected = 0;
                                for (int spin = 0; (spin < (1 << 22)) &&
                                     !__hip_atomic_compare_exchange_strong(&P->lock, &expected, 1, __ATOMIC_ACQUIRE,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); spin++) {
                                        expected = 0;
                                        __builtin_amdgcn_s_sleep(2);
                                }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        PSTAT(16, PSTAT_CLOCK() - t_in);
                        const int F = __builtin_amdgcn_readfirstlane(P->n_free);
                        const int R = __builtin_amdgcn_readfirstlane(P->count[POOL_READY]);
                        const int S = __builtin_amdgcn_readfirstlane(P->count[POOL_SERVICE]);
                        const int head_r = __builtin_amdgcn_readfirstlane(P->head[POOL_READY]);
                        const int head_s = __builtin_amdgcn_readfirstlane(P->head[POOL_SERVICE]);
                        /* what each role would have to work on */
                        const int lean_out = min(s_own, F), lean_in = min(R, e_own + lean_out);
                        const int lean_n = r_own + lean_in;
                        const int serv_out = min(r_own, F), serv_in = min(S, e_own + serv_out);
                        const int serv_n = s_own + serv_in;
                        if (exhausted && (r_own + s_own + R + S < 64))
                                role = ROLE_AS_IS;
                        else if ((lean_n >= kPoolLeanMin) || (exhausted && (lean_n > serv_n)))
                                role = ROLE_LEAN;
                        else
                                role = ROLE_SERVICE;
                        /* who goes */
                        const bool goes = (role == ROLE_LEAN) ? waits : ((role == ROLE_SERVICE) ? ready : false);
                        const int n_out = (role == ROLE_LEAN) ? lean_out : ((role == ROLE_SERVICE) ? serv_out : 0);
                        const int out_kind = (role == ROLE_LEAN) ? POOL_SERVICE : POOL_READY;
                        const int out_rank = lane_rank(__ballot(goes));
                        const bool push = goes & (out_rank < n_out);
                        int out_slot = 0;
                        if (push) out_slot = P->free_slot[F - 1 - out_rank];
                        /* who comes: into the lanes that are or become free -- ready rays for a
                         * lean wave, waiting ones for a serving wave, either kind at the end */
                        const bool free_lane = !has | push;
                        const int in_rank = lane_rank(__ballot(free_lane));
                        int n_in_r = 0, n_in_s = 0;
                        if (role == ROLE_LEAN) n_in_r = lean_in;
                        if (role == ROLE_SERVICE) n_in_s = serv_in;
                        if (role == ROLE_AS_IS) n_in_r = min(R, e_own), n_in_s = min(S, e_own - n_in_r);
                        const bool pull_r = free_lane & (in_rank < n_in_r);
                        const bool pull_s = free_lane & !pull_r & (in_rank < n_in_r + n_in_s);
                        int in_slot = 0;
                        if (pull_r) in_slot = P->list[POOL_READY][(head_r + in_rank) % kPoolSlots];
                        if (pull_s) in_slot = P->list[POOL_SERVICE][(head_s + in_rank - n_in_r) % kPoolSlots];
                        if (push) {
                                const int q = out_slot;
                                P->d[0][q] = bx, P->d[1][q] = by, P->d[2][q] = bz;
                                P->d[3][q] = dx, P->d[4][q] = dy, P->d[5][q] = dz;
                                P->d[6][q] = len, P->d[7][q] = ds, P->d[8][q] = line.s;
                                P->d[9][q] = line.lat[0], P->d[10][q] = line.lat[1], P->d[11][q] = line.lat[2], P->d[12][q] = line.lat[3];
                                P->d[13][q] = line.lon[0], P->d[14][q] = line.lon[1], P->d[15][q] = line.lon[2], P->d[16][q] = line.lon[3];
                                P->d[17][q] = line.alt[0], P->d[18][q] = line.alt[1], P->d[19][q] = line.alt[2], P->d[20][q] = line.alt[3];
                                P->d[21][q] = line.k4, P->d[22][q] = line.tau;
                                P->i[0][q] = (int)ray, P->i[1][q] = count, P->i[2][q] = m, P->i[3][q] = k;
                                P->i[4][q] = state | (lined_ ? 4 : 0) | (line.valid ? 8 : 0) | (stopped_ ? 16 : 0);
                                P->i[5][q] = (int)cell.id, P->i[6][q] = (int)cell.lo, P->i[7][q] = (int)cell.hi;
                                /* the steps a ray took are counted by the wave that took them */
                                my_steps += (ull)(count - count0);
                                ray = -1;
                        }
                        if (pull_r | pull_s) {
                                const int q = in_slot;
                                bx = P->d[0][q], by = P->d[1][q], bz = P->d[2][q];
                                dx = P->d[3][q], dy = P->d[4][q], dz = P->d[5][q];
                                len = P->d[6][q], ds = P->d[7][q], line.s = P->d[8][q];
                                line.lat[0] = P->d[9][q], line.lat[1] = P->d[10][q], line.lat[2] = P->d[11][q], line.lat[3] = P->d[12][q];
                                line.lon[0] = P->d[13][q], line.lon[1] = P->d[14][q], line.lon[2] = P->d[15][q], line.lon[3] = P->d[16][q];
                                line.alt[0] = P->d[17][q], line.alt[1] = P->d[18][q], line.alt[2] = P->d[19][q], line.alt[3] = P->d[20][q];
                                line.k4 = P->d[21][q], line.tau = P->d[22][q];
                                ray = P->i[0][q], count = P->i[1][q], m = P->i[2][q], k = P->i[3][q];
                                count0 = count;
                                const int bits = P->i[4][q];
                                state = bits & 3, lined_ = (bits & 4) != 0, line.valid = (bits & 8) != 0, stopped_ = (bits & 16) != 0;
                                cell.id = (unsigned)P->i[5][q], cell.lo = (unsigned)P->i[6][q], cell.hi = (unsigned)P->i[7][q];
                        }
                        /* the lists: slots that were read are free again, the ones written are listed */
                        const int n_in = n_in_r + n_in_s;
                        if (pull_r | pull_s) P->free_slot[F - n_out + in_rank] = (unsigned short)in_slot;
                        /* (a ring's tail is where it was: what came out of it came from its head) */
                        const int tail = ((out_kind == POOL_READY) ? head_r + R : head_s + S) + out_rank;
                        if (push) P->list[out_kind][tail % kPoolSlots] = (unsigned short)out_slot;
                        if ((threadIdx.x & 63) == 0) {
                                P->n_free = F - n_out + n_in;
                                P->head[POOL_READY] = (head_r + n_in_r) % kPoolSlots;
                                P->head[POOL_SERVICE] = (head_s + n_in_s) % kPoolSlots;
                                P->count[POOL_READY] = R - n_in_r + ((out_kind == POOL_READY) ? n_out : 0);
                                P->count[POOL_SERVICE] = S - n_in_s + ((out_kind == POOL_SERVICE) ? n_out : 0);
                        }
                        pool_empty = (R - n_in_r + S - n_in_s + n_out) == 0;
                        pool_free = F - n_out + n_in;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if ((threadIdx.x & 63) == 0)
                                __hip_atomic_store(&P->lock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        PSTAT(0, 1), PSTAT((role == ROLE_AS_IS) ? 3 : role, 1), PSTAT(4, PSTAT_CLOCK() - t_in), PSTAT(13, n_out), PSTAT(14, n_in);
                        PSTAT(17, r_own), PSTAT(18, s_own), PSTAT(19, R), PSTAT(20, S);
                }
        };

        for (;;) {
                if (POOLED && pooled) pool_exchange();
                /* ---- refill idle lanes from the queue ---- */
                for (;;) {
                        /* (pooled: a lean wave takes no new ray -- it would wait for a general
                         * iteration -- and a block whose pool is nearly full has rays enough) */
                        if (POOLED && pooled && ((role == ROLE_LEAN) || (pool_free < kPoolRefillFree))) break;
                        const bool need = (ray < 0) && !dead;
                        const ull mask = __ballot(need);
                        if (mask == 0) break;
                        /* Over paged tiles a new ray may need one that is not resident: its first
                         * sample then takes the exact lookup, with its dependent loads, and the
                         * ray leaves for the list -- a trip that costs the whole wave six times a
                         * plain one.  With half the tiles away half of the new rays do, and if a
                         * wave took new rays whenever a lane was free EVERY trip of it would be
                         * such a one (C3 with 8 of 16 tiles: the closed-form pass 28 ms for
                         * 6.4): the free lanes wait until there are kPagedRefill of them. */
                        if (CAN_FAULT && (ph.pg.faulted != nullptr) && !exhausted &&
                            (__popcll(mask) < kPagedRefill) && (__ballot(ray >= 0) != 0))
                                break;
                        if (pool_next >= pool_end) {
                                if (exhausted) {
                                        if (need) dead = true;
                                        break;
                                }
                                ull base = 0;
                                if ((threadIdx.x & 63) == 0)
                                        base = atomicAdd(queue, (ull)ph.chunk);
                                base = __shfl(base, 0, 64);
                                pool_next = (long)base;
                                pool_end = min((long)base + ph.chunk, n);
                                if ((long)base >= n) {
                                        exhausted = true;
                                        pool_next = pool_end = 0;
                                }
                                continue;
                        }
                        const long avail = pool_end - pool_next;
                        const int rank = __builtin_amdgcn_mbcnt_hi(
                            (unsigned)(mask >> 32),
                            __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                        if (need && (rank < avail)) {
                                ray = pool_next + rank;
                                if (ph.ids != nullptr)
                                        ray = ph.ids[(ray < n_front) ? ray : capacity - 1 - (ray - n_front)];
                                if (MODEL) line.valid = false, line.s = 0., line.tau = kLineTau0;
                                stopped_ = false;
                                if (!MODEL && (ph.dir_copy != nullptr)) {
                                        /* (the rays in the library's order: see PhaseIO) */
                                        const long src = ph.out.order_of[ray];
                                        bx = ph.pos_in[3 * src], by = ph.pos_in[3 * src + 1], bz = ph.pos_in[3 * src + 2];
                                        dx = ph.dir_in[3 * src], dy = ph.dir_in[3 * src + 1], dz = ph.dir_in[3 * src + 2];
                                        ph.dir_copy[3 * ray] = dx, ph.dir_copy[3 * ray + 1] = dy, ph.dir_copy[3 * ray + 2] = dz;
                                        if (flags & TRACE_CARRY_MEDIUM)
                                                index[2 * ray] = ph.index_in[2 * src], index[2 * ray + 1] = ph.index_in[2 * src + 1];
                                } else {
                                        bx = pos[3 * ray], by = pos[3 * ray + 1], bz = pos[3 * ray + 2];
                                        dx = dir[3 * ray], dy = dir[3 * ray + 1], dz = dir[3 * ray + 2];
                                }
                                len = 0., count = 0, state = ST_INIT;
                                if (ph.accumulate) len = length[ray], count = n_steps[ray];
                                count0 = count;
                                lined_ = MODEL && (count >= ph.line_after);
                                if (CAN_FAULT && (ph.accumulate == 2)) {
                                        /* a ray that waited for a tile: it carries on
                                         * with the step it was about to take (a fresh
                                         * sample of its position could need the tile it
                                         * came from, which may be gone) */
                                        const double w = ph.pg.tentative[ray];
                                        if (w >= 0.) {
                                                state = ST_STEP, ds = w;
                                                m = index[2 * ray], k = index[2 * ray + 1];
                                        }
                                }
                        }
                        pool_next += min((long)__popcll(mask), avail);
                }
                if (__ballot(ray >= 0) == 0) {
                        if (!(POOLED && pooled)) break;
                        /* pooled: the wave leaves when the queue is dry and the pool was empty at
                         * its last look; else it looks again (a kernel must always end: a wave that
                         * finds nothing a million times over leaves too -- its rays, if any were
                         * left, are then missing from the totals, which the callers check) */
                        if ((exhausted && pool_empty) || (++idle_trips > 1000000)) break;
                        if (role == ROLE_LEAN) role = ROLE_SERVICE; /* (cannot be: a lean wave has rays) */
                        __builtin_amdgcn_s_sleep(8);
                        continue;
                }
                idle_trips = 0;
                /* ---- creep loop (phase B, sparse waves) ---------------------------
                 * What is left at the end of a launch is a handful of rays
                 * skimming the ground with ~0.5 m steps for thousands of steps.
                 * While every live lane of the wave is such a ray -- stepping on
                 * its line -- a step needs no state machine: this loop does just
                 * that (laying a new line or fetching a new cell when needed),
                 * and hands any lane that needs more (a boundary, the step cap,
                 * another state) back to the general iteration below WITHOUT
                 * having committed that step.  It calls the same functions on
                 * the same values as the general path, so results do not depend
                 * on whether it engaged. */
                if (MODEL && (MODE != TAMD_MODE_ONE_MAP) &&
                    !((MODE == TAMD_MODE_ONE_STACK) && ctx.stack.regular) && /* -> the lean loop below */
                    (__popcll(__ballot(ray >= 0)) <= ph.creep_lanes)) {
                        for (int it = 0; it < 4096; it++) {
                                bool fail = false;
                                double qx = 0, qy = 0, qz = 0;
                                Sample s;
                                if (ray >= 0) {
                                        fail = (state != ST_STEP) || (count + 1 >= max_steps) || !lined_;
                                        if (!fail) {
                                                const double sl = line.s + ds;
                                                qx = d_along<FAST>(bx, dx, ds), qy = d_along<FAST>(by, dy, ds), qz = d_along<FAST>(bz, dz, ds);
                                                /* a new line starts at q: B is at -ds on it */
                                                if (f_sample_on_line<MODE>(v, ctx, qx, qy, qz, dx, dy,
                                                        dz, line, sl, s,
                                                        (MODE != TAMD_MODE_GENERIC) ? &cell : nullptr))
                                                        line.s = -ds;
                                                fail = (s.m != m) || (s.fault.centre >= 0);
                                        }
                                }
                                /* a lane that must leave has sampled q but not moved:
                                 * the general iteration samples the same q again, and
                                 * gets the same bits (line and cell now serve q) */
                                if (__ballot(fail) != 0) break;
                                if (ray >= 0) {
                                        bx = qx, by = qy, bz = qz;
                                        line.s += ds;
                                        line.tau += kLineDrift;
                                        len += ds;
                                        count++;
                                        k = s.k;
                                        my_samples++;
                                        ds = d_step_length(v, s.alt, s.e0, s.e1, s.m);
                                }
                        }
                }

                /* ---- lean steps (one map, a regular stack) ------------------------
                 * The lined pass's work horse: a step of a ray on its line over one
                 * grid is the line (nine fused operations), the cell (its four nodes
                 * decoded to the patch's coefficients once per cell; fetched here when
                 * the ray walks into the next one) and the reference's tests -- ~75
                 * vector instructions where a closed-form sample takes a thousand.
                 * A lane that needs anything else (a new line, a crossing and its
                 * bisection, the rim of the grid, the step cap) stops WITHOUT having
                 * committed that step, and a general iteration below takes it: the
                 * same functions on the same values, so results do not depend on
                 * where a step was taken.  "Does any lane have to leave?" is a
                 * wave-wide question (compare, ballot, branch: the vector and
                 * scalar units wait for each other): it is asked once per
                 * kCreepUnroll steps, and a lane that cannot take one of them takes
                 * none of the following.  In a wave of a handful of rays (the end of
                 * a launch: C2's longest ray takes 11 326 steps) the group ends when
                 * any lane stopped; in a busy one the lanes that stopped wait while
                 * dense_go others step on.  The SIMD's issue slots bound this loop
                 * (three waves of it keep the vector unit ~90 % busy): what counts is
                 * the instructions of a step and the share of lanes that take it. */
                const int live = __popcll(__ballot(ray >= 0));
                const bool sparse = (live <= ph.creep_lanes);
                if (creep_wait > 0) creep_wait--;
                if (MODEL &&
                    ((MODE == TAMD_MODE_ONE_MAP) || ((MODE == TAMD_MODE_ONE_STACK) && ctx.stack.regular)) &&
                    (sparse || ((ph.dense_go > 0) && (creep_wait == 0)) || (POOLED && (role == ROLE_LEAN))) &&
                    !(POOLED && (role == ROLE_SERVICE))) {
                        /* one map: the grid.  A regular stack: the tile the cached cell is
                         * in -- the shared tile shape at that tile's origin, computed as
                         * f_stack_elevation computes it; a point `interior` to it (same
                         * guard as there) gets that tile from the directory too */
                        constexpr bool STACK = (MODE == TAMD_MODE_ONE_STACK);
                        const tamd_grid & g = STACK ? ctx.stack.proto : ctx.grid;
                        /* The cached cell, decoded once per entry (and after a trip that
                         * changed it): its origin, its node coordinates as doubles and its
                         * four elevations -- a step then needs no conversion between
                         * integers and doubles (a quarter of the rate of the other
                         * instructions, and on the chain).  Same values as f_grid_locate /
                         * f_grid_blend produce: for an interior point (double)(int)hx ==
                         * trunc(hx), and the clamp of the cell index does nothing.
                         * "Still in the cached cell" is asked of the fractions fx = hx - cx,
                         * fy = hy - cy with a margin of `guard` -- g < f < 1 - g, as a test of f's
                         * upper word against those of g and 1 - g -- so that a point that
                         * passes is `interior` to the grid (or the tile: the guards of
                         * f_grid_locate and f_stack_elevation) whichever cell it is in; the few
                         * within the margin of a cell's edge take the way of a cell change,
                         * which makes that test on the point itself. */
                        constexpr double guard = STACK ? kSeamGuard : 1e-6;
                        /* the upper word of guard, plus one; that of 1 - guard */
                        constexpr unsigned lo_word = STACK ? 0x3e112e0cu : 0x3eb0c6f8u;
                        constexpr unsigned span_words = (STACK ? 0x3fefffffu : 0x3feffffdu) - lo_word;
                        const double mx = (double)(g.nx - 1) - guard, my = (double)(g.ny - 1) - guard;
                        double x0, y0, cx, cy, z00, z10, z01, z11;
                        bool cached;
                        auto decode_nodes = [&]() {
                                if (g.is_signed) {
                                        z00 = (double)(int16_t)(cell.lo & 0xffffu), z10 = (double)((int)cell.lo >> 16);
                                        z01 = (double)(int16_t)(cell.hi & 0xffffu), z11 = (double)((int)cell.hi >> 16);
                                } else {
                                        z00 = (double)(cell.lo & 0xffffu), z10 = (double)(cell.lo >> 16);
                                        z01 = (double)(cell.hi & 0xffffu), z11 = (double)(cell.hi >> 16);
                                }
                                z00 = __builtin_fma(z00, g.dz, g.z0), z10 = __builtin_fma(z10, g.dz, g.z0);
                                z01 = __builtin_fma(z01, g.dz, g.z0), z11 = __builtin_fma(z11, g.dz, g.z0);
                                /* f_patch's coefficients, as f_grid_blend forms them */
                                z11 = (z11 - z10) - (z01 - z00), z10 = z10 - z00, z01 = z01 - z00;
                        };
                        auto decode_cell = [&]() {
                                cached = (cell.id != ~0u);
                                const unsigned slot = STACK ? (cell.id >> 24) : 0u;
                                const unsigned cell_index = STACK ? (cell.id & 0xffffffu) : cell.id;
                                const unsigned tile_y = STACK ? slot / (unsigned)ctx.stack.nlon : 0u;
                                const unsigned tile_x = STACK ? slot - tile_y * (unsigned)ctx.stack.nlon : 0u;
                                x0 = STACK ? ctx.stack.lon0 + (int)tile_x * ctx.stack.dlon : g.x0;
                                y0 = STACK ? ctx.stack.lat0 + (int)tile_y * ctx.stack.dlat : g.y0;
                                const unsigned cell_iy = cached ? cell_index / (unsigned)g.nx : 0u;
                                const unsigned cell_ix = cached ? cell_index - cell_iy * (unsigned)g.nx : 0u;
                                cy = cached ? (double)cell_iy : -1.;
                                cx = cached ? (double)cell_ix : -1.;
                                decode_nodes();
                        };
                        decode_cell();
                        /* The test of a lean step is SUFFICIENT for the step the general
                         * iteration would accept, not equivalent to it -- a lane that fails
                         * it has committed nothing and the general iteration decides: with
                         * t = altitude - elevation and sgn = -1 in the rock, +1 above it,
                         *   sgn t > tau                  => the line serves as to drift and
                         *                                   truncation near the boundary (|t| > tau)
                         *                                   AND the medium is the ray's;
                         *   k4 min(max(|t|, 1), 400) > s^4   => the line serves as to its reach,
                         *                                   which is then below kLineRange too
                         *                                   (kLeanClearance: (5.56e11 x 400)^(1/4) =
                         *                                   3862 m at the equator, where k4 is largest);
                         * and a step within kCreepUnroll of the cap is left to the general
                         * iteration.  Two compares a step decide it where the reference's
                         * tests, one by one, took a dozen and as many scalar instructions
                         * between them: that chain is what a step of a lone wave waits for. */
                        const double sgn = (m == 0) ? -1. : 1.; /* a lane's medium does not change in here */
                        const int count_in = count;
                        const unsigned long long t_lean = PSTAT_CLOCK();
                        (void)t_lean;
                        for (int it = 0; it < 4096; it++) {
                                /* no short-circuits below: every lane computes
                                 * everything (garbage is harmless, nothing is
                                 * committed on failure) and the tests are AND-ed */
                                bool going = (ray >= 0) & (state == ST_STEP) & lined_ & line.valid & cached &
                                    (count + kCreepUnroll < max_steps);
                                PSTAT(5, 1), PSTAT(7, __popcll(__ballot(going)));
#pragma unroll
                                for (int u = 0; u < kCreepUnroll; u++) {
                                        const double sl = line.s + ds;
                                        double lat, lon, alt;
                                        f_line_eval(line, sl, lat, lon, alt);
                                        /* f_grid_locate without its rim fallback */
                                        const double hx = (lon - x0) * g.inv_dx;
                                        const double hy = (lat - y0) * g.inv_dy;
                                        /* still in the cached cell, and not within `guard` of its
                                         * edges <=> guard < hx - cx < 1 - guard and the same in y:
                                         * for a double, "its upper word, unsigned, is in [that of
                                         * guard + 1, that of 1 - guard)" (a negative one has the
                                         * sign bit there, a NaN the exponent's) */
                                        double fx = hx - cx, fy = hy - cy;
                                        if (going & (max(d_upper_word(fx) - lo_word, d_upper_word(fy) - lo_word) >= span_words)) {
                                                /* another cell of the same grid: what
                                                 * f_grid_elevation does on a cache miss */
                                                const double tx = __builtin_trunc(hx), ty = __builtin_trunc(hy);
                                                const int ix = (int)tx, iy = (int)ty; /* (NaN: 0) */
                                                going = (hx > guard) & (hx < mx) & (hy > guard) & (hy < my);
                                                if (going) {
                                                        const unsigned id = (unsigned)iy * (unsigned)g.nx + (unsigned)ix;
                                                        d_cell_fetch(STACK ? ctx.slots[cell.id >> 24] : g.nodes, g.nbx, ix, iy, cell.lo, cell.hi);
                                                        cell.id = STACK ? ((cell.id & 0xff000000u) | id) : id;
                                                        cx = tx, cy = ty;
                                                        fx = hx - tx, fy = hy - ty;
                                                        decode_nodes();
                                                }
                                        }
                                        /* f_grid_blend */
                                        const double elevation = f_patch(z00, z10, z01, z11, fx, fy) + ctx.offset;
                                        const double t = alt - elevation;
                                        const double clearance = fabs(t);
                                        const double s2 = sl * sl;
                                        going = going & (__builtin_fma(sgn, t, -line.tau) > 0.) &
                                            (__builtin_fma(line.k4, fmin(fmax(clearance, 1.), kLeanClearance), -(s2 * s2)) > 0.);
                                        if (going) {
                                                bx = d_along<FAST>(bx, dx, ds), by = d_along<FAST>(by, dy, ds), bz = d_along<FAST>(bz, dz, ds);
                                                line.tau = line.tau + kLineDrift;
                                                line.s = line.s + ds; /* == sl */
                                                len = len + ds;
                                                count++;
                                                /* d_step_length for one surface: both of its
                                                 * cases are |alt - elevation| */
                                                ds = fmax(clearance * v.slope, v.resolution);
                                        }
                                }
                                const bool stopped = (ray >= 0) & !going;
                                const int n_stopped = __popcll(__ballot(stopped));
                                if (n_stopped == 0) continue;
                                if (POOLED && pooled) stopped_ = stopped;
                                if (POOLED && (role == ROLE_LEAN)) break; /* the pool takes them: see RayPool */
                                if (!sparse) {
                                        /* a busy wave: the lanes that stopped wait while
                                         * enough of the others step on (a group of lean
                                         * steps costs a third of a general iteration, which
                                         * then serves every lane that waits at once); where
                                         * the first group already loses most lanes -- rays
                                         * high above the ground, a new cell every step --
                                         * the wave does not try again for a while */
                                        if (live - n_stopped >= ph.dense_go) continue;
                                        if (it == 0) creep_wait = kCreepBackoff;
                                        break;
                                }
                                break;
                        }
                        my_samples += (ull)(count - count_in); /* a lean step is a sample */
                        PSTAT(11, PSTAT_CLOCK() - t_lean), PSTAT(6, wave_sum((ull)(count - count_in)));
                }
                if (POOLED && (role == ROLE_LEAN)) continue; /* back to the pool: no general iteration */

                /* `drain`: once the queue is dry a wave of phase A hands its rays over
                 * as they stand (between two steps) instead of stepping its last few
                 * to the hand-over count with most lanes idle -- measured with the
                 * hand-over at 512 steps: the queue of C2 is dry after 2.4 ms and the
                 * last wave left at 4.5 ms.  Phase B packs them again. */
                const bool drain = !MODEL && (ph.park_after > 0) && exhausted && (ray >= 0) &&
                    (state == ST_STEP) && (__popcll(__ballot(ray >= 0)) <= ph.drain_lanes);
                bool park = drain;
                const unsigned long long t_gen = PSTAT_CLOCK();
                (void)t_gen;
                PSTAT(8, 1), PSTAT(9, __popcll(__ballot(ray >= 0)));
                TileFault fault = { -1, 0, 0 }; /* the tiles to page in, if any */
                double fx = 0, fy = 0, fz = 0; /* where the ray goes back to, then */
                bool defer = false; /* MODEL: the lane waits for a closed form (see below) */
                bool crossed = false; /* CROSS: the step crossed a boundary: the ray goes on the list */
                if ((ray >= 0) && !drain) {
                        /* ---- one sample at q = B + d * t ---- */
                        double t = 0.;
                        if (state == ST_STEP) t = ds;
                        if (!CROSS && (state == ST_BISECT))
                                t = FAST ? f_bracket_point(ds0, ds1, c0, c1, halvings & 0xffff) : 0.5 * (ds0 + ds1);
                        double qx = bx, qy = by, qz = bz;
                        if (state != ST_INIT) /* B + d*0 == B, but d may be garbage */
                                qx = d_along<FAST>(bx, dx, t), qy = d_along<FAST>(by, dy, t), qz = d_along<FAST>(bz, dz, t);

                        Sample s;
                        if (LINED) {
                                /* B's parameter: -t on a new line (its origin is q),
                                 * and a STEP sample then moves B to q */
                                CellCache * const cache = (MODE != TAMD_MODE_GENERIC) ? &cell : nullptr;
                                bool relay = !f_line_try<MODE>(v, ctx, line, line.s + t, s, cache, !CROSS && (state == ST_BISECT));
                                /* A closed form is a thousand instructions for the
                                 * whole wave, whoever needs it: in a busy wave the
                                 * lanes that do (a ray's first sample, a line at its
                                 * end) wait until there are kRelayBatch of them, or
                                 * until only they are left, or kRelayPatience general
                                 * iterations.  Nothing is committed for a lane that
                                 * waits: it takes this very sample again.  (One map: C2
                                 * -2.4 %.  Through a stack the same costs 6-9 %, with 20
                                 * bytes more of scratch in a kernel held to 168 registers.) */
                                if ((MODE == TAMD_MODE_ONE_MAP) && !sparse) {
                                        const int n_need = __popcll(__ballot(relay));
                                        const int n_here = __popcll(__ballot(true));
                                        const int waited = __builtin_amdgcn_readfirstlane(relay_wait);
                                        const bool now = (n_need >= kRelayBatch) | (n_need == n_here) |
                                            (waited >= kRelayPatience);
                                        relay_wait = ((n_need == 0) | now) ? 0 : waited + 1;
                                        defer = relay & !now;
                                        relay = relay & now;
                                }
                                PSTAT(10, __popcll(__ballot(relay)));
                                if (relay) {
                                        f_line_relay<MODE>(v, ctx, qx, qy, qz, dx, dy, dz, line, s, cache);
                                        line.s = -t;
                                }
                                if ((state == ST_STEP) & !defer) line.s += t;
                        } else
                                d_sample<MODE, FAST>(v, ctx, qx, qy, qz, s,
                                    (FAST && (MODE != TAMD_MODE_GENERIC)) ? &cell : nullptr);
                        /* (a ray that phase A handed over before its line starts -- when
                         * its queue ran dry: which rays, depends on the scheduling -- samples
                         * its position again here: not one of the trace's samples, so that
                         * the count is the same from run to run) */
                        my_samples += (defer || (MODEL && !lined_ && (state == ST_INIT))) ? 0 : 1;
                        if (CAN_FAULT && !defer && (s.fault.centre >= 0)) {
                                /* a tile that is not resident: the ray goes back to
                                 * the arrays as it was BEFORE this sample (before the
                                 * crossing step, if it was bisecting: the bracket is
                                 * not kept) and on the list for the next round */
                                fault = s.fault;
                                if (state == ST_INIT) home = -1; /* a new ray: nothing to keep */
                                const double back = (!CROSS && (state == ST_BISECT)) ? ds : 0.;
                                fx = bx - dx * back, fy = by - dy * back, fz = bz - dz * back;
                        }

                        if (POOLED) stopped_ = defer;
                        /* ---- bookkeeping ----
                         * STEP and BISECT are handled together, as selects rather
                         * than branches: in a busy wave every case is present in
                         * some lane on every trip, so branching buys nothing and
                         * costs exec-mask juggling.  Only INIT (once per ray) and
                         * the two endings (located, done) stay branches. */
                        bool done = false, located = false;
                        if (CAN_FAULT && !defer && (fault.centre < 0) && (state == ST_STEP)) home = s.slot;
                        if ((fault.centre >= 0) | defer) {
                                /* nothing: see below */
                        } else if (state == ST_INIT) {
                                m = s.m, k = s.k;
                                ds = (m >= 0) ? d_step_length(v, s.alt, s.e0, s.e1, m) : 0.;
                                if ((flags & TRACE_CARRY_MEDIUM) && (m >= 0)) {
                                        /* The caller knows which medium the ray
                                         * is in; the sample only sizes the step.
                                         * If the two disagree the ray sits ON the
                                         * boundary between them (the bisection
                                         * left it within 1e-8 m): the distance to
                                         * the nearest surface is ~0 and the
                                         * reference's cached sample would give the
                                         * minimum step [ref stepper.c:812-813]. */
                                        const int given = index[2 * ray];
                                        if ((given >= 0) && (given <= v.n_layers) && (given != m)) {
                                                m = given;
                                                ds = v.resolution;
                                        }
                                }
                                state = ST_STEP;
                                done = (m < 0) || (count >= max_steps);
                        } else if (CROSS) {
                                /* every sample is a STEP sample: it stands, or the ray
                                 * goes on the list of the crossings (below) */
                                const bool same = (s.m == m);
                                const double ds_next = d_step_length(v, s.alt, s.e0, s.e1, s.m);
                                bx = qx, by = qy, bz = qz; /* [ref stepper.c:824] */
                                crossed = !same;           /* [ref stepper.c:832-838] */
                                bm = s.m, bk = s.k;
                                if (MODEL) line.tau = same ? line.tau + kLineDrift : line.tau;
                                len = same ? len + ds : len;
                                k = same ? s.k : k;
                                ds = same ? ds_next : ds; /* a crossing keeps the tentative length */
                                count += same ? 1 : 0;
                                const bool capped = same & (count >= max_steps);
                                done = capped;
                                my_capped += capped ? 1 : 0;
                                park = same & !capped & (ph.park_after > 0) & (count >= ph.park_after);
                                if (MODEL && same && !capped && !park && !lined_ &&
                                    (count >= ph.line_after)) {
                                        lined_ = true; /* (see the other branch) */
                                        line.valid = false, line.s = 0.;
                                        state = ST_INIT;
                                }
                        } else {
                                const bool stepping = (state == ST_STEP);
                                const bool same = (s.m == m);
                                const bool accept = stepping & same;   /* the step stands */
                                const bool cross = stepping & !same;   /* [ref stepper.c:832-838] */
                                const bool other = !same;              /* a sample of another medium */
                                const double ds_next = d_step_length(v, s.alt, s.e0, s.e1, s.m);
                                /* a STEP sample always moves B to q */
                                bx = stepping ? qx : bx, by = stepping ? qy : by, bz = stepping ? qz : bz;
                                if (MODEL) line.tau = accept ? line.tau + kLineDrift : line.tau;
                                len = accept ? len + ds : len;
                                k = accept ? s.k : k;
                                bm = other ? s.m : bm, bk = other ? s.k : bk;
                                /* the bracket [ref stepper.c:836, :849-858] */
                                ds0 = cross ? -ds : ((!stepping & same) ? t : ds0);
                                ds1 = cross ? 0. : ((!stepping & other) ? t : ds1);
                                int moved_twice = 0; /* bit 16 / 17 of halvings: the last sample moved ds0 / ds1 */
                                if (FAST) {
                                        /* the clearance where the crossing step began is
                                         * what sized it (more, if the resolution did: a guess) */
                                        const double cl = fmin(fabs(s.alt - s.e0), fabs(s.alt - s.e1));
                                        const bool again0 = !stepping & same & ((halvings & 0x10000) != 0);
                                        const bool again1 = !stepping & other & ((halvings & 0x20000) != 0);
                                        c0 = cross ? ds / v.slope : ((!stepping & same) ? cl : (again1 ? 0.5 * c0 : c0));
                                        c1 = (cross | (!stepping & other)) ? cl : (again0 ? 0.5 * c1 : c1);
                                        moved_twice = stepping ? 0 : (same ? 0x10000 : 0x20000);
                                }
                                ds = accept ? ds_next : ds; /* a crossing keeps the tentative length */
                                count += accept ? 1 : 0;
                                /* a bracket of finite doubles is below 1e-8 after at
                                 * most ~1100 halvings; the cap only guards against
                                 * non-finite input (a kernel must always end) */
                                halvings = stepping ? 0 : (((halvings & 0xffff) + 1) | moved_twice);
                                state = cross ? ST_BISECT : state;
                                const bool capped = accept & (count >= max_steps);
                                done = capped;
                                my_capped += capped ? 1 : 0;
                                /* on to the next phase (always at the same step count:
                                 * the line a ray lays there is part of its arithmetic) */
                                park = accept & !capped & (ph.park_after > 0) & (count >= ph.park_after);
                                if (MODEL && accept && !capped && !park && !lined_ &&
                                    (count >= ph.line_after)) {
                                        /* phase B: from here on the ray steps on its line,
                                         * laid by a fresh sample of its position -- what a
                                         * ray handed over at this very step goes through */
                                        lined_ = true;
                                        line.valid = false, line.s = 0.;
                                        state = ST_INIT;
                                }
                                located = (state == ST_BISECT) &
                                    (!(ds1 - ds0 > 1E-08) | ((halvings & 0xffff) > 1200));
                        }
                        if (located) { /* [ref stepper.c:861-863] */
                                bx = d_along<FAST>(bx, dx, ds1), by = d_along<FAST>(by, dy, ds1), bz = d_along<FAST>(bz, dz, ds1);
                                len += ds + ds1;
                                count++;
                                m = bm, k = bk;
                                done = true;
                        }
                        if (done) {
                                if (ph.out.order_of != nullptr) { /* to the caller's arrays, the ray's place there */
                                        const long o = ph.out.order_of[ray];
                                        ph.out.pos[3 * o] = bx, ph.out.pos[3 * o + 1] = by, ph.out.pos[3 * o + 2] = bz;
                                        ph.out.index[2 * o] = m, ph.out.index[2 * o + 1] = k;
                                        ph.out.length[o] = len;
                                        ph.out.n_steps[o] = count;
                                } else {
                                        pos[3 * ray] = bx, pos[3 * ray + 1] = by, pos[3 * ray + 2] = bz;
                                        index[2 * ray] = m, index[2 * ray + 1] = k;
                                        if (length) length[ray] = len;
                                        if (n_steps) n_steps[ray] = count;
                                }
                                my_rays++;
                                my_steps += (ull)(count - count0);
                                ray = -1;
                        }
                }
                PSTAT(12, PSTAT_CLOCK() - t_gen);
                /* ---- park over-long rays (phase A; whole wave takes part) ---- */
                if (!MODEL && (ph.ds_mark != nullptr) && (ray >= 0) && (state == ST_STEP) &&
                    (count == ph.mark_at))
                        ph.ds_mark[ray] = ds;
                const ull pmask = __ballot(park);
                if (pmask != 0) {
                        /* Phase A sorts what it hands over.  A launch of phase B ends with
                         * the chip all but empty, waiting for the few rays of thousands of
                         * steps -- the later one of those was drawn from the queue, the
                         * longer.  A ray whose steps shrank from d0 (at step mark_at) to ds
                         * now will, at that rate, be down to the minimum step after
                         *   ln(ds / resolution) / (ln(d0 / ds) / (count - mark_at))
                         * more: a crude figure that tells what matters -- the rays heading
                         * straight for the ground have little there (of C2's rays at step 32
                         * the 40 % below 120 hold none of the 4 % that take over 500 further
                         * steps, nor any of the 0.1 % over 2 000), and they go to the back of
                         * the list.  Where a ray is listed changes when phase B takes it, not
                         * what comes out. */
                        bool back = false;
                        if (!MODEL && (ph.n_parked_back != nullptr) && park && (count > ph.mark_at)) {
                                const float shrink = __logf((float)ph.ds_mark[ray] / (float)ds);
                                const float togo = __logf((float)ds / (float)v.resolution);
                                back = (shrink > 0.f) &&
                                    !(togo * (float)(count - ph.mark_at) > ph.long_if * shrink);
                        }
                        const ull bmask = __ballot(back);
                        const ull fmask = pmask & ~bmask;
                        const int leader = __builtin_ctzll(pmask);
                        ull base = 0, base_back = 0;
                        if ((int)(threadIdx.x & 63) == leader) {
                                if (fmask != 0) base = atomicAdd(ph.n_parked, (ull)__popcll(fmask));
                                if (bmask != 0) base_back = atomicAdd(ph.n_parked_back, (ull)__popcll(bmask));
                        }
                        base = __shfl(base, leader, 64);
                        base_back = __shfl(base_back, leader, 64);
                        if (park) {
                                const ull mine = back ? bmask : fmask;
                                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32),
                                    __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0));
                                const long place = back ? capacity - 1 - (long)(base_back + rank) : (long)(base + rank);
                                ph.parked[place] = (int)ray;
                                if (!MODEL && (ph.sort_key != nullptr)) {
                                        /* The lined pass ends waiting for its longest rays, and the
                                         * later one of those is drawn, the longer: the front of the
                                         * list is ordered by how shallow a ray goes -- the sine of its
                                         * elevation angle, d . up (up ~ B / |B|: the geocentric
                                         * vertical, 0.2 degrees off at most) -- the shallowest first:
                                         * they are the ones that skim the ground for thousands of
                                         * steps (measured by ordering a batch's INPUT that way: C2 3.37
                                         * -> 3.01 ms).  The back of the list (rays that cannot be long)
                                         * keeps its place behind everything: the largest key. */
                                        const double up = __builtin_fma(dx, bx, __builtin_fma(dy, by, dz * bz)) *
                                            __builtin_amdgcn_rsq(__builtin_fma(bx, bx, __builtin_fma(by, by, bz * bz)));
                                        /* (one byte: 253 levels up to 14 degrees -- one pass of the sort;
                                         * 254 is a place nobody filled, 255 the back) */
                                        /* (tried instead: the steps the ray would take over flat ground at
                                         * that angle from its clearance, on a log scale: no better) */
                                        const double scaled = fmin(fabs(up) * 1024., 253.);
                                        ph.sort_key[place] = back ? (unsigned char)255 : (unsigned char)scaled;
                                }
                                pos[3 * ray] = bx, pos[3 * ray + 1] = by, pos[3 * ray + 2] = bz;
                                index[2 * ray] = m, index[2 * ray + 1] = k;
                                length[ray] = len;
                                n_steps[ray] = count;
                                my_steps += (ull)(count - count0);
                                ray = -1;
                        }
                }
                /* ---- list the rays that crossed a boundary (whole wave takes part) ---- */
                if (CROSS) {
                        const ull cmask = __ballot(crossed);
                        if (cmask != 0) {
                                const int leader = __builtin_ctzll(cmask);
                                ull base = 0;
                                if ((int)(threadIdx.x & 63) == leader)
                                        base = atomicAdd(ph.cross.count, (ull)__popcll(cmask));
                                base = __shfl(base, leader, 64);
                                if (crossed) {
                                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(cmask >> 32),
                                            __builtin_amdgcn_mbcnt_lo((unsigned)cmask, 0));
                                        ph.cross.ray[base + rank] = (int)ray;
                                        ph.cross.ds[base + rank] = ds;
                                        ph.cross.other[base + rank] = cross_pack(bm, bk);
                                        /* B is the tentative point; m, k the medium it left */
                                        pos[3 * ray] = bx, pos[3 * ray + 1] = by, pos[3 * ray + 2] = bz;
                                        index[2 * ray] = m, index[2 * ray + 1] = k;
                                        length[ray] = len;
                                        n_steps[ray] = count;
                                        my_steps += (ull)(count - count0);
                                        ray = -1;
                                }
                        }
                }
                /* ---- list the rays that wait for a tile (whole wave takes part) ---- */
                if (CAN_FAULT && (ph.pg.faulted != nullptr)) {
                        const bool waits = fault.centre >= 0;
                        if (waits) {
                                pos[3 * ray] = fx, pos[3 * ray + 1] = fy, pos[3 * ray + 2] = fz;
                                if (state != ST_INIT) /* else: what the caller gave, or nothing */
                                        index[2 * ray] = m, index[2 * ray + 1] = k;
                                else if (!(flags & TRACE_CARRY_MEDIUM))
                                        index[2 * ray] = -1, index[2 * ray + 1] = -1;
                                if (length) length[ray] = len;
                                if (n_steps) n_steps[ray] = count;
                                /* the step it was about to take (a bisecting ray went
                                 * back before its crossing step: the same one) */
                                ph.pg.tentative[ray] = (state != ST_INIT) ? ds : -1.;
                                my_steps += (ull)(count - count0);
                        }
                        /* a bisecting ray also wants the tile of its crossing sample */
                        page_fault(ph.pg, fault, ray, (!CROSS && (state == ST_BISECT)) ? home : -1);
                        if (waits) ray = -1;
                }
        }

#ifdef TRACE_POOL_STATS
        if (MODEL && ((threadIdx.x & 63) == 0))
                for (int i = 0; i < 24; i++)
                        if (pstat_[i]) atomicAdd(&g_pool_stats[i], pstat_[i]);
#endif
        block_tally(stats, my_rays, my_steps, my_samples, my_capped);
}

/* Where a ray STARTS, as one byte: the cell of a 16 x 16 raster over the map or the stack's
 * lattice that holds its origin (rays outside go to the rim's cells).  The trace then takes the
 * rays cell by cell (run_trace: one pass of a radix sort on these keys): the 64 rays of a wave
 * start within a few kilometres of each other and meet the same tiles, the same pages and -- for a
 * while -- the same cache lines.  Ordering a batch's INPUT that way was worth 19 % on C3 (16 tiles,
 * 415 MB: 25.7 -> 20.8 ms), 7 % on C4, 4 % on C5; this does it inside the library, whatever order
 * the caller's rays come in, and leaves the caller's arrays where they are (the passes read the
 * rays through the ordered list of their numbers). */
#ifndef SPATIAL_CELLS
#define SPATIAL_CELLS 16
#endif
#if SPATIAL_CELLS <= 16
#define SPATIAL_KEY_T unsigned char
#define SPATIAL_BITS 8
#else
#define SPATIAL_KEY_T unsigned short
#define SPATIAL_BITS 16
#endif
template <int MODE>
__global__ void k_ray_cells(tamd_view v, long n, const double * __restrict__ pos,
    SPATIAL_KEY_T * __restrict__ key, int * __restrict__ id)
{
        const tamd_meta mt = v.metas[0];
        constexpr int kCells = SPATIAL_CELLS;
        double x0, y0, sx, sy; /* the box: longitude, latitude; cells / its extent */
        if (MODE == TAMD_MODE_ONE_MAP) {
                const tamd_grid & g = v.grids[mt.src];
                x0 = g.x0, y0 = g.y0;
                sx = kCells / (g.dx * (g.nx - 1)), sy = kCells / (g.dy * (g.ny - 1));
        } else {
                const tamd_stack & st = v.stacks[mt.src];
                x0 = st.lon0, y0 = st.lat0;
                sx = kCells / (st.dlon * st.nlon), sy = kCells / (st.dlat * st.nlat);
        }
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
                double lat, lon, alt;
                f_to_geodetic(pos[3 * r], pos[3 * r + 1], pos[3 * r + 2], lat, lon, alt);
                const int bx = (int)fmin(fmax((lon - x0) * sx, 0.), (double)(kCells - 1));
                const int by = (int)fmin(fmax((lat - y0) * sy, 0.), (double)(kCells - 1)); /* (NaN: 0) */
                key[r] = (SPATIAL_KEY_T)(by * kCells + bx);
                id[r] = (int)r;
        }
}

/* The least waves a SIMD the kernel must fit (registers: 512 / waves).  The lined
 * pass is bound by what the SIMD issues, with a memory wait every few steps: a
 * third wave is worth more than the few values that go to scratch for it (one
 * map: 169 registers -> 168, none; a stack: 188 -> 168 and 64 bytes). */
#ifndef TRACE_LINED_WAVES
#define TRACE_LINED_WAVES 3
#endif
#ifndef TRACE_A_WAVES
#define TRACE_A_WAVES 1
#endif
template <int MODE, bool FAST, bool MODEL, bool PAGED>
constexpr int trace_waves()
{
        return (FAST && MODEL && !PAGED && (MODE != TAMD_MODE_GENERIC)) ? TRACE_LINED_WAVES :
               (FAST && !MODEL && !PAGED && (MODE != TAMD_MODE_GENERIC)) ? TRACE_A_WAVES : 1;
}

template <int MODE, bool FAST, bool MODEL, bool PAGED, bool CROSS, bool POOL = false>
__global__ void __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(trace_waves<MODE, FAST, MODEL, PAGED>())))
k_trace(tamd_view v, long n,
    double * __restrict__ pos, const double * __restrict__ dir, int max_steps,
    int * __restrict__ index, double * __restrict__ length, int * __restrict__ n_steps,
    int flags, PhaseIO ph, ull * __restrict__ stats, ull * __restrict__ queue)
{
        trace_body<MODE, FAST, MODEL, PAGED, CROSS, POOL>(v, n, pos, dir, max_steps, index, length, n_steps, flags,
            ph, stats, queue);
}

/* The crossings of a trace, every lane busy: for each listed ray the bracket
 * [-ds, 0] behind its tentative point q is narrowed below 1e-8 m [ref
 * stepper.c:832-864] -- by halving, on closed-form samples, in the reference's
 * arithmetic; in the fast one by false position (f_bracket_point) on the line
 * that the closed form lays AT q: it has no drift (the samples all leave from q)
 * and serves every sample within its reach, ~500 m; beyond, a sample is a closed
 * form that lays the next line.  The first sample is q again: the trace kernel
 * decided there that the medium changed (`other`), by this very closed form or
 * by the ray's line within its error bound of it; where the two disagree (a
 * boundary within 1e-9 m of q) the trace kernel's word stands, so that a listed
 * ray always ends here.  A ray that needs a tile which is not resident goes back
 * before its step and on the pager's list, as in k_trace. */
template <int MODE, bool FAST, bool PAGED>
__global__ void __launch_bounds__(256) k_cross(tamd_view v, double * __restrict__ pos,
    const double * __restrict__ dir, int * __restrict__ index, double * __restrict__ length,
    int * __restrict__ n_steps, CrossList cross, Paging pg, ull * __restrict__ stats, RayOut out)
{
        constexpr bool CAN_FAULT = PAGED && (MODE != TAMD_MODE_ONE_MAP);
        OneCtx ctx;
        d_load_ctx<MODE, FAST>(v, ctx);
        const long n = (long)*cross.count;
        ull my_rays = 0, my_samples = 0;
        for (long i0 = blockIdx.x * (long)blockDim.x; i0 < n; i0 += (long)gridDim.x * blockDim.x) {
                const long i = i0 + threadIdx.x; /* whole waves go round (page_fault) */
                TileFault fault = { -1, 0, 0 };
                int home = -1;
                long r = -1;
                if (i < n) {
                        r = cross.ray[i];
                        const double ds = cross.ds[i];
                        int bm, bk;
                        cross_unpack(cross.other[i], bm, bk);
                        const double px = pos[3 * r], py = pos[3 * r + 1], pz = pos[3 * r + 2];
                        const double dx = dir[3 * r], dy = dir[3 * r + 1], dz = dir[3 * r + 2];
                        const int medium0 = index[2 * r];
                        CellCache cell = { ~0u, 0u, 0u, -1, nullptr };
                        CellCache * cache = (FAST && (MODE != TAMD_MODE_GENERIC)) ? &cell : nullptr;
                        Sample s;
                        RayLine line;
                        double at = 0.; /* q's parameter on the line */
                        if (FAST) {
                                f_to_geodetic(px, py, pz, s.lat, s.lon, s.alt, &line, dx, dy, dz);
                                d_classify<MODE, true>(v, ctx, s, cache);
                        } else
                                d_sample<MODE, false>(v, ctx, px, py, pz, s, cache);
                        my_samples++;
                        if (CAN_FAULT) fault = s.fault, home = s.slot;
                        const bool agreed = (fault.centre < 0) && (s.m != medium0);
                        if (agreed) bm = s.m, bk = s.k;
                        double ds0 = -ds, ds1 = 0.;
                        /* the clearances at the two ends (f_bracket_point): the one where
                         * the step began is what sized it (more, if the resolution did) */
                        double c0 = ds / v.slope;
                        double c1 = agreed ? fmin(fabs(s.alt - s.e0), fabs(s.alt - s.e1)) : 0.;
                        int taken = 0, last = 0; /* last: the end the previous sample moved (1: ds0, 2: ds1) */
                        while ((fault.centre < 0) && (ds1 - ds0 > 1E-08) && (taken <= 1200)) {
                                const double t = FAST ? f_bracket_point(ds0, ds1, c0, c1, taken) :
                                                        0.5 * (ds0 + ds1);
                                const double qx = d_along<FAST>(px, dx, t), qy = d_along<FAST>(py, dy, t), qz = d_along<FAST>(pz, dz, t);
                                Sample s2;
                                if (FAST) {
                                        if (!f_line_try<MODE>(v, ctx, line, at + t, s2, cache, true)) {
                                                f_line_relay<MODE>(v, ctx, qx, qy, qz, dx, dy, dz, line, s2, cache);
                                                at = -t; /* a new line, laid at this sample */
                                        }
                                } else
                                        d_sample<MODE, false>(v, ctx, qx, qy, qz, s2, cache);
                                my_samples++, taken++;
                                if (CAN_FAULT && (s2.fault.centre >= 0)) {
                                        fault = s2.fault;
                                } else if (s2.m == medium0) {
                                        ds0 = t;
                                        if (FAST) {
                                                c0 = fmin(fabs(s2.alt - s2.e0), fabs(s2.alt - s2.e1));
                                                if (last == 1) c1 = 0.5 * c1; /* the Illinois rule */
                                                last = 1;
                                        }
                                } else {
                                        ds1 = t;
                                        bm = s2.m, bk = s2.k;
                                        if (FAST) {
                                                c1 = fmin(fabs(s2.alt - s2.e0), fabs(s2.alt - s2.e1));
                                                if (last == 2) c0 = 0.5 * c0;
                                                last = 2;
                                        }
                                }
                        }
                        if (fault.centre >= 0) {
                                /* back before the step, which the next round takes again */
                                pos[3 * r] = px - dx * ds, pos[3 * r + 1] = py - dy * ds, pos[3 * r + 2] = pz - dz * ds;
                                pg.tentative[r] = ds;
                        } else if (out.order_of != nullptr) { /* (RayOut: the caller's arrays) */
                                const long o = out.order_of[r];
                                out.pos[3 * o] = d_along<FAST>(px, dx, ds1), out.pos[3 * o + 1] = d_along<FAST>(py, dy, ds1),
                                out.pos[3 * o + 2] = d_along<FAST>(pz, dz, ds1);
                                out.index[2 * o] = bm, out.index[2 * o + 1] = bk;
                                out.length[o] = length[r] + (ds + ds1);
                                out.n_steps[o] = n_steps[r] + 1;
                                my_rays++;
                        } else { /* [ref stepper.c:861-863] */
                                pos[3 * r] = d_along<FAST>(px, dx, ds1), pos[3 * r + 1] = d_along<FAST>(py, dy, ds1),
                                pos[3 * r + 2] = d_along<FAST>(pz, dz, ds1);
                                index[2 * r] = bm, index[2 * r + 1] = bk;
                                length[r] = length[r] + (ds + ds1);
                                n_steps[r] = n_steps[r] + 1;
                                my_rays++;
                        }
                }
                if (CAN_FAULT && (pg.faulted != nullptr)) page_fault(pg, fault, r, home);
        }
        block_tally(stats, my_rays, my_rays, my_samples, 0);
}

/* ---- a whole scattering walk per ray ----------------------------------------
 *
 * turtle_stepper_scatter_n over a geometry with every tile resident: each lane
 * takes a ray through ALL its generations, the ray's state in registers from
 * its first step to its last -- where the generation-by-generation form
 * (k_step + k_bisect per generation) streams 136 bytes of it out and in again
 * at every step, which is what bounds that form.  Same state
 * machine as k_trace, one sample per live lane and trip whatever the lane is
 * doing (stepping, or bisecting a crossing: lanes need not be at the same
 * generation), with two differences: an accepted step ends a generation (a new
 * direction from Philox(first + ray, generation; seed)), and a located crossing
 * does not end the ray but moves it into the medium it entered, from the last
 * sample the bisection took there [ref stepper.c:849-858: that sample is what
 * turtle_stepper_step publishes and caches for the next call].  Same arithmetic
 * on the same values as the generation-by-generation form: same bits.
 * Persistent waves, rays from a queue (wave-aggregated draws), as in k_trace. */
struct WalkIO {
        ull seed;
        long first;      /* the global index of ray 0 */
        int first_step;  /* the generation of the first step */
        int n_steps;     /* generations to take */
};

#ifndef WALK_WAVES_ATTR
#define WALK_WAVES_ATTR
#endif
template <int MODE, bool FAST>
__global__ void __launch_bounds__(256) WALK_WAVES_ATTR k_walk(tamd_view v, long n, double * __restrict__ pos,
    double * __restrict__ alt, double * __restrict__ elev, int * __restrict__ index,
    double * __restrict__ length, int * __restrict__ steps, WalkIO io, ull * __restrict__ stats,
    ull * __restrict__ queue)
{
        long pool_next = 0, pool_end = 0; /* wave-uniform */
        bool exhausted = false;            /* wave-uniform */
        OneCtx ctx;
        d_load_ctx<MODE, FAST>(v, ctx);
        CellCache cell = { ~0u, 0u, 0u, -1, nullptr };
        CellCache * cache = (FAST && (MODE != TAMD_MODE_GENERIC)) ? &cell : nullptr;

        long ray = -1;
        bool dead = false;
        int state = ST_STEP, count = 0;
        double bx = 0, by = 0, bz = 0, dx = 0, dy = 0, dz = 0, len = 0;
        double ds = 0, ds0 = 0, ds1 = 0;
        double s_alt = 0, s_e0 = 0, s_e1 = 0; /* the sample the ray stands on */
        double b_alt = 0, b_e0 = 0, b_e1 = 0; /* the bisection's last sample of the new medium */
        int m = -1, k = -1, bm = -1, bk = -1, halvings = 0;
        ull my_rays = 0, my_steps = 0, my_samples = 0, my_plain = 0;

        for (;;) {
                /* ---- refill idle lanes from the queue (as k_trace) ---- */
                for (;;) {
                        const bool need = (ray < 0) && !dead;
                        const ull mask = __ballot(need);
                        if (mask == 0) break;
                        if (pool_next >= pool_end) {
                                if (exhausted) {
                                        if (need) dead = true;
                                        break;
                                }
                                ull base = 0;
                                if ((threadIdx.x & 63) == 0) base = atomicAdd(queue, (ull)kChunk);
                                base = __shfl(base, 0, 64);
                                pool_next = (long)base;
                                pool_end = min((long)base + kChunk, n);
                                if ((long)base >= n) {
                                        exhausted = true;
                                        pool_next = pool_end = 0;
                                }
                                continue;
                        }
                        const long avail = pool_end - pool_next;
                        const int rank = __builtin_amdgcn_mbcnt_hi(
                            (unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                        if (need && (rank < avail)) {
                                ray = pool_next + rank;
                                m = index[2 * ray], k = index[2 * ray + 1];
                                if ((m < 0) || (io.n_steps <= 0)) {
                                        ray = -1; /* has left the data: no further step */
                                } else {
                                        bx = pos[3 * ray], by = pos[3 * ray + 1], bz = pos[3 * ray + 2];
                                        s_alt = alt[ray], s_e0 = elev[2 * ray], s_e1 = elev[2 * ray + 1];
                                        /* the sum goes on where it stands: the same roundings
                                         * in one call as in several */
                                        len = length[ray], count = 0, state = ST_STEP;
                                        ds = d_step_length(v, s_alt, s_e0, s_e1, m);
                                        d_isotropic((ull)(io.first + ray), (ull)io.first_step, io.seed, dx, dy, dz);
                                }
                        }
                        pool_next += min((long)__popcll(mask), avail);
                }
                if (__ballot(ray >= 0) == 0) {
                        if (exhausted) break;
                        continue; /* (every ray drawn had left the data: draw again) */
                }
                if (ray >= 0) {
                        /* ---- one sample at q = B + d * t ---- */
                        const double t = (state == ST_STEP) ? ds : 0.5 * (ds0 + ds1);
                        const double qx = bx + dx * t, qy = by + dy * t, qz = bz + dz * t;
                        Sample s;
                        d_sample<MODE, FAST>(v, ctx, qx, qy, qz, s, cache);
                        my_samples++;
                        /* ---- bookkeeping: STEP and BISECT together, as selects (k_trace) ---- */
                        const bool stepping = (state == ST_STEP);
                        const bool same = (s.m == m);
                        const bool accept = stepping & same;
                        const bool cross = stepping & !same;
                        const bool other = !same;
                        bx = stepping ? qx : bx, by = stepping ? qy : by, bz = stepping ? qz : bz;
                        bm = other ? s.m : bm, bk = other ? s.k : bk;
                        b_alt = other ? s.alt : b_alt, b_e0 = other ? s.e0 : b_e0, b_e1 = other ? s.e1 : b_e1;
                        ds0 = cross ? -ds : ((!stepping & same) ? t : ds0);
                        ds1 = cross ? 0. : ((!stepping & other) ? t : ds1);
                        halvings = stepping ? 0 : halvings + 1;
                        state = cross ? ST_BISECT : state;
                        my_steps += stepping ? 1 : 0;
                        my_plain += accept ? 1 : 0;
                        const bool located = (state == ST_BISECT) & !cross &
                            (!(ds1 - ds0 > 1E-08) | (halvings > 1200));
                        bool ended = accept;
                        if (accept) {
                                len += ds, k = s.k;
                                s_alt = s.alt, s_e0 = s.e0, s_e1 = s.e1;
                        }
                        if (located) { /* [ref stepper.c:861-863] */
                                bx = bx + dx * ds1, by = by + dy * ds1, bz = bz + dz * ds1;
                                len += ds + ds1;
                                m = bm, k = bk;
                                s_alt = b_alt, s_e0 = b_e0, s_e1 = b_e1;
                                state = ST_STEP;
                                ended = true;
                        }
                        if (ended) { /* a generation is over */
                                count++;
                                if ((m < 0) || (count >= io.n_steps)) {
                                        pos[3 * ray] = bx, pos[3 * ray + 1] = by, pos[3 * ray + 2] = bz;
                                        alt[ray] = s_alt;
                                        elev[2 * ray] = (m >= 0) ? s_e0 : 0., elev[2 * ray + 1] = (m >= 0) ? s_e1 : 0.;
                                        index[2 * ray] = m, index[2 * ray + 1] = k;
                                        length[ray] = len, steps[ray] += count;
                                        my_rays++;
                                        ray = -1;
                                } else {
                                        ds = d_step_length(v, s_alt, s_e0, s_e1, m);
                                        d_isotropic((ull)(io.first + ray), (ull)(io.first_step + count), io.seed,
                                            dx, dy, dz);
                                }
                        }
                }
        }
        block_tally(stats, my_rays, my_steps, my_samples, my_plain);
}

__global__ void k_philox(long n, ull seed, ull stream, long first, unsigned * __restrict__ out)
{
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n;
             r += (long)gridDim.x * blockDim.x) {
                const ull id = (ull)(first + r);
                unsigned c[4] = { (unsigned)id, (unsigned)(id >> 32), (unsigned)stream,
                        (unsigned)(stream >> 32) };
                philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
                for (int i = 0; i < 4; i++) out[4 * r + i] = c[i];
        }
}

__global__ void k_isotropic(long n, ull seed, ull stream, long first, double * __restrict__ dir)
{
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n;
             r += (long)gridDim.x * blockDim.x) {
                double x, y, z;
                d_isotropic((ull)(first + r), stream, seed, x, y, z);
                dir[3 * r] = x, dir[3 * r + 1] = y, dir[3 * r + 2] = z;
        }
}

/* hits[m + 1] and a linear path-length histogram, exact integer counts.
 * Per-block LDS counters (32-bit) flushed with one 64-bit atomic per bin. */
__global__ void __launch_bounds__(256) k_tally(long n, const int * __restrict__ index,
    const double * __restrict__ length, int n_media, ull * __restrict__ hits,
    int n_bins, double scale, ull * __restrict__ histogram)
{
        extern __shared__ unsigned int lds[];
        unsigned int * h_hits = lds;                 /* n_media + 1 */
        unsigned int * h_bins = lds + (n_media + 1); /* n_bins + 1 */
        const int total = n_media + 1 + n_bins + 1;
        for (int i = threadIdx.x; i < total; i += blockDim.x) lds[i] = 0;
        __syncthreads();
        for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n;
             r += (long)gridDim.x * blockDim.x) {
                const int m = index[2 * r];
                if ((m >= -1) && (m < n_media)) atomicAdd(&h_hits[m + 1], 1u);
                const double t = length[r] * scale;
                int b = n_bins; /* overflow, also NaN and negatives */
                if ((t >= 0.) && (t < (double)n_bins)) b = (int)t;
                atomicAdd(&h_bins[b], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const unsigned int c = lds[i];
                if (c == 0) continue;
                if (i <= n_media)
                        atomicAdd(&hits[i], (ull)c);
                else
                        atomicAdd(&histogram[i - (n_media + 1)], (ull)c);
        }
}

} /* namespace */

/* ======================================================================== */
/*                         host side of the device layer                    */
/* ======================================================================== */

/* Per host THREAD: the device it works on, its stream, its scratch arena and
 * its blocks of bookkeeping memory.  The reference's rule is one stepper (and one
 * client) per thread over shared maps and stacks [ref include/turtle.h:129-132,
 * :620-626, examples/example-pthread.c:66-125]; here a thread also has a device:
 * the one LOCAL_RANK names (else 0) until it calls turtle_amd_device_set, so one
 * process can drive several GPUs, a thread each.  What threads share -- map
 * nodes, a stack's tiles -- is uploaded per device and changed under one lock
 * (host.h: tamd_geometry_lock). */
struct Ctx {
        int device = -1, cus = 0;
        hipStream_t own_stream = nullptr, stream = nullptr;
        int math_strict = 0;
        int in_flight = 1; /* batches the thread keeps in flight (tamd_dev_in_flight_set) */
        void * scratch = nullptr;
        size_t scratch_size = 0, scratch_used = 0;
        void * block[2] = { nullptr, nullptr }; /* grow-only: the pager's lists, a stack's own tables */
        size_t block_size[2] = { 0, 0 };
        void * pinned = nullptr; /* host memory the device can copy from / to without staging */
        size_t pinned_size = 0;
        void release()
        {
                if (device < 0) return;
                if (hipSetDevice(device) != hipSuccess) return;
                if (own_stream != nullptr) (void)hipStreamSynchronize(own_stream), (void)hipStreamDestroy(own_stream);
                if (scratch != nullptr) (void)hipFree(scratch);
                for (int i = 0; i < 2; i++)
                        if (block[i] != nullptr) (void)hipFree(block[i]);
                if (pinned != nullptr) (void)hipHostFree(pinned);
                pinned = nullptr, pinned_size = 0;
                own_stream = stream = nullptr, scratch = nullptr, scratch_size = scratch_used = 0;
                block[0] = block[1] = nullptr, block_size[0] = block_size[1] = 0;
        }
};
static thread_local char g_error[512] = "";
static thread_local Ctx g_ctx;
/* A thread that ends without turtle_amd_thread_release() (a pool's worker, an OpenMP
 * thread) gives back its stream, arena, blocks and pinned buffer here -- while the
 * process lives: at process exit the HIP runtime may already be gone, and the
 * main thread's context is left to it. */
static thread_local struct CtxGuard {
        bool armed = false; /* (set by tamd_dev_select on any thread but the main one) */
        ~CtxGuard()
        {
                if (armed) g_ctx.release();
        }
} g_ctx_guard;
#define g_stream (g_ctx.stream)
#define g_cus (g_ctx.cus)
#define g_math_strict (g_ctx.math_strict)

static int fail(const char * what, hipError_t e)
{
        snprintf(g_error, sizeof(g_error), "%s: %s (HIP error %d)", what,
            hipGetErrorString(e), (int)e);
        return 1;
}

#define HIP_TRY(call)                                                          \
        do {                                                                   \
                const hipError_t e_ = (call);                                  \
                if (e_ != hipSuccess) return fail(#call, e_);                  \
        } while (0)

extern "C" const char * tamd_dev_error(void) { return g_error; }

extern "C" int tamd_dev_count(void)
{
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess) return 0;
        return count;
}

#ifdef TRACE_POOL_STATS
extern "C" int tamd_dev_pool_stats(unsigned long long * out, int reset)
{
        HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pool_stats), sizeof(g_pool_stats)));
        if (reset) {
                static unsigned long long zero[32];
                HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pool_stats), zero, sizeof(zero)));
        }
        return 0;
}
#endif

extern "C" int tamd_dev_select(int device)
{
        const int count = tamd_dev_count();
        if (count <= 0) {
                snprintf(g_error, sizeof(g_error),
                    "no HIP device is visible: libturtle_amd has no CPU path");
                return 1;
        }
        if ((device < 0) || (device >= count)) {
                snprintf(g_error, sizeof(g_error),
                    "invalid device index %d (have %d)", device, count);
                return 1;
        }
        if (g_ctx.device == device) {
                HIP_TRY(hipSetDevice(device));
                return 0;
        }
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
                snprintf(g_error, sizeof(g_error),
                    "device %d is %s: libturtle_amd carries gfx950 code only",
                    device, prop.gcnArchName);
                return 1;
        }
        /* what this thread held on its previous device goes (its stream too: a
         * stream handed in by turtle_amd_stream_set belonged to that device) */
        g_ctx.release();
        HIP_TRY(hipSetDevice(device));
        g_ctx.device = device;
        g_ctx_guard.armed = ((long)syscall(SYS_gettid) != (long)getpid()); /* not the main thread: see CtxGuard */
        g_ctx.cus = prop.multiProcessorCount;
        HIP_TRY(hipStreamCreateWithFlags(&g_ctx.own_stream, hipStreamNonBlocking));
        g_ctx.stream = g_ctx.own_stream;
        return 0;
}

extern "C" int tamd_dev_init(void)
{
        if (g_ctx.device >= 0) {
                HIP_TRY(hipSetDevice(g_ctx.device)); /* HIP's current device is per thread too */
                return 0;
        }
        int device = 0;
        const char * env = getenv("LOCAL_RANK");
        if ((env != nullptr) && (*env != 0)) {
                const int count = tamd_dev_count();
                if (count > 0) device = atoi(env) % count;
        }
        return tamd_dev_select(device);
}

extern "C" int tamd_dev_current(void) { return g_ctx.device; }

/* what the calling thread holds on its device (a worker calls it before it ends:
 * nothing is freed behind a thread's back, the runtime may be gone by then) */
extern "C" void tamd_dev_release(void)
{
        g_ctx.release();
        g_ctx.device = -1;
}
extern "C" int tamd_dev_cus(void) { return (tamd_dev_init() == 0) ? g_ctx.cus : 0; }

extern "C" int tamd_dev_stream_set(void * stream)
{
        if (tamd_dev_init()) return 1;
        g_ctx.stream = (stream != nullptr) ? (hipStream_t)stream : g_ctx.own_stream;
        return 0;
}

extern "C" int tamd_dev_sync(void)
{
        if (tamd_dev_init()) return 1;
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        return 0;
}

/* every stream of `device` (before memory that other threads' launches may still
 * read is freed); leaves the calling thread on its own device */
/* Every stream of that device has drained (non-zero: it could not be told -- the
 * caller then LEAKS what it meant to free there, rather than free memory that a
 * launch may still read).  The calling thread is back on its own device on every
 * path. */
extern "C" int tamd_dev_sync_device(int device)
{
        if (device < 0) return 0;
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if ((g_ctx.device >= 0) && (g_ctx.device != device)) {
                const hipError_t back = hipSetDevice(g_ctx.device);
                if (e == hipSuccess) e = back;
        }
        return (e == hipSuccess) ? 0 : fail("tamd_dev_sync_device", e);
}

extern "C" int tamd_dev_malloc(void ** ptr, size_t bytes)
{
        *ptr = nullptr;
        if (tamd_dev_init()) return 1;
        HIP_TRY(hipMalloc(ptr, bytes ? bytes : 1));
        return 0;
}

extern "C" void tamd_dev_free(void * ptr)
{
        if (ptr != nullptr) (void)hipFree(ptr);
}

/* memory of another device than the calling thread's */
extern "C" void tamd_dev_free_on(int device, void * ptr)
{
        if (ptr == nullptr) return;
        if ((device >= 0) && (device != g_ctx.device)) (void)hipSetDevice(device);
        (void)hipFree(ptr);
        if ((device >= 0) && (device != g_ctx.device) && (g_ctx.device >= 0)) (void)hipSetDevice(g_ctx.device);
}

extern "C" int tamd_dev_h2d(void * dst, const void * src, size_t bytes)
{
        if (tamd_dev_init()) return 1;
        if (bytes == 0) return 0;
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        return 0;
}

extern "C" int tamd_dev_d2h(void * dst, const void * src, size_t bytes)
{
        if (tamd_dev_init()) return 1;
        if (bytes == 0) return 0;
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        return 0;
}

extern "C" int tamd_dev_zero(void * dst, size_t bytes)
{
        if (tamd_dev_init()) return 1;
        HIP_TRY(hipMemsetAsync(dst, 0, bytes, g_ctx.stream));
        return 0;
}

extern "C" int tamd_dev_pinned(void ** ptr, size_t bytes)
{
        *ptr = nullptr;
        if (tamd_dev_init()) return 1;
        if (bytes > g_ctx.pinned_size) {
                HIP_TRY(hipStreamSynchronize(g_ctx.stream));
                if (g_ctx.pinned != nullptr) (void)hipHostFree(g_ctx.pinned);
                g_ctx.pinned = nullptr, g_ctx.pinned_size = 0;
                HIP_TRY(hipHostMalloc(&g_ctx.pinned, bytes, hipHostMallocDefault));
                g_ctx.pinned_size = bytes;
        }
        *ptr = g_ctx.pinned;
        return 0;
}

/* page-locked host memory that outlives the call (a stack's staging buffers for its
 * tiles: a copy from it is queued, not waited for) */
extern "C" int tamd_dev_host_alloc(void ** ptr, size_t bytes)
{
        *ptr = nullptr;
        if (tamd_dev_init()) return 1;
        HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
        return 0;
}

extern "C" void tamd_dev_host_free(void * ptr)
{
        if (ptr != nullptr) (void)hipHostFree(ptr);
}

extern "C" int tamd_dev_copy_async(void * dst, const void * src, size_t bytes, int to_device)
{
        if (tamd_dev_init()) return 1;
        if (bytes == 0) return 0;
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
            g_ctx.stream));
        return 0;
}

extern "C" void tamd_scratch_reset(void) { g_ctx.scratch_used = 0; }

extern "C" int tamd_scratch_get(void ** ptr, size_t bytes)
{
        *ptr = nullptr;
        if (tamd_dev_init()) return 1;
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (g_ctx.scratch_used + need > g_ctx.scratch_size) {
                if (g_ctx.scratch_used != 0) {
                        /* pieces already handed out would dangle: the host layer
                         * sizes the arena up front with one oversize request */
                        snprintf(g_error, sizeof(g_error), "scratch arena exhausted");
                        return 1;
                }
                HIP_TRY(hipStreamSynchronize(g_ctx.stream));
                if (g_ctx.scratch) (void)hipFree(g_ctx.scratch);
                g_ctx.scratch = nullptr, g_ctx.scratch_size = 0;
                const size_t size = need + (need >> 2) + (1u << 20);
                HIP_TRY(hipMalloc(&g_ctx.scratch, size));
                g_ctx.scratch_size = size;
        }
        *ptr = (char *)g_ctx.scratch + g_ctx.scratch_used;
        g_ctx.scratch_used += need;
        return 0;
}

/* One of the calling thread's grow-only blocks (0: the pager's lists and counters,
 * 1: the tables of a stack's own batch calls), at least `bytes` long; *grown is
 * set when it is a new allocation (what it held is gone) */
extern "C" int tamd_dev_block(int which, void ** ptr, size_t bytes, int * grown)
{
        *ptr = nullptr;
        if (grown != nullptr) *grown = 0;
        if (tamd_dev_init()) return 1;
        if (bytes > g_ctx.block_size[which]) {
                if (g_ctx.block[which] != nullptr) {
                        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
                        (void)hipFree(g_ctx.block[which]);
                        g_ctx.block[which] = nullptr, g_ctx.block_size[which] = 0;
                }
                HIP_TRY(hipMalloc(&g_ctx.block[which], bytes));
                g_ctx.block_size[which] = bytes;
                if (grown != nullptr) *grown = 1;
        }
        *ptr = g_ctx.block[which];
        return 0;
}

static int grid_for(long n, int block)
{
        long blocks = (n + block - 1) / block;
        /* (4, 16 or 64 blocks per CU: the same or worse, measured on the step kernel) */
        const long cap = (long)(g_cus > 0 ? g_cus : 256) * 8;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        return (int)blocks;
}

#define LAUNCH_CHECK(name)                                                     \
        do {                                                                   \
                const hipError_t e_ = hipGetLastError();                       \
                if (e_ != hipSuccess) return fail("launch " name, e_);         \
        } while (0)

extern "C" int tamd_k_ecef_from_geodetic(long n, const double * lat,
    const double * lon, const double * elev, double * ecef)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_ecef_from_geodetic, dim3(grid_for(n, 256)), dim3(256), 0,
            g_stream, n, lat, lon, elev, ecef);
        LAUNCH_CHECK("k_ecef_from_geodetic");
        return 0;
}

extern "C" int tamd_k_ecef_to_geodetic(
    long n, const double * ecef, double * lat, double * lon, double * alt)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        if (g_math_strict)
                hipLaunchKernelGGL(k_ecef_to_geodetic<false>, dim3(grid_for(n, 256)),
                    dim3(256), 0, g_stream, n, ecef, lat, lon, alt);
        else
                hipLaunchKernelGGL(k_ecef_to_geodetic<true>, dim3(grid_for(n, 256)),
                    dim3(256), 0, g_stream, n, ecef, lat, lon, alt);
        LAUNCH_CHECK("k_ecef_to_geodetic");
        return 0;
}

extern "C" int tamd_k_ecef_from_horizontal(long n, const double * lat,
    const double * lon, const double * az, const double * el, double * dir)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_ecef_from_horizontal, dim3(grid_for(n, 256)), dim3(256), 0,
            g_stream, n, lat, lon, az, el, dir);
        LAUNCH_CHECK("k_ecef_from_horizontal");
        return 0;
}

extern "C" int tamd_k_ecef_to_horizontal(long n, const double * lat,
    const double * lon, const double * dir, double * az, double * el)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_ecef_to_horizontal, dim3(grid_for(n, 256)), dim3(256), 0,
            g_stream, n, lat, lon, dir, az, el);
        LAUNCH_CHECK("k_ecef_to_horizontal");
        return 0;
}

extern "C" int tamd_k_elevation(struct tamd_view view, long n, const double * a,
    const double * b, double * z, int * inside, struct tamd_paging pg)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_elevation, dim3(grid_for(n, 256)), dim3(256), 0, g_stream,
            view, n, a, b, z, inside, pg);
        LAUNCH_CHECK("k_elevation");
        return 0;
}

extern "C" int tamd_k_project(struct tamd_proj proj, int inverse, long n, const double * a,
    const double * b, double * c, double * d)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_project, dim3(grid_for(n, 256)), dim3(256), 0, g_stream, proj,
            inverse, n, a, b, c, d);
        LAUNCH_CHECK("k_project");
        return 0;
}

extern "C" int tamd_k_gradient(struct tamd_view view, long n, const double * a,
    const double * b, double * ga, double * gb, int * inside, struct tamd_paging pg)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_gradient, dim3(grid_for(n, 256)), dim3(256), 0, g_stream, view,
            n, a, b, ga, gb, inside, pg);
        LAUNCH_CHECK("k_gradient");
        return 0;
}

extern "C" int tamd_k_position(struct tamd_view view, long n, const double * lat,
    const double * lon, const double * height, int layer, double * pos,
    int * data_index, struct tamd_paging pg)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_position, dim3(grid_for(n, 256)), dim3(256), 0, g_stream,
            view, n, lat, lon, height, layer, pos, data_index, pg);
        LAUNCH_CHECK("k_position");
        return 0;
}

/* n single steps: the step kernel, then -- with a direction and scratch for
 * the list -- the bisection of the rays that crossed a boundary.  stats /
 * queue: as for a trace (queue[2 * stride] counts the listed rays), or NULL. */
static int run_step(struct tamd_view view, long n, double * pos, const double * dir,
    double * lat, double * lon, double * alt, double * elev, double * step, int * index,
    int flags, CrossList cross, Paging pg, ull * stats, StepWalk walk)
{
        const dim3 grid(grid_for(n, 256)), block(256);
        const bool strict = g_math_strict || !view.fast_ok;
#define STEP_CASE(MODE)                                                                        \
        do {                                                                                   \
                if (strict)                                                                    \
                        hipLaunchKernelGGL((k_step<MODE, false>), grid, block, 0, g_stream, view, n,   \
                            pos, dir, lat, lon, alt, elev, step, index, flags, cross, pg, stats, walk); \
                else if (MODE == TAMD_MODE_GENERIC)                                            \
                        hipLaunchKernelGGL((k_step<MODE, true>), grid, block, 0, g_stream, view, n,    \
                            pos, dir, lat, lon, alt, elev, step, index, flags, cross, pg, stats, walk); \
                else                                                                           \
                        hipLaunchKernelGGL((k_step_fast<MODE>), grid, block, 0, g_stream, view, n,     \
                            pos, dir, lat, lon, alt, elev, step, index, flags, cross, pg, stats, walk); \
                LAUNCH_CHECK("k_step");                                                        \
                if (cross.ray == nullptr) break;                                               \
                /* the listed rays are a few percent of n, and their number is on the        \
                 * device: a grid for a tenth of n, striding over whatever there is */         \
                const dim3 few(grid_for(n / 10 + 1, 256));                                     \
                if (strict)                                                                    \
                        hipLaunchKernelGGL((k_bisect<MODE, false>), few, block, 0, g_stream, view,     \
                            pos, dir, lat, lon, alt, elev, step, index, flags, cross, pg, stats, walk); \
                else                                                                           \
                        hipLaunchKernelGGL((k_bisect<MODE, true>), few, block, 0, g_stream, view,      \
                            pos, dir, lat, lon, alt, elev, step, index, flags, cross, pg, stats, walk); \
                LAUNCH_CHECK("k_bisect");                                                      \
        } while (0)
        if (view.mode == TAMD_MODE_ONE_MAP)
                STEP_CASE(TAMD_MODE_ONE_MAP);
        else if (view.mode == TAMD_MODE_ONE_STACK)
                STEP_CASE(TAMD_MODE_ONE_STACK);
        else
                STEP_CASE(TAMD_MODE_GENERIC);
#undef STEP_CASE
        return 0;
}

extern "C" int tamd_k_step(struct tamd_view view, long n, double * pos,
    const double * dir, double * lat, double * lon, double * alt, double * elev,
    double * step, int * index, int flags, struct tamd_paging pg)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        const CrossList none = { nullptr, nullptr, nullptr, nullptr };
        const StepWalk no_walk = { 0, 0, 0, 0, nullptr, nullptr };
        return run_step(view, n, pos, dir, lat, lon, alt, elev, step, index, flags, none, pg,
            nullptr, no_walk);
}

/* Waves per SIMD the trace kernel is launched with.  It is fp64-VALU bound
 * with a dependent 4-node gather per sample, so a few waves per SIMD are
 * enough to cover the gather latency; TURTLE_AMD_TRACE_WAVES overrides the
 * default for experiments. */
static int trace_blocks_per_cu(const void * kernel)
{
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, 256, 0) !=
                hipSuccess ||
            blocks < 1)
                blocks = 1;
        /* batches in flight share the SIMDs: a kernel that takes one block a CU fewer than fit
         * leaves registers for a wave of another batch's kernel beside its own (three C2 batches
         * in flight: 2.38 -> 2.29 ms a pass over 10 passes, 2.39 -> 2.18 over 20; C4 25.5 -> 24.7;
         * through a stack no change -- two blocks a CU whatever fits: C2 the same, C4 24.1, but C3
         * 23.9 -> 25.3; alone a kernel would lose either way: C2 3.55 -> 3.7 with two) */
        if ((g_ctx.in_flight > 1) && (blocks > 2)) {
                static int share = -1; /* experiments: 0: as many as fit, 1: one fewer, 2: two */
                if (share < 0) {
                        const char * e = getenv("TURTLE_AMD_IN_FLIGHT_SHARE");
                        share = ((e != nullptr) && (*e != 0)) ? atoi(e) : 1;
                }
                if (share == 1) blocks -= 1;
                if (share == 2) blocks = 2;
        }
        const char * env = getenv("TURTLE_AMD_TRACE_WAVES");
        if ((env != nullptr) && (*env != 0)) {
                const int waves = atoi(env); /* per SIMD == blocks of 256 per CU */
                if ((waves >= 1) && (waves < blocks)) blocks = waves;
        }
        return blocks;
}

extern "C" void tamd_dev_math_set(int strict) { g_ctx.math_strict = strict ? 1 : 0; }
extern "C" void tamd_dev_in_flight_set(int batches) { g_ctx.in_flight = (batches > 1) ? batches : 1; }
extern "C" int tamd_dev_in_flight_get(void) { return g_ctx.in_flight; }
extern "C" int tamd_dev_math_get(void) { return g_ctx.math_strict; }

template <int MODE, bool FAST, bool MODEL, bool PAGED, bool CROSS, bool POOL = false>
static int launch_trace_(struct tamd_view view, long n, bool n_on_device, double * pos,
    const double * dir, int max_steps, int * index, double * length, int * n_steps,
    int flags, PhaseIO ph, ull * stats, ull * queue)
{
        const void * kernel = (const void *)k_trace<MODE, FAST, MODEL, PAGED, CROSS, POOL>;
        long blocks = (long)g_cus * trace_blocks_per_cu(kernel);
        const long useful = (n + 255) / 256;
        if (!n_on_device && (blocks > useful)) blocks = useful;
        if (n_on_device) {
                /* phase B: the rays phase A handed over -- the long ones (a few
                 * percent of n) and whatever was in flight when its queue ran dry
                 * (up to one ray per lane): as many blocks as fit, or as there can
                 * be work for (TURTLE_AMD_TAIL_DIV: fewer, for experiments) */
                static int div = 0;
                if (div == 0) {
                        const char * env = getenv("TURTLE_AMD_TAIL_DIV");
                        div = ((env != nullptr) && (*env != 0)) ? atoi(env) : 1;
                        if (div < 1) div = 1;
                }
                long wide = useful / div;
                if (wide < (long)g_cus) wide = (long)g_cus;
                if (blocks > wide) blocks = wide;
        }
        hipLaunchKernelGGL((k_trace<MODE, FAST, MODEL, PAGED, CROSS, POOL>), dim3((unsigned)blocks), dim3(256),
            0, g_stream, view, n, pos, dir, max_steps, index, length, n_steps, flags, ph, stats,
            queue);
        LAUNCH_CHECK("k_trace");
        return 0;
}

/* the instance for this call: PAGED where tiles may have to come in, CROSS where
 * there is a list for the crossings (always, for a lined pass) */
template <int MODE, bool FAST, bool MODEL>
static int launch_trace(struct tamd_view view, long n, bool n_on_device, double * pos,
    const double * dir, int max_steps, int * index, double * length, int * n_steps,
    int flags, PhaseIO ph, ull * stats, ull * queue)
{
#define TRACE_ARGS view, n, n_on_device, pos, dir, max_steps, index, length, n_steps, flags, ph, stats, queue
        constexpr bool CAN_PAGE = (MODE != TAMD_MODE_ONE_MAP);
        const bool paged = CAN_PAGE && (ph.pg.faulted != nullptr);
        const bool cross = (ph.cross.ray != nullptr);
        if (MODEL && !cross) {
                snprintf(g_error, sizeof(g_error), "k_trace: a lined pass needs the crossing list");
                return 1;
        }
        if (paged) {
                if (cross) return launch_trace_<MODE, FAST, MODEL, CAN_PAGE, true>(TRACE_ARGS);
                if constexpr (!MODEL) return launch_trace_<MODE, FAST, MODEL, CAN_PAGE, false>(TRACE_ARGS);
        }
        /* the lined pass of one map / one stack with its rays pooled per block (RayPool): an
         * instance of its own, so that the one without is what it was */
        if constexpr (FAST && MODEL && (MODE != TAMD_MODE_GENERIC))
                if (cross && (ph.pool > 0)) return launch_trace_<MODE, FAST, MODEL, false, true, true>(TRACE_ARGS);
        if (cross) return launch_trace_<MODE, FAST, MODEL, false, true>(TRACE_ARGS);
        if constexpr (!MODEL) return launch_trace_<MODE, FAST, MODEL, false, false>(TRACE_ARGS);
        return 1;
#undef TRACE_ARGS
}

/* the crossings the passes listed (their number is on the device: a grid for
 * all of n, striding over whatever there is) */
template <int MODE, bool FAST>
static int launch_cross(struct tamd_view view, long n, double * pos, const double * dir, int * index,
    double * length, int * n_steps, CrossList cross, Paging pg, ull * stats,
    RayOut out = { nullptr, nullptr, nullptr, nullptr, nullptr })
{
        constexpr bool CAN_PAGE = (MODE != TAMD_MODE_ONE_MAP);
        const bool paged = CAN_PAGE && (pg.faulted != nullptr);
        const void * kernel = paged ? (const void *)k_cross<MODE, FAST, CAN_PAGE> :
                                      (const void *)k_cross<MODE, FAST, false>;
        long blocks = (long)g_cus * trace_blocks_per_cu(kernel);
        const long useful = (n + 255) / 256;
        if (blocks > useful) blocks = useful;
        if (paged)
                hipLaunchKernelGGL((k_cross<MODE, FAST, CAN_PAGE>), dim3((unsigned)blocks), dim3(256), 0,
                    g_stream, view, pos, dir, index, length, n_steps, cross, pg, stats, out);
        else
                hipLaunchKernelGGL((k_cross<MODE, FAST, false>), dim3((unsigned)blocks), dim3(256), 0,
                    g_stream, view, pos, dir, index, length, n_steps, cross, pg, stats, out);
        LAUNCH_CHECK("k_cross");
        return 0;
}

static int env_int(const char * name, int fallback)
{
        const char * env = getenv(name);
        return ((env != nullptr) && (*env != 0)) ? atoi(env) : fallback;
}

/* Step counts at which a ray moves on to the next phase of a fast trace (0: no
 * further phase), and the rays a wave of phase A may still hold when it hands
 * over after the queue ran dry.  TURTLE_AMD_* override them for experiments. */
/* The lean steps of the lined pass (one map, a regular stack) cost a tenth of a
 * closed form in instructions: with the line from step 32 on instead of 512, C2
 * (1 M rays) goes from 7.2 to 6.0 ms, one map at 10 M rays from 36.1 to 35.4, C3
 * (a stack, 10 M rays) from 42.4 to 39.8 ms -- the last only with the stack's lined
 * kernel at three waves a SIMD (trace_waves(); at two, 47.5).  Layered geometries
 * have no lean loop. */
static int park_threshold(int mode, long n)
{
        static int value = -2;
        (void)n;
        if (value == -2) value = env_int("TURTLE_AMD_PARK", -1);
        if (value >= 0) return value;
        return (mode == TAMD_MODE_GENERIC) ? 512 : 32;
}
/* Few in a small batch, where the launch waits for single rays in all-but-empty
 * waves (C2, 1 M rays: 8 lanes 6.85 ms, 32 lanes 7.08, 64 lanes 7.4); more in a large
 * one, where phase B is a matter of throughput (C2 at 4 M rays: 20.5 -> 19.6 ms with 32;
 * C3, 10 M: 42.6 -> 41.8).  The loop gives the same bits whenever it engages. */
static int creep_lanes(long n)
{
        static int value = -2;
        if (value == -2) value = env_int("TURTLE_AMD_CREEP_LANES", -1);
        if (value >= 0) return value;
        return (n >= 2000000) ? 4 * kCreepLanes : kCreepLanes;
}
static int dense_go(void)
{
        static int value = -1;
        if (value < 0) value = env_int("TURTLE_AMD_DENSE_GO", 24);
        return value;
}
/* What phase A takes for a long ray when it sorts its hand-over (see there; 0: unsorted) */
static int sort_long_if(void)
{
        static int value = -1;
        if (value < 0) value = env_int("TURTLE_AMD_SORT_LONG", 120);
        return value;
}
/* Do the lined pass's waves exchange rays through LDS (RayPool)?  The same bits either way
 * (test_ray_pool_changes_no_bit); what it is worth, measured (round 4, one MI355X, each alone):
 * one map, 12.5 M rays (C4) 27.0 -> 26.0-26.4 ms; one map, 1 M rays (C2) 3.37 -> 3.55-3.72 ms
 * (a batch that small is as long as its longest rays' own chains, and a ray moves slower in a
 * wave that is kept full); a stack, 10 M rays (C3) 25.4 -> 31.7-33.9 ms (the stack's pooled
 * kernel spills 312 bytes a lane at three waves a SIMD); one map at 3 / 4 / 6 M rays: 8.29 ->
 * 8.42, 10.84 -> 11.15, 13.95 -> 13.74 ms.  So: one map, from 6 M rays on.
 * TURTLE_AMD_POOL=0 / 1: never / wherever the kernel exists. */
static int pool_on(int mode, long n)
{
        static int value = -2;
        if (value == -2) value = env_int("TURTLE_AMD_POOL", -1);
        if (value >= 0) return value;
        return (mode == TAMD_MODE_ONE_MAP) && (n >= 6000000);
}
/* Is the hand-over ordered between the passes (run_trace)?  A batch of a few million rays is as
 * long as its longest rays' own chains, and drawing those first is worth more than the sort costs
 * (TURTLE_AMD_SORT_KEY: 0 never, 1 always, else up to that many rays). */
static int sort_hand_over(int mode, long n)
{
        static long value = -2;
        (void)mode;
        if (value == -2) {
                const char * env = getenv("TURTLE_AMD_SORT_KEY");
                value = ((env != nullptr) && (*env != 0)) ? atol(env) : -1;
        }
        if (value == 0) return 0;
        if (value == 1) return 1;
        return n <= ((value > 1) ? value : 4000000L);
}
/* Does a trace take its rays in the order of where they start (k_ray_cells)?  TURTLE_AMD_SPATIAL:
 * 0 never, 1 always, else from that many rays on. */
static int spatial_order(int mode, long n)
{
        static long value = -2;
        if (value == -2) {
                const char * env = getenv("TURTLE_AMD_SPATIAL");
                value = ((env != nullptr) && (*env != 0)) ? atol(env) : -1;
        }
        if ((mode == TAMD_MODE_GENERIC) || (value == 0)) return 0;
        if (value == 1) return 1;
        /* (measured, one pass alone, off -> on: one map at 1 / 2 / 4 / 12.5 M rays 3.13 -> 3.20, 5.91 ->
         * 5.95, 11.17 -> 10.92, 26.7 -> 26.3 ms; a 4 x 4 stack at 1 / 10 M rays 4.10 -> 4.10, 25.5 -> 22.8) */
        return n >= ((value > 1) ? value : 3000000L);
}
/* the room behind the lists of a trace (internal.h, TAMD_TRACE_SORT_ROOM), on a 256-byte boundary */
static char * sort_room_of(int * parked, long n)
{
        const uintptr_t at = (uintptr_t)(parked + TAMD_TRACE_SORT_INTS * n);
        return (char *)((at + 255) & ~(uintptr_t)255);
}
static int drain_lanes(void)
{
        static int value = -1;
        if (value < 0) value = env_int("TURTLE_AMD_DRAIN", 64);
        return value;
}

/* One round of a trace: all the rays (pg.ids == NULL), or the ones the last
 * round listed because they needed a tile (they carry on from the arrays).
 *
 * The passes step; a ray whose step crossed a boundary goes on a list (CROSS, see
 * trace_body) and k_cross locates the crossings at the end, packed.
 * Fast arithmetic steps in two passes: A takes every ray by the closed form up to
 * park_threshold() steps (32, or 512: see there) and hands over what is left; B
 * takes those to their crossing on their lines.  A ray changes pass at a fixed
 * step count, or (below it, when A's queue ran dry) where its arithmetic does not
 * depend on the pass: see LINED.  (A third pass for the rays beyond a second
 * threshold, a few to a wave on an otherwise empty chip, was wired in until round
 * 3 and always off: on C2 every threshold from 256 to 2 048 made the trace slower,
 * 6.9-7.8 ms against 6.0 ms.)
 * Without scratch for the lists (`parked` NULL: a batch beyond 2^31 rays) there
 * is one pass, which bisects in place. */
constexpr int kQ = 16; /* words between two counters of a trace: see run_trace */
template <int MODE>
static int run_trace(struct tamd_view view, long n, double * pos, const double * dir,
    int max_steps, int * index, double * length, int * n_steps, int flags, int * parked,
    double * cross_ds, Paging pg, ull * stats, ull * queue)
{
        const bool again = (pg.ids != nullptr);
        const bool sort_room = (flags & TAMD_TRACE_SORT_ROOM) != 0;
        flags &= ~TAMD_TRACE_SORT_ROOM;
        if (again) flags |= TRACE_CARRY_MEDIUM;
        const int resume = again ? 2 : 0;
        const bool strict = g_math_strict || !view.fast_ok;
        /* lists: parked[0 .. n) from A to B (from both ends), parked[n .. 3n) the crossings;
         * counters, kQ words (a cache line or two) apart -- every wave of a pass adds to them:
         * queue[0], [kQ]: the work queues of A, B; queue[2 kQ], [3 kQ]: the lengths of the lists;
         * queue[4 kQ]: of the first list's far end */
        const bool listed = (parked != nullptr) && (cross_ds != nullptr) && (length != nullptr) &&
            (n_steps != nullptr);
        const CrossList none = { nullptr, nullptr, nullptr, nullptr };
        const CrossList cross = { parked + n, cross_ds, queue + 3 * kQ, parked + 2 * n };
        const PhaseIO one = { pg.ids, pg.n_in, nullptr, nullptr, 0, resume, pg, 0, 0, kChunk,
                creep_lanes(n), dense_go(), listed ? cross : none };
        if (!listed) {
                if (strict)
                        return launch_trace<MODE, false, false>(view, n, again, pos, dir, max_steps, index,
                            length, n_steps, flags, one, stats, queue);
                return launch_trace<MODE, true, false>(view, n, again, pos, dir, max_steps, index,
                    length, n_steps, flags, one, stats, queue);
        }
        if (strict) {
                if (launch_trace<MODE, false, false>(view, n, again, pos, dir, max_steps, index, length,
                        n_steps, flags, one, stats, queue))
                        return 1;
                return launch_cross<MODE, false>(view, n, pos, dir, index, length, n_steps, cross, pg, stats);
        }
        const int park = park_threshold(MODE, n);
        if ((park <= 0) || (max_steps <= park)) {
                if (launch_trace<MODE, true, false>(view, n, again, pos, dir, max_steps, index, length,
                        n_steps, flags, one, stats, queue))
                        return 1;
                return launch_cross<MODE, true>(view, n, pos, dir, index, length, n_steps, cross, pg, stats);
        }
        PhaseIO a = { pg.ids, pg.n_in, parked, queue + 2 * kQ, park, resume, pg, drain_lanes(), 0,
                kChunk, creep_lanes(n), dense_go(), cross };
        PhaseIO b = { parked, queue + 2 * kQ, nullptr, nullptr, 0, 1, pg, 0, park, kChunk,
                creep_lanes(n), dense_go(), cross };
        b.pool = pool_on(MODE, n);
        const int long_if = sort_long_if();
        if (!again && (long_if > 0)) {
                /* (a later round of a paged trace takes rays at any step count: unsorted) */
                a.n_parked_back = queue + 4 * kQ, a.ds_mark = cross_ds + 2 * n, a.mark_at = park / 2;
                a.long_if = (float)long_if;
                b.n_dev_back = queue + 4 * kQ;
        }
        /* Room to ORDER the hand-over (internal.h, TAMD_TRACE_SORT_ROOM): keys beside the list, a
         * radix sort (hipCUB) of the whole list's n places between the two passes -- places nobody
         * filled carry a key (254) between the front's (0 .. 253) and the back's (255), so the front comes out first, in
         * order, and the back stays at the far end, where the lined pass looks for it.  For batches
         * small enough to be as long as their longest rays (sort_hand_over()). */
        /* The rays in the order of where they start (k_ray_cells): keys and numbers into the sort's
         * room, one pass of the radix sort.  The passes then WORK in that order -- a ray's state
         * between the passes (position, medium, path length, step count, its direction) lives at its
         * place in the ordered list, in arrays of the library's own -- while the caller's arrays are
         * touched twice a ray: phase A reads a new ray from its place there, and whichever kernel
         * ENDS a ray (k_cross for nearly all) writes its results to that place (RayOut); no pass of
         * its own copies anything.  (Tried first: the passes working in the caller's arrays through
         * the ordered list -- every hand-over then goes to a place of its own instead of next to
         * its wave's: half the gain on C3, a loss on C4; and copies made by kernels of their own:
         * 0.9 + 1.4 ms for C4's 12.5 M rays, more than the order gains there.)  Not over paged tiles
         * (the pager's lists name rays by their place in the CALLER's arrays, round after round). */
        RayOut out = { nullptr, nullptr, nullptr, nullptr, nullptr };
        if constexpr (MODE != TAMD_MODE_GENERIC) {
                if (sort_room && !again && (pg.faulted == nullptr) && spatial_order(MODE, n)) {
                        hipcub::DoubleBuffer<SPATIAL_KEY_T> cell((SPATIAL_KEY_T *)(parked + 5 * n),
                            (SPATIAL_KEY_T *)(parked + 6 * n));
                        hipcub::DoubleBuffer<int> list(parked, parked + 4 * n);
                        size_t bytes = 0;
                        if ((hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, cell, list, (int)n, 0, SPATIAL_BITS, g_stream) ==
                                hipSuccess) &&
                            (bytes <= TAMD_TRACE_SORT_TEMP)) {
                                hipLaunchKernelGGL(k_ray_cells<MODE>, dim3(grid_for(n, 256)), dim3(256), 0, g_stream,
                                    view, n, pos, cell.Current(), list.Current());
                                LAUNCH_CHECK("k_ray_cells");
                                char * const room = sort_room_of(parked, n);
                                if (hipcub::DeviceRadixSort::SortPairs((void *)room, bytes, cell, list, (int)n, 0, SPATIAL_BITS,
                                        g_stream) != hipSuccess)
                                        return fail("hipcub::DeviceRadixSort", hipGetLastError());
                                /* the arrays the passes work in, and the list, which must outlive them
                                 * (they use parked[0, n) and the sort's room again): 72 bytes a ray */
                                char * copies = room + TAMD_TRACE_SORT_TEMP;
                                double * const pos_s = (double *)copies;
                                double * const dir_s = pos_s + 3 * n;
                                double * const length_s = dir_s + 3 * n;
                                int * const index_s = (int *)(length_s + n);
                                int * const n_steps_s = index_s + 2 * n;
                                int * const kept = n_steps_s + n;
                                HIP_TRY(hipMemcpyAsync(kept, list.Current(), (size_t)n * sizeof(int),
                                    hipMemcpyDeviceToDevice, g_stream));
                                out.order_of = kept, out.pos = pos, out.index = index, out.length = length,
                                out.n_steps = n_steps;
                                a.out = out, b.out = out;
                                a.pos_in = pos, a.dir_in = dir, a.index_in = index, a.dir_copy = dir_s;
                                pos = pos_s, dir = dir_s, index = index_s, length = length_s, n_steps = n_steps_s;
                        }
                }
        }
        const bool order = sort_room && !again && (a.n_parked_back != nullptr) && sort_hand_over(MODE, n);
        hipcub::DoubleBuffer<unsigned char> keys((unsigned char *)(parked + 5 * n), (unsigned char *)(parked + 6 * n));
        hipcub::DoubleBuffer<int> ids(parked, parked + 4 * n);
        void * const sort_temp = (void *)sort_room_of(parked, n);
        size_t temp_bytes = 0;
        bool sorting = false;
        if (order) {
                if ((hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, ids, (int)n, 0, 8, g_stream) ==
                        hipSuccess) &&
                    (temp_bytes <= TAMD_TRACE_SORT_TEMP)) {
                        sorting = true;
                        a.sort_key = keys.Current();
                        HIP_TRY(hipMemsetAsync(keys.Current(), 0xFE, (size_t)n, g_stream));
                }
        }
        if (launch_trace<MODE, true, false>(view, n, again, pos, dir, max_steps, index, length,
                n_steps, flags, a, stats, queue))
                return 1;
        if (sorting) {
                if (hipcub::DeviceRadixSort::SortPairs(sort_temp, temp_bytes, keys, ids, (int)n, 0, 8, g_stream) !=
                    hipSuccess)
                        return fail("hipcub::DeviceRadixSort", hipGetLastError());
                b.ids = ids.Current();
        }
        if (launch_trace<MODE, true, true>(view, n, true, pos, dir, max_steps, index, length,
                n_steps, flags | TRACE_CARRY_MEDIUM, b, stats, queue + 1 * kQ))
                return 1;
        return launch_cross<MODE, true>(view, n, pos, dir, index, length, n_steps, cross, pg, stats, out);
}

/* queue: five counters (see run_trace); parked: room for 3 n ray ids and cross_ds
 * for 3 n doubles (the lists of the passes; what the hand-over sorts by in the last n),
 * or NULL.  pg: the round of a paged
 * geometry (paging.c), all NULL otherwise; the counters in `stats` add up over the
 * rounds of a call. */
extern "C" int tamd_k_trace(struct tamd_view view, long n, double * pos,
    const double * dir, int max_steps, int * index, double * length, int * n_steps,
    int flags, int * parked, double * cross_ds, struct tamd_paging pg, unsigned long long * stats,
    unsigned long long * queue)
{
        if (tamd_dev_init()) return 1;
        if ((pos == nullptr) || (dir == nullptr) || (index == nullptr) || (stats == nullptr) ||
            (queue == nullptr) ||
            ((pg.faulted != nullptr) &&
                ((pg.tentative == nullptr) || (length == nullptr) || (n_steps == nullptr) ||
                    (pg.n_faulted == nullptr) || (pg.wanted == nullptr) || (pg.wanted_first == nullptr)))) {
                /* (a kernel that writes through a null pointer can take the node down) */
                snprintf(g_error, sizeof(g_error), "tamd_k_trace: a required array is missing");
                return 1;
        }
        if (pg.ids == nullptr) HIP_TRY(hipMemsetAsync(stats, 0, 4 * sizeof(ull), g_stream));
        HIP_TRY(hipMemsetAsync(queue, 0, 5 * kQ * sizeof(ull), g_stream));
        if (n <= 0) return 0;
        const int carry = ((flags & TURTLE_AMD_TRACE_RESUME) ? TRACE_CARRY_MEDIUM : 0) |
            ((parked != nullptr) ? (flags & TAMD_TRACE_SORT_ROOM) : 0);
        if (view.mode == TAMD_MODE_ONE_MAP)
                return run_trace<TAMD_MODE_ONE_MAP>(view, n, pos, dir, max_steps, index, length,
                    n_steps, carry, parked, cross_ds, pg, stats, queue);
        if (view.mode == TAMD_MODE_ONE_STACK)
                return run_trace<TAMD_MODE_ONE_STACK>(view, n, pos, dir, max_steps, index, length,
                    n_steps, carry, parked, cross_ds, pg, stats, queue);
        return run_trace<TAMD_MODE_GENERIC>(view, n, pos, dir, max_steps, index, length, n_steps,
            carry, parked, cross_ds, pg, stats, queue);
}

/* n single steps with a direction, in two passes (see k_step); cross_ray /
 * cross_ds: scratch for n entries, or NULL to bisect in place */
extern "C" int tamd_k_step_dir(struct tamd_view view, long n, double * pos,
    const double * dir, double * lat, double * lon, double * alt, double * elev,
    double * step, int * index, int flags, int * cross_ray, double * cross_ds,
    struct tamd_paging pg, unsigned long long * stats, unsigned long long * queue)
{
        if (tamd_dev_init()) return 1;
        if (pg.ids == nullptr) HIP_TRY(hipMemsetAsync(stats, 0, 4 * sizeof(ull), g_stream));
        HIP_TRY(hipMemsetAsync(queue, 0, 3 * sizeof(ull), g_stream));
        if (n <= 0) return 0;
        const CrossList cross = { cross_ray, (cross_ray != nullptr) ? cross_ds : nullptr, queue + 2, nullptr };
        const StepWalk no_walk = { 0, 0, 0, 0, nullptr, nullptr };
        return run_step(view, n, pos, dir, lat, lon, alt, elev, step, index, flags, cross, pg,
            stats, no_walk);
}

/* One generation of a scattering walk: as tamd_k_step_dir with
 * TURTLE_AMD_STEP_RESUME, the directions drawn in the kernels from Philox(first +
 * ray, stream; seed) and the step added to length[] / steps[] */
extern "C" int tamd_k_step_walk(struct tamd_view view, long n, double * pos, double * alt,
    double * elev, int * index, unsigned long long seed, unsigned long long stream, long first,
    double * length, int * steps, int * cross_ray, double * cross_ds, struct tamd_paging pg,
    unsigned long long * stats, unsigned long long * queue)
{
        if (tamd_dev_init()) return 1;
        /* (stats add up over the generations of a walk: the caller zeroes them) */
        HIP_TRY(hipMemsetAsync(queue, 0, 3 * sizeof(ull), g_stream));
        if (n <= 0) return 0;
        const CrossList cross = { cross_ray, cross_ds, queue + 2, nullptr };
        const StepWalk walk = { 1, seed, stream, first, length, steps };
        return run_step(view, n, pos, nullptr, nullptr, nullptr, alt, elev, nullptr, index,
            TURTLE_AMD_STEP_RESUME, cross, pg, stats, walk);
}

/* A whole walk in one launch (k_walk): every tile resident, nothing listed.
 * stats are NOT zeroed (the caller does); queue[0] is. */
extern "C" int tamd_k_walk(struct tamd_view view, long n, double * pos, double * alt, double * elev,
    int * index, unsigned long long seed, long first, int first_step, int n_steps, double * length,
    int * steps, unsigned long long * stats, unsigned long long * queue)
{
        if (tamd_dev_init()) return 1;
        HIP_TRY(hipMemsetAsync(queue, 0, sizeof(ull), g_stream));
        if (n <= 0) return 0;
        const WalkIO io = { seed, first, first_step, n_steps };
        const bool strict = g_math_strict || !view.fast_ok;
#define WALK_CASE(MODE)                                                                        \
        do {                                                                                   \
                const void * kernel = strict ? (const void *)k_walk<MODE, false> :            \
                                               (const void *)k_walk<MODE, true>;               \
                long blocks = (long)g_cus * trace_blocks_per_cu(kernel);                       \
                const long useful = (n + 255) / 256;                                           \
                if (blocks > useful) blocks = useful;                                          \
                if (strict)                                                                    \
                        hipLaunchKernelGGL((k_walk<MODE, false>), dim3((unsigned)blocks), dim3(256), 0,   \
                            g_stream, view, n, pos, alt, elev, index, length, steps, io, stats, queue);  \
                else                                                                           \
                        hipLaunchKernelGGL((k_walk<MODE, true>), dim3((unsigned)blocks), dim3(256), 0,    \
                            g_stream, view, n, pos, alt, elev, index, length, steps, io, stats, queue);  \
        } while (0)
        if (view.mode == TAMD_MODE_ONE_MAP)
                WALK_CASE(TAMD_MODE_ONE_MAP);
        else if (view.mode == TAMD_MODE_ONE_STACK)
                WALK_CASE(TAMD_MODE_ONE_STACK);
        else
                WALK_CASE(TAMD_MODE_GENERIC);
#undef WALK_CASE
        LAUNCH_CHECK("k_walk");
        return 0;
}

extern "C" int tamd_k_philox(long n, unsigned long long seed, unsigned long long stream,
    long first, unsigned * out)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_philox, dim3(grid_for(n, 256)), dim3(256), 0, g_stream, n, seed,
            stream, first, out);
        LAUNCH_CHECK("k_philox");
        return 0;
}

extern "C" int tamd_k_isotropic(long n, unsigned long long seed, unsigned long long stream,
    long first, double * dir)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        hipLaunchKernelGGL(k_isotropic, dim3(grid_for(n, 256)), dim3(256), 0, g_stream, n,
            seed, stream, first, dir);
        LAUNCH_CHECK("k_isotropic");
        return 0;
}

extern "C" int tamd_k_tally(long n, const int * index, const double * length,
    int n_media, unsigned long long * hits, int n_bins, double length_max,
    unsigned long long * histogram)
{
        if (tamd_dev_init()) return 1;
        if (n <= 0) return 0;
        const size_t lds = (size_t)(n_media + 1 + n_bins + 1) * sizeof(unsigned int);
        if (lds > 64 * 1024) {
                snprintf(g_error, sizeof(g_error), "too many tally bins (%d)", n_bins);
                return 1;
        }
        const double scale = (double)n_bins / length_max;
        hipLaunchKernelGGL(k_tally, dim3(grid_for(n, 256)), dim3(256), lds, g_stream, n,
            index, length, n_media, hits, n_bins, scale, histogram);
        LAUNCH_CHECK("k_tally");
        return 0;
}
