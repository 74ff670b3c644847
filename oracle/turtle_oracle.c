/*
 * turtle_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see the header).
 *
 * CPU restatement of the reference stepper hot path.  Arithmetic follows the
 * reference expression by expression (same operand order, no FMA contraction:
 * build with -ffp-contract=off) so that, on the same libm, results are
 * bit-identical to the reference at local_range = 0 and at local_range > 0.
 * Citations are file:line under /root/reference.
 */
#include "turtle_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.14159265358979323846 /* ecef.c:30-33 */
#define DEG (ORC_PI / 180.)

/* WGS84 constants, ecef.c:36-38 */
static const double WGS84_A = 6378137;
static const double WGS84_B = 6356752.3142;
static const double WGS84_E = 0.081819190842622;

#define DOMAIN_ERROR 6 /* enum turtle_return, include/turtle.h:35-62 */

/* ---- ECEF ------------------------------------------------------------- */

/* ecef.c:41-55 */
void orc_ecef_from_geodetic(
    double latitude, double longitude, double elevation, double ecef[3])
{
        const double a = WGS84_A, e = WGS84_E;
        const double s = sin(latitude * ORC_PI / 180.);
        const double c = cos(latitude * ORC_PI / 180.);
        const double R = a / sqrt(1. - e * e * s * s);
        ecef[0] = (R + elevation) * c * cos(longitude * ORC_PI / 180.);
        ecef[1] = (R + elevation) * c * sin(longitude * ORC_PI / 180.);
        ecef[2] = (R * (1. - e * e) + elevation) * s;
}

/* ecef.c:63-130 (Olson 1996) */
void orc_ecef_to_geodetic(const double ecef[3], double * latitude,
    double * longitude, double * altitude)
{
        const double a = WGS84_A;
        const double e2 = WGS84_E * WGS84_E;
        const double a1 = a * e2;
        const double a2 = a1 * a1;
        const double a3 = 0.5 * a1 * e2;
        const double a4 = 2.5 * a2;
        const double a5 = a1 + a3;
        const double a6 = 1. - e2;

        /* poles, ecef.c:77-84 */
        if ((ecef[0] == 0.) && (ecef[1] == 0.)) {
                if (latitude != NULL) *latitude = (ecef[2] >= 0.) ? 90. : -90.;
                if (longitude != NULL) *longitude = 0.;
                if (altitude != NULL) *altitude = fabs(ecef[2]) - WGS84_B;
                return;
        }

        if (longitude != NULL)
                *longitude = atan2(ecef[1], ecef[0]) * 180. / ORC_PI;
        if ((latitude == NULL) && (altitude == NULL)) return;

        const double zp = fabs(ecef[2]);
        const double w2 = ecef[0] * ecef[0] + ecef[1] * ecef[1];
        const double w = sqrt(w2);
        const double z2 = ecef[2] * ecef[2];
        const double r2 = w2 + z2;
        const double r = sqrt(r2);
        const double s2 = z2 / r2;
        const double c2 = w2 / r2;

        double c, s, ss, la;
        const double u0 = a2 / r;
        const double v0 = a3 - a4 / r;
        if (c2 > 0.3) { /* ecef.c:101-107 */
                s = (zp / r) * (1. + c2 * (a1 + u0 + s2 * v0) / r);
                la = asin(s);
                ss = s * s;
                c = sqrt(1. - ss);
        } else { /* ecef.c:108-115 */
                c = (w / r) * (1. - s2 * (a5 - u0 - c2 * v0) / r);
                la = acos(c);
                ss = 1. - c * c;
                s = sqrt(ss);
        }

        /* ecef.c:117-129 */
        const double g = 1. - e2 * ss;
        const double rg = a / sqrt(g);
        const double rf = a6 * rg;
        const double u = w - rg * c;
        const double v = zp - rf * s;
        const double f = c * u + s * v;
        const double m = c * v - s * u;
        const double p = m / (rf / g + f);

        la += p;
        if (ecef[2] < 0.) la = -la;
        if (latitude != NULL) *latitude = la * 180. / ORC_PI;
        if (altitude != NULL) *altitude = f + 0.5 * m * p;
}

/* ecef.c:136-154 */
static void enu_basis(
    double latitude, double longitude, double e[3], double n[3], double u[3])
{
        const double lambda = longitude * ORC_PI / 180.;
        const double phi = latitude * ORC_PI / 180.;
        const double sl = sin(lambda), cl = cos(lambda);
        const double sp = sin(phi), cp = cos(phi);
        e[0] = -sl, e[1] = cl, e[2] = 0.;
        n[0] = -cl * sp, n[1] = -sl * sp, n[2] = cp;
        u[0] = cl * cp, u[1] = sl * cp, u[2] = sp;
}

/* ecef.c:160-176 */
void orc_ecef_from_horizontal(double latitude, double longitude,
    double azimuth, double elevation, double direction[3])
{
        double e[3], n[3], u[3];
        enu_basis(latitude, longitude, e, n, u);
        const double az = azimuth * ORC_PI / 180.;
        const double el = elevation * ORC_PI / 180.;
        const double ce = cos(el);
        const double r[3] = { ce * sin(az), ce * cos(az), sin(el) };
        int i;
        for (i = 0; i < 3; i++)
                direction[i] = r[0] * e[i] + r[1] * n[i] + r[2] * u[i];
}

/* ecef.c:178-207 */
void orc_ecef_to_horizontal(double latitude, double longitude,
    const double direction[3], double * azimuth, double * elevation)
{
        double e[3], n[3], u[3];
        enu_basis(latitude, longitude, e, n, u);
        const double x =
            e[0] * direction[0] + e[1] * direction[1] + e[2] * direction[2];
        const double y =
            n[0] * direction[0] + n[1] * direction[1] + n[2] * direction[2];
        const double z =
            u[0] * direction[0] + u[1] * direction[1] + u[2] * direction[2];
        double r = direction[0] * direction[0] + direction[1] * direction[1] +
            direction[2] * direction[2];
        if (r <= FLT_EPSILON) return; /* outputs untouched, ecef.c:194 */
        r = sqrt(r);
        if (azimuth != NULL) *azimuth = atan2(x, y) * 180. / ORC_PI;
        if (elevation != NULL) {
                const double arg = z / r;
                if (arg > 1.)
                        *elevation = 90.;
                else if (arg < -1.)
                        *elevation = -90.;
                else
                        *elevation = asin(arg) * 180. / ORC_PI;
        }
}

/* ---- projections -------------------------------------------------------- */

/* projection.c:238-244 */
static double lambert_latitude_to_iso(double latitude, double e)
{
        const double phi = latitude * ORC_PI / 180.;
        const double s = sin(phi);
        return log(tan(0.25 * ORC_PI + 0.5 * phi) * pow((1. - e * s) / (1. + e * s), 0.5 * e));
}

/* projection.c:253-268 */
static double lambert_iso_to_latitude(double L, double e)
{
        const double epsilon = FLT_EPSILON;
        const double eL = exp(L);
        double phi0 = 2. * atan(eL) - 0.5 * ORC_PI;
        for (;;) {
                const double s = sin(phi0);
                double phi1 =
                    2. * atan(pow((1. + e * s) / (1. - e * s), 0.5 * e) * eL) - 0.5 * ORC_PI;
                if (fabs(phi1 - phi0) <= epsilon) return phi1 / ORC_PI * 180.;
                phi0 = phi1;
        }
}

/* projection.c:329-349: e, n, c, lambda_c, xs, ys */
static const double LAMBERT[6][6] = {
        { 0.08248325676, 0.7604059656, 11603796.98, 0.04079234433, 600000.0, 5657616.674 },
        { 0.08248325676, 0.7289686274, 11745793.39, 0.04079234433, 600000.0, 6199695.768 },
        { 0.08248325676, 0.7289686274, 11745793.39, 0.04079234433, 600000.0, 8199695.768 },
        { 0.08248325676, 0.6959127966, 11947992.52, 0.04079234433, 600000.0, 6791905.085 },
        { 0.08248325676, 0.6712679322, 12136281.99, 0.04079234433, 234.358, 7239161.542 },
        { 0.08181919112, 0.7253743710, 11755528.70, 0.05235987756, 700000.0, 12657560.145 }
};

void orc_project(const struct orc_proj * proj, double latitude, double longitude,
    double * x, double * y)
{
        if (proj->type == ORC_PROJ_LAMBERT) { /* projection.c:286-295 */
                const double * P = LAMBERT[proj->lambert_tag];
                const double L = lambert_latitude_to_iso(latitude, P[0]);
                const double cenL = P[2] * exp(-P[1] * L);
                const double lambda = longitude / 180. * ORC_PI;
                const double theta = P[1] * (lambda - P[3]);
                *x = P[4] + cenL * sin(theta);
                *y = P[5] - cenL * cos(theta);
                return;
        }
        /* projection.c:377-408 */
        const double a = 6378.137E+03;
        const double f = 1. / 298.257223563;
        const double E0 = 5E+05;
        const double N0 = (proj->hemisphere > 0) ? 0. : 1E+07;
        const double k0 = 0.9996;
        const double n = f / (2. - f);
        const double A = a / (1. + n) * (1. + n * n * (0.25 + 0.0625 * n * n));
        const double alpha[3] = { n * (0.5 + n * (-2. / 3. + 5. / 16. * n)),
                n * n * (13. / 48. - 3. / 5. * n), 61. / 240. * n * n * n };
        const double c = 2. * sqrt(n) / (1. + n);
        const double s = sin(latitude * ORC_PI / 180.);
        const double t = sinh(atanh(s) - c * atanh(c * s));
        const double dl = (longitude - proj->longitude_0) * ORC_PI / 180.;
        const double zeta = atan2(t, cos(dl));
        const double eta = atanh(sin(dl) / sqrt(1. + t * t));
        double xs = 0., ys = 0.;
        int i;
        for (i = 0; i < 3; i++) {
                xs += alpha[i] * cos(2. * (i + 1) * zeta) * sinh(2. * (i + 1) * eta);
                ys += alpha[i] * sin(2. * (i + 1) * zeta) * cosh(2. * (i + 1) * eta);
        }
        *x = E0 + k0 * A * (eta + xs);
        *y = N0 + k0 * A * (zeta + ys);
}

void orc_unproject(const struct orc_proj * proj, double x, double y,
    double * latitude, double * longitude)
{
        if (proj->type == ORC_PROJ_LAMBERT) { /* projection.c:304-318 */
                const double * P = LAMBERT[proj->lambert_tag];
                const double dx = x - P[4];
                const double dy = y - P[5];
                const double R = sqrt(dx * dx + dy * dy);
                const double gamma = atan2(dx, -dy);
                *longitude = (P[3] + gamma / P[1]) * 180. / ORC_PI;
                const double L = -log(R / P[2]) / P[1];
                *latitude = lambert_iso_to_latitude(L, P[0]);
                return;
        }
        /* projection.c:417-448 */
        const double a = 6378.137E+03;
        const double f = 1. / 298.257223563;
        const double E0 = 5E+05;
        const double N0 = (proj->hemisphere > 0) ? 0. : 1E+07;
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
        int i;
        for (i = 0; i < 3; i++) {
                zeta -= beta[i] * sin(2. * (i + 1) * zeta0) * cosh(2. * (i + 1) * eta0);
                eta -= beta[i] * cos(2. * (i + 1) * zeta0) * sinh(2. * (i + 1) * eta0);
        }
        const double chi = asin(sin(zeta) / cosh(eta));
        double s = 0.;
        for (i = 0; i < 3; i++) s += delta[i] * sin(2. * (i + 1) * chi);
        *latitude = (chi + s) * 180. / ORC_PI;
        *longitude = proj->longitude_0 + atan2(sinh(eta), cos(zeta)) * 180. / ORC_PI;
}

void orc_project_n(const struct orc_proj * proj, long n, const double * latitude,
    const double * longitude, double * x, double * y)
{
        long r;
        for (r = 0; r < n; r++) orc_project(proj, latitude[r], longitude[r], x + r, y + r);
}

void orc_unproject_n(const struct orc_proj * proj, long n, const double * x,
    const double * y, double * latitude, double * longitude)
{
        long r;
        for (r = 0; r < n; r++) orc_unproject(proj, x[r], y[r], latitude + r, longitude + r);
}

/* ---- single grid ------------------------------------------------------ */


/* map.c:41-44 and hgt.c:127-131 */
double orc_grid_node(const struct orc_grid * g, int ix, int iy)
{
        if (g->layout == ORC_LAYOUT_HGT) {
                const uint16_t raw = g->data[(g->ny - 1 - iy) * g->nx + ix];
                const uint16_t host = (uint16_t)((raw >> 8) | (raw << 8));
                return (int16_t)host; /* ntohs on a little-endian host */
        }
        return g->z0 + g->data[iy * g->nx + ix] * g->dz;
}

/* map.c:229-277 */
int orc_grid_elevation(const struct orc_grid * g, double x, double y, double * z)
{
        if (isnan(x) || isnan(y)) return 0; /* map.c:233-240 */

        double hx = (x - g->x0) / g->dx;
        double hy = (y - g->y0) / g->dy;
        /* map.c:247-255: inclusive upper edge.  The reference converts to int
         * before this test; do the test first so the cast is always defined
         * (same result for every in-range value). */
        if ((hx > g->nx - 1) || (hx < 0) || (hy > g->ny - 1) || (hy < 0))
                return 0;
        int ix = (int)hx;
        int iy = (int)hy;
        if (ix == g->nx - 1) { /* map.c:256-265 */
                ix--;
                hx = 1.;
        } else
                hx -= ix;
        if (iy == g->ny - 1) {
                iy--;
                hy = 1.;
        } else
                hy -= iy;

        const double z00 = orc_grid_node(g, ix, iy);
        const double z10 = orc_grid_node(g, ix + 1, iy);
        const double z01 = orc_grid_node(g, ix, iy + 1);
        const double z11 = orc_grid_node(g, ix + 1, iy + 1);
        *z = z00 * (1. - hx) * (1. - hy) + z01 * (1. - hx) * hy +
            z10 * hx * (1. - hy) + z11 * hx * hy; /* map.c:272-273 */
        return 1;
}


/* map.c:280-378.  Mirrors the reference as it is, including its slip at
 * map.c:352-353: for a point in the first half-row of the grid (iy == 0,
 * hy <= 0.5) the y-gradient is stored into *gx and *gy is left untouched. */
int orc_grid_gradient(
    const struct orc_grid * g, double x, double y, double * gx, double * gy)
{
        if (isnan(x) || isnan(y)) return 0;
        double hx = (x - g->x0) / g->dx;
        double hy = (y - g->y0) / g->dy;
        if ((hx > g->nx - 1) || (hx < 0) || (hy > g->ny - 1) || (hy < 0)) return 0;
        int ix = (int)hx;
        int iy = (int)hy;
        if (ix == g->nx - 1) {
                ix--;
                hx = 1.;
        } else
                hx -= ix;
        if (iy == g->ny - 1) {
                iy--;
                hy = 1.;
        } else
                hy -= iy;
        const double z00 = orc_grid_node(g, ix, iy);
        const double z10 = orc_grid_node(g, ix + 1, iy);
        const double z01 = orc_grid_node(g, ix, iy + 1);
        const double z11 = orc_grid_node(g, ix + 1, iy + 1);

        if (hx <= 0.5) { /* map.c:324-335 */
                const double gx1 = (z10 - z00) * (1. - hy) + (z11 - z01) * hy;
                if (ix == 0) {
                        *gx = gx1 / g->dx;
                } else {
                        const double z_10 = orc_grid_node(g, ix - 1, iy);
                        const double z_11 = orc_grid_node(g, ix - 1, iy + 1);
                        const double gx0 = (z00 - z_10) * (1. - hy) + (z01 - z_11) * hy;
                        const double ax = hx + 0.5;
                        *gx = (gx0 * (1. - ax) + gx1 * ax) / g->dx;
                }
        } else { /* map.c:336-348 */
                const double gx0 = (z10 - z00) * (1. - hy) + (z11 - z01) * hy;
                if (ix == g->nx - 2) {
                        *gx = gx0 / g->dx;
                } else {
                        const double z20 = orc_grid_node(g, ix + 2, iy);
                        const double z21 = orc_grid_node(g, ix + 2, iy + 1);
                        const double gx1 = (z20 - z10) * (1. - hy) + (z21 - z11) * hy;
                        const double ax = hx - 0.5;
                        *gx = (gx0 * (1. - ax) + gx1 * ax) / g->dx;
                }
        }

        if (hy <= 0.5) { /* map.c:350-361 */
                const double gy1 = (z01 - z00) * (1. - hx) + (z11 - z10) * hx;
                if (iy == 0) {
                        *gx = gy1 / g->dy; /* sic, map.c:353 */
                } else {
                        const double z0_1 = orc_grid_node(g, ix, iy - 1);
                        const double z1_1 = orc_grid_node(g, ix + 1, iy - 1);
                        const double gy0 = (z00 - z0_1) * (1. - hx) + (z10 - z1_1) * hx;
                        const double ay = hy + 0.5;
                        *gy = (gy0 * (1. - ay) + gy1 * ay) / g->dy;
                }
        } else { /* map.c:362-374 */
                const double gy0 = (z01 - z00) * (1. - hx) + (z11 - z10) * hx;
                if (iy == g->ny - 2) {
                        *gy = gy0 / g->dy;
                } else {
                        const double z02 = orc_grid_node(g, ix, iy + 2);
                        const double z12 = orc_grid_node(g, ix + 1, iy + 2);
                        const double gy1 = (z02 - z01) * (1. - hx) + (z12 - z11) * hx;
                        const double ay = hy - 0.5;
                        *gy = (gy0 * (1. - ay) + gy1 * ay) / g->dy;
                }
        }
        return 1;
}

void orc_grid_gradient_n(const struct orc_grid * grid, long n, const double * x,
    const double * y, double * gx, double * gy, int * inside)
{
        long r;
        for (r = 0; r < n; r++)
                inside[r] = orc_grid_gradient(grid, x[r], y[r], gx + r, gy + r);
}

/* ---- tile directory, all tiles resident -------------------------------- */

/* half-open membership test used for resident tiles, stack.c:307-311,:320-321 */
static int tile_holds(const struct orc_grid * g, double latitude, double longitude)
{
        const double hx = (longitude - g->x0) / g->dx;
        const double hy = (latitude - g->y0) / g->dy;
        return (hx >= 0.) && (hx < g->nx - 1) && (hy >= 0.) && (hy < g->ny - 1);
}

/* stack.c:338-361.  With every tile resident the MRU order cannot change the
 * answer: the list scan (stack.c:300-335) finds the one tile whose half-open
 * box holds the point; failing that, turtle_stack_load_ (:399-450) selects the
 * tile by the directory formula and the bilinear lookup applies its inclusive
 * edge test to it. */
int orc_stack_elevation(const struct orc_geometry * geometry,
    const struct orc_stack * st, double latitude, double longitude, double * z)
{
        const int n = st->nlat * st->nlon;
        int i;
        *z = 0.;
        for (i = 0; i < n; i++) {
                if (st->tile[i] < 0) continue;
                const struct orc_grid * g = &geometry->grids[st->tile[i]];
                if (tile_holds(g, latitude, longitude))
                        return orc_grid_elevation(g, longitude, latitude, z);
        }
        /* stack.c:413-424 */
        if ((longitude < st->lon0) || (latitude < st->lat0)) return 0;
        const double fx = (longitude - st->lon0) / st->dlon;
        if (!(fx < st->nlon)) return 0; /* (int)fx >= nlon, NaN-safe */
        const int ix = (int)fx;
        const double fy = (latitude - st->lat0) / st->dlat;
        if (!(fy < st->nlat)) return 0;
        const int iy = (int)fy;
        const int t = st->tile[iy * st->nlon + ix];
        if (t < 0) return 0;
        const int inside =
            orc_grid_elevation(&geometry->grids[t], longitude, latitude, z);
        if (!inside) *z = 0.;
        return inside;
}

/* ---- stepper ---------------------------------------------------------- */

/* stepper.c:547-570 */
void orc_stepper_init(struct orc_stepper * s, const struct orc_geometry * geometry)
{
        memset(s, 0, sizeof(*s));
        s->geometry = geometry;
        s->local_range = 1.;
        s->slope_factor = 0.4;
        s->resolution_factor = 1E-02;
        s->last.index[0] = s->last.index[1] = -1;
        orc_stepper_reset(s);
}

/* stepper.c:602-615 */
void orc_stepper_reset(struct orc_stepper * s)
{
        int i;
        for (i = 0; i < 3; i++) {
                s->last.position[i] = DBL_MAX;
                s->reference_ecef[i] = DBL_MAX;
        }
}

/* stepper.c:37-51 */
static void exact_geodetic(
    struct orc_stepper * s, const double * position, double * geo)
{
        s->n_transforms++;
        orc_ecef_to_geodetic(position, geo, geo + 1, geo + 2);
        const int gi = s->geometry->geoid;
        if (gi >= 0) {
                double undulation;
                const double lo = (geo[1] >= 0) ? geo[1] : geo[1] + 360.;
                if (orc_grid_elevation(
                        &s->geometry->grids[gi], lo, geo[0], &undulation))
                        geo[2] -= undulation;
        }
}

/* stepper.c:85-171, for the one "geodetic" transform (n0 = 0, n1 = 3) */
static void get_geographic(
    struct orc_stepper * s, const double * position, double * geo)
{
        if (s->local_range <= 0.) { /* stepper.c:97-106 */
                exact_geodetic(s, position, geo);
                return;
        }

        double local[3], range = 0.;
        int i, j;
        for (i = 0; i < 3; i++) { /* stepper.c:109-116 */
                double r = position[i] - s->reference_ecef[i];
                local[i] = r;
                r = fabs(r);
                if (r > range) range = r;
        }
        if (range < s->local_range) { /* stepper.c:118-128 */
                for (i = 0; i < 3; i++) {
                        geo[i] = s->reference_geographic[i];
                        for (j = 0; j < 3; j++)
                                geo[i] += s->jacobian[i][j] * local[j];
                }
                return;
        }

        exact_geodetic(s, position, geo); /* stepper.c:131-133 */

        double step = 0.; /* stepper.c:138-142 */
        for (i = 0; i < 3; i++) {
                const double d = fabs(position[i] - s->last.position[i]);
                if (d > step) step = d;
        }
        if (step < 0.33 * s->local_range) { /* stepper.c:144-162 */
                memcpy(s->reference_ecef, position, sizeof(s->reference_ecef));
                memcpy(s->reference_geographic, geo,
                    sizeof(s->reference_geographic));
                for (i = 0; i < 3; i++) {
                        double r[3] = { position[0], position[1], position[2] };
                        r[i] += 10.;
                        double geo1[3];
                        exact_geodetic(s, r, geo1);
                        for (j = 0; j < 3; j++)
                                s->jacobian[j][i] = 0.1 * (geo1[j] - geo[j]);
                }
        }
}

/* Elevation of one data source at (lat, lon): stepper.c:199-264 (step form)
 * and :282-324 (elevation form) reduce to the same three lookups. */
static int source_elevation(const struct orc_geometry * geometry,
    const struct orc_meta * m, double latitude, double longitude, double * z)
{
        switch (m->kind) {
        case ORC_FLAT: /* stepper.c:252-264 */
                *z = 0.;
                return 1;
        case ORC_MAP: { /* stepper.c:240-248, :304-315 */
                const struct orc_grid * g = &geometry->grids[m->src];
                if (g->proj.type >= 0) {
                        double x, y;
                        orc_project(&g->proj, latitude, longitude, &x, &y);
                        return orc_grid_elevation(g, x, y, z);
                }
                /* geodetic grid: x = longitude, y = latitude */
                return orc_grid_elevation(g, longitude, latitude, z);
        }
        default: /* stepper.c:223-224 */
                return orc_stack_elevation(
                    geometry, &geometry->stacks[m->src], latitude, longitude, z);
        }
}

/* stepper.c:703-756 (without the error plumbing).  Returns 1 when the sample
 * was computed afresh, 0 when the cached `last` sample was reused; the
 * reference raises its "no valid data" error only on the fresh path
 * (stepper.c:751-754 sits outside the cached branch, :745-748). */
static int stepper_sample(struct orc_stepper * s, const double * position,
    struct orc_sample * sample)
{
        const struct orc_geometry * G = s->geometry;
        if ((position[0] == s->last.position[0]) &&
            (position[1] == s->last.position[1]) &&
            (position[2] == s->last.position[2])) { /* stepper.c:745-748 */
                if (sample != &s->last) memcpy(sample, &s->last, sizeof(*sample));
                return 0;
        }

        s->n_samples++;
        sample->index[0] = sample->index[1] = -1; /* stepper.c:713-716 */
        sample->elevation[0] = -DBL_MAX;
        sample->elevation[1] = DBL_MAX;
        int has_geodetic = 0, layer;
        for (layer = 0; layer < G->n_layers; layer++) {
                int k, data_index = 0;
                for (k = G->layer_first[layer]; k < G->layer_first[layer + 1];
                     k++, data_index++) {
                        const struct orc_meta * m = &G->metas[k];
                        if (!has_geodetic) {
                                get_geographic(s, position, sample->geographic);
                                if (sample == &s->last) /* stepper.c:730-733 */
                                        memcpy(s->last.position, position,
                                            sizeof(s->last.position));
                                has_geodetic = 1;
                        }
                        double elevation;
                        const int inside = source_elevation(G, m,
                            sample->geographic[0], sample->geographic[1],
                            &elevation);
                        if (!inside) continue;
                        elevation += m->offset; /* stepper.c:737 */
                        if (elevation >= sample->geographic[2]) {
                                /* check_layer, stepper.c:690-694 */
                                sample->index[0] = layer;
                                sample->index[1] = data_index;
                                sample->elevation[1] = elevation;
                                return 1;
                        }
                        sample->index[0] = layer + 1; /* stepper.c:695-699 */
                        sample->index[1] = data_index;
                        sample->elevation[0] = elevation;
                        break;
                }
        }
        return 1;
}

/* stepper.c:758-778 */
static void publish(const struct orc_stepper * s, double * latitude,
    double * longitude, double * altitude, double * elevation, int * index)
{
        if (latitude != NULL) *latitude = s->last.geographic[0];
        if (longitude != NULL) *longitude = s->last.geographic[1];
        if (altitude != NULL) *altitude = s->last.geographic[2];
        if (elevation != NULL) {
                const int ok = s->last.index[0] >= 0;
                elevation[0] = ok ? s->last.elevation[0] : 0.;
                elevation[1] = ok ? s->last.elevation[1] : 0.;
        }
        if (index != NULL) {
                index[0] = s->last.index[0];
                index[1] = s->last.index[1];
        }
}

/* stepper.c:780-875 */
int orc_stepper_step(struct orc_stepper * s, double * position,
    const double * direction, double * latitude, double * longitude,
    double * altitude, double * elevation, double * step_length, int * index)
{
        const int fresh = stepper_sample(s, position, &s->last);
        if (fresh && (s->last.index[0] < 0) && (index == NULL))
                return DOMAIN_ERROR; /* stepper.c:751-754, :788-790 */
        if (s->last.index[0] < 0) { /* stepper.c:791-796 */
                publish(s, latitude, longitude, altitude, elevation, index);
                if (step_length != NULL) *step_length = 0;
                return 0;
        }

        double ds = 0.; /* stepper.c:799-813 */
        int i;
        for (i = 0; i < 2; i++) {
                if ((s->last.index[0] == 0) && (i == 0))
                        continue;
                else if ((s->last.index[0] == s->geometry->n_layers) && (i == 1))
                        break;
                const double dsi =
                    fabs(s->last.geographic[2] - s->last.elevation[i]);
                if ((dsi < ds) || (ds <= 0.)) ds = dsi;
        }
        ds *= s->slope_factor;
        if (ds < s->resolution_factor) ds = s->resolution_factor;

        if (direction == NULL) { /* stepper.c:816-821 */
                publish(s, latitude, longitude, altitude, elevation, index);
                if (step_length != NULL) *step_length = ds;
                return 0;
        }

        for (i = 0; i < 3; i++) position[i] += direction[i] * ds;

        const int medium0 = s->last.index[0];
        stepper_sample(s, position, &s->last);
        int medium1 = s->last.index[0];

        if (medium0 != medium1) { /* stepper.c:832-864 */
                double ds0 = -ds, ds1 = 0.;
                struct orc_sample sample2;
                memcpy(&sample2, &s->last, sizeof(sample2));
                while (ds1 - ds0 > 1E-08) {
                        const double ds2 = 0.5 * (ds0 + ds1);
                        double position2[3] = { position[0] + direction[0] * ds2,
                                position[1] + direction[1] * ds2,
                                position[2] + direction[2] * ds2 };
                        stepper_sample(s, position2, &sample2);
                        const int medium2 = sample2.index[0];
                        if (medium2 == medium0) {
                                ds0 = ds2;
                        } else {
                                medium1 = medium2;
                                ds1 = ds2;
                                memcpy(sample2.position, position2,
                                    sizeof(sample2.position));
                                memcpy(&s->last, &sample2, sizeof(s->last));
                        }
                }
                ds += ds1;
                for (i = 0; i < 3; i++) position[i] += direction[i] * ds1;
        }
        (void)medium1;

        publish(s, latitude, longitude, altitude, elevation, index);
        if (step_length != NULL) *step_length = ds;
        if ((s->last.index[0] < 0) && (index == NULL))
                return DOMAIN_ERROR; /* stepper.c:870-873 */
        return 0;
}

/* stepper.c:877-931 */
int orc_stepper_position(struct orc_stepper * s, double latitude,
    double longitude, double height, int layer_index, double * position,
    int * data_index)
{
        const struct orc_geometry * G = s->geometry;
        if ((layer_index < 0) || (layer_index >= G->n_layers))
                return DOMAIN_ERROR;

        int k, index = 0;
        for (k = G->layer_first[layer_index]; k < G->layer_first[layer_index + 1];
             k++, index++) {
                const struct orc_meta * m = &G->metas[k];
                double elevation = 0.;
                if (!source_elevation(G, m, latitude, longitude, &elevation))
                        continue;
                elevation += m->offset;
                if (G->geoid >= 0) { /* stepper.c:905-914 */
                        double undulation;
                        const double lo =
                            (longitude >= 0) ? longitude : longitude + 360.;
                        if (orc_grid_elevation(&G->grids[G->geoid], lo, latitude,
                                &undulation))
                                elevation += undulation;
                }
                orc_ecef_from_geodetic(
                    latitude, longitude, elevation + height, position);
                if (data_index != NULL) *data_index = index;
                return 0;
        }
        if (data_index != NULL) {
                *data_index = -1;
                return 0;
        }
        return DOMAIN_ERROR;
}

/* ---- batch drivers ---------------------------------------------------- */

struct trace_job {
        const struct orc_geometry * geometry;
        double slope, resolution, range;
        long begin, end;
        double * position;
        const double * direction;
        int max_steps;
        int * index;
        double * length;
        int * n_steps;
        long steps, samples, transforms;
};

static void * trace_worker(void * arg)
{
        struct trace_job * job = arg;
        struct orc_stepper s;
        orc_stepper_init(&s, job->geometry);
        s.slope_factor = job->slope;
        s.resolution_factor = job->resolution;
        s.local_range = job->range;
        long r;
        for (r = job->begin; r < job->end; r++) {
                double * pos = job->position + 3 * r;
                const double * dir = job->direction + 3 * r;
                int idx[2];
                double total = 0.;
                int n = 0;
                orc_stepper_reset(&s);
                orc_stepper_step(&s, pos, NULL, NULL, NULL, NULL, NULL, NULL, idx);
                const int medium = idx[0];
                if (medium >= 0) {
                        while (n < job->max_steps) {
                                double ds;
                                orc_stepper_step(&s, pos, dir, NULL, NULL, NULL,
                                    NULL, &ds, idx);
                                total += ds;
                                n++;
                                if (idx[0] != medium) break;
                        }
                }
                if (job->index != NULL) {
                        job->index[2 * r] = idx[0];
                        job->index[2 * r + 1] = idx[1];
                }
                if (job->length != NULL) job->length[r] = total;
                if (job->n_steps != NULL) job->n_steps[r] = n;
                job->steps += n;
        }
        job->samples = s.n_samples;
        job->transforms = s.n_transforms;
        return NULL;
}

long orc_trace_n(const struct orc_geometry * geometry, double slope,
    double resolution, double range, long n, double * position,
    const double * direction, int max_steps, int * index, double * length,
    int * n_steps, int threads, long * n_samples, long * n_transforms)
{
        if (threads < 1) threads = 1;
        if (threads > 1024) threads = 1024;
        struct trace_job * jobs = calloc(threads, sizeof(*jobs));
        pthread_t * tid = calloc(threads, sizeof(*tid));
        int t;
        for (t = 0; t < threads; t++) {
                struct trace_job * j = &jobs[t];
                j->geometry = geometry;
                j->slope = slope, j->resolution = resolution, j->range = range;
                j->begin = n * t / threads;
                j->end = n * (t + 1) / threads;
                j->position = position, j->direction = direction;
                j->max_steps = max_steps;
                j->index = index, j->length = length, j->n_steps = n_steps;
        }
        if (threads == 1)
                trace_worker(&jobs[0]);
        else {
                for (t = 0; t < threads; t++)
                        pthread_create(&tid[t], NULL, trace_worker, &jobs[t]);
                for (t = 0; t < threads; t++) pthread_join(tid[t], NULL);
        }
        long steps = 0, samples = 0, transforms = 0;
        for (t = 0; t < threads; t++) {
                steps += jobs[t].steps;
                samples += jobs[t].samples;
                transforms += jobs[t].transforms;
        }
        if (n_samples != NULL) *n_samples = samples;
        if (n_transforms != NULL) *n_transforms = transforms;
        free(jobs);
        free(tid);
        return steps;
}

void orc_step_n(const struct orc_geometry * geometry, double slope,
    double resolution, long n, double * position, const double * direction,
    double * latitude, double * longitude, double * altitude,
    double * elevation, double * step_length, int * index)
{
        struct orc_stepper s;
        orc_stepper_init(&s, geometry);
        s.slope_factor = slope;
        s.resolution_factor = resolution;
        s.local_range = 0.;
        long r;
        for (r = 0; r < n; r++) {
                int idx[2];
                orc_stepper_reset(&s);
                orc_stepper_step(&s, position + 3 * r,
                    direction ? direction + 3 * r : NULL,
                    latitude ? latitude + r : NULL,
                    longitude ? longitude + r : NULL,
                    altitude ? altitude + r : NULL,
                    elevation ? elevation + 2 * r : NULL,
                    step_length ? step_length + r : NULL, idx);
                if (index != NULL) {
                        index[2 * r] = idx[0];
                        index[2 * r + 1] = idx[1];
                }
        }
}

void orc_position_n(const struct orc_geometry * geometry, long n,
    const double * latitude, const double * longitude, const double * height,
    int layer_index, double * position, int * data_index)
{
        struct orc_stepper s;
        orc_stepper_init(&s, geometry);
        long r;
        for (r = 0; r < n; r++) {
                int di;
                orc_stepper_position(&s, latitude[r], longitude[r], height[r],
                    layer_index, position + 3 * r, &di);
                if (data_index != NULL) data_index[r] = di;
        }
}

void orc_ecef_to_geodetic_n(long n, const double * ecef, double * latitude,
    double * longitude, double * altitude)
{
        long r;
        for (r = 0; r < n; r++)
                orc_ecef_to_geodetic(
                    ecef + 3 * r, latitude + r, longitude + r, altitude + r);
}

void orc_ecef_from_geodetic_n(long n, const double * latitude,
    const double * longitude, const double * elevation, double * ecef)
{
        long r;
        for (r = 0; r < n; r++)
                orc_ecef_from_geodetic(
                    latitude[r], longitude[r], elevation[r], ecef + 3 * r);
}

void orc_ecef_from_horizontal_n(long n, const double * latitude,
    const double * longitude, const double * azimuth, const double * elevation,
    double * direction)
{
        long r;
        for (r = 0; r < n; r++)
                orc_ecef_from_horizontal(latitude[r], longitude[r], azimuth[r],
                    elevation[r], direction + 3 * r);
}

void orc_ecef_to_horizontal_n(long n, const double * latitude,
    const double * longitude, const double * direction, double * azimuth,
    double * elevation)
{
        long r;
        for (r = 0; r < n; r++)
                orc_ecef_to_horizontal(latitude[r], longitude[r],
                    direction + 3 * r, azimuth + r, elevation + r);
}

void orc_grid_elevation_n(const struct orc_grid * grid, long n,
    const double * x, const double * y, double * z, int * inside)
{
        long r;
        for (r = 0; r < n; r++) {
                z[r] = 0.;
                inside[r] = orc_grid_elevation(grid, x[r], y[r], z + r);
        }
}

void orc_stack_elevation_n(const struct orc_geometry * geometry, int stack,
    long n, const double * latitude, const double * longitude, double * z,
    int * inside)
{
        long r;
        for (r = 0; r < n; r++)
                inside[r] = orc_stack_elevation(geometry,
                    &geometry->stacks[stack], latitude[r], longitude[r], z + r);
}
