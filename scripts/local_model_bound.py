#!/usr/bin/env python3
"""The truncation of the walk kernel's second-order local model (device.hip, LocalModel) against a
40-digit evaluation of the transform: the largest |error| / d^3 over latitudes to 80 degrees, heights
from -10 km to 100 km, any direction, d from 1 m to 2 km.  The kernel's bound (kModelC3) must hold it."""
import random

from mpmath import mp, mpf, sqrt, sin, cos, atan2, atan, tan, pi

mp.dps = 40
A = mpf(6378137)
E = mpf("0.081819190842622")
E2 = E * E


def from_geodetic(lat, lon, h):
    s, c = sin(lat), cos(lat)
    n = A / sqrt(1 - E2 * s * s)
    return ((n + h) * c * cos(lon), (n + h) * c * sin(lon), (n * (1 - E2) + h) * s)


def to_geodetic(x, y, z):
    lon = atan2(y, x)
    w = sqrt(x * x + y * y)
    lat = atan2(z, w * (1 - E2))
    for _ in range(60):
        s = sin(lat)
        n = A / sqrt(1 - E2 * s * s)
        h = w / cos(lat) - n
        new = atan2(z, w * (1 - E2 * n / (n + h)))
        if abs(new - lat) < mpf(10) ** -38:
            lat = new
            break
        lat = new
    s = sin(lat)
    n = A / sqrt(1 - E2 * s * s)
    h = w / cos(lat) - n
    return lat, lon, h


def model(lat0, lon0, h0):
    S, C, sl, cl = sin(lat0), cos(lat0), sin(lon0), cos(lon0)
    iw2 = 1 / (1 - E2 * S * S)
    rn = A * sqrt(iw2)
    rm = (1 - E2) * rn * iw2
    k = E2 * S * C * iw2
    rm1, rn1 = 3 * rm * k, rn * k
    re = rn + h0
    rho, nu = 1 / (rm + h0), 1 / (re * C)
    a3 = -S * nu * rho / 2
    a4 = -rm1 * rho ** 3 / 2
    b2 = nu * nu * (S * (1 + re * rho) - rn1 * rho * C) / 2
    b3 = -nu * nu * C
    c1, c2 = rho / 2, 1 / (2 * re)

    def at(dx, dy, dz):
        a = cl * dx + sl * dy
        Ee = cl * dy - sl * dx
        N = C * dz - S * a
        U = C * a + S * dz
        lat = lat0 + N * (rho - rho * rho * U + a4 * N) + a3 * Ee * Ee
        lon = lon0 + Ee * (nu + b2 * N + b3 * U)
        h = h0 + U + c1 * N * N + c2 * Ee * Ee
        return lat, lon, h
    return at, rm + h0, re * C


random.seed(5)
worst = {}
ratio = 0.0
for trial in range(4000):
    lat0 = mpf(random.uniform(-80, 80)) * pi / 180
    lon0 = mpf(random.uniform(-179, 179)) * pi / 180
    h0 = mpf(random.choice([random.uniform(-1e4, 1e4), random.uniform(-100, 3000), random.uniform(0, 1e5)]))
    o = from_geodetic(lat0, lon0, h0)
    at, m_lat, m_lon = model(lat0, lon0, h0)
    for d in (1, 10, 50, 100, 200, 500, 2000):
        u = [random.gauss(0, 1) for _ in range(3)]
        nrm = sum(v * v for v in u) ** 0.5
        dx, dy, dz = (mpf(d * v / nrm) for v in u)
        lat, lon, h = to_geodetic(o[0] + dx, o[1] + dy, o[2] + dz)
        ml, mo, mh = at(dx, dy, dz)
        err = max(abs(h - mh), abs(lat - ml) * m_lat, abs(lon - mo) * m_lon)
        key = (d, "|lat| > 60" if abs(lat0) > pi / 3 else "|lat| <= 60")
        worst[key] = max(worst.get(key, 0), float(err / mpf(d) ** 3))
        # the kernel's bound: kModelC3 (1.3 + tan^2 lat) d^3
        ratio = max(ratio, float(err / (mpf("2.5e-14") * (mpf("1.3") + tan(lat0) ** 2) * mpf(d) ** 3)))
for key in sorted(worst):
    print(f"d = {key[0]:5d} m, {key[1]:11s}: worst |error| / d^3 = {worst[key]:.3e} m^-2")
print(f"worst error / (2.5e-14 (1.3 + tan^2 lat) d^3) = {ratio:.3f} (must stay below 1)")
print(f"overall: {max(worst.values()):.3e}; 1 / R^2 = {1 / 6.371e6 ** 2:.3e}")
