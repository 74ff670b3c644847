"""Synthetic terrain and ray recipes for tests and bench (harness inputs).

No real DEM data exists in the build container or on the GPU box, so every
configuration in BASELINE.json runs on terrain regenerated from a formula
(SURVEY.md 8d).  Pure numpy; nothing here computes elevations, transforms or
steps -- that is the library's job.
"""
from __future__ import annotations

import os

import numpy as np

HGT_N = 3601  # SRTMGL1 nodes per tile edge (reference: src/turtle/io/hgt.c:98-104)


def c1_gradient_nodes(nx: int = 256, ny: int = 256) -> np.ndarray:
    """C1 map elevations z[iy, ix] = 200 + 1500*ix/(nx-1) (flat gradient along x)."""
    ix = np.arange(nx, dtype=np.float64)
    return np.broadcast_to(200.0 + 1500.0 * ix / (nx - 1), (ny, nx)).copy()


def srtm_like_nodes(lat0: int, lon0: int, n: int = HGT_N) -> np.ndarray:
    """Elevations of one 1x1 degree tile, as int16 [row south->north, col west->east].

    z = round(500 + 400 sin(0.01 J) cos(0.013 I)) with GLOBAL node indices
    J = (lon0)*(n-1)+j, I = (lat0)*(n-1)+i counted from (0N, 0E), so that
    adjacent tiles agree on their shared edge row/column like real SRTM tiles.
    """
    j = (lon0 * (n - 1) + np.arange(n, dtype=np.int64)).astype(np.float64)
    i = (lat0 * (n - 1) + np.arange(n, dtype=np.int64)).astype(np.float64)
    z = 500.0 + 400.0 * np.sin(0.01 * j)[None, :] * np.cos(0.013 * i)[:, None]
    return np.rint(z).astype(np.int16)


def hgt_name(lat0: int, lon0: int, n: int = HGT_N) -> str:
    ns = "N" if lat0 >= 0 else "S"
    ew = "E" if lon0 >= 0 else "W"
    # a bare name or an SRTMGL1 suffix means 3601 nodes, anything else 1201
    suffix = "" if n == HGT_N else ".SRTMGL3"
    return f"{ns}{abs(lat0):02d}{ew}{abs(lon0):03d}{suffix}.hgt"


def hgt_bytes(nodes_s2n: np.ndarray) -> bytes:
    """Serialise south->north int16 nodes as an .hgt payload (big-endian, north row first)."""
    return np.ascontiguousarray(nodes_s2n[::-1, :]).astype(">i2").tobytes()


def write_hgt(directory: str, lat0: int, lon0: int, n: int = HGT_N) -> str:
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, hgt_name(lat0, lon0, n))
    with open(path, "wb") as f:
        f.write(hgt_bytes(srtm_like_nodes(lat0, lon0, n)))
    return path


def uniform_rays(n: int, lat_range, lon_range, seed: int = 0x5EED2026,
                 margin: float = 0.1, el_range=(-10.0, -1.0), jump: int = 0):
    """The common ray recipe: origin (lat, lon) uniform inside the
    box shrunk by `margin` of its span, azimuth U[0,360), elevation U[el_range].
    Counter-based Philox stream: `jump` = r selects the r-th jumped stream, so
    rank r of a sharded run draws block r of one global ray array."""
    bits = np.random.Philox(seed)
    if jump:
        bits = bits.jumped(jump)
    rng = np.random.Generator(bits)
    u = rng.random((4, n))
    dlat = lat_range[1] - lat_range[0]
    dlon = lon_range[1] - lon_range[0]
    lat = lat_range[0] + dlat * (margin + (1 - 2 * margin) * u[0])
    lon = lon_range[0] + dlon * (margin + (1 - 2 * margin) * u[1])
    az = 360.0 * u[2]
    el = el_range[0] + (el_range[1] - el_range[0]) * u[3]
    return lat, lon, az, el
