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


def geotiff_name(lat0: int, lon0: int) -> str:
    """ASTER-GDEM2's tile name (the terrain BASELINE's C5 names)"""
    ns = "N" if lat0 >= 0 else "S"
    ew = "E" if lon0 >= 0 else "W"
    return f"ASTGTM2_{ns}{abs(lat0):02d}{ew}{abs(lon0):03d}_dem.tif"


def geotiff_bytes(nodes_s2n: np.ndarray, lon0: float, lat1: float, dx: float, dy: float) -> bytes:
    """The grid as an uncompressed single-strip GeoTIFF of int16 samples, little-endian, scan
    lines north->south, with the two GeoTIFF tags the reference reads: ModelPixelScale (33550)
    and ModelTiepoint (33922) [ref src/turtle/io/geotiff16.c:205-214]."""
    import struct
    ny, nx = nodes_s2n.shape
    data = np.ascontiguousarray(nodes_s2n[::-1, :]).astype("<i2").tobytes()
    scale = struct.pack("<3d", dx, dy, 0.0)
    tie = struct.pack("<6d", 0.0, 0.0, 0.0, lon0, lat1, 0.0)
    n_tags = 12
    ifd_at = 8
    extra_at = ifd_at + 2 + 12 * n_tags + 4
    scale_at, tie_at = extra_at, extra_at + len(scale)
    data_at = tie_at + len(tie)
    def tag(code, kind, count, value):
        return struct.pack("<HHII", code, kind, count, value)
    tags = [tag(256, 4, 1, nx), tag(257, 4, 1, ny), tag(258, 3, 1, 16), tag(259, 3, 1, 1),
            tag(262, 3, 1, 1), tag(273, 4, 1, data_at), tag(277, 3, 1, 1), tag(278, 4, 1, ny),
            tag(279, 4, 1, len(data)), tag(339, 3, 1, 2), tag(33550, 12, 3, scale_at),
            tag(33922, 12, 6, tie_at)]
    return (b"II" + struct.pack("<HI", 42, ifd_at) + struct.pack("<H", n_tags) + b"".join(tags) +
            struct.pack("<I", 0) + scale + tie + data)


def write_geotiff(directory: str, lat0: int, lon0: int, n: int = HGT_N) -> str:
    """One 1x1 degree tile of the synthetic terrain as ASTER-GDEM2 ships it: GeoTIFF, int16"""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, geotiff_name(lat0, lon0))
    step = 1.0 / (n - 1)
    with open(path, "wb") as f:
        f.write(geotiff_bytes(srtm_like_nodes(lat0, lon0, n), float(lon0), float(lat0 + 1), step, step))
    return path
