"""Terrain/geometry recipes shared by the oracle tests and the GPU parity tests.

Each recipe returns the pieces both sides need: numpy node arrays (south->north)
plus the oracle-side OracleGeometry.  The product side builds its own objects
from the same node arrays through the C-ABI (see tests/amd_build.py).
"""
from __future__ import annotations

import hashlib

import numpy as np

from oracle import ffi as O
from turtle_amd import synth

C1_X, C1_Y, C1_Z = (3.0, 4.0), (45.0, 46.0), (0.0, 2000.0)


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def c1_nodes():
    return synth.c1_gradient_nodes()


def c1_oracle(layers=None, geoid_nodes=None):
    grids = [O.default_grid(c1_nodes(), C1_X, C1_Y, C1_Z)]
    geoid = -1
    if geoid_nodes is not None:
        grids.append(O.default_grid(geoid_nodes, (0.0, 360.0), (-90.0, 90.0), (-40.0, 40.0)))
        geoid = 1
    if layers is None:
        layers = [[(O.MAP, 0, 0.0)]]
    return O.OracleGeometry(grids=grids, layers=layers, geoid=geoid)


def two_layer_spec():
    """flat + C1 map per layer, offsets -0.5 and 0 (tests/golden/generate.py g6)."""
    return [[(O.FLAT, 0, off), (O.MAP, 0, off)] for off in (-0.5, 0.0)]


def hgt_oracle(lat0=45, lon0=3, n=synth.HGT_N):
    nodes = synth.srtm_like_nodes(lat0, lon0, n)
    g = O.OracleGeometry(grids=[O.hgt_grid(lat0, lon0, nodes)],
                         layers=[[(O.MAP, 0, 0.0)]])
    return nodes, g


def mosaic_oracle(tiles, n, lat0, lon0, nlat, nlon):
    """A stack over 1x1 degree HGT tiles; `tiles` lists the (lat, lon) present."""
    grids, table = [], -np.ones((nlat, nlon), dtype=np.int32)
    for la, lo in tiles:
        table[la - lat0, lo - lon0] = len(grids)
        grids.append(O.hgt_grid(la, lo, synth.srtm_like_nodes(la, lo, n)))
    stack = dict(lat0=float(lat0), lon0=float(lon0), dlat=1.0, dlon=1.0,
                 nlat=nlat, nlon=nlon, tile=table)
    return O.OracleGeometry(grids=grids, stacks=[stack], layers=[[(O.STACK, 0, 0.0)]])
