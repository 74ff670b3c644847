"""Build product-side (libturtle_amd) objects for the recipes of terrains.py,
through the public C API only."""
from __future__ import annotations

import os

import turtle_amd as TA
from turtle_amd import synth

import terrains as T


def c1_map():
    return TA.Map.create(T.c1_nodes(), T.C1_X, T.C1_Y, T.C1_Z)


def geoid_map(nodes):
    return TA.Map.create(nodes, (0.0, 360.0), (-90.0, 90.0), (-40.0, 40.0))


def c1_stepper(m):
    st = TA.Stepper()
    st.add_map(m, 0.0)
    return st


def two_layer_stepper(m, geoid=None):
    st = TA.Stepper()
    if geoid is not None:
        st.geoid_set(geoid)
    for off in (-0.5, 0.0):
        st.add_layer()
        st.add_flat(off)
        st.add_map(m, off)
    return st


def hgt_tile(tmpdir, lat0=45, lon0=3, n=synth.HGT_N):
    return TA.Map.load(synth.write_hgt(str(tmpdir), lat0, lon0, n))


def mosaic(tmpdir, tiles, n, fmt="hgt"):
    """a stack over synthetic 1x1 degree tiles: SRTM's .hgt, or ASTER-GDEM2's GeoTIFF"""
    d = os.path.join(str(tmpdir), "mosaic")
    for la, lo in tiles:
        (synth.write_geotiff if fmt == "tif" else synth.write_hgt)(d, la, lo, n)
    with open(os.path.join(d, "README.txt"), "w") as f:
        f.write("not a map\n")
    return TA.Stack(d, 0)
