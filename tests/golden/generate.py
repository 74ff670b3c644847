#!/usr/bin/env python3
"""Generate tests/golden/*.npz by driving the REAL reference.

Run in the build container only (needs oracle/_ref/libturtle_ref.so, which
oracle/Makefile compiles from /root/reference in place):

    make -C oracle ref && python tests/golden/generate.py

Every fixture stores its INPUTS (ray origins/directions, query points, the
terrain recipe parameters) next to the reference's OUTPUTS, so nothing depends
on an RNG at test time.  Terrain is regenerated from the formulas in
turtle_amd/synth.py; the SHA-256 of each regenerated payload is stored so a
drift of the recipe is caught before it masquerades as a parity failure.

Fixture families:
  G1 ecef.npz          ECEF <-> geodetic / horizontal known answers
  G2 bilinear.npz      turtle_map_elevation on the C1 map incl. edges, NaN
  G3 c1_traces.npz     C1: 1000 full traces at local_range 0 and 1
  G4 hgt_traces.npz    10 000 rays on the synthetic 3601^2 SRTMGL1 tile
  G5 stack.npz         tile-directory edge cases + traces across a 2x2 mosaic
  G6 layers.npz        multi-layer / offset / flat / geoid step outputs+traces
  G7 steps.npz         per-step records (pos, ds, index) for 16 rays
"""
from __future__ import annotations

import hashlib
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_ffi as R  # noqa: E402
from turtle_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def rays_on(stepper, lat, lon, az, el, height, layer=0):
    pos = np.empty((lat.size, 3))
    for k in range(lat.size):
        rc, p, di = stepper.position(lat[k], lon[k], height, layer)
        assert rc == 0 and di >= 0
        pos[k] = p
    direction = R.ecef_from_horizontal(lat, lon, az, el)
    return pos, direction


# ---------------------------------------------------------------- G1
def g1_ecef():
    rng = np.random.Generator(np.random.Philox(101))
    n = 4096
    lat = rng.uniform(-90, 90, n)
    lon = rng.uniform(-180, 180, n)
    alt = rng.uniform(-1e4, 1e5, n)
    # force both branches of the c^2 > 0.3 test and the special cases
    lat[:8] = [0, 90, -90, 45.5, 56.7, 56.8, -56.79, 89.999999]
    lon[:8] = [90, 0, 0, 3.5, 180, -180, 0, 12]
    alt[:8] = [1000, 1000, 1000, 1000, 0, 0, -500, 1e5]
    ecef = R.ecef_from_geodetic(lat, lon, alt)
    # exact poles and a far/near-centre point
    extra = np.array([[0, 0, 6356752.3142 + 1000.0], [0, 0, -6356752.3142 - 1000.0],
                      [0, 0, 0.0], [1.0, 0, 0], [0, 1e7, 0], [-6378137.0, 0, 0],
                      [1e-300, 0, 6.4e6]])
    ecef_all = np.vstack([ecef, extra])
    la, lo, al = R.ecef_to_geodetic(ecef_all)
    az = rng.uniform(-180, 180, n)
    el = rng.uniform(-90, 90, n)
    az[:4] = [60, 0, 90, 180]
    el[:4] = [30, 90, -90, 0]
    direction = R.ecef_from_horizontal(lat, lon, az, el)
    scale = rng.uniform(0.1, 10, n)[:, None]
    scale[:16] = 1.0
    dir_scaled = direction * scale
    dir_scaled[5] = 0.0  # r <= FLT_EPSILON: outputs untouched (stay 0)
    az2, el2 = R.ecef_to_horizontal(lat, lon, dir_scaled)
    save("ecef.npz", lat=lat, lon=lon, alt=alt, ecef=ecef, ecef_all=ecef_all,
         to_lat=la, to_lon=lo, to_alt=al, az=az, el=el, direction=direction,
         dir_scaled=dir_scaled, to_az=az2, to_el=el2)


# ---------------------------------------------------------------- C1 map
C1_X, C1_Y, C1_Z = (3.0, 4.0), (45.0, 46.0), (0.0, 2000.0)


def c1_map():
    nodes = synth.c1_gradient_nodes()
    return nodes, R.RefMap.create(nodes, C1_X, C1_Y, C1_Z)


def g2_bilinear(m, nodes):
    rng = np.random.Generator(np.random.Philox(202))
    n = 2048
    x = rng.uniform(C1_X[0] - 0.05, C1_X[1] + 0.05, n)
    y = rng.uniform(C1_Y[0] - 0.05, C1_Y[1] + 0.05, n)
    dx = 1.0 / 255
    special = [
        (3.0, 45.0), (4.0, 46.0), (3.0, 46.0), (4.0, 45.0),  # corners
        (4.0, 45.5), (3.5, 46.0), (3.0, 45.5), (3.5, 45.0),  # edges
        (np.nextafter(4.0, 5), 45.5), (np.nextafter(3.0, 0), 45.5),
        (3.5, np.nextafter(46.0, 47)), (3.5, np.nextafter(45.0, 0)),
        (np.nextafter(4.0, 0), np.nextafter(46.0, 0)),
        (3.0 + 254 * dx, 45.0 + 254 * dx), (3.0 + 254.5 * dx, 45.0 + 0.5 * dx),
        (np.nan, 45.5), (3.5, np.nan), (np.inf, 45.5), (3.5, -np.inf),
        (3.0 + dx, 45.0 + dx), (-3.5, 45.5), (3.5, -45.5),
    ]
    for k, (a, b) in enumerate(special):
        x[k], y[k] = a, b
    z, inside = m.elevation(x, y)
    ix = rng.integers(0, 256, 64)
    iy = rng.integers(0, 256, 64)
    node = np.array([m.node(int(a), int(b)) for a, b in zip(ix, iy)])
    save("bilinear.npz", x=x, y=y, z=z, inside=inside, node_ix=ix, node_iy=iy,
         node_xyz=node, nodes_sha=np.array(sha(nodes)))


def g3_c1_traces(m):
    lat, lon, az, el = synth.uniform_rays(1000, C1_Y, C1_X, seed=303)
    out = {}
    for rng_name, local_range in (("r0", 0.0), ("r1", 1.0)):
        st = R.RefStepper()
        st.add_map(m, 0.0)
        st.range_set(local_range)
        if "position" not in out:
            pos, direction = rays_on(st, lat, lon, az, el, 500.0)
            out.update(position=pos, direction=direction)
        t = st.trace(out["position"], out["direction"])
        for k, v in t.items():
            out[f"{rng_name}_{k}"] = v
        st.destroy()
    save("c1_traces.npz", lat=lat, lon=lon, az=az, el=el, **out)


def g7_steps(m):
    lat, lon, az, el = synth.uniform_rays(16, C1_Y, C1_X, seed=707)
    st = R.RefStepper()
    st.add_map(m, 0.0)
    st.range_set(0.0)
    pos, direction = rays_on(st, lat, lon, az, el, 500.0)
    t = st.trace(pos, direction, record=True)
    st.destroy()
    save("steps.npz", position=pos, direction=direction, record=t["record"],
         index=t["index"], length=t["length"], n_steps=t["n_steps"],
         final=t["position"])


# ---------------------------------------------------------------- G4
def g4_hgt(tmp):
    path = synth.write_hgt(tmp, 45, 3)
    nodes = synth.srtm_like_nodes(45, 3)
    m = R.RefMap.load(path)
    # node decode spot checks (row flip + big-endian), hgt.c:127-131
    rng = np.random.Generator(np.random.Philox(404))
    ix = rng.integers(0, 3601, 256)
    iy = rng.integers(0, 3601, 256)
    node = np.array([m.node(int(a), int(b))[2] for a, b in zip(ix, iy)])
    assert np.array_equal(node, nodes[iy, ix].astype(np.float64))
    lat, lon, az, el = synth.uniform_rays(10000, (45.0, 46.0), (3.0, 4.0), seed=404)
    st = R.RefStepper()
    st.add_map(m, 0.0)
    st.range_set(0.0)
    pos, direction = rays_on(st, lat, lon, az, el, 500.0)
    t0 = st.trace(pos, direction)
    st.range_set(1.0)
    t1 = st.trace(pos, direction)
    st.destroy()
    # elevation KATs on the big tile, x = lon, y = lat
    qx = rng.uniform(2.99, 4.01, 4096)
    qy = rng.uniform(44.99, 46.01, 4096)
    qz, qin = m.elevation(qx, qy)
    m.destroy()
    save("hgt_traces.npz", lat=lat, lon=lon, az=az, el=el, position=pos,
         direction=direction, nodes_sha=np.array(sha(nodes)),
         node_ix=ix, node_iy=iy, node_z=node, qx=qx, qy=qy, qz=qz, qin=qin,
         **{f"r0_{k}": v for k, v in t0.items()},
         **{f"r1_{k}": v for k, v in t1.items()})


# ---------------------------------------------------------------- G5
def g5_stack(tmp):
    d = os.path.join(tmp, "mosaic")
    n = 1201
    tiles = [(45, 3), (45, 4), (46, 3)]  # (46, 4) deliberately missing
    for la, lo in tiles:
        synth.write_hgt(d, la, lo, n)
    with open(os.path.join(d, "README.txt"), "w") as f:
        f.write("not a map\n")  # unknown extension must be skipped (stack.c:80-83)
    stack = R.RefStack(d, 0)
    stack.load()
    rng = np.random.Generator(np.random.Philox(505))
    m = 4096
    lat = rng.uniform(44.9, 47.1, m)
    lon = rng.uniform(2.9, 5.1, m)
    e = 1e-7
    special = [
        (45.5, 3.5), (45.0, 3.0), (47.0, 5.0), (46.0, 4.0), (46.0, 3.5),
        (45.5, 4.0), (47.0, 3.5), (45.5, 5.0), (46.5, 4.5), (46.0 + e, 4.0 + e),
        (46.0 - e, 4.0 - e), (47.0 - e, 3.5), (47.0 + e, 3.5), (45.5, 5.0 - e),
        (45.0 - e, 3.5), (45.5, 3.0 - e), (np.nextafter(46.0, 0), 3.25),
        (np.nextafter(46.0, 47), 3.25), (45.25, np.nextafter(4.0, 0)),
        (45.25, np.nextafter(4.0, 5)), (np.nextafter(47.0, 0), 3.5),
        (45.5, np.nextafter(5.0, 0)), (np.nan, 3.5), (45.5, np.nan),
        (-45.5, 3.5), (45.5, -3.5), (1e9, 3.5),
    ]
    for k, (a, b) in enumerate(special):
        lat[k], lon[k] = a, b
    z, inside = stack.elevation(lat, lon)

    # single-tile stack: exclusive outer upper edge (SURVEY 8c probe facts)
    d1 = os.path.join(tmp, "single")
    synth.write_hgt(d1, 45, 3, n)
    one = R.RefStack(d1, 0)
    one.load()
    lat1 = np.array([45.5, 45.0, 46.0, 46.0000001, 45.5, 46.0, np.nextafter(46.0, 0)])
    lon1 = np.array([3.5, 3.0, 4.0, 3.5, 4.0, 3.5, np.nextafter(4.0, 0)])
    z1, in1 = one.elevation(lat1, lon1)
    one.destroy()

    # traces across the mosaic seams, some leaving through the missing tile
    la, lo, az, el = synth.uniform_rays(2000, (45.0, 47.0), (3.0, 5.0), seed=515,
                                        margin=0.02, el_range=(-6.0, -0.2))
    st = R.RefStepper()
    st.add_stack(stack, 0.0)
    st.range_set(0.0)
    pos = np.empty((la.size, 3))
    ok = np.zeros(la.size, dtype=bool)
    for k in range(la.size):
        rc, p, di = st.position(la[k], lo[k], 300.0, 0)
        ok[k] = di >= 0
        pos[k] = p if ok[k] else 0.0
    direction = R.ecef_from_horizontal(la, lo, az, el)
    pos, direction = pos[ok], direction[ok]
    t = st.trace(pos, direction)
    st.destroy()
    stack.destroy()
    save("stack.npz", n=np.array(n), tiles=np.array(tiles), lat=lat, lon=lon, z=z,
         inside=inside, lat1=lat1, lon1=lon1, z1=z1, in1=in1, ray_lat=la[ok],
         ray_lon=lo[ok], position=pos, direction=direction,
         **{f"t_{k}": v for k, v in t.items()})


# ---------------------------------------------------------------- G6
def g6_layers(m_c1):
    """Two layers, each flat + map (offsets -0.5 / 0), mirroring the shape of
    tests/test-turtle.c:255-409 with geodetic data only; then a geoid."""
    rows = []

    def record(tag, st, pos, direction):
        o = st.step(pos, direction)
        rows.append((tag, pos, direction, o))

    # the reference queries the geoid at lon in [0, 360) (stepper.c:45-48)
    gn2 = np.linspace(-30.0, 30.0, 361)[None, :] * np.ones((181, 1))
    geoid360 = R.RefMap.create(gn2, (0.0, 360.0), (-90.0, 90.0), (-40.0, 40.0))

    out = {}
    for gname, g in (("nogeoid", None), ("geoid", geoid360)):
        st = R.RefStepper()
        if g is not None:
            st.geoid_set(g)
        st.range_set(0.0)
        for off in (-0.5, 0.0):
            st.add_layer()
            st.add_flat(off)
            st.add_map(m_c1, off)
        lat0, lon0 = 45.756546, 3.4485671  # on the map
        lat1, lon1 = 40.0, 10.0            # only the flat data holds it
        P, Dd, O = [], [], []
        for (la, lo) in ((lat0, lon0), (lat1, lon1)):
            up = R.ecef_from_horizontal([la], [lo], [0.0], [90.0])[0]
            side = R.ecef_from_horizontal([la], [lo], [35.0], [-2.0])[0]
            for layer in (0, 1):
                for h in (-0.25, 0.5, -0.1, -0.5, 10.0, -10.0):
                    rc, p, di = st.position(la, lo, h, layer)
                    assert rc == 0
                    for d in (None, up, side):
                        for slope in (0.4, 2.0):
                            st.slope_set(slope)
                            st.reset()
                            o = st.step(p, d)
                            P.append(p)
                            Dd.append(np.zeros(3) if d is None else d)
                            O.append([o["rc"], 0 if d is None else 1, slope,
                                      *o["position"], o["latitude"], o["longitude"],
                                      o["altitude"], *o["elevation"], o["step"],
                                      *o["index"], di, layer, h])
        st.slope_set(0.4)
        # traces through the layered geometry, starting in the top medium
        la, lo, az, el = synth.uniform_rays(300, C1_Y, C1_X, seed=606)
        pos = np.empty((la.size, 3))
        for k in range(la.size):
            rc, p, di = st.position(la[k], lo[k], 200.0, 1)
            pos[k] = p
        direction = R.ecef_from_horizontal(la, lo, az, el)
        t = st.trace(pos, direction)
        # continue each ray through the next medium (index changes twice)
        t2 = st.trace(t["position"], direction)
        st.destroy()
        out.update({f"{gname}_P": np.array(P), f"{gname}_D": np.array(Dd),
                    f"{gname}_O": np.array(O), f"{gname}_tpos": pos,
                    f"{gname}_tdir": direction,
                    **{f"{gname}_t_{k}": v for k, v in t.items()},
                    **{f"{gname}_t2_{k}": v for k, v in t2.items()}})
    geoid360.destroy()
    save("layers.npz", geoid_nodes=gn2, **out)


def main():
    if not R.available():
        sys.exit("oracle/_ref/libturtle_ref.so missing: run `make -C oracle ref` "
                 "in the build container")
    tmp = tempfile.mkdtemp(prefix="turtle_golden_")
    try:
        g1_ecef()
        nodes, m = c1_map()
        g2_bilinear(m, nodes)
        g3_c1_traces(m)
        g7_steps(m)
        g6_layers(m)
        m.destroy()
        g4_hgt(tmp)
        g5_stack(tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    errs = R.errors()
    print(f"reference raised {len(errs)} handled errors during generation")


if __name__ == "__main__":
    main()
