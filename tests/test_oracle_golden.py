"""Pin the CPU restatement (oracle/) against the reference's own outputs.

tests/golden/*.npz were produced by tests/golden/generate.py driving the real
reference library.  The restatement follows the reference expression by
expression, so on the same libm it must agree BIT FOR BIT at local_range = 0
and 1; the asserts below are exact unless a comment says otherwise.
"""
import numpy as np

from oracle import ffi as O
import terrains as T
from turtle_amd import synth


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_g1_ecef(golden):
    g = golden("ecef")
    assert eq(O.ecef_from_geodetic(g["lat"], g["lon"], g["alt"]), g["ecef"])
    la, lo, al = O.ecef_to_geodetic(g["ecef_all"])
    assert eq(la, g["to_lat"]) and eq(lo, g["to_lon"]) and eq(al, g["to_alt"])
    assert eq(O.ecef_from_horizontal(g["lat"], g["lon"], g["az"], g["el"]), g["direction"])
    az, el = O.ecef_to_horizontal(g["lat"], g["lon"], g["dir_scaled"])
    assert eq(az, g["to_az"]) and eq(el, g["to_el"])


def test_g1_reference_test_ecef_assertions():
    """tests/test-turtle.c:582-625 re-expressed."""
    p = O.ecef_from_geodetic([45.5], [3.5], [1000.0])
    la, lo, al = O.ecef_to_geodetic(p)
    assert abs(la[0] - 45.5) < 1e-8 and abs(lo[0] - 3.5) < 1e-8 and abs(al[0] - 1000) < 1e-8
    d = O.ecef_from_horizontal([45.5], [3.5], [60.0], [30.0])
    az, el = O.ecef_to_horizontal([45.5], [3.5], d)
    assert abs(az[0] - 60) < 1e-8 and abs(el[0] - 30) < 1e-8
    for lat, lon in ((90.0, 0.0), (-90.0, 0.0), (0.0, 90.0)):
        p = O.ecef_from_geodetic([lat], [lon], [1000.0])
        la, lo, al = O.ecef_to_geodetic(p)
        if abs(lat) == 90:
            # cos(pi/2) is 6e-17, not 0: the reference test passes because
            # its check is ck_assert_double_eq on the *rounded* result
            assert abs(la[0] - lat) < 1e-9 and abs(al[0] - 1000) < 1e-8
        else:
            assert la[0] == lat and lo[0] == lon and abs(al[0] - 1000) < 1e-8


def test_g2_bilinear(golden):
    g = golden("bilinear")
    assert T.sha(T.c1_nodes()) == str(g["nodes_sha"])
    geo = T.c1_oracle()
    z, inside = geo.grid_elevation(0, g["x"], g["y"])
    assert eq(inside, g["inside"])
    ok = inside == 1
    assert eq(z[ok], g["z"][ok])
    for ix, iy, xyz in zip(g["node_ix"], g["node_iy"], g["node_xyz"]):
        v = O.lib().orc_grid_node(O.C.byref(geo.grids[0]), int(ix), int(iy))
        assert v == xyz[2]


def _check_trace(t, g, prefix):
    assert eq(t["index"], g[prefix + "_index"])
    assert eq(t["n_steps"], g[prefix + "_n_steps"])
    assert eq(t["length"], g[prefix + "_length"])
    assert eq(t["position"], g[prefix + "_position"])


def test_g3_c1_traces(golden):
    g = golden("c1_traces")
    geo = T.c1_oracle()
    pos, di = geo.position(g["lat"], g["lon"], 500.0)
    assert eq(pos, g["position"]) and (di == 0).all()
    assert eq(O.ecef_from_horizontal(g["lat"], g["lon"], g["az"], g["el"]), g["direction"])
    for prefix, rng in (("r0", 0.0), ("r1", 1.0)):
        t = geo.trace(g["position"], g["direction"], local_range=rng)
        _check_trace(t, g, prefix)
    assert 300 < g["r0_n_steps"].mean() < 500  # ~393 steps/ray


def test_g3_threads_do_not_change_results(golden):
    g = golden("c1_traces")
    geo = T.c1_oracle()
    t = geo.trace(g["position"], g["direction"], local_range=0.0, threads=4)
    _check_trace(t, g, "r0")


def test_g7_per_step_records(golden):
    g = golden("steps")
    geo = T.c1_oracle()
    rec = g["record"]
    # replay each ray one turtle_stepper_step at a time (orc_step_n: fresh
    # history per call, exact transform)
    for r in range(g["position"].shape[0]):
        rows = rec[rec[:, 0] == r]
        pos = g["position"][r].copy()
        for row in rows:
            o = geo.step(pos[None, :], g["direction"][r][None, :])
            assert eq(o["position"][0], row[2:5])
            assert o["step"][0] == row[5]
            assert eq(o["index"][0], row[6:8].astype(np.int32))
            pos = o["position"][0]


def test_g4_hgt_tile(golden):
    g = golden("hgt_traces")
    nodes, geo = T.hgt_oracle()
    assert T.sha(nodes) == str(g["nodes_sha"])
    for ix, iy, z in zip(g["node_ix"], g["node_iy"], g["node_z"]):
        assert O.lib().orc_grid_node(O.C.byref(geo.grids[0]), int(ix), int(iy)) == z
    z, inside = geo.grid_elevation(0, g["qx"], g["qy"])
    assert eq(inside, g["qin"])
    assert eq(z[inside == 1], g["qz"][inside == 1])
    pos, di = geo.position(g["lat"], g["lon"], 500.0)
    assert eq(pos, g["position"])
    for prefix, rng in (("r0", 0.0), ("r1", 1.0)):
        t = geo.trace(g["position"], g["direction"], local_range=rng, threads=4)
        _check_trace(t, g, prefix)


def test_g11_c3_seam_full_size_tiles(golden):
    """C3's shape at full tile size: rays across the seams of a 2x2 mosaic of 3601^2 tiles, the
    restatement against the reference's own trace, bit for bit."""
    g = golden("c3_seam")
    tiles = [tuple(t) for t in g["tiles"]]
    assert T.sha(synth.srtm_like_nodes(45, 3)) == str(g["nodes_sha"])
    geo = T.mosaic_oracle(tiles, synth.HGT_N, 45, 3, 2, 2)
    pos, di = geo.position(g["lat"], g["lon"], 300.0)
    assert eq(pos, g["position"]) and (di == 0).all()
    t = geo.trace(g["position"], g["direction"], threads=4)
    _check_trace(t, g, "t")
    # the rays do what the fixture is for: they end in another tile than they started in
    lat1, lon1, _ = O.ecef_to_geodetic(t["position"])
    moved = (np.floor(lat1) != np.floor(g["lat"])) | (np.floor(lon1) != np.floor(g["lon"]))
    assert moved.sum() > 300 and (t["index"][:, 0] == -1).any()


def test_g5_stack(golden):
    g = golden("stack")
    n = int(g["n"])
    geo = T.mosaic_oracle([tuple(t) for t in g["tiles"]], n, 45, 3, 2, 2)
    z, inside = geo.stack_elevation(0, g["lat"], g["lon"])
    assert eq(inside, g["inside"])
    assert eq(z, g["z"])
    one = T.mosaic_oracle([(45, 3)], n, 45, 3, 1, 1)
    z1, in1 = one.stack_elevation(0, g["lat1"], g["lon1"])
    assert eq(in1, g["in1"]) and eq(z1, g["z1"])
    # SURVEY 8c probe facts: (45.5,3.5) in, (45,3) in, (46,4) OUT, (46.0000001,3.5) out
    assert list(in1[:4]) == [1, 1, 0, 0]
    pos, di = geo.position(g["ray_lat"], g["ray_lon"], 300.0)
    assert eq(pos, g["position"])
    t = geo.trace(g["position"], g["direction"])
    _check_trace(t, g, "t")
    assert (t["index"][:, 0] == -1).any() and (t["index"][:, 0] == 0).any()


def test_g6_layers(golden):
    g = golden("layers")
    for name, geoid_nodes in (("nogeoid", None), ("geoid", g["geoid_nodes"])):
        geo = T.c1_oracle(layers=T.two_layer_spec(), geoid_nodes=geoid_nodes)
        P, D, Oq = g[name + "_P"], g[name + "_D"], g[name + "_O"]
        for slope in (0.4, 2.0):
            for has_dir in (0, 1):
                sel = (Oq[:, 2] == slope) & (Oq[:, 1] == has_dir)
                o = geo.step(P[sel], D[sel] if has_dir else None, slope=slope)
                ref = Oq[sel]
                assert (ref[:, 0] == 0).all()
                assert eq(o["position"], ref[:, 3:6])
                assert eq(o["latitude"], ref[:, 6]) and eq(o["longitude"], ref[:, 7])
                assert eq(o["altitude"], ref[:, 8])
                assert eq(o["elevation"], ref[:, 9:11])
                assert eq(o["step"], ref[:, 11])
                assert eq(o["index"], ref[:, 12:14].astype(np.int32))
        # stepper_position on both layers, incl. the data index it reports
        t = geo.trace(g[name + "_tpos"], g[name + "_tdir"])
        _check_trace(t, g, name + "_t")
        t2 = geo.trace(t["position"], g[name + "_tdir"])
        _check_trace(t2, g, name + "_t2")


def test_g8_gradient(golden):
    """turtle_map_gradient incl. the slip at map.c:353 (13 rows of the fixture)."""
    g = golden("gradient")
    geo = O.OracleGeometry(grids=[O.default_grid(g["nodes"], T.C1_X, T.C1_Y, T.C1_Z)],
                           layers=[[(O.MAP, 0, 0.0)]])
    gx, gy, inside = geo.grid_gradient(0, g["x"], g["y"])
    assert eq(inside, g["inside"]) and eq(gx, g["gx"]) and eq(gy, g["gy"])
    assert ((g["gy"] == -7.0) & (g["inside"] == 1)).sum() > 0  # the slip is exercised


UTM_X, UTM_Y, UTM_Z = (495000.0, 497000.0), (5066000.0, 5068000.0), (0.0, 1000.0)


def utm_oracle(g, with_geoid):
    grids = [O.default_grid(g["nodes"], UTM_X, UTM_Y, UTM_Z, projection="UTM 31N")]
    geoid = -1
    if with_geoid:
        grids.append(O.default_grid(g["geoid_nodes"], (0.0, 360.0), (-90.0, 90.0), (-40.0, 40.0)))
        geoid = 1
    return O.OracleGeometry(grids=grids, layers=[[(O.FLAT, 0, -5.0), (O.MAP, 0, 0.0)]],
                            geoid=geoid)


def test_g9_projections(golden):
    """projection.c forward/inverse for the six Lambert variants and UTM."""
    g = golden("projection")
    for k, name in enumerate(g["names"]):
        x, y = O.project(str(name), g[f"p{k}_lat"], g[f"p{k}_lon"])
        assert eq(x, g[f"p{k}_x"]) and eq(y, g[f"p{k}_y"]), name
        la, lo = O.unproject(str(name), g[f"p{k}_x"], g[f"p{k}_y"])
        assert eq(la, g[f"p{k}_ulat"]) and eq(lo, g[f"p{k}_ulon"]), name
        assert np.abs(la - g[f"p{k}_lat"]).max() < 1e-7  # it is a projection pair


def test_g9_projected_map_in_the_stepper(golden):
    """stepper.c:65-83, :243-248, :304-311: a UTM map under the stepper."""
    g = golden("projection")
    for name in ("nogeoid", "geoid"):
        geo = utm_oracle(g, name == "geoid")
        pos, di = geo.position(g[name + "_lat"], g[name + "_lon"], 150.0)
        assert eq(pos, g[name + "_pos"]) and eq(di, g[name + "_di"])
        o = geo.step(g[name + "_pos"][::4])
        rows = g[name + "_rows"]
        assert eq(o["latitude"], rows[:, 0]) and eq(o["longitude"], rows[:, 1])
        assert eq(o["altitude"], rows[:, 2]) and eq(o["elevation"], rows[:, 3:5])
        assert eq(o["step"], rows[:, 5]) and eq(o["index"], rows[:, 6:8].astype(np.int32))
        t = geo.trace(g[name + "_pos"], g[name + "_dir"])
        _check_trace(t, g, name + "_t")
