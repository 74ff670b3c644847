"""The scalar entry points answered on the host (turtle_amd/csrc/scalar.c; the caller's option
turtle_amd_scalar_set(TURTLE_AMD_SCALAR_HOST)): the host restatement against the REFERENCE's golden
vectors, bit for bit -- it is the reference's arithmetic with the exact transform, on the host's own
libm.  Runs without a GPU: the functions are called directly (the public entry points still want a
device: the option says where a point is computed, it is no fallback)."""
import ctypes as C
import os

import numpy as np
import pytest

import turtle_amd as TA
from turtle_amd import binding, synth
import terrains as T

D = C.c_double
L = binding.lib()
for f in ("tamd_h_stepper_step", "tamd_h_stepper_position", "tamd_h_map_elevation", "tamd_h_stack_elevation"):
    getattr(L, f).restype = C.c_int
L.tamd_h_to_geodetic.restype = None


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


def step(st, pos, direction):
    """one turtle_stepper_step on the host: (position, step, index, lat, lon, alt, elevation)"""
    p = (D * 3)(*pos)
    d = None if direction is None else (D * 3)(*direction)
    la, lo, al, ds = D(), D(), D(), D()
    el = (D * 2)()
    idx = (C.c_int * 2)(-9, -9)
    msg = C.create_string_buffer(4200)
    rc = L.tamd_h_stepper_step(st.h, p, d, C.byref(la), C.byref(lo), C.byref(al), el, C.byref(ds), idx, msg, 4200)
    assert rc == 0, (rc, msg.value)
    return np.array(p[:]), ds.value, np.array(idx[:], dtype=np.int32), la.value, lo.value, al.value, np.array(el[:])


def trace(st, pos, direction, max_steps=100000):
    """the reference harness's loop [ref examples/example-stepper.c:128-140] over the host calls"""
    n = pos.shape[0]
    index = np.empty((n, 2), dtype=np.int32)
    length, nsteps, out = np.zeros(n), np.zeros(n, dtype=np.int32), pos.copy()
    for r in range(n):
        p, _, idx, *_ = step(st, pos[r], None)
        medium, total, k = idx[0], 0.0, 0
        if medium >= 0:
            while k < max_steps:
                p, ds, idx, *_ = step(st, p, direction[r])
                total += ds
                k += 1
                if idx[0] != medium:
                    break
        index[r], length[r], nsteps[r], out[r] = idx, total, k, p
    return dict(index=index, length=length, n_steps=nsteps, position=out)


def position(st, lat, lon, height, layer=0):
    p = (D * 3)()
    di = C.c_int(-9)
    msg = C.create_string_buffer(4200)
    rc = L.tamd_h_stepper_position(st.h, D(lat), D(lon), D(height), layer, p, C.byref(di), msg, 4200)
    assert rc == 0
    return np.array(p[:]), di.value


def test_ecef_known_answers(golden):
    g = golden("ecef")
    la, lo, al = D(), D(), D()
    for k in range(g["ecef_all"].shape[0]):
        e = (D * 3)(*g["ecef_all"][k])
        L.tamd_h_to_geodetic(e, C.byref(la), C.byref(lo), C.byref(al))
        assert (la.value, lo.value, al.value) == (g["to_lat"][k], g["to_lon"][k], g["to_alt"][k])


def test_bilinear_known_answers(golden):
    g = golden("bilinear")
    m = TA.Map.create(T.c1_nodes(), T.C1_X, T.C1_Y, T.C1_Z)
    z = D()
    for x, y, zz, ii in zip(g["x"], g["y"], g["z"], g["inside"]):
        inside = L.tamd_h_map_elevation(m.h, D(x), D(y), C.byref(z))
        assert inside == ii and (not ii or z.value == zz)
    m.destroy()


def test_c1_traces_and_per_step_records(golden):
    g = golden("c1_traces")
    m = TA.Map.create(T.c1_nodes(), T.C1_X, T.C1_Y, T.C1_Z)
    st = TA.Stepper()
    st.add_map(m, 0.0)
    for k in range(0, 64):
        p, di = position(st, g["lat"][k], g["lon"][k], 500.0)
        assert eq(p, g["position"][k]) and di == 0
    sel = slice(0, 300)
    t = trace(st, g["position"][sel], g["direction"][sel])
    for key in ("index", "n_steps", "length", "position"):
        assert eq(t[key], g["r0_" + key][sel]), key       # the reference at local range 0
    s = golden("steps")
    rec = s["record"]
    for r in range(s["position"].shape[0]):
        pos = s["position"][r].copy()
        for row in rec[rec[:, 0] == r]:
            pos, ds, idx, *_ = step(st, pos, s["direction"][r])
            assert eq(pos, row[2:5]) and ds == row[5] and eq(idx, row[6:8].astype(np.int32))
    st.destroy()
    m.destroy()


def test_layers_offsets_flat_and_geoid(golden):
    g = golden("layers")
    import amd_build as B
    m = B.c1_map()
    for name, geoid_nodes in (("nogeoid", None), ("geoid", g["geoid_nodes"])):
        geoid = None if geoid_nodes is None else B.geoid_map(geoid_nodes)
        st = B.two_layer_stepper(m, geoid)
        P, Dq, Oq = g[name + "_P"], g[name + "_D"], g[name + "_O"]
        for k in range(P.shape[0]):
            st.slope = float(Oq[k, 2])
            L.turtle_stepper_reset(st.h)
            pos, ds, idx, la, lo, al, el = step(st, P[k], Dq[k] if Oq[k, 1] else None)
            ref = Oq[k]
            assert eq(pos, ref[3:6]) and (la, lo, al) == tuple(ref[6:9]) and eq(el, ref[9:11])
            assert ds == ref[11] and eq(idx, ref[12:14].astype(np.int32))
        st.slope = 0.4
        t = trace(st, g[name + "_tpos"], g[name + "_tdir"])
        for key in ("index", "n_steps", "length", "position"):
            assert eq(t[key], g[name + "_t_" + key]), key
        st.destroy()
        if geoid is not None:
            geoid.destroy()
    m.destroy()


def test_stack_lookups_loads_and_traces(golden, tmp_path):
    g = golden("stack")
    import amd_build as B
    n = int(g["n"])
    stack = B.mosaic(tmp_path, [tuple(t) for t in g["tiles"]], n)
    z, inside = D(), C.c_int()
    msg = C.create_string_buffer(4200)
    for la, lo, zz, ii in zip(g["lat"], g["lon"], g["z"], g["inside"]):
        assert L.tamd_h_stack_elevation(stack.h, D(la), D(lo), C.byref(z), C.byref(inside), msg, 4200) == 0
        assert inside.value == ii and z.value == zz
    st = TA.Stepper()
    st.add_stack(stack, 0.0)
    sel = slice(0, 400)
    t = trace(st, g["position"][sel], g["direction"][sel])
    for key in ("index", "n_steps", "length", "position"):
        assert eq(t[key], g["t_" + key][sel]), key
    st.destroy()
    stack.destroy()
    # the reference's stack test [ref tests/test-turtle.c:628-690]: size 3 over 4 tiles, the
    # tiles in memory after every query
    small = TA.Stack(os.path.join(str(tmp_path), "mosaic"), 2)
    for (la, lo), want in (((45.5, 3.5), 1), ((45.5, 4.5), 2), ((46.5, 3.5), 2), ((45.5, 3.5), 2)):
        assert L.tamd_h_stack_elevation(small.h, D(la), D(lo), C.byref(z), C.byref(inside), msg, 4200) == 0
        assert inside.value == 1 and small.resident == want
    small.destroy()
