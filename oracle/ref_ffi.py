"""ctypes front-end of the REAL reference (oracle/_ref/libturtle_ref.so).

Exists only in the build container, where oracle/Makefile compiles the
reference from /root/reference in place.  Used by tests/golden/generate.py to
produce the committed fixtures and by tests/test_oracle_vs_reference.py (which
skips when the library is absent, e.g. on the GPU box).  TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libturtle_ref.so")


def available() -> bool:
    return os.path.exists(LIB_PATH)


class MapInfo(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("x", C.c_double * 2),
                ("y", C.c_double * 2), ("z", C.c_double * 2),
                ("encoding", C.c_char_p)]


HANDLER = C.CFUNCTYPE(None, C.c_int, C.c_void_p, C.c_char_p)
_lib = None
_errors = []


@HANDLER
def _collect(code, function, message):
    _errors.append((code, message.decode()))


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.turtle_stepper_range_get.restype = C.c_double
        L.turtle_stepper_slope_get.restype = C.c_double
        L.turtle_stepper_resolution_get.restype = C.c_double
        L.turtle_error_handler_set(_collect)  # never exit() the interpreter
        _lib = L
    return _lib


def errors():
    out = list(_errors)
    _errors.clear()
    return out


D = C.c_double


class RefMap:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def create(cls, nodes_s2n, x, y, z, projection=None):
        ny, nx = nodes_s2n.shape
        info = MapInfo(nx, ny, (D * 2)(*x), (D * 2)(*y), (D * 2)(*z), None)
        h = C.c_void_p()
        rc = lib().turtle_map_create(C.byref(h), C.byref(info),
                                     projection.encode() if projection else None)
        assert rc == 0, errors()
        m = cls(h)
        fill = lib().turtle_map_fill
        for iy in range(ny):
            for ix in range(nx):
                rc = fill(h, ix, iy, D(float(nodes_s2n[iy, ix])))
                assert rc == 0, errors()
        return m

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        rc = lib().turtle_map_load(C.byref(h), path.encode())
        assert rc == 0, errors()
        return cls(h)

    def elevation(self, x, y):
        x = np.asarray(x, dtype=np.float64)
        z = np.zeros(x.size)
        inside = np.zeros(x.size, dtype=np.int32)
        f = lib().turtle_map_elevation
        zz, ii = D(), C.c_int()
        for k in range(x.size):
            zz.value = 0.0
            f(self.h, D(x[k]), D(y[k]), C.byref(zz), C.byref(ii))
            z[k], inside[k] = zz.value, ii.value
        return z, inside

    def gradient(self, x, y, fill=-7.0):
        x = np.asarray(x, dtype=np.float64)
        gx, gy = np.full(x.size, fill), np.full(x.size, fill)
        inside = np.zeros(x.size, dtype=np.int32)
        f = lib().turtle_map_gradient
        a, b, ii = D(), D(), C.c_int()
        for k in range(x.size):
            a.value = b.value = fill
            f(self.h, D(x[k]), D(y[k]), C.byref(a), C.byref(b), C.byref(ii))
            gx[k], gy[k], inside[k] = a.value, b.value, ii.value
        return gx, gy, inside

    def node(self, ix, iy):
        x, y, z = D(), D(), D()
        rc = lib().turtle_map_node(self.h, ix, iy, C.byref(x), C.byref(y), C.byref(z))
        assert rc == 0
        return x.value, y.value, z.value

    def destroy(self):
        lib().turtle_map_destroy(C.byref(self.h))


class RefStack:
    def __init__(self, path, size=0):
        self.h = C.c_void_p()
        rc = lib().turtle_stack_create(C.byref(self.h), path.encode(), size, None, None)
        assert rc == 0, errors()

    def load(self):
        rc = lib().turtle_stack_load(self.h)
        assert rc == 0, errors()

    def elevation(self, latitude, longitude):
        lat = np.asarray(latitude, dtype=np.float64)
        z = np.zeros(lat.size)
        inside = np.zeros(lat.size, dtype=np.int32)
        f = lib().turtle_stack_elevation
        zz, ii = D(), C.c_int()
        for k in range(lat.size):
            zz.value = 0.0
            f(self.h, D(lat[k]), D(longitude[k]), C.byref(zz), C.byref(ii))
            z[k], inside[k] = zz.value, ii.value
        return z, inside

    def gradient(self, latitude, longitude, fill=-7.0):
        lat = np.asarray(latitude, dtype=np.float64)
        glat, glon = np.full(lat.size, fill), np.full(lat.size, fill)
        inside = np.zeros(lat.size, dtype=np.int32)
        f = lib().turtle_stack_gradient
        a, b, ii = D(), D(), C.c_int()
        for k in range(lat.size):
            a.value = b.value = fill
            f(self.h, D(lat[k]), D(longitude[k]), C.byref(a), C.byref(b), C.byref(ii))
            glat[k], glon[k], inside[k] = a.value, b.value, ii.value
        return glat, glon, inside

    def destroy(self):
        lib().turtle_stack_destroy(C.byref(self.h))


class RefStepper:
    def __init__(self):
        self.h = C.c_void_p()
        assert lib().turtle_stepper_create(C.byref(self.h)) == 0

    def add_layer(self):
        assert lib().turtle_stepper_add_layer(self.h) == 0

    def add_flat(self, offset):
        assert lib().turtle_stepper_add_flat(self.h, D(offset)) == 0

    def add_map(self, m, offset):
        assert lib().turtle_stepper_add_map(self.h, m.h, D(offset)) == 0

    def add_stack(self, s, offset):
        assert lib().turtle_stepper_add_stack(self.h, s.h, D(offset)) == 0

    def geoid_set(self, m):
        lib().turtle_stepper_geoid_set(self.h, m.h if m is not None else None)

    def range_set(self, v):
        lib().turtle_stepper_range_set(self.h, D(v))

    def slope_set(self, v):
        lib().turtle_stepper_slope_set(self.h, D(v))

    def resolution_set(self, v):
        lib().turtle_stepper_resolution_set(self.h, D(v))

    def reset(self):
        lib().turtle_stepper_reset(self.h)

    def position(self, lat, lon, height, layer):
        pos = (D * 3)(0, 0, 0)
        di = C.c_int(-2)
        rc = lib().turtle_stepper_position(self.h, D(lat), D(lon), D(height),
                                           layer, pos, C.byref(di))
        return rc, np.array(pos[:]), di.value

    def step(self, pos, direction):
        """One turtle_stepper_step; returns dict of every output."""
        p = (D * 3)(*pos)
        d = None if direction is None else (D * 3)(*direction)
        la, lo, al, ds = D(), D(), D(), D()
        el = (D * 2)()
        idx = (C.c_int * 2)()
        rc = lib().turtle_stepper_step(self.h, p, d, C.byref(la), C.byref(lo),
                                       C.byref(al), el, C.byref(ds), idx)
        return dict(rc=rc, position=np.array(p[:]), latitude=la.value,
                    longitude=lo.value, altitude=al.value,
                    elevation=np.array(el[:]), step=ds.value,
                    index=np.array(idx[:], dtype=np.int32))

    def trace(self, position, direction, max_steps=100000, record=False):
        """The harness loop (examples/example-stepper.c:128-140 shape)."""
        L = lib()
        pos = np.array(position, dtype=np.float64).reshape(-1, 3)
        dire = np.asarray(direction, dtype=np.float64).reshape(-1, 3)
        n = pos.shape[0]
        index = np.empty((n, 2), dtype=np.int32)
        length = np.empty(n)
        nsteps = np.empty(n, dtype=np.int32)
        rec = []
        p, d = (D * 3)(), (D * 3)()
        ds = D()
        idx = (C.c_int * 2)()
        step = L.turtle_stepper_step
        for r in range(n):
            p[:] = pos[r]
            d[:] = dire[r]
            step(self.h, p, None, None, None, None, None, None, idx)
            medium = idx[0]
            total, k = 0.0, 0
            if medium >= 0:
                while k < max_steps:
                    step(self.h, p, d, None, None, None, None, C.byref(ds), idx)
                    total += ds.value
                    k += 1
                    if record:
                        rec.append((r, k, p[0], p[1], p[2], ds.value, idx[0], idx[1]))
                    if idx[0] != medium:
                        break
            pos[r] = p[:]
            index[r] = idx[:]
            length[r] = total
            nsteps[r] = k
        out = dict(position=pos, index=index, length=length, n_steps=nsteps)
        if record:
            out["record"] = np.array(rec, dtype=np.float64)
        return out

    def destroy(self):
        lib().turtle_stepper_destroy(C.byref(self.h))


def ecef_to_geodetic(ecef):
    e = np.asarray(ecef, dtype=np.float64).reshape(-1, 3)
    out = np.empty((e.shape[0], 3))
    f = lib().turtle_ecef_to_geodetic
    la, lo, al = D(), D(), D()
    for k in range(e.shape[0]):
        f((D * 3)(*e[k]), C.byref(la), C.byref(lo), C.byref(al))
        out[k] = la.value, lo.value, al.value
    return out[:, 0].copy(), out[:, 1].copy(), out[:, 2].copy()


def ecef_from_geodetic(lat, lon, elev):
    lat = np.asarray(lat, dtype=np.float64)
    out = np.empty((lat.size, 3))
    f = lib().turtle_ecef_from_geodetic
    v = (D * 3)()
    for k in range(lat.size):
        f(D(lat[k]), D(lon[k]), D(elev[k]), v)
        out[k] = v[:]
    return out


def ecef_from_horizontal(lat, lon, az, el):
    lat = np.asarray(lat, dtype=np.float64)
    out = np.empty((lat.size, 3))
    f = lib().turtle_ecef_from_horizontal
    v = (D * 3)()
    for k in range(lat.size):
        f(D(lat[k]), D(lon[k]), D(az[k]), D(el[k]), v)
        out[k] = v[:]
    return out


def ecef_to_horizontal(lat, lon, direction):
    lat = np.asarray(lat, dtype=np.float64)
    d = np.asarray(direction, dtype=np.float64).reshape(-1, 3)
    az, el = np.zeros(lat.size), np.zeros(lat.size)
    f = lib().turtle_ecef_to_horizontal
    a, e = D(), D()
    for k in range(lat.size):
        a.value = e.value = 0.0
        f(D(lat[k]), D(lon[k]), (D * 3)(*d[k]), C.byref(a), C.byref(e))
        az[k], el[k] = a.value, e.value
    return az, el


class RefProjection:
    def __init__(self, name):
        self.h = C.c_void_p()
        rc = lib().turtle_projection_create(C.byref(self.h), name.encode())
        assert rc == 0, errors()

    def project(self, lat, lon):
        lat = np.asarray(lat, dtype=np.float64)
        x, y = np.empty(lat.size), np.empty(lat.size)
        f = lib().turtle_projection_project
        a, b = D(), D()
        for k in range(lat.size):
            f(self.h, D(lat[k]), D(lon[k]), C.byref(a), C.byref(b))
            x[k], y[k] = a.value, b.value
        return x, y

    def unproject(self, x, y):
        x = np.asarray(x, dtype=np.float64)
        lat, lon = np.empty(x.size), np.empty(x.size)
        f = lib().turtle_projection_unproject
        a, b = D(), D()
        for k in range(x.size):
            f(self.h, D(x[k]), D(y[k]), C.byref(a), C.byref(b))
            lat[k], lon[k] = a.value, b.value
        return lat, lon

    def destroy(self):
        lib().turtle_projection_destroy(C.byref(self.h))


# ---- the timing harness (oracle/ref_driver.c) --------------------------------

DRIVER_PATH = os.path.join(HERE, "_ref", "libturtle_ref_driver.so")


def driver_available() -> bool:
    return os.path.exists(DRIVER_PATH) and os.path.exists(LIB_PATH)


def trace_map(map_path, position, direction, local_range=1.0, slope=0.4, resolution=1e-2,
              max_steps=100000, threads=1):
    """n rays through the map at `map_path`, stepped by the REAL reference until
    their medium changes (the example harness's loop), `threads` pthreads."""
    L = C.CDLL(DRIVER_PATH)
    L.ref_trace_map_n.restype = C.c_long
    pos = np.array(position, dtype=np.float64, order="C").reshape(-1, 3)
    dire = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    n = pos.shape[0]
    index = np.empty((n, 2), dtype=np.int32)
    length = np.empty(n, dtype=np.float64)
    nsteps = np.empty(n, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    seconds = C.c_double(0.0)   # the stepping alone: the map is loaded before the clock starts
    total = L.ref_trace_map_n(os.fsencode(map_path), D(local_range), D(slope), D(resolution),
                              C.c_long(n), vp(pos), vp(dire), C.c_int(max_steps), vp(index),
                              vp(length), vp(nsteps), C.c_int(threads), C.byref(seconds))
    if total < 0:
        raise RuntimeError(f"the reference could not load {map_path}")
    return dict(position=pos, index=index, length=length, n_steps=nsteps, total_steps=int(total),
                seconds=seconds.value)


def stack_run(stack_path, position, direction, walk_steps=0, stack_size=0, local_range=1.0, slope=0.4,
              resolution=1e-2, max_steps=100000, threads=1, locked=True):
    """n rays through the turtle_stack over the .hgt tiles in `stack_path`, stepped by the REAL
    reference: ONE stack with lock / unlock shared by `threads` pthreads, a client per worker (the
    reference's threaded example).  walk_steps = 0: each ray to its first boundary, direction[n][3];
    walk_steps > 0: a scattering walk of that many steps, direction[walk_steps][n][3]."""
    L = C.CDLL(DRIVER_PATH)
    L.ref_stack_n.restype = C.c_long
    pos = np.array(position, dtype=np.float64, order="C").reshape(-1, 3)
    n = pos.shape[0]
    dire = np.ascontiguousarray(direction, dtype=np.float64)
    assert dire.size == 3 * n * max(1, walk_steps)
    index = np.empty((n, 2), dtype=np.int32)
    length = np.empty(n, dtype=np.float64)
    nsteps = np.empty(n, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    seconds = C.c_double(0.0)   # the stepping alone: the tiles are loaded before the clock starts
    if locked:
        total = L.ref_stack_n(os.fsencode(stack_path), C.c_int(stack_size), D(local_range), D(slope),
                              D(resolution), C.c_long(n), vp(pos), vp(dire), C.c_int(max_steps),
                              C.c_int(walk_steps), vp(index), vp(length), vp(nsteps), C.c_int(threads),
                              C.byref(seconds))
    else:
        # one thread, no lock / unlock: the stepper looks the stack up itself, no client (ref_driver.c)
        L.ref_stack_unlocked_n.restype = C.c_long
        total = L.ref_stack_unlocked_n(os.fsencode(stack_path), C.c_int(stack_size), D(local_range), D(slope),
                                       D(resolution), C.c_long(n), vp(pos), vp(dire), C.c_int(max_steps),
                                       C.c_int(walk_steps), vp(index), vp(length), vp(nsteps), C.byref(seconds))
    if total < 0:
        raise RuntimeError(f"the reference could not make a stack of {stack_path}")
    return dict(position=pos, index=index, length=length, n_steps=nsteps, total_steps=int(total),
                seconds=seconds.value)
