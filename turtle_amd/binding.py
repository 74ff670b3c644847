"""ctypes binding of libturtle_amd.so (see include/turtle_amd.h).

Arrays may be numpy arrays (HOST space: copied through HBM by the library) or
torch CUDA tensors (DEVICE space: used in place, launches are asynchronous on
the stream given to :func:`set_stream`).  Outputs are allocated in the same
space as the inputs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HOST, DEVICE = 0, 1
STEP_RESUME = 1
TRACE_RESUME = 1
SCATTER_START = 1

RETURN_NAMES = [
    "SUCCESS", "BAD_ADDRESS", "BAD_EXTENSION", "BAD_FORMAT", "BAD_PROJECTION",
    "BAD_JSON", "DOMAIN_ERROR", "LIBRARY_ERROR", "LOCK_ERROR", "MEMORY_ERROR",
    "PATH_ERROR", "UNLOCK_ERROR",
]


class TurtleError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code
        self.name = RETURN_NAMES[code] if 0 <= code < len(RETURN_NAMES) else str(code)


def library_path() -> str:
    # TURTLE_AMD_LIBRARY: another build of the same library (kernel experiments)
    return os.environ.get("TURTLE_AMD_LIBRARY") or os.path.join(HERE, "libturtle_amd.so")


def build(force: bool = False) -> str:
    """Compile the library in-tree (gcc + hipcc --offload-arch=gfx950)."""
    args = ["make", "-s", "-C", os.path.join(HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return library_path()


class _MapInfo(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("x", C.c_double * 2),
                ("y", C.c_double * 2), ("z", C.c_double * 2), ("encoding", C.c_char_p)]


_HANDLER = C.CFUNCTYPE(None, C.c_int, C.c_void_p, C.c_char_p)
_pending = []


@_HANDLER
def _on_error(code, function, message):
    # never let the default handler exit() the interpreter: record, and let
    # the wrapper raise TurtleError when the call returns
    _pending.append((code, message.decode(errors="replace")))


_lib = None


def lib():
    """The loaded C library.  Raises if it has not been built: there is no
    fallback implementation."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or make -C turtle_amd/csrc); turtle_amd has no "
                "pure-Python or CPU implementation")
        L = C.CDLL(path)
        for name in ("turtle_stepper_range_get", "turtle_stepper_slope_get",
                     "turtle_stepper_resolution_get"):
            getattr(L, name).restype = C.c_double
        L.turtle_error_function.restype = C.c_char_p
        L.turtle_stepper_geoid_get.restype = C.c_void_p
        L.turtle_error_handler_set(_on_error)
        _lib = L
    return _lib


def _check(rc):
    pend = list(_pending)
    _pending.clear()
    if rc != 0:
        msg = pend[-1][1] if pend else f"turtle error #{rc}"
        raise TurtleError(rc, msg)
    if pend:  # void functions report only through the handler
        raise TurtleError(pend[-1][0], pend[-1][1])


# ---- array plumbing ---------------------------------------------------------

def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _space_of(*arrays):
    dev = [a for a in arrays if a is not None and _is_torch(a) and a.is_cuda]
    return DEVICE if dev else HOST


def _as(a, space, dtype=np.float64):
    """Contiguous array of the right dtype in the right space (or None)."""
    if a is None:
        return None
    if space == DEVICE:
        import torch
        tdt = torch.float64 if dtype == np.float64 else torch.int32
        if not _is_torch(a):
            a = torch.as_tensor(np.asarray(a, dtype=dtype), device="cuda")
        return a.to(dtype=tdt).contiguous()
    if _is_torch(a):
        a = a.cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def _new(shape, space, dtype=np.float64, like=None, zero=False):
    if space == DEVICE:
        import torch
        tdt = torch.float64 if dtype == np.float64 else (
            torch.int32 if dtype == np.int32 else torch.int64)
        dev = like.device if (like is not None and _is_torch(like)) else "cuda"
        return (torch.zeros if zero else torch.empty)(shape, dtype=tdt, device=dev)
    return (np.zeros if zero else np.empty)(shape, dtype=dtype)


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


# ---- device management ------------------------------------------------------

def device_count() -> int:
    return lib().turtle_amd_device_count()


def compute_units() -> int:
    return lib().turtle_amd_compute_units()


def set_stream(stream=None):
    """Launch on the given stream: a torch.cuda.Stream, a raw hipStream_t
    integer, or None for the library's own stream."""
    handle = None
    if stream is not None:
        raw = int(getattr(stream, "cuda_stream", stream))
        if raw == 0:
            raise ValueError(
                "the legacy default stream has handle 0, which the C API reads as "
                "'use the library stream': pass a torch.cuda.Stream() and make it "
                "current with torch.cuda.stream(...)")
        handle = C.c_void_p(raw)
    _check(lib().turtle_amd_stream_set(handle))


def set_in_flight(batches):
    """Hint: the batches this thread keeps in flight (turtle_amd_in_flight_set)."""
    lib().turtle_amd_in_flight_set(int(batches))


def get_in_flight():
    return int(lib().turtle_amd_in_flight_get())


def synchronize():
    _check(lib().turtle_amd_synchronize())


def set_math(mode):
    """'fast' (default) or 'strict' arithmetic in the trace kernel
    (enum turtle_amd_math in turtle_amd.h)."""
    lib().turtle_amd_math_set({"fast": 0, "strict": 1}[mode])


def get_math():
    return "strict" if lib().turtle_amd_math_get() else "fast"


def set_scalar(where):
    """where the scalar (one point a call) drop-in functions compute: 'device' (default: the
    kernels with n = 1) or 'host' (turtle_amd/csrc/scalar.c; enum turtle_amd_scalar)"""
    lib().turtle_amd_scalar_set({"device": 0, "host": 1}[where])


def get_scalar():
    return "host" if lib().turtle_amd_scalar_get() else "device"


# ---- ECEF -------------------------------------------------------------------

def ecef_from_geodetic(latitude, longitude, elevation):
    sp = _space_of(latitude, longitude, elevation)
    la, lo, el = (_as(v, sp) for v in (latitude, longitude, elevation))
    out = _new((la.shape[0], 3), sp, like=la)
    _check(lib().turtle_ecef_from_geodetic_n(C.c_long(la.shape[0]), _ptr(la), _ptr(lo),
                                             _ptr(el), _ptr(out), sp))
    return out


def ecef_to_geodetic(ecef):
    sp = _space_of(ecef)
    e = _as(ecef, sp).reshape(-1, 3)
    n = e.shape[0]
    la, lo, al = (_new((n,), sp, like=e) for _ in range(3))
    _check(lib().turtle_ecef_to_geodetic_n(C.c_long(n), _ptr(e), _ptr(la), _ptr(lo),
                                           _ptr(al), sp))
    return la, lo, al


def ecef_from_horizontal(latitude, longitude, azimuth, elevation):
    sp = _space_of(latitude, longitude, azimuth, elevation)
    la, lo, az, el = (_as(v, sp) for v in (latitude, longitude, azimuth, elevation))
    out = _new((la.shape[0], 3), sp, like=la)
    _check(lib().turtle_ecef_from_horizontal_n(C.c_long(la.shape[0]), _ptr(la), _ptr(lo),
                                               _ptr(az), _ptr(el), _ptr(out), sp))
    return out


def ecef_to_horizontal(latitude, longitude, direction):
    sp = _space_of(latitude, longitude, direction)
    la, lo = _as(latitude, sp), _as(longitude, sp)
    d = _as(direction, sp).reshape(-1, 3)
    az = _new((la.shape[0],), sp, like=la, zero=True)
    el = _new((la.shape[0],), sp, like=la, zero=True)
    _check(lib().turtle_ecef_to_horizontal_n(C.c_long(la.shape[0]), _ptr(la), _ptr(lo),
                                             _ptr(d), _ptr(az), _ptr(el), sp))
    return az, el


def scalar_ecef_to_geodetic(ecef):
    e = (C.c_double * 3)(*ecef)
    la, lo, al = C.c_double(), C.c_double(), C.c_double()
    lib().turtle_ecef_to_geodetic(e, C.byref(la), C.byref(lo), C.byref(al))
    _check(0)
    return la.value, lo.value, al.value


def scalar_ecef_from_geodetic(latitude, longitude, elevation):
    out = (C.c_double * 3)()
    lib().turtle_ecef_from_geodetic(C.c_double(latitude), C.c_double(longitude),
                                    C.c_double(elevation), out)
    _check(0)
    return np.array(out[:])


# ---- projections ---------------------------------------------------------------

class Projection:
    """struct turtle_projection handle ("Lambert 93", "UTM 31N", ...)."""

    def __init__(self, name):
        self.h = C.c_void_p()
        _check(lib().turtle_projection_create(C.byref(self.h), name.encode()))

    @property
    def name(self):
        f = lib().turtle_projection_name
        f.restype = C.c_char_p
        v = f(self.h)
        return None if v is None else v.decode()

    def project(self, latitude, longitude):
        sp = _space_of(latitude, longitude)
        la, lo = _as(latitude, sp), _as(longitude, sp)
        x, y = _new((la.shape[0],), sp, like=la), _new((la.shape[0],), sp, like=la)
        _check(lib().turtle_projection_project_n(self.h, C.c_long(la.shape[0]), _ptr(la),
                                                 _ptr(lo), _ptr(x), _ptr(y), sp))
        return x, y

    def unproject(self, x, y):
        sp = _space_of(x, y)
        x, y = _as(x, sp), _as(y, sp)
        la, lo = _new((x.shape[0],), sp, like=x), _new((x.shape[0],), sp, like=x)
        _check(lib().turtle_projection_unproject_n(self.h, C.c_long(x.shape[0]), _ptr(x),
                                                   _ptr(y), _ptr(la), _ptr(lo), sp))
        return la, lo

    def project_scalar(self, latitude, longitude):
        x, y = C.c_double(-1.0), C.c_double(-1.0)
        _check(lib().turtle_projection_project(self.h, C.c_double(latitude),
                                               C.c_double(longitude), C.byref(x), C.byref(y)))
        return x.value, y.value

    def destroy(self):
        if self.h:
            lib().turtle_projection_destroy(C.byref(self.h))
        self.h = None


# ---- maps / stacks -----------------------------------------------------------

class Map:
    """struct turtle_map handle."""

    def __init__(self, handle, owner=True):
        self.h = handle
        self._owner = owner

    @classmethod
    def create(cls, nodes_s2n=None, x=(0, 1), y=(0, 1), z=(0, 1), shape=None,
               projection=None):
        """turtle_map_create (+ turtle_map_fill of every node if nodes given)."""
        ny, nx = nodes_s2n.shape if nodes_s2n is not None else shape
        info = _MapInfo(nx, ny, (C.c_double * 2)(*x), (C.c_double * 2)(*y),
                        (C.c_double * 2)(*z), None)
        h = C.c_void_p()
        _check(lib().turtle_map_create(C.byref(h), C.byref(info),
                                       projection.encode() if projection else None))
        m = cls(h)
        if nodes_s2n is not None:
            fill = lib().turtle_map_fill
            for iy in range(ny):
                row = nodes_s2n[iy]
                for ix in range(nx):
                    rc = fill(h, ix, iy, C.c_double(float(row[ix])))
                    if rc:
                        _check(rc)
        return m

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        _check(lib().turtle_map_load(C.byref(h), os.fsencode(path)))
        return cls(h)

    def dump(self, path):
        _check(lib().turtle_map_dump(self.h, os.fsencode(path)))

    def fill(self, ix, iy, z):
        _check(lib().turtle_map_fill(self.h, ix, iy, C.c_double(z)))

    def node(self, ix, iy):
        x, y, z = C.c_double(), C.c_double(), C.c_double()
        _check(lib().turtle_map_node(self.h, ix, iy, C.byref(x), C.byref(y), C.byref(z)))
        return x.value, y.value, z.value

    def meta(self):
        info = _MapInfo()
        proj = C.c_char_p()
        lib().turtle_map_meta(self.h, C.byref(info), C.byref(proj))
        return dict(nx=info.nx, ny=info.ny, x=tuple(info.x), y=tuple(info.y),
                    z=tuple(info.z), encoding=info.encoding.decode(),
                    projection=None if proj.value is None else proj.value.decode())

    def elevation(self, x, y):
        """Batch bilinear lookup -> (z, inside)."""
        sp = _space_of(x, y)
        x, y = _as(x, sp), _as(y, sp)
        n = x.shape[0]
        z = _new((n,), sp, like=x)
        inside = _new((n,), sp, np.int32, like=x)
        _check(lib().turtle_map_elevation_n(self.h, C.c_long(n), _ptr(x), _ptr(y), _ptr(z),
                                            _ptr(inside), sp))
        return z, inside

    def gradient(self, x, y, fill=0.0):
        """Batch gradient -> (gx, gy, inside); outputs start at `fill`."""
        sp = _space_of(x, y)
        x, y = _as(x, sp), _as(y, sp)
        n = x.shape[0]
        gx, gy = _new((n,), sp, like=x), _new((n,), sp, like=x)
        gx[...] = fill
        gy[...] = fill
        inside = _new((n,), sp, np.int32, like=x)
        _check(lib().turtle_map_gradient_n(self.h, C.c_long(n), _ptr(x), _ptr(y), _ptr(gx),
                                           _ptr(gy), _ptr(inside), sp))
        return gx, gy, inside

    def elevation_scalar(self, x, y, want_inside=True):
        z, inside = C.c_double(-12345.0), C.c_int(-1)
        rc = lib().turtle_map_elevation(self.h, C.c_double(x), C.c_double(y), C.byref(z),
                                        C.byref(inside) if want_inside else None)
        _check(rc)
        return z.value, inside.value

    def destroy(self):
        if self.h and self._owner:
            lib().turtle_map_destroy(C.byref(self.h))
        self.h = None


class Stack:
    """struct turtle_stack handle."""

    def __init__(self, path, size=0, lock=None, unlock=None):
        self.h = C.c_void_p()
        self._lock = (lock, unlock)
        _check(lib().turtle_stack_create(C.byref(self.h), os.fsencode(path), size, lock,
                                         unlock))

    def load(self):
        _check(lib().turtle_stack_load(self.h))

    def clear(self):
        _check(lib().turtle_stack_clear(self.h))

    @property
    def resident(self):
        """tiles in memory right now"""
        return int(lib().turtle_amd_stack_resident(self.h))

    def elevation(self, latitude, longitude):
        sp = _space_of(latitude, longitude)
        la, lo = _as(latitude, sp), _as(longitude, sp)
        n = la.shape[0]
        z = _new((n,), sp, like=la)
        inside = _new((n,), sp, np.int32, like=la)
        _check(lib().turtle_stack_elevation_n(self.h, C.c_long(n), _ptr(la), _ptr(lo),
                                              _ptr(z), _ptr(inside), sp))
        return z, inside

    def gradient(self, latitude, longitude, fill=0.0):
        sp = _space_of(latitude, longitude)
        la, lo = _as(latitude, sp), _as(longitude, sp)
        n = la.shape[0]
        glat, glon = _new((n,), sp, like=la), _new((n,), sp, like=la)
        glat[...] = fill
        glon[...] = fill
        inside = _new((n,), sp, np.int32, like=la)
        _check(lib().turtle_stack_gradient_n(self.h, C.c_long(n), _ptr(la), _ptr(lo),
                                             _ptr(glat), _ptr(glon), _ptr(inside), sp))
        return glat, glon, inside

    def elevation_scalar(self, latitude, longitude, want_inside=True):
        z, inside = C.c_double(-12345.0), C.c_int(-1)
        _check(lib().turtle_stack_elevation(
            self.h, C.c_double(latitude), C.c_double(longitude), C.byref(z),
            C.byref(inside) if want_inside else None))
        return z.value, inside.value

    def destroy(self):
        if self.h:
            lib().turtle_stack_destroy(C.byref(self.h))
        self.h = None


# ---- stepper -----------------------------------------------------------------

class Stepper:
    """struct turtle_stepper handle: same verbs as the C API."""

    def __init__(self):
        self.h = C.c_void_p()
        self._keep = []
        _check(lib().turtle_stepper_create(C.byref(self.h)))

    def clone(self):
        """A second stepper over the same geometry and settings (turtle_amd_stepper_clone): one
        stepper is one stream of calls, a batch more in flight takes a stepper more."""
        other = Stepper.__new__(Stepper)
        other.h = C.c_void_p()
        other._keep = list(self._keep)
        _check(lib().turtle_amd_stepper_clone(self.h, C.byref(other.h)))
        return other

    def add_layer(self):
        _check(lib().turtle_stepper_add_layer(self.h))

    def add_flat(self, offset=0.0):
        _check(lib().turtle_stepper_add_flat(self.h, C.c_double(offset)))

    def add_map(self, m, offset=0.0):
        self._keep.append(m)
        _check(lib().turtle_stepper_add_map(self.h, m.h, C.c_double(offset)))

    def add_stack(self, s, offset=0.0):
        self._keep.append(s)
        _check(lib().turtle_stepper_add_stack(self.h, s.h, C.c_double(offset)))

    def geoid_set(self, m):
        self._keep.append(m)
        lib().turtle_stepper_geoid_set(self.h, m.h if m is not None else None)

    range = property(lambda s: lib().turtle_stepper_range_get(s.h),
                     lambda s, v: lib().turtle_stepper_range_set(s.h, C.c_double(v)))
    slope = property(lambda s: lib().turtle_stepper_slope_get(s.h),
                     lambda s, v: lib().turtle_stepper_slope_set(s.h, C.c_double(v)))
    resolution = property(
        lambda s: lib().turtle_stepper_resolution_get(s.h),
        lambda s, v: lib().turtle_stepper_resolution_set(s.h, C.c_double(v)))

    # -- batch --
    def position(self, latitude, longitude, height, layer=0, out=None):
        sp = _space_of(latitude, longitude)
        la, lo = _as(latitude, sp), _as(longitude, sp)
        n = la.shape[0]
        if np.isscalar(height):
            h = _new((n,), sp, like=la)
            h[...] = height
        else:
            h = _as(height, sp)
        pos = out if out is not None else _new((n, 3), sp, like=la, zero=True)
        di = _new((n,), sp, np.int32, like=la)
        _check(lib().turtle_stepper_position_n(self.h, C.c_long(n), _ptr(la), _ptr(lo),
                                               _ptr(h), layer, _ptr(pos), _ptr(di), sp))
        return pos, di

    def step(self, position, direction=None, resume=None, outputs=True):
        """turtle_stepper_step_n.  `position` is updated IN PLACE when it
        already is a contiguous float64 array/tensor.  `resume` = the dict a
        previous call returned for these positions (TURTLE_AMD_STEP_RESUME)."""
        sp = _space_of(position, direction)
        pos = _as(position, sp).reshape(-1, 3)
        d = None if direction is None else _as(direction, sp).reshape(-1, 3)
        n = pos.shape[0]
        flags = 0
        if resume is not None:
            flags = STEP_RESUME
            out = resume
        else:
            out = dict(
                latitude=_new((n,), sp, like=pos) if outputs else None,
                longitude=_new((n,), sp, like=pos) if outputs else None,
                altitude=_new((n,), sp, like=pos), elevation=_new((n, 2), sp, like=pos),
                index=_new((n, 2), sp, np.int32, like=pos))
        out["step"] = _new((n,), sp, like=pos)
        _check(lib().turtle_stepper_step_n(
            self.h, C.c_long(n), _ptr(pos), _ptr(d), _ptr(out["latitude"]),
            _ptr(out["longitude"]), _ptr(out["altitude"]), _ptr(out["elevation"]),
            _ptr(out["step"]), _ptr(out["index"]), flags, sp))
        out["position"] = pos
        return out

    def walk(self, position, direction=None, state=None):
        """turtle_stepper_walk_n: a walk's steps with the least state between the calls.
        direction None begins it (returns the state: position, next, index, step); else `state`
        = the dict the last call returned for these positions, updated in place."""
        if state is None:
            sp = _space_of(position)
            pos = _as(position, sp).reshape(-1, 3)
            n = pos.shape[0]
            state = dict(position=pos, next=_new((n,), sp, like=pos), step=_new((n,), sp, like=pos),
                         index=_new((n, 2), sp, np.int32, like=pos))
        sp = _space_of(state["position"], direction)
        n = state["position"].shape[0]
        d = None if direction is None else _as(direction, sp).reshape(-1, 3)
        _check(lib().turtle_stepper_walk_n(self.h, C.c_long(n), _ptr(state["position"]), _ptr(d),
                                           _ptr(state["next"]), _ptr(state["step"]), _ptr(state["index"]), sp))
        return state

    def scatter(self, position, seed, n_steps, first_ray=0, first_step=0, state=None):
        """turtle_stepper_scatter_n: `n_steps` generations of single steps in
        Philox(first_ray + r, generation; seed) directions, sums kept on the device.
        `state` = the dict a previous call returned (continues that walk: its
        arrays are updated in place); None starts one at `position`."""
        flags = 0
        if state is None:
            sp = _space_of(position)
            pos = _as(position, sp).reshape(-1, 3)
            n = pos.shape[0]
            state = dict(position=pos, altitude=_new((n,), sp, like=pos),
                         elevation=_new((n, 2), sp, like=pos),
                         index=_new((n, 2), sp, np.int32, like=pos),
                         length=_new((n,), sp, like=pos), steps=_new((n,), sp, np.int32, like=pos))
            flags = SCATTER_START
        sp = _space_of(state["position"])
        n = state["position"].shape[0]
        _check(lib().turtle_stepper_scatter_n(
            self.h, C.c_long(n), _ptr(state["position"]), C.c_ulonglong(seed), C.c_long(first_ray),
            first_step, n_steps, _ptr(state["altitude"]), _ptr(state["elevation"]),
            _ptr(state["index"]), _ptr(state["length"]), _ptr(state["steps"]), flags, sp))
        return state

    def trace(self, position, direction, max_steps=100000, want=("length", "n_steps"),
              resume_index=None):
        """turtle_stepper_trace_n.  `resume_index` = the index array a previous
        trace returned for these rays (TURTLE_AMD_TRACE_RESUME)."""
        sp = _space_of(position, direction)
        pos = _as(position, sp).reshape(-1, 3)
        d = _as(direction, sp).reshape(-1, 3)
        n = pos.shape[0]
        flags = 0
        if resume_index is not None:
            flags = TRACE_RESUME
            index = _as(resume_index, sp, np.int32).reshape(-1, 2)
            index = index.clone() if _is_torch(index) else index.copy()
        else:
            index = _new((n, 2), sp, np.int32, like=pos)
        length = _new((n,), sp, like=pos) if "length" in want else None
        nsteps = _new((n,), sp, np.int32, like=pos) if "n_steps" in want else None
        _check(lib().turtle_stepper_trace_n(self.h, C.c_long(n), _ptr(pos), _ptr(d),
                                            max_steps, _ptr(index), _ptr(length),
                                            _ptr(nsteps), flags, sp))
        return dict(position=pos, index=index, length=length, n_steps=nsteps)

    def trace_into(self, pos, d, index, length, nsteps, max_steps=100000):
        """Device-resident trace with caller-owned tensors (no allocation):
        the timed call of bench.py."""
        _check(lib().turtle_stepper_trace_n(self.h, C.c_long(pos.shape[0]), _ptr(pos),
                                            _ptr(d), max_steps, _ptr(index), _ptr(length),
                                            _ptr(nsteps), 0, DEVICE))

    @property
    def rounds(self):
        """rounds the last batch call took (1: no tile had to be paged in)"""
        return int(lib().turtle_amd_stepper_rounds(self.h))

    def trace_stats(self):
        s = (C.c_ulonglong * 4)()
        _check(lib().turtle_stepper_trace_stats(self.h, s))
        return dict(rays=s[0], steps=s[1], samples=s[2], capped=s[3])

    # -- scalar (the reference's own entry points) --
    def step_scalar(self, position, direction=None, want_index=True):
        p = (C.c_double * 3)(*position)
        d = None if direction is None else (C.c_double * 3)(*direction)
        la, lo, al, ds = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        el = (C.c_double * 2)()
        idx = (C.c_int * 2)(-9, -9)
        rc = lib().turtle_stepper_step(self.h, p, d, C.byref(la), C.byref(lo), C.byref(al),
                                       el, C.byref(ds), idx if want_index else None)
        _check(rc)
        return dict(position=np.array(p[:]), latitude=la.value, longitude=lo.value,
                    altitude=al.value, elevation=np.array(el[:]), step=ds.value,
                    index=np.array(idx[:], dtype=np.int32))

    def position_scalar(self, latitude, longitude, height, layer=0, want_index=True,
                        initial=(0.0, 0.0, 0.0)):
        p = (C.c_double * 3)(*initial)
        di = C.c_int(-9)
        rc = lib().turtle_stepper_position(
            self.h, C.c_double(latitude), C.c_double(longitude), C.c_double(height), layer,
            p, C.byref(di) if want_index else None)
        _check(rc)
        return np.array(p[:]), di.value

    def destroy(self):
        if self.h:
            _check(lib().turtle_stepper_destroy(C.byref(self.h)))
        self.h = None


def tally(index, length, n_media, n_bins, length_max, hits=None, histogram=None):
    """turtle_amd_tally_n: uint64 hit counts per final medium (-1..n_media-1)
    and a linear path-length histogram with an overflow bin; accumulates into
    `hits` / `histogram` when given."""
    sp = _space_of(index, length)
    idx = _as(index, sp, np.int32)
    ln = _as(length, sp)
    if hits is None:
        hits = _new((n_media + 1,), sp, np.int64, like=ln, zero=True)
    if histogram is None:
        histogram = _new((n_bins + 1,), sp, np.int64, like=ln, zero=True)
    _check(lib().turtle_amd_tally_n(C.c_long(ln.shape[0]), _ptr(idx), _ptr(ln), n_media,
                                    _ptr(hits), n_bins, C.c_double(length_max),
                                    _ptr(histogram), sp))
    return hits, histogram


def isotropic(n, seed, stream, first_ray=0, device=True, out=None):
    """turtle_amd_isotropic_n: n unit vectors from Philox(ray, stream; seed)."""
    sp = DEVICE if device else HOST
    d = out if out is not None else _new((n, 3), sp)
    _check(lib().turtle_amd_isotropic_n(C.c_long(n), C.c_ulonglong(seed),
                                        C.c_ulonglong(stream), C.c_long(first_ray), _ptr(d), sp))
    return d


def philox(n, seed, stream, first_ray=0):
    """Raw Philox-4x32-10 blocks [n][4] (host array), for known-answer tests."""
    w = np.empty((n, 4), dtype=np.uint32)
    _check(lib().turtle_amd_philox_n(C.c_long(n), C.c_ulonglong(seed), C.c_ulonglong(stream),
                                     C.c_long(first_ray), _ptr(w), HOST))
    return w
