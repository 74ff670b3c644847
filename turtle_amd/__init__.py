"""turtle_amd -- MI355X-native TURTLE ray/terrain stepper.

The product is the C-ABI shared library ``libturtle_amd.so`` (sources in
``turtle_amd/csrc``, header ``include/turtle_amd.h``).  This package is the thin
Python binding used by the tests and by ``bench.py``: it loads the library with
ctypes and mirrors the C API one to one (same names, same argument meaning,
same error codes).  There is no Python or CPU implementation of the path here:
if the library is missing or no gfx950 device is usable, calls fail loudly.
"""
from . import synth  # noqa: F401
from .binding import (  # noqa: F401
    DEVICE, HOST, STEP_RESUME, Map, Projection, Stack, Stepper, TurtleError, build, device_count,
    compute_units, ecef_from_geodetic, ecef_from_horizontal, ecef_to_geodetic,
    ecef_to_horizontal, get_math, isotropic, lib, philox, library_path, set_math, set_scalar, get_scalar, set_stream, set_in_flight, get_in_flight, synchronize,
    tally,
)
