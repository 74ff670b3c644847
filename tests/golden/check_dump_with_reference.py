#!/usr/bin/env python3
"""Build-container check (not a test: it needs the compiled reference and its
dlopen()ed libpng / libtiff): maps dumped by turtle_amd's turtle_map_dump are read
back by the REFERENCE's turtle_map_load, node for node and with the same meta data.

Run: python tests/golden/check_dump_with_reference.py   (prints OK lines)"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def reexec_with_links():
    d = tempfile.mkdtemp(prefix="turtle_links_")
    for stem in ("libtiff", "libpng"):
        hits = sorted(glob.glob(f"/usr/lib/x86_64-linux-gnu/{stem}*.so.*"))
        if hits:
            os.symlink(hits[0], os.path.join(d, stem + ".so"))
    env = dict(os.environ, LD_LIBRARY_PATH=d + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               TURTLE_LINKS_READY="1")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=env))


def main():
    import ctypes as C
    import turtle_amd as TA
    from oracle import ref_ffi as R
    L = R.lib()
    tmp = tempfile.mkdtemp(prefix="turtle_dump_")
    ny, nx = 23, 41
    iy, ix = np.mgrid[0:ny, 0:nx]
    cases = {
        # name: (nodes, x, y, z, projection, formats)
        "int16": (np.rint(300.0 * np.sin(ix / 5.0) + 11.0 * iy - 400.0), (3.0, 4.0), (45.0, 46.0),
                  (-32767.0, 32768.0), None, ("png", "tif")),
        "scaled": (np.rint(50.0 * np.cos(iy / 3.0) + 7.0 * ix + 500.0), (495000.0, 497000.0),
                   (5066000.0, 5068000.0), (0.0, 2000.0), "UTM 31N", ("png",)),
    }
    for name, (nodes, x, y, z, proj, formats) in cases.items():
        m = TA.Map.create(nodes, x, y, z, projection=proj)
        for ext in formats:
            path = os.path.join(tmp, f"{name}.{ext}")
            m.dump(path)
            back = R.RefMap.load(path)
            info = R.MapInfo()
            pname = C.c_char_p()
            L.turtle_map_meta(back.h, C.byref(info), C.byref(pname))
            mine = m.meta()
            assert (info.nx, info.ny) == (nx, ny)
            assert np.allclose(tuple(info.x), mine["x"], rtol=0, atol=1e-9 * max(1, abs(x[1])))
            assert np.allclose(tuple(info.y), mine["y"], rtol=0, atol=1e-9 * max(1, abs(y[1])))
            assert tuple(info.z) == mine["z"], (tuple(info.z), mine["z"])
            assert (pname.value.decode() if pname.value else None) == proj
            worst = 0.0
            for j in range(ny):
                for i in range(nx):
                    zz = C.c_double()
                    L.turtle_map_node(back.h, i, j, None, None, C.byref(zz))
                    worst = max(worst, abs(zz.value - m.node(i, j)[2]))
            assert worst == 0.0, worst
            back.destroy()
            print(f"OK  {name}.{ext}: the reference reads back {nx}x{ny} nodes and the meta data unchanged")
        m.destroy()


if __name__ == "__main__":
    if os.environ.get("TURTLE_LINKS_READY") != "1":
        reexec_with_links()
    main()
