#!/usr/bin/env python3
"""Build container only: the GeoTIFF tiles that turtle_amd/synth.py writes for C5 (the ASTER-GDEM2
stand-ins) are read by the REAL reference -- through libtiff, dlopen()ed under a name that exists
here only behind a symlink: same shim as generate_files.py -- and give, node for node and
tile-box for tile-box, what its HGT reader gives for the same synthetic tile, alone and as a stack."""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import ref_ffi as R
    from turtle_amd import synth
    tmp = tempfile.mkdtemp(prefix="turtle_tif_")
    n = 1201
    a = R.RefMap.load(synth.write_geotiff(os.path.join(tmp, "tif"), 45, 3, n))
    b = R.RefMap.load(synth.write_hgt(os.path.join(tmp, "hgt"), 45, 3, n))
    rng = np.random.default_rng(1)
    for ix, iy in zip(rng.integers(0, n, 2000), rng.integers(0, n, 2000)):
        assert a.node(int(ix), int(iy)) == b.node(int(ix), int(iy)), (ix, iy)
    x, y = rng.uniform(2.99, 4.01, 4096), rng.uniform(44.99, 46.01, 4096)
    za, ia = a.elevation(x, y)
    zb, ib = b.elevation(x, y)
    assert np.array_equal(ia, ib) and np.array_equal(za, zb)
    for la, lo in ((45, 4), (46, 3), (46, 4)):
        synth.write_geotiff(os.path.join(tmp, "tif"), la, lo, n)
        synth.write_hgt(os.path.join(tmp, "hgt"), la, lo, n)
    sa, sb = R.RefStack(os.path.join(tmp, "tif"), 0), R.RefStack(os.path.join(tmp, "hgt"), 0)
    lat, lon = rng.uniform(44.9, 47.1, 4096), rng.uniform(2.9, 5.1, 4096)
    za, ia = sa.elevation(lat, lon)
    zb, ib = sb.elevation(lat, lon)
    assert np.array_equal(ia, ib) and np.array_equal(za, zb)
    print("the reference reads synth's GeoTIFF tiles as it reads its HGT tiles: nodes, elevations, stack")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    d = tempfile.mkdtemp(prefix="turtle_libs_")
    for stem in ("libtiff", "libpng"):
        hits = sorted(glob.glob(f"/usr/lib/x86_64-linux-gnu/{stem}*.so.*"))
        if hits:
            os.symlink(hits[0], os.path.join(d, stem + ".so"))
    env = dict(os.environ, LD_LIBRARY_PATH=d + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env))
