#!/usr/bin/env python3
"""G10: data files written BY the reference (turtle_map_dump) for the ingest
tests, plus what the reference reads back from them.

The reference dlopen()s "libtiff.so"/"libpng.so", which exist here only under
their versioned names: this script re-executes itself with a private directory
of symlinks on LD_LIBRARY_PATH (build container only).

  geotiff_utm.tif   201x201 int16 GeoTIFF, UTM-ish frame     -> geotiff.npz
  map_utm.png       201x201 PNG-16 + JSON header, "UTM 31N"   -> png.npz
                    (the reference's own map format, tests/test-turtle.c:67-95)
  geoid_small.grd   text grid in the EGM96 ww15mgh.grd layout  -> text.npz
  dem_small.asc     ESRI ASCII grid with a nodata value        -> text.npz
                    (both AUTHORED here -- the reference cannot write them --
                    and read back by the reference)
"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))


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
    from oracle import ref_ffi as R
    L = R.lib()
    rng = np.random.Generator(np.random.Philox(1010))
    i = np.arange(201, dtype=np.float64)
    nodes = np.rint(400.0 + 300.0 * np.sin(i / 13.0)[None, :] * np.cos(i / 29.0)[:, None]
                    - 2.0 * i[:, None])
    # a map whose 16-bit codes ARE the elevations (z0 = -32767, dz = 1), as a
    # GeoTIFF stores them [ref io/geotiff16.c:186-187, :230-238]
    m = R.RefMap.create(nodes, (495000.0, 497000.0), (5066000.0, 5068000.0),
                        (-32767.0, 32768.0))
    path = os.path.join(OUT, "geotiff_utm.tif")
    rc = L.turtle_map_dump(m.h, path.encode())
    assert rc == 0, R.errors()
    m.destroy()
    back = R.RefMap.load(path)
    info = R.MapInfo()
    L.turtle_map_meta(back.h, C.byref(info), None)
    ix = rng.integers(0, 201, 300)
    iy = rng.integers(0, 201, 300)
    node = np.array([back.node(int(a), int(b)) for a, b in zip(ix, iy)])
    qx = rng.uniform(494990.0, 497010.0, 1000)
    qy = rng.uniform(5065990.0, 5068010.0, 1000)
    qz, qin = back.elevation(qx, qy)
    back.destroy()
    np.savez_compressed(os.path.join(OUT, "geotiff.npz"), nodes=nodes, nx=info.nx, ny=info.ny,
                        x=np.array(info.x[:]), y=np.array(info.y[:]), z=np.array(info.z[:]),
                        ix=ix, iy=iy, node=node, qx=qx, qy=qy, qz=qz, qin=qin)
    print("geotiff_utm.tif", os.path.getsize(path), "bytes; meta", info.nx, info.ny, info.x[:],
          info.y[:], info.z[:], "errors:", R.errors())

    # ---- PNG-16 with the JSON "topography" header ----
    nodes = 350.0 + 300.0 * np.sin(i / 11.0)[None, :] * np.cos(i / 19.0)[:, None] + 0.5 * i[:, None]
    m = R.RefMap.create(nodes, (495000.0, 497000.0), (5066000.0, 5068000.0), (0.0, 1000.0),
                        "UTM 31N")
    path = os.path.join(OUT, "map_utm.png")
    rc = L.turtle_map_dump(m.h, path.encode())
    assert rc == 0, R.errors()
    m.destroy()
    back = R.RefMap.load(path)
    info = R.MapInfo()
    proj = C.c_char_p()
    L.turtle_map_meta(back.h, C.byref(info), C.byref(proj))
    projection = proj.value.decode()  # points into the map: read before destroying it
    node = np.array([back.node(int(a), int(b)) for a, b in zip(ix, iy)])
    qz, qin = back.elevation(qx, qy)
    back.destroy()
    np.savez_compressed(os.path.join(OUT, "png.npz"), nodes=nodes, nx=info.nx, ny=info.ny,
                        x=np.array(info.x[:]), y=np.array(info.y[:]), z=np.array(info.z[:]),
                        projection=np.array(projection), ix=ix, iy=iy, node=node,
                        qx=qx, qy=qy, qz=qz, qin=qin)
    print("map_utm.png", os.path.getsize(path), "bytes; meta", info.nx, info.ny, info.x[:],
          info.y[:], info.z[:], projection, "errors:", R.errors())

    # ---- text formats: we write the files, the reference reads them ----
    lat = np.arange(-2.0, 2.0 + 1e-9, 0.25)          # 17 rows
    lon = np.arange(10.0, 14.0 + 1e-9, 0.5)          # 9 columns
    und = 12.5 * np.sin(lon / 3.0)[None, :] - 7.0 * np.cos(lat * 1.3)[:, None]
    grd = os.path.join(OUT, "geoid_small.grd")
    with open(grd, "w") as f:
        f.write(f"  {lat[0]:.6f}   {lat[-1]:.6f}   {lon[0]:.6f}  {lon[-1]:.6f}   0.250000   0.500000\n")
        for row in und:                               # file order = rows iy = 0, 1, ...
            f.write(" ".join(f"{v:9.3f}" for v in row) + "\n")
    asc = os.path.join(OUT, "dem_small.asc")
    dem = np.rint(800.0 + 300.0 * np.sin(np.arange(12) / 2.0)[None, :]
                  * np.cos(np.arange(10) / 3.0)[:, None])
    dem[2, 3] = -9999.0
    with open(asc, "w") as f:
        f.write("ncols 12\nnrows 10\nxllcorner 3.0\nyllcorner 45.0\ncellsize 0.01\n"
                "NODATA_value -9999\n")
        for row in dem:                               # north row first
            f.write(" ".join(f"{v:.1f}" for v in row) + "\n")
    out = {}
    for tag, path in (("grd", grd), ("asc", asc)):
        back = R.RefMap.load(path)
        info = R.MapInfo()
        L.turtle_map_meta(back.h, C.byref(info), None)
        node = np.array([[back.node(a, b)[2] for a in range(info.nx)] for b in range(info.ny)])
        out.update({f"{tag}_nx": info.nx, f"{tag}_ny": info.ny, f"{tag}_x": np.array(info.x[:]),
                    f"{tag}_y": np.array(info.y[:]), f"{tag}_z": np.array(info.z[:]),
                    f"{tag}_node": node})
        back.destroy()
        print(os.path.basename(path), info.nx, info.ny, info.x[:], info.y[:], info.z[:])
    np.savez_compressed(os.path.join(OUT, "text.npz"), **out)
    print("errors:", R.errors())


if __name__ == "__main__":
    if os.environ.get("TURTLE_LINKS_READY") != "1":
        reexec_with_links()
    main()
