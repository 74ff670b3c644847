"""Host-side logic of libturtle_amd that needs no GPU: handles, file ingest,
tile directory, stepper configuration rules, the error convention.
Mirrors the CPU-checkable parts of tests/test-turtle.c (test_map :412-513,
test_io_hgt :1049-1089, test_stack :628-690, test_client :697-775,
test_stepper defaults :893-901)."""
import os
import re

import numpy as np
import pytest

import turtle_amd as TA
from turtle_amd import synth

HAS_GPU = TA.device_count() > 0


def test_map_create_meta_node_fill():
    nodes = np.zeros((201, 201))
    m = TA.Map.create(shape=(201, 201), x=(495000.0, 497000.0), y=(5066000.0, 5068000.0),
                      z=(0.0, 1000.0))
    meta = m.meta()
    assert (meta["nx"], meta["ny"]) == (201, 201)
    assert meta["x"] == (495000.0, 497000.0) and meta["y"] == (5066000.0, 5068000.0)
    assert meta["z"][0] == 0.0 and abs(meta["z"][1] - 1000.0) < 1e-9
    assert meta["encoding"] == "none" and meta["projection"] is None
    m.fill(3, 7, 1000.0)
    m.fill(4, 7, 500.0)
    x, y, z = m.node(3, 7)
    assert (x, y, z) == (495000.0 + 3 * 10.0, 5066000.0 + 7 * 10.0, 1000.0)
    # 16-bit quantisation: round((z-z0)/dz)*dz [ref map.c:41-51]
    dz = 1000.0 / 65535
    assert m.node(4, 7)[2] == 0.0 + round(500.0 / dz) * dz
    with pytest.raises(TA.TurtleError) as e:
        m.fill(201, 0, 1.0)
    assert e.value.name == "DOMAIN_ERROR" and "point is outside of map" in str(e.value)
    with pytest.raises(TA.TurtleError) as e:
        m.fill(0, 0, 1001.0)
    assert "elevation is outside of map span" in str(e.value)
    with pytest.raises(TA.TurtleError) as e:
        m.node(-1, 0)
    assert e.value.name == "DOMAIN_ERROR"
    m.destroy()
    assert nodes.shape == (201, 201)


def test_map_create_rejects_bad_input():
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.create(shape=(0, 4))
    assert e.value.name == "DOMAIN_ERROR" and "invalid input parameter(s)" in str(e.value)
    with pytest.raises(TA.TurtleError):
        TA.Map.create(shape=(4, 4), z=(1.0, 1.0))


def test_error_message_shape():
    """"{ <function> [#<code>], <file>:<line> } <text>" [ref error.c:108-138]"""
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.load("/nonexistent/N45E003.hgt")
    assert e.value.name == "PATH_ERROR"
    assert re.match(r"\{ turtle_map_load \[#10\], .*map\.c:\d+ \} could not open file "
                    r"`/nonexistent/N45E003.hgt'", str(e.value))
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.load("/tmp/whatever.xyz")
    assert e.value.name == "BAD_EXTENSION" and "unsuported file format `xyz'" in str(e.value)
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.load("/tmp/noextension")
    assert e.value.name == "BAD_EXTENSION" and "missing file extension" in str(e.value)


def test_hgt_ingest_decodes_once(tmp_path):
    """Big-endian, north row first on disk -> native int16, south row first in
    memory; turtle_map_node sees the same values as the reference's get_z
    [ref io/hgt.c:127-131]; name parsing [ref io/hgt.c:59-104]."""
    n = 1201
    nodes = synth.srtm_like_nodes(45, 3, n)
    nodes[5, 7] = -32768  # a void: NOT masked by the reference
    nodes[0, 0], nodes[n - 1, n - 1] = -12, 3210
    path = os.path.join(tmp_path, "N45E003.SRTMGL3.hgt")
    open(path, "wb").write(synth.hgt_bytes(nodes))
    m = TA.Map.load(path)
    meta = m.meta()
    assert (meta["nx"], meta["ny"]) == (n, n)
    assert meta["x"] == (3.0, 4.0) and meta["y"] == (45.0, 46.0)
    assert meta["z"] == (-32767.0, -32767.0 + 65535)
    for ix, iy in ((0, 0), (n - 1, n - 1), (7, 5), (600, 17), (1200, 0), (0, 1200)):
        x, y, z = m.node(ix, iy)
        assert z == float(nodes[iy, ix])
        assert x == 3.0 + ix * (1.0 / (n - 1)) and y == 45.0 + iy * (1.0 / (n - 1))
    m.fill(2, 2, 10.0)  # test-turtle.c:1080-1085 shape
    assert m.node(2, 2)[2] == 10.0
    m.destroy()
    # west/south names and the SRTMGL1 suffix
    p2 = os.path.join(tmp_path, "S12W077.SRTMGL1.hgt")
    open(p2, "wb").write(b"\0" * (2 * 3601 * 3601))
    m = TA.Map.load(p2)
    meta = m.meta()
    assert (meta["nx"], meta["x"], meta["y"]) == (3601, (-77.0, -76.0), (-12.0, -11.0))
    m.destroy()
    p3 = os.path.join(tmp_path, "N45E003.hgt")
    open(p3, "wb").write(b"\0" * 100)  # truncated
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.load(p3)
    assert e.value.name == "BAD_FORMAT" and "missing data" in str(e.value)
    p4 = os.path.join(tmp_path, "X45E003.hgt")
    open(p4, "wb").write(b"\0" * 100)
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.load(p4)
    assert e.value.name == "BAD_FORMAT" and "invalid hgt filename" in str(e.value)


def test_stack_directory_scan(tmp_path):
    d = os.path.join(tmp_path, "topo")
    for la, lo in ((45, 2), (46, 2), (45, 3)):
        synth.write_hgt(d, la, lo, 1201)
    open(os.path.join(d, "notes.txt"), "w").write("skipped")
    os.makedirs(os.path.join(d, "subdir"))
    s = TA.Stack(d, 3)
    s.clear()  # nothing loaded yet: fine
    s.destroy()
    with pytest.raises(TA.TurtleError) as e:
        TA.Stack(os.path.join(tmp_path, "missing"))
    assert e.value.name == "PATH_ERROR" and "could not access" in str(e.value)
    empty = os.path.join(tmp_path, "empty")
    os.makedirs(empty)
    s = TA.Stack(empty)
    s.load()  # an empty stack loads nothing [ref stack.c:260-261]
    s.destroy()


def test_stack_residency_budget(tmp_path):
    """stack_size, turtle_stack_load / clear [ref stack.c:150, :228-297]: tiles come
    in, in directory order, up to the limit the caller gave; without a limit all of
    them [ref tests/test-turtle.c:664-684: 3 of 4 with size 3, 4 with size 0]."""
    d = os.path.join(tmp_path, "grid")
    tiles = [(la, lo) for la in range(40, 45) for lo in range(5, 10) if (la, lo) != (42, 7)]
    for la, lo in tiles:
        synth.write_hgt(d, la, lo, 1201)
    s = TA.Stack(d, 0)
    assert s.resident == 0          # nothing is read at creation [ref stack.c:46-226]
    s.load()
    assert s.resident == len(tiles) == 24
    s.clear()
    assert s.resident == 0
    s.destroy()
    for size, expect in ((20, 20), (16, 16), (3, 3), (1, 1)):
        s = TA.Stack(d, size)
        s.load()
        assert s.resident == expect
        s.load()                    # full: a second call changes nothing
        assert s.resident == expect
        s.destroy()


def test_stack_lock_consistency_and_client(tmp_path):
    import ctypes as C
    LOCKER = C.CFUNCTYPE(C.c_int)
    calls = []

    @LOCKER
    def lock():
        calls.append("lock")
        return 0

    d = os.path.join(tmp_path, "topo")
    synth.write_hgt(d, 45, 3, 1201)
    with pytest.raises(TA.TurtleError) as e:
        TA.Stack(d, 0, lock, None)
    assert e.value.name == "BAD_ADDRESS" and "inconsistent lock & unlock" in str(e.value)
    L = TA.lib()
    plain = TA.Stack(d, 0)
    h = C.c_void_p()
    rc = L.turtle_client_create(C.byref(h), plain.h)  # [ref client.c:52-55]
    from turtle_amd import binding as Bn
    with pytest.raises(TA.TurtleError) as e:
        Bn._check(rc)
    assert e.value.name == "BAD_ADDRESS" and "stack has no lock" in str(e.value)
    plain.destroy()
    locked = TA.Stack(d, 0, lock, lock)
    Bn._check(L.turtle_client_create(C.byref(h), locked.h))
    locked.clear()
    assert calls == ["lock", "lock"]
    Bn._check(L.turtle_client_destroy(C.byref(h)))
    assert h.value is None
    st = TA.Stepper()
    st.add_stack(locked, 0.0)  # creates (and later destroys) its own client
    st.destroy()
    locked.destroy()


def test_in_flight_hint():
    assert TA.get_in_flight() == 1
    TA.set_in_flight(3)
    assert TA.get_in_flight() == 3
    TA.set_in_flight(0)                  # (anything below two is one)
    assert TA.get_in_flight() == 1


def test_stepper_defaults_and_setters():
    st = TA.Stepper()
    assert st.range == 1.0 and st.slope == 0.4 and st.resolution == 1e-2
    st.range, st.slope, st.resolution = 10.0, 1.0, 1e-3
    assert (st.range, st.slope, st.resolution) == (10.0, 1.0, 1e-3)
    m = TA.Map.create(shape=(2, 2))
    st.geoid_set(m)
    assert TA.lib().turtle_stepper_geoid_get(st.h) == m.h.value
    st.destroy()
    m.destroy()


def _layer_exists(st, layer):
    """turtle_stepper_position checks the layer index before any device work
    [ref stepper.c:883-886]: DOMAIN_ERROR means "no such layer"."""
    try:
        st.position_scalar(45.0, 3.0, 0.0, layer)
    except TA.TurtleError as e:
        return e.name != "DOMAIN_ERROR" or "no valid data" not in str(e)
    return True


@pytest.mark.skipif(HAS_GPU, reason="uses the no-device failure to stop before compute")
def test_stepper_layer_rules():
    m = TA.Map.create(shape=(2, 2))
    st = TA.Stepper()
    assert not _layer_exists(st, 0)
    st.add_flat(0.0)  # the first data creates layer 0 [ref stepper.c:394-396]
    assert _layer_exists(st, 0) and not _layer_exists(st, 1)
    st.add_layer()
    st.add_layer()  # an empty top layer is reused [ref stepper.c:366-368]
    st.add_map(m, 1.0)
    st.add_map(m, 2.0)
    assert _layer_exists(st, 1) and not _layer_exists(st, 2)
    assert not _layer_exists(st, -1)
    st.destroy()
    m.destroy()


@pytest.mark.skipif(HAS_GPU, reason="uses the no-device failure to stop before compute")
def test_stepper_clone():
    """turtle_amd_stepper_clone: the layers as they were added, the geoid and the settings; the
    clone borrows the same data and is destroyed on its own."""
    m, g = TA.Map.create(shape=(2, 2)), TA.Map.create(shape=(2, 2))
    st = TA.Stepper()
    st.add_flat(-1.0)
    st.add_layer()
    st.add_map(m, 1.0)
    st.add_map(m, 2.0)
    st.geoid_set(g)
    st.range, st.slope, st.resolution = 0.0, 0.5, 1e-3
    c = st.clone()
    assert (c.range, c.slope, c.resolution) == (0.0, 0.5, 1e-3)
    assert TA.lib().turtle_stepper_geoid_get(c.h) == g.h.value
    assert _layer_exists(c, 0) and _layer_exists(c, 1) and not _layer_exists(c, 2)
    c.add_layer()
    c.add_flat(5.0)                      # the clone grows on its own
    assert _layer_exists(c, 2) and not _layer_exists(st, 2)
    c.destroy()
    assert _layer_exists(st, 1) and st.slope == 0.5
    empty = TA.Stepper().clone()         # nothing added yet: nothing to copy
    assert not _layer_exists(empty, 0)
    empty.destroy()
    st.destroy()
    m.destroy()
    g.destroy()


def test_projection_names():
    """The name parser [ref projection.c:98-171], quirks included."""
    for name in ("Lambert I", "Lambert II", "Lambert IIe", "Lambert III", "Lambert IV",
                 "Lambert 93", "UTM 31N", "UTM 3.5N", "UTM 19S", "  UTM 31N"):
        p = TA.Projection(name)
        assert p.name == name
        p.destroy()
    for bad, text in (("", "missing projection specifier"), ("Mercator", "invalid projection"),
                      ("UTM", "invalid UTM specifier"), ("UTM 31X", "invalid UTM hemisphere"),
                      ("Lambert V", "invalid projection"), ("UTM 3.5", "invalid extended UTM")):
        with pytest.raises(TA.TurtleError) as e:
            TA.Projection(bad)
        assert e.value.name == "BAD_PROJECTION" and text in str(e.value), bad
    m = TA.Map.create(shape=(3, 3), x=(0, 10), y=(0, 10), z=(0, 1), projection="Lambert 93")
    assert m.meta()["projection"] == "Lambert 93"
    assert TA.lib().turtle_map_projection(m.h) != 0
    m.destroy()
    m = TA.Map.create(shape=(3, 3))
    assert m.meta()["projection"] is None and not TA.lib().turtle_map_projection(m.h)
    m.destroy()
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.create(shape=(3, 3), projection="nowhere")
    assert e.value.name == "BAD_PROJECTION"


def _write_tiff(path, nodes_n2s, byteorder, x0, y_top, dx, dy, compression=1, rows_per_strip=None):
    """A minimal GeoTIFF-16 writer for the ingest tests (baseline TIFF, strips)."""
    import struct
    e = "<" if byteorder == "II" else ">"
    ny, nx = nodes_n2s.shape
    rps = rows_per_strip or ny
    n_strips = (ny + rps - 1) // rps
    data = nodes_n2s.astype(e + "i2").tobytes()
    entries = []
    blob = b""
    base = 8 + len(data)

    def extra(payload):
        nonlocal blob
        off = base + len(blob)
        blob += payload
        return off

    offs = [8 + 2 * nx * rps * k for k in range(n_strips)]
    cnts = [2 * nx * min(rps, ny - rps * k) for k in range(n_strips)]
    entries += [(256, 4, 1, nx), (257, 4, 1, ny), (258, 3, 1, 16), (259, 3, 1, compression),
                (262, 3, 1, 1), (277, 3, 1, 1), (278, 4, 1, rps)]
    if n_strips == 1:
        entries += [(273, 4, 1, offs[0]), (279, 4, 1, cnts[0])]
    else:
        entries += [(273, 4, n_strips, extra(struct.pack(e + f"{n_strips}I", *offs))),
                    (279, 4, n_strips, extra(struct.pack(e + f"{n_strips}I", *cnts)))]
    entries += [(33550, 12, 3, extra(struct.pack(e + "3d", dx, dy, 0.0))),
                (33922, 12, 6, extra(struct.pack(e + "6d", 0, 0, 0, x0, y_top, 0)))]
    entries.sort()
    ifd_at = base + len(blob)
    ifd = struct.pack(e + "H", len(entries))
    for tag, typ, cnt, val in entries:
        if typ == 3:
            ifd += struct.pack(e + "HHIHH", tag, typ, cnt, val, 0)
        else:
            ifd += struct.pack(e + "HHII", tag, typ, cnt, val)
    ifd += struct.pack(e + "I", 0)
    with open(path, "wb") as f:
        f.write(byteorder.encode() + struct.pack(e + "HI", 42, ifd_at) + data + blob + ifd)


def test_geotiff_ingest(tmp_path):
    """Native GeoTIFF-16 reader against a file WRITTEN BY THE REFERENCE
    (tests/golden/geotiff_utm.tif, see generate_files.py) and what the reference
    read back from it [ref io/geotiff16.c:165-258]; plus big-endian, multi-strip
    and refused layouts from a writer of our own."""
    here = os.path.dirname(os.path.abspath(__file__))
    g = dict(np.load(os.path.join(here, "golden", "geotiff.npz")))
    m = TA.Map.load(os.path.join(here, "golden", "geotiff_utm.tif"))
    meta = m.meta()
    assert (meta["nx"], meta["ny"]) == (int(g["nx"]), int(g["ny"]))
    assert meta["x"] == tuple(g["x"]) and meta["y"] == tuple(g["y"]) and meta["z"] == tuple(g["z"])
    assert meta["encoding"] == "tif" and meta["projection"] is None
    for ix, iy, ref in zip(g["ix"], g["iy"], g["node"]):
        assert m.node(int(ix), int(iy)) == tuple(ref)
    m.destroy()
    nodes = g["nodes"]  # south -> north
    for order, rps in (("MM", None), ("II", 7), ("MM", 1)):
        p = os.path.join(tmp_path, f"t_{order}_{rps}.tif")
        _write_tiff(p, nodes[::-1], order, 495000.0, 5068000.0, 10.0, 10.0, rows_per_strip=rps)
        m = TA.Map.load(p)
        assert m.meta()["x"] == (495000.0, 497000.0) and m.meta()["y"] == (5066000.0, 5068000.0)
        for ix, iy in ((0, 0), (200, 200), (17, 133), (133, 17)):
            assert m.node(ix, iy)[2] == nodes[iy, ix]
        m.destroy()
    p = os.path.join(tmp_path, "lzw.tif")
    _write_tiff(p, nodes[::-1], "II", 0.0, 0.0, 1.0, 1.0, compression=5)
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.load(p)
    assert e.value.name == "BAD_FORMAT"
    open(os.path.join(tmp_path, "junk.tif"), "wb").write(b"not a tiff at all")
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.load(os.path.join(tmp_path, "junk.tif"))
    assert e.value.name == "BAD_FORMAT"


def test_png_map_ingest(tmp_path):
    """Native reader of the reference's PNG-16 map format against a file
    WRITTEN BY THE REFERENCE (tests/golden/map_utm.png) and what the reference
    read back from it [ref io/png16.c:183-448]: JSON header with C99 hex floats,
    projection name, big-endian samples, north row first."""
    here = os.path.dirname(os.path.abspath(__file__))
    g = dict(np.load(os.path.join(here, "golden", "png.npz")))
    m = TA.Map.load(os.path.join(here, "golden", "map_utm.png"))
    meta = m.meta()
    assert (meta["nx"], meta["ny"]) == (int(g["nx"]), int(g["ny"]))
    assert meta["x"] == tuple(g["x"]) and meta["y"] == tuple(g["y"]) and meta["z"] == tuple(g["z"])
    assert meta["encoding"] == "png" and meta["projection"] == str(g["projection"]) == "UTM 31N"
    for ix, iy, ref in zip(g["ix"], g["iy"], g["node"]):
        assert m.node(int(ix), int(iy)) == tuple(ref)
    m.destroy()
    bad = os.path.join(tmp_path, "bad.png")
    open(bad, "wb").write(b"\\x89PNG\\r\\n\\x1a\\n" + b"\\0" * 40)
    with pytest.raises(TA.TurtleError) as e:
        TA.Map.load(bad)
    assert e.value.name == "BAD_FORMAT"


def test_text_grid_ingest():
    """.grd and .asc readers against what the REFERENCE read from the same
    files (tests/golden/text.npz): meta incl. the 16-bit quantisation range
    found by the reference's own scan, and every node, bit for bit
    [ref io/grd.c:45-157, io/asc.c:45-150]."""
    here = os.path.dirname(os.path.abspath(__file__))
    g = dict(np.load(os.path.join(here, "golden", "text.npz")))
    for tag, name in (("grd", "geoid_small.grd"), ("asc", "dem_small.asc")):
        m = TA.Map.load(os.path.join(here, "golden", name))
        meta = m.meta()
        assert (meta["nx"], meta["ny"]) == (int(g[tag + "_nx"]), int(g[tag + "_ny"]))
        assert meta["x"] == tuple(g[tag + "_x"]) and meta["y"] == tuple(g[tag + "_y"])
        assert meta["z"] == tuple(g[tag + "_z"]) and meta["encoding"] == tag
        node = np.array([[m.node(a, b)[2] for a in range(meta["nx"])] for b in range(meta["ny"])])
        assert np.array_equal(node, g[tag + "_node"])
        m.destroy()


def _png_decode(path):
    """A PNG reader of the test's own (zlib + the five scan-line filters):
    width, height, the tEXt payloads, the 16-bit samples (rows as stored)."""
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    at, texts, idat, head = 8, [], b"", None
    while at < len(raw):
        n, kind = struct.unpack(">I4s", raw[at:at + 8])
        body = raw[at + 8:at + 8 + n]
        assert struct.unpack(">I", raw[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body)
        if kind == b"IHDR":
            head = struct.unpack(">IIBBBBB", body)
        elif kind == b"tEXt":
            texts.append(body)
        elif kind == b"IDAT":
            idat += body
        at += 12 + n
    w, h, depth, colour, _, _, interlace = head
    assert (depth, colour, interlace) == (16, 0, 0)
    data = bytearray(zlib.decompress(idat))
    stride, bpp = 2 * w, 2
    rows, prev = [], bytearray(stride)
    for i in range(h):
        f = data[i * (stride + 1)]
        cur = bytearray(data[i * (stride + 1) + 1:(i + 1) * (stride + 1)])
        for k in range(stride):
            a = cur[k - bpp] if k >= bpp else 0
            b = prev[k]
            c = prev[k - bpp] if k >= bpp else 0
            if f == 1:
                cur[k] = (cur[k] + a) & 255
            elif f == 2:
                cur[k] = (cur[k] + b) & 255
            elif f == 3:
                cur[k] = (cur[k] + (a + b) // 2) & 255
            elif f == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                cur[k] = (cur[k] + (a if (pa <= pb and pa <= pc) else (b if pb <= pc else c))) & 255
        rows.append(np.frombuffer(bytes(cur), dtype=">u2").astype(np.uint16))
        prev = cur
    return w, h, texts, np.array(rows)


def test_map_dump(tmp_path):
    """turtle_map_dump [ref map.c:165-180].  The PNG written from a map that was
    loaded from a file WRITTEN BY THE REFERENCE (tests/golden/map_utm.png) holds
    the same samples and, byte for byte, the same "Comment" header as that file
    [ref png16.c:456-545]; both formats read back as the map that was dumped;
    the reference's refusals [ref geotiff16.c:266-278, io.c:101-103]."""
    here = os.path.dirname(os.path.abspath(__file__))
    theirs = os.path.join(here, "golden", "map_utm.png")
    m = TA.Map.load(theirs)
    ours = os.path.join(tmp_path, "again.png")
    m.dump(ours)
    w0, h0, t0, s0 = _png_decode(theirs)
    w1, h1, t1, s1 = _png_decode(ours)
    assert (w0, h0) == (w1, h1) and np.array_equal(s0, s1)
    assert t0 == t1 and t1[0].startswith(b"Comment\x00{\"topography\" : {\"x0\" : 0x")
    back = TA.Map.load(ours)
    assert back.meta() == m.meta()
    for ix, iy in ((0, 0), (3, 7), (m.meta()["nx"] - 1, m.meta()["ny"] - 1)):
        assert back.node(ix, iy) == m.node(ix, iy)
    back.destroy()
    with pytest.raises(TA.TurtleError) as e:
        m.dump(os.path.join(tmp_path, "projected.tif"))   # z scale and projection
    assert e.value.name == "BAD_FORMAT" and "unsupported z scale" in str(e.value)
    m.destroy()

    # an int16-scaled geodetic map [ref tests/test-turtle.c:1093-1135], asymmetric on
    # purpose: a writer that flipped the rows would be caught
    nx, ny = 31, 17
    t = TA.Map.create(None, (3.0, 4.0), (45.0, 46.0), (-32767.0, 32768.0), shape=(ny, nx))
    for iy in range(ny):
        for ix in range(nx):
            t.fill(ix, iy, float(7 * iy - 3 * ix))
    for ext in ("tif", "png"):
        p = os.path.join(tmp_path, "int16." + ext)
        t.dump(p)
        back = TA.Map.load(p)
        mb, mt = back.meta(), t.meta()
        assert (mb["nx"], mb["ny"], mb["z"]) == (nx, ny, mt["z"])
        assert np.allclose(mb["x"], mt["x"], atol=1e-12) and np.allclose(mb["y"], mt["y"], atol=1e-12)
        for iy in range(ny):
            for ix in range(0, nx, 5):
                assert back.node(ix, iy)[2] == 7 * iy - 3 * ix
        back.destroy()
    for name, code in (("x.hgt", "BAD_FORMAT"), ("x.grd", "BAD_FORMAT"), ("x.asc", "BAD_FORMAT"),
                       ("x.jpg", "BAD_EXTENSION"), ("noextension", "BAD_EXTENSION")):
        with pytest.raises(TA.TurtleError) as e:
            t.dump(os.path.join(tmp_path, name))
        assert e.value.name == code
    with pytest.raises(TA.TurtleError) as e:
        t.dump(os.path.join(tmp_path, "no", "such", "dir.png"))
    assert e.value.name == "PATH_ERROR" and "turtle_map_dump" in str(e.value)
    t.destroy()
