#!/usr/bin/env python3
"""bench.py -- ray-steps/s of the stepper path on the BASELINE workloads.

Headline (BASELINE.json configs[1], "C2"): 1 M rays per GPU through one
synthetic 3601x3601 SRTMGL1 tile (turtle_amd.synth), rays from the common
recipe of SURVEY.md 8d: origin uniform over the tile 500 m above ground,
azimuth U[0,360), elevation U[-10,-1] deg, slope 0.4, resolution 1e-2, traced
until the medium changes (hit or exit).  One "step" of this benchmark = one
turtle_stepper_trace_n call over the whole batch, inputs resident in HBM.

With no --workload (what the driver runs) the line also carries, under "also",
one measured pass each of C3 (10 M rays, 4x4 mosaic through a stack; and again with
8 of its 16 tiles resident), C4 (12.5 M rays: one rank of the 8-GPU config) and C5 (10 M
scattering rays x 256 single steps over a 10x10 mosaic), and every workload
reports a parity count: the outputs the GPU just produced against the CPU
restatement (oracle/) on its first 100 000 rays.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), tile
replicated, rays block-sharded (rank r draws its own Philox block), no
data-path collective; the per-step tally (hit counts + 1024-bin path-length
histogram, uint64) is all-reduced.  Weak scaling: per-GPU work is fixed.
--workload c4 is BASELINE configs[3]: 12.5 M rays per GPU (100 M on 8).

Output (rank 0): one full JSON record per workload as it ends (`"leg": name`), then --
the LAST line, the one the driver parses, kept below 6 KB -- the headline with one short
object per further workload under "also".  The headline's `value`, `ms_per_step`,
`kernel.ms` and `roofline` all come from ONE timed region: `--steps` passes, one batch
of rays on the GPU at a time (BASELINE's "1 M parallel rays"); what several batches in
flight on streams of their own reach is measured in a second region and reported
beside it (`in_flight`).
"""
from __future__ import annotations

import argparse
import concurrent.futures
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
VALU_CLOCK_HZ = 2.4e9       # MI355X peak engine clock (MI355X_MICROARCH.md)
TRACE_KERNEL = "k_trace (phase A: every ray by the closed form up to its hand-over step, 32 or 512; phase B: the rest on their lines) + k_cross (every crossing, packed)"
STEP_KERNELS = "k_step_fast + k_bisect per generation (single steps, directions drawn in the kernel)"
STEP_N_KERNELS = "k_step_fast + k_bisect per generation (turtle_stepper_step_n, TURTLE_AMD_STEP_RESUME, the caller's directions)"
WALK_N_KERNELS = "k_step_fast + k_bisect per generation (turtle_stepper_walk_n: one double of state a ray, the caller's directions)"
WALK_KERNEL = "k_walk (a ray's whole walk in one launch: state in registers, directions drawn in the kernel)"
SEED = 0x5EED2026
PARITY_RAYS = 100_000

WORKLOADS = {
    # name: (tiles (lat0, lon0, nlat, nlon), through a stack?, default rays/GPU, text)
    "c2": ((45, 3, 1, 1), False, 1_000_000,
           "C2: 1M rays/GPU, one 3601x3601 SRTMGL1 tile, trace to first boundary"),
    "c3": ((45, 3, 4, 4), True, 10_000_000,
           "C3: 10M rays/GPU, 4x4 mosaic of 3601x3601 tiles through a stack (all "
           "tiles resident in HBM), trace to first boundary"),
    "c4": ((45, 3, 1, 1), False, 12_500_000,
           "C4: 12.5M rays/GPU (100M on 8 GPUs, block-sharded), one 3601x3601 SRTMGL1 tile "
           "replicated, trace to first boundary, RCCL reduce of hits + path-length histogram"),
    "c5": ((40, 0, 10, 10), True, 10_000_000,
           "C5: 10M scattering rays/GPU, {steps} single steps each with a new isotropic "
           "direction per step (Philox(ray, step)), 10x10 mosaic of 3601x3601 tiles"),
}


def device_source_hash():
    """What the committed counter measurements are keyed on: the kernels' source."""
    h = hashlib.sha256()
    for name in ("device.hip", "internal.h"):
        with open(os.path.join(ROOT, "turtle_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(workload, rays, math):
    """HBM-side bytes per launch from the committed PMC passes (profiles/*_pmc.json,
    written by scripts/profile_round.sh): counters cannot be read from inside
    the timed process, so the figure is the offline measurement of the same
    command BY THE SAME KERNEL SOURCE (its hash is in the file), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if (d.get("workload"), d.get("rays_per_gpu"), d.get("math")) != (workload, rays, math):
            continue
        if d.get("source_hash") != device_source_hash():
            continue
        valu = sum(k.get("SQ_INSTS_VALU", 0.0) for k in d.get("kernels", {}).values())
        # (passes without FETCH_SIZE / WRITE_SIZE: no figure, not zero)
        return d.get("traffic_bytes_per_launch") or None, os.path.basename(path), valu or None
    return None, None, None


def host_cores():
    """CPU threads this process may really use: the cgroup quota if there is
    one (a GPU box shows every core of the host but grants a share), else the
    affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def node_barrier(env):
    """every rank of the job (they share the node: one process per GPU)"""
    if env["world"] > 1:
        import torch.distributed as dist
        dist.barrier()


class Terrain:
    """The synthetic tiles of a workload as files -- SRTM's .hgt, or (C5: `fmt` "tif")
    the GeoTIFF-16 tiles ASTER-GDEM2 ships -- written ONCE per node (local rank 0 writes,
    the others wait at a barrier) and loaded by every rank through the C API; rank 0 keeps
    the node arrays for the CPU checker."""

    def __init__(self, TA, tiles, use_stack, env, stack_size=0, fmt="hgt"):
        from turtle_amd import synth
        self.tiles, self.use_stack, self.env, self.fmt = tiles, use_stack, env, fmt
        lat0, lon0, nlat, nlon = tiles
        self.cells = [(lat0 + i, lon0 + j) for i in range(nlat) for j in range(nlon)]
        # one directory for the job: every rank derives the same name
        job = os.environ.get("MASTER_PORT", str(os.getpid())) if env["world"] > 1 else str(os.getpid())
        self.tmp = os.path.join(tempfile.gettempdir(), f"turtle_bench_{job}_{lat0}_{lon0}_{nlat}x{nlon}_{fmt}")
        self.nodes, self.hgt_dir = {}, None
        name = synth.geotiff_name if fmt == "tif" else synth.hgt_name
        if env["rank"] == 0:
            shutil.rmtree(self.tmp, ignore_errors=True)
            os.makedirs(self.tmp)

            def make(cell):
                nodes = synth.srtm_like_nodes(*cell)
                with open(os.path.join(self.tmp, name(*cell)), "wb") as f:
                    if fmt == "tif":
                        step = 1.0 / (synth.HGT_N - 1)
                        f.write(synth.geotiff_bytes(nodes, float(cell[1]), float(cell[0] + 1), step, step))
                    else:
                        f.write(synth.hgt_bytes(nodes))
                return cell, nodes
            with concurrent.futures.ThreadPoolExecutor(max(1, min(8, host_cores()))) as pool:
                self.nodes = dict(pool.map(make, self.cells))
        node_barrier(env)
        self.stepper = TA.Stepper()
        if use_stack:
            self.handle = TA.Stack(self.tmp, stack_size)
            self.handle.load()      # up to stack_size tiles (0: all of them: no paging rounds)
            self.stepper.add_stack(self.handle, 0.0)
        else:
            self.handle = TA.Map.load(os.path.join(self.tmp, name(lat0, lon0)))
            self.stepper.add_map(self.handle, 0.0)
        self.lat_range = (float(lat0), float(lat0 + nlat))
        self.lon_range = (float(lon0), float(lon0 + nlon))

    def hgt_files(self):
        """the same tiles as .hgt files, for the reference on the CPU: its GeoTIFF reader
        dlopen()s "libtiff.so", a name this image has only with a version behind it"""
        from turtle_amd import synth
        if self.fmt == "hgt":
            return self.tmp
        if self.hgt_dir is None:
            self.hgt_dir = self.tmp + "_as_hgt"
            shutil.rmtree(self.hgt_dir, ignore_errors=True)
            os.makedirs(self.hgt_dir)

            def make(cell):
                with open(os.path.join(self.hgt_dir, synth.hgt_name(*cell)), "wb") as f:
                    f.write(synth.hgt_bytes(self.nodes[cell]))
            with concurrent.futures.ThreadPoolExecutor(max(1, min(8, host_cores()))) as pool:
                list(pool.map(make, self.cells))
        return self.hgt_dir

    def oracle(self):
        """oracle/ geometry of the same terrain (the checker, never the product)"""
        from oracle import ffi as O
        lat0, lon0, nlat, nlon = self.tiles
        grids, table = [], []
        for i in range(nlat):
            for j in range(nlon):
                table.append(len(grids))
                grids.append(O.hgt_grid(lat0 + i, lon0 + j, self.nodes[(lat0 + i, lon0 + j)]))
        if self.use_stack:
            stack = dict(lat0=float(lat0), lon0=float(lon0), dlat=1.0, dlon=1.0, nlat=nlat,
                         nlon=nlon, tile=np.array(table, dtype=np.int32))
            return O.OracleGeometry(grids=grids, stacks=[stack], layers=[[(O.STACK, 0, 0.0)]])
        return O.OracleGeometry(grids=grids, layers=[[(O.MAP, 0, 0.0)]])

    def close(self):
        self.stepper.destroy()
        self.handle.destroy()
        node_barrier(self.env)      # nobody reads the files any more
        if self.env["rank"] == 0:
            shutil.rmtree(self.tmp, ignore_errors=True)
            if self.hgt_dir is not None:
                shutil.rmtree(self.hgt_dir, ignore_errors=True)


def parity_counts(index, length, ref_index, ref_length, steps=None, ref_steps=None):
    """identical medium, |dL| / L <= 1e-6; the rays that miss are COUNTED, and so are
    the rays whose step count differs (a ray grazing a surface within 1e-9 m may take one more
    or less)"""
    index, ref_index = np.asarray(index), np.asarray(ref_index)
    flipped = index[:, 0] != ref_index[:, 0]
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(np.asarray(length) - ref_length) / np.maximum(np.abs(ref_length), 1e-300)
    rel = np.where(np.isfinite(rel), rel, 0.0)
    out = {"rays": int(index.shape[0]), "medium_mismatch": int(flipped.sum()),
           "beyond_1e-6": int((~flipped & (rel > 1e-6)).sum()),
           "max_rel_path_length": float(rel[~flipped].max(initial=0.0)),
           "checker": "oracle/ C restatement (reference arithmetic, exact transform), same rays"}
    if steps is not None and ref_steps is not None:
        ds = np.abs(np.asarray(steps).astype(np.int64) - np.asarray(ref_steps).astype(np.int64))
        out["step_count_mismatch"] = int((ds != 0).sum())
        out["max_step_count_difference"] = int(ds.max(initial=0))
    return out


def cpu_baseline(terrain, pos, d, cores):
    """The reference itself (oracle/_ref, kind "reference") where its build
    travelled with the repo and the terrain is a single map, else the CPU
    restatement (oracle/, kind "port"): a bounded sample of the same rays, all
    host cores, exact transform (range 0) and the reference's default local-linear
    approximation (range 1).  Returns the timings and the range-0 results."""
    from oracle import ref_ffi as R
    from turtle_amd import synth
    geo = terrain.oracle()
    out = {}
    n_rays = pos.shape[0]
    geo.trace(pos[:20000], d[:20000], local_range=0.0, threads=cores)   # warm up
    res = None
    for tag, rng in (("range1", 1.0), ("range0", 0.0)):
        t0 = time.perf_counter()
        res = geo.trace(pos, d, local_range=rng, threads=cores)
        dt = time.perf_counter() - t0
        out[tag] = res["total_steps"] / dt
    t0 = time.perf_counter()
    one = geo.trace(pos[: max(1, n_rays // cores)], d[: max(1, n_rays // cores)],
                    local_range=0.0, threads=1)
    out["one_core"] = one["total_steps"] / (time.perf_counter() - t0)
    if R.driver_available():
        if terrain.use_stack:
            # the reference's threaded pattern [ref examples/example-pthread.c:66-125]: one
            # stack with lock / unlock shared by the threads, a client per worker
            path = terrain.hgt_files()
            run = lambda p, dd, rng, th: R.stack_run(path, p, dd, local_range=rng, threads=th)
        else:
            lat0, lon0 = terrain.tiles[:2]
            path = os.path.join(terrain.tmp, synth.hgt_name(lat0, lon0))
            run = lambda p, dd, rng, th: R.trace_map(path, p, dd, local_range=rng, threads=th)
        run(pos[:20000], d[:20000], 1.0, cores)   # warm up
        for tag, rng in (("ref_range1", 1.0), ("ref_range0", 0.0)):
            a = run(pos, d, rng, cores)
            out[tag] = a["total_steps"] / a["seconds"]
        out["ref_equal"] = bool(np.array_equal(a["index"], res["index"]) and
                                np.array_equal(a["length"], res["length"]))
        a = run(pos[: max(1, n_rays // cores)], d[: max(1, n_rays // cores)], 1.0, 1)
        out["ref_one_core"] = a["total_steps"] / a["seconds"]
        out["ref_how"] = ("a locked stack shared, a client per thread" if terrain.use_stack else
                          "a stepper per thread over a shared map")
    return out, res


def cpu_baseline_entry(cpu, cores, n_rays):
    """`sample` says what was timed in under 200 characters; the other figures are fields"""
    port = {"port_range0": cpu["range0"], "port_range1": cpu["range1"], "port_one_core": cpu["one_core"]}
    if "ref_range1" in cpu:
        return {"value": cpu["ref_range1"], "unit": "ray-steps/s", "cores": cores, "kind": "reference",
                "sample": f"{n_rays} rays of the same recipe through the reference itself (oracle/_ref), "
                          f"{cores} pthreads, {cpu.get('ref_how', 'one stepper per thread')}, its default "
                          f"local range of 1 m",
                "range0": cpu["ref_range0"], "one_core": cpu["ref_one_core"],
                "equal_to_the_restatement_bit_for_bit": cpu["ref_equal"], **port}
    return {"value": cpu["range0"], "unit": "ray-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_rays} rays of the same recipe, oracle/ C restatement, {cores} pthreads, exact "
                      f"transform (range 0)", **port}


def run_workload(name, args, env, headline):
    """One workload: terrain, rays, warm-up, the timed passes, the numbers.
    `headline`: args.steps timed passes with the all-reduce of the tally between
    them and the barriers of the contract; else one timed pass."""
    import torch
    import torch.distributed as dist
    import turtle_amd as TA
    from turtle_amd import sharding

    world, rank, dev = env["world"], env["rank"], env["dev"]
    tiles, use_stack, default_rays, text = WORKLOADS[name]
    terrain = Terrain(TA, tiles, use_stack, env, args.stack_size if use_stack else 0,
                      fmt=("tif" if (name == "c5" and args.tiles == "auto") or args.tiles == "tif" else "hgt"))
    stepper = terrain.stepper
    n_block = args.rays or default_rays
    blocks = max(1, args.blocks)
    n = n_block * blocks
    # ---- rays: rank r draws block(s) r of the Philox stream; set-up on the GPU ----
    parts = [sharding.rank_rays(n_block, rank * blocks + b, terrain.lat_range, terrain.lon_range)
             for b in range(blocks)]
    lat, lon, az, el = (np.concatenate([p[i] for p in parts]) for i in range(4))
    if args.sort > 0:
        nb = args.sort
        bx = np.minimum((nb * (lon - terrain.lon_range[0]) /
                         (terrain.lon_range[1] - terrain.lon_range[0])).astype(int), nb - 1)
        by = np.minimum((nb * (lat - terrain.lat_range[0]) /
                         (terrain.lat_range[1] - terrain.lat_range[0])).astype(int), nb - 1)
        order = np.argsort(by * nb + bx, kind="stable")
        lat, lon, az, el = lat[order], lon[order], az[order], el[order]
    t_lat, t_lon, t_az, t_el = (torch.as_tensor(v, device=dev) for v in (lat, lon, az, el))
    pos0, di = stepper.position(t_lat, t_lon, 500.0)
    direction = TA.ecef_from_horizontal(t_lat, t_lon, t_az, t_el)
    assert int((di != 0).sum()) == 0
    del t_lat, t_lon, t_az, t_el
    scatter = name == "c5"
    gens = getattr(args, "step_n", 0) if scatter else 0     # > 0: the walk through turtle_stepper_step_n
    compact = bool(getattr(args, "compact", False))         # ... through turtle_stepper_walk_n instead
    if gens:
        args.scatter_steps = gens
    # ---- batches in flight: a stepper, a stream and a set of arrays each (one stepper is
    # one stream of calls, as one turtle_stepper is one thread's in the reference) ----
    width = args.in_flight if args.in_flight > 0 else (1 if (scatter or (use_stack and args.stack_size)) else 3)
    main_stream = torch.cuda.current_stream()
    n_media, n_bins, lmax = 2, 1024, 65536.0
    t_hits, t_hist, t_steps, t_size = sharding.tally_layout(n_media, n_bins)

    class Flight:
        def __init__(self, k):
            self.stepper, self.stream = stepper, main_stream
            if k > 0:
                self.stepper = stepper.clone()
                self.stream = torch.cuda.Stream(device=dev)
            self.pos = torch.empty_like(pos0)
            self.index = torch.empty((n, 2), dtype=torch.int32, device=dev)
            self.length = torch.empty(n, dtype=torch.float64, device=dev)
            self.nsteps = torch.empty(n, dtype=torch.int32, device=dev)
            self.tally = torch.zeros(t_size, dtype=torch.int64, device=dev)
            self.gen_ms = []

        def enter(self):
            torch.cuda.set_stream(self.stream)
            TA.set_stream(self.stream)

    flights = [Flight(k) for k in range(width)]
    pos, index, length, nsteps, tally = (flights[0].pos, flights[0].index, flights[0].length,
                                         flights[0].nsteps, flights[0].tally)
    first_ray = rank * n
    walk_state = {}

    dirs = []
    if gens:
        # the CALLER's directions, as a Monte-Carlo would hand them over: drawn beforehand on the
        # device (the library's Philox, so that the CPU checker can draw the same)
        dirs = [TA.isotropic(n, SEED, k, first_ray=first_ray) for k in range(gens)]

    def step_walk(f, count):
        """a pass of the step_n leg: sample the origins, then `gens` generations of
        turtle_stepper_step_n, each resumed from the sample the one before returned
        (TURTLE_AMD_STEP_RESUME), latitude and longitude not asked for.  `count`: an UNTIMED
        replica that also sums what the caller of a real walk would (lengths, steps)."""
        st = f.stepper.walk(f.pos) if compact else f.stepper.step(f.pos, None, outputs=False)
        if count:
            total = torch.zeros(n, dtype=torch.float64, device=dev)
            taken = torch.zeros(n, dtype=torch.int32, device=dev)
        else:
            f.gen_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            f.gen_ev[0].record()
        for k in range(gens):
            if count:
                alive = st["index"][:, 0] >= 0
            st = f.stepper.walk(None, dirs[k], state=st) if compact else \
                f.stepper.step(st["position"], dirs[k], resume=st)
            if count:
                total += torch.where(alive, st["step"], torch.zeros_like(total))
                taken += alive.to(torch.int32)
        if count:
            return {"index": st["index"].clone(), "length": total, "steps": taken}
        f.gen_ev[1].record()
        f.gen_ms.append(f.gen_ev)
        return None

    if args.sort_steps and not scatter:
        pos.copy_(pos0)
        stepper.trace_into(pos, direction, index, length, nsteps, args.max_steps)
        order = torch.argsort(nsteps, descending=args.sort_steps > 0, stable=True)
        pos0, direction = pos0[order].contiguous(), direction[order].contiguous()

    def one_pass(f):
        if not scatter:
            f.stepper.trace_into(f.pos, direction, f.index, f.length, f.nsteps, args.max_steps)
            return
        if gens:
            step_walk(f, False)
            f.walk = walk_state["w"]
            return
        # C5: turtle_stepper_scatter_n samples the origins, then takes scatter-steps
        # single steps per ray, each resumed from the sample of the one before,
        # directions drawn and sums kept in the kernels
        f.walk = walk_state["w"] = f.stepper.scatter(f.pos, SEED, args.scatter_steps, first_ray=first_ray)

    def reduce_tally(f):
        f.tally.zero_()
        if scatter:
            w = f.walk
            TA.tally(w["index"], w["length"], n_media, n_bins, lmax, f.tally[t_hits], f.tally[t_hist])
            f.tally[t_steps] = w["steps"].sum(dtype=torch.int64)
        else:
            TA.tally(f.index, f.length, n_media, n_bins, lmax, f.tally[t_hits], f.tally[t_hist])
            f.tally[t_steps] = f.nsteps.sum(dtype=torch.int64)
        sharding.all_reduce_tally(f.tally, world)   # RCCL, ~8 KB: the only collective

    def passes(count, lanes, events=None):
        for k in range(count):
            f = lanes[k % len(lanes)]
            f.enter()
            f.pos.copy_(pos0)                # a pass advances positions in place
            if events is not None:
                events[k][0].record()        # HIP events on the launch stream
            one_pass(f)
            if events is not None:
                events[k][1].record()
            reduce_tally(f)
        flights[0].enter()

    paged = bool(use_stack and args.stack_size)
    steps = args.steps if headline else (1 if (scatter or paged) else 3)
    warmup = args.warmup if headline else 1
    torch.cuda.synchronize()                 # the rays are there for every stream
    if gens:
        flights[0].pos.copy_(pos0)
        walk_state["w"] = step_walk(flights[0], True)
    passes(max(warmup, width), flights)      # (every stepper's scratch comes with its first pass)

    def new_events(count):
        return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                for _ in range(count)]

    def region(count, lanes):
        """`count` passes bracketed by a barrier + synchronize on both sides; the MAX over
        ranks of the wall time, and each pass's own HIP events (on its launch stream)"""
        ev = new_events(count)
        TA.set_in_flight(len(lanes))         # (a hint: the kernels share the SIMDs with their like)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        passes(count, lanes, ev)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        TA.set_in_flight(1)
        t_all = torch.tensor([dt], dtype=torch.float64,
                             device=dev if env["backend"] == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        return float(t_all.item()), float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # THE timed region: one batch on the GPU at a time.  value, ms_per_step, kernel.ms and
    # the roofline are all this region's.
    elapsed, kernel_ms = region(steps, flights[:1])
    stats = stepper.trace_stats()            # of the last pass on this rank
    if gens:
        # step_n keeps no totals: the replica's count, and SURVEY's 1.05 samples a step
        taken = int(walk_state["w"]["steps"].sum(dtype=torch.int64).item())
        stats = {"rays": n, "steps": taken, "samples": int(round(1.05 * taken)), "capped": 0}
        ms = [a.elapsed_time(b) for a, b in flights[0].gen_ms[-steps:]]
        gen_ms = float(np.mean(ms)) / gens
    total_steps_per_pass = int(tally[t_steps].item())   # all ranks (all-reduced)
    flight = None
    if width > 1:
        # beside it: `width` batches in flight (a trace ends with a few rays of thousands of
        # steps, which then step beside the bulk of the next batch); same bits
        count = max(steps, 3 * width)
        dt, span = region(count, flights)
        flight = {"batches": width, "passes": count, "ms_per_pass": 1e3 * dt / count,
                  "value": total_steps_per_pass * count / dt, "ms_a_pass_spans": span}
    out = None
    if rank == 0:
        value = total_steps_per_pass * steps / elapsed
        # a walk over resident tiles is ONE launch (k_walk); over paged tiles, or with
        # TURTLE_AMD_WALK=steps, two kernels per generation
        by_steps = bool(args.stack_size) or os.environ.get("TURTLE_AMD_WALK") == "steps" or bool(gens)
        launches = (args.scatter_steps if by_steps else 1) if scatter else 1
        if scatter:
            # SURVEY 8d, single-step batch mode: 8 B of nodes per sample + 48 in (pos, dir
            # -- drawn in the kernel here, but the state it stands for) + 24 (pos out) + 8
            # (ds) + 8 (index[2]) = 96 B per step
            samples_per_step = stats["samples"] / max(1, stats["steps"])
            alg_bytes = (8.0 * samples_per_step + 88.0) * stats["steps"]
            per_launch = alg_bytes / launches
            kernel = {"name": ((WALK_N_KERNELS if compact else STEP_N_KERNELS) if gens else STEP_KERNELS)
                      if by_steps else WALK_KERNEL,
                      "ms": kernel_ms, "launches_per_step": launches,
                      "ms_per_generation": gen_ms if gens else kernel_ms / args.scatter_steps,
                      "steps_per_pass": stats["steps"], "samples_per_pass": stats["samples"],
                      "samples_per_step": samples_per_step,
                      "gpu_steps_per_s": stats["steps"] / (kernel_ms * 1e-3)}
            bytes_note = "SURVEY 8d single-step mode: (8 B x samples/step + 88 B) per step"
        else:
            # SURVEY 8d, trace mode: 4 x 2 B nodes per sample + per ray 48 B in (pos, dir)
            # + 16 B out (medium, n_steps, path length)
            alg_bytes = 8.0 * stats["samples"] + 64.0 * stats["rays"]
            per_launch = alg_bytes
            kernel = {"name": TRACE_KERNEL, "ms": kernel_ms, "launches_per_step": 1,
                      "steps_per_launch": stats["steps"], "samples_per_launch": stats["samples"],
                      "samples_per_step": stats["samples"] / max(1, stats["steps"]),
                      "gpu_steps_per_s": stats["steps"] / (kernel_ms * 1e-3),
                      "rays_stopped_at_max_steps": stats["capped"]}
            bytes_note = "SURVEY 8d trace mode: 8 B x samples + 64 B x rays"
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src, valu = measured_traffic(
            name + (("_walk_n" if compact else "_step_n") if gens else ""), n, TA.get_math())
        if use_stack and args.stack_size:
            # the counters were taken with every tile resident: they say nothing of a paged pass
            traffic, traffic_src, valu = None, None, None
        simds = 4 * TA.compute_units()
        valu_frac = (4.0 * valu / (simds * kernel_ms * 1e-3 * VALU_CLOCK_HZ)) if valu else None
        out = {
            "value": value, "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
            "config": {"workload": text.replace("{steps}", str(args.scatter_steps)) +
                       ((" -- through turtle_stepper_walk_n, the caller's directions" if compact else
                         " -- through turtle_stepper_step_n, the caller's directions") if gens else "") + (f" -- stack_size {args.stack_size}: tiles paged host->HBM "
                                           f"by demand, {stepper.rounds} rounds in the last pass"
                                           if (use_stack and args.stack_size) else ""),
                       "rays_per_gpu": n, "max_steps": args.max_steps,
                       "slope": 0.4, "resolution": 1e-2, "math": TA.get_math(),
                       "parallelism": f"rays x{world}", "in_flight": 1},
            "kernel": kernel,
            "in_flight": flight,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": per_launch,
                         "valu_issue_frac": valu_frac, "bytes": bytes_note,
                         # where bandwidth is the question (single steps): what the counters say
                         # the launch really moved, over its time, against the HBM peak
                         "traffic_frac_of_hbm": (traffic / (kernel_ms / launches * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                 if (traffic and scatter) else None)},
            "tally": {"hits": [int(v) for v in tally[t_hits].tolist()],
                      "sha256": hashlib.sha256(tally.cpu().numpy().tobytes()).hexdigest()[:16]},
        }
        # ---- parity of what was just timed, and the CPU beside it ----
        if not args.no_cpu and world == 1:
            cores = host_cores()
            m = min(n, PARITY_RAYS if not headline else args.cpu_rays)
            if scatter:
                m = min(m, 2 * PARITY_RAYS)      # a scalar CPU walk: 2.4e6 steps/s
            p_host, d_host = pos0[:m].cpu().numpy(), direction[:m].cpu().numpy()
            if scatter:
                w = walk_state["w"]
                geo = terrain.oracle()
                ref_pos = p_host.copy()
                total, alive = np.zeros(m), np.ones(m, dtype=bool)
                t0 = time.perf_counter()
                cpu_steps = 0
                o = geo.step(ref_pos)
                alive = o["index"][:, 0] >= 0
                for k in range(args.scatter_steps):
                    dk = TA.isotropic(m, SEED, k, first_ray=first_ray, device=False)
                    o = geo.step(ref_pos, dk)
                    # the reference would stop stepping a ray that has left the data
                    ref_pos = np.where(alive[:, None], o["position"], ref_pos)
                    total += np.where(alive, o["step"], 0.0)
                    cpu_steps += int(alive.sum())
                    alive &= o["index"][:, 0] >= 0
                dt = time.perf_counter() - t0
                ref_index = np.where(alive[:, None], o["index"], -1)
                out["parity"] = parity_counts(w["index"][:m].cpu().numpy(),
                                              w["length"][:m].cpu().numpy(), ref_index, total)
                out["parity"]["note"] = ("a walk: a ray whose medium differs once has diverged for "
                                         "good, and so has every later step of it")
                port = (f"oracle/ C restatement, one thread (its single-step entry point is scalar): "
                        f"{cpu_steps / dt:.4g} steps/s")
                out["cpu_baseline"] = {
                    "value": cpu_steps / dt, "unit": "ray-steps/s", "cores": 1, "kind": "port",
                    "sample": f"{m} rays x {args.scatter_steps} steps, {port}, directions from the "
                              f"library's Philox"}
                from oracle import ref_ffi as R
                if R.driver_available():
                    # the reference itself, its threaded pattern, all granted cores; directions
                    # drawn on the GPU by the library's Philox, a chunk of rays at a time
                    path = terrain.hgt_files()
                    chunk, secs, ref_steps, equal = 25_000, 0.0, 0, True
                    for lo_ in range(0, m, chunk):
                        hi_ = min(m, lo_ + chunk)
                        dirs = np.stack([TA.isotropic(hi_ - lo_, SEED, k, first_ray=first_ray + lo_,
                                                      device=False) for k in range(args.scatter_steps)])
                        a = R.stack_run(path, p_host[lo_:hi_], dirs, walk_steps=args.scatter_steps,
                                        local_range=1.0, threads=cores)
                        secs += a["seconds"]
                        ref_steps += a["total_steps"]
                    out["cpu_baseline"] = {
                        "value": ref_steps / secs, "unit": "ray-steps/s", "cores": cores, "kind": "reference",
                        "sample": f"{m} rays x {args.scatter_steps} steps through the reference itself "
                                  f"(oracle/_ref; one turtle_stack with lock / unlock shared by {cores} "
                                  f"pthreads, a client per worker: the reference's threaded example; its "
                                  f"default local range of 1 m; the same tiles as .hgt files -- its GeoTIFF "
                                  f"reader wants libtiff under a name this image lacks); {ref_steps} steps "
                                  f"against the restatement's {cpu_steps}.  For comparison the {port}"}
            else:
                cpu, res = cpu_baseline(terrain, p_host, d_host, cores)
                out["parity"] = parity_counts(index[:m].cpu().numpy(), length[:m].cpu().numpy(),
                                              res["index"], res["length"], nsteps[:m].cpu().numpy(),
                                              res["n_steps"])
                out["cpu_baseline"] = cpu_baseline_entry(cpu, cores, m)
    for f in flights[1:]:
        f.stepper.destroy()
    terrain.close()
    return out


def run_micro(env, n=20_000_000):
    """The path's bandwidth-shaped batch kernels alone, once each (the
    ECEF transforms, the bilinear lookup, stepper_position): points/s and algorithmic bytes (the arrays
    a call reads and writes + 8 B of nodes where it looks a cell up) over the kernel's time."""
    import torch
    import turtle_amd as TA
    dev = env["dev"]
    terrain = Terrain(TA, (45, 3, 1, 1), False, env)
    try:
        g = torch.Generator(device=dev)
        g.manual_seed(SEED)
        lat = 45.05 + 0.9 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
        lon = 3.05 + 0.9 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
        h = 1000.0 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
        ecef = TA.ecef_from_geodetic(lat, lon, h)
        out = {}
        for name, call, nbytes in (
                ("ecef_from_geodetic_n", lambda: TA.ecef_from_geodetic(lat, lon, h), 24 + 24),
                ("ecef_to_geodetic_n", lambda: TA.ecef_to_geodetic(ecef), 24 + 24),
                ("map_elevation_n", lambda: terrain.handle.elevation(lon, lat), 16 + 8 + 4 + 8),
                ("stepper_position_n", lambda: terrain.stepper.position(lat, lon, 500.0), 24 + 24 + 4 + 8)):
            call()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                call()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            out[name] = {"points_per_s": n / (ms * 1e-3), "ms": ms, "bytes_per_point": nbytes,
                         "gbs": n * nbytes / (ms * 1e-3) / 1e9,
                         "frac_of_hbm": n * nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        return {"points": n, "kernels": out}
    finally:
        terrain.close()


DEFAULT_ALSO = "c3,c3@8,c4,c5,c5!step_n,c5!walk_n,c2!strict,c3!strict"
LINE_LIMIT = 6000           # bytes of the last stdout line (the driver keeps an 8 KB tail)


def _short(text, limit=200):
    return text if len(text) <= limit else text[: limit - 3] + "..."


def _round(x, digits=5):
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _round(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, digits) for v in x]
    return x


def short_leg(r):
    """one short object per further workload: numbers only"""
    out = {"value": r["value"], "ms_per_pass": r["ms_per_step"], "passes": r["steps"],
           "kernel_ms": r["kernel"]["ms"], "frac": r["roofline"]["frac"],
           "traffic": r["roofline"]["traffic"]}
    if r["kernel"].get("ms_per_generation") is not None:
        out["ms_per_generation"] = r["kernel"]["ms_per_generation"]
    if r["roofline"].get("traffic_frac_of_hbm") is not None:
        out["traffic_frac_of_hbm"] = r["roofline"]["traffic_frac_of_hbm"]
    if r.get("in_flight"):
        out["in_flight"] = {k: r["in_flight"][k] for k in ("batches", "value", "ms_per_pass")}
    if "parity" in r:
        out["medium_mismatch"] = r["parity"]["medium_mismatch"]
        out["beyond_1e-6"] = r["parity"]["beyond_1e-6"]
        out["parity_rays"] = r["parity"]["rays"]
    if "cpu_baseline" in r:
        out["cpu"] = {"value": r["cpu_baseline"]["value"], "kind": r["cpu_baseline"]["kind"],
                      "cores": r["cpu_baseline"]["cores"]}
    return out


def final_line(head, extra, args, world, backend, comm_size, micro=None):
    """The LAST stdout line: the headline (everything from one timed region, one batch at a
    time) and one short object per further workload; the full records went out before it."""
    kernel = head["kernel"]
    roof = head["roofline"]
    line = {
        "metric": "ray-steps/sec (whole node) through 3601^2 SRTM tile",
        "value": head["value"], "unit": "ray-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {k: head["config"][k] for k in ("workload", "rays_per_gpu", "max_steps", "math",
                                                   "parallelism", "in_flight")},
        "backend": backend if world > 1 else None, "comm_world_size": comm_size,
        "kernel": {"name": _short(kernel["name"], 120), "ms": kernel["ms"],
                   "launches_per_step": kernel["launches_per_step"],
                   "steps_per_launch": kernel.get("steps_per_launch", kernel.get("steps_per_pass")),
                   "samples_per_step": kernel["samples_per_step"]},
        "in_flight": ({k: head["in_flight"][k] for k in ("batches", "value", "ms_per_pass")}
                      if head.get("in_flight") else None),
        "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic",
                                          "traffic_source", "valu_issue_frac")},
        "tally": head["tally"],
    }
    line["config"]["workload"] = _short(line["config"]["workload"], 160)
    if "parity" in head:
        line["parity"] = {k: v for k, v in head["parity"].items() if k not in ("checker", "note")}
    if "cpu_baseline" in head:
        c = head["cpu_baseline"]
        line["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"],
                                "kind": c["kind"], "sample": _short(c["sample"], 200)}
    if extra:
        line["also"] = {k: short_leg(r) for k, r in extra.items()}
    if micro is not None:
        line["micro"] = {k: {"points_per_s": v["points_per_s"], "frac_of_hbm": v["frac_of_hbm"]}
                         for k, v in micro["kernels"].items()}
    line = _round(line)
    if len(json.dumps(line)) >= LINE_LIMIT:      # never lose the headline to a long line
        line["also"] = {k: {"value": v["value"], "ms_per_pass": v["ms_per_pass"]}
                        for k, v in line.get("also", {}).items()}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="default: c2 as the headline, c3 and c5 once each under 'also'")
    ap.add_argument("--also", default=None,
                    help="comma-separated workloads measured once each after the headline "
                         "('none': nothing; default: c3,c5 on one GPU when no --workload is given)")
    ap.add_argument("--rays", type=int, default=0, help="rays per GPU and block (0 = the workload's)")
    ap.add_argument("--blocks", type=int, default=1,
                    help="blocks of --rays rays per rank: --gpus 1 --blocks 8 traces the ray "
                         "array an 8-rank run shards (same tally, to the bit)")
    ap.add_argument("--stack-size", type=int, default=0,
                    help="c3 / c5: tiles the stack keeps in memory (0: all; SURVEY's second C3 leg: 8)")
    ap.add_argument("--max-steps", type=int, default=100_000)
    ap.add_argument("--scatter-steps", type=int, default=256, help="c5: steps per ray")
    ap.add_argument("--sort", type=int, default=0,
                    help="experiment: order the rays by origin in an NxN grid of bins")
    ap.add_argument("--sort-steps", type=int, default=0,
                    help="experiment: order the rays by their step count (1: longest first, "
                         "-1: shortest first), known from a trace made beforehand")
    ap.add_argument("--cpu-rays", type=int, default=1_000_000,
                    help="rays of the headline's CPU-baseline sample (default: the whole C2 batch)")
    ap.add_argument("--no-cpu", action="store_true", help="no CPU baseline, no parity count")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="batches in flight: steppers on streams of their own that take the passes in "
                         "turn, so that the few long rays a trace ends with step beside the bulk of the "
                         "next batch (0: three for a trace over resident terrain, else one)")
    ap.add_argument("--step-n", type=int, default=0,
                    help="c5 only: the walk through turtle_stepper_step_n with TURTLE_AMD_STEP_RESUME and "
                         "the CALLER's directions (drawn beforehand), this many generations a pass")
    ap.add_argument("--compact", action="store_true",
                    help="with --step-n: through turtle_stepper_walk_n (one double of state a ray between the calls)")
    ap.add_argument("--generations", type=int, default=64, help="generations of the default c5!step_n leg")
    ap.add_argument("--tiles", choices=("auto", "hgt", "tif"), default="auto",
                    help="tile files: SRTM's .hgt, ASTER-GDEM2's GeoTIFF-16 (auto: tif for c5, hgt else)")
    args = ap.parse_args()

    # `bench.py --gpus N` on its own: start the N ranks as a CHILD job (before anything here
    # has touched a GPU: a process that has must never exec), relay rank 0's line and code
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE {world}")
    # one process per GPU; TURTLE_BENCH_BACKEND=gloo lets several ranks share a
    # GPU to rehearse the N > 1 path on a one-GPU box (RCCL needs a GPU per rank)
    backend = os.environ.get("TURTLE_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count())
    os.environ["LOCAL_RANK"] = str(local)      # the C library picks its device from it
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import turtle_amd as TA

    # one non-default stream carries everything: the library's launches, torch's
    # copies and the timing events (the legacy stream's handle 0 cannot be
    # handed to a C API that reads NULL as "your own stream")
    if os.environ.get("TURTLE_AMD_MATH"):
        TA.set_math(os.environ["TURTLE_AMD_MATH"])   # experiments: fast | strict
    stream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(stream)
    TA.set_stream(stream)
    env = {"world": world, "rank": rank, "dev": torch.device("cuda", local), "backend": backend}

    head_name = args.workload or "c2"
    head = run_workload(head_name, args, env, headline=True)
    if rank == 0:
        print(json.dumps({"leg": head_name, **head}), flush=True)
    also = args.also
    if also is None:
        also = DEFAULT_ALSO if (args.workload is None and world == 1) else "none"
    extra = {}
    for leg in [w for w in also.split(",") if w and w != "none"]:
        name = leg
        sub = argparse.Namespace(**vars(args))
        sub.rays, sub.blocks, sub.sort, sub.sort_steps, sub.stack_size, sub.step_n, sub.compact = 0, 1, 0, 0, 0, 0, False
        if name.endswith("@8"):      # C3's second leg: the same workload, 8 of its 16 tiles resident
            name, sub.stack_size = name[:-2], 8
        if name.endswith("!step_n"):  # C5 through turtle_stepper_step_n, the caller's directions
            name, sub.step_n = name[:-7], args.generations
        if name.endswith("!walk_n"):  # ... through turtle_stepper_walk_n: one double of state a ray
            name, sub.step_n, sub.compact = name[:-7], args.generations, True
        strict = name.endswith("!strict")   # the reference's arithmetic, operand for operand
        if strict:
            name, sub.no_cpu = name[:-7], True
            TA.set_math("strict")
        try:
            r = run_workload(name, sub, env, headline=False)
        finally:
            if strict:
                TA.set_math(os.environ.get("TURTLE_AMD_MATH", "fast"))
        if rank == 0:
            key = (name + (f"_stack_size_{sub.stack_size}" if sub.stack_size else "")
                   + (("_walk_n" if sub.compact else "_step_n") if sub.step_n else "") + ("_strict" if strict else ""))
            print(json.dumps({"leg": key, **r}), flush=True)
            extra[key] = r

    micro = None
    if args.workload is None and world == 1 and args.also is None:
        micro = run_micro(env)
        print(json.dumps({"leg": "micro", **micro}), flush=True)
    if rank == 0:
        print(json.dumps(final_line(head, extra, args, world, env["backend"],
                                    dist.get_world_size() if world > 1 else 1, micro)), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
