#!/usr/bin/env python3
"""bench.py -- ray-steps/s of the trace kernel on the BASELINE workload.

Workload (BASELINE.json configs[1], "C2"): 1 M rays per GPU through one
synthetic 3601x3601 SRTMGL1 tile (turtle_amd.synth), rays from the common
recipe of SURVEY.md 8d: origin uniform over the tile 500 m above ground,
azimuth U[0,360), elevation U[-10,-1] deg, slope 0.4, resolution 1e-2, traced
until the medium changes (hit or exit).  One "step" of this benchmark = one
turtle_stepper_trace_n call over the whole batch, inputs resident in HBM.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), tile
replicated, rays block-sharded (rank r draws its own Philox block), no
data-path collective; the per-step tally (hit counts + 1024-bin path-length
histogram, uint64) is all-reduced.  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
VALU_CLOCK_HZ = 2.4e9   # MI355X peak engine clock (MI355X_MICROARCH.md)
TRACE_KERNEL = "k_trace (phase A: all rays up to 512 steps; phase B: the parked long rays)"
STEP_KERNELS = "k_step + k_bisect per generation (single steps; rays that cross a boundary bisected packed)"


def measured_traffic(workload, rays, math):
    """HBM-side bytes per trace_n call from the committed PMC passes
    (profiles/*_pmc.json, written by scripts/profile_round.sh): counters cannot
    be read from inside the timed process, so the figure is the offline
    measurement of the same command, or None when none matches."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if (d.get("workload"), d.get("rays_per_gpu"), d.get("math")) == (workload, rays, math):
            valu = sum(k.get("SQ_INSTS_VALU", 0.0) for k in d.get("kernels", {}).values())
            return d.get("traffic_bytes_per_launch"), os.path.basename(path), valu
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


WORKLOADS = {
    # name: (tiles (lat0, lon0, nlat, nlon), through a stack?, default rays/GPU, text)
    "c2": ((45, 3, 1, 1), False, 1_000_000,
           "C2: 1M rays/GPU, one 3601x3601 SRTMGL1 tile, trace to first boundary"),
    "c5": ((40, 0, 10, 10), True, 10_000_000,
           "C5: 10M scattering rays/GPU, 256 single steps each with a new isotropic "
           "direction per step (Philox(ray, step)), 10x10 mosaic of 3601x3601 tiles"),
    "c3": ((45, 3, 4, 4), True, 10_000_000,
           "C3: 10M rays/GPU, 4x4 mosaic of 3601x3601 tiles through a stack (all "
           "tiles resident in HBM), trace to first boundary"),
}


def cpu_baseline(tiles, use_stack, n_rays, seed):
    """The CPU restatement (oracle/, kind "port") on a bounded sample of the
    same workload, all host cores, exact transform (range 0) and the
    reference's default local-linear approximation (range 1)."""
    from oracle import ffi as O
    from turtle_amd import synth
    lat0, lon0, nlat, nlon = tiles
    grids, table = [], []
    for i in range(nlat):
        for j in range(nlon):
            table.append(len(grids))
            grids.append(O.hgt_grid(lat0 + i, lon0 + j, synth.srtm_like_nodes(lat0 + i, lon0 + j)))
    if use_stack:
        stack = dict(lat0=float(lat0), lon0=float(lon0), dlat=1.0, dlon=1.0, nlat=nlat,
                     nlon=nlon, tile=np.array(table, dtype=np.int32))
        geo = O.OracleGeometry(grids=grids, stacks=[stack], layers=[[(O.STACK, 0, 0.0)]])
    else:
        geo = O.OracleGeometry(grids=grids, layers=[[(O.MAP, 0, 0.0)]])
    lat, lon, az, el = synth.uniform_rays(
        n_rays, (float(lat0), float(lat0 + nlat)), (float(lon0), float(lon0 + nlon)), seed=seed)
    pos, _ = geo.position(lat, lon, 500.0)
    d = O.ecef_from_horizontal(lat, lon, az, el)
    cores = host_cores()
    out = {}
    geo.trace(pos[:20000], d[:20000], local_range=0.0, threads=cores)   # warm up
    for tag, rng in (("range0", 0.0), ("range1", 1.0)):
        t0 = time.perf_counter()
        r = geo.trace(pos, d, local_range=rng, threads=cores)
        dt = time.perf_counter() - t0
        out[tag] = r["total_steps"] / dt
        out[tag + "_s"] = dt
    t0 = time.perf_counter()
    one = geo.trace(pos[: max(1, n_rays // cores)], d[: max(1, n_rays // cores)],
                    local_range=0.0, threads=1)
    out["one_core"] = one["total_steps"] / (time.perf_counter() - t0)
    # The REAL reference, when its build travelled with the repo (oracle/_ref/, made
    # by oracle/Makefile in the build container) and the terrain is a single map: the
    # same rays through turtle_stepper_step in the example harness's loop
    # (oracle/ref_driver.c), one stepper per thread; the clock runs over the stepping.
    from oracle import ref_ffi as R
    if (not use_stack) and R.driver_available():
        tmp = tempfile.mkdtemp(prefix="turtle_ref_")
        path = synth.write_hgt(tmp, lat0, lon0)
        R.trace_map(path, pos[:20000], d[:20000], local_range=1.0, threads=cores)   # warm up
        for tag, rng in (("ref_range1", 1.0), ("ref_range0", 0.0)):
            a = R.trace_map(path, pos, d, local_range=rng, threads=cores)
            out[tag] = a["total_steps"] / a["seconds"]
        same = geo.trace(pos, d, local_range=0.0, threads=cores)
        out["ref_equal"] = bool(np.array_equal(a["index"], same["index"]) and
                                np.array_equal(a["length"], same["length"]))
        a = R.trace_map(path, pos[: max(1, n_rays // cores)], d[: max(1, n_rays // cores)],
                        local_range=1.0, threads=1)
        out["ref_one_core"] = a["total_steps"] / a["seconds"]
    return out, cores, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--rays", type=int, default=0, help="rays per GPU (0 = the workload's)")
    ap.add_argument("--max-steps", type=int, default=100_000)
    ap.add_argument("--scatter-steps", type=int, default=256, help="c5: steps per ray")
    ap.add_argument("--sort", type=int, default=0,
                    help="experiment: order the rays by origin in an NxN grid of bins")
    ap.add_argument("--sort-steps", type=int, default=0,
                    help="experiment: order the rays by their step count (1: longest first, "
                         "-1: shortest first), known from a trace made beforehand")
    ap.add_argument("--cpu-rays", type=int, default=1_000_000,
                    help="rays of the CPU-baseline sample (default: the whole C2 batch)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
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
    from turtle_amd import sharding, synth

    # ---- terrain: synthetic SRTMGL1 tile(s), loaded through the C API ----
    tiles, use_stack, default_rays, workload_text = WORKLOADS[args.workload]
    lat0, lon0, nlat, nlon = tiles
    lat_range, lon_range = (float(lat0), float(lat0 + nlat)), (float(lon0), float(lon0 + nlon))
    tmp = tempfile.mkdtemp(prefix=f"turtle_bench_{rank}_")
    for i in range(nlat):
        for j in range(nlon):
            synth.write_hgt(tmp, lat0 + i, lon0 + j)
    stepper = TA.Stepper()
    if use_stack:
        terrain = TA.Stack(tmp, 0)
        terrain.load()                         # every tile resident: no paging rounds
        stepper.add_stack(terrain, 0.0)
    else:
        terrain = TA.Map.load(os.path.join(tmp, synth.hgt_name(lat0, lon0)))
        stepper.add_map(terrain, 0.0)
    # one non-default stream carries everything: the library's launches, torch's
    # copies and the timing events (the legacy stream's handle 0 cannot be
    # handed to a C API that reads NULL as "your own stream")
    if os.environ.get("TURTLE_AMD_MATH"):
        TA.set_math(os.environ["TURTLE_AMD_MATH"])   # experiments: fast | strict
    stream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(stream)
    TA.set_stream(stream)

    # ---- rays: rank r draws block r of the Philox stream; set-up on the GPU ----
    n = args.rays or default_rays
    lat, lon, az, el = sharding.rank_rays(n, rank, lat_range, lon_range)
    if args.sort > 0:
        nb = args.sort
        bx = np.minimum((nb * (lon - lon_range[0]) / (lon_range[1] - lon_range[0])).astype(int), nb - 1)
        by = np.minimum((nb * (lat - lat_range[0]) / (lat_range[1] - lat_range[0])).astype(int), nb - 1)
        order = np.argsort(by * nb + bx, kind="stable")
        lat, lon, az, el = lat[order], lon[order], az[order], el[order]
    dev = torch.device("cuda", local)
    t_lat, t_lon, t_az, t_el = (torch.as_tensor(v, device=dev) for v in (lat, lon, az, el))
    pos0, di = stepper.position(t_lat, t_lon, 500.0)
    direction = TA.ecef_from_horizontal(t_lat, t_lon, t_az, t_el)
    assert int((di != 0).sum()) == 0
    pos = torch.empty_like(pos0)
    index = torch.empty((n, 2), dtype=torch.int32, device=dev)
    length = torch.empty(n, dtype=torch.float64, device=dev)
    nsteps = torch.empty(n, dtype=torch.int32, device=dev)
    n_media, n_bins, lmax = 2, 1024, 65536.0
    t_hits, t_hist, t_steps, t_size = sharding.tally_layout(n_media, n_bins)
    tally = torch.zeros(t_size, dtype=torch.int64, device=dev)

    if args.sort_steps and args.workload != "c5":
        pos.copy_(pos0)
        stepper.trace_into(pos, direction, index, length, nsteps, args.max_steps)
        order = torch.argsort(nsteps, descending=args.sort_steps > 0, stable=True)
        pos0, direction = pos0[order].contiguous(), direction[order].contiguous()
    scatter = args.workload == "c5"
    first_ray = rank * n
    direction_k = torch.empty_like(pos0) if scatter else None
    walk = {}

    def one_step():
        if not scatter:
            stepper.trace_into(pos, direction, index, length, nsteps, args.max_steps)
            return
        # C5: sample the origins once, then scatter-steps single steps, each
        # resuming from the previous call's sample (one sample per step)
        state = stepper.step(pos, None, outputs=False)
        moved = torch.zeros(n, dtype=torch.int32, device=dev)
        total = torch.zeros(n, dtype=torch.float64, device=dev)
        for k in range(args.scatter_steps):
            TA.isotropic(n, 0x5EED2026, k, first_ray, out=direction_k)
            inside = state["index"][:, 0] >= 0
            state = stepper.step(state["position"], direction_k, resume=state)
            moved.add_(inside)
            total.add_(state["step"])
        index.copy_(state["index"])
        length.copy_(total)
        nsteps.copy_(moved)
        walk["kernel_launches"] = args.scatter_steps

    def reduce_tally():
        tally.zero_()
        TA.tally(index, length, n_media, n_bins, lmax, tally[t_hits], tally[t_hist])
        tally[t_steps] = nsteps.sum(dtype=torch.int64)
        sharding.all_reduce_tally(tally, world)   # RCCL, ~8 KB: the only collective

    for _ in range(args.warmup):
        pos.copy_(pos0)                      # a trace advances positions in place
        one_step()
        reduce_tally()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pos.copy_(pos0)
        ev[k][0].record()                    # HIP events on the launch stream
        one_step()
        ev[k][1].record()
        reduce_tally()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t_all = torch.tensor([elapsed], dtype=torch.float64,
                         device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    elapsed = float(t_all.item())

    stats = stepper.trace_stats()            # of the last launch on this rank
    total_steps_per_pass = int(tally[t_steps].item())   # all ranks (all-reduced)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if rank == 0:
        value = total_steps_per_pass * args.steps / elapsed
        # algorithmic bytes of one launch: 4 x 2 B nodes per sample
        # + per ray 48 B in (pos, dir) + 44 B out (pos, index, length, n_steps)
        alg_bytes = 8.0 * stats["samples"] + 92.0 * stats["rays"]
        launches = 1
        if scatter:
            # single-step mode streams the ray state every step: in 80 B (pos, dir,
            # alt, elev[2], index[2]) + out 64 B (pos, alt, elev[2], step, index[2])
            alg_bytes = 8.0 * stats["samples"] + 144.0 * n
            launches = args.scatter_steps
        achieved = alg_bytes * launches / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src, valu = measured_traffic(args.workload, n, TA.get_math())
        # the bound that does apply: wave64 VALU instructions issue at one per 4
        # cycles per SIMD (counted offline: SQ_INSTS_VALU of the same command)
        simds = 4 * TA.compute_units()
        valu_frac = (4.0 * valu / (simds * kernel_ms * 1e-3 * VALU_CLOCK_HZ)) if valu else None
        line = {
            "metric": "ray-steps/sec (whole node) through 3601^2 SRTM tile",
            "value": value, "unit": "ray-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_text, "rays_per_gpu": n, "max_steps": args.max_steps,
                       "slope": 0.4, "resolution": 1e-2, "math": TA.get_math(),
                       "parallelism": f"rays x{world}"},
            "kernel": {"name": STEP_KERNELS if scatter else TRACE_KERNEL, "ms": kernel_ms,
                       "launches_per_step": launches,
                       "steps_per_launch": stats["steps"],
                       "samples_per_launch": stats["samples"],
                       "samples_per_step": stats["samples"] / max(1, stats["steps"]),
                       "gpu_steps_per_s": stats["steps"] * launches / (kernel_ms * 1e-3),
                       "capped_rays": stats["capped"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes * launches,
                         "valu_issue_frac": valu_frac,
                         "note": "achieved = algorithmic bytes / kernel time; traffic = "
                                 "FETCH_SIZE+WRITE_SIZE bytes per launch (PMC, offline). The "
                                 "kernel is fp64-VALU/latency shaped, not bandwidth shaped: "
                                 "valu_issue_frac = wave-VALU instructions x 4 cycles / (SIMDs x "
                                 "kernel time x 2.4 GHz), whole launch (the bulk phase alone: "
                                 "0.76); see DESIGN.md"},
            "tally": {"hits": [int(v) for v in tally[t_hits].tolist()]},
        }
        if not args.no_cpu and world == 1:
            cpu, cores, _ = cpu_baseline(tiles, use_stack, args.cpu_rays, 0x5EED2026)
            port = (f"oracle/ C restatement, {cores} pthreads: {cpu['range0']:.4g} steps/s with the "
                    f"exact transform (range 0), {cpu['range1']:.4g} at range 1, "
                    f"{cpu['one_core']:.4g} on one core")
            if "ref_range1" in cpu:
                line["cpu_baseline"] = {
                    "value": cpu["ref_range1"], "unit": "ray-steps/s", "cores": cores,
                    "kind": "reference",
                    "sample": f"{args.cpu_rays} rays of the same recipe through the reference itself "
                              f"(oracle/_ref, turtle_stepper_step in the example harness's loop, one "
                              f"stepper per thread, {cores} pthreads, its default local range of 1 m); "
                              f"with the exact transform (range 0, what the GPU computes): "
                              f"{cpu['ref_range0']:.4g} steps/s; one core: {cpu['ref_one_core']:.4g}; "
                              f"results equal to the restatement's bit for bit: {cpu['ref_equal']}. "
                              f"For comparison the {port}"}
            else:
                line["cpu_baseline"] = {
                    "value": cpu["range0"], "unit": "ray-steps/s", "cores": cores,
                    "kind": "port",
                    "sample": f"{args.cpu_rays} rays of the same recipe, {port}"}
        print(json.dumps(line), flush=True)

    stepper.destroy()
    terrain.destroy()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
