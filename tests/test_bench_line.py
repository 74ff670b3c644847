"""bench.py's last stdout line is what the driver parses, from an 8 KB tail: it must stay
short whatever the workloads report, and carry the headline keys."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _record(name, wordy=400):
    """a workload's full record as run_workload builds it, with texts as long as they get"""
    return {
        "value": 6.4912345678e10, "ms_per_step": 3.3512345678, "steps": 20,
        "config": {"workload": name + " " + "w" * wordy, "rays_per_gpu": 12_500_000, "max_steps": 100000,
                   "slope": 0.4, "resolution": 1e-2, "math": "fast", "parallelism": "rays x1", "in_flight": 1},
        "kernel": {"name": "k" * wordy, "ms": 3.3012345678, "launches_per_step": 1,
                   "steps_per_launch": 217_412_345, "samples_per_launch": 230_730_385,
                   "samples_per_step": 1.0612345678, "gpu_steps_per_s": 6.5e10, "rays_stopped_at_max_steps": 0,
                   "ms_per_generation": 0.93123456},
        "in_flight": {"batches": 3, "passes": 20, "ms_per_pass": 2.15812345, "value": 1.00812345e11,
                      "ms_a_pass_spans": 5.9},
        "roofline": {"bound": "hbm", "achieved": 570.12345678, "peak": 8000.0, "unit": "GB/s",
                     "frac": 0.0712345678, "traffic": 3.3012345e9, "traffic_source": "r04_c2_pmc.json",
                     "algorithmic_bytes_per_launch": 1.91e9, "valu_issue_frac": 0.4612345, "bytes": "b" * wordy,
                     "traffic_frac_of_hbm": 0.2612345},
        "tally": {"hits": [940123, 59877], "sha256": "0123456789abcdef"},
        "parity": {"rays": 1_000_000, "medium_mismatch": 0, "beyond_1e-6": 0, "max_rel_path_length": 2.1e-8,
                   "checker": "c" * wordy, "step_count_mismatch": 3, "max_step_count_difference": 1},
        "cpu_baseline": {"value": 1.9812345e8, "unit": "ray-steps/s", "cores": 16, "kind": "reference",
                         "sample": "s" * wordy, "range0": 1.7e8, "one_core": 1.59e7,
                         "equal_to_the_restatement_bit_for_bit": True, "port_range0": 1.6e8,
                         "port_range1": 1.9e8, "port_one_core": 1.5e7},
    }


def test_last_line_is_short_and_carries_the_headline():
    import bench
    args = argparse.Namespace(steps=20, warmup=5)
    legs = {k: _record(k) for k in bench.DEFAULT_ALSO.replace("@8", "_stack_size_8").replace("!", "_").split(",")}
    micro = {"points": 20_000_000, "kernels": {f"kernel_{i}_of_the_path_n": {
        "points_per_s": 1.0123456e11, "ms": 0.4941234, "bytes_per_point": 48, "gbs": 4861.1234, "frac_of_hbm": 0.60761234}
        for i in range(4)}}
    line = bench.final_line(_record("c2"), legs, args, 1, "nccl", 1, micro)
    assert set(line["micro"]) == set(micro["kernels"])
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "kernel", "roofline", "cpu_baseline",
                "parity", "in_flight", "also"):
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert len(line["cpu_baseline"]["sample"]) <= 200
    assert line["kernel"]["ms"] <= line["ms_per_step"]           # one timed region, one batch at a time
    assert set(line["also"]) == set(legs)
    for leg in line["also"].values():
        assert {"value", "ms_per_pass", "frac", "medium_mismatch", "beyond_1e-6", "cpu"} <= set(leg)
    # twenty workloads with texts ten times as long: the headline still fits
    many = {f"leg{i}": _record("x", 4000) for i in range(20)}
    text = json.dumps(bench.final_line(_record("c2", 4000), many, args, 8, "nccl", 8))
    assert len(text) < bench.LINE_LIMIT, len(text)
    assert json.loads(text)["comm_world_size"] == 8 and json.loads(text)["backend"] == "nccl"
