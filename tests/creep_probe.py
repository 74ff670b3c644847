"""Helper of test_gpu_properties.test_creep_loop_changes_no_bit: traces a fixed batch
through one map and through a one-tile stack and stores the results.  Run as a child
process, because the library reads TURTLE_AMD_CREEP_LANES once."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA                      # noqa: E402
from turtle_amd import synth                 # noqa: E402


def main(out_path, workdir):
    n_nodes = 1201
    path = synth.write_hgt(os.path.join(workdir, "map"), 45, 3, n_nodes)
    synth.write_hgt(os.path.join(workdir, "stack"), 45, 3, n_nodes)
    lat, lon, az, el = synth.uniform_rays(40000, (45.0, 46.0), (3.0, 4.0), seed=123,
                                          el_range=(-3.0, -0.2))      # shallow: long rays
    out = {}
    for tag in ("map", "stack"):
        st = TA.Stepper()
        if tag == "map":
            terrain = TA.Map.load(path)
            st.add_map(terrain, 0.0)
        else:
            terrain = TA.Stack(os.path.join(workdir, "stack"), 0)
            st.add_stack(terrain, 0.0)
        pos, _ = st.position(lat, lon, 300.0)
        d = TA.ecef_from_horizontal(lat, lon, az, el)
        t = st.trace(pos.copy(), d)
        for k in ("position", "index", "length", "n_steps"):
            out[f"{tag}_{k}"] = np.asarray(t[k])
        st.destroy()
        terrain.destroy()
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
