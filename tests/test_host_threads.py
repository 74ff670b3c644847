"""The host's scalar path under the reference's threaded pattern, with ThreadSanitizer (CPU only:
sanitizers do not run on the GPU pool), and the environment switch that selects that path."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from turtle_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "turtle_amd", "csrc")


def _build(tmp, sanitizer):
    """libturtle_amd with its host objects under `sanitizer` (the device object as built),
    and tests/c/host_stack_threads.c against it"""
    out = os.path.join(tmp, "san")
    os.makedirs(out, exist_ok=True)
    flags = ["-O1", "-g", "-std=gnu99", "-fPIC", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objs = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".c"):
            objs.append(os.path.join(out, f[:-2] + ".o"))
            subprocess.check_call(["gcc"] + flags + ["-c", os.path.join(CSRC, f), "-o", objs[-1]])
    lib = os.path.join(out, "libturtle_amd.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", lib] + objs +
                          [os.path.join(CSRC, "build", "device.o"), "-lm", "-lz", "-lpthread",
                           f"-fsanitize={sanitizer}"])
    exe = os.path.join(out, "host_stack_threads")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", f"-fsanitize={sanitizer}",
                           "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "host_stack_threads.c"), "-L" + out,
                           "-lturtle_amd", "-lpthread", "-lm", "-Wl,-rpath," + out, "-o", exe])
    return exe


@pytest.mark.parametrize("size", [1, 2])
def test_threads_share_a_small_locked_stack_on_the_host(tmp_path, size):
    """a lookup on the host must not read a tile another thread's load frees.
    Four threads, a locked stack that keeps 1 or 2 of its 4 tiles: every answer right, and
    ThreadSanitizer sees no race (it makes the process exit non-zero if it does)."""
    if not os.path.exists(os.path.join(CSRC, "build", "device.o")):
        pytest.skip("the library has not been built here")
    tiles = os.path.join(str(tmp_path), "tiles")
    for la in (45, 46):
        for lo in (3, 4):
            synth.write_hgt(tiles, la, lo, 1201)
    exe = _build(str(tmp_path), "thread")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66 report_signal_unsafe=0")
    run = subprocess.run([exe, tiles, str(size), "4", "160"], capture_output=True, text=True, timeout=600,
                         env=env)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "0 wrong answers" in run.stdout


def test_environment_selects_the_host_for_scalar_calls():
    """TURTLE_AMD_SCALAR=host, read once at first use: a caller relinked against the library,
    its source unchanged, gets the scalar calls answered on the host; turtle_amd_scalar_set wins."""
    code = ("import turtle_amd as TA; L = TA.lib(); print(L.turtle_amd_scalar_get()); "
            "L.turtle_amd_scalar_set(0); print(L.turtle_amd_scalar_get()); "
            "L.turtle_amd_scalar_set(1); print(L.turtle_amd_scalar_get())")
    for value, first in (("host", 1), ("device", 0), (None, 0), ("nonsense", 0)):
        env = {k: v for k, v in os.environ.items() if k != "TURTLE_AMD_SCALAR"}
        if value is not None:
            env["TURTLE_AMD_SCALAR"] = value
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-800:]
        assert [int(v) for v in out.stdout.split()] == [first, 0, 1], (value, out.stdout)
