"""The per-frame protocol's RATES (run on the MI355X box: pytest -m gpu).  This file sorts first on purpose: the legs below are a
latency-bound chain of dependent launches, and a parent process that holds a HIP context of its own -- pytest after any other GPU
test -- slows them by 40 % through its idle hardware queues (tests/test_gpu_bench_contract.py says so where it checks the same
legs' plumbing).  Here the pytest process has not touched HIP yet; when it has (another order, another selection), the test
skips instead of asserting on a disturbed number.

BASELINE.md's target for the unchanged update() / detect() loop is >= 10 000 frames/s at 1920x1080; VERDICT round 4 asks for the
three hand-over kinds to be held to numbers: pinned pool and recycled (registered) ring >= 9.5 k, fresh pageable arrays -- the
copy path, PCIe-staged by the runtime -- >= 7 k (measured 10.4 k / 10.2 k / 8.0 k, profiles/r05_bench_per_frame_*.json)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def hip_untouched():
    mod = sys.modules.get("metdetpy_amd._lib")
    return mod is None or mod._lib is None


@pytest.mark.parametrize("frames,steps,floor", [("pinned", 8000, 9500), ("recycled", 8000, 9500), ("pageable", 4000, 7000)])
def test_per_frame_protocol_rate(frames, steps, floor):
    if not hip_untouched():
        pytest.skip("this pytest process already holds a HIP context: its idle queues disturb the children's latency-bound chain")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    best = 0.0
    for _ in range(2):  # (best of two: the first process on a fresh box also pays the runtime's one-time set-up)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--protocol", "per-frame", "--frames", frames, "--steps", str(steps),
                              "--unique", "100", "--cpu-seconds", "0", "--parity-frames", "0", "--no-secondary"], capture_output=True, text=True,
                             timeout=600, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        best = max(best, d["value"])
    assert best >= floor, (frames, best, d["per_frame"])
    st = d["per_frame"]["update_stats"]
    if frames == "pageable":
        assert st["direct_frames"] == 0 and st["auto_registered"] == 0  # nothing is pinned behind a caller's back (ABI 4)
    else:
        assert st["direct_frames"] > 0.98 * (st["direct_frames"] + st["copied_frames"])
