"""Host-fed loader preprocessing: eight 4K BGR frames in ordinary (pageable) NumPy memory -> 960x540 gray, per call.
MDH_PRE_STAGE=0 python tools/bench_pre_host.py: the copy straight from the caller's memory instead of through the library's pinned buffers."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from metdetpy_amd.imgproc import Preprocessor
rng = np.random.default_rng(3)
f = rng.integers(0, 256, (8, 2160, 3840, 3), dtype=np.uint8)
pre = Preprocessor((2160, 3840), (960, 540), None)
pre.batch(f); pre.batch(f)
t = time.perf_counter()
for _ in range(10): out = pre.batch(f)
dt = (time.perf_counter() - t) / 10
print(f"{f.nbytes/1e6:.0f} MB host -> device + preprocess: {dt*1e3:.2f} ms per call, {f.nbytes/dt/1e9:.1f} GB/s, checksum {int(out.sum())}")
pre.close()
