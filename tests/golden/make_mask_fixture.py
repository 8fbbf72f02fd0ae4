"""Build the mask fixtures used by BASELINE config C3 from the reference's own test/mask-east.jpg
(run in the build container only; PIL decodes the JPEG, the loader rule resize -> BGR2GRAY -> (>128 -> 1)
is the oracle's restatement of fileio.load_mask, fileio.py:250-292).  Only the binarised result is committed."""
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, "/root/repo")
from oracle import m3_oracle as mo

bgr = np.array(Image.open("/root/reference/test/mask-east.jpg").convert("RGB"))[..., ::-1].copy()
for (W, H) in [(3840, 2160), (960, 540)]:
    np.savez_compressed(f"/root/repo/tests/golden/mask_east_{W}x{H}.npz", mask=mo.load_mask_jpg(bgr, (W, H)))
