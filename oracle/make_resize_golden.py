"""Writes tests/golden/resize_pil.json: digest of PIL.Image.resize on a seeded frame (run in the build container, where
Pillow — the reference's dependency for this step — is installed).  TEST INFRASTRUCTURE ONLY."""
import hashlib
import json
import os

import numpy as np
from PIL import Image
import PIL

seed, h, w, out = 20250705, 480, 640, 224
frame = np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
res = np.asarray(Image.fromarray(frame).resize((out, out)))
gold = dict(seed=seed, h=h, w=w, out=out, pillow=PIL.__version__, sha256=hashlib.sha256(res.tobytes()).hexdigest(),
            first_pixels=res[:2, :4].tolist())
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resize_pil.json")
json.dump(gold, open(path, "w"), indent=1)
print(path, gold["sha256"])
