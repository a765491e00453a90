"""Makes tests/golden/reference_renders/cornell_nee2_1728x1117.png from the reference's screenshot
renders/importance_sampling/0_1-NEE2.png (3680x2514 RGBA: window shadow, 56-px title bar, then the
3456x2234 content, which is the 1728x1117 render with every pixel replicated 2x2)."""
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/renders/importance_sampling/0_1-NEE2.png"
im = np.array(Image.open(src))
a = im[..., 3]
ys, xs = np.where(a == 255)
x0, y0 = xs.min(), ys.min() + 56
content = im[y0:y0 + 2234, x0:x0 + 3456, :3]
small = np.ascontiguousarray(content[::2, ::2])
d = np.abs(small.astype(int) - content[1::2, 1::2].astype(int)).max(-1)
ok = np.ones(d.shape, bool)
ok[:430, 1290:] = False
ok[:2] = False
ok[1100:] = False
assert (d[ok] == 0).all(), "not a 2x2 replication outside the UI overlay"
Image.fromarray(small).save("tests/golden/reference_renders/cornell_nee2_1728x1117.png", optimize=True)
print(small.shape)
