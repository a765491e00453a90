"""Compare a render of the default Cornell scene with the reference's own screenshot
renders/importance_sampling/0_1-NEE2.png (de-doubled to 1728x1117, tests/golden/reference_renders/).
The ImGui panel in that screenshot records the parameters: single render, 100 samples, bounce limit 5.
The screenshot is in the Display-P3 space of the author's screen; radiances are 1/2.4 of the snapshot's
defaults, i.e. the light's emissionStrength was 1 when it was taken."""
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine  # noqa: E402

shot = np.array(Image.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_renders", "cornell_nee2_1728x1117.png"))).astype(np.float64) / 255.0
H, W = shot.shape[:2]
mask = np.ones((H, W), bool)
mask[:430, 1290:] = False
mask[:2] = False
mask[1100:] = False


def eotf(v):
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)


def oetf(x):
    x = np.clip(x, 0, 1)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)


P3_TO_SRGB = np.array([[1.2249, -0.2247, 0.0], [-0.0420, 1.0419, 0.0], [-0.0197, -0.0786, 1.0979]])
ref_lin = eotf(shot) @ P3_TO_SRGB.T          # what the reference stored, decoded to linear sRGB
ref8 = np.floor(oetf(ref_lin) * 255 + 0.5)    # ... and as the 8-bit sRGB-encoded value of its image

scene = engine.Scene()
scene.prepare_storage_buffers()
strength = float(os.environ.get("STRENGTH", "1.0"))
scene.arrays().materials[3].emissionStrength = strength
r = engine.Renderer(0)
r.upload_scene(scene)
lit = mask & (ref8.sum(-1) > 0)
imgs = {}
for fc in (0, 1, 2, 3):
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=100, bounceLimit=5, frameCount=fc)
    img = r.render(pc, W, H)[..., :3].astype(np.float64)
    imgs[fc] = img
    q = np.floor(oetf(img) * 255.0 + 0.5)
    d = np.abs(q - ref8).max(-1)
    dl = d[lit]
    rel = (img[lit] / np.maximum(ref_lin[lit], 1e-4))
    print(f"frameCount {fc}: lit pixels {lit.sum()}  |d8| exact {np.mean(dl == 0) * 100:5.1f} %  <=1 {np.mean(dl <= 1) * 100:5.1f} %  <=2 {np.mean(dl <= 2) * 100:5.1f} %  <=4 {np.mean(dl <= 4) * 100:5.1f} %  "
          f"<=8 {np.mean(dl <= 8) * 100:5.1f} %  mean {dl.mean():.2f}   median linear ratio mine/ref {np.median(rel):.4f}", flush=True)
# noise against a converged image: do the reference's per-pixel deviations follow this renderer's for some seed?
conv = np.zeros_like(imgs[0])
NCONV = 24
for fc in range(100, 100 + NCONV):
    pc = engine.push_constants(W, H, singleRender=1, sampleLimit=100, bounceLimit=5, frameCount=fc)
    conv += r.render(pc, W, H)[..., :3]
conv /= NCONV
sel = lit & (conv.max(-1) < 0.9)
scale = ref_lin[sel].sum() / conv[sel].sum()
print(f"converged image ({NCONV * 100} spp): reference / converged = {scale:.4f}")
nref = (ref_lin - scale * conv)[sel]
for fc in (0, 1, 2, 3):
    nme = (imgs[fc] - conv)[sel]
    print(f"frameCount {fc}: correlation of the per-pixel deviations from the converged image, reference vs this renderer: "
          f"{np.corrcoef(nme.ravel(), nref.ravel())[0, 1]:.4f}   (rms mine {nme.std():.5f} ref {nref.std():.5f})")
# what independent noise looks like: two of my own renders with different seeds
q0, q1 = np.floor(oetf(imgs[0]) * 255 + 0.5), np.floor(oetf(imgs[1]) * 255 + 0.5)
dn = np.abs(q0 - q1).max(-1)[lit]
print(f"two seeds of this renderer against each other: <=1 {np.mean(dn <= 1) * 100:5.1f} %  <=2 {np.mean(dn <= 2) * 100:5.1f} %  <=4 {np.mean(dn <= 4) * 100:5.1f} %  mean {dn.mean():.2f}")
np.save(os.path.join("gpurun_out", "cornell_ref_cmp.npy"), imgs[0].astype(np.float32))
