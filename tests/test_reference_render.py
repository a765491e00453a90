"""Pin against the one output of the reference whose parameters are known.

tests/golden/reference_renders/cornell_nee2_1728x1117.png is the render area of the reference's own
screenshot renders/importance_sampling/0_1-NEE2.png, de-doubled (the screenshot is an exact 2x2 pixel
replication of the 1728x1117 image; made by tools/extract_reference_render.py). Its ImGui panel records
single render, 100 samples, bounce limit 5, debug -1, and the default Cornell scene is on screen.
Two things had to be inferred and are checked by the numbers below rather than assumed:
  * the screenshot is in the author's display space (embedded ICC profile: Display-P3 primaries, sRGB
    tone curve), so pure sRGB green shows as (0.46, 0.99, 0.30); it is converted back here;
  * all radiances are 1/2.4 of what the snapshot's defaults give, i.e. the light's emissionStrength was
    1 (an ImGui material edit) when it was taken; light transport is linear in it.
What is compared: silhouette and light edges to the pixel, radiometry per channel and region, the
per-pixel noise level, and (GPU, full frame) that the reference's noise follows this renderer's
frameCount-0 RNG streams more than any other seed. There is no bit-level vector to compare with."""
import os

import numpy as np
import pytest
from PIL import Image

from ray_tracer_amd import engine

HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 1728, 1117
P3_TO_SRGB = np.array([[1.2249, -0.2247, 0.0], [-0.0420, 1.0419, 0.0], [-0.0197, -0.0786, 1.0979]])


def eotf(v):
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)


def reference_linear():
    shot = np.array(Image.open(os.path.join(HERE, "golden", "reference_renders", "cornell_nee2_1728x1117.png"))).astype(np.float64) / 255.0
    assert shot.shape == (H, W, 3)
    valid = np.ones((H, W), bool)
    valid[:430, 1290:] = False   # ImGui panel
    valid[:2] = False            # window border
    valid[:, :2] = False
    valid[:, -2:] = False
    valid[1100:] = False         # rounded window corners
    return eotf(shot) @ P3_TO_SRGB.T, shot, valid


def screenshot_scene():
    s = engine.Scene()
    s.prepare_storage_buffers()
    s.arrays().materials[3].emissionStrength = 1.0
    return s


def constants(frameCount=0):
    return engine.push_constants(W, H, singleRender=1, sampleLimit=100, bounceLimit=5, frameCount=frameCount)


def test_fixture_is_display_p3_of_pure_wall_colours():
    ref, shot, valid = reference_linear()
    g = ref[400:600, 440:520].mean((0, 1))   # left wall: material 2 = (0,1,0)
    r = ref[400:600, 1200:1260].mean((0, 1))  # right wall: material 1 = (1,0,0)
    assert abs(g[0]) < 2e-3 * g[1] + 1e-3 and abs(g[2]) < 2e-3 * g[1] + 1e-3 and g[1] > 0.05
    assert abs(r[1]) < 2e-3 * r[0] + 1e-3 and abs(r[2]) < 2e-3 * r[0] + 1e-3 and r[0] > 0.05


def test_oracle_rows_match_the_reference_render():
    from oracle import pyoracle
    ref, shot, valid = reference_linear()
    rows = dict(row0=120, rowStride=115, nRows=8)
    ys = [rows["row0"] + k * rows["rowStride"] for k in range(rows["nRows"])]
    img, _ = pyoracle.render(screenshot_scene(), constants(0), W, H, **rows)
    img = img[..., :3].astype(np.float64)
    r = ref[ys]
    v = valid[ys]
    # silhouette of the box opening: first and last lit pixel of every row
    for k, y in enumerate(ys):
        mine = np.where(img[k].sum(-1) > 0)[0]
        theirs = np.where((shot[y].sum(-1) > 0) & v[k])[0]
        assert abs(int(mine.min()) - int(theirs.min())) <= 1, (y, mine.min(), theirs.min())
        if v[k].all():
            assert abs(int(mine.max()) - int(theirs.max())) <= 1, (y, mine.max(), theirs.max())
    # the light (row 120 crosses it): saturated span, edges to the pixel
    sat_mine = np.where(img[0].min(-1) >= 1.0)[0]
    sat_ref = np.where(shot[ys[0]].min(-1) >= 1.0)[0]
    assert abs(int(sat_mine.min()) - int(sat_ref.min())) <= 1 and abs(int(sat_mine.max()) - int(sat_ref.max())) <= 1
    # radiometry, per channel, over everything that is lit and not clipped
    sel = v & (img.sum(-1) > 0) & (img.max(-1) < 0.9) & (r.max(-1) < 0.9)
    ratio = img[sel].sum(0) / r[sel].sum(0)
    assert np.all(np.abs(ratio - 1.0) < 0.03), ratio
    # ... and per wall: left third (green bounce), middle, right third (red bounce)
    x = np.arange(W)[None, :].repeat(len(ys), 0)
    for lo, hi in ((371, 700), (700, 1030), (1030, 1357)):
        m = sel & (x >= lo) & (x < hi)
        lum = img[m].sum() / r[m].sum()
        assert abs(lum - 1.0) < 0.04, (lo, hi, lum)


@pytest.mark.gpu
def test_full_frame_matches_the_reference_render(renderer):
    ref, shot, valid = reference_linear()
    r = renderer
    r.set_tuning("pipeline", -1)
    r.upload_scene(screenshot_scene())
    imgs = [r.render(constants(fc), W, H)[..., :3].astype(np.float64) for fc in range(3)]
    conv = np.zeros_like(imgs[0])
    n = 16
    for fc in range(100, 100 + n):
        conv += r.render(constants(fc), W, H)[..., :3]
    conv /= n
    lit = valid & (conv.sum(-1) > 0)
    # silhouette: the set of lit pixels
    lit_ref = valid & (shot.sum(-1) > 0)
    inter, union = (lit & lit_ref).sum(), (lit | lit_ref).sum()
    assert inter / union > 0.998, inter / union
    sel = lit & (conv.max(-1) < 0.9) & (ref.max(-1) < 0.9)
    # radiometry per region and channel (converged image of this renderer against the 100-sample reference)
    regions = {"back wall": (450, 550, 800, 900), "green wall": (400, 600, 440, 520), "red wall": (400, 600, 1200, 1260),
               "floor": (860, 900, 700, 1000), "ceiling": (40, 70, 600, 700), "tall box": (500, 700, 900, 1000),
               "short box top": (690, 710, 650, 800)}
    for name, (y0, y1, x0, x1) in regions.items():
        a, b = conv[y0:y1, x0:x1].reshape(-1, 3).sum(0), ref[y0:y1, x0:x1].reshape(-1, 3).sum(0)
        for c in range(3):
            if b[c] > 1e-3 * b.max():
                assert abs(a[c] / b[c] - 1.0) < 0.03, (name, c, a[c] / b[c])
    scale = ref[sel].sum() / conv[sel].sum()
    assert abs(scale - 1.0) < 0.02, scale
    # the same estimator: per-pixel noise of 100 samples is as large as the reference's (which also carries 8-bit quantisation)
    nref = (ref - scale * conv)[sel]
    nme = (imgs[1] - conv)[sel]
    assert 0.85 < nme.std() / nref.std() < 1.05, (nme.std(), nref.std())
    # the same RNG streams: the reference's deviations follow frameCount 0 of this renderer, not the other seeds
    corr = [np.corrcoef((imgs[fc] - conv)[sel].ravel(), nref.ravel())[0, 1] for fc in range(3)]
    assert corr[0] > max(corr[1], corr[2]) + 0.04, corr
