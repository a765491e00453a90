#!/usr/bin/env python3
"""Eyeball check for the texture semantics: dread.obj with dread_alb.png in the bare Cornell box, to be looked at
next to the reference's renders/dread_texture.png (an earlier commit of the author: no cubes, parameters unrecorded).
usage: tools/render_dread.py out.png [spp]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, render  # noqa: E402

out = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
s = engine.Scene()
for m in (engine.default_material(), engine.default_material(albedo=(1, 0, 0)), engine.default_material(albedo=(0, 1, 0)),
          engine.default_material(albedo=(0, 0, 0), emissionColor=(1, 1, 1), emissionStrength=2.4)):
    s.add_material(m)
for i in range(10):
    s.set_sphere(i, (0, 0, 0), 0.0, 0)
s.cornell_box()
n0 = s.counts()["materials"]
s.read_obj(os.path.join(engine.ASSET_DIR, "dread.obj"), engine.placement(position=(-0.35, 0.43, 0.0), scale=0.85, rotation=(0, 25, 0)), 0)
slot = s.add_texture(os.path.join(engine.ASSET_DIR, "dread_alb.png"))
for mi in range(n0, s.counts()["materials"]):
    m = s.material(mi)
    m.albedoIndex = slot
    s.set_material(mi, m)
W, H = 864, 558
pc = engine.push_constants(W, H, singleRender=1, sampleLimit=spp)
r = engine.Renderer(0)
r.upload_scene(s)
r.upload_textures(engine.load_textures(s))
img = r.render(pc, W, H)
from PIL import Image
Image.fromarray(render.srgb8(img)[..., :3]).save(out)
print("wrote", out)
