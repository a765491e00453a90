"""Scene configurations C1–C5 of BASELINE.md §3 (SURVEY §8d "Concrete inputs").

The large assets of the reference (bunny_full.obj, dragon.obj, sponza.obj) are
not shipped with it. When a real file is present under
`assets/` (or $RT_ASSET_DIR) it is loaded through read_obj; otherwise a seeded
procedural stand-in with the same triangle count is generated and the scene is
labelled "synthetic-<N>-tris". Stand-ins go through the same
triangle / centroid / BVH path as an OBJ group (rt_scene_add_mesh).

Geometry generation is plain numpy on the host: it is input synthesis, not
part of the hot path.
"""
import os

import numpy as np

from . import engine

BUNNY_TRIS = 69451
DRAGON_TRIS = 871414
SPONZA_TRIS = 262267


def _asset(name):
    for d in (os.environ.get("RT_ASSET_DIR"), engine.ASSET_DIR):
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return None


# --------------------------------------------------------------------------
# mesh generators: return (positions[T,3,3], normals[T,3,3]) float32
# --------------------------------------------------------------------------
def _tri_normals_smooth(verts, faces):
    """Area-weighted vertex normals."""
    v = verts.astype(np.float64)
    fn = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, faces[:, k], fn)
    ln = np.linalg.norm(vn, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return (vn / ln).astype(np.float32)


def _expand(verts, faces, normals):
    return verts[faces].astype(np.float32), normals[faces].astype(np.float32)


def _debris(rng, n, center, extent, size):
    """n small free triangles scattered in a box (leaves, chips): used to hit exact counts."""
    if n <= 0:
        return np.zeros((0, 3, 3), np.float32), np.zeros((0, 3, 3), np.float32)
    c = center + (rng.random((n, 1, 3)) - 0.5) * extent
    tri = c + (rng.random((n, 3, 3)) - 0.5) * size
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    nrm = nrm / ln
    return tri.astype(np.float32), np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32)


def blob(ntris, seed=1, radius=1.0, center=(0, 0, 0), bumps=6, amp=0.18):
    """Closed lat-long surface with smooth seeded displacement, exactly ntris triangles."""
    rng = np.random.default_rng(seed)
    ntris = int(ntris)
    if ntris < 8:
        return _debris(rng, ntris, np.asarray(center, np.float64), radius, radius * 0.3)
    c = max(3, int(np.sqrt(ntris / 2.0) * 1.4))
    r = max(2, ntris // (2 * c) + 1)          # 2c(r-1) <= ntris
    while 2 * c * (r - 1) > ntris:
        r -= 1
    r = max(r, 2)
    theta = np.linspace(0, np.pi, r + 1)       # rings
    phi = np.linspace(0, 2 * np.pi, c, endpoint=False)
    T, P = np.meshgrid(theta[1:-1], phi, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    d = np.concatenate([[[0, 1, 0]], d, [[0, -1, 0]]], 0)
    k = rng.normal(size=(bumps, 3)) * 2.5
    ph = rng.uniform(0, 2 * np.pi, bumps)
    disp = 1.0 + amp * np.mean(np.sin(d @ k.T + ph), axis=1)
    verts = d * disp[:, None] * radius + np.asarray(center)
    faces = []
    ring = lambda i: 1 + i * c                 # noqa: E731  first vertex of ring i (0..r-2)
    idx = np.arange(c)
    nxt = (idx + 1) % c
    faces.append(np.stack([np.zeros(c, int), ring(0) + nxt, ring(0) + idx], 1))
    for i in range(r - 2):
        a, b = ring(i), ring(i + 1)
        faces.append(np.stack([a + idx, a + nxt, b + idx], 1))
        faces.append(np.stack([a + nxt, b + nxt, b + idx], 1))
    last = len(verts) - 1
    faces.append(np.stack([np.full(c, last), ring(r - 2) + idx, ring(r - 2) + nxt], 1))
    faces = np.concatenate(faces, 0)
    pos, nrm = _expand(verts, faces, _tri_normals_smooth(verts, faces))
    rest = ntris - len(pos)
    dp, dn = _debris(rng, rest, np.asarray(center, np.float64), radius * 2.2, radius * 0.04)
    return np.concatenate([pos, dp]), np.concatenate([nrm, dn])


def grid_patch(origin, du, dv, nu, nv, disp=None, flip=False):
    """Rectangle origin + s*du + t*dv tessellated nu x nv, optional displacement along its normal."""
    origin, du, dv = (np.asarray(x, np.float64) for x in (origin, du, dv))
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    n = np.cross(du, dv)
    n /= np.linalg.norm(n)
    p = origin + s[..., None] * du + t[..., None] * dv
    if disp is not None:
        p = p + disp(s, t)[..., None] * n
    verts = p.reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b = a + (nv + 1)
    f1 = np.stack([a, b, a + 1], 1)
    f2 = np.stack([b, b + 1, a + 1], 1)
    faces = np.concatenate([f1, f2], 0)
    if flip:
        faces = faces[:, ::-1]
    return _expand(verts, faces, _tri_normals_smooth(verts, faces))


def cylinder(base, axis, radius, nseg, nring, bulge=0.0):
    """Open cylinder (column): 2*nseg*nring triangles."""
    base, axis = np.asarray(base, np.float64), np.asarray(axis, np.float64)
    h = np.linalg.norm(axis)
    w = axis / h
    u = np.cross(w, [1.0, 0, 0] if abs(w[0]) < 0.9 else [0, 0, 1.0])
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    t, a = np.meshgrid(np.linspace(0, 1, nring + 1), np.linspace(0, 2 * np.pi, nseg, endpoint=False), indexing="ij")
    rad = radius * (1.0 + bulge * np.sin(np.pi * t))
    p = base + t[..., None] * axis + rad[..., None] * (np.cos(a)[..., None] * u + np.sin(a)[..., None] * v)
    verts = p.reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nring), np.arange(nseg), indexing="ij")
    a0 = (i * nseg + j).ravel()
    a1 = (i * nseg + (j + 1) % nseg).ravel()
    b0, b1 = a0 + nseg, a1 + nseg
    faces = np.concatenate([np.stack([a0, a1, b0], 1), np.stack([a1, b1, b0], 1)], 0)
    return _expand(verts, faces, _tri_normals_smooth(verts, faces))


def _cat(parts):
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


# --------------------------------------------------------------------------
# Sponza stand-in: an atrium along +z, y-down like the reference's world,
# 25 material groups (one RenderObject + BVH each, as read_obj makes per
# usemtl group, src/vk_engine.cpp:960-1002) whose regions overlap in space.
# --------------------------------------------------------------------------
_SPONZA_WEIGHTS = [  # (group, share of triangles)
    ("floor", 0.03), ("walls_lower", 0.05), ("walls_upper", 0.05), ("arches_lower", 0.07), ("arches_upper", 0.07),
    ("columns_a", 0.06), ("columns_b", 0.06), ("columns_c", 0.05), ("ceiling_beams", 0.03), ("roof_frame", 0.02),
    ("curtain_red", 0.06), ("curtain_green", 0.06), ("curtain_blue", 0.06), ("fabric_a", 0.03), ("fabric_b", 0.03),
    ("vase_round", 0.04), ("vase_hanging", 0.03), ("vase_plant", 0.05), ("flagpoles", 0.02), ("chains", 0.02),
    ("lion", 0.04), ("lion_background", 0.02), ("details", 0.02), ("bricks", 0.02), ("leaf", 0.01),
]


def sponza_standin_groups(ntris=SPONZA_TRIS, seed=1):
    """25 (name, positions, normals) groups totalling exactly ntris triangles."""
    rng = np.random.default_rng(seed)
    L, Wd, Hh = 12.0, 5.0, 9.0           # half length (z), half width (x), height; floor at y=+0.5, up is -y
    yf = 0.5
    shares = np.array([w for _, w in _SPONZA_WEIGHTS])
    counts = np.floor(shares / shares.sum() * ntris).astype(int)
    counts[-1] += ntris - counts.sum()
    groups = []

    def wave(ax, ay, fx, fy, ph=0.0):
        return lambda s, t: ax * np.sin(fx * s * 2 * np.pi + ph) + ay * np.sin(fy * t * 2 * np.pi)

    def side(n):  # grid resolution giving about n triangles for a square-ish patch
        return max(1, int(np.sqrt(max(n, 2) / 2.0)))

    def build(name, gi, n):
        parts = []
        if name == "floor":
            k = side(n)
            parts.append(grid_patch([-Wd, yf, -L], [2 * Wd, 0, 0], [0, 0, 2 * L], k, k, wave(0.01, 0.01, 9, 23), flip=True))
        elif name in ("walls_lower", "walls_upper"):
            y0 = yf if name == "walls_lower" else yf - Hh / 2
            k = side(n / 4)
            for sx in (-1, 1):
                parts.append(grid_patch([sx * Wd, y0, -L], [0, 0, 2 * L], [0, -Hh / 2, 0], 2 * k, k // 2 + 1,
                                        wave(0.02, 0.02, 40, 6), flip=sx > 0))
            for sz in (-1, 1):
                parts.append(grid_patch([-Wd, y0, sz * L], [2 * Wd, 0, 0], [0, -Hh / 2, 0], k, k // 2 + 1,
                                        wave(0.02, 0.02, 12, 6), flip=sz < 0))
        elif name in ("arches_lower", "arches_upper"):
            y0 = yf - 3.0 if name == "arches_lower" else yf - 3.0 - Hh / 2
            nb = 10
            per = max(8, n // (2 * nb))
            seg = max(3, int(np.sqrt(per / 2)))
            for sx in (-1, 1):
                for b in range(nb):
                    z0 = -L + (b + 0.5) * (2 * L / nb)
                    parts.append(cylinder([sx * (Wd - 1.5), y0, z0 - 1.0], [0, 0, 2.0], 0.9, seg, seg, bulge=0.0))
        elif name.startswith("columns"):
            off = {"columns_a": 0.0, "columns_b": 0.8, "columns_c": 1.6}[name]
            ncol = 22
            per = max(8, n // ncol)
            seg = max(3, int(np.sqrt(per / 2)))
            ring = max(1, per // (2 * seg))
            for i in range(ncol):
                sx = -1 if i % 2 else 1
                z0 = -L + 1.0 + (i // 2) * (2 * L - 2.0) / (ncol // 2 - 1) + off * 0.3
                ytop = yf - (Hh / 2 if name != "columns_c" else Hh)
                parts.append(cylinder([sx * (Wd - 1.5 + off * 0.2), yf, z0], [0, ytop - yf, 0], 0.22 + 0.03 * off, seg, ring, bulge=0.08))
        elif name in ("ceiling_beams", "roof_frame"):
            nb = 14
            per = max(8, n // nb)
            seg = max(3, int(np.sqrt(per / 2)))
            ring = max(1, per // (2 * seg))
            y0 = yf - Hh / 2 if name == "ceiling_beams" else yf - Hh
            for i in range(nb):
                z0 = -L + (i + 0.5) * 2 * L / nb
                parts.append(cylinder([-Wd, y0, z0], [2 * Wd, 0, 0], 0.12, seg, ring))
        elif name.startswith("curtain") or name.startswith("fabric"):
            lane = {"curtain_red": -3, "curtain_green": -1, "curtain_blue": 1, "fabric_a": 3, "fabric_b": 5}[name]
            npan = 4
            k = side(n / npan)
            for i in range(npan):
                sx = -1 if i % 2 else 1
                z0 = lane * 1.8 + (i // 2) * 0.9 - 0.5
                parts.append(grid_patch([sx * (Wd - 1.45), yf - Hh / 2 - 0.2, z0], [0, 0, 1.6], [0, 3.2, 0], k, k,
                                        wave(0.12, 0.03, 5 + i, 2, ph=i), flip=sx > 0))
        elif name in ("vase_round", "vase_hanging", "vase_plant"):
            nv = 8
            per = n // nv
            for i in range(nv):
                sx = -1 if i % 2 else 1
                z0 = -L + 2.0 + (i // 2) * (2 * L - 4.0) / (nv // 2 - 1)
                if name == "vase_round":
                    c = [sx * 2.0, yf - 0.45, z0]
                elif name == "vase_hanging":
                    c = [sx * 2.6, yf - Hh / 2 + 1.0, z0 + 1.0]
                else:
                    c = [sx * 2.0, yf - 1.2, z0]
                parts.append(blob(per, seed=seed * 100 + gi * 10 + i, radius=0.45 if name != "vase_plant" else 0.6, center=c,
                                  bumps=4 if name != "vase_plant" else 14, amp=0.15 if name != "vase_plant" else 0.6))
        elif name in ("flagpoles", "chains"):
            nb = 12
            per = max(8, n // nb)
            seg = max(3, int(np.sqrt(per / 4)))
            ring = max(1, per // (2 * seg))
            for i in range(nb):
                sx = -1 if i % 2 else 1
                z0 = -L + 1.5 + (i // 2) * (2 * L - 3.0) / (nb // 2 - 1)
                if name == "flagpoles":
                    parts.append(cylinder([sx * (Wd - 1.4), yf - Hh / 2 - 0.5, z0], [-sx * 1.8, -0.6, 0], 0.03, seg, ring))
                else:
                    parts.append(cylinder([sx * 2.6, yf - Hh / 2, z0 + 1.0], [0, 0.8, 0], 0.015, seg, ring))
        elif name in ("lion", "lion_background"):
            k = side(n)
            bump = (lambda s, t: 0.25 * np.exp(-((s - 0.5) ** 2 + (t - 0.5) ** 2) * 18) * (1 + 0.3 * np.sin(40 * s) * np.sin(37 * t))) \
                if name == "lion" else wave(0.01, 0.01, 30, 30)
            d = 0.02 if name == "lion" else 0.0
            parts.append(grid_patch([-1.2, yf - 0.6, L - 0.05 - d], [2.4, 0, 0], [0, -2.4, 0], k, k, bump, flip=True))
        elif name in ("details", "bricks"):
            nb = 40
            per = max(8, n // nb)
            k = side(per)
            for i in range(nb):
                sx = -1 if i % 2 else 1
                z0 = -L + (i // 2 + 0.5) * 2 * L / (nb // 2)
                y0 = yf - (0.8 if name == "bricks" else Hh / 2 + 0.9)
                parts.append(grid_patch([sx * (Wd - 0.03), y0, z0 - 0.4], [0, 0, 0.8], [0, -0.5, 0], k, k,
                                        wave(0.01, 0.01, 6, 6), flip=sx > 0))
        return parts

    for gi, ((name, _), n) in enumerate(zip(_SPONZA_WEIGHTS, counts)):
        n = int(n)
        if name == "leaf":  # free small triangles around the plants
            pos, nrm = _debris(rng, n, np.array([0.0, yf - 1.3, 0.0]), np.array([4.6, 1.2, 2 * L - 4]), 0.06)
        else:
            # raise the tessellation until the group has at least n triangles, then trim
            mult = 1.0
            while True:
                parts = build(name, gi, int(n * mult) + 8)
                if sum(len(p[0]) for p in parts) >= n:
                    break
                mult *= 1.15
            pos, nrm = _cat(parts)
            pos, nrm = pos[:n], nrm[:n]
        groups.append((name, pos, nrm))
    assert sum(len(g[1]) for g in groups) == ntris
    return groups


# --------------------------------------------------------------------------
# configurations
# --------------------------------------------------------------------------
def cornell(spheres=True):
    """C1: default Cornell scene (+ dielectric, mirror and diffuse spheres)."""
    s = engine.Scene()
    s.prepare_storage_buffers()
    if spheres:
        s.set_sphere(0, (0.0, 0.1, -0.3), 0.4, 5)
        s.set_sphere(1, (0.5, 0.1, 0.0), 0.4, 4)
        s.set_sphere(2, (-0.5, 0.1, 0.0), 0.4, 0)
    return s, "cornell-51-tris"


def cornell_with_model(obj_name, standin_tris, material, seed, scale=0.7, position=(0.0, 0.53, 0.0)):
    s = engine.Scene()
    s.prepare_storage_buffers()
    p = engine.placement(position=position, scale=scale, samplerIndex=1)
    path = _asset(obj_name)
    if path:
        s.read_obj(path, p, material)
        label = f"{obj_name}"
    else:
        pos, nrm = blob(standin_tris, seed=seed, radius=1.0, center=(0, -1.0, 0))
        s.add_mesh(f"standin:{obj_name}", pos, nrm, p, material)
        label = f"synthetic-{standin_tris}-tris"
    return s, label


def cornell_bunny():
    """C2: Cornell + Stanford bunny, diffuse (src/vk_engine.cpp:745-749 placement)."""
    return cornell_with_model("bunny_full.obj", BUNNY_TRIS, 0, seed=2)


def cornell_dragon():
    """C3: Cornell + Stanford dragon, mirror."""
    return cornell_with_model("dragon.obj", DRAGON_TRIS, 4, seed=3)


def _bake_y(pos, nrm, position, scale, angle_deg):
    """Vertices and normals of a mesh placed by T * Ry * S (uniform S), in world space."""
    a = np.radians(angle_deg)
    c, sn = np.cos(a), np.sin(a)
    R = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]])      # glm::rotate about +y
    p = pos.reshape(-1, 3).astype(np.float64) * scale @ R.T + np.asarray(position, np.float64)
    n = nrm.reshape(-1, 3).astype(np.float64) @ R.T
    return p.astype(np.float32).reshape(-1, 9), n.astype(np.float32).reshape(-1, 9)


def sponza(dragons=0, ntris=SPONZA_TRIS, seed=1, flatten=False):
    """C4 (dragons=0) / C5 (dragons=16): Sponza with materials from sponza.mtl
    (albedo = Ka*Kd, textures not sampled: SURVEY F3), a Cornell-style emitter at
    the rectangle NEE is hard-wired to (raytrace.comp:368-387), environment on.
    flatten=True is C5's second variant: no instancing, every dragon is its own
    mesh with its placement baked into the vertices and its own BVH."""
    s = engine.Scene()
    s._l.rt_scene_set_sphere  # noqa: B018  (ten zeroed spheres, as prepare_storage_buffers)
    for i in range(10):
        s.set_sphere(i, (0, 0, 0), 0.0, 0)
    for m in (engine.default_material(), engine.default_material(albedo=(1, 0, 0)),
              engine.default_material(albedo=(0, 1, 0)),
              engine.default_material(albedo=(0, 0, 0), emissionColor=(1, 1, 1), emissionStrength=2.4),
              engine.default_material(reflectance=1.0), engine.default_material(ior=2.0)):
        s.add_material(m)
    real = _asset("sponza.obj") or _asset(os.path.join("sponza2", "sponza_tri.obj"))
    if real:
        # as the reference places it (src/vk_engine.cpp:726-729: scale 1, no rotation, origin): an identity transform, so its
        # usemtl groups take the traversal's identity path like the stand-in's. The file's own units then apply: the camera of
        # sponza_camera() and the Cornell emitter at y = -1.5 are set for the stand-in's atrium (24 x 10 x 9 units) and have to be
        # moved by the caller for a model in other units (--camera-position / --fov of the CLI)
        s.read_obj(real, engine.placement(), 0)
        label = os.path.basename(real)
    else:
        mtl = _asset("sponza.mtl")
        first = s.counts()["materials"]
        s.read_mtl(mtl)
        nmat = s.counts()["materials"] - first
        for gi, (name, pos, nrm) in enumerate(sponza_standin_groups(ntris, seed)):
            s.add_mesh(f"standin:sponza/{name}", pos, nrm, engine.placement(), first + (gi % nmat))
        label = f"synthetic-{ntris}-tris"
    s.read_obj(os.path.join(engine.ASSET_DIR, "light2.obj"), engine.placement(position=(0, -1.5, 0), frontOnly=True), 3)
    if dragons:
        path = _asset("dragon.obj")
        if not path or flatten:
            pos, nrm = blob(DRAGON_TRIS, seed=3, radius=1.0, center=(0, -1.0, 0))
        for i in range(dragons):
            gx, gz = i % 4, i // 4
            where = (-3.0 + 2.0 * gx, 0.3, -7.5 + 5.0 * gz)
            p = engine.placement(position=where, scale=0.8, rotation=(0, 22.5 * i, 0))
            if flatten:
                fp, fn = _bake_y(pos, nrm, where, 0.8, 22.5 * i)
                s.add_mesh(f"flat:dragon/{i}", fp, fn, engine.placement(), 4 if i % 2 else 0)
            elif path:
                s.read_obj(path, p, 4 if i % 2 else 0)
            else:
                s.add_mesh("standin:dragon.obj", pos, nrm, p, 4 if i % 2 else 0)
        label += f"+{dragons}x{'dragon.obj' if path and not flatten else 'synthetic-%d-tris' % DRAGON_TRIS}"
        if flatten:
            label += "-flattened"
    return s, label


def sponza_camera(width, height, **kw):
    """Camera inside the atrium, looking down its length (+z)."""
    kw.setdefault("pos", (0.0, -1.3, -10.5))
    kw.setdefault("cameraAngles", (2.0, 0.0, 0.0))
    kw.setdefault("fov", 65.0)
    kw.setdefault("environmentOn", True)
    return engine.push_constants(width, height, **kw)


CONFIGS = {
    "cornell": lambda: cornell(True),
    "bunny": cornell_bunny,
    "dragon": cornell_dragon,
    "sponza": lambda: sponza(0),
    "sponza_dragons": lambda: sponza(16),
    "sponza_dragons_flat": lambda: sponza(16, flatten=True),
}
