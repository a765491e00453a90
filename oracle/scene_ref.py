"""TEST INFRASTRUCTURE — independent Python restatement of the reference's
scene preparation, used only to cross-check the C++ host code
(ray_tracer_amd/csrc/scene.cpp) on small meshes:

    read_obj   src/vk_engine.cpp:800-1037   (v / vt / vn / f, one unshared
                                             TrianglePoint per corner, centroids)
    build_bvh  src/vk_engine.cpp:1169-1337  (binned SAH, 20 bins, the
                                             rightArea quirk of :1321, Hoare
                                             partition, children in pairs)

All arithmetic is done on numpy float32 scalars so every operation rounds to
binary32 exactly as the C++ does. Pure-Python loops: small cases only.
"""
import numpy as np

F = np.float32
BINS = 20


def parse_obj(path):
    """Positions / normals / uvs and faces of a triangulated OBJ -> per-triangle corner data
    in file order: tri_pos[T,3,3], tri_nrm[T,3,3], tri_uv[T,3,2] (float32)."""
    pos, nrm, uv = [], [], []
    tp, tn, tu = [], [], []
    include_uv = False
    with open(path) as f:
        for line in f.read().split("\n"):
            prefix = line.split(" ")[0] if " " in line else line
            if prefix == "v":
                pos.append([F(x) for x in line[2:].split(" ")[:3]])
            elif prefix == "vn":
                nrm.append([F(x) for x in line[3:].split(" ")[:3]])
            elif prefix == "vt":
                parts = line.split(" ")
                uv.append([F(parts[1]), F(parts[2])])
            elif prefix == "f":
                corners = sum(1 for ch in line[:-1] if ch == " ")
                toks = [t for t in line.split(" ")[1:] if t != ""][:corners]
                cp, cn, cu = [], [], []
                for t in toks:
                    fields = t.split("/")
                    vi = int(fields[0]) - 1
                    ti = int(fields[1]) - 1 if len(fields) > 1 and fields[1] != "" else None
                    ni = int(fields[2]) - 1 if len(fields) > 2 and fields[2] != "" else None
                    if ti is not None:
                        include_uv = True
                    cp.append(pos[vi])
                    cn.append(nrm[ni] if nrm else [F(0)] * 3)
                    cu.append(uv[ti] if include_uv else [F(0), F(0)])
                tp.append(cp[:3]); tn.append(cn[:3]); tu.append(cu[:3])
    return np.array(tp, F), np.array(tn, F), np.array(tu, F)


def centroids_of(tri_pos):
    c = np.zeros((len(tri_pos), 3), F)
    for t in range(len(tri_pos)):
        for a in range(3):
            s = F(0)
            for k in range(3):
                s = F(s + tri_pos[t, k, a])
            c[t, a] = F(s / F(3))
    return c


class Box:
    def __init__(self):
        self.lo = [F(1e30)] * 4
        self.hi = [F(-1e30)] * 4

    def grow_point(self, p):
        for i in range(3):
            self.lo[i] = p[i] if p[i] < self.lo[i] else self.lo[i]
            self.hi[i] = p[i] if self.hi[i] < p[i] else self.hi[i]

    def grow_box(self, b):
        for i in range(4):
            self.lo[i] = b.lo[i] if b.lo[i] < self.lo[i] else self.lo[i]
            self.hi[i] = b.hi[i] if self.hi[i] < b.hi[i] else self.hi[i]

    def area(self):
        with np.errstate(over="ignore", invalid="ignore"):
            x = F(self.hi[0] - self.lo[0]); y = F(self.hi[1] - self.lo[1]); z = F(self.hi[2] - self.lo[2])
            return F(F(F(x * y) + F(y * z)) + F(z * x))


def build_bvh(tri_pos):
    """Returns (nodes, order): nodes = list of [minx,maxx,miny,maxy,minz,maxz,index,triCount]
    in the reference's numbering (root 0, children in pairs), order = triangle permutation."""
    n = len(tri_pos)
    order = list(range(n))
    cent = centroids_of(tri_pos)
    nodes = [None] * (2 * n - 1)
    used = [1]

    def bounds(first, count):
        b = Box()
        for i in range(first, first + count):
            for k in range(3):
                b.grow_point(tri_pos[order[i], k])
        return b

    def set_node(idx, first, count):
        b = bounds(first, count)
        nodes[idx] = [b.lo[0], b.hi[0], b.lo[1], b.hi[1], b.lo[2], b.hi[2], first, count]

    def find_split(first, count):
        best, axis, pos = F(1e30), 0, F(0)
        for a in range(3):
            cs = [cent[order[i], a] for i in range(first, first + count)]
            mn, mx = F(1e30), F(-1e30)
            for c in cs:
                mn = mn if mn < c else c
                mx = c if mx < c else mx
            if mn == mx:
                continue
            bins = [Box() for _ in range(BINS)]
            cnt = [0] * BINS
            scale = F(F(BINS) / F(mx - mn))
            for i in range(first, first + count):
                f = np.floor(F(F(cent[order[i], a] - mn) * scale))
                bi = int(min(F(BINS - 1), f))
                cnt[bi] += 1
                for k in range(3):
                    bins[bi].grow_point(tri_pos[order[i], k])
            la, ra = [F(0)] * (BINS - 1), [F(0)] * (BINS - 1)
            lc, rc = [F(0)] * (BINS - 1), [F(0)] * (BINS - 1)
            lb, rb = Box(), Box()
            ls = rs = 0
            for i in range(BINS - 1):
                ls += cnt[i]; lc[i] = F(ls)
                lb.grow_box(bins[i]); la[i] = lb.area()
                rs += cnt[BINS - 1 - i]; rc[BINS - 2 - i] = F(rs)
                rb.grow_box(bins[BINS - 1 - i])
                ra[i] = rb.area()               # the quirk of src/vk_engine.cpp:1321
                ra[BINS - 2 - i] = rb.area()
            scale = F(F(mx - mn) / F(BINS))
            for i in range(BINS - 1):
                with np.errstate(over="ignore", invalid="ignore"):
                    cost = F(F(lc[i] * la[i]) + F(rc[i] * ra[i]))
                if cost < best:
                    axis, pos, best = a, F(mn + F(scale * F(i + 1))), cost
        return best, axis, pos

    def subdivide(idx, depth):
        first, count = nodes[idx][6], nodes[idx][7]
        if count <= 2 or depth >= 64:
            return
        best, axis, pos = find_split(first, count)
        nd = nodes[idx]
        x = F(nd[1] - nd[0]); y = F(nd[3] - nd[2]); z = F(nd[5] - nd[4])
        parent_area = F(F(F(x * y) + F(y * z)) + F(z * x))
        if best >= F(F(count) * parent_area):
            return
        i, j = first, first + count - 1
        while i <= j:
            if cent[order[i], axis] < pos:
                i += 1
            else:
                order[i], order[j] = order[j], order[i]
                j -= 1
        left = i - first
        if left == 0 or left == count:
            return
        child = used[0]
        used[0] += 2
        set_node(child, first, left)
        set_node(child + 1, i, count - left)
        nodes[idx][6], nodes[idx][7] = child, 0
        subdivide(child, depth + 1)
        subdivide(child + 1, depth + 1)

    set_node(0, 0, n)
    subdivide(0, 0)
    return nodes[: used[0]], order
