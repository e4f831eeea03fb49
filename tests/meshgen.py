"""Procedural OBJ / BMP inputs for the mesh-path tests (the reference's objs/*.obj are not available on
the GPU box; these files are generated, not copied)."""
import struct

import numpy as np


def write_cube_obj(path, h=0.1):
    """axis-aligned cube +-h in the 'v//vn' face format (the shape of the reference's objs/cube.obj: 8 v, 12 tris)"""
    v = [(-h, -h, -h), (-h, -h, h), (-h, h, h), (-h, h, -h), (h, h, -h), (h, h, h), (h, -h, h), (h, -h, -h)]
    n = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1)]
    f = [(1, 2, 4, 4), (4, 2, 3, 4), (8, 1, 5, 6), (5, 1, 4, 6), (5, 6, 8, 1), (8, 6, 7, 1), (6, 3, 7, 3), (7, 3, 2, 3),
         (4, 3, 5, 2), (5, 3, 6, 2), (1, 8, 2, 5), (2, 8, 7, 5)]
    with open(path, "w") as fp:
        for p in v:
            fp.write("v %g %g %g\n" % p)
        fp.write("\n")
        for q in n:
            fp.write("vn %g %g %g\n" % q)
        fp.write("\n")
        for a, b, c, k in f:
            fp.write("f %d//%d %d//%d %d//%d\n" % (a, k, b, k, c, k))
    return path


def write_sphere_obj(path, rings=24, segs=32, radius=(0.9, 1.3, 0.7), center=(0.3, 1.0, -0.2), quads=True):
    """textured ellipsoid in the 'v/vt/vn' format; optionally quads (exercise the fan triangulation)"""
    vs, vts, faces = [], [], []
    for i in range(rings + 1):
        th = np.pi * i / rings
        for j in range(segs + 1):
            ph = 2 * np.pi * j / segs
            vs.append((center[0] + radius[0] * np.sin(th) * np.cos(ph), center[1] + radius[1] * np.cos(th),
                       center[2] + radius[2] * np.sin(th) * np.sin(ph)))
            vts.append((j / segs * 0.999, i / rings * 0.999))
    idx = lambda i, j: i * (segs + 1) + j + 1
    for i in range(rings):
        for j in range(segs):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            if quads and 0 < i < rings - 1 and (i + j) % 3 == 0:
                faces.append((a, b, c, d))
            else:
                if i > 0:
                    faces.append((a, b, d))
                if i < rings - 1:
                    faces.append((b, c, d))
    with open(path, "w") as fp:
        fp.write("# generated\n")
        for p in vs:
            fp.write("v %.6f %.6f %.6f\n" % p)
        for t in vts:
            fp.write("vt %.6f %.6f\n" % t)
        fp.write("vn 0 1 0\n")
        for f in faces:
            fp.write("f " + " ".join("%d/%d/1" % (k, k) for k in f) + "\n")
    return path


def write_soup_obj(path, n=300, seed=0):
    """random triangles of assorted sizes/orientations (plain 'f a b c' format)"""
    rng = np.random.default_rng(seed)
    with open(path, "w") as fp:
        for t in range(n):
            c = rng.random(3) * 2 - 1
            s = 10 ** rng.uniform(-2.0, -0.3)
            for _ in range(3):
                p = c + rng.normal(size=3) * s
                fp.write("v %.6f %.6f %.6f\n" % tuple(p))
        # a few exactly axis-aligned triangles (degenerate plane components)
        fp.write("v 0.5 0.5 0.5\nv 0.5 0.9 0.5\nv 0.5 0.5 0.9\nv -0.5 0.2 0.1\nv 0.1 0.2 0.1\nv -0.5 0.2 0.7\n")
        for t in range(n + 2):
            fp.write("f %d %d %d\n" % (3 * t + 1, 3 * t + 2, 3 * t + 3))
    return path


def write_bmp(path, w=64, h=64, seed=0):
    """24-bit BMP, rows as stored (Scene::loadBMP ignores row order/padding; w*3 is a multiple of 4 here)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx // 8 + yy // 8) % 2) * 200 + 30], -1).astype(np.uint8)
    img[rng.random((h, w)) < 0.05] = 255
    data = img[..., ::-1].tobytes()  # BGR
    hdr = b"BM" + struct.pack("<IHHI", 54 + len(data), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(data), 2835, 2835, 0, 0)
    with open(path, "wb") as fp:
        fp.write(hdr + data)
    return path


# Well-formed corner cases of the OBJ grammar the reference's loader accepts (objloader.cpp:24-130):
# 'v/vt' faces, 3-component vt, comment/group/material lines, exponents, a w component, CRLF line ends,
# a file without a trailing newline, faces with more than four vertices (dropped by buildVBOs).
QUIRK_OBJS = {
    "vt_only": "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0.5\nvt 0 0\nvt 1 0\nvt 0 1\nvt 1 1\nf 1/1 2/2 3/3\nf 2/2 4/4 3/3\n",
    "quad_tex": "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0.25\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\n",
    "quad_plain": "v 0 0 0\nv 2 0 0\nv 2 0 2\nv 0 0.5 2\nf 1 2 3 4\nf 4 3 2 1\n",
    "poly5": "v 0 0 0\nv 1 0 0\nv 1.5 1 0\nv 0.5 2 0.3\nv -0.5 1 0\nf 1 2 3 4 5\nf 1 2 3\n",
    "comments": "# hello\n\no thing\ng grp\nusemtl m\ns off\nv 0 0 0\nv 1 0 0\nv 0 1 0\n# mid\nf 1 2 3\n",
    "sci": "v 1e-1 0 0\nv 1.5E0 2e-2 0\nv 0 1 -3.25e-1\nf 1 2 3\n",
    "vw": "v 0 0 0 1\nv 1 0 0 1\nv 0 1 0 1\nf 1 2 3\n",
    "vt3": "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0 0\nvt 1 0 0\nvt 0 1 0\nf 1/1 2/2 3/3\n",
    "crlf": "v 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\n",
    "noeol": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3",
    "neg_y": "v 0 -3 0\nv 1 -2 5\nv 0 1 7\nf 1 2 3\n",
    "tetra": "v 0 0 0\nv 2 0 0\nv 0 2 0\nv 0 0 2\nf 1 2 3\nf 1 3 4\nf 1 4 2\nf 2 4 3\n",
}


def write_generated_objs(directory):
    """every generated OBJ of the loader tests -> {name: path} (deterministic bytes)"""
    import os
    out = {"cube": write_cube_obj(os.path.join(directory, "cube.obj")),
           "sphere": write_sphere_obj(os.path.join(directory, "sphere.obj")),
           "sphere_tris": write_sphere_obj(os.path.join(directory, "sphere_tris.obj"), rings=9, segs=7, quads=False),
           "soup": write_soup_obj(os.path.join(directory, "soup.obj"))}
    for name, text in QUIRK_OBJS.items():
        path = os.path.join(directory, "quirk_%s.obj" % name)
        with open(path, "wb") as fp:
            fp.write(text.encode())
        out["quirk_" + name] = path
    return out


def write_colonnade_obj(path, n_cols=8, length=40.0, col_radius=0.15, col_height=4.0, segs=12, z_off=1.5,
                        beam=0.2, slab=None, floor=None):
    """Stand-in for config 5's crytek-sponza (sponza.obj is not in the reference checkout): two rows of
    n_cols prismatic columns (quad sides, triangle-fan caps) along x, a lintel box on each row and
    optionally a floor slab (x half-width, z half-width) or a floor of nx x nz QUADS (x0, x1, z0, z1[, nx, nz[, y]]) in a plane y = const (one face: the voxel
    grid is N cells along each axis of the mesh's own box, so a surface costs its share of that box's FACE times N^2 voxels, whatever
    its size in metres); textured ('v/vt' faces).  x is the longest axis,
    so the voxel grid edge is `length`."""
    vs, vts, faces = [], [], []

    def add_v(p, t):
        vs.append(p)
        vts.append(t)
        return len(vs)

    def box(x0, x1, y0, y1, z0, z1):
        c = [add_v((x, y, z), ((x - x0) / max(x1 - x0, 1e-9) * 0.999, (y - y0 + z - z0) / max(y1 - y0 + z1 - z0, 1e-9) * 0.999))
             for x in (x0, x1) for y in (y0, y1) for z in (z0, z1)]
        for a, b, cc, d in ((0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)):
            faces.append((c[a], c[b], c[cc], c[d]))

    pitch = length / n_cols
    for row, z in enumerate((-z_off, z_off)):
        for k in range(n_cols):
            cx = -length / 2 + pitch * (k + 0.5)
            ring0, ring1 = [], []
            for s in range(segs):
                a = 2 * np.pi * s / segs + 0.1 * row
                px, pz = cx + col_radius * np.cos(a), z + col_radius * np.sin(a)
                ring0.append(add_v((px, 0.0, pz), (s / segs * 0.999, 0.0)))
                ring1.append(add_v((px, col_height, pz), (s / segs * 0.999, 0.999)))
            top = add_v((cx, col_height, z), (0.5, 0.5))
            for s in range(segs):
                t = (s + 1) % segs
                faces.append((ring0[s], ring0[t], ring1[t], ring1[s]))
                faces.append((ring1[s], ring1[t], top))
        if beam:
            box(-length / 2, length / 2, col_height, col_height + beam, z - beam / 2, z + beam / 2)
    if slab:
        box(-slab[0], slab[0], -0.05, 0.0, -slab[1], slab[1])
    if floor:
        # tiles, each with its own texture coordinates: a triangle takes ONE colour, the texel at its first vertex (voxelization.cu:113-127)
        x0, x1, z0, z1 = floor[:4]
        nx, nz = (floor[4], floor[5]) if len(floor) > 4 else (1, 1)
        fy = floor[6] if len(floor) > 6 else 0.0
        for i in range(nx):
            for j in range(nz):
                xa, xb = x0 + (x1 - x0) * i / nx, x0 + (x1 - x0) * (i + 1) / nx
                za, zb = z0 + (z1 - z0) * j / nz, z0 + (z1 - z0) * (j + 1) / nz
                uv = (((i * 37 + j * 11) % 64) / 64.0 + 0.004, ((i * 13 + j * 29) % 64) / 64.0 + 0.004)
                c = [add_v((x, fy, z), (uv[0] + 0.003 * k, uv[1] + 0.002 * k)) for k, (x, z) in enumerate(((xa, za), (xa, zb), (xb, zb), (xb, za)))]
                faces.append((c[0], c[1], c[2], c[3]))
    with open(path, "w") as fp:
        fp.write("# generated colonnade\n")
        for p in vs:
            fp.write("v %.6f %.6f %.6f\n" % p)
        for t in vts:
            fp.write("vt %.6f %.6f\n" % t)
        for f in faces:
            fp.write("f " + " ".join("%d/%d" % (k, k) for k in f) + "\n")
    return path
