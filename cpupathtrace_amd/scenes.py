"""Scene construction for tests, smoke() and bench.py: the callers' side of the hot path, in numpy.

The reference builds its scenes in application code (demo/main.cpp, benchmark/main.cpp, test/render_test.cpp) with
``makePlane`` / ``makeBox`` / ``io::loadMesh`` and hands ``Scene::Scene`` a vector of objects.  This module restates those
callers so the same flat scene description (objects in construction order) can be given to the C-ABI (include/pt_hip.h),
to the CPU oracle and to the compiled reference.  All arithmetic is done in float32, one rounding per operation, in the
reference's order, so the triangles are bit-identical to what the reference's generators produce
(checked against the reference's own makePlane/makeBox/mat4 in tests/test_oracle_vs_reference.py).

Citations are file:line under /root/reference.
"""
import ctypes
import ctypes.util

import numpy as np

F = np.float32

OBJ_TRIANGLE, OBJ_SPHERE = 0, 1
BSDF_LAMBERTIAN, BSDF_GLASS, BSDF_MIRROR = 0, 1, 2
APERTURE_NONE, APERTURE_CIRCULAR, APERTURE_HEXAGONAL = 0, 1, 2
NO_MATERIAL = 0xFFFFFFFF

MATERIAL_DTYPE = np.dtype([("diffuse", "<f4", 4), ("specular", "<f4", 4), ("emission", "<f4", 4), ("ior", "<f4"), ("bsdf", "<i4"),
                           ("one_way", "<i4"), ("pad", "<i4")])

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.cosf.restype = ctypes.c_float
_libm.cosf.argtypes = [ctypes.c_float]
_libm.sinf.restype = ctypes.c_float
_libm.sinf.argtypes = [ctypes.c_float]


def make_plane(a, b, cull=False):
    """makePlane, src/scene/mesh.cpp:294-337.  Returns (n, 3, 3) float32 triangle corners (0 or 2 triangles)."""
    a = np.asarray(a, dtype=F)
    b = np.asarray(b, dtype=F)
    eps = F(1e-4)
    plane_dim = -1
    for i in range(3):
        if np.abs(a[i] - b[i]) < eps:
            plane_dim = i
    others_separate = True
    for i in range(3):
        if i == plane_dim:
            continue
        if np.abs(a[i] - b[i]) < eps:
            others_separate = False
    if plane_dim < 0 or not others_separate:
        return np.zeros((0, 3, 3), dtype=F)
    dim1 = 1 if plane_dim == 0 else 0
    v2 = a.copy()
    v4 = b.copy()
    v2[dim1] = b[dim1]
    v4[dim1] = a[dim1]
    return np.array([[a, v2, b], [b, v4, a]], dtype=F)


def make_box(a, b, cull=False):
    """makeBox, src/scene/mesh.cpp:339-375.  Returns (n, 3, 3) float32 (0 or 12 triangles)."""
    a = np.asarray(a, dtype=F)
    b = np.asarray(b, dtype=F)
    eps = F(1e-4)
    for i in range(3):
        if np.abs(a[i] - b[i]) < eps:
            return np.zeros((0, 3, 3), dtype=F)
    out = []
    for i in range(3):
        plane_a = a.copy()
        plane_b = a.copy()
        for dim in range(3):
            if dim == i:
                continue
            plane_a[dim] = a[dim]
            plane_b[dim] = b[dim]
        out.append(make_plane(plane_a, plane_b, cull))
        plane_a[i] = b[i]
        plane_b[i] = b[i]
        out.append(make_plane(plane_a, plane_b, cull))
    return np.concatenate(out, axis=0)


def mat4_apply(m, pts):
    """mat4<float>::operator*(vec3), include/PathTrace/util/matrix.h:49-56: affine product, then multiply by 1/w."""
    m = np.asarray(m, dtype=F).reshape(4, 4)
    p = np.asarray(pts, dtype=F).reshape(-1, 3)
    rows = []
    for r in range(4):
        acc = np.zeros(len(p), dtype=F)  # dot() accumulates from 0, vector.h:193-201
        acc = acc + m[r, 0] * p[:, 0]
        acc = acc + m[r, 1] * p[:, 1]
        acc = acc + m[r, 2] * p[:, 2]
        acc = acc + m[r, 3] * F(1.0)
        rows.append(acc)
    inv_w = F(1.0) / rows[3]
    return np.stack([rows[0] * inv_w, rows[1] * inv_w, rows[2] * inv_w], axis=1).reshape(np.shape(pts))


def _dot3(a, b):
    acc = np.zeros(a.shape[:-1], dtype=F)
    for k in range(3):
        acc = acc + a[..., k] * b[..., k]
    return acc


def _normalize(v):
    inv = F(1.0) / np.sqrt(_dot3(v, v))  # vector.h:161-167
    return v * inv[..., None]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def face_normals(tri):
    """Triangle::Triangle, src/scene/object.cpp:118-124: all three vertex normals = normalize((b-a) x (c-a))."""
    tri = np.asarray(tri, dtype=F).reshape(-1, 3, 3)
    with np.errstate(all="ignore"):
        fn = _normalize(_cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]))
    return np.repeat(fn[:, None, :], 3, axis=1)


class SceneBuilder:
    """Collects objects in construction order, like the std::vector handed to Scene::Scene (scene.h:32)."""

    def __init__(self):
        self.kind = []
        self.tri_pos, self.tri_nrm, self.tri_cull, self.tri_mat = [], [], [], []
        self.sph, self.sph_mat = [], []
        self.materials = []
        self.light_pos, self.light_spec = [], []

    def material(self, diffuse=(1, 1, 1, 1), ior=1.0, emission=(0, 0, 0, 0), bsdf=BSDF_LAMBERTIAN, one_way=False, specular=(1, 1, 1, 1)):
        """ConstantMaterial(diffuse, ior, emission) + BSDF -> ConstantMaterialHandler (material.h:53-68, object.h:26-40)."""
        m = np.zeros((), dtype=MATERIAL_DTYPE)
        m["diffuse"], m["specular"], m["emission"] = diffuse, specular, emission
        m["ior"], m["bsdf"], m["one_way"] = ior, bsdf, 1 if one_way else 0
        self.materials.append(m)
        return len(self.materials) - 1

    def triangles(self, tri, material=NO_MATERIAL, cull=False, normals=None):
        tri = np.asarray(tri, dtype=F).reshape(-1, 3, 3)
        n = len(tri)
        if n == 0:
            return
        self.kind.append(np.full(n, OBJ_TRIANGLE, dtype=np.uint8))
        self.tri_pos.append(tri.reshape(n, 9))
        nrm = face_normals(tri) if normals is None else np.asarray(normals, dtype=F).reshape(n, 3, 3)
        self.tri_nrm.append(nrm.reshape(n, 9))
        self.tri_cull.append(np.full(n, 1 if cull else 0, dtype=np.uint8))
        self.tri_mat.append(np.full(n, material, dtype=np.uint32))

    def sphere(self, origin, radius, material=NO_MATERIAL):
        self.kind.append(np.array([OBJ_SPHERE], dtype=np.uint8))
        self.sph.append(np.array([origin[0], origin[1], origin[2], radius], dtype=F))
        self.sph_mat.append(np.uint32(material))

    def point_light(self, pos, spectrum):
        self.light_pos.append(np.asarray(pos, dtype=F))
        self.light_spec.append(np.asarray(spectrum, dtype=F))

    def build(self):
        def cat(parts, shape, dtype):
            return np.concatenate(parts, axis=0).astype(dtype, copy=False) if parts else np.zeros(shape, dtype=dtype)

        return {
            "obj_kind": cat(self.kind, (0,), np.uint8),
            "tri_pos": cat(self.tri_pos, (0, 9), F),
            "tri_nrm": cat(self.tri_nrm, (0, 9), F),
            "tri_cull": cat(self.tri_cull, (0,), np.uint8),
            "tri_material": cat(self.tri_mat, (0,), np.uint32),
            "sph": np.array(self.sph, dtype=F).reshape(-1, 4),
            "sph_material": np.array(self.sph_mat, dtype=np.uint32).reshape(-1),
            "materials": np.array(self.materials, dtype=MATERIAL_DTYPE).reshape(-1),
            "light_pos": np.array(self.light_pos, dtype=F).reshape(-1, 3),
            "light_spectrum": np.array(self.light_spec, dtype=F).reshape(-1, 4),
        }


def camera(origin, look_at, up, focal_length, height, aspect_ratio, aperture_width=0.0, aperture_height=0.0, aperture_kind=APERTURE_NONE,
           hex_ratio=0.0, focal_plane_dist=0.0):
    """Arguments of Camera::Camera, include/PathTrace/camera.h:92,108-109."""
    return dict(origin=origin, look_at=look_at, up=up, focal_length=focal_length, height=height, aspect_ratio=aspect_ratio,
                aperture_width=aperture_width, aperture_height=aperture_height, aperture_kind=aperture_kind, hex_ratio=hex_ratio,
                focal_plane_dist=focal_plane_dist)


def options(width, height, min_spp, max_spp, epsilon=1e-3):
    """RenderOptions, include/PathTrace/worker.h:14-31."""
    return dict(image_width=width, image_height=height, min_sample_count=min_spp, max_sample_count=max_spp, epsilon=epsilon)


# ---------------------------------------------------------------------------------------------------------------------
# The reference's own scenes
# ---------------------------------------------------------------------------------------------------------------------

def empty_scene():
    """test/render_test.cpp:14-29."""
    return SceneBuilder().build(), camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 1.0, 1.0, 1.0)


def simple_scene():
    """test/render_test.cpp:31-52: default-material sphere + point light."""
    sb = SceneBuilder()
    sb.point_light((0.0, 1.0, 0.0), (1, 1, 1, 1))
    sb.sphere((0.0, 0.0, 0.6), 0.5)
    return sb.build(), camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 0.1, 1.0, 1.0)


def advanced_scene():
    """test/render_test.cpp:54-90: glass sphere (IOR left at 1), emissive Lambertian sphere, ground triangle, point light."""
    sb = SceneBuilder()
    sb.point_light((0.0, 1.0, 0.0), (1, 1, 1, 1))
    sb.sphere((0.1, 0.1, 1.0), 0.5, sb.material((1.0, 1.0, 1.0, 1.5), bsdf=BSDF_GLASS))
    sb.sphere((-0.1, 0.2, 2.0), 0.6, sb.material((0.8, 0.4, 0.6, 1.0), 1.0, (0.2, 0.1, 0.3, 1.0)))
    sb.triangles([[(5.0, -1.0, 5.0), (0.0, -1.0, -5.0), (-5.0, -1.0, 5.0)]], sb.material((0.4, 0.6, 0.4, 1.0)))
    return sb.build(), camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 0.2, 0.5, 1.94)


def two_spheres_scene():
    """test/scene/scene_test.cpp:8-47."""
    sb = SceneBuilder()
    sb.sphere((-1.0, -1.0, -1.0), 1.0)
    sb.sphere((1.0, 1.0, 1.0), 1.0)
    return sb.build()


def box_scene(aspect_ratio=-1.0):
    """benchmark/main.cpp:34-57 renderSceneBox: unit box + two-sided emissive quad under the ceiling."""
    sb = SceneBuilder()
    sb.triangles(make_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)))
    light = sb.material((1, 1, 1, 1), 1.0, (1, 1, 1, 1))
    sb.triangles(make_plane((-0.25, F(1.0) - F(0.01), -0.25), (0.25, F(1.0) - F(0.01), 0.25)), light)
    return sb.build(), camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, aspect_ratio)


def oneway_mirror_scene(aspect_ratio=-1.0):
    """Box of benchmark/main.cpp:34-57 with the two mirror kinds of src/scene/propagation.cpp:178-217 in it: a MirrorBRDF(one_way = true)
    pane across the middle of the box (reflects rays that meet its front, lets rays from behind pass: propagation.cpp:186-190,210-212),
    a tilted two-sided mirror quad behind it and a two-sided mirror sphere in front of it, so that paths meet the pane from both sides."""
    sb = SceneBuilder()
    sb.triangles(make_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), sb.material((0.9, 0.8, 0.7, 1.0)))
    light = sb.material((1, 1, 1, 1), 1.0, (1, 1, 1, 1))
    sb.triangles(make_plane((-0.25, F(1.0) - F(0.01), -0.25), (0.25, F(1.0) - F(0.01), 0.25)), light)
    pane = sb.material((1, 1, 1, 1), bsdf=BSDF_MIRROR, one_way=True, specular=(0.9, 0.95, 1.0, 1.0))
    sb.triangles(make_plane((-0.7, -0.8, 0.1), (0.6, 0.5, 0.1)), pane)                      # normal along z: front faces the camera or not by winding
    sb.triangles([[(-0.9, -0.9, 0.8), (0.9, -0.9, 0.5), (0.0, 0.9, 0.9)]], sb.material((1, 1, 1, 1), bsdf=BSDF_MIRROR, specular=(1.0, 0.9, 0.8, 1.0)))
    sb.sphere((0.45, -0.55, -0.4), 0.3, sb.material((0, 0, 1, 1), bsdf=BSDF_MIRROR, one_way=True))
    sb.sphere((-0.5, -0.6, -0.3), 0.25, sb.material((0, 0, 1, 1), bsdf=BSDF_MIRROR))
    return sb.build(), camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, aspect_ratio)


def dragon_box_scene(mesh_pos, mesh_nrm, aspect_ratio=-1.0, copies=1):
    """benchmark/main.cpp:59-105 renderSceneDragonBox with `mesh` in the dragon's place (glass, IOR 1.5, two-sided).

    mesh_pos/mesh_nrm are already transformed as io::loadMesh(path, T, false, true) would return them."""
    sb = SceneBuilder()
    sb.triangles(make_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)))
    light = sb.material((1, 1, 1, 1), 1.0, (1, 1, 1, 1))
    sb.triangles(make_plane((-0.25, F(1.0) - F(0.01), -0.25), (0.25, F(1.0) - F(0.01), 0.25)), light, cull=True)
    glass = sb.material((1, 1, 1, 1), 1.5, bsdf=BSDF_GLASS)
    sb.triangles(mesh_pos, glass, cull=False, normals=mesh_nrm)
    return sb.build(), camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, aspect_ratio)


def dragon_grid_scene(mesh_pos, mesh_nrm, aspect_ratio=-1.0, grid=4):
    """BASELINE.json configs[4]: grid x grid transformed COPIES of the mesh (the reference has no instancing: every copy is its
    own set of Triangle objects), one per unit cell of a wall of cells in the x-y plane, inside DragonBox
    (benchmark/main.cpp:59-105) widened `grid` times in x and y.  Camera, light quad and viewing distance are DragonBox's scaled
    the same way, so the box fills the frame exactly as it does there."""
    sb = SceneBuilder()
    half = F(grid)  # box half-extent in x and y: one 2 x 2 cell per copy
    sb.triangles(make_box((-half, -half, -1.0), (half, half, 1.0)))
    light = sb.material((1, 1, 1, 1), 1.0, (1, 1, 1, 1))
    sb.triangles(make_plane((-0.25 * grid, half - F(0.01), -0.25), (0.25 * grid, half - F(0.01), 0.25)), light, cull=True)
    glass = sb.material((1, 1, 1, 1), 1.5, bsdf=BSDF_GLASS)
    pos = np.asarray(mesh_pos, dtype=F).reshape(-1, 3, 3)
    for gy in range(grid):
        for gx in range(grid):
            shift = np.array([F(2 * gx + 1) - half, F(2 * gy + 1) - half, 0], dtype=F)
            sb.triangles(pos + shift, glass, cull=False, normals=mesh_nrm)
    cam = camera((0, 0, -1.0 - 2.0 * grid), (0, 0, 0), (0, 1, 0), 1.0, 1.0, aspect_ratio)
    return sb.build(), cam


DRAGON_BOX_TRANSFORM = [[0.01, 0, 0, 0], [0, 0.01, 0, -0.5], [0, 0, 0.01, 0], [0, 0, 0, 1]]  # benchmark/main.cpp:80-83
DEMO_DRAGON_TRANSFORM = [[0.005, 0, 0, 0.4], [0, 0.005, 0, -0.8], [0, 0, 0.005, -0.75], [0, 0, 0, 1]]  # demo/main.cpp:141-144


def cornell_scene(width=256, height=256, mesh_pos=None, mesh_nrm=None):
    """demo/main.cpp:47-203: Cornell box, mirror sphere, rotated tall box, thin-lens camera; the dragon is optional because
    assets/xyzrgb_dragon.obj is not part of the reference mount (.MISSING_LARGE_BLOBS)."""
    epsilon = F(1.0e-3)
    aspect_ratio = F(width) / F(height)
    cam = camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, float(-aspect_ratio), 0.05, 0.05, APERTURE_CIRCULAR, 0.0, 3.5)
    sb = SceneBuilder()
    ground_y, ceiling_y, walls_x, walls_z = F(-1.0), F(1.0), F(1.0), F(1.0)

    ground = make_plane((20.0, ground_y, -20.0), (-20.0, ground_y, 20.0), True)
    ceiling = make_plane((-20.0, ceiling_y, -20.0), (20.0, ceiling_y, 20.0), True)
    ceiling_light = make_plane((-0.25, ceiling_y - epsilon, -0.25), (0.25, ceiling_y - epsilon, 0.25), True)
    walls = [
        (make_plane((-walls_x, ground_y, -walls_z), (walls_x, ceiling_y, -walls_z), True), (0, 0, 1, 1)),
        (make_plane((-walls_x, ground_y, -walls_z), (-walls_x, ceiling_y, walls_z), True), (1, 0, 0, 1)),
        (make_plane((walls_x, ground_y, walls_z), (-walls_x, ceiling_y, walls_z), True), (1, 1, 1, 1)),
        (make_plane((walls_x, ground_y, walls_z), (walls_x, ceiling_y, -walls_z), True), (0, 1, 0, 1)),
    ]
    wall_mats = [sb.material(color) for _, color in walls]
    ground_mat = sb.material((1, 1, 1, 1))
    ceiling_mat = sb.material((1, 1, 1, 1))
    light_mat = sb.material((1, 1, 1, 1), 1.0, (1, 1, 1, 1))
    # moveObjects order, demo/main.cpp:133-136
    sb.triangles(ground, ground_mat, cull=True)
    sb.triangles(ceiling, ceiling_mat, cull=True)
    sb.triangles(ceiling_light, light_mat, cull=True)
    for (tris, _), m in zip(walls, wall_mats):
        sb.triangles(tris, m, cull=True)

    if mesh_pos is not None:
        sb.triangles(mesh_pos, sb.material((1, 1, 1, 1), 1.5, bsdf=BSDF_GLASS), cull=False, normals=mesh_nrm)

    radius = F(0.5)
    sb.sphere((0.5, F(-1.0) + radius, 0.5), radius, sb.material((0, 0, 1, 1), bsdf=BSDF_MIRROR, one_way=False))

    box = make_box(np.array([-1, -1, -1], dtype=F) * F(0.3), np.array([1, 1, 1], dtype=F) * F(0.3))
    rot_y = 0.25
    c, s = _libm.cosf(rot_y), _libm.sinf(rot_y)  # std::cos/std::sin on a float, demo/main.cpp:186-189
    transformation = [[c, 0.0, s, -0.5], [0.0, 3.0, 0.0, -0.25], [-s, 0.0, c, 0.5], [0.0, 0.0, 0.0, 1.0]]
    sb.triangles(mat4_apply(transformation, box.reshape(-1, 3)).reshape(-1, 3, 3), sb.material((1, 1, 1, 1)))
    return sb.build(), cam


# ---------------------------------------------------------------------------------------------------------------------
# Procedural stand-in for assets/xyzrgb_dragon.obj (absent from the reference mount)
# ---------------------------------------------------------------------------------------------------------------------

def bumpy_sphere_vertices(nu, nv, radius=40.0, bump=0.2, centre=(0.0, 50.0, 0.0)):
    """The stand-in mesh as an indexed mesh in OBJ terms: float32 vertices (untransformed) and 0-based triangle faces."""
    theta = (np.arange(nu, dtype=np.float64) * (2.0 * np.pi / nu))
    phi = (np.arange(1, nv, dtype=np.float64) * (np.pi / nv))
    tt, pp = np.meshgrid(theta, phi, indexing="xy")  # (nv-1, nu)
    r = radius * (1.0 + bump * np.sin(5.0 * tt) * np.sin(7.0 * pp))
    ring = np.stack([r * np.sin(pp) * np.cos(tt) + centre[0], r * np.cos(pp) + centre[1], r * np.sin(pp) * np.sin(tt) + centre[2]], axis=-1)
    verts = np.concatenate([[[centre[0], centre[1] + radius, centre[2]]], ring.reshape(-1, 3), [[centre[0], centre[1] - radius, centre[2]]]])
    verts = verts.astype(F)
    n_ring = nv - 1
    top, bottom = 0, len(verts) - 1

    def vid(j, i):
        return 1 + j * nu + (i % nu)

    i = np.arange(nu)
    faces = [np.stack([np.full(nu, top), vid(0, i + 1), vid(0, i)], axis=1)]
    for j in range(n_ring - 1):
        a, b, c, d = vid(j, i), vid(j, i + 1), vid(j + 1, i), vid(j + 1, i + 1)
        faces.append(np.stack([a, b, d], axis=1))
        faces.append(np.stack([a, d, c], axis=1))
    faces.append(np.stack([np.full(nu, bottom), vid(n_ring - 1, i), vid(n_ring - 1, i + 1)], axis=1))
    return verts, np.concatenate(faces, axis=0)


def bumpy_sphere_mesh(nu, nv, transform=None, radius=40.0, bump=0.2, centre=(0.0, 50.0, 0.0)):
    """Closed UV sphere r = radius * (1 + bump * sin(5 theta) * sin(7 phi)) with 2*nu*(nv-1) triangles and smooth vertex
    normals computed the way io::loadMesh(..., smooth=true) does (normalised sum of the adjacent faces' unit normals,
    src/scene/mesh.cpp:228-267).  nu = nv = 1900 gives 7,216,200 triangles, the scale of the XYZ RGB dragon."""
    verts, faces = bumpy_sphere_vertices(nu, nv, radius, bump, centre)
    if transform is not None:
        verts = mat4_apply(transform, verts)

    tri = verts[faces]  # (n, 3, 3)
    with np.errstate(all="ignore"):
        fn = _normalize(_cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]))
    vn = np.zeros((len(verts), 3), dtype=np.float64)
    for k in range(3):
        for comp in range(3):
            vn[:, comp] += np.bincount(faces[:, k], weights=fn[:, comp], minlength=len(verts))
    vn = vn.astype(F)
    with np.errstate(all="ignore"):
        vn = _normalize(vn)
    nrm = vn[faces]
    return tri.astype(F), nrm.astype(F)
