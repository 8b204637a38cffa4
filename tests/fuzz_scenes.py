"""Seeded random scenes that exercise EVERY primitive kind (sphere, plane, quad, cube, mesh) and EVERY material kind,
built directly as ABI arrays (through the oracle-side f32 helpers for matrices / quads)."""
import ctypes as C

import numpy as np


def random_scene(abi, host, seed, exact_only, n_prims=14, mesh_tris=60, only_kinds=None, lambert_only=False, identity_meshes=False, no_metal=False):
    from oracle import scene_loader as L
    rng = np.random.default_rng(seed)
    F = np.float32
    mats = []

    def mat(kind, albedo=(0, 0, 0), aux=(0, 0, 0), p0=0.0, eta=(0, 0, 0), k=(0, 0, 0)):
        m = abi.Material(); m.kind = kind
        m.albedo[:] = [float(F(v)) for v in albedo]; m.aux[:] = [float(F(v)) for v in aux]; m.p0 = float(F(p0))
        m.eta[:] = [float(F(v)) for v in eta]; m.k[:] = [float(F(v)) for v in k]
        mats.append(m); return len(mats) - 1

    col = lambda lo=0.1, hi=0.95: tuple(rng.uniform(lo, hi, 3))
    kinds = [mat(abi.MAT_LAMBERT_SOLID, col()), mat(abi.MAT_LAMBERT_CHECKER, col(), col(), p0=1.0 / rng.uniform(0.2, 1.5)),
             mat(abi.MAT_METAL, col(), p0=0.0), mat(abi.MAT_METAL, col(), p0=rng.uniform(0.05, 0.6)),
             mat(abi.MAT_DIELECTRIC, p0=rng.uniform(1.2, 1.9)), mat(abi.MAT_EMISSIVE, tuple(rng.uniform(1, 6, 3))),
             mat(abi.MAT_PLASTIC, col(), p0=rng.uniform(1.2, 1.8)), mat(abi.MAT_NULL), mat(abi.MAT_LAMBERT_SOLID, col())]
    if lambert_only:                                                 # what the Lambert-only lockstep kernel (k_render_ctr_simple) is picked for:
        del mats[:]                                                  # the materials ARRAY holds nothing but Lambert (solid) / Emissive / Null
        kinds = [mat(abi.MAT_LAMBERT_SOLID, col()), mat(abi.MAT_EMISSIVE, tuple(rng.uniform(1, 6, 3))), mat(abi.MAT_NULL), mat(abi.MAT_LAMBERT_SOLID, col())]
    if no_metal:                                                     # the wavefront kernels' pruned instantiations: no MI355RT_MAT_METAL in the list
        kinds = [k for k in kinds if mats[k].kind != abi.MAT_METAL]
    if not exact_only:
        cu = ((0.2, 1.09, 1.42), (3.91, 2.57, 2.30))
        kinds += [mat(abi.MAT_ROUGH_GGX, col(), p0=rng.uniform(0.02, 0.5), eta=cu[0], k=cu[1]),
                  mat(abi.MAT_ROUGH_BECKMANN, col(), p0=rng.uniform(0.02, 0.5), eta=cu[0], k=cu[1])]

    def matrix():
        q = L.quat_from_euler_yxz_deg(F(rng.uniform(-180, 180)), F(rng.uniform(-180, 180)), F(rng.uniform(-180, 180)))
        return L.mat4_from_scale_rotation_translation([F(v) for v in rng.uniform(0.4, 2.0, 3)], q, [F(v) for v in rng.uniform(-3, 3, 3)])

    prims, tri_chunks, meshes = [], [], []
    if only_kinds is not None:
        order = [only_kinds[i % len(only_kinds)] for i in range(n_prims)]
    else:
        order = [abi.PRIM_SPHERE, abi.PRIM_PLANE, abi.PRIM_QUAD, abi.PRIM_CUBE, abi.PRIM_MESH] + list(rng.integers(0, 5, n_prims - 5))
    n_tri = 0
    for kind in order:
        p = abi.Primitive(); p.kind = int(kind); p.material = int(kinds[rng.integers(0, len(kinds))])
        if kind == abi.PRIM_SPHERE:
            p.data[0:4] = [float(F(v)) for v in rng.uniform(-3, 3, 3)] + [float(F(rng.uniform(0.3, 1.2)))]
        elif kind == abi.PRIM_PLANE:
            n = L.normalized(L.v3(*rng.normal(size=3)))
            p.data[0:6] = [0.0, float(F(-4.0 - rng.uniform(0, 1))), 0.0] + [float(v) for v in (L.v3(0, 1, 0) if rng.random() < 0.5 else n)]
        elif kind == abi.PRIM_QUAD:
            p.data[0:15] = [float(v) for v in L.quad_from_matrix(matrix())]
        elif kind == abi.PRIM_CUBE:
            m = matrix(); p.data[0:16] = [float(v) for v in m]; p.data[16:32] = [float(v) for v in L.mat4_inverse(m)]
        else:
            m = matrix(); p.data[0:16] = [float(v) for v in m]; p.data[16:32] = [float(v) for v in L.mat4_inverse(m)]
            if identity_meshes:                                        # untransformed meshes (OBJ data in world space): what k_render_ctr_wf_nometal_ident is picked for
                ident = [1.0 if (k % 5) == 0 else 0.0 for k in range(16)]
                p.data[0:16] = ident; p.data[16:32] = ident
            nt = mesh_tris if mesh_tris > 0 else int(rng.integers(1, 4))          # mesh_tris <= 0: tiny meshes of 1..3 triangles
            v = rng.uniform(-1, 1, size=(nt, 3, 3)).astype(F)
            v[: nt // 4, :, 2] = F(0.25)                                   # a coplanar patch: zero-thickness leaf boxes (App. B-1)
            idx = np.arange(nt * 3).reshape(nt, 3)
            tris = L._triangles_from_indexed(v.reshape(-1, 3), idx)
            mesh = abi.Mesh(); mesh.first_triangle, mesh.triangle_count = n_tri, len(tris)
            n_tri += len(tris); tri_chunks.append(tris); meshes.append(mesh); p.mesh = len(meshes) - 1
        prims.append(p)

    sc = L.LoadedScene()
    sc.materials, sc.primitives, sc.meshes = mats, prims, meshes
    sc.triangles = np.concatenate(tri_chunks, axis=0) if tri_chunks else np.zeros((0, 12), F)
    sc.finalize()
    sc.c.miss_color[:] = [float(F(v)) for v in rng.uniform(0.2, 0.8, 3)]
    sc._keep = host.attach_bvh(sc)
    sc.camera = L.camera_new((0.0, 1.0, 9.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), F(50.0), F(4.0 / 3.0))
    return sc
