"""The product's C++ loader (csrc/host) against the independent numpy restatement (oracle/scene_loader.py):
both follow src/tungsten/parser.rs and must produce bit-identical POD arrays."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import SCENES


def _bytes(ptr, n, T):
    if n == 0:
        return np.zeros((0,), np.uint8)
    return np.frombuffer((T * n).from_address(C.addressof(ptr.contents)), dtype=np.uint8).reshape(n, C.sizeof(T)).copy()


@pytest.mark.parametrize("name", sorted(SCENES))
def test_cpp_loader_equals_oracle_loader(name, native, abi):
    host, _ = native
    from oracle import scene_loader
    kw = dict(skip_unknown_primitives=(name == "teapot"))
    a = host.LoadedScene(SCENES[name], **kw)
    b = scene_loader.load_scene(SCENES[name], **kw)
    for f, _ in a.settings._fields_:
        assert getattr(a.settings, f) == getattr(b.settings, f)
    assert np.array_equal(np.frombuffer(a.camera, np.uint8), np.frombuffer(b.camera, np.uint8))
    sa, sb = a.c, b.c
    assert (sa.n_primitives, sa.n_materials, sa.n_meshes, sa.n_triangles) == (sb.n_primitives, sb.n_materials, sb.n_meshes, sb.n_triangles)
    assert np.array_equal(_bytes(sa.primitives, sa.n_primitives, abi.Primitive), _bytes(sb.primitives, sb.n_primitives, abi.Primitive))
    assert np.array_equal(_bytes(sa.materials, sa.n_materials, abi.Material), _bytes(sb.materials, sb.n_materials, abi.Material))
    assert np.array_equal(_bytes(sa.triangles, sa.n_triangles, abi.Triangle), _bytes(sb.triangles, sb.n_triangles, abi.Triangle))
    assert list(sa.miss_color) == [0.5, 0.5, 0.5]


def test_wo3_four_index_reader_is_opt_in_and_agrees_between_loaders(native, abi):
    """mesh_object.rs:190-192 steps 3 u32 per triangle through a file that stores 4.  The default
    reproduces that; wo3_four_index_stride=1 reads every triangle.  Both loaders must agree bit-for-bit in both modes."""
    host, _ = native
    from oracle import scene_loader
    kw = dict(skip_unknown_primitives=True)
    bug = host.LoadedScene(SCENES["teapot"], **kw)
    fix = host.LoadedScene(SCENES["teapot"], wo3_four_index_stride=True, **kw)
    ofix = scene_loader.load_scene(SCENES["teapot"], wo3_four_index_stride=True, **kw)
    assert bug.c.n_triangles == 19369 + 11968                           # SURVEY 8(d) cfg 3: what survives the 3-index stride
    assert fix.c.n_triangles == ofix.c.n_triangles > 2 * bug.c.n_triangles
    assert np.array_equal(_bytes(fix.c.triangles, fix.c.n_triangles, abi.Triangle), _bytes(ofix.c.triangles, ofix.c.n_triangles, abi.Triangle))
    import struct
    total = 0
    for m in ("Mesh000.wo3", "Mesh001.wo3"):
        b = open(os.path.join(os.path.dirname(SCENES["teapot"]), "models", m), "rb").read()
        nv = struct.unpack_from("<Q", b, 0)[0]
        nt = struct.unpack_from("<Q", b, 8 + nv * 32)[0]
        assert len(b) == 8 + nv * 32 + 8 + nt * 16                       # the file really holds 4 u32 per triangle
        total += nt
    assert total == 126048 and fix.c.n_triangles == 124840               # all of them, minus the degenerate ones (mesh_object.rs:128-134)


def test_scene_inventory_matches_survey(native, abi):
    host, _ = native
    c = host.LoadedScene(SCENES["cornell"])
    kinds = [c.c.primitives[i].kind for i in range(c.c.n_primitives)]
    assert kinds == [abi.PRIM_QUAD] * 5 + [abi.PRIM_CUBE] * 2 + [abi.PRIM_QUAD]       # JSON order = hit order (hittable.rs:50)
    light = c.c.materials[c.c.primitives[7].material]
    assert light.kind == abi.MAT_EMISSIVE and list(light.albedo) == [17.0, 12.0, 4.0]   # quad.emission wins over bsdf (parser.rs:707-711)
    assert (c.settings.width, c.settings.height, c.settings.samples_per_pixel, c.settings.max_depth) == (1024, 1024, 64, 64)
    v = host.LoadedScene(SCENES["veach"])
    r = [v.c.primitives[i].data[3] for i in range(v.c.n_primitives) if v.c.primitives[i].kind == abi.PRIM_SPHERE]
    assert r == pytest.approx([1.0, 0.5, 0.05])                                           # radius <- scale fallback (parser.rs:550-564)
    em = [v.c.materials[v.c.primitives[i].material].albedo[0] for i in range(v.c.n_primitives) if v.c.primitives[i].kind == abi.PRIM_SPHERE]
    assert em == pytest.approx([300 / (4 * np.pi ** 2 * rr * rr) for rr in (1.0, 0.5, 0.05)], rel=1e-6)   # parser.rs:567-575
    s = host.LoadedScene(SCENES["semesterbild"])
    assert s.c.n_triangles == 4748 and s.c.meshes[0].node_count == 3351 and s.c.meshes[0].max_depth == 11
    t = host.LoadedScene(SCENES["teapot"], skip_unknown_primitives=True)
    assert [t.c.meshes[i].triangle_count for i in range(2)] == [19369, 11968]            # WO3 3-index stride bug kept (App. B-2)
    assert t.c.materials[1].kind == abi.MAT_LAMBERT_CHECKER and t.c.materials[1].p0 == pytest.approx(1 / 20)


def test_unknown_primitive_is_a_hard_error_like_serde(native, abi):
    host, _ = native
    with pytest.raises(RuntimeError, match="unknown variant `infinite_sphere`"):
        host.LoadedScene(SCENES["teapot"])


def test_overrides_and_defaults(native, tmp_path, abi):
    host, _ = native
    a = host.LoadedScene(SCENES["cornell"], width=400, height=300, spp=16, max_depth=4)
    assert (a.settings.width, a.settings.height, a.settings.samples_per_pixel, a.settings.max_depth) == (400, 300, 16, 4)
    assert a.camera.half_width == pytest.approx(a.camera.half_height * 400 / 300, rel=1e-6)   # aspect follows W/H (parser.rs:294-297)
    p = tmp_path / "min.json"
    p.write_text(json.dumps({"camera": {"transform": {"position": [0, 0, 5], "look_at": {"x": 0, "y": 0, "z": 0}, "up": [0, 1, 0]}, "fov": 40},
                             "primitives": [{"type": "sphere", "transform": {}, "bsdf": "nope"},
                                            {"type": "plane", "point": [0, -1, 0], "normal": [0, 2, 0], "material": {"Metal": {"albedo": [0.8, 0.8, 0.8], "fuzz": 3.0}}}],
                             "bsdfs": [{"name": "c", "type": "conductor"}]}))
    m = host.LoadedScene(str(p))
    assert (m.settings.width, m.settings.height, m.settings.samples_per_pixel, m.settings.max_depth) == (800, 600, 16, 10)   # parser.rs:255-258
    assert m.c.n_materials == 2                         # "conductor" skipped; magenta fallback + inline metal
    assert m.c.materials[0].kind == abi.MAT_LAMBERT_SOLID and list(m.c.materials[0].albedo) == [1.0, 0.0, 1.0]
    assert m.c.materials[1].kind == abi.MAT_METAL and m.c.materials[1].p0 == 1.0            # fuzz clamped (material.rs:79-84)
    assert list(m.c.primitives[1].data[3:6]) == [0.0, 1.0, 0.0]                              # Plane::new normalises
    from oracle import scene_loader
    o = scene_loader.load_scene(str(p))
    assert np.array_equal(_bytes(m.c.primitives, 2, abi.Primitive), _bytes(o.c.primitives, 2, abi.Primitive))
    assert np.array_equal(_bytes(m.c.materials, 2, abi.Material), _bytes(o.c.materials, 2, abi.Material))


def test_png_writer_roundtrip(native, tmp_path):
    host, _ = native
    from PIL import Image
    rng = np.random.default_rng(1)
    img = rng.integers(0, 1 << 24, size=(7, 13), dtype=np.uint32)
    path = str(tmp_path / "x.png")
    host.write_png(path, img, 13, 7)
    got = np.array(Image.open(path).convert("RGB")).astype(np.uint32)
    assert np.array_equal((got[..., 0] << 16) | (got[..., 1] << 8) | got[..., 2], img)


def test_pfm_writer_roundtrip(native, tmp_path):
    """Portable FloatMap: 'PF', width height, negative scale = little-endian, rows bottom-up, raw f32 (bit-preserving)."""
    host, _ = native
    rng = np.random.default_rng(2)
    img = rng.standard_normal((5, 9, 3)).astype(np.float32)
    img[0, 0] = [np.inf, -0.0, np.float32(1e-42)]                  # special values and a denormal survive untouched
    path = str(tmp_path / "x.pfm")
    host.write_pfm(path, img, 9, 5)
    raw = open(path, "rb").read()
    header = b"PF\n9 5\n-1.0\n"
    assert raw.startswith(header) and len(raw) == len(header) + 5 * 9 * 3 * 4
    got = np.frombuffer(raw[len(header):], dtype="<f4").reshape(5, 9, 3)[::-1]
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))


def _read_exr(raw):
    """Minimal reader for what mi355rt_write_exr emits, written from the OpenEXR file-layout document (not from the writer):
    magic, version, attributes until an empty name, offset table, then one chunk per scan line."""
    import struct
    assert raw[:4] == bytes([0x76, 0x2F, 0x31, 0x01]) and struct.unpack_from("<I", raw, 4)[0] == 2
    p, attrs = 8, {}
    while raw[p] != 0:
        e = raw.index(b"\0", p); name = raw[p:e].decode(); p = e + 1
        e = raw.index(b"\0", p); typ = raw[p:e].decode(); p = e + 1
        size = struct.unpack_from("<i", raw, p)[0]; p += 4
        attrs[name] = (typ, raw[p:p + size]); p += size
    p += 1
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    assert attrs["displayWindow"][1] == attrs["dataWindow"][1] and (x0, y0) == (0, 0)
    W, H = x1 + 1, y1 + 1
    ch, q, names = attrs["channels"][1], 0, []
    while ch[q] != 0:
        e = ch.index(b"\0", q); names.append(ch[q:e].decode()); q = e + 1
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", ch, q); q += 16
        assert ptype == 2 and xs == 1 and ys == 1                    # FLOAT, no subsampling
    assert names == sorted(names) == ["B", "G", "R"]
    offsets = struct.unpack_from(f"<{H}Q", raw, p)
    img = np.zeros((H, W, 3), np.float32)
    for y in range(H):
        yy, nbytes = struct.unpack_from("<ii", raw, offsets[y])
        assert yy == y and nbytes == 12 * W
        rows = np.frombuffer(raw, "<f4", 3 * W, offsets[y] + 8).reshape(3, W)
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = rows[0], rows[1], rows[2]
    assert offsets[-1] + 8 + 12 * W == len(raw)
    return img


def test_exr_writer_roundtrip(native, tmp_path):
    """OpenEXR scan-line file, FLOAT B/G/R, uncompressed: parsed back by an independent reader, bit-preserving."""
    host, _ = native
    rng = np.random.default_rng(5)
    img = rng.standard_normal((7, 13, 3)).astype(np.float32)
    img[0, 0] = [np.inf, -0.0, np.float32(1e-42)]
    path = str(tmp_path / "x.exr")
    host.write_exr(path, img, 13, 7)
    got = _read_exr(open(path, "rb").read())
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))


def _mini_scene(tmp_path, camera_extra=None, prims=None, name="s.json"):
    cam = {"transform": {"position": [0, 0, 5], "look_at": {"x": 0, "y": 0, "z": 0}, "up": [0, 1, 0]}, "fov": 40}
    cam.update(camera_extra or {})
    p = tmp_path / name
    p.write_text(json.dumps({"camera": cam, "primitives": prims or [], "bsdfs": [{"name": "m", "type": "lambert", "albedo": 0.5}]}))
    return str(p)


@pytest.mark.parametrize("res,ok", [([320, 200], True), (64, True), ([320], False), ([320, 200, 3], False), ([], False), ("320x200", False)])
def test_resolution_is_a_number_or_exactly_two_numbers(res, ok, native, tmp_path):
    """ResolutionConfig is `Square(usize) | Explicit([usize; 2])`, untagged (/root/reference/src/tungsten/parser.rs:66-72): serde takes
    an array of exactly two -- any other length matches no variant and the whole scene fails to load."""
    host, _ = native
    from oracle import scene_loader
    path = _mini_scene(tmp_path, {"resolution": res})
    if ok:
        want = (res, res) if isinstance(res, int) else tuple(res)
        a, b = host.LoadedScene(path), scene_loader.load_scene(path)
        assert (a.settings.width, a.settings.height) == want == (b.settings.width, b.settings.height)
    else:
        with pytest.raises(RuntimeError, match="did not match any variant of untagged enum ResolutionConfig"):
            host.LoadedScene(path)
        with pytest.raises(ValueError, match="ResolutionConfig"):
            scene_loader.load_scene(path)


def test_obj_points_and_lines_are_dropped_like_tobj_does(native, abi, tmp_path):
    """tobj 4.0.3 GPU_LOAD_OPTIONS has ignore_points and ignore_lines set (/root/reference/src/mesh/mesh_object.rs:64): an `f` with
    one or two corners never reaches mesh.indices.  The mesh keeps its triangles (it does not vanish with "Invalid index data
    length"); a short face whose corner is out of range still fails the mesh, as tobj's parser does."""
    host, _ = native
    from oracle import scene_loader
    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2\nf 1 2 3\nf 4\nf 2 4 3\nf 3 1\n")
    prims = [{"type": "mesh", "file": "m.obj", "bsdf": "m", "transform": {}}]
    path = _mini_scene(tmp_path, prims=prims)
    a, b = host.LoadedScene(path), scene_loader.load_scene(path)
    assert a.c.n_meshes == 1 and a.c.n_triangles == 2 == b.c.n_triangles
    assert np.array_equal(_bytes(a.c.triangles, 2, abi.Triangle), _bytes(b.c.triangles, 2, abi.Triangle))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nf 1 9\n")
    bad = _mini_scene(tmp_path, prims=[{"type": "mesh", "file": "bad.obj", "bsdf": "m", "transform": {}}], name="bad.json")
    c = host.LoadedScene(bad)                                   # the mesh fails to load and is skipped with a warning (parser.rs:690-700); the scene loads
    assert c.c.n_meshes == 0 and c.c.n_primitives == 0
