"""ctypes mirror of include/mi355rt.h (the C ABI).  Pure declarations: no library is loaded here.

Every Structure below must stay field-for-field identical to the header; tests/test_abi.py checks
the sizes against the values the C compiler reports (`mi355rt_host` exports them).
"""
import ctypes as C

ABI_VERSION = 4

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_IO, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6

MAT_LAMBERT_SOLID, MAT_LAMBERT_CHECKER, MAT_METAL, MAT_DIELECTRIC, MAT_EMISSIVE, MAT_PLASTIC, \
    MAT_ROUGH_GGX, MAT_ROUGH_BECKMANN, MAT_NULL, MAT_TEXTURE = range(10)
PRIM_SPHERE, PRIM_PLANE, PRIM_QUAD, PRIM_CUBE, PRIM_MESH = range(5)
RNG_CTR, RNG_REF = 0, 1
FLAG_FIXED_AABB = 1

f32, u32, u64 = C.c_float, C.c_uint32, C.c_uint64


class Camera(C.Structure):
    _fields_ = [("position", f32 * 3), ("forward", f32 * 3), ("right", f32 * 3), ("true_up", f32 * 3),
                ("half_width", f32), ("half_height", f32)]


class Settings(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("samples_per_pixel", u32), ("max_depth", u32)]


class Material(C.Structure):
    _fields_ = [("kind", u32), ("albedo", f32 * 3), ("aux", f32 * 3), ("p0", f32), ("p1", f32),
                ("eta", f32 * 3), ("k", f32 * 3), ("texture", u32)]


class Texture(C.Structure):
    _fields_ = [("rgba8", C.POINTER(C.c_uint8)), ("width", u32), ("height", u32)]


class Primitive(C.Structure):
    _fields_ = [("kind", u32), ("material", u32), ("mesh", u32), ("_pad", u32), ("data", f32 * 32)]


class Triangle(C.Structure):
    _fields_ = [("v0", f32 * 3), ("v1", f32 * 3), ("v2", f32 * 3), ("normal", f32 * 3)]


class BvhNode(C.Structure):
    _fields_ = [("bmin", f32 * 3), ("bmax", f32 * 3), ("left", u32), ("right", u32),
                ("first_index", u32), ("index_count", u32)]


class Mesh(C.Structure):
    _fields_ = [("first_triangle", u32), ("triangle_count", u32), ("first_node", u32), ("node_count", u32),
                ("first_index", u32), ("index_count", u32), ("max_depth", u32), ("_pad", u32)]


class Scene(C.Structure):
    _fields_ = [("primitives", C.POINTER(Primitive)), ("n_primitives", u32),
                ("materials", C.POINTER(Material)), ("n_materials", u32),
                ("meshes", C.POINTER(Mesh)), ("n_meshes", u32),
                ("triangles", C.POINTER(Triangle)), ("n_triangles", u32),
                ("nodes", C.POINTER(BvhNode)), ("n_nodes", u32),
                ("tri_indices", C.POINTER(u32)), ("n_tri_indices", u32),
                ("miss_color", f32 * 3), ("sky_width", u32), ("sky_height", u32),
                ("sky_rgb", C.POINTER(f32)), ("textures", C.POINTER(Texture)), ("n_textures", u32)]


class Options(C.Structure):
    _fields_ = [("abi_version", u32), ("rng_mode", u32), ("seed", u64),
                ("row_begin", u32), ("row_end", u32), ("strip_rows", u32), ("n_parts", u32), ("part", u32),
                ("flags", u32), ("workspace_bytes", u64)]

    @classmethod
    def make(cls, rng_mode=RNG_CTR, seed=0, row_begin=0, row_end=0, strip_rows=1, n_parts=1, part=0,
             workspace_bytes=0, flags=0):
        return cls(ABI_VERSION, rng_mode, seed, row_begin, row_end, strip_rows, n_parts, part, flags, workspace_bytes)


class Stats(C.Structure):
    _fields_ = [("render_kernel_ms", C.c_double), ("resolve_kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("samples", u64), ("rays", u64), ("rows_rendered", u32), ("bands", u32),
                ("grid_blocks", u32), ("block_threads", u32), ("kernel_vgprs", u32), ("kernel_sgprs", u32)]


class LoadOverrides(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("samples_per_pixel", u32), ("max_depth", u32),
                ("skip_unknown_primitives", u32), ("wo3_four_index_stride", u32)]


STRUCT_SIZES = {  # what sizeof() must report in C
    "mi355rt_camera": 56, "mi355rt_settings": 16, "mi355rt_material": 64, "mi355rt_primitive": 144,
    "mi355rt_triangle": 48, "mi355rt_bvh_node": 40, "mi355rt_mesh": 32,
}


def rows_selected(height, opt=None):
    """Rows an Options selects, in output order (mirror of the rule in mi355rt.h)."""
    if opt is None:
        return list(range(height))
    rb, re_ = opt.row_begin, (opt.row_end or height)
    strip, parts, part = max(opt.strip_rows, 1), max(opt.n_parts, 1), opt.part
    return [y for y in range(rb, re_) if (y // strip) % parts == part]
