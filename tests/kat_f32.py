"""Independent known answers for the per-function arithmetic of the hot path, written in numpy float32 straight from the
Rust sources (file:line cited per function) -- NOT from oracle/rt_oracle.cpp and NOT from the HIP kernels.  Every
operation is a numpy float32 scalar operation (correctly rounded IEEE f32 for + - * / sqrt), in the reference's
operation order, so results must equal the oracle's / the GPU's bit for bit wherever only those operations occur.
Used by tests/test_kat_functions.py (oracle, CPU) and its gpu twin.
"""
import numpy as np

f32 = np.float32
EPS = f32(1e-4)           # renderer.rs:17
PI = f32(np.pi)           # std::f32::consts::PI


# ---- vec3.rs ----------------------------------------------------------------------------------------------------
def V(x, y, z):
    return np.array([x, y, z], dtype=np.float32)


def dot(a, b):                                   # vec3.rs:17-19
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def cross(a, b):                                 # vec3.rs:21-27
    return V(f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0]))


def length(a):                                   # vec3.rs:29-35
    return f32(np.sqrt(dot(a, a)))


def normalized(a):                               # vec3.rs:37-44
    ln = length(a)
    if ln < EPS:
        return a.copy()
    return a * f32(f32(1.0) / ln)


def ray_new(o, d):                               # ray.rs:12-17: Ray::new normalises again
    return o, normalized(d)


def near_zero(a):                                # vec3.rs:63-66
    s = f32(1e-8)
    return abs(a[0]) < s and abs(a[1]) < s and abs(a[2]) < s


def to_world(local, n):                          # vec3.rs:72-81
    up = V(0, 0, 1) if abs(n[2]) < f32(0.999) else V(0, 1, 0)
    t = normalized(cross(n, up))
    b = cross(n, t)
    return (t * local[0] + b * local[1]) + n * local[2]


# ---- rand 0.9.1 float conversions + Philox4x32-10 (Random123) --------------------------------------
def u01(w):
    return f32(f32(w >> 8) * f32(1.0 / 16777216.0))


def range11(w):
    v12 = np.array([(w >> 9) | 0x3F800000], dtype=np.uint32).view(np.float32)[0]
    return f32(f32(f32(v12 - f32(1.0)) * f32(2.0)) + f32(-1.0))


def philox(k0, k1, c0, c1, c2, c3):
    M0, M1, W0, W1, M32 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + W0) & M32, (k1 + W1) & M32
    return [c0, c1, c2, c3]


def pcg4d(x, y, z, w):
    """Jarzynski & Olano, "Hash Functions for GPU Rendering" (JCGT 9(3), 2020), listing of pcg4d, on Python integers."""
    M32 = 0xFFFFFFFF
    x, y, z, w = [(v * 1664525 + 1013904223) & M32 for v in (x, y, z, w)]
    x = (x + y * w) & M32; y = (y + z * x) & M32; z = (z + x * y) & M32; w = (w + y * z) & M32
    x, y, z, w = [v ^ (v >> 16) for v in (x, y, z, w)]
    x = (x + y * w) & M32; y = (y + z * x) & M32; z = (z + x * y) & M32; w = (w + y * z) & M32
    return [x, y, z, w]


def ctr_block(k0, k1, x, s, ray, j):
    """Counter-mode generator of the product since round 5 (rt_rng.h, CTR_GEN 2): a per-path base pcg4d(x, sample, key lo, key hi); block j of
    the event after ray `ray` = pcg4d(base.x, base.y, base.z + ray, base.w + j)."""
    b = pcg4d(x, s, k0, k1)
    return pcg4d(b[0], b[1], (b[2] + ray) & 0xFFFFFFFF, (b[3] + j) & 0xFFFFFFFF)


class CtrDraws:
    """Draw addressing of the counter mode: the scatter event of `ray` reads random::<f32>() number k from
    block 0 word k, and rejection try j of random_in_unit_sphere from block j words 1..3."""

    def __init__(self, k0, k1, x, s, ray):
        self.key = (k0, k1, x, s, ray)

    def block(self, j):
        k0, k1, x, s, ray = self.key
        return ctr_block(k0, k1, x, s, ray, j)

    def uniform(self, k):
        return u01(self.block(0)[k])

    def unit_ball(self):                          # vec3.rs:46-61
        j = 0
        while True:
            w = self.block(j)
            p = V(range11(w[1]), range11(w[2]), range11(w[3]))
            if dot(p, p) < f32(1.0):
                return p
            j += 1


# ---- material.rs / tungsten/materials.rs -----------------------------------------------------------------------------
def reflect(v, n):                               # material.rs:194-206
    if np.isnan(v).any() or np.isnan(n).any() or not n.any():
        return V(np.nan, np.nan, np.nan)
    return v - (n * f32(2.0)) * dot(v, n)


def powi5(x):                                    # llvm.powi.f32(x, 5): x * ((x*x) * (x*x))
    x2 = f32(x * x)
    return f32(x * f32(x2 * x2))


def schlick(cosine, ref_idx):                    # material.rs:221-227 == tungsten/materials.rs:23-27
    r0 = f32(f32(f32(1.0) - ref_idx) / f32(f32(1.0) + ref_idx))
    r0 = f32(r0 * r0)
    return f32(r0 + f32(f32(f32(1.0) - r0) * powi5(f32(f32(1.0) - cosine))))


def lambert_dir(n, p_hit, ball):                 # material.rs:54-62
    d = n + normalized(ball)
    if near_zero(d):
        d = n
    return ray_new(p_hit + n * EPS, normalized(d))


def scatter_lambert(albedo, rd, p, n, draws):    # material.rs:47-71
    o, d = lambert_dir(n, p, draws.unit_ball())
    return True, o, d, albedo


def checker_value(on, off, inv_scale, p):        # tungsten/materials.rs:89-99 (Rust `%` keeps the sign)
    def cell(v):
        return int(np.floor(f32(v * inv_scale)))
    s = cell(p[0]) + cell(p[1]) + cell(p[2])
    rem = abs(s) % 2 * (1 if s >= 0 else -1)
    return on if rem == 0 else off


def scatter_metal(albedo, fuzz, rd, p, n, draws):     # material.rs:87-110
    refl = reflect(normalized(rd), n)
    fz = refl + draws.unit_ball() * fuzz if fuzz > 0 else refl
    if not (dot(fz, n) > 0):
        return False, None, None, None
    o, d = ray_new(p + n * EPS, normalized(fz))
    return True, o, d, albedo


def scatter_dielectric(ior, front, rd, p, n, draws):  # material.rs:122-162
    ratio = f32(f32(1.0) / ior) if front else f32(ior / f32(1.0))
    unit = normalized(rd)
    cos_t = min(dot(-unit, n), f32(1.0))
    sin2 = f32(f32(1.0) - f32(cos_t * cos_t))
    cannot = f32(f32(ratio * ratio) * sin2) > f32(1.0)
    refl = schlick(cos_t, f32(f32(1.0) / ratio))
    if cannot or refl > draws.uniform(0):
        d = reflect(unit, n)
    else:                                          # refract(), material.rs:208-219
        ct = min(dot(-unit, n), f32(1.0))
        perp = (unit + n * ct) * ratio
        par2 = f32(f32(1.0) - dot(perp, perp))
        d = reflect(unit, n) if par2 < 0 else perp + n * f32(-np.sqrt(par2))
    o = p + n * EPS if dot(d, n) > 0 else p - n * EPS
    o, d = ray_new(o, normalized(d))
    return True, o, d, V(1, 1, 1)


def scatter_plastic(albedo, ior, rd, p, n, draws):    # tungsten/materials.rs:29-65
    dn = dot(rd, n)
    cosine = f32(f32(ior * dn) / length(rd)) if dn > 0 else f32(f32(-dn) / length(rd))
    prob = schlick(cosine, ior)
    if draws.uniform(0) < prob:
        r = normalized(rd - (n * f32(2.0)) * dot(rd, n))           # Vec3::reflect, vec3.rs:68-70
        o, d = ray_new(p + n * EPS, r)
        return True, o, d, V(0.9, 0.9, 0.9)
    o, d = lambert_dir(n, p, draws.unit_ball())
    return True, o, d, albedo


def fresnel_conductor(cos_theta, eta, k):        # tungsten/materials.rs:184-202
    c = min(max(cos_theta, f32(0.0)), f32(1.0))
    cos2 = V(*[f32(c * c)] * 3)
    sin2 = V(1, 1, 1) - cos2
    eta2, k2 = eta * eta, k * k
    t0 = eta2 - k2 - sin2
    a2b2 = np.sqrt(t0 * t0 + V(4, 4, 4) * eta2 * k2)
    t1 = a2b2 + cos2
    a = np.sqrt((a2b2 + t0) * V(0.5, 0.5, 0.5))
    t2 = V(*[f32(f32(2.0) * c)] * 3) * a
    rs = (t1 - t2) / (t1 + t2)
    t3 = cos2 * a2b2 + sin2 * sin2
    rp = rs * ((t3 - t2) / (t3 + t2))
    return (rs + rp) * V(0.5, 0.5, 0.5)


def ggx_g1(ndx, rough):                          # tungsten/materials.rs:205-216
    if ndx <= 0:
        return f32(0.0)
    a = f32(rough * rough)
    k = f32(a / f32(2.0))
    den = f32(f32(ndx * f32(f32(1.0) - k)) + k)
    return f32(1.0) if den < EPS else f32(ndx / den)


def beckmann_g(a, ndv, ndl):                     # tungsten/materials.rs:223-234
    def lam(x):
        t = f32(f32(1.0) / f32(a * x))
        if t < f32(1.6):
            num = f32(f32(f32(1.0) - f32(f32(1.259) * t)) + f32(f32(f32(0.396) * t) * t))
            den = f32(f32(f32(3.535) * t) + f32(f32(f32(2.181) * t) * t))
            return f32(num / den)
        return f32(0.0)
    return f32(f32(1.0) / f32(f32(f32(1.0) + lam(ndv)) + lam(ndl)))


def scatter_rough(albedo, rough, eta, k, ggx, rd, p, n, draws):   # tungsten/materials.rs:236-290, 306-377
    """Transcendentals (ln, atan, sin, cos) are evaluated in float64 and rounded once: callers compare with a few-ulp tolerance."""
    v = -normalized(rd)
    u1 = max(draws.uniform(0), f32(1e-6))
    u2 = draws.uniform(1)
    ln_u1 = f32(np.log(np.float64(u1)))
    if ggx:
        a = f32(rough * rough)
        arg = f32(f32(f32(a * a) * f32(-ln_u1)) / f32(f32(1.0) - u1))
    else:
        arg = f32(-f32(f32(rough * rough) * ln_u1))
    if np.isnan(arg) or np.isinf(arg) or arg < 0:
        h = to_world(V(0, 0, 1), n)
    else:
        theta = f32(np.arctan(np.float64(f32(np.sqrt(arg)))))
        phi = f32(f32(f32(2.0) * PI) * u2)
        st, ct = f32(np.sin(np.float64(theta))), f32(np.cos(np.float64(theta)))
        h = to_world(V(f32(st * f32(np.cos(np.float64(phi)))), f32(st * f32(np.sin(np.float64(phi)))), ct), n)
    l = reflect(-v, h)
    if dot(l, n) <= 0:
        return False, None, None, None
    ndl, ndv = max(dot(n, l), f32(0)), max(dot(n, v), f32(0))
    ndh, vdh = max(dot(n, h), f32(0)), max(dot(v, h), f32(0))
    g = f32(ggx_g1(ndv, rough) * ggx_g1(ndl, rough)) if ggx else beckmann_g(rough, ndv, ndl)
    f = fresnel_conductor(vdh, eta, k)
    num = (f * g) * vdh
    den = f32(f32(ndv * ndh) + EPS)
    col = albedo * (num / den) if den > EPS else V(0, 0, 0)
    o, d = ray_new(p + n * EPS, normalized(l))
    return True, o, d, col


# ---- objects/sphere.rs:15-53, tungsten/objects/quad.rs:83-132, hittable.rs:19-26 -------------------------------------
def set_face(rd, outward):
    front = dot(rd, outward) < 0
    return (outward if front else -outward), front


def hit_sphere(center, radius, ro, rd, t_min, t_max):
    oc = ro - center
    a = dot(rd, rd)
    half_b = dot(oc, rd)
    c = f32(dot(oc, oc) - f32(radius * radius))
    disc = f32(f32(half_b * half_b) - f32(a * c))
    if disc < 0:
        return None
    sq = f32(np.sqrt(disc))
    root = f32(f32(-half_b - sq) / a)
    if root <= t_min or root >= t_max:
        root = f32(f32(-half_b + sq) / a)
        if root <= t_min or root >= t_max:
            return None
    p = ro + rd * root
    n, front = set_face(rd, (p - center) / radius)
    return root, p, n, front


def quad_from_corners(base, e0, e1):
    """Quad::new's derived fields (quad.rs:60-79) from the world-space base / edges."""
    n = normalized(cross(e0, e1))
    return dict(base=base, e0=e0, e1=e1, n=n, d=dot(n, base), inv0=f32(f32(1.0) / dot(e0, e0)), inv1=f32(f32(1.0) / dot(e1, e1)))


def hit_quad(q, ro, rd, t_min, t_max):
    denom = dot(q["n"], rd)
    if abs(denom) < EPS:
        return None
    t = f32(f32(q["d"] - dot(q["n"], ro)) / denom)
    if t <= t_min or t >= t_max:
        return None
    p = ro + rd * t
    v = p - q["base"]
    l0, l1 = f32(dot(v, q["e0"]) * q["inv0"]), f32(dot(v, q["e1"]) * q["inv1"])
    lo, hi = f32(-EPS), f32(f32(1.0) + EPS)
    if l0 < lo or l0 > hi or l1 < lo or l1 > hi:
        return None
    n, front = set_face(rd, q["n"])
    return t, p, n, front


def texture_value(rgba, h_offset, n):                    # tungsten/parser.rs:222-241 (rgba: uint8 [H, W, 4])
    """acos / atan2 are evaluated in float64 and rounded once; a normal that lands within an ulp of a texel border may
    pick the neighbour on another libm -- the test cases keep clear of borders."""
    H, W = rgba.shape[:2]
    theta = f32(np.arccos(np.float64(n[1])))
    phi = f32(f32(np.arctan2(np.float64(n[2]), np.float64(n[0]))) + PI)
    u = f32(phi / f32(f32(2.0) * PI))
    v = f32(theta / PI)
    u = f32(np.fmod(f32(u + h_offset), f32(1.0)))
    x = int(f32(max(u, f32(0.0)) * f32(W - 1)))
    y = int(f32(max(v, f32(0.0)) * f32(H - 1)))
    px = rgba[min(y, H - 1), min(x, W - 1)]
    return V(f32(px[0]) / f32(255.0), f32(px[1]) / f32(255.0), f32(px[2]) / f32(255.0)), (f32(max(u, f32(0.0)) * f32(W - 1)), f32(max(v, f32(0.0)) * f32(H - 1)))


def scatter_texture(albedo, rgba, h_offset, rd, p, n, draws):     # tungsten/parser.rs:205-243
    o, d = lambert_dir(n, p, draws.unit_ball())
    tex, _ = texture_value(rgba, h_offset, n)
    return True, o, d, albedo * tex
