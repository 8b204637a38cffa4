// rt_intersect.h -- hittable.rs / objects/*.rs / mesh + BVH: probe-form hit tests, the stackless two-link walk, finish_hit()
// Part of the device code of libmi355rt.so; included by rt_kernels.hip only (one translation unit: every kernel sees the same
// inlined device functions, and build.kernel_hash() covers every file of this directory).
#pragma once
#include "rt_math.h"

namespace mi355rt {

// ---------------------------------------------------------------------------------------------------
// Intersection.  The list walk (hittable.rs:45-58) only needs to know WHICH primitive is closest so far and at what t; the
// HitRecord (hittable.rs:10-27) of all but the last winner is never looked at.  So the walk carries a 4-register candidate
// (Cand) instead of the 8-register record, and finish_hit() builds the record of the winner once per ray, with exactly the
// arithmetic the reference's hit() performs for it (same inputs, same operations, same order -> same bits).  Measured
// reason: the compiler keeps a loop-carried record in two register sets and copies it at every nesting level of every
// primitive test -- 24-33 v_mov per quad, a third of its instructions; the copies scale with the size of the state.
// ---------------------------------------------------------------------------------------------------
// `q0` = the first 16 bytes of the hit material (kind, albedo), read from the copy in the primitive record (DevPrim.mat0) together
// with the record itself -- the lockstep kernels of mesh-free lists only (finish_hit's CARRY_Q0): there the material read no longer
// waits for the record read (cornell 15.85 -> 15.62 ms).  The wavefront kernels have no registers to carry it (with the BVH walk:
// teapot +3.5 %; mesh-free at 64 VGPRs: 18 -> 33 spilled registers, veach-mis -0.2 % of time for 2.2x the HBM bytes) and read
// the material record after the hit record as before (profiles/r03_ab_material_head_in_primitive.txt).
struct Hit { float t; f3 p; f3 n; uint32_t mat_ff; float4 q0; };            // the finished record; `mat_ff` = material | front_face << 31
constexpr uint32_t CAND_NONE = 0xFFFFFFFFu;
struct Cand {
    float t;            // closest hit distance so far (world), +inf while idx == CAND_NONE
    uint32_t idx;       // list index of the primitive that owns it
    float aux;          // cube: the object-space slab distance t_hit (cube.rs:98); mesh: the walk's object-space best_t
    uint32_t aux2;      // mesh: the winning triangle (index into the leaf-ordered array)
};
DI void cand_reset(Cand& c) { c.t = __builtin_inff(); c.idx = CAND_NONE; c.aux = 0.f; c.aux2 = 0u; }
// The lockstep kernels of mesh-free scenes carry the cube's object-space hit point as well (3 more registers that only the cube
// loop touches): finish_cube() then needs neither the object-space ray nor `aux` again (-42 instructions per shaded cube hit).
struct CandP : Cand { f3 po; };
DI void cand_reset(CandP& c) { cand_reset(static_cast<Cand&>(c)); c.po = mk(0.f, 0.f, 0.f); }

DI void set_face(Hit& h, f3 rd, f3 outward, uint32_t material) {                 // hittable.rs:19-26
    bool front = dot(rd, outward) < 0.0f;
    h.n = front ? outward : -outward;
    h.mat_ff = material | (front ? 0x80000000u : 0u);
}

// objects/sphere.rs:15-53 (rejections folded into one predicate; sqrt of a negative discriminant is discarded)
// Every test below PROBES: it reads the candidate only as t_max and hands back fresh values (accepted?, t, aux); the one
// place that changes the loop-carried candidate is cand_take()'s selects.  (A test that assigned the candidate inside its own
// branches made the compiler carry two copies of it through the structurised switch: ~10 v_mov per quad, ~25 per cube.)
struct Probe { float t, aux; f3 po; };
typedef float rec16_t __attribute__((ext_vector_type(16)));
DI bool cand_take(Cand& c, bool acc, uint32_t i, float t) { c.t = acc ? t : c.t; c.idx = acc ? i : c.idx; return acc; }
DI bool cand_take(Cand& c, bool acc, uint32_t i, const Probe& o) { c.aux = acc ? o.aux : c.aux; return cand_take(c, acc, i, o.t); }
DI bool cand_take(CandP& c, bool acc, uint32_t i, const Probe& o) {
    c.po.x = acc ? o.po.x : c.po.x; c.po.y = acc ? o.po.y : c.po.y; c.po.z = acc ? o.po.z : c.po.z;
    return cand_take(static_cast<Cand&>(c), acc, i, o.t);
}
DI bool hit_sphere(cprim_t pr, uint32_t i, f3 ro, f3 rd, float t_min, Cand& c) {
    f3 center = mk(pr->d[0], pr->d[1], pr->d[2]); float radius = pr->d[3];
    f3 oc = ro - center;
    float a = dot(rd, rd);
    float half_b = dot(oc, rd);
    float cc = dot(oc, oc) - radius * radius;
    float disc = half_b * half_b - a * cc;
    bool acc = false; float t = 0.f;
    if (!(disc < 0.0f)) {                                   // a wave whose lanes all miss the sphere skips the sqrt and the two divisions
        float sqrtd = sqrtf(disc);
        float r0 = (-half_b - sqrtd) / a, r1 = (-half_b + sqrtd) / a;
        const float t_max = c.t;
        const bool ok0 = !(r0 <= t_min || r0 >= t_max), ok1 = !(r1 <= t_min || r1 >= t_max);
        acc = ok0 || ok1; t = ok0 ? r0 : r1;
    }
    return cand_take(c, acc, i, t);
}

// objects/plane.rs:26-56
DI bool hit_plane(cprim_t pr, uint32_t i, f3 ro, f3 rd, float t_min, Cand& c) {
    f3 p1 = mk(pr->d[0], pr->d[1], pr->d[2]), n = mk(pr->d[3], pr->d[4], pr->d[5]);
    float denom = dot(n, rd);
    float t = dot(n, p1 - ro) / denom;
    return cand_take(c, !(fabsf(denom) < EPS) && !(t <= t_min || t >= c.t), i, t);
}

// tungsten/objects/quad.rs:83-132.  The two cheap rejections (parallel ray, t out of range) are folded into one
// predicate so the wave takes a single branch into the parallelogram test; the arithmetic is unchanged (the
// division also runs for |denom| < EPS lanes, whose result is discarded).
// FASTD: t by div_bounded() (rt_math.h).  t is used only where |denom| >= EPS, and |denom| <= |n||d| ~ 1 (the host refuses quads whose normal
// is not of unit scale: rt_api.cpp build_device_scene), so the divisor is in range; a
// numerator below 2^-100 gives a |t| below 2^-86 either way (rejected: t <= t_min), one of 2^100 or more sends the whole wave to the
// compiler's division (ballot); infinities and NaN come out of v_div_fixup_f32 as they do there.
template <bool FASTD = false>
DI bool hit_quad(cprim_t pr, uint32_t i, f3 ro, f3 rd, float t_min, Cand& c) {
    const rec16_t q = *reinterpret_cast<const __attribute__((address_space(4))) rec16_t*>(pr->d);     // the whole record: ONE scalar load
    f3 n = mk(q[0], q[1], q[2]);
    float denom = dot(n, rd);
    const float num = q[3] - dot(n, ro);
    float t;
    if (FASTD && __builtin_expect(__ballot(fabsf(num) >= 0x1p100f) == 0ull, 1)) t = div_bounded(num, denom); else
    t = num / denom;
    // Branch-free, and since round 5 also in the ISA: the predicate is built with `&` / `|` on the comparison results, not `&&` / `||`.  With the
    // short-circuit operators the compiler fenced the parallelogram test with two exec-mask regions (s_and_saveexec, s_cbranch_execz, s_or exec,
    // mask merges: ~12 scalar instructions per quad) that a wave of incoherent rays never skips, and read the record in four pieces with a wait in
    // front of each region.  The kernels' time follows the TOTAL instruction count, scalar ones included (profiles/r05/ab_lockstep_ray_stock.txt),
    // so: one s_load_dwordx16 for the record (its layout puts what every ray needs first, rt_device.h), one wait, straight-line arithmetic,
    // two selects on the candidate.  cornell 13.03 -> 12.61 ms of kernel (profiles/r05/ab_scalar_diet.txt).
    const bool candidate = !(fabsf(denom) < EPS) & !((t <= t_min) | (t >= c.t));
    f3 hit_pos = ro + rd * t;
    f3 v = hit_pos - mk(q[4], q[5], q[6]);
    float l0 = dot(v, mk(q[7], q[8], q[9])) * q[13];
    float l1 = dot(v, mk(q[10], q[11], q[12])) * q[14];
    const float lo = -EPS, hi = 1.0f + EPS;
    // lo <= l <= hi as ONE comparison: the median of (l, lo, hi) is l exactly when l lies between them; for a NaN l v_med3_f32 returns the minimum of the
    // other two, which no NaN equals -- false, like the two comparisons it replaces (-0.25 %).
    // (Measured and not kept, profiles/r05/ab_scalar_diet.txt: ONE test of the ray's origin per walk against a host-derived bound instead of the 2^100 test of
    // every quad's numerator, with a copy of the run loop without it: +1.5 % -- at this point the loop's layout weighs more than six instructions.)
    return cand_take(c, candidate & ((__builtin_amdgcn_fmed3f(l0, lo, hi) == l0) & (__builtin_amdgcn_fmed3f(l1, lo, hi) == l1)), i, t);
}

// glam Mat4 * Vec4 pieces on the DevPrim cube/mesh record (see rt_device.h for the layout).  PrimPtr is the wave-uniform
// constant-address-space pointer of the list walk (scalar loads) or a per-lane global pointer in finish_hit().
template <class PrimPtr> DI f3 xform_w2o_point(PrimPtr pr, f3 p) {            // (w2o * (p, 1)).xyz
    const auto* m = pr->d;
    return mk(((m[0] * p.x + m[3] * p.y) + m[6] * p.z) + m[9], ((m[1] * p.x + m[4] * p.y) + m[7] * p.z) + m[10],
              ((m[2] * p.x + m[5] * p.y) + m[8] * p.z) + m[11]);
}
template <class PrimPtr> DI f3 xform_w2o_dir(PrimPtr pr, f3 v) {              // (w2o * (v, 0)).xyz ; zd = w_axis * 0.0f keeps -0.0 behaviour
    const auto* m = pr->d;
    return mk(((m[0] * v.x + m[3] * v.y) + m[6] * v.z) + m[12], ((m[1] * v.x + m[4] * v.y) + m[7] * v.z) + m[13],
              ((m[2] * v.x + m[5] * v.y) + m[8] * v.z) + m[14]);
}
template <class PrimPtr> DI f3 xform_o2w_point(PrimPtr pr, f3 p) {            // (o2w * (p, 1)).xyz
    const auto* m = pr->d + 16;
    return mk(((m[0] * p.x + m[3] * p.y) + m[6] * p.z) + m[9], ((m[1] * p.x + m[4] * p.y) + m[7] * p.z) + m[10],
              ((m[2] * p.x + m[5] * p.y) + m[8] * p.z) + m[11]);
}
template <class PrimPtr> DI f3 xform_normal(PrimPtr pr, f3 n) {               // (w2o.transpose() * (n, 0)).xyz
    const auto* m = pr->d;
    return mk(((m[0] * n.x + m[1] * n.y) + m[2] * n.z) + m[31], ((m[3] * n.x + m[4] * n.y) + m[5] * n.z) + m[32],
              ((m[6] * n.x + m[7] * n.y) + m[8] * n.z) + m[33]);
}
DI float glam_signum(float v) { if (v != v) return v; return copysignf(1.0f, v); }

// Untransformed meshes (world_to_object == identity; the host checks it: rt_api.cpp xform_is_identity).  glam's Mat4 * Vec4 computes, per component,
// ((1*x + (+-0)*y) + (+-0)*z) + (+-0): for finite y, z the middle terms are zeros, and x + (+-0) == x for every x != 0 -- the object-space ray IS the
// world-space ray, bit for bit, unless a component is a zero (whose SIGN the sum could change) or not finite (0 * inf = NaN).  The kernels instantiated
// for such scenes (MESH_IDENT) therefore skip the two matrix products of mesh_setup for waves whose rays all pass this test, and take the general
// form otherwise -- a wave-uniform choice, once per pass.
DI bool ray_nonzero_finite(f3 ro, f3 rd) {
    const float mn = fminf(fminf(fminf(fabsf(ro.x), fabsf(ro.y)), fabsf(ro.z)), fminf(fminf(fabsf(rd.x), fabsf(rd.y)), fabsf(rd.z)));
    const float mx = fmaxf(fmaxf(fmaxf(fabsf(ro.x), fabsf(ro.y)), fabsf(ro.z)), fmaxf(fmaxf(fabsf(rd.x), fabsf(rd.y)), fabsf(rd.z)));
    return (mn > 0.0f) && (mx < __builtin_inff()) && !has_nan(ro) && !has_nan(rd);              // (fminf / fmaxf ignore a NaN operand: tested apart)
}

// objects/cube.rs:59-158, the part that decides whether and where the cube is hit; the face normal (cube.rs:105-143) is
// computed by finish_hit() for the winner only.
DI uint32_t cube_axis(f3 po) {                                                          // cube.rs:112-133 as selects
    const float ax = fabsf(po.x), ay = fabsf(po.y), az = fabsf(po.z);
    const float tol = 1e-4f;
    return (fabsf(ax - 0.5f) < tol) ? 0u : (fabsf(ay - 0.5f) < tol) ? 1u : (fabsf(az - 0.5f) < tol) ? 2u
         : (ax > ay && ax > az) ? 0u : (ay > az) ? 1u : 2u;
}
template <bool FASTR = false, class C>
DI bool hit_cube(cprim_t pr, uint32_t i, f3 ro_w, f3 rd_w, float t_min, C& c) {
    const rec16_t m = *reinterpret_cast<const __attribute__((address_space(4))) rec16_t*>(pr->d);     // w2o (3 x 4) and zd: ONE scalar load
    f3 ro = mk(((m[0] * ro_w.x + m[3] * ro_w.y) + m[6] * ro_w.z) + m[9], ((m[1] * ro_w.x + m[4] * ro_w.y) + m[7] * ro_w.z) + m[10],
               ((m[2] * ro_w.x + m[5] * ro_w.y) + m[8] * ro_w.z) + m[11]);                             // xform_w2o_point
    f3 rd = mk(((m[0] * rd_w.x + m[3] * rd_w.y) + m[6] * rd_w.z) + m[12], ((m[1] * rd_w.x + m[4] * rd_w.y) + m[7] * rd_w.z) + m[13],
               ((m[2] * rd_w.x + m[5] * rd_w.y) + m[8] * rd_w.z) + m[14]);                             // xform_w2o_dir
    float ix, iy, iz; recip3<FASTR>(rd.x, rd.y, rd.z, ix, iy, iz);                  // (FASTR: the kernels of mesh-free lists, like normalized<FASTN>)
    float t1x = (-0.5f - ro.x) * ix, t2x = (0.5f - ro.x) * ix;
    float t1y = (-0.5f - ro.y) * iy, t2y = (0.5f - ro.y) * iy;
    float t1z = (-0.5f - ro.z) * iz, t2z = (0.5f - ro.z) * iz;
    float t_enter = fmaxf(fminf(t1x, t2x), fmaxf(fminf(t1y, t2y), fminf(t1z, t2z)));
    float t_exit = fminf(fmaxf(t1x, t2x), fminf(fmaxf(t1y, t2y), fmaxf(t1z, t2z)));
    const float t_hit = (t_enter > 0.0f) ? t_enter : t_exit;
    const float t_max = c.t;
    const bool candidate = !((t_exit < t_enter) | (t_exit <= 0.0f)) & !((t_hit >= t_max) | (t_hit <= t_min) | (t_hit < EPS));   // cube.rs:90-103, one branch (`|` / `&`: one exec-mask region, not three)
    Probe o; o.t = 0.f; o.aux = t_hit;
    bool acc = false;
    if (candidate) {
        f3 po = ro + rd * t_hit;
        o.po = po;
        f3 pw = xform_o2w_point(pr, po);
        o.t = dot(pw - ro_w, rd_w);                                                         // cube.rs:145-153: the same dot product twice
        acc = !((o.t < 0.0f) | (o.t < t_min) | (o.t > t_max));
    }
    return cand_take(c, acc, i, o);
}
// The record of a cube hit.  The object-space normal is +-e_axis, normalize_or_zero() of such a vector is the vector
// itself (1/sqrt(1) == 1), and the world normal normalized(w2o^T * (n, 0)) therefore takes one of 6 values per cube, which
// the host precomputed with the same f32 operations (DevPrim.d[34..51], rt_api.cpp cube_normal_table).
template <class PrimPtr>
DI f3 cube_po(PrimPtr pr, const Cand& c, f3 ro_w, f3 rd_w) {                                // cube.rs:104, from the slab distance the candidate kept
    const f3 ro = xform_w2o_point(pr, ro_w), rd = xform_w2o_dir(pr, rd_w);
    return ro + rd * c.aux;
}
template <class PrimPtr> DI f3 cube_po(PrimPtr, const CandP& c, f3, f3) { return c.po; }  // ... or the point itself
template <class PrimPtr, class C>
DI void finish_cube(PrimPtr pr, const C& c, f3 ro_w, f3 rd_w, f3& p, f3& outward) {
    const f3 po = cube_po(pr, c, ro_w, rd_w);
    p = xform_o2w_point(pr, po);
    const uint32_t axis = cube_axis(po);
    const float cc = (axis == 0u) ? po.x : ((axis == 1u) ? po.y : po.z);
    f3 nw;
    if (cc != cc) nw = normalized(xform_normal(pr, mk(0.f, 0.f, 0.f)));                     // NaN signum -> normalize_or_zero -> the zero vector (cube.rs:134)
    else {
        const uint32_t code = 2u * axis + (__float_as_uint(cc) >> 31);                      // glam signum: the sign bit decides, also for +-0
        const auto* t = pr->d + 34u + 3u * code;
        nw = mk(t[0], t[1], t[2]);
    }
    outward = nw;
}

// mesh/mesh_object.rs:263-329 + acceleration/bvh.rs:78-170 + acceleration/aabb.rs:27-45.
// Stackless walk over two-link nodes (rt_device.h): hit inner -> left child, everything else -> the escape link, which
// reproduces the reference's left-then-right recursion exactly; `best_t` plays the role of the recursion's shrinking
// t_max.  The walk is split into setup / node step / leaf / finalize so that the state-machine kernel can interleave
// the traversals of different lanes; hit_mesh() composes them into the plain per-lane loop.
struct MeshTrav {
    f3 ro, rd;                 // object-space ray (direction normalised twice, mesh_object.rs:289)
    float ix, iy, iz;          // 1/d, aabb.rs:29 (same value at every node)
    float len_raw;             // |w2o * d_world| for the (sic) t_world formula
    uint32_t node;             // next node to visit; NODE_END: the walk is over
    float best_t; uint32_t best_tri;
    uint32_t leaf_a, leaf_b;   // pending leaf (first triangle, count); leaf_b == 0: none
};
// FAST: the short reciprocal / square root of rt_math.h (same bits; a template argument because the kernels' register allocation
// decides whether the shorter code is also the faster one).
// identity (wave-uniform): the mesh is untransformed and every active lane's ray passed ray_nonzero_finite()
template <bool FAST = false, class PrimPtr>
DI void mesh_setup(PrimPtr pr, f3 ro_w, f3 rd_w, float t_max, MeshTrav& m, bool identity = false) {   // mesh_object.rs:264-291
    f3 rd_raw;
    if (identity) { m.ro = ro_w; rd_raw = rd_w; }
    else { m.ro = xform_w2o_point(pr, ro_w); rd_raw = xform_w2o_dir(pr, rd_w); }
    m.len_raw = len(rd_raw);
    if constexpr (FAST) {
        const f3 once = (m.len_raw < EPS) ? rd_raw : rd_raw * recip_normal_range(m.len_raw);     // normalized(): the length is the one above
        m.rd = normalized<true>(once);
        recip3<true>(m.rd.x, m.rd.y, m.rd.z, m.ix, m.iy, m.iz);
    } else {
    m.rd = normalized(normalized(rd_raw));
    m.ix = 1.0f / m.rd.x; m.iy = 1.0f / m.rd.y; m.iz = 1.0f / m.rd.z;
    }
    m.node = pr->node_begin;
    m.best_t = t_max; m.best_tri = 0xFFFFFFFFu; m.leaf_b = 0; m.leaf_a = 0;
}
// Visit m.node (box test, aabb.rs:27-45).  Afterwards m.node is the next node to visit and, when a leaf was hit, its
// triangles are pending (m.leaf_b > 0) and must be tested before the walk goes on.
// FIXED_AABB: MI355RT_FLAG_FIXED_AABB -- a box is missed only when t_max < t_min (the reference misses on <=, aabb.rs:41).
// USE_LDS: nodes below `lds_count` are read from the workgroup's LDS copy (ds_read_b128), the rest from global memory.
typedef float lds_v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) lds_v4f* lds_nodes_t;
// USE_LDS (the reference build's state machine only; the product kernels read nodes from L1 / L2): 0 = global memory; 1 = the LDS copy when EVERY lane
//          of the wave is below `lds_count` (a wave-uniform choice; a per-lane choice was measured too: the LDS readers then wait for the slowest global load).
// SPEC (wavefront kernel's WALK stage): the walk does not stop at a hit leaf.  The leaf is left pending (leaf_a / leaf_b, and
// `resume` = the node behind it) and the walk goes on with the unchanged best_t; a SECOND leaf while one is pending stalls the lane
// in front of that leaf's node (it is visited again after the leaf phase).  See rt_wavefront.h for why this is exact.
template <bool FIXED_AABB = false, int USE_LDS = 0, bool SPEC = false>
DI void mesh_step(const float4* __restrict__ n4, lds_nodes_t lds, uint32_t lds_count, float t_min, MeshTrav& m, uint32_t* resume = nullptr, bool* stalled = nullptr) {
    // 32-bit byte offset from the uniform base: the load takes the base from SGPRs instead of a 64-bit per-lane address
    // The choice between the LDS copy and global memory is made for the WAVE (the LDS copy is a copy: global memory holds every
    // node): a per-lane choice would make the LDS readers wait for the other lanes' global loads (both paths fill the
    // same registers) and serialise the two latencies.  Whole array in LDS (semesterbild): always the LDS path.
    float4 q0, q1;
    if (USE_LDS != 0 && __ballot(m.node >= lds_count) == 0ull) {
        lds_nodes_t lq = reinterpret_cast<lds_nodes_t>(reinterpret_cast<const __attribute__((address_space(3))) char*>(lds) + (m.node << 5));
        const lds_v4f l0 = lq[0], l1 = lq[1];
        q0 = make_float4(l0.x, l0.y, l0.z, l0.w); q1 = make_float4(l1.x, l1.y, l1.z, l1.w);
    } else {
        const float4* __restrict__ nq = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(n4) + (m.node << 5));
        q0 = nq[0]; q1 = nq[1];
    }
    const uint32_t a = __float_as_uint(q0.w), b = __float_as_uint(q1.w);
    // aabb.rs:31-44 returns false at the first axis whose interval is empty.  tmin only grows and tmax only shrinks from axis
    // to axis (f32::max / f32::min ignore a NaN operand, so they never move the other way), hence an interval that is empty
    // after some axis is still empty after the last one and vice versa: ONE test after the z axis decides the same.
    float tmin = t_min, tmax = m.best_t;
    {   float t0 = (q0.x - m.ro.x) * m.ix, t1 = (q1.x - m.ro.x) * m.ix; if (m.ix < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        tmin = fmaxf(tmin, t0); tmax = fminf(tmax, t1); }
    {   float t0 = (q0.y - m.ro.y) * m.iy, t1 = (q1.y - m.ro.y) * m.iy; if (m.iy < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        tmin = fmaxf(tmin, t0); tmax = fminf(tmax, t1); }
    {   float t0 = (q0.z - m.ro.z) * m.iz, t1 = (q1.z - m.ro.z) * m.iz; if (m.iz < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        tmin = fmaxf(tmin, t0); tmax = fminf(tmax, t1); }
    const bool ok = !(FIXED_AABB ? (tmax < tmin) : (tmax <= tmin));
    // Branch-free successor: hit inner node -> its left child `a`; missed node or leaf -> the escape link (a hit leaf's
    // triangles are tested first: leaf_b > 0 holds the walk until mesh_leaf() has run).
    const uint32_t count = b >> NODE_LINK_BITS, esc = b & NODE_END;
    if (SPEC) {
        const bool leaf = ok && count != 0u;
        const bool second = leaf && m.leaf_b != 0u, take_leaf = leaf && m.leaf_b == 0u;
        m.node = second ? m.node : ((ok && count == 0u) ? a : esc);
        *resume = take_leaf ? esc : *resume;
        m.leaf_a = take_leaf ? a : m.leaf_a;
        m.leaf_b = take_leaf ? count : m.leaf_b;
        *stalled = second;
    } else {
        const bool take_leaf = ok && count != 0u;
        m.node = (ok && count == 0u) ? a : esc;
        m.leaf_a = take_leaf ? a : m.leaf_a;
        m.leaf_b = take_leaf ? count : m.leaf_b;
    }
}
// Moeller-Trumbore over the pending leaf, bvh.rs:91-138
DI void mesh_leaf(const float4* __restrict__ t4, float t_min, MeshTrav& m) {
    for (uint32_t k = 0; k < m.leaf_b; ++k) {
        const float4* __restrict__ tq = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(t4) + (m.leaf_a + k) * 48u);
        const float4 r0 = tq[0], r1 = tq[1], r2 = tq[2];
        const f3 v0 = mk(r0.x, r0.y, r0.z), e1 = mk(r0.w, r1.x, r1.y), e2 = mk(r1.z, r1.w, r2.x);
        f3 hh = cross(m.rd, e2);
        float aa = dot(e1, hh);
        float f = 1.0f / aa;                                 // (the short reciprocal of rt_math.h, guarded by a ballot on |aa| >= 2^126, measured +-0.1 %: not taken)
        f3 s = m.ro - v0;
        float u = f * dot(s, hh);
        f3 q = cross(s, e1);
        float v = f * dot(m.rd, q);
        float t = f * dot(e2, q);
        // bvh.rs:99-116, the four `continue`s as one predicate (same values, one branch)
        const bool hit = !(fabsf(aa) < EPS) && (u >= 0.0f && u <= 1.0f) && !(v < 0.0f || u + v > 1.0f) && (t > t_min && t < m.best_t);
        if (hit) { m.best_t = t; m.best_tri = m.leaf_a + k; }
    }
    m.leaf_b = 0;
}
// The end of Mesh::hit that decides acceptance (mesh_object.rs:312-318); the record is built by finish_mesh() for the winner.
DI bool mesh_accept(uint32_t i, const MeshTrav& m, f3 rd_w, float t_min, Cand& c) {
    float t_world = m.best_t * m.len_raw / len(rd_w);                   // (sic) mesh_object.rs:312-314
    const bool acc = (m.best_tri != 0xFFFFFFFFu) && !(t_world < t_min || t_world > c.t);
    c.aux2 = acc ? m.best_tri : c.aux2;
    Probe o; o.t = t_world; o.aux = m.best_t;
    return cand_take(c, acc, i, o);
}
// mesh_object.rs:264-310 for the winning triangle: the object-space ray is recomputed exactly as mesh_setup() computed it.
template <class PrimPtr>
DI void finish_mesh(PrimPtr pr, const float4* __restrict__ t4, const Cand& c, f3 ro_w, f3 rd_w, f3& p, f3& outward) {
    const f3 ro = xform_w2o_point(pr, ro_w);
    const f3 rd = normalized(normalized(xform_w2o_dir(pr, rd_w)));
    const float4 r2 = t4[3 * (size_t)c.aux2 + 2];
    f3 tn = mk(r2.y, r2.z, r2.w);
    f3 pos_obj = ro + rd * c.aux;
    f3 n_obj = (dot(rd, tn) < 0.0f) ? tn : -tn;                         // bvh.rs:118-124
    p = xform_o2w_point(pr, pos_obj);
    outward = normalized(xform_normal(pr, n_obj));
}
DI bool hit_mesh(cprim_t pr, uint32_t i, const DevNode* __restrict__ nodes, const DevTri* __restrict__ tris, f3 ro_w, f3 rd_w,
                 float t_min, Cand& c) {
    const float4* __restrict__ n4 = reinterpret_cast<const float4*>(nodes);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(tris);
    MeshTrav m; mesh_setup(pr, ro_w, rd_w, c.t, m);
    while (m.node != NODE_END) {
        mesh_step(n4, nullptr, 0u, t_min, m);
        if (m.leaf_b) mesh_leaf(t4, t_min, m);
    }
    return mesh_accept(i, m, rd_w, t_min, c);
}

// The HitRecord of the list's winner (hittable.rs:10-27), once per ray.  Lanes of a wave may have different winners, so
// the primitive record is read per lane here (global loads; L1/L2 resident).
// KINDS: bit k set = a primitive of kind k (MI355RT_PRIM_*) may occur in the list.  Like the material sets (MATS): set_scene knows which kinds a list holds and picks an
// instantiation whose set covers them; the run checks of the other kinds in the walk and their branches here are compiled out (rt_device.h, PRIMS_*).
template <bool HAS_MESH, bool SHARED_TAIL = !HAS_MESH, bool CARRY_Q0 = false, uint32_t KINDS = PRIMS_ALL, class C>
DI void finish_hit(const DevPrim* __restrict__ prims, const DevTri* __restrict__ tris, const C& c, f3 ro, f3 rd, Hit& h) {
#define MI_KIND(k) (((KINDS >> (k)) & 1u) != 0u)
    const DevPrim* __restrict__ pr = prims + c.idx;
    const uint32_t kind = pr->kind;
    if constexpr (CARRY_Q0) h.q0 = *reinterpret_cast<const float4*>(pr->mat0);
    // Mesh-free lists: each kind only says where the hit is and which way its surface faces; HitRecord::set_face_normal
    // (hittable.rs:19-26) then runs once for all lanes of the wave, whatever their winners are (cornell -2.5 %).  With meshes in
    // the list every kind finishes its own record (measured: the shared tail costs the wavefront kernel 3-4 %).
    constexpr bool shared_tail = SHARED_TAIL;
    if (shared_tail) {
        f3 p = ro + rd * c.t, outward;                                        // sphere.rs:35, plane.rs:40, quad.rs:103
        if (MI_KIND(MI355RT_PRIM_QUAD) && kind == MI355RT_PRIM_QUAD) {        // quad.rs:103-131
            outward = mk(pr->d[0], pr->d[1], pr->d[2]);                     // dot(ray.direction, normal): the same sum of the same products as `denom`
        } else if (MI_KIND(MI355RT_PRIM_CUBE) && (kind == MI355RT_PRIM_CUBE || KINDS == PRIMS_QUAD_CUBE)) {      // (two kinds: what is not a quad is a cube)
            finish_cube(pr, c, ro, rd, p, outward);
        } else if (MI_KIND(MI355RT_PRIM_SPHERE) && kind == MI355RT_PRIM_SPHERE) {   // sphere.rs:35-52
            outward = divf(p - mk(pr->d[0], pr->d[1], pr->d[2]), pr->d[3]);
        } else if (MI_KIND(MI355RT_PRIM_PLANE) && kind == MI355RT_PRIM_PLANE) {     // plane.rs:40-55
            outward = mk(pr->d[3], pr->d[4], pr->d[5]);
        } else if (HAS_MESH) {
            finish_mesh(pr, reinterpret_cast<const float4*>(tris), c, ro, rd, p, outward);
        }
        h.t = c.t; h.p = p;
        set_face(h, rd, outward, pr->material);
    } else {
        h.t = c.t;
        if (kind == MI355RT_PRIM_QUAD) {
            h.p = ro + rd * c.t;
            set_face(h, rd, mk(pr->d[0], pr->d[1], pr->d[2]), pr->material);
        } else if (kind == MI355RT_PRIM_CUBE) {
            f3 outward; finish_cube(pr, c, ro, rd, h.p, outward);
            set_face(h, rd, outward, pr->material);
        } else if (kind == MI355RT_PRIM_SPHERE) {
            h.p = ro + rd * c.t;
            set_face(h, rd, divf(h.p - mk(pr->d[0], pr->d[1], pr->d[2]), pr->d[3]), pr->material);
        } else if (kind == MI355RT_PRIM_PLANE) {
            h.p = ro + rd * c.t;
            set_face(h, rd, mk(pr->d[3], pr->d[4], pr->d[5]), pr->material);
        } else if (HAS_MESH) {
            f3 outward; finish_mesh(pr, reinterpret_cast<const float4*>(tris), c, ro, rd, h.p, outward);
            set_face(h, rd, outward, pr->material);
        }
    }
}

#undef MI_KIND
DI uint32_t prim_material_kind(const RenderParams& P, uint32_t idx) { return __float_as_uint(P.prims[idx].mat0[0]); }   // one read, not two dependent ones

// hittable.rs:45-58 -- HittableList::hit with t_min = EPSILON, t_max = INFINITY (renderer.rs:24)
template <bool HAS_MESH, uint32_t KINDS = PRIMS_ALL, class C>
DI void walk_list(cprim_t prims, uint32_t n_prims, const DevNode* __restrict__ nodes, const DevTri* __restrict__ tris, f3 ro, f3 rd, C& c) {
#define MI_KIND(k) (((KINDS >> (k)) & 1u) != 0u)
    // Same order as the list, but the dispatch on the kind (wave-uniform: a scalar branch) is taken once per RUN of equal kinds
    // (DevPrim.run_end, host-computed) and each kind has its own tight loop: the structurised switch inside one loop carried the
    // candidate through a chain of merge blocks with register copies at every one of them.
    // The kinds are tried in a fixed cyclic order, each as `if (the run at i is of this kind) loop over the run`: plain nested
    // structured control flow (a `switch` here is lowered to a chain of flow blocks, each with its own copies of the candidate).
    // (Round 5 measured the tidier form again -- ONE read of the run's header, kind and run_end together, and a wave-uniform `else if` chain over the
    // kinds: cornell +2.8 %, profiles/r05/ab_scalar_diet.txt.)
    uint32_t i = 0;
    while (i < n_prims) {
#define MI_RUN(KIND, CALL) if (i < n_prims && prims[i].kind == (KIND)) { const uint32_t end = min(prims[i].run_end, n_prims); do { CALL; } while (++i < end); }
        if (MI_KIND(MI355RT_PRIM_QUAD)) { MI_RUN(MI355RT_PRIM_QUAD,   hit_quad<!HAS_MESH>(prims + i, i, ro, rd, EPS, c)) }
        if (MI_KIND(MI355RT_PRIM_CUBE)) { MI_RUN(MI355RT_PRIM_CUBE,   hit_cube<!HAS_MESH>(prims + i, i, ro, rd, EPS, c)) }
        if (MI_KIND(MI355RT_PRIM_SPHERE)) { MI_RUN(MI355RT_PRIM_SPHERE, hit_sphere(prims + i, i, ro, rd, EPS, c)) }
        if (MI_KIND(MI355RT_PRIM_PLANE)) { MI_RUN(MI355RT_PRIM_PLANE,  hit_plane(prims + i, i, ro, rd, EPS, c)) }
        if (HAS_MESH) { MI_RUN(MI355RT_PRIM_MESH, hit_mesh(prims + i, i, nodes, tris, ro, rd, EPS, c)) }
        else if (KINDS == PRIMS_ALL) { if (i < n_prims && prims[i].kind >= MI355RT_PRIM_MESH) ++i; }   // cannot happen (the host picks this kernel only for mesh-free lists); keeps the loop finite
        else if (i < n_prims && ((KINDS >> prims[i].kind) & 1u) == 0u) ++i;                               // likewise for a pruned set of kinds
#undef MI_RUN
#undef MI_KIND
    }
}
// CARRY_PO: the candidate keeps the cube's object-space hit point (CandP).  On for both mesh-free kernels: the Lambert-only one
// (cornell -1.8 % at 72 VGPRs) and the general one, which needs 80 VGPRs = 6 waves per SIMD for it (veach-mis: +1.9 % at 72 with
// spills, -2.6 % at 80; see MI355RT_OCC_LOCKSTEP).
template <bool HAS_MESH, bool CARRY_PO = false, uint32_t KINDS = PRIMS_ALL>
DI bool hit_scene(cprim_t prims, uint32_t n_prims, const DevNode* __restrict__ nodes, const DevTri* __restrict__ tris,
                  f3 ro, f3 rd, Hit& best) {
    typename std::conditional<CARRY_PO && !HAS_MESH, CandP, Cand>::type c; cand_reset(c);
    walk_list<HAS_MESH, KINDS>(prims, n_prims, nodes, tris, ro, rd, c);
    if (c.idx == CAND_NONE) return false;
    finish_hit<HAS_MESH, !HAS_MESH, !HAS_MESH, KINDS>((const DevPrim*)prims, tris, c, ro, rd, best);          // (the lockstep kernels' entry: q0 rides along when there is no mesh)
    return true;
}


}  // namespace mi355rt
