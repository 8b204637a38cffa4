"""mi355rt_context_set_scene picks the kernel.  For a mesh-free scene that mixes a rough conductor with another scattering material the
materials only say that the wavefront form MAY pay; a probe render decides (rays per path >= 1.6: the sorted SHADE passes have something
to sort).  veach-mis (2.5 rays per path) lands on the wavefront kernel (variant 11) -- every other test scene with a rough conductor does.
Here the other branch: the same kinds of material in a scene that is mostly sky stays on the lockstep kernel (variant 9), and the two kernels
agree on it bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def sparse_scene(abi, host, width=160, height=120, spp=8, depth=12):
    from oracle import scene_loader as L
    F = np.float32

    def mat(kind, albedo=(0, 0, 0), p0=0.0, eta=(0, 0, 0), k=(0, 0, 0)):
        m = abi.Material(); m.kind = kind
        m.albedo[:] = [float(F(v)) for v in albedo]; m.p0 = float(F(p0)); m.eta[:] = [float(F(v)) for v in eta]; m.k[:] = [float(F(v)) for v in k]
        return m

    mats = [mat(abi.MAT_ROUGH_GGX, (0.9, 0.8, 0.7), p0=0.2, eta=(0.2, 1.09, 1.42), k=(3.91, 2.57, 2.30)), mat(abi.MAT_LAMBERT_SOLID, (0.6, 0.3, 0.2)),
            mat(abi.MAT_EMISSIVE, (4.0, 4.0, 3.0))]
    prims = []
    for centre, radius, material in (((-1.5, 0.0, 0.0), 0.45, 0), ((1.4, 0.3, -1.0), 0.5, 1), ((0.0, 2.6, -2.0), 0.3, 2)):
        p = abi.Primitive(); p.kind = abi.PRIM_SPHERE; p.material = material
        p.data[0:4] = [float(F(v)) for v in centre] + [float(F(radius))]
        prims.append(p)
    sc = L.LoadedScene()
    sc.materials, sc.primitives, sc.meshes = mats, prims, []
    sc.triangles = np.zeros((0, 12), F)
    sc.finalize()
    sc.c.miss_color[:] = [0.5, 0.5, 0.5]
    sc.camera = L.camera_new((0.0, 1.0, 9.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), F(50.0), F(width / height))
    sc.settings = abi.Settings(width, height, spp, depth)
    return sc


def test_sparse_rough_conductor_scene_stays_on_the_lockstep_kernel(native, abi):
    host, device = native
    sc = sparse_scene(abi, host)
    n = sc.settings.width * sc.settings.height
    outs = {}
    for forced in (None, 11, 9):
        ctx = device.Context(0)
        try:
            if forced is not None:
                ctx.set_knob("kernel", forced)
            ctx.set_scene(sc, sc.camera, sc.settings)
            if forced is None:
                assert ctx.kernel_variant() == 9                  # the probe found ~1.1 rays per path: nothing to sort, the queues would only cost
            else:
                assert ctx.kernel_variant() == forced
            packed = torch.zeros(n, dtype=torch.int32, device="cuda")
            linear = torch.zeros(n * 3, dtype=torch.float32, device="cuda")
            st = ctx.render(packed.data_ptr(), linear.data_ptr(), abi.Options.make(), None, want_stats=True)
            outs[forced] = (packed.cpu().numpy(), linear.cpu().numpy().view(np.uint32), st.rays, st.samples)
        finally:
            ctx.close()
    assert 1.0 < outs[None][2] / outs[None][3] < 1.6                 # the figure the probe estimates, on the full render
    for forced in (11, 9):
        assert np.array_equal(outs[forced][0], outs[None][0]) and np.array_equal(outs[forced][1], outs[None][1]) and outs[forced][2:] == outs[None][2:], forced
