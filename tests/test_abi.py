"""The C-ABI libraries load without a GPU and export exactly what include/mi355rt.h declares; the ctypes
mirror matches the C compiler's struct sizes; the product path refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, pkg


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "mi355rt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(native):
    host, device = native
    b = pkg("build")
    libs = [C.CDLL(b.DEVICE_SO), C.CDLL(b.HOST_SO)]
    names = _declared_functions()
    assert len(names) >= 18
    missing = [n for n in names if not any(hasattr(L, n) for L in libs)]
    assert not missing, missing
    for n in device.EXPORTS:
        assert hasattr(libs[0], n)


def test_struct_sizes_match_the_c_compiler(native, abi):
    host, _ = native
    out = (C.c_uint32 * 16)()
    n = host.lib().mi355rt_host_struct_sizes(out, 16)
    got = list(out)[:n]
    want = [C.sizeof(t) for t in (abi.Camera, abi.Settings, abi.Material, abi.Primitive, abi.Triangle, abi.BvhNode, abi.Mesh,
                                  abi.Scene, abi.Options, abi.Stats, abi.LoadOverrides)]
    assert got == want
    assert [C.sizeof(abi.Camera), C.sizeof(abi.Material), C.sizeof(abi.Primitive), C.sizeof(abi.Triangle), C.sizeof(abi.BvhNode)] == [56, 64, 144, 48, 40]


def test_abi_version_and_row_selection(native, abi):
    _, device = native
    L = device.lib()
    assert L.mi355rt_abi_version() == abi.ABI_VERSION
    st = abi.Settings(8, 10, 1, 1)
    n = C.c_uint32()
    assert L.mi355rt_rows_selected(C.byref(st), None, C.byref(n)) == 0 and n.value == 10
    opt = abi.Options.make(strip_rows=2, n_parts=3, part=1)
    assert L.mi355rt_rows_selected(C.byref(st), C.byref(opt), C.byref(n)) == 0
    assert n.value == len(abi.rows_selected(10, opt)) == 4            # strips (2,3) and (8,9)
    bad = abi.Options.make(n_parts=2, part=2)
    assert L.mi355rt_rows_selected(C.byref(st), C.byref(bad), C.byref(n)) == abi.ERR_INVALID
    assert b"row selection" in L.mi355rt_last_error()
    zero = abi.Settings(0, 10, 1, 1)
    assert L.mi355rt_rows_selected(C.byref(zero), None, C.byref(n)) == abi.ERR_INVALID


def test_no_cpu_fallback_without_a_device(native, abi):
    """Without a GPU the render entry points must fail loudly (never compute on the CPU)."""
    _, device = native
    h = C.c_void_p()
    rc = device.lib().mi355rt_context_create(0, C.byref(h))
    if rc == 0:
        device.lib().mi355rt_context_destroy(h)
        pytest.skip("a GPU is visible here")
    assert rc == abi.ERR_NO_DEVICE
    assert b"no CPU path" in device.lib().mi355rt_last_error()
    sc = abi.Scene(); sc.miss_color[:] = [0.5] * 3
    cam = abi.Camera(); st = abi.Settings(4, 4, 1, 1)
    with pytest.raises(device.RenderError) as e:
        device.render(sc, cam, st)
    assert e.value.rc == abi.ERR_NO_DEVICE


def test_product_libraries_read_no_environment_and_hold_no_retired_kernels(native):
    """include/mi355rt.h promises that nothing but the structs selects behaviour: neither library may import getenv.  The
    earliest generation of the mesh path (the state machine) is compiled into the tests' reference build only."""
    import subprocess
    b = pkg("build")
    for so in (b.DEVICE_SO, b.HOST_SO):
        undef = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
        assert "getenv" not in undef, so
    syms = subprocess.run(["nm", b.DEVICE_SO], capture_output=True, text=True, check=True).stdout
    kernels = set(re.findall(r"k_render_ctr_[a-z_]+", syms))
    assert kernels == {"k_render_ctr_nomesh", "k_render_ctr_simple", "k_render_ctr_simple_qc", "k_render_ctr_nospec", "k_render_ctr_mesh",
                       "k_render_ctr_wf", "k_render_ctr_wf_nometal", "k_render_ctr_wf_nometal_ident", "k_render_ctr_wf_nometal_shallow", "k_render_ctr_wf_meshfree", "k_render_ctr_wf_fixaabb"}, kernels
