"""N > 1 path on CPU: world_size-2 gloo processes run the product's strip plan + gather + de-interleave
(raytracer-rust_amd/distributed.py) with the oracle injected as the renderer, and must reproduce the
single-process image bit-for-bit (RNG keyed by absolute row, so tiling cannot change pixels)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SCENES, pkg


def _worker(rank, world, port, strip_rows, out_path, W, H, collective="all_gather"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from oracle import scene_loader
    abi, rtdist = pkg("abi"), pkg("distributed")
    sc = scene_loader.load_scene(SCENES["cornell"], width=W, height=H, spp=2, max_depth=4)
    plan = rtdist.make_plan(H, W, world, strip_rows)
    opt = plan.options_for(abi, rank)
    packed, _, _ = oracle.render(sc, sc.camera, sc.settings, opt, threads=1, want_linear=False)   # stand-in for ctx.render
    assert packed.shape[0] == len(plan.rows[rank])
    local = torch.zeros((plan.max_rows, W), dtype=torch.int32)
    local[:packed.shape[0]] = torch.from_numpy(packed.astype(np.int32))
    img = rtdist.gather_image(local, plan, rank, collective=collective)
    if rank == 0:
        np.save(out_path, img.numpy().astype(np.uint32))
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("strip_rows,H", [(4, 24), (3, 20), (1, 7)])     # even split, ragged split (padding), single rows
def test_two_rank_gather_is_bit_identical(strip_rows, H, tmp_path, oracle_mod, abi):
    from oracle import scene_loader
    W = 16
    out = str(tmp_path / "img.npy")
    port = 29500 + (os.getpid() + strip_rows * 7 + H) % 2000
    mp.spawn(_worker, args=(2, port, strip_rows, out, W, H), nprocs=2, join=True)
    got = np.load(out)
    sc = scene_loader.load_scene(SCENES["cornell"], width=W, height=H, spp=2, max_depth=4)
    want, _, _ = oracle_mod.render(sc, sc.camera, sc.settings, abi.Options.make(), threads=1, want_linear=False)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("world,collective", [(2, "gather"), (8, "all_gather")])
def test_rooted_gather_and_the_eight_rank_plan(world, collective, tmp_path, oracle_mod, abi):
    """The 8-GPU decomposition of the bench (strip plan chosen by make_plan itself) with 8 gloo ranks, and the rooted-gather form."""
    from oracle import scene_loader
    W, H = 8, 40
    out = str(tmp_path / "img.npy")
    port = 31500 + (os.getpid() + world) % 2000
    mp.spawn(_worker, args=(world, port, None, out, W, H, collective), nprocs=world, join=True)
    got = np.load(out)
    sc = scene_loader.load_scene(SCENES["cornell"], width=W, height=H, spp=2, max_depth=4)
    want, _, _ = oracle_mod.render(sc, sc.camera, sc.settings, abi.Options.make(), threads=1, want_linear=False)
    assert np.array_equal(got, want)


def test_plan_covers_every_row_once():
    rtdist = pkg("distributed")
    for H, world, strip in [(600, 8, None), (600, 1, None), (1080, 8, None), (7, 3, 2), (5, 8, 1)]:
        plan = rtdist.make_plan(H, 4, world, strip)
        allrows = sorted(y for r in plan.rows for y in r)
        assert allrows == list(range(H))
        pos = plan.perm.tolist()
        assert len(set(pos)) == H and max(pos) < world * plan.max_rows
        for r, rr in enumerate(plan.rows):
            assert [pos[y] for y in rr] == [r * plan.max_rows + i for i in range(len(rr))]
    assert rtdist.make_plan(600, 4, 8).strip_rows * 8 * (600 // (rtdist.make_plan(600, 4, 8).strip_rows * 8)) == 600   # no padding at 8 ranks


@pytest.mark.parametrize("W,H", [(800, 600), (1920, 1080)])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_plans_are_balanced_and_unpadded(W, H, world, abi):
    """What bench.py does per rank for --gpus 2 / 4 / 8 at the two BASELINE image sizes (rows are the reference's unit,
    /root/reference/src/renderer.rs:87-91): make_plan -> options_for -> (render) -> gather_image.  Load balance is the first thing an
    8-GPU line is judged on: every rank must get the same number of rows (= samples) with zero padding rows in the gather buffer,
    the ABI's own row selection must agree with the plan, and the de-interleave must put every row where it belongs."""
    rtdist = pkg("distributed")
    plan = rtdist.make_plan(H, W, world)
    assert plan.max_rows * world == H, "padding rows in the gather buffer"
    assert {len(r) for r in plan.rows} == {H // world}, "ranks render different numbers of rows"
    assert 1 <= plan.strip_rows <= 4                                         # narrow strips: sky rows and object rows mix on every rank
    for rank in range(world):
        opt = plan.options_for(abi, rank)
        assert (opt.strip_rows, opt.n_parts, opt.part) == (plan.strip_rows, world, rank)
        assert abi.rows_selected(H, opt) == plan.rows[rank]                  # the library renders exactly the rows the plan gathers
    # the de-interleave of gather_image: rank-major stacked strips -> image order (row y carries the value y)
    stacked = torch.cat([torch.tensor(plan.rows[r], dtype=torch.int32).unsqueeze(1).expand(-1, 3) for r in range(world)])
    image = stacked.index_select(0, plan.perm_on("cpu"))
    assert torch.equal(image[:, 0], torch.arange(H, dtype=torch.int32))
    # per-rank cost balance on the real picture: with strips of <= 4 rows every rank's rows are spread over the whole image height
    for rr in plan.rows:
        assert rr[0] < plan.strip_rows * world and rr[-1] >= H - plan.strip_rows * world


def test_bench_exports_the_ipc_mode_before_torch_is_imported():
    """RCCL's intra-node transport needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool (dmabuf IPC only); bench.py must have it in the
    environment before the HIP runtime comes up, i.e. textually before its `import torch`."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for fn in ("def main(argv=None):", "def main_single_process(args):"):          # a rank under a launcher; one process driving every device
        body = src[src.index(fn):]
        assert 0 < body.index('\n    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < body.index("\n    import torch\n")
    launch = src[src.index("def self_launch("):src.index("def main_single_process(")]    # the parent that starts the ranks: in the children's environment,
    assert 'env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")' in launch and "import torch" not in launch   # and no torch in the parent at all
