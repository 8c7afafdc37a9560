"""CPU (no GPU): the C-ABI library builds, loads and exports every symbol include/centernet_hip.h declares; host logic
(state_dict surface, flat Adam, gradient buckets over gloo with world_size 2, synthetic data) behaves."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def built_lib():
    from centernet_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _hip


def test_library_exports_every_declared_symbol(built_lib):
    protos = built_lib.parse_header()
    assert len(protos) >= 40
    lib = built_lib.lib()
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/centernet_hip.h but not exported"
    assert lib.cn_version() >= 100
    assert lib.cn_last_error() is not None
    # every family the hot path needs is present
    for fam in ("cn_conv2d_fwd", "cn_conv2d_wgrad", "cn_dcn_im2col", "cn_dcn_col2im", "cn_bn_train_fwd", "cn_bn_train_bwd",
                "cn_maxpool_fwd", "cn_dwdeconv_fwd", "cn_focal_fwd", "cn_gather_l1_fwd", "cn_ctdet_decode",
                "cn_multi_pose_decode", "cn_adam_step", "cn_nchw_to_nhwc"):
        assert fam in protos


def test_missing_extension_fails_loudly(monkeypatch):
    from centernet_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libcenternet_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _hip.lib()


def test_ops_reject_cpu_tensors(built_lib):
    with pytest.raises(RuntimeError, match="no CPU path"):
        built_lib.call("cn_add", torch.zeros(8), torch.zeros(8), torch.zeros(8), 8, 0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "centernet-pytorch-lightning_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} reaches into oracle/"


@pytest.mark.parametrize("arch", ["res_18", "res_101", "dla_34", "resdcn_18", "resdcn_101", "hourglass"])
def test_state_dict_surface_matches_reference_layout(arch):
    from centernet_amd.models import create_model
    from oracle import models_ref
    a = create_model(arch).state_dict()
    b = models_ref.create_model(arch).state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    if arch == "dla_34":
        assert len(a) == 386 and a["dla_up.ida_0.proj_1.conv.conv_offset_mask.weight"].shape == (27, 512, 3, 3)
        assert a["ida_up.up_2.weight"].shape == (64, 1, 8, 8)


def test_task_module_surface():
    from centernet_amd import CenterNetDetection, CenterNetMultiPose
    m = CenterNetDetection("dla_34")
    assert (m.head_conv, m.num_stacks, m.padding, m.down_ratio, m.max_objs) == (256, 1, 31, 4, 128)
    assert list(m.heads[0].heads) == ["heatmap", "width_height", "regression"]
    assert m.heads[0].heatmap.fc[2].bias.detach().unique().item() == pytest.approx(-2.19)
    assert m.hparams.wh_weight == 0.1 and m.hparams.learning_rate == 1e-4 and len(m.valid_ids) == 80
    p = CenterNetMultiPose("res_18")
    assert p.head_conv == 64 and list(p.heads[0].heads)[3:] == ["heatmap_keypoints", "keypoints", "heatmap_keypoints_offset"]
    assert p.test_max_per_image == 20 and p.hparams.hp_weight == 1
    hg = CenterNetDetection("hourglass")
    assert (hg.head_conv, hg.num_stacks, hg.padding, len(hg.heads), hg.backbone.out_channels) == (256, 2, 127, 2, 256)
    opt, sched = CenterNetDetection("res_18", learning_rate_milestones=[2, 4]).configure_optimizers()
    assert sched[0]["interval"] == "epoch" and opt[0].param_groups[0]["lr"] == 1e-4


def test_flat_adam_cpu_matches_torch_adam():
    from centernet_amd.engine import FlatAdam
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref.load_state_dict(net.state_dict())
    o1, o2 = FlatAdam(net.parameters(), lr=1e-2), torch.optim.Adam(ref.parameters(), lr=1e-2)
    for it in range(4):
        x = torch.randn(6, 7)
        o1.zero_grad(); o2.zero_grad()
        net(x).square().mean().backward(); ref(x).square().mean().backward()
        assert net[0].weight.grad.data_ptr() == o1.flat_g.data_ptr()      # autograd accumulates straight into the flat buffer
        o1.step(); o2.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def _ddp_worker(rank, world, port, out, between_graphs=False):
    import torch.distributed as dist
    from centernet_amd.engine import FlatAdam, GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 8))
    dead = torch.nn.Linear(3, 3)                        # never used: no gradient, like the reference's dead DLA projections
    params = list(net.parameters()) + list(dead.parameters())
    opt = FlatAdam(params, lr=1e-2)
    sync = GradSync(opt, bucket_bytes=256, hooks=not between_graphs)      # graph mode: no hooks, one collective after backward
    sync.broadcast_state(net)
    assert len(sync.buckets) > 2
    for it in range(3):
        g = torch.Generator().manual_seed(1000 * it + rank)
        x = torch.randn(4, 16, generator=g)
        opt.zero_grad()
        loss = net(x).square().mean()
        if between_graphs:
            loss.backward(); sync.allreduce_all()
        else:
            sync.begin(); loss.backward(); sync.finish()
        opt.step()
    out[rank] = opt.flat_p.clone()
    dist.destroy_process_group()


@pytest.mark.parametrize("between_graphs", [False, True])
def test_gradient_buckets_over_gloo_world2(between_graphs):
    """N>1 path on CPU: ranks end bit-identical, and equal to one process that averages the two gradients itself — with the
    backward-overlapped buckets of eager mode and with the single collective graph mode runs between its two hipGraphs."""
    from centernet_amd.engine import FlatAdam
    port = 29000 + os.getpid() % 2000 + (7 if between_graphs else 0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, port, out, between_graphs), nprocs=2, join=True)
    assert torch.equal(out[0], out[1])
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 8))
    dead = torch.nn.Linear(3, 3)
    opt = FlatAdam(list(net.parameters()) + list(dead.parameters()), lr=1e-2)
    for it in range(3):
        opt.zero_grad()
        for r in range(2):
            g = torch.Generator().manual_seed(1000 * it + r)
            (net(torch.randn(4, 16, generator=g)).square().mean() / 2).backward()
        opt.step()
    n = sum(p.numel() for p in net.parameters())
    torch.testing.assert_close(out[0][:n], opt.flat_p[:n], rtol=1e-5, atol=1e-6)


def test_synthetic_batches():
    from centernet_amd import synth
    x, t = synth.ctdet_batch(3, 2, 128, 128)
    assert x.shape == (2, 3, 128, 128) and t["heatmap"].shape == (2, 80, 32, 32) and t["indices"].dtype == torch.int64
    assert t["regression_mask"].dtype == torch.bool and int(t["regression_mask"].sum()) >= 2
    assert float(t["heatmap"].max()) == 1.0
    x2, t2 = synth.ctdet_batch(3, 2, 128, 128)
    assert torch.equal(x, x2) and all(torch.equal(t[k], t2[k]) for k in t)
    _, p = synth.pose_batch(3, 2, 128, 128)
    assert p["heatmap"].shape == (2, 1, 32, 32) and p["keypoints_mask"].shape == (2, 128, 34)
    assert p["heatmap_keypoints_indices"].shape == (2, 128 * 17) and p["heatmap_keypoints"].shape == (2, 17, 32, 32)
