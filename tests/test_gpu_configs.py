"""GPU: the BASELINE.json configurations at FULL size (C2 ResNet-18 bf16 bs=32 512x512, C3 DLA-34 ctdet bf16 bs=64 512x512 — the
headline — and C5 DLA-34 multi_pose bf16 bs=32 512x512), whole-network bf16 accuracy against the ORACLE (not against this package's own fp32 run), the reference-shaped
backbone <-> head seam (NCHW fp32 in both directions) and the eval -> train -> eval cache regression.

Full-size runs cannot be compared with a CPU run of the whole batch in seconds, so they are held to size-independent
properties (sorted scores, score == heat[class, index], every pick a 3x3 maximum, finite gradients) plus an oracle comparison
of a 2-image sub-batch in eval mode with a bf16-aware tolerance (stated at each assert)."""
import numpy as np
import pytest
import torch

from centernet_amd import rng, synth
from oracle import models_ref, ops_ref
from conftest import strided

pytestmark = pytest.mark.gpu
DEV = "cuda"
torch.set_num_threads(16)       # the GPU host has 256 hardware threads; torch's intra-op pool gets slower beyond ~16


def _model(arch, seed, dtype, task="ctdet"):
    from centernet_amd.centernet_detection import CenterNetDetection
    from centernet_amd.centernet_multi_pose import CenterNetMultiPose
    m = (CenterNetDetection if task == "ctdet" else CenterNetMultiPose)(arch, compute_dtype=dtype)
    rng.fill_state_dict(m, seed)
    return m.to(DEV)


def _to_dev(batch):
    x, tgt = batch
    return x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()}


def _repeat(batch, B):
    x, tgt = batch
    rep = (B + x.shape[0] - 1) // x.shape[0]
    return (x.repeat(rep, 1, 1, 1)[:B], {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B] for k, v in tgt.items()})


def _peak_properties(heat, scores, inds, clses, K):
    """heat [B,C,H,W] post-sigmoid; scores [B,K] fp32, inds [B,K] flat y*W+x, clses [B,K]."""
    B, C, H, W = heat.shape
    assert bool((scores[:, :-1] >= scores[:, 1:]).all()), "scores sorted descending"
    flat = heat.reshape(B, -1)
    at = clses.long() * H * W + inds.long()
    assert torch.equal(torch.gather(flat, 1, at), scores), "score == heat[class, index]"
    pooled = torch.nn.functional.max_pool2d(heat, 3, 1, 1)
    peaks = torch.where(pooled == heat, heat, torch.zeros_like(heat)).reshape(B, -1)
    assert torch.equal(torch.gather(peaks, 1, at), scores), "every pick is a 3x3 local maximum"
    assert bool(((peaks > scores[:, -1:]).sum(1) <= K - 1).all()), "no unpicked peak beats the K-th score"


def _rel_range_err(got, ref):
    """max |got - ref| in units of the bf16-aware scale of the map: its RANGE (what the network's arithmetic error is relative
    to) plus 1/3 bf16 ulp of its largest magnitude per percent — the head output is stored in bf16 before it becomes the public
    fp32 map, so a near-constant map (heat logits of an untrained net: -2.19 +- 0.05) carries 2^-9 * 2.19 of pure storage
    rounding however accurate the network is.  scale = range + 2^-8 * max|ref| / 3e-2, so that `err < 3e-2` reads: within 3 % of
    the range plus one bf16 ulp of the magnitude."""
    ref = ref.double()
    scale = ((ref.max() - ref.min()) + ref.abs().max() * 2.0 ** -8 / 3e-2).clamp_min(1e-12)
    d = got.double().cpu() - ref
    return float(d.abs().max() / scale), float(d.pow(2).mean().sqrt() / scale)


def _assert_bf16_close(tag, got, ref, max_tol=6e-2, rms_tol=2e-2):
    """bf16 whole-network tolerance, stated: the WORST of the map's ~10^5..10^6 elements within 6 % of the scale above (measured
    on MI355X in eval mode: 2.6-3.7 % for ResNet-18 / DLA-34 at 256-512 px), the RMS error within 2 % (measured 0.4-1.1 %; a
    near-constant heat map alone carries 0.7 % of bf16 storage rounding)."""
    mx, rms = _rel_range_err(got, ref)
    print(f"{tag}: max err / scale = {mx:.3e}, rms err / scale = {rms:.3e}")
    assert mx < max_tol and rms < rms_tol, (tag, mx, rms)


# ------------------------------------------------------------------------------------------------ C2
def test_c2_res18_bf16_bs32_512_full_size():
    """BASELINE config C2: ResNet-18 ctdet, bf16, batch 32, 512x512 — one hipGraph-replayed train step + ctdet_decode."""
    from centernet_amd.decode.ctdet import ctdet_decode
    from centernet_amd.engine import TrainStep
    seed, B, size, K = 41, 32, 512, 100
    m = _model("res_18", seed, torch.bfloat16).train()
    small = synth.ctdet_batch(seed, 4, size, size)
    batch = _to_dev(_repeat(small, B))
    kept = {}
    orig = m.loss

    def loss_and_keep(outputs, target):
        r = orig(outputs, target)
        kept["out"] = outputs[-1]
        return r
    m.loss = loss_and_keep

    def decode():
        o = kept["out"]
        return ctdet_decode(o["heatmap"].detach(), o["width_height"].detach(), reg=o["regression"].detach(), K=K, return_aux=True)

    step = TrainStep(m, lr=1e-4, distributed=False, graph=True, post_forward=decode)
    l0 = float(step(batch))
    assert step.graph, "hipGraph capture fell back to eager"
    l1 = float(step(batch))
    torch.cuda.synchronize()
    assert np.isfinite(l0) and np.isfinite(l1)
    assert bool(torch.isfinite(step.opt.flat_g).all()) and float(step.opt.flat_g.abs().max()) > 0, "finite, non-trivial gradients"
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
    det, inds, clses = step.post_out
    assert det.shape == (B, K, 6) and bool(torch.isfinite(det).all())
    heat = kept["out"]["heatmap"].detach()                  # sigmoid (in place) of this step's head map
    _peak_properties(heat, det[..., 4], inds, clses, K)
    assert torch.equal(det[..., 5], clses.float())

    # 2-image sub-batch, eval mode (BN folded into the conv epilogues), against the fp32 torch-CPU oracle with the same weights
    m2 = _model("res_18", seed, torch.bfloat16).eval()
    ref = models_ref.CenterNetRef("res_18")
    rng.fill_state_dict(ref, seed)
    ref.eval()
    x2 = small[0][:2]
    with torch.no_grad():
        out = m2(x2.to(DEV))[0]
        out_ref = ref(x2)[0]
    for k in ("heatmap", "width_height", "regression"):
        _assert_bf16_close(f"C2 eval bf16 vs oracle {k}", out[k], out_ref[k])


# ------------------------------------------------------------------------------------------------ C3 (the headline configuration)
def test_c3_dla34_ctdet_bf16_bs64_512_full_size():
    """BASELINE config C3 = what bench.py times: DLA-34 ctdet (DCNv2 up path), bf16, batch 64, 512x512 — hipGraph-replayed train
    steps + ctdet_decode of the step's own head maps (forked after the forward pass, like the bench)."""
    from centernet_amd.decode.ctdet import ctdet_decode
    from centernet_amd.engine import TrainStep
    seed, B, size, K = 44, 64, 512, 100
    m = _model("dla_34", seed, torch.bfloat16).train()
    small = synth.ctdet_batch(seed, 4, size, size)
    batch = _to_dev(_repeat(small, B))
    kept = {}
    orig = m.loss

    def loss_and_keep(outputs, target):
        r = orig(outputs, target)
        kept["out"] = outputs[-1]
        return r
    m.loss = loss_and_keep

    def decode():
        o = kept["out"]
        return ctdet_decode(o["heatmap"].detach(), o["width_height"].detach(), reg=o["regression"].detach(), K=K, return_aux=True)

    step = TrainStep(m, lr=1e-4, distributed=False, graph=True, post_forward=decode)
    l0 = float(step(batch))
    assert step.graph, "hipGraph capture fell back to eager"
    l1 = float(step(batch))
    torch.cuda.synchronize()
    assert np.isfinite(l0) and np.isfinite(l1)
    assert bool(torch.isfinite(step.opt.flat_g).all()) and float(step.opt.flat_g.abs().max()) > 0, "finite, non-trivial gradients"
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
    dcn_g = [p.grad for n_, p in m.named_parameters() if "conv_offset_mask.weight" in n_]
    assert len(dcn_g) == 16 and all(float(g_.abs().max()) > 0 for g_ in dcn_g), "every DCN offset conv received a gradient"
    det, inds, clses = step.post_out
    assert det.shape == (B, K, 6) and bool(torch.isfinite(det).all())
    heat = kept["out"]["heatmap"].detach()                  # sigmoid (in place) of this step's head map
    _peak_properties(heat, det[..., 4], inds, clses, K)
    assert torch.equal(det[..., 5], clses.float())
    # the decode of the full batch == the oracle's decode of the same head maps, on the first two images (bit-exact rule)
    o = {k: v.detach()[:2].cpu() for k, v in kept["out"].items()}
    det_ref = ops_ref.ctdet_decode(o["heatmap"], o["width_height"], o["regression"], K=K)
    assert torch.equal(det[:2].cpu(), det_ref), "ctdet_decode bit-exact against the oracle on this step's head maps"

    # 2-image sub-batch, eval mode, against the fp32 torch-CPU oracle with the same weights (sampling offsets are non-zero:
    # rng.fill_state_dict gives conv_offset_mask small random weights)
    m2 = _model("dla_34", seed, torch.bfloat16).eval()
    ref = models_ref.CenterNetRef("dla_34")
    rng.fill_state_dict(ref, seed)
    ref.eval()
    x2 = small[0][:2]
    with torch.no_grad():
        out = m2(x2.to(DEV))[0]
        out_ref = ref(x2)[0]
    for k in ("heatmap", "width_height", "regression"):
        # 12 % worst element here (6 % for C2 / C5; rms stays at 2 %): measured on this seed 6.6 % / 0.65 % rms on the heat map and
        # 9.8 % / 1.4 % rms on the size map, identical with the VALU-blend and the matrix-core-blend DCN kernels (CN_DISABLE_DCN_BM=1)
        _assert_bf16_close(f"C3 eval bf16 vs oracle {k}", out[k], out_ref[k], max_tol=0.12)


# ------------------------------------------------------------------------------------------------ C5
def test_c5_dla34_multi_pose_bf16_bs32_512_full_size():
    """BASELINE config C5: DLA-34 multi_pose (hm + wh + reg + hm_hp + hp_offset + hps heads), bf16, batch 32, 512x512 — one
    hipGraph-replayed train step + multi_pose_decode."""
    from centernet_amd.decode.multi_pose import multi_pose_decode
    from centernet_amd.engine import TrainStep
    seed, B, size, K = 42, 32, 512, 100
    m = _model("dla_34", seed, torch.bfloat16, task="pose").train()
    small = synth.pose_batch(seed, 4, size, size)
    batch = _to_dev(_repeat(small, B))
    kept = {}
    orig = m.loss

    def loss_and_keep(outputs, target):
        r = orig(outputs, target)
        kept["out"] = outputs[-1]
        return r
    m.loss = loss_and_keep

    def decode():
        o = {k: v.detach() for k, v in kept["out"].items()}
        return multi_pose_decode(o["heatmap"], o["width_height"], o["keypoints"], reg=o["regression"],
                                 hm_hp=o["heatmap_keypoints"], hp_offset=o["heatmap_keypoints_offset"], K=K)

    step = TrainStep(m, lr=1e-4, distributed=False, graph=True, post_forward=decode)
    l0 = float(step(batch))
    assert step.graph, "hipGraph capture fell back to eager"
    l1 = float(step(batch))
    torch.cuda.synchronize()
    assert np.isfinite(l0) and np.isfinite(l1)
    assert bool(torch.isfinite(step.opt.flat_g).all()) and float(step.opt.flat_g.abs().max()) > 0
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
    det = step.post_out
    assert det.shape == (B, K, 57) and bool(torch.isfinite(det).all())
    heat = kept["out"]["heatmap"].detach()                  # [B,1,128,128] after the in-place sigmoid of the loss
    sc = det[..., 4]
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())
    pooled = torch.nn.functional.max_pool2d(heat, 3, 1, 1)
    peaks = torch.where(pooled == heat, heat, torch.zeros_like(heat)).reshape(B, -1)
    top = torch.topk(peaks, K, dim=1).values
    assert torch.equal(top, sc), "centre scores == the K largest 3x3 peaks of the heat map"
    assert bool((det[..., 39] == 0).all()), "single class"
    hs = det[..., 40:57]
    assert bool(((hs >= 0) & (hs <= 1)).all())
    # the decode of the full batch == the oracle's decode of the same head maps, on the first two images (bit-exact rule)
    o = {k: v.detach()[:2].cpu() for k, v in kept["out"].items()}
    det_ref = ops_ref.multi_pose_decode(o["heatmap"], o["width_height"], o["keypoints"], reg=o["regression"],
                                        hm_hp=o["heatmap_keypoints"], hp_offset=o["heatmap_keypoints_offset"], K=K)
    assert torch.equal(det[:2].cpu(), det_ref), "multi_pose_decode bit-exact against the oracle on this step's head maps"

    m2 = _model("dla_34", seed, torch.bfloat16, task="pose").eval()
    ref = models_ref.CenterNetRef("dla_34", task="pose")
    rng.fill_state_dict(ref, seed)
    ref.eval()
    x2 = small[0][:2]
    with torch.no_grad():
        out = m2(x2.to(DEV))[0]
        out_ref = ref(x2)[0]
    for k in out_ref:
        _assert_bf16_close(f"C5 eval bf16 vs oracle {k}", out[k], out_ref[k])


# ------------------------------------------------------------------------------------------------ bf16 vs the oracle
@pytest.mark.parametrize("arch,size", [("res_18", 256), ("dla_34", 256)])
def test_network_bf16_vs_oracle(arch, size):
    """Whole-network bf16 accuracy against the fp32 torch-CPU ORACLE (same weights, same inputs), training-mode BN, batch 16:
    loss within 2 % relative, every head map within 35 % (worst element) / 5 % (rms) of the scale of `_rel_range_err` (measured on
    MI355X: ResNet-18 2.7-3.9 % / 0.6-1.0 %; DLA-34 heat map 3.8 % / 1.0 %, size map 21 % / 3.8 %, offset map 28 % / 4.5 % — the
    worst of 2 x 16 x 64 x 64 elements of maps whose last conv is sigma = 0.001-initialised; with batch 4 this bound had to be 50 %
    and the rms bound 8 %)."""
    seed, B = 43, 16
    ref = models_ref.CenterNetRef(arch)
    rng.fill_state_dict(ref, seed)
    ref.train()
    x, tgt = synth.ctdet_batch(seed, B, size, size)
    out_ref = ref(x)
    raw_ref = {k: v.detach().clone() for k, v in out_ref[0].items()}
    loss_ref, st_ref = ref.loss(out_ref, tgt)
    m = _model(arch, seed, torch.bfloat16).train()
    xg, tg = _to_dev((x, tgt))
    outs = m(xg)
    raw = {k: v.detach().clone() for k, v in outs[0].items()}
    loss, st = m.loss(outs, tg)
    loss.backward()
    assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)
    for k in raw_ref:
        # training-mode BN re-normalises every layer by the statistics of THIS batch; with batch 16 the deepest level still averages
        # over 16 x 8 x 8 = 1 024 samples per channel (batch 4 left 256: bf16 noise in a mean / variance then moved whole channels
        # of DLA-34's sigma = 0.001-initialised size / offset maps by up to 26 % of their range, which is why this test used to
        # allow 50 %).  Eval mode (C2 / C3 / C5 tests) is held to 6 % / 2 %.
        _assert_bf16_close(f"{arch} train bf16 vs oracle {k}", raw[k], raw_ref[k], max_tol=0.35, rms_tol=5e-2)
    for k in ("loss", "hm_loss", "wh_loss", "off_loss"):
        rel = abs(float(st[k]) - float(st_ref[k])) / abs(float(st_ref[k]))
        print(f"{arch} train bf16 vs oracle {k}: {float(st[k]):.5f} vs {float(st_ref[k]):.5f} (rel {rel:.2e})")
        assert rel < (2e-2 if k in ("loss", "hm_loss") else 5e-2), (k, rel)


@pytest.mark.parametrize("which", ["dla_level3_tree", "ida_up_dcn", "res_layer"])
def test_bf16_gradient_paths_of_subnetworks(which):
    """A gradient PATH lost (or doubled) in the bf16-only plumbing — epilogue statistics, slab-form weight gradients, the Root conv
    over a concatenation, skip gradients folded into data-gradient epilogues — shows as a relative L2 error of ~1 on the tensors
    behind it.  The fp32 goldens cannot see bf16-only code, and on the WHOLE network bf16 gradients of a random-weight net are
    dominated by cancellation noise (measured: median relative L2 0.5 on ResNet-18, 1.1 on DLA-34 — tools/grad_modes.py), so the
    check runs on sub-networks a few layers deep: every parameter's and the input's bf16 gradient within 35 % (L2) of this
    package's own fp32-mode gradient (the mode the reference-made goldens pin), same weights and inputs, training-mode BN.
    Measured worst tensors: two ResNet blocks 9 %, IDAUp with two DCNv2 layers 12 %, DLA level-3 tree (8 convs, 2 Roots) 22 %; with
    the Root's first child's gradient dropped (the bug this test was written after) everything upstream reads ~100 %."""
    from centernet_amd import ops
    from centernet_amd.models.backbones import pose_dla_dcn as dla
    from centernet_amd.models.backbones import msra_resnet as res

    def build():
        torch.manual_seed(11)
        if which == "dla_level3_tree":       # nested trees, project, two Roots (one over four children), max-pools, residual blocks
            return dla.Tree(2, dla.BasicBlock, 64, 128, 2, level_root=True), (4, 32, 32, 64)
        if which == "ida_up_dcn":            # DeformConv (DCNv2 + BN + ReLU) -> depthwise up-conv -> merge -> DeformConv
            return dla.IDAUp(64, [64, 128], [1, 2]), None
        return torch.nn.Sequential(res.BasicBlock(64, 64), res.BasicBlock(64, 64)), (4, 16, 16, 64)

    grads = {}
    for dt in (torch.float32, torch.bfloat16):
        net, shape = build()
        rng.fill_state_dict(net, 9)
        net = net.to(DEV).train()
        g = torch.Generator().manual_seed(5)
        if which == "ida_up_dcn":
            for m in net.modules():          # non-zero sampling offsets (the reference zero-initialises conv_offset_mask)
                if hasattr(m, "conv_offset_mask"):
                    torch.nn.init.normal_(m.conv_offset_mask.weight, std=0.02)
            xs = [ops.mark_nhwc(torch.randn(4, 16, 16, 64, generator=g).to(DEV).to(dt).requires_grad_(True)),
                  ops.mark_nhwc(torch.randn(4, 8, 8, 128, generator=g).to(DEV).to(dt).requires_grad_(True))]
            layers = list(xs)
            net(layers, 0, 2)
            y, ins = layers[-1], xs
        else:
            x = ops.mark_nhwc(torch.randn(*shape, generator=g).to(DEV).to(dt).requires_grad_(True))
            y, ins = net(x), [x]
        r = torch.randn(y.shape, generator=g).to(DEV)
        (y.float() * r).sum().backward()
        grads[dt] = {**{n: p.grad.detach().double() for n, p in net.named_parameters() if p.grad is not None},
                     **{f"input{i}": t.grad.detach().double() for i, t in enumerate(ins)}}
    rel = {}
    scale = max(float(v.norm()) / v.numel() ** 0.5 for v in grads[torch.float32].values())
    for n, g32 in grads[torch.float32].items():
        assert n in grads[torch.bfloat16], f"{n}: no bf16 gradient"
        if float(g32.norm()) / g32.numel() ** 0.5 < 1e-6 * scale:     # conv biases in front of a BatchNorm: exactly zero gradient
            continue
        rel[n] = float((grads[torch.bfloat16][n] - g32).norm()) / float(g32.norm())
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:4]
    print(f"{which}: bf16 vs fp32-mode gradients, relative L2 over {len(rel)} tensors: median {np.median(list(rel.values())):.3f}, "
          f"worst {[(n, round(v, 3)) for n, v in worst]}")
    assert worst[0][1] < 0.35, worst


# ------------------------------------------------------------------------------------------------ backbone <-> head seam
class _TorchBackbone(torch.nn.Module):
    """A third-party backbone as the reference's registry expects one (models/__init__.py:6-19): plain torch, NCHW fp32 in,
    `[Tensor[B, out_channels, H/4, W/4]]` out."""

    def __init__(self, out_channels=64):
        super().__init__()
        self.out_channels = out_channels
        self.conv = torch.nn.Conv2d(3, out_channels, 3, stride=4, padding=1)

    def forward(self, x):
        return [torch.relu(self.conv(x))]


def test_reference_shaped_seam_nchw_both_ways(golden):
    """tests/test_models.py:12-39 of the reference: backbone(x)[0] is [1, C, 128, 128]; CenterHead(backbone output) gives
    [1, classes, 128, 128].  (a) this package's backbone with `nchw_out=True` returns the reference's contract and matches the
    reference-made golden feature map; (b) a plain-torch NCHW backbone composes with this package's CenterHead, values and
    gradients equal to the oracle head; (c) this package's NCHW backbone output feeds a plain-torch head."""
    from centernet_amd.models import create_model, _model_factory
    from centernet_amd.models.heads import CenterHead
    # (a) shape contract of tests/test_models.py on the 512x512 input
    net = create_model("dla_34", compute_dtype=torch.bfloat16).to(DEV).eval()      # public default = the reference contract
    with torch.no_grad():
        y = net(torch.rand(1, 3, 512, 512, device=DEV))
    assert isinstance(y, list) and y[0].shape == (1, 64, 128, 128) and y[0].dtype == torch.float32
    head = CenterHead({"heatmap": 80, "width_height": 2}, net.out_channels, 64).to(DEV)
    with torch.no_grad():
        o = head(y[0])
    assert o["heatmap"].shape == (1, 80, 128, 128) and o["width_height"].shape == (1, 2, 128, 128)
    # ... and values: the reference-made golden feature map of the DLA-34 fixture (fp32 compute)
    g = golden("dla34_eval.npz")
    seed, size = int(g["seed"]), int(g["size"])
    from centernet_amd.centernet_detection import CenterNetDetection
    m = CenterNetDetection("dla_34", compute_dtype=torch.float32)
    rng.fill_state_dict(m, seed)
    m = m.to(DEV).eval()
    m.backbone.nchw_out = True
    x, _ = synth.ctdet_batch(seed, 2, size, size)
    with torch.no_grad():
        feat = m.backbone(x.to(DEV))[0]
        outs = m(x.to(DEV))[0]                              # heads fed the NCHW fp32 map: same numbers as the NHWC route
    assert feat.shape == (2, 64, size // 4, size // 4)
    ref_s = g["feat_s"]
    assert np.abs(strided(feat).cpu().numpy() - ref_s).max() < 1e-4 * np.abs(ref_s).max() + 1e-6
    for k in ("heatmap", "width_height", "regression"):
        ref_k = g[f"{k}_s"]
        assert np.abs(strided(outs[k]).cpu().numpy() - ref_k).max() < 1e-4 * np.abs(ref_k).max() + 1e-6, k

    # (b) third-party NCHW backbone registered in the factory -> this package's head, against the oracle head (fp32)
    _model_factory["stub"] = lambda num_layers, compute_dtype, **kw: _TorchBackbone(64)
    try:
        bb = create_model("stub").to(DEV)
    finally:
        del _model_factory["stub"]
    head = CenterHead({"heatmap": 80, "width_height": 2, "regression": 2}, 64, 64, compute_dtype=torch.float32).to(DEV)
    ref_head = models_ref.CenterHead({"heatmap": 80, "width_height": 2, "regression": 2}, 64, 64)
    rng.fill_state_dict(head, 5)
    ref_head.load_state_dict({k: v.cpu() for k, v in head.state_dict().items()})
    ref_bb = _TorchBackbone(64)
    ref_bb.load_state_dict({k: v.cpu() for k, v in bb.state_dict().items()})
    xi = rng.t_uniform(5, "img", (1, 3, 512, 512))
    out = head(bb(xi.to(DEV))[0])
    out_ref = ref_head(ref_bb(xi)[0])
    w = {k: rng.t_normal(5, "w" + k, tuple(v.shape)) for k, v in out_ref.items()}
    sum((out[k] * w[k].to(DEV)).sum() for k in out).backward()
    sum((out_ref[k] * w[k]).sum() for k in out_ref).backward()
    for k in out_ref:
        assert out[k].shape == out_ref[k].shape == (1, {"heatmap": 80}.get(k, 2), 128, 128)
        assert torch.allclose(out[k].detach().cpu(), out_ref[k].detach(), rtol=1e-4, atol=1e-5), k
    gb, gr = bb.conv.weight.grad.cpu(), ref_bb.conv.weight.grad     # the gradient crosses the seam back into plain torch
    # 5e-4: the last hop is ATen's own conv weight gradient on the GPU (MIOpen picks its algorithm from what ran before in the
    # process: 3e-5 when this test runs alone, 1.5e-4 in the middle of the whole suite); the HIP head's part of the chain is held to
    # 1e-4 by the output check above and by test_head_conv_fused_relu_backward
    assert float((gb - gr).norm() / gr.norm()) < 5e-4

    # (c) this package's backbone (NCHW fp32 out) -> a plain torch head
    plain = torch.nn.Conv2d(64, 5, 1).to(DEV)
    net.train()
    z = plain(net(torch.rand(1, 3, 128, 128, device=DEV))[0])
    assert z.shape == (1, 5, 32, 32)
    z.sum().backward()
    assert net.base.level2.tree1.conv1.weight.grad is not None and bool(torch.isfinite(net.base.level2.tree1.conv1.weight.grad).all())


def test_center_head_rejects_wrong_layout():
    from centernet_amd.models.heads import CenterHead
    head = CenterHead({"heatmap": 3}, 64, 64).to(DEV)
    with pytest.raises(ValueError):
        head(torch.zeros(1, 32, 32, 64, device=DEV))         # an untagged NHWC tensor is not silently reinterpreted


# ------------------------------------------------------------------------------------------------ eval -> train -> eval
@pytest.mark.parametrize("arch,graph", [("dla_34", False), ("res_18", True), ("resdcn_18", False)])
def test_eval_after_training_uses_fresh_weights_and_statistics(arch, graph):
    """The optimizer and the training-mode BN kernel update parameters / running statistics through raw pointers, which
    `Tensor._version` does not see: the eval-mode caches (packed + BN-folded weights) must not survive a training step.
    eval -> TrainStep x2 -> eval has to equal a freshly built model loaded with the same state_dict."""
    from centernet_amd.engine import TrainStep
    seed = 44
    m = _model(arch, seed, torch.float32)
    x, tgt = synth.ctdet_batch(seed, 2, 128, 128)
    batch = _to_dev((x, tgt))
    m.eval()
    with torch.no_grad():
        before = {k: v.clone() for k, v in m(batch[0])[0].items()}     # fills the eval caches
    m.train()
    step = TrainStep(m, lr=1e-3, distributed=False, graph=graph)
    for _ in range(2):
        step(batch)
    torch.cuda.synchronize()
    m.eval()
    with torch.no_grad():
        after = {k: v.clone() for k, v in m(batch[0])[0].items()}
    fresh = _model(arch, seed + 1, torch.float32)
    fresh.load_state_dict(m.state_dict())
    fresh.eval()
    with torch.no_grad():
        want = fresh(batch[0])[0]
    for k in want:
        assert torch.allclose(after[k], want[k], rtol=1e-5, atol=1e-6), f"{k}: stale eval cache after training"
        assert not torch.allclose(after[k], before[k], rtol=1e-3, atol=1e-4), f"{k}: training did not change the output?"


def test_failed_forward_leaves_the_bn_statistics_sinks_clean(monkeypatch):
    """The statistics sinks of the conv + BN fusion are persistent and must be all-zero when armed.  A forward pass that dies
    between a producer (which filled a sink) and its BatchNorm must not poison the next step: TrainStep clears the sinks on the
    failure path, and the step after the failure equals the step of an untouched model."""
    from centernet_amd import ops
    from centernet_amd.engine import TrainStep
    seed, B, size = 53, 4, 128
    batch = _to_dev(synth.ctdet_batch(seed, B, size, size))
    m = _model("res_18", seed, torch.bfloat16).train()
    step = TrainStep(m, lr=0.0, distributed=False, graph=False)
    real, calls = ops.batch_norm_act, {"n": 0}

    def dying(x, bn, residual=None, relu=True):
        calls["n"] += 1
        if calls["n"] == 4:
            assert getattr(x, "_bn_part", None) is not None, "the 4th BN's producer is expected to have filled a sink"
            raise RuntimeError("injected failure between a producer and its BatchNorm")
        return real(x, bn, residual, relu)

    monkeypatch.setattr(ops, "batch_norm_act", dying)
    with pytest.raises(RuntimeError, match="injected failure"):
        step(batch)
    monkeypatch.setattr(ops, "batch_norm_act", real)
    torch.cuda.synchronize()
    sinks = [b for ring in ops.BnStats._rings.values() for b in ring]
    assert sinks and all(float(b.abs().max()) == 0.0 for b in sinks), "sinks cleared on the failure path"
    for bn in (mod for mod in m.modules() if hasattr(mod, "_pending")):      # running statistics touched by the dead step: start clean
        bn.reset_running_stats()
    loss_after = float(step(batch))
    m2 = _model("res_18", seed, torch.bfloat16).train()
    loss_ref = float(TrainStep(m2, lr=0.0, distributed=False, graph=False)(batch))
    assert loss_after == pytest.approx(loss_ref, rel=1e-5), (loss_after, loss_ref)


@pytest.mark.gpu
def test_mixed_precision_prefix_tightens_the_training_mode_maps():
    """round-4 VERDICT item 5: DLA.fp32_levels = 3 (fp32 compute and storage for base_layer .. level2, bf16 above) against bf16
    throughout, both against this package's fp32 mode.  profiles/r05_mixed_precision.txt has the full-size numbers (7-10 % worst /
    0.9-1.5 % rms at +33 ms per step: measured, not adopted); here the selectable mode is held to 'at least 1.5x tighter in rms'
    on the regression maps of a small batch (tools/mixed_precision.py 8 256: 4.1 -> 1.5 %, 4.9 -> 1.9 %).  The heatmap of a
    randomly initialised head is logits of -2.19 +- a few hundredths stored in bf16 (step 2^-7): at this size its error is that
    output rounding whatever the backbone computes in (12.4 % rms of a tiny range with and without the prefix), so it is only held
    to 'not worse'."""
    from centernet_amd.centernet_detection import CenterNetDetection
    x = synth.ctdet_batch(77, 8, 256, 256)[0].cuda()

    def maps(dt, lv):
        m = CenterNetDetection("dla_34", compute_dtype=dt)
        rng.fill_state_dict(m, 77)
        m = m.cuda().train()
        m.backbone.base.fp32_levels = lv
        with torch.no_grad():
            return {k: v.float().cpu() for k, v in m(x)[0].items()}
    ref, low, mix = maps(torch.float32, 0), maps(torch.bfloat16, 0), maps(torch.bfloat16, 3)
    for k in ref:
        _, r_low = _rel_range_err(low[k], ref[k])
        _, r_mix = _rel_range_err(mix[k], ref[k])
        assert r_mix < (r_low * 1.02 if k == "heatmap" else r_low / 1.5), (k, r_low, r_mix)
