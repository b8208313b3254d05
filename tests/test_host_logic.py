"""Host-side mirror of the reference interface, on CPU: registry/config surface, module state,
the small torch-level helpers, against the goldens recorded from the reference."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_calib
from dhd_amd import synthetic as syn


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_registry_builds_reference_type_names():
    import dhd_amd
    for name in ('MGHS', 'MGHS_Depth', 'MGHS_Stereo', 'SFA'):
        assert name in dhd_amd.NECKS
    m = dhd_amd.build_neck(dict(type='SFA', in_channels=512, out_channels=256))
    assert sum(p.numel() for p in m.parameters()) == 1469728  # measured on the reference's mix.py
    with pytest.raises(KeyError):
        dhd_amd.build_neck(dict(type='LSSViewTransformer'))
    assert dhd_amd.build_neck(None) is None


def test_mghs_constructor_state_matches_reference():
    import dhd_amd
    cfg = syn.dhd_s_config()
    m = dhd_amd.build_neck(dict(cfg, type='MGHS'))
    g0 = golden('g0_frustum')
    assert m.D == 44 and m.H == 65 and tuple(m.frustum.shape) == (44, 16, 44, 3)
    assert np.array_equal(m.frustum[0, 0, :, 0].numpy(), g0['dhds_u'])
    assert np.array_equal(m.frustum[0, :, 0, 1].numpy(), g0['dhds_v'])
    assert np.array_equal(m.frustum[:, 0, 0, 2].numpy(), g0['dhds_d'])
    assert m.grid_size.tolist() == [200.0, 200.0, 1.0] and m.grid_lower_bound.tolist() == [-40.0, -40.0, -1.0]
    keys = set(m.state_dict())
    for k in ('depth_net.weight', 'depth_net.bias', 'height_net.reduce_conv.0.weight', 'height_net.bn.running_mean',
              'height_net.depth_mlp.fc1.weight', 'height_net.depth_se.conv_reduce.weight',
              'height_net.depth_conv.0.conv1.weight', 'height_net.depth_conv.3.aspp1.atrous_conv.weight',
              'height_net.depth_conv.4.conv_offset.weight', 'height_net.depth_conv.4.weight',
              'height_net.depth_conv.5.weight'):
        assert k in keys, k
    assert tuple(m.depth_net.weight.shape) == (44 + 64, 256, 1, 1)
    assert sum(p.numel() for p in m.parameters()) == 6801909
    md = dhd_amd.build_neck(dict(cfg, type='MGHS_Depth', collapse_z=False, depthnet_cfg=dict(use_dcn=False),
                                 grid_config=dict(cfg['grid_config'], depth=[1.0, 45.0, 0.5])))
    assert md.D == 88 and 'depth_net.context_conv.weight' in md.state_dict() and md.loss_depth_weight == 3.0
    ms = dhd_amd.build_neck(dict(cfg, type='MGHS_Stereo', heightnet_cfg=dict(use_dcn=False), depthnet_cfg=dict(use_dcn=False, stereo=True)))
    assert tuple(ms.cv_frustum.shape) == (44, 64, 176, 3) and tuple(ms.frustum.shape) == (44, 16, 44, 3)
    assert 'depth_net.cost_volumn_net.0.weight' in ms.state_dict()


def test_mlp_input_and_losses_match_reference_fixtures():
    import dhd_amd
    g = golden('g4_loss')
    cfg = syn.dhd_s_config()
    cfg['input_size'] = (64, 176)
    m = dhd_amd.MGHS(**cfg)
    calib = [T(a) for a in golden_calib(g)]
    assert np.array_equal(m.get_mlp_input(*calib).numpy(), g['mlp_input'])
    gd, gh, hp = T(g['gt_depth']), T(g['gt_height']), T(g['height_prob'])
    np.testing.assert_array_equal(m.get_downsampled_gt_height(gh).numpy(), g['gt_height_onehot'])
    np.testing.assert_array_equal(m.get_downsampled_gt_depth(gd).numpy(), g['gt_depth_onehot_init'])
    assert abs(float(m.get_height_loss(gd, gh, hp)) - float(g['loss_height_init'])) < 1e-6
    # the state view_transform leaves behind (mask_3_grid: depth interval 0.5 while D stays 44)
    m._set_grid(m.mask_3_grid)
    np.testing.assert_array_equal(m.get_downsampled_gt_depth(gd).numpy(), g['gt_depth_onehot_after_forward'])
    assert abs(float(m.get_height_loss(gd, gh, hp)) - float(g['loss_height_after_forward'])) < 1e-6
    sparse = m.downsample_sparse_map(gh, 16)
    assert tuple(sparse.shape) == (1, 2, 4, 11) and float(sparse.min()) >= -1.5


def test_height_map_helpers_follow_reference_semantics():
    import dhd_amd
    from oracle import mghs_oracle as O
    cfg = syn.dhd_s_config()
    m = dhd_amd.MGHS(**cfg)
    idx = syn.height_index(9, (3, 16, 44), 65)
    hm = m.height_feature_to_height_map(T(syn.height_probs_from_index(idx, 65)), m.height_range)
    masks = m.create_mask_3(hm, *m.mask_range)
    band = np.full(idx.shape, 255, np.uint8)
    for k, mk in enumerate(masks):
        band[mk.numpy()] = k
    assert np.array_equal(band, O.band_index(idx, cfg['height_range'], cfg['mask_range']))
    with pytest.raises(ValueError):
        m.height_feature_to_height_map(torch.zeros(2, 3, 4), m.height_range)


def test_voxel_pooling_prepare_v2_mirror_on_cpu_matches_reference_lists():
    """The API-parity index routine is plain torch and must reproduce the reference's lists
    (canonical order) from the reference's own coordinates."""
    import dhd_amd
    from conftest import small_dhds_cfg
    g = golden('g2b_small_dhds')
    cfg = small_dhds_cfg()
    m = dhd_amd.MGHS(**dict(cfg, heightnet_cfg=dict(use_dcn=False)))
    coor = T(g['coor'])
    grids = [dict(x=[-40, 40, 0.4], y=[-40, 40, 0.4], z=[-1, 5.4, 6.4]), cfg['mask_1_grid'], cfg['mask_2_grid'], cfg['mask_3_grid']]
    for k, grid in enumerate(grids):
        m.create_grid_infos(**{a: grid[a] for a in 'xyz'})
        rb, rd, rf, st, ln = m.voxel_pooling_prepare_v2(coor)
        for a, n in ((rb, 'ranks_bev'), (rd, 'ranks_depth'), (rf, 'ranks_feat'), (st, 'interval_starts'), (ln, 'interval_lengths')):
            assert a.dtype == torch.int32 and np.array_equal(a.numpy(), g[n + str(k)]), (n, k)
    far = coor + 1000.0
    assert m.voxel_pooling_prepare_v2(far) == (None, None, None, None, None)


def test_heightnet_and_depthnet_run_on_cpu_with_reference_shapes():
    from dhd_amd import HeightNet, DepthNet
    torch.manual_seed(0)
    hn = HeightNet(32, 32, 65).eval()
    x, mlp = torch.randn(4, 32, 4, 11), torch.randn(2, 2, 27)
    assert tuple(hn(x, mlp).shape) == (4, 65, 4, 11)
    dn = DepthNet(32, 32, 16, 44, use_dcn=False, aspp_mid_channels=24).eval()
    assert tuple(dn(x, mlp).shape) == (4, 44 + 16, 4, 11)
    # zero-initialised offsets: the deformable conv equals a plain grouped 3x3 conv
    dcn = hn.depth_conv[4]
    y = torch.randn(2, 32, 5, 7)
    ref = torch.nn.functional.conv2d(y, dcn.weight, padding=1, groups=4)
    assert torch.allclose(dcn(y), ref, atol=1e-5)


def test_synthetic_inputs_are_bit_reproducible():
    a = syn.lift_inputs(5, 1, 2, 44, 4, 11, 8, 65)
    b = syn.lift_inputs(5, 1, 2, 44, 4, 11, 8, 65)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert float(a[0].min()) > 0 and a[2].max() < 65
    import hashlib
    assert hashlib.sha256(syn.hash_u32(7, 1000).tobytes()).hexdigest()[:16] == hashlib.sha256(syn.hash_u32(7, 1000).tobytes()).hexdigest()[:16]
    c = syn.make_calibration(3, 2, 6)
    assert c[0].shape == (2, 6, 4, 4) and np.allclose(np.linalg.det(c[0][:, :, :3, :3].astype(np.float64)), 1.0, atol=1e-5)


def test_stereo_gen_grid_matches_the_matmul_formulation():
    """DepthNet.gen_grid (reference depthnet.py:249-305) is written with broadcast multiply-adds instead of a
    (B,N,D,H,W,3,3) @ (...,3,1) matmul (17.8 M GEMM batches at B = 3 fault in the ROCm BLAS path): same values."""
    from dhd_amd.depthnet import DepthNet
    torch.manual_seed(0)
    B, N, D, H, W = 2, 3, 8, 6, 10
    dn = DepthNet(32, 32, 16, 8, use_dcn=False, aspp_mid_channels=16, stereo=True, bias=5.0)
    cal = [T(a) for a in syn.make_calibration(1, B, N, (64, 176))]
    metas = dict(k2s_sensor=cal[0], intrins=cal[2], post_rots=cal[3], post_trans=cal[4], frustum=torch.rand(D, H, W, 3) * 40 + 1)
    got = dn.gen_grid(metas, B, N, D, H, W, 64, 176)
    pts = metas['frustum'] - metas['post_trans'].view(B, N, 1, 1, 1, 3)
    pts = torch.inverse(metas['post_rots']).view(B, N, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1))
    pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]), 5)
    combine = metas['k2s_sensor'][:, :, :3, :3].matmul(torch.inverse(metas['intrins']))
    pts = combine.view(B, N, 1, 1, 1, 3, 3).matmul(pts) + metas['k2s_sensor'][:, :, :3, 3].reshape(B, N, 1, 1, 1, 3, 1)
    neg = pts[..., 2, 0] < 1e-3
    pts = metas['intrins'].view(B, N, 1, 1, 1, 3, 3).matmul(pts)
    pts = pts[..., :2, :] / pts[..., 2:3, :]
    pts = metas['post_rots'][..., :2, :2].view(B, N, 1, 1, 1, 2, 2).matmul(pts).squeeze(-1) + metas['post_trans'][..., :2].view(B, N, 1, 1, 1, 2)
    ref = torch.stack([(pts[..., 0] / 175.0 * 2.0 - 1.0).masked_fill(neg, -2), (pts[..., 1] / 63.0 * 2.0 - 1.0).masked_fill(neg, -2)], -1)
    assert got.shape == (B * N, D * H, W, 2)
    assert torch.allclose(got, ref.view(B * N, D * H, W, 2), rtol=1e-5, atol=1e-4)


def test_stereo_grid_and_cost_volume_match_reference_fixture():
    """DepthNet.gen_grid and the PyTorch formulation of calculate_cost_volumn against golden G8, recorded from the
    reference's own DepthNet methods (models/model_utils/depthnet.py:249-361)."""
    from dhd_amd.depthnet import DepthNet
    g = golden('g8_stereo')
    d, h, w = g['frustum'].shape[:3]
    bn, c = g['curr'].shape[:2]
    dn = DepthNet(32, 32, 16, d, use_dcn=False, aspp_mid_channels=16, stereo=True, bias=float(g['bias']))
    metas = dict(k2s_sensor=T(g['k2s_sensor']), intrins=T(g['intrins']), post_rots=T(g['post_rots']), post_trans=T(g['post_trans']),
                 frustum=T(g['frustum']), cv_feat_list=[T(g['prev']), T(g['curr'])])
    grid = dn.gen_grid(metas, 1, bn, d, h, w, h * 4, w * 4)
    np.testing.assert_allclose(grid.numpy(), g['grid'], rtol=1e-5, atol=1e-5)
    cv = dn.calculate_cost_volumn(metas)
    np.testing.assert_allclose(cv.numpy(), g['cost_volume'], rtol=1e-4, atol=1e-6)


class _Wrapped(torch.nn.Module):
    """Stand-in for the runner's MMDataParallel: `.module` is the detector."""

    def __init__(self, module):
        super().__init__()
        self.module = module


class _Runner:
    def __init__(self, model, work_dir=None):
        import logging
        self.model, self.epoch, self.work_dir, self.rank = _Wrapped(model), 0, work_dir, 0
        self.logger = logging.getLogger('test')


def ema_fixture_net(g, device='cpu'):
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.BatchNorm2d(5), torch.nn.Linear(7, 3))
    net.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('init.')})
    return net.to(device)


def ema_fixture_step(net, it):
    """The model drift golden G9 was recorded with (tests/golden/make_golden.py)."""
    with torch.no_grad():
        j = 0
        for v in net.state_dict().values():
            if v.dtype.is_floating_point:
                v.add_(torch.from_numpy(np.float32(0.05) * syn.hash_signed(950 + 10 * it + j, tuple(v.shape))).to(v.device))
                j += 1
            else:
                v.add_(1)


def test_ema_hook_on_cpu_matches_reference_fixture(tmp_path):
    import dhd_amd
    g = golden('g9_ema')
    runner = _Runner(ema_fixture_net(g), str(tmp_path))
    hook = dhd_amd.build_hook(dict(type='MEGVIIEMAHook', init_updates=10560, priority='NORMAL'))   # DHD-S.py:272-278
    hook.before_run(runner)
    assert runner.ema_model.updates == 10560 and not runner.ema_model.ema.training
    assert all(not p.requires_grad for p in runner.ema_model.ema.parameters())
    for it in range(3):
        ema_fixture_step(runner.model.module, it)
        hook.after_train_iter(runner)
        for k, v in runner.ema_model.ema.state_dict().items():
            assert np.array_equal(v.numpy(), g[f'ema{it}.{k}']), (it, k)
    assert runner.ema_model.updates == int(g['updates'])
    hook.after_train_epoch(runner)
    cpt = torch.load(tmp_path / 'epoch_1_ema.pth')
    assert cpt['updates'] == 10563 and cpt['epoch'] == 0
    resumed = dhd_amd.MEGVIIEMAHook(resume=str(tmp_path / 'epoch_1_ema.pth'))
    resumed.before_run(runner)
    assert runner.ema_model.updates == 10563
    for k, v in runner.ema_model.ema.state_dict().items():
        assert np.array_equal(v.numpy(), g[f'ema2.{k}'])


def test_control_hooks_follow_reference_schedule():
    import dhd_amd
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    runner = _Runner(net)
    seq = dhd_amd.build_hook(dict(type='SequentialControlHook', temporal_start_epoch=1))
    seq.before_run(runner)
    assert runner.model.module.with_prev is False
    for epoch, flag in ((0, False), (1, False), (2, True)):            # strictly after the start epoch
        runner.epoch = epoch
        seq.before_train_epoch(runner)
        assert runner.model.module.with_prev is flag
    sync = dhd_amd.build_hook(dict(type='SyncbnControlHook', syncbn_start_epoch=1))
    runner.epoch = 0
    sync.before_train_epoch(runner)
    assert isinstance(runner.model.module[1], torch.nn.BatchNorm2d) and not sync.is_syncbn
    runner.epoch = 1
    sync.before_train_epoch(runner)
    assert isinstance(runner.model.module[1], torch.nn.SyncBatchNorm) and sync.is_syncbn
    with pytest.raises(KeyError):
        dhd_amd.build_hook(dict(type='NoSuchHook'))


def swin_from_fixture(g, **kw):
    from dhd_amd.swin import SwinTransformer
    net = SwinTransformer(embed_dims=16, patch_size=4, window_size=4, depths=(2, 2, 2), num_heads=(2, 4, 8), strides=(4, 2, 2),
                          out_indices=(1, 2), drop_path_rate=0.1, with_cp=False, return_stereo_feat=True, **kw).eval()
    ref_state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('state.')}
    own = net.state_dict()
    assert list(own) == list(ref_state)                                    # same keys in the same order: checkpoints load
    for k, v in own.items():
        assert v.shape == ref_state[k].shape and v.dtype == ref_state[k].dtype, k
        if 'relative_position_index' in k:
            assert torch.equal(v, ref_state[k]), k                         # the bias lookup table itself
    net.load_state_dict(ref_state)
    return net


def test_swin_backbone_matches_reference_fixture():
    """Golden G10: the reference's models/backbones/swin.py (13 x 19 tokens: window padding, odd sizes in
    PatchMerging, shifted windows in every second block), outputs, stereo feature and input gradient."""
    g = golden('g10_swin')
    net = swin_from_fixture(g)
    x = torch.from_numpy(g['x']).requires_grad_()
    outs = net(x)
    assert len(outs) == 3
    for i, o in enumerate(outs):
        ref = g[f'out{i}']
        assert o.shape == ref.shape
        assert np.abs(o.detach().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), i
    ws = [torch.from_numpy(syn.hash_signed(2000 + i, tuple(o.shape))) for i, o in enumerate(outs)]
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    assert np.abs(x.grad.numpy() - g['x_grad']).max() <= 2e-5 * np.abs(g['x_grad']).max()
    with torch.no_grad():
        s0 = net.forward_first_stage(x)
    assert np.abs(s0.numpy() - g['stage0']).max() <= 2e-5 * np.abs(g['stage0']).max()
    assert np.allclose(s0.numpy(), outs[0].detach().numpy(), atol=1e-5)    # the stereo feature is stage 0's raw output
    # gradient checkpointing (the reference default) changes nothing
    cp = swin_from_fixture(g)
    for st in cp.stages:
        st.with_cp = True
    x2 = torch.from_numpy(g['x']).requires_grad_()
    o2 = cp(x2)
    sum((o * w).sum() for o, w in zip(o2, ws)).backward()
    assert all(torch.equal(a, b) for a, b in zip(o2, outs)) and torch.allclose(x2.grad, x.grad, atol=1e-6)


def test_swin_official_checkpoint_conversion_and_builder():
    import dhd_amd
    from dhd_amd.swin import convert_official_swin
    # PatchMerging: the official order concatenates x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2]
    C = 3
    w = torch.arange(4 * C, dtype=torch.float32)[None].repeat(2, 1)         # value = official input channel index
    out = convert_official_swin({'layers.0.downsample.reduction.weight': w, 'layers.0.downsample.norm.bias': w[0],
                                 'layers.0.blocks.1.attn.qkv.weight': w, 'layers.1.blocks.0.mlp.fc1.bias': w[0],
                                 'layers.1.blocks.0.mlp.fc2.bias': w[0], 'patch_embed.proj.weight': w, 'head.weight': w, 'norm.weight': w})
    assert sorted(out) == sorted(['stages.0.downsample.reduction.weight', 'stages.0.downsample.norm.bias',
                                  'stages.0.blocks.1.attn.w_msa.qkv.weight', 'stages.1.blocks.0.ffn.layers.0.0.bias',
                                  'stages.1.blocks.0.ffn.layers.1.bias', 'patch_embed.projection.weight', 'norm.weight'])
    got = out['stages.0.downsample.reduction.weight'][0]
    # unfold channel c*4 + kh*2 + kw reads official group (kh, kw) -> kw*2 + kh, channel c
    want = torch.tensor([(kw * 2 + kh) * C + c for c in range(C) for kh in range(2) for kw in range(2)], dtype=torch.float32)
    assert torch.equal(got, want) and torch.equal(out['stages.0.downsample.norm.bias'], want)
    net = dhd_amd.build_backbone(dict(type='SwinTransformer', embed_dims=8, depths=[1, 1], num_heads=[1, 2], strides=(4, 2),
                                      out_indices=(0, 1), window_size=3, frozen_stages=2, output_missing_index_as_none=False,
                                      act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN', requires_grad=True)))
    net.init_weights()
    assert not any(p.requires_grad for p in net.patch_embed.parameters()) and not any(p.requires_grad for p in net.stages[0].parameters())
    net.train()
    assert not net.stages[0].training and net.stages[1].training
    outs = net(torch.rand(1, 3, 30, 41))
    assert [tuple(o.shape) for o in outs] == [(1, 8, 8, 11), (1, 16, 4, 6)]


def test_batchnorm2d_subclass_is_transparent_off_the_gpu():
    """dhd_amd.batchnorm.BatchNorm2d: same state-dict keys and the parent's numerics for CPU tensors / eval mode;
    SyncBatchNorm conversion (SyncbnControlHook) still recognises it."""
    from dhd_amd.batchnorm import BatchNorm2d
    torch.manual_seed(0)
    a, b = BatchNorm2d(6), torch.nn.BatchNorm2d(6)
    assert list(a.state_dict()) == list(b.state_dict()) and isinstance(a, torch.nn.BatchNorm2d)
    b.load_state_dict(a.state_dict())
    x = torch.randn(3, 6, 8, 8)
    assert not a._hip_ok(x) and not a._hip_ok(x, force=True)
    assert torch.equal(a(x), b(x)) and torch.equal(a.running_var, b.running_var)
    a.eval(); b.eval()
    assert torch.equal(a(x), b(x))
    conv = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.Sequential(BatchNorm2d(4)))
    assert isinstance(conv[0], torch.nn.SyncBatchNorm)


def test_batchnorm_relu_and_residual_arguments_off_the_gpu():
    """BatchNorm2d.forward(x, relu=..., residual=...): off the GPU (and in eval mode) the ReLU / residual add + ReLU are applied with
    torch operators after the parent's normalisation -- the same values and gradients as the reference's module code
    `relu(bn(x))` / `relu(bn(x) + identity)` (resnet.py Bottleneck.forward); the callers that use it keep their state-dict keys."""
    import copy
    from dhd_amd.batchnorm import BatchNorm2d
    from dhd_amd.detector import Bottleneck
    torch.manual_seed(1)
    for layout in (torch.contiguous_format, torch.channels_last):
        a = BatchNorm2d(8).train()
        b = copy.deepcopy(a)
        x = torch.randn(3, 8, 5, 7).contiguous(memory_format=layout).requires_grad_()
        r = torch.randn(3, 8, 5, 7).requires_grad_()
        xr, rr = x.detach().clone().requires_grad_(), r.detach().clone().requires_grad_()
        ya, yb = a(x, relu=True), torch.relu(torch.nn.BatchNorm2d.forward(b, xr))
        za, zb = a(x, residual=r), torch.relu(torch.nn.BatchNorm2d.forward(b, xr) + rr)
        assert torch.equal(ya, yb) and torch.equal(za, zb)
        (ya.sum() + (za * za).sum()).backward()
        (yb.sum() + (zb * zb).sum()).backward()
        assert torch.allclose(x.grad, xr.grad, atol=1e-6) and torch.allclose(r.grad, rr.grad, atol=1e-6)
        assert torch.equal(a.running_mean, b.running_mean) and int(a.num_batches_tracked) == int(b.num_batches_tracked) == 2
        a.eval(), b.eval()
        assert torch.equal(a(x, residual=r), torch.relu(b(xr) + rr))
    blk = Bottleneck(16, 4)
    assert [k for k, _ in blk.named_parameters()] == ['conv1.weight', 'bn1.weight', 'bn1.bias', 'conv2.weight', 'bn2.weight', 'bn2.bias',
                                                       'conv3.weight', 'bn3.weight', 'bn3.bias']
    y = blk(torch.randn(2, 16, 6, 6))
    assert y.shape == (2, 16, 6, 6) and float(y.detach().min()) >= 0.0


def test_callers_of_the_fused_batchnorm_survive_sync_batchnorm_conversion():
    """SyncbnControlHook (core/hook/syncbncontrol.py:18-32, on from epoch 0 in DHD-L.py) replaces every BatchNorm2d by
    nn.SyncBatchNorm, whose forward takes no relu / residual arguments: the callers go through batchnorm.bn_act, which applies the
    tail with torch operators for any module that is not this repo's BatchNorm2d.  Same values as before the conversion (eval
    mode: SyncBatchNorm runs on CPU tensors only there)."""
    import copy
    from dhd_amd.batchnorm import bn_act
    from dhd_amd.depthnet import BasicBlock
    from dhd_amd.detector import Bottleneck, ConvModule, _DoubleConv
    torch.manual_seed(2)
    x = torch.randn(2, 16, 6, 6)
    for m in (Bottleneck(16, 4), BasicBlock(16, 16), ConvModule(16, 8, 3, padding=1, norm=True, act=True), _DoubleConv(16, 8)):
        for bn in (b for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d)):
            bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 1.5)
        m.eval()
        ref = m(x)
        conv = torch.nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(m)).eval()
        assert any(isinstance(b, torch.nn.SyncBatchNorm) for b in conv.modules())
        assert torch.allclose(conv(x), ref, atol=1e-6)
    gn = torch.nn.GroupNorm(2, 16)
    assert torch.equal(bn_act(gn, x, relu=True), torch.relu(gn(x))) and torch.equal(bn_act(gn, x, residual=x), torch.relu(gn(x) + x))


def test_to_layout_off_the_gpu_and_its_gradient_layout():
    """dhd_amd.layout.to_layout on CPU tensors (torch's copy): values unchanged, the requested format, identity when the tensor
    already has it (or has both: one channel / one pixel), and the gradient handed back in the PRODUCER's layout."""
    from dhd_amd.layout import to_layout, _format_of
    cl, nchw = torch.channels_last, torch.contiguous_format
    x = torch.randn(2, 6, 4, 5)
    y = to_layout(x, cl)
    assert _format_of(y) == cl and torch.equal(x, y) and to_layout(y, cl) is y and to_layout(x, nchw) is x
    assert _format_of(torch.randn(2, 1, 4, 5)) == 'both' and to_layout(torch.randn(2, 6, 1, 1), cl).is_contiguous()
    assert _format_of(x[:, ::2]) is None and _format_of(torch.randn(3, 4)) is None
    assert to_layout(x[:, ::2], cl).is_contiguous(memory_format=cl)
    for src, dst in ((nchw, cl), (cl, nchw)):
        t = torch.randn(2, 6, 4, 5).contiguous(memory_format=src).requires_grad_()
        g = torch.randn(2, 6, 4, 5).contiguous(memory_format=dst)
        (to_layout(t, dst) * g).sum().backward()
        assert torch.equal(t.grad, g) and _format_of(t.grad) == src


def test_batchnorm_deferred_counters_match_the_per_layer_updates():
    """batchnorm.defer_counters / flush_counters (the detector's one-launch-per-step update of every `num_batches_tracked`): outputs,
    running statistics and the counters after a flush equal those of plain nn.BatchNorm2d behaviour -- also for a layer that is
    called twice per step -- and eval mode / momentum=None keep the parent's path."""
    import copy
    from dhd_amd.batchnorm import BatchNorm2d, defer_counters, flush_counters
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 1), BatchNorm2d(8))
    ref = copy.deepcopy(net)
    defer_counters(net)
    x = torch.randn(2, 3, 6, 6)
    for step in range(3):
        a = net(x) + net[4](net[3](net(x).relu()))       # net[4] runs three times per step, net[1] twice
        b = ref(x) + ref[4](ref[3](ref(x).relu()))
        assert torch.equal(a, b)
        assert int(net[1].num_batches_tracked) == 2 * step and int(ref[1].num_batches_tracked) == 2 * (step + 1)   # not yet flushed
        flush_counters(net)
        for m, r in ((net[1], ref[1]), (net[4], ref[4])):
            assert int(m.num_batches_tracked) == int(r.num_batches_tracked)
            assert torch.equal(m.running_mean, r.running_mean) and torch.equal(m.running_var, r.running_var)
    assert int(net[4].num_batches_tracked) == 9
    net.eval(); ref.eval()
    assert torch.equal(net(x), ref(x))
    flush_counters(net)
    assert int(net[4].num_batches_tracked) == 9
    cum = BatchNorm2d(4, momentum=None).train()           # cumulative average: the counter is an input, never deferred
    defer_counters(cum)
    cum(torch.randn(2, 4, 3, 3))
    assert int(cum.num_batches_tracked) == 1 and cum._pending == 0
    # a reader of the state (checkpoint, EMA copy, deepcopy + state_dict) between the forward and the owner's flush sees the
    # reference's counts: state_dict() flushes the layer's own pending calls, and nothing is counted twice afterwards
    net.train(); ref.train()
    net(x); ref(x)
    assert net[1]._pending == 1
    sd = net.state_dict()
    assert int(sd['1.num_batches_tracked']) == int(ref[1].num_batches_tracked) and net[1]._pending == 0
    clone = copy.deepcopy(net)
    net(x); ref(x)
    assert int(copy.deepcopy(net).state_dict()['4.num_batches_tracked']) == int(ref[4].num_batches_tracked)
    flush_counters(net)
    assert int(net[4].num_batches_tracked) == int(ref[4].num_batches_tracked) == int(clone[4].num_batches_tracked) + 1


def _g13_case(name, device='cpu'):
    """Build the mirrored HeightNet / DepthNet with golden G13's hashed parameters and inputs."""
    from dhd_amd.depthnet import DepthNet, HeightNet
    g = golden('g13_depthnet')
    net = HeightNet(32, 32, 65) if name == 'height' else DepthNet(32, 32, 8, 44)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    sd = syn.hashed_state(shapes, 1310 if name == 'height' else 1350)
    for k in sd:
        if k.endswith('conv_offset.bias'):
            sd[k] = (sd[k] * 20).astype(np.float32)
    missing = net.load_state_dict({k: T(v) for k, v in sd.items()}, strict=False)
    assert all('num_batches_tracked' in k for k in missing.missing_keys) and not missing.unexpected_keys
    x = T(syn.hash_signed(1302, (6, 32, 6, 10))).to(device).requires_grad_()
    return g, net.to(device), x, T(g['mlp_input']).to(device)


def check_g13(name, device='cpu', tol=2e-4):
    g, net, x, mlp = _g13_case(name, device)
    for mode in ('train', 'eval'):
        net.train(mode == 'train')
        for mod in net.modules():      # the fixture was recorded with ASPP's Dropout(0.5) switched off
            if isinstance(mod, torch.nn.Dropout):
                mod.eval()
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        out = net(x, mlp)
        ref = g[f'{name}.{mode}.out']
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=1e-3)
        w = T(syn.hash_signed(1303, tuple(out.shape))).to(device)
        x.grad = None
        net.zero_grad()
        (out * w).sum().backward()
        ref = g[f'{name}.{mode}.xgrad']
        np.testing.assert_allclose(x.grad.cpu().numpy(), ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=1e-3)
        if mode == 'train':
            n = 0
            for k, p in net.named_parameters():
                key = f'{name}.pgrad.{k}'
                if key in g.files:
                    ref = g[key]
                    np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2 * tol * max(1.0, np.abs(ref).max()), rtol=2e-3,
                                               err_msg=key)
                    n += 1
            assert n >= 40
        net.load_state_dict(sd0)


@pytest.mark.parametrize('name', ['height', 'depth'])
def test_heightnet_depthnet_wiring_matches_reference_fixture(name):
    """Golden G13: the reference's HeightNet / DepthNet (depthnet.py) run with stand-ins for the two un-vendored
    blocks (mmdet BasicBlock, mmcv DCN restated from the published algorithm).  The mirror on CPU (DCN through its
    grid_sample formulation): outputs, input gradient, parameter gradients, train and eval BatchNorm."""
    check_g13(name)


def test_affine_inverse_matches_torch_inverse():
    """detector._affine_inverse (the capturable closed form used while a step is recorded into a HIP graph) against
    torch.inverse on ego poses: rotation + translation of hundreds of metres, float64."""
    from dhd_amd.detector import _affine_inverse
    g = torch.Generator().manual_seed(0)
    m = torch.eye(4, dtype=torch.double).repeat(5, 1, 1, 1)
    m[..., :3, :3] = torch.linalg.qr(torch.randn(5, 1, 3, 3, dtype=torch.double, generator=g))[0]
    m[..., :3, 3] = torch.randn(5, 1, 3, dtype=torch.double, generator=g) * 500
    ref = torch.inverse(m)
    assert (ref - _affine_inverse(m)).abs().max() < 1e-11
    assert torch.equal(ref.float(), _affine_inverse(m).float()) or (ref.float() - _affine_inverse(m).float()).abs().max() < 1e-4


def test_resnet_honours_frozen_stages_norm_eval_and_warns_about_pretrained():
    from dhd_amd.detector import ResNet
    with pytest.warns(UserWarning, match='not implemented'):
        net = ResNet(depth=50, frozen_stages=1, norm_eval=True, pretrained='torchvision://resnet50')
    net.train()
    assert not net.conv1.weight.requires_grad and not net.layer1[0].conv1.weight.requires_grad
    assert net.layer2[0].conv1.weight.requires_grad
    assert not net.bn1.training and not net.layer3[0].bn1.training      # norm_eval: every BatchNorm in eval mode
    plain = ResNet(depth=50).train()
    assert plain.bn1.training and plain.conv1.weight.requires_grad


def test_cu_gemm_lds_tile_swizzle_is_bank_conflict_free():
    """dhd_amd/csrc/sfa_gemm_cu.h: the activation tile in LDS is [part][pixel p][k] bf16 with the 16-byte unit u of a pixel row
    stored at u ^ swz(p).  Simulated against the LDS rules of MI355X_MICROARCH.md (ds_read_b128: 4 groups of 16 lanes, bank =
    (addr / 4) mod 64, 4 banks per lane; ds_write_b64: 4 x 16 contiguous lanes, bank = (addr / 4) mod 32, 2 banks per lane):
    both the staging stores (lane = 8 g + q: rows 4g..4g+3, pixel 4q + e) and the fragment reads (lane = (pixel n, k half h))
    touch every bank at most once per group, for C = 256 (8 waves) and C = 128 (4 waves)."""
    def swz(p):
        return (((p >> 2) & 7) ^ (p & 1)) | (((p >> 1) & 1) << 3)

    assert all(swz(4 * q + e) == swz(4 * q) ^ swz(e) for q in range(8) for e in range(4))   # the kernel's base ^ constant form
    read_groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    read_groups += [[lane + 32 for lane in g] for g in read_groups]
    for c, waves in ((256, 8), (128, 4)):
        rowb = 2 * c
        for ks in range(c // 16):
            for grp in read_groups:
                banks = {}
                for lane in grp:
                    n, h = lane & 31, lane >> 5
                    a = n * rowb + (((2 * ks + h) ^ swz(n)) << 4)
                    for w in range(4):
                        banks[(a // 4 + w) % 64] = banks.get((a // 4 + w) % 64, 0) + 1
                assert max(banks.values()) == 1, (c, ks)
        for wv in range(waves):
            kbase = wv * 32
            for e in range(4):
                for g0 in range(0, 64, 16):
                    banks = {}
                    for lane in range(g0, g0 + 16):
                        g, q = lane >> 3, lane & 7
                        p, kq = 4 * q + e, (kbase >> 2) + g
                        a = p * rowb + (((kq >> 1) ^ swz(p)) << 4) + ((kq & 1) << 3)
                        for w in range(2):
                            banks[(a // 4 + w) % 32] = banks.get((a // 4 + w) % 32, 0) + 1
                    assert max(banks.values()) == 1, (c, wv, e)


def test_half_gemm_lds_layouts_are_bank_conflict_free():
    """dhd_amd/csrc/sfa_half.h.  (1) pw_gemm_cuh_kernel: the 64-pixel activation tile [pixel p][k] of a half type, 16-byte unit
    u of a row at u ^ swz(p), swz(p) = ((p >> 3) & 7 ^ (p & 1) << 2) | ((p >> 1) & 1) << 3: staging stores (ds_write_b64, lane =
    8 g + q: rows 4g..4g+3, pixel 8q + e) and fragment reads (ds_read_b128, lane = (pixel n, k half h), rows n and n + 32) touch
    every bank once per lane group.  (2) pw_wgrad_h_kernel: a row's eight fragments (ds_write_b128, 8 contiguous lanes) land in
    eight different 16-byte slots once unit u of fragment (ks, hh) is stored at u ^ (2 ks + hh), and the fragment reads stay
    conflict free.  Rules: MI355X_MICROARCH.md, LDS."""
    def swz(p):
        return (((p >> 3) & 7) ^ ((p & 1) << 2)) | (((p >> 1) & 1) << 3)

    assert all(swz(8 * q + e) == swz(8 * q) ^ swz(e) for q in range(8) for e in range(8)) and swz(32) == 4
    g0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
    g1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
    read_groups = [g0, g1, [x + 32 for x in g0], [x + 32 for x in g1]]
    for c, waves in ((256, 8), (128, 4)):
        rowb = 2 * c
        for ks in range(c // 16):
            for half in range(2):
                for grp in read_groups:
                    banks = {}
                    for lane in grp:
                        n, h = lane & 31, lane >> 5
                        p = n + 32 * half
                        a = p * rowb + (((2 * ks + h) ^ swz(p)) << 4)
                        for w in range(4):
                            banks[(a // 4 + w) % 64] = banks.get((a // 4 + w) % 64, 0) + 1
                    assert max(banks.values()) == 1, (c, ks, half)
        for wv in range(waves):
            kbase = wv * 32
            for e in range(8):
                for l0 in range(0, 64, 16):
                    banks = {}
                    for lane in range(l0, l0 + 16):
                        g, q = lane >> 3, lane & 7
                        p, kq = 8 * q + e, (kbase >> 2) + g
                        a = p * rowb + (((kq >> 1) ^ swz(p)) << 4) + ((kq & 1) << 3)
                        for w in range(2):
                            banks[(a // 4 + w) % 32] = banks.get((a // 4 + w) % 32, 0) + 1
                    assert max(banks.values()) == 1, (c, wv, e)
    # (2) weight-gradient staging: unit index within one (k-step, operand) image = tile * 64 + ((row & 31) + 32 hh) ^ ch8
    for ot in (256, 128):
        for wv in range(8):
            for j in range(ot // 64):
                for l0 in range(0, 64, 8):      # ds_write_b128: groups of 8 contiguous lanes, 32 banks
                    slots = set()
                    for lane in range(l0, l0 + 8):
                        row, ch8 = 64 * j + 8 * wv + (lane >> 3), lane & 7
                        ks, hh = ch8 >> 1, ch8 & 1
                        unit = (ks * 2) * (ot // 32) * 64 + (row >> 5) * 64 + (((row & 31) + 32 * hh) ^ ch8)
                        slots.add(unit % 8)
                    assert len(slots) == 8
        for ks in range(4):
            for grp in read_groups:
                slots = set()
                for lane in grp:
                    slots.add(((lane ^ (lane >> 5)) ^ (2 * ks)) % 16)
                assert len(slots) == 16
            # the read finds what the write put there: lane (r, h) of k-step ks reads fragment (row r, half h)
            for lane in range(64):
                r, h = lane & 31, lane >> 5
                assert ((lane ^ h) ^ (2 * ks)) == ((r + 32 * h) ^ (2 * ks + h))


def test_trace_ranges_wrap_the_operators_only_when_enabled(monkeypatch):
    """dhd_amd.trace: named roctx ranges (torch.cuda.nvtx) around the operators, off by default (a disabled range is one
    global check); the decorated autograd functions keep their names and signatures."""
    import torch
    from dhd_amd import trace, mghs_op, mix, bev_pool_v2 as _  # noqa: F401
    import importlib
    bp = importlib.import_module('dhd_amd.bev_pool_v2')
    calls = []
    monkeypatch.setattr(torch.cuda.nvtx, 'range_push', lambda name: calls.append(('push', name)))
    monkeypatch.setattr(torch.cuda.nvtx, 'range_pop', lambda: calls.append(('pop',)))

    @trace.traced('unit.range')
    def f(a, b=2):
        if a < 0:
            raise ValueError('neg')
        return a + b

    was = trace.enabled()
    try:
        trace.enable(False)
        assert f(1) == 3 and calls == []
        trace.enable(True)
        assert f(1, b=5) == 6 and calls == [('push', 'unit.range'), ('pop',)]
        with pytest.raises(ValueError):
            f(-1)
        assert calls[-1] == ('pop',) and len(calls) == 4          # the range is closed on the way out of an exception
    finally:
        trace.enable(was)
    assert f.__name__ == 'f'
    for fn in (mghs_op._MGHSPool.forward, mghs_op._MGHSPool.backward, mix._FusedStage.forward, mix._FusedStage.backward,
               bp._FusedPool.forward, bp.QuickCumsumCuda.backward, mghs_op.lift):
        assert hasattr(fn, '__wrapped__')
