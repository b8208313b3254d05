"""The CPU oracle against the fixtures recorded from the reference's own Python
(tests/golden/make_golden.py) and against the reference's only known-answer test."""
import numpy as np
import pytest

from conftest import golden, golden_calib, small_dhds_cfg
from dhd_amd import synthetic as syn
from oracle import mghs_oracle as O


def grids_of(cfg):
    return [O.FULL_GRID] + [{a: cfg[k][a] for a in 'xyz'} for k in ('mask_1_grid', 'mask_2_grid', 'mask_3_grid')]


def test_reference_known_answer_test():
    # ops/bev_pool_v2/bev_pool.py:163-194
    depth = np.array([0.3, 0.4, 0.2, 0.1, 0.7, 0.6, 0.8, 0.9], np.float32).reshape(1, 1, 2, 2, 2)
    feat = np.ones((1, 1, 2, 2, 2), np.float32)
    rd, rf, rb = (np.array(v, np.int32) for v in ([0, 4, 1, 6], [0, 0, 1, 2], [0, 0, 1, 1]))
    st, ln = np.array([0, 2], np.int32), np.array([2, 2], np.int32)
    out = O.bev_pool_v2(depth, feat, rd, rf, rb, (1, 1, 2, 2, 2), st, ln)
    assert out.shape == (1, 2, 1, 2, 2)
    assert abs(float(out.sum()) - 4.4) < 1e-6
    dg, fg = O.bev_pool_v2_backward(np.ones((1, 1, 2, 2, 2), np.float32), depth, feat, rd, rf, rb)
    np.testing.assert_allclose(dg.reshape(-1), [2, 2, 0, 0, 2, 0, 2, 0], atol=1e-6)
    np.testing.assert_allclose(fg.reshape(-1), [1, 1, .4, .4, .8, .8, 0, 0], atol=1e-6)


def test_frustum_axes_match_torch_linspace_arange():
    g0 = golden('g0_frustum')
    for name in ('dhds', 'smoke', 'dhdl', 'stereo'):
        cfg = g0[name + '_cfg']
        u, v, d = O.frustum_axes(list(cfg[:3]), (int(cfg[3]), int(cfg[4])), int(cfg[5]))
        for a, k in ((u, 'u'), (v, 'v'), (d, 'd')):
            assert np.array_equal(a, g0[f'{name}_{k}']), (name, k)


@pytest.mark.parametrize('name', ['g2_smoke', 'g2b_small_dhds', 'g2c_no_band', 'g2d_out_of_grid'])
def test_small_cases_bit_exact_indices_and_values(name):
    g = golden(name)
    cfg = syn.smoke_config() if name == 'g2_smoke' else small_dhds_cfg()
    calib = golden_calib(g)
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    fr = g['frustum']
    assert np.array_equal(axes[0], fr[0, 0, :, 0]) and np.array_equal(axes[1], fr[0, :, 0, 1])
    assert np.array_equal(axes[2], fr[:, 0, 0, 2])
    # per-point arithmetic, given the reference's own small matrices: bit-exact
    coor = O.ego_coor(axes, calib[0], calib[2], calib[3], calib[4], calib[5], g['ref_inv_post_rot'], g['ref_combine'])
    assert np.array_equal(coor, g['coor'])
    assert np.array_equal(O.band_index(g['height_idx'], cfg['height_range'], cfg['mask_range']), g['band'])
    for k, grid in enumerate(grids_of(cfg)):
        assert np.array_equal(O.voxel_rank(coor, grid), g[f'rank_map{k}'])
        rb, rd, rf, st, ln = O.prepare_v2(coor, grid)
        if rb is None:
            assert len(g[f'ranks_bev{k}']) == 0
            continue
        for a, n in ((rb, 'ranks_bev'), (rd, 'ranks_depth'), (rf, 'ranks_feat'), (st, 'interval_starts'),
                     (ln, 'interval_lengths')):
            assert np.array_equal(a, g[n + str(k)]), (n, k)
    outs = O.view_transform(cfg, calib, g['depth'], g['tran_feat'], g['height_idx'], g['ref_inv_post_rot'],
                            g['ref_combine'])
    for k, o in enumerate(outs):
        assert o.shape == g[f'out{k}'].shape
        np.testing.assert_allclose(o, g[f'out{k}'], atol=1e-6, rtol=0)
    ws = [syn.hash_signed(int(g['seed_w']) + k, o.shape) for k, o in enumerate(outs)]
    dg, fg = O.view_transform_backward(cfg, calib, g['depth'], g['tran_feat'], g['height_idx'], ws,
                                       g['ref_inv_post_rot'], g['ref_combine'])
    np.testing.assert_allclose(dg, g['depth_grad'], atol=5e-6, rtol=0)
    np.testing.assert_allclose(fg, g['feat_grad'], atol=5e-6, rtol=0)


def test_own_inverse_is_close_and_flips_few_points():
    """inv3x3 is the LAPACK algorithm, not MKL's kernel: matrices agree to ~1 ulp and the voxel
    map changes only for points sitting on a cell boundary."""
    g = golden('g2b_small_dhds')
    cfg = small_dhds_cfg()
    calib = golden_calib(g)
    ipr, comb = O.camera_matrices(calib[0], calib[2], calib[3])
    np.testing.assert_allclose(ipr, g['ref_inv_post_rot'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(comb, g['ref_combine'], rtol=1e-5, atol=1e-7)
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    coor = O.ego_coor(axes, calib[0], calib[2], calib[3], calib[4], calib[5])
    assert np.abs(coor - g['coor']).max() < 1e-4
    for k, grid in enumerate(grids_of(cfg)):
        bad = (O.voxel_rank(coor, grid) != g[f'rank_map{k}']).sum()
        assert bad <= 3, (k, bad)


@pytest.mark.parametrize('batch', [1, 2, 4])
def test_full_dhds_size_hashes_and_samples(batch):
    import hashlib
    g = golden(f'g3_dhds_b{batch}')
    cfg = syn.dhd_s_config()
    calib = golden_calib(g)
    s_cal, s_in, s_w = (int(v) for v in g['seeds'])
    depth, feat, hidx = syn.lift_inputs(s_in, batch, 6, 44, 16, 44, 64, 65)
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    coor = O.ego_coor(axes, calib[0], calib[2], calib[3], calib[4], calib[5], g['ref_inv_post_rot'], g['ref_combine'])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(coor) == str(g['coor_sha'])
    for k, grid in enumerate(grids_of(cfg)):
        rm = O.voxel_rank(coor, grid)
        assert sha(rm) == str(g[f'rank_map_sha{k}'])
        assert int((rm >= 0).sum()) == int(g[f'n_kept{k}'])
        assert len(np.unique(rm[rm >= 0])) == int(g[f'n_intervals{k}'])
    outs = O.view_transform(cfg, calib, depth, feat, hidx, g['ref_inv_post_rot'], g['ref_combine'])
    for k, o in enumerate(outs):
        np.testing.assert_allclose(o.reshape(-1)[g[f'out_pos{k}']], g[f'out_val{k}'], atol=2e-5, rtol=1e-5)
        s = g[f'out_sum{k}']
        assert abs(o.astype(np.float64).sum() - s[0]) < 1e-3 * max(1.0, s[1]) * 1e-3
        assert int(np.count_nonzero(o)) == int(s[2])
    ws = [syn.hash_signed(s_w + k, o.shape) for k, o in enumerate(outs)]
    dg, fg = O.view_transform_backward(cfg, calib, depth, feat, hidx, ws, g['ref_inv_post_rot'], g['ref_combine'])
    np.testing.assert_allclose(dg.reshape(-1)[g['depth_grad_pos']], g['depth_grad_val'], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(fg.reshape(-1)[g['feat_grad_pos']], g['feat_grad_val'], atol=1e-4, rtol=1e-5)


def test_mlp_input_and_height_loss_labels():
    g = golden('g4_loss')
    calib = golden_calib(g)
    assert np.array_equal(O.get_mlp_input(*calib), g['mlp_input'])
    cfg = syn.dhd_s_config()
    np.testing.assert_array_equal(
        O.downsampled_gt_height(g['gt_height'], 16, cfg['height_range'], cfg['height_interval']),
        g['gt_height_onehot'])
    for phase in ('init', 'after_forward'):
        dcfg = list(g[f'depth_cfg_{phase}'])
        np.testing.assert_array_equal(O.downsampled_gt_depth(g['gt_depth'], 16, dcfg, 44),
                                      g[f'gt_depth_onehot_{phase}'])
        loss = O.height_loss(g['gt_depth'], g['gt_height'], g['height_prob'], 16, dcfg, 44, cfg['height_range'],
                             cfg['height_interval'], cfg['loss_height_weight'])
        assert abs(loss - float(g[f'loss_height_{phase}'])) < 1e-5 * max(1.0, abs(loss))
    # the quirk: after one forward the depth interval read by the loss is 0.5, not 1.0
    assert list(g['depth_cfg_init']) == [1.0, 45.0, 1.0] and list(g['depth_cfg_after_forward']) == [1.0, 45.0, 0.5]
    assert not np.array_equal(g['gt_depth_onehot_init'], g['gt_depth_onehot_after_forward'])


def test_sfa_stage_eval_mode():
    g = golden('g5_sfa')
    sd = {k[3:]: g[k] for k in g.files if k.startswith('sd.')}
    p = 'mysk_7.'
    bn = lambda i: (sd[f'{p}spacial_leanring.{i}.weight'], sd[f'{p}spacial_leanring.{i}.bias'],
                    sd[f'{p}spacial_leanring.{i}.running_mean'], sd[f'{p}spacial_leanring.{i}.running_var'])
    out = O.sfa_stage(g['x'], sd[p + 'fc.0.weight'], sd[p + 'fc.0.bias'], sd[p + 'fc.2.weight'], sd[p + 'fc.2.bias'],
                      sd[p + 'spacial_leanring.0.weight'], sd[p + 'spacial_leanring.0.bias'], bn(1),
                      sd[p + 'spacial_leanring.3.weight'], sd[p + 'spacial_leanring.3.bias'], bn(4))
    np.testing.assert_allclose(out, g['eval.stage'], atol=2e-6, rtol=1e-5)


def test_occupancy_losses_oracle_vs_reference_code():
    """oracle.occ_losses (predictor.loss restated in float64) against golden G6, recorded from the reference's
    own semkitti_loss.py; the cross-entropy part against its textbook definition."""
    from dhd_amd.detector import NUSC_CLASS_FREQUENCIES
    g = golden('g6_occ_losses')
    cw = (1 / np.log(NUSC_CLASS_FREQUENCIES + 0.001)).astype(np.float32)
    l_ce, l_sem, l_geo = O.occ_losses(g['logits'], g['labels'], g['mask_camera'], cw)
    assert abs(l_sem - float(g['sem_scal'])) < 1e-6 and abs(l_geo - float(g['geo_scal'])) < 1e-6
    z = g['logits'].astype(np.float64)
    lp = z - np.log(np.exp(z).sum(1, keepdims=True))
    t, cam = g['labels'], g['mask_camera'].astype(bool)
    keep = cam & (t != 255)
    ce = -(lp[keep, t[keep]] * cw[t[keep]]).sum() / sum((t[cam] == i).sum() * float(cw[i]) for i in range(18))
    assert abs(l_ce - ce) < 1e-9


def test_rasterise_oracle_vs_reference_code():
    """oracle.points_to_maps against golden G7 (recorded from the reference's PointToMultiViewDepthandHeight):
    identical everywhere except where the winning sort key is shared by several points (the reference's unstable
    argsort keeps an arbitrary one of them), and there the reference's value is one of the tied points."""
    g = golden('g7_rasterise')
    h, w = (int(v) for v in g['size'])
    dm, hm, mk, ties = O.points_to_maps(g['points'], h, w, 1, tuple(g['depth_range']), return_ties=True)
    assert np.array_equal(mk, g['height_mask']) and mk.sum() > 3000
    ok = ~ties
    assert np.array_equal(dm[ok], g['depth_map'][ok]) and np.array_equal(hm[ok], g['height_map'][ok])
    assert 0 < ties.sum() < 10
    p = g['points']
    for y, x in np.argwhere(ties):
        cand = p[(np.rint(p[:, 0]) == x) & (np.rint(p[:, 1]) == y)]
        assert any(c[2] == g['depth_map'][y, x] and c[3] == g['height_map'][y, x] for c in cand)


def test_ema_update_matches_reference_fixture():
    """Golden G9: the reference's ModelEMA (core/hook/ema.py) driven for three iterations from updates = 10560."""
    g = golden('g9_ema')
    keys = [k[5:] for k in g.files if k.startswith('init.')]
    ema = {k: g['init.' + k] for k in keys}
    model = {k: g['init.' + k].copy() for k in keys}
    updates = int(g['updates']) - 3
    for it in range(3):
        j = 0
        for k in keys:
            if model[k].dtype.kind == 'f':
                model[k] = (model[k] + np.float32(0.05) * syn.hash_signed(950 + 10 * it + j, model[k].shape)).astype(np.float32)
                j += 1
        updates += 1
        d = O.ema_decay(float(g['decay']), updates)
        for k in keys:
            if ema[k].dtype.kind == 'f':
                ema[k] = O.ema_update(ema[k], model[k], d)
                assert np.array_equal(ema[k], g[f'ema{it}.{k}']), (it, k)
            else:
                assert np.array_equal(g[f'ema{it}.{k}'], g['init.' + k])   # integer buffers are left alone


SFA_GRAD_KEYS = {'fc1_w': 'fc.0.weight', 'fc1_b': 'fc.0.bias', 'fc2_w': 'fc.2.weight', 'fc2_b': 'fc.2.bias',
                 'conv1_w': 'spacial_leanring.0.weight', 'conv1_b': 'spacial_leanring.0.bias',
                 'bn1_w': 'spacial_leanring.1.weight', 'bn1_b': 'spacial_leanring.1.bias',
                 'conv2_w': 'spacial_leanring.3.weight', 'conv2_b': 'spacial_leanring.3.bias',
                 'bn2_w': 'spacial_leanring.4.weight', 'bn2_b': 'spacial_leanring.4.bias'}


def g5b_inputs():
    """Inputs of golden G5b (the reference's mix.SFA(256, 128)): hashed parameters, x, loss weights."""
    from dhd_amd.mix import SFA
    g = golden('g5b_sfa_c128')
    s_sd, s_x, _ = (int(v) for v in g['seeds'])
    shapes = {k: tuple(v.shape) for k, v in SFA(256, 128).state_dict().items() if v.dtype.is_floating_point}
    sd = syn.hashed_state(shapes, s_sd)
    x = syn.hash_signed(s_x, (2, 256, 10, 16)) * np.float32(0.7) + np.float32(0.1)
    return g, sd, x


def stage_args(sd, prefix='mysk_7.'):
    bn = lambda i: tuple(sd[f'{prefix}spacial_leanring.{i}.{n}'] for n in ('weight', 'bias', 'running_mean', 'running_var'))
    return (sd[prefix + 'fc.0.weight'], sd[prefix + 'fc.0.bias'], sd[prefix + 'fc.2.weight'], sd[prefix + 'fc.2.bias'],
            sd[prefix + 'spacial_leanring.0.weight'], sd[prefix + 'spacial_leanring.0.bias'], bn(1),
            sd[prefix + 'spacial_leanring.3.weight'], sd[prefix + 'spacial_leanring.3.bias'], bn(4))


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_sfa_stage_c128_forward_and_backward_vs_reference(mode):
    """Golden G5b: the reference's channel_spatial_stage at C = 128 (the fused operator's shape), eval and train
    BatchNorm: output, input gradient and all 12 parameter gradients of the oracle's float64 restatement."""
    g, sd, x = g5b_inputs()
    wst = syn.hash_signed(5253, (2, 128, 10, 16))
    out, dx, grads = O.sfa_stage(x, *stage_args(sd), training=(mode == 'train'), out_grad=wst)
    np.testing.assert_allclose(out, g[f'{mode}.stage'], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(dx, g[f'{mode}.stage_xgrad'], atol=3e-6, rtol=1e-5)
    for k, name in SFA_GRAD_KEYS.items():
        ref = g[f'{mode}.stage_pgrad.{name}']
        np.testing.assert_allclose(grads[k], ref, atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=1e-5, err_msg=k)


def dhdm_small_cfg():
    cfg = small_dhds_cfg()
    cfg['grid_config'] = dict(cfg['grid_config'], depth=[1.0, 45.0, 0.5])
    cfg['collapse_z'] = False
    return cfg


def test_mghs_depth_view_transform_vs_reference():
    """Golden G11 (MGHS_Depth.view_transform, lss_heightmap.py:793-856): D = 88, (B,C,1,Dy,Dx) + (B,C,16,Dy,Dx) with the
    bands stacked low / mid / high along z, gradients, and the grid_config the call leaves behind."""
    g = golden('g11_mghs_depth_small')
    cfg = dhdm_small_cfg()
    calib = golden_calib(g)
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    assert len(axes[2]) == 88 and np.array_equal(axes[2], g['frustum'][:, 0, 0, 2])
    coor = O.ego_coor(axes, calib[0], calib[2], calib[3], calib[4], calib[5], g['ref_inv_post_rot'], g['ref_combine'])
    assert np.array_equal(coor, g['coor'])
    for k, grid in enumerate(grids_of(cfg)):
        assert np.array_equal(O.voxel_rank(coor, grid), g[f'rank_map{k}'])
    bev, bev_w_z = O.mghs_depth_view_transform(cfg, calib, g['depth'], g['tran_feat'], g['height_idx'],
                                               g['ref_inv_post_rot'], g['ref_combine'])
    assert bev.shape == g['out0'].shape == (2, 8, 1, 200, 200) and bev_w_z.shape == g['out1'].shape == (2, 8, 16, 200, 200)
    np.testing.assert_allclose(bev, g['out0'], atol=1e-6, rtol=0)
    np.testing.assert_allclose(bev_w_z, g['out1'], atol=1e-6, rtol=0)
    # reset to the full grid afterwards (:848-854), unlike MGHS which stays on mask_3_grid
    assert list(g['final_grid_z']) == [-1, 5.4, 6.4] and list(g['final_grid_size']) == [200, 200, 1]
    # gradients of <bev, w0> + <bev_w_z, w1>: the collapsed-layout backward with w1 cut into its three z ranges
    s_w = int(g['seed_w'])
    w0, w1 = syn.hash_signed(s_w, bev.shape), syn.hash_signed(s_w + 1, bev_w_z.shape)
    flat = lambda a: np.ascontiguousarray(a.transpose(0, 2, 1, 3, 4)).reshape(a.shape[0], -1, 200, 200)
    ws = [flat(w0), flat(w1[:, :, 0:4]), flat(w1[:, :, 4:8]), flat(w1[:, :, 8:16])]
    dg, fg = O.view_transform_backward(cfg, calib, g['depth'], g['tran_feat'], g['height_idx'], ws,
                                       g['ref_inv_post_rot'], g['ref_combine'])
    np.testing.assert_allclose(dg, g['depth_grad'], atol=5e-6, rtol=0)
    np.testing.assert_allclose(fg, g['feat_grad'], atol=5e-6, rtol=0)


def test_mghs_depth_view_transform_dhdl_size_b2_vs_reference():
    """Golden G15: the reference's MGHS_Depth.view_transform at the DHD-L geometry (6 x 512x1408 -> 32x88 maps, D = 88,
    C = 64, collapse_z=False) with B = 2 (DHD-L.py samples_per_gpu): 2.97 M frustum points.  Ego coordinates and every
    grid's point -> voxel map by SHA-256, the pooled `bev_feat` / stacked `bev_feat_w_z` by samples, sums and counts."""
    import hashlib
    g = golden('g15_mghs_depth_dhdl_b2')
    cfg = syn.dhd_s_config()
    cfg['grid_config'] = dict(cfg['grid_config'], depth=[1.0, 45.0, 0.5])
    cfg['input_size'] = (512, 1408)
    cfg['collapse_z'] = False
    calib = golden_calib(g)
    _, s_in, _ = (int(v) for v in g['seeds'])
    depth, feat, hidx = syn.lift_inputs(s_in, 2, 6, 88, 32, 88, 64, 65)
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    coor = O.ego_coor(axes, calib[0], calib[2], calib[3], calib[4], calib[5], g['ref_inv_post_rot'], g['ref_combine'])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert coor.shape == (2, 6, 88, 32, 88, 3) and sha(coor) == str(g['coor_sha'])
    for k, grid in enumerate(grids_of(cfg)):
        rm = O.voxel_rank(coor, grid)
        assert sha(rm) == str(g[f'rank_map_sha{k}'])
        assert int((rm >= 0).sum()) == int(g[f'n_kept{k}'])
    outs = O.mghs_depth_view_transform(cfg, calib, depth, feat, hidx, g['ref_inv_post_rot'], g['ref_combine'])
    assert outs[0].shape == (2, 64, 1, 200, 200) and outs[1].shape == (2, 64, 16, 200, 200)
    for k, o in enumerate(outs):
        np.testing.assert_allclose(o.reshape(-1)[g[f'out_pos{k}']], g[f'out_val{k}'], atol=3e-5, rtol=1e-5)
        s = g[f'out_sum{k}']
        assert abs(o.astype(np.float64).sum() - s[0]) < 1e-6 * s[1] + 1e-3
        assert int(np.count_nonzero(o)) == int(s[2])


def test_dcn_oracle_vs_the_grid_sample_formulation():
    """oracle/dcn_oracle.py (explicit neighbour gathers, mmcv's published algorithm) against the package's CPU
    formulation (F.grid_sample, zero padding, align_corners) in float64, offsets far outside the image included."""
    import torch
    from oracle import dcn_oracle as D
    from dhd_amd.depthnet import DCN
    m = DCN(8, 12, 3, padding=1, groups=4).double()
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(syn.hash_signed(1, (12, 2, 3, 3)).astype(np.float64)))
        m.conv_offset.weight.copy_(torch.from_numpy(0.3 * syn.hash_signed(2, (18, 8, 3, 3)).astype(np.float64)))
        m.conv_offset.bias.copy_(torch.from_numpy(4.0 * syn.hash_signed(3, (18,)).astype(np.float64)))
    x = torch.from_numpy(syn.hash_signed(4, (2, 8, 7, 9)).astype(np.float64)).requires_grad_()
    off = m.conv_offset(x).detach()
    assert off.abs().max() > 7      # taps well outside the 7 x 9 image
    holder = torch.nn.Parameter(off.clone())

    class _Fixed(torch.nn.Module):
        def forward(self, _x):
            return holder
    m.conv_offset = _Fixed()
    y = m(x)
    gy = torch.from_numpy(syn.hash_signed(5, tuple(y.shape)).astype(np.float64))
    y.backward(gy)
    xn, wn = x.detach().numpy(), m.weight.detach().numpy()
    assert np.abs(D.deform_conv2d(xn, off.numpy(), wn, 1, 1, 4) - y.detach().numpy()).max() < 1e-12
    dx, doff, dw = D.deform_conv2d_backward(gy.numpy(), xn, off.numpy(), wn, 1, 1, 4)
    assert np.abs(dx - x.grad.numpy()).max() < 1e-12
    assert np.abs(doff - holder.grad.numpy()).max() < 1e-12
    assert np.abs(dw - m.weight.grad.numpy()).max() < 1e-11


@pytest.mark.parametrize('name', ['g2_smoke', 'g2b_small_dhds', 'g2c_no_band', 'g2d_out_of_grid'])
def test_torch_cpu_twin_vs_reference_fixtures(name):
    """oracle/mghs_torch_cpu.py (the multi-threaded torch-CPU restatement that bench.py times as `cpu_baseline`)
    against the reference's own results: frustum, ego coordinates and index lists equal, outputs / gradients to rounding."""
    import torch
    from oracle import mghs_torch_cpu as TC
    g = golden(name)
    cfg = syn.smoke_config() if name == 'g2_smoke' else small_dhds_cfg()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    calib = [t(a) for a in golden_calib(g)]
    fr = TC.frustum(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    assert np.array_equal(fr.numpy(), g['frustum'])
    inv, comb = t(g['ref_inv_post_rot']), t(g['ref_combine'])
    coor = TC.get_ego_coor(fr, calib[0], calib[2], calib[3], calib[4], calib[5], inv, comb)
    assert np.array_equal(coor.numpy(), g['coor'])
    for k, grid in enumerate(grids_of(cfg)):
        rb, rd, rf, st, ln = TC.prepare_v2(coor, grid)
        if rb is None:
            assert len(g[f'ranks_bev{k}']) == 0
            continue
        order = np.lexsort((rd.numpy(), rb.numpy()))          # canonical order inside a voxel (the argsort is unstable)
        assert np.array_equal(rb.numpy()[order], g[f'ranks_bev{k}']) and np.array_equal(rd.numpy()[order], g[f'ranks_depth{k}'])
        assert np.array_equal(rf.numpy()[order], g[f'ranks_feat{k}'])
        assert np.array_equal(st.numpy(), g[f'interval_starts{k}']) and np.array_equal(ln.numpy(), g[f'interval_lengths{k}'])
    depth, feat = t(g['depth']).requires_grad_(), t(g['tran_feat']).requires_grad_()
    height = t(syn.height_probs_from_index(g['height_idx'], len(cfg['height_range'])))
    outs = TC.view_transform(cfg, fr, calib, depth, feat, height, inv, comb)
    for k, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g[f'out{k}'], atol=1e-6, rtol=0)
    loss = sum((o * t(syn.hash_signed(int(g['seed_w']) + k, tuple(o.shape)))).sum() for k, o in enumerate(outs))
    if loss.requires_grad:
        loss.backward()
        np.testing.assert_allclose(depth.grad.numpy(), g['depth_grad'], atol=5e-6, rtol=0)
        np.testing.assert_allclose(feat.grad.numpy(), g['feat_grad'], atol=5e-6, rtol=0)
    else:
        assert not g['depth_grad'].any() and not g['feat_grad'].any()


def g14_inputs(k, shape):
    """Sample k of golden G14 (same hashes as tests/golden/make_golden.py)."""
    logits = 3.0 * syn.hash_signed(1400 + k, tuple(shape) + (18,))
    gt = (syn.hash_u32(1410 + k, int(np.prod(shape))) % 18).astype(np.uint8).reshape(shape)
    gt[::7, ::5] = 255
    cam = syn.hash_uniform(1420 + k, tuple(shape)) < 0.4
    return logits.astype(np.float32), gt, cam


def test_evaluation_histogram_vs_reference_metric():
    """Golden G14: the reference's Metric_mIoU (core/evaluation/occ_metrics.py) fed three samples; oracle.occ_confusion
    (get_occ + hist_info restated) must accumulate the same 18 x 18 histogram, per-class IoU and mIoU."""
    g = golden('g14_miou')
    shape = tuple(int(v) for v in g['shape'])
    hist = np.zeros((18, 18))
    for k in range(3):
        logits, gt, cam = g14_inputs(k, shape)
        _, h, _ = O.occ_confusion(logits.reshape(-1, 18), gt.reshape(-1), cam.reshape(-1))
        hist += h
    assert np.array_equal(hist, g['hist'])
    with np.errstate(divide='ignore', invalid='ignore'):
        iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))
    np.testing.assert_allclose(iu, g['per_class_iou'], rtol=1e-12)
    assert round(float(np.nanmean(iu[:17]) * 100), 2) == float(g['miou'])
