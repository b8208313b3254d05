"""Config surface: the reference's projects/configs/DHD/*.py must load unchanged (they inherit two
base files from an un-vendored mmdetection3d checkout).  The reference tree only exists in the build
container; without it the file-based checks are skipped and the bundled-base checks still run."""
import os

import pytest

REF_CFG = '/root/reference/projects/configs/DHD'


def test_base_files_and_merge_semantics(tmp_path):
    from dhd_amd.config import Config
    base = tmp_path / 'base.py'
    base.write_text("a = dict(x=1, y=dict(p=1, q=2))\nlst = [1, 2]\n")
    child = tmp_path / 'child.py'
    child.write_text("_base_ = ['./base.py', '../../nowhere/default_runtime.py']\n"
                     "a = dict(y=dict(q=3), z=4)\nb = dict(_delete_=True, k=1)\nimport os\n")
    c = Config.fromfile(str(child))
    assert c.a == dict(x=1, y=dict(p=1, q=3), z=4) and c.a.y.q == 3 and c.lst == [1, 2]
    assert c.dist_params.backend == 'nccl' and 'os' not in c          # bundled default_runtime.py found by name
    c.merge_from_dict({'a.y.p': 7, 'lst.0': 9, 'new.key': 'v'})
    assert c.a.y.p == 7 and c.lst[0] == 9 and c.new.key == 'v'
    with pytest.raises(AttributeError):
        c.missing
    with pytest.raises(FileNotFoundError):
        bad = tmp_path / 'bad.py'
        bad.write_text("_base_ = ['./does_not_exist.py']\n")
        Config.fromfile(str(bad))


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference tree not present on this box')
@pytest.mark.parametrize('name', ['DHD-S', 'DHD-M', 'DHD-L'])
def test_reference_configs_load_unchanged(name):
    from dhd_amd.config import Config
    cfg = Config.fromfile(os.path.join(REF_CFG, name + '.py'))
    assert cfg.plugin is True and cfg.plugin_dir == 'projects/mmdet3d_plugin/'
    assert cfg.dist_params.backend == 'nccl' and cfg.optimizer.type == 'AdamW' and cfg.optimizer.lr == 2e-4
    assert cfg.data.train.type == 'NuScenesDatasetOccpancy' and cfg.runner.max_epochs == 24
    vt = cfg.model.img_view_transformer
    assert vt.type == {'DHD-S': 'MGHS', 'DHD-M': 'MGHS_Stereo', 'DHD-L': 'MGHS_Stereo'}[name]
    assert vt.mask_range == [-1.0, 0.6, 2.2, 5.4] and len(vt.height_range) == 65
    assert cfg.data.samples_per_gpu == {'DHD-S': 4, 'DHD-M': 3, 'DHD-L': 2}[name]


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference tree not present on this box')
def test_dhd_s_config_builds_the_detector_and_matches_the_packaged_copy():
    import dhd_amd
    from dhd_amd.config import Config
    from dhd_amd.detector import dhd_s_model_cfg
    cfg = Config.fromfile(os.path.join(REF_CFG, 'DHD-S.py'))
    ours = dhd_s_model_cfg()
    ref = cfg.model
    assert set(ref) == set(ours)
    for k in ours:
        a, b = ref[k], ours[k]
        if isinstance(b, dict):
            for kk in b:
                va, vb = a[kk], b[kk]
                assert (list(va) if isinstance(va, tuple) else va) == (list(vb) if isinstance(vb, tuple) else vb), (k, kk)
            assert set(a) == set(b), k
        else:
            assert a == b, k
    model = dhd_amd.build_detector(cfg.model)
    assert type(model).__name__ == 'DHD' and type(model.img_view_transformer).__name__ == 'MGHS'
    m = dhd_amd.build_neck(Config.fromfile(os.path.join(REF_CFG, 'DHD-M.py')).model.img_view_transformer)
    assert type(m).__name__ == 'MGHS_Stereo' and m.D == 88 and m.collapse_z is False


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference tree not present on this box')
def test_dhd_m_config_builds_the_temporal_stereo_detector():
    """projects/configs/DHD/DHD-M.py (DHD_stereo + MGHS_Stereo, ResNet-50, one adjacent frame + one stereo
    reference frame) builds unchanged; parameter names follow the reference's attribute names."""
    import dhd_amd
    from dhd_amd.config import Config
    cfg = Config.fromfile(os.path.join(REF_CFG, 'DHD-M.py'))
    m = dhd_amd.build_detector(cfg.model)
    assert type(m).__name__ == 'DHD_stereo' and m.num_frame == 3 and m.temporal_frame == 2 and m.extra_ref_frames == 1
    assert m.pre_process and not m.align_after_view_transfromation
    keys = set(m.state_dict())
    for k in ('pre_process_net.layers.0.0.conv1.weight', 'pre_process_net_3d.layers.0.0.conv1.weight',
              'img_view_transformer.depth_net.cost_volumn_net.0.weight', 'img_view_transformer.height_net.depth_conv.4.conv_offset.weight',
              'img_bev_encoder_backbone.inc.double_conv.0.weight', 'mix.mysk_7.spacial_leanring.0.weight', 'occ_head.predicter.2.bias'):
        assert k in keys, k
    assert m.mix.mysk_7.channels == 512 and m.img_view_transformer.D == 88
    from dhd_amd.detector import dhd_m_model_cfg
    ours, ref = dhd_m_model_cfg(), cfg.model
    norm = lambda v: list(v) if isinstance(v, tuple) else v
    assert set(ours) == set(ref)
    for k, b in ours.items():
        if isinstance(b, dict):
            assert {kk: norm(vv) for kk, vv in b.items()} == {kk: norm(ref[k][kk]) for kk in b if kk in ref[k]}, k
        else:
            assert b == ref[k], k


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference tree not present on this box')
def test_dhd_l_config_builds_with_the_swin_backbone():
    """projects/configs/DHD/DHD-L.py (Swin-B, 512 x 1408 images, FPN_LSS necks) builds unchanged."""
    import dhd_amd
    from dhd_amd.config import Config
    from dhd_amd.detector import dhd_l_model_cfg
    cfg = Config.fromfile(os.path.join(REF_CFG, 'DHD-L.py'))
    m = dhd_amd.build_detector(cfg.model)
    assert type(m).__name__ == 'DHD_stereo' and type(m.img_backbone).__name__ == 'SwinTransformer'
    assert sum(p.numel() for p in m.img_backbone.parameters()) == 86_879_608   # Swin-B, 12 x 12 windows, norms of stages 2 and 3 only
    keys = set(m.state_dict())
    for k in ('img_backbone.patch_embed.projection.weight', 'img_backbone.stages.2.blocks.17.attn.w_msa.relative_position_bias_table',
              'img_backbone.stages.2.blocks.17.ffn.layers.0.0.weight', 'img_backbone.stages.2.downsample.reduction.weight',
              'img_backbone.norm2.weight', 'img_backbone.norm3.bias', 'img_neck.conv.0.weight',
              'img_bev_encoder_backbone.layers.0.0.conv1.weight', 'img_bev_encoder_neck.up2.4.bias'):
        assert k in keys, k
    assert 'img_backbone.norm0.weight' not in keys and m.img_backbone.stages[2].blocks[17].attn.w_msa.relative_position_bias_table.shape == (529, 16)
    vt = m.img_view_transformer
    assert vt.D == 88 and tuple(vt.frustum.shape[:3]) == (88, 32, 88)
    ours, ref = dhd_l_model_cfg(), cfg.model
    norm = lambda v: list(v) if isinstance(v, tuple) else v
    assert set(ours) == set(ref)
    for k, b in ours.items():
        if isinstance(b, dict):
            assert {kk: norm(vv) for kk, vv in b.items()} == {kk: norm(vv) for kk, vv in ref[k].items()}, k
        else:
            assert b == ref[k], k
