"""The caller of the hot path: the `DHD` occupancy detector (single frame, DHD-S) and the stock dense
modules it wires together, with the reference's registry names, constructor kwargs and
state-dict keys (projects/mmdet3d_plugin/models/detectors/DHD_model.py:10-241 on top of
bevdet.py:11-78 / bevdet_occ.py:12-21).

Only MGHS and SFA run hand-written HIP kernels (dhd_amd/lss_heightmap.py, mix.py).  Everything in
this file is plain convolution / linear / loss code that runs on PyTorch-ROCm's MIOpen/hipBLASLt
(MFMA) kernels, as BASELINE.json's north_star prescribes for the dense parts.  The blocks that the
reference imports from un-vendored packages are restated from their published definitions
(numerics UNPINNED): mmdet 2.25.1 `ResNet` / `Bottleneck` / `CrossEntropyLoss`, mmcv
`ConvModule` (conv [+ norm] [+ act] with the conv under `.conv`).

  ResNet            mmdet ResNet-50, style='pytorch', out_indices (2,3)      DHD-S.py:44-54
  CustomFPN         models/necks/fpn.py:10-203
  CustomResNet      models/backbones/resnet.py:10-80
  FPN_LSS           models/necks/lss_fpn.py:11-74
  UNet              models/backbones/unet.py:6-142
  Identity          models/necks/identity.py
  predictor         models/dense_heads/occ_head.py:32-153 (+ losses/semkitti_loss.py:136-226)
  DHD               models/detectors/DHD_model.py:10-241
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .batchnorm import BatchNorm2d, bn_act
from .depthnet import BasicBlock
from .registry import BACKBONES, DETECTORS, HEADS, NECKS, build_backbone, build_head, build_neck


# ------------------------------------------------------------------ image backbone / neck

class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)  # style='pytorch'
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        # relu(bn(.)) and relu(bn(.) + identity) as one operator each (batchnorm.BatchNorm2d.forward)
        out = bn_act(self.bn1, self.conv1(x), relu=True)
        out = bn_act(self.bn2, self.conv2(out), relu=True)
        return bn_act(self.bn3, self.conv3(out), residual=identity)


# A/B switch: the configs' with_cp=True trades a second forward of the image backbone for activation memory (32 GB cards); on 288 GB it
# only costs time.  The default keeps the configs' behaviour.
_NO_CHECKPOINT = bool(__import__('os').environ.get('DHD_NO_CHECKPOINT'))


@BACKBONES.register_module()
class ResNet(nn.Module):
    """ResNet-50/101 trunk; returns the stages listed in out_indices."""
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth=50, num_stages=4, out_indices=(2, 3), frozen_stages=-1, norm_cfg=None, norm_eval=False,
                 with_cp=False, style='pytorch', pretrained=None, **_):
        super().__init__()
        blocks = self.arch[depth][:num_stages]
        self.out_indices, self.with_cp, self.norm_eval = tuple(out_indices), with_cp, norm_eval
        if not -1 <= frozen_stages <= num_stages:
            raise ValueError(f'ResNet: frozen_stages={frozen_stages} outside [-1, num_stages={num_stages}]')
        self.frozen_stages = frozen_stages
        self.pretrained = pretrained
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = 64
        self.res_layers = []
        for i, n in enumerate(blocks):
            planes, stride = 64 * 2 ** i, 1 if i == 0 else 2
            down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), BatchNorm2d(planes * 4))
            layer = [Bottleneck(inplanes, planes, stride, down)]
            inplanes = planes * 4
            layer += [Bottleneck(inplanes, planes) for _ in range(n - 1)]
            name = f'layer{i + 1}'
            setattr(self, name, nn.Sequential(*layer))
            self.res_layers.append(name)
        if pretrained:
            # mmdet's ResNet.init_weights loads `pretrained` (DHD-S.py:53: 'torchvision://resnet50').  A local file in
            # torchvision's key layout (conv1, bn1, layer1..4; its `fc` is ignored) is loaded; schemes that need a network or
            # torchvision itself cannot be served here: say so instead of silently training from random initialisation
            import os
            import warnings
            if os.path.isfile(str(pretrained)):
                from .checkpoint import load_checkpoint
                rep = load_checkpoint(self, pretrained, strict=False, quiet=True)['_load_report']
                # expected leftovers: torchvision's `fc.*` (unexpected) and `num_batches_tracked` (already filtered).  A file with
                # other key prefixes (a detector checkpoint: `img_backbone.*`) matches nothing and would leave the backbone at
                # its random initialisation without a word
                missing = [k for k in rep['missing']]
                if rep['loaded'] == 0:
                    raise RuntimeError(f'ResNet(pretrained={pretrained!r}): no entry of the file matches this backbone (keys such as '
                                       f'{rep["unexpected"][:3]}); for a detector checkpoint use load_checkpoint(..., prefix="img_backbone")')
                if missing or rep['mismatched']:
                    warnings.warn(f'ResNet(pretrained={pretrained!r}): {len(missing)} backbone entries not in the file (e.g. {missing[:3]}), '
                                  f'{len(rep["mismatched"])} with another shape; they keep their random initialisation', stacklevel=2)
            else:
                warnings.warn(f'ResNet(pretrained={pretrained!r}): checkpoint loading is not implemented for this scheme (no network, no '
                              'torchvision), weights stay at their random initialisation; pass a local file or load a state dict explicitly',
                              stacklevel=2)
        self._freeze_stages()   # as mmdet's ResNet.__init__: frozen parameters never reach an optimizer built before .train()

    def forward(self, x):
        x = self.maxpool(bn_act(self.bn1, self.conv1(x), relu=True))
        outs = []
        for i, name in enumerate(self.res_layers):
            layer = getattr(self, name)
            if self.with_cp and x.requires_grad and not _NO_CHECKPOINT:
                for blk in layer:
                    x = checkpoint(blk, x, use_reentrant=False)
            else:
                x = layer(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def _freeze_stages(self):
        """mmdet ResNet._freeze_stages: stem (frozen_stages >= 0) and the first `frozen_stages` stages in eval mode, no gradients."""
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            layer = getattr(self, f'layer{i}')
            layer.eval()
            for p in layer.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        """mmdet ResNet.train: frozen stages stay frozen; with norm_eval every BatchNorm keeps its running statistics."""
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def forward_first_stage(self, x):
        """Stem + first residual stage only: the stereo reference feature of BEVStereo4D (bevstereo4d.py:29-40)."""
        x = self.maxpool(bn_act(self.bn1, self.conv1(x), relu=True))
        return getattr(self, self.res_layers[0])(x)


class ConvModule(nn.Module):
    """conv (+ BN) (+ ReLU); the convolution lives under `.conv`, the norm under `.bn` (mmcv layout)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, norm=False, act=False, bias='auto'):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=(not norm) if bias == 'auto' else bias)
        self.bn = BatchNorm2d(cout) if norm else None
        self.act = nn.ReLU(inplace=True) if act else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            return bn_act(self.bn, x, relu=self.act is not None)
        return x if self.act is None else self.act(x)


@NECKS.register_module()
class CustomFPN(nn.Module):
    """Top-down FPN that only materialises the levels in out_ids."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, out_ids=[], **_):
        super().__init__()
        self.in_channels, self.out_ids, self.start_level = in_channels, list(out_ids), start_level
        end = len(in_channels) if end_level == -1 else end_level
        self.lateral_convs = nn.ModuleList(ConvModule(in_channels[i], out_channels, 1) for i in range(start_level, end))
        self.fpn_convs = nn.ModuleList(ConvModule(out_channels, out_channels, 3, padding=1)
                                       for i in range(start_level, end) if i in self.out_ids)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, inputs):
        lat = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
        return [self.fpn_convs[k](lat[i]) for k, i in enumerate(self.out_ids)]


# ------------------------------------------------------------------ BEV / voxel encoders

@BACKBONES.register_module()
class CustomResNet(nn.Module):
    def __init__(self, numC_input, num_layer=[2, 2, 2], num_channels=None, stride=[2, 2, 2], backbone_output_ids=None,
                 norm_cfg=None, with_cp=False, block_type='Basic'):
        super().__init__()
        assert block_type == 'Basic' and len(num_layer) == len(stride)
        num_channels = [numC_input * 2 ** (i + 1) for i in range(len(num_layer))] if num_channels is None else num_channels
        self.backbone_output_ids = range(len(num_layer)) if backbone_output_ids is None else backbone_output_ids
        layers, cur = [], numC_input
        for i, n in enumerate(num_layer):
            blocks = [BasicBlock(cur, num_channels[i], stride=stride[i],
                                 downsample=nn.Conv2d(cur, num_channels[i], 3, stride[i], 1))]
            cur = num_channels[i]
            blocks += [BasicBlock(cur, cur) for _ in range(n - 1)]
            layers.append(nn.Sequential(*blocks))
        self.layers = nn.Sequential(*layers)
        self.with_cp = with_cp

    def forward(self, x):
        feats = []
        for i, layer in enumerate(self.layers):
            x = checkpoint(layer, x, use_reentrant=False) if self.with_cp else layer(x)
            if i in self.backbone_output_ids:
                feats.append(x)
        return feats


_PLAIN_UPSAMPLE = bool(__import__('os').environ.get('DHD_PLAIN_UPSAMPLE'))   # A/B switch: leave nn.Upsample to torch


class _UpsampleBilinear(torch.autograd.Function):
    """csrc/upsample.hip: bilinear, align_corners=True, NCHW or channels_last, in the tensor's own dtype."""

    @staticmethod
    def _geom(x, size):
        from . import _lib
        nhwc = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
        n, c, h, w = x.shape
        return (_lib_dtype(x.dtype), int(nhwc), n, c, h, w, int(size[0]), int(size[1]))

    @staticmethod
    def forward(ctx, x, size):
        from . import _lib
        geom = _UpsampleBilinear._geom(x, size)
        if not geom[1]:
            x = x.contiguous()
        lib = _lib.load()
        with torch.cuda.device(x.device):
            y = torch.empty((geom[2], geom[3], geom[6], geom[7]), dtype=x.dtype, device=x.device,
                            memory_format=torch.channels_last if geom[1] else torch.contiguous_format)
            _lib.check(lib.dhd_upsample_bilinear_forward(_lib.ptr(x), *geom, _lib.ptr(y), _lib.stream_ptr(x.device)), 'dhd_upsample_bilinear_forward')
        ctx.geom = geom
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        geom = ctx.geom
        gy = gy.contiguous(memory_format=torch.channels_last if geom[1] else torch.contiguous_format)
        lib = _lib.load()
        with torch.cuda.device(gy.device):
            gx = torch.empty((geom[2], geom[3], geom[4], geom[5]), dtype=gy.dtype, device=gy.device,
                             memory_format=torch.channels_last if geom[1] else torch.contiguous_format)
            _lib.check(lib.dhd_upsample_bilinear_backward(_lib.ptr(gy), *geom, _lib.ptr(gx), _lib.stream_ptr(gy.device)), 'dhd_upsample_bilinear_backward')
        return gx, None


def _lib_dtype(dt):
    return {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}.get(dt, -1)


class Upsample(nn.Upsample):
    """nn.Upsample whose bilinear / align_corners=True case (the only one the DHD configs use: lss_fpn.py:27,43, unet.py:86) runs
    on the library's kernels (csrc/upsample.hip) on GPU tensors, in the tensor's own layout (NCHW or channels_last) and dtype.
    torch.autocast would run upsample_bilinear2d in float32 (it is on autocast's float32 list): a half input comes back as a
    float32 tensor four / sixteen times its size, the `cat` behind it promotes its other operand, and the convolution that
    follows casts everything back to half; torch's own kernels take 0.6-1.0 ms per call at the BEV encoder's sizes (NCHW forward)
    and 2.3 ms (channels_last backward).  The kernels interpolate in float32 registers and round once on the way out -- what the
    consumer's cast of the float32 result produces up to one unit in the last place (tests/test_detector.py).  DHD_PLAIN_UPSAMPLE=1
    leaves everything to torch (A/B switch)."""

    def forward(self, x):
        if _PLAIN_UPSAMPLE or not (x.is_cuda and x.dim() == 4 and self.mode == 'bilinear' and self.align_corners):
            return super().forward(x)
        from . import _lib
        h, w = x.shape[2:]
        if self.size is not None:
            size = (self.size, self.size) if isinstance(self.size, int) else tuple(self.size)
        else:
            sf = self.scale_factor if isinstance(self.scale_factor, (tuple, list)) else (self.scale_factor, self.scale_factor)
            size = (int(h * sf[0]), int(w * sf[1]))        # floor(in * scale), as F.interpolate
        geom = _UpsampleBilinear._geom(x, size)
        if geom[0] < 0 or x.data_ptr() % 16 or not _lib.load().dhd_upsample_bilinear_supported(*geom):   # (16-byte vector loads)
            if x.dtype in (torch.float16, torch.bfloat16) and torch.is_autocast_enabled():
                with torch.autocast('cuda', enabled=False):
                    return super().forward(x)
            return super().forward(x)
        return _UpsampleBilinear.apply(x, size)


@NECKS.register_module()
class FPN_LSS(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor=4, input_feature_index=(0, 2), norm_cfg=None,
                 extra_upsample=2, lateral=None, use_input_conv=False):
        super().__init__()
        self.input_feature_index = input_feature_index
        self.extra_upsample = extra_upsample is not None
        self.up = Upsample(scale_factor=scale_factor, mode='bilinear', align_corners=True)
        f = 2 if self.extra_upsample else 1
        self.conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels * f, 3, padding=1, bias=False), BatchNorm2d(out_channels * f), nn.ReLU(inplace=True),
            nn.Conv2d(out_channels * f, out_channels * f, 3, padding=1, bias=False), BatchNorm2d(out_channels * f), nn.ReLU(inplace=True))
        if self.extra_upsample:
            self.up2 = nn.Sequential(
                Upsample(scale_factor=extra_upsample, mode='bilinear', align_corners=True),
                nn.Conv2d(out_channels * f, out_channels, 3, padding=1, bias=False), BatchNorm2d(out_channels), nn.ReLU(inplace=True),
                nn.Conv2d(out_channels, out_channels, 1, padding=0))
        self.lateral = lateral is not None
        if self.lateral:
            self.lateral_conv = nn.Sequential(nn.Conv2d(lateral, lateral, 1, bias=False), BatchNorm2d(lateral), nn.ReLU(inplace=True))

    def forward(self, feats):
        x2, x1 = feats[self.input_feature_index[0]], feats[self.input_feature_index[1]]
        if self.lateral:
            x2 = self.lateral_conv(x2)
        x = self.conv(torch.cat([x2, self.up(x1)], dim=1))
        return self.up2(x) if self.extra_upsample else x


class _DoubleConv(nn.Module):
    def __init__(self, cin, cout, mid=None):
        super().__init__()
        mid = mid or cout
        self.double_conv = nn.Sequential(nn.Conv2d(cin, mid, 3, padding=1, bias=False), BatchNorm2d(mid), nn.ReLU(inplace=True),
                                         nn.Conv2d(mid, cout, 3, padding=1, bias=False), BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        c1, b1, _, c2, b2, _ = self.double_conv      # conv, BN, ReLU twice; each BN takes its ReLU along
        return bn_act(b2, c2(bn_act(b1, c1(x), relu=True)), relu=True)


class _Down(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _DoubleConv(cin, cout))

    def forward(self, x):
        return self.maxpool_conv(x)


class _Up(nn.Module):
    def __init__(self, cin, cout, bilinear):
        super().__init__()
        if bilinear:
            self.up = Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = _DoubleConv(cin, cout, cin // 2)
        else:
            self.up = nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2)
            self.conv = _DoubleConv(cin, cout)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        dy, dx = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([x2, x1], dim=1))


class _OutConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


@BACKBONES.register_module()
class UNet(nn.Module):
    """4-level U-Net at full BEV resolution: the per-band voxel encoder."""

    def __init__(self, n_channels, n_classes, bilinear=False):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        f = 2 if bilinear else 1
        self.inc = _DoubleConv(n_channels, 64)
        self.down1, self.down2, self.down3 = _Down(64, 128), _Down(128, 256), _Down(256, 512)
        self.down4 = _Down(512, 1024 // f)
        self.up1, self.up2 = _Up(1024, 512 // f, bilinear), _Up(512, 256 // f, bilinear)
        self.up3, self.up4 = _Up(256, 128 // f, bilinear), _Up(128, 64, bilinear)
        self.outc = _OutConv(64, n_classes)

    def forward(self, x):
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x = self.up1(self.down4(x4), x4)
        x = self.up2(x, x3)
        x = self.up3(x, x2)
        return self.outc(self.up4(x, x1))


@NECKS.register_module()
class Identity(nn.Module):
    def forward(self, x):
        return x


# ------------------------------------------------------------------ head and losses

NUSC_CLASS_FREQUENCIES = np.array([944004, 1897170, 152386, 2391677, 16957802, 724139, 189027, 2074468, 413451, 2384460,
                                   5916653, 175883646, 4275424, 51393615, 61411620, 105975596, 116424404, 1892500630])


def _neg_log_clamped(x):
    """binary_cross_entropy_with_logits(inverse_sigmoid(x), 1) == -log(x'), where inverse_sigmoid's two
    while-loops (semkitti_loss.py:8-16) move x into [1e-5, 1-1e-5) in one step for x in [0, 1]."""
    x = x.float()
    x = torch.where(x >= 1 - 1e-5, x - 1e-5, x)
    x = torch.where(x < 1e-5, x + 1e-5, x)
    return -torch.log(x)


def sem_scal_loss_with_mask(pred, target, camera_mask, ignore_index=255):
    """semkitti_loss.py:171-226 without the per-class host round trips: the Python `if tensor > 0`
    branches become masks (same value, no device->host syncs)."""
    with torch.autocast(device_type=pred.device.type, enabled=False):
        p = F.softmax(pred.float(), dim=1)
        valid = ((target != ignore_index) & camera_mask.bool()).float()
        n_cls = p.shape[1]
        onehot = F.one_hot(target.clamp(0, n_cls - 1), n_cls).float() * valid[:, None]  # (M, n_cls)
        pv = p * valid[:, None]
        tgt_count = onehot.sum(0)[:-1]                    # voxels of class i
        p_sum = pv.sum(0)[:-1]
        nom = (pv * onehot).sum(0)[:-1]
        n_valid = valid.sum()
        neg_count = n_valid - tgt_count
        spec_nom = ((valid[:, None] - pv) * (valid[:, None] - onehot)).sum(0)[:-1]
        present = tgt_count > 0
        lp = torch.where(p_sum > 0, _neg_log_clamped(nom / (p_sum + 1e-5)), torch.zeros_like(p_sum))
        lr = _neg_log_clamped(nom / (tgt_count + 1e-5))
        ls = torch.where(neg_count > 0, _neg_log_clamped(spec_nom / (neg_count + 1e-5)), torch.zeros_like(p_sum))
        per_class = torch.where(present, lp + lr + ls, torch.zeros_like(lp))
        return per_class.sum() / present.float().sum()


def geo_scal_loss_with_mask(pred, target, camera_mask, ignore_index=255, non_empty_idx=0):
    """semkitti_loss.py:136-169."""
    p = F.softmax(pred, dim=1)
    empty = p[:, non_empty_idx]
    valid = ((target != ignore_index) & camera_mask.bool()).float()
    nonempty_t = (target != non_empty_idx).float() * valid
    empty_t = valid - nonempty_t
    nonempty_p = (1 - empty) * valid
    inter = (nonempty_t * nonempty_p).sum()
    precision = inter / (nonempty_p.sum() + 1e-5)
    recall = inter / (nonempty_t.sum() + 1e-5)
    spec = (empty_t * empty).sum() / (empty_t.sum() + 1e-5)
    with torch.autocast(device_type=pred.device.type, enabled=False):
        return _neg_log_clamped(precision) + _neg_log_clamped(recall) + _neg_log_clamped(spec)


class CrossEntropyLoss(nn.Module):
    """mmdet-style softmax CE: per-element class weights, an element mask (`weight`) and an external
    normaliser (`avg_factor`) (models/losses/cross_entropy_loss.py:12-63,201-302)."""

    def __init__(self, use_sigmoid=False, ignore_index=255, loss_weight=1.0, class_weight=None, **_):
        super().__init__()
        assert not use_sigmoid
        self.ignore_index, self.loss_weight = ignore_index, loss_weight
        self.class_weight = None if class_weight is None else torch.as_tensor(class_weight, dtype=torch.float32)

    def forward(self, cls_score, label, weight=None, avg_factor=None):
        cw = None if self.class_weight is None else self.class_weight.to(cls_score.device)
        loss = F.cross_entropy(cls_score.float(), label, weight=cw, reduction='none', ignore_index=self.ignore_index)
        if weight is not None:
            loss = loss * weight.float()
        loss = loss.mean() if avg_factor is None else loss.sum() / avg_factor
        return self.loss_weight * loss


@HEADS.register_module()
class predictor(nn.Module):
    """FlashOcc-style channel-to-height head: 3x3 conv, then an MLP emitting Dz*num_classes per BEV cell."""

    def __init__(self, in_dim=256, out_dim=256, Dz=16, use_mask=True, weight_ce=1, weight_geo=1, weight_sem=1,
                 num_classes=18, use_predicter=True, class_balance=False, loss_occ=None):
        super().__init__()
        self.in_dim, self.out_dim, self.Dz = in_dim, out_dim, Dz
        self.final_conv = ConvModule(in_dim, out_dim if use_predicter else num_classes * Dz, 3, padding=1, bias=True)
        self.use_predicter = use_predicter
        if use_predicter:
            self.predicter = nn.Sequential(nn.Linear(out_dim, out_dim * 2), nn.Softplus(), nn.Linear(out_dim * 2, num_classes * Dz))
        self.use_mask, self.num_classes, self.class_balance = use_mask, num_classes, class_balance
        loss_cfg = dict(loss_occ or {})
        loss_cfg.pop('type', None)
        if class_balance:
            self.cls_weights = torch.from_numpy(1 / np.log(NUSC_CLASS_FREQUENCIES[:num_classes] + 0.001))
            loss_cfg['class_weight'] = self.cls_weights
        self.loss_occ = CrossEntropyLoss(**loss_cfg)
        self.weight_ce, self.weight_geo, self.weight_sem = weight_ce, weight_geo, weight_sem

    def forward(self, img_feats):
        x = self.final_conv(img_feats).permute(0, 3, 2, 1)  # (B, Dx, Dy, C)
        if self.use_predicter:
            b, dx, dy = x.shape[:3]
            x = self.predicter(x).view(b, dx, dy, self.Dz, self.num_classes)
        return x

    def loss(self, occ_pred, voxel_semantics, mask_camera):
        if not (self.use_mask and self.class_balance):
            raise NotImplementedError
        preds = occ_pred.reshape(-1, self.num_classes)
        from . import occ_loss
        if occ_loss.supported(preds):
            # one HIP operator: two streaming passes, no host round trips (csrc/occ_loss.hip)
            scale = getattr(self.loss_occ, 'loss_weight', 1.0)
            cw = getattr(self, '_cls_weights_dev', None)   # device copy made once (a per-step host-to-device copy also
            if cw is None or cw.device != preds.device:    # cannot be captured into a HIP graph)
                cw = self._cls_weights_dev = self.cls_weights.to(device=preds.device, dtype=torch.float32)
            l_ce, l_sem, l_geo = occ_loss.occ_losses(preds, voxel_semantics, mask_camera, cw,
                                                     ignore_index=self.loss_occ.ignore_index, non_empty_idx=17)
            return dict(loss_occ=self.weight_ce * scale * l_ce, loss_voxel_sem_scal=self.weight_sem * l_sem,
                        loss_voxel_geo_scal=self.weight_geo * l_geo)
        sem = voxel_semantics.long().reshape(-1)
        mask = mask_camera.to(torch.int32).reshape(-1)
        # sum_i (#valid voxels of class i) * w_i, without a Python loop of host-visible sums
        counts = torch.bincount(sem[mask.bool()], minlength=256)[:self.num_classes]
        avg = (counts.double() * self.cls_weights.to(counts.device)).sum()
        return dict(
            loss_occ=self.weight_ce * self.loss_occ(preds, sem, weight=mask, avg_factor=avg.float()),
            loss_voxel_sem_scal=self.weight_sem * sem_scal_loss_with_mask(preds, sem, mask),
            loss_voxel_geo_scal=self.weight_geo * geo_scal_loss_with_mask(preds, sem, mask, non_empty_idx=17))

    def get_occ(self, occ_pred, img_metas=None):
        from . import occ_loss
        if occ_pred.is_cuda and occ_pred.shape[-1] == occ_loss.NUM_CLASSES:
            pred, _ = occ_loss.occ_argmax_hist(occ_pred.float())  # argmax of the logits = argmax of their softmax
            return list(pred.view(occ_pred.shape[:-1]).cpu().numpy())
        return list(occ_pred.softmax(-1).argmax(-1).cpu().numpy().astype(np.uint8))


# ------------------------------------------------------------------ detector

def _affine_inverse(m):
    """Inverse of (..., 4, 4) affine matrices [[A, t], [0, 1]] by the adjugate of A: element-wise tensor ops only."""
    a = m[..., :3, :3]
    r0, r1, r2 = a[..., 0, :], a[..., 1, :], a[..., 2, :]
    c0, c1, c2 = torch.cross(r1, r2, dim=-1), torch.cross(r2, r0, dim=-1), torch.cross(r0, r1, dim=-1)
    det = (r0 * c0).sum(-1, keepdim=True)
    inv_a = torch.stack((c0, c1, c2), dim=-1) / det.unsqueeze(-1)
    t = m[..., :3, 3:4]
    top = torch.cat((inv_a, -(inv_a @ t)), dim=-1)
    bottom = torch.zeros_like(m[..., 3:4, :])
    bottom[..., 0, 3] = 1.0
    return torch.cat((top, bottom), dim=-2)


def _inverse3(m):
    """Inverse of (..., 3, 3) matrices by the adjugate (element-wise tensor ops only)."""
    r0, r1, r2 = m[..., 0, :], m[..., 1, :], m[..., 2, :]
    c0, c1, c2 = torch.cross(r1, r2, dim=-1), torch.cross(r2, r0, dim=-1), torch.cross(r0, r1, dim=-1)
    det = (r0 * c0).sum(-1, keepdim=True)
    return torch.stack((c0, c1, c2), dim=-1) / det.unsqueeze(-1)


def small_inverse(m):
    """torch.inverse, except while the current stream is being captured into a HIP graph (dhd_amd/graph.py): the solver
    library behind torch.inverse allocates and synchronises, so 3x3 matrices and affine 4x4 matrices (last row 0,0,0,1:
    every pose / calibration matrix of the detectors) are then inverted in closed form."""
    if m.is_cuda and torch.cuda.is_current_stream_capturing():
        return _inverse3(m) if m.shape[-1] == 3 else _affine_inverse(m)
    return torch.inverse(m)


@DETECTORS.register_module()
class DHD(nn.Module):
    def __init__(self, img_backbone=None, img_neck=None, img_view_transformer=None, img_bev_encoder_backbone=None,
                 img_bev_encoder_neck=None, occ_head=None, upsample=False, img_voxel_encoder0_backbone=None,
                 img_voxel_encoder0_neck=None, img_voxel_encoder1_backbone=None, img_voxel_encoder1_neck=None,
                 img_voxel_encoder2_backbone=None, img_voxel_encoder2_neck=None, mix=None, **_):
        super().__init__()
        self.img_backbone = build_backbone(img_backbone)
        self.img_neck = build_neck(img_neck)
        self.img_view_transformer = build_neck(img_view_transformer)
        self.img_bev_encoder_backbone = build_backbone(img_bev_encoder_backbone)
        self.img_bev_encoder_neck = build_neck(img_bev_encoder_neck)
        self.occ_head = build_head(occ_head)
        self.img_voxel_encoder0 = build_backbone(img_voxel_encoder0_backbone)
        self.img_voxel_neck0 = build_neck(img_voxel_encoder0_neck)
        self.img_voxel_encoder1 = build_backbone(img_voxel_encoder1_backbone)
        self.img_voxel_neck1 = build_neck(img_voxel_encoder1_neck)
        self.img_voxel_encoder2 = build_backbone(img_voxel_encoder2_backbone)
        self.img_voxel_neck2 = build_neck(img_voxel_encoder2_neck)
        self.mix = build_neck(mix)
        self.upsample = upsample
        # one launch per step for all `num_batches_tracked` counters instead of one per layer (batchnorm.BatchNorm2d.defer_counter)
        from .batchnorm import defer_counters
        defer_counters(self)
        self._layout = {}

    # sub-modules that exchange tensors directly share a layout group
    _LAYOUT_GROUPS = dict(img_neck='img_backbone', img_bev_encoder_neck='img_bev_encoder_backbone', img_voxel_neck0='img_voxel_encoder0',
                          img_voxel_neck1='img_voxel_encoder1', img_voxel_neck2='img_voxel_encoder2')
    _DENSE_PARTS = ('img_backbone', 'img_view_transformer', 'img_bev_encoder_backbone', 'img_voxel_encoder0', 'img_voxel_encoder1',
                    'img_voxel_encoder2', 'occ_head')

    def use_channels_last(self, on=True, parts=None):
        """Run dense convolution stacks in NHWC (`torch.channels_last`): their 4-D weights are converted once and the tensor at
        the stack's entry is brought to its layout (`_enter`); inside a stack every module keeps the layout it is given.
        MIOpen's fp16 / bf16 implicit-GEMM solvers on gfx950 are NHWC kernels -- with NCHW tensors every convolution is wrapped
        in transposes (15 % of the kernel time of a DHD-S fp16 step).  The custom operators (MGHS, the SFA stage, the losses)
        take and return NCHW as before.  `parts`: names out of `_DENSE_PARTS` (default: all).  Results are those of the NCHW
        model up to the convolution solvers' rounding."""
        fmt = torch.channels_last if on else torch.contiguous_format
        for part in (self._DENSE_PARTS if parts is None else parts):
            if part not in self._DENSE_PARTS:
                raise ValueError(f'{part!r} is not one of {self._DENSE_PARTS}')
            self._layout[part] = fmt
            for name in [part] + [k for k, v in self._LAYOUT_GROUPS.items() if v == part]:
                m = getattr(self, name, None)
                if m is not None:
                    m.to(memory_format=fmt)
        return self

    def _enter(self, part, t):
        """`t` in the layout of dense stack `part` (NCHW for everything that is not a stack, e.g. 'mix'): a copy only where the
        producer's layout differs, made by the library's tiled transpose, whose gradient returns in the producer's layout."""
        from .layout import to_layout
        return to_layout(t, self._layout.get(part, torch.contiguous_format))

    @property
    def with_img_neck(self):
        return self.img_neck is not None

    def image_encoder(self, img, stereo=False):
        B, N, C, H, W = img.shape
        x = self.img_backbone(self._enter('img_backbone', img.view(B * N, C, H, W)))
        stereo_feat = None
        if stereo:
            stereo_feat, x = x[0], x[1:]
        if self.with_img_neck:
            x = self.img_neck(x)
            if isinstance(x, (list, tuple)):
                x = x[0]
        x = self._enter('img_view_transformer', x)
        return x.view(B, N, *x.shape[1:]), stereo_feat

    def prepare_inputs(self, inputs):  # noqa: D401
        """sensor -> key-ego transforms in float64, then float32 (bevdet.py:60-78)."""
        assert len(inputs) == 7
        imgs, s2e, e2g, intrins, post_rots, post_trans, bda = inputs
        B, N = imgs.shape[:2]
        s2e, e2g = s2e.view(B, N, 4, 4), e2g.view(B, N, 4, 4)
        key = e2g[:, 0:1].double()
        # torch.inverse goes through the solver library (workspace allocation + synchronisation): not capturable.
        # While the step is being captured into a HIP graph (dhd_amd/graph.py) the ego pose, an affine matrix with last
        # row (0,0,0,1), is inverted in closed form instead (float64; equal to the LU inverse to ~1e-15 before the cast)
        key_inv = small_inverse(key)
        s2k = (key_inv @ e2g.double() @ s2e.double()).float()
        return [imgs, s2k, e2g, intrins, post_rots, post_trans, bda]

    @staticmethod
    def _first(x):
        return x[0] if isinstance(x, (list, tuple)) else x

    def bev_encoder(self, x):
        return self._first(self.img_bev_encoder_neck(self.img_bev_encoder_backbone(self._enter('img_bev_encoder_backbone', x))))

    def voxel_encoder0(self, x):
        return self._first(self.img_voxel_neck0(self.img_voxel_encoder0(self._enter('img_voxel_encoder0', x))))

    def voxel_encoder1(self, x):
        return self._first(self.img_voxel_neck1(self.img_voxel_encoder1(self._enter('img_voxel_encoder1', x))))

    def voxel_encoder2(self, x):
        return self._first(self.img_voxel_neck2(self.img_voxel_encoder2(self._enter('img_voxel_encoder2', x))))

    def extract_img_feat(self, img_inputs, img_metas=None, **kwargs):
        imgs, s2k, e2g, intrins, post_rots, post_trans, bda = self.prepare_inputs(img_inputs)
        x, _ = self.image_encoder(imgs)
        vt = self.img_view_transformer
        mlp_input = vt.get_mlp_input(s2k, e2g, intrins, post_rots, post_trans, bda)
        x_2d, depth, height, low, mid, high = vt([x, s2k, e2g, intrins, post_rots, post_trans, bda, mlp_input])
        x_2d, x_3d = self.encode_maps(x_2d, low, mid, high)
        return x_2d, x_3d, depth, height

    def encode_maps(self, x_2d, low, mid, high):
        """The four BEV maps of the view transform -> (x_2d, x_3d): `bev_encoder` on the collapsed map, one UNet per height band,
        the three band features concatenated (DHD_model.py:107-113)."""
        x_2d = self.bev_encoder(x_2d)
        x_3d = torch.cat((self.voxel_encoder0(low), self.voxel_encoder1(mid), self.voxel_encoder2(high)), dim=1)
        return x_2d, x_3d

    def occ_logits(self, img_feats):
        """[x_2d, x_3d] -> voxel logits (B, Dx, Dy, Dz, n_cls): cat -> mix -> occ_head (DHD_model.py:196-198, :224-226)."""
        return self.occ_head(self._enter('occ_head', self.mix(self._enter('mix', torch.cat(img_feats, dim=1)))))

    def extract_feat(self, points, img_inputs, img_metas=None, **kwargs):
        x_2d, x_3d, depth, height = self.extract_img_feat(img_inputs, img_metas, **kwargs)
        return x_2d, x_3d, None, depth, height

    def forward_occ_train(self, img_feats, voxel_semantics, mask_camera):
        return self.occ_head.loss(self.occ_logits(img_feats), voxel_semantics, mask_camera)

    def forward_train(self, points=None, img_metas=None, img_inputs=None, **kwargs):
        x_2d, x_3d, _, depth, height = self.extract_feat(points, img_inputs=img_inputs, img_metas=img_metas, **kwargs)
        losses = dict(loss_height=self.img_view_transformer.get_height_loss(kwargs['gt_depth'], kwargs['gt_height'], height))
        losses.update(self.forward_occ_train([x_2d, x_3d], kwargs['voxel_semantics'], kwargs['mask_camera']))
        return losses

    def simple_test(self, points, img_metas, img=None, rescale=False, **kwargs):
        x_2d, x_3d, _, _, _ = self.extract_feat(points, img_inputs=img, img_metas=img_metas, **kwargs)
        return self.simple_test_occ([x_2d, x_3d], img_metas)

    def simple_test_occ(self, img_feats, img_metas=None):
        return self.occ_head.get_occ(self.occ_logits(img_feats), img_metas)

    def flush_bn_counters(self):
        """Add the BatchNorm calls counted on the host since the last flush to the `num_batches_tracked` buffers (one launch).
        `forward(return_loss=True)` does it once per step; a caller that drives `forward_train` or a sub-module itself calls this
        after its step.  Reading a `state_dict()` flushes too (batchnorm.BatchNorm2d._flush_own), so checkpoints and EMA copies
        always see the reference's counts; `_pending` itself is host state that a HIP-graph replay does not advance (graph.py)."""
        from .batchnorm import flush_counters
        flush_counters(self)

    def forward(self, return_loss=True, **kwargs):
        if not return_loss:
            return self.simple_test(**kwargs)
        losses = self.forward_train(**kwargs)
        if self.training:
            self.flush_bn_counters()
        return losses


@DETECTORS.register_module()
class DHD_stereo(DHD):
    """The temporal-stereo detector of DHD-M / DHD-L (models/detectors/DHD_model.py:245-666 on top of
    bevstereo4d.py:11-140 and bevdet4d.py:21-300): the key frame and `num_adj` adjacent frames go through the view
    transformer (adjacent ones without gradient), one extra reference frame only provides the stereo feature of the
    oldest adjacent frame; per-frame BEV features are concatenated on the channel axis.  img_inputs carry all
    frames: imgs (B, N_views*N_frames, 3, H, W), view index major."""

    def __init__(self, pre_process=None, pre_process_net_3d=None, align_after_view_transfromation=False, num_adj=1,
                 with_prev=True, **kwargs):
        super().__init__(**kwargs)
        self.pre_process = pre_process is not None
        if self.pre_process:
            self.pre_process_net = build_backbone(pre_process)
            self.pre_process_net_3d = build_backbone(pre_process_net_3d)
        self.align_after_view_transfromation = align_after_view_transfromation
        self.with_prev = with_prev
        self.extra_ref_frames = 1
        self.temporal_frame = num_adj + 1
        self.num_frame = self.temporal_frame + self.extra_ref_frames
        self.grid = None

    # ---- inputs (bevdet4d.py:208-300)
    def prepare_inputs(self, img_inputs, stereo=False):
        imgs = img_inputs[0]
        B, NT, C, H, W = imgs.shape
        F_, N = self.num_frame, NT // self.num_frame
        imgs = [t.squeeze(2) for t in torch.split(imgs.view(B, N, F_, C, H, W), 1, 2)]
        s2e, e2g, intrins, post_rots, post_trans, bda = img_inputs[1:7]
        s2e, e2g = s2e.view(B, F_, N, 4, 4), e2g.view(B, F_, N, 4, 4)
        key_inv = small_inverse(e2g[:, 0, 0].double())[:, None, None]
        s2k = (key_inv @ e2g.double() @ s2e.double()).float()
        curr2adj = None
        if stereo:
            t = self.temporal_frame
            cur = e2g[:, :t].double() @ s2e[:, :t].double()
            adj = e2g[:, 1:t + 1].double() @ s2e[:, 1:t + 1].double()
            c2a = (small_inverse(adj) @ cur).float()
            curr2adj = [p.squeeze(1) for p in torch.split(c2a, 1, 1)] + [None] * self.extra_ref_frames
            assert len(curr2adj) == F_
        per_frame = [[p.squeeze(1) for p in torch.split(v, 1, 1)] for v in
                     (s2k, e2g, intrins.view(B, F_, N, 3, 3), post_rots.view(B, F_, N, 3, 3), post_trans.view(B, F_, N, 3))]
        return [imgs] + per_frame + [bda, curr2adj]

    # ---- BEV alignment of a previous frame (bevdet4d.py:43-138); unused by the shipped configs
    def gen_grid(self, inp, sensor2keyegos, bda, bda_adj=None):
        B, C, H, W = inp.shape
        if self.grid is None or self.grid.shape[:2] != (H, W) or self.grid.device != inp.device:
            xs = torch.linspace(0, W - 1, W, dtype=inp.dtype, device=inp.device).view(1, W).expand(H, W)
            ys = torch.linspace(0, H - 1, H, dtype=inp.dtype, device=inp.device).view(H, 1).expand(H, W)
            self.grid = torch.stack((xs, ys, torch.ones_like(xs)), -1)
        grid = self.grid.view(1, H, W, 3).expand(B, H, W, 3).view(B, H, W, 3, 1)
        c02l0, c12l0 = sensor2keyegos[0][:, 0:1], sensor2keyegos[1][:, 0:1]
        bda_ = torch.zeros((B, 1, 4, 4), dtype=grid.dtype, device=grid.device)
        bda_[:, :, :3, :3] = bda.unsqueeze(1)
        bda_[:, :, 3, 3] = 1
        c02l0 = bda_.matmul(c02l0)
        c12l0 = (bda_ if bda_adj is None else bda_adj).matmul(c12l0)
        l02l1 = c02l0.matmul(small_inverse(c12l0))[:, 0].view(B, 1, 1, 4, 4)
        l02l1 = l02l1[:, :, :, [True, True, False, True], :][:, :, :, :, [True, True, False, True]]
        vt = self.img_view_transformer
        key = (str(grid.device), grid.dtype, str(inp.device), inp.dtype, W, H, tuple(vt.grid_interval.tolist()),
               tuple(vt.grid_lower_bound.tolist()))   # `norm` below is built with inp's dtype / device
        if getattr(self, '_f2b_key', None) != key:   # constants of the module: built (host -> device) once, not per step
            feat2bev = torch.zeros((3, 3), dtype=grid.dtype)
            feat2bev[0, 0], feat2bev[1, 1] = vt.grid_interval[0], vt.grid_interval[1]
            feat2bev[0, 2], feat2bev[1, 2] = vt.grid_lower_bound[0], vt.grid_lower_bound[1]
            feat2bev[2, 2] = 1
            self._f2b = (feat2bev.view(1, 3, 3).to(grid.device), torch.inverse(feat2bev).view(1, 3, 3).to(grid.device),
                         torch.tensor([W - 1.0, H - 1.0], dtype=inp.dtype).to(inp.device))
            self._f2b_key = key
        feat2bev, feat2bev_inv, norm = self._f2b
        tf = feat2bev_inv.matmul(l02l1).matmul(feat2bev)
        grid = tf.matmul(grid)
        return grid[:, :, :, :2, 0] / norm.view(1, 1, 1, 2) * 2.0 - 1.0

    def shift_feature(self, inp, sensor2keyegos, bda, bda_adj=None):
        grid = self.gen_grid(inp, sensor2keyegos, bda, bda_adj=bda_adj)
        return F.grid_sample(inp, grid.to(inp.dtype), align_corners=True)

    # ---- stereo reference feature = first residual stage of the backbone (bevstereo4d.py:18-50)
    def extract_stereo_ref_feat(self, x):
        B, N, C, H, W = x.shape
        return self.img_backbone.forward_first_stage(x.view(B * N, C, H, W))

    def prepare_bev_feat(self, img, sensor2keyego, ego2global, intrin, post_rot, post_tran, bda, mlp_input, feat_prev_iv,
                         k2s_sensor, extra_ref_frame):
        if extra_ref_frame:
            return None, None, None, None, self.extract_stereo_ref_feat(img)
        x, stereo_feat = self.image_encoder(img, stereo=True)
        vt = self.img_view_transformer
        cvf = getattr(self, '_cv_frustum_dev', None)   # device copy of the stereo frustum template, made once
        if cvf is None or cvf.device != x.device or cvf.dtype != x.dtype:
            cvf = self._cv_frustum_dev = vt.cv_frustum.to(x)
        metas = dict(k2s_sensor=k2s_sensor, intrins=intrin, post_rots=post_rot, post_trans=post_tran,
                     frustum=cvf, cv_downsample=4, downsample=vt.downsample, grid_config=vt.grid_config,
                     cv_feat_list=[feat_prev_iv, stereo_feat])
        bev_2d, bev_3d, depth, height = vt([x, sensor2keyego, ego2global, intrin, post_rot, post_tran, bda, mlp_input], metas)
        if self.pre_process and bev_3d.dim() == 5:
            b2 = self.pre_process_net(torch.cat(bev_2d.unbind(dim=2), 1))[0]
            b3 = self.pre_process_net_3d(torch.cat(bev_3d.unbind(dim=2), 1))[0]
            bev_2d = torch.stack(torch.chunk(b2, 1, dim=1), dim=2)
            bev_3d = torch.stack(torch.chunk(b3, 16, dim=1), dim=2)
        return bev_2d, bev_3d, depth, height, stereo_feat

    def extract_img_feat(self, img_inputs, img_metas=None, pred_prev=False, sequential=False, **kwargs):
        if sequential or pred_prev:
            raise NotImplementedError('sequential / pred_prev inference (DHD_model.py:401-402,459-484) is not mirrored')
        imgs, s2ks, e2gs, intrins, post_rots, post_trans, bda, curr2adj = self.prepare_inputs(img_inputs, stereo=True)
        vt = self.img_view_transformer
        list_2d, list_3d = [], []
        depth_key = height_key = feat_prev_iv = None
        for fid in range(self.num_frame - 1, -1, -1):
            key_frame = fid == 0
            extra_ref = fid == self.num_frame - self.extra_ref_frames
            if not (key_frame or self.with_prev):
                continue
            s2k, e2g = (s2ks[0], e2gs[0]) if self.align_after_view_transfromation else (s2ks[fid], e2gs[fid])
            mlp_input = vt.get_mlp_input(s2ks[0], e2gs[0], intrins[fid], post_rots[fid], post_trans[fid], bda)
            args = (imgs[fid], s2k, e2g, intrins[fid], post_rots[fid], post_trans[fid], bda, mlp_input, feat_prev_iv,
                    curr2adj[fid], extra_ref)
            if key_frame:
                b2, b3, depth_key, height_key, feat_curr = self.prepare_bev_feat(*args)
            else:
                with torch.no_grad():
                    b2, b3, _, _, feat_curr = self.prepare_bev_feat(*args)
                # Under autocast the half-precision copies of the weights made in this no_grad pass sit in
                # autocast's cast cache without a grad_fn; the key frame (processed last) would reuse them and
                # its weights would silently receive no gradient.  Drop them.
                if torch.is_autocast_enabled():
                    torch.clear_autocast_cache()
            if not extra_ref:
                list_2d.append(b2)
                list_3d.append(b3)
            if not key_frame:
                feat_prev_iv = feat_curr
        if not self.with_prev:
            n_prev = self.num_frame - self.extra_ref_frames - 1
            k2, k3 = list_2d[0], list_3d[0]
            z2 = k2.new_zeros((k2.shape[0], k2.shape[1] * n_prev) + tuple(k2.shape[2:]))
            z3 = k3.new_zeros((k3.shape[0], k3.shape[1] * n_prev) + tuple(k3.shape[2:]))
            list_2d, list_3d = [z2, k2], [z3, k3]
        if self.align_after_view_transfromation:
            if list_2d[0].dim() != 4:
                raise NotImplementedError('aligning (B,C,Dz,Dy,Dx) features: the reference passes them to a 4-D grid_sample')
            for adj_id in range(self.num_frame - 2):
                pair = [s2ks[0], s2ks[self.num_frame - 2 - adj_id]]
                list_2d[adj_id] = self.shift_feature(list_2d[adj_id], pair, bda)
                list_3d[adj_id] = self.shift_feature(list_3d[adj_id], pair, bda)
        bev_2d = torch.cat(list_2d, dim=1)
        bev_3d = torch.cat(list_3d, dim=1)                    # (B, C*frames, 16, Dy, Dx)
        bev_2d = torch.cat(bev_2d.unbind(dim=2), 1)
        colz = lambda t: torch.cat(t.unbind(dim=2), 1)
        x_2d = self.bev_encoder(bev_2d)
        x_3d = torch.cat((self.voxel_encoder0(colz(bev_3d[:, :, :4])), self.voxel_encoder1(colz(bev_3d[:, :, 4:8])),
                          self.voxel_encoder2(colz(bev_3d[:, :, 8:]))), dim=1)
        return x_2d, x_3d, depth_key, height_key

    def forward_train(self, points=None, img_metas=None, img_inputs=None, **kwargs):
        x_2d, x_3d, _, depth, height = self.extract_feat(points, img_inputs=img_inputs, img_metas=img_metas, **kwargs)
        loss_depth, loss_height = self.img_view_transformer.get_depth_and_height_loss(kwargs['gt_depth'], kwargs['gt_height'],
                                                                                      depth, height)
        losses = dict(loss_depth=loss_depth, loss_height=loss_height)
        losses.update(self.forward_occ_train([x_2d, x_3d], kwargs['voxel_semantics'], kwargs['mask_camera']))
        return losses


def dhd_s_model_cfg(**overrides):
    """The `model = dict(...)` block of projects/configs/DHD/DHD-S.py:42-155, verbatim values."""
    from .synthetic import dhd_s_config
    n = 64
    cfg = dict(
        type='DHD',
        img_backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(2, 3), frozen_stages=-1,
                          norm_cfg=dict(type='BN', requires_grad=True), norm_eval=False, with_cp=True, style='pytorch',
                          pretrained='torchvision://resnet50'),
        img_neck=dict(type='CustomFPN', in_channels=[1024, 2048], out_channels=256, num_outs=1, start_level=0, out_ids=[0]),
        img_view_transformer=dict(type='MGHS', **dhd_s_config()),
        img_bev_encoder_backbone=dict(type='CustomResNet', numC_input=n, num_channels=[n * 2, n * 4, n * 8]),
        img_bev_encoder_neck=dict(type='FPN_LSS', in_channels=n * 8 + n * 2, out_channels=256),
        img_voxel_encoder0_backbone=dict(type='UNet', n_channels=n * 4, n_classes=64),
        img_voxel_encoder0_neck=dict(type='Identity'),
        img_voxel_encoder1_backbone=dict(type='UNet', n_channels=n * 4, n_classes=128),
        img_voxel_encoder1_neck=dict(type='Identity'),
        img_voxel_encoder2_backbone=dict(type='UNet', n_channels=n * 8, n_classes=64),
        img_voxel_encoder2_neck=dict(type='Identity'),
        mix=dict(type='SFA', in_channels=512, out_channels=256),
        occ_head=dict(type='predictor', in_dim=256, out_dim=256, Dz=16, use_mask=True, num_classes=18, use_predicter=True,
                      class_balance=True, weight_ce=10.0, weight_geo=0.2, weight_sem=0.2,
                      loss_occ=dict(type='CrossEntropyLoss', use_sigmoid=False, ignore_index=255, loss_weight=1.0)))
    cfg.update(overrides)
    return cfg


def dhd_m_model_cfg(input_size=(256, 704)):
    """The model block of projects/configs/DHD/DHD-M.py:41-170 (values verbatim), with a reduced image size."""
    n = 64
    band = lambda z: {'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': z, 'depth': [1.0, 45.0, 0.5]}
    return dict(
        type='DHD_stereo', align_after_view_transfromation=False, num_adj=1,
        img_backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 2, 3), frozen_stages=-1,
                          norm_cfg=dict(type='BN', requires_grad=True), norm_eval=False, with_cp=True, style='pytorch'),
        img_neck=dict(type='CustomFPN', in_channels=[1024, 2048], out_channels=256, num_outs=1, start_level=0, out_ids=[0]),
        img_view_transformer=dict(
            type='MGHS_Stereo', grid_config={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 6.4], 'depth': [1.0, 45.0, 0.5]},
            input_size=input_size, height_range=[round(-1.0 + 0.1 * i, 1) for i in range(65)], height_interval=0.1,
            mask_range=[-1.0, 0.6, 2.2, 5.4], mask_1_grid=band([-1, 0.6, 0.4]), mask_2_grid=band([0.6, 2.2, 0.4]),
            mask_3_grid=band([2.2, 5.4, 0.4]), in_channels=256, out_channels=n, sid=False, collapse_z=False,
            loss_height_weight=0.1, loss_depth_weight=0.05,
            depthnet_cfg=dict(use_dcn=False, aspp_mid_channels=96, stereo=True, bias=5.), downsample=16),
        img_bev_encoder_backbone=dict(type='UNet', n_channels=n * 2, n_classes=512),
        img_bev_encoder_neck=dict(type='Identity'),
        pre_process=dict(type='CustomResNet', numC_input=n, num_layer=[1], num_channels=[n], stride=[1], backbone_output_ids=[0]),
        pre_process_net_3d=dict(type='CustomResNet', numC_input=n * 16, num_layer=[1], num_channels=[n * 16], stride=[1],
                                backbone_output_ids=[0]),
        img_voxel_encoder0_backbone=dict(type='UNet', n_channels=n * 4 * 2, n_classes=128), img_voxel_encoder0_neck=dict(type='Identity'),
        img_voxel_encoder1_backbone=dict(type='UNet', n_channels=n * 4 * 2, n_classes=256), img_voxel_encoder1_neck=dict(type='Identity'),
        img_voxel_encoder2_backbone=dict(type='UNet', n_channels=n * 8 * 2, n_classes=128), img_voxel_encoder2_neck=dict(type='Identity'),
        mix=dict(type='SFA', in_channels=1024, out_channels=512),
        occ_head=dict(type='predictor', in_dim=512, out_dim=256, Dz=16, use_mask=True, num_classes=18, use_predicter=True,
                      class_balance=True, weight_ce=10.0, weight_geo=0.2, weight_sem=0.2,
                      loss_occ=dict(type='CrossEntropyLoss', use_sigmoid=False, ignore_index=255, loss_weight=1.0)))


def dhd_l_model_cfg(input_size=(512, 1408)):
    """The model block of projects/configs/DHD/DHD-L.py:41-185 (values verbatim): Swin-B image backbone, FPN_LSS
    necks, 512-channel view-transformer input at 1/16 of 512 x 1408 (fH x fW = 32 x 88, D = 88)."""
    n = 64
    band = lambda z: {'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': z, 'depth': [1.0, 45.0, 0.5]}
    return dict(
        type='DHD_stereo', align_after_view_transfromation=False, num_adj=1,
        img_backbone=dict(type='SwinTransformer', pretrain_img_size=224, patch_size=4, window_size=12, mlp_ratio=4, embed_dims=128,
                          depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], strides=(4, 2, 2, 2), out_indices=(2, 3), qkv_bias=True,
                          qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, use_abs_pos_embed=False,
                          return_stereo_feat=True, act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN', requires_grad=True),
                          pretrain_style='official', output_missing_index_as_none=False),
        img_neck=dict(type='FPN_LSS', in_channels=512 + 1024, out_channels=512, extra_upsample=None, input_feature_index=(0, 1),
                      scale_factor=2),
        img_view_transformer=dict(
            type='MGHS_Stereo', grid_config={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 6.4], 'depth': [1.0, 45.0, 0.5]},
            input_size=input_size, height_range=[round(-1.0 + 0.1 * i, 1) for i in range(65)], height_interval=0.1,
            mask_range=[-1.0, 0.6, 2.2, 5.4], mask_1_grid=band([-1, 0.6, 0.4]), mask_2_grid=band([0.6, 2.2, 0.4]),
            mask_3_grid=band([2.2, 5.4, 0.4]), in_channels=512, out_channels=n, sid=False, collapse_z=False,
            loss_height_weight=0.1, loss_depth_weight=0.05,
            depthnet_cfg=dict(use_dcn=False, aspp_mid_channels=96, stereo=True, bias=5.),
            heightnet_cfg=dict(use_dcn=False, aspp_mid_channels=96), downsample=16),
        img_bev_encoder_backbone=dict(type='CustomResNet', with_cp=True, numC_input=n * 2, num_channels=[n * 2, n * 4, n * 8]),
        img_bev_encoder_neck=dict(type='FPN_LSS', in_channels=n * 8 + n * 2, out_channels=256),
        pre_process=dict(type='CustomResNet', numC_input=n, num_layer=[1], num_channels=[n], stride=[1], backbone_output_ids=[0]),
        pre_process_net_3d=dict(type='CustomResNet', numC_input=n * 16, num_layer=[1], num_channels=[n * 16], stride=[1],
                                backbone_output_ids=[0]),
        img_voxel_encoder0_backbone=dict(type='UNet', n_channels=n * 4 * 2, n_classes=64), img_voxel_encoder0_neck=dict(type='Identity'),
        img_voxel_encoder1_backbone=dict(type='UNet', n_channels=n * 4 * 2, n_classes=128), img_voxel_encoder1_neck=dict(type='Identity'),
        img_voxel_encoder2_backbone=dict(type='UNet', n_channels=n * 8 * 2, n_classes=64), img_voxel_encoder2_neck=dict(type='Identity'),
        mix=dict(type='SFA', in_channels=512, out_channels=256),
        occ_head=dict(type='predictor', in_dim=256, out_dim=256, Dz=16, use_mask=True, num_classes=18, use_predicter=True,
                      class_balance=True, weight_ce=10.0, weight_geo=0.2, weight_sem=0.2,
                      loss_occ=dict(type='CrossEntropyLoss', use_sigmoid=False, ignore_index=255, loss_weight=1.0)))
