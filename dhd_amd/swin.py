"""Swin Transformer image backbone of DHD-L (projects/configs/DHD/DHD-L.py:43-66; reference module
models/backbones/swin.py, itself the mmdet backbone with BEVDet's `return_stereo_feat`).

A dense caller of the hot path: its stage outputs feed FPN_LSS and the view transformer.  State-dict keys
follow the reference (`patch_embed.projection`, `stages.i.blocks.j.attn.w_msa.qkv`, `...ffn.layers.0.0`,
`stages.i.downsample.reduction`, `norm{i}`), so its checkpoints load.  Differences in how it runs:
window attention goes through `F.scaled_dot_product_attention` with one additive term (relative-position
bias + shift mask) instead of materialised score matrices, and the shift masks are built once per feature
map size instead of in every block's forward (swin.py:420-447 rebuilds them 24x per image)."""
import warnings
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils import checkpoint

from .registry import BACKBONES


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class DropPath(nn.Module):
    """Stochastic depth per sample (mmcv's `DropPath`, the `dropout_layer` of swin.py:566,575)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = keep + torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device)
        return x.div(keep) * mask.floor()


class FFN(nn.Module):
    """mmcv-full 1.5.3 `FFN` with num_fcs = 2 as swin.py:569-577 builds it: keys layers.0.0 / layers.1."""

    def __init__(self, embed_dims, feedforward_channels, ffn_drop=0., drop_path=0.):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.GELU(), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))
        self.dropout_layer = DropPath(drop_path)

    def forward(self, x, identity=None):
        return (x if identity is None else identity) + self.dropout_layer(self.layers(x))


class PatchEmbed(nn.Module):
    """swin.py:79-170: non-overlapping conv patches (input padded up to the patch size), tokens (B, L, C)."""

    def __init__(self, in_channels=3, embed_dims=96, kernel_size=4, stride=4, norm=True):
        super().__init__()
        self.patch_size = _pair(kernel_size)
        self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size=kernel_size, stride=stride)
        self.norm = nn.LayerNorm(embed_dims) if norm else None

    def forward(self, x):
        ph, pw = self.patch_size
        H, W = x.shape[2:]
        if H % ph or W % pw:
            x = F.pad(x, (0, (pw - W % pw) % pw, 0, (ph - H % ph) % ph))
        x = self.projection(x)
        self.DH, self.DW = x.shape[2], x.shape[3]
        x = x.flatten(2).transpose(1, 2)
        return x if self.norm is None else self.norm(x)


class PatchMerging(nn.Module):
    """swin.py:174-241: 2x2 neighbourhoods -> 4C channels in nn.Unfold's order (c, kh, kw), LN, Linear 4C->2C."""

    def __init__(self, in_channels, out_channels, stride=2, norm=True):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.norm = nn.LayerNorm(stride ** 2 * in_channels) if norm else None
        self.reduction = nn.Linear(stride ** 2 * in_channels, out_channels, bias=False)

    def forward(self, x, hw_shape):
        B, L, C = x.shape
        H, W = hw_shape
        s = self.stride
        x = x.view(B, H, W, C)
        if H % s or W % s:
            x = F.pad(x, (0, 0, 0, W % s, 0, H % s))  # the reference pads by the remainder (swin.py:229-230)
        Hp, Wp = x.shape[1] // s, x.shape[2] // s
        x = x[:, :Hp * s, :Wp * s].reshape(B, Hp, s, Wp, s, C).permute(0, 1, 3, 5, 2, 4).reshape(B, Hp * Wp, C * s * s)
        if self.norm is not None:
            x = self.norm(x)
        return self.reduction(x), ((H + 1) // 2, (W + 1) // 2)


class WindowMSA(nn.Module):
    """swin.py:244-350: multi-head attention inside a window with a learned relative-position bias."""

    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None, attn_drop_rate=0., proj_drop_rate=0.):
        super().__init__()
        self.embed_dims, self.num_heads, self.window_size = embed_dims, num_heads, window_size
        self.scale = qk_scale or (embed_dims // num_heads) ** -0.5
        Wh, Ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        ys, xs = torch.meshgrid(torch.arange(Wh), torch.arange(Ww), indexing='ij')
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        index = (ys[:, None] - ys[None, :] + Wh - 1) * (2 * Ww - 1) + (xs[:, None] - xs[None, :] + Ww - 1)
        self.register_buffer('relative_position_index', index.contiguous())
        self.qkv = nn.Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop_rate)
        self.proj = nn.Linear(embed_dims, embed_dims)
        self.proj_drop = nn.Dropout(proj_drop_rate)

    def init_weights(self):
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, mask=None):
        """x: (B, nW, N, C) windows; mask: (nW, N, N) additive shift mask or None."""
        B, nW, N, C = x.shape
        nh = self.num_heads
        # windows x heads as the "head" axis of a 4-D attention call: the additive term (nW * heads, N, N) is then
        # shared by the whole batch without being copied per image, and the fused attention kernels apply
        qkv = self.qkv(x).view(B, nW, N, 3, nh, C // nh).permute(3, 0, 1, 4, 2, 5).reshape(3, B, nW * nh, N, C // nh)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 0, 1)
        bias = bias.unsqueeze(0).expand(nW, nh, N, N) if mask is None else bias.unsqueeze(0) + mask.unsqueeze(1)
        out = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=bias.reshape(1, nW * nh, N, N).to(qkv.dtype),
                                             dropout_p=self.attn_drop.p if self.training else 0., scale=self.scale)
        out = out.view(B, nW, nh, N, C // nh)
        return self.proj_drop(self.proj(out.transpose(2, 3).reshape(B, nW, N, C)))


def shift_window_mask(H_pad, W_pad, window, shift, device):
    """swin.py:420-447: windows that straddle the cyclic seam must not attend across it (-100 between regions)."""
    img = torch.zeros(H_pad, W_pad, device=device)
    cuts = (slice(0, -window), slice(-window, -shift), slice(-shift, None))
    for i, hs in enumerate(cuts):
        for j, ws in enumerate(cuts):
            img[hs, ws] = 3 * i + j
    win = img.view(H_pad // window, window, W_pad // window, window).permute(0, 2, 1, 3).reshape(-1, window * window)
    diff = win.unsqueeze(1) - win.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


_WIN_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_PLAIN_WINDOWS = bool(__import__('os').environ.get('DHD_PLAIN_WINDOWS'))   # A/B switch: F.pad + torch.roll + permute copies


class _WindowRows(torch.autograd.Function):
    """csrc/window.hip: pad + cyclic shift + window partition (reverse=False), or its inverse (reverse=True), as one row gather;
    the gradient of either is the other one applied to the incoming gradient.  `out_dtype` lets the partition emit the autocast
    dtype the qkv projection would cast its input to anyway (its gradient comes back in the input's dtype)."""

    @staticmethod
    def forward(ctx, x, H, W, ws, sh, reverse, out_dtype):
        from . import _lib
        x = x.contiguous()
        B, C = x.shape[0], x.shape[-1]
        nh, nw = -(-H // ws), -(-W // ws)
        shape = (B, H, W, C) if reverse else (B, nh * nw, ws * ws, C)
        with torch.cuda.device(x.device):
            out = torch.empty(shape, dtype=out_dtype, device=x.device)
            _lib.check(_lib.load().dhd_window_rows(_lib.ptr(x), _lib.ptr(out), _WIN_DTYPES[x.dtype], _WIN_DTYPES[out_dtype], B, H, W, C, ws, sh,
                                                   int(reverse), _lib.stream_ptr(x.device)), 'dhd_window_rows')
        ctx.args = (H, W, ws, sh, reverse, x.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        H, W, ws, sh, reverse, in_dtype = ctx.args
        return _WindowRows.apply(g, H, W, ws, sh, not reverse, in_dtype), None, None, None, None, None, None


def _windows_on_gpu(x):
    return (not _PLAIN_WINDOWS) and x.is_cuda and x.dtype in _WIN_DTYPES and x.shape[-1] % 8 == 0 and x.data_ptr() % 16 == 0


class ShiftWindowMSA(nn.Module):
    """swin.py:353-513."""

    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True, qk_scale=None, attn_drop_rate=0.,
                 proj_drop_rate=0., drop_path=0.):
        super().__init__()
        assert 0 <= shift_size < window_size
        self.window_size, self.shift_size = window_size, shift_size
        self.w_msa = WindowMSA(embed_dims, num_heads, _pair(window_size), qkv_bias, qk_scale, attn_drop_rate, proj_drop_rate)
        self.drop = DropPath(drop_path)

    def forward(self, query, hw_shape, masks=None):
        B, L, C = query.shape
        H, W = hw_shape
        ws, sh = self.window_size, self.shift_size
        x = query.view(B, H, W, C)
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        if _windows_on_gpu(x):
            # the same data movement as below in two launches instead of six copies (csrc/window.hip); under autocast the
            # windows leave in the dtype the qkv projection casts its input to, and come back in the projection's dtype
            mask = None
            if sh > 0:
                key = (H + pad_b, W + pad_r, ws, sh, x.device)
                if masks is None:
                    mask = shift_window_mask(H + pad_b, W + pad_r, ws, sh, x.device)
                else:
                    if key not in masks:
                        masks[key] = shift_window_mask(H + pad_b, W + pad_r, ws, sh, x.device)
                    mask = masks[key]
            odt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() and x.dtype == torch.float32 else x.dtype
            win = _WindowRows.apply(x, H, W, ws, sh, False, odt)
            win = self.w_msa(win, mask)
            x = _WindowRows.apply(win, H, W, ws, sh, True, win.dtype)
            return self.drop(x.view(B, H * W, C))
        if pad_r or pad_b:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = H + pad_b, W + pad_r
        mask = None
        if sh > 0:
            x = torch.roll(x, shifts=(-sh, -sh), dims=(1, 2))
            key = (Hp, Wp, ws, sh, x.device)
            if masks is None:
                mask = shift_window_mask(Hp, Wp, ws, sh, x.device)
            else:
                if key not in masks:
                    masks[key] = shift_window_mask(Hp, Wp, ws, sh, x.device)
                mask = masks[key]
        nh, nw = Hp // ws, Wp // ws
        win = x.view(B, nh, ws, nw, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, nh * nw, ws * ws, C)
        win = self.w_msa(win, mask)
        x = win.view(B, nh, nw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if sh > 0:
            x = torch.roll(x, shifts=(sh, sh), dims=(1, 2))
        if pad_r or pad_b:
            x = x[:, :H, :W].contiguous()
        return self.drop(x.view(B, H * W, C))


class SwinBlock(nn.Module):
    """swin.py:516-592: x + attn(LN(x)), then x + ffn(LN(x))."""

    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False, qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0, qkv_bias, qk_scale,
                                   attn_drop_rate, drop_rate, drop_path_rate)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.ffn = FFN(embed_dims, feedforward_channels, drop_rate, drop_path_rate)

    def forward(self, x, hw_shape, masks=None):
        x = x + self.attn(self.norm1(x), hw_shape, masks)
        return self.ffn(self.norm2(x), identity=x)


class SwinBlockSequence(nn.Module):
    """swin.py:595-677: `depth` blocks alternating plain / shifted windows, then the optional PatchMerging.
    Returns (x_down, down_hw, x, hw)."""

    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size=7, qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., downsample=None, with_cp=True):
        super().__init__()
        rates = drop_path_rate if isinstance(drop_path_rate, list) else [drop_path_rate] * depth
        self.blocks = nn.ModuleList(
            SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, i % 2 == 1, qkv_bias, qk_scale, drop_rate,
                      attn_drop_rate, rates[i]) for i in range(depth))
        self.downsample = downsample
        self.with_cp = with_cp

    def forward(self, x, hw_shape, masks=None):
        for block in self.blocks:
            if self.with_cp and x.requires_grad:
                x = checkpoint.checkpoint(block, x, hw_shape, masks, use_reentrant=False)
            else:
                x = block(x, hw_shape, masks)
        if self.downsample is not None:
            down, down_hw = self.downsample(x, hw_shape)
            return down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


def convert_official_swin(ckpt):
    """Key / layout translation of a checkpoint of the official Swin repository (the reference's `swin_convert`,
    swin.py:25-76): layers -> stages, attn -> attn.w_msa, mlp.fc1/fc2 -> ffn.layers.0.0 / ffn.layers.1,
    patch_embed.proj -> projection, and the PatchMerging weights from the official (kw, kh, c) channel order to
    nn.Unfold's (c, kh, kw)."""
    out = OrderedDict()
    for k, v in ckpt.items():
        if k.startswith('head'):
            continue
        if k.startswith('layers'):
            if 'attn.' in k:
                k = k.replace('attn.', 'attn.w_msa.')
            elif 'mlp.fc1.' in k:
                k = k.replace('mlp.fc1.', 'ffn.layers.0.0.')
            elif 'mlp.fc2.' in k:
                k = k.replace('mlp.fc2.', 'ffn.layers.1.')
            elif 'mlp.' in k:
                k = k.replace('mlp.', 'ffn.')
            elif 'downsample' in k and ('reduction.' in k or 'norm.' in k):
                lead = v.shape[:-1]
                v = v.reshape(*lead, 2, 2, v.shape[-1] // 4).permute(*range(len(lead)), len(lead) + 2, len(lead) + 1,
                                                                     len(lead)).reshape(*lead, -1)
            k = k.replace('layers', 'stages', 1)
        elif k.startswith('patch_embed') and 'proj' in k:
            k = k.replace('proj', 'projection')
        out[k] = v
    return out


@BACKBONES.register_module()
class SwinTransformer(nn.Module):
    """swin.py:680-976.  `forward` returns the normalised outputs of `out_indices` as (B, C, H, W), preceded by
    the un-normalised stage-0 output when `return_stereo_feat` (the temporal-stereo feature of DHD-M/L)."""

    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3), qkv_bias=True,
                 qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, use_abs_pos_embed=False,
                 act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN'), pretrain_style='official', pretrained=None, init_cfg=None,
                 with_cp=True, return_stereo_feat=False, output_missing_index_as_none=False, frozen_stages=-1):
        super().__init__()
        assert pretrain_style in ('official', 'mmcls')
        assert act_cfg.get('type', 'GELU') == 'GELU' and norm_cfg.get('type', 'LN') == 'LN', 'only GELU / LN are mirrored'
        assert strides[0] == patch_size, 'Use non-overlapping patch embed.'
        if not (isinstance(pretrained, str) or pretrained is None):
            raise TypeError('pretrained must be a str or None')
        pretrain_img_size = _pair(pretrain_img_size)
        self.out_indices, self.use_abs_pos_embed = out_indices, use_abs_pos_embed
        self.pretrain_style, self.pretrained, self.init_cfg, self.frozen_stages = pretrain_style, pretrained, init_cfg, frozen_stages
        self.patch_embed = PatchEmbed(in_channels, embed_dims, patch_size, strides[0], patch_norm)
        if use_abs_pos_embed:
            n = (pretrain_img_size[0] // patch_size) * (pretrain_img_size[1] // patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, n, embed_dims))
        self.drop_after_pos = nn.Dropout(p=drop_rate)
        dpr = [r.item() for r in torch.linspace(0, drop_path_rate, sum(depths))]  # stochastic depth decay rule
        self.stages = nn.ModuleList()
        ch = embed_dims
        for i, depth in enumerate(depths):
            down = PatchMerging(ch, 2 * ch, strides[i + 1], patch_norm) if i < len(depths) - 1 else None
            self.stages.append(SwinBlockSequence(ch, num_heads[i], mlp_ratio * ch, depth, window_size, qkv_bias, qk_scale,
                                                 drop_rate, attn_drop_rate, dpr[:depth], down, with_cp))
            dpr = dpr[depth:]
            if down is not None:
                ch = down.out_channels
        self.num_features = [int(embed_dims * 2 ** i) for i in range(len(depths))]
        for i in out_indices:
            self.add_module(f'norm{i}', nn.LayerNorm(self.num_features[i]))
        self.output_missing_index_as_none = output_missing_index_as_none
        self.return_stereo_feat = return_stereo_feat
        self._masks = {}  # shift masks per (padded size, window, shift, device): shared by all blocks
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1 and self.use_abs_pos_embed:
            self.absolute_pos_embed.requires_grad = False
        if self.frozen_stages >= 2:
            self.drop_after_pos.eval()
            for i in range(self.frozen_stages - 1):
                self.stages[i].eval()
                for p in self.stages[i].parameters():
                    p.requires_grad = False

    def init_weights(self):
        if self.pretrained is None:
            if self.use_abs_pos_embed:
                nn.init.trunc_normal_(self.absolute_pos_embed, std=0.02)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=.02)
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)
                elif isinstance(m, nn.LayerNorm):
                    nn.init.zeros_(m.bias)
                    nn.init.ones_(m.weight)
                elif isinstance(m, WindowMSA):
                    m.init_weights()
            return
        ckpt = torch.load(self.pretrained, map_location='cpu')
        state = ckpt.get('state_dict', ckpt.get('model', ckpt))
        if self.pretrain_style == 'official':
            state = convert_official_swin(state)
        if next(iter(state)).startswith('module.'):
            state = {k[7:]: v for k, v in state.items()}
        own = self.state_dict()
        for k in [k for k in state if 'relative_position_bias_table' in k and k in own]:
            (L1, h1), (L2, h2) = state[k].shape, own[k].shape
            if h1 != h2:
                warnings.warn(f'Error in loading {k}, pass')
            elif L1 != L2:   # another window size: bicubic resize of the (2w-1)^2 table
                S1, S2 = int(L1 ** 0.5), int(L2 ** 0.5)
                t = F.interpolate(state[k].permute(1, 0).reshape(1, h1, S1, S1), size=(S2, S2), mode='bicubic', align_corners=False)
                state[k] = t.view(h2, L2).permute(1, 0).contiguous()
        self.load_state_dict(state, strict=False)

    def _tokens(self, x):
        x = self.patch_embed(x)
        if self.use_abs_pos_embed:
            x = x + self.absolute_pos_embed
        return self.drop_after_pos(x), (self.patch_embed.DH, self.patch_embed.DW)

    def _to_map(self, out, hw, i):
        return out.view(-1, *hw, self.num_features[i]).permute(0, 3, 1, 2).contiguous()

    def forward_first_stage(self, x):
        """Stereo feature of the extra reference frame (detectors/bevstereo4d.py:41-54): stage 0 only."""
        x, hw = self._tokens(x)
        _, _, out, out_hw = self.stages[0](x, hw, self._masks)
        return self._to_map(out, out_hw, 0)

    def forward(self, x):
        x, hw = self._tokens(x)
        outs = []
        for i, stage in enumerate(self.stages):
            x, hw, out, out_hw = stage(x, hw, self._masks)
            if i == 0 and self.return_stereo_feat:
                outs.append(self._to_map(out, out_hw, 0))
            if i in self.out_indices:
                outs.append(self._to_map(getattr(self, f'norm{i}')(out), out_hw, i))
            elif self.output_missing_index_as_none:
                outs.append(None)
        return outs

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self
