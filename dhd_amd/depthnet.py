"""HeightNet / DepthNet: the dense producers of the lift inputs (height logits, depth logits,
context features).  Same constructor arguments, forward signatures and state-dict key names as
the reference's models/model_utils/depthnet.py (HeightNet :418-652, DepthNet :172-415, ASPP
:42-116, Mlp :119-147, SELayer :150-169) so reference checkpoints map one-to-one.

These are convolution stacks: they run on PyTorch-ROCm's MIOpen/hipBLASLt (MFMA) kernels, not on
hand-written HIP -- BASELINE.json's north_star reserves MFMA for exactly this dense work.  Two
third-party blocks the reference pulls from un-vendored packages are restated here from their
published definitions (parity UNPINNED: no reference test or fixture covers them):
  * mmdet 2.25.1 `BasicBlock`  (3x3 conv-BN-ReLU-3x3 conv-BN + identity / downsample, ReLU);
  * mmcv-full 1.5.3 `DCN` (= DeformConv2dPack: zero-initialised 3x3 offset conv + deformable
    convolution v1, offsets ordered (dy, dx) per tap, bilinear sampling with zero padding).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .batchnorm import BatchNorm2d, bn_act


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = bn_act(self.bn1, self.conv1(x), relu=True)
        return bn_act(self.bn2, self.conv2(out), residual=identity)


class _DeformIm2col(torch.autograd.Function):
    """x (B,C,H,W), offset (B,2kk,H,W) -> col (B, C*kk, H*W) through libdhd_amd.so (csrc/deform.hip).  `col_dtype`: float32, or the
    autocast half type -- the GEMM behind the sampling runs in it, so the 155 MB column matrix of the DHD-S HeightNet is written
    once as 78 MB of half and its gradient is read as half, with no cast kernels in between.  x is staged as float32 NCHW (the
    sampling kernels want consecutive cells on consecutive lanes and 4-byte gathers).  Backward: the gather form of
    col2im (dhd_deform_col2im_t) where the library takes the shape, else the float32 LDS-atomic form."""

    @staticmethod
    def forward(ctx, x, offset, k, pad, dil, col_dtype=torch.float32):
        from . import _lib
        if not x.is_cuda:
            raise _lib.DhdError(f'DCN input must live on the GPU: dhd_amd runs only as HIP kernels (got {x.device})')
        # float32 NCHW staging copy of x (8.6 -> 17 MB at the DHD-S size): the library also reads a half x (x_dtype), but its
        # corner gathers as 2-byte loads ran col2offset at 466 us against 138 us on the float32 copy (profiles/r6)
        x = x.float().contiguous()
        offset = _lib.require_gpu_tensor(offset.float().contiguous(), torch.float32, 'DCN offsets')
        b, c, h, w = x.shape
        dev = x.device
        with torch.cuda.device(dev):
            col = torch.empty((b, c * k * k, h * w), dtype=col_dtype, device=dev)
            _lib.check(_lib.load().dhd_deform_im2col_t(_lib.ptr(x), _lib.dtype_code(x.dtype), _lib.ptr(offset), _lib.ptr(col),
                                                       _lib.dtype_code(col_dtype), b, c, h, w, k, pad, dil, _lib.stream_ptr(dev)),
                       'dhd_deform_im2col_t')
        ctx.save_for_backward(x, offset)
        ctx.args = (k, pad, dil, col_dtype)
        return col

    @staticmethod
    def backward(ctx, dcol):
        from . import _lib
        x, offset = ctx.saved_tensors
        k, pad, dil, col_dtype = ctx.args
        b, c, h, w = x.shape
        dev = x.device
        lib = _lib.load()
        with torch.cuda.device(dev):
            doff = torch.empty_like(offset)
            if lib.dhd_deform_col2im_gather_supported(_lib.dtype_code(col_dtype), h, w, k):
                dx = torch.empty_like(x)      # x's dtype and strides
                dcol = dcol.to(col_dtype).contiguous()
                ws = torch.empty(lib.dhd_deform_col2im_workspace_bytes(b, h, w, k), dtype=torch.uint8, device=dev)
                _lib.check(lib.dhd_deform_col2im_t(_lib.ptr(dcol), _lib.dtype_code(col_dtype), _lib.ptr(x), _lib.dtype_code(x.dtype),
                                                   _lib.ptr(offset), _lib.ptr(dx), _lib.ptr(doff), b, c, h, w, k, pad, dil, _lib.ptr(ws),
                                                   ws.numel(), _lib.stream_ptr(dev)), 'dhd_deform_col2im_t')
            else:
                xf = x.float().contiguous()
                dx = torch.empty_like(xf)
                dcol = dcol.float().contiguous()
                _lib.check(lib.dhd_deform_col2im(_lib.ptr(dcol), _lib.ptr(xf), _lib.ptr(offset), _lib.ptr(dx), _lib.ptr(doff), b, c, h, w,
                                                 k, pad, dil, _lib.stream_ptr(dev)), 'dhd_deform_col2im')
                dx = dx.to(x.dtype)
        return dx, doff, None, None, None, None


class DCN(nn.Module):
    """Deformable convolution v1 with its own offset branch (mmcv DeformConv2dPack).  On the GPU the sampling
    (deformable im2col and its two backward passes) runs in libdhd_amd.so; the grid_sample formulation
    below is the same math for CPU tensors and serves as the cross-check in the tests."""

    use_hip = True

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 deform_groups=1, im2col_step=128, bias=False):
        super().__init__()
        assert not bias and stride == 1 and deform_groups == 1
        self.k, self.padding, self.dilation, self.groups = kernel_size, padding, dilation, groups
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size, kernel_size))
        n = in_channels * kernel_size * kernel_size
        bound = 1.0 / n ** 0.5
        nn.init.uniform_(self.weight, -bound, bound)
        self.conv_offset = nn.Conv2d(in_channels, 2 * kernel_size * kernel_size, kernel_size, stride=1,
                                     padding=padding, dilation=dilation, bias=True)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        b, c, h, w = x.shape
        k = self.k
        g = self.groups
        wgt = self.weight.reshape(g, self.out_channels // g, (c // g) * k * k)
        if self.use_hip and x.is_cuda and h * w * 4 <= 48 * 1024:
            # sampling in HIP (float32 arithmetic); the GEMM with the layer's weight follows the ambient autocast dtype, and so
            # does the column matrix.  matmul with the weight broadcast over the batch: the (B, g, K, HW) operand is used where it
            # lies (einsum re-laid it out with a 155 MB copy each way)
            cdt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else torch.float32
            if cdt not in (torch.float16, torch.bfloat16):
                cdt = torch.float32
            col = _DeformIm2col.apply(x, self.conv_offset(x), k, self.padding, self.dilation, cdt)
            out = torch.matmul(wgt, col.view(b, g, (c // g) * k * k, h * w))      # (g, o, K) x (B, g, K, HW) -> (B, g, o, HW)
            return out.reshape(b, self.out_channels, h, w)
        offset = self.conv_offset(x).view(b, k * k, 2, h, w)
        ys, xs = torch.meshgrid(torch.arange(h, device=x.device, dtype=x.dtype),
                                torch.arange(w, device=x.device, dtype=x.dtype), indexing='ij')
        cols = []
        for t in range(k * k):
            ky, kx = divmod(t, k)
            py = ys + (ky * self.dilation - self.padding) + offset[:, t, 0]
            px = xs + (kx * self.dilation - self.padding) + offset[:, t, 1]
            grid = torch.stack((2 * px / max(w - 1, 1) - 1, 2 * py / max(h - 1, 1) - 1), -1)
            cols.append(F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=True))
        col = torch.stack(cols, 2)  # (B, C, k*k, H, W)
        col = col.view(b, g, (c // g) * k * k, h * w)
        out = torch.einsum('gok,bgkp->bgop', wgt, col)
        return out.reshape(b, self.out_channels, h, w)


class _ASPPModule(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, padding, dilation, BatchNorm):
        super().__init__()
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size, stride=1, padding=padding, dilation=dilation,
                                     bias=False)
        self.bn = BatchNorm(planes)
        self.relu = nn.ReLU()
        _kaiming(self)

    def forward(self, x):
        return self.relu(self.bn(self.atrous_conv(x)))


def _kaiming(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class ASPP(nn.Module):
    def __init__(self, inplanes, mid_channels=256, BatchNorm=nn.BatchNorm2d):
        super().__init__()
        self.aspp1 = _ASPPModule(inplanes, mid_channels, 1, 0, 1, BatchNorm)
        self.aspp2 = _ASPPModule(inplanes, mid_channels, 3, 6, 6, BatchNorm)
        self.aspp3 = _ASPPModule(inplanes, mid_channels, 3, 12, 12, BatchNorm)
        self.aspp4 = _ASPPModule(inplanes, mid_channels, 3, 18, 18, BatchNorm)
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)),
                                             nn.Conv2d(inplanes, mid_channels, 1, stride=1, bias=False),
                                             BatchNorm(mid_channels), nn.ReLU())
        self.conv1 = nn.Conv2d(int(mid_channels * 5), inplanes, 1, bias=False)
        self.bn1 = BatchNorm(inplanes)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)
        _kaiming(self)

    def forward(self, x):
        branches = [self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x)]
        pooled = self.global_avg_pool[1](self.global_avg_pool[0](x))          # AdaptiveAvgPool2d((1, 1)) -> 1x1 conv
        if pooled.shape[2:] == (1, 1) and pooled.stride() != (pooled.shape[1], 1, 1, 1):   # (is_contiguous() ignores size-1 axes)
            # a channels_last convolution returns the (N, C, 1, 1) map with strides (C, 1, C, C); MIOpen's half-precision training
            # BatchNorm SEGFAULTS on that at N = 2 (experiments/bn_1x1_crash_probe.py: plain strides, float32 or N = 24 are fine).
            # The same bytes with plain strides:
            pooled = pooled.reshape(pooled.shape[0], pooled.shape[1]).reshape(pooled.shape)
        pooled = self.global_avg_pool[3](self.global_avg_pool[2](pooled))     # BatchNorm -> ReLU
        # F.interpolate(pooled, size, 'bilinear', align_corners=True) of a 1 x 1 map (mmdet3d depthnet ASPP.forward) is that value
        # everywhere (scale 0: cell 0, lambda 0) -- a broadcast view instead of a float32 up-sampling kernel under autocast
        if pooled.shape[2:] == (1, 1):
            pooled = pooled.expand(-1, -1, *x.shape[2:])
            if branches[0].is_contiguous(memory_format=torch.channels_last) and not branches[0].is_contiguous():
                # a stride-0 view among channels_last branches makes `cat` answer in NCHW, and conv1 then re-lays the
                # (B, 5 mid, H, W) tensor out twice per step (2 x 93 us at the DHD-S size); one small dense copy instead
                pooled = pooled.contiguous(memory_format=torch.channels_last)
            branches.append(pooled)
        else:
            branches.append(F.interpolate(pooled, size=x.shape[2:], mode='bilinear', align_corners=True))
        x = self.relu(self.bn1(self.conv1(torch.cat(branches, dim=1))))
        return self.dropout(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.ReLU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class SELayer(nn.Module):
    def __init__(self, channels, act_layer=nn.ReLU, gate_layer=nn.Sigmoid):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.act1 = act_layer()
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)
        self.gate = gate_layer()

    def forward(self, x, x_se):
        return x * self.gate(self.conv_expand(self.act1(self.conv_reduce(x_se))))


class _AddChannelBias(torch.autograd.Function):
    """y + bias[None, :, None, None] whose bias gradient is a (1 x rows) @ (rows x C) product when the gradient is channels_last.
    torch's `sum((0, 2, 3))` of a channels_last tensor with an ODD channel count (HeightNet's 65 height bins) runs 111-136 us on
    MI355X for 1.1 M elements (experiments/bias_sum_probe.py; 8 us in NCHW, 13 us for 108 channels): the single largest
    reduction of the view transformer's backward."""

    @staticmethod
    def forward(ctx, y, bias):
        return y + bias.to(y.dtype).view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        c = g.shape[1]
        if g.is_cuda and not g.is_contiguous() and g.is_contiguous(memory_format=torch.channels_last):
            rows = g.permute(0, 2, 3, 1).reshape(-1, c)                      # a view: (N H W, C)
            gb = torch.ones(1, rows.shape[0], dtype=torch.float32, device=g.device).matmul(rows.float()).view(c)
        else:
            gb = g.sum((0, 2, 3), dtype=torch.float32)
        return g, gb


class HeadConv1x1(nn.Conv2d):
    """nn.Conv2d(cin, cout, 1) (same parameters and state-dict keys) with the bias added -- and its gradient reduced -- by
    _AddChannelBias; used where cout is odd (the 65-bin height head)."""

    def forward(self, x):
        if self.bias is None or not x.is_cuda:
            return super().forward(x)
        return _AddChannelBias.apply(self._conv_forward(x, self.weight, None), self.bias)


def _depth_conv_stack(mid_channels, depth_channels, conv_in, downsample, use_dcn, use_aspp, aspp_mid_channels):
    layers = [BasicBlock(conv_in, mid_channels, downsample=downsample),
              BasicBlock(mid_channels, mid_channels), BasicBlock(mid_channels, mid_channels)]
    if use_aspp:
        layers.append(ASPP(mid_channels, mid_channels if aspp_mid_channels < 0 else aspp_mid_channels))
    if use_dcn:
        layers.append(DCN(mid_channels, mid_channels, kernel_size=3, padding=1, groups=4, im2col_step=128))
    head = HeadConv1x1 if depth_channels % 2 else nn.Conv2d
    layers.append(head(mid_channels, depth_channels, kernel_size=1, stride=1, padding=0))
    return nn.Sequential(*layers)


class _LiftNetBase(nn.Module):
    """Shared trunk of HeightNet and DepthNet: 3x3 reduce conv, camera-aware SE gate driven by the
    27-vector of get_mlp_input, 3 residual blocks [+ASPP] [+DCN] + 1x1 classifier."""

    def _build(self, in_channels, mid_channels, depth_channels, use_dcn, use_aspp, with_cp, stereo, bias,
               aspp_mid_channels):
        self.reduce_conv = nn.Sequential(nn.Conv2d(in_channels, mid_channels, 3, stride=1, padding=1),
                                         BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.bn = nn.BatchNorm1d(27)
        self.depth_mlp = Mlp(27, mid_channels, mid_channels)
        self.depth_se = SELayer(mid_channels)
        conv_in, downsample = mid_channels, None
        if stereo:
            conv_in += depth_channels
            downsample = nn.Conv2d(conv_in, mid_channels, 1, 1, 0)
            cv = []
            for _ in range(2):
                cv += [nn.Conv2d(depth_channels, depth_channels, 3, stride=2, padding=1), BatchNorm2d(depth_channels)]
            self.cost_volumn_net = nn.Sequential(*cv)
            self.bias = bias
        self._stack_args = (mid_channels, depth_channels, conv_in, downsample, use_dcn, use_aspp, aspp_mid_channels)
        self.with_cp = with_cp
        self.depth_channels = depth_channels

    # ---- stereo cost volume (depthnet.py:249-361 / :492-603) ------------------------------------
    @staticmethod
    def _mat_vec(m, comps):
        """(B,N,r,c) matrices applied to per-point vectors given as c component tensors (B,N,D,H,W) -> r tensors.
        Written with broadcast multiply-adds: the reference's `matmul` over (B,N,D,H,W,3,3)@(…,3,1) becomes a batched
        GEMM with B*N*D*H*W = 17.8 M batches at B = 3, which faults in the ROCm BLAS path (fine at B <= 2)."""
        mb = m[:, :, None, None, None]
        return [sum(mb[..., i, j] * comps[j] for j in range(m.shape[-1])) for i in range(m.shape[-2])]

    def gen_grid(self, metas, B, N, D, H, W, hi, wi):
        """Current-frame stereo frustum -> sampling grid in the adjacent frame's image (reference depthnet.py:249-305)."""
        pts = metas['frustum'] - metas['post_trans'].view(B, N, 1, 1, 1, 3)
        from .detector import small_inverse   # torch.inverse, or a closed form while a HIP graph is being captured
        x, y, z = self._mat_vec(small_inverse(metas['post_rots']), list(pts.unbind(-1)))
        x, y = x * z, y * z
        rots = metas['k2s_sensor'][:, :, :3, :3].contiguous()
        trans = metas['k2s_sensor'][:, :, :3, 3].contiguous()
        combine = rots.matmul(small_inverse(metas['intrins']))
        x, y, z = [v + trans[:, :, i, None, None, None] for i, v in enumerate(self._mat_vec(combine, [x, y, z]))]
        neg = z < 1e-3
        x, y, z = self._mat_vec(metas['intrins'], [x, y, z])
        x, y = x / z, y / z
        x, y = self._mat_vec(metas['post_rots'][..., :2, :2], [x, y])
        x = x + metas['post_trans'][:, :, 0, None, None, None]
        y = y + metas['post_trans'][:, :, 1, None, None, None]
        px = (x / (wi - 1.0) * 2.0 - 1.0).masked_fill(neg, -2)
        py = (y / (hi - 1.0) * 2.0 - 1.0).masked_fill(neg, -2)
        return torch.stack([px, py], dim=-1).view(B * N, D * H, W, 2)

    def calculate_cost_volumn(self, metas):
        prev, curr = metas['cv_feat_list']
        group = 4
        _, c, hf, wf = curr.shape
        hi, wi = hf * 4, wf * 4
        B, N, _ = metas['post_trans'].shape
        D, H, W, _ = metas['frustum'].shape
        grid = self.gen_grid(metas, B, N, D, H, W, hi, wi).to(curr.dtype)
        prev = prev.view(B * N, -1, H, W)
        curr = curr.view(B * N, -1, H, W)
        if self.use_hip_cost_volume and curr.is_cuda and c % group == 0 and c <= 1024 and D <= 256:
            return self._hip_cost_volume(prev, curr, grid, D, (c // group - 1) * group)
        cost = 0
        warped = None
        for f in range(curr.shape[1] // group):
            warped = F.grid_sample(prev[:, f * group:(f + 1) * group], grid, align_corners=True, padding_mode='zeros')
            diff = curr[:, f * group:(f + 1) * group].unsqueeze(2) - warped.view(B * N, -1, D, H, W)
            cost = cost + diff.abs().sum(dim=1)
        if not self.bias == 0:
            invalid = warped[:, 0].view(B * N, D, H, W) == 0
            cost = torch.where(invalid, cost + self.bias, cost)
        return (-cost).softmax(dim=1)

    use_hip_cost_volume = True

    def _hip_cost_volume(self, prev, curr, grid, n_depth, flag_channel):
        """The same cost volume in one HIP kernel (csrc/deform.hip: stereo_cost_volume_kernel); float32."""
        from . import _lib, mghs_op
        bn, c, h, w = curr.shape
        dev = curr.device
        prev_l = mghs_op._nchw_to_nhwc(prev.float().contiguous())
        curr_l = mghs_op._nchw_to_nhwc(curr.float().contiguous())
        grid = grid.float().contiguous()
        with torch.cuda.device(dev):
            out = torch.empty((bn, n_depth, h, w), dtype=torch.float32, device=dev)
            _lib.check(_lib.load().dhd_stereo_cost_volume(_lib.ptr(prev_l), _lib.ptr(curr_l), _lib.ptr(grid), bn, c, h, w, n_depth,
                                                          float(self.bias), flag_channel, _lib.ptr(out), _lib.stream_ptr(dev)),
                       'dhd_stereo_cost_volume')
        return out.to(curr.dtype)

    def _gated(self, x, mlp_input, mlp, se):
        return se(x, mlp(mlp_input)[..., None, None])

    def _depth_branch(self, x, mlp_input, stereo_metas):
        depth = self._gated(x, mlp_input, self.depth_mlp, self.depth_se)
        if stereo_metas is not None:
            if stereo_metas['cv_feat_list'][0] is None:
                bn, _, h, w = x.shape
                scale = float(stereo_metas['downsample']) / stereo_metas['cv_downsample']
                cv = torch.zeros((bn, self.depth_channels, int(h * scale), int(w * scale))).to(x)
            else:
                with torch.no_grad():
                    cv = self.calculate_cost_volumn(stereo_metas)
            depth = torch.cat([depth, self.cost_volumn_net(cv)], dim=1)
        if self.with_cp:
            return checkpoint(self.depth_conv, depth, use_reentrant=False)
        return self.depth_conv(depth)


class HeightNet(_LiftNetBase):
    def __init__(self, in_channels, mid_channels, depth_channels, use_dcn=True, use_aspp=True, with_cp=False,
                 stereo=False, bias=0.0, aspp_mid_channels=-1):
        super().__init__()
        self._build(in_channels, mid_channels, depth_channels, use_dcn, use_aspp, with_cp, stereo, bias,
                    aspp_mid_channels)
        self.depth_conv = _depth_conv_stack(*self._stack_args)

    def forward(self, x, mlp_input, stereo_metas=None):
        """x (B*N, C, fH, fW), mlp_input (B, N, 27) -> height logits (B*N, H, fH, fW)."""
        mlp_input = self.bn(mlp_input.reshape(-1, mlp_input.shape[-1]))
        x = self.reduce_conv(x)
        return self._depth_branch(x, mlp_input, stereo_metas)


class DepthNet(_LiftNetBase):
    def __init__(self, in_channels, mid_channels, context_channels, depth_channels, use_dcn=True, use_aspp=True,
                 with_cp=False, stereo=False, bias=0.0, aspp_mid_channels=-1):
        super().__init__()
        self._build(in_channels, mid_channels, depth_channels, use_dcn, use_aspp, with_cp, stereo, bias,
                    aspp_mid_channels)
        self.context_conv = nn.Conv2d(mid_channels, context_channels, kernel_size=1, stride=1, padding=0)
        self.context_mlp = Mlp(27, mid_channels, mid_channels)
        self.context_se = SELayer(mid_channels)
        self.depth_conv = _depth_conv_stack(*self._stack_args)

    def forward(self, x, mlp_input, stereo_metas=None):
        """-> (B*N, D + C_context, fH, fW): depth logits then context."""
        mlp_input = self.bn(mlp_input.reshape(-1, mlp_input.shape[-1]))
        x = self.reduce_conv(x)
        context = self.context_conv(self._gated(x, mlp_input, self.context_mlp, self.context_se))
        depth = self._depth_branch(x, mlp_input, stereo_metas)
        return torch.cat([depth, context], dim=1)
