#!/usr/bin/env python3
"""Benchmark of the DHD-S view-transform hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:
    MGHS : height argmax -> band, context re-layout, geometry + grouping (prepare),
           4-grid pooling forward (bev + low/mid/high), pooling backward (depth/context grads)
    SFA  : attention stage forward + backward on cat[x_2d, x_3d] (B,512,200,200)
at the DHD-S shapes of projects/configs/DHD/DHD-S.py (6 cameras 256x704 -> 16x44 features,
D=44, C=64, grids 200x200x{1,4,4,8}, samples_per_gpu=4).  Samples are independent, so ranks
shard them with no data-path collective ("weak" scaling); rank 0 prints ONE JSON line.

The line also carries
  roofline     : achieved HBM GB/s of the dominant kernel (mghs_stream_fwd), algorithmic bytes
                  / mean launch duration from HIP events on the launch stream
  cpu_baseline : the CPU oracle (numpy MGHS + torch-CPU SFA stage) timed on this box's host cores
                 on a small sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dhd_amd import _lib, dist as ddist, mghs_op, synthetic as syn  # noqa: E402
from dhd_amd.mix import channel_spatial_stage  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def pmc_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/<round>/pmc_summary.json:
    separate FETCH_SIZE / WRITE_SIZE runs of this same script, gfx950 correction applied).  PMC
    counters cannot be read from inside the run, so this is the stored measurement for the same
    per-GPU batch, or None."""
    best = None
    prof = os.path.join(ROOT, 'profiles')
    for rnd in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, rnd, 'pmc_summary.json')
        if os.path.exists(f):
            d = json.load(open(f))
            if d.get('samples_per_gpu') == batch and kernel in d.get('kernels', {}):
                best = d['kernels'][kernel]['hbm_bytes_per_launch']
    return best


def sfa_forward_traffic(batch):
    """Sum of the committed PMC traffic of the kernels one dhd_sfa_stage_forward call launches, or None."""
    calls = {'plane_mean_kernel': 1, 'fc_forward_kernel': 1, 'pack_weight6_kernel': 2, 'pw_gemm6_kernel<8,true,false,0>': 1,
             'pw_gemm6_kernel<8,false,true,0>': 1, 'stat_reduce_kernel': 2, 'bn_train_finalize_kernel': 2, 'blend2_bn_kernel': 1}
    parts = [pmc_traffic(k, batch) for k in calls]
    if any(p is None for p in parts):
        return None
    return int(sum(p * n for p, n in zip(parts, calls.values())))


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--batch', type=int, default=4, help='samples per GPU (DHD-S.py:243 samples_per_gpu=4)')
    p.add_argument('--no-sfa', action='store_true', help='time the MGHS part only')
    p.add_argument('--geometry', choices=['dhd-s', 'dhd-m', 'dhd-l'], default='dhd-s',
                   help='view-transform geometry of the hot path: DHD-S (D=44, 16x44), DHD-M (D=88), DHD-L (D=88, 32x88 from 512x1408)')
    p.add_argument('--workload', choices=['hotpath', 'e2e', 'occ_loss', 'ema'], default='hotpath',
                   help="hotpath: MGHS + SFA stage (default). e2e: the whole DHD-S detector (dense parts on MIOpen/hipBLASLt), "
                        "forward_train + backward + AdamW step, DDP over RCCL when --gpus > 1")
    p.add_argument('--amp', choices=['off', 'bf16', 'fp16'], default='off', help='autocast dtype of the dense modules (e2e)')
    p.add_argument('--model', choices=['dhd-s', 'dhd-m', 'dhd-l'], default='dhd-s',
                   help='e2e: DHD-S (single frame), DHD-M (temporal stereo) or DHD-L (Swin-B, 512x1408 images, temporal stereo)')
    p.add_argument('--no-ema', action='store_true', help='e2e: leave out the per-iteration weight EMA (MEGVIIEMAHook) of the configs')
    p.add_argument('--cpu-samples', type=int, default=2, help='samples for the CPU baseline leg (0 = skip)')
    return p.parse_args()


class HotPath:
    """Device-resident inputs + the exact C-ABI call sequence of MGHS.view_transform fwd/bwd."""

    GEOMETRY = {'dhd-s': ((256, 704), 1.0), 'dhd-m': ((256, 704), 0.5), 'dhd-l': ((512, 1408), 0.5)}  # input size, depth step

    def __init__(self, dev, batch, seed, with_sfa, geometry='dhd-s'):
        self.dev, self.B = dev, batch
        cfg = self.cfg = syn.dhd_s_config()
        (ih, iw), dstep = self.GEOMETRY[geometry]
        cfg['input_size'] = (ih, iw)
        cfg['grid_config'] = dict(cfg['grid_config'], depth=[1.0, 45.0, dstep])
        N, D, fh, fw, C = 6, int(round(44 / dstep)), ih // 16, iw // 16, 64
        self.dims = (N, D, fh, fw, C)
        self.calib_np = syn.make_calibration(seed, batch, N, cfg['input_size'])
        depth, feat, hidx = syn.lift_inputs(seed + 1, batch, N, D, fh, fw, C, 65)
        self.inputs_np = (depth, feat, hidx)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        u = torch.linspace(0, iw - 1, fw, dtype=torch.float)
        v = torch.linspace(0, ih - 1, fh, dtype=torch.float)
        d = torch.arange(1.0, 45.0, dstep, dtype=torch.float)
        s2e, _, intrin, post_rot, post_tran, bda = [t(a) for a in self.calib_np]
        self.calib, self._keep = mghs_op.make_calib(s2e, intrin, post_rot, post_tran, bda,
                                                    (u.to(dev), v.to(dev), d.to(dev)))
        full = {'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 6.4]}
        grids = [mghs_op.grid_from_cfg(g) for g in (full, cfg['mask_1_grid'], cfg['mask_2_grid'], cfg['mask_3_grid'])]
        self.plan = mghs_op.Plan(batch, N, D, fh, fw, C, grids)
        self.depth, self.feat = t(depth), t(feat)
        self.height = t(syn.height_probs_from_index(hidx, 65))
        self.ws = self.plan.new_workspace(dev)
        g = torch.Generator(device='cpu').manual_seed(seed)
        self.out_grads = [torch.randn(s, generator=g).to(dev) for s in self.plan.out_shapes()]
        self.with_sfa = with_sfa
        if with_sfa:
            torch.manual_seed(seed)
            self.stage = channel_spatial_stage(512).to(dev).train()
            self.stage_params = list(self.stage.parameters())
            self.x = torch.randn(batch, 512, 200, 200, generator=g).to(dev).requires_grad_()
            self.gy = torch.randn(batch, 256, 200, 200, generator=g).to(dev)
        self.ev = []  # (start, end) HIP events around the dominant kernel, one pair per timed step
        self.ev_sfa = []  # (start, after forward, after backward) events around the SFA stage operator
        # algorithmic bytes of the forward pooling per launch (fused form): dense outputs
        # written once + depth read once + context read once.  The streaming kernel is charged with ALL
        # of them although depth/context are read by the gather kernel before it (conservative by 1%).
        self.pool_fwd_bytes = batch * (4 * C * 17 * 200 * 200 + 4 * N * D * fh * fw + 4 * N * fh * fw * C)

    def step(self, record):
        cfg = self.cfg
        band = mghs_op.height_band(self.height, cfg['height_range'], cfg['mask_range'])
        feat_nhwc = mghs_op._nchw_to_nhwc(self.feat)
        mghs_op.prepare(self.plan, self.calib, band, self.ws)
        if record:
            # HIP events on the launch stream, around the streaming kernel only
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            outs = mghs_op.pool_forward_phases(self.plan, self.depth, feat_nhwc, self.ws, between=e0.record)
            e1.record()
            self.ev.append((e0, e1))
        else:
            outs = mghs_op.pool_forward(self.plan, self.depth, feat_nhwc, self.ws)
        dg, fg = mghs_op.pool_backward(self.plan, self.depth, feat_nhwc, self.out_grads, self.ws)
        fg_nchw = mghs_op._nhwc_to_nchw(fg)
        if self.with_sfa:
            self.x.grad = None
            for prm in self.stage_params:  # optimizer.zero_grad(set_to_none=True), the PyTorch default
                prm.grad = None
            if record:
                # HIP events on the launch stream around the stage's forward and backward calls
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                y = self.stage(self.x)
                e[1].record()
                y.backward(self.gy)
                e[2].record()
                self.ev_sfa.append(e)
            else:
                y = self.stage(self.x)
                y.backward(self.gy)
        return outs, dg, fg_nchw


class EndToEnd:
    """DHD-S exactly as projects/configs/DHD/DHD-S.py:42-155 (random init, synthetic 6-camera batch,
    SURVEY.md 8d config 2): forward_train -> sum of the four losses -> backward -> grad clip 5 -> AdamW."""

    def __init__(self, dev, batch, seed, world, amp, model='dhd-s', ema=True):
        import dhd_amd
        from dhd_amd.detector import dhd_l_model_cfg, dhd_m_model_cfg, dhd_s_model_cfg
        torch.manual_seed(seed)
        # dhd-m: DHD-M.py (DHD_stereo: key frame + 1 adjacent + 1 stereo reference frame, D = 88, SFA with C = 512)
        # dhd-l: DHD-L.py (the same wiring on a Swin-B backbone, 512 x 1408 images, 32 x 88 feature maps with 512 channels)
        frames = 1 if model == 'dhd-s' else 3
        cfg = {'dhd-s': dhd_s_model_cfg, 'dhd-m': dhd_m_model_cfg, 'dhd-l': dhd_l_model_cfg}[model]()
        self.model = dhd_amd.build_detector(cfg).to(dev).train()
        if model == 'dhd-l':
            self.model.img_backbone.init_weights()   # trunc-normal init of the Swin linears / bias tables (swin.py:876-890)
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        self.n_params = sum(p.numel() for p in self.params)
        self.net = self.model
        if world > 1:
            self.net = torch.nn.parallel.DistributedDataParallel(self.model, device_ids=[dev.index], bucket_cap_mb=64,
                                                                 gradient_as_bucket_view=True)
        self.opt = torch.optim.AdamW(self.params, lr=2e-4, weight_decay=1e-2, fused=True)  # DHD-S.py:262
        # custom_hooks of all three configs (DHD-S.py:272-278): weight EMA after every iteration
        self.ema = dhd_amd.ModelEMA(self.model, 0.9990, updates=10560) if ema else None
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        N, (H, W) = 6, ((512, 1408) if model == 'dhd-l' else (256, 704))
        per = [syn.make_calibration(seed + 7 * f, batch, N, (H, W)) for f in range(frames)]
        calib = [t(np.concatenate([p[k] for p in per], 1)) for k in range(5)] + [t(per[0][5])]
        for f in range(1, frames):  # the ego vehicle moves 0.8 m per frame
            calib[1][:, f * N:(f + 1) * N, 0, 3] += 0.8 * f
        g = torch.Generator(device='cpu').manual_seed(seed)
        imgs = torch.randn(batch, N * frames, 3, H, W, generator=g).to(dev)
        sel = torch.rand(batch, N, H, W, generator=g) < 0.02
        self.kw = dict(
            img_inputs=[imgs] + calib,
            gt_depth=torch.where(sel, 1 + 44 * torch.rand(batch, N, H, W, generator=g), torch.zeros(())).to(dev),
            gt_height=torch.where(sel, -1 + 6.4 * torch.rand(batch, N, H, W, generator=g), torch.zeros(())).to(dev),
            voxel_semantics=torch.randint(0, 18, (batch, 200, 200, 16), generator=g).to(dev),
            mask_camera=(torch.rand(batch, 200, 200, 16, generator=g) < 0.3).to(dev))
        self.amp = {'off': None, 'bf16': torch.bfloat16, 'fp16': torch.float16}[amp]
        self.scaler = torch.amp.GradScaler('cuda') if amp == 'fp16' else None
        self.B = batch

    def step(self, record):
        self.opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=self.amp, enabled=self.amp is not None):
            losses = self.net(return_loss=True, **self.kw)
            loss = sum(losses.values())
        if self.scaler is not None:
            self.scaler.scale(loss).backward()
            self.scaler.unscale_(self.opt)
            torch.nn.utils.clip_grad_norm_(self.params, 5.0)
            self.scaler.step(self.opt)
            self.scaler.update()
        else:
            loss.backward()
            torch.nn.utils.clip_grad_norm_(self.params, 5.0)  # DHD-S.py:263
            self.opt.step()
        if self.ema is not None:
            self.ema.update(None, self.model)
        return loss


def run_e2e(a, rank, world, dev):
    job = EndToEnd(dev, a.batch, 1000 + rank, world, a.amp, a.model, not a.no_ema)
    for _ in range(a.warmup):
        job.step(False)

    def fence():
        torch.cuda.synchronize()
        ddist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = job.step(True)
    fence()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps(dict(
            metric=f'samples/sec (6-cam fwd+bwd) {a.model.upper()} end-to-end', value=a.batch * world * a.steps / elapsed, unit='samples/s',
            n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps, higher_is_better=True,
            scaling='weak', vs_baseline=None, dtype={'off': 'f32', 'bf16': 'bf16', 'fp16': 'f16'}[a.amp], data='synthetic',
            config=dict(workload={'dhd-m': 'DHD-M (DHD_stereo: key + adjacent + stereo reference frame, D=88) whole detector: ResNet-50 + FPN',
                                  'dhd-l': 'DHD-L (configs[3]: DHD_stereo on 512x1408 images, D=88, 32x88 feature maps) whole detector: '
                                           'Swin-B + FPN_LSS',
                                  'dhd-s': 'DHD-S (configs[1]/[2]) whole detector: ResNet-50 + FPN'}[a.model] +
                                 ', MGHS (HIP), BEV encoder, 3 UNets, SFA (HIP stage), predictor + losses (HIP); '
                                 'forward_train + backward + grad-clip + AdamW' + ('' if a.no_ema else ' + weight EMA (HIP)') + '; random init',
                        samples_per_gpu=a.batch, global_batch=a.batch * world, params=job.n_params,
                        parallelism=f'DDP x{world} (RCCL bucketed all-reduce overlapped with backward)' if world > 1 else 'single GPU',
                        final_loss=float(loss)))), flush=True)
    ddist.shutdown()


def cpu_baseline(hp, n_samples):
    """Oracle timing on the host: numpy MGHS view_transform fwd+bwd (the reference's op sequence,
    4x geometry + 4x sort + pool + permute) and, for the SFA stage, the reference formula in
    torch-CPU fp32 with all host threads."""
    from oracle import mghs_oracle as O  # checker / baseline only
    cfg = hp.cfg
    n = min(n_samples, hp.B)
    calib = [a[:n] for a in hp.calib_np]
    depth, feat, hidx = (a[:n * 6] for a in hp.inputs_np)
    t0 = time.perf_counter()
    outs = O.view_transform(cfg, calib, depth, feat, hidx)
    gr = [np.ones_like(o) for o in outs]
    O.view_transform_backward(cfg, calib, depth, feat, hidx, gr)
    t_mghs = time.perf_counter() - t0
    t_sfa = 0.0
    if hp.with_sfa:
        torch.set_num_threads(os.cpu_count())
        st = channel_spatial_stage(512)
        x = torch.randn(n, 512, 200, 200, requires_grad=True)
        t0 = time.perf_counter()
        xb, xv = torch.split(x, 256, dim=1)
        a1 = st.fc(x.mean(-1).mean(-1))[:, :, None, None]
        xb1, xv1 = a1 * xb, (1 - a1) * xv
        a2 = torch.sigmoid(st.spacial_leanring(xb1 + xv1))
        (a2 * xb1 + (1 - a2) * xv1).sum().backward()
        t_sfa = time.perf_counter() - t0
    return dict(value=n / (t_mghs + t_sfa), unit='samples/s', cores=os.cpu_count(), kind='port',
                sample=f'{n} sample(s) of the same workload: oracle/mghs_oracle.py view_transform fwd+bwd '
                       f'({t_mghs:.2f} s, numpy, mostly 1 thread)' +
                       (f' + SFA stage fwd+bwd in torch-CPU fp32 ({t_sfa:.2f} s, {os.cpu_count()} threads)' if hp.with_sfa else ''))


def run_occ_loss(a, rank, world, dev):
    """The caller row after the hot path: predictor.loss on B x 200x200x16 voxels x 18 classes,
    forward + backward through dhd_occ_loss_forward/backward.  Roofline: the gradient pass (reads and writes the
    (M,18) logits matrix once each), timed with HIP events on the launch stream."""
    from dhd_amd import occ_loss
    from dhd_amd.detector import NUSC_CLASS_FREQUENCIES
    m = a.batch * 200 * 200 * 16
    g = torch.Generator(device='cpu').manual_seed(2000 + rank)
    logits = torch.randn(m, 18, generator=g).to(dev).requires_grad_()
    labels = torch.randint(0, 18, (m,), generator=g).to(torch.uint8).to(dev)
    cam = (torch.rand(m, generator=g) < 0.3).to(torch.uint8).to(dev)
    cw = torch.from_numpy((1 / np.log(NUSC_CLASS_FREQUENCIES + 0.001)).astype(np.float32)).to(dev)
    ones = torch.ones(3, device=dev)
    ev = []

    def step(record):
        logits.grad = None
        l = occ_loss._OccLosses.apply(logits, labels, cam, cw, 255, 17)
        occ_loss._OccLosses.events = ev if record else None  # HIP events right around the gradient kernel's launch
        l.backward(ones)
        occ_loss._OccLosses.events = None

    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
        grad_bytes = m * (2 * 18 * 4 + 2)  # logits read + gradient written + label and mask bytes
        achieved = grad_bytes / (kern_ms * 1e-3) / 1e9
        line = dict(metric='samples/sec (fwd+bwd) DHD-S occupancy-head losses', value=a.batch * world * a.steps / elapsed, unit='samples/s',
                    n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps, higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                    config=dict(workload='DHD-S predictor.loss (class-balanced masked CE + sem_scal + geo_scal) on B x 200x200x16 voxels x 18 '
                                         'classes, forward + backward', samples_per_gpu=a.batch, global_batch=a.batch * world,
                                parallelism=f'sample-sharded x{world}, no data-path collective'),
                    roofline=dict(bound='hbm', kernel='occ_loss_grad', achieved=achieved,
                                  peak=HBM_PEAK_GBPS, unit='GB/s', frac=achieved / HBM_PEAK_GBPS, traffic=pmc_traffic('occ_loss_grad', a.batch),
                                  launch_ms=kern_ms, algorithmic_bytes=grad_bytes))
        if world == 1 and a.cpu_samples > 0:
            from oracle import mghs_oracle as O  # checker / CPU baseline only
            n1 = 200 * 200 * 16
            z, t, c = logits[:n1].detach().cpu().numpy(), labels[:n1].cpu().numpy().astype(np.int64), cam[:n1].cpu().numpy()
            t0 = time.perf_counter()
            O.occ_losses(z, t, c, cw.cpu().numpy())
            dt = time.perf_counter() - t0
            line['cpu_baseline'] = dict(value=1.0 / dt, unit='samples/s', cores=1, kind='port',
                                        sample='1 sample (640 000 voxels), oracle.occ_losses forward only (numpy float64, 1 thread)')
        print(json.dumps(line), flush=True)
    ddist.shutdown()


def run_ema(a, rank, world, dev):
    """The caller row after the optimizer step: MEGVIIEMAHook.after_train_iter on the DHD-S
    detector's whole state dict.  A step = one EMA update; 12 algorithmic bytes per value (EMA read + write,
    model read).  The reference's own formulation (two eager ops per state-dict entry) is timed beside it."""
    import dhd_amd
    from dhd_amd.detector import dhd_s_model_cfg
    from dhd_amd.ema import ModelEMA
    torch.manual_seed(3000 + rank)
    model = dhd_amd.build_detector(dhd_s_model_cfg()).to(dev).train()
    ema = ModelEMA(model, 0.9990, updates=10560)
    n_val = sum(v.numel() for v in model.state_dict().values() if v.dtype.is_floating_point)
    n_tensors = sum(1 for v in model.state_dict().values() if v.dtype.is_floating_point)
    ev = []

    def step(record):
        ema.events = ev if record else None  # HIP events right around the kernel's launch
        ema.update(None, model)
        ema.events = None

    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    torch.cuda.synchronize(); ddist.barrier(); torch.cuda.synchronize()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0, dev)

    def eager():  # ema.py:55-59 verbatim in behaviour
        with torch.no_grad():
            msd = model.state_dict()
            for k, v in ema.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= 0.999
                    v += (1.0 - 0.999) * msd[k].detach()
    for _ in range(2):
        eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eager()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / 5 * 1e3
    if rank == 0:
        kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
        achieved = 12 * n_val / (kern_ms * 1e-3) / 1e9
        print(json.dumps(dict(
            metric='EMA updates/sec, DHD-S state dict', value=world * a.steps / elapsed, unit='updates/s', n_gpus=world, steps=a.steps,
            warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps, higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32',
            data='synthetic', config=dict(workload=f'MEGVIIEMAHook.after_train_iter on the DHD-S detector: {n_val} float32 values in '
                                                    f'{n_tensors} tensors, one launch', parallelism=f'replicas x{world}'),
            roofline=dict(bound='hbm', kernel='ema_update_kernel', achieved=achieved, peak=HBM_PEAK_GBPS, unit='GB/s',
                          frac=achieved / HBM_PEAK_GBPS, traffic=pmc_traffic('ema_update_kernel', a.batch), launch_ms=kern_ms,
                          algorithmic_bytes=12 * n_val),
            pytorch_eager_ms=eager_ms)), flush=True)
    ddist.shutdown()


def main():
    a = parse()
    rank, local, world = ddist.env_world()
    if world != a.gpus and world > 1:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    ddist.init_from_env(backend='nccl', device=dev)  # RCCL; used only for the barrier / MAX around the timed region
    _lib.load()
    if a.workload == 'e2e':
        return run_e2e(a, rank, world, dev)
    if a.workload == 'ema':
        return run_ema(a, rank, world, dev)
    if a.workload == 'occ_loss':
        return run_occ_loss(a, rank, world, dev)
    hp = HotPath(dev, a.batch, 1000 + rank, not a.no_sfa, a.geometry)

    for _ in range(a.warmup):
        hp.step(False)

    def fence():
        torch.cuda.synchronize()
        ddist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        hp.step(True)
    fence()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0, dev)

    if rank == 0:
        kern_ms = float(np.mean([s.elapsed_time(e) for s, e in hp.ev]))
        achieved = hp.pool_fwd_bytes / (kern_ms * 1e-3) / 1e9
        line = dict(
            metric=f'samples/sec (6-cam fwd+bwd) {a.geometry.upper()} view-transform hot path', value=a.batch * world * a.steps / elapsed,
            unit='samples/s', n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps,
            higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
            config=dict(workload=('DHD-S (configs[1])' if a.geometry == 'dhd-s' else a.geometry.upper() + ' geometry (configs[3-4])') + ' hot path: MGHS 4-grid lift-splat fwd+bwd incl. geometry/grouping'
                                 + ('' if a.no_sfa else ' + SFA attention stage fwd+bwd') +
                                 f'; geometry {a.geometry}: 6 cams -> {hp.dims[2]}x{hp.dims[3]}, D={hp.dims[1]}, C=64, grids 200x200x{{1,4,4,8}}; dense backbone/encoder convs not in the step',
                        samples_per_gpu=a.batch, global_batch=a.batch * world, parallelism=f'sample-sharded x{world}, no data-path collective'),
            roofline=dict(bound='hbm', kernel='mghs_stream_fwd', achieved=achieved, peak=HBM_PEAK_GBPS, unit='GB/s',
                          frac=achieved / HBM_PEAK_GBPS, traffic=pmc_traffic('mghs_stream_fwd', a.batch), launch_ms=kern_ms,
                          algorithmic_bytes=hp.pool_fwd_bytes))
        if hp.ev_sfa:
            # second roofline, for the SFA stage operator as a whole (a dozen kernels per call): SURVEY 8(d) gives its forward
            # algorithmic traffic as x read twice + u/out written + the two 1x1 convs reading and writing (B,C,H,W) once each
            fwd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in hp.ev_sfa]))
            bwd_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in hp.ev_sfa]))
            c, hw = 256, 200 * 200
            fwd_bytes = a.batch * 4 * hw * (2 * 2 * c + 2 * c + 4 * c)
            gemm_flop = 2.0 * c * c * hw * a.batch
            line['roofline_sfa_stage'] = dict(
                bound='hbm', kernel='dhd_sfa_stage_forward (plane_mean, fc, 2 x pw_gemm6, stat reductions, blend2_bn)',
                achieved=fwd_bytes / (fwd_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBPS, unit='GB/s',
                frac=fwd_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, traffic=sfa_forward_traffic(a.batch), launch_ms=fwd_ms,
                algorithmic_bytes=fwd_bytes,
                backward_ms=bwd_ms, gemm_tflops_fp32_equivalent=6 * gemm_flop / ((fwd_ms + bwd_ms) * 1e-3) / 1e12,
                note='float32 GEMMs computed as 6 bf16 MFMA products each (exact three-way split); f32-MFMA peak is 157 TFLOP/s')
        if world == 1 and a.cpu_samples > 0:
            line['cpu_baseline'] = cpu_baseline(hp, a.cpu_samples)
        print(json.dumps(line), flush=True)
    ddist.shutdown()


if __name__ == '__main__':
    main()
