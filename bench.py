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
  roofline            : achieved HBM GB/s of the dominant kernel (mghs_stream_fwd), algorithmic bytes
                         / mean launch duration from HIP events on the launch stream
  roofline_bwd        : the same for the pooling backward (mghs_stream_bwd + mghs_pixel_bwd, 177.7 MB/sample)
  roofline_operator   : the standalone operator drop-in (dhd_bev_pool_v2_forward/backward on the full grid with
                        reference-style index lists, 13.8 MB/sample), timed after the main loop
  roofline_sfa_stage  : the SFA stage operator's forward as a whole (328 MB/sample)
  cpu_baseline        : the torch-CPU twin of the reference's op sequence (oracle/mghs_torch_cpu.py) on this box's
                        host cores, all threads, 3 warm-ups + median of 5, B=1 and B=4 (rank 0, N=1 only)
  e2e                 : the whole DHD-S detector fwd+bwd+optimizer (fp32 and fp16 autocast), 3 warm-ups + 5 steps;
                        under DDP for N>1, with the exposed all-reduce time (DDP step - no_sync step)
`--gpus N` without a launcher environment re-executes itself under torch.distributed.run with N ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# MIOpen's user find-db (dhd_amd/miopen_db/: built once by experiments/miopen_find_job.py -- an exhaustive FIND over the convolution
# problems of the DHD-S step on an MI355X; this image ships no gfx950 find-db, so without it every convolution runs the solver a
# heuristic picks).  Must be in the environment before MIOpen initialises; DHD_NO_MIOPEN_DB=1 leaves MIOpen on its defaults (A/B).
# A FRESH scratch copy per run (tempfile.mkdtemp: private, unpredictable name) is used so that MIOpen's own write-backs never touch the
# tracked files and no run reads what an earlier run -- or an earlier version of the tracked db -- left behind; ranks and child
# processes inherit the path through the environment.  The db's content hash goes into the JSON line (`miopen_db`).
MIOPEN_DB = dict(source=None, sha256_16=None)
if 'MIOPEN_USER_DB_PATH' not in os.environ and not os.environ.get('DHD_NO_MIOPEN_DB'):
    _src = os.path.join(ROOT, 'dhd_amd', 'miopen_db')
    if os.path.isdir(_src) and any(f.endswith('.ufdb.txt') for f in os.listdir(_src)):
        import atexit
        import hashlib
        import shutil
        import tempfile
        _dst = tempfile.mkdtemp(prefix='dhd_amd_miopen_db_')
        _h = hashlib.sha256()
        for _f in sorted(os.listdir(_src)):
            if _f.endswith('.txt'):
                shutil.copy(os.path.join(_src, _f), _dst)
                _h.update(_f.encode() + b'\0' + open(os.path.join(_src, _f), 'rb').read())
        os.environ['MIOPEN_USER_DB_PATH'] = _dst
        os.environ['DHD_MIOPEN_DB_SHA'] = _h.hexdigest()[:16]
        atexit.register(lambda d=_dst, pid=os.getpid(): os.getpid() == pid and shutil.rmtree(d, ignore_errors=True))
if os.environ.get('DHD_MIOPEN_DB_SHA'):
    MIOPEN_DB = dict(source='dhd_amd/miopen_db (scratch copy)', sha256_16=os.environ['DHD_MIOPEN_DB_SHA'])
elif 'MIOPEN_USER_DB_PATH' in os.environ:
    MIOPEN_DB = dict(source='MIOPEN_USER_DB_PATH from the environment', sha256_16=None)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dhd_amd import _lib, dist as ddist, mghs_op, synthetic as syn  # noqa: E402
from dhd_amd.mix import channel_spatial_stage  # noqa: E402


def event_mean(pairs):
    """Mean of HIP-event intervals in ms, without the samples a host stall inflated.  An event pair brackets asynchronous
    launches: when the host pauses between recording the first event and issuing the kernel (a Python garbage collection took
    36-39 ms at that exact place in two runs of three, turning a 26 us average into 1.9 ms), the stream idles and the interval
    shows the pause, not the kernel.  Samples above three times the median are such pauses and are left out (the timed
    regions also run with the collector off, see `no_gc`)."""
    v = np.array([a.elapsed_time(b) for a, b in pairs], dtype=np.float64)
    if v.size == 0:
        return float('nan')
    keep = v <= 3.0 * np.median(v)
    EVENT_DROPS['dropped'] += int((~keep).sum())
    EVENT_DROPS['total'] += int(v.size)
    return float(v[keep].mean())


# how many event samples event_mean() has left out in this process (reported in the JSON line as `event_samples_dropped`)
EVENT_DROPS = {'dropped': 0, 'total': 0}


class no_gc:
    """Python's cyclic collector off inside a timed region (collected once on entry), restored on exit."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()

    def __exit__(self, *a):
        import gc
        if self.was:
            gc.enable()

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def kernel_source_sha256():
    """SHA-256 over the kernel sources the library is built from (csrc/*.hip, *.h and the C header): the identity a
    committed PMC measurement is valid for."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'dhd_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'dhd_amd', 'csrc', '*.h'))
                   + [os.path.join(ROOT, 'include', 'dhd_amd.h')])
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0' + open(f, 'rb').read())
    return h.hexdigest()


def pmc_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/<round>/pmc_summary.json:
    separate FETCH_SIZE / WRITE_SIZE runs of this same script, gfx950 correction applied).  PMC counters cannot be
    read from inside the run, so this is the stored measurement -- returned only if it was collected for the same
    per-GPU batch AND from the kernel sources this library was built from (`source_sha256`); otherwise None."""
    best = None
    prof = os.path.join(ROOT, 'profiles')
    sha = kernel_source_sha256()
    for rnd in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, rnd, 'pmc_summary.json')
        if os.path.exists(f):
            d = json.load(open(f))
            if d.get('samples_per_gpu') == batch and d.get('source_sha256') == sha and kernel in d.get('kernels', {}):
                best = d['kernels'][kernel]['hbm_bytes_per_launch']
    return best


def pmc_summary(batch):
    """kernel -> HBM bytes per launch from the committed PMC summary that matches this batch and these kernel sources, or {}."""
    prof = os.path.join(ROOT, 'profiles')
    sha = kernel_source_sha256()
    best = {}
    for rnd in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, rnd, 'pmc_summary.json')
        if os.path.exists(f):
            d = json.load(open(f))
            if d.get('samples_per_gpu') == batch and d.get('source_sha256') == sha:
                best = {k: v['hbm_bytes_per_launch'] for k, v in d.get('kernels', {}).items()}
    return best


def sfa_forward_traffic(batch, terms=2):
    """Sum of the committed PMC traffic of the kernels one dhd_sfa_stage_forward call launches (training mode; `terms` = 2:
    the default bf16x3 GEMMs, 3: bf16x6), or None.  The two GEMMs are found by their template arguments <terms, tiles, K steps,
    TWO_IN, RELU, EPI, ...>: conv1 = two inputs / no ReLU / epilogue 0, conv2 = one input / ReLU / epilogue 0.  (The counter
    passes run both precisions, so the small finalize kernel's figure is an average over them: +-5 MB of ~1.8 GB.)"""
    k = pmc_summary(batch)
    import re
    if terms == 2:   # default precision: the one-CU-per-tile kernels <K steps, waves, TWO_IN, RELU, EPI, RECORD, ...>
        conv1 = [v for n, v in k.items() if re.match(r'pw_gemm_cu_kernel<\d+,\d+,true,false,0,', n)]
        conv2 = [v for n, v in k.items() if re.match(r'pw_gemm_cu_kernel<\d+,\d+,false,true,0,', n)]
    else:
        conv1 = [v for n, v in k.items() if re.match(rf'pw_gemm_res_kernel<{terms},\d+,\d+,true,false,0,', n)]
        conv2 = [v for n, v in k.items() if re.match(rf'pw_gemm_res_kernel<{terms},\d+,\d+,false,true,0,', n)]
    calls = {'plane_mean_pack_kernel': 1, 'fc_forward_kernel': 1, 'bn_stats_finalize_kernel': 2, 'blend2_bn_kernel': 1}
    if len(conv1) != 1 or len(conv2) != 1 or any(n not in k for n in calls):
        return None
    return int(conv1[0] + conv2[0] + sum(k[n] * m for n, m in calls.items()))


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
    p.add_argument('--layout', default=None,
                   help="e2e: memory format of the dense modules: 'nchw', 'channels_last' or 'channels_last:<sub-module>,...' "
                        "(default: $DHD_E2E_LAYOUT, else the measured-best stacks for the half-precision DHD-S step and nchw otherwise)")
    p.add_argument('--no-ema', action='store_true', help='e2e: leave out the per-iteration weight EMA (MEGVIIEMAHook) of the configs')
    p.add_argument('--cpu-samples', type=int, default=4, help='largest batch of the CPU baseline leg (0 = skip)')
    p.add_argument('--deterministic', action='store_true', help='hotpath: order the entries of every voxel by point id (bit-reproducible forward)')
    p.add_argument('--no-graph', action='store_true', help='e2e: issue the step eagerly instead of replaying it as one HIP graph (N = 1)')
    p.add_argument('--dist-backend', choices=['nccl', 'gloo'], default='nccl',
                   help="process-group backend; 'gloo' lets several ranks share one GPU (testing the N > 1 code path on a one-GPU box)")
    p.add_argument('--no-e2e', action='store_true', help='hotpath: leave out the end-to-end DHD-S sub-record')
    p.add_argument('--no-operator', action='store_true', help='hotpath: leave out the standalone bev_pool_v2 operator timing')
    p.add_argument('--sfa-gemm', choices=['bf16x3', 'bf16x6', 'f32'], default=None,
                   help="hotpath: precision of the SFA stage's C x C GEMMs in the main timed loop (default: the library default, bf16x3; "
                        "the bf16x6 step time is reported beside it either way)")
    p.add_argument('--bucket-mb', type=int, default=64, help='e2e under DDP: gradient bucket size (MB)')
    p.add_argument('--no-ddp-static-graph', dest='ddp_static_graph', action='store_false',
                   help='e2e under DDP: DistributedDataParallel(static_graph=False) (default: static_graph=True -- the step has the same '
                        'autograd graph every iteration; exercised with two ranks over gloo incl. the no_sync() comparison)')
    p.add_argument('--repeats', type=int, default=7,
                   help='hotpath: the timed loop of EXACTLY --steps steps is run this many times in the process (barrier + synchronize on both '
                        'sides each time); ms_per_step / value are the median loop, min / max / all are reported beside it')
    p.add_argument('--fresh-procs', type=int, default=2,
                   help='hotpath, N = 1: repeat the whole measurement in this many fresh processes (other allocator placement of the 0.7 GB '
                        'outputs: one build differs by up to 11 % on the writer between two processes on one box)')
    p.add_argument('--no-dhdl', action='store_true', help='hotpath: leave out the MGHS-only record at the DHD-L geometry (configs[3]/[4], B = 2)')
    p.add_argument('--pmc-pass', action='store_true',
                   help='hotpath: only what a counter-collection pass needs -- one timed loop, the bf16x6 loop and the three-step operator; no '
                        'DHD-L / fresh-process children, no half-precision or fused-operator launches (they share kernel names with the '
                        'hot path at other sizes and would be averaged into its per-launch traffic)')
    p.add_argument('--child', action='store_true', help='(internal) a --fresh-procs child: print the timing statistics only')
    p.add_argument('--ddp-graph', dest='ddp_graph', action='store_true', default=None,
                   help='e2e under DDP: capture the whole step, RCCL all-reduces included, into a HIP graph.  Default (neither flag): TRY it '
                        'when the backend is RCCL ("nccl"), after every eager number of the record has been measured, under a watchdog '
                        '(--ddp-graph-timeout); on a capture error / timeout the record keeps the eager step and says why')
    p.add_argument('--no-ddp-graph', dest='ddp_graph', action='store_false', help='e2e under DDP: do not attempt the graph capture')
    p.add_argument('--ddp-graph-timeout', type=float, default=240.0,
                   help='seconds the DDP graph attempt (capture + replays, all legs) may take before the line is printed without it')
    p.add_argument('--stub-model', action='store_true',
                   help='(test only) --workload e2e on a few-kilobyte stand-in detector that runs on CPU tensors: exercises the launcher, the '
                        'process group, DDP and the record plumbing without a GPU (tests/test_distributed.py); the line is marked "stub"')
    a = p.parse_args()
    if a.pmc_pass:
        a.no_dhdl, a.fresh_procs, a.repeats, a.no_e2e, a.cpu_samples = True, 0, 1, True, 0
    return a


class HotPath:
    """Device-resident inputs + the exact C-ABI call sequence of MGHS.view_transform fwd/bwd."""

    GEOMETRY = {'dhd-s': ((256, 704), 1.0), 'dhd-m': ((256, 704), 0.5), 'dhd-l': ((512, 1408), 0.5)}  # input size, depth step

    def __init__(self, dev, batch, seed, with_sfa, geometry='dhd-s', deterministic=False, sfa_gemm=None):
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
        # the context gradient comes back in tran_feat's own (B*N, C, fH, fW) layout (DHD_MGHS_FEAT_GRAD_NCHW)
        self.plan = mghs_op.Plan(batch, N, D, fh, fw, C, grids, deterministic=deterministic, feat_grad_nchw=True)
        self.depth, self.feat = t(depth), t(feat)
        self.height = t(syn.height_probs_from_index(hidx, 65))
        self.ws = self.plan.new_workspace(dev)
        g = torch.Generator(device='cpu').manual_seed(seed)
        self.out_grads = [torch.randn(s, generator=g).to(dev) for s in self.plan.out_shapes()]
        self.with_sfa = with_sfa
        if with_sfa:
            torch.manual_seed(seed)
            self.stage = channel_spatial_stage(512).to(dev).train()
            self.stage.gemm = sfa_gemm     # None: the library default (bf16x3); 'bf16x6': float32-level products
            self.stage_params = list(self.stage.parameters())
            self.x = torch.randn(batch, 512, 200, 200, generator=g).to(dev).requires_grad_()
            self.gy = torch.randn(batch, 256, 200, 200, generator=g).to(dev)
        self.ev = []  # (start, end) HIP events around the dominant kernel, one pair per timed step
        self.ev_bwd = []  # (start, end) around dhd_mghs_backward (mghs_stream_bwd + mghs_pixel_bwd)
        self.ev_sfa = []  # (start, after forward, after backward) events around the SFA stage operator
        # algorithmic bytes of the forward pooling per launch (fused form): dense outputs
        # written once + depth read once + context read once.  The streaming kernel is charged with ALL
        # of them although depth/context are read by the gather kernel before it (conservative by 1%).
        self.pool_fwd_bytes = batch * (4 * C * 17 * 200 * 200 + 4 * N * D * fh * fw + 4 * N * fh * fw * C)
        # backward: out_grad read once + depth / context read + depth_grad / feat_grad written
        self.pool_bwd_bytes = batch * (4 * C * 17 * 200 * 200 + 2 * (4 * N * D * fh * fw + 4 * N * fh * fw * C))

    def make_events(self, n_steps):
        """HIP events of the timed loop, created BEFORE it (hipEventCreate inside the loop cost up to 0.1 ms apiece on some
        boxes: more than the MGHS-only step itself)."""
        ev = lambda: torch.cuda.Event(enable_timing=True)
        self._free_events = [[ev() for _ in range(7)] for _ in range(n_steps)]

    def step(self, record):
        cfg = self.cfg
        pool = self._free_events.pop() if record and getattr(self, '_free_events', None) else None
        if record and pool is None:
            pool = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        # dhd_mghs_lift: height argmax -> band, context re-layout, geometry + grouping (4 launches)
        _, feat_nhwc = mghs_op.lift(self.plan, self.calib, self.height, cfg['height_range'], cfg['mask_range'], self.feat, self.ws)
        if record:
            # HIP events on the launch stream, around the streaming kernel only
            e0, e1 = pool[0], pool[1]
            outs = mghs_op.pool_forward_phases(self.plan, self.depth, feat_nhwc, self.ws, between=e0.record)
            e1.record()
            self.ev.append((e0, e1))
        else:
            outs = mghs_op.pool_forward(self.plan, self.depth, feat_nhwc, self.ws)
        if record:
            b0, b1 = pool[2], pool[3]
            b0.record()
        dg, fg = mghs_op.pool_backward(self.plan, self.depth, feat_nhwc, self.out_grads, self.ws)
        if record:
            b1.record()
            self.ev_bwd.append((b0, b1))
        fg_nchw = fg   # already (B*N, C, fH, fW)
        if self.with_sfa:
            self.x.grad = None
            for prm in self.stage_params:  # optimizer.zero_grad(set_to_none=True), the PyTorch default
                prm.grad = None
            if record:
                # HIP events on the launch stream around the stage's forward and backward calls
                e = pool[4:7]
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


class _StubDetector(torch.nn.Module):
    """(test only, --stub-model) the detector's call interface -- forward(return_loss=True, img_inputs=[imgs, ...], **labels) ->
    dict of losses -- on a few-kilobyte network, so that the self-launch, the process group, DDP and the record plumbing of the e2e
    workload can run in a CPU-only container.  Never part of a reported number: every line it produces carries "stub": true."""

    def __init__(self):
        super().__init__()
        nn = torch.nn
        self.body = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 16, 1))
        self.head = nn.Linear(16, 18)

    def forward(self, return_loss=True, img_inputs=None, voxel_semantics=None, **_):
        imgs = img_inputs[0]
        x = self.body(imgs.flatten(0, 1)).mean((2, 3)).view(imgs.shape[0], imgs.shape[1], -1).mean(1)
        return dict(loss_occ=torch.nn.functional.cross_entropy(self.head(x).float(), voxel_semantics), loss_height=(x.float() ** 2).mean())


class EndToEnd:
    """DHD-S exactly as projects/configs/DHD/DHD-S.py:42-155 (random init, synthetic 6-camera batch,
    SURVEY.md 8d config 2): forward_train -> sum of the four losses -> backward -> grad clip 5 -> AdamW."""

    def __init__(self, dev, batch, seed, world, amp, model='dhd-s', ema=True, graph=False, bucket_mb=64, static_graph=False,
                 ddp_graph=False, layout=None, stub=False):
        import dhd_amd
        from dhd_amd.detector import dhd_l_model_cfg, dhd_m_model_cfg, dhd_s_model_cfg
        torch.manual_seed(seed)
        self.dev_type = dev.type
        if stub:
            return self._init_stub(dev, batch, seed, world, amp, graph, bucket_mb, static_graph, ddp_graph)
        # dhd-m: DHD-M.py (DHD_stereo: key frame + 1 adjacent + 1 stereo reference frame, D = 88, SFA with C = 512)
        # dhd-l: DHD-L.py (the same wiring on a Swin-B backbone, 512 x 1408 images, 32 x 88 feature maps with 512 channels)
        frames = 1 if model == 'dhd-s' else 3
        cfg = {'dhd-s': dhd_s_model_cfg, 'dhd-m': dhd_m_model_cfg, 'dhd-l': dhd_l_model_cfg}[model]()
        self.model = dhd_amd.build_detector(cfg).to(dev).train()
        if model == 'dhd-l':
            self.model.img_backbone.init_weights()   # trunc-normal init of the Swin linears / bias tables (swin.py:876-890)
        # Measured per dense stack on MI355X with the committed find-db, DHD-S fp16 step (docs/LAB_NOTEBOOK.md R5.5): NCHW 65.9 ms;
        # image encoder in channels_last 59.9; + UNets + head 59.5; + the library's NHWC BatchNorm(+ReLU) 57.1; + its bilinear
        # upsample kernels 56.5; + the BEV encoder (which lost 3.2 ms in channels_last on torch's NHWC upsample backward) 55.1.
        # float32 (with its own NHWC find-db entries): 143.7 -> 127.9 ms.
        # DHD-M fp16 B = 3 with find-db entries for its NHWC problems (experiments/e2e_dhdm_layout_ab.sh): 140.8 -> 122.4 ms.
        # DHD-L bf16 B = 2 (Swin-B: the convolutions are the smaller part): heuristics 379.1 -> find-db NCHW 362.1 -> channels_last 356.1 ms.
        default = 'channels_last' if (model == 'dhd-s' or (model, amp) in (('dhd-m', 'fp16'), ('dhd-l', 'bf16'))) else 'nchw'
        self.layout = layout or os.environ.get('DHD_E2E_LAYOUT') or default
        if self.layout.startswith('channels_last'):          # 'channels_last' or 'channels_last:part,part' (detector.use_channels_last)
            parts = self.layout.partition(':')[2]
            self.model.use_channels_last(True, parts.split(',') if parts else None)
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        self.n_params = sum(p.numel() for p in self.params)
        self.net = self.model
        if world > 1:
            self.net = torch.nn.parallel.DistributedDataParallel(self.model, device_ids=[dev.index], bucket_cap_mb=bucket_mb,
                                                                 gradient_as_bucket_view=True, static_graph=bool(static_graph))
        self.opt = torch.optim.AdamW(self.params, lr=2e-4, weight_decay=1e-2, fused=True, capturable=bool(graph))  # DHD-S.py:262
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
        # persistent half copies of the convolution / linear weights, refreshed by one multi-tensor copy after the optimizer step,
        # instead of autocast's cast per weight and forward (and once more in the checkpointed recomputation): dhd_amd/amp_weights.py.
        # After the EMA's deepcopy on purpose (a copy made later would fall back to autocast's path anyway).  DHD_NO_HALF_WEIGHT_CACHE=1: A/B
        self.wcache = None
        if self.amp is not None and not os.environ.get('DHD_NO_HALF_WEIGHT_CACHE'):
            self.wcache = dhd_amd.HalfWeightCache(self.model, self.amp)
        self.B = batch
        self.graphed = None
        self.graph_error = None
        # N > 1: the captured step would contain the RCCL all-reduces (capturable in principle); opt-in, see --ddp-graph.  gloo's
        # collectives are staged through the host by its own threads: a capture attempt invalidates the stream's capture and, on
        # this runtime, leaves the context unusable for the eager fallback (profiles/r5/two_ranks: "operation failed due to a
        # previous error during capture") -- so with a non-RCCL backend the capture is not attempted and the line says why
        self.want_graph = bool(graph) and (world == 1 or bool(ddp_graph))
        if self.want_graph and world > 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() != 'nccl':
            self.want_graph = False
            self.graph_error = (f'not attempted: --ddp-graph needs RCCL (backend "nccl"); "{torch.distributed.get_backend()}" stages its '
                                'collectives through the host and cannot be captured into a HIP graph')

    def _init_stub(self, dev, batch, seed, world, amp, graph, bucket_mb, static_graph, ddp_graph):
        """--stub-model (test only): the same step protocol (zero_grad, autocast, losses dict -> backward -> clip -> AdamW, DDP when
        world > 1) on a stand-in module of a few thousand parameters that accepts CPU tensors."""
        self.model = _StubDetector().to(dev).train()
        self.layout = 'stub'
        self.params = list(self.model.parameters())
        self.n_params = sum(p.numel() for p in self.params)
        self.net = self.model
        if world > 1:
            self.net = torch.nn.parallel.DistributedDataParallel(self.model, device_ids=[dev.index] if dev.type == 'cuda' else None,
                                                                 bucket_cap_mb=bucket_mb, gradient_as_bucket_view=True,
                                                                 static_graph=bool(static_graph))
        cuda = dev.type == 'cuda'
        self.opt = torch.optim.AdamW(self.params, lr=2e-4, weight_decay=1e-2, fused=cuda, capturable=bool(graph) and cuda)
        self.ema = None
        g = torch.Generator(device='cpu').manual_seed(seed)
        self.kw = dict(img_inputs=[torch.randn(batch, 6, 3, 16, 16, generator=g).to(dev)],
                       voxel_semantics=torch.randint(0, 18, (batch,), generator=g).to(dev))
        self.amp = {'off': None, 'bf16': torch.bfloat16, 'fp16': torch.float16}[amp]
        self.scaler = torch.amp.GradScaler(dev.type) if amp == 'fp16' else None
        self.B, self.graphed, self.graph_error = batch, None, None
        self.want_graph = bool(graph) and cuda and (world == 1 or bool(ddp_graph))
        if bool(graph) and not cuda:
            self.graph_error = 'not attempted: CPU tensors (--stub-model)'

    def capture(self):
        """After the eager warm-up: the whole step as one HIP graph (dhd_amd/graph.py); falls back to eager on failure."""
        if not self.want_graph:
            return
        from dhd_amd.graph import GraphedStep
        try:
            self.graphed = GraphedStep(lambda: self._eager_step(), warmup=2, emas=[self.ema] if self.ema is not None else [])
        except Exception as e:  # noqa: BLE001 -- report and keep measuring eagerly
            self.graph_error = f'{type(e).__name__}: {e}'[:300]
            self.graphed = None
            try:
                torch.cuda.synchronize()
            except Exception as e2:  # noqa: BLE001 -- a capture that poisoned the context: the caller keeps its eager numbers
                self.graph_error += f' | then synchronize: {type(e2).__name__}: {e2}'[:200]

    def step(self, record):
        if self.graphed is not None:
            return self.graphed()
        return self._eager_step()

    def _eager_step(self):
        self.opt.zero_grad(set_to_none=True)
        with torch.autocast(self.dev_type, dtype=self.amp, enabled=self.amp is not None):
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
        if getattr(self, 'wcache', None) is not None:
            self.wcache.refresh()
        if self.ema is not None:
            self.ema.update(None, self.model)
        return loss


def fence():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    ddist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class Watchdog:
    """A host timer around a step that may hang on the device (the first graph replay of RCCL collectives on hardware this code
    was never run on): when it expires, `on_expire()` runs (rank 0 prints the line from what has been measured) and the process
    leaves with os._exit(0) -- no destructors, no collective."""

    def __init__(self, seconds, on_expire):
        import threading
        self.seconds, self.on_expire = seconds, on_expire
        self._timer = threading.Timer(seconds, self._fire)
        self._timer.daemon = True

    def _fire(self):
        try:
            self.on_expire(f'watchdog: the DDP graph attempt exceeded {self.seconds:.0f} s; the record holds the eager measurements')
        finally:
            sys.stdout.flush()
            os._exit(0)

    def __enter__(self):
        self._timer.start()
        return self

    def __exit__(self, *exc):
        self._timer.cancel()
        return False


# set when a graph capture of the N > 1 record raised outside its own try blocks (a context left unusable by a failed capture):
# the line is then printed from what was measured and the process leaves without the closing barrier (main, run_e2e)
CAPTURE_TROUBLE = []


def leave():
    """ddist.shutdown(), unless a capture left the context in doubt: then no collective, no destructors, exit code 0."""
    if CAPTURE_TROUBLE:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    ddist.shutdown()


def resolve_ddp_graph(a, world):
    """--ddp-graph / --no-ddp-graph / neither (= try when the backend is RCCL)."""
    if world == 1 or a.no_graph:
        return False
    return (a.dist_backend == 'nccl') if a.ddp_graph is None else bool(a.ddp_graph)


def e2e_measure(a, rank, world, dev, legs, model, warmup, steps, emit_partial=None, stub=False, want_util=False):
    """The whole-detector step, per leg (amp, tag, with_cp).  N = 1: eager + HIP-graph step.  N > 1, in this order:
      1. the SAME binary's single-GPU step on every rank at once (no DDP wrapper, no collective): eager and graphed -- the
         like-for-like denominators of a scaling efficiency (its eager step here, its graph step in 3);
      2. the DDP step eagerly (bucketed all-reduce overlapped with backward), each rank's own time next to the MAX, and the
         same under no_sync() -> exposed (non-overlapped) all-reduce time;
      3. LAST, under a watchdog, every graph capture: the single-GPU graph steps, then -- when asked for or by default over RCCL
         (resolve_ddp_graph) -- the DDP step captured into a HIP graph.  Every eager number above is already in the record if a
         capture fails, poisons the context or hangs.
    Returns (out, scaling): `out[tag]` as before; `scaling` = the N > 1 summary that goes to the top level of the line."""
    out, scaling, pending, singles = {}, {}, [], []
    per_rank = {}
    backend = a.dist_backend if world > 1 else None
    try_ddp_graph = resolve_ddp_graph(a, world)

    def timed(job, n, tag=None):
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            job.step(True)
        fence()
        mine = time.perf_counter() - t0
        if tag is not None and world > 1:   # every rank's own time next to the MAX that defines the step
            per_rank[tag] = [round(1e3 * v / n, 3) for v in ddist.gather_values(mine)]
        return ddist.max_over_ranks(mine, dev) / n

    def make(amp, w, with_cp, ddp_graph=False):
        job = EndToEnd(dev, a.batch, 1000 + rank, w, amp, model, not a.no_ema, graph=not a.no_graph, bucket_mb=a.bucket_mb,
                       static_graph=a.ddp_static_graph, ddp_graph=ddp_graph, layout=a.layout, stub=stub)
        if not with_cp and not stub:
            job.model.img_backbone.with_cp = False
        return job

    def guarded(fn):
        """fn() on every rank; (result, [errors over all ranks]).  A rank that fails (out of memory, a bad kernel ...) must not
        leave the others waiting in a collective, and rank 0 must be able to report a failure it did not see."""
        res, err = None, None
        try:
            res = fn()
        except Exception as exc:  # noqa: BLE001
            err = f'{type(exc).__name__}: {exc}'[:300]
        return res, ddist.gather_errors(err)

    n_params = None
    for amp, tag, with_cp in legs:
        rec = dict(steps=steps, warmup=warmup)
        if world > 1:
            # 1. single-GPU step of the same binary, all ranks at once
            single, errs = guarded(lambda: make(amp, 1, with_cp))
            if errs:
                out[tag] = dict(error=errs)
                continue
            for _ in range(warmup):
                single.step(False)
            sg = dict(ms_per_step_eager=1e3 * timed(single, max(2, steps // 2)))
            sg['note'] = 'no DDP wrapper, no collective; every rank runs it at the same time, MAX over ranks'
            rec['single_gpu_same_binary'] = sg
            # its HIP-graph step is measured AFTER every eager number of the record (phase 3 below): on the development box a graph
            # capture in the process made the next DDP-over-gloo float32 step 100x slower (226 ms -> 24-35 s per step; fp16 unaffected;
            # profiles/r6/two_rank_e2e_scaling_gloo.txt) -- whatever a capture does to the context, no eager measurement may come after one
            if single.want_graph:
                singles.append((tag, single))
            else:
                if single.graph_error:
                    sg['hip_graph_error'] = single.graph_error
                del single
                if torch.cuda.is_available():
                    torch.cuda.empty_cache()
        job, errs = guarded(lambda: make(amp, world, with_cp, ddp_graph=try_ddp_graph))
        if errs:
            out[tag] = dict(rec, error=errs)
            continue
        n_params = job.n_params
        rec['layout'] = job.layout
        for _ in range(warmup):
            job.step(False)
        if world == 1:
            eager = timed(job, 2) if job.want_graph else None
            job.capture()
            per_step = timed(job, steps, tag)
            rec.update(samples_per_s=a.batch / per_step, ms_per_step=1e3 * per_step, hip_graph=job.graphed is not None)
            if eager is not None:
                rec['ms_per_step_eager'] = 1e3 * eager
            if job.graph_error:
                rec['hip_graph_error'] = job.graph_error
        else:
            # 2. DDP, eager
            per_step = timed(job, steps, tag)
            rec.update(samples_per_s=a.batch * world / per_step, ms_per_step=1e3 * per_step, ms_per_step_eager=1e3 * per_step,
                       hip_graph=False, ms_per_step_by_rank=per_rank.get(tag), ddp_graph_requested=a.ddp_graph,
                       ddp_graph_attempt=bool(try_ddp_graph and job.want_graph))
            with job.net.no_sync():
                job.step(False)
                rec['ms_per_step_no_allreduce'] = 1e3 * timed(job, steps)
            rec['exposed_allreduce_ms'] = max(0.0, rec['ms_per_step_eager'] - rec['ms_per_step_no_allreduce'])
            rec.update(allreduce_bytes=4 * job.n_params, bucket_mb=a.bucket_mb, static_graph=bool(a.ddp_static_graph))
            if job.graph_error:     # e.g. "not attempted: --ddp-graph needs RCCL"
                rec['hip_graph_error'] = job.graph_error
            elif not rec['ddp_graph_attempt']:
                rec['hip_graph_error'] = 'not attempted: ' + ('--no-graph' if a.no_graph else '--no-ddp-graph' if a.ddp_graph is False else
                                                              f'backend "{a.dist_backend}" is not RCCL (the default tries the capture over "nccl" only)')
            if rec['ddp_graph_attempt']:
                pending.append((tag, job))
        if not with_cp:
            rec['note'] = 'img_backbone.with_cp = False (the configs set True): an A/B beside the fp16 leg, not the configs\' step'
        if want_util and tag == 'fp16' and world == 1 and not stub:
            try:
                util = e2e_module_utilisation(job)
                util['whole_step_tflops'] = round(util['sum_gflop'] / rec['ms_per_step'], 1)          # GFLOP / ms = TFLOP/s
                util['whole_step_frac_of_mfma_peak'] = round(util['sum_gflop'] / rec['ms_per_step'] / MFMA_PEAK_TFLOPS, 4)
                rec['mfma_utilisation'] = util
            except Exception as exc:  # noqa: BLE001
                rec['mfma_utilisation'] = dict(error=f'{type(exc).__name__}: {exc}'[:300])
        out[tag] = rec
        if not any(job is j for _, j in pending):
            del job
            if torch.cuda.is_available():
                torch.cuda.empty_cache()

    def summary():
        legs_out = {}
        for tag, rec in out.items():
            if not isinstance(rec, dict) or 'ms_per_step_eager' not in rec or world == 1:
                continue
            sg = rec.get('single_gpu_same_binary', {})
            per = lambda ms, n: None if ms is None else a.batch * n / (ms * 1e-3)
            legs_out[tag] = dict(
                ddp_ms_per_step_eager=rec['ms_per_step_eager'], ddp_ms_per_step_graph=rec.get('ms_per_step_graph'),
                ddp_samples_per_s_eager=per(rec['ms_per_step_eager'], world), ddp_samples_per_s_graph=per(rec.get('ms_per_step_graph'), world),
                ddp_graph_error=rec.get('hip_graph_error'),
                single_gpu_ms_per_step_eager=sg.get('ms_per_step_eager'), single_gpu_ms_per_step_graph=sg.get('ms_per_step_graph'),
                single_gpu_samples_per_s_eager=per(sg.get('ms_per_step_eager'), 1), single_gpu_samples_per_s_graph=per(sg.get('ms_per_step_graph'), 1),
                ms_per_step_no_allreduce=rec.get('ms_per_step_no_allreduce'), exposed_allreduce_ms=rec.get('exposed_allreduce_ms'),
                ms_per_step_by_rank=rec.get('ms_per_step_by_rank'))
        if not legs_out:
            return None
        return dict(n_gpus=world, samples_per_gpu=a.batch, backend=('RCCL ("nccl")' if backend == 'nccl' else f'{backend} (test only)'),
                    bucket_mb=a.bucket_mb, allreduce_bytes=None if n_params is None else 4 * n_params, static_graph=bool(a.ddp_static_graph),
                    legs=legs_out,
                    note='north_star\'s scaling number: the whole-detector step under DDP at this N, eager and (if the capture worked) as one '
                         'HIP graph, beside the single-GPU step of the same binary measured in the same process -- compare eager with eager and '
                         'graph with graph.  The top-level `value` of this line is the collective-free hot path and says nothing about the fabric.')

    # 3. every graph capture of the N > 1 record, last, under the watchdog: the single-GPU graph steps, then the DDP graph attempt
    if pending or singles:
        if emit_partial is not None:
            wd = Watchdog(a.ddp_graph_timeout, lambda why: emit_partial(out, dict(summary() or {}, ddp_graph_abandoned=why)))
        else:
            wd = Watchdog(a.ddp_graph_timeout, lambda why: None)
        fence()
        try:
            with wd:
                for tag, single in singles:
                    sg = out[tag]['single_gpu_same_binary']
                    single.capture()
                    if ddist.gather_errors(single.graph_error):
                        single.graphed = None
                    if single.graphed is not None:
                        try:
                            sg['ms_per_step_graph'] = 1e3 * timed(single, steps)
                        except Exception as exc:  # noqa: BLE001
                            sg['hip_graph_error'] = f'replay: {type(exc).__name__}: {exc}'[:300]
                    if single.graph_error:
                        sg['hip_graph_error'] = single.graph_error
                singles.clear()
                if torch.cuda.is_available():
                    torch.cuda.empty_cache()
                for tag, job in pending:
                    rec = out[tag]
                    # torch's notes on DDP under graph capture ask for >= 11 DDP-enabled eager iterations before it (bucket rebuild,
                    # static-graph bookkeeping): warm-up + timed steps so far are 3 + 5 (+ the no_sync ones); 8 more
                    for _ in range(8):
                        job.step(False)
                    job.capture()
                    failed = ddist.gather_errors(job.graph_error)   # a capture that failed on one rank only: every rank stays eager
                    if failed:
                        job.graphed = None
                        rec['hip_graph_error'] = failed
                        continue
                    try:
                        g = timed(job, steps)
                        rec.update(ms_per_step_graph=1e3 * g, hip_graph=True, ms_per_step=1e3 * g, samples_per_s=a.batch * world / g)
                    except Exception as exc:  # noqa: BLE001
                        rec['hip_graph_error'] = f'replay: {type(exc).__name__}: {exc}'[:300]
                        break
        except Exception as exc:  # noqa: BLE001 -- e.g. a collective that fails because a capture left the context unusable
            CAPTURE_TROUBLE.append(f'{type(exc).__name__}: {exc}'[:300])
        pending.clear()
    sc = summary()
    if CAPTURE_TROUBLE and sc is not None:
        sc['graph_phase_abandoned'] = CAPTURE_TROUBLE[0]
    return out, sc, n_params


E2E_WORKLOAD = {'dhd-m': 'DHD-M (DHD_stereo: key + adjacent + stereo reference frame, D=88) whole detector: ResNet-50 + FPN',
                'dhd-l': 'DHD-L (configs[3]: DHD_stereo on 512x1408 images, D=88, 32x88 feature maps) whole detector: Swin-B + FPN_LSS',
                'dhd-s': 'DHD-S (configs[1]/[2]) whole detector: ResNet-50 + FPN'}


def parallelism_text(a, world):
    if world == 1:
        return 'single GPU'
    return (f'DDP x{world} ({"RCCL" if a.dist_backend == "nccl" else "gloo (test only)"} bucketed all-reduce, {a.bucket_mb} MB buckets, '
            'overlapped with backward)')


def run_e2e(a, rank, world, dev):
    """--workload e2e: the whole-detector step as the line's own metric (one leg: --amp)."""
    tag = {'off': 'fp32', 'bf16': 'bf16', 'fp16': 'fp16'}[a.amp]
    printed = []

    def emit(out, scaling):
        if rank != 0 or printed:
            return
        printed.append(True)
        rec = out.get(tag, {})
        ok = 'ms_per_step' in rec
        print(json.dumps(dict(
            metric=f'samples/sec (6-cam fwd+bwd) {a.model.upper()} end-to-end', value=rec['samples_per_s'] if ok else None, unit='samples/s',
            n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=rec.get('ms_per_step'), higher_is_better=True,
            scaling='weak', vs_baseline=None, dtype={'off': 'f32', 'bf16': 'bf16', 'fp16': 'f16'}[a.amp], data='synthetic',
            **(dict(stub=True) if a.stub_model else {}),
            config=dict(workload=('STUB detector (test only, --stub-model): plumbing, not a measurement' if a.stub_model else
                                  E2E_WORKLOAD[a.model] + ', MGHS (HIP), BEV encoder, 3 UNets, SFA (HIP stage), predictor + losses (HIP); '
                                  'forward_train + backward + grad-clip + AdamW' + ('' if a.no_ema else ' + weight EMA (HIP)') + '; random init'),
                        samples_per_gpu=a.batch, global_batch=a.batch * world, params=rec.get('n_params'),
                        parallelism=parallelism_text(a, world), hip_graph=rec.get('hip_graph'), hip_graph_error=rec.get('hip_graph_error'),
                        layout=rec.get('layout')),
            e2e={tag: rec}, e2e_scaling=scaling, distributed=dist_report, miopen_db=MIOPEN_DB)), flush=True)

    dist_report = ddist.rank_report()
    out, scaling, n_params = e2e_measure(a, rank, world, dev, [(a.amp, tag, True)], a.model, a.warmup, a.steps, emit_partial=emit,
                                         stub=a.stub_model)
    if tag in out and isinstance(out[tag], dict):
        out[tag]['n_params'] = n_params
    emit(out, scaling)
    leave()


def hbm_calibration(dev, nbytes, reps=20, warmup=3):
    """What this box's HBM does for the plain streams a kernel can be compared with, measured in THIS run with the same
    timer as the kernel (HIP events on the launch stream around each launch): hipMemsetAsync, a linear grid-stride fill
    with 16-byte non-temporal stores, and a linear read, each over `nbytes` (the dominant kernel's algorithmic bytes).
    `roofline.frac` stays achieved / spec peak; `frac_of_fill` = fill_ms / launch_ms says how far the kernel is from the
    fastest store stream of the same box, which is what round-over-round comparisons should use (boxes differ by +-8 %)."""
    lib = _lib.load()
    nbytes = int(nbytes) // 16 * 16
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    s = _lib.stream_ptr(dev)
    out = {}
    for name, pattern in (('memset', 0), ('fill', 1), ('read', 2)):
        ev = []
        for it in range(warmup + reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.dhd_hbm_calibrate(_lib.ptr(buf), nbytes, pattern, s), 'dhd_hbm_calibrate')
            e1.record()
            if it >= warmup:
                ev.append((e0, e1))
        torch.cuda.synchronize()
        out[name] = event_mean(ev)
    del buf
    return out, nbytes


def lift_timing(hp, reps=20, warmup=3):
    """dhd_mghs_lift (band ids + context re-layout + geometry + grouping, every frame: training) against
    dhd_mghs_lift_static (static rig at inference, SURVEY 8f-1: camera matrices and the full-height grid's grouping reused,
    only the band grids' entries redone), HIP events around the call, on a workspace with its own scratch."""
    cfg = hp.cfg
    ws = hp.plan.new_workspace(hp.dev, private_scratch=True)
    out = {}
    for name, static in (('lift_us', False), ('lift_static_us', True)):
        ev = []
        for it in range(warmup + reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            mghs_op.lift(hp.plan, hp.calib, hp.height, cfg['height_range'], cfg['mask_range'], hp.feat, ws, static=static)
            e1.record()
            if it >= warmup:
                ev.append((e0, e1))
        torch.cuda.synchronize()
        out[name] = 1e3 * event_mean(ev)
    return out


def mghs_amp_record(hp, steps, warmup, dtype=torch.float16):
    """The MGHS node as an autocast region sees it (configs[1]: DHD-S fp16, DHD-S.py:281; the reference's operator returns float32,
    bev_pool.py:20-21, and the convolutions on either side cast): per step lift + pooling forward + backward,
      `f32_nodes_plus_casts`: float32 tensors out, cast to half for the consumer, half gradients cast back to float32 (what autocast
                              does around a float32 node: 704 MB -> 352 MB and back at B = 4),
      `half_io`:              dhd_tensor_view.dtype = DHD_F16: the writer emits the half tensors (bit-identical to the cast), the
                              backward reads the half gradients.
    HIP events around the whole step and around the streaming writer; `steps` steps after `warmup`."""
    import ctypes as C
    from dhd_amd.mghs_op import _alloc_outputs, _views
    lib = _lib.load()
    cfg, plan, dev = hp.cfg, hp.plan, hp.dev
    gh = [g.to(dtype) for g in hp.out_grads]
    out = {}
    for mode in ('f32_nodes_plus_casts', 'half_io'):
        odt = dtype if mode == 'half_io' else torch.float32
        ev_step, ev_wr = [], []
        for it in range(warmup + steps):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            st = _lib.stream_ptr(dev)
            e[0].record()
            _, feat_nhwc = mghs_op.lift(plan, hp.calib, hp.height, cfg['height_range'], cfg['mask_range'], hp.feat, hp.ws)
            outs = _alloc_outputs(plan, 'collapsed', dev, odt)
            arr = _views(plan, 'collapsed', outs)
            _lib.check(lib.dhd_mghs_forward_gather(C.byref(plan.desc), _lib.ptr(hp.depth), _lib.ptr(feat_nhwc), C.byref(hp.ws.c), st),
                       'dhd_mghs_forward_gather')
            e[1].record()
            _lib.check(lib.dhd_mghs_forward_stream_views(C.byref(plan.desc), _lib.ptr(hp.depth), _lib.ptr(feat_nhwc), C.byref(arr),
                                                         C.byref(hp.ws.c), st), 'dhd_mghs_forward_stream_views')
            e[2].record()
            if mode == 'f32_nodes_plus_casts':
                consumed = [o.to(dtype) for o in outs]                  # the consumer's cast of the float32 tensors
                grads = [g.float() for g in gh]                        # the half gradients cast back for the float32 node
            else:
                consumed, grads = outs, gh
            garr = _views(plan, 'collapsed', grads)
            dg, fg = torch.empty_like(hp.depth), torch.empty_like(hp.feat)
            _lib.check(lib.dhd_mghs_backward_views(C.byref(plan.desc), _lib.ptr(hp.depth), _lib.ptr(feat_nhwc), C.byref(garr), _lib.ptr(dg),
                                                   _lib.ptr(fg), C.byref(hp.ws.c), st), 'dhd_mghs_backward_views')
            e[3].record()
            if it >= warmup:
                ev_step.append((e[0], e[3]))
                ev_wr.append((e[1], e[2]))
            del outs, consumed, grads
        torch.cuda.synchronize()
        out[mode] = dict(mghs_step_ms=event_mean(ev_step), writer_ms=event_mean(ev_wr))
    if hp.with_sfa:
        # the SFA stage operator in the same two regimes: x arrives in half (the concatenated encoder outputs of an autocast region)
        xh = hp.x.detach().to(dtype).requires_grad_()
        gyh = hp.gy.to(dtype)
        out['half_edges'] = dict(out['half_io'])     # the MGHS part is the same measurement; the stage differs (below)
        for mode in ('f32_nodes_plus_casts', 'half_edges', 'half_io'):
            # half_edges = round 4's form (ABI 3: out / gout / gx in half, float32 storage inside), half_io = the default since
            # round 5: HALF STORAGE (dhd_sfa_weights.storage_dtype, ABI 4) -- x read in half, y1 / y2 / g2 / g1 / du kept in half
            hp.stage.half_storage = mode == 'half_io'
            ev = []
            for it in range(warmup + steps):
                xh.grad = None
                for prm in hp.stage_params:
                    prm.grad = None
                e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                e[0].record()
                # float32 node: the caller widens x and narrows the result, autograd does the same to the gradients on the way back
                y = hp.stage(xh.float()).to(dtype) if mode == 'f32_nodes_plus_casts' else hp.stage(xh)
                y.backward(gyh)
                e[1].record()
                if it >= warmup:
                    ev.append(tuple(e))
            torch.cuda.synchronize()
            out[mode]['sfa_stage_ms'] = event_mean(ev)
            out[mode]['hot_path_ms'] = out[mode]['mghs_step_ms'] + out[mode]['sfa_stage_ms']
        hp.stage.half_storage = True
        del xh, gyh
    out['dtype'] = str(dtype).replace('torch.', '')
    out['writer_half_GBps'] = hp.pool_fwd_bytes / 2 / (out['half_io']['writer_ms'] * 1e-3) / 1e9   # half the output bytes (+ the 1 % of inputs)
    out['note'] = ('MGHS part of the hot path only (lift + pooling forward + backward), B = %d; writer_ms = the streaming writer alone '
                   '(half: 352 MB instead of 704 MB at B = 4); sfa_stage_ms = the stage operator forward + backward with a half x: widened and '
                   'narrowed around a float32 node (f32_nodes_plus_casts), dhd_sfa_weights.io_dtype only (half_edges: out / gout / gx in half, x '
                   'widened once, float32 storage inside), or dhd_sfa_weights.storage_dtype (half_io: every tensor of the stage in half, single-'
                   'product half GEMMs with float32 accumulation, float32 statistics and parameter gradients)' % hp.B)
    return out


def operator_roofline(hp, steps, warmup, fused=True):
    """The operator-level drop-in on its own (include/dhd_amd.h section 1 = bev_pool.cpp:30-39,74-85): the full-height
    grid with reference-style index lists (ranks_depth / ranks_feat / ranks_bev sorted by voxel, interval starts /
    lengths; for the backward re-grouped by ranks_feat as bev_pool.py:47-57 does).  Timed region per launch, as the
    reference's wrapper issues it: zero-fill of the caller-owned output(s) + the kernel.  Algorithmic bytes:
    out + depth + context + 12 B per kept point."""
    import ctypes as C
    from dhd_amd.bev_pool_v2 import bev_pool_v2 as bev_pool_v2_op
    lib = _lib.load()
    dev, B = hp.dev, hp.B
    N, D, fh, fw, Cc = hp.dims
    rank, _ = mghs_op.voxel_index(hp.plan, hp.calib, 0)
    pid = torch.nonzero(rank >= 0).flatten()
    rb = rank[pid].long()
    order = torch.argsort(rb, stable=True)
    rb, rd = rb[order].int().contiguous(), pid[order].int().contiguous()
    pix = (rd.long() // (D * fh * fw)) * (fh * fw) + rd.long() % (fh * fw)
    rf = pix.int().contiguous()
    _, ln = torch.unique_consecutive(rb, return_counts=True)
    st = (torch.cumsum(ln, 0) - ln).int().contiguous()
    ln = ln.int().contiguous()
    o2 = torch.argsort(rf, stable=True)
    rb2, rd2, rf2 = rb[o2].contiguous(), rd[o2].contiguous(), rf[o2].contiguous()
    _, ln2 = torch.unique_consecutive(rf2, return_counts=True)
    st2 = (torch.cumsum(ln2, 0) - ln2).int().contiguous()
    ln2 = ln2.int().contiguous()
    depth = hp.depth.view(B, N, D, fh, fw)
    feat = mghs_op._nchw_to_nhwc(hp.feat).view(B, N, fh, fw, Cc)
    out = torch.empty(B, 1, 200, 200, Cc, device=dev)
    og = torch.randn(B, 1, 200, 200, Cc, device=dev)
    dgrad, fgrad = torch.empty_like(depth), torch.empty_like(feat)
    s = _lib.stream_ptr(dev)
    ev = []

    def once(record):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        out.zero_()
        _lib.check(lib.dhd_bev_pool_v2_forward(_lib.ptr(depth), _lib.ptr(feat), _lib.ptr(out), _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb),
                                               _lib.ptr(ln), _lib.ptr(st), Cc, int(ln.numel()), s), 'dhd_bev_pool_v2_forward')
        e[1].record()
        dgrad.zero_()
        fgrad.zero_()
        _lib.check(lib.dhd_bev_pool_v2_backward(_lib.ptr(og), _lib.ptr(dgrad), _lib.ptr(fgrad), _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(rd2),
                                                _lib.ptr(rf2), _lib.ptr(rb2), _lib.ptr(ln2), _lib.ptr(st2), Cc, int(ln2.numel()), s),
                   'dhd_bev_pool_v2_backward')
        e[2].record()
        if record:
            ev.append(e)
    for _ in range(warmup):
        once(False)
    for _ in range(steps):
        once(True)
    torch.cuda.synchronize()
    # the same operator through its Python surface (zero-fill, kernel, permute; backward with the argsort re-grouping)
    import importlib
    from dhd_amd.bev_pool_v2 import bev_pool_v2 as bev_pool_v2_op
    bev_mod = importlib.import_module('dhd_amd.bev_pool_v2')   # the module (dhd_amd.bev_pool_v2 the attribute is the function)
    dt, ft = depth.clone().requires_grad_(), feat.clone().requires_grad_()
    shape = (B, 1, 200, 200, Cc)
    ogp = og.permute(0, 4, 1, 2, 3).contiguous()

    def python_op_ms(use_fused, clear=(), events=None):
        """Wall clock per forward + backward call: 3 warm-up calls, then 5 loops of `steps` calls each (synchronised on both sides),
        the MEDIAN loop -- one host stall of a few ms inside a single 20-call loop would otherwise be the figure."""
        loops = []
        for rep in range(6):
            n = 3 if rep == 0 else steps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                dt.grad = ft.grad = None
                for cache in clear:
                    cache.clear()
                if events is not None and rep > 0:
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                    e[0].record()
                bev_pool_v2_op(dt, ft, rd, rf, rb, shape, st, ln, fused=use_fused).backward(ogp)
                if events is not None and rep > 0:
                    e[1].record()
                    events.append(tuple(e))
            torch.cuda.synchronize()
            if rep > 0:
                loops.append((time.perf_counter() - t0) / n * 1e3)
        return float(np.median(loops))

    py_ms = python_op_ms(False)
    # the same with the regrouping redone in every backward (index lists rebuilt per call, as voxel_pooling_v2 does in training)
    py_uncached_ms = python_op_ms(False, clear=(bev_mod._regroup_cache,))
    # fused=True: the (B, C, Dz, Dy, Dx) tensor written once, its gradient read once in that layout
    fev = []
    py_fused_ms = python_op_ms(fused, events=fev)
    py_fused_gpu_ms = event_mean(fev)
    # the same with the voxel -> row map and the regrouping rebuilt in every call (index lists that change per call)
    py_fused_uncached_ms = python_op_ms(fused, clear=(bev_mod._regroup_cache, bev_mod._state_cache))
    fwd_ms = event_mean([(e[0], e[1]) for e in ev])
    bwd_ms = event_mean([(e[1], e[2]) for e in ev])
    n_kept = int(rb.numel())
    plane = B * 4 * Cc * 200 * 200
    small = B * (4 * N * D * fh * fw + 4 * N * fh * fw * Cc)
    fwd_bytes = plane + small + 12 * n_kept
    bwd_bytes = plane + 2 * small + 12 * n_kept
    ach = (fwd_bytes + bwd_bytes) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9
    return dict(bound='hbm', kernel='bev_pool_v2_fwd_kernel + bev_pool_v2_bwd_kernel (dhd_bev_pool_v2_forward/backward, incl. the '
                                    'zero-fill of the caller-owned outputs)', achieved=ach, peak=HBM_PEAK_GBPS, unit='GB/s',
                frac=ach / HBM_PEAK_GBPS,
                traffic=(lambda p: None if None in p else int(sum(p)))([pmc_traffic(k, B) for k in ('bev_pool_v2_fwd_vec_kernel', 'bev_pool_v2_bwd_vec_kernel')]),
                launch_ms=fwd_ms + bwd_ms, forward_ms=fwd_ms, backward_ms=bwd_ms,
                algorithmic_bytes=fwd_bytes + bwd_bytes, forward_bytes=fwd_bytes, backward_bytes=bwd_bytes, kept_points=n_kept,
                intervals=int(ln.numel()), python_op_fwd_bwd_ms=py_ms, python_op_fwd_bwd_regroup_every_call_ms=py_uncached_ms,
                python_op_fused_fwd_bwd_ms=py_fused_ms, python_op_fused_fwd_bwd_event_ms=py_fused_gpu_ms,
                python_op_fused_fwd_bwd_lists_rebuilt_every_call_ms=py_fused_uncached_ms,
                python_op_fused_frac=(fwd_bytes + bwd_bytes) / (py_fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                note='full-height grid only (Dz=1); traffic = PMC bytes of the two kernels (the zero-fills of the caller-owned outputs are '
                     'torch fills, not counted); python_op_fwd_bwd_ms = dhd_amd.bev_pool_v2(...).backward() incl. zero-fill and permute, with '
                     'the backward re-grouping (bev_pool.py:47-57: argsort + gathers + run-length scan in the reference; here '
                     'dhd_bev_pool_v2_regroup, a device counting sort) reused while the same index tensors come back; '
                     '..._regroup_every_call_ms redoes it in every backward; python_op_fused_* = bev_pool_v2(..., fused=True): the '
                     '(B, C, Dz, Dy, Dx) tensor written once by the segment writer, its gradient read once in that layout '
                     '(dhd_bev_pool_v2_fused_forward / _backward), wall clock per call and HIP events around the same calls, the voxel -> row '
                     'map and the regrouping reused while the same index tensors come back; ..._lists_rebuilt_every_call_ms redoes both')


def cpu_baseline(hp, max_batch, warmups=3, reps=5, budget_s=75.0):
    """SURVEY 8(d) / BASELINE.md 2.3: the torch-CPU twin of the reference's op sequence (oracle/mghs_torch_cpu.py: 4 x
    geometry, 4 x index preparation with argsort, pool as index_add_, permute, cat; backward by autograd) plus the SFA
    stage formula on torch-CPU modules, on the same synthetic inputs as the GPU run, B = 1 and B = max_batch,
    `warmups` warm-ups and the median of `reps` runs each.

    Threads: SURVEY asks for torch.set_num_threads(os.cpu_count()).  On the GPU box (256 hardware threads) that setting
    makes this op sequence ~200x SLOWER than 8 threads (measured: 46 s against 0.24 s per sample for the forward; the
    185 856 batched 3x3 matmuls and the index ops are dominated by fork/join overhead), which is neither a fair baseline
    nor affordable inside a benchmark run.  The thread count is therefore calibrated: one B = 1 forward+backward at
    8, 16, 32, ... threads up to os.cpu_count(), stopping at the first count that is slower than its predecessor; the
    fastest count is used and reported as `cores`, the calibration times are reported too.  A wall-clock budget bounds
    the leg: when it runs out the remaining repetitions are dropped (the counts actually used are reported)."""
    from oracle import mghs_torch_cpu as TC  # checker / baseline only
    cfg = hp.cfg
    t_leg = time.perf_counter()
    fr = TC.frustum(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x))

    def inputs(b):
        calib = [t(x[:b]) for x in hp.calib_np]
        depth, feat, hidx = (x[:b * 6] for x in hp.inputs_np)
        return calib, t(depth).requires_grad_(), t(feat).requires_grad_(), t(syn.height_probs_from_index(hidx, 65))

    def mghs_once(calib, dt, ft, height):
        dt.grad = ft.grad = None
        t0 = time.perf_counter()
        outs = TC.view_transform(cfg, fr, calib, dt, ft, height)
        t1 = time.perf_counter()
        torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
        return t1 - t0, time.perf_counter() - t1

    # thread calibration on one sample
    one = inputs(1)
    calib_times, best, n = {}, None, 8
    top = os.cpu_count() or 8
    while True:
        n = min(n, top)
        torch.set_num_threads(n)
        mghs_once(*one)                                   # warm-up at this thread count
        calib_times[n] = sum(mghs_once(*one))
        if best is not None and calib_times[n] > calib_times[best]:
            break
        best = n
        if n >= top or time.perf_counter() - t_leg > 0.3 * budget_s:
            break
        n *= 2
    threads = best
    torch.set_num_threads(threads)

    res = {}
    for b in sorted({1, min(max_batch, hp.B)}):
        args = inputs(b)
        times = []
        for it in range(warmups + reps):
            ft_, bt_ = mghs_once(*args)
            if it >= warmups:
                times.append((ft_, bt_))
            if time.perf_counter() - t_leg > budget_s and times:
                break
        res[b] = dict(mghs_fwd_s=float(np.median([x[0] for x in times])), mghs_bwd_s=float(np.median([x[1] for x in times])),
                      mghs_reps=len(times))
        if hp.with_sfa:
            st = channel_spatial_stage(512).train()
            x = torch.randn(b, 512, 200, 200, requires_grad=True)
            gy = torch.randn(b, 256, 200, 200)
            times = []
            for it in range(warmups + reps):
                x.grad = None
                t0 = time.perf_counter()
                TC.sfa_stage(st, x).backward(gy)
                if it >= warmups:
                    times.append(time.perf_counter() - t0)
                if time.perf_counter() - t_leg > 1.5 * budget_s and times:
                    break
            res[b]['sfa_stage_fwd_bwd_s'] = float(np.median(times))
            res[b]['sfa_reps'] = len(times)
            del x, gy
        r = res[b]
        r['samples_per_s'] = b / (r['mghs_fwd_s'] + r['mghs_bwd_s'] + r.get('sfa_stage_fwd_bwd_s', 0.0))
        r['mghs_only_samples_per_s'] = b / (r['mghs_fwd_s'] + r['mghs_bwd_s'])
    top_b = max(res)
    return dict(value=res[top_b]['samples_per_s'], unit='samples/s', cores=threads, kind='port',
                sample=f'the same synthetic workload at B={top_b} (value) and B=1: oracle/mghs_torch_cpu.py view_transform fwd+bwd '
                       f'(torch-CPU twin of the reference op sequence; the reference has no CPU pool, bev_pool.cpp:7-14)'
                       + (' + SFA stage fwd+bwd on torch-CPU modules' if hp.with_sfa else '') +
                       f'; {threads} torch threads (fastest of the calibrated counts, host has {top} hardware threads), '
                       f'{warmups} warm-ups, median of up to {reps}',
                host_threads=top, thread_calibration_s={str(k): v for k, v in calib_times.items()},
                by_batch={str(b): r for b, r in res.items()}, leg_seconds=time.perf_counter() - t_leg)


MFMA_PEAK_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA peak of MI355X (MI355X_MICROARCH.md: ~2.5 PF dense, 2 495 TF measured)


def e2e_module_utilisation(job, reps=4):
    """Where the dense 98 % of BASELINE.json's metric goes: every top-level module of the detector run ALONE,
    forward + backward, on the inputs it receives in a real step (captured by hooks in one eager forward), under the job's
    autocast dtype.  Per module: analytic FLOPs of its convolutions / linear layers (2 x MACs, counted by hooks from the actual
    shapes; x 3 for forward + data gradient + weight gradient), the time of forward + backward by HIP events, achieved TFLOP/s
    and the fraction of the dense half-precision MFMA peak.  The HIP nodes (img_view_transformer = MGHS + depth_net /
    HeightNet, mix = SFA) carry their dense sub-layers' FLOPs only; their pooling / blending work is HBM-bound by design."""
    import torch.nn as nn
    model = job.model
    tops = [(n, m) for n, m in model.named_children() if sum(p.numel() for p in m.parameters()) > 0 or n in ('mix',)]
    captured = {}
    hooks = []

    def det(x):
        if torch.is_tensor(x):
            return x.detach()
        if isinstance(x, (list, tuple)):
            return type(x)(det(v) for v in x)
        return x

    for name, m in tops:
        hooks.append(m.register_forward_pre_hook(lambda mod, args, kwargs, name=name: captured.setdefault(name, (det(args), det(kwargs))), with_kwargs=True))
    with torch.autocast('cuda', dtype=job.amp, enabled=job.amp is not None):
        model(return_loss=True, **job.kw)
    for h in hooks:
        h.remove()
    flops = {}

    def count(mod, args, out, owner):
        x = args[0]
        if isinstance(mod, (nn.Conv2d, nn.Conv3d)):
            macs = out.numel() * (mod.in_channels // mod.groups) * int(np.prod(mod.kernel_size))
        elif isinstance(mod, (nn.ConvTranspose2d, nn.ConvTranspose3d)):
            macs = x.numel() * (mod.out_channels // mod.groups) * int(np.prod(mod.kernel_size))
        else:
            macs = out.numel() * mod.in_features
        flops[owner] = flops.get(owner, 0) + 2 * macs

    def leaves(x):
        if torch.is_tensor(x):
            return [x] if x.is_floating_point() else []
        if isinstance(x, (list, tuple)):
            return [t for v in x for t in leaves(v)]
        if isinstance(x, dict):
            return [t for v in x.values() for t in leaves(v)]
        return []

    def with_grad(x):
        if torch.is_tensor(x):
            return x.clone().requires_grad_() if x.is_floating_point() and x.numel() > 4096 else x
        if isinstance(x, (list, tuple)):
            return type(x)(with_grad(v) for v in x)
        return x

    rec, total_ms, total_flop = {}, 0.0, 0.0
    for name, m in tops:
        if name not in captured:
            continue
        args, kwargs = captured[name]
        fh = [mm.register_forward_hook(lambda mod, a_, o_, owner=name: count(mod, a_, o_, owner))
              for mm in m.modules() if isinstance(mm, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d, nn.Linear))]
        grads = None
        times = []
        try:
            for it in range(reps + 2):
                ins = with_grad(args)
                for p_ in m.parameters():
                    p_.grad = None
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                with torch.autocast('cuda', dtype=job.amp, enabled=job.amp is not None):
                    out = m(*ins, **kwargs)
                outs = [o for o in leaves(out) if o.requires_grad]
                if grads is None:
                    grads = [torch.ones_like(o) for o in outs]
                if outs:
                    torch.autograd.backward(outs, grads)
                e1.record()
                if it == 0:
                    for h in fh:
                        h.remove()
                if it >= 2:
                    times.append((e0, e1))
                del out, outs, ins
            torch.cuda.synchronize()
            ms = float(np.median([a_.elapsed_time(b_) for a_, b_ in times]))
        except Exception as exc:  # noqa: BLE001 -- a module that cannot run alone is reported, not fatal
            for h in fh:
                h.remove()
            rec[name] = dict(error=f'{type(exc).__name__}: {exc}'[:160])
            continue
        f3 = 3.0 * flops.get(name, 0)
        rec[name] = dict(gflop_fwd_bwd=round(f3 / 1e9, 1), ms_fwd_bwd=round(ms, 3), tflops=round(f3 / (ms * 1e-3) / 1e12, 1) if ms > 0 else None,
                         frac_of_mfma_peak=round(f3 / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if ms > 0 else None)
        total_ms += ms
        total_flop += f3
        for p_ in m.parameters():
            p_.grad = None
    del captured
    torch.cuda.empty_cache()
    return dict(modules=rec, sum_ms=round(total_ms, 2), sum_gflop=round(total_flop / 1e9, 1), peak_tflops=MFMA_PEAK_TFLOPS,
                note='each top-level module alone, forward + backward on the inputs of a real step, eager, HIP events; FLOPs = 3 x 2 x MACs '
                     'of its Conv / ConvTranspose / Linear layers from the actual shapes; peak = dense fp16 MFMA (2.5 PF); the kernel-category '
                     'shares (layout transposes, casts, fp32 kernels) of the same step: profiles/r5/e2e_dhds_fp16_kernel_categories.txt')


def e2e_subrecord(a, rank, world, dev, warmup=3, steps=5, emit_partial=None):
    """north_star's target number, observed by the driver inside the default line: the whole DHD-S detector
    (forward_train + backward + grad clip + AdamW + weight EMA) in fp32 and under fp16 autocast, B samples per GPU,
    `warmup` warm-ups + `steps` timed steps (e2e_measure).  N > 1: DDP over RCCL (64 MB buckets, overlapped with backward); the
    exposed (non-overlapped) all-reduce time is measured as step(DDP) - step(DDP.no_sync()); returns (record, e2e_scaling)."""
    # third leg, single GPU only, NOT the configs' behaviour and therefore never the headline of this sub-record: the fp16 step without
    # the image backbone's activation checkpointing (DHD-S.py: with_cp=True buys memory on 32 GB cards; on 288 GB it only costs the
    # second backbone forward) -- reported beside the faithful fp16 leg as an A/B
    legs = [('off', 'fp32', True), ('fp16', 'fp16', True)] + ([('fp16', 'fp16_no_activation_checkpointing', False)] if world == 1 else [])
    cfg = lambda n_params: dict(workload='DHD-S (configs[1]/[2]) whole detector: ResNet-50 + FPN, MGHS (HIP), BEV encoder, 3 UNets, SFA (HIP stage), '
                                         'predictor + losses (HIP); forward_train + backward + grad-clip + AdamW + weight EMA (HIP); random init',
                                samples_per_gpu=a.batch, global_batch=a.batch * world, params=n_params, parallelism=parallelism_text(a, world))
    partial = None
    if emit_partial is not None:
        partial = lambda out, scaling: emit_partial(dict(out, config=cfg(None)), scaling)
    out, scaling, n_params = e2e_measure(a, rank, world, dev, legs, 'dhd-s', warmup, steps, emit_partial=partial, want_util=True)
    out['config'] = cfg(n_params)
    return out, scaling


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
        kern_ms = event_mean(ev)
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
        kern_ms = event_mean(ev)
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
    stub_cpu = a.stub_model and a.workload == 'e2e' and not torch.cuda.is_available()
    if not torch.cuda.is_available() and not stub_cpu:
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # no launcher environment: start one rank per GPU ourselves (the reference's tools/dist_train.sh:11-20 does the same
        # with torch.distributed.launch), then this process becomes the launcher
        have = torch.cuda.device_count()
        if have < a.gpus and a.dist_backend != 'gloo':   # gloo (test only): ranks may share a GPU (or have none: --stub-model)
            raise SystemExit(f'--gpus {a.gpus} requested but only {have} GPU(s) are visible')
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
                                  '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank, local, world = ddist.env_world()
    if world != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or without a launcher)')
    if stub_cpu:
        dev = torch.device('cpu')
    else:
        if a.dist_backend == 'gloo':
            local = local % torch.cuda.device_count()   # ranks may share a device (RCCL refuses that, gloo does not)
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    # RCCL ("nccl"); the hot path uses it only for the barrier / MAX around the timed region, the e2e sub-record for DDP
    ddist.init_from_env(backend=a.dist_backend, device=dev if a.dist_backend == 'nccl' else None)
    _lib.load()
    if a.workload == 'e2e':
        return run_e2e(a, rank, world, dev)
    if a.workload == 'ema':
        return run_ema(a, rank, world, dev)
    if a.workload == 'occ_loss':
        return run_occ_loss(a, rank, world, dev)
    hp = HotPath(dev, a.batch, 1000 + rank, not a.no_sfa, a.geometry, deterministic=a.deterministic, sfa_gemm=a.sfa_gemm)

    for _ in range(a.warmup):
        hp.step(False)

    # R repeats of the timed loop of exactly K steps; each repeat has its own fences and its own event samples
    def timed_loop(record=True):
        hp.ev, hp.ev_bwd, hp.ev_sfa = [], [], []
        hp.make_events((a.steps + 3) // 4)
        with no_gc():   # a cyclic collection inside the timed steps is a 30-40 ms host pause in a 0.1 s region
            fence()
            t0 = time.perf_counter()
            for k in range(a.steps):
                # HIP events around the dominant kernel / the backward / the SFA stage on every 4th timed step (and the first): each
                # recorded step carries seven extra marker packets and host calls, 35 us on a 0.34 ms MGHS-only step
                hp.step(record and k % 4 == 0)
            fence()
            mine = time.perf_counter() - t0
            el = ddist.max_over_ranks(mine, dev)
        own_ms.append(1e3 * mine / a.steps)
        parts = dict(writer_ms=event_mean(hp.ev), mghs_bwd_ms=event_mean(hp.ev_bwd)) if record else {}
        if record and hp.ev_sfa:
            parts['sfa_fwd_ms'] = event_mean([(e[0], e[1]) for e in hp.ev_sfa])
            parts['sfa_bwd_ms'] = event_mean([(e[1], e[2]) for e in hp.ev_sfa])
        return el, parts

    own_ms = []   # this rank's own loop times (the headline is the MAX over ranks per loop)
    reps = [timed_loop() for _ in range(max(1, a.repeats))]
    # every rank describes itself (backend, world size as the process group reports it, device, own step time): identical on all ranks
    dist_report = ddist.rank_report(ms_per_step_own=float(np.median(own_ms)))
    per_step = sorted(1e3 * el / a.steps for el, _ in reps)
    elapsed = float(np.median([el for el, _ in reps]))
    stats = lambda v: dict(median=float(np.median(v)), min=float(np.min(v)), max=float(np.max(v)))
    part_stats = {k: stats([p[k] for _, p in reps]) for k in reps[0][1]}
    n_event_samples = len(hp.ev) * len(reps)
    if a.child:   # a fresh-process repeat: the statistics only
        if rank == 0:
            print(json.dumps(dict(ms_per_step=stats(per_step), parts=part_stats, repeats=len(reps), steps=a.steps,
                                  pool_fwd_bytes=hp.pool_fwd_bytes, pool_bwd_bytes=hp.pool_bwd_bytes)), flush=True)
        ddist.shutdown()
        return

    # the same step with float32-level GEMM products in the SFA stage (bf16x6), W warm-ups + K timed steps
    elapsed_x6 = None
    if hp.with_sfa and (a.sfa_gemm or 'bf16x3') != 'bf16x6':
        main_gemm, hp.stage.gemm = hp.stage.gemm, 'bf16x6'
        for _ in range(a.warmup):
            hp.step(False)
        elapsed_x6 = float(np.median([timed_loop(False)[0] for _ in range(min(3, max(1, a.repeats)))]))
        hp.stage.gemm = main_gemm
    cal_ms, cal_bytes = hbm_calibration(dev, hp.pool_fwd_bytes)   # every rank runs it, rank 0 reports
    lift_us = lift_timing(hp)

    line = None
    if rank == 0:
        kern_ms = part_stats['writer_ms']['median']
        achieved = hp.pool_fwd_bytes / (kern_ms * 1e-3) / 1e9
        line = dict(
            metric=f'samples/sec (6-cam fwd+bwd) {a.geometry.upper()} view-transform hot path', value=a.batch * world * a.steps / elapsed,
            unit='samples/s', n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps,
            higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
            config=dict(workload=('DHD-S (configs[1])' if a.geometry == 'dhd-s' else a.geometry.upper() + ' geometry (configs[3-4])') + ' hot path: MGHS 4-grid lift-splat fwd+bwd incl. geometry/grouping'
                                 + ('' if a.no_sfa else ' + SFA attention stage fwd+bwd') +
                                 f'; geometry {a.geometry}: 6 cams -> {hp.dims[2]}x{hp.dims[3]}, D={hp.dims[1]}, C=64, grids 200x200x{{1,4,4,8}}; dense backbone/encoder convs not in the step',
                        samples_per_gpu=a.batch, global_batch=a.batch * world, parallelism=f'sample-sharded x{world}, no data-path collective' +
                        ('' if world == 1 else ': `value` is the collective-free hot path (N independent replicas of the per-GPU work); the DDP / RCCL '
                         'step north_star asks for at this N is the top-level `e2e_scaling` record'),
                        deterministic_forward=bool(a.deterministic),
                        sfa_gemm=None if a.no_sfa else (a.sfa_gemm or 'bf16x3') +
                        ' (float32 operands cut into bf16 parts for the bf16 MFMA, float32 accumulate; bf16x3 = 2 parts / 3 products per a*b, '
                        'stage output within 2.2e-5 of float64; bf16x6 = 3 parts / 6 products, float32-level; storage, element-wise '
                        'arithmetic and statistics are float32 in both)'),
            roofline=dict(bound='hbm', kernel='mghs_stream_fwd', achieved=achieved, peak=HBM_PEAK_GBPS, unit='GB/s',
                          frac=achieved / HBM_PEAK_GBPS, traffic=pmc_traffic('mghs_stream_fwd', a.batch), launch_ms=kern_ms,
                          algorithmic_bytes=hp.pool_fwd_bytes,
                          # same-run, same-timer calibration streams over the same number of bytes (hbm_calibration)
                          memset_ms=cal_ms['memset'], fill_ms=cal_ms['fill'], read_ms=cal_ms['read'], calibration_bytes=cal_bytes,
                          frac_of_fill=cal_ms['fill'] / kern_ms, frac_of_memset=cal_ms['memset'] / kern_ms,
                          fill_GBps=cal_bytes / (cal_ms['fill'] * 1e-3) / 1e9, read_GBps=cal_bytes / (cal_ms['read'] * 1e-3) / 1e9,
                          launch_ms_min=part_stats['writer_ms']['min'], launch_ms_max=part_stats['writer_ms']['max'],
                          event_samples=n_event_samples))
        # the protocol: R in-process repeats of the K-step loop; the headline is the MEDIAN loop
        line['distributed'] = dist_report
        line['miopen_db'] = MIOPEN_DB
        line.update(repeats=len(reps), ms_per_step_min=per_step[0], ms_per_step_max=per_step[-1], ms_per_step_first=1e3 * reps[0][0] / a.steps,
                    ms_per_step_all=[round(v, 5) for v in (1e3 * el / a.steps for el, _ in reps)], parts=part_stats)
        line['prepare'] = dict(lift_us, note='dhd_mghs_lift = height argmax -> band + context re-layout + geometry + grouping (4 launches, '
                               'every frame in training); dhd_mghs_lift_static = the same for a static rig at inference (SURVEY 8f-1)')
        if elapsed_x6 is not None:
            line['ms_per_step_bf16x6'] = 1e3 * elapsed_x6 / a.steps
            line['value_bf16x6'] = a.batch * world * a.steps / elapsed_x6
        bwd_ms = part_stats['mghs_bwd_ms']['median']
        bwd_ach = hp.pool_bwd_bytes / (bwd_ms * 1e-3) / 1e9
        bwd_traffic = (lambda p: None if None in p else int(sum(p)))([pmc_traffic(k, a.batch) for k in ('mghs_stream_bwd', 'mghs_pixel_bwd')])
        # `frac_effective`: ALGORITHMIC bytes (the whole out_grad, 177.7 MB/sample) over the time -- the kernels skip the out_grad lines
        # that hold no point, so this is a rate of useful work, not of HBM traffic; `frac_hbm`: the bytes the counters saw over the same
        # time (null without a PMC record of these sources).  (round 4 called the first one `frac`.)
        line['roofline_bwd'] = dict(bound='hbm', kernel='mghs_stream_bwd + mghs_pixel_bwd (dhd_mghs_backward)', achieved=bwd_ach,
                                    peak=HBM_PEAK_GBPS, unit='GB/s', frac_effective=bwd_ach / HBM_PEAK_GBPS,
                                    frac_hbm=None if bwd_traffic is None else bwd_traffic / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    traffic=bwd_traffic, launch_ms=bwd_ms, algorithmic_bytes=hp.pool_bwd_bytes)
        if 'sfa_fwd_ms' in part_stats:
            # second roofline, for the SFA stage operator as a whole (a dozen kernels per call): SURVEY 8(d) gives its forward
            # algorithmic traffic as x read twice + u/out written + the two 1x1 convs reading and writing (B,C,H,W) once each
            fwd_ms, bwd_ms = part_stats['sfa_fwd_ms']['median'], part_stats['sfa_bwd_ms']['median']
            c, hw = 256, 200 * 200
            fwd_bytes = a.batch * 4 * hw * (2 * 2 * c + 4 * c)   # 2 reads of (2C,H,W) + 4 passes over (C,H,W) = 328 MB/sample
            gemm_flop = 2.0 * c * c * hw * a.batch
            line['roofline_sfa_stage'] = dict(
                bound='hbm', kernel='dhd_sfa_stage_forward (plane_mean, fc, 2 x pw_gemm_cu, stat reductions, blend2_bn)',
                achieved=fwd_bytes / (fwd_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBPS, unit='GB/s',
                frac=fwd_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, traffic=sfa_forward_traffic(a.batch, 3 if a.sfa_gemm == 'bf16x6' else 2), launch_ms=fwd_ms,
                algorithmic_bytes=fwd_bytes,
                backward_ms=bwd_ms, gemm_tflops_fp32_equivalent=6 * gemm_flop / ((fwd_ms + bwd_ms) * 1e-3) / 1e12,
                note='six C x C GEMMs per forward+backward; GEMM precision as config.sfa_gemm (include/dhd_amd.h: dhd_sfa_weights.gemm); '
                     'f32-MFMA peak is 157 TFLOP/s')
    if rank == 0 and world == 1 and a.geometry == 'dhd-s' and not a.no_dhdl and not a.child:
        # configs[3] / [4]: the view transform of DHD-L.py (6 x 512x1408 -> 32 x 88 maps, D = 88, B = 2 samples per GPU), MGHS part only,
        # in a child process: 2.97 M points instead of 0.74 M -- the point-proportional kernels weigh more than the writers there
        t_stage = time.perf_counter()
        c = fresh_process_repeats(a, n=1, geometry='dhd-l', batch=2, no_sfa=True)[0]
        if 'error' not in c:
            wr, bw = c['parts']['writer_ms']['median'], c['parts']['mghs_bwd_ms']['median']
            c = dict(workload='DHD-L geometry (configs[3]/[4]): 6 cams 512x1408 -> 32x88, D = 88, C = 64, grids 200x200x{1,4,4,8}, B = 2; MGHS '
                              'lift + pooling forward + backward only', samples_per_gpu=2, ms_per_step=c['ms_per_step'], parts=c['parts'],
                     bound='hbm', kernel='mghs_stream_fwd', algorithmic_bytes=c['pool_fwd_bytes'], launch_ms=wr,
                     achieved=c['pool_fwd_bytes'] / (wr * 1e-3) / 1e9, peak=HBM_PEAK_GBPS, unit='GB/s',
                     frac=c['pool_fwd_bytes'] / (wr * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     step_frac_of_hbm_floor=(c['pool_fwd_bytes'] + c['pool_bwd_bytes']) / (HBM_PEAK_GBPS * 1e9) / (c['ms_per_step']['median'] * 1e-3),
                     backward_ms=bw)
        line['roofline_dhdl'] = c
        print(f'[bench] DHD-L geometry record {time.perf_counter() - t_stage:.1f} s', file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and a.fresh_procs > 0:
        t_stage = time.perf_counter()
        line['fresh_processes'] = fresh_process_repeats(a)
        print(f'[bench] {a.fresh_procs} fresh-process repeats {time.perf_counter() - t_stage:.1f} s', file=sys.stderr, flush=True)
    t_stage = time.perf_counter()
    if hp.plan.half_outputs_supported and not a.pmc_pass:
        with no_gc():
            amp_rec = mghs_amp_record(hp, max(5, min(a.steps, 20)), 3)   # every rank runs it, rank 0 reports
        if rank == 0:
            line['hotpath_amp'] = amp_rec
    if a.geometry == 'dhd-s' and not a.no_operator:
        with no_gc():
            op_roof = operator_roofline(hp, max(5, min(a.steps, 20)), 3, fused=not a.pmc_pass)   # every rank runs it, rank 0 reports
        if rank == 0:
            line['roofline_operator'] = op_roof
    if rank == 0 and world == 1 and a.cpu_samples > 0:
        line['cpu_baseline'] = cpu_baseline(hp, a.cpu_samples)
        print(f'[bench] cpu_baseline leg {time.perf_counter() - t_stage:.1f} s', file=sys.stderr, flush=True)
    t_stage = time.perf_counter()
    if a.geometry == 'dhd-s' and not a.no_sfa and not a.no_e2e:
        del hp
        torch.cuda.empty_cache()
        printed = []

        def emit_partial(e2e, scaling):   # the watchdog's way out (a hung DDP graph replay): the line from what has been measured
            if rank == 0 and not printed:
                printed.append(True)
                line['e2e'], line['event_samples_dropped'] = e2e, dict(EVENT_DROPS)
                if scaling is not None:
                    line['e2e_scaling'] = scaling
                print(json.dumps(line), flush=True)
        scaling = None
        try:
            e2e, scaling = e2e_subrecord(a, rank, world, dev, emit_partial=emit_partial)
        except Exception as exc:  # noqa: BLE001 -- the hot-path line must still be printed
            e2e = dict(error=f'{type(exc).__name__}: {exc}'[:400])
        if rank == 0:
            line['e2e'] = e2e
            if scaling is not None:
                line['e2e_scaling'] = scaling
            print(f'[bench] e2e leg {time.perf_counter() - t_stage:.1f} s', file=sys.stderr, flush=True)
    if rank == 0:
        line['event_samples_dropped'] = dict(EVENT_DROPS)   # event intervals event_mean() left out as host stalls (> 3 x median)
        print(json.dumps(line), flush=True)
    leave()


def fresh_process_repeats(a, n=None, geometry=None, batch=None, no_sfa=None):
    """The same W + R x K measurement in `--fresh-procs` new processes (N = 1): another placement of the 0.7 GB outputs and
    scratch by the allocator, another clock / thermal state.  Returns their statistics (what --child prints).  With `geometry` /
    `batch` / `no_sfa`: another workload of the hot path in a child (the DHD-L geometry record of the default line)."""
    import subprocess
    out = []
    no_sfa = a.no_sfa if no_sfa is None else no_sfa
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(a.steps), '--warmup', str(a.warmup), '--repeats', str(a.repeats),
           '--batch', str(batch or a.batch), '--geometry', geometry or a.geometry, '--child', '--fresh-procs', '0', '--cpu-samples', '0', '--no-e2e',
           '--no-operator']
    cmd += (['--no-sfa'] if no_sfa else []) + (['--deterministic'] if a.deterministic else []) + (['--sfa-gemm', a.sfa_gemm] if a.sfa_gemm else [])
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    for _ in range(a.fresh_procs if n is None else n):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            rows = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            out.append(json.loads(rows[-1]) if r.returncode == 0 and rows else dict(error=(r.stderr or r.stdout)[-300:]))
        except Exception as exc:  # noqa: BLE001 -- the main line must still be printed
            out.append(dict(error=f'{type(exc).__name__}: {exc}'[:300]))
    return out


if __name__ == '__main__':
    main()
