# Stand-in for mmdetection3d v1.0.0rc4 `configs/_base_/default_runtime.py`, which the reference's
# configs inherit (projects/configs/DHD/DHD-S.py:1-2) but does not vendor (doc/install.md:27-32).
# Restated from the published file; only the keys, not the mmcv hook machinery, matter here.
checkpoint_config = dict(interval=1)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook'), dict(type='TensorboardLoggerHook')])
dist_params = dict(backend='nccl')  # = RCCL on ROCm
log_level = 'INFO'
work_dir = None
load_from = None
resume_from = None
workflow = [('train', 1)]
opencv_num_threads = 0
mp_start_method = 'fork'
