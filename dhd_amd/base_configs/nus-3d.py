# Stand-in for mmdetection3d v1.0.0rc4 `configs/_base_/datasets/nus-3d.py` (un-vendored base of
# projects/configs/DHD/*.py).  Restated from the published file; every key below is overridden or
# unused by the DHD configs except as a merge target.
point_cloud_range = [-50, -50, -5, 50, 50, 3]
class_names = ['car', 'truck', 'trailer', 'bus', 'construction_vehicle', 'bicycle', 'motorcycle', 'pedestrian',
               'traffic_cone', 'barrier']
dataset_type = 'NuScenesDataset'
data_root = 'data/nuscenes/'
input_modality = dict(use_lidar=True, use_camera=False, use_radar=False, use_map=False, use_external=False)
file_client_args = dict(backend='disk')
train_pipeline = [
    dict(type='LoadPointsFromFile', coord_type='LIDAR', load_dim=5, use_dim=5, file_client_args=file_client_args),
    dict(type='LoadPointsFromMultiSweeps', sweeps_num=10, file_client_args=file_client_args),
    dict(type='LoadAnnotations3D', with_bbox_3d=True, with_label_3d=True),
    dict(type='GlobalRotScaleTrans', rot_range=[-0.3925, 0.3925], scale_ratio_range=[0.95, 1.05], translation_std=[0, 0, 0]),
    dict(type='RandomFlip3D', flip_ratio_bev_horizontal=0.5),
    dict(type='PointsRangeFilter', point_cloud_range=point_cloud_range),
    dict(type='ObjectRangeFilter', point_cloud_range=point_cloud_range),
    dict(type='ObjectNameFilter', classes=class_names),
    dict(type='PointShuffle'),
    dict(type='DefaultFormatBundle3D', class_names=class_names),
    dict(type='Collect3D', keys=['points', 'gt_bboxes_3d', 'gt_labels_3d'])
]
test_pipeline = [
    dict(type='LoadPointsFromFile', coord_type='LIDAR', load_dim=5, use_dim=5, file_client_args=file_client_args),
    dict(type='LoadPointsFromMultiSweeps', sweeps_num=10, file_client_args=file_client_args),
    dict(type='MultiScaleFlipAug3D', img_scale=(1333, 800), pts_scale_ratio=1, flip=False,
         transforms=[dict(type='DefaultFormatBundle3D', class_names=class_names, with_label=False),
                     dict(type='Collect3D', keys=['points'])])
]
eval_pipeline = [
    dict(type='LoadPointsFromFile', coord_type='LIDAR', load_dim=5, use_dim=5, file_client_args=file_client_args),
    dict(type='LoadPointsFromMultiSweeps', sweeps_num=10, file_client_args=file_client_args),
    dict(type='DefaultFormatBundle3D', class_names=class_names, with_label=False),
    dict(type='Collect3D', keys=['points'])
]
data = dict(
    samples_per_gpu=4,
    workers_per_gpu=4,
    train=dict(type=dataset_type, data_root=data_root, ann_file=data_root + 'nuscenes_infos_train.pkl',
               pipeline=train_pipeline, classes=class_names, modality=input_modality, test_mode=False, box_type_3d='LiDAR'),
    val=dict(type=dataset_type, data_root=data_root, ann_file=data_root + 'nuscenes_infos_val.pkl', pipeline=test_pipeline,
             classes=class_names, modality=input_modality, test_mode=True, box_type_3d='LiDAR'),
    test=dict(type=dataset_type, data_root=data_root, ann_file=data_root + 'nuscenes_infos_val.pkl', pipeline=test_pipeline,
              classes=class_names, modality=input_modality, test_mode=True, box_type_3d='LiDAR'))
evaluation = dict(interval=24, pipeline=eval_pipeline)
