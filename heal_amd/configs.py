"""Hypes dictionaries for the BASELINE configurations, built in code with the reference's key names
(what `yaml_utils.load_yaml` returns for opencood/hypes_yaml/opv2v/LiDAROnly/lidar_pyramid.yaml,
MoreModality/HEAL/stage2/m1_single_pyramid.yaml, Single/m1_pointpillar_pretrain.yaml and
MoreModality/HEAL/final_infer/m1m2m3m4.yaml, inference-relevant sections only).
`dump_yaml` writes them out as a config.yaml that both loaders parse.
"""
import copy

import yaml

from heal_amd.opencood.hypes_yaml.yaml_utils import load_general_params

FULL_RANGE = [-102.4, -102.4, -3, 102.4, 102.4, 1]
ANCHOR_YAW = [0, 90]
DIR_ARGS = {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": ANCHOR_YAW}


def _pointpillar_modality(lidar_range, aligner="identity"):
    aligner_args = {"core_method": "identity"} if aligner == "identity" else {
        "core_method": "convnext", "spatial_align": False, "args": {"num_of_blocks": 3, "dim": 64}}
    return {
        "core_method": "point_pillar",
        "sensor_type": "lidar",
        "encoder_args": {
            "voxel_size": [0.4, 0.4, 4],
            "lidar_range": list(lidar_range),
            "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
            "point_pillar_scatter": {"num_features": 64},
        },
        "backbone_args": {"layer_nums": [3], "layer_strides": [2], "num_filters": [64]},
        "aligner_args": aligner_args,
    }


def _fusion_backbone():
    return {"resnext": True, "layer_nums": [3, 5, 8], "layer_strides": [1, 2, 2], "num_filters": [64, 128, 256],
            "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128], "anchor_number": 2}


def _shrink_header():
    return {"kernal_size": [3], "stride": [1], "padding": [1], "dim": [256], "input_dim": 384}


def _common(lidar_range, max_cav):
    r = list(lidar_range)
    return {
        "yaml_parser": "load_general_params",
        "train_params": {"batch_size": 1, "max_cav": max_cav},
        "cav_lidar_range": r,
        "preprocess": {
            "core_method": "SpVoxelPreprocessor",
            "args": {"voxel_size": [0.4, 0.4, 4], "max_points_per_voxel": 32, "max_voxel_train": 32000,
                     "max_voxel_test": 70000},
            "cav_lidar_range": r,
        },
        "postprocess": {
            "core_method": "VoxelPostprocessor",
            "gt_range": r,
            "anchor_args": {"cav_lidar_range": r, "l": 3.9, "w": 1.6, "h": 1.56, "r": list(ANCHOR_YAW),
                            "feature_stride": 2, "num": 2},
            "target_args": {"pos_threshold": 0.6, "neg_threshold": 0.45, "score_threshold": 0.2},
            "order": "hwl",
            "max_num": 150,
            "nms_thresh": 0.15,
            "dir_args": copy.deepcopy(DIR_ARGS),
        },
        # the loss block of the reference's *_pyramid.yaml files (training side)
        "loss": {"core_method": "point_pillar_pyramid_loss", "args": {
            "pos_cls_weight": 2.0,
            "cls": {"type": "SigmoidFocalLoss", "alpha": 0.25, "gamma": 2.0, "weight": 1.0},
            "reg": {"type": "WeightedSmoothL1Loss", "sigma": 3.0, "codewise": True, "weight": 2.0},
            "dir": {"type": "WeightedSoftmaxClassificationLoss", "weight": 0.2, "args": copy.deepcopy(DIR_ARGS)},
            "depth": {"weight": 1.0},
            "pyramid": {"relative_downsample": [1, 2, 4], "weight": [0.4, 0.2, 0.1]}}},
    }


def lidar_pyramid(lidar_range=FULL_RANGE, max_cav=5):
    """PointPillars + PyramidFusion collaborative model (BASELINE configs 3 and the LiDAR part of 4)."""
    h = _common(lidar_range, max_cav)
    h["name"] = "heal_amd_opv2v_lidar_pyramid"
    h["model"] = {"core_method": "heter_pyramid_collab", "args": {
        "lidar_range": list(lidar_range), "supervise_single": True,
        "m1": _pointpillar_modality(lidar_range, "identity"),
        "fusion_backbone": _fusion_backbone(), "shrink_header": _shrink_header(),
        "in_head": 256, "anchor_number": 2, "dir_args": copy.deepcopy(DIR_ARGS)}}
    return load_general_params(h)


def m1_single_pyramid(lidar_range=FULL_RANGE):
    """Single-agent PointPillars through the pyramid backbone (BASELINE configs 1/2)."""
    h = _common(lidar_range, 1)
    h["name"] = "heal_amd_opv2v_m1_single_pyramid"
    h["model"] = {"core_method": "heter_pyramid_single", "args": {
        "ego_modality": "m1", "lidar_range": list(lidar_range), "fix_encoder": False,
        "m1": _pointpillar_modality(lidar_range, "convnext"),
        "fusion_backbone": _fusion_backbone(), "shrink_header": _shrink_header(),
        "in_head": 256, "anchor_number": 2, "dir_args": copy.deepcopy(DIR_ARGS)}}
    return load_general_params(h)


def m1_late(lidar_range=FULL_RANGE):
    """Single-agent PointPillars detector of the late-fusion / pre-training recipe (configs 1/2)."""
    h = _common(lidar_range, 1)
    h["name"] = "heal_amd_opv2v_m1_pointpillar_late"
    m1 = _pointpillar_modality(lidar_range, "identity")
    m1["layers_args"] = {"layer_nums": [3, 5, 8], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
                         "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]}
    m1["shrink_header"] = _shrink_header()
    m1["head_args"] = {"in_head": 256}
    h["model"] = {"core_method": "heter_model_late", "args": {
        "ego_modality": "m1", "lidar_range": list(lidar_range), "m1": m1, "anchor_number": 2,
        "dir_args": copy.deepcopy(DIR_ARGS)}}
    return load_general_params(h)


def _camera_modality(lidar_range, final_dim, encoder, cam_bound=51.2):
    grid_conf = {"xbound": [-cam_bound, cam_bound, 0.4], "ybound": [-cam_bound, cam_bound, 0.4], "zbound": [-10, 10, 20.0],
                 "ddiscr": [2, 50, 48], "mode": "LID"}
    data_aug_conf = {"resize_lim": [0.65, 0.7] if encoder == "EfficientNet" else [0.56, 0.61],
                     "final_dim": list(final_dim), "rot_lim": [-3.6, 3.6], "H": 600, "W": 800, "rand_flip": False,
                     "bot_pct_lim": [0.0, 0.05], "cams": ["camera0", "camera1", "camera2", "camera3"], "Ncams": 4}
    return {
        "core_method": "lift_splat_shoot",
        "sensor_type": "camera",
        "encoder_args": {"anchor_number": 2, "grid_conf": grid_conf, "data_aug_conf": data_aug_conf,
                         "img_downsample": 8, "img_features": 128, "use_depth_gt": False,
                         "depth_supervision": True, "camera_encoder": encoder},
        "camera_mask_args": {"cav_lidar_range": list(lidar_range), "grid_conf": copy.deepcopy(grid_conf)},
        "backbone_args": {"layer_nums": [3], "layer_strides": [2], "num_filters": [64], "inplanes": 128},
        "aligner_args": {"core_method": "convnext", "spatial_align": False,
                         "args": {"num_of_blocks": 3, "dim": 64}},
    }


def _second_modality(lidar_range):
    return {
        "core_method": "second",
        "sensor_type": "lidar",
        "encoder_args": {"voxel_size": [0.1, 0.1, 0.1], "lidar_range": list(lidar_range),
                         "mean_vfe": {"num_point_features": 4},
                         "spconv": {"num_features_in": 4, "num_features_out": 64},
                         "map2bev": {"feature_num": 128}},
        "backbone_args": {"layer_nums": [3], "layer_strides": [1], "num_filters": [64], "inplanes": 128},
        "aligner_args": {"core_method": "convnext", "spatial_align": False,
                         "args": {"num_of_blocks": 3, "dim": 64}},
    }


def heal_heter(modalities=("m1", "m2", "m3", "m4"), lidar_range=FULL_RANGE, max_cav=5, cam_bound=51.2, cam_dims=None):
    """HEAL final-infer collaborative model with several modalities (m1 PointPillars LiDAR,
    m2 Lift-Splat EfficientNet 384x512, m4 Lift-Splat Resnet101 336x448; m3 SECOND when built) --
    MoreModality/HEAL/final_infer/m1m2m3m4.yaml, BASELINE config 4.  `cam_bound` / `cam_dims` ({"m2": (H, W), ...})
    shrink the camera BEV grid and the image sizes for reduced-size parity tests."""
    cam_dims = dict({"m2": (384, 512), "m4": (336, 448)}, **(cam_dims or {}))
    h = _common(lidar_range, max_cav)
    h["name"] = "heal_amd_opv2v_" + "".join(modalities)
    args = {"ego_modality": "m1", "lidar_range": list(lidar_range), "supervise_single": True}
    for m in modalities:
        if m == "m1":
            args[m] = _pointpillar_modality(lidar_range, "identity")
        elif m == "m2":
            args[m] = _camera_modality(lidar_range, cam_dims["m2"], "EfficientNet", cam_bound)
        elif m == "m3":
            args[m] = _second_modality(lidar_range)
        elif m == "m4":
            args[m] = _camera_modality(lidar_range, cam_dims["m4"], "Resnet101", cam_bound)
        else:
            raise NotImplementedError(f"modality {m}")
    args.update({"fusion_backbone": _fusion_backbone(), "shrink_header": _shrink_header(), "in_head": 256,
                 "anchor_number": 2, "dir_args": copy.deepcopy(DIR_ARGS)})
    h["model"] = {"core_method": "heter_pyramid_collab", "args": args}
    return load_general_params(h)


def _v2xvit_args():
    return {"transformer": {"encoder": {
        "num_blocks": 1, "depth": 3, "use_roi_mask": True, "use_RTE": False, "RTE_ratio": 0,
        "cav_att_config": {"dim": 256, "use_hetero": True, "use_RTE": False, "RTE_ratio": 0, "heads": 8,
                           "dim_head": 32, "dropout": 0.3},
        "pwindow_att_config": {"dim": 256, "heads": [16, 8, 4], "dim_head": [16, 32, 64], "dropout": 0.3,
                               "window_size": [4, 8, 16], "relative_pos_embedding": True,
                               "fusion_method": "split_attn"},
        "feed_forward": {"mlp_dim": 256, "dropout": 0.3},
        "sttf": {"voxel_size": [0.4, 0.4, 4], "downsample_rate": 4}}}}


def _baseline_modality(base, strides, inplanes=None):
    m = {k: v for k, v in base.items() if k in ("core_method", "sensor_type", "encoder_args", "camera_mask_args")}
    m["backbone_args"] = {"layer_nums": [3, 5, 8], "layer_strides": list(strides), "num_filters": [64, 128, 256],
                          "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]}
    if inplanes:
        m["backbone_args"]["inplanes"] = inplanes
    m["shrink_header"] = {"kernal_size": [3], "stride": [2], "padding": [1], "dim": [256], "input_dim": 384}
    return m


def lidar_baseline(fusion_method="v2xvit", lidar_range=FULL_RANGE, max_cav=5, modality="m1"):
    """HeterModelBaseline (LiDAROnly/lidar_v2xvit.yaml; BASELINE config 5 with modality='m3'):
    encoder -> plain BEV backbone -> shrinker -> single-scale fusion (v2xvit | att | max) -> heads."""
    h = _common(lidar_range, max_cav)
    h["name"] = f"heal_amd_opv2v_{modality}_{fusion_method}"
    if modality == "m1":
        mod = _baseline_modality(_pointpillar_modality(lidar_range), (2, 2, 2))
    elif modality == "m3":
        mod = _baseline_modality(_second_modality(lidar_range), (1, 2, 2), inplanes=128)
    else:
        raise NotImplementedError(modality)
    args = {"ego_modality": modality, "lidar_range": list(lidar_range), modality: mod, "fusion_method": fusion_method,
            "in_head": 256, "anchor_number": 2, "dir_args": copy.deepcopy(DIR_ARGS)}
    if fusion_method == "v2xvit":
        args["v2xvit"] = _v2xvit_args()
    elif fusion_method == "att":
        args["att"] = {"feat_dim": 256}
    h["model"] = {"core_method": "heter_model_baseline", "args": args}
    # encoder + backbone + stride-2 shrinker leave the map at 1/4 of the 0.4 m anchor grid (lidar_v2xvit.yaml: feature_stride 4)
    h["postprocess"]["anchor_args"]["feature_stride"] = 4
    return load_general_params(h)


def oldstyle_pointpillar(fusion_method=None, lidar_range=FULL_RANGE, max_cav=5, compression=0):
    """Old-style YAML (v2xsim2/visualization.yaml `model` block): core_method point_pillar, or
    point_pillar_baseline when a fusion_method (max | att | v2xvit) is given.  Uses the `processed_lidar` key."""
    h = _common(lidar_range, max_cav)
    args = {"voxel_size": [0.4, 0.4, 4], "lidar_range": list(lidar_range), "anchor_number": 2, "max_cav": max_cav,
            "backbone_fix": False,
            "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
            "point_pillar_scatter": {"num_features": 64},
            "base_bev_backbone": {"layer_nums": [3, 5, 8], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
                                  "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]},
            "shrink_header": {"kernal_size": [3], "stride": [1], "padding": [1], "dim": [256], "input_dim": 384},
            "dir_args": copy.deepcopy(DIR_ARGS)}
    core = "point_pillar"
    if fusion_method is not None:
        core = "point_pillar_baseline"
        args["fusion_method"] = fusion_method
        if fusion_method == "att":
            args["att"] = {"feat_dim": 256}
        elif fusion_method == "v2xvit":
            args["v2xvit"] = _v2xvit_args()
        if compression:
            args["compression"] = compression
    h["name"] = f"heal_amd_{core}"
    h["model"] = {"core_method": core, "args": args}
    return load_general_params(h)


def oldstyle_lss(encoder="EfficientNet", final_dim=(384, 512), lidar_range=FULL_RANGE, max_cav=5):
    """Old-style camera detector (opencood/models/lift_splat_shoot.py): core_method lift_splat_shoot, `image_inputs`
    key, resnet18-trunk BevEncode on the 256x256 pooled grid, stride-2 shrink header to the 128x128 anchor map."""
    h = _common(lidar_range, max_cav)
    enc = _camera_modality(lidar_range, final_dim, encoder)["encoder_args"]
    args = dict(enc, bevout_feature=128,
                shrink_header={"kernal_size": [3], "stride": [2], "padding": [1], "dim": [128], "input_dim": 128},
                dir_args=copy.deepcopy(DIR_ARGS))
    h["name"] = "heal_amd_lift_splat_shoot"
    h["model"] = {"core_method": "lift_splat_shoot", "args": args}
    return load_general_params(h)


def dump_yaml(hypes, path):
    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        if hasattr(o, "tolist"):
            return o.tolist()
        return o
    with open(path, "w") as f:
        yaml.safe_dump(plain(hypes), f, sort_keys=False)
