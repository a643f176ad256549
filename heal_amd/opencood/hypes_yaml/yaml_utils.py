"""YAML config surface (reference: opencood/hypes_yaml/yaml_utils.py:14-49 load_yaml, :337-369 load_general_params,
and the post-parsers of the old-style model files: :97-137 load_point_pillar_params, :140-180 load_second_params,
:295-334 load_lift_splat_shoot_params; :234-249 save_yaml).  The reference's own YAML files parse unchanged."""
import math
import os
import re

import numpy as np
import yaml

_FLOAT_RE = re.compile(u'''^(?:
     [-+]?(?:[0-9][0-9_]*)\\.[0-9_]*(?:[eE][-+]?[0-9]+)?
    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
    |\\.[0-9_]+(?:[eE][-+][0-9]+)?
    |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\\.[0-9_]*
    |[-+]?\\.(?:inf|Inf|INF)
    |\\.(?:nan|NaN|NAN))$''', re.X)


class _Loader(yaml.Loader):
    """yaml.Loader with the float resolver that also accepts '1e-10' (no dot)."""


_Loader.add_implicit_resolver(u'tag:yaml.org,2002:float', _FLOAT_RE, list(u'-+0123456789.'))


def load_general_params(param):
    """Derive anchor grid sizes W/H/D from the lidar range and voxel size (yaml_utils.py:337-369)."""
    cav_lidar_range = param['preprocess']['cav_lidar_range']
    voxel_size = param['preprocess']['args']['voxel_size']
    anchor_args = param['postprocess']['anchor_args']
    vw, vh, vd = voxel_size[0], voxel_size[1], voxel_size[2]
    anchor_args['vw'], anchor_args['vh'], anchor_args['vd'] = vw, vh, vd
    anchor_args['W'] = math.ceil((cav_lidar_range[3] - cav_lidar_range[0]) / vw)
    anchor_args['H'] = math.ceil((cav_lidar_range[4] - cav_lidar_range[1]) / vh)
    anchor_args['D'] = math.ceil((cav_lidar_range[5] - cav_lidar_range[2]) / vd)
    param['postprocess'].update({'anchor_args': anchor_args})
    return param


def _grid_size(param):
    r = param['preprocess']['cav_lidar_range']
    return np.round((np.array(r[3:6]) - np.array(r[0:3])) / np.array(param['preprocess']['args']['voxel_size'])).astype(np.int64)


def load_point_pillar_params(param):
    """yaml_utils.py:97-137 (opencood/models/point_pillar*.py): the scatter grid size + the anchor grid (ceil)."""
    param['model']['args']['point_pillar_scatter']['grid_size'] = _grid_size(param)
    return load_general_params(param)


def load_lift_splat_shoot_params(param):
    """yaml_utils.py:295-334 (opencood/models/lift_splat_shoot*.py): the anchor grid only."""
    return load_general_params(param)


def load_second_params(param):
    """yaml_utils.py:140-180 (opencood/models/second*.py): `grid_size` of the sparse encoder; W/H/D truncate (int(),
    not ceil like the other parsers)."""
    r = param['preprocess']['cav_lidar_range']
    vw, vh, vd = param['preprocess']['args']['voxel_size'][:3]
    param['model']['args']['grid_size'] = _grid_size(param)
    a = param['postprocess']['anchor_args']
    a['vw'], a['vh'], a['vd'] = vw, vh, vd
    a['W'], a['H'], a['D'] = int((r[3] - r[0]) / vw), int((r[4] - r[1]) / vh), int((r[5] - r[2]) / vd)
    param['postprocess'].update({'anchor_args': a})
    return param


def save_yaml(data, save_name):
    """yaml_utils.py:234-249."""
    with open(save_name, 'w') as outfile:
        yaml.dump(data, outfile, default_flow_style=False)


_PARSERS = {"load_general_params": load_general_params, "load_point_pillar_params": load_point_pillar_params,
            "load_second_params": load_second_params, "load_lift_splat_shoot_params": load_lift_splat_shoot_params}


def load_yaml(file, opt=None):
    """Load a yaml file; `opt.model_dir` (if set) replaces `file` with <model_dir>/config.yaml; the
    `yaml_parser` key selects the post-parser (load_general_params on the HEAL path, the old-style model files' parsers;
    voxel / bev / stage-1 parsers belong to detectors outside the scope and raise)."""
    if opt and getattr(opt, "model_dir", None):
        file = os.path.join(opt.model_dir, 'config.yaml')
    with open(file, 'r') as stream:
        param = yaml.load(stream, Loader=_Loader)
    if "yaml_parser" in param:
        name = param["yaml_parser"]
        if name not in _PARSERS:
            raise NotImplementedError(f"yaml_parser '{name}' belongs to a detector outside the hot-path scope")
        param = _PARSERS[name](param)
    return param


def update_ranges(param, new_range):
    """tools/inference.py:54-73 + common_utils.update_dict: replace every *_range key recursively."""
    for k, v in list(param.items()):
        if isinstance(v, dict):
            update_ranges(v, new_range)
        elif k in ("cav_lidar_range", "lidar_range", "gt_range"):
            param[k] = list(new_range)
    return param
