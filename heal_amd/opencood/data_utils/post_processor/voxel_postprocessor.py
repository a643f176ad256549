"""VoxelPostprocessor (reference: opencood/data_utils/post_processor/voxel_postprocessor.py):
generate_anchor_box (:30-83) on the host, post_process (:245-405) on the gfx950 decode + rotated-NMS
kernel K8 (heal_decode_nms)."""
import math
import sys

import numpy as np
import torch

from heal_amd import ops


class VoxelPostprocessor:
    def __init__(self, anchor_params, train):
        self.params = anchor_params
        self.train = train
        self.anchor_num = self.params['anchor_args']['num']
        self._anchor_cache = {}

    def generate_anchor_box(self):
        a = self.params['anchor_args']
        W, H = a['W'], a['H']
        l, w, h = a['l'], a['w'], a['h']
        r = [math.radians(e) for e in a['r']]
        assert self.anchor_num == len(r)
        vh, vw = a['vh'], a['vw']
        xrange = [a['cav_lidar_range'][0], a['cav_lidar_range'][3]]
        yrange = [a['cav_lidar_range'][1], a['cav_lidar_range'][4]]
        feature_stride = a.get('feature_stride', 2)
        x = np.linspace(xrange[0] + vw, xrange[1] - vw, W // feature_stride)
        y = np.linspace(yrange[0] + vh, yrange[1] - vh, H // feature_stride)
        cx, cy = np.meshgrid(x, y)
        cx = np.tile(cx[..., np.newaxis], self.anchor_num)
        cy = np.tile(cy[..., np.newaxis], self.anchor_num)
        cz = np.ones_like(cx) * -1.0
        w = np.ones_like(cx) * w
        l = np.ones_like(cx) * l
        h = np.ones_like(cx) * h
        r_ = np.ones_like(cx)
        for i in range(self.anchor_num):
            r_[..., i] = r[i]
        if self.params['order'] == 'hwl':
            return np.stack([cx, cy, cz, h, w, l, r_], axis=-1)
        if self.params['order'] == 'lhw':
            return np.stack([cx, cy, cz, l, h, w, r_], axis=-1)
        sys.exit('Unknown bbx order.')

    def _anchors_f32(self, anchor_box, device):
        """anchors as contiguous fp32 on the device (delta_to_boxes3d does `.float()`), cached."""
        key = (anchor_box.data_ptr() if isinstance(anchor_box, torch.Tensor) else id(anchor_box), str(device))
        hit = self._anchor_cache.get(key)
        if hit is None:
            t = anchor_box if isinstance(anchor_box, torch.Tensor) else torch.from_numpy(np.asarray(anchor_box))
            hit = t.to(device=device, dtype=torch.float32).contiguous()
            self._anchor_cache = {key: hit}
        return hit

    def post_process(self, data_dict, output_dict):
        """-> (pred_box3d [K,8,3], scores [K]) or (None, None).  Intermediate / single-agent form:
        one entry (the ego) in output_dict; batch size 1 (voxel_postprocessor.py:314)."""
        if self.params['order'] != 'hwl':
            raise NotImplementedError("the decode kernel implements order 'hwl' (PointPillars / HEAL configs)")
        if len(output_dict) != 1:
            raise NotImplementedError("late-fusion post-processing over several cavs is not on the hot path")
        cav_id = next(iter(output_dict.keys()))
        out = output_dict[cav_id]
        cav = data_dict[cav_id]
        cls = out['cls_preds'] if 'cls_preds' in out else out['psm']
        reg = out['reg_preds'] if 'reg_preds' in out else out['rm']
        dirp = out.get('dir_preds', out.get('dm'))
        if 'iou_preds' in out:
            raise NotImplementedError("iou_preds rescoring is not used by the HEAL configs")
        anchors = self._anchors_f32(cav['anchor_box'], cls.device)
        tfm = cav['transformation_matrix']
        tfm = tfm.detach().cpu().numpy() if isinstance(tfm, torch.Tensor) else np.asarray(tfm)
        dir_args = self.params.get('dir_args', {'dir_offset': 0.7853, 'num_bins': 2})
        return ops.decode_nms(cls, reg, dirp, anchors, self.params['target_args']['score_threshold'],
                              dir_args['dir_offset'], dir_args['num_bins'], self.params['nms_thresh'],
                              tfm.astype(np.float32), self.params['gt_range'])
