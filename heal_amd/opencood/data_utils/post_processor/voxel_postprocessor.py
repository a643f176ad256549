"""VoxelPostprocessor (reference: opencood/data_utils/post_processor/voxel_postprocessor.py):
generate_anchor_box (:30-83) on the host, post_process (:245-405) on the gfx950 decode + rotated-NMS
kernel K8 (heal_decode_nms), generate_label (:85-207) on heal_label_assign, collate_batch (:210-243) and the
inherited generate_gt_bbx (base_postprocessor.py:47-107) on the host."""
import math
import os
import sys

import numpy as np
import torch

from heal_amd import ops


_DEFERRED_ANCHORS = None


def resolve_deferred_labels(target_dict):
    """Turn the deferred label inputs of a collated batch (see VoxelPostprocessor.defer) into `pos_equal_one`,
    `neg_equal_one`, `targets` -- float64 tensors on the labels' device, exactly what the non-deferred path delivers --
    and store them in `target_dict`.  A no-op for ordinary label dictionaries and for already resolved ones."""
    if 'deferred_gt_box_center' not in target_dict or 'pos_equal_one' in target_dict:
        return target_dict
    gt, mask, anchors = target_dict['deferred_gt_box_center'], target_dict['deferred_mask'], target_dict['deferred_anchors']
    post = VoxelPostprocessor({'order': 'hwl', 'anchor_args': {'num': int(anchors.shape[2])},
                               'target_args': {'pos_threshold': target_dict['deferred_pos_threshold'],
                                               'neg_threshold': target_dict['deferred_neg_threshold']}}, train=True)
    post.defer = False
    anchors_np = anchors.detach().cpu().numpy()
    frames = [post.generate_label(gt_box_center=gt[b].detach().cpu().numpy(), anchors=anchors_np,
                                  mask=mask[b].detach().cpu().numpy()) for b in range(gt.shape[0])]
    for k, v in VoxelPostprocessor.collate_batch(frames).items():
        target_dict[k] = v.to(gt.device)
    return target_dict


class VoxelPostprocessor:
    def __init__(self, anchor_params, train):
        self.params = anchor_params
        self.train = train
        self.anchor_num = self.params['anchor_args']['num']
        self._anchor_cache = {}
        # Deferred labels (`postprocess.defer_to_device: true` or HEAL_DEFER_VOXELIZE=1, the switch of the deferred
        # voxeliser): generate_label runs inside the datasets' forked DataLoader workers, where no HIP context can be
        # created.  train=False: the labels are unused, zeros are returned.  train=True: it only packs its inputs and
        # the assignment happens on the device when a loss first reads the labels (resolve_deferred_labels); datasets
        # whose collate reads the label tensors by key (the heter datasets) need num_workers=0 and no deferral instead.
        self.defer = bool(self.params.get('defer_to_device', False)) or os.environ.get("HEAL_DEFER_VOXELIZE", "0") == "1"
        # labels are never read (tools/inference.py): `postprocess.inference_only: true` or HEAL_INFERENCE_ONLY=1
        self.inference_only = (bool(self.params.get('inference_only', False))
                               or os.environ.get("HEAL_INFERENCE_ONLY", "0") == "1")

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

    # ---------------------------------------------------------------------------------------------- training labels
    @staticmethod
    def _standup_boxes(boxes, order):
        """boxes_to_corners_3d + corner2d_to_standup_box (box_utils.py:152-204,225-248) on a float32 device tensor
        [n,7]: min / max of the 8 rotated corners in x and y -> [n,4] (x1,y1,x2,y2).  float32 like the reference, whose
        check_numpy_to_torch converts every numpy input with .float()."""
        b = boxes[:, [0, 1, 2, 5, 4, 3, 6]] if order == 'hwl' else boxes
        t = torch.tensor([[1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1],
                          [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1]], dtype=torch.float32, device=boxes.device) / 2
        c = b[:, None, 3:6] * t[None]
        cosa, sina = torch.cos(b[:, 6])[:, None], torch.sin(b[:, 6])[:, None]
        x = (c[..., 0] * cosa + c[..., 1] * (-sina)) + c[..., 2] * 0.0   # rows of points @ [[c,s,0],[-s,c,0],[0,0,1]]
        y = (c[..., 0] * sina + c[..., 1] * cosa) + c[..., 2] * 0.0
        x = x + b[:, None, 0]
        y = y + b[:, None, 1]
        return torch.stack([x.min(1)[0], y.min(1)[0], x.max(1)[0], y.max(1)[0]], 1).contiguous()

    def generate_label(self, **kwargs):
        """voxel_postprocessor.py:85-207 with the IoU / assignment core on the GPU (heal_label_assign).
        gt_box_center (max_num,7), anchors (H,W,A,7), mask (max_num) as numpy -> dict of numpy float64 arrays
        pos_equal_one (H,W,A), neg_equal_one (H,W,A), targets (H,W,7A), like the reference."""
        assert self.params['order'] == 'hwl', 'Currently Voxel only supporthwl bbx order.'
        gt_box_center, anchors, masks = kwargs['gt_box_center'], kwargs['anchors'], kwargs['mask']
        if self.defer and self.inference_only:
            # tools/inference.py never reads the anchor labels the datasets prepare for every sample (the heter datasets
            # even collate them per agent), so under an EXPLICIT inference-only switch no assignment is run: all-zero arrays
            # of the reference's shapes.  `train=False` alone is not that switch: tools/train.py builds its validation set
            # with train=False and computes the loss on these labels (train.py:153-154)
            H, W, A = anchors.shape[:3]
            return {'pos_equal_one': np.zeros((H, W, A)), 'neg_equal_one': np.zeros((H, W, A)),
                    'targets': np.zeros((H, W, A * 7))}
        if self.defer:
            global _DEFERRED_ANCHORS
            _DEFERRED_ANCHORS = anchors       # one anchor grid per process; collate_batch (same worker) attaches it once
            t = self.params['target_args']
            return {'deferred_gt_box_center': np.asarray(gt_box_center), 'deferred_mask': np.asarray(masks),
                    'deferred_pos_threshold': float(t['pos_threshold']), 'deferred_neg_threshold': float(t['neg_threshold'])}
        dev = torch.device("cuda", torch.cuda.current_device())
        H, W, A = anchors.shape[:3]
        key = (id(anchors), anchors.shape)
        hit = self._anchor_cache.get(("label",) + key)
        if hit is None:
            a64 = torch.from_numpy(np.ascontiguousarray(anchors).reshape(-1, 7)).to(dev)
            hit = (a64, self._standup_boxes(a64.float(), self.params['order']), anchors)
            self._anchor_cache[("label",) + key] = hit
        a64, a_boxes = hit[0], hit[1]
        gt_all = torch.from_numpy(np.ascontiguousarray(gt_box_center)).to(dev)
        valid = torch.from_numpy(np.ascontiguousarray(masks) == 1).to(dev)
        gt_valid = gt_all[valid]
        g_boxes = self._standup_boxes(gt_valid.float(), self.params['order']) if gt_valid.shape[0] else \
            torch.zeros((0, 4), dtype=torch.float32, device=dev)
        t = self.params['target_args']
        assigned, neg = ops.label_assign(a_boxes, g_boxes, t['pos_threshold'], t['neg_threshold'])
        pos_idx = torch.nonzero(assigned >= 0)[:, 0]
        # the reference indexes gt_box_center (all rows) with the index into the masked subset (:172-190)
        g = gt_all[assigned[pos_idx].long()].double()
        an = a64[pos_idx]
        an_d = torch.sqrt(an[:, 4] ** 2 + an[:, 5] ** 2)
        tgt = torch.zeros((a64.shape[0], 7), dtype=torch.float64, device=dev)
        tgt[pos_idx] = torch.stack([(g[:, 0] - an[:, 0]) / an_d, (g[:, 1] - an[:, 1]) / an_d, (g[:, 2] - an[:, 2]) / an[:, 3],
                                    torch.log(g[:, 3] / an[:, 3]), torch.log(g[:, 4] / an[:, 4]),
                                    torch.log(g[:, 5] / an[:, 5]), g[:, 6] - an[:, 6]], 1)
        pos = (assigned >= 0).double()
        return {'pos_equal_one': pos.view(H, W, A).cpu().numpy(),
                'neg_equal_one': neg.double().view(H, W, A).cpu().numpy(),
                'targets': tgt.view(H, W, A * 7).cpu().numpy()}

    @staticmethod
    def collate_batch(label_batch_list):
        """voxel_postprocessor.py:210-243: stack the per-frame label dictionaries of generate_label."""
        if label_batch_list and 'deferred_gt_box_center' in label_batch_list[0]:
            first = label_batch_list[0]
            return {'deferred_gt_box_center': torch.from_numpy(np.array([f['deferred_gt_box_center'] for f in label_batch_list])),
                    'deferred_mask': torch.from_numpy(np.array([f['deferred_mask'] for f in label_batch_list])),
                    'deferred_anchors': torch.from_numpy(np.ascontiguousarray(_DEFERRED_ANCHORS)),
                    'deferred_pos_threshold': first['deferred_pos_threshold'],
                    'deferred_neg_threshold': first['deferred_neg_threshold']}
        keys = ("targets", "pos_equal_one", "neg_equal_one")
        return {k: torch.from_numpy(np.array([frame[k] for frame in label_batch_list])) for k in keys}

    def generate_gt_bbx(self, data_dict):
        """base_postprocessor.py:47-107: ground-truth corners [N,8,3] in the ego frame for evaluation.  Every cav's valid
        `object_bbx_center` rows go to corners (`params['order']`), through its `transformation_matrix_clean`; objects
        seen by several cavs are kept once (first occurrence of each id, ids visited in `set` order like the
        reference); boxes with any corner outside `gt_range` (x, y and z) are dropped."""
        from heal_amd.opencood.utils import box_utils
        corners, ids = [], []
        for cav_content in data_dict.values():
            centers = cav_content["object_bbx_center"][cav_content["object_bbx_mask"] == 1]
            c = box_utils.boxes_to_corners_3d(centers, self.params["order"])
            corners.append(box_utils.project_box3d(c.float(), cav_content["transformation_matrix_clean"]))
            ids += cav_content["object_ids"]
        corners = torch.vstack(corners)
        picked = corners[[ids.index(x) for x in set(ids)]]
        kept = box_utils.mask_boxes_outside_range_numpy(picked.cpu().numpy(), self.params["gt_range"], order=None)
        return torch.from_numpy(kept).to(device=corners.device)

    @staticmethod
    def delta_to_boxes3d(deltas, anchors):
        """voxel_postprocessor.py:407-453: regression maps (N,7A,H,W) + anchors (H,W,A,7) -> boxes (N,H*W*A,7)."""
        N = deltas.shape[0]
        d = deltas.permute(0, 2, 3, 1).contiguous().view(N, -1, 7)
        a = anchors.to(d.device).view(-1, 7).float()
        a_d = torch.sqrt(a[:, 4] ** 2 + a[:, 5] ** 2)
        out = torch.zeros_like(d)
        out[..., 0] = d[..., 0] * a_d + a[:, 0]
        out[..., 1] = d[..., 1] * a_d + a[:, 1]
        out[..., 2] = d[..., 2] * a[:, 3] + a[:, 2]
        out[..., 3:6] = torch.exp(d[..., 3:6]) * a[:, 3:6]
        out[..., 6] = d[..., 6] + a[:, 6]
        return out

    def _post_process_multi(self, data_dict, output_dict):
        """Late fusion (voxel_postprocessor.py:277-405 with several cavs): every cav's candidates are decoded, projected
        with its own transformation matrix and pooled before the filters and ONE rotated NMS.  Same steps, in the
        reference's order, as tensor ops on the device + heal_nms_quads; the single-cav hot path above stays on the
        fused kernel."""
        from heal_amd.opencood.utils import box_utils
        from heal_amd.opencood.utils.common_utils import limit_period
        boxes3d_list, scores_list = [], []
        thr = self.params['target_args']['score_threshold']
        for cav_id, out in output_dict.items():
            assert cav_id in data_dict
            cav = data_dict[cav_id]
            cls = out['cls_preds'] if 'cls_preds' in out else out['psm']
            reg = out['reg_preds'] if 'reg_preds' in out else out['rm']
            dirp = out.get('dir_preds', out.get('dm'))
            if 'iou_preds' in out:
                raise NotImplementedError("iou_preds rescoring is not used by the HEAL configs")
            prob = torch.sigmoid(cls.permute(0, 2, 3, 1)).reshape(1, -1)
            box3d = self.delta_to_boxes3d(reg, cav['anchor_box'] if isinstance(cav['anchor_box'], torch.Tensor)
                                          else torch.from_numpy(np.asarray(cav['anchor_box'])))
            assert box3d.shape[0] == 1
            mask = torch.gt(prob, thr).view(-1)
            boxes3d, scores = box3d[0][mask], prob[0][mask]
            if dirp is not None and len(boxes3d) != 0:
                dir_args = self.params['dir_args']
                nb = dir_args['num_bins']
                labels = torch.max(dirp.permute(0, 2, 3, 1).contiguous().reshape(-1, nb)[mask], dim=-1)[1]
                period = 2 * np.pi / nb
                rot = limit_period(boxes3d[..., 6] - dir_args['dir_offset'], 0, period)
                boxes3d[..., 6] = rot + dir_args['dir_offset'] + period * labels.to(dirp.dtype)
                boxes3d[..., 6] = limit_period(boxes3d[..., 6], 0.5, 2 * np.pi)
            if len(boxes3d) != 0:
                corners = box_utils.boxes_to_corners_3d(boxes3d, order=self.params['order'])
                tfm = cav['transformation_matrix']
                tfm = tfm if isinstance(tfm, torch.Tensor) else torch.from_numpy(np.asarray(tfm))
                boxes3d_list.append(box_utils.project_box3d(corners, tfm.to(corners.device).float()))
                scores_list.append(scores)
        if not boxes3d_list:
            return None, None
        pred, scores = torch.vstack(boxes3d_list), torch.cat(scores_list)
        keep = torch.logical_and(box_utils.remove_large_pred_bbx(pred), box_utils.remove_bbx_abnormal_z(pred))
        pred, scores = pred[keep], scores[keep]
        keep = torch.from_numpy(box_utils.nms_rotated(pred, scores, self.params['nms_thresh']).astype(np.int64)).to(pred.device)
        pred, scores = pred[keep], scores[keep]
        kept, mask = box_utils.mask_boxes_outside_range_numpy(pred.cpu().numpy(), self.params['gt_range'], order=None,
                                                              return_mask=True)
        return torch.from_numpy(kept).to(pred.device), scores[torch.from_numpy(mask).to(pred.device)]

    def _anchors_f32(self, anchor_box, device):
        """anchors as contiguous fp32 on the device (delta_to_boxes3d does `.float()`), cached."""
        key = (anchor_box.data_ptr() if isinstance(anchor_box, torch.Tensor) else id(anchor_box), str(device))
        hit = self._anchor_cache.get(key)
        if hit is None:
            t = anchor_box if isinstance(anchor_box, torch.Tensor) else torch.from_numpy(np.asarray(anchor_box))
            hit = (t.to(device=device, dtype=torch.float32).contiguous(), anchor_box)  # keep the source alive: key = address
            self._anchor_cache = {k: v for k, v in self._anchor_cache.items() if isinstance(k, tuple) and k and k[0] == "label"}
            self._anchor_cache[key] = hit
        return hit[0]

    def post_process(self, data_dict, output_dict):
        """-> (pred_box3d [K,8,3], scores [K]) or (None, None).  Intermediate / single-agent form:
        one entry (the ego) in output_dict; batch size 1 (voxel_postprocessor.py:314)."""
        if self.params['order'] != 'hwl':
            raise NotImplementedError("the decode kernel implements order 'hwl' (PointPillars / HEAL configs)")
        if len(output_dict) != 1:
            return self._post_process_multi(data_dict, output_dict)
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
