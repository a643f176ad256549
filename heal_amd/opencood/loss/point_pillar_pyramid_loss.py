"""Loss of the pyramid-fusion models (reference: opencood/loss/point_pillar_pyramid_loss.py:11-149): the detection +
depth losses on the fused heads and a focal occupancy loss on every pyramid level's single-agent foreground map
(`occ_single_list`), against the anchor labels max-pooled to the level's resolution."""
import torch
import torch.nn.functional as F

from heal_amd.opencood.data_utils.post_processor.voxel_postprocessor import resolve_deferred_labels
from heal_amd.opencood.loss.point_pillar_depth_loss import PointPillarDepthLoss
from heal_amd.opencood.loss.point_pillar_loss import sigmoid_focal_loss


class PointPillarPyramidLoss(PointPillarDepthLoss):
    _LOG_FIELDS = PointPillarDepthLoss._LOG_FIELDS + (("Pyramid Loss", "pyramid_loss", "Pyramid_loss"),)

    def __init__(self, args):
        super().__init__(args)
        self.pyramid = args['pyramid']
        self.relative_downsample = self.pyramid['relative_downsample']
        self.pyramid_weight = self.pyramid['weight']
        self.num_levels = len(self.relative_downsample)

    def forward(self, output_dict, target_dict, suffix=""):
        target_dict = resolve_deferred_labels(target_dict)   # the occupancy loss reads the anchor labels as well
        if output_dict['pyramid'] == 'collab':
            return self.forward_collab(output_dict, target_dict, suffix)
        if output_dict['pyramid'] == 'single':
            return self.forward_single(output_dict, target_dict, suffix)
        raise RuntimeError("output_dict['pyramid'] must be 'collab' or 'single'")

    def _occ(self, output_dict, target_dict):
        return self.calc_occ_loss(output_dict['occ_single_list'], target_dict['pos_equal_one'],
                                  target_dict['neg_equal_one'], target_dict['pos_equal_one'].shape[0])

    def forward_single(self, output_dict, target_dict, suffix):
        """heter_pyramid_single (:29-45)."""
        total_loss = PointPillarDepthLoss.forward(self, output_dict, target_dict, suffix)
        occ_loss = self._occ(output_dict, target_dict)
        total_loss += occ_loss
        self.loss_dict.update({'pyramid_loss': occ_loss.item(), 'total_loss': total_loss.item()})
        return total_loss

    def forward_collab(self, output_dict, target_dict, suffix):
        """heter_pyramid_collab (:47-68): fused heads with suffix "", the per-agent occupancy maps with "_single"."""
        if suffix == "":
            return PointPillarDepthLoss.forward(self, output_dict, target_dict)
        assert suffix == "_single"
        occ_loss = self._occ(output_dict, target_dict)
        self.loss_dict = {'pyramid_loss': occ_loss.item(), 'total_loss': occ_loss.item()}
        return occ_loss

    def calc_occ_loss(self, occ_single_list, positives, negatives, batch_size):
        """:71-107.  A cell is foreground if either anchor is positive, background if both anchors are negative; level i
        pools foreground with max and background with min over relative_downsample[i] x relative_downsample[i]."""
        total = 0
        fg = torch.logical_or(positives[..., 0], positives[..., 1]).unsqueeze(-1).float()
        bg = torch.logical_and(negatives[..., 0], negatives[..., 1]).unsqueeze(-1).float()
        for i, occ in enumerate(occ_single_list):
            k = self.relative_downsample[i]
            pos = F.max_pool2d(fg.permute(0, 3, 1, 2), kernel_size=k).permute(0, 2, 3, 1).view(batch_size, -1, 1)
            neg = (1 - F.max_pool2d((1 - bg).permute(0, 3, 1, 2), kernel_size=k).permute(0, 2, 3, 1)).view(batch_size, -1, 1)
            pos_normalizer = pos.sum(1, keepdim=True).float()
            preds = occ.permute(0, 2, 3, 1).contiguous().view(batch_size, -1, 1)
            weights = pos * self.pos_cls_weight + neg * 1.0
            weights /= torch.clamp(pos_normalizer, min=1.0)
            level = sigmoid_focal_loss(preds, pos, weights=weights, **self.cls).sum() / batch_size
            total += level * self.pyramid_weight[i]
        return total
