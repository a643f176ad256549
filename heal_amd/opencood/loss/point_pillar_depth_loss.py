"""PointPillarLoss + the depth-bin supervision of the Lift-Splat encoders (reference:
opencood/loss/point_pillar_depth_loss.py:10-181).  `depth_items*` entries of the model output are
(depth_logit [N,D,H,W], depth_gt_indices [N,H,W] (, fg_mask))."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from heal_amd.opencood.loss.point_pillar_loss import PointPillarLoss


class FocalLoss(nn.Module):
    """point_pillar_depth_loss.py:97-181: multi-class focal loss over the class axis 1, optional 3-tap smoothing of the
    one-hot target ([0.2, 0.9, 0.2]).  Unlike the reference the smoothing kernel is not pinned to "cuda" at
    construction: it follows the input's device."""

    def __init__(self, alpha, gamma=2.0, reduction='none', smooth_target=False, eps=None):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.smooth_target, self.eps = alpha, gamma, reduction, smooth_target, eps
        if smooth_target:
            self.smooth_kernel = nn.Conv1d(1, 1, kernel_size=3, stride=1, padding=1, bias=False)
            self.smooth_kernel.weight = nn.Parameter(torch.tensor([[[0.2, 0.9, 0.2]]]), requires_grad=False)

    def forward(self, input, target):
        D = input.shape[1]
        one_hot = F.one_hot(target, num_classes=D).to(input)
        if self.smooth_target:
            flat = self.smooth_kernel.to(input.device)(one_hot.view(-1, D).float().unsqueeze(1)).squeeze(1)
            one_hot = flat.view(*target.shape, D)
        one_hot = one_hot.permute(0, 3, 1, 2)
        focal = -self.alpha * torch.pow(-input.softmax(1) + 1.0, self.gamma) * input.log_softmax(1)
        loss = torch.einsum('bc...,bc...->b...', (one_hot, focal))
        if self.reduction == 'none':
            return loss
        if self.reduction == 'mean':
            return torch.mean(loss)
        if self.reduction == 'sum':
            return torch.sum(loss)
        raise NotImplementedError(f"Invalid reduction mode: {self.reduction}")


class PointPillarDepthLoss(PointPillarLoss):
    _LOG_FIELDS = PointPillarLoss._LOG_FIELDS + (("Depth Loss", "depth_loss", "Depth_loss"),)

    def __init__(self, args):
        super().__init__(args)
        self.depth = args['depth']
        self.depth_weight = self.depth['weight']
        self.smooth_target = bool(self.depth.get('smooth_target'))
        self.use_fg_mask = bool(self.depth.get('use_fg_mask'))
        self.fg_weight, self.bg_weight = 3.25, 0.25
        self.depth_loss_func = FocalLoss(alpha=0.25, gamma=2.0, reduction="none", smooth_target=self.smooth_target)

    def forward(self, output_dict, target_dict, suffix=""):
        total_loss = super().forward(output_dict, target_dict, suffix)
        all_depth_loss = 0
        for name in [k for k in output_dict.keys() if k.startswith(f"depth_items{suffix}")]:
            item = output_dict[name]
            depth_loss = self.depth_loss_func(item[0], item[1])
            if self.use_fg_mask:
                fg = item[-1]
                depth_loss *= (fg > 0) * self.fg_weight + (fg == 0) * self.bg_weight
            all_depth_loss += depth_loss.mean() * self.depth_weight
        total_loss += all_depth_loss
        self.loss_dict.update({'depth_loss': all_depth_loss})  # like the reference: 'total_loss' in the dict is not updated
        return total_loss
