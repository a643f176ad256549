"""Old-style intermediate-fusion PointPillars: host mirror of
opencood/models/point_pillar_baseline.py:16-135.  fusion_method max / att / v2xvit (the fusion modules on the hot
path); disconet / v2vnet are outside the scope and raise."""
from heal_amd.opencood.models.fuse_modules.fusion_in_one import build_fusion
from heal_amd.opencood.models.point_pillar import _PillarDetector
from heal_amd.opencood.models.sub_modules.naive_compress import NaiveCompressor
from heal_amd.opencood.utils.transformation_utils import normalize_pairwise_tfm


class PointPillarBaseline(_PillarDetector):
    def __init__(self, args):
        super().__init__(args)
        self.fusion_net = build_fusion(args)
        if args.get("backbone_fix", False):
            self.backbone_fix()

    def before_heads(self, args):
        self.compression = "compression" in args
        if self.compression:
            self.naive_compressor = NaiveCompressor(self.out_channel, args["compression"])

    def backbone_fix(self):
        """point_pillar_baseline.py:71-97: freeze everything but the fusion net (fine-tuning switch)."""
        frozen = [self.pillar_vfe, self.scatter, self.backbone, self.cls_head, self.reg_head]
        if self.compression:
            frozen.append(self.naive_compressor)
        if self.shrink_flag:
            frozen.append(self.shrink_conv)
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, data_dict):
        canvas, x = self.bev_features(data_dict)
        affine_matrix = normalize_pairwise_tfm(data_dict["pairwise_t_matrix"], canvas.shape[2], canvas.shape[3],
                                               self.voxel_size[0])
        if self.compression:
            x = self.naive_compressor(x)
        return self.predictions(self.fusion_net(x, data_dict["record_len"], affine_matrix))
