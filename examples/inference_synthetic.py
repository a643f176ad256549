"""tools/inference.py-style driver written ONLY against the reference's import names (`opencood.*`), running on the
MI355X implementation through heal_amd.compat.install_as_opencood().

    python examples/inference_synthetic.py [--agents 3]

What it shows (route A): the YAML surface (`yaml_utils.load_yaml`), model discovery by name
(`train_utils.create_model`), `model(batch_data['ego'])`, and post-processing through
`inference_utils.inference_intermediate_fusion(batch, model, dataset)` -- none of which mentions heal_amd.
The dataset is a ten-line synthetic stand-in (the reference's datasets read OPV2V from disk: out of scope)."""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import heal_amd.compat as compat
from heal_amd import configs
from heal_amd.pipeline import Scene, calibrate_heads, fill_deterministic

compat.install_as_opencood()
from opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor  # noqa: E402
from opencood.hypes_yaml import yaml_utils  # noqa: E402
from opencood.tools import inference_utils, train_utils  # noqa: E402


class SyntheticDataset:
    """The two members the inference helpers use: a post-processor and post_process()."""

    def __init__(self, hypes, device):
        self.post_processor = VoxelPostprocessor(hypes["postprocess"], train=False)
        self.anchor_box = torch.from_numpy(self.post_processor.generate_anchor_box()).to(device)

    def post_process(self, batch_data, output_dict):
        pred, score = self.post_processor.post_process(batch_data, output_dict)
        return pred, score, None  # no ground truth in a synthetic scene


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    with tempfile.TemporaryDirectory() as d:  # the YAML round trip a real run starts with
        path = os.path.join(d, "lidar_pyramid.yaml")
        configs.dump_yaml(configs.lidar_pyramid(max_cav=max(5, a.agents)), path)
        hypes = yaml_utils.load_yaml(path)
    model = train_utils.create_model(hypes)        # HeterPyramidCollab, found by name like the reference does
    model = fill_deterministic(model, 3).to(dev).eval()   # stands in for load_state_dict(checkpoint)
    dataset = SyntheticDataset(hypes, dev)
    scene = Scene(a.agents, seed=11, device=dev)
    # untrained heads: shift the classification bias so that a realistic number of anchors pass the score threshold
    calibrate_heads(model, scene.model_input(), hypes["postprocess"]["target_args"]["score_threshold"], 400)
    batch = {"ego": dict(scene.model_input(), anchor_box=dataset.anchor_box,
                         transformation_matrix=torch.eye(4, device=dev))}
    batch = train_utils.to_device(batch, dev)
    with torch.no_grad():
        out = inference_utils.inference_intermediate_fusion(batch, model, dataset)
    boxes, scores = out["pred_box_tensor"], out["pred_score"]
    n = 0 if boxes is None else int(boxes.shape[0])
    print(f"{type(model).__module__}.{type(model).__name__}: {a.agents} agents -> {n} boxes"
          + ("" if n == 0 else f", best score {float(scores.max()):.3f}, corners tensor {tuple(boxes.shape)}"))


if __name__ == "__main__":
    main()
