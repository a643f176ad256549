"""K8 timing: heal_decode_nms at BASELINE size (2 x 256 x 256 anchors, ~600 candidates above the score threshold) standalone:
event pair around the call and the per-call period inside a captured graph.  Usage: python scripts/k8_bench.py  (on the GPU box)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_amd import configs, ops  # noqa: E402
from heal_amd.pipeline import Scene, ScenePipeline  # noqa: E402


def main():
    hypes = configs.lidar_pyramid(max_cav=5)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st), torch.no_grad():
        pipe = ScenePipeline(hypes, "cuda:0", seed=0)
        scene = Scene(5, seed=4, device="cuda:0", modalities=["m1"] * 5)
        pipe.calibrate_cls_bias(scene)
        out = pipe.forward(scene)
        anchors = pipe.post._anchors_f32(pipe.anchor_box, pipe.device)
        P = pipe.post.params
        args = (out["cls_preds"], out["reg_preds"], out["dir_preds"], anchors, P["target_args"]["score_threshold"], 0.7853, 2,
                P["nms_thresh"], np.eye(4, dtype=np.float32), P["gt_range"])
        fn = lambda: ops.decode_nms(*args, sync=False)
        c, s, n = fn()
        torch.cuda.synchronize()
        res = {"kept": int(n.item()), "candidates": int((torch.sigmoid(out["cls_preds"]) > 0.2).sum())}
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res["event_pair_us"] = float(np.median(ts))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                fn()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        res["in_graph_us"] = float(np.median(ts))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
