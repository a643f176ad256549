"""Host mirror of the model-calling helpers of opencood/tools/inference_utils.py (the `serve()` analogue):
same names, arguments and return dicts.  `dataset` only has to provide `post_process(batch_data, output_dict) ->
(pred_box_tensor, pred_score, gt_box_tensor)` (and `post_process_no_fusion` for `inference_no_fusion`), as the
reference's datasets do by delegating to VoxelPostprocessor."""
from collections import OrderedDict


def _pack(pred_box_tensor, pred_score, gt_box_tensor):
    return {"pred_box_tensor": pred_box_tensor, "pred_score": pred_score, "gt_box_tensor": gt_box_tensor}


def inference_late_fusion(batch_data, model, dataset):
    """inference_utils.py:18-47: every cav runs the single-agent model; boxes are pooled in post_process."""
    output_dict = OrderedDict()
    for cav_id, cav_content in batch_data.items():
        output_dict[cav_id] = model(cav_content)
    return _pack(*dataset.post_process(batch_data, output_dict))


def inference_no_fusion(batch_data, model, dataset, single_gt=False):
    """inference_utils.py:51-86: the ego alone (labels from all cavs unless single_gt)."""
    if single_gt:
        batch_data = {'ego': batch_data['ego']}
    output_dict_ego = OrderedDict()
    output_dict_ego['ego'] = model(batch_data['ego'])
    return _pack(*dataset.post_process_no_fusion(batch_data, output_dict_ego))


def inference_early_fusion(batch_data, model, dataset):
    """inference_utils.py:123-153 (also the body of inference_intermediate_fusion, :156-174)."""
    output_dict = OrderedDict()
    output_dict['ego'] = model(batch_data['ego'])
    ret = _pack(*dataset.post_process(batch_data, output_dict))
    if "depth_items" in output_dict['ego']:
        ret.update({"depth_items": output_dict['ego']['depth_items']})
    return ret


def inference_intermediate_fusion(batch_data, model, dataset):
    """inference_utils.py:156-174."""
    return inference_early_fusion(batch_data, model, dataset)
