"""Run one of HEAL's own, unmodified drivers (opencood/tools/inference.py, train.py, ...) on the MI355X implementation.

    python examples/run_reference_tool.py /path/to/HEAL inference --model_dir <dir> --fusion_method intermediate

heal_amd.compat.overlay_reference() keeps the checkout's `opencood` package (datasets, evaluation, visualisation, the
driver itself) and substitutes, under the reference's module names, the model files, losses, pcdet IoU/NMS API and the
voxel pre / post processors of this repo.  spconv, the CUDA extensions and the Cython module of the checkout are never
imported.  (tests/test_overlay.py checks the import / discovery side of this in the build container; a dataset is
needed to go further.)"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    checkout, tool = os.path.abspath(sys.argv[1]), sys.argv[2]
    if not os.path.isdir(os.path.join(checkout, "opencood")):
        raise SystemExit(f"{checkout} does not contain an `opencood` package")
    # DataLoader workers are forked and cannot create a HIP context: voxelisation and label assignment are deferred to
    # the main process (SpVoxelPreprocessor / VoxelPostprocessor `defer` mode)
    os.environ.setdefault("HEAL_DEFER_VOXELIZE", "1")
    if tool.startswith("inference"):
        # the inference drivers never read the anchor labels the datasets build per sample: skip the assignment.  NOT for
        # train.py, whose validation pass computes the loss on them (train.py:153-154)
        os.environ.setdefault("HEAL_INFERENCE_ONLY", "1")
    from heal_amd import compat
    compat.overlay_reference(checkout)
    sys.argv = [os.path.join(checkout, "opencood", "tools", tool + ".py")] + sys.argv[3:]
    runpy.run_module("opencood.tools." + tool, run_name="__main__", alter_sys=True)


if __name__ == "__main__":
    main()
