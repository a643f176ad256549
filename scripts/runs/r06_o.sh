python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "concurrent_modality" 2>&1 | grep -B30 "short test summary" | head -60
