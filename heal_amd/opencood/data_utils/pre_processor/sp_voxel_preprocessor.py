"""SpVoxelPreprocessor (reference: opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:18-174)
with the spconv CPU voxel generator replaced by the gfx950 voxeliser K1 (heal_voxelize).

`preprocess(pcd_np)` keeps the reference contract -- numpy in, dict of numpy out -- so it costs one
host->device and one device->host copy; the model-side fast path feeds device point clouds straight
to the encoder instead (heter_encoders.PointPillar, key 'points').

Deferred mode (`preprocess.args.defer_to_device: true` in the YAML, or HEAL_DEFER_VOXELIZE=1): `preprocess` does not
touch the GPU -- it runs inside the datasets' forked DataLoader workers, where a HIP context cannot be created -- and
returns the cloud itself; `collate_batch` turns the clouds into a list of tensors, `train_utils.to_device` moves them
(0.9 MB per agent instead of the 4-16 MB of padded voxels), and the encoder voxelises every agent of the modality in
one launch chain on the device (K1, heter_encoders 'points' path) with the caps carried along in the dictionary."""
import os
import sys

import numpy as np
import torch

from heal_amd import ops


class SpVoxelPreprocessor:
    def __init__(self, preprocess_params, train, device="cuda"):
        self.params = preprocess_params
        self.train = train
        self.device = torch.device(device)
        self.lidar_range = self.params['cav_lidar_range']
        self.voxel_size = self.params['args']['voxel_size']
        self.max_points_per_voxel = self.params['args']['max_points_per_voxel']
        self.max_voxels = self.params['args']['max_voxel_train'] if train else self.params['args']['max_voxel_test']
        grid_size = (np.array(self.lidar_range[3:6]) - np.array(self.lidar_range[0:3])) / np.array(self.voxel_size)
        self.grid_size = np.round(grid_size).astype(np.int64)
        self.defer = bool(self.params['args'].get('defer_to_device', False)) or \
            os.environ.get("HEAL_DEFER_VOXELIZE", "0") == "1"

    def preprocess_device(self, points, batch_idx=0):
        """points: [N,4] f32 device tensor -> device tensors (voxels, coords (b,z,y,x), num_points)."""
        return ops.voxelize(points, self.lidar_range, self.voxel_size, self.max_points_per_voxel,
                            self.max_voxels, batch_idx=batch_idx, sync=True)

    def preprocess(self, pcd_np):
        if self.defer:
            return {'points': np.ascontiguousarray(pcd_np[:, :4], dtype=np.float32),
                    'max_points_per_voxel': int(self.max_points_per_voxel), 'max_voxels': int(self.max_voxels)}
        pts = torch.from_numpy(np.ascontiguousarray(pcd_np[:, :4], dtype=np.float32)).to(self.device)
        voxels, coords, num = self.preprocess_device(pts)
        return {'voxel_features': voxels.cpu().numpy(),
                'voxel_coords': coords[:, 1:].contiguous().cpu().numpy(),  # (z,y,x) like spconv
                'voxel_num_points': num.cpu().numpy()}

    def collate_batch(self, batch):
        if isinstance(batch, list):
            return self.collate_batch_list(batch)
        if isinstance(batch, dict):
            return self.collate_batch_dict(batch)
        sys.exit('Batch has too be a list or a dictionarn')

    @staticmethod
    def _collate_deferred(clouds, max_points, max_voxels):
        first = lambda v: int(v[0] if isinstance(v, (list, tuple)) else v)  # noqa: E731 - merged dictionaries hold lists
        return {'points': [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)) for c in clouds],
                'max_points_per_voxel': first(max_points), 'max_voxels': first(max_voxels)}

    @staticmethod
    def collate_batch_list(batch):
        if batch and 'points' in batch[0]:
            return SpVoxelPreprocessor._collate_deferred([b['points'] for b in batch], batch[0]['max_points_per_voxel'],
                                                         batch[0]['max_voxels'])
        feats = [b['voxel_features'] for b in batch]
        nums = [b['voxel_num_points'] for b in batch]
        coords = [np.pad(b['voxel_coords'], ((0, 0), (1, 0)), mode='constant', constant_values=i)
                  for i, b in enumerate(batch)]
        return {'voxel_features': torch.from_numpy(np.concatenate(feats)),
                'voxel_coords': torch.from_numpy(np.concatenate(coords)),
                'voxel_num_points': torch.from_numpy(np.concatenate(nums))}

    @staticmethod
    def collate_batch_dict(batch):
        if 'points' in batch:
            return SpVoxelPreprocessor._collate_deferred(batch['points'], batch['max_points_per_voxel'], batch['max_voxels'])
        coords = [np.pad(c, ((0, 0), (1, 0)), mode='constant', constant_values=i)
                  for i, c in enumerate(batch['voxel_coords'])]
        return {'voxel_features': torch.from_numpy(np.concatenate(batch['voxel_features'])),
                'voxel_coords': torch.from_numpy(np.concatenate(coords)),
                'voxel_num_points': torch.from_numpy(np.concatenate(batch['voxel_num_points']))}
