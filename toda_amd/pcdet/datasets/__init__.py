"""Dataset registry and dataloader builder (reference pcdet/datasets/__init__.py:25-97).  Real
dataset readers (KITTI / nuScenes / Waymo / ...) are out of scope: the registry holds the
synthetic dataset that exposes the same attributes and the same collate contract."""
from functools import partial

import torch
from torch.utils.data import DataLoader
from torch.utils.data import DistributedSampler as _DistributedSampler

from ..utils import common_utils
from .dataset import DatasetTemplate
from .synthetic import SyntheticLidarDataset, SyntheticPairDataset
from .mixup_dataset import SyntheticMixupPairDataset
from .two_dataset import SyntheticMixDataset

__all__ = {
    "DatasetTemplate": DatasetTemplate,
    "SyntheticLidarDataset": SyntheticLidarDataset,
    "SyntheticPairDataset": SyntheticPairDataset,
    "SyntheticMixDataset": SyntheticMixDataset,
    "SyntheticMixupPairDataset": SyntheticMixupPairDataset,
}


class DistributedSampler(_DistributedSampler):
    """Non-shuffling-capable sampler for evaluation (reference :45-65)."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        super().__init__(dataset, num_replicas=num_replicas, rank=rank)
        self.shuffle = shuffle

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            order = torch.randperm(len(self.dataset), generator=g).tolist()
        else:
            order = list(range(len(self.dataset)))
        order += order[:(self.total_size - len(order))]
        return iter(order[self.rank:self.total_size:self.num_replicas])


def build_dataloader(dataset_cfg, class_names, batch_size, dist, root_path=None, workers=4, logger=None,
                     training=True, merge_all_iters_to_one_epoch=False, total_epochs=0):
    dataset = __all__[dataset_cfg.DATASET](dataset_cfg=dataset_cfg, class_names=class_names, root_path=root_path,
                                           training=training, logger=logger)
    if merge_all_iters_to_one_epoch:
        dataset.merge_all_iters_to_one_epoch(merge=True, epochs=total_epochs)
    if dist:
        rank, world = common_utils.get_dist_info()
        sampler = _DistributedSampler(dataset) if training else DistributedSampler(dataset, world, rank, shuffle=False)
    else:
        sampler = None
    device_resident = bool(getattr(dataset, "on_device", False))   # samples are CUDA tensors: nothing to pin, no worker processes
    loader = DataLoader(dataset, batch_size=batch_size, pin_memory=not device_resident, num_workers=0 if device_resident else workers,
                        shuffle=(sampler is None) and training, collate_fn=dataset.collate_batch, drop_last=False,
                        sampler=sampler, timeout=0)
    return dataset, loader, sampler
