"""Anchor grids/shifts (reference yolort/models/anchor_utils.py:9-67).

The fused post-process evaluates grids and shifts in closed form inside the decode kernel
(grid = cell (x, y); shift = anchor (w, h) in pixels), so this module only exists for API parity
(`YOLO(anchor_generator=...)`, the reference's known-answer test) and for custom post_process hooks.
It is index plumbing on the feature-map device, not arithmetic on activations.
"""
from typing import List, Tuple

import torch
from torch import Tensor, nn


class AnchorGenerator(nn.Module):
    def __init__(self, strides: List[int], anchor_grids: List[List[float]]):
        super().__init__()
        assert len(strides) == len(anchor_grids)
        self.strides = strides
        self.anchor_grids = anchor_grids
        self.num_layers = len(anchor_grids)
        self.num_anchors = len(anchor_grids[0]) // 2

    def _generate_grids(self, grid_sizes, dtype=torch.float32, device=torch.device("cpu")) -> List[Tensor]:
        grids = []
        for height, width in grid_sizes:
            xs = torch.arange(width, dtype=torch.int32, device=device).to(dtype)
            ys = torch.arange(height, dtype=torch.int32, device=device).to(dtype)
            gx = xs.view(1, width).expand(height, width)   # [..., 0] = x (column index)
            gy = ys.view(height, 1).expand(height, width)  # [..., 1] = y (row index)
            grids.append(torch.stack((gx, gy), 2).expand((1, self.num_anchors, height, width, 2)))
        return grids

    def _generate_shifts(self, grid_sizes, dtype=torch.float32, device=torch.device("cpu")) -> List[Tensor]:
        anchors = torch.as_tensor(self.anchor_grids, dtype=torch.float32, device=device).to(dtype)
        strides = torch.as_tensor(self.strides, dtype=torch.float32, device=device).to(dtype)
        anchors = anchors.view(self.num_layers, -1, 2) / strides.view(-1, 1, 1)
        shifts = []
        for i, (height, width) in enumerate(grid_sizes):
            s = (anchors[i].clone() * self.strides[i]).view(1, self.num_anchors, 1, 1, 2)
            shifts.append(s.expand(1, self.num_anchors, height, width, 2).contiguous().to(dtype))
        return shifts

    def forward(self, feature_maps: List[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
        grid_sizes = [fm.shape[-2:] for fm in feature_maps]
        dtype, device = feature_maps[0].dtype, feature_maps[0].device
        return self._generate_grids(grid_sizes, dtype, device), self._generate_shifts(grid_sizes, dtype, device)
