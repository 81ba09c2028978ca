"""LiDAR (sparse-voxel + point) branch of the U2MKD hot path on MI355X.

Mirrors the reference's operator-level wiring on top
of ``u2mkd_amd.torchsparse``; module / parameter names match the reference so
its checkpoints load unchanged.
"""
from .point_voxel import (initial_voxelize, point_to_voxel, voxel_to_point, fetch_idx, prepare_geometry,
                          SparseSyncBatchNorm)
from .blocks import BasicConvolutionBlock, BasicDeconvolutionBlock, ResidualBlock
from .spvcnn import SPVCNN
from .sphereformer import SphereFormer
from .spvcnn_spformer import SPVCNN_SPFORMER, spformer_kwargs

__all__ = ['initial_voxelize', 'point_to_voxel', 'voxel_to_point', 'fetch_idx', 'SparseSyncBatchNorm',
           'BasicConvolutionBlock', 'BasicDeconvolutionBlock', 'ResidualBlock', 'SPVCNN', 'SphereFormer',
           'SPVCNN_SPFORMER', 'spformer_kwargs']
