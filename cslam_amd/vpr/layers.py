"""Aggregation layers of the two VPR heads (cslam/vpr/cosplace_utils/layers.py:12-39,
cslam/vpr/netvlad.py:28-129).  Parameter names follow the reference (checkpoint compatible)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class GeM(nn.Module):
    """Generalised-mean pooling over the spatial dims: (mean(clamp(x, eps)^p))^(1/p), p learnt
    (layers.py:12-19)."""

    def __init__(self, p=3.0, eps=1e-6):
        super().__init__()
        self.p = nn.Parameter(torch.ones(1) * p)
        self.eps = eps

    def forward(self, x):
        pooled = x.clamp(min=self.eps).pow(self.p).mean(dim=(-2, -1), keepdim=True)
        return pooled.pow(1.0 / self.p)


class Flatten(nn.Module):
    """[B, C, 1, 1] → [B, C] (layers.py:25-30)."""

    def forward(self, x):
        return x[:, :, 0, 0]


class L2Norm(nn.Module):
    """x / max(||x||_2, 1e-12) along `dim` (layers.py:33-38)."""

    def __init__(self, dim=1):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return F.normalize(x, p=2.0, dim=self.dim)


class NetVLADLayer(nn.Module):
    """NetVLAD pooling (netvlad.py:28-129): soft-assign every local descriptor to K clusters by a
    1x1 convolution + softmax, accumulate the assignment-weighted residuals to the centroids,
    intra-normalise, flatten, L2-normalise.

    The reference loops over the clusters to save memory; here the residual sum is expanded,
        V[n,k,:] = sum_s a[n,k,s] x[n,:,s]  -  c[k,:] * sum_s a[n,k,s],
    i.e. one batched GEMM [K x S] x [S x C] per image (hipBLASLt through torch) + a rank-1 term."""

    def __init__(self, num_clusters=64, dim=128, normalize_input=True, vladv2=False):
        super().__init__()
        self.num_clusters = num_clusters
        self.dim = dim
        self.alpha = 0
        self.vladv2 = vladv2
        self.normalize_input = normalize_input
        self.conv = nn.Conv2d(dim, num_clusters, kernel_size=(1, 1), bias=vladv2)
        self.centroids = nn.Parameter(torch.rand(num_clusters, dim))

    def forward(self, x):
        n, c = x.shape[:2]
        if self.normalize_input:
            x = F.normalize(x, p=2, dim=1)
        soft_assign = F.softmax(self.conv(x).view(n, self.num_clusters, -1), dim=1)   # [N, K, S]
        x_flat = x.view(n, c, -1)                                                     # [N, C, S]
        vlad = torch.bmm(soft_assign, x_flat.transpose(1, 2))                         # [N, K, C]
        vlad = vlad - self.centroids.unsqueeze(0) * soft_assign.sum(dim=2, keepdim=True)
        vlad = F.normalize(vlad, p=2, dim=2)
        vlad = vlad.reshape(n, -1)
        return F.normalize(vlad, p=2, dim=1)
