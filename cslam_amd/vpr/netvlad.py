"""NetVLAD matcher — host mirror of cslam/vpr/netvlad.py:132-236 (VGG16 conv5 features →
NetVLADLayer → PCA → L2) with the PCA applied on the device as one GEMM."""
import pickle
from os.path import isfile

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .backbones import vgg16_features
from .layers import NetVLADLayer
from .preprocess import preprocess


class NetVLAD(object):
    """NetVLAD matcher"""

    def __init__(self, params, node, model=None, pca=None):
        self.params = params
        self.node = node
        self.enable = self.params['frontend.nn_checkpoint'].lower() != 'disable'
        if not self.enable:
            return
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        if model is not None:
            self.model = model
        else:
            encoder, encoder_dim = vgg16_features(cut_last=2)
            self.model = nn.Module()
            self.model.add_module('encoder', encoder)
            self.model.add_module('pool', NetVLADLayer(num_clusters=64, dim=encoder_dim, vladv2=False))
            resume_ckpt = self.params['frontend.nn_checkpoint']
            if isfile(resume_ckpt):
                print("=> loading checkpoint '{}'".format(resume_ckpt))
                checkpoint = torch.load(resume_ckpt, map_location=lambda storage, loc: storage)
                self.model.load_state_dict(checkpoint['state_dict'])
                print("=> loaded checkpoint '{}' (epoch {})".format(resume_ckpt, checkpoint['epoch']))
            else:
                print("Error: Checkpoint path is incorrect")
        self.model = self.model.to(self.device)
        self.model.eval()
        if pca is None:
            pca = pickle.load(open(self.params['frontend.netvlad.pca_checkpoint'], 'rb'))
        # sklearn PCA.transform: (X - mean_) @ components_.T [/ sqrt(explained_variance_) if whiten]
        comp = np.asarray(pca.components_, dtype=np.float32)
        if getattr(pca, "whiten", False):
            comp = comp / np.sqrt(np.asarray(pca.explained_variance_, dtype=np.float32))[:, None]
        self.pca_components = torch.from_numpy(np.ascontiguousarray(comp.T)).to(self.device)
        mean = getattr(pca, "mean_", None)
        self.pca_mean = (torch.zeros(comp.shape[1]) if mean is None
                         else torch.from_numpy(np.asarray(mean, dtype=np.float32))).to(self.device)

    def compute_embeddings_device(self, keyframes):
        """uint8 [B, H, W, 3] → PCA-reduced, L2-normalised descriptors [B, d] on the device."""
        with torch.no_grad():
            x = preprocess(keyframes, self.params["frontend.image_crop_size"], self.device)
            vlad = self.model.pool(self.model.encoder(x))
            reduced = (vlad - self.pca_mean) @ self.pca_components
            return F.normalize(reduced, p=2, dim=1).float().contiguous()

    def compute_embedding(self, keyframe):
        """Load image to device and extract the global image descriptor (np.array)."""
        if self.enable:
            return self.compute_embeddings_device(keyframe)[0].cpu().numpy()
        # Random descriptor if disabled.  Use this option only for testing
        return np.random.rand(128)
