"""CosPlace matcher — host mirror of cslam/vpr/cosplace.py:27-100 with a device-resident,
batched path next to the reference's per-image one."""
from os.path import isfile

import numpy as np
import torch

from .network import GeoLocalizationNet
from .preprocess import preprocess


class CosPlace(object):
    """CosPlace matcher"""

    def __init__(self, params, node, model=None):
        """
        Args:
            params (dict): parameters (`frontend.nn_checkpoint`, `frontend.cosplace.descriptor_dim`,
                `frontend.cosplace.backbone`, `frontend.image_crop_size`)
            node: ROS 2 node (logger)
            model: an already built GeoLocalizationNet (tests / synthetic benchmarks); the
                reference always loads `frontend.nn_checkpoint`
        """
        self.params = params
        self.node = node
        self.enable = self.params['frontend.nn_checkpoint'].lower() != 'disable'
        if not self.enable:
            self.descriptor_dim = self.params.get('frontend.cosplace.descriptor_dim', 64)
            return
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.descriptor_dim = self.params['frontend.cosplace.descriptor_dim']
        if model is not None:
            self.model = model.to(self.device)
        else:
            self.model = GeoLocalizationNet(self.params['frontend.cosplace.backbone'], self.descriptor_dim)
            resume_ckpt = self.params['frontend.nn_checkpoint']
            if isfile(resume_ckpt):
                self.node.get_logger().info("loading checkpoint '{}'".format(resume_ckpt))
                checkpoint = torch.load(resume_ckpt, map_location=lambda storage, loc: storage)
                self.model.load_state_dict(checkpoint)
                self.model = self.model.to(self.device)
            else:
                self.node.get_logger().error("Error: Checkpoint path is incorrect {}".format(resume_ckpt))
                exit()
        self.model.eval()

    def compute_embeddings_device(self, keyframes):
        """Batch of images (uint8 [B, H, W, 3]) → descriptors [B, D] float32 ON THE DEVICE, ready
        for `NearestNeighborsMatching.add_items_device` (no host round trip)."""
        with torch.no_grad():
            x = preprocess(keyframes, self.params["frontend.image_crop_size"], self.device)
            return self.model(x).float().contiguous()

    def compute_embedding(self, keyframe):
        """Load image to device and extract the global image descriptor

        Args:
            keyframe (image): image to match

        Returns:
            np.array: global image descriptor
        """
        if self.enable:
            return self.compute_embeddings_device(keyframe)[0].cpu().numpy()
        # Random descriptor if disabled.  Use this option only for testing
        return np.random.rand(self.descriptor_dim)
