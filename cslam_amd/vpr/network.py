"""CosPlace network (cslam/vpr/cosplace_utils/network.py:29-45): backbone → L2Norm → GeM → Flatten
→ Linear → L2Norm.  Same module tree as the reference, so its checkpoints load unchanged."""
import torch
import torch.nn as nn

from .backbones import resnet_trunk, vgg16_features
from .layers import Flatten, GeM, L2Norm


def get_backbone(backbone_name):
    if backbone_name.startswith("ResNet"):
        return resnet_trunk(backbone_name)
    if backbone_name == "VGG16":
        return vgg16_features(cut_last=2)
    raise ValueError(f"backbone {backbone_name!r} is not available without torchvision "
                     "(ResNet18/50/101/152 and VGG16 are)")


class GeoLocalizationNet(nn.Module):
    def __init__(self, backbone, fc_output_dim):
        super().__init__()
        self.backbone, features_dim = get_backbone(backbone)
        self.aggregation = nn.Sequential(L2Norm(), GeM(), Flatten(), nn.Linear(features_dim, fc_output_dim),
                                         L2Norm())

    def forward(self, image: torch.Tensor):
        return self.aggregation(self.backbone(image)).detach()
