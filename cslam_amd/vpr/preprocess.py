"""Image preprocessing of both VPR heads on the device (cosplace.py:65-71, netvlad.py:188-194:
CenterCrop(crop) → Resize(224, bicubic) → ToTensor → Normalize(ImageNet)).  The reference goes
through PIL on the host; this is the batched torch equivalent (bicubic with antialiasing, which
is what PIL's resize does when shrinking) — close to, not bit-identical with, PIL."""
import torch
import torch.nn.functional as F

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def preprocess(images, crop_size, device, out_size=224):
    """images: uint8 [B, H, W, 3] (numpy or tensor, RGB) → float32 [B, 3, out, out'] on `device`;
    the smaller side becomes `out_size` like torchvision's Resize(int)."""
    x = torch.as_tensor(images)
    if x.dim() == 3:
        x = x.unsqueeze(0)
    x = x.to(device, non_blocking=True).permute(0, 3, 1, 2).float().div_(255.0)
    h, w = x.shape[-2:]
    ch, cw = min(crop_size, h), min(crop_size, w)
    top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
    x = x[..., top:top + ch, left:left + cw]
    if ch <= cw:
        nh, nw = out_size, max(1, int(out_size * cw / ch))
    else:
        nh, nw = max(1, int(out_size * ch / cw)), out_size
    x = F.interpolate(x, size=(nh, nw), mode="bicubic", antialias=True, align_corners=False).clamp_(0.0, 1.0)
    mean = torch.tensor(IMAGENET_DEFAULT_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_DEFAULT_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std
