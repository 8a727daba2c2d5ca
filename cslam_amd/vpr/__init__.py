"""Visual place recognition front step on PyTorch-ROCm: the CosPlace
and NetVLAD heads of cslam/vpr/, producing global descriptors ON THE DEVICE so they can be written
straight into the kNN pool (`NearestNeighborsMatching.add_items_device`) instead of travelling
`descriptor.tolist()` → list → ndarray → torch like the reference
(global_descriptor_loop_closure_detection.py:161, loop_closure_sparse_matching.py:64).

The backbones are plain torch modules whose parameter names equal torchvision's, so the
reference's checkpoints load unchanged; torchvision itself is not needed."""
from .layers import GeM, Flatten, L2Norm, NetVLADLayer  # noqa: F401
from .network import GeoLocalizationNet  # noqa: F401
