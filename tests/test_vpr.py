"""VPR front step: the aggregation heads against outputs of the
reference's own layer classes (tests/golden/vpr_ref.npz, produced by gen_golden.gen_vpr), the
backbones' checkpoint-compatible structure, and — on the GPU — descriptors going from the head
into the kNN pool without touching the host.

Tolerance: fp32 torch ops in a different association order (GeM via mean instead of avg_pool2d,
NetVLAD residuals as one GEMM minus a rank-1 term instead of a per-cluster loop) → 2e-6 on
unit-norm outputs.  The backbones cannot be pinned (torchvision is not in the image): structure
and parameter names only."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "vpr_ref.npz"))


def test_cosplace_head_matches_reference_layers():
    from cslam_amd.vpr.layers import Flatten, GeM, L2Norm
    fc = torch.nn.Linear(96, 40)
    gem = GeM()
    with torch.no_grad():
        fc.weight.copy_(torch.from_numpy(G["cos_fc_w"]))
        fc.bias.copy_(torch.from_numpy(G["cos_fc_b"]))
        gem.p.fill_(float(G["cos_p"]))
        x = torch.from_numpy(G["cos_feats"])
        pooled = torch.nn.Sequential(L2Norm(), gem, Flatten())(x)
        out = torch.nn.Sequential(L2Norm(), gem, Flatten(), fc, L2Norm())(x)
    np.testing.assert_allclose(pooled.numpy(), G["cos_gem"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(out.numpy(), G["cos_out"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(out.numpy(), axis=1), 1.0, atol=1e-6)


@pytest.mark.parametrize("tag,v2", [("v1", False), ("v2", True)])
def test_netvlad_layer_matches_reference(tag, v2):
    from cslam_amd.vpr.layers import NetVLADLayer
    layer = NetVLADLayer(num_clusters=6, dim=24, vladv2=v2).eval()
    with torch.no_grad():
        layer.conv.weight.copy_(torch.from_numpy(G[f"vlad_{tag}_conv_w"]))
        if v2:
            layer.conv.bias.copy_(torch.from_numpy(G[f"vlad_{tag}_conv_b"]))
        layer.centroids.copy_(torch.from_numpy(G[f"vlad_{tag}_centroids"]))
        out = layer(torch.from_numpy(G[f"vlad_{tag}_x"]))
    np.testing.assert_allclose(out.numpy(), G[f"vlad_{tag}_out"], rtol=0, atol=2e-6)
    # parameter names of the reference's layer (checkpoint compatibility)
    assert set(layer.state_dict()) == {"centroids", "conv.weight"} | ({"conv.bias"} if v2 else set())


def test_geolocalization_net_structure_is_checkpoint_compatible():
    from cslam_amd.vpr import GeoLocalizationNet
    net = GeoLocalizationNet("ResNet18", 64).eval()
    keys = list(net.state_dict())
    # `list(resnet18.children())[:-2]` as a Sequential + the aggregation head (network.py:33-39,69)
    for k in ("backbone.0.weight", "backbone.1.running_var", "backbone.4.0.conv1.weight",
              "backbone.5.0.downsample.0.weight", "backbone.7.1.bn2.bias", "aggregation.1.p",
              "aggregation.3.weight", "aggregation.3.bias"):
        assert k in keys, k
    assert not any(k.startswith("backbone.8") for k in keys)          # no avgpool / fc
    n_backbone = sum(v.numel() for k, v in net.state_dict().items()
                     if k.startswith("backbone") and "num_batches" not in k and "running" not in k)
    assert n_backbone == 11_176_512                                     # torchvision resnet18 minus fc
    with torch.no_grad():
        out = net(torch.rand(2, 3, 96, 128))
    assert out.shape == (2, 64) and not out.requires_grad
    np.testing.assert_allclose(out.norm(dim=1).numpy(), 1.0, atol=1e-5)
    r50 = GeoLocalizationNet("ResNet50", 32)
    assert r50.aggregation[3].in_features == 2048
    assert sum(p.numel() for n, p in r50.named_parameters() if n.startswith("backbone")) == 23_508_032
    with pytest.raises(ValueError):
        GeoLocalizationNet("EfficientNet_B0", 32)


def test_preprocess_center_crop_resize_normalise():
    from cslam_amd.vpr.preprocess import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD, preprocess
    img = np.full((2, 300, 400, 3), 128, np.uint8)
    out = preprocess(img, 256, torch.device("cpu"))
    assert out.shape == (2, 3, 224, 224)
    for c in range(3):
        np.testing.assert_allclose(out[:, c].numpy(), (128 / 255 - IMAGENET_DEFAULT_MEAN[c]) / IMAGENET_DEFAULT_STD[c],
                                   atol=1e-5)
    # a crop larger than the image keeps the image; the smaller side becomes 224
    out = preprocess(np.zeros((100, 150, 3), np.uint8), 376, torch.device("cpu"))
    assert out.shape == (1, 3, 224, 336)


def test_disabled_matchers_return_random_descriptors_like_the_reference():
    from cslam_amd.vpr.cosplace import CosPlace
    from cslam_amd.vpr.netvlad import NetVLAD
    cp = CosPlace({"frontend.nn_checkpoint": "disable", "frontend.cosplace.descriptor_dim": 48}, node=None)
    assert cp.compute_embedding(np.zeros((8, 8, 3), np.uint8)).shape == (48,)
    nv = NetVLAD({"frontend.nn_checkpoint": "Disable"}, node=None)
    assert nv.compute_embedding(np.zeros((8, 8, 3), np.uint8)).shape == (128,)


def test_netvlad_pca_on_device_equals_sklearn_transform():
    from sklearn.decomposition import PCA
    from cslam_amd.vpr.layers import NetVLADLayer
    from cslam_amd.vpr.netvlad import NetVLAD
    rs = np.random.RandomState(0)
    pca = PCA(n_components=16, whiten=True).fit(rs.randn(200, 6 * 24).astype(np.float32))
    model = torch.nn.Module()
    model.add_module("encoder", torch.nn.Sequential(torch.nn.Conv2d(3, 24, 3, 4, 1), torch.nn.ReLU()))
    model.add_module("pool", NetVLADLayer(num_clusters=6, dim=24))
    nv = NetVLAD({"frontend.nn_checkpoint": "random-init", "frontend.image_crop_size": 64}, node=None,
                 model=model, pca=pca)
    img = rs.randint(0, 255, size=(3, 80, 90, 3)).astype(np.uint8)
    got = nv.compute_embeddings_device(img).cpu().numpy()
    with torch.no_grad():
        from cslam_amd.vpr.preprocess import preprocess
        vlad = model.pool(model.encoder(preprocess(img, 64, nv.device))).cpu().numpy()
    want = pca.transform(vlad)
    want /= np.linalg.norm(want, axis=1, keepdims=True)
    np.testing.assert_allclose(got, want, atol=2e-5)


@pytest.mark.gpu
def test_descriptors_flow_from_head_to_pool_without_host(gpu):
    """cfg5 front step: images → CosPlace head on the GPU → device rows appended to the local
    pool and matched against another robot's pool; results equal the reference-style host path
    (descriptor → numpy → add_item / search_best) bit for bit (the pool stores the same floats)."""
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.vpr import GeoLocalizationNet
    from cslam_amd.vpr.cosplace import CosPlace

    class _Log:
        def info(self, *a, **k):
            pass
        error = debug = warn = info

    class FakeNode:
        def get_logger(self):
            return _Log()
    torch.manual_seed(0)
    params = {"frontend.nn_checkpoint": "random-init", "frontend.cosplace.descriptor_dim": 64,
              "frontend.cosplace.backbone": "ResNet18", "frontend.image_crop_size": 200}
    cp = CosPlace(params, FakeNode(), model=GeoLocalizationNet("ResNet18", 64))
    rs = np.random.RandomState(1)
    images = rs.randint(0, 255, size=(6, 240, 320, 3)).astype(np.uint8)
    emb = cp.compute_embeddings_device(images)
    assert emb.is_cuda and emb.shape == (6, 64)
    np.testing.assert_allclose(emb.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    # per-image path of the reference API gives the same descriptor
    np.testing.assert_allclose(cp.compute_embedding(images[2]), emb[2].cpu().numpy(), atol=2e-6)

    lp = {"robot_id": 0, "max_nb_robots": 2, "frontend.sensor_type": "stereo",
          "frontend.similarity_threshold": -1.0, "frontend.nb_best_matches": 10,
          "frontend.intra_loop_min_inbetween_keyframes": 2}
    a = LoopClosureSparseMatching(dict(lp), FakeNode())     # device path
    b = LoopClosureSparseMatching(dict(lp), FakeNode())     # host path
    other = rs.rand(9, 64).astype(np.float32)
    for lc in (a, b):
        lc.other_robots_nnsm[1].add_items(other, list(range(100, 109)))
    got = a.add_local_global_descriptors_device(emb, range(6))
    want = []
    emb_host = emb.cpu().numpy()
    for k in range(6):
        want += b.add_local_global_descriptor(emb_host[k], k)
    assert [tuple(e) for e in got] == [tuple(e) for e in want]
    np.testing.assert_array_equal(np.asarray(a.local_nnsm.data), np.asarray(b.local_nnsm.data))
    assert a.candidate_selector.candidate_edges.keys() == b.candidate_selector.candidate_edges.keys()


@pytest.mark.gpu
def test_heads_on_the_gpu_match_the_reference_layers(gpu):
    """The aggregation heads run on the GPU (PyTorch-ROCm) in the product: CosPlace's L2Norm / GeM /
    Linear / L2Norm and both NetVLAD variants on cuda:0 against the outputs of the REFERENCE's own
    layer classes (tests/golden/vpr_ref.npz, gen_golden.gen_vpr) - not against this package's CPU
    path.  fp32 on rocBLAS / MIOpen in another association order: 5e-6 on unit-norm outputs."""
    from cslam_amd.vpr.layers import Flatten, GeM, L2Norm, NetVLADLayer
    dev = torch.device("cuda:0")
    fc = torch.nn.Linear(96, 40).to(dev)
    gem = GeM().to(dev)
    with torch.no_grad():
        fc.weight.copy_(torch.from_numpy(G["cos_fc_w"]))
        fc.bias.copy_(torch.from_numpy(G["cos_fc_b"]))
        gem.p.fill_(float(G["cos_p"]))
        x = torch.from_numpy(G["cos_feats"]).to(dev)
        pooled = torch.nn.Sequential(L2Norm(), gem, Flatten())(x)
        out = torch.nn.Sequential(L2Norm(), gem, Flatten(), fc, L2Norm())(x)
    assert out.is_cuda
    np.testing.assert_allclose(pooled.cpu().numpy(), G["cos_gem"], rtol=5e-5, atol=1e-7)
    np.testing.assert_allclose(out.cpu().numpy(), G["cos_out"], rtol=0, atol=5e-6)
    for tag, v2 in (("v1", False), ("v2", True)):
        layer = NetVLADLayer(num_clusters=6, dim=24, vladv2=v2).eval().to(dev)
        with torch.no_grad():
            layer.conv.weight.copy_(torch.from_numpy(G[f"vlad_{tag}_conv_w"]))
            if v2:
                layer.conv.bias.copy_(torch.from_numpy(G[f"vlad_{tag}_conv_b"]))
            layer.centroids.copy_(torch.from_numpy(G[f"vlad_{tag}_centroids"]))
            got = layer(torch.from_numpy(G[f"vlad_{tag}_x"]).to(dev))
        assert got.is_cuda
        np.testing.assert_allclose(got.cpu().numpy(), G[f"vlad_{tag}_out"], rtol=0, atol=5e-6)
