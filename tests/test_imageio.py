"""Input pipeline (PIL-exact resize) and evaluator wire format (COCO RLE): oracle pinning, host functions of the C-ABI
(CPU) and the kernels (-m gpu).  Everything here is integer / byte work: the bar is bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import imageio as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "imageio_golden.npz")


def _cases(seed, n, lo=8, hi=200, out_hi=260):
    rng = np.random.default_rng(seed)
    for it in range(n):
        h, w = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        nh, nw = int(rng.integers(4, out_hi)), int(rng.integers(4, out_hi))
        if it % 5 == 0:
            nw = w
        if it % 7 == 0:
            nh = h
        yield rng.integers(0, 256, (h, w, 3), dtype=np.uint8), nh, nw


# ------------------------------------------------------------------------------------------------ oracle pinning (CPU)
def test_oracle_resize_matches_golden_pillow_outputs():
    g = np.load(GOLD)
    i = 0
    while f"in{i}" in g:
        nh, nw = g[f"size{i}"]
        assert np.array_equal(O.resize_bilinear_u8(g[f"in{i}"], int(nh), int(nw)), g[f"out{i}"]), i
        i += 1
    assert i >= 8


def test_oracle_resize_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    for img, nh, nw in _cases(0, 25):
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(O.resize_bilinear_u8(img, nh, nw), ref), (img.shape, nh, nw)
    img = np.random.default_rng(3).integers(0, 256, (700, 31, 3), dtype=np.uint8)          # 17.5x down-scaling
    ref = np.asarray(Image.fromarray(img).resize((9, 40), Image.BILINEAR))
    assert np.array_equal(O.resize_bilinear_u8(img, 40, 9), ref)


def test_oracle_rle_known_answers_and_round_trip():
    assert O.rle_encode(np.zeros((2, 3), np.uint8)) == [6] and O.rle_to_string([6]) == b"6"
    assert O.rle_encode(np.array([[0, 1], [1, 1]])) == [1, 3] and O.rle_to_string([1, 3]) == b"13"
    assert O.rle_encode(np.ones((2, 2), np.uint8)) == [0, 4]                   # a mask that starts with 1: empty zero run
    assert O.rle_to_string([5, 2, 7, 1]) == b"527O"                             # 4th count stored as 1 - 2 = -1 -> 'O'
    assert O.rle_to_string([40]) == b"X1"                                        # 40 = 8 + 32: low group 8 | continuation, then 1
    rng = np.random.default_rng(5)
    for _ in range(20):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        m = rng.random((h, w)) > rng.random()
        c = O.rle_encode(m)
        assert sum(c) == h * w and all(x > 0 for x in c[1:])
        assert O.rle_from_string(O.rle_to_string(c)) == c
        assert np.array_equal(O.rle_decode(c, h, w), m)


# ------------------------------------------------------------------------------------------------ host side of the C-ABI (CPU)
def test_host_coefficients_and_rle_string_match_the_oracle():
    from ape_amd import ops

    rng = np.random.default_rng(1)
    for _ in range(60):
        a, b = int(rng.integers(1, 3000)), int(rng.integers(1, 1500))
        bo, ko = O.precompute_coeffs(a, b)
        bn, kn = ops.resize_coeffs(a, b)
        assert np.array_equal(bn.numpy(), bo) and np.array_equal(kn.numpy(), ko), (a, b)
    for _ in range(30):
        c = rng.integers(0, 3000000, int(rng.integers(1, 60))).tolist()
        assert ops.rle_to_string(c) == O.rle_to_string(c)
    assert ops.rle_to_string([5, 2, 7, 1]) == b"527O"


def test_shortest_edge_size_matches_oracle():
    from ape_amd.engine import shortest_edge_size

    rng = np.random.default_rng(2)
    for _ in range(200):
        h, w = int(rng.integers(16, 5000)), int(rng.integers(16, 5000))
        assert shortest_edge_size(h, w, 1024, 1024) == O.shortest_edge_size(h, w, 1024, 1024)
        assert shortest_edge_size(h, w, 800, 1333) == O.shortest_edge_size(h, w, 800, 1333)


def test_predictor_preprocess_host_logic(fake_ops):
    """DefaultPredictor.preprocess (ops swapped for their definitions) == the reference's host pipeline (defaults.py:213-220)"""
    from ape_amd.engine import DefaultPredictor

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

    rng = np.random.default_rng(4)
    for fmt in ("RGB", "BGR"):
        pred = DefaultPredictor(model=_M(), short_edge_length=96, max_size=128, input_format=fmt)
        for h, w in [(60, 90), (200, 100), (96, 96), (300, 1000)]:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            got = pred.preprocess(img)
            ref = O.predictor_input(img, 96, 128, fmt)
            assert got.dtype == torch.float32 and np.array_equal(got.numpy(), ref), (fmt, h, w)


def test_instances_to_coco_json_host_logic(fake_ops):
    from ape_amd import evaluation
    from ape_amd.structures import make_instances

    rng = np.random.default_rng(6)
    masks = torch.from_numpy(rng.random((3, 20, 30)) > 0.6)
    masks[1] = False
    boxes = torch.tensor([[1.0, 2.0, 11.0, 22.0], [0.0, 0.0, 5.0, 5.0], [3.0, 4.0, 6.0, 9.0]])
    inst = make_instances((20, 30), boxes, torch.tensor([0.9, 0.5, 0.1]), torch.tensor([7, 0, 3]), masks)
    monkey_cuda = torch.Tensor.to                                     # encode_masks uploads host masks; on CPU keep them
    try:
        torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else monkey_cuda(self, *a, **k)
        res = evaluation.instances_to_coco_json(inst, 42)
    finally:
        torch.Tensor.to = monkey_cuda
    assert len(res) == 3 and res[0]["image_id"] == 42 and res[0]["category_id"] == 7
    assert res[0]["bbox"] == [1.0, 2.0, 10.0, 20.0] and abs(res[2]["score"] - 0.1) < 1e-6
    for k in range(3):
        seg = res[k]["segmentation"]
        assert seg["size"] == [20, 30] and isinstance(seg["counts"], str)
        c = O.rle_from_string(seg["counts"].encode())
        assert np.array_equal(O.rle_decode(c, 20, 30), masks[k].numpy())
    assert res[1]["segmentation"]["counts"] == O.rle_to_string([600]).decode()


# ------------------------------------------------------------------------------------------------ kernels (GPU)
@pytest.mark.gpu
def test_resize_kernel_bit_exact_with_pillow():
    from PIL import Image
    from ape_amd import ops

    n = 0
    for img, nh, nw in list(_cases(7, 30)) + list(_cases(8, 6, lo=300, hi=900, out_hi=700)):
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)) if (nh, nw) != img.shape[:2] else img
        d = torch.from_numpy(img).cuda()
        got = ops.resize_bilinear_u8(d, nh, nw)
        assert np.array_equal(got.cpu().numpy(), ref), (img.shape, nh, nw)
        chw = ops.resize_bilinear_u8(d, nh, nw, float_chw=True, flip=True)
        assert np.array_equal(chw.cpu().numpy(), ref[:, :, ::-1].transpose(2, 0, 1).astype(np.float32))
        n += 1
    assert n == 36
    g = np.load(GOLD)                                                          # the committed Pillow outputs as well
    for i in range(8):
        nh, nw = (int(v) for v in g[f"size{i}"])
        assert np.array_equal(ops.resize_bilinear_u8(torch.from_numpy(g[f"in{i}"]).cuda(), nh, nw).cpu().numpy(), g[f"out{i}"])


@pytest.mark.gpu
def test_resize_kernel_camera_sized_image_into_canvas():
    """a 12 MP image -> long side 1024 (ResizeShortestEdge(1024, 1024)), written into the corner of the 1024^2 canvas"""
    from PIL import Image
    from ape_amd import ops
    from ape_amd.engine import shortest_edge_size

    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((4000, 3000), Image.BICUBIC))           # structured content
    nh, nw = shortest_edge_size(3000, 4000, 1024, 1024)
    assert (nh, nw) == (768, 1024)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    canvas = torch.full((3, 1024, 1024), -1.0, device="cuda")
    ops.resize_bilinear_u8(torch.from_numpy(img).cuda(), nh, nw, float_chw=True, out=canvas[:, :nh, :nw])
    assert np.array_equal(canvas[:, :nh, :nw].cpu().numpy(), ref.transpose(2, 0, 1).astype(np.float32))
    assert (canvas[:, nh:, :] == -1).all()
    # extreme down-scaling: the tile height shrinks until the source rows fit the LDS
    tall = rng.integers(0, 256, (5000, 64, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(tall).resize((32, 50), Image.BILINEAR))
    assert np.array_equal(ops.resize_bilinear_u8(torch.from_numpy(tall).cuda(), 50, 32).cpu().numpy(), ref)


def _blobs(rng, n, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((n, h, w), bool)
    for i in range(n):
        for _ in range(int(rng.integers(1, 4))):
            cy, cx, ry, rx = rng.random() * h, rng.random() * w, (0.02 + 0.3 * rng.random()) * h, (0.02 + 0.3 * rng.random()) * w
            out[i] |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
    return out


@pytest.mark.gpu
def test_rle_kernel_matches_oracle():
    from ape_amd import ops

    rng = np.random.default_rng(10)
    for (h, w) in [(1, 1), (7, 5), (128, 256), (129, 257), (300, 500), (427, 640)]:
        m = np.concatenate([_blobs(rng, 3, h, w), (rng.random((2, h, w)) > 0.5), np.zeros((1, h, w), bool), np.ones((1, h, w), bool)])
        cap = h * w + 1
        counts, nruns = ops.rle_encode(torch.from_numpy(m).cuda(), cap=max(cap, 2))
        counts, nruns = counts.cpu().numpy(), nruns.cpu().numpy()
        for i in range(m.shape[0]):
            ref = O.rle_encode(m[i])
            assert nruns[i] == len(ref), (h, w, i)
            assert counts[i, : nruns[i]].tolist() == ref, (h, w, i)
            assert ops.rle_to_string(torch.from_numpy(counts[i, : nruns[i]].copy())) == O.rle_to_string(ref)
    c, n = ops.rle_encode(torch.zeros((0, 8, 8), dtype=torch.uint8, device="cuda"))
    assert c.shape[0] == 0 and n.shape[0] == 0


@pytest.mark.gpu
def test_rle_full_size_round_trip_and_truncation():
    from ape_amd import evaluation, ops

    rng = np.random.default_rng(12)
    m = _blobs(rng, 6, 1024, 1024)
    m[5] = rng.random((1024, 1024)) > 0.5                                   # ~0.5 M runs: exceeds every sensible buffer
    d = torch.from_numpy(m).cuda()
    counts, nruns = ops.rle_encode(d, cap=4096)
    nr = nruns.cpu().numpy()
    assert nr[5] > 4096 and (nr[:5] <= 4096).all()                          # truncated encoding is reported, not silently cut
    hc = counts.cpu().numpy()
    for i in range(5):
        c = hc[i, : nr[i]].tolist()
        assert sum(c) == 1024 * 1024
        assert np.array_equal(O.rle_decode(c, 1024, 1024), m[i])
    rles = evaluation.encode_masks(d)                                        # grows the buffer for the noisy mask
    for i in (0, 5):
        c = O.rle_from_string(rles[i]["counts"])
        assert rles[i]["size"] == [1024, 1024] and np.array_equal(O.rle_decode(c, 1024, 1024), m[i])
