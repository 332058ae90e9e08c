"""ape_amd.evaluation.box_ap: known-answer checks of the COCO box-AP protocol (restated; pycocotools is not installable here)."""
import torch

from ape_amd.evaluation import box_ap


def test_hand_computed_case():
    """1 class, 2 ground-truth boxes; detections: 0.9 -> GT1 (IoU 1), 0.8 -> nothing, 0.7 -> GT2 with IoU 0.62.
    t <= 0.60: tp 1,1,2 / fp 0,1,1 -> recall .5,.5,1, precision 1,.5,.667 -> envelope 1,.667,.667: 51 recall points at 1 and
    50 at 2/3 -> 0.835; t >= 0.65: only the first detection matches -> 51 / 101.  AP = (3 * 0.83498 + 7 * 0.50495) / 10."""
    gt = [(torch.tensor([[0.0, 0.0, 100.0, 100.0], [200.0, 200.0, 300.0, 300.0]]), torch.tensor([3, 3]))]
    w = 62.0 / 1.0                      # [200, 200+w] x [200, 300] inside GT2: IoU = w / 100
    det = [(torch.tensor([[0.0, 0.0, 100.0, 100.0], [500.0, 500.0, 600.0, 600.0], [200.0, 200.0, 200.0 + w, 300.0]]),
            torch.tensor([0.9, 0.8, 0.7]), torch.tensor([3, 3, 3]))]
    r = box_ap(det, gt)
    want = (3 * (51 + 50 * 2.0 / 3.0) / 101 + 7 * 51 / 101) / 10
    assert abs(r["AP"] - want) < 1e-9 and abs(r["AP50"] - (51 + 50 * 2.0 / 3.0) / 101) < 1e-9 and abs(r["AP75"] - 51 / 101) < 1e-9
    assert r["classes"] == 1 and r["gt"] == 2 and r["dets"] == 3


def test_properties():
    g = torch.Generator().manual_seed(0)

    def img():
        xy, wh = torch.rand(30, 2, generator=g) * 500, torch.rand(30, 2, generator=g) * 100 + 10
        return torch.cat([xy, xy + wh], 1), torch.randint(0, 5, (30,), generator=g)

    gt = [img() for _ in range(4)]
    det = [(b, torch.rand(len(b), generator=g), c) for b, c in gt]
    assert box_ap(det, gt)["AP"] == 1.0                                               # the ground truth itself, any scores
    wrong = [(b, s, (c + 1) % 5) for b, s, c in det]
    assert box_ap(wrong, gt)["AP"] == 0.0                                             # right boxes, wrong classes
    fp_first = [(torch.cat([b + 1000, b]), torch.cat([s + 1, s]), torch.cat([c, c])) for b, s, c in det]
    assert abs(box_ap(fp_first, gt)["AP"] - 0.5) < 1e-12                              # every true positive behind as many false ones
    fp_last = [(torch.cat([b, b + 1000]), torch.cat([s + 1, s]), torch.cat([c, c])) for b, s, c in det]
    assert box_ap(fp_last, gt)["AP"] == 1.0                                           # false positives behind full recall cost nothing
    jit = [(b + torch.randn(b.shape, generator=g) * 3, s, c) for b, s, c in det]
    r = box_ap(jit, gt)
    assert r["AP50"] > r["AP"] > 0.3 and r["AP50"] > r["AP75"]
    # a class without ground truth does not enter the mean; a class with ground truth and no detection counts as 0
    extra = [(torch.cat([b, b[:1]]), torch.cat([s, s[:1]]), torch.cat([c, torch.tensor([7])])) for b, s, c in det]
    assert box_ap(extra, gt)["AP"] == 1.0
    miss = [(b[c != 0], s[c != 0], c[c != 0]) for b, s, c in det]
    assert abs(box_ap(miss, gt)["AP"] - 0.8) < 1e-12
    # max_dets counts per (image, CATEGORY) like COCOeval's evaluateImg: 30 true positives behind 120 better-scored false ones, i.e.
    # per class n true behind 4 n false -- with a cap of min-class-count the kept ones are all false, with the default 100 nothing is cut
    many_b = torch.cat([gt[0][0]] + [gt[0][0] + 2000] * 4)
    many = [(many_b, torch.cat([torch.rand(30, generator=g)] + [torch.rand(30, generator=g) + 2] * 4), torch.cat([gt[0][1]] * 5))]
    fewest = int(torch.bincount(gt[0][1], minlength=5).min())
    assert fewest >= 1 and box_ap(many, gt[:1], max_dets=4 * fewest)["AP"] == 0.0 and box_ap(many, gt[:1])["AP"] > 0.0
    assert box_ap(many, gt[:1])["dets"] == 150
