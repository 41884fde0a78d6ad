"""Cross-checks of the two oracle pieces whose upstream packages are absent offline (torchmetrics 0.11.4, torchvision 0.11):
the oracle against implementations this repo did not write.  They are not the pinned packages, so DESIGN.md keeps calling
both "parity unpinned"; what these tests establish is agreement with an independent implementation of the same published
algorithm (same standing as tests/test_vit_crosscheck.py for the ViT).

* ECE (classification_train_separately.py:21, 413-423): scikit-learn's calibration_curve does the uniform binning and the
  per-bin means; the weighted |acc - conf| sum is three numpy lines here.
* crop-and-resize (diffusion/utils.py:282-312): Pillow's float ('F' mode) bilinear resize, which for an up-scale is the same
  half-pixel-centre, edge-clamped two-tap filter as torchvision's tensor Resize.
"""
import numpy as np
import torch

from oracle import ref_cpu


def _ece_sklearn(probs: np.ndarray, target: np.ndarray, n_bins: int) -> float:
    from sklearn.calibration import calibration_curve
    conf = probs.max(axis=1).astype(np.float64)
    acc = (probs.argmax(axis=1) == target).astype(np.float64)
    acc_bin, conf_bin = calibration_curve(acc, conf, n_bins=n_bins, strategy="uniform")
    # calibration_curve drops the empty bins; recover the sizes of the ones it kept with the same edges
    counts = np.histogram(conf, bins=np.linspace(0.0, 1.0, n_bins + 1))[0]
    counts = counts[counts > 0]
    assert counts.shape == acc_bin.shape
    return float(np.sum(np.abs(acc_bin - conf_bin) * counts / counts.sum()))


def test_ece_oracle_agrees_with_sklearn_binning():
    g = torch.Generator().manual_seed(11)
    for n, c, sharp in ((48, 2, 1.0), (500, 2, 3.0), (1000, 7, 2.0), (37, 3, 0.3)):
        probs = torch.softmax(torch.randn(n, c, generator=g) * sharp, dim=1)
        target = torch.randint(0, c, (n,), generator=g)
        want = _ece_sklearn(probs.numpy(), target.numpy(), 10)
        got = float(ref_cpu.multiclass_calibration_error_l1(probs, target, 10))
        assert abs(got - want) < 2e-6, (n, c, got, want)      # fp32 oracle against a float64 computation
    # the call shape of the report: already-averaged probabilities through convert_to_prob once more (quirk kept)
    probs = torch.softmax(torch.randn(200, 2, generator=g), dim=1)
    target = torch.randint(0, 2, (200,), generator=g)
    twice = ref_cpu.convert_to_prob(probs, 0.1737)
    assert abs(float(ref_cpu.compute_ece_as_reference(probs, target, 0.1737)) - _ece_sklearn(twice.numpy(), target.numpy(), 10)) < 2e-6


def test_ece_bin_edges_are_right_closed():
    """torchmetrics buckets with torch.bucketize(conf, edges) - 1: a confidence exactly on an edge belongs to the bin BELOW it
    ((lo, hi]); scikit-learn's searchsorted puts it in the bin above, so this case is checked by hand instead.
    Four rows, confidences 0.6 (edge), 0.6, 0.75, 1.0; correct: yes, no, yes, yes."""
    edge = float(torch.linspace(0, 1, 11)[6])                 # the fp32 edge itself, so `conf == edge` really holds
    probs = torch.tensor([[edge, 1 - edge], [edge, 1 - edge], [0.75, 0.25], [1.0, 0.0]])
    target = torch.tensor([0, 1, 0, 0])
    # bins: (0.5, 0.6] holds rows 0, 1: |0.5 - 0.6| * 2/4;  (0.7, 0.8] holds row 2: |1 - 0.75| / 4;  (0.9, 1.0] row 3: 0
    want = abs(0.5 - edge) * 0.5 + 0.25 * 0.25
    assert abs(float(ref_cpu.multiclass_calibration_error_l1(probs, target, 10)) - want) < 1e-7


def test_crop_and_resize_oracle_agrees_with_pillow_float_bilinear():
    from PIL import Image
    g = torch.Generator().manual_seed(5)
    for side, frac in ((20, 0.25), (224, 0.1), (64, 0.5)):
        crop = int(side * (1 - frac))
        x = torch.rand(3, 3, side, side, generator=g)
        corners = [(int(t), int(l)) for t, l in torch.randint(0, side - crop + 1, (3, 2), generator=g)]
        got = ref_cpu.crop_and_resize(x, corners, crop).numpy()
        for b, (top, left) in enumerate(corners):
            for ch in range(3):
                window = np.ascontiguousarray(x[b, ch, top:top + crop, left:left + crop].numpy())
                want = np.asarray(Image.fromarray(window, mode="F").resize((side, side), Image.BILINEAR))
                # the two differ only in how the tap position is rounded (fp32 source coordinate in torch, double in Pillow):
                # up to ~side * 2^-24 of a pixel, times a pixel-to-pixel difference of at most 1
                assert np.abs(got[b, ch] - want).max() < 3e-5, (side, b, ch)
