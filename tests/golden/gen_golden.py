#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  The reference modules are imported unmodified from /root/reference:

* diffusion/diffusion_utils.py      -- imports as is (math + torch only)
* diffusion/latent_model.py         -- its line 4 imports two torchvision resnet constructors
  that the hot path never touches (only ResNetEncoder uses them, arch != 'linear');
  torchvision is absent from this image, so an empty placeholder module object is put in
  sys.modules for that one import line.  No reference arithmetic is replaced.
* mapping/models/mlp.py             -- imports as is
* diffusion/classification_train_separately.py -- module-level helpers + three methods
  (convert_to_prob, compute_ensemble_confidence, majority_voting_for_mc_samples); needs the
  same kind of placeholder for absent, unused third-party imports.

Only arrays (inputs + expected outputs) and seeds are written; no reference source text.
Usage:  python tests/golden/gen_golden.py [--skip-full]            (re)write the fixtures
        python tests/golden/gen_golden.py --check [--skip-full]    regenerate into a temporary directory and compare with the COMMITTED
                                                                   files: same key sets, every array bit-identical (dtype, shape,
                                                                   bytes); exit status 1 and a list of differences otherwise.
                                                                   tests/test_golden_fixtures.py runs this whenever /root/reference
                                                                   is present, so a drifted or hand-edited fixture cannot go unnoticed.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import ref_cpu  # only for the seeded synthetic-parameter initialisers  # noqa: E402


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    sys.path.insert(0, os.path.join(REF, "diffusion"))
    sys.path.insert(0, os.path.join(REF, "mapping", "models"))
    import diffusion_utils as du                      # noqa
    tv = _placeholder("torchvision")
    tvm = _placeholder("torchvision.models")
    tvr = _placeholder("torchvision.models.resnet", resnet18=None, resnet50=None)
    tv.models = tvm; tvm.resnet = tvr
    import latent_model as lm                         # noqa
    import mlp as ref_mlp                             # noqa
    return du, lm, ref_mlp


def import_reference_runner():
    """Best effort: the runner module drags in many absent third-party packages."""
    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return _Any()
        def __getattr__(self, k): return _Any()
    for name in ["statsmodels", "statsmodels.api", "torchmetrics", "torchmetrics.classification",
                 "autoattack", "foolbox", "foolbox.attacks", "foolbox.criteria",
                 "torchvision.transforms", "torchvision.datasets", "torchvision.utils",
                 "torchvision.transforms.functional", "medmnist", "tensorboardX"]:
        if name not in sys.modules:
            m = _placeholder(name)
            m.__getattr__ = lambda k, _A=_Any: _A()   # type: ignore
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "diffusion"))
    try:
        import classification_train_separately as runner   # noqa
    finally:
        os.chdir(cwd)
    return runner


def ns(**kw):
    return argparse.Namespace(**kw)


def make_config(D, H, Fd, C, T, dataset="ChestXRay"):
    return ns(diffusion=ns(timesteps=T), model=ns(data_dim=D, arch="linear", feature_dim=Fd, hidden_dim=H),
              data=ns(num_classes=C, dataset=dataset))


def build_ref_model(lm, D, H, Fd, C, T, seed, denoiser=False, guidance=True):
    cfg = make_config(D, H, Fd, C, T)
    model = lm.ConditionalModel(cfg, guidance=guidance)
    params = ref_cpu.init_cond_model_params(D, H, Fd, C, T, guidance, seed=seed, denoiser=denoiser)
    missing = model.load_state_dict(params, strict=True)
    model.eval()
    return model, params


def gen_schedule(du):
    out = {}
    for T in (10, 100, 1000):
        betas = du.make_beta_schedule(schedule="linear", num_timesteps=T, start=1e-4, end=0.02)
        betas = betas.float()
        alphas = 1.0 - betas                                   # runner :219-220
        omabs = torch.sqrt(1 - alphas.cumprod(dim=0))          # runner :222-224
        out[f"betas_{T}"] = betas.numpy(); out[f"alphas_{T}"] = alphas.numpy(); out[f"omabs_{T}"] = omabs.numpy()
    for sched in ("cosine", "cosine_anneal", "quad", "sigmoid", "const", "jsd"):
        out[f"betas_{sched}_50"] = du.make_beta_schedule(schedule=sched, num_timesteps=50, start=1e-4, end=0.02).float().numpy()
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **out)
    print("schedule.npz", len(out))


def run_ref_loop(du, model, x, yhat, T, alphas, omabs, seed):
    B, C = yhat.shape
    torch.manual_seed(seed)
    noise = torch.stack([torch.randn(B, C) for _ in range(T)])   # same draw order as :139,:67
    torch.manual_seed(seed)
    with torch.no_grad():
        seq = du.p_sample_loop(model, x, yhat, yhat, T, alphas, omabs, only_last_sample=False)
    torch.manual_seed(seed)
    with torch.no_grad():
        y0 = du.p_sample_loop(model, x, yhat, yhat, T, alphas, omabs, only_last_sample=True)
    assert torch.equal(y0, seq[-1])
    return noise, torch.stack(seq)


def gen_sampler_small(du, lm, only=None):
    """Small/medium-dim ConditionalModel: whole state_dict + every intermediate y_t."""
    cases = [  # name, D, H, F, C, T, B, seed
        ("s0", 48, 64, 64, 2, 10, 3, 11),
        ("s1", 192, 128, 128, 2, 100, 32, 12),
        ("s2", 96, 64, 80, 3, 25, 1, 13),        # C=3, B=1, F not a power of two
        ("s3", 160, 96, 256, 2, 1000, 5, 14),    # T=1000: 1/sqrt(abar) amplification (random weights: expansive chain)
        ("s4", 160, 96, 128, 2, 1000, 5, 15),    # T=1000 with the denoiser-structured init: contractive chain, |y_t| = O(1)
        ("s5", 64, 48, 80, 3, 12, 4, 16),        # guidance=False (lin1 sees y alone, latent_model.py:157-158,172) + a per-row t vector
    ]
    for name, D, H, Fd, C, T, B, seed in cases:
        if only and name not in only:
            continue
        model, params = build_ref_model(lm, D, H, Fd, C, T, seed, denoiser=(name == "s4"), guidance=(name != "s5"))
        g = torch.Generator().manual_seed(seed + 100)
        x = torch.rand(B, D, generator=g)
        yhat = torch.softmax(torch.randn(B, C, generator=g), dim=1)
        betas = du.make_beta_schedule("linear", T, 1e-4, 0.02).float()
        alphas = 1.0 - betas
        omabs = torch.sqrt(1 - alphas.cumprod(dim=0))
        noise, seq = run_ref_loop(du, model, x, yhat, T, alphas, omabs, seed + 200)
        # single eps_theta calls at chosen t (incl. 0 and T-1), with a softplus-threshold edge case
        ts = sorted({0, 1, T // 2, T - 1})
        eps = {}
        with torch.no_grad():
            for t in ts:
                yy = seq[min(T - 1 - t, T - 1)]
                eps[t] = model(x, yy, torch.tensor([t]), yhat)
            big = model(x * 40.0, seq[0] * 30.0, torch.tensor([T - 1]), yhat)   # drives softplus inputs > 20
            t_rows = torch.tensor([(T - 1 - 3 * b) % T for b in range(B)])       # ConditionalModel.forward with t of shape [B]
            eps_rows = model(x, seq[1], t_rows, yhat)
        save = {("p." + k): v.numpy() for k, v in params.items()}
        save.update(x=x.numpy(), yhat=yhat.numpy(), noise=noise.numpy(), seq=seq.numpy(),
                    alphas=alphas.numpy(), omabs=omabs.numpy(), eps_ts=np.array(ts),
                    eps=np.stack([eps[t].numpy() for t in ts]), eps_big=big.numpy(), t_rows=t_rows.numpy(), eps_rows=eps_rows.numpy(),
                    dims=np.array([D, H, Fd, C, T, B, seed]))
        np.savez_compressed(os.path.join(OUT, f"sampler_{name}.npz"), **save)
        print(f"sampler_{name}.npz  y0[0]={seq[-1][0].tolist()}")


def gen_sampler_full(du, lm):
    """Config dims (D=150528, F=H=4096, C=2): weights regenerated from a seed, outputs only."""
    D, H, Fd, C, T, B, seed = 150528, 4096, 4096, 2, 10, 4, 1000
    model, params = build_ref_model(lm, D, H, Fd, C, T, seed)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(B, C, generator=g), dim=1)
    betas = du.make_beta_schedule("linear", T, 1e-4, 0.02).float()
    alphas = 1.0 - betas
    omabs = torch.sqrt(1 - alphas.cumprod(dim=0))
    noise, seq = run_ref_loop(du, model, x, yhat, T, alphas, omabs, 4321)
    with torch.no_grad():
        xe = model.norm(model.encoder_x(x))
    np.savez_compressed(os.path.join(OUT, "sampler_full.npz"), yhat=yhat.numpy(), noise=noise.numpy(),
                        seq=seq.numpy(), xe=xe.numpy(), dims=np.array([D, H, Fd, C, T, B, seed]),
                        x_seed=np.array(1234))
    print("sampler_full.npz", seq[-1])


def gen_classifier(ref_mlp):
    """mlp.Classifier at its hard-coded 196*768 input; weights from seed, outputs only."""
    B, seed = 3, 2000
    params = ref_cpu.init_classifier_params(196 * 768, seed=seed)
    m = ref_mlp.Classifier(num_classes=2, in_features=196 * 768)
    m.load_state_dict(params, strict=True)
    m.eval()
    g = torch.Generator().manual_seed(77)
    tok = torch.randn(B, 196, 768, generator=g)
    with torch.no_grad():
        out = m(tok, dataset="ChestXRay")
    np.savez_compressed(os.path.join(OUT, "classifier_full.npz"), out=out.numpy(),
                        dims=np.array([B, seed, 77]))
    print("classifier_full.npz", out)


def gen_aggregation(runner):
    g = torch.Generator().manual_seed(5)
    save = {}
    for name, n_s, B, C, temp in (("a0", 100, 7, 2, 0.1737), ("a1", 5, 4, 2, 0.3162), ("a2", 6, 9, 3, 0.1737)):
        samples = [torch.randn(B, C, generator=g) * 0.8 + 0.5 for _ in range(n_s)]
        if name == "a1":     # forced ties + the raw-argmax vs closest-to-1 disagreement (SURVEY a-16)
            samples = samples[:4]
            samples[0][0] = torch.tensor([2.5, 1.2]); samples[1][0] = torch.tensor([0.1, 0.9])
            samples[2][0] = torch.tensor([2.5, 1.2]); samples[3][0] = torch.tensor([0.1, 0.9])
        d = runner.Diffusion.__new__(runner.Diffusion)
        d.temperature = temp
        vote = runner.majority_voting_for_mc_samples([s.clone() for s in samples])
        p1 = d.convert_to_prob(samples[0].clone())
        lst = [s.clone() for s in samples]
        prob = d.compute_ensemble_confidence(lst)
        save[name + "_samples"] = torch.stack(samples).numpy(); save[name + "_vote"] = vote.numpy()
        save[name + "_prob"] = prob.numpy(); save[name + "_p1"] = p1.numpy()
        save[name + "_mutated"] = torch.stack(lst).numpy(); save[name + "_temp"] = np.array(temp)
    np.savez_compressed(os.path.join(OUT, "aggregation.npz"), **save)
    print("aggregation.npz")


def gen_report(runner):
    """compute_mean_piws_for_class / calculate_variances / accuracy of the reference on synthetic per-sample
    probabilities (what pred_mc holds after compute_ensemble_confidence, quirk Q4)."""
    g = torch.Generator().manual_seed(11)
    save = {}
    for name, S, N, C in (("r0", 100, 64, 2), ("r1", 25, 37, 3), ("r2", 8, 5, 2)):
        probs = [torch.softmax(torch.randn(N, C, generator=g) * 1.5, dim=1) for _ in range(S)]
        mv = torch.randint(0, C, (N,), generator=g)
        label = torch.randint(0, C, (N,), generator=g)
        if name == "r2":
            mv = torch.zeros(N, dtype=torch.long); label = torch.zeros(N, dtype=torch.long)   # empty selections -> NaN / 0
        pc, pi = runner.compute_mean_piws_for_class([p.clone() for p in probs], mv, label)
        vc, vi = runner.calculate_variances([p.clone() for p in probs], mv, label)
        save[name + "_probs"] = torch.stack(probs).numpy(); save[name + "_mv"] = mv.numpy(); save[name + "_label"] = label.numpy()
        save[name + "_piw_c"] = pc.numpy(); save[name + "_piw_i"] = pi.numpy()
        save[name + "_var_c"] = vc.numpy(); save[name + "_var_i"] = vi.numpy()
    np.savez_compressed(os.path.join(OUT, "report.npz"), **save)
    print("report.npz")


def gen_perturb(runner):
    """add_noise / adjust_brightness / adjust_contrast / down_up_sample / random_cover_new of the reference
    (diffusion/utils.py, star-imported by the runner module) on small images.  random_crop_and_resize needs
    torchvision's Resize and is not runnable here."""
    import random as pyrandom
    g = torch.Generator().manual_seed(21)
    x = torch.rand(3, 3, 24, 20, generator=g)
    save = {"x": x.numpy()}
    torch.manual_seed(5)
    z = torch.randn_like(x)
    torch.manual_seed(5)
    save["z"] = z.numpy(); save["noise_0p3"] = runner.add_noise(x, 0.3).numpy()
    save["bright_p0p4"] = runner.adjust_brightness(x, 0.4).numpy(); save["bright_m0p3"] = runner.adjust_brightness(x, -0.3).numpy()
    save["contrast_1p7"] = runner.adjust_contrast(x, 1.7).numpy(); save["contrast_0p4"] = runner.adjust_contrast(x, 0.4).numpy()
    save["downup_2"] = runner.down_up_sample(x, 2).numpy(); save["downup_3"] = runner.down_up_sample(x, 3).numpy()
    pyrandom.seed(9)
    save["cover_0p05_2"] = runner.random_cover_new(x, (0.05, 2)).numpy()
    np.savez_compressed(os.path.join(OUT, "perturb.npz"), **save)
    print("perturb.npz")


def generate(skip_full, only=None):
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    du, lm, ref_mlp = import_reference()
    if only:
        gen_sampler_small(du, lm, only=set(only))
        return
    gen_schedule(du)
    gen_sampler_small(du, lm)
    try:
        runner = import_reference_runner()
        gen_aggregation(runner)
        gen_report(runner)
        gen_perturb(runner)
    except Exception as e:  # ordinary Python error -> recorded, aggregation stays self-pinned
        print("runner import failed:", type(e).__name__, e)
    if not skip_full:
        gen_classifier(ref_mlp)
        gen_sampler_full(du, lm)


def compare_trees(fresh_dir, committed_dir):
    """Differences between freshly generated fixtures and the committed ones, as a list of strings (empty = identical)."""
    diffs = []
    for f in sorted(os.listdir(fresh_dir)):
        if not f.endswith(".npz"):
            continue
        path = os.path.join(committed_dir, f)
        if not os.path.exists(path):
            diffs.append(f"{f}: generated but not committed")
            continue
        new, old = np.load(os.path.join(fresh_dir, f)), np.load(path)
        if sorted(new.files) != sorted(old.files):
            diffs.append(f"{f}: key sets differ: only generated {sorted(set(new.files) - set(old.files))}, "
                         f"only committed {sorted(set(old.files) - set(new.files))}")
        for k in sorted(set(new.files) & set(old.files)):
            a, b = new[k], old[k]
            if a.dtype != b.dtype or a.shape != b.shape or a.tobytes() != b.tobytes():
                diffs.append(f"{f}[{k}]: committed array is not what the generator produces "
                             f"({b.dtype}{b.shape} vs {a.dtype}{a.shape})")
    return diffs


def main():
    global OUT
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true", help="leave out the two config-dim fixtures (2.6 GB of weights, minutes)")
    ap.add_argument("--only", nargs="*", default=None, help="regenerate only these sampler cases (e.g. s4) and nothing else")
    ap.add_argument("--check", action="store_true", help="regenerate to a temp dir and compare with the committed fixtures")
    a = ap.parse_args()
    if not a.check:
        generate(a.skip_full, a.only)
        return 0
    import tempfile
    committed = OUT
    with tempfile.TemporaryDirectory(prefix="nd_golden_") as tmp:
        OUT = tmp
        generate(a.skip_full, a.only)
        diffs = compare_trees(tmp, committed)
        n = len([f for f in os.listdir(tmp) if f.endswith(".npz")])
    OUT = committed
    for d in diffs:
        print("DIFF", d)
    print(f"checked {n} fixture files against {committed}: {'identical' if not diffs else str(len(diffs)) + ' difference(s)'}")
    return 1 if diffs else 0


if __name__ == "__main__":
    sys.exit(main())
