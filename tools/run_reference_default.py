"""The reference's own default invocation (testing_scripts/test.sh:24 with configs/chest_x_ray.yml's shape) at REAL sizes, end to end
through the drop-in CLI: checkpoints in the reference's three layouts written to disk (synthetic weights: ViT-B/16 state_dict,
five 150528->4096->2048->128->2 mapping MLPs, five noise estimators D=150528 F=H=4096 with T+1-row embeddings), YAML in the
reference's key layout (timesteps 1000, batch_size 70), mc_trials = 20 (hard-coded in the reference), synthetic test batches.
GPU box only; needs ~30 GB under --dir.     python tools/run_reference_default.py [--dir /tmp/nd_ref_default] [--timesteps 1000] [--batches 1]
Prints the report lines of test_atk and the step*images/s it logs."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from nested_diffusion_amd import synthetic
from nested_diffusion_amd import main as nd_main

ap = argparse.ArgumentParser()
ap.add_argument("--dir", default="/tmp/nd_ref_default")
ap.add_argument("--timesteps", type=int, default=1000)
ap.add_argument("--batches", type=int, default=1)
ap.add_argument("--batch_size", type=int, default=70)
a = ap.parse_args()
D, H, F, C, K, T = 150528, 4096, 4096, 2, 5, a.timesteps
ck = os.path.join(a.dir, "ckpt")
os.makedirs(os.path.join(ck, "MLPs"), exist_ok=True)
t0 = time.time()
cpu = lambda sd: {k: v.cpu() for k, v in sd.items()}
torch.save(cpu(synthetic.vit_state(seed=7)), os.path.join(ck, "vit_base_patch16_224_ChestXRay.pth"))
for i in range(K):
    torch.save(cpu(synthetic.classifier_state(196 * 768, seed=2000 + i)), os.path.join(ck, "MLPs", f"block_{i}.pth"))
paths = []
for i in range(K):
    p = os.path.join(a.dir, f"diffu{i}_ckpt_best.pth")
    torch.save({"noise_estimator": cpu(synthetic.cond_model_state(D, H, F, C, T, seed=1000 + i, denoiser=T >= 500)), "optimizer": {}, "epoch": 1}, p)
    paths.append(p)
print(f"checkpoints written in {time.time() - t0:.0f} s", flush=True)
cfg = {"data": {"dataset": "ChestXRay", "seed": 4444, "num_classes": C, "num_workers": 0, "dataroot": "PATH"},
       "model": {"type": "simple", "data_dim": D, "feature_dim": F, "hidden_dim": H, "arch": "linear", "var_type": "fixedlarge"},
       "diffusion": {"beta_schedule": "linear", "beta_start": 0.0001, "beta_end": 0.02, "timesteps": T, "vis_step": 100, "num_figs": 10,
                     "include_guidance": True, "apply_aux_cls": True, "trained_aux_cls_ckpt_path": ck,
                     "trained_diffusion_ckpt_path": [paths], "aux_cls": {"arch": "sevit"}},
       "training": {"image_folder": "training_image_samples"}, "testing": {"batch_size": a.batch_size}}
ypath = os.path.join(a.dir, "chest_x_ray.yml")
yaml.safe_dump(cfg, open(ypath, "w"))
argv = ["--test", "--device", "0", "--thread", "8", "--loss", "card_onehot_conditional", "--config", ypath, "--exp", os.path.join(a.dir, "results"),
        "--doc", "chest_x_ray", "--n_splits", "1", "--noise_perturbation", "0", "--low_resolution", "0", "--brightness", "0", "--contrast", "1",
        "--crop", "0", "--attack_name", "None", "--eps", "0", "--ni", "--preprocess", "grayscaled", "--seed", "7",
        "--synthetic_batches", str(a.batches)]
t0 = time.time()
rc = nd_main.main(argv)
print(f"main() returned {rc} after {time.time() - t0:.1f} s", flush=True)
log = open(os.path.join(a.dir, "results", "logs", "chest_x_ray", "split_0", "stdout.txt")).read()
assert "Traceback" not in log, log[-3000:]
print(log[-1500:])
