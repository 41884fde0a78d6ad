"""Child process of tests/test_gpu_configs.py: one rank (or the single process) of BASELINE configs[3]'s PER-RANK workload at
config dims -- K = 5 members, T = 100, D = 150528, F = H = 4096, fp32 -- running Diffusion.test_atk on ONE global test batch.
Every rank builds the same seeded synthetic weights, uploads and perturbs ITS rows of the batch with the random choices made for
the whole batch (runner._rank_batches / perturb), samples, and takes part in the batch's single all-gather.  Rank 0 writes the gathered
class probabilities and votes.  World size, rank and rendezvous come from the environment (as under torch.distributed.run)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--global-batch", type=int, default=64)
    ap.add_argument("--timesteps", type=int, default=100)
    ap.add_argument("--members", type=int, default=5)
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="rehearse ranks 0..N-1 of an N-rank job one after the other in THIS process (tests/_emulated_dist.py): a GPU box "
                         "admits at most 6 processes on its card, so 8 ranks x 32 rows cannot run there as 8 processes")
    a = ap.parse_args()
    from nested_diffusion_amd import dist as nd_dist, synthetic
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    rank, local, world = nd_dist.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ns = argparse.Namespace
    D, H, F, C, T, K, B = 3 * 224 * 224, 4096, 4096, 2, a.timesteps, a.members, a.global_batch
    cfg = ns(data=ns(dataset="ChestXRay", num_classes=C), model=ns(data_dim=D, hidden_dim=H, feature_dim=F, arch="linear"),
             diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                          trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
             testing=ns(batch_size=B))
    vit = VisionTransformer(synthetic.vit_state(seed=7, device=dev), 12, dev)
    mlps = [Classifier(synthetic.classifier_state(196 * 768, seed=2000 + k, device=dev), dev) for k in range(K)]
    g = torch.Generator().manual_seed(31)
    x = torch.rand(B, 3, 224, 224, generator=g)
    target = torch.randint(0, C, (B,), generator=g)

    def one_rank():
        """What one rank (or the single process) of the job does, start to finish: same --seed everywhere (the reference's set_seed)."""
        states = [synthetic.cond_model_state(D, H, F, C, T, seed=1000 + k, device=dev) for k in range(K)]
        runner = Diffusion(ns(seed=4242, mc_trials=1, noise_perturbation=0.02), cfg, device=dev, conditioner=GuidingConditioner(vit, mlps),
                           noise_estimator_states=states)
        del states
        runner.test_atk(test_loader=[(x, target)])
        res = {"prob": runner.last_probs.cpu(), "accuracy": float(runner.last_report["accuracy"]), "rows_per_rank": runner.engine.max_batch,
               "step_kernel": runner.engine.step_plan(runner.engine.max_batch)["kernel"]}
        runner.engine = None
        del runner
        torch.cuda.empty_cache()
        return res

    if a.emulate_world:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _emulated_dist import emulate_rank
        sink, res = {}, None
        for r in range(a.emulate_world):
            emulate_rank(r, a.emulate_world, sink)
            res = one_rank()                      # after the LAST rank the sink holds every shard: res is what the real gather returns
            print(f"emulated rank {r}/{a.emulate_world}: rows {nd_dist.shard_bounds(B, r, a.emulate_world)}", flush=True)
        emulate_rank()
        res.update(world=a.emulate_world, backend="emulated", shards=sorted(sink))
        torch.save(res, a.out)
        return
    res = one_rank()
    if rank == 0:
        res.update(world=world, backend=torch.distributed.get_backend() if world > 1 else None)
        torch.save(res, a.out)
    nd_dist.shutdown()


if __name__ == "__main__":
    main()
