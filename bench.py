#!/usr/bin/env python3
"""bench.py -- denoising-step*images/s of the nested-diffusion inference hot path on MI355X.

Workload (BASELINE.json metric "K=5,T=100,224^2", configs[2]): K=5 ensemble members, T=100 steps,
B=32 synthetic 3x224x224 images PER GPU, mc=1 trial, fp32, config dims D=150528, F=H=4096.
One "step" = one pass of the WHOLE hot path over one batch, inputs already resident in HBM:
  ViT-prefix + mapping MLPs -> softmax -> encoder hoist (norm(encoder_x(x)), once per member and batch)
  -> K x T reverse-diffusion steps (one hipGraph) -> convert_to_prob / mean / vote (-> RCCL all-gather for N>1).
value = N * B*K*mc*T * steps / wall time (weak scaling: per-GPU batch fixed, no data-path collective).

  python bench.py [--gpus N --steps K --warmup W]      (N>1: launched by torch.distributed.run, one rank per GPU)
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the run) and
`cpu_baseline` (the CPU oracle in as-written mode on a bounded sample; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling


def ns(**kw):
    return argparse.Namespace(**kw)


def build_runner(args, device):
    from nested_diffusion_amd import synthetic
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    D, H, F, C, T, K = 3 * 224 * 224, 4096, 4096, 2, args.timesteps, args.members
    cfg = ns(data=ns(dataset="ChestXRay", num_classes=C), model=ns(data_dim=D, hidden_dim=H, feature_dim=F, arch="linear"),
             diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                          trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
             testing=ns(batch_size=args.batch))
    vit = VisionTransformer(synthetic.vit_state(seed=7, device=device), 12, device, dtype=getattr(args, "dtype", "f32"))
    mlps = [Classifier(synthetic.classifier_state(196 * 768, seed=2000 + k, device=device), device, dtype=getattr(args, "dtype", "f32"))
            for k in range(K)]
    states = [synthetic.cond_model_state(D, H, F, C, T, seed=1000 + k, device=device) for k in range(K)]
    runner = Diffusion(ns(seed=1234, mc_trials=args.mc, fp16=getattr(args, "dtype", "f32") == "f16"), cfg, device=device,
                       conditioner=GuidingConditioner(vit, mlps),
                       noise_estimator_states=states)
    runner.load_noise_estimators(max_batch=args.batch, mc_trials=args.mc)
    cpu_member = None
    if args.cpu_baseline:
        cpu_member = {k: v.cpu() for k, v in states[0].items()}
    return runner, cfg, cpu_member


def cpu_baseline(member_cpu, x_flat_cpu, yhat_cpu, T_full, budget_s=20.0, eng=None, temperature=0.1737):
    """CPU oracle, as-written mode (encoder re-evaluated every step, eager), one member, same B and dims.
    Bounded sample: as many denoising steps as fit ~budget_s (every step is identical work).  The same
    (member 0, noise, steps) sample is also run through the HIP path: `delta_vs_cpu` is the parity of the run."""
    from oracle import ref_cpu
    B, C = yhat_cpu.shape
    alphas, omabs = ref_cpu.schedule_tables("linear", T_full, 1e-4, 0.02)
    t0 = time.perf_counter()
    ref_cpu.p_sample_loop(member_cpu, x_flat_cpu, yhat_cpu, yhat_cpu, 2, alphas, omabs, torch.randn(2, B, C), True, hoist=False)
    per_step = (time.perf_counter() - t0) / 2
    T_s = int(max(4, min(T_full, budget_s / max(per_step, 1e-3))))
    noise = torch.randn(T_s, B, C)
    t0 = time.perf_counter()
    y_cpu = ref_cpu.p_sample_loop(member_cpu, x_flat_cpu, yhat_cpu, yhat_cpu, T_s, alphas, omabs, noise, True, hoist=False)
    dt = time.perf_counter() - t0
    delta = None
    if eng is not None:
        dev = eng.device
        y_gpu = eng.sample(yhat_cpu.to(dev)[None], yhat_cpu.to(dev)[None], noise.to(dev)[None], member0=0, n_members=1, mc=1, T=T_s)[0].cpu()
        p_gpu, p_cpu = ref_cpu.convert_to_prob(y_gpu, temperature), ref_cpu.convert_to_prob(y_cpu, temperature)
        delta = {"max_abs_y0": float((y_gpu - y_cpu).abs().max()), "max_abs_class_prob": float((p_gpu - p_cpu).abs().max()),
                 "criterion": "class probabilities within 1e-3 (fp32)"}
    return {"delta_vs_cpu": delta, "value": B * T_s / dt, "unit": "denoising-step*images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/ref_cpu.py p_sample_loop as written (encoder re-evaluated each step), 1 of K members, "
                      f"B={B}, {T_s} of T={T_full} steps, fp32 torch CPU, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--members", type=int, default=5)
    ap.add_argument("--timesteps", type=int, default=100)
    ap.add_argument("--mc", type=int, default=1)
    ap.add_argument("--dtype", default="f32", choices=("f32", "f16"),
                    help="operand dtype of the weight-streaming layers; f16 is the secondary fp16-operand mode (BASELINE config 5), "
                         "never the headline (the reference computes in fp32)")
    ap.add_argument("--no-cpu-baseline", dest="cpu_baseline", action="store_false")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    from nested_diffusion_amd import dist as nd_dist
    rank, local, world = nd_dist.init_from_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    args.cpu_baseline = args.cpu_baseline and rank == 0 and world == 1 and args.dtype == "f32"

    from nested_diffusion_amd import synthetic
    runner, cfg, cpu_member = build_runner(args, device)
    eng = runner.engine
    B, K, T, mc, C = args.batch, args.members, args.timesteps, args.mc, 2
    images = synthetic.images(B, seed=1234 + rank, device=device)        # resident in HBM before the timed region
    torch.cuda.synchronize(device)

    def step():
        out = runner.predict_batch(images)
        if world > 1:
            out["prob_all"] = nd_dist.all_gather_rows(out["prob"], B * world, world)   # the single collective
        return out

    eng.set_profiling(True)                       # 8 probed steps per replay: event nodes inside the graph
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize(device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(args.steps):
        out = step()
    ev[1].record()
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    head_us, lin2_us, lin3_us, n_probe = eng.profile_read()

    # stage breakdown (outside the timed region, torch events on the launch stream)
    def timed(fn, reps=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize(device)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) / reps
    from nested_diffusion_amd import ops
    flat = torch.flatten(images, 1)
    yhat = out["yhat"]
    noise = torch.randn(K, T, B * mc, C, device=device)
    stages = {
        "conditioner_ms": timed(lambda: runner.compute_guiding_prediction(images, include_full_vit=False)),
        "encoder_hoist_ms": timed(lambda: eng.encode(flat)),
        "sampler_ms": timed(lambda: eng.sample(yhat, yhat, noise, mc=mc, T=T)),
    }

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    units = B * K * mc * T
    value = world * units * args.steps / dt
    F = 4096
    M = B * mc
    # algorithmic bytes of ONE launch of the dominant kernel (k_skinny_fused, lin2 block, all K members in the grid):
    # weights F*F + gain/shift rows 2*F + activations in M*F + out M*F, fp32  (DESIGN.md "roofline accounting")
    wb = 4.0 if args.dtype == "f32" else 2.0
    alg_bytes = K * (wb * F * F + 4.0 * 2 * F + wb * 2 * M * F)
    # The bracketed intervals include the dispatch gap of the graph nodes (~3 us: rocprofv3's begin->end durations of the same
    # kernels are that much shorter, profiles/).  record_node_us is an EMPTY interval (two record nodes back to back, ~6 us),
    # reported for reference; nothing is subtracted, so `achieved` is the conservative figure.
    ovh_us = getattr(eng, "probe_overhead_us", 0.0) if n_probe else 0.0
    avg_us = 0.5 * (lin2_us + lin3_us) if n_probe else float("nan")
    achieved = alg_bytes / (avg_us * 1e-6) / 1e9 if n_probe else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_skinny_bytes_per_launch")
        except Exception:
            traffic = None
    line = {
        "metric": "denoising-steps*images/sec (K=5,T=100,224^2)", "value": value, "unit": "denoising-step*images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.dtype == "f32" else "f16 operands / f32 accumulate (secondary mode, not the reference's arithmetic)",
        "data": "synthetic",
        "config": {"workload": f"K={K} members, T={T} steps, B={B} images/GPU (3x224x224), mc={mc}, D=150528, F=H=4096; "
                               "whole hot path per step: ViT-prefix+mapping MLPs, encoder hoist, K*T reverse steps, aggregation",
                   "global_batch": B * world, "parallelism": f"dp{world} (batch-sharded, all K members per GPU, one all-gather)"},
        "roofline": {"bound": "hbm", "kernel": "k_skinny<2,6,4,2,{0,1},true> (lin2 / lin3+lin4 ConditionalLinear blocks, K members per launch)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                     "traffic": traffic, "alg_bytes_per_launch": alg_bytes, "avg_launch_us": avg_us,
                     "probe": {"head_us": head_us, "lin2_us": lin2_us, "lin3_us": lin3_us, "record_node_us": ovh_us, "steps_probed": n_probe}},
        "stages_ms": stages,
        "sampler_only_value": units / (stages["sampler_ms"] * 1e-3),
    }
    if args.cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cpu_member, flat.cpu(), yhat[0].cpu(), T, args.cpu_seconds, eng=eng,
                                            temperature=runner.temperature)
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
