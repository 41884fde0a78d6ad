#!/usr/bin/env python3
"""bench.py -- denoising-step*images/s of the nested-diffusion inference hot path on MI355X.

Workload (BASELINE.json metric "K=5,T=100,224^2", configs[2]): K=5 ensemble members, T=100 steps,
B=32 synthetic 3x224x224 images PER GPU, mc=1 trial, fp32, config dims D=150528, F=H=4096.
One "step" = one pass of the WHOLE hot path over one batch, inputs already resident in HBM:
  ViT-prefix + mapping MLPs -> softmax -> encoder hoist (norm(encoder_x(x)), once per member and batch)
  -> K x T reverse-diffusion steps (one hipGraph) -> convert_to_prob / mean / vote (-> RCCL all-gather for N>1).
value = N * B*K*mc*T * steps / wall time (weak scaling: per-GPU batch fixed, no data-path collective).

  python bench.py [--gpus N --steps K --warmup W]
N > 1: either launched by torch.distributed.run (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env), or
invoked plainly -- the parent then starts the N ranks itself as child processes BEFORE it has touched the GPU, relays rank 0's
line and exits non-zero if any rank did (launch_ranks below).
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
F32_MFMA_PEAK_TF = 157.3   # dense f32-input MFMA (v_mfma_f32_16x16x4_f32), same guide
F16_MFMA_PEAK_TF = 2500.0  # dense f16 MFMA
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA (same guide; AMD's 5 PF headline includes 2:1 sparsity)


def ns(**kw):
    return argparse.Namespace(**kw)


SEED_VIT, SEED_MLP, SEED_MEMBER = 7, 2000, 1000


def build_runner(args, device):
    from nested_diffusion_amd import synthetic
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    D, H, F, C, T, K = 3 * 224 * 224, 4096, 4096, 2, args.timesteps, args.members
    cfg = ns(data=ns(dataset="ChestXRay", num_classes=C), model=ns(data_dim=D, hidden_dim=H, feature_dim=F, arch="linear"),
             diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                          trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
             testing=ns(batch_size=args.batch))
    vit = VisionTransformer(synthetic.vit_state(seed=SEED_VIT, device=device), 12, device, dtype=args.dtype)
    mlps = [Classifier(synthetic.classifier_state(196 * 768, seed=SEED_MLP + k, device=device), device, dtype=args.dtype) for k in range(K)]
    # long schedules (T >= 500, BASELINE configs[4]): members with the denoiser-structured init, whose chains stay O(1) like a
    # trained estimator's (synthetic.make_denoiser); a random eps_theta amplifies y_T by 1/sqrt(abar_T) ~ 160 at T = 1000
    states = [synthetic.cond_model_state(D, H, F, C, T, seed=SEED_MEMBER + k, device=device, denoiser=T >= 500) for k in range(K)]
    runner = Diffusion(ns(seed=1234, mc_trials=args.mc, fp16=args.dtype == "f16"), cfg, device=device,
                       conditioner=GuidingConditioner(vit, mlps), noise_estimator_states=states)
    runner.load_noise_estimators(max_batch=args.batch, mc_trials=args.mc)
    return runner, cfg


def host_copies(args, device):
    """The same seeded synthetic weights again (the device generator is deterministic), one tensor set at a time -> host."""
    from nested_diffusion_amd import synthetic
    D, H, F, C, T, K = 3 * 224 * 224, 4096, 4096, 2, args.timesteps, args.members
    cpu = lambda sd: {k: v.cpu() for k, v in sd.items()}
    vit = cpu(synthetic.vit_state(seed=SEED_VIT, device=device))
    mlps = [cpu(synthetic.classifier_state(196 * 768, seed=SEED_MLP + k, device=device)) for k in range(K)]
    members = [cpu(synthetic.cond_model_state(D, H, F, C, T, seed=SEED_MEMBER + k, device=device, denoiser=T >= 500)) for k in range(K)]
    return vit, mlps, members


def _cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, len(phys) or None, usable


def cpu_baseline(members_cpu, vit_cpu, mlps_cpu, images_cpu, out_gpu, noise, T_full, temperature, budget_s=15.0):
    """The CPU oracle on this box's host cores, rank 0, N = 1.
    value          as-written mode (the reference's cost model: encoder re-evaluated every step, eager, members one after the
                   other), one member, same B and dims, on a bounded number of steps (every step is identical work), at the
                   fastest of a sweep of torch thread counts (the default -- every logical CPU the box shows -- oversubscribes
                   the container's CPU share);
    hoisted_value  the same arithmetic with the t-invariant encoder evaluated once per member (bit-identical results on the
                   CPU): all K members x all T steps -- this full run is also what `delta_vs_cpu` compares the timed HIP
                   run's outputs with (same weights, images and noise; conditioner included)."""
    from oracle import ref_cpu
    K = len(members_cpu)
    B = images_cpu.shape[0]
    C = out_gpu["prob"].shape[1]
    x_flat = images_cpu.flatten(1)
    model, phys, usable = _cpu_info()
    alphas, omabs = ref_cpu.schedule_tables("linear", T_full, 1e-4, 0.02)
    yhat0 = out_gpu["yhat"][0].cpu()
    default_threads = torch.get_num_threads()
    sweep = {}
    cands = sorted({n for n in (8, 16, 32, 64, usable, default_threads) if 1 <= n <= max(usable, default_threads)})
    z3 = torch.randn(3, B, C)
    for n in cands:
        torch.set_num_threads(n)
        ref_cpu.p_sample_loop(members_cpu[0], x_flat, yhat0, yhat0, 1, alphas, omabs, z3[:1], True, hoist=False)       # warm the pool
        t0 = time.perf_counter()
        ref_cpu.p_sample_loop(members_cpu[0], x_flat, yhat0, yhat0, 3, alphas, omabs, z3, True, hoist=False)
        sweep[n] = 3 * B / (time.perf_counter() - t0)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    T_s = int(max(4, min(T_full, budget_s * sweep[best] / B)))
    zs = torch.randn(T_s, B, C)
    t0 = time.perf_counter()
    ref_cpu.p_sample_loop(members_cpu[0], x_flat, yhat0, yhat0, T_s, alphas, omabs, zs, True, hoist=False)
    dt = time.perf_counter() - t0
    # full hot path on the CPU, hoisted: conditioner (prefix shared across members: same values) + K members x T steps
    t0 = time.perf_counter()
    logits = ref_cpu.compute_guiding_prediction(vit_cpu, mlps_cpu, images_cpu, 12, 12, full_vit=False, share_prefix=True)
    yhat = [torch.softmax(l, dim=1) for l in logits]
    t_cond = time.perf_counter() - t0
    t0 = time.perf_counter()
    nz = noise.cpu().reshape(K, T_full, 1, B, C).permute(0, 2, 1, 3, 4).contiguous()                # oracle layout [K, mc, T, B, C]
    raw, vote, prob = ref_cpu.ensemble_predict(members_cpu, x_flat, yhat, T_full, alphas, omabs, nz, temperature, hoist=True)
    t_samp = time.perf_counter() - t0
    ref = torch.stack(raw)
    delta = {"max_abs_class_prob": float((out_gpu["prob"].cpu() - prob).abs().max()),
             "max_abs_y0": float((out_gpu["samples"].cpu() - ref).abs().max()),
             "max_abs_yhat": float((out_gpu["yhat"].cpu() - torch.stack(yhat)).abs().max()),
             "votes_equal": bool(torch.equal(out_gpu["vote"].cpu(), vote)),
             "scope": f"whole hot path, all K={K} members x all T={T_full} steps, B={B}: ViT prefix + mapping MLPs + encoders + "
                      "sampler + aggregation, same weights / images / noise as one HIP step of this run",
             "criterion": "class probabilities within 1e-3 (fp32)"}
    torch.set_num_threads(default_threads)
    return {"delta_vs_cpu": delta, "value": B * T_s / dt, "unit": "denoising-step*images/s", "cores": best, "kind": "port",
            "cpu_model": model, "physical_cores": phys, "usable_logical_cpus": usable, "torch_default_threads": default_threads,
            "thread_sweep": {str(k): round(v, 1) for k, v in sweep.items()},
            "hoisted_value": K * T_full * B / t_samp, "hoisted_sampler_s": t_samp, "conditioner_s": t_cond,
            "sample": f"oracle/ref_cpu.py p_sample_loop as written (encoder re-evaluated each step), 1 of K members, "
                      f"B={B}, {T_s} of T={T_full} steps, fp32 torch CPU on {best} threads, {dt:.1f} s; hoisted_value: all {K} members x "
                      f"{T_full} steps with the encoder evaluated once per member, {t_samp:.1f} s"}


def shape_traffic(dtype: str, M: int, K: int):
    """(HBM-side bytes per step-block launch, where the number comes from) for this run's shape, REPLAYED from the committed PMC
    profile profiles/traffic.json (tools/profile_round.sh pmc -> tools/pmc_traffic.py); (None, why) when the shape has no entry."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    key = f"{dtype}_M{M}_K{K}"
    if not os.path.exists(tpath):
        return None, "null: profiles/traffic.json is missing"
    try:
        prof = json.load(open(tpath))
        ent = prof.get("entries", {}).get(key)
    except Exception as e:
        return None, f"null: profiles/traffic.json unreadable: {e}"
    if ent is None:
        return None, (f"null: no PMC profile is committed for this shape ({key}: dtype, rows per member, members); profiles/traffic.json "
                      f"holds {sorted(prof.get('entries', {}))}")
    return ent["step_block_bytes_per_launch"], (f"profiles/traffic.json[{key}] <- {prof.get('source')}: rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE, "
                                                 "separate passes) of the step-block kernels at this shape, REPLAYED from the committed profile, "
                                                 "not measured by this run; FETCH_SIZE counts Infinity-Cache hits too")


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, one GPU each), let rank 0 print the JSON line on the inherited stdout, and return non-zero
    if any rank failed (the others are then terminated instead of waiting in a collective).  The parent has made NO GPU call
    when it gets here (`torch.cuda.device_count()` does not initialise the runtime on this image) and makes none afterwards:
    a process that has initialised the GPU must not exec or fork workers."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    shared = os.environ.get("ND_DIST_BACKEND", "") == "gloo"      # rehearsal: ranks share devices, collective over gloo
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if n_dev < n and not shared:
        raise SystemExit(f"--gpus {n} but only {n_dev} GPU(s) visible (ND_DIST_BACKEND=gloo lets ranks share a device for rehearsal)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))      # HSA_ENABLE_IPC_MODE_LEGACY: nested_diffusion_amd/dist.py
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            try:
                code = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:                      # exactly the children started above
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--members", type=int, default=5)
    ap.add_argument("--timesteps", type=int, default=100)
    ap.add_argument("--mc", type=int, default=1, help="Monte-Carlo trials per member (the reference hard-codes 20: secondary line)")
    ap.add_argument("--dtype", default="f32", choices=("f32", "f16"),
                    help="operand dtype of the weight-streaming layers; f16 is the secondary fp16-operand mode (BASELINE config 5), "
                         "never the headline (the reference computes in fp32)")
    ap.add_argument("--no-cpu-baseline", dest="cpu_baseline", action="store_false")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    from nested_diffusion_amd import dist as nd_dist
    # ND_FORCE_DIST=1: run the N > 1 branches below (process-group init on the production backend, barriers, the device all-reduce of
    # the step time, the batch's all-gather) in a ONE-rank group -- what a 1-GPU box can rehearse of the RCCL call sequence
    force_dist = os.environ.get("ND_FORCE_DIST", "0") == "1"
    rank, local, world = nd_dist.init_from_env(force=force_dist)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist_on = world > 1 or force_dist
    n_ranks_seen = torch.distributed.get_world_size() if dist_on else 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # the CPU leg runs the whole K x T job once (for delta_vs_cpu): only at the headline's mc = 1
    args.cpu_baseline = args.cpu_baseline and rank == 0 and world == 1 and args.dtype == "f32" and args.mc == 1

    from nested_diffusion_amd import synthetic
    runner, cfg = build_runner(args, device)
    eng = runner.engine
    eng.seed(1234, first_image=rank * args.batch)
    B, K, T, mc, C = args.batch, args.members, args.timesteps, args.mc, 2
    # the batch is resident in HBM before the timed region, in the library's input buffer (a loader's H2D copy would land there)
    images = eng.batch_buffers(B, mc, T, (3, 224, 224))["images"]
    images.copy_(synthetic.images(B, seed=1234 + rank, device=device))
    torch.cuda.synchronize(device)

    def step(noise=None):
        # ONE library call = one hipGraph launch: conditioner, encoder hoist, K*T reverse steps (noise drawn in-library), aggregation
        out = runner.predict_batch(images, noise=noise, clone=False)
        if dist_on:
            out["prob_all"] = nd_dist.all_gather_rows(out["prob"], B * world, world, force_collective=force_dist)   # the single collective
        return out

    def barrier():
        # RCCL: name the device, or the first barrier of a group guesses it from the rank (and warns)
        if torch.distributed.get_backend() == "nccl":
            torch.distributed.barrier(device_ids=[local])
        else:
            torch.distributed.barrier()

    def timed_steps(n):
        torch.cuda.synchronize(device)
        if dist_on:
            barrier()
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(n):
            out = step()
        torch.cuda.synchronize(device)
        if dist_on:
            barrier()
            torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if dist_on:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, out

    # THE timed region: W warmup steps, then exactly K steps.  Event-record nodes sit inside the sampler graph (8 probed
    # denoising steps per replay, on the stream the kernels are launched on): they are what `roofline.achieved` is measured
    # with, and they make `value` slightly conservative (unprobed_ms_per_step below is the same loop without them).
    eng.set_profiling(True)
    for _ in range(args.warmup):
        step()
    dt, out = timed_steps(args.steps)
    head_us, pair_us, rec_us, n_probe = eng.profile_read()
    eng.set_profiling(False)
    chk = step()
    # the gathered tensor holds this rank's rows where they belong (out["prob"] is a fixed buffer: compared before the next step)
    collective_checked = bool(torch.equal(chk["prob_all"][rank * B:(rank + 1) * B], chk["prob"])) if dist_on else None
    dt_unprobed, _ = timed_steps(args.steps)

    # the same loop with every batch handed over as a HOST tensor by a loader, through the runner's own batch loop (runner._rank_batches,
    # what test_atk iterates): pinned staging, a non-blocking copy on a side stream into one of two device buffers, issued one batch
    # ahead so that batch n + 1's 19.3 MB cross PCIe under batch n's graph.  Reported beside `value`, never as `value` (rank 0, N = 1).
    host_in = None
    if rank == 0 and world == 1:
        host_images = images.cpu().pin_memory()          # as the runner's DataLoader(pin_memory=True) hands batches over
        n_h = max(1, min(args.steps, 50))
        tgt = torch.zeros(B, dtype=torch.int64)

        def host_loop(n):
            for x, _ in runner._rank_batches([(host_images, tgt)] * n, 0, B, B):
                runner.predict_batch(x, clone=False)
        host_loop(2)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        host_loop(n_h)
        torch.cuda.synchronize(device)
        host_in = (time.perf_counter() - t0) / n_h

    # stage breakdown (outside the timed region, torch events on the launch stream)
    def timed(fn, reps=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize(device)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) / reps
    flat = torch.flatten(images, 1)
    yhat = out["yhat"].clone()
    noise = torch.randn(K, T, B * mc, C, device=device)
    stages = {
        "conditioner_ms": timed(lambda: runner.compute_guiding_prediction(images, include_full_vit=False)),
        "encoder_hoist_ms": timed(lambda: eng.encode(flat)),
        "sampler_ms": timed(lambda: eng.sample(yhat, yhat, noise, mc=mc, T=T)),
    }

    if rank != 0:
        if dist_on:
            torch.distributed.destroy_process_group()
        return
    units = B * K * mc * T
    value = world * units * args.steps / dt
    F = 4096
    M = B * mc
    wb = 4.0 if args.dtype == "f32" else 2.0
    # mean duration of one step-block launch: the lin2 and lin3(+lin4) launches of a probed step sit in ONE event interval; the
    # empty interval recorded right behind it is what a record node adds to any interval
    avg_us = 0.5 * (pair_us - rec_us) if n_probe else float("nan")
    ovh_us = rec_us if n_probe else 0.0
    plan = eng.step_plan(M, K)
    if plan["kernel"] == "k_skinny":
        # ALGORITHMIC bytes of ONE launch of the dominant kernel (a ConditionalLinear block of all K members): weights F*F,
        # gain/shift rows 2*F, activations in M*F and out M*F
        alg = K * (wb * F * F + 4.0 * 2 * F + wb * 2 * M * F)
        achieved = alg / (avg_us * 1e-6) / 1e9 if n_probe else None
        roof = {"bound": "hbm", "kernel": plan["name"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "alg_bytes_per_launch": alg,
                "label": "HBM + Infinity Cache: algorithmic bytes DELIVERED to the CUs per second over the 8 TB/s HBM3E peak"}
        # Part of the step weights is kept resident in the 256 MiB Infinity Cache across steps (default-policy loads, the rest is
        # streamed from HBM with nontemporal loads), so `achieved` is not a DRAM rate.  `hbm_side` subtracts the resident bytes:
        # the rate the DRAM itself has to sustain for this launch time.
        keep2, keep3 = eng.resident_weight_bytes()
        res = 0.5 * (keep2 + keep3)
        roof["hbm_side"] = ({"achieved": (alg - res) / (avg_us * 1e-6) / 1e9, "frac": (alg - res) / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "unit": "GB/s", "resident_bytes_per_launch": res,
                             "note": f"lin2 launches read {keep2 / 1e6:.0f} MB and lin3 launches {keep3 / 1e6:.0f} MB of weights out of the "
                                     "Infinity Cache (a memory-side cache: those bytes still cross the same fabric links to the XCDs)"}
                            if n_probe else None)
        traffic, why = shape_traffic(args.dtype, M, K)
        roof["traffic"] = traffic
        # the same launches also issue K*2*M*F*F exact-f32 MFMA flop: at M = 32 rows that is 34 us of matrix-pipe time per launch,
        # so the kernel sits against BOTH the stream and the f32 matrix pipe (PMC SQ_VALU_MFMA_BUSY 0.55, profiles/r02_pmc_mfma.csv)
        roof["co_limit_mfma_frac"] = (K * 2.0 * M * F * F / (avg_us * 1e-6) / 1e12 / F32_MFMA_PEAK_TF) if (n_probe and args.dtype == "f32") else None
        roof["traffic_source"] = why
        if args.dtype == "f32" and K == 5 and M == 32:
            # REPLAYED, like `traffic`: a timing ablation of this very shape (variant library, every weight load of k_skinny served from
            # cache, same instructions and MFMAs) -- what the launch would take with unlimited bandwidth for the weights
            roof["launch_us_with_weights_from_cache"] = {"value": 50.8, "source": "profiles/r06_skinny_weight_traffic_ablation.txt (50.3-51.3 us against "
                                                        "57.2-57.8 on that box): the weight bytes are worth 1.12 x; the rest is the six-fragment workgroups' "
                                                        "exact-f32 MFMAs (45 us at 2.4 GHz), ramp and epilogue"}
    else:
        # M = B*mc rows > 128: the blocks are compute-bound GEMMs (2*M*F*F flop per member and launch).  fp32 arithmetic runs on the
        # bf16 matrix pipe with exact products (nine bf16 pair products per fp32 product, csrc/nd_b9.hpp): the peak for USEFUL fp32
        # flop is then the dense bf16 peak / 9; the f32-input MFMA form (ND_STEP_F32_MFMA=1) has 157.3 TFLOP/s
        alg = K * 2.0 * M * F * F
        achieved = alg / (avg_us * 1e-6) / 1e12 if n_probe else None
        b9 = args.dtype == "f32" and plan.get("b9")
        peak = (BF16_MFMA_PEAK_TF / 9.0 if b9 else F32_MFMA_PEAK_TF) if args.dtype == "f32" else F16_MFMA_PEAK_TF
        roof = {"bound": "mfma", "kernel": plan["name"], "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": (achieved / peak) if achieved else None, "alg_flop_per_launch": alg,
                "label": ("useful fp32 TFLOP/s over (dense bf16 MFMA peak 2500 TFLOP/s) / 9: each fp32 product costs nine bf16 MFMA products; "
                          f"the same launch is {achieved / F32_MFMA_PEAK_TF:.2f} of the 157.3 TFLOP/s f32-input MFMA peak it used to be priced against"
                          if (b9 and achieved) else "fp32 TFLOP/s over the dense MFMA peak of the operand type")}
        roof["traffic"], roof["traffic_source"] = shape_traffic(args.dtype, M, K)
    roof["avg_launch_us"] = avg_us
    roof["probe"] = {"head_interval_us": head_us, "lin2_plus_lin3_interval_us": pair_us, "record_node_us": ovh_us, "pairs_probed": n_probe,
                     "note": "HIP event-record NODES inside the timed batch graph, on the launch stream, around pairs of consecutive steps (i, i+1): "
                             "one interval around head(i), ONE around the two ConditionalLinear launches of step i, one around the whole unrecorded "
                             "step i+1; (whole step) - (two launches) = the head alone, and record_node_us = head interval - head alone is what a "
                             "record node adds to a loaded interval; avg_launch_us = (lin2_plus_lin3_interval_us - record_node_us) / 2, to be compared "
                             "with the rocprofv3 begin->end averages of the two k_skinny / k_cond_gemm_b9 rows in profiles/ (a profiled run clocks "
                             "2-3 % lower)"}
    line = {
        "metric": "denoising-steps*images/sec (K=5,T=100,224^2)", "value": value, "unit": "denoising-step*images/s",
        "n_gpus": world, "n_ranks_seen": n_ranks_seen,
        "dist_backend": torch.distributed.get_backend() if dist_on else None,
        "collective_checked": collective_checked,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.dtype == "f32" else "f16 operands / f32 accumulate (secondary mode, not the reference's arithmetic)",
        "data": "synthetic",
        "config": {"workload": f"K={K} members, T={T} steps, B={B} images/GPU (3x224x224), mc={mc}, D=150528, F=H=4096; "
                               "whole hot path per step: ViT-prefix+mapping MLPs, encoder hoist, K*T reverse steps, aggregation",
                   "global_batch": B * world, "parallelism": f"dp{world} (batch-sharded, all K members per GPU, one all-gather)"},
        "roofline": roof,
        "probes_in_timed_region": True,
        "unprobed_ms_per_step": 1e3 * dt_unprobed / args.steps,
        "unprobed_value": world * units * args.steps / dt_unprobed,
        "pcie_inclusive": ({"ms_per_step": 1e3 * host_in, "value": units / host_in,
                            "note": "every batch handed over as a pinned host tensor through the runner's loader loop (side-stream H2D one batch ahead, "
                                    "event wait, device-to-device copy into the library's input buffer), transfers inside the timing; not the headline"}
                           if host_in else None),
        "stages_ms": stages,
        "sampler_only_value": units / (stages["sampler_ms"] * 1e-3),
    }
    if args.cpu_baseline:
        g = torch.Generator(device=device).manual_seed(4321)
        nz = torch.randn(K, T, B * mc, C, device=device, generator=g)
        out_fixed = {k: v.clone() for k, v in step(nz).items()}       # one more (untimed) HIP step on recorded noise
        torch.cuda.synchronize(device)
        vit_cpu, mlps_cpu, members_cpu = host_copies(args, device)
        line["cpu_baseline"] = cpu_baseline(members_cpu, vit_cpu, mlps_cpu, images.cpu(), out_fixed, nz, T, runner.temperature,
                                            args.cpu_seconds)
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
