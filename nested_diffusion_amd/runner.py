"""Mirror of the reference runner's inference path: class Diffusion of
diffusion/classification_train_separately.py -- __init__ schedule block (:215-226), conditioner
loading (:249-275), compute_guiding_prediction (:330-348), convert_to_prob (:392-398),
compute_ensemble_confidence (:425-447), majority_voting_for_mc_samples (:51-68) and the hot loop
of test_atk (:749-794), plus the report metrics and test_calibrate.  Training and attacks are out of scope.

Host code is PyTorch plumbing; every tensor operation of the hot path runs in libnd_hip.so.
"""
from __future__ import annotations

import logging
import os
import time
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .diffusion_utils import make_beta_schedule
from .engine import EnsembleEngine
from .mapping import GuidingConditioner, load_checkpoint_object, load_conditioner
from . import dist as nd_dist

CHEST = ['ChestXRay', 'ChestXRayAtkFGSM', 'ChestXRayAtkPGD', 'ChestXRayAtkBIM', 'ChestXRayAtkAUTOPGD', 'ChestXRayAtkCW',
         'ChestXRayValidate']
ISIC = ['ISICSkinCancer', 'ISICSkinCancerAtkFGSM', 'ISICSkinCancerAtkPGD', 'ISICSkinCancerAtkBIM',
        'ISICSkinCancerAtkAUTOPGD', 'ISICSkinCancerAtkCW', 'ISICSkinCancerValidate']


def majority_voting_for_mc_samples(predictions: Sequence[torch.Tensor]) -> torch.Tensor:
    """classification_train_separately.py:51-68 -- mode over samples of argmax(raw y_0); ties -> smallest label."""
    samples = torch.stack([p if p.is_cuda else p.cuda() for p in predictions]).float()
    _, vote, _ = ops.aggregate(samples, 1.0)
    return vote.to(predictions[0].device)


def set_seed(seed) -> None:
    """:31-38, called from Diffusion.__init__ (:198): torch (CPU + every GPU), numpy and python `random`."""
    import random
    import numpy as np
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def temperature_for(dataset: str) -> float:
    """classification_train_separately.py:318-327."""
    if dataset in CHEST:
        return 0.1737
    if dataset in ISIC:
        return 0.3162
    raise NotImplementedError(dataset)


class Diffusion(object):
    def __init__(self, args, config, device=None, conditioner: Optional[GuidingConditioner] = None,
                 noise_estimator_states: Optional[List[Dict[str, torch.Tensor]]] = None):
        """`conditioner` / `noise_estimator_states` let synthetic runs (no checkpoints on disk) inject
        weights; otherwise they are read from the reference's checkpoint layout."""
        self.args, self.config = args, config
        self.seed = getattr(args, "seed", 0)
        if self.seed is not None:
            set_seed(self.seed)                                              # :198
        # operand dtype of the weight-streaming layers: --fp16 (or model.operand_dtype: f16 in the YAML) selects the fp16
        # mode of BASELINE config 5; the reference itself only has fp32
        self.operand_dtype = "f16" if getattr(args, "fp16", False) else getattr(config.model, "operand_dtype", "f32")
        if device is None:
            device = torch.device("cuda")
        self.device = torch.device(device)
        self.num_timesteps = config.diffusion.timesteps
        self.mc_trials = int(getattr(args, "mc_trials", 20) or 20)          # hard-coded 20 at :770
        # schedule (:215-226): same torch ops as the reference, fp32
        betas = make_beta_schedule(schedule=config.diffusion.beta_schedule, num_timesteps=self.num_timesteps,
                                   start=config.diffusion.beta_start, end=config.diffusion.beta_end)
        betas = betas.float()                                    # T-float init tables: host ops in the reference's
        alphas = 1.0 - betas                                     # order (CPU cumprod), pinned by tests/golden/schedule.npz
        omabs = torch.sqrt(1 - alphas.cumprod(dim=0))
        if config.diffusion.beta_schedule == "cosine":
            omabs = omabs * 0.9999
        self.betas = betas.to(self.device)
        self.alphas = alphas.to(self.device)
        self.one_minus_alphas_bar_sqrt = omabs.to(self.device)
        self.temperature = temperature_for(config.data.dataset)
        self.selected_block_indices = [0, 1, 2, 3, 4]                       # :275
        if conditioner is None:
            if config.diffusion.aux_cls.arch != "sevit":
                raise NotImplementedError("only aux_cls.arch == 'sevit' is on the hot path")
            ds = "ChestXRay" if config.data.dataset in CHEST else "ISICSkinCancer"
            conditioner = load_conditioner(config.diffusion.trained_aux_cls_ckpt_path, ds, self.device, dtype=self.operand_dtype)
        self.cond_pred_model = conditioner
        self.num_noise_estimators_required = len(conditioner.mlps) + 1      # :274 (the +1 is never sampled, Q1)
        self._states = noise_estimator_states
        self.engine: Optional[EnsembleEngine] = None
        self.members: List[int] = []
        self.bytes_uploaded = 0          # image bytes this rank has sent over PCIe (its shard of every batch, nothing else)

    # ---- conditioner -------------------------------------------------------------------------
    def compute_guiding_prediction(self, x, include_full_vit: bool = True):
        """:330-348."""
        return self.cond_pred_model.compute_guiding_prediction(x, include_full_vit)

    # ---- aggregation --------------------------------------------------------------------------
    def convert_to_prob(self, logits: torch.Tensor) -> torch.Tensor:
        """:392-398 -- softmax(-(y-1)^2 / T)."""
        dev = logits.device
        x = logits.float().to(self.device).reshape(1, -1, logits.shape[-1]).contiguous()
        _, _, probs = ops.aggregate(x, self.temperature, return_probs=True)
        return probs.reshape(logits.shape).to(dev)

    def compute_ensemble_confidence(self, outputs: List[torch.Tensor]) -> torch.Tensor:
        """:425-447 -- replaces every list entry by its probabilities (quirk Q4) and returns the mean."""
        dev = outputs[0].device
        samples = torch.stack([o.to(self.device) for o in outputs]).float().contiguous()
        prob, _, probs = ops.aggregate(samples, self.temperature, return_probs=True)
        for i in range(len(outputs)):
            outputs[i] = probs[i].to(dev)
        return prob.to(dev)

    # ---- noise estimators ---------------------------------------------------------------------
    def load_noise_estimators(self, max_batch: int, mc_trials: Optional[int] = None) -> None:
        """:684-697.  Members actually sampled are selected_block_indices ∩ available checkpoints
        (the reference asks for len(mlps)+1 paths and dies on the 6th, quirk Q1)."""
        cfg = self.config
        mc = mc_trials or self.mc_trials
        if self._states is None:
            paths = cfg.diffusion.trained_diffusion_ckpt_path[0]
            states = []
            for i in range(min(len(paths), self.num_noise_estimators_required)):
                state = load_checkpoint_object(paths[i])      # {'noise_estimator': state_dict, 'optimizer': ..., 'epoch': ...} (:1120-1126)
                states.append(state["noise_estimator"])
                logging.info("Diffusion model %d loaded", i)
            self._states = states
        self.members = [i for i in self.selected_block_indices if i < len(self._states)]
        K = len(self.members)
        self.engine = EnsembleEngine(cfg.data.num_classes, cfg.model.data_dim, cfg.model.hidden_dim, cfg.model.feature_dim,
                                     self.num_timesteps, n_members=K, max_batch=max_batch, max_rows=max_batch * mc,
                                     device=self.device, dtype=self.operand_dtype)
        for slot, i in enumerate(self.members):
            self.engine.load_member(slot, self._states[i])
        self.engine.set_schedule(self.alphas, self.one_minus_alphas_bar_sqrt)
        self._seeded_first = None
        self._seed_noise(0)            # in-library noise (predict_batch without a noise tensor)
        self._states = None            # device copies live in the engine

    def _seed_noise(self, first_image: int) -> None:
        """Seed the library's generator ONCE per (run, shard): nd_seed resets the device-side batch counter, so seeding again before
        every pass over the data (each Nelder-Mead evaluation of `--calib` under ND_CALIB_RESAMPLE=1, or test_atk after it) would
        replay identical noise.  Later passes just let the counter run on."""
        if self._seeded_first != first_image:
            self.engine.seed(self.seed or 0, first_image=first_image)
            self._seeded_first = first_image

    # ---- the hot path (:749-794) --------------------------------------------------------------
    @torch.no_grad()
    def predict_batch(self, images_224: torch.Tensor, noise: Optional[torch.Tensor] = None, mc_trials: Optional[int] = None,
                      clone: bool = True, use_graph: bool = True):
        """images [B,3,224,224] on the GPU -> dict(samples [K*mc, B, C] raw y_0 member-major then trial,
        vote [B], prob [B, C], probs [K*mc, B, C], yhat [K, B, C]).  noise: optional [K, T, B*mc, C] in reference draw order
        (row = trial*B + image); None = drawn inside the library (the reference draws inside its loop, diffusion_utils.py:67,139).
        ONE library call: the conditioner (:753), the softmax (:755-758), the encoder hoist, the K x mc sampling loops
        (:767-777) and the aggregation (:786-789) are one hipGraph launch after the first batch of a shape."""
        if self.engine is None:
            raise RuntimeError("call load_noise_estimators() first")
        if self.members != list(range(len(self.members))) or len(self.members) > len(self.cond_pred_model.mlps):
            raise RuntimeError(f"member k is conditioned on mapping MLP k: the loaded noise estimators {self.members} must be "
                               f"0..K-1 with K <= {len(self.cond_pred_model.mlps)} mapping MLPs")
        mc = mc_trials or self.mc_trials
        eng, T = self.engine, self.num_timesteps
        images_224 = ops._f32(images_224, "images")
        cond = self.cond_pred_model.handle(images_224.shape[0], images_224.shape[-1])
        return eng.predict_batch(cond, images_224, noise, mc, T, self.temperature, use_graph=use_graph, clone=clone)

    # ---- world-size independent randomness ---------------------------------------------------------
    def draw_noise(self, B_total: int, lo: int, hi: int, mc: Optional[int] = None) -> torch.Tensor:
        """torch-generator draws for rows [lo, hi) of a test batch of B_total images: [K, T, (hi-lo)*mc, C], for callers that want
        explicit noise (test_atk itself uses the library's generator, which has the same property at no cost).
        Every rank holds the same --seed (the reference's set_seed, :31-38), draws the draws of the WHOLE batch
        [K, T, mc, B_total, C] and keeps its own images, so image i sees the same K*T*mc draws at any world size
        (and rows of different ranks are not copies of each other)."""
        mc = mc or self.mc_trials
        K, T, C = len(self.members), self.num_timesteps, self.config.data.num_classes
        z = torch.randn(K, T, mc, B_total, C, device=self.device)
        return z[:, :, :, lo:hi].reshape(K, T, mc * (hi - lo), C).contiguous()

    def _perturbs(self) -> bool:
        """Does any flag of the robustness protocol change the pixels on the device (perturb: :726-737)?"""
        a = self.args
        covered = getattr(a, "covered", (0.0, 0.0)) or (0.0, 0.0)
        return bool((getattr(a, "noise_perturbation", 0.0) or 0.0) > 0.0 or (getattr(a, "low_resolution", 0) or 0) > 1
                    or (getattr(a, "brightness", 0.0) or 0.0) != 0.0 or getattr(a, "contrast", 1.0) not in (1.0, None)
                    or covered[0] > 0 or (getattr(a, "crop", 0.0) or 0.0) > 0)

    def _rank_batches(self, test_loader, lo: int, hi: int, B_total: int):
        """The rank's slice of every batch of the loader, on the device: yields (images [hi-lo, 3, S, S] perturbed, targets [hi-lo]).
        A loader that is already sharded (data.get_test_loader(shard=...): its .shard attribute) hands over the rank's rows only -- only
        those files were decoded; a full-batch loader (the reference's, or a caller's) is sliced on the HOST.  The slice is pinned
        (as handed over, or through a staging buffer) and copied on a side stream, never on the compute stream
        (classification_train_separately.py:722 is a blocking .to(device) of the whole batch on the compute stream):
          * no device-side perturbation configured: STRAIGHT into the library's input buffer, as soon as the previous batch's graph has
            read that buffer (nd_set_input_flag: about a quarter into the batch) -- the transfer runs beside the previous batch's sampler,
            with no staging buffer on the device and no device-to-device copy;
          * otherwise: into one of two device buffers one batch ahead (perturb then makes new tensors out of it)."""
        main = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device)
        sharded = getattr(test_loader, "shard", None) == (lo, hi)
        pins, bufs, evs, held = [None, None], [None, None], [None, None], [None, None]
        tpins, tbufs = [None, None], [None, None]

        def rows_of(item):
            images_raw, target = item
            x = images_raw if sharded else images_raw[lo:hi]
            t = target if sharded else target[lo:hi]
            if x.shape[0] != hi - lo:
                raise ValueError(f"loader handed {x.shape[0]} rows for the shard [{lo}, {hi})")
            return x, t

        def host_stage(k, x, t):
            """pinned source of the batch's H2D copy (the tensor itself if the loader pinned it) and of its targets"""
            if evs[k] is not None:
                evs[k].synchronize()                        # the copy that last read staging buffer k (two batches ago)
            if x.dtype == torch.float32 and x.is_pinned():  # a loader with pin_memory=True: nothing to stage
                src = x
            else:
                if pins[k] is None or pins[k].shape != x.shape:
                    pins[k] = torch.empty(x.shape, dtype=torch.float32, pin_memory=True)
                if x.dtype == torch.float32 and x.is_contiguous():
                    # one memcpy on this thread.  (Tensor.copy_ spreads a 19 MB copy over every logical CPU the box shows; under a
                    # container's CPU quota that takes 20-60 ms instead of 2 -- measured: bench.py's pcie_inclusive leg)
                    import numpy as np
                    np.copyto(pins[k].numpy(), x.numpy())
                else:
                    pins[k].copy_(x)
                src = pins[k]
            # the targets ride along (pinned too: a pageable H2D copy on the compute stream would hold the HOST until the running batch
            # has finished, and the launches behind it would start late)
            if tpins[k] is None or tpins[k].shape != t.shape or tpins[k].dtype != t.dtype:
                tpins[k] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                tbufs[k] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            tpins[k].copy_(t)
            held[k] = src                                   # a caller's pinned batch stays alive until its copy has been waited for
            return src

        first = None
        it = iter(test_loader)
        for first in it:
            break
        if first is None:
            return
        x0, _ = rows_of(first)
        direct = (not x0.is_cuda) and not self._perturbs() and self.engine is not None and x0.dim() == 4
        if direct:
            eng = self.engine
            eng.enable_input_flag()
            fixed = eng.batch_buffers(hi - lo, self.mc_trials, self.num_timesteps, tuple(x0.shape[1:]))["images"]
            k, item = 0, first
            while item is not None:
                x, t = rows_of(item)
                src = host_stage(k, x, t)
                eng.inputs_consumed()                       # every batch launched so far has read `fixed`
                with torch.cuda.stream(side):
                    fixed.copy_(src, non_blocking=True)
                    tbufs[k].copy_(tpins[k], non_blocking=True)
                    evs[k] = torch.cuda.Event()
                    evs[k].record(side)
                main.wait_event(evs[k])
                self.bytes_uploaded += x.numel() * 4
                tdev = tbufs[k]
                k ^= 1
                yield fixed, tdev
                item = next(it, None)
            return

        def stage(k, item):
            x, t = rows_of(item)
            if x.is_cuda:                                   # a caller's loader that already lives on the device
                return x.to(self.device, torch.float32), t.to(self.device), None
            if bufs[k] is None or bufs[k].shape != x.shape:
                bufs[k] = torch.empty(x.shape, dtype=torch.float32, device=self.device)
            src = host_stage(k, x, t)
            side.wait_stream(main)                          # bufs[k]'s last reader (batch n - 1, already enqueued) before it is overwritten
            with torch.cuda.stream(side):
                bufs[k].copy_(src, non_blocking=True)
                tbufs[k].copy_(tpins[k], non_blocking=True)
                evs[k] = torch.cuda.Event()
                evs[k].record(side)
            self.bytes_uploaded += x.numel() * 4
            return bufs[k], tbufs[k], evs[k]

        nxt = stage(0, first)
        k = 1
        for item in it:
            cur, nxt = nxt, stage(k, item)
            k ^= 1
            yield self._finish_batch(cur, main, lo, hi, B_total)
        yield self._finish_batch(nxt, main, lo, hi, B_total)

    def _finish_batch(self, staged, main, lo, hi, B_total):
        x, t, ev = staged
        if ev is not None:
            main.wait_event(ev)
        return self.perturb(x, lo, hi, B_total), t

    # ---- temperature calibration (:449-629, driven by main.py:356-361) ---------------------------
    def test_calibrate(self, temp=None, test_loader=None):
        """ECE of the ensemble at scaling temperature `temp` on the validation set (:449-629).  The reference
        re-runs the whole sampler for every temperature the Nelder-Mead search tries although the samples y_0 do not
        depend on it; here the raw samples are drawn once and cached (set ND_CALIB_RESAMPLE=1 for the as-written,
        stochastic objective), so each further evaluation is one aggregation + one calibration-error kernel."""
        if temp is not None:
            self.temperature = float(temp[0] if hasattr(temp, "__len__") else temp)
        resample = bool(int(os.environ.get("ND_CALIB_RESAMPLE", "0")))
        if resample or getattr(self, "_calib_cache", None) is None:
            rank, world = nd_dist.rank_world()
            B = self.config.testing.batch_size
            lo, hi = nd_dist.shard_bounds(B, rank, world)
            if test_loader is None:
                from .data import get_test_loader
                test_loader = get_test_loader(self.args, self.config, shard=(lo, hi) if world > 1 else None)
            if self.engine is None:
                self.load_noise_estimators(max_batch=max(hi - lo, 1))
            self._seed_noise(lo)
            samples, targets = [], []
            for images, target in self._rank_batches(test_loader, lo, hi, B):
                out = self.predict_batch(images)
                S = out["samples"].shape[0]
                flat = out["samples"].permute(1, 0, 2).reshape(hi - lo, -1)                      # [B_local, S*C]
                flat = torch.cat([flat, target.to(torch.float32)[:, None]], dim=1).contiguous()  # + the rows' targets
                flat = nd_dist.all_gather_rows(flat, B, world)
                samples.append(flat[:, :-1].reshape(B, S, -1).permute(1, 0, 2).contiguous())
                targets.append(flat[:, -1].to(torch.int64))
            self._calib_cache = (torch.cat(samples, dim=1).contiguous(), torch.cat(targets))
        samples, targets = self._calib_cache
        prob, vote, _ = ops.aggregate(samples, self.temperature)              # compute_ensemble_confidence (:612)
        rep = ops.report(prob, prob, prob, vote, targets, self.temperature, n_bins=10)   # compute_ece (:619)
        ece = float(rep["ece"])
        print(f"Ours ECE: {ece} \n")
        logging.info(f"Ours ECE: {ece} \n")
        return ece

    # ---- input perturbations (:726-737), in the reference's order ------------------------------
    def perturb(self, images_224: torch.Tensor, lo: int = 0, hi: Optional[int] = None, B_total: Optional[int] = None) -> torch.Tensor:
        """images_224: rows [lo, hi) of a test batch of B_total images (default: the whole batch).  Every rank holds the same
        --seed (set_seed, :31-38) and makes the reference's RNG calls for the WHOLE batch in the reference's order -- the device
        draw of add_noise (utils.py:274), python `random` for the cover rectangles (:321-343), torch.randint for the crop corners
        (:296-300) -- and applies rows [lo, hi) of them to its own images: image i gets the same noise, windows and crops at
        world 1, 2, 4, 8, and only the shard's pixels are ever moved or touched."""
        from . import perturb as P
        a = self.args
        n_local = images_224.shape[0]
        hi = lo + n_local if hi is None else hi
        B_total = n_local if B_total is None else B_total
        if hi - lo != n_local or not (0 <= lo <= hi <= B_total):
            raise ValueError(f"perturb: {n_local} rows for the shard [{lo}, {hi}) of {B_total}")
        if (getattr(a, "noise_perturbation", 0.0) or 0.0) > 0.0:
            z = torch.randn((B_total,) + tuple(images_224.shape[1:]), dtype=torch.float32, device=images_224.device)
            images_224 = P.add_noise(images_224, a.noise_perturbation, z=z[lo:hi])
        if (getattr(a, "low_resolution", 0) or 0) > 1:
            images_224 = P.down_up_sample(images_224, a.low_resolution)
        if (getattr(a, "brightness", 0.0) or 0.0) != 0.0:
            images_224 = P.adjust_brightness(images_224, a.brightness)
        if getattr(a, "contrast", 1.0) not in (1.0, None):
            images_224 = P.adjust_contrast(images_224, a.contrast)
        covered = getattr(a, "covered", (0.0, 0.0)) or (0.0, 0.0)
        if covered[0] > 0:
            H, W = images_224.shape[-2:]
            _, rects = P.pick_cover_regions(B_total, H, W, covered[0], int(covered[1]))
            images_224 = P.random_cover_new(images_224, covered, rects=rects[lo:hi])
        if (getattr(a, "crop", 0.0) or 0.0) > 0:
            corners = P.pick_crop_corners(B_total, images_224.shape[-1], a.crop)
            images_224 = P.random_crop_and_resize(images_224, a.crop, corners=corners[lo:hi])
        return images_224

    # ---- test loop ----------------------------------------------------------------------------
    def test_atk(self, test_loader=None):
        """:631-840: input perturbations (:726-737), the hot path, and the report the reference prints (accuracy, ECE,
        per-class PIW and variances, :801-838).  Adversarial attacks (:738-739) need ViT gradients and raise."""
        args, config = self.args, self.config
        if getattr(args, "attack_name", None) not in (None, "None"):
            raise NotImplementedError("adversarial attacks need gradients through the ViT: out of scope")
        rank, world = nd_dist.rank_world()
        B = config.testing.batch_size
        lo, hi = nd_dist.shard_bounds(B, rank, world)
        if test_loader is None:
            from .data import get_test_loader
            test_loader = get_test_loader(args, config, shard=(lo, hi) if world > 1 else None)   # each rank decodes its own rows only
        self.load_noise_estimators(max_batch=max(hi - lo, 1))
        # the sampler's draws come from the library's counter-based generator keyed on the GLOBAL image index (lo = this rank's
        # first image) and the batch counter: image i sees the same K*T*mc draws at any world size
        self._seed_noise(lo)
        mv_class, target_class, prob_mc, piw_mc, var_mc = [], [], [], [], []
        n_step_img, t0 = 0, time.time()
        for images, target in self._rank_batches(test_loader, lo, hi, B):   # :715-737, this rank's rows only
            out = self.predict_batch(images, clone=False)
            # spread of the K*mc per-sample probabilities per image (what the reference keeps in pred_mc, quirk Q4)
            piw, var = ops.sample_stats(out["probs"])
            packed = torch.cat([out["prob"], piw, var, out["vote"].to(torch.float32)[:, None], target.to(torch.float32)[:, None]], dim=1)
            packed = nd_dist.all_gather_rows(packed, B, world)               # the single RCCL all-gather of a batch
            C = out["prob"].shape[1]
            prob_mc.append(packed[:, :C]); piw_mc.append(packed[:, C:2 * C]); var_mc.append(packed[:, 2 * C:3 * C])
            mv_class.append(packed[:, 3 * C].to(torch.int64)); target_class.append(packed[:, 3 * C + 1].to(torch.int64))
            n_step_img += B * len(self.members) * self.mc_trials * self.num_timesteps
        torch.cuda.synchronize(self.device)
        dt = time.time() - t0
        if self.engine.loop_form() == "one_launch":
            self.engine.persist_status()                                     # raises if a barrier wait of the one-launch loop was abandoned
        rep = ops.report(torch.cat(piw_mc), torch.cat(var_mc), torch.cat(prob_mc), torch.cat(mv_class), torch.cat(target_class),
                         self.temperature, n_bins=10)                        # :801-815
        acc = rep["accuracy"]
        if rank == 0:
            msg = (f"Majority voting accuracy for MC: {rep['accuracy'] :.4f} \n" +
                   f"ECE: {rep['ece'] :.4f} \n" +
                   f"Average correct PIW per class: {rep['piw_correct']} \n" +
                   f"Average incorrect PIW per class: {rep['piw_incorrect']} \n" +
                   f"Average correct variances per class: {rep['var_correct']} \n" +
                   f"Average incorrect variances per class: {rep['var_incorrect']}")
            print(msg)                                                       # :820-825
            logging.info(msg + " \n")                                        # :829-838
            logging.info("throughput: %.1f denoising-step*images/s over %d GPU(s)", n_step_img / max(dt, 1e-9), world)
        self.last_report = rep
        self.last_probs = torch.cat(prob_mc)
        return acc
