"""Host-side owner of one nd_handle: device memory (torch tensors), streams, and the calls into
libnd_hip.so.  PyTorch is plumbing here -- all arithmetic of the hot path runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import MEMBER_WEIGHT_FIELDS, NdBatchOut, NdConfig, NdMemberWeights, check, ptr


def _require_gpu(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise _lib.NdError("nested_diffusion_amd needs an MI355X (torch device 'cuda'); there is no CPU fallback")
    return device


class EnsembleEngine:
    """K ConditionalModel members (latent_model.py:108-167) resident on one GPU.

    Replaces the reference's per-batch load/.to(device)/.to(cpu) shuttle of each noise estimator
    (classification_train_separately.py:685-697, 773, 780): all members stay in HBM.
    """

    def __init__(self, y_dim: int, data_dim: int, hidden_dim: int, feature_dim: int, n_steps: int,
                 n_members: int = 1, max_batch: int = 32, max_rows: Optional[int] = None, device="cuda", dtype="f32"):
        """dtype 'f32': the reference's arithmetic.  'f16': the five large weight matrices and their input
        activations in fp16 (fp32 accumulation / epilogues / state) -- BASELINE config 5, not a reference mode."""
        self.device = _require_gpu(device)
        self.lib = _lib.load()
        max_rows = max_rows if max_rows is not None else max_batch
        self.dtype = _lib.dtype_code(dtype)
        self.cfg = NdConfig(y_dim, data_dim, hidden_dim, feature_dim, n_steps, n_members, max_batch, max_rows, self.dtype)
        self.C, self.D, self.H, self.F, self.T, self.K = y_dim, data_dim, hidden_dim, feature_dim, n_steps, n_members
        self.max_batch, self.max_rows = max_batch, max_rows
        h = C.c_void_p()
        check(self.lib.nd_create(C.byref(self.cfg), C.byref(h)), "nd_create")
        self.h = h
        nbytes = self.lib.nd_workspace_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise _lib.NdError("nd_workspace_bytes: " + self.lib.nd_last_error().decode())
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (self.workspace.data_ptr() + 255) & ~255
        check(self.lib.nd_bind_workspace(self.h, base, nbytes), "nd_bind_workspace")
        self._weights: Dict[int, Dict[str, torch.Tensor]] = {}
        self._sched: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._static: Dict[tuple, Dict[str, torch.Tensor]] = {}
        self._batch_static: Dict[tuple, Dict[str, torch.Tensor]] = {}

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.nd_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- stream ------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    # -- weights -----------------------------------------------------------------------------
    def load_member(self, k: int, state_dict: Dict[str, torch.Tensor]) -> None:
        """state_dict = ConditionalModel.state_dict() (checkpoint key 'noise_estimator',
        classification_train_separately.py:689-690).  Shapes are checked strictly; lin1 may be [F, 2C] (include_guidance: True,
        the shipped configs) or [F, C] (guidance=False)."""
        C_, D, H, F, T = self.C, self.D, self.H, self.F, self.T
        want = {"encoder_x.0.weight": (H, D), "encoder_x.3.weight": (H, H), "encoder_x.6.weight": (F, H),
                "lin1.lin.weight": (F, 2 * C_), "lin2.lin.weight": (F, F), "lin3.lin.weight": (F, F),
                "lin4.weight": (C_, F), "lin4.bias": (C_,)}
        for name in ("lin1", "lin2", "lin3"):
            want[name + ".embed.weight"] = (T + 1, F)
        held: Dict[str, torch.Tensor] = {}
        w = NdMemberWeights()
        for field, key in MEMBER_WEIGHT_FIELDS:
            if key not in state_dict:
                raise KeyError(f"state_dict is missing '{key}'")
            t = self._dev(state_dict[key])
            if key == "lin1.lin.weight" and tuple(t.shape) == (F, C_):
                # member built with guidance=False: lin1 takes y_t alone (latent_model.py:157-158, 172).  The library's step
                # head always reads [y_t, yhat]; zero weights on the yhat half add 0 * yhat = +-0 per term: same sums.
                t = torch.cat([t, torch.zeros_like(t)], dim=1).contiguous()
            if key in want and tuple(t.shape) != want[key]:
                raise ValueError(f"'{key}' has shape {tuple(t.shape)}, expected {want[key]}")
            held[key] = t
            setattr(w, field, t.data_ptr())
        # the library repacks / folds everything into its workspace and synchronises before returning,
        # so `held` (and the caller's tensors) can be released afterwards
        check(self.lib.nd_load_member(self.h, k, C.byref(w), self._stream()), "nd_load_member")
        del held

    def set_schedule(self, alphas: torch.Tensor, one_minus_alphas_bar_sqrt: torch.Tensor) -> None:
        a, s = self._dev(alphas), self._dev(one_minus_alphas_bar_sqrt)
        if a.numel() != s.numel():
            raise ValueError("alphas and one_minus_alphas_bar_sqrt differ in length")
        self._sched = (a, s)
        check(self.lib.nd_set_schedule(self.h, ptr(a), ptr(s), a.numel(), self._stream()), "nd_set_schedule")

    def seed(self, seed: int, first_image: int = 0) -> None:
        """Seed the in-library noise generator (used when no noise tensor is passed): Philox keyed on (seed; global image index,
        member, trial, draw, batch counter).  `first_image` = global index of this rank's first image, so the draws of an image
        do not depend on how the batch is sharded.  Resets the batch counter."""
        check(self.lib.nd_seed(self.h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_image) & 0xFFFFFFFF), "nd_seed")

    # -- compute -----------------------------------------------------------------------------
    def encode(self, x: torch.Tensor, member0: int = 0, n_members: Optional[int] = None) -> None:
        """xe = norm(encoder_x(x)) for a member range, cached in the workspace (latent_model.py:170-171)."""
        n_members = self.K - member0 if n_members is None else n_members
        x = self._dev(x)
        if x.dim() != 2 or x.shape[1] != self.D:
            raise ValueError(f"x must be [B, {self.D}], got {tuple(x.shape)}")
        check(self.lib.nd_encode(self.h, member0, n_members, ptr(x), x.shape[0], self._stream()), "nd_encode")
        self._x_keepalive = x

    def member_buffer(self, k: int, which: int, rows: int) -> torch.Tensor:
        """Copy of an internal buffer: which = 0 xe [rows,F], 1 h1, 2 h2 (tests only)."""
        out = torch.empty(rows, self.F, dtype=torch.float32, device=self.device)
        check(self.lib.nd_member_buffer(self.h, k, which, ptr(out), rows, self._stream()), "nd_member_buffer")
        return out

    def eps_theta(self, member: int, y: torch.Tensor, yhat: torch.Tensor, t: int, mc: int = 1) -> torch.Tensor:
        """ConditionalModel.forward's t-dependent trunk on the cached xe (latent_model.py:172-184)."""
        y, yhat = self._dev(y), self._dev(yhat)
        B = yhat.shape[0]
        if y.shape != (B * mc, self.C):
            raise ValueError(f"y must be [{B * mc}, {self.C}]")
        out = torch.empty_like(y)
        check(self.lib.nd_eps_theta(self.h, member, ptr(y), ptr(yhat), int(t), ptr(out), B, mc, self._stream()), "nd_eps_theta")
        return out

    def p_sample(self, member: int, y: torch.Tensor, yhat: torch.Tensor, ymean: torch.Tensor, t: int,
                 z: Optional[torch.Tensor] = None, mc: int = 1) -> torch.Tensor:
        """One reverse step: p_sample (t >= 1, draw z supplied) or p_sample_t_1to0 (t == 0)
        -- diffusion_utils.py:54-111."""
        y, yhat, ymean = self._dev(y), self._dev(yhat), self._dev(ymean)
        z = self._dev(z) if z is not None else None
        B = yhat.shape[0]
        out = torch.empty_like(y)
        check(self.lib.nd_p_sample(self.h, member, ptr(y), ptr(yhat), ptr(ymean), ptr(z), int(t), ptr(out), B, mc,
                                   self._stream()), "nd_p_sample")
        return out

    def set_loop_form(self, one_launch: bool, skew_us: float = -1.0) -> None:
        """Form of the T-step loop (nd_set_loop_form): per-step kernel nodes of a hipGraph, or ONE launch per p_sample_loop where
        the shape allows (same bits; needs the device to itself while it runs).  skew_us < 0 keeps the current member skew."""
        check(self.lib.nd_set_loop_form(self.h, 1 if one_launch else 0, float(skew_us)), "nd_set_loop_form")

    def loop_form(self) -> str:
        """'one_launch' or 'graph_nodes': what the most recently recorded / enqueued sampling loop ran as."""
        return "one_launch" if self.lib.nd_loop_form(self.h) else "graph_nodes"

    def persist_status(self, reset: bool = False) -> bool:
        """True when no barrier wait of a one-launch loop has been abandoned (synchronises the device); raises otherwise."""
        rc = self.lib.nd_persist_status(self.h, 1 if reset else 0)
        if rc < 0:
            check(rc, "nd_persist_status")
        if rc != 0:
            raise _lib.NdError("one-launch sampling loop: a barrier wait was abandoned (the grid was not resident at once -- another "
                               "process on this device?); results of that call are invalid.  ND_PERSIST=0 selects the per-step form")
        return True

    def set_profiling(self, enable: bool) -> None:
        check(self.lib.nd_set_profiling(self.h, 1 if enable else 0), "nd_set_profiling")

    def profile_read(self):
        """(head_us, pair_us, record_us, n_probed_pairs) of the last sample() / predict_batch(); synchronises the stream.
        head_us: interval around the step head (+ one record node); pair_us: ONE interval around the two ConditionalLinear launches
        (lin2, lin3+lin4; + one record node); record_us: what a record node adds to a loaded interval, calibrated from a whole
        unrecorded step (nd_profile_read).  Mean step-block launch = (pair_us - record_us) / 2; head alone = head_us - record_us."""
        torch.cuda.current_stream(self.device).synchronize()
        us = (C.c_float * 4)()
        n = C.c_int(0)
        check(self.lib.nd_profile_read(self.h, us, C.byref(n)), "nd_profile_read")
        self.probe_overhead_us = float(us[3])
        return float(us[0]), float(us[1]), float(us[3]), int(n.value)

    def probe_nodes(self) -> int:
        """Event-record nodes in the most recently built sampling / batch graph (4 per probed step with profiling on, else 0)."""
        return int(self.lib.nd_profile_probe_nodes(self.h))

    def resident_weight_bytes(self) -> Tuple[int, int]:
        """(lin2, lin3) weight bytes per step launch that are kept Infinity-Cache resident across steps."""
        a, b = self.lib.nd_resident_weight_bytes(self.h, 0), self.lib.nd_resident_weight_bytes(self.h, 1)
        if a < 0 or b < 0:
            raise _lib.NdError("nd_resident_weight_bytes: members not loaded")
        return int(a), int(b)

    def step_plan(self, M: int, n_members: Optional[int] = None) -> Dict[str, object]:
        """Which kernel the lin2 / lin3 blocks of a step run at M = B*mc rows (host-side plan, no launch)."""
        out = (C.c_int * 8)()
        n_members = self.K if n_members is None else n_members
        check(self.lib.nd_step_plan(self.F, int(M), int(n_members), self.dtype, out), "nd_step_plan")
        tile = bool(out[0])
        # the tiled blocks run on the bf16 matrix pipe with exact fp32 products where the library keeps frag32b3 copies of lin2 / lin3
        # (fp32 handles with F % 32 == 0 and at most 8 members per launch; ND_STEP_F32_MFMA=1 keeps the f32-input MFMA kernel)
        b9 = tile and self.dtype == _lib.ND_DTYPE_F32 and self.F % 32 == 0 and n_members <= 8 and not os.environ.get("ND_STEP_F32_MFMA")
        name = (("k_cond_gemm_b9<{0,1}> (LDS-tiled 128x128 ConditionalLinear blocks on the bf16 matrix pipe, nine exact bf16 pair products "
                 "per fp32 product, K members per launch)" if b9 else
                 "k_cond_gemm<{0,1}> (LDS-tiled 128x128 f32-MFMA ConditionalLinear blocks, K members per launch)") if tile else
                "k_skinny<MT,NF,4,U,{0,1}> (weight-streaming lin2 / lin3+lin4 ConditionalLinear blocks, K members per launch)")
        plan = {"kernel": "k_cond_gemm" if tile else "k_skinny", "b9": bool(b9), "name": name, "workgroups": out[1], "whole_tiles": out[2],
                "remainder_tiles": out[3], "split": out[4], "partials": out[5], "TM": out[6], "TN": out[7]}
        if not tile:
            # the weight stream's geometry (nd_skinny_plan): NF = weight-fragment slots per workgroup (1 at K = 1 member, 5-6 at K = 5),
            # MT = row fragments per pass (nd_pick_mt), grid = (workgroups, row passes, k-slabs)
            o6 = (C.c_int * 6)()
            check(self.lib.nd_skinny_plan(self.F, self.F, int(M), int(n_members), self.dtype, 1, o6), "nd_skinny_plan")
            mt = self.lib.nd_skinny_row_fragments(int(M))             # the launcher's own choice (nd_pick_mt), not a copy of it
            if mt < 1:
                check(mt, "nd_skinny_row_fragments")
            plan["stream"] = {"grid": (o6[0], o6[1], o6[2]), "NF": o6[3], "MT": mt, "chunks_per_slab": o6[4], "threads": o6[5]}
        return plan

    def static_buffers(self, n_members: int, B: int, mc: int, T: int, seq: bool) -> Dict[str, torch.Tensor]:
        """Fixed-address I/O tensors so the hipGraph of a (members, B, mc, T) shape is built once."""
        key = (n_members, B, mc, T, seq)
        buf = self._static.get(key)
        if buf is None:
            M, C_ = B * mc, self.C
            dev = self.device
            buf = {"yhat": torch.empty(n_members, B, C_, device=dev), "ymean": torch.empty(n_members, B, C_, device=dev),
                   "noise": torch.empty(n_members, T, M, C_, device=dev), "y0": torch.empty(n_members, M, C_, device=dev)}
            if seq:
                buf["seq"] = torch.empty(n_members, T + 1, M, C_, device=dev)
            self._static[key] = buf
        return buf

    def sample(self, yhat: torch.Tensor, ymean: torch.Tensor, noise: Optional[torch.Tensor], member0: int = 0,
               n_members: Optional[int] = None, mc: int = 1, T: Optional[int] = None, return_seq: bool = False,
               use_graph: bool = True) -> torch.Tensor:
        """p_sample_loop for a member range x mc trials (diffusion_utils.py:133-163).
        yhat, ymean: [n_members, B, C]; noise: [n_members, T, B*mc, C] in the reference's draw order, or None: the library draws
        it (Philox, see seed()).  Returns y_0 [n_members, B*mc, C] (or the whole trajectory [n_members, T+1, B*mc, C])."""
        n_members = self.K - member0 if n_members is None else n_members
        T = self.T if T is None else T
        if yhat.dim() != 3 or yhat.shape[0] != n_members or yhat.shape[2] != self.C:
            raise ValueError(f"yhat must be [{n_members}, B, {self.C}], got {tuple(yhat.shape)}")
        B = yhat.shape[1]
        M = B * mc
        if tuple(ymean.shape) != tuple(yhat.shape):
            raise ValueError("ymean must have yhat's shape")
        if noise is not None and tuple(noise.shape) != (n_members, T, M, self.C):
            raise ValueError(f"noise must be [{n_members}, {T}, {M}, {self.C}], got {tuple(noise.shape)}")
        if use_graph:
            buf = self.static_buffers(n_members, B, mc, T, return_seq)
            buf["yhat"].copy_(yhat); buf["ymean"].copy_(ymean)
            if noise is not None:
                buf["noise"].copy_(noise)
            yh, ym, nz, y0 = buf["yhat"], buf["ymean"], (buf["noise"] if noise is not None else None), buf["y0"]
            seq = buf.get("seq")
        else:
            yh, ym, nz = self._dev(yhat), self._dev(ymean), (self._dev(noise) if noise is not None else None)
            y0 = torch.empty(n_members, M, self.C, device=self.device)
            seq = torch.empty(n_members, T + 1, M, self.C, device=self.device) if return_seq else None
        check(self.lib.nd_sample(self.h, member0, n_members, ptr(yh), ptr(ym), ptr(nz), ptr(y0), ptr(seq), B, mc, T,
                                 1 if use_graph else 0, self._stream()), "nd_sample")
        out = seq if return_seq else y0
        return out.clone() if use_graph else out

    # -- the whole hot path of a batch -------------------------------------------------------------
    def batch_buffers(self, B: int, mc: int, T: int, image_shape) -> Dict[str, torch.Tensor]:
        """Fixed-address input / output tensors of predict_batch for one (B, mc, T): the batch graph is recorded once per set.
        buf['images'] is the input buffer: a caller that writes its batch THERE (e.g. the H2D copy of a loader) saves the
        device-to-device copy predict_batch otherwise makes."""
        key = (B, mc, T, tuple(image_shape))
        buf = self._batch_static.get(key)
        if buf is None:
            K, C_, dev = self.K, self.C, self.device
            buf = {"images": torch.empty(B, *image_shape, device=dev), "noise": None,
                   "samples": torch.empty(K * mc, B, C_, device=dev), "prob": torch.empty(B, C_, device=dev),
                   "vote": torch.empty(B, dtype=torch.int64, device=dev), "probs": torch.empty(K * mc, B, C_, device=dev),
                   "yhat": torch.empty(K, B, C_, device=dev)}
            self._batch_static[key] = buf
        return buf

    def enable_input_flag(self) -> None:
        """Ask the library to publish, per predict_batch call, the moment its input buffer has been read (nd_set_input_flag): a pinned
        host word receives the count of calls whose inputs are consumed -- about a quarter into the batch -- so a loader can refill
        batch_buffers(...)['images'] for the next batch under this batch's sampler (runner._rank_batches)."""
        if getattr(self, "_input_flag", None) is None:
            self._input_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._batch_calls = 0
            check(self.lib.nd_set_input_flag(self.h, self._input_flag.data_ptr()), "nd_set_input_flag")

    def inputs_consumed(self, timeout_s: float = 30.0) -> None:
        """Block (host) until every predict_batch issued so far has read its input buffer.  No-op without enable_input_flag()."""
        if getattr(self, "_input_flag", None) is None:
            return
        import time
        t0 = time.perf_counter()
        while int(self._input_flag[0]) < self._batch_calls:
            if time.perf_counter() - t0 > timeout_s:
                raise _lib.NdError(f"inputs-consumed signal stuck at {int(self._input_flag[0])} of {self._batch_calls} batches")
            time.sleep(2e-5)

    def predict_batch(self, cond, images: torch.Tensor, noise: Optional[torch.Tensor], mc: int, T: int, temperature: float,
                      use_graph: bool = True, clone: bool = True) -> Dict[str, torch.Tensor]:
        """classification_train_separately.py:749-794 for one batch in ONE library call (nd_predict_batch; one hipGraph launch
        after the first call of a shape): guiding prediction, softmax, encoder hoist, K x mc p_sample_loops, aggregation.
        cond: nd_cond handle (mapping.GuidingConditioner.handle).  images [B, Cin, H, W]; noise [K, T, B*mc, C] or None (Philox).
        clone=False returns the fixed output buffers themselves (overwritten by the next call)."""
        B = images.shape[0]
        if images.dim() != 4 or images[0].numel() != self.D:
            raise ValueError(f"images must be [B, Cin, H, W] with Cin*H*W = {self.D}, got {tuple(images.shape)}")
        M = B * mc
        buf = self.batch_buffers(B, mc, T, images.shape[1:])
        if images.data_ptr() != buf["images"].data_ptr():
            buf["images"].copy_(images)
        nz = None
        if noise is not None:
            if tuple(noise.shape) != (self.K, T, M, self.C):
                raise ValueError(f"noise must be [{self.K}, {T}, {M}, {self.C}], got {tuple(noise.shape)}")
            if buf["noise"] is None:
                buf["noise"] = torch.empty(self.K, T, M, self.C, device=self.device)
            buf["noise"].copy_(noise)
            nz = buf["noise"]
        out = NdBatchOut(ptr(buf["samples"]), ptr(buf["prob"]), ptr(buf["vote"]), ptr(buf["probs"]), ptr(buf["yhat"]))
        check(self.lib.nd_predict_batch(self.h, cond, ptr(buf["images"]), ptr(nz), C.byref(out), B, mc, T, float(temperature),
                                        1 if use_graph else 0, self._stream()), "nd_predict_batch")
        if getattr(self, "_input_flag", None) is not None:
            self._batch_calls += 1
        keys = ("samples", "vote", "prob", "probs", "yhat")
        return {k: (buf[k].clone() if clone else buf[k]) for k in keys}
