"""`SPMM` -- drop-in for the reference's SPMM_models.SPMM on the pretraining path.

Same constructor (`SPMM(tokenizer=None, config=None, loader_len=0, no_train=False)`, SPMM_models.py:17), same
`forward(property_original, text_input_ids, text_attention_mask, alpha=0)` -> `(loss_mlm, loss_mpm*5, loss_ita, loss_itm)`
(:79,:256), same `state_dict()` keys / [out,in] fp32 layout, same `training_step` /
`configure_optimizers` semantics (:338-380) without the pytorch_lightning dependency (not installed on the target image).
All arithmetic runs in the HIP kernels of libspmm_hip.so; there is no eager / CPU fallback."""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Optional

import torch
from torch import nn

from . import ops
from .config import SPMMConfig, is_buffer, student_of
from .options import EngineOptions

try:                                    # the reference subclasses pl.LightningModule (SPMM_models.py:16); keep that when it is importable
    import pytorch_lightning as _pl
    _Base, _HAS_PL = _pl.LightningModule, True
except Exception:                       # not installed on the target image: the members SPMM_models.py:345-386 uses are provided below
    _Base, _HAS_PL = nn.Module, False
from .params import ParamStore
from .step import PretrainStep


class _CosineSchedule:
    """scheduler/cosine_lr.py:69-96 as configured by scheduler_factory.py:27-42 for SPMM_pretrain.py:62-63
    (t_mul=1, decay_rate=1, cycle_limit=1, warmup_prefix=True, t_in_epochs=True): only `step(epoch)` is used."""

    def __init__(self, sched: dict):
        self.s = sched

    def lr_at(self, t: int) -> float:
        s = self.s
        warm, base = s["warmup_epochs"], s["lr"]
        if t < warm:
            return s["warmup_lr"] + t * (base - s["warmup_lr"]) / warm
        t -= warm
        if t // s["epochs"] < 1:
            return s["min_lr"] + 0.5 * (base - s["min_lr"]) * (1 + math.cos(math.pi * t / s["epochs"]))
        return s["min_lr"]


class _FusedAdamW:
    """AdamW(lr, weight_decay on all params, betas (0.9,0.999), eps 1e-8) SPMM_models.py:340 + clip_grad_norm_(5.) :361
    as three launches over the flat arena; exposes `param_groups[0]['lr']` like a torch optimizer."""

    def __init__(self, store: ParamStore, eng: PretrainStep, lr: float, weight_decay: float):
        self.store, self.eng = store, eng
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": (0.9, 0.999), "eps": 1e-8}]
        dev = store.device
        self.normsq = torch.zeros(1, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scalars = torch.zeros(ops.adam_scalars_bytes() // 4, device=dev)

    def zero_grad(self, set_to_none: bool = False):
        ops.zero_(self.store.grad)

    def step(self, fill_lr: bool = True):
        g = self.param_groups[0]
        # a second step() without a backward in between: the previous step's off-path shadow rebuild (weight-gradient stream) must not
        # be overtaken by this step's master update / rebuild
        self.eng.pre_backward_wait()
        if fill_lr:                                  # (a host value: set outside a captured graph, see SPMM.fused_step_graphed)
            self.eng.lr.fill_(g["lr"])
        self.normsq.zero_()
        ops.grad_sqnorm(self.store.grad, self.normsq)
        ops.adamw_step(self.store.flat, self.store.grad, self.store.adam_m, self.store.adam_v, self.store.shadow, lr=self.eng.lr,
                       beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"], normsq=self.normsq,
                       max_norm=5.0, step=self.step_count, nan_flag=self.eng.nan_flag, scalars=self.scalars)
        # the transposed weight shadows are operands of the NEXT backward's data-gradient GEMMs only
        self.store.refresh_shadows(transposed_only=True, part="forward")
        self.eng.off_path(lambda: (self.store.refresh_shadows(transposed_only=True, part="transposed"), self.eng.refresh_padded_shadows()))

    @property
    def grad_norm(self) -> torch.Tensor:
        return self.scalars[4]


class _StepFn(torch.autograd.Function):
    """Autograd boundary: the four losses are functions of every trainable parameter; backward runs the engine's own
    backward with the incoming loss gradients as device-side scales and hands back views of the flat gradient arena."""

    @staticmethod
    def forward(ctx, model, prop, ids, mask, kw, *params):
        losses = model.engine.forward(prop, ids, mask, **kw)
        ctx.model = model
        return tuple(losses[i].clone() for i in range(4))

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        m = ctx.model
        m.engine.gscale.copy_(torch.stack([g.reshape(()).to(torch.float32) for g in (g0, g1, g2, g3)]))
        m.store.grad.zero_()
        m.engine.backward()
        grads = tuple(m.store.g(n) for n in m._param_names)
        return (None, None, None, None, None) + grads


class SPMM(_Base):
    def __init__(self, tokenizer=None, config=None, loader_len=0, no_train=False, device=None, spmm_config: Optional[SPMMConfig] = None,
                 options: Optional[EngineOptions] = None):
        super().__init__()
        self.options = options if options is not None else EngineOptions.from_env()
        if not torch.cuda.is_available() and not ops._DRY_RUN:
            raise RuntimeError("spmm_amd.SPMM needs an MI355X (HIP device); there is no CPU fallback")
        self.automatic_optimization = False
        self.config = config
        self.tokenizer = tokenizer
        self.training_step_outputs = []
        if device is None:
            device = "cpu" if ops._DRY_RUN else f"cuda:{torch.cuda.current_device()}"
        self.device_ = torch.device(device)
        self.cfg = spmm_config if spmm_config is not None else SPMMConfig.from_reference_dict(config)
        self.no_train = no_train
        self._grad_sync = None
        self._graphs = {}            # fused_step_graphed: shape -> "warm" | (CUDAGraph, static buffers)
        self.store = ParamStore(self.cfg, self.device_, train=not no_train)
        self.engine = PretrainStep(self.cfg, self.store, self.device_, self.options)
        self._param_names = []
        # register every tensor under the reference's state_dict name (dots are legal in _parameters/_buffers keys)
        for name, t in self.store.named_tensors():
            if is_buffer(name):
                self._buffers[name] = t
            elif self.store.kind[name] in ("tied_w", "tied_b"):
                continue                                            # added to state_dict() as aliases
            else:
                p = nn.Parameter(t, requires_grad=(student_of(name) is None) and not no_train)
                self._parameters[name] = p
                if p.requires_grad:
                    self._param_names.append(name)
        self._init_weights()
        if not no_train and config is not None:
            self.mlm_probability = config.get("mlm_probability", 0.15)
            self.warmup_steps = config["schedular"]["warmup_epochs"]
            self.loader_len = loader_len
            self.momentum = config["momentum"]
            self.queue_size = config["queue_size"]
        # sub-module facades with the reference's call signatures
        from .facade import BertFacade, LinearFacade, MaskedLMFacade, MtrHeadFacade
        object.__setattr__(self, "text_encoder", MaskedLMFacade(self, "text_encoder.", self.cfg.text))
        object.__setattr__(self, "text_encoder_m", MaskedLMFacade(self, "text_encoder_m.", self.cfg.text))
        object.__setattr__(self, "property_encoder", BertFacade(self, "property_encoder.", self.cfg.prop, False))
        object.__setattr__(self, "property_encoder_m", BertFacade(self, "property_encoder_m.", self.cfg.prop, False))
        for nm in ("property_proj", "text_proj", "itm_head", "property_embed", "property_proj_m", "text_proj_m"):
            object.__setattr__(self, nm, LinearFacade(self, nm))
        object.__setattr__(self, "property_mtr_head", MtrHeadFacade(self))
        if not _HAS_PL:                     # read-only trainer-backed properties under Lightning
            self.current_epoch = 0
            self.global_rank = 0
        self.global_step_ = 0
        self._optimizer = None
        self._scheduler = None
        self.logged = {}

    # ---- init (xbert.py:742-752, SPMM_models.py:31-44, :62, :66, :72-77) ------------------------------------------------
    @torch.no_grad()
    def _init_weights(self):
        st, cfg = self.store, self.cfg
        g = torch.Generator(device="cpu").manual_seed(torch.initial_seed() % (2 ** 31))
        for name, shape, kind in st.spec:
            if student_of(name) is not None or is_buffer(name) or kind in ("tied_w", "tied_b"):
                continue
            t = st.w(name)
            in_bert = ("encoder." in name) or (".bert." in name) or (".cls." in name)
            if kind in ("emb",) or (kind == "lin_w" and in_bert):
                t.copy_(torch.randn(shape, generator=g) * cfg.text.initializer_range)
            elif kind == "lin_w":
                t.copy_((torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(shape[-1]))
            elif kind == "lin_b":
                if in_bert:
                    t.zero_()
                else:
                    fan_in = {"property_embed.bias": 1, "itm_head.bias": 2 * cfg.text.hidden_size}.get(name, cfg.text.hidden_size)
                    t.copy_((torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in))
            elif kind == "ln_w":
                t.fill_(1.0)
            elif kind in ("ln_b", "zero"):
                t.zero_()
            elif kind == "temp":
                t.fill_(cfg.temp)
        for nm in ("prop_queue", "text_queue"):
            q = torch.randn(st.shape[nm], generator=g)
            st.buffers[nm].copy_(torch.nn.functional.normalize(q, dim=0))
        st.refresh_shadows()
        st.copy_params()
        self.engine.invalidate_banks()

    # ---- state_dict with the reference's 758 keys ------------------------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = OrderedDict() if destination is None else destination
        for name, t in self.store.named_tensors():
            out[prefix + name] = t if keep_vars else t.detach()
        return out

    def load_state_dict(self, state_dict, strict: bool = True):
        # an off-path shadow rebuild of the last optimiser step may still be running on the weight-gradient stream: the shadows written
        # below (current stream) must land AFTER it
        self.engine.pre_backward_wait()
        missing, unexpected = self.store.load_state_dict(state_dict, strict=strict)
        self.engine.refresh_padded_shadows()
        self.engine.invalidate_banks()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    @property
    def temp(self):
        return self._parameters["temp"]

    if not _HAS_PL:
        @property
        def device(self):
            return self.device_

        @property
        def global_step(self):
            return self.global_step_

        @global_step.setter
        def global_step(self, v):
            self.global_step_ = int(v)

        def log(self, name, value, prog_bar=False, **kw):        # self.log(...) of SPMM_models.py:365-370
            self.logged[name] = value

        def manual_backward(self, loss):                         # SPMM_models.py:360
            loss.backward()

    def lr_scheduler_step(self, scheduler, *args):               # SPMM_models.py:345: manual stepping only
        pass

    def on_train_epoch_end(self):
        """SPMM_models.py:382-386: mean of the last (up to) 1000 steps' losses, printed by rank 0."""
        if self.training_step_outputs:
            tmp = torch.stack(self.training_step_outputs[-1000:]).float().mean(dim=0).tolist()
            if self.global_rank == 0:
                print(f"\n mean loss: {tmp[0]:.4f}, {tmp[1]:.4f}, {tmp[2]:.4f}, {tmp[3]:.4f}")
        self.training_step_outputs.clear()

    # ---- checkpoints in the reference's Lightning layout (SPMM_pretrain.py:24-37; consumers d_pv2smiles_batched.py:133-146,
    #      d_regression.py:153-162, SPMM_models_rxn.py:16-27) ----------------------------------------------------------------
    def save_checkpoint(self, path: str, **extra):
        """Lightning-layout dict: 'state_dict' (what every reference consumer reads) plus what a resumed run needs to continue
        bit-for-bit -- Adam moments and step count, the schedule position / lr, the dropout seed (Lightning's ckpt_path resume,
        SPMM_pretrain.py:37, restores optimizer and scheduler state too)."""
        sd = {k: v.detach().cpu().clone() for k, v in self.state_dict().items()}
        # the dropout / sampling counter is saved WITHOUT the rank's offset (engine.py): every rank re-applies its own on load
        ck = dict(state_dict=sd, epoch=int(self.current_epoch), global_step=int(self.global_step_),
                  rng_seed=(int(self.engine.seed.item()) - self.engine.seed_rank_offset) % (1 << 62))
        if self._optimizer is not None:
            o = self._optimizer
            ck["optimizer_states"] = [dict(adam_m=self.store.adam_m.detach().cpu().clone(), adam_v=self.store.adam_v.detach().cpu().clone(),
                                           step_count=int(o.step_count.item()), param_groups=[dict(g) for g in o.param_groups])]
        ck.update(extra)
        torch.save(ck, path)

    def load_checkpoint(self, path_or_dict, strict: bool = False, weights_only: bool = False):
        """Accepts the Lightning dict ('state_dict'), the legacy 'model' key, or a bare state_dict; applies the legacy
        `_unk -> _mask` rename and drops nothing the arena knows.  strict=False mirrors SPMM_pretrain.py:26.  Optimizer state,
        epoch / global step and the dropout seed are restored when the checkpoint carries them -- unless `weights_only` (the
        reference's `--checkpoint` warm start, SPMM_pretrain.py:24-26: a fresh optimiser, schedule and step count on old weights;
        only `trainer.fit(ckpt_path=)` :37 continues a run)."""
        ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, str) else path_or_dict
        sd = ck.get("state_dict", ck.get("model", ck))
        sd = {k.replace("_unk", "_mask"): v for k, v in sd.items()}
        res = self.load_state_dict(sd, strict=strict)
        if isinstance(ck, dict) and not weights_only:
            if "rng_seed" in ck:
                self.engine.seed.fill_((int(ck["rng_seed"]) + self.engine.seed_rank_offset) % (1 << 62))
            if "epoch" in ck and isinstance(ck["epoch"], int) and not _HAS_PL:
                self.current_epoch = ck["epoch"]
            if "global_step" in ck:
                self.global_step_ = int(ck["global_step"])
            st = ck.get("optimizer_states")
            if st and not self.no_train and "adam_m" in st[0]:
                o = self.optimizers()
                self.store.adam_m.copy_(st[0]["adam_m"])
                self.store.adam_v.copy_(st[0]["adam_v"])
                o.step_count.fill_(int(st[0]["step_count"]))
                for g, saved in zip(o.param_groups, st[0]["param_groups"]):
                    g.update(saved)
        return res

    @property
    def queue_ptr(self):
        return self.store.buffers["queue_ptr"]

    # ---- forward / training_step -------------------------------------------------------------------------------------------
    def forward(self, property_original, text_input_ids, text_attention_mask, alpha=0, *, mpm_mask=None, neg_idx=None, aux=None):
        """SPMM.forward SPMM_models.py:79-256.  Extra keyword-only arguments inject the reference's recorded random draws."""
        eng = self.engine
        eng.train_mode = self.training
        eng.alpha.fill_(float(alpha))
        kw = dict(mpm_mask=mpm_mask, neg_idx=neg_idx, aux=aux, gather=self._gather_fn())
        dev = self.device_
        args = (property_original.to(dev), text_input_ids.to(dev), text_attention_mask.to(dev))
        if torch.is_grad_enabled() and not self.no_train:
            params = [self._parameters[n] for n in self._param_names]
            losses = _StepFn.apply(self, *args, kw, *params)
        else:
            kw["save"] = False
            l = eng.forward(*args, **kw)
            losses = tuple(l[i].clone() for i in range(4))
        return losses

    def _gather_fn(self):
        d = torch.distributed
        if d.is_available() and d.is_initialized() and (d.get_world_size() > 1 or self.options.force_dist):
            from .parallel import all_gather_features
            return all_gather_features
        return None

    def configure_optimizers(self):
        """SPMM_models.py:338-343."""
        opt = _FusedAdamW(self.store, self.engine, lr=self.config["optimizer"]["lr"], weight_decay=self.config["optimizer"]["weight_decay"])
        sch = _CosineSchedule(self.config["schedular"])
        opt.param_groups[0]["lr"] = sch.lr_at(0)          # Scheduler.__init__ installs the warm-up start value
        self._optimizer, self._scheduler = opt, sch
        return [opt], [sch]

    def optimizers(self):
        if self._optimizer is None:
            self.configure_optimizers()
        return self._optimizer

    def lr_schedulers(self):
        if self._scheduler is None:
            self.configure_optimizers()
        return self._scheduler

    def fused_step(self, prop, ids, mask, alpha, *, mpm_mask=None, neg_idx=None, grad_sync=None, n_tokens=None):
        """zero_grad -> forward -> backward (unit loss weights, SPMM_models.py:358) -> [grad all-reduce] -> clip -> AdamW.
        `n_tokens` (optional, host int from the data pipeline: sum of the attention mask, every row a non-empty prefix) sizes the
        packed GEMMs without the one blocking device read the step otherwise needs.  Returns the device tensor of the four losses."""
        eng, opt = self.engine, self.optimizers()
        eng.train_mode = self.training
        check = self._schedule_check_begin(grad_sync)
        eng.alpha.fill_(float(alpha))
        eng.gscale.fill_(1.0)
        # (single rank only: beside RCCL's stream the weight-gradient stream shares a hardware slot with the caller's stream -- queue index
        #  4 = 0 mod 4 -- and two streams that wait for each other there run the step at 74-76 ms instead of 57, profiles/r06_dp_one_rank.txt)
        eng._off_path_ok = grad_sync is None or (self.options.dp_four_streams and getattr(grad_sync, "exclusive", True) is False)
        eng.off_path(lambda: ops.zero_(self.store.grad))        # nothing reads or writes a gradient before the backward
        dev = self.device_
        losses = eng.forward(prop.to(dev), ids.to(dev), mask.to(dev), mpm_mask=mpm_mask, neg_idx=neg_idx, gather=self._gather_fn(),
                             n_tokens=n_tokens)
        if hasattr(grad_sync, "layer_done"):             # overlapped: slices are reduced as their layers finish backward
            grad_sync.begin(self.store.grad)
            eng.layer_done_cb = grad_sync.layer_done
            # The exchange's kernels on RCCL's stream are a third chip-filling stream next to the backward chain and the asynchronous
            # weight gradients: measured with a one-rank RCCL group (tools/dist_variants.sh) that combination runs the step in 75-78 ms
            # against 62 with the weight gradients back on the backward's own stream (59 without any exchange) -- the same collapse
            # as two weight-gradient streams.  So the weight-gradient stream rests while slices are exchanged.
            wg_async = eng.wgrad_async
            eng.wgrad_async = wg_async and not getattr(grad_sync, "exclusive", False)
            try:
                # a collective kernel is resident on RCCL's stream for much of this backward: the persistent NT GEMM, which counts on
                # one workgroup per CU, loses 30-55 % beside one, the per-tile launch 3-8 % (tools/gemm_bench contend)
                with ops.nt_tiles_per_workgroup(getattr(grad_sync, "tiles_under_comm", False)):
                    eng.backward()
            finally:
                eng.layer_done_cb = None
                eng.wgrad_async = wg_async
            grad_sync.finish()
        else:
            eng.backward()
            if grad_sync is not None:
                grad_sync(self.store.grad)
        opt.step()
        if check is not None:
            check.record(torch.cuda.current_stream())
        return losses

    # Data-parallel runs only (EngineOptions.schedule_check): the first SCHEDULE_CHECK_STEPS steps THAT CARRY A GRADIENT EXCHANGE try the
    # schedules this package cannot choose between without the node it runs on -- 4 steps on the side streams and 4 on one stream to warm up
    # (allocator pools grow while batches of new packed sizes arrive), then timed steps ROTATING between the candidates (four samples each,
    # none always measured on the warmer clocks / pools): "three streams" (side streams, NT GEMMs one workgroup per tile under the exchange),
    # "one stream", and -- with EngineOptions.nt_under_comm = "auto" -- "three streams, persistent NT" (faster by 1.3 ms per step when the
    # collectives' kernels are short, slower when ring kernels hold CUs for milliseconds: DESIGN.md 6).  The fastest of the two multi-stream
    # forms runs from then on (the persistent one only if its median wins by >= 1.5 %), the single stream only if ITS median wins by 13 % or
    # more (with some orders of stream creation next to RCCL's stream the multi-stream step ran at 80 ms instead of 60, EXPERIMENTS.md 1.4;
    # this package has never run next to a multi-rank communicator).  All candidates give the same results bit for bit; the events are read
    # once, at the decision, which is logged with the medians (streams.log()).
    _SCHED_WARM = (4, 4)              # untimed warm-up steps: three streams, then one stream
    _SCHED_SAMPLES = 4                # timed samples per candidate
    SCHEDULE_CHECK_STEPS = 8 + 3 * 4  # (two candidates when nt_under_comm is not "auto": the last third of the steps then re-times "three streams")

    def _schedule_check_begin(self, grad_sync):
        eng = self.engine
        if grad_sync is None:         # a warm-up / evaluation-style step without an exchange neither counts nor latches the decision
            return None
        st = getattr(self, "_sched", None)
        if st is None:
            on = (eng.opt.schedule_check and eng.multi_stream and self.device_.type == "cuda" and not getattr(ops, "_DRY_RUN", False))
            cands = ["three streams", "one stream"] + (["three streams, persistent NT"] if getattr(grad_sync, "nt_auto", False) else ["three streams"])
            st = self._sched = {"n": 0, "on": on, "ev": [], "cands": cands, "tiles0": getattr(grad_sync, "tiles_under_comm", True)}
        if not st["on"]:
            return None
        n = st["n"]
        w3, w1 = self._SCHED_WARM
        if n >= self.SCHEDULE_CHECK_STEPS:
            ms = {}
            for e0, e1, which in st["ev"]:
                e1.synchronize()
                ms.setdefault(which, []).append(e0.elapsed_time(e1))
            med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
            names = sorted(med)
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                # the step ends when the slowest rank does: every rank decides on the slowest rank's medians (and so decides the same)
                t = torch.tensor([med[k] for k in names], dtype=torch.float32, device=self.device_)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                med = {k: float(t[i]) for i, k in enumerate(names)}
            multi, single = med["three streams"], med["one stream"]
            pers = med.get("three streams, persistent NT")
            keep_persistent = pers is not None and pers <= 0.985 * multi
            best_multi = pers if keep_persistent else multi
            keep_single = single <= 0.87 * best_multi
            eng.force_one_stream = keep_single
            if hasattr(grad_sync, "tiles_under_comm"):
                grad_sync.tiles_under_comm = st["tiles0"] and not keep_persistent
            st["on"], st["ev"] = False, []
            kept = "one stream" if keep_single else ("three streams, persistent NT" if keep_persistent else "three streams")
            st["decision"] = {"three_streams_ms": round(multi, 2), "one_stream_ms": round(single, 2),
                              "three_streams_persistent_nt_ms": None if pers is None else round(pers, 2),
                              "samples_each": min(len(v) for v in ms.values()), "kept": kept}
            from . import streams
            streams.note("schedule check: medians " + ", ".join(f"{k} {v:.1f} ms" for k, v in sorted(med.items())) + f" -> {kept} kept")
            return None
        st["n"] = n + 1
        if n < w3 + w1:
            eng.force_one_stream = n >= w3
            return None
        which = st["cands"][(n - w3 - w1) % 3]
        eng.force_one_stream = which == "one stream"
        if hasattr(grad_sync, "tiles_under_comm"):
            grad_sync.tiles_under_comm = st["tiles0"] and which != "three streams, persistent NT"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        st["ev"].append((e0, e1, which))
        return e1

    def schedule_decision(self):
        """-> the data-parallel schedule check's decision and the medians it rests on (None before it was taken / when it did not run)."""
        return (getattr(self, "_sched", None) or {}).get("decision")

    def fused_step_graphed(self, prop, ids, mask, alpha, *, mpm_mask=None, neg_idx=None):
        """The same step as ONE hipGraph launch (single rank; SPMM_models.py:348-380 has no counterpart -- this removes the ~26 ms
        of host enqueue per step).  Every step-varying scalar already lives in device memory (lr, alpha, dropout seed, Adam step
        count, queue pointer, NaN flag); the three that come from the host are written before the replay.  The captured step uses
        the DENSE text layout: the packed layout's row count changes with every batch and sizes ~120 GEMMs, a graph cannot follow
        it -- so replay trades the packing gain (~26 % of the text rows at U{L/2..L} lengths) for a near-zero host cost.
        First call of a shape: eager (lazy initialisations happen outside capture); second: capture + replay; then replay."""
        eng, opt = self.engine, self.optimizers()
        dev = self.device_
        key = (tuple(ids.shape), mpm_mask is not None, neg_idx is not None, bool(self.training))
        eng.train_mode = self.training
        eng.alpha.fill_(float(alpha))
        eng.gscale.fill_(1.0)
        eng.lr.fill_(opt.param_groups[0]["lr"])
        # An eager fused_step() before this call leaves Engine._pre_bwd pending (its optimiser's off-path shadow rebuild): waiting on an
        # event recorded outside the capture is a StreamCaptureIsolation error, so drain it here and keep the maintenance inline.
        eng._off_path_ok = False
        eng.pre_backward_wait()
        st = self._graphs.get(key)
        if st is None:                                   # first step of this shape: eager, dense layout
            pack, eng.pack_text = eng.pack_text, False
            try:
                losses = self._step_body(prop.to(dev), ids.to(dev), mask.to(dev), mpm_mask, neg_idx)
            finally:
                eng.pack_text = pack
            self._graphs[key] = "warm"
            return losses
        if st == "warm":
            static = dict(prop=prop.to(dev).clone(), ids=ids.to(dev).clone(), mask=mask.to(dev).clone(),
                          mpm=None if mpm_mask is None else mpm_mask.to(dev).clone(),
                          neg=None if neg_idx is None else tuple(t.to(dev).clone() for t in neg_idx))
            pack, eng.pack_text = eng.pack_text, False
            graph = torch.cuda.CUDAGraph()
            try:
                torch.cuda.synchronize()
                with torch.cuda.graph(graph):
                    static["losses"] = self._step_body(static["prop"], static["ids"], static["mask"], static["mpm"], static["neg"])
            finally:
                eng.pack_text = pack
            st = self._graphs[key] = (graph, static)
        graph, static = st
        static["prop"].copy_(prop, non_blocking=True)
        static["ids"].copy_(ids, non_blocking=True)
        static["mask"].copy_(mask, non_blocking=True)
        if static["mpm"] is not None:
            static["mpm"].copy_(mpm_mask, non_blocking=True)
        if static["neg"] is not None:
            for d, t in zip(static["neg"], neg_idx):
                d.copy_(t, non_blocking=True)
        graph.replay()
        return static["losses"]

    def _step_body(self, prop, ids, mask, mpm_mask, neg_idx):
        """zero_grad -> forward -> backward -> clip -> AdamW on device-resident scalars only (what a hipGraph can hold)."""
        eng, opt = self.engine, self.optimizers()
        ops.zero_(self.store.grad)
        losses = eng.forward(prop, ids, mask, mpm_mask=mpm_mask, neg_idx=neg_idx, gather=None)
        eng.backward()
        opt.step(fill_lr=False)
        return losses

    def training_step(self, train_batch, batch_idx):
        """SPMM_models.py:348-380 (tokenisation :353 is the caller's job when `text` is already a tensor pair)."""
        prop, text = train_batch[0], train_batch[1]
        draws = train_batch[2] if len(train_batch) > 2 else {}
        if isinstance(text, (tuple, list)) and torch.is_tensor(text[0]):
            ids, mask = text
        else:
            ti = self.tokenizer(text, padding="longest", truncation=True, max_length=100, return_tensors="pt")
            ids, mask = ti.input_ids[:, 1:], ti.attention_mask[:, 1:]
        alpha = self.config["alpha"] if self.current_epoch > 0 else self.config["alpha"] * min(1., batch_idx / self.loader_len)
        opt, sch = self.optimizers(), self.lr_schedulers()
        from .parallel import grad_sync_fn
        if self._grad_sync is None:
            self._grad_sync = grad_sync_fn(self.store, self.options) or False
        n_tokens = draws.get("n_tokens")
        if torch.is_tensor(mask) and mask.device.type == "cpu":       # a host mask (the tokenizer's): count / verify it here, for free
            lens = mask.sum(1)
            prefix = bool((lens > 0).all()) and bool(((torch.arange(mask.shape[1])[None, :] < lens[:, None]) == (mask != 0)).all())
            if n_tokens is not None and (not prefix or int(n_tokens) != int(lens.sum())):
                raise ValueError(f"n_tokens={n_tokens} contradicts the attention mask (sum {int(lens.sum())}, prefix rows: {prefix})")
            if n_tokens is None and prefix:
                n_tokens = int(lens.sum())
        losses = self.fused_step(prop, ids, mask, alpha, grad_sync=self._grad_sync or None, mpm_mask=draws.get("mpm_mask"),
                                 neg_idx=draws.get("neg_idx"), n_tokens=n_tokens)
        if self.global_rank == 0:
            self.logged = {"lr": opt.param_groups[0]["lr"], "losses": losses}      # device tensor: no host read per step
        step_size, warm = 100, self.warmup_steps
        if self.current_epoch > 0 and batch_idx == 0:
            opt.param_groups[0]["lr"] = sch.lr_at(self.current_epoch + warm)
        elif self.current_epoch == 0 and batch_idx % step_size == 0 and batch_idx <= warm * step_size:
            opt.param_groups[0]["lr"] = sch.lr_at(batch_idx // step_size)
        out = losses[:4].detach().clone()
        self.training_step_outputs.append(out)
        if len(self.training_step_outputs) > 1000:         # only the last 1000 feed on_train_epoch_end (:383)
            del self.training_step_outputs[:-1000]
        self.global_step_ += 1
        return out
