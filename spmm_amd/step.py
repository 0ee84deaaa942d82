"""One SPMM pretraining step on the engine: SPMM.forward (SPMM_models.py:79-256) and its backward."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import ops
from .engine import STREAM_TOKENS_MAX, BF, LOSS_ITA, LOSS_ITM, LOSS_MLM, LOSS_MPM, Engine, Group, KVSource, SelfKV, _ceil


class PretrainStep(Engine):
    def _pack_plan(self, mask32: torch.Tensor, B: int, Lt: int, n_tokens: Optional[int] = None):
        """Row bookkeeping for the packed text passes: valid rows of the dense [B*Lt] layout in order, per-sequence start
        and length.  The packed row count sizes the GEMMs, so the host must know it: either the data pipeline says so
        (`n_tokens`: the tokenizer's attention mask is a host tensor, its sum costs nothing there -- and the caller then vouches
        that every mask row is a non-empty prefix, which is what padding='longest' produces) or it is read back from the device,
        one blocking read per step.  Returns None -- dense fallback -- when a sequence does not start with a valid token
        (position 0 is what the losses read) or nothing would be saved."""
        if n_tokens is not None:
            M = int(n_tokens)
            if M >= B * Lt:
                return None
            # The caller vouches for the hint (SPMM.training_step derives it from the tokenizer's host mask itself).  Best effort
            # against a wrong one, without a read-back: the mismatch is detected on the device (spmm_pack_plan raises `hint_bad`) and
            # raises the NaN flag (AdamW, EMA and enqueue become no-ops, as for a non-finite loss, SPMM_models.py:132-134), and the
            # per-sequence bookkeeping is clamped to the rows the hint sized; launches sized from other derived quantities may still misbehave.
        else:
            lens = mask32.sum(1)
            prefix = (torch.arange(Lt, device=mask32.device)[None, :] < lens[:, None]) == (mask32 != 0)
            stats = torch.stack([lens.sum(), (lens > 0).sum(), prefix.all().to(lens.dtype)]).cpu()
            M, nonempty, is_prefix = int(stats[0]), int(stats[1]), int(stats[2])
            if nonempty != B or not is_prefix or M >= B * Lt:   # holes in the mask: the packed index would not be the position
                return None
        if M < 1:
            return None
        # valid rows first, original order kept; whatever the hint was, no index leaves the M rows it sized (csrc/plan.hip)
        return ops.pack_plan(mask32, M, self.hint_bad)

    # ------------------------------------------------------------------------------- fusion batch, packed path
    def _s6_forward_cls(self, B, Lt, M, pk, mask32, neg, y1, y2, prop_embeds, text_embeds, save):
        """The seven student fusion passes (:137-198, :224-231, :243-250) as one batch, with the TOP fusion layer reduced to the rows a
        loss reads.  The three ITM pass pairs feed only position 0 of their last hidden states to the ITM head (:199-201), so in the last
        layer their other rows matter only as self-attention keys / values: that layer runs on [6B position-0 rows | every row of the LM
        pass | every row of the causal PV pass], the position-0 queries attending keys / values projected from the full sequences
        (engine.SelfKV).  Exact: the rows left out reach no loss, and their gradients are exactly zero in the reference too.
        Batch layout (csrc/plan.hip): [PV queries pe | pe[neg] | pe] [text packed te | te] [LM pass] [causal PV] [text negatives, PACKED].
        Which sequences were drawn as text negatives is device data, so the LENGTH of that last part -- and with it the batch's row count --
        is known only on the device: the batch is allocated for B x Lt such rows and every launch over it reads the real row count from
        device memory (`Engine._dyn`, "device-side row counts" in include/spmm_hip.h): no padding row is computed, no host read sizes
        anything.  Every index array comes from ONE launch (spmm_fusion_plan); the batch and the top layer's input are two row gathers."""
        cfg = self.cfg
        ct = cfg.text
        H, Lp, f, n = ct.hidden_size, cfg.n_props + 1, ct.fusion_layer, ct.num_hidden_layers
        fp = ops.fusion_plan(neg, pk, Lp)
        o_tp = 3 * B * Lp
        o_lm = o_tp + 2 * M
        o_12 = o_lm + B * Lt
        o_8 = o_12 + B * Lp
        Rcap = fp["Rcap"]
        X6 = ops.gather_rows2(self._new(Rcap, H), y1, fp["idx6"], y2)
        src_pv = KVSource(prop_embeds.view(B * Lp, H), B, Lp).preset(4 * B, fp["start_p"], fp["list_p"])
        src_text = KVSource(text_embeds, B, Lt, row0=pk["row0"], length=pk["len"], pack_idx=pk["rows"]).preset(4 * B, fp["start_t"], fp["list_t"])
        ar = fp["ar"]
        g_lo = [Group(0, 3 * B, Lp, None, 3 * B).bind(src_text, fp["kvidx_pv"], 0),
                Group(o_tp, 2 * B, Lt, None, 2 * B, q_row0=fp["qrow0_tp"], q_len=fp["qlen_tp"], nrows=2 * M).bind(src_pv, fp["kvidx_tp"], 0),
                Group(o_lm, B, Lt, mask32, 0).bind(src_pv, ar, 2 * B),
                Group(o_12, B, Lp, None, 0).bind(src_text, ar, 3 * B),
                Group(o_8, B, Lt, None, B, q_row0=fp["row0_8"], q_len=fp["len_8"], nrows=B * Lt).bind(src_pv, ar, 3 * B)]
        self._dyn = (Rcap, fp["rows_dev"])
        try:
            y, tape_lo = self.stack_fwd("text_encoder.bert.", ct, range(f, n - 1), True, X6, g_lo, save)
        finally:
            self._dyn = None
        ntop = 6 * B + B * Lt + B * Lp
        Xtop = ops.gather_rows2(self._new(ntop, H), y, fp["idx_top"])
        skv_a = SelfKV(y[:o_lm])                                # rows of the PV and packed-text ITM sequences
        skv_b = SelfKV(y[o_8:], rows_dev=fp["mn_dev"])          # rows of the packed text negatives (device-side count)
        g_top = [Group(0, 3 * B, 1, None, 3 * B, self_src=skv_a, skv_row0=fp["skv_row0_pv"], skv_len=fp["skv_len_pv"], skv_L=Lp).bind(src_text, fp["kvidx_pv"], 0),
                 Group(3 * B, 2 * B, 1, None, 2 * B, self_src=skv_a, skv_row0=fp["skv_row0_tx"], skv_len=fp["skv_len_tx"], skv_L=Lt).bind(src_pv, fp["kvidx_tp"], 0),
                 Group(5 * B, B, 1, None, B, self_src=skv_b, skv_row0=fp["row0_8"], skv_len=fp["len_8"], skv_L=Lt).bind(src_pv, ar, 3 * B),
                 Group(6 * B, B, Lt, mask32, 0).bind(src_pv, ar, 2 * B),
                 Group(6 * B + B * Lt, B, Lp, None, 0).bind(src_text, ar, 3 * B)]
        ytop, sv_top, _ = self._layer_fwd(f"text_encoder.bert.encoder.layer.{n - 1}.", ct, True, Xtop, g_top, save)
        return dict(fp=fp, ytop=ytop, sv_top=sv_top, g_top=g_top, skv=(skv_a, skv_b), g_lo=g_lo, tape_lo=tape_lo, src_text=src_text, src_pv=src_pv,
                    o_tp=o_tp, o_lm=o_lm, o_12=o_12, o_8=o_8, ntop=ntop, Rcap=Rcap)

    def _s6_backward_cls(self, T, dYtop, d_pe, d_te):
        """Backward of `_s6_forward_cls`: dYtop = d(loss)/d(top layer output) on its [6B | B Lt | B Lp] rows -> d(loss)/d(batch rows)."""
        S6, B, Lt, M = T["S6"], T["B"], T["Lt"], T["M"]
        cfg = self.cfg
        ct = cfg.text
        H, Lp, f, n = ct.hidden_size, cfg.n_props + 1, ct.fusion_layer, ct.num_hidden_layers
        fp, o_lm, o_8 = S6["fp"], S6["o_lm"], S6["o_8"]
        dkv_acc = {id(S6["src_text"]): d_te[:M], id(S6["src_pv"]): d_pe.view(B * Lp, H)}
        dX6 = self._new(S6["Rcap"], H)
        S6["skv"][0].dx = dX6[:o_lm]                             # the ITM sequences' rows: gradient through the top layer's keys / values ...
        S6["skv"][1].dx = dX6[o_8:]
        dXtop = self._layer_bwd(f"text_encoder.bert.encoder.layer.{n - 1}.", ct, S6["sv_top"], dYtop, S6["g_top"], dkv_acc)
        if self.layer_done_cb is not None:
            self._layer_done(f"text_encoder.bert.encoder.layer.{n - 1}.")
        ops.add_rows_bf16(dX6, fp["idx_top"][:6 * B], dXtop[:6 * B])     # ... plus, at position 0, what came through the queries
        dX6[o_lm:o_8].copy_(dXtop[6 * B:])
        self._dyn = (S6["Rcap"], fp["rows_dev"])
        try:
            return self.stack_bwd("text_encoder.bert.", ct, range(f, n - 1), S6["tape_lo"], dX6, S6["g_lo"], dkv_acc=dkv_acc)
        finally:
            self._dyn = None

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, prop: torch.Tensor, ids: torch.Tensor, mask: torch.Tensor, *, mpm_mask: Optional[torch.Tensor] = None,
                neg_idx: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, gather: Optional[Callable] = None,
                save: bool = True, aux: Optional[dict] = None, n_tokens: Optional[int] = None) -> torch.Tensor:
        """Returns the device tensor losses[0:4] = (loss_mlm, 5*loss_mpm, loss_ita, loss_itm).  `alpha` is read from
        self.alpha (device).  Mutates temp (clamp), the momentum arena (EMA), the queues and the queue pointer exactly as
        the reference's forward does."""
        cfg, P = self.cfg, self.P
        ct, cp = cfg.text, cfg.prop
        H, E, Lp, f, n = ct.hidden_size, cfg.embed_dim, cfg.n_props + 1, ct.fusion_layer, ct.num_hidden_layers
        B, Lt = ids.shape
        if B % 4 or cfg.queue_size % 4:
            raise ValueError("batch size and queue size must be multiples of 4")
        self._salt = 0
        if self.train_mode:
            self.seed.add_(1)               # new dropout masks / negative draws every step; backward re-reads the same value
        ops.zero_(self.step_zero)             # losses, d(ita)/d(temp), the non-finite flag, the token-hint flag
        temp = P.w("temp").view(1)
        ops.clamp_scalar(temp, 0.01, 0.5)                                             # :80-81
        ids32 = ids.to(torch.int32).contiguous()
        mask32 = mask.to(torch.int32).contiguous()
        prop = prop.to(torch.float32).contiguous()
        if mpm_mask is None:                                                          # :85 bernoulli(0.5)
            mpm_mask = (torch.rand(B, cfg.n_props, device=self.dev) < 0.5).to(torch.float32)
        mpm_mask = mpm_mask.to(torch.float32).contiguous()

        # ---- S1..S4: the student and momentum unimodal encoders (:90-106) batched with their causal twins (:215-224, :242).
        # The text chains (S2, S4) and the PV chains (S1, S3) share nothing until the fusion layers: two streams.
        self._one_stream = self.force_one_stream or B * Lt > STREAM_TOKENS_MAX
        pk = self._pack_plan(mask32, B, Lt, n_tokens) if (self.pack_text and aux is None and Lt <= ops.ATTN_MAXL) else None   # packed layouts: what the attention kernels hold on chip
        if n_tokens is not None:
            self.nan_flag.bitwise_or_(self.hint_bad)
        M = pk["M"] if pk else B * Lt
        ids2 = torch.cat([ids32, ids32])
        # fp32 residual stream (EngineOptions.resid_fp32): every hidden state travels as a bf16 tensor (GEMM operand) and
        # its fp32 twin (residual additions, loss-head inputs); `f32(t)` starts a twin, the *_32 names below mirror every regrouping
        r32 = self.opt.resid_fp32 and aux is None
        f32 = (lambda t: t.float()) if r32 else (lambda t: None)
        side = self._fork()
        with self._on(side):
            x2, esv2 = self.embed_text("text_encoder.bert.", ct, ids2, 2 * B, Lt, save)
            x2_32 = f32(x2)
            # P2 (and P4, P6, P8) feed only position 0 of their outputs to a loss (:95, :105, :201), and a padding token is
            # never attended as a key, so its rows influence nothing: those passes run on the packed valid rows.  The
            # student's LM pass (P10) keeps every row -- its loss counts the padding targets (:233).
            if pk:
                x2d = x2
                x2 = ops.gather_rows2(self._new(M + B * Lt, H), x2d, pk["gidx2"])
                if r32:
                    x2_32 = torch.cat([x2_32[:B * Lt].index_select(0, pk["rows"]), x2_32[B * Lt:]])
                g2 = [Group(0, B, Lt, None, B, q_row0=pk["row0"], q_len=pk["len"], nrows=M), Group(M, B, Lt, mask32, 0)]
            else:
                g2 = [Group(0, 2 * B, Lt, torch.cat([mask32, mask32]), B)]
            y2, tape2 = self.stack_fwd("text_encoder.bert.", ct, range(0, f), True, x2, g2, save, X32=x2_32)
            text_embeds, hidden10 = y2[:M], y2[M:]
            text_embeds_32, hidden10_32 = (self.last32[:M], self.last32[M:]) if r32 else (None, None)
        side_m = self._fork(1)
        ema_done = None
        with self._on(side_m):
            # EMA of the momentum parameters (:99 / :266-269) at the head of the momentum stream: the student chains do not
            # read them, so they start at once; the PV momentum pass below waits on `ema_done`
            ops.ema_update(P.flat_m, P.flat, P.shadow_m, cfg.momentum)
            P.refresh_frag(True)                 # (fragment-ordered images of the momentum cross-attention output projections)
            if side_m is not None:
                ema_done = torch.cuda.Event()
                ema_done.record(side_m)
            # momentum text branch (:104-105, :215-222), no tape
            x4, _ = self.embed_text("text_encoder_m.bert.", ct, ids2, 2 * B, Lt, False)
            x4_32 = f32(x4)
            if pk:
                # The teacher's LM logits are read only where the label is a real token (:236-237), and a causal position
                # sees nothing to its right: P9 (unlike P10) is packed too.
                x4d = x4
                x4 = ops.gather_rows2(self._new(2 * M, H), x4d, pk["gidx4"])
                if r32:
                    x4_32 = torch.cat([x4_32[:B * Lt].index_select(0, pk["rows"]), x4_32[B * Lt:].index_select(0, pk["rows"])])
                g4 = [g2[0], Group(M, B, Lt, None, 0, q_row0=pk["row0"], q_len=pk["len"], nrows=M)]
            else:
                g4 = g2
            cls_m = bool(pk) and self.opt.cls_only_top and not r32 and f >= 1
            if cls_m:
                # text_embeds_m feeds only position 0 to a loss (text_feat_m, :105): the momentum text encoder's LAST unimodal layer runs on
                # [position 0 of the B packed sequences | every row of the causal copy], the position-0 queries attending keys / values
                # projected from the full sequences (engine.SelfKV; the student's counterpart is a key / value source of the fusion layers)
                y4, _ = self.stack_fwd("text_encoder_m.bert.", ct, range(0, f - 1), True, x4, g4, False)
                x4t = ops.gather_rows2(self._new(B + M, H), y4, pk["idx_m"])
                g4t = [Group(0, B, 1, None, B, self_src=SelfKV(y4[:M]), skv_row0=pk["row0"], skv_len=pk["len"], skv_L=Lt),
                       Group(B, B, Lt, None, 0, q_row0=pk["row0"], q_len=pk["len"], nrows=M)]
                y4t, _, _ = self._layer_fwd(f"text_encoder_m.bert.encoder.layer.{f - 1}.", ct, False, x4t, g4t, False)
                text_embeds_m, hidden9 = y4t[:B], y4t[B:]          # (position-0 rows only)
            else:
                y4, _ = self.stack_fwd("text_encoder_m.bert.", ct, range(0, f), True, x4, g4, False, X32=x4_32)
                text_embeds_m, hidden9 = y4[:M], y4[M:]
            text_embeds_m_32, hidden9_32 = (self.last32[:M], self.last32[M:]) if r32 else (None, None)
        x1, esv1 = self.embed_pv("property_encoder.", cp, prop, mpm_mask, 2 * B, B, save)
        g1 = [Group(0, 2 * B, Lp, None, B)]
        y1, tape1 = self.stack_fwd("property_encoder.", cp, range(cp.num_hidden_layers), False, x1, g1, save, X32=f32(x1))
        prop_embeds, prop_embeds_causal = y1[:B * Lp], y1[B * Lp:]
        prop_embeds_32, prop_embeds_causal_32 = (self.last32[:B * Lp], self.last32[B * Lp:]) if r32 else (None, None)
        if ema_done is not None:
            torch.cuda.current_stream().wait_event(ema_done)
        x3, _ = self.embed_pv("property_encoder_m.", cp, prop, mpm_mask, B, B, False)
        prop_embeds_m, _ = self.stack_fwd("property_encoder_m.", cp, range(cp.num_hidden_layers), False, x3,
                                          [Group(0, B, Lp, None, B)], False, X32=f32(x3))
        prop_embeds_m_32 = self.last32 if r32 else None
        self._join(side)
        self._join(side_m)
        if pk:
            g5 = [Group(0, B, Lt, None, 0, kv=prop_embeds_m, Lkv=Lp, kv_mask=None, q_row0=pk["row0"], q_len=pk["len"], nrows=M)]
        else:
            g5 = [Group(0, B, Lt, mask32, 0, kv=prop_embeds_m, Lkv=Lp, kv_mask=None)]
        # The teacher's fusion pass (S5: P9b + LM head, small M) is needed only by the LM loss at the very end: it runs on
        # the side stream underneath the features / ITA / S6 work below.
        side5 = self._fork()
        with self._on(side5):
            y5, _ = self.stack_fwd("text_encoder_m.bert.", ct, range(f, n), True, hidden9, g5, False, X32=hidden9_32)
            logits_m, _ = self.lm_head_fwd("text_encoder_m.", ct, y5, False)
            if pk:                                               # the loss kernel indexes [B, Lt, V]
                V_ = logits_m.shape[1]
                if V_ % 4 == 0:                                  # rows of 4 V bytes = 2 V bf16-sized elements: the row gather serves them
                    dense = torch.empty(B * Lt, V_, dtype=logits_m.dtype, device=self.dev)
                    ops.gather_rows2(dense.view(BF), logits_m.view(BF), pk["inv"])
                    logits_m = dense
                else:
                    logits_m = torch.zeros(B * Lt, V_, dtype=logits_m.dtype, device=self.dev).index_copy_(0, pk["rows"], logits_m)

        # ---- features, similarity banks, ITA loss and its gradient w.r.t. the student features (:92-131)
        bank = self._banks(B)
        J, Jp = bank["J"], bank["Jp"]
        A3 = self._new(4 * B, 3 * E)
        feats = {}
        twins = {"property_proj": prop_embeds_32, "text_proj": text_embeds_32, "property_proj_m": prop_embeds_m_32, "text_proj_m": text_embeds_m_32}
        for k, (proj, X, L, w3, qT) in enumerate((("property_proj", prop_embeds, Lp, None, None),
                                                  ("text_proj", text_embeds, Lt, None, None),
                                                  ("property_proj_m", prop_embeds_m, Lp, *bank["prop"]),
                                                  ("text_proj_m", text_embeds_m, Lt, *bank["text"]))):
            if proj == "text_proj_m" and cls_m:                  # X holds the position-0 rows already
                raw, feat, nrm, cls = self._feat_fwd(proj, X, 1, B, save, X32=None)
            else:
                raw, feat, nrm, cls = self._feat_fwd(proj, X, L, B, save, cls_rows=pk["row0_64"] if (pk and proj.startswith("text_proj")) else None,
                                                     X32=twins[proj])
            ops.l2norm_fwd(raw, feat, nrm, a3=A3[k * B:(k + 1) * B], w3=None if w3 is None else w3[:B], yT=qT)
            feats[proj] = (feat, nrm, cls)
        S_text = self._new(4 * B, J, dtype=torch.float32)     # rows: i2t | t2t | i2t_m | t2t_m
        S_prop = self._new(4 * B, J, dtype=torch.float32)     # rows: i2i | t2i | i2i_m | t2i_m
        ops.gemm_nt(A3, bank["text"][0], S_text, epi=ops.EPI_F32, div=temp)
        ops.gemm_nt(A3, bank["prop"][0], S_prop, epi=ops.EPI_F32, div=temp)
        dS_text, dS_prop = self._new(2 * B, Jp), self._new(2 * B, Jp)
        for S, dS in ((S_text, dS_text), (S_prop, dS_prop)):
            ops.ita_rows(S[:2 * B], S[2 * B:], dS, B=B, J=J, alpha=self.alpha, temp=temp, losses=self.losses, slot=LOSS_ITA,
                         dtemp=self.dtemp_ita, nan_flag=self.nan_flag)
        dfeat = None
        if save:   # d loss_ita / d [prop_feat ; text_feat], before the queue is overwritten (:208)
            dfeat = self._zeros(2 * B, E, dtype=torch.float32)
            sp = max(1, min(Jp // 64, 16))      # K = B + Q split 16 ways: 52 us at K = 36 992 (64 ways: 102 us, the fp32 atomics dominate; tools/bench_splitk.py)
            ops.gemm_nt(dS_text, bank["text"][1], dfeat, epi=ops.EPI_F32_ATOMIC, splits=sp, div=temp)
            ops.gemm_nt(dS_prop, bank["prop"][1], dfeat, epi=ops.EPI_F32_ATOMIC, splits=sp, div=temp)

        # ---- hard negatives (:154-178): prop negatives from sim_t2i rows, text negatives from sim_i2t rows
        neg = self._zeros(2 * B, dtype=torch.int64)
        fp, ft = (None, None) if neg_idx is None else (neg_idx[0].to(torch.int64).contiguous(), neg_idx[1].to(torch.int64).contiguous())
        ops.sample_neg(S_prop[B:2 * B], B, neg[:B], forced=fp, seed=self.seed, salt=self._next_salt())
        ops.sample_neg(S_text[:B], B, neg[B:], forced=ft, seed=self.seed, salt=self._next_salt())

        # ---- S6: the fusion layers over all seven student fusion passes at once (:137-198, :224-231, :243-250)
        # Layout, index arrays and the CLS-only top layer of the packed path: `_s6_forward_cls` (csrc/plan.hip); the general form below
        # serves dense masks, `aux` inspection and the fp32 residual stream.
        cls_top = bool(pk) and self.opt.cls_only_top and not r32 and aux is None
        if cls_top:
            S6 = self._s6_forward_cls(B, Lt, M, pk, mask32, neg, y1, y2, prop_embeds, text_embeds, save)
            ytop = S6["ytop"]
            ops.itm_head(ytop[:3 * B], H, ytop[3 * B:6 * B], H, H, P.w("itm_head.weight"), P.w("itm_head.bias"), n=3 * B, B=B, losses=self.losses,
                         slot=LOSS_ITM, logits=None)
            ytext = hid10_cls = ytop[6 * B:6 * B + B * Lt]
            ypv = ypv_h = ytop                                   # (tape names of the general path)
            hp12_cls = ytop[6 * B + B * Lt:]
            src_text = src_pv = neg_rows = cls_text = itm_text = g6 = tape6 = None
            Mn = B * Lt
        else:
            # Cross-attention sources: the four PV-query passes read [te, te, te[neg], te], the four text-query passes
            # [pe, pe[neg], pe, pe] -- B unique sequences each (KVSource): K/V are projected once on the unique rows and the
            # attention kernels follow the sequence -> source map.
            pe = prop_embeds.view(B, Lp * H)
            pe_neg = ops.gather_rows(self._new(B, Lp * H), pe, neg[:B])
            ar = torch.arange(B, dtype=torch.int64, device=self.dev)
            mask_neg = mask32.index_select(0, neg[B:])
            qpv = torch.cat([pe, pe_neg, pe, prop_embeds_causal.view(B, Lp * H)]).view(4 * B * Lp, H)
            if r32:
                pe32 = prop_embeds_32.view(B, Lp * H)
                qpv_32 = torch.cat([pe32, pe32.index_select(0, neg[:B]), pe32, prop_embeds_causal_32.view(B, Lp * H)]).view(4 * B * Lp, H)
            src_pv = KVSource(prop_embeds.view(B * Lp, H), B, Lp)
            if pk:
                src_text = KVSource(text_embeds, B, Lt, row0=pk["row0"], length=pk["len"], pack_idx=pk["rows"])
                neg_rows = pk["row0_64"].index_select(0, neg[B:])[:, None] + torch.arange(Lt, device=self.dev)[None, :]
                neg_rows = torch.where(mask_neg.bool(), neg_rows, torch.full_like(neg_rows, M)).view(-1)   # dense (i, l) -> row of te
                r0 = 4 * B * Lp
                # Text negatives as queries (second half of P8) stay dense [B, Lt]: which sequences were drawn is device data, and
                # packing them behind a second host read measured no gain.  Rows past the negative's length are zero-filled and
                # masked as keys.
                Mn = B * Lt
                te_neg = torch.cat([text_embeds, self._zeros(1, H)]).index_select(0, neg_rows)
                qtext = torch.cat([text_embeds, text_embeds, te_neg, hidden10])
                if r32:
                    te_neg_32 = torch.cat([text_embeds_32, self._zeros(1, H, dtype=torch.float32)]).index_select(0, neg_rows)
                    qtext_32 = torch.cat([text_embeds_32, text_embeds_32, te_neg_32, hidden10_32])
                gt = [Group(r0, 2 * B, Lt, None, 2 * B, q_row0=torch.cat([pk["row0"], pk["row0"] + M]), q_len=torch.cat([pk["len"], pk["len"]]),
                            nrows=2 * M).attend(src_pv, torch.cat([ar, neg[:B]])),
                      Group(r0 + 2 * M, 2 * B, Lt, torch.cat([mask_neg, mask32]), B).attend(src_pv, torch.cat([ar, ar]))]
                cls_text = torch.cat([pk["row0_64"], pk["row0_64"] + M, 2 * M + ar * Lt])
                kvmask_qpv = None                                   # key padding is implied by the source lengths
            else:
                src_text = KVSource(text_embeds, B, Lt)
                te = text_embeds.view(B, Lt * H)
                te_neg = ops.gather_rows(self._new(B, Lt * H), te, neg[B:])
                qtext = torch.cat([te, te, te_neg, hidden10.view(B, Lt * H)]).view(4 * B * Lt, H)
                if r32:
                    te32 = text_embeds_32.view(B, Lt * H)
                    qtext_32 = torch.cat([te32, te32, te32.index_select(0, neg[B:]), hidden10_32.view(B, Lt * H)]).view(4 * B * Lt, H)
                kvmask_qpv = torch.cat([mask32, mask32, mask_neg, mask32])
                gt = [Group(4 * B * Lp, 4 * B, Lt, kvmask_qpv, 3 * B).attend(src_pv, torch.cat([ar, neg[:B], ar, ar]))]
                neg_rows, cls_text, Mn = None, torch.arange(3 * B, dtype=torch.int64, device=self.dev) * Lt, B * Lt
            X6 = torch.cat([qpv, qtext])
            g6 = [Group(0, 4 * B, Lp, None, 3 * B, kv_mask=kvmask_qpv).attend(src_text, torch.cat([ar, ar, neg[B:], ar]))] + gt
            src_text.finalize()
            src_pv.finalize()
            y6, tape6 = self.stack_fwd("text_encoder.bert.", ct, range(f, n), True, X6, g6, save, X32=torch.cat([qpv_32, qtext_32]) if r32 else None)
            ypv, ytext = y6[:4 * B * Lp], y6[4 * B * Lp:]
            # the loss heads read the fp32 twins in that mode (forward and backward: the tape keeps what the forward read)
            ypv_h, ytext_h = (self.last32[:4 * B * Lp], self.last32[4 * B * Lp:]) if r32 else (ypv, ytext)

            # ---- ITM head (:199-206) on the position-0 rows of the first 3B sequences of both halves
            vl_logits = self._new(3 * B, 2, dtype=torch.float32) if aux is not None else None
            itm_text = ytext_h.index_select(0, cls_text)
            ops.itm_head(ypv_h, Lp * H, itm_text, H, H, P.w("itm_head.weight"), P.w("itm_head.bias"), n=3 * B, B=B, losses=self.losses,
                         slot=LOSS_ITM, logits=vl_logits)

        # ---- queue (:208, :272-286)
        feat_pm, feat_tm = feats["property_proj_m"][0], feats["text_proj_m"][0]
        if gather is not None:
            # ONE all-gather per step (SPMM_models.py:273-274 issues two): [prop_feat_m | text_feat_m | NaN flag].  The flag column
            # makes the NaN decision (skip enqueue + optimiser step) identical on every rank, so the replicas cannot diverge.
            g = gather(torch.cat([feat_pm, feat_tm, self.nan_flag.to(torch.float32).expand(B, 1)], 1))
            feat_pm, feat_tm = g[:, :E].contiguous(), g[:, E:2 * E].contiguous()
            self.nan_flag.copy_((g[:, 2 * E].abs().sum() > 0).to(torch.int32).view(1))
        ptr = P.buffers["queue_ptr"]
        # NaN step: the reference returns at :132-134, before the enqueue at :208 -- queue and pointer stay untouched
        ops.enqueue(feat_pm, P.buffers["prop_queue"], *bank["prop"], ptr, Bloc=B, advance=False, skip_flag=self.nan_flag)
        ops.enqueue(feat_tm, P.buffers["text_queue"], *bank["text"], ptr, Bloc=B, advance=True, skip_flag=self.nan_flag)

        # ---- LM loss (:211-238)
        self._join(side5)
        hid10 = hid10_cls if cls_top else ytext[ytext.shape[0] - B * Lt:]
        logits, lmsv = self.lm_head_fwd("text_encoder.", ct, hid10, save)
        ops.lm_loss(logits, logits_m, ids32, nseq=B, L=Lt, V=ct.vocab_size, alpha=self.alpha, ws=self.icount[0:1], losses=self.losses,
                    slot=LOSS_MLM)

        # ---- MPM (:241-256)
        hp12 = hp12_cls if cls_top else ypv[3 * B * Lp:]
        mt, mpre = self._new(B * Lp, H), self._new(B * Lp, H)
        ops.gemm_nt(hp12, P.wb("property_mtr_head.0.weight"), mt, bias=P.w("property_mtr_head.0.bias"), epi=ops.EPI_GELU, C2=mpre)
        mln = self._new(B * Lp, H)
        mmean, mrstd = self._new(B * Lp, dtype=torch.float32), self._new(B * Lp, dtype=torch.float32)
        if r32:
            mln_h = self._new(B * Lp, H, dtype=torch.float32)
            ops.ln_fwd_r32(mt, None, P.w("property_mtr_head.2.weight"), P.w("property_mtr_head.2.bias"), mln, y32=mln_h, zout=mt, mean=mmean,
                           rstd=mrstd, eps=ct.layer_norm_eps)
        else:
            mln_h = mln
            ops.ln_fwd(mt, None, P.w("property_mtr_head.2.weight"), P.w("property_mtr_head.2.bias"), mln, zout=mt, mean=mmean, rstd=mrstd,
                       eps=ct.layer_norm_eps)
        pred = self._new(B, cfg.n_props, dtype=torch.float32) if aux is not None else None
        ops.mpm_head(mln_h, Lp, H, P.w("property_mtr_head.3.weight"), P.w("property_mtr_head.3.bias"), prop, mpm_mask, B=B,
                     ws=self.icount[1:2], losses=self.losses, slot=LOSS_MPM, pred=pred)

        if aux is not None:
            aux.update(prop_embeds=prop_embeds, text_embeds=text_embeds, prop_embeds_m=prop_embeds_m, text_embeds_m=text_embeds_m,
                       prop_feat=feats["property_proj"][0], text_feat=feats["text_proj"][0], prop_feat_m=feats["property_proj_m"][0],
                       text_feat_m=feats["text_proj_m"][0], sim_i2t=S_text[:B], sim_t2i=S_prop[B:2 * B], sim_i2t_m=S_text[2 * B:3 * B],
                       vl_output=vl_logits, mlm_output=logits.view(B, Lt, -1)[:, :-1], logits_m=logits_m.view(B, Lt, -1)[:, :-1],
                       pred=pred, prop_neg_idx=neg[:B], text_neg_idx=neg[B:], mpm_mask=mpm_mask,
                       prop_embeds_causal=prop_embeds_causal)
        if save:
            self.tape = dict(B=B, Lt=Lt, pk=pk, M=M, Mn=Mn, src_text=src_text, src_pv=src_pv, neg_rows=neg_rows, cls_text=cls_text, itm_text=itm_text,
                             prop=prop, mpm_mask=mpm_mask, ids32=ids32, ids2=ids2, esv1=esv1, g1=g1, tape1=tape1,
                             esv2=esv2, g2=g2, tape2=tape2, feats=feats, dfeat=dfeat, neg=neg, g6=g6, tape6=tape6, ypv=ypv_h,
                             ytext=ytext, logits=logits, logits_m=logits_m, lmsv=lmsv, hp12=hp12, mpre=mpre, mt=mt, mln=mln_h,
                             mmean=mmean, mrstd=mrstd, S6=S6 if cls_top else None)
        return self.losses[:4]

    # ----------------------------------------------------------------------------------------------- backward
    def backward(self):
        """Accumulates d(sum_k gscale[k] * loss_k)/d(param) into the flat gradient arena (self.P.grad)."""
        T = self.tape
        if T is None:
            raise RuntimeError("backward() without a taped forward()")
        self.pre_backward_wait()                                   # gradient arena zeroed, transposed weight shadows rebuilt (Engine.off_path)
        cfg, P = self.cfg, self.P
        ct, cp = cfg.text, cfg.prop
        H, E, Lp, f, n = ct.hidden_size, cfg.embed_dim, cfg.n_props + 1, ct.fusion_layer, ct.num_hidden_layers
        B, Lt = T["B"], T["Lt"]
        gs = self.gscale
        scratch = self.loss_scratch
        pk, M = T["pk"], T["M"]
        S6 = T["S6"]                                               # packed path with the CLS-only top layer (_s6_forward_cls), else None
        if S6 is not None:
            # gradient w.r.t. the top layer's output on its [6B position-0 rows | LM pass | causal PV pass] rows: every row is written below
            dYtop = self._new(S6["ntop"], H)
            dY_mpm, dY_lm = dYtop[6 * B + B * Lt:], dYtop[6 * B:6 * B + B * Lt]
        else:
            nt6 = T["ytext"].shape[0]                              # text-query rows of S6: 2M + 2B*Lt packed, 4B*Lt dense
            dY6 = self._zeros(4 * B * Lp + nt6, H)
            dYpv, dYtext = dY6[:4 * B * Lp], dY6[4 * B * Lp:]
            dY_mpm, dY_lm = dYpv[3 * B * Lp:], dYtext[nt6 - B * Lt:]

        # ---- MPM head
        dmln = self._new(B * Lp, H)
        ops.mpm_head(T["mln"], Lp, H, P.w("property_mtr_head.3.weight"), P.w("property_mtr_head.3.bias"), T["prop"], T["mpm_mask"], B=B,
                     ws=self.icount[1:2], losses=scratch, slot=LOSS_MPM, dh=dmln, dw=P.g("property_mtr_head.3.weight"),
                     db=P.g("property_mtr_head.3.bias"), gscale=gs[1:2])
        dmz = self._new(B * Lp, H)
        ops.ln_bwd(dmln, T["mt"], T["mmean"], T["mrstd"], P.w("property_mtr_head.2.weight"), dmz,
                   dgamma=P.g("property_mtr_head.2.weight"), dbeta=P.g("property_mtr_head.2.bias"))
        dmpre = self._gelu_bwd(dmz, T["mpre"])
        self._wgrad(dmpre, T["hp12"], P.g("property_mtr_head.0.weight"), P.g("property_mtr_head.0.bias"))
        ops.gemm_nt(dmpre, self._wT("property_mtr_head.0", P.w("property_mtr_head.0.weight")), dY_mpm)

        # ---- LM head
        V = ct.vocab_size
        dlogits = self._new(B * Lt, _ceil(V, 64))
        ops.lm_loss(T["logits"], T["logits_m"], T["ids32"], nseq=B, L=Lt, V=V, alpha=self.alpha, ws=self.icount[0:1], losses=scratch,
                    slot=LOSS_MLM, dlogits=dlogits, gscale=gs[0:1])
        self.lm_head_bwd("text_encoder.", ct, T["lmsv"], dlogits, out=dY_lm)

        # ---- ITM head: gradients of the position-0 rows of the first 3B sequences of both halves
        neg_p, neg_t = T["neg"][:B], T["neg"][B:]
        d_pe = self._zeros(B, Lp * H, dtype=torch.float32)          # hub gradients of prop_embeds / text_embeds (fp32);
        d_te = self._zeros(M + 1, H, dtype=torch.float32)           # text rows packed like text_embeds, + one dump row
        if S6 is not None:
            ytop = S6["ytop"]
            ops.itm_head(ytop[:3 * B], H, ytop[3 * B:6 * B], H, H, P.w("itm_head.weight"), P.w("itm_head.bias"), n=3 * B, B=B, losses=scratch,
                         slot=LOSS_ITM, dxa=dYtop[:3 * B], dxb=dYtop[3 * B:6 * B], dW=P.g("itm_head.weight"), db=P.g("itm_head.bias"), gscale=gs[3:4])
            # ---- S6 backward; the cross-attention K/V data gradients land directly on the unique sources (KVSource)
            dX6 = self._s6_backward_cls(T, dYtop, d_pe, d_te)
            dXpv = dX6[:3 * B * Lp].view(3 * B, Lp * H)
            dXt = dX6[S6["o_tp"]:S6["o_lm"]]
            ops.acc_rows(d_pe, dXpv[0:B])
            ops.acc_rows(d_pe, dXpv[B:2 * B], idx=neg_p, atomic=True)
            ops.acc_rows(d_pe, dXpv[2 * B:3 * B])
            ops.acc_rows(d_te[:M], dXt[:M])
            ops.acc_rows(d_te[:M], dXt[M:2 * M])
            ops.acc_rows(d_te, dX6[S6["o_8"]:], idx=S6["fp"]["neg_rows"], atomic=True)       # packed negatives; rows past their count: skipped
            dXt_lm, dX12 = dX6[S6["o_lm"]:S6["o_12"]], dX6[S6["o_12"]:S6["o_8"]]
        else:
            ditm = self._new(3 * B, H)
            ops.itm_head(T["ypv"], Lp * H, T["itm_text"], H, H, P.w("itm_head.weight"), P.w("itm_head.bias"), n=3 * B, B=B, losses=scratch,
                         slot=LOSS_ITM, dxa=dYpv, dxb=ditm, dW=P.g("itm_head.weight"), db=P.g("itm_head.bias"), gscale=gs[3:4])
            dYtext.index_copy_(0, T["cls_text"], ditm)
            # ---- S6 backward; the cross-attention K/V data gradients land directly on the unique sources (KVSource)
            dX6 = self.stack_bwd("text_encoder.bert.", ct, range(f, n), T["tape6"], dY6, T["g6"],
                                 dkv_acc={id(T["src_text"]): d_te[:M], id(T["src_pv"]): d_pe.view(B * Lp, H)})
            dXpv, dXt = dX6[:4 * B * Lp].view(4 * B, Lp * H), dX6[4 * B * Lp:]
            ops.acc_rows(d_pe, dXpv[0:B])
            ops.acc_rows(d_pe, dXpv[B:2 * B], idx=neg_p, atomic=True)
            ops.acc_rows(d_pe, dXpv[2 * B:3 * B])
            if pk:
                ops.acc_rows(d_te[:M], dXt[:M])
                ops.acc_rows(d_te[:M], dXt[M:2 * M])
                nr = T["neg_rows"]                                 # (rows past a negative's length: skipped)
                ops.acc_rows(d_te, dXt[2 * M:2 * M + T["Mn"]], idx=torch.where(nr < M, nr, torch.full_like(nr, -1)), atomic=True)
            else:
                dte, dxt = d_te[:M].view(B, Lt * H), dXt.view(4 * B, Lt * H)
                ops.acc_rows(dte, dxt[0:B])
                ops.acc_rows(dte, dxt[B:2 * B])
                ops.acc_rows(dte, dxt[2 * B:3 * B], idx=neg_t, atomic=True)
            dXt_lm, dX12 = dXt[nt6 - B * Lt:], dXpv[3 * B:].reshape(B * Lp, H)

        # ---- ITA: stored d loss / d features -> projections -> CLS rows
        for k, proj in enumerate(("property_proj", "text_proj")):
            feat, nrm, cls = T["feats"][proj]
            dproj = self._new(B, E)
            ops.l2norm_bwd(T["dfeat"][k * B:(k + 1) * B], feat, nrm, dproj, gscale=gs[2:3])
            self._wgrad(dproj, cls, P.g(proj + ".weight"), P.g(proj + ".bias"))
            dcls = self._new(B, H)
            ops.gemm_nt(dproj, self._wT(proj, P.w(proj + ".weight")), dcls)
            if k == 0:
                ops.acc_rows(d_pe[:, :H], dcls)
            elif pk:
                ops.acc_rows(d_te, dcls, idx=pk["row0_64"])
            else:
                ops.acc_rows(d_te[:M].view(B, Lt * H)[:, :H], dcls)
        ops.axpy_scalar(P.g("temp").view(1), self.dtemp_ita, scale_ptr=gs[2:3])

        # ---- S2 backward (text layers 0..f-1 on P2 | P10a) and the text embeddings -- on the side stream, next to ...
        side = self._fork()
        with self._on(side):
            dY2 = self._new(M + B * Lt, H)
            ops.cast_f32_bf16(d_te[:M].view(-1), dY2[:M].view(-1))
            dY2[M:].copy_(dXt_lm)
            dX2 = self.stack_bwd("text_encoder.bert.", ct, range(0, f), T["tape2"], dY2, T["g2"])
            if pk:                                              # back to the dense layout of the embedding kernels (padding rows: zero)
                dX2 = ops.gather_rows2(self._new(2 * B * Lt, H), dX2, pk["inv"])
            dz2 = self._embed_ln_bwd("text_encoder.bert.", ct, T["esv2"], dX2)
            tp = "text_encoder.bert.embeddings."
            ops.embed_bwd(0, dz2, nseq=2 * B, L=Lt, H=H, dpos=P.g(tp + "position_embeddings.weight"),
                          dtype0=P.g(tp + "token_type_embeddings.weight"), ids=T["ids2"], dword=P.g(tp + "word_embeddings.weight"))

        # ---- ... S1 backward (PV encoder on P1 | P11) and the PV embedding
        dY1 = self._new(2 * B * Lp, H)
        ops.cast_f32_bf16(d_pe.view(-1), dY1[:B * Lp].view(-1))
        dY1[B * Lp:].copy_(dX12)
        # the first layers' weight gradients (the last ones this chain reaches) stay on this stream (EngineOptions.pv_wgrad_inline)
        nl1 = cp.num_hidden_layers
        k1 = min(max(int(self.opt.pv_wgrad_inline), 0), nl1) if side is not None else 0
        dX1 = self.stack_bwd("property_encoder.", cp, range(k1, nl1), T["tape1"][k1:], dY1, T["g1"])
        wa, self.wgrad_async = self.wgrad_async, False
        dX1 = self.stack_bwd("property_encoder.", cp, range(0, k1), T["tape1"][:k1], dX1, T["g1"])
        self.wgrad_async = wa
        dz1 = self._embed_ln_bwd("property_encoder.", cp, T["esv1"], dX1)
        pp = "property_encoder.embeddings."
        ops.embed_bwd(1, dz1, nseq=2 * B, L=Lp, H=H, dpos=P.g(pp + "position_embeddings.weight"),
                      dtype0=P.g(pp + "token_type_embeddings.weight"), pv_x=T["prop"], pv_mask=T["mpm_mask"], src_mod=B,
                      d_w=P.g("property_embed.weight"), d_b=P.g("property_embed.bias"), d_cls=P.g("property_cls"),
                      d_masktok=P.g("property_mask"))
        self._join(side)
        self.wgrad_join(release=True)                           # the weight gradients of the side stream are complete from here on
        self._wg_pending = False
        self.tape = None
