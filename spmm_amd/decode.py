"""PV -> SMILES k-beam decoding (BASELINE.json configs[3]): N molecules x k beams per launch with
a key/value cache, returning per molecule the hypotheses the reference's one-molecule, whole-prefix-per-step search
(`generate`, d_pv2smiles_single.py:26-51; `evaluate`, d_pv2smiles_batched.py:18-59) returns.  That sequential search is
restated in oracle/decode_oracle.py -- test infrastructure, the yard-stick of tests/ -- not here."""
from __future__ import annotations

from typing import List, Tuple

import torch

CLS_ID, SEP_ID = 2, 3           # vocab_bpe_300.txt:3-4
GRAPH_BELOW_ROWS = 200          # beam rows below which one hipGraph replay per position is the default: with the bookkeeping in one launch a position is a
#                                 chain of ~130 dependent kernels (1.95 ms at 100 rows, 2.2 at 1 000, 3.6 at 5 000) and replay gains 7 % at 100 rows, 2 % at
#                                 250, nothing from 500 on -- and a replayed graph has a fixed batch: no compaction of finished molecules
last_run: dict = {}             # what the last eager beam_search_batched did: molecules, compactions, final_batch, positions
COMPACT_BELOW = 0.75            # the batch is re-gathered once at most this share of its molecules is still live
FUSED_BEAM_STEP = True          # beam bookkeeping of a position as one HIP launch (spmm_beam_step); False: the tensor-op form (BeamBook.update)


def _pick(p: torch.Tensor, k: int, stochastic: bool, generator=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The two k-candidate branches of `generate` (d_pv2smiles_single.py:37-44): the k most probable next tokens, or k
    tokens drawn without replacement from the next-token distribution.  p: [..., V] probabilities -> (log-probs, ids) [..., k]."""
    if stochastic:
        flat = p.reshape(-1, p.shape[-1])
        ids = torch.multinomial(flat, num_samples=k, replacement=False, generator=generator)
        return torch.log(flat.gather(1, ids)).reshape(*p.shape[:-1], k), ids.reshape(*p.shape[:-1], k)
    top = torch.topk(p, k=k, dim=-1)
    return torch.log(top.values), top.indices


@torch.no_grad()
def encode_properties(model, prop: torch.Tensor, prop_mask: torch.Tensor | None = None) -> torch.Tensor:
    """d_pv2smiles_batched.py:24-27: PV [B,53] -> prop_embeds [B,54,H].  prop_mask ([53] or [B,53], 1 = property unknown)
    substitutes the learned mask token for those entries, as conditional generation on a subset of properties does
    (d_pv2smiles_single.py:66-70)."""
    feat = model.property_embed(prop.unsqueeze(2))
    if prop_mask is not None:
        mk = prop_mask.to(feat.device).to(feat.dtype).reshape(-1, prop.shape[1])[..., None]
        unk = model.property_mask.detach().to(feat.device).to(feat.dtype).expand(feat.shape[0], feat.shape[1], -1)
        feat = feat * (1 - mk) + unk * mk
    cls = model.property_cls
    properties = torch.cat([cls.expand(feat.size(0), -1, -1).to(feat.dtype).to(feat.device), feat], dim=1)
    return model.property_encoder(inputs_embeds=properties, return_dict=True).last_hidden_state


# ------------------------------------------------------------------------------------------------------------------
# Batched decoding: N molecules x k beams per launch, key/value cache, no host round trips inside the loop.
# ------------------------------------------------------------------------------------------------------------------
class BeamBook:
    """The beam bookkeeping of the reference's search for N independent molecules at once, as tensor ops (no `.item()`):
    per molecule it makes exactly the decisions d_pv2smiles_batched.py:29-57 makes -- candidates ending in [SEP] are moved to
    `final` in row-major order and struck out with -1e5, the molecule stops once it holds >= k finals, the k best of the
    k*k candidates survive."""

    def __init__(self, N: int, k: int, max_steps: int, device, fused: bool = False):
        self.N, self.k, self.Lmax = N, k, max_steps + 3
        self.F = 2 * k                                    # < k finals before the last appending step, <= k appended by it (one [SEP] per beam)
        it = torch.int32 if fused else torch.long         # fused: the state csrc/decode.hip::beam_step_kernel updates in place
        self.fused = fused
        self.tokens = torch.zeros(N, k, self.Lmax, dtype=it, device=device)
        self.tokens[:, :, 0] = CLS_ID
        self.t = 1                                        # tokens held by every live beam
        self.cur_p = torch.zeros(N, k, device=device)
        self.fin_p = torch.full((N, self.F + 1), -float("inf"), device=device)           # slot F is a write-only dump
        self.fin_len = torch.zeros(N, self.F + 1, dtype=it, device=device)
        self.fin_tok = torch.zeros(N, self.F + 1, self.Lmax, dtype=it, device=device)
        self.fin_n = torch.zeros(N, dtype=it, device=device)
        self.done = torch.zeros(N, dtype=torch.bool, device=device)
        self.n_done = torch.zeros(1, dtype=torch.int32, device=device) if fused else None
        self.mol = None                                   # fused, after compact(): int32 indices of the molecules still decoded

    def first(self, values: torch.Tensor, indices: torch.Tensor):
        """values/indices [N,k]: log-probs and ids of the k best successors of [CLS]."""
        self.tokens[:, :, 1] = indices
        self.cur_p = values.clone()
        self.t = 2

    def update(self, values: torch.Tensor, indices: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """values/indices [N,k,k]: per live beam the k best next tokens.  Returns (parent [N,k], token [N,k]) of the new beams."""
        N, k, t, F = self.N, self.k, self.t, self.F
        k2_p = (self.cur_p[:, :, None] + values).reshape(N, k * k)
        idx = indices.reshape(N, k * k)
        ends = (idx == SEP_ID) & ~self.done[:, None]
        e = ends.long()
        slot = torch.where(ends, self.fin_n[:, None] + torch.cumsum(e, 1) - e, torch.full_like(e, F))
        self.fin_p.scatter_(1, slot, k2_p)
        self.fin_len.scatter_(1, slot, torch.full_like(e, t + 1))
        cand = self.tokens[:, :, None, :].expand(N, k, k, self.Lmax).reshape(N, k * k, self.Lmax).clone()
        cand[:, :, t] = SEP_ID
        self.fin_tok.scatter_(1, slot[:, :, None].expand(N, k * k, self.Lmax), cand)
        self.fin_n = self.fin_n + e.sum(1)
        k2_p = torch.where(ends, torch.full_like(k2_p, -1e5), k2_p)
        new_p, flat = torch.topk(k2_p, k, dim=1)
        parent = flat // k
        tok = idx.gather(1, flat)
        live = ~(self.done | (self.fin_n >= k))           # a molecule that just reached k finals breaks before this update
        new_tokens = self.tokens.gather(1, parent[:, :, None].expand(N, k, self.Lmax)).clone()
        new_tokens[:, :, t] = tok
        self.tokens = torch.where(live[:, None, None], new_tokens, self.tokens)
        self.cur_p = torch.where(live[:, None], new_p, self.cur_p)
        self.done = self.done | (self.fin_n >= k)
        self.t = t + 1
        return parent, tok

    def update_dev(self, values: torch.Tensor, indices: torch.Tensor, t64: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """`update` with the number of tokens held by the live beams (t64: int64 [1]) in device memory and every piece of state
        updated in place, so that the call sequence is identical at every step (graph replay).  Same decisions as `update`."""
        N, k, F, L = self.N, self.k, self.F, self.Lmax
        k2_p = (self.cur_p[:, :, None] + values).reshape(N, k * k)
        idx = indices.reshape(N, k * k)
        ends = (idx == SEP_ID) & ~self.done[:, None]
        e = ends.long()
        slot = torch.where(ends, self.fin_n[:, None] + torch.cumsum(e, 1) - e, torch.full_like(e, F))
        self.fin_p.scatter_(1, slot, k2_p)
        self.fin_len.scatter_(1, slot, (t64 + 1).expand(N, k * k).contiguous())
        cand = self.tokens[:, :, None, :].expand(N, k, k, L).reshape(N, k * k, L).clone()
        cand.index_fill_(2, t64, SEP_ID)
        self.fin_tok.scatter_(1, slot[:, :, None].expand(N, k * k, L), cand)
        self.fin_n.add_(e.sum(1))
        k2_p = torch.where(ends, torch.full_like(k2_p, -1e5), k2_p)
        new_p, flat = torch.topk(k2_p, k, dim=1)
        parent = flat // k
        tok = idx.gather(1, flat)
        live = ~(self.done | (self.fin_n >= k))
        new_tokens = self.tokens.gather(1, parent[:, :, None].expand(N, k, L)).clone()
        new_tokens.scatter_(2, t64.view(1, 1, 1).expand(N, k, 1), tok[:, :, None])
        self.tokens.copy_(torch.where(live[:, None, None], new_tokens, self.tokens))
        self.cur_p.copy_(torch.where(live[:, None], new_p, self.cur_p))
        self.done.logical_or_(self.fin_n >= k)
        return parent, tok

    def live_slots(self) -> torch.Tensor:
        """Positions (in the current, possibly compacted batch) of the molecules that are not done yet."""
        done = self.done if self.mol is None else self.done[self.mol.long()]
        return (~done).nonzero().squeeze(1)

    def compact(self, keep: torch.Tensor):
        """Keep only the batch slots `keep` (from live_slots()); the state arrays stay whole, the kernel indexes them through `mol`."""
        cur = torch.arange(self.N, dtype=torch.int32, device=self.done.device) if self.mol is None else self.mol
        self.mol = cur[keep].contiguous()

    def step_fused(self, logits: torch.Tensor, anc: torch.Tensor | None = None, t_ptr: torch.Tensor | None = None, t_off: int = 0,
                   ids_out: torch.Tensor | None = None, rowmap: torch.Tensor | None = None) -> torch.Tensor:
        """`update` (and the decoder's ancestry reorder) as one HIP launch on the next-token logits [N*k, V]: softmax, the k best successors
        per beam, finals, survivors, token histories, `anc` -- all in place.  Returns the tokens to feed next (int32 [N*k]).  With t_ptr
        (device int32 [1]) the number of tokens held comes from device memory (*t_ptr + t_off) and self.t is left to the caller."""
        from . import ops
        ids = ops.beam_step(logits, self, t=self.t, t_ptr=t_ptr, t_off=t_off, anc=anc, ids_out=ids_out, rowmap=rowmap)
        if t_ptr is None:
            self.t += 1
        return ids

    def all_done(self) -> bool:
        """Host check (one small device read): every molecule holds its k finals."""
        return int(self.n_done.item()) == self.N if self.fused else bool(self.done.all())

    def results(self) -> List[List[Tuple[float, List[int]]]]:
        k, F = self.k, self.F
        p = self.fin_p[:, :F]
        order = torch.sort(p, dim=1, descending=True, stable=True).indices[:, :k].cpu()
        p, ln, tk, n = p.cpu(), self.fin_len[:, :F].cpu(), self.fin_tok[:, :F].cpu(), self.fin_n.cpu()
        out = []
        for m in range(self.N):
            hyp = []
            for j in order[m].tolist():
                if j < int(n[m]):
                    hyp.append((float(p[m, j]), tk[m, j, : int(ln[m, j])].tolist()))
            out.append(hyp)
        return out


class RecomputeDecoder:
    """Step function with the reference's cost model: re-runs the whole prefix of every beam through the module API
    (`model.text_encoder(..., is_decoder=True, return_logits=True)`), so it works with the CPU oracle as well."""

    def __init__(self, model, prop_embeds: torch.Tensor, k: int, Lmax: int):
        self.m, self.k = model, k
        self.kv = prop_embeds.repeat_interleave(k, dim=0)
        self.tok = torch.zeros(self.kv.shape[0], Lmax, dtype=torch.long, device=prop_embeds.device)

    def step(self, ids: torch.Tensor, t: int) -> torch.Tensor:
        self.tok[:, t] = ids
        text = self.tok[:, : t + 1]
        return self.m.text_encoder(text, attention_mask=torch.ones_like(text), encoder_hidden_states=self.kv,
                                   encoder_attention_mask=torch.ones(self.kv.shape[:-1], dtype=torch.long, device=self.kv.device),
                                   return_dict=True, is_decoder=True, return_logits=True)[:, -1, :]

    def reorder(self, parent: torch.Tensor, t: int):
        N, k = parent.shape
        L = self.tok.shape[1]
        self.tok = self.tok.view(N, k, L).gather(1, parent[:, :, None].expand(N, k, L)).reshape(N * k, L)


class CachedDecoder:
    """One new token per beam per step on the HIP engine: per-layer self-attention K/V cache [R, Lmax, H] that is never
    moved (the ancestry table `anc[r, j]` names the cache row holding position j of row r's hypothesis), cross-attention
    K/V projected once per molecule and shared by its k beams (cache slots sketched at
    xbert.py:291-295,480,1344-1348)."""

    def __init__(self, model, prop_embeds: torch.Tensor, k: int, Lmax: int):
        from . import ops
        from .engine import BF
        self.ops, self.BF = ops, BF
        eng = model.engine
        self.eng, self.P, self.c, self.k = eng, eng.P, model.cfg.text, k
        self.pfx = "text_encoder."
        c, dev = self.c, model.device_
        N, Lkv, H = prop_embeds.shape
        assert Lmax <= 256 and H == c.hidden_size and c.hidden_size // c.num_attention_heads == 64
        self.N, self.R, self.Lmax, self.Lp, self.H = N, N * k, Lmax, Lkv, H
        nl = c.num_hidden_layers
        # head-major cache [R, nH, Lmax, 64]: the positions of a (row, head) are one contiguous 128-B-per-key stream for the wave that reads them
        self.kc = [torch.empty(self.R, c.num_attention_heads, Lmax, 64, dtype=BF, device=dev) for _ in range(nl)]
        self.vc = [torch.empty(self.R, c.num_attention_heads, Lmax, 64, dtype=BF, device=dev) for _ in range(nl)]
        self.anc = torch.arange(self.R, dtype=torch.int32, device=dev)[:, None].repeat(1, Lmax).contiguous()
        self.rows = torch.arange(self.R, dtype=torch.int32, device=dev)
        self.rowmap = None                               # after compact(): cache row of every beam row still decoded (None: the row itself)
        kv_src = prop_embeds.to(dev).to(BF).reshape(N * Lkv, H).contiguous()
        self.xkv = {}
        for l in range(c.fusion_layer, nl):
            pf = f"{self.pfx}bert.encoder.layer.{l}.crossattention.self."
            KV = torch.empty(N * Lkv, 2 * H, dtype=BF, device=dev)
            ops.gemm_nt(kv_src, self.P.fused(pf, ("key", "value"), "weight"), KV, bias=self.P.fused(pf, ("key", "value"), "bias", what="w"))
            self.xkv[l] = KV

    def _new(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.BF, device=self.anc.device)

    @torch.no_grad()
    def compact(self, keep: torch.Tensor):
        """Drop every molecule but the batch slots `keep` (int64, ascending) from the decoded batch: activations shrink to len(keep) * k
        rows.  The K/V caches are NOT moved -- the ancestry table keeps naming the cache rows, and `rowmap` says where a surviving row
        writes its next position -- only the ancestry rows and the (per-molecule) cross-attention keys / values are gathered."""
        k, L, Lp = self.k, self.Lmax, self.Lp
        cur = self.rows if self.rowmap is None else self.rowmap
        self.rowmap = cur.view(self.N, k)[keep].reshape(-1).contiguous()
        self.anc = self.anc.view(self.N, k, L)[keep].reshape(-1, L).contiguous()
        for l, KV in self.xkv.items():
            self.xkv[l] = KV.view(self.N, Lp, KV.shape[1])[keep].reshape(-1, KV.shape[1]).contiguous()
        self.N = int(keep.numel())
        self.R = self.N * k

    def _attn_out(self, pf, ctx, resid):
        ops, P, c = self.ops, self.P, self.c
        x = self._new(self.R, self.H)
        ops.gemm_nt(ctx, P.wb(pf + "output.dense.weight"), x, bias=P.w(pf + "output.dense.bias"))
        y = self._new(self.R, self.H)
        ops.ln_fwd(x, resid, P.w(pf + "output.LayerNorm.weight"), P.w(pf + "output.LayerNorm.bias"), y, eps=c.layer_norm_eps)
        return y

    @torch.no_grad()
    def step(self, ids: torch.Tensor, t: int, t_dev: torch.Tensor | None = None) -> torch.Tensor:
        """ids [R]: the token at position t of every beam -> fp32 logits [R, V] for position t + 1.
        t_dev (int32 [1], device): the position comes from device memory instead (t is then ignored), which makes the whole
        launch sequence independent of the step -- capturable once as a hipGraph and replayed (beam_search_batched(graph=True))."""
        ops, P, c, R, H, nH = self.ops, self.P, self.c, self.R, self.H, self.c.num_attention_heads
        bp = self.pfx + "bert."
        x = self._new(R, H)
        ops.embed_step_ln_fwd(ids.to(torch.int32).contiguous(), t, x, pos_ptr=t_dev, word=P.w(bp + "embeddings.word_embeddings.weight"),
                              pos=P.w(bp + "embeddings.position_embeddings.weight"), type0=P.w(bp + "embeddings.token_type_embeddings.weight"),
                              gamma=P.w(bp + "embeddings.LayerNorm.weight"), beta=P.w(bp + "embeddings.LayerNorm.bias"), eps=c.layer_norm_eps)
        for l in range(c.num_hidden_layers):
            lp = f"{bp}encoder.layer.{l}."
            pf = lp + "attention."
            QKV = self._new(R, 3 * H)
            ops.gemm_nt(x, P.fused(pf + "self.", ("query", "key", "value"), "weight"), QKV,
                        bias=P.fused(pf + "self.", ("query", "key", "value"), "bias", what="w"))
            ctx = self._new(R, H)                       # (the launch also moves the new key / value rows into the cache)
            ops.decode_attn(QKV[:, :H], self.kc[l], self.vc[l], ctx, nH=nH, Lkv=self.Lmax if t_dev is not None else t + 1,
                            seq_stride=self.Lmax * H, tok_stride=64, head_stride=self.Lmax * 64, anc=self.anc, group=self.k, t_ptr=t_dev, knew=QKV[:, H:2 * H], vnew=QKV[:, 2 * H:],
                            rowmap=self.rowmap)
            a = self._attn_out(pf, ctx, x)
            if l >= c.fusion_layer:
                pf = lp + "crossattention."
                q = self._new(R, H)
                ops.gemm_nt(a, P.wb(pf + "self.query.weight"), q, bias=P.w(pf + "self.query.bias"))
                KV = self.xkv[l]
                ops.decode_attn(q, KV[:, :H], KV[:, H:], ctx, nH=nH, Lkv=self.Lp, seq_stride=self.Lp * 2 * H, tok_stride=2 * H, kv_div=self.k, group=self.k)
                a = self._attn_out(pf, ctx, a)
            h = self._new(R, c.intermediate_size)
            ops.gemm_nt(a, P.wb(lp + "intermediate.dense.weight"), h, bias=P.w(lp + "intermediate.dense.bias"), epi=ops.EPI_GELU)
            x2 = self._new(R, H)
            ops.gemm_nt(h, P.wb(lp + "output.dense.weight"), x2, bias=P.w(lp + "output.dense.bias"))
            x = self._new(R, H)
            ops.ln_fwd(x2, a, P.w(lp + "output.LayerNorm.weight"), P.w(lp + "output.LayerNorm.bias"), x, eps=c.layer_norm_eps)
        logits, _ = self.eng.lm_head_fwd(self.pfx, c, x, False)
        return logits

    @torch.no_grad()
    def reorder(self, parent: torch.Tensor, t: int):
        """New beam b of molecule n continues old beam parent[n, b]; positions < t are inherited, position t is its own."""
        N, k, L = self.N, self.k, self.Lmax
        self.anc = self.anc.view(N, k, L).gather(1, parent[:, :, None].expand(N, k, L)).reshape(N * k, L).contiguous()
        self.anc[:, t:] = self.rows[:, None]

    @torch.no_grad()
    def reorder_dev(self, parent: torch.Tensor, t_next64: torch.Tensor):
        """`reorder` with the next position in device memory and the table updated in place (static address for graph replay)."""
        N, k, L = self.N, self.k, self.Lmax
        self.anc.copy_(self.anc.view(N, k, L).gather(1, parent[:, :, None].expand(N, k, L)).reshape(N * k, L))
        self.anc.index_copy_(1, t_next64, self.rows[:, None])


@torch.no_grad()
def beam_search_batched(model, props: torch.Tensor, k: int = 5, max_steps: int = 100, cached: bool | None = None,
                        sync_every: int = 4, prop_mask: torch.Tensor | None = None, stochastic: bool = False,
                        generator=None, graph: bool | None = None, compact: bool = True) -> List[List[Tuple[float, List[int]]]]:
    """The reference's beam search for N molecules at once (props [N,53]); result[n] is what the one-molecule search
    (oracle/decode_oracle.py::beam_search) returns for props[n].
    cached=True (default on the HIP model) decodes one token per step against the K/V cache; cached=False re-runs the prefix
    through the module API (any model exposing it, e.g. the CPU oracle).  prop_mask: properties to leave unspecified
    (encode_properties).  stochastic=True draws the k candidates of every beam from the next-token distribution instead of
    taking the k most probable (d_pv2smiles_single.py:37-40); `generator` seeds those draws.  graph=True (cached, deterministic)
    captures one decode position -- ~230 launches -- as a hipGraph and replays it: the per-position host cost drops from ~2.3 ms
    of launch overhead to one graph launch, which is what small batches are bound by.  graph=None (default): replay when the batch
    is launch-bound -- fewer than GRAPH_BELOW_ROWS beam rows -- and the search is deterministic.  compact=True (eager fused path): finished
    molecules are dropped from the batch as the search goes (same results; the reference decodes one molecule at a time and simply stops)."""
    last_run.clear()
    if cached is None:
        cached = hasattr(model, "engine")
    if graph is None:
        # Replay is OPT-IN (graph=True) since round 5: every call re-captures its ~130 kernels (a graph is tied to this call's buffers), capture
        # is process-global on the capture stream's device -- another thread touching the GPU meanwhile aborts it -- and a replayed graph
        # cannot drop finished molecules (`compact`).  It gains 7 % at 100 beam rows and nothing from 500 on (GRAPH_BELOW_ROWS).
        graph = False
    prop_embeds = encode_properties(model, props, prop_mask)
    N, dev = prop_embeds.shape[0], prop_embeds.device
    if cached:
        model.engine.train_mode = False
    dec = (CachedDecoder if cached else RecomputeDecoder)(model, prop_embeds, k, max_steps + 3)
    # one launch per position for the beam bookkeeping (csrc/decode.hip::beam_step_kernel: k <= 8 beams, vocabulary <= 512, histories <= 256
    # tokens -- the tensor-op bookkeeping serves everything else)
    fused = bool(cached and not stochastic and k <= 8 and FUSED_BEAM_STEP and model.cfg.text.vocab_size <= 512 and max_steps + 3 <= 256)
    book = BeamBook(N, k, max_steps, dev, fused=fused)
    ids = torch.full((N * k,), CLS_ID, dtype=torch.long, device=dev)
    logits = dec.step(ids, 0).view(N, k, -1)[:, 0]                       # all k rows hold the same [CLS] prefix
    values, indices = _pick(torch.softmax(logits.float(), dim=-1), k, stochastic, generator)
    book.first(values, indices)
    ids = indices.reshape(N * k)
    if graph and cached and not stochastic:
        return _decode_graphed(dec, book, ids, N, k, max_steps, sync_every)
    n_cur = N
    last_run.update(molecules=N, compactions=0, final_batch=N, positions=0)
    for s in range(max_steps):
        logits = dec.step(ids, s + 1)
        if fused:
            ids = book.step_fused(logits, dec.anc, rowmap=dec.rowmap)
        else:
            values, indices = _pick(torch.softmax(logits.view(N, k, -1).float(), dim=-1), k, stochastic, generator)
            parent, tok = book.update(values, indices)
            dec.reorder(parent, s + 2)
            ids = tok.reshape(N * k)
        if s % sync_every == sync_every - 1:
            if not fused:
                if book.all_done():
                    last_run["positions"] = s + 1
                    break
                continue
            n_live = N - int(book.n_done.item())         # the one host read of the loop
            if n_live == 0:
                last_run["positions"] = s + 1
                break
            if compact:
                # molecules that hold their k finals stop costing anything: once a quarter of the batch is done the rest is gathered into
                # a smaller batch (caches stay where they are, CachedDecoder.compact)
                if n_live <= COMPACT_BELOW * n_cur and n_cur - n_live >= 4:
                    keep = book.live_slots()
                    dec.compact(keep)
                    book.compact(keep)
                    ids = ids.view(n_cur, k)[keep].reshape(-1).contiguous()
                    n_cur = int(keep.numel())
                    last_run["compactions"] += 1
                    last_run["final_batch"] = n_cur
        last_run["positions"] = s + 1
    return book.results()


# ------------------------------------------------------------------------------------------------------------------
# SMILES -> PV: 53 autoregressive regression steps
# ------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def smiles_to_pv(model, text_ids: torch.Tensor, text_mask: torch.Tensor, n_props: int = 53) -> torch.Tensor:
    """Predict the (normalised) property vector of every SMILES in the batch, one property per step, as
    d_smiles2pv.py:14-52 does: the text is encoded once by the unimodal text layers; at step i the PV prefix
    [CLS, p_0..p_{i-1}] goes through the (bidirectional) PV encoder, then causally through the fusion layers with
    cross-attention to the text, and `property_mtr_head` reads property i off the last position.  The whole batch advances
    together; the prefix is re-encoded every step because the PV encoder is bidirectional (nothing to cache).
    text_ids / text_mask: [B, Lt] without the tokenizer's own [CLS] (the caller drops column 0, SPMM_models.py:357).
    Returns [B, n_props] in the normalised space (de-normalise with the dataset's mean/std as d_smiles2pv.py:49 does)."""
    text_embeds = model.text_encoder.bert(text_ids, attention_mask=text_mask, return_dict=True, mode="text").last_hidden_state
    B, dev = text_embeds.shape[0], text_embeds.device
    cls = model.property_cls
    prefix = cls.detach().to(dev).to(text_embeds.dtype).expand(B, -1, -1)
    out = []
    for _ in range(n_props):
        pv = model.property_encoder(inputs_embeds=prefix, return_dict=True).last_hidden_state
        ones = torch.ones(pv.shape[:-1], dtype=torch.long, device=dev)
        fused = model.text_encoder.bert(encoder_embeds=pv, attention_mask=ones, encoder_hidden_states=text_embeds,
                                        encoder_attention_mask=text_mask.to(dev), return_dict=True, is_decoder=True,
                                        mode="fusion").last_hidden_state
        nxt = model.property_mtr_head(fused[:, -1:, :]).reshape(B)            # only the last position is read (:25)
        out.append(nxt)
        prefix = torch.cat([prefix, model.property_embed(nxt.reshape(B, 1, 1)).to(prefix.dtype)], dim=1)
    return torch.stack(out, dim=-1)


def _decode_graphed(dec: "CachedDecoder", book: BeamBook, ids: torch.Tensor, N: int, k: int, max_steps: int, sync_every: int):
    """The loop of beam_search_batched with every step-dependent scalar in device memory: two eager positions (they also set the
    kernels' one-time attributes), then one captured position replayed for the rest."""
    dev = ids.device
    ids_s = ids.to(torch.int32).clone() if book.fused else ids.clone()      # static input of the graph
    t_dev = torch.ones(1, dtype=torch.int32, device=dev)         # position of the token in ids_s

    def one_position():
        logits = dec.step(ids_s, 0, t_dev=t_dev)
        if book.fused:
            book.step_fused(logits, dec.anc, t_ptr=t_dev, t_off=1, ids_out=ids_s)      # ids_s: the kernel's output and the next position's input
        else:
            t64 = t_dev.to(torch.int64)
            values, indices = _pick(torch.softmax(logits.view(N, k, -1).float(), dim=-1), k, False)
            parent, tok = book.update_dev(values, indices, t64 + 1)
            dec.reorder_dev(parent, t64 + 1)
            ids_s.copy_(tok.reshape(N * k))
        t_dev.add_(1)

    eager = min(2, max_steps)
    for _ in range(eager):
        one_position()
    steps_left = max_steps - eager
    if steps_left > 0 and not book.all_done():
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):      # other threads' GPU work does not abort this capture
            one_position()
        g.replay()                                               # capture records, it does not execute: this is position `eager`
        for s in range(1, steps_left):
            if s % sync_every == 0 and book.all_done():
                break
            g.replay()
    book.t = int(t_dev.item()) + 1
    return book.results()
