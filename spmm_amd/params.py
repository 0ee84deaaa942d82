"""Parameter arena: every trainable tensor of the student lives in ONE flat fp32 buffer (plus flat grad / Adam m / Adam v
/ bf16 compute shadow), the momentum twins in a second flat buffer with the SAME offsets, so that clip, AdamW and the EMA
are single multi-tensor launches.  Tensors are exposed under the reference's state_dict names ([out,in] fp32 layout,
SURVEY.md section 5); q/k/v weights (and biases) of an attention block are laid out back to back so the fused QKV GEMM can
read them as one [3H,H] matrix without copies."""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import ops
from .config import SPMMConfig, is_buffer, momentum_twin, state_spec, student_of

ALIGN = 64   # elements; keeps every tensor 256-B (fp32) / 128-B (bf16) aligned


def _layout_order(spec) -> List[str]:
    """Flat order of the student's trainable tensors: q.w,k.w,v.w then q.b,k.b,v.b contiguous per attention block."""
    names = [n for n, _, k in spec if not is_buffer(n) and student_of(n) is None and k not in ("tied_w", "tied_b")]
    done, out = set(), []
    for n in names:
        if n in done:
            continue
        if n.endswith(".self.query.weight"):
            base = n[:-len("query.weight")]
            grp = [base + f"{x}.{y}" for y in ("weight", "bias") for x in ("query", "key", "value")]
            out += grp
            done.update(grp)
        else:
            out.append(n)
            done.add(n)
    assert sorted(out) == sorted(names)
    return out


class ParamStore:
    def __init__(self, cfg: SPMMConfig, device, train: bool = True):
        self.cfg, self.device, self.train = cfg, device, train
        self.spec = state_spec(cfg)
        self.shape = {n: s for n, s, _ in self.spec}
        self.kind = {n: k for n, _, k in self.spec}
        self.order = _layout_order(self.spec)
        self.offset: Dict[str, int] = {}
        off = 0
        for n in self.order:
            self.offset[n] = off
            numel = int(math.prod(self.shape[n])) if self.shape[n] else 1
            off += (numel + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        f32 = dict(dtype=torch.float32, device=device)
        self.flat = torch.zeros(self.total, **f32)
        self.flat_m = torch.zeros(self.total, **f32)
        self.shadow = torch.zeros(self.total, dtype=torch.bfloat16, device=device)
        self.shadow_m = torch.zeros(self.total, dtype=torch.bfloat16, device=device)
        if train:
            self.grad = torch.zeros(self.total, **f32)
            self.adam_m = torch.zeros(self.total, **f32)
            self.adam_v = torch.zeros(self.total, **f32)
        H, E, Q = cfg.text.hidden_size, cfg.embed_dim, cfg.queue_size
        self.buffers: Dict[str, torch.Tensor] = {}
        for n, s, k in self.spec:
            if k == "posid":
                self.buffers[n] = torch.arange(s[1], device=device).expand(1, -1).clone()
            elif k == "queue":
                self.buffers[n] = torch.zeros(s, **f32)
            elif k == "ptr":
                self.buffers[n] = torch.zeros(1, dtype=torch.long, device=device)
        self._wT: Dict[str, torch.Tensor] = {}     # transposed bf16 shadows for dgrad, keyed by (fused) name
        self._wF: Dict[str, tuple] = {}            # fused cross-attention: name -> (bf16 shadow view, fragment-ordered image, is momentum)

    # ---- views -----------------------------------------------------------------------------------------------
    def _resolve(self, name: str):
        """-> (flat buffer selector, student name) ; momentum names map onto the twin arena."""
        st = student_of(name)
        mom = st is not None
        base = st if mom else name
        if self.kind.get(base) == "tied_w":
            base = base.replace("cls.predictions.decoder.weight", "bert.embeddings.word_embeddings.weight")
        elif self.kind.get(base) == "tied_b":
            base = base.replace("cls.predictions.decoder.bias", "cls.predictions.bias")
        return mom, base

    def _view(self, buf: torch.Tensor, base: str, shape=None):
        shape = self.shape[base] if shape is None else shape
        n = int(math.prod(shape)) if shape else 1
        return buf[self.offset[base]: self.offset[base] + n].view(shape)

    def w(self, name: str) -> torch.Tensor:            # fp32 master
        mom, base = self._resolve(name)
        return self._view(self.flat_m if mom else self.flat, base)

    def wb(self, name: str) -> torch.Tensor:           # bf16 compute shadow [out,in]
        mom, base = self._resolve(name)
        return self._view(self.shadow_m if mom else self.shadow, base)

    def g(self, name: str) -> torch.Tensor:            # fp32 gradient
        mom, base = self._resolve(name)
        assert not mom
        return self._view(self.grad, base)

    def fused(self, prefix: str, names, kind: str, what: str = "wb") -> torch.Tensor:
        """Rows of several adjacent tensors as one matrix/vector, e.g. fused(p+'.self.', ('query','key','value'), 'weight')."""
        first = f"{prefix}{names[0]}.{kind}"
        mom, base = self._resolve(first)
        shp = self.shape[base]
        rows = sum(self.shape[self._resolve(f"{prefix}{n}.{kind}")[1]][0] for n in names)
        # adjacency check
        o = self.offset[base]
        for n in names:
            b = self._resolve(f"{prefix}{n}.{kind}")[1]
            assert self.offset[b] == o, (b, self.offset[b], o)
            o += int(math.prod(self.shape[b]))
        buf = {"wb": self.shadow_m if mom else self.shadow, "w": self.flat_m if mom else self.flat,
               "g": None if mom else getattr(self, "grad", None)}[what]
        full = (rows,) + tuple(shp[1:])
        return buf[self.offset[base]: self.offset[base] + int(math.prod(full))].view(full)

    def wT(self, key: str, src: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Transposed bf16 shadow [in,out] of a (fused) student weight; refreshed by refresh_shadows()."""
        if key not in self._wT:
            assert src is not None
            R, C = src.shape
            self._wT[key] = torch.empty(C, R, dtype=torch.bfloat16, device=self.device)
            self._wT_src = getattr(self, "_wT_src", {})
            self._wT_src[key] = src
            ops.cast_transpose(src, None, self._wT[key])
        return self._wT[key]

    def wF(self, name: str) -> torch.Tensor:
        """Fragment-ordered bf16 image of a cross-attention output-projection weight (csrc/xattn.hip: every wave-level load of the
        MFMA A operand is 1 KiB contiguous); built from the bf16 shadow on first use, refreshed by refresh_frag()."""
        if name not in self._wF:
            src = self.wb(name)
            self._wF[name] = (src, torch.empty(src.numel(), dtype=torch.bfloat16, device=self.device), student_of(name) is not None)
            ops.xattn_pack_wo(src, self._wF[name][1])
        return self._wF[name][1]

    def refresh_frag(self, momentum: bool):
        """Re-pack the fragment-ordered images after their shadows changed (student: optimiser step; momentum twins: EMA)."""
        for src, out, mom in self._wF.values():
            if mom == momentum:
                ops.xattn_pack_wo(src, out)

    # ---- maintenance -----------------------------------------------------------------------------------------
    def refresh_shadows(self, transposed_only: bool = False, part: str = "all"):
        """bf16 shadows <- fp32 masters (after load_state_dict / optimiser step).  AdamW and the EMA kernels already
        refresh the flat shadows; the transposed dgrad shadows are rebuilt here, one launch for all of them.
        part: "forward" = only what the next FORWARD reads (fragment-ordered images of the fused cross-attention),
        "transposed" = only the data-gradient GEMMs' transposed shadows (read by the next BACKWARD: Engine.off_path), "all" = both."""
        if not transposed_only:
            ops.cast_f32_bf16(self.flat, self.shadow)
            ops.cast_f32_bf16(self.flat_m, self.shadow_m)
        if part in ("all", "forward"):
            self.refresh_frag(False)
            if not transposed_only:
                self.refresh_frag(True)
        if part == "forward":
            return
        srcs = getattr(self, "_wT_src", {})
        if not srcs:
            return
        if getattr(self, "_ct_n", -1) != len(srcs):       # (re)build the device descriptor table when shadows were added
            import struct
            blob, tile0 = b"", 0
            for key, src in srcs.items():
                R, C = src.shape
                ntr, ntc = (R + 63) // 64, (C + 63) // 64
                blob += struct.pack("<QQiiii", src.data_ptr(), self._wT[key].data_ptr(), R, C, tile0, ntc)
                tile0 += ntr * ntc
            assert len(blob) == len(srcs) * 32
            self._ct_desc = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.device)
            self._ct_n, self._ct_tiles = len(srcs), tile0
        ops.cast_transpose_multi(self._ct_desc, self._ct_n, self._ct_tiles)

    def copy_params(self):
        """copy_params SPMM_models.py:259-263: momentum twins start as copies of the student."""
        self.flat_m.copy_(self.flat)
        self.shadow_m.copy_(self.shadow)

    # ---- state_dict ------------------------------------------------------------------------------------------
    def named_tensors(self):
        """(name, tensor) in the reference's state_dict order; parameters are views of the flat arenas."""
        for n, _, _ in self.spec:
            yield n, (self.buffers[n] if n in self.buffers else self.w(n))

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        missing = [n for n, _, _ in self.spec if n not in sd]
        unexpected = [k for k in sd if k not in self.shape]
        bad = [k for k in sd if k in self.shape and tuple(sd[k].shape) != tuple(self.shape[k]) and sd[k].numel() != int(math.prod(self.shape[k]) if self.shape[k] else 1)]
        if bad:
            raise RuntimeError(f"size mismatch for {bad[:5]}")
        if strict and (missing or unexpected):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]} unexpected {unexpected[:5]}")
        with torch.no_grad():
            for n, t in self.named_tensors():
                if n in sd:
                    if self.kind[n] in ("tied_w", "tied_b"):
                        continue            # aliases of word_embeddings / predictions.bias, loaded through those
                    t.copy_(sd[n].to(t.device).reshape(t.shape))
        self.refresh_shadows()
        return missing, unexpected
