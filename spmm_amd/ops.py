"""Tensor-level wrappers over the C ABI (include/spmm_hip.h).  Every function enqueues HIP kernels on
torch's current stream and returns immediately; torch is used for device memory and streams only."""
import ctypes
import os

import torch

from ._lib import lib

EPI_BF16, EPI_GELU, EPI_F32, EPI_F32_ATOMIC, EPI_GELU_GRAD, EPI_F32_ACC, EPI_GELU_DERIV, EPI_MUL = range(8)
BF16 = torch.bfloat16


_DRY_RUN = False        # tests only: validate every call against the header's prototype without launching anything
_dry_log = []


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st():
    if _DRY_RUN:
        return None
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_DEBUG_SYNC = os.environ.get("SPMM_DEBUG_SYNC") == "1"      # hang diagnosis: name every launch on stderr and drain the GPU after it


def _call(name, *args):
    if not _DRY_RUN:
        if _DEBUG_SYNC:
            import sys
            r = lib().call(name, *args)
            print(f"[launch] {name} {[a for a in args if isinstance(a, int) and not isinstance(a, bool)][:8]}", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            return r
        return lib().call(name, *args)
    res, argtypes = lib().protos[name]
    if len(args) != len(argtypes):
        raise TypeError(f"{name}: {len(args)} arguments, header declares {len(argtypes)}")
    for i, (a, t) in enumerate(zip(args, argtypes)):
        try:
            t(a) if not isinstance(a, ctypes.c_void_p) else None
        except Exception as e:   # noqa: BLE001
            raise TypeError(f"{name}: argument {i} = {a!r} is not a {t.__name__}") from e
        if t is ctypes.c_void_p and not (a is None or isinstance(a, (ctypes.c_void_p, int))):
            raise TypeError(f"{name}: argument {i} must be a pointer, got {type(a)}")
    _dry_log.append(name)


def _row_stride(t):
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0)


GEMM_AUTO, GEMM_AUTO_TILES = 0, 16      # SPMM_GEMM_AUTO / SPMM_GEMM_AUTO_TILES (include/spmm_hip.h)
_nt_auto = GEMM_AUTO


class nt_tiles_per_workgroup:
    """While active, automatically chosen NT GEMMs run one workgroup per tile instead of one persistent workgroup per CU: for the
    stretch of a step during which a collective kernel holds CUs on RCCL's stream (the data-parallel backward).  Same results bit
    for bit (same tiles, same accumulation order)."""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global _nt_auto
        self._prev = _nt_auto
        if self.on:
            _nt_auto = GEMM_AUTO_TILES
        return self

    def __exit__(self, *exc):
        global _nt_auto
        _nt_auto = self._prev
        return False


def gemm_nt(A, W, C, *, bias=None, epi=EPI_BF16, R=None, G=None, C2=None, alpha=1.0, div=None, splits=1, K=None, colsum=None, kernel=0,
            M_dev=None):
    """C[M,N] (+)= A[M,K] @ W[N,K]^T.  A, W bf16 (row stride free, unit column stride).  kernel: 0 = chosen from the shape,
    1 / 2 / 3 / 8 / 9 force a tile kernel (SPMM_GEMM_* in include/spmm_hip.h)."""
    if kernel == 0:
        kernel = _nt_auto
    M, Ka = A.shape
    N = W.shape[0]
    K = Ka if K is None else K
    assert A.dtype == BF16 and W.dtype == BF16 and W.shape[1] >= K and C.shape[0] >= M and C.shape[1] >= N
    if epi in (EPI_BF16, EPI_GELU, EPI_GELU_GRAD, EPI_GELU_DERIV, EPI_MUL):
        assert C.dtype == BF16
    else:
        assert C.dtype == torch.float32
    _call("spmm_gemm_nt", _p(A), _row_stride(A), _p(W), _row_stride(W), M, N, K, splits, _p(bias), _p(div),
               float(alpha), _p(R), 0 if R is None else _row_stride(R), _p(G), 0 if G is None else _row_stride(G),
               _p(C), _row_stride(C), _p(C2), 0 if C2 is None else _row_stride(C2), epi, _p(colsum), int(kernel), _p(M_dev), _st())
    return C


def gemm_tn(A, B, C, *, alpha=1.0, splits=None, kernel=0, M_dev=None):
    """C[N,K] (fp32) += alpha * A[M,N]^T @ B[M,K]  -- weight gradients straight from token-major activations.
    kernel: 0 = chosen from the shape, 1 = 128x128 tiles, 8 = 256x256 8-phase (N, K multiples of 8)."""
    M, N = A.shape
    K = B.shape[1]
    assert A.dtype == BF16 and B.dtype == BF16 and C.dtype == torch.float32 and B.shape[0] == M and tuple(C.shape) == (N, K)
    if splits is None:
        splits = 1 if _DRY_RUN else lib().cdll.spmm_gemm_tn_splits(M, N, K, int(kernel))
    ws = torch.empty(splits * N * K, dtype=torch.float32, device=C.device) if splits > 1 else None
    _call("spmm_gemm_tn", _p(A), _row_stride(A), _p(B), _row_stride(B), M, N, K, splits, float(alpha), _p(C), _row_stride(C), _p(ws),
          int(kernel), _p(M_dev), _st())
    return C


def colsum_bf16(x, out, *, R_dev=None):
    R, C = x.shape
    _call("spmm_colsum_bf16", _p(x), _row_stride(x), R, C, _p(out), _p(R_dev), _st())
    return out


def attn_fwd(Q, K, V, O, lse, *, nseq, nH, Lq, Lkv, kmask=None, causal_from=None, is_cross=False, dropout_p=0.0,
             seed=None, salt=0, kv_seq=None, q_row0=None, q_len=None, kv_row0=None, kv_len=None, q_off=0, kv_off=0):
    """Q [nseq*Lq, >=nH*64] etc. (2-D views with row strides), O [nseq*Lq, nH*64]; packed layouts via the row0/len arrays."""
    cf = nseq if causal_from is None else causal_from
    _call("spmm_attn_fwd", _p(Q), _row_stride(Q), _p(K), _row_stride(K), _p(V), _row_stride(V), _p(kmask), _p(kv_seq), _p(q_row0),
               _p(q_len), _p(kv_row0), _p(kv_len), _p(O),
               _row_stride(O), _p(lse), nseq, nH, Lq, Lkv, cf, int(is_cross), float(dropout_p), _p(seed), salt, int(q_off), int(kv_off), _st())
    return O


def attn_bwd(Q, K, V, O, lse, dO, dQ, dK, dV, *, nseq, nH, Lq, Lkv, kmask=None, causal_from=None, is_cross=False,
             dropout_p=0.0, seed=None, salt=0, kv_seq=None, q_row0=None, q_len=None, kv_row0=None, kv_len=None, q_off=0, kv_off=0,
             d_mode=0, dbuf=None):
    cf = nseq if causal_from is None else causal_from
    _call("spmm_attn_bwd", _p(Q), _row_stride(Q), _p(K), _row_stride(K), _p(V), _row_stride(V), _p(kmask), _p(kv_seq), _p(q_row0),
               _p(q_len), _p(kv_row0), _p(kv_len), _p(O),
               _row_stride(O), _p(lse), _p(dO), _row_stride(dO), _p(dQ), _row_stride(dQ), _p(dK), _row_stride(dK), _p(dV),
               _row_stride(dV), nseq, nH, Lq, Lkv, cf, int(is_cross), float(dropout_p), _p(seed), salt, int(q_off), int(kv_off), int(d_mode),
               _p(dbuf), _st())


def xattn_supported(H, nH, Lq, Lkv):
    return bool(_DRY_RUN or lib().cdll.spmm_xattn_supported(int(H), int(nH), int(Lq), int(Lkv)))


def xattn_pack_wo(W, out=None):
    """Fragment-ordered bf16 image of an output-projection weight [H, H] (bf16, [out, in]) for xattn_fwd."""
    H = W.shape[0]
    assert W.dtype == BF16 and W.shape[1] == H
    out = torch.empty(H * H, dtype=BF16, device=W.device) if out is None else out
    _call("spmm_xattn_pack_wo", _p(W), _row_stride(W), _p(out), H, _st())
    return out


def xattn_fwd(Q, K, V, WoF, bo, R, gamma, beta, Y, *, nseq, nH, Lq, Lkv, eps=1e-12, kmask=None, kv_seq=None, q_row0=None, q_len=None,
              kv_row0=None, kv_len=None, Z=None, mean=None, rstd=None, CTX=None, lse=None, attn_dropout_p=0.0, salt_a=0,
              hidden_dropout_p=0.0, salt_h=0, seed=None, row_base=0):
    """Fused cross-attention block: Y = LN(dropout_h(softmax(Q K^T / 8 + mask) V . Wo^T + bo) + R)  (csrc/xattn.hip).  Q, R, Y, Z, CTX:
    [rows, H] views of the group's query rows; K, V: [source rows, >= H] views; optional outputs Z / mean / rstd / CTX / lse."""
    _call("spmm_xattn_fwd", _p(Q), _row_stride(Q), _p(K), _row_stride(K), _p(V), _row_stride(V), _p(kmask), _p(kv_seq), _p(q_row0), _p(q_len),
          _p(kv_row0), _p(kv_len), _p(WoF), _p(bo), _p(R), _row_stride(R), _p(gamma), _p(beta), float(eps), _p(Y), _row_stride(Y), _p(Z),
          0 if Z is None else _row_stride(Z), _p(mean), _p(rstd), _p(CTX), 0 if CTX is None else _row_stride(CTX), _p(lse), nseq, nH, Lq, Lkv,
          float(attn_dropout_p), salt_a, float(hidden_dropout_p), salt_h, _p(seed), int(row_base), _st())
    return Y


def ln_fwd(x, res, gamma, beta, y, *, zout=None, mean=None, rstd=None, eps=1e-12, dropout_p=0.0, seed=None, salt=0, rows_dev=None):
    rows, H = x.shape
    _call("spmm_ln_fwd", _p(x), _p(res), _p(gamma), _p(beta), _p(y), _p(zout), _p(mean), _p(rstd), rows, H, float(eps),
               float(dropout_p), _p(seed), salt, _p(rows_dev), _st())
    return y


def ln_fwd_r32(x, res32, gamma, beta, y, *, y32=None, zout=None, mean=None, rstd=None, eps=1e-12, dropout_p=0.0, seed=None, salt=0):
    """ln_fwd with the residual read in fp32 and the normalised row written as bf16 (y) and fp32 (y32)."""
    rows, H = x.shape
    assert res32 is None or res32.dtype == torch.float32
    assert y32 is None or y32.dtype == torch.float32
    _call("spmm_ln_fwd_r32", _p(x), _p(res32), _p(gamma), _p(beta), _p(y), _p(y32), _p(zout), _p(mean), _p(rstd), rows, H, float(eps),
          float(dropout_p), _p(seed), salt, _st())
    return y


def ln_bwd(dy, z, mean, rstd, gamma, dz, *, dy2=None, dx=None, dgamma=None, dbeta=None, dropout_p=0.0, seed=None, salt=0,
           drop_on_dy=False, dxsum=None, rows_dev=None, beta_from_y=None):
    """beta_from_y: `z` is the LayerNorm's OUTPUT y; the normalised values are recovered as (y - beta) / gamma (mean may be None)."""
    rows, H = dy.shape
    _call("spmm_ln_bwd", _p(dy), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dz), _p(dx), _p(dgamma), _p(dbeta),
               rows, H, float(dropout_p), _p(seed), salt, int(drop_on_dy), _p(dxsum), _p(rows_dev), _p(beta_from_y), _st())
    return dz


def embed_ln_fwd(mode, y, *, nseq, L, H, pos, type0, gamma, beta, ids=None, word=None, pv_x=None, pv_mask=None, pv_w=None,
                 pv_b=None, pv_cls=None, pv_masktok=None, src_mod=1, zout=None, mean=None, rstd=None, eps=1e-12,
                 dropout_p=0.0, seed=None, salt=0):
    _call("spmm_embed_ln_fwd", mode, _p(ids), _p(word), _p(pos), _p(type0), _p(pv_x), _p(pv_mask), _p(pv_w), _p(pv_b),
               _p(pv_cls), _p(pv_masktok), src_mod, _p(gamma), _p(beta), _p(y), _p(zout), _p(mean), _p(rstd), nseq, L, H,
               float(eps), float(dropout_p), _p(seed), salt, _st())
    return y


def embed_bwd(mode, dz, *, nseq, L, H, dpos, dtype0, ids=None, dword=None, pv_x=None, pv_mask=None, src_mod=1, d_w=None,
              d_b=None, d_cls=None, d_masktok=None):
    _call("spmm_embed_bwd", mode, _p(dz), _p(ids), _p(pv_x), _p(pv_mask), src_mod, _p(dword), _p(dpos), _p(dtype0),
               _p(d_w), _p(d_b), _p(d_cls), _p(d_masktok), nseq, L, H, _st())


def transpose_bf16(x, out, *, Rpad=None, colsum=None):
    R, C = x.shape
    Rpad = out.shape[1] if Rpad is None else Rpad
    _call("spmm_transpose_bf16", _p(x), _row_stride(x), _p(out), _row_stride(out), R, C, Rpad, _p(colsum), _st())
    return out


def cast_transpose(w32, out, outT):
    R, C = w32.shape
    _call("spmm_cast_transpose", _p(w32), _p(out), _p(outT), R, C, _st())


def cast_transpose_multi(descs_dev, ndesc, total_tiles):
    _call("spmm_cast_transpose_multi", _p(descs_dev), ndesc, total_tiles, _st())


def cast_f32_bf16(x, out):
    _call("spmm_cast_f32_bf16", _p(x), _p(out), x.numel(), _st())
    return out


def cast_bf16_f32(x, out):
    _call("spmm_cast_bf16_f32", _p(x), _p(out), x.numel(), _st())
    return out


def acc_rows(dst, src, *, idx=None, atomic=False):
    rows, H = src.shape
    _call("spmm_acc_rows", _p(dst), _row_stride(dst), _p(src), _row_stride(src), _p(idx), rows, H, int(atomic), _st())


def gather_rows(dst, src, idx):
    rows, H = dst.shape
    _call("spmm_gather_rows", _p(dst), _p(src), _p(idx), rows, H, _st())
    return dst


SRC_B = 1 << 40        # gather_rows2: index bit that selects the second source (csrc/plan.hip)


def gather_rows2(dst, srcA, idx, srcB=None):
    """dst[r] = 0 if idx[r] < 0 else (srcB if idx[r] & SRC_B else srcA)[idx[r] & (SRC_B - 1)]; rows of H bf16-sized elements."""
    rows, H = dst.shape
    assert idx.dtype == torch.int64 and idx.numel() >= rows and dst.is_contiguous() and srcA.is_contiguous() and (srcB is None or srcB.is_contiguous())
    assert dst.dtype == BF16 and srcA.dtype == BF16 and srcA.shape[1] == H and (srcB is None or (srcB.dtype == BF16 and srcB.shape[1] == H))
    _call("spmm_gather_rows2", _p(dst), _p(srcA), _p(srcB), _p(idx), rows, H, _st())
    return dst


def add_rows_bf16(dst, idx, src):
    """dst[idx[r]] += src[r] (bf16 rows, every index at most once)."""
    rows, H = src.shape
    assert dst.dtype == BF16 and src.dtype == BF16 and idx.dtype == torch.int64 and dst.is_contiguous() and src.is_contiguous() and dst.shape[1] == H
    _call("spmm_add_rows_bf16", _p(dst), _p(idx), _p(src), rows, H, _st())
    return dst


def zero_(t):
    """t.zero_() without a tensor-library launch (contiguous, 16-byte granular)."""
    n = t.numel() * t.element_size()
    if n == 0:
        return t
    if not t.is_contiguous():                      # a column slice of a row-major matrix: rows of equal length at a constant stride
        es = t.element_size()
        if t.dim() == 2 and t.stride(1) == 1 and (t.shape[1] * es) % 16 == 0 and (t.stride(0) * es) % 16 == 0 and t.data_ptr() % 16 == 0:
            _call("spmm_zero_rows", _p(t), t.shape[0], t.shape[1] * es, t.stride(0) * es, _st())
            return t
        return t.zero_()
    if n % 16 or t.data_ptr() % 16:
        return t.zero_()
    _call("spmm_zero_bytes", _p(t), n, _st())
    return t


def gelu_bwd(dz, pre, out=None):
    """out = dz * gelu'(pre) (erf-GELU), bf16."""
    assert dz.dtype == BF16 and pre.dtype == BF16 and dz.is_contiguous() and pre.is_contiguous() and dz.shape == pre.shape
    out = torch.empty_like(dz) if out is None else out
    _call("spmm_gelu_bwd", _p(dz), _p(pre), _p(out), dz.numel(), _st())
    return out


def pack_plan(mask32, M, bad):
    """Packed-row plan of the text passes from the [B, Lt] attention mask and the host's valid-token count M (csrc/plan.hip)."""
    B, Lt = mask32.shape
    dev = mask32.device
    alloc = torch.zeros if _DRY_RUN else torch.empty          # (dry runs launch nothing: index 0 keeps the host-side indexing valid)
    i32 = alloc(2 * B, dtype=torch.int32, device=dev)
    i64 = alloc(B + M + (M + B * Lt) + 2 * M + 2 * B * Lt + (B + M), dtype=torch.int64, device=dev)
    o = [0]

    def cut(n):
        o[0] += n
        return i64[o[0] - n:o[0]]
    row0_64, rows, gidx2, gidx4, inv, idx_m = cut(B), cut(M), cut(M + B * Lt), cut(2 * M), cut(2 * B * Lt), cut(B + M)
    _call("spmm_pack_plan", _p(mask32), B, Lt, M, _p(i32[:B]), _p(i32[B:]), _p(row0_64), _p(rows), _p(gidx2), _p(gidx4), _p(inv), _p(idx_m), _p(bad), _st())
    return dict(M=M, rows=rows, row0=i32[B:], row0_64=row0_64, len=i32[:B], gidx2=gidx2, gidx4=gidx4, inv=inv, idx_m=idx_m)


FUSION_SMALL = dict(ar=(0, 1), kvidx_pv=(1, 3), kvidx_tp=(4, 2), qrow0_tp=(6, 2), qlen_tp=(8, 2), row0_8=(10, 1), len_8=(11, 1),
                    skv_row0_pv=(12, 3), skv_len_pv=(15, 3), skv_row0_tx=(18, 2), skv_len_tx=(20, 2), start_t=(22, 2), list_t=(24, 4),
                    start_p=(28, 2), list_p=(30, 4))          # (offset, length) in units of B: the layout csrc/plan.hip writes; then rows_dev[2]


def fusion_plan(neg, pk, Lp):
    """Index arrays of the fusion batch from the sampled negatives (csrc/plan.hip::fusion_plan_kernel) -> dict of views.  `Rcap` = rows the
    batch is allocated for; `rows_dev` / `mn_dev` (int32 [1] device views) = rows it really has / rows of the packed text negatives."""
    B = pk["len"].numel()
    Lt = (pk["inv"].numel() // 2) // B
    M, dev = pk["M"], neg.device
    Rcap = 4 * B * Lp + 2 * M + 2 * B * Lt
    ntop = 6 * B + B * Lt + B * Lp
    alloc = torch.zeros if _DRY_RUN else torch.empty
    i64 = alloc(Rcap + B * Lt + ntop, dtype=torch.int64, device=dev)
    idx6, neg_rows, idx_top = i64[:Rcap], i64[Rcap:Rcap + B * Lt], i64[Rcap + B * Lt:]
    small = alloc(35 * B + 2, dtype=torch.int32, device=dev)     # 34 B-sized tables (FUSION_SMALL), then rows_dev, mn_dev
    _call("spmm_fusion_plan", _p(neg), _p(pk["len"]), _p(pk["row0"]), B, Lt, Lp, M, _p(idx6), _p(neg_rows), _p(idx_top), _p(small), _st())
    out = dict(idx6=idx6, neg_rows=neg_rows, idx_top=idx_top, Rcap=Rcap, rows_dev=small[34 * B:34 * B + 1], mn_dev=small[34 * B + 1:34 * B + 2])
    for k, (o, n) in FUSION_SMALL.items():
        out[k] = small[o * B:(o + n) * B]
    out["start_t"], out["start_p"] = out["start_t"][:B + 1], out["start_p"][:B + 1]
    return out


def l2norm_fwd(x, y, nrm, *, a3=None, w3=None, yT=None):
    rows, E = y.shape
    _call("spmm_l2norm_fwd", _p(x), _row_stride(x), _p(y), _p(nrm), _p(a3), _p(w3), _p(yT),
               0 if yT is None else _row_stride(yT), rows, E, _st())


def l2norm_bwd(dy, y, nrm, dx, *, gscale=None):
    rows, E = y.shape
    _call("spmm_l2norm_bwd", _p(dy), _p(y), _p(nrm), _p(gscale), _p(dx), rows, E, _st())


def ita_rows(S, SM, dS, *, B, J, alpha, temp, losses, slot, dtemp, nan_flag=None):
    nrows = S.shape[0]
    _call("spmm_ita_rows", _p(S), _p(SM), _row_stride(S), nrows, B, J, _p(alpha), _p(temp), _p(dS), _row_stride(dS),
               dS.shape[1], _p(losses), slot, _p(dtemp), _p(nan_flag), _st())


def sample_neg(S, B, out, *, forced=None, seed=None, salt=0, offset=0):
    _call("spmm_sample_neg", _p(S), _row_stride(S), B, _p(forced), _p(seed), salt, _p(out), offset, _st())


def lm_loss(logits, logits_m, ids, *, nseq, L, V, alpha, ws, losses, slot, dlogits=None, gscale=None):
    _call("spmm_lm_loss", _p(logits), _p(logits_m), _row_stride(logits), _p(ids), nseq, L, V, _p(alpha), _p(ws),
               _p(gscale), _p(dlogits), 0 if dlogits is None else _row_stride(dlogits),
               V if dlogits is None else dlogits.shape[1], _p(losses), slot, _st())


def itm_head(xa, stride_a, xb, stride_b, H, W, bias, *, n, B, losses, slot, logits=None, dxa=None, dxb=None, dW=None,
             db=None, gscale=None):
    do_bwd = dxa is not None
    assert xa.dtype == xb.dtype and xa.dtype in (BF16, torch.float32)
    _call("spmm_itm_head", _p(xa), stride_a, _p(xb), stride_b, H, _p(W), _p(bias), n, B, _p(gscale), _p(losses), slot,
               _p(logits), _p(dxa), _p(dxb), _p(dW), _p(db), int(do_bwd), int(xa.dtype == torch.float32), _st())


def mpm_head(h, Lp, H, w, bias, target, mask, *, B, ws, losses, slot, pred=None, dh=None, dw=None, db=None, gscale=None):
    do_bwd = dh is not None
    _call("spmm_mpm_head", _p(h), Lp, H, _p(w), _p(bias), _p(target), _p(mask), B, _p(ws), _p(gscale), _p(losses), slot,
               _p(pred), _p(dh), _p(dw), _p(db), int(do_bwd), int(h.dtype == torch.float32), _st())


def rows_linear(x, W, bias, out, *, act=0):
    """out[r,n] = act(bias[n] + x[r,:] . W[n,:]) in fp32; x [rows, K] fp32 or bf16 (row stride free), W [N, K] fp32 contiguous."""
    rows, K = x.shape
    N = W.shape[0]
    assert W.dtype == torch.float32 and W.is_contiguous() and W.shape[1] == K and out.dtype == torch.float32 and tuple(out.shape) == (rows, N)
    assert x.dtype in (torch.float32, BF16) and x.stride(1) == 1
    _call("spmm_rows_linear", _p(x), int(x.dtype == BF16), _row_stride(x), _p(W), _p(bias), _p(out), _row_stride(out), rows, N, K, int(act), _st())
    return out


def enqueue(feats, queue, w3, qT, ptr, *, Bloc, advance=True, skip_flag=None):
    n, E = feats.shape
    _call("spmm_enqueue", _p(feats), n, E, _p(queue), queue.shape[1], _p(w3), _p(qT), _row_stride(qT), Bloc, _p(ptr),
               int(advance), _p(skip_flag), _st())


def queue_shadow(queue, w3, qT, *, Bloc):
    E, Q = queue.shape
    _call("spmm_queue_shadow", _p(queue), E, Q, _p(w3), _p(qT), _row_stride(qT), Bloc, _st())


def clamp_scalar(p, lo, hi):
    _call("spmm_clamp_scalar", _p(p), float(lo), float(hi), _st())


def adam_scalars_bytes():
    return lib().cdll.spmm_adam_scalars_bytes()


_sqn_ws = {}


def grad_sqnorm(g, out):
    ws = _sqn_ws.get(g.device)
    if ws is None:
        nbytes = 4096 if _DRY_RUN else lib().cdll.spmm_grad_sqnorm_workspace_bytes()
        ws = _sqn_ws[g.device] = torch.zeros(nbytes // 4, dtype=torch.float32, device=g.device)
    _call("spmm_grad_sqnorm", _p(g), g.numel(), _p(out), _p(ws), _st())


def adamw_step(p, g, m, v, shadow, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.02, normsq, max_norm=5.0, step,
               nan_flag=None, scalars):
    _call("spmm_adamw_step", _p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), _p(lr), beta1, beta2, eps, weight_decay,
               _p(normsq), max_norm, _p(step), _p(nan_flag), _p(scalars), _st())


def ema_update(pm, p, shadow, momentum):
    _call("spmm_ema_update", _p(pm), _p(p), _p(shadow), pm.numel(), float(momentum), _st())


def axpy_scalar(dst, src, *, scale_ptr=None, scale=1.0):
    _call("spmm_axpy_scalar", _p(dst), _p(src), _p(scale_ptr), float(scale), dst.numel(), _st())


def embed_step_ln_fwd(ids32, pos_index, y, *, word, pos, type0, gamma, beta, eps=1e-12, pos_ptr=None):
    """One decode step of BertEmbeddings: y[r] = LN(word[ids[r]] + pos[pos_index] + type0)."""
    rows, H = y.shape
    assert ids32.dtype == torch.int32 and ids32.numel() == rows and y.dtype == BF16
    _call("spmm_embed_step_ln_fwd", _p(ids32), int(pos_index), _p(pos_ptr), _p(word), _p(pos), _p(type0), _p(gamma), _p(beta), _p(y), rows, H,
          float(eps), _st())
    return y


def decode_attn(q, K, V, out, *, nH, Lkv, seq_stride, tok_stride, head_stride=64, anc=None, kv_div=1, group=1, scale=0.125, t_ptr=None, knew=None, vnew=None,
                rowmap=None):
    """Single-query attention over a K/V cache; q/out: [R, >=nH*64] bf16 views, K/V: bf16 views whose element (s, j, h*64+d)
    sits at s*seq_stride + j*tok_stride + h*head_stride + d from their first element (head_stride 64: token-major rows).  knew / vnew ([R, nH*64] bf16 views, self-attention
    only): key / value of the newest position, which the launch also writes into the cache."""
    R = q.shape[0]
    assert q.dtype == BF16 and K.dtype == BF16 and V.dtype == BF16 and out.dtype == BF16
    assert anc is None or (anc.dtype == torch.int32 and anc.shape[0] == R)
    assert (knew is None) == (vnew is None) and (knew is None or (knew.dtype == BF16 and vnew.dtype == BF16 and _row_stride(knew) == _row_stride(vnew)))
    _call("spmm_decode_attn", _p(q), _row_stride(q), _p(K), _p(V), int(seq_stride), int(tok_stride), int(head_stride), _p(anc),
          0 if anc is None else _row_stride(anc), int(kv_div), int(group), _p(out), _row_stride(out), R, nH, Lkv, float(scale), _p(t_ptr),
          _p(knew), _p(vnew), 0 if knew is None else _row_stride(knew), _p(rowmap), _st())
    return out


def beam_step(logits, book, *, t=0, t_ptr=None, t_off=0, anc=None, ids_out=None, parent_out=None, rowmap=None):
    """One position of the batched k-beam search on a decode.BeamBook with int32 state (csrc/decode.hip::beam_step_kernel): updates
    the book (and the ancestry table `anc` of the K/V cache) in place and returns the tokens to feed next, int32 [N*k]."""
    _, k, L = book.tokens.shape
    mol = getattr(book, "mol", None)                      # int32 [N]: the live molecules of a compacted batch (None: all of them)
    N = book.tokens.shape[0] if mol is None else mol.numel()
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.shape[0] == N * k and logits.stride(1) == 1
    assert (mol is None or mol.dtype == torch.int32) and (rowmap is None or (rowmap.dtype == torch.int32 and rowmap.numel() == N * k))
    assert book.tokens.dtype == torch.int32 and book.fin_tok.dtype == torch.int32 and book.fin_len.dtype == torch.int32 and book.fin_n.dtype == torch.int32
    assert book.tokens.is_contiguous() and book.fin_tok.is_contiguous() and book.cur_p.is_contiguous() and book.fin_p.is_contiguous()
    assert anc is None or (anc.dtype == torch.int32 and anc.shape[0] == N * k and anc.stride(1) == 1)
    if ids_out is None:
        ids_out = torch.empty(N * k, dtype=torch.int32, device=logits.device)
    _call("spmm_beam_step", _p(logits), logits.stride(0), N, k, logits.shape[1], L, book.F, int(t), _p(t_ptr), int(t_off), _p(book.tokens),
          _p(book.cur_p), _p(book.fin_p), _p(book.fin_len), _p(book.fin_tok), _p(book.fin_n), _p(book.done), _p(anc),
          0 if anc is None else anc.stride(0), _p(ids_out), _p(parent_out), _p(book.n_done), _p(mol), _p(rowmap), _st())
    return ids_out


def segment_sum_bf16(src, start, lst, out):
    """out[u] = sum_{k in [start[u], start[u+1])} src[list[k]] over rows of W bf16 elements (fp32 accumulation)."""
    U, W = out.shape
    assert src.dtype == BF16 and out.dtype == BF16 and src.shape[1] == W and src.is_contiguous() and out.is_contiguous()
    assert start.dtype == torch.int32 and lst.dtype == torch.int32 and start.numel() == U + 1
    _call("spmm_segment_sum_bf16", _p(src), _p(start), _p(lst), _p(out), U, W, _st())
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Sequences longer than the 256 rows the attention kernels keep on chip (up to the 512 positions of config_bert.json):
# queries and keys are processed in <= 128-long chunks by several launches and merged here.  Forward merges the per-key-chunk
# outputs with their log-sum-exp weights (exact, also with dropout: the mask does not depend on the normalisation); backward
# first sums D[q] = sum_kv P dP over the key chunks (d_mode 1), then runs the gradient launches with that D (d_mode 2).
# Dense layouts only (optionally with shared key/value sources).
# ---------------------------------------------------------------------------------------------------------------------
ATTN_MAXL = 256          # csrc/attention.hip: K / V panel of a head in LDS, a wave's 32 x Lkv score tile in registers


def _chunks(L, step=128):
    return [(o, min(step, L - o)) for o in range(0, L, step)]


def attn_fwd_long(Q, K, V, O, lse, *, nseq, nH, Lq, Lkv, kmask=None, causal_from=None, is_cross=False, dropout_p=0.0, seed=None,
                  salt=0, kv_seq=None):
    if Lq <= ATTN_MAXL and Lkv <= ATTN_MAXL:
        return attn_fwd(Q, K, V, O, lse, nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, kmask=kmask, causal_from=causal_from, is_cross=is_cross,
                        dropout_p=dropout_p, seed=seed, salt=salt, kv_seq=kv_seq)
    dev, H = Q.device, nH * 64
    nsrc = K.shape[0] // Lkv
    ar_q = torch.arange(nseq, dtype=torch.int32, device=dev)
    ar_k = torch.arange(nsrc, dtype=torch.int32, device=dev)
    Ov = O.view(nseq, Lq, nH, 64)
    for qi, (q0, ql) in enumerate(_chunks(Lq)):
        q_row0, q_len = ar_q * Lq + q0, torch.full((nseq,), ql, dtype=torch.int32, device=dev)
        outs, lses = [], []
        for ci, (k0, kl) in enumerate(_chunks(Lkv)):
            Oc = torch.empty(nseq * Lq, H, dtype=BF16, device=dev)
            lc = torch.empty(nseq, nH, ql, dtype=torch.float32, device=dev)
            attn_fwd(Q, K, V, Oc, lc, nseq=nseq, nH=nH, Lq=ql, Lkv=kl, kmask=None if kmask is None else kmask[:, k0:k0 + kl].contiguous(),
                     causal_from=causal_from, is_cross=is_cross, dropout_p=dropout_p, seed=seed, salt=salt + 7919 * (qi * 8 + ci + 1),
                     kv_seq=kv_seq, q_row0=q_row0, q_len=q_len, kv_row0=ar_k * Lkv + k0,
                     kv_len=torch.full((nsrc,), kl, dtype=torch.int32, device=dev), q_off=q0, kv_off=k0)
            outs.append(Oc.view(nseq, Lq, nH, 64)[:, q0:q0 + ql].float())
            lses.append(lc)
        L = torch.stack(lses)                                   # [chunks, nseq, nH, ql]
        tot = torch.logsumexp(L, dim=0)
        acc = sum(o * torch.exp(l - tot).permute(0, 2, 1)[..., None] for o, l in zip(outs, lses))
        Ov[:, q0:q0 + ql] = acc.to(BF16)
        if lse is not None:
            lse[:, :, q0:q0 + ql] = tot
    return O


def attn_bwd_long(Q, K, V, O, lse, dO, dQ, dK, dV, *, nseq, nH, Lq, Lkv, kmask=None, causal_from=None, is_cross=False, dropout_p=0.0,
                  seed=None, salt=0, kv_seq=None):
    """dK / dV: [nseq*Lkv, H] views when kv_seq is given (per query sequence), else [sources*Lkv, H]."""
    if Lq <= ATTN_MAXL and Lkv <= ATTN_MAXL:
        return attn_bwd(Q, K, V, O, lse, dO, dQ, dK, dV, nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, kmask=kmask, causal_from=causal_from,
                        is_cross=is_cross, dropout_p=dropout_p, seed=seed, salt=salt, kv_seq=kv_seq)
    dev, H = Q.device, nH * 64
    nsrc = K.shape[0] // Lkv
    nkv_out = nseq if kv_seq is not None else nsrc
    ar_q = torch.arange(nseq, dtype=torch.int32, device=dev)
    ar_k = torch.arange(nsrc, dtype=torch.int32, device=dev)
    dQ32 = torch.zeros(nseq, Lq, H, dtype=torch.float32, device=dev)
    dK32 = torch.zeros(nkv_out, Lkv, H, dtype=torch.float32, device=dev)
    dV32 = torch.zeros(nkv_out, Lkv, H, dtype=torch.float32, device=dev)
    for qi, (q0, ql) in enumerate(_chunks(Lq)):
        q_row0, q_len = ar_q * Lq + q0, torch.full((nseq,), ql, dtype=torch.int32, device=dev)
        lse_i = lse[:, :, q0:q0 + ql].contiguous()
        calls = []
        for ci, (k0, kl) in enumerate(_chunks(Lkv)):
            calls.append(dict(nseq=nseq, nH=nH, Lq=ql, Lkv=kl, kmask=None if kmask is None else kmask[:, k0:k0 + kl].contiguous(),
                              causal_from=causal_from, is_cross=is_cross, dropout_p=dropout_p, seed=seed,
                              salt=salt + 7919 * (qi * 8 + ci + 1), kv_seq=kv_seq, q_row0=q_row0, q_len=q_len, kv_row0=ar_k * Lkv + k0,
                              kv_len=torch.full((nsrc,), kl, dtype=torch.int32, device=dev), q_off=q0, kv_off=k0))
        dQc = torch.empty(nseq * Lq, H, dtype=BF16, device=dev)
        D = torch.zeros(nseq, nH, ql, dtype=torch.float32, device=dev)
        for (k0, kl), kw in zip(_chunks(Lkv), calls):                      # pass 1: D over all key chunks
            Dc = torch.empty_like(D)
            tmp = torch.empty((nseq if kv_seq is not None else nsrc) * (kl if kv_seq is not None else Lkv), H, dtype=BF16, device=dev)
            attn_bwd(Q, K, V, O, lse_i, dO, dQc, tmp, tmp, d_mode=1, dbuf=Dc, **kw)
            D += Dc
        for (k0, kl), kw in zip(_chunks(Lkv), calls):                      # pass 2: gradients with the complete D
            if kv_seq is not None:                                           # per query sequence, dense within the chunk
                dKc = torch.zeros(nseq * kl, H, dtype=BF16, device=dev); dVc = torch.zeros_like(dKc)
            else:                                                            # written in place at the sources' rows
                dKc = torch.zeros(nsrc * Lkv, H, dtype=BF16, device=dev); dVc = torch.zeros_like(dKc)
            attn_bwd(Q, K, V, O, lse_i, dO, dQc, dKc, dVc, d_mode=2, dbuf=D, **kw)
            dQ32[:, q0:q0 + ql] += dQc.view(nseq, Lq, H)[:, q0:q0 + ql].float()
            if kv_seq is not None:
                dK32[:, k0:k0 + kl] += dKc.view(nseq, kl, H).float(); dV32[:, k0:k0 + kl] += dVc.view(nseq, kl, H).float()
            else:
                dK32[:, k0:k0 + kl] += dKc.view(nsrc, Lkv, H)[:, k0:k0 + kl].float(); dV32[:, k0:k0 + kl] += dVc.view(nsrc, Lkv, H)[:, k0:k0 + kl].float()
    dQ.copy_(dQ32.view(nseq * Lq, H))
    dK.copy_(dK32.view(nkv_out * Lkv, H))
    dV.copy_(dV32.view(nkv_out * Lkv, H))
