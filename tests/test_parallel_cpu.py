"""World-size-2 gloo test of the data-parallel exchange (spmm_amd/parallel.py): feature all-gather + bucketed gradient
averaging reproduce the single-process result on the global batch.  The oracle supplies features/gradients on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import spmm_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spmm_amd import parallel
    torch.manual_seed(0)
    cfg = O.tiny_cfg()
    for c in (cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sd = O.closed_form_state_dict(cfg)
    names = O.trainable_names(cfg)
    for n in names:
        sd[n].requires_grad_(True)
    O._finish_tied(sd)
    Bg, Lt = 8, 16
    prop, ids, mask = O.synthetic_batch(Bg, Lt, seed=5)
    mpm = (torch.arange(Bg * 53).reshape(Bg, 53) % 2).float()
    bl = Bg // world
    sl = slice(rank * bl, (rank + 1) * bl)
    neg = (torch.arange(bl).roll(1), torch.arange(bl).roll(1))
    losses = O.spmm_forward(sd, cfg, prop[sl], ids[sl], mask[sl], 0.2, mpm_mask=mpm[sl], neg_idx=neg, train=True,
                            gather=parallel.all_gather_features)
    sum(losses).backward()
    flat = torch.cat([sd[n].grad.reshape(-1) if sd[n].grad is not None else torch.zeros(sd[n].numel()) for n in names])
    local = flat.clone()
    parallel.allreduce_mean_(flat, bucket_elems=50_000)             # several buckets
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    parallel.assert_replicas_identical(sd["prop_queue"], "prop_queue")
    parallel.assert_replicas_identical(sd["queue_ptr"], "queue_ptr")
    if rank == 0:
        # plain numpy (pickled by value): torch tensors would travel as shared-memory handles that die with the worker
        q.put((flat.numpy(), torch.stack(gathered).mean(0).numpy(), sd["prop_queue"].detach().numpy().copy(), int(sd["queue_ptr"])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_grad_average():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat, ref_mean, queue, ptr = q.get(timeout=120)
    flat, ref_mean, queue = torch.from_numpy(flat), torch.from_numpy(ref_mean), torch.from_numpy(queue)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.allclose(flat, ref_mean, atol=1e-7)
    # the queue received the features of the GLOBAL batch (8 columns from ptr 0), identically on every rank
    assert ptr == 8 % 16
    cfg = O.tiny_cfg()
    init = O.closed_form_state_dict(cfg)["prop_queue"]
    assert not torch.allclose(queue[:, :8], init[:, :8]) and torch.equal(queue[:, 8:], init[:, 8:])


def _overlap_worker(rank, world, port, q, wire="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPMM_DRY_RUN="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spmm_amd import parallel
    from spmm_amd.config import tiny_config, state_spec
    from spmm_amd.params import _layout_order, ALIGN
    import math
    cfg = tiny_config()
    spec = state_spec(cfg)
    order = _layout_order(spec)
    shape = {n: s for n, s, _ in spec}
    offset, off = {}, 0
    for n in order:                                    # the arithmetic of ParamStore.__init__ (needs no device)
        offset[n] = off
        off += ((int(math.prod(shape[n])) if shape[n] else 1) + ALIGN - 1) // ALIGN * ALIGN
    sync = parallel.OverlappedGradSync(order, offset, off, wire=wire)
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(off, generator=g)
    local = grad.clone()
    sync.begin(grad)
    nt, npv, f = cfg.text.num_hidden_layers, cfg.prop.num_hidden_layers, cfg.text.fusion_layer
    called = []
    for pfx, layers in (("text_encoder.bert.", range(nt - 1, f - 1, -1)), ("text_encoder.bert.", range(f - 1, -1, -1)),
                        ("property_encoder.", range(npv - 1, -1, -1))):      # the order backward finishes layers in
        for i in layers:
            sync.layer_done(f"{pfx}encoder.layer.{i}.")
            called.append(f"{pfx}encoder.layer.{i}.")
    covered = sum(hi - lo for lo, hi in sync._done)
    sync.layer_done("text_encoder_m.bert.encoder.layer.0.")                   # unknown prefixes are ignored
    sync.finish()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    if rank == 0:
        q.put((grad.numpy(), torch.stack(gathered).mean(0).numpy(), covered, off, len(called)))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_grad_sync_two_ranks():
    """Per-layer asynchronous all-reduces + the final sweep over what no layer covered average the whole arena exactly once."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, want, covered, total, nl = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.allclose(torch.from_numpy(got), torch.from_numpy(want), atol=1e-7)
    assert 0.5 * total < covered < total and nl >= 3          # layers carry most of the bytes, the sweep the rest


def test_overlapped_grad_sync_bf16_reduce_scatter_all_gather_two_ranks():
    """SPMM_GRAD_WIRE=bf16: every slice travels as bf16 through reduce-scatter + all-gather (half the bytes per link).  Stated
    tolerance against the fp32 mean: two bf16 roundings (the cast and the sum), |err| <= 2^-7 * (|g0| + |g1|) / 2 per element."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q, "bf16")) for r in range(world)]
    for p in procs:
        p.start()
    got, want, covered, total, nl = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got, want = torch.from_numpy(got), torch.from_numpy(want)
    g = [torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    bound = 2.0 ** -7 * (g[0].abs() + g[1].abs()) / 2 + 1e-6
    assert ((got - want).abs() <= bound).all(), float(((got - want).abs() - bound).max())
    assert (got - want).abs().max() > 1e-5                    # it really went through bf16
