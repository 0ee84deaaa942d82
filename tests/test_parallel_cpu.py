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
