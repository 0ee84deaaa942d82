"""smoke(): one tiny pretraining step on cuda:0, checked against the CPU oracle (oracle/ is test infrastructure and is
imported here only as the checker)."""
import torch


def smoke_step():
    import spmm_oracle as O            # checker only
    from .config import tiny_config
    from .model import SPMM

    torch.manual_seed(0)
    cfg = tiny_config()
    ocfg = O.tiny_cfg()
    sd = O.closed_form_state_dict(ocfg)
    B, Lt = 4, 16
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=7)
    mpm = (torch.arange(B * 53).reshape(B, 53) % 3 == 0).float()
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(2))
    model = SPMM(config=None, spmm_config=cfg, no_train=False)
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    model.eval()
    losses = model(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=(neg[0].cuda(), neg[1].cuda()))
    got = torch.stack([l.detach() for l in losses]).cpu()
    with torch.no_grad():
        ref = torch.stack(list(O.spmm_forward(sd, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)))
    err = (got - ref).abs().max().item()
    print(f"smoke: losses hip={got.tolist()} oracle={ref.tolist()} max|diff|={err:.3e}")
    if not torch.isfinite(got).all() or err > 5e-2:
        raise RuntimeError(f"smoke step deviates from the oracle: {got.tolist()} vs {ref.tolist()}")
    losses[0].sum().backward() if False else None
    return got, ref
