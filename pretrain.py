"""SPMM pretraining driver -- the caller of the hot path, with the flags and the inline config dict of the reference's
SPMM_pretrain.py (:41-67) so that a user of the reference finds the same entry point:

  python pretrain.py --data_path ./data/pretrain.txt --vocab_filename ./vocab_bpe_300.txt --output_dir ./Pretrain
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 pretrain.py ...          (one process per GPU, RCCL)
  python pretrain.py --synthetic 256 --tiny --batch_size 4 --max_steps 64                              (BASELINE configs[0] plumbing)

What differs from the reference, on purpose:
  * no pytorch_lightning: `spmm_amd.trainer.Trainer` runs the epochs, logs the five scalars (+ molecules/s) and writes
    Lightning-layout checkpoints `checkpoint_epoch=N.ckpt` every `--ckpt_every` steps (SPMM_pretrain.py:29-37);
  * devices come from the launcher (WORLD_SIZE) instead of the hard-coded ngpu=8 (:13); `loader_len` is the per-rank number
    of batches, which is what `len(data_loader) // torch.cuda.device_count()` (:23) means;
  * RDKit (SMILES canonicalisation + the 53 descriptors, dataset.py:36-40) is not on the target image: `--data_path` lines
    are used as written and the property vectors come from `--property_path` (a [N,53] .npy, already z-scored) -- or
    `--synthetic N` generates the SURVEY.md section 8d recipe (randn PVs, random pieces of the vocabulary)."""
import argparse
import os
import sys
from pathlib import Path

# HIP maps streams onto 4 hardware queues by default and streams sharing a queue serialise; the step uses five
# (main, two side streams, weight gradients, RCCL): ask for 8 before the runtime initialises.
if os.environ.get("SPMM_DIST_BACKEND") != "gloo":          # (not for several gloo ranks on one GPU: see bench.py)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
if int(os.environ.get("WORLD_SIZE", "1")) > 1:             # dmabuf IPC for RCCL; must be in place before the HIP runtime initialises
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class SyntheticPretrainData(torch.utils.data.Dataset):
    """N (property vector, token ids) samples: PV ~ N(0,1)^53 (the pipeline z-scores them, dataset.py:38); ids = [CLS] pieces [SEP]
    with U{4..V-1} pieces and a length in [Lt/2, Lt] -- the token ids the tokenizer would produce (SPMM_models.py:353-357)."""

    def __init__(self, n: int, seq_len: int, vocab_size: int = 300, seed: int = 42):
        g = torch.Generator().manual_seed(seed)
        self.prop = torch.randn(n, 53, generator=g)
        self.ids = torch.zeros(n, seq_len, dtype=torch.long)
        lens = torch.randint(max(seq_len // 2, 3), seq_len + 1, (n,), generator=g)
        for i in range(n):
            L = int(lens[i])
            self.ids[i, 0] = 2
            self.ids[i, 1:L - 1] = torch.randint(4, vocab_size, (L - 2,), generator=g)
            self.ids[i, L - 1] = 3

    def __len__(self):
        return self.prop.shape[0]

    def __getitem__(self, i):
        return self.prop[i], self.ids[i]


class SmilesFileData(torch.utils.data.Dataset):
    """Lines of SMILES (used as written) + a [N,53] float32 .npy of normalised property vectors (dataset.py:13-40 without RDKit)."""

    def __init__(self, data_path: str, property_path: str, limit: int = 50000000):
        import numpy as np
        with open(data_path) as f:
            self.smiles = [l.strip() for _, l in zip(range(limit), f) if l.strip()]
        self.prop = torch.from_numpy(np.load(property_path)).float()
        if self.prop.shape != (len(self.smiles), 53):
            raise ValueError(f"{property_path}: expected [{len(self.smiles)}, 53] property vectors, found {tuple(self.prop.shape)}")

    def __len__(self):
        return len(self.smiles)

    def __getitem__(self, i):
        return self.prop[i], "[CLS]" + self.smiles[i]


def batches(dataset, batch_size: int, rank: int, world: int, device):
    """drop_last batches of the rank's shard (sample r, r+W, ... of every global batch), shuffle=False like SPMM_pretrain.py:18."""
    class _Loader:
        def __len__(self):
            return len(dataset) // (batch_size * world)

        def __iter__(self):
            for b in range(len(self)):
                idx = [b * batch_size * world + rank + world * j for j in range(batch_size)]
                items = [dataset[i] for i in idx]
                prop = torch.stack([p for p, _ in items]).to(device)
                if torch.is_tensor(items[0][1]):
                    ids = torch.stack([t for _, t in items])
                    keep = int((ids != 0).any(0).nonzero().max()) + 1          # padding='longest'
                    ids = ids[:, :keep]
                    n_tokens = int((ids != 0).sum())            # host-side: sizes the packed GEMMs without a device read per step
                    ids = ids.to(device)
                    yield prop, (ids, (ids != 0).long()), {"n_tokens": n_tokens}
                else:
                    yield prop, [t for _, t in items]
    return _Loader()


def main(args, config):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.dry_run:                                 # CPU plumbing check (BASELINE configs[0]): every launch validated, none executed
        from spmm_amd import ops
        ops._DRY_RUN = True
        device = torch.device("cpu")
    else:
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        device = torch.device(f"cuda:{local}")
        if world > 1:
            torch.distributed.init_process_group("nccl", device_id=device)
    torch.manual_seed(args.seed)

    from spmm_amd.model import SPMM
    from spmm_amd.tokenizer import SmilesWordPiece
    from spmm_amd.trainer import Trainer
    from spmm_amd.parallel import broadcast_state_

    if rank == 0:
        print("Creating dataset")
    tokenizer = SmilesWordPiece(args.vocab_filename) if os.path.exists(args.vocab_filename) else None
    if args.synthetic:
        dataset = SyntheticPretrainData(args.synthetic, args.seq_len, seed=args.seed)
    else:
        if tokenizer is None:
            raise SystemExit(f"--vocab_filename {args.vocab_filename} not found (needed to tokenise --data_path)")
        if not args.property_path:
            raise SystemExit("--property_path (a [N,53] .npy of normalised property vectors) is required with --data_path: RDKit is not available here")
        dataset = SmilesFileData(args.data_path, args.property_path)
    loader = batches(dataset, config["batch_size"], rank, world, device)
    if rank == 0:
        print("#data:", len(dataset), "batches per rank:", len(loader), "world:", world)

    model = SPMM(config=config, tokenizer=tokenizer, loader_len=len(loader), device=device)
    if args.checkpoint and not args.resume:          # weights only (SPMM_pretrain.py:24-26); --resume continues optimizer / epoch / seed too
        model.load_checkpoint(torch.load(args.checkpoint, map_location="cpu"), weights_only=True)
    if world > 1:
        broadcast_state_([model.store.flat, model.store.flat_m] + [model.store.buffers[k] for k in ("prop_queue", "text_queue")])
        model.store.refresh_shadows()
        model.engine.invalidate_banks()
    trainer = Trainer(max_epochs=args.epochs if args.epochs else config["schedular"]["epochs"], output_dir=args.output_dir,
                      every_n_train_steps=args.ckpt_every, log_every_n_steps=args.log_every, max_steps=args.max_steps)
    trainer.fit(model, loader, None, ckpt_path=args.checkpoint if (args.checkpoint and args.resume) else None)
    if world > 1:
        torch.distributed.destroy_process_group()
    return trainer


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    # the reference's flags (SPMM_pretrain.py:41-48)
    parser.add_argument("--checkpoint", default="")
    parser.add_argument("--data_path", default="./data/chemformer_parsed2_shuffle.txt")
    parser.add_argument("--resume", default=False, type=lambda s: str(s).lower() in ("1", "true", "yes"))
    parser.add_argument("--output_dir", default="./Pretrain")
    parser.add_argument("--vocab_filename", default="./vocab_bpe_300.txt")
    parser.add_argument("--seed", default=42, type=int)
    # additions
    parser.add_argument("--property_path", default="", help="[N,53] .npy of normalised property vectors for --data_path")
    parser.add_argument("--synthetic", type=int, default=0, help="train on N synthetic (PV, token ids) samples instead of --data_path")
    parser.add_argument("--seq_len", type=int, default=128, help="longest synthetic sequence")
    parser.add_argument("--tiny", action="store_true", help="2-layer / 128-d encoders (configs/config_bert_tiny.json), embed_dim 64, queue 16")
    parser.add_argument("--batch_size", type=int, default=0, help="per-GPU batch (default: the reference's 96)")
    parser.add_argument("--epochs", type=int, default=0)
    parser.add_argument("--max_steps", type=int, default=None)
    parser.add_argument("--ckpt_every", type=int, default=10000)
    parser.add_argument("--log_every", type=int, default=50)
    parser.add_argument("--dry_run", action="store_true", help="no GPU: validate every kernel call against the C ABI without launching")
    args = parser.parse_args()

    cfg_dir = os.path.join(ROOT, "configs")
    pretrain_config = {                                # SPMM_pretrain.py:51-65, key for key
        "property_width": 768,
        "embed_dim": 256,
        "batch_size": 96,
        "temp": 0.07,
        "mlm_probability": 0.15,
        "queue_size": 36864,
        "momentum": 0.995,
        "alpha": 0.4,
        "bert_config_text": os.path.join(cfg_dir, "config_bert.json"),
        "bert_config_property": os.path.join(cfg_dir, "config_bert_property.json"),
        "schedular": {"sched": "cosine", "lr": 5e-5, "epochs": 30, "min_lr": 1e-5,
                      "decay_rate": 1, "warmup_lr": 5e-5, "warmup_epochs": 20, "cooldown_epochs": 0},
        "optimizer": {"opt": "adamW", "lr": 5e-5, "weight_decay": 0.02},
    }
    if args.tiny:
        pretrain_config.update(embed_dim=64, queue_size=16, bert_config_text=os.path.join(cfg_dir, "config_bert_tiny.json"),
                               bert_config_property=os.path.join(cfg_dir, "config_bert_property_tiny.json"))
    if args.batch_size:
        pretrain_config["batch_size"] = args.batch_size
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    main(args, pretrain_config)
