"""Minimal trainer with the surface SPMM_pretrain.py uses from pytorch_lightning (`Trainer(max_epochs=...).fit(model, loader,
ckpt_path=...)` + `ModelCheckpoint(dirpath, every_n_train_steps)`, SPMM_pretrain.py:29-37) -- pytorch_lightning is not
installed on the target image, and the step itself (forward, backward, gradient exchange, clip, AdamW, EMA) is one fused
sequence of HIP launches inside `SPMM.training_step`, so there is nothing for a generic trainer to wrap.

One process per GPU (torch.distributed over RCCL, started by torch.distributed.run); rank r trains on samples r, r+W, ...
of every global batch, which is what Lightning's injected DistributedSampler does for the reference."""
from __future__ import annotations

import os
import time
from typing import Iterable, Optional

import torch


def _dist():
    d = torch.distributed
    return d if (d.is_available() and d.is_initialized()) else None


class Trainer:
    def __init__(self, max_epochs: int = 30, output_dir: str = "./Pretrain", every_n_train_steps: int = 10000, log_every_n_steps: int = 50,
                 max_steps: Optional[int] = None, filename: str = "checkpoint_{epoch}", quiet: bool = False):
        self.max_epochs, self.output_dir, self.every_n = max_epochs, output_dir, every_n_train_steps
        self.log_every, self.max_steps, self.filename, self.quiet = log_every_n_steps, max_steps, filename, quiet
        d = _dist()
        self.rank = d.get_rank() if d else 0
        self.world = d.get_world_size() if d else 1
        self.global_step = 0
        self.history = []                 # (global_step, lr, [4 losses], molecules/s) at every logging point (rank 0)

    def _ckpt_path(self, epoch: int) -> str:
        # Lightning renders filename='checkpoint_{epoch}' as 'checkpoint_epoch=3.ckpt'
        return os.path.join(self.output_dir, self.filename.replace("{epoch}", f"epoch={epoch}") + ".ckpt")

    def save(self, model, epoch: int) -> Optional[str]:
        if self.rank != 0:
            return None
        os.makedirs(self.output_dir, exist_ok=True)
        path = self._ckpt_path(epoch)
        model.global_step = self.global_step
        model.save_checkpoint(path)
        return path

    def fit(self, model, train_loader: Iterable, val_loader=None, ckpt_path: Optional[str] = None):
        """`train_loader` yields (properties [B,53], text) with text = list of '[CLS]'+SMILES strings (tokenised by
        model.tokenizer, SPMM_models.py:353) or an (input_ids, attention_mask) tensor pair; an optional third element is a dict
        of recorded random draws (`mpm_mask`, `neg_idx`) replayed by parity tests."""
        start_epoch, skip = 0, 0
        if ckpt_path:
            model.load_checkpoint(ckpt_path)
            start_epoch, self.global_step = int(model.current_epoch), int(getattr(model, "global_step", 0))
            if hasattr(train_loader, "__len__") and len(train_loader) > 0:
                skip = self.global_step - start_epoch * len(train_loader)      # batches of the interrupted epoch already trained on
                if skip >= len(train_loader):
                    start_epoch, skip = start_epoch + 1, 0
        model.train()
        model.global_rank = self.rank
        model.optimizers()
        done = False
        for epoch in range(start_epoch, self.max_epochs):
            model.current_epoch = epoch
            t0, seen = time.perf_counter(), 0
            for batch_idx, batch in enumerate(train_loader):
                if skip > 0:                                     # resumed mid-epoch: batch_idx (alpha ramp, lr cadence) stays aligned
                    skip -= 1
                    continue
                out = model.training_step(batch, batch_idx)
                self.global_step += 1
                seen += int(batch[0].shape[0]) * self.world
                if self.log_every and (self.global_step % self.log_every == 0):
                    vals = [float(v) for v in out.cpu()]                     # the only host read of the window
                    dt = time.perf_counter() - t0
                    rate = seen / dt if dt > 0 else 0.0
                    t0, seen = time.perf_counter(), 0
                    if self.rank == 0:
                        lr = model.optimizers().param_groups[0]["lr"]
                        self.history.append((self.global_step, lr, vals, rate))
                        if not self.quiet:
                            print(f"epoch {epoch} step {self.global_step}: lr {lr:.3e} loss_mlm {vals[0]:.4f} loss_mpm {vals[1]:.4f} "
                                  f"loss_ita {vals[2]:.4f} loss_itm {vals[3]:.4f} | {rate:.1f} molecules/s", flush=True)
                if self.every_n and self.global_step % self.every_n == 0:
                    self.save(model, epoch)
                if self.max_steps is not None and self.global_step >= self.max_steps:
                    done = True
                    break
            model.on_train_epoch_end()
            if done:
                break
        self.save(model, model.current_epoch)
        return model
