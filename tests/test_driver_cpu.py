"""BASELINE configs[0]: the pretrain driver on 256 synthetic SMILES+PV samples, 2-layer / 128-d encoders, batch 4, on a CPU-only
box.  The product has no CPU arithmetic path, so this is the PLUMBING check the config asks for: the driver runs in
`--dry_run` mode, where every kernel call of every step is validated against the C ABI prototypes (argument count and types)
without being launched -- data -> batches -> training_step -> schedule -> Lightning-layout checkpoint."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pretrain_driver_plumbing_dry_run(tmp_path):
    out = tmp_path / "Pretrain"
    cmd = [sys.executable, os.path.join(ROOT, "pretrain.py"), "--synthetic", "256", "--tiny", "--batch_size", "4", "--seq_len", "16",
           "--max_steps", "64", "--dry_run", "--output_dir", str(out), "--log_every", "32", "--ckpt_every", "40"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "#data: 256 batches per rank: 64" in r.stdout and "step 64:" in r.stdout and "mean loss" in r.stdout
    ck = torch.load(out / "checkpoint_epoch=0.ckpt", map_location="cpu")
    assert {"state_dict", "epoch", "global_step", "optimizer_states", "rng_seed"} <= set(ck)
    assert ck["global_step"] == 64 and ck["optimizer_states"][0]["step_count"] == 0      # dry run: no kernel advanced the device counter
    sd = ck["state_dict"]
    assert "text_encoder.bert.encoder.layer.1.crossattention.self.key.weight" in sd and "prop_queue" in sd and "text_encoder_m.cls.predictions.decoder.weight" in sd
    assert sd["property_proj.weight"].shape == (64, 128)


def test_driver_flags_match_the_reference_entry_point():
    """SPMM_pretrain.py:41-48: --checkpoint --data_path --resume --output_dir --vocab_filename --seed, same defaults."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "pretrain.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0
    for flag in ("--checkpoint", "--data_path", "--resume", "--output_dir", "--vocab_filename", "--seed"):
        assert flag in r.stdout
    src = open(os.path.join(ROOT, "pretrain.py")).read()
    for key in ("'property_width'", "'embed_dim'", "'batch_size'", "'temp'", "'mlm_probability'", "'queue_size'", "'momentum'", "'alpha'",
                "'bert_config_text'", "'bert_config_property'", "'schedular'", "'optimizer'"):
        assert key.replace("'", '"') in src
