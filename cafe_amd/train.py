"""Thin training driver: `python -m cafe_amd.train --model dlrm --workload cfg2 --num-batches 200`.

What dlrm_s_pytorch.py / wdl.py / dcn.py do around the embedding layer, reduced to what reporting a loss needs
(SURVEY.md 8(f) rank 1): build the model (sizes as dlrm_s_pytorch.py:1237-1242), iterate batches, in the
reference's order per step (:1640-1658), print the running loss every --print-freq steps (:1675-1712), optionally
evaluate on the test split (loss and accuracy, :1717-1760 without the AUC pass) and save/load a checkpoint that
includes the sketch (the reference's `sketch_buffer` is dead, :443; lib.update() on load, :1510-1511).

Data: files in the reference's binary format (cafe_amd.data) or, by default, synthetic batches of the named
workload (seeded Zipf ids, U[0,1) dense, Bernoulli(0.25) labels).  Flag names are the reference's where one exists."""
import argparse
import json
import sys
import time

import numpy as np
import torch

from . import config as cfg
from .ctr_models import CafeCTRModel
from .data import BinaryCTRDataset


def parse(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--model", choices=("dlrm", "wdl", "dcn"), default="dlrm")
    p.add_argument("--workload", default="cfg2", help="cfg1..cfg5 (cafe_amd.config) when no data files are given")
    p.add_argument("--arch-sparse-feature-size", type=int, default=None)
    p.add_argument("--arch-mlp-bot", default=None, help="e.g. 13-512-256-64-16")
    p.add_argument("--arch-mlp-top", default=None, help="e.g. 512-256-1 (the input width is derived)")
    p.add_argument("--compress-rate", type=float, default=None)
    p.add_argument("--hash-rate", type=float, default=None)
    p.add_argument("--sketch-threshold", type=int, default=None)
    p.add_argument("--adjust-threshold", type=int, default=1)
    p.add_argument("--sketch-alpha", type=float, default=1.0000005)
    p.add_argument("--mini-batch-size", type=int, default=None)
    p.add_argument("--learning-rate", type=float, default=0.1)
    p.add_argument("--num-batches", type=int, default=100)
    p.add_argument("--nepochs", type=int, default=1)
    p.add_argument("--numpy-rand-seed", type=int, default=123)
    p.add_argument("--print-freq", type=int, default=20)
    p.add_argument("--test-freq", type=int, default=-1)
    p.add_argument("--test-num-batches", type=int, default=20)
    p.add_argument("--fused-insert", action="store_true",
                   help="one sketch insert per step instead of one per field (decay/reset evaluated per step)")
    p.add_argument("--cat-path"); p.add_argument("--dense-path"); p.add_argument("--label-path"); p.add_argument("--count-path")
    p.add_argument("--num-fields", type=int, default=26)
    p.add_argument("--save-model", default="")
    p.add_argument("--load-model", default="")
    return p.parse_args(argv)


def synthetic_batches(w, dense_dim, num_batches, seed, device, start=0):
    rng = np.random.default_rng(seed)
    for k in range(num_batches):
        id_seed = int(rng.integers(1 << 31))
        if k < start:           # resumed run: the generator advances exactly as if the batch had been produced
            if dense_dim:
                rng.random((w.batch, dense_dim), dtype=np.float32)
            rng.random((w.batch, 1))
            continue
        ids = torch.from_numpy(w.make_ids(id_seed, 1)[0]).to(device)
        x = torch.from_numpy(rng.random((w.batch, dense_dim), dtype=np.float32)).to(device) if dense_dim else None
        y = torch.from_numpy((rng.random((w.batch, 1)) < 0.25).astype(np.float32)).to(device)
        yield x, ids, y


def save_checkpoint(path, model, step):
    """Model parameters + tables + the complete sketch state (every slot, scalars, free list)."""
    emb = model.embedding
    torch.save({"step": int(step), "head": model.head.state_dict(), "plain": model.plain.state_dict(),
                "weight_high": emb.weight_high.cpu(),
                "weight_hash": [w.cpu() for w in emb.weight_hash], "sketch": emb.sketch.state_tensors()}, path)


def load_checkpoint(path, model):
    ck = torch.load(path, map_location="cpu", weights_only=True)
    emb = model.embedding
    model.head.load_state_dict(ck["head"])
    if "plain" in ck:
        model.plain.load_state_dict(ck["plain"])
    emb.weight_high.copy_(ck["weight_high"])
    for w, src in zip(emb.weight_hash, ck["weight_hash"]):
        w.copy_(src)
    emb.sketch.load_state_tensors(ck["sketch"])
    return int(ck["step"])


def evaluate(model, batches):
    tot, n, hit = 0.0, 0, 0
    with torch.no_grad():
        for x, ids, y in batches:
            p = model(x, ids)
            tot += float(torch.nn.functional.binary_cross_entropy(p, y, reduction="sum"))
            hit += int(((p > 0.5).float() == y).sum())
            n += y.numel()
    return tot / max(n, 1), hit / max(n, 1)


def main(argv=None):
    a = parse(argv)
    if not torch.cuda.is_available():
        sys.exit("cafe_amd.train needs a GPU: there is no CPU path")
    torch.manual_seed(a.numpy_rand_seed)
    dev = torch.device("cuda", 0)
    files = a.cat_path is not None
    if files:
        dense_dim = 13 if a.dense_path else 0
        train = BinaryCTRDataset(a.cat_path, a.label_path, a.count_path, a.dense_path, a.num_fields, 13, "train")
        test = BinaryCTRDataset(a.cat_path, a.label_path, a.count_path, a.dense_path, a.num_fields, 13, "test")
        counts, dim = train.counts, a.arch_sparse_feature_size or 16
        batch = a.mini_batch_size or 128
        cr, hr, thr = a.compress_rate or 0.001, a.hash_rate or 0.3, a.sketch_threshold or 500
        w = None
    else:
        w = cfg.workload(a.workload, a.mini_batch_size)
        counts, dim, batch = w.counts, a.arch_sparse_feature_size or w.dim, w.batch
        dense_dim = 0 if a.workload == "cfg5" else 13
        cr, hr = a.compress_rate or w.compress_rate, a.hash_rate or w.hash_rate
        thr = a.sketch_threshold or w.threshold
    ln_bot = [int(v) for v in a.arch_mlp_bot.split("-")] if a.arch_mlp_bot else None
    ln_top = [0] + [int(v) for v in a.arch_mlp_top.split("-")] if a.arch_mlp_top else None
    model = CafeCTRModel(a.model, counts, dim, dense_dim=dense_dim, compress_rate=cr, hash_rate=hr, threshold=thr,
                         adjust_thres=a.adjust_threshold, alpha=a.sketch_alpha, ln_bot=ln_bot, ln_top=ln_top, device=dev,
                         seed=a.numpy_rand_seed, per_field_calls=not a.fused_insert)
    opt = torch.optim.SGD(model.dense_parameters(), lr=a.learning_rate)
    step0 = load_checkpoint(a.load_model, model) if a.load_model else 0
    print(json.dumps({"model": a.model, "fields": len(counts), "dim": dim, "batch": batch, "hotn": model.hotn,
                      "cold_rows": int(sum(model.hash_sizes)), "ids": int(sum(counts)), "start_step": step0}), flush=True)
    step, run_loss, run_n, t0 = step0, None, 0, time.time()
    # a resumed run continues at the saved iteration of the saved epoch (dlrm_s_pytorch.py:1512-1525 ld_k, ld_j)
    per_epoch = a.num_batches if a.num_batches else ((len(train) + batch - 1) // batch if files else 0)
    epoch0, skip = (step0 // per_epoch, step0 % per_epoch) if per_epoch else (0, 0)
    for epoch in range(epoch0, a.nepochs):
        first = skip if epoch == epoch0 else 0
        it = (train.batches(batch, dev, start=first) if files
              else synthetic_batches(w, dense_dim, a.num_batches, a.numpy_rand_seed + epoch, dev, start=first))
        for j, (x, ids, y) in enumerate(it, start=first):
            if a.num_batches and j >= a.num_batches:
                break
            loss = model.train_step(x, ids, y, opt, a.learning_rate)
            run_loss = loss if run_loss is None else run_loss + loss
            run_n += 1
            step += 1
            if step % a.print_freq == 0:
                torch.cuda.synchronize()
                dt = time.time() - t0
                print(f"epoch {epoch} step {step}: loss {float(run_loss) / run_n:.6f}  {dt * 1e3 / run_n:.2f} ms/it", flush=True)
                run_loss, run_n, t0 = None, 0, time.time()
            if a.test_freq > 0 and step % a.test_freq == 0:
                tb = (test.batches(batch, dev) if files
                      else synthetic_batches(w, dense_dim, a.test_num_batches, 987654321, dev))
                tl, acc = evaluate(model, (b for k, b in enumerate(tb) if k < a.test_num_batches))
                print(f"step {step}: test loss {tl:.6f} accuracy {acc:.4f}", flush=True)
    if a.save_model:
        save_checkpoint(a.save_model, model, step)
    return model


if __name__ == "__main__":
    main()
