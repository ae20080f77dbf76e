#!/usr/bin/env python3
"""Generate tests/golden/loader_kaggle.npz from the REFERENCE's own data loader: the unmodified
/root/reference/dlrm_data_pytorch.py (KaggleDataset, :315-392, behind torch's DataLoader with
collate_wrapper_criteo_offset2, :409-421, exactly as make_criteo_data_and_loaders builds them, :655-690) on a tiny
synthetic set of the binary files the reference memory-maps (:637-649).

Runs only in the build container.  The fixture is data only: the raw arrays the .bin files hold and, per batch, what
the reference's loader yields (X_int, lS_o, lS_i, T).  The reference hard-codes the Kaggle row count in its train/test
split (45840617 * 6 // 7, :342); on a smaller file its 'train' split is therefore the whole file, which is what is
recorded here; cafe_amd.data splits at 6/7 of the actual row count (tests/test_data_loader.py compares accordingly).

    python tests/golden/make_loader_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import torch
    sys.path.insert(0, "/root/reference")
    import dlrm_data_pytorch as ref                     # the reference module, unmodified

    rng = np.random.default_rng(2024)
    S, F, B = 896, 26, 128
    counts = np.array([1460, 583, 9000, 2200, 305, 24, 1251, 633, 3, 931, 568, 8351, 3194, 27, 1499, 5461, 10, 565, 217, 4,
                       7046, 18, 15, 2861, 105, 1425], np.int64)
    cat = np.stack([(rng.zipf(1.1, size=S) - 1) % c for c in counts], axis=1).astype(np.int32)
    dense = rng.random((S, 13), dtype=np.float32)
    label = (rng.random(S) < 0.25).astype(np.int32)
    count_file = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)        # 27 running sums (:645-649)
    new_count = np.zeros(26)
    for i in range(26):
        new_count[i] = count_file[i + 1] - count_file[i]                          # :647-649
    train = ref.KaggleDataset(new_count, cat, dense, label, "train", False, False, 0.001, None, 0.3)
    loader = torch.utils.data.DataLoader(train, batch_size=B, shuffle=False, num_workers=0,
                                         collate_fn=ref.collate_wrapper_criteo_offset2, pin_memory=False, drop_last=False)
    arrs = dict(cat=cat, dense=dense, label=label, count_file=count_file, batch=B)
    nb = 0
    for k, (X, lS_o, lS_i, T) in enumerate(loader):
        arrs[f"X_{k}"] = X.numpy()
        arrs[f"lS_o_{k}"] = lS_o.numpy()
        arrs[f"lS_i_{k}"] = lS_i.numpy()
        arrs[f"T_{k}"] = T.numpy()
        nb += 1
    arrs["num_batches"] = nb
    out = os.path.join(HERE, "loader_kaggle.npz")
    np.savez_compressed(out, **arrs)
    print(f"loader_kaggle: {nb} batches, {os.path.getsize(out) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
