"""Data path of the hot path's callers: the reference's on-disk format, read in batches
whose ids go to HBM once and stay there.

Format (dlrm_data_pytorch.py:637-649 Kaggle, :572-583 Avazu; the same layout per day for Terabyte, :214-226):

    cat    int32   [S, F]   row-major, per-field LOCAL ids (F = 26, Avazu 22)
    dense  float32 [S, 13]  (absent for Avazu)
    label  int32   [S]
    count  int32   [F + 1]  running sum of the field cardinalities: n_k = count[k + 1] - count[k]   (:646-649)

Split: the first `train_len` samples train, the rest test (Kaggle 6/7 of the file, :342-350; Avazu S - 4 218 938,
:105-111).  Collation is the reference's one-index-per-bag form (collate_wrapper_criteo_offset2/3, :395-421):
lS_i = X_cat[:, k] per field, offsets = arange(B); here that is ids[F, B] int64 on the device and the offsets are
implicit.  The last batch is short (drop_last=False, :601, :672).

The reference converts every sample through Python tuples and torch.tensor per batch on the host; here a batch is
one slice of the memory map, copied through a pinned buffer, transposed on the device."""
import os

import numpy as np
import torch


def field_counts(count_path, num_fields):
    """dlrm_data_pytorch.py:580-583 / :646-649."""
    c = np.fromfile(count_path, dtype=np.int32)
    if c.size != num_fields + 1:
        raise ValueError(f"{count_path}: expected {num_fields + 1} int32 values, found {c.size}")
    n = np.diff(c.astype(np.int64))
    if (n <= 0).any():
        raise ValueError(f"{count_path}: counts must be a strictly increasing running sum")
    return [int(x) for x in n]


class BinaryCTRDataset:
    def __init__(self, cat_path, label_path, count_path, dense_path=None, num_fields=26, dense_dim=13, split="train",
                 train_len=None):
        self.F = int(num_fields)
        self.dense_dim = int(dense_dim) if dense_path else 0
        self.counts = field_counts(count_path, self.F)
        nbytes = os.path.getsize(cat_path)
        if nbytes % (4 * self.F):
            raise ValueError(f"{cat_path}: size {nbytes} is not a whole number of int32[{self.F}] samples")
        S = nbytes // (4 * self.F)
        cat = np.memmap(cat_path, dtype=np.int32, mode="r", shape=(S, self.F))
        lab = np.memmap(label_path, dtype=np.int32, mode="r", shape=(S,))
        den = np.memmap(dense_path, dtype=np.float32, mode="r", shape=(S, self.dense_dim)) if dense_path else None
        if train_len is None:
            train_len = S * 6 // 7                                  # :342
        if not 0 <= train_len <= S:
            raise ValueError("train_len outside the file")
        sl = slice(0, train_len) if split == "train" else slice(train_len, S)
        if split not in ("train", "test"):
            raise ValueError(split)
        self.cat, self.label, self.dense = cat[sl], lab[sl], (den[sl] if den is not None else None)

    def __len__(self):
        return len(self.label)

    def num_batches(self, batch_size):
        return (len(self) + batch_size - 1) // batch_size

    def batches(self, batch_size, device="cuda", start=0):
        """Yields (dense [B, 13] float32 | None, ids [F, B] int64, labels [B, 1] float32) on `device`."""
        dev = torch.device(device)
        pin = dev.type == "cuda"
        B = int(batch_size)
        stage_c = torch.empty((B, self.F), dtype=torch.int32, pin_memory=pin)
        stage_l = torch.empty((B,), dtype=torch.int32, pin_memory=pin)
        stage_d = torch.empty((B, self.dense_dim), dtype=torch.float32, pin_memory=pin) if self.dense is not None else None
        done = None
        for lo in range(start * B, len(self), B):
            hi = min(lo + B, len(self))
            n = hi - lo
            if done is not None:
                done.synchronize()                                   # the staging buffers are free again
            stage_c[:n].numpy()[...] = self.cat[lo:hi]
            stage_l[:n].numpy()[...] = self.label[lo:hi]
            ids = stage_c[:n].to(dev, non_blocking=True).t().contiguous().to(torch.int64)
            y = stage_l[:n].to(dev, non_blocking=True).to(torch.float32).view(-1, 1)
            x = None
            if stage_d is not None:
                stage_d[:n].numpy()[...] = self.dense[lo:hi]
                x = stage_d[:n].to(dev, non_blocking=True)
                if not pin:
                    x = x.clone()
            if pin:
                done = torch.cuda.Event()
                done.record()
            yield x, ids, y


def write_synthetic(directory, counts, samples, dense_dim=13, seed=123, zipf_a=1.05, click_rate=0.25):
    """Synthetic files in the reference's format (seeded Zipf ids, U[0,1) dense features, Bernoulli labels: SURVEY
    8(d)); returns the paths as keyword arguments for BinaryCTRDataset."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(seed)
    F = len(counts)
    cat = np.empty((samples, F), dtype=np.int32)
    for k, n in enumerate(counts):
        loc = (rng.zipf(zipf_a, size=samples) - 1) % n
        if k == 0 and n > 1:
            loc = np.maximum(loc, 1)                                # global id 0 means "empty slot" 
        cat[:, k] = loc
    paths = {"cat_path": os.path.join(directory, "sparse.bin"), "label_path": os.path.join(directory, "label.bin"),
             "count_path": os.path.join(directory, "count.bin")}
    cat.tofile(paths["cat_path"])
    (rng.random(samples) < click_rate).astype(np.int32).tofile(paths["label_path"])
    np.concatenate([[0], np.cumsum(counts)]).astype(np.int32).tofile(paths["count_path"])
    if dense_dim:
        paths["dense_path"] = os.path.join(directory, "dense.bin")
        rng.random((samples, dense_dim), dtype=np.float32).tofile(paths["dense_path"])
    return paths
