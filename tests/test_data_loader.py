"""cafe_amd.data: the reference's binary dataset format (dlrm_data_pytorch.py:637-649, :572-583) read back batch by
batch -- CPU test of the host logic (the same code moves batches to HBM on a GPU box)."""
import numpy as np
import pytest
import torch

from cafe_amd import data as cd


@pytest.mark.parametrize("dense_dim,F", [(13, 26), (0, 22)])
def test_roundtrip_and_collation(tmp_path, dense_dim, F):
    rng = np.random.default_rng(0)
    counts = [int(c) for c in rng.integers(3, 5000, size=F)]
    S, B = 1000, 96
    paths = cd.write_synthetic(str(tmp_path), counts, S, dense_dim=dense_dim, seed=4)
    raw = np.fromfile(paths["cat_path"], dtype=np.int32).reshape(S, F)
    lab = np.fromfile(paths["label_path"], dtype=np.int32)
    den = np.fromfile(paths["dense_path"], dtype=np.float32).reshape(S, dense_dim) if dense_dim else None
    assert (raw >= 0).all() and (raw < np.asarray(counts)[None, :]).all() and raw[:, 0].min() >= 1
    train = cd.BinaryCTRDataset(num_fields=F, split="train", **paths)
    test = cd.BinaryCTRDataset(num_fields=F, split="test", **paths)
    assert train.counts == counts
    tl = S * 6 // 7                                               # dlrm_data_pytorch.py:342
    assert len(train) == tl and len(test) == S - tl
    seen = 0
    for k, (x, ids, y) in enumerate(train.batches(B, device="cpu")):
        n = min(B, tl - seen)
        assert ids.dtype == torch.int64 and ids.shape == (F, n) and ids.is_contiguous()
        assert np.array_equal(ids.numpy(), raw[seen:seen + n].T)              # lS_i[k] = X_cat[:, k]  (:404, :417)
        assert y.shape == (n, 1) and np.array_equal(y.numpy()[:, 0], lab[seen:seen + n].astype(np.float32))
        if dense_dim:
            assert np.array_equal(x.numpy(), den[seen:seen + n])
        else:
            assert x is None
        seen += n
    assert seen == tl and k + 1 == train.num_batches(B)           # last batch is short, not dropped
    first_test = next(test.batches(B, device="cpu"))
    assert np.array_equal(first_test[1].numpy(), raw[tl:tl + B].T)


def test_bad_files_are_rejected(tmp_path):
    paths = cd.write_synthetic(str(tmp_path), [10, 20, 30], 50, dense_dim=0)
    with pytest.raises(ValueError):
        cd.BinaryCTRDataset(num_fields=4, **paths)                # count file has 4 entries, not 5
    with open(paths["cat_path"], "ab") as f:
        f.write(b"\\0\\0")
    with pytest.raises(ValueError):
        cd.BinaryCTRDataset(num_fields=3, **paths)


def test_batches_equal_the_reference_loader(tmp_path):
    """tests/golden/loader_kaggle.npz: what the reference's own KaggleDataset + DataLoader(collate_wrapper_criteo_offset2)
    (dlrm_data_pytorch.py:315-392, :409-421, :655-690) yield for a tiny file set, recorded by
    tests/golden/make_loader_golden.py.  The reference's 'train' split of a small file is the whole file (its split
    point is the hard-coded Kaggle row count, :342); here the split is at 6/7 of the rows, so the reference's batches
    0..5 are this loader's train split and its batch 6 is the test split.  train_len = S reproduces the reference's
    split literally."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loader_kaggle.npz"))
    B, nb = int(g["batch"]), int(g["num_batches"])
    paths = {"cat_path": str(tmp_path / "sparse.bin"), "label_path": str(tmp_path / "label.bin"),
             "count_path": str(tmp_path / "count.bin"), "dense_path": str(tmp_path / "dense.bin")}
    g["cat"].tofile(paths["cat_path"]); g["label"].tofile(paths["label_path"])
    g["count_file"].tofile(paths["count_path"]); g["dense"].tofile(paths["dense_path"])
    S = len(g["label"])
    assert S * 6 // 7 == 6 * B and nb == 7

    def check(batch, k):
        x, ids, y = batch
        assert x.dtype == torch.float32 and np.array_equal(x.numpy(), g[f"X_{k}"])              # X_int
        assert ids.dtype == torch.int64 and np.array_equal(ids.numpy(), g[f"lS_i_{k}"])         # torch.stack(lS_i): [26, B]
        assert y.dtype == torch.float32 and np.array_equal(y.numpy(), g[f"T_{k}"])              # [B, 1]
        # lS_o (one index per bag: offsets = arange(B), :418) is implied by the [F, B] layout
        assert np.array_equal(g[f"lS_o_{k}"], np.tile(np.arange(ids.shape[1]), (ids.shape[0], 1)))

    train = cd.BinaryCTRDataset(num_fields=26, split="train", **paths)
    test = cd.BinaryCTRDataset(num_fields=26, split="test", **paths)
    assert train.counts == [int(c) for c in np.diff(g["count_file"])]
    k = -1
    for k, b in enumerate(train.batches(B, device="cpu")):
        check(b, k)
    assert k == 5
    check(next(test.batches(B, device="cpu")), 6)
    whole = cd.BinaryCTRDataset(num_fields=26, split="train", train_len=S, **paths)             # the reference's split
    for k, b in enumerate(whole.batches(B, device="cpu")):
        check(b, k)
    assert k == nb - 1
