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
