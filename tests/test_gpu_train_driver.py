"""cafe_amd.train end to end on the GPU: files in the reference's binary format -> model -> steps -> checkpoint
(including the complete sketch) -> resume through the driver itself (it continues at the saved iteration, like the
reference's ld_j, dlrm_s_pytorch.py:1512-1525).  A run resumed from its checkpoint must continue exactly like the
uninterrupted run: same sketch decisions (bit-exact), tables, scores and head within 1e-5 relative (both runs are this
package on the same GPU; they differ only in the order of the float atomics of rows that span several tiles).  One field (60 ids) is
too small to be a sketch field at this compress rate and gets a plain table (dlrm_s_pytorch.py:323-343)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, what, rtol=1e-5):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= rtol, f"{what}: max rel err {err:.3e}"


@pytest.mark.parametrize("model_kind,dense", [("dlrm", True), ("wdl", False)])
def test_files_checkpoint_resume(tmp_path, model_kind, dense, capsys):
    from cafe_amd import data as cd, train
    counts = [3000, 150, 9000, 40000, 60, 700, 120]
    paths = cd.write_synthetic(str(tmp_path / "d"), counts, 4000, dense_dim=13 if dense else 0, seed=9, zipf_a=1.2)
    common = ["--model", model_kind, "--cat-path", paths["cat_path"], "--label-path", paths["label_path"],
              "--count-path", paths["count_path"], "--num-fields", str(len(counts)), "--arch-sparse-feature-size", "16",
              "--compress-rate", "0.05", "--hash-rate", "0.3", "--sketch-threshold", "2", "--mini-batch-size", "128",
              "--learning-rate", "0.2", "--print-freq", "4", "--arch-mlp-top", "64-1"]
    if dense:
        common += ["--dense-path", paths["dense_path"], "--arch-mlp-bot", "13-32-16"]
    ck = str(tmp_path / "ck.pt")
    full = train.main(common + ["--num-batches", "16", "--test-freq", "8", "--test-num-batches", "2"])
    out = capsys.readouterr().out
    assert "step 16: loss" in out and "test loss" in out
    a = train.main(common + ["--num-batches", "8", "--save-model", ck])
    del a
    b = train.main(common + ["--num-batches", "16", "--load-model", ck])
    assert '"start_step": 8' in capsys.readouterr().out
    assert full.plain_idx == [4] and len(full.embedding.weight_hash) == 6
    sa, sb = full.embedding.sketch.state(), b.embedding.sketch.state()
    for key in ("mval", "mdic", "mflip", "lval", "ldic", "lt", "lflip", "fifo"):
        assert np.array_equal(np.asarray(sa[key]), np.asarray(sb[key])), key
    for key in ("num", "real_n", "clk", "flip", "batch_num"):
        assert sa[key] == sb[key], key
    _close(sb["mcnt"], sa["mcnt"], "main scores")
    _close(b.embedding.weight_high.cpu().numpy(), full.embedding.weight_high.cpu().numpy(), "hot table")
    for f in range(len(full.embedding.weight_hash)):
        _close(b.embedding.weight_hash[f].cpu().numpy(), full.embedding.weight_hash[f].cpu().numpy(), f"cold {f}")
    _close(b.plain[0].weight.detach().cpu().numpy(), full.plain[0].weight.detach().cpu().numpy(), "plain table")
    for pa, pb in zip(full.head.parameters(), b.head.parameters()):
        _close(pb.detach().cpu().numpy(), pa.detach().cpu().numpy(), "head")
    assert (np.asarray(sa["mdic"]) != 0).sum() > 10            # ids did turn hot
