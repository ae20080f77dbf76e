"""Thin DLRM / WDL / DCN models on the CAFE embedding layer.

Only what is needed to report losses on the hot path's configurations: the heads are stock torch ops on the GPU
(they are not on the path this package accelerates); the embedding layer is cafe_amd.CafeEmbeddingCollection, i.e.
the fused HIP launches.  Structure and initialisation follow the reference:

* DLRM: bottom MLP -> dot interaction of [x] + embeddings, strictly-lower-triangle terms, concatenated behind x ->
  top MLP whose last layer is a sigmoid (dlrm_s_pytorch.py:213-251 create_mlp, :779-820 interact_features,
  :825-851 sequential_forward).
* WDL: z = [dense, embeddings]; sigmoid(deep(z) + wide(z)) with deep = MLP [in, 256, 256, 1] (wdl.py:410-414,
  :704-721).  The reference asks create_mlp for a sigmoid on the last deep layer as well (sigmoid_layer =
  ln_top.size - 2), so the deep logit is squashed twice; kept, the losses are the reference's.
* DCN: three cross layers x_{l+1} = x0 * (x_l w_l) + b_l next to MLP [in, 256, 256, 256] (last layer sigmoid, as
  above), a final Linear(in + 256, 1) and a sigmoid (dcn.py:411-422, :712-742).  The reference keeps the cross
  weights in plain Python lists (dcn.py:415-420; ArtifactEvaluation/models.py:99-108 too), so its optimizer never
  sees them and they stay at their N(0, 1e-4) initialisation; `train_cross=False` (default) reproduces that,
  True registers them.
* No dense features (Avazu-shaped data): the root scripts hard-code 13 dense inputs (wdl.py:410, dcn.py:412); the
  `dense is None` handling is ArtifactEvaluation/models.py:57-58, :87-88, :127-128.
* Table sizes: cafe_amd.config (dlrm_s_pytorch.py:1237-1242, :309-322).  A field is a sketch field when
  n > 2000 * compress_rate (dlrm_s_pytorch.py:309); smaller fields get a plain full table, as the reference gives them
  (dlrm_s_pytorch.py:323-343: nn.EmbeddingBag(n, m), U(-sqrt(1/max(n, 5)), +)), outside the fused path (stock torch
  ops).  Global id offsets and hotn count every field (tot_nums, dlrm_s_pytorch.py:353 and :1237-1240).  cfg4
  (compress_rate 0.1) has two such fields.

Call order of a training step is the reference's (dlrm_s_pytorch.py:1640-1658): forward, loss.backward(),
insert_grad (scores -> sketch insert -> migration), optimizer step; the embedding tables' SGD runs inside the fused
backward (CafeEmbeddingCollection.backward_update), the dense parameters use torch.optim.SGD."""
import numpy as np
import torch
import torch.nn as nn

from . import config as cfg
from .collection import CafeEmbeddingCollection


def create_mlp(ln, sigmoid_layer=-1, rng=None):
    """dlrm_s_pytorch.py:213-251: Linear + ReLU (Sigmoid at `sigmoid_layer`), weights N(0, sqrt(2/(m+n))), biases
    N(0, sqrt(1/m)) drawn from numpy like the reference does."""
    rng = np.random if rng is None else rng
    layers = []
    for i in range(len(ln) - 1):
        n, m = int(ln[i]), int(ln[i + 1])
        ll = nn.Linear(n, m, bias=True)
        w = rng.normal(0.0, np.sqrt(2 / (m + n)), size=(m, n)).astype(np.float32)
        b = rng.normal(0.0, np.sqrt(1 / m), size=m).astype(np.float32)
        with torch.no_grad():
            ll.weight.copy_(torch.from_numpy(w))
            ll.bias.copy_(torch.from_numpy(b))
        layers.append(ll)
        layers.append(nn.Sigmoid() if i == sigmoid_layer else nn.ReLU())
    return nn.Sequential(*layers)


def interact_dot(x, ly, itself=False):
    """dlrm_s_pytorch.py:779-809 (arch_interaction_op == "dot"); ly: list of [B, D] or one [F, B, D] tensor."""
    if isinstance(ly, torch.Tensor):
        feats = ly.transpose(0, 1)                                  # [B, F, D]
        T = feats if x is None else torch.cat([x[:, None, :], feats], dim=1)
    else:
        b, d = ly[0].shape
        T = torch.cat(([x] if x is not None else []) + list(ly), dim=1).view(b, -1, d)
    Z = torch.bmm(T, T.transpose(1, 2))
    ni = Z.shape[1]
    off = 1 if itself else 0
    li = torch.tensor([i for i in range(ni) for _ in range(i + off)], device=Z.device)
    lj = torch.tensor([j for i in range(ni) for j in range(i + off)], device=Z.device)
    zflat = Z[:, li, lj]
    return zflat if x is None else torch.cat([x, zflat], dim=1)


class CafeCTRHead(nn.Module):
    """Everything behind the embedding layer: pure torch, device-agnostic (the parity tests run a copy on the CPU
    behind the oracle's embedding layer)."""

    def __init__(self, kind, num_fields, dim, dense_dim=13, ln_bot=None, ln_top=None, seed=123, train_cross=False,
                 interaction_itself=False):
        super().__init__()
        if kind not in ("dlrm", "wdl", "dcn"):
            raise ValueError(kind)
        self.kind, self.dim, self.dense_dim, self.F = kind, int(dim), int(dense_dim), int(num_fields)
        self.interaction_itself = interaction_itself
        self.train_cross = train_cross
        F = self.F
        rng = np.random.RandomState(seed)                         # --numpy-rand-seed, dlrm_s_pytorch.py:1068
        in_dim = self.dim * F + self.dense_dim
        if kind == "dlrm":
            if self.dense_dim:
                ln_bot = [self.dense_dim, 512, 256, self.dim] if ln_bot is None else list(ln_bot)
                if ln_bot[0] != self.dense_dim or ln_bot[-1] != self.dim:
                    raise ValueError("ln_bot must map dense_dim to the embedding dim (dlrm_s_pytorch.py:1257-1267)")
                self.bot_l = create_mlp(ln_bot, -1, rng)
            nf = F + (1 if self.dense_dim else 0)
            n_int = nf * (nf + 1) // 2 if interaction_itself else nf * (nf - 1) // 2
            first = n_int + (self.dim if self.dense_dim else 0)
            ln_top = [first, 1024, 1024, 512, 256, 1] if ln_top is None else [first] + list(ln_top)[1:]
            self.top_l = create_mlp(ln_top, len(ln_top) - 2, rng)
        elif kind == "wdl":
            ln_top = [in_dim, 256, 256, 1]
            self.wide = nn.Linear(in_dim, 1)
            self.top_l = create_mlp(ln_top, len(ln_top) - 2, rng)
            nn.init.normal_(self.wide.weight, mean=0, std=0.0001)
        else:
            ln_top = [in_dim, 256, 256, 256]
            self.cross_layer_n = 3
            for i in range(self.cross_layer_n):
                w = torch.normal(0.0, 0.0001, size=(in_dim, 1))
                b = torch.zeros(in_dim)
                if train_cross:
                    self.register_parameter(f"cross_weight_{i}", nn.Parameter(w))
                    self.register_parameter(f"cross_bias_{i}", nn.Parameter(b))
                else:                                             # the reference's optimizer never sees them
                    self.register_buffer(f"cross_weight_{i}", w)
                    self.register_buffer(f"cross_bias_{i}", b)
            self.last_layer = nn.Linear(in_dim + 256, 1)
            self.top_l = create_mlp(ln_top, len(ln_top) - 2, rng)

    def forward(self, dense, emb):
        """emb: [F, B, D] -> click probability [B, 1]."""
        B = emb.shape[1]
        if self.kind == "dlrm":
            x = self.bot_l(dense) if self.dense_dim else None
            return self.top_l(interact_dot(x, emb, self.interaction_itself))
        z = emb.transpose(0, 1).reshape(B, -1)
        if self.dense_dim:
            z = torch.cat([dense, z], dim=1)
        if self.kind == "wdl":
            return torch.sigmoid(self.top_l(z) + self.wide(z))
        x1 = z
        for i in range(self.cross_layer_n):                       # dcn.py:712-721
            x1 = z * torch.mm(x1, getattr(self, f"cross_weight_{i}")) + getattr(self, f"cross_bias_{i}")
        return torch.sigmoid(self.last_layer(torch.cat([self.top_l(z), x1], dim=1)))


class _Lookup(torch.autograd.Function):
    """Forward = the fused query + gather launch; backward only keeps the output gradient for backward_update."""

    @staticmethod
    def forward(ctx, anchor, owner, ids):
        ctx.owner = owner
        return owner.embedding.forward(ids, out=owner._lookup_out(ids, 0))

    @staticmethod
    def backward(ctx, g):
        ctx.owner._emb_grad = g.contiguous()
        return None, None, None


class CafeCTRModel(nn.Module):
    """kind: "dlrm" | "wdl" | "dcn".  dense_dim = 0 for data without dense features."""

    def __init__(self, kind, counts, dim, dense_dim=13, compress_rate=0.001, hash_rate=0.3, threshold=500,
                 adjust_thres=1, alpha=1.0000005, ln_bot=None, ln_top=None, device=None, seed=123, train_cross=False,
                 interaction_itself=False, per_field_calls=True):
        super().__init__()
        self.kind, self.dim, self.dense_dim = kind, int(dim), int(dense_dim)
        self.counts = [int(c) for c in counts]
        F = len(self.counts)
        self.sk_idx = [i for i, n in enumerate(self.counts) if n > 2000 * compress_rate]        # dlrm_s_pytorch.py:309
        self.plain_idx = [i for i in range(F) if i not in set(self.sk_idx)]
        if not self.sk_idx:
            raise ValueError("no field is large enough to be a sketch field (dlrm_s_pytorch.py:309)")
        self.per_field_calls = per_field_calls
        self.hotn = cfg.hot_rows(self.counts, dim, compress_rate, hash_rate)
        sk_counts = [self.counts[i] for i in self.sk_idx]
        self.hash_sizes = cfg.cold_rows(sk_counts, compress_rate, hash_rate)
        offsets = cfg.field_offsets(self.counts)[self.sk_idx]
        self.embedding = CafeEmbeddingCollection(sk_counts, dim, self.hotn, self.hash_sizes, offsets=offsets,
                                                 threshold=threshold, adjust_thres=adjust_thres, alpha=alpha, device=device,
                                                 seed=seed)
        rng = np.random.RandomState(seed + 17)
        self.plain = nn.ModuleList()
        for i in self.plain_idx:
            n = self.counts[i]
            e = nn.Embedding(n, dim)
            b = np.sqrt(1 / max(n, 5))
            with torch.no_grad():
                e.weight.copy_(torch.from_numpy(rng.uniform(-b, b, size=(n, dim)).astype(np.float32)))
            self.plain.append(e)
        self._order = None
        if self.plain_idx:
            order = self.sk_idx + self.plain_idx
            self._order = torch.tensor([order.index(i) for i in range(F)], dtype=torch.long)
        dev = self.embedding.device
        self.head = CafeCTRHead(kind, len(self.counts), dim, dense_dim, ln_bot, ln_top, seed, train_cross,
                                interaction_itself)
        self._anchor = nn.Parameter(torch.zeros(()))              # gives autograd a reason to call _Lookup.backward
        self._emb_grad = None
        self.loss_fn = nn.BCELoss(reduction="mean")               # dlrm_s_pytorch.py:462, wdl.py:425, dcn.py:433
        self.to(dev)

    def forward(self, dense, ids):
        """dense: [B, dense_dim] or None; ids: [F, B] int64 local ids on the device."""
        if not self.plain_idx:
            return self.head(dense, _Lookup.apply(self._anchor, self, ids))
        sk = _Lookup.apply(self._anchor, self, ids[self.sk_idx].contiguous())
        pl = torch.stack([e(ids[i]) for e, i in zip(self.plain, self.plain_idx)])
        emb = torch.cat([sk, pl]).index_select(0, self._order.to(sk.device))
        return self.head(dense, emb)

    def _lookup_out(self, ids, ahead):
        """Two output arrays used by alternate steps (ahead = 1: the one the NEXT forward will write), so that the next
        step's forward can be named while this step's update is being queued (CafeEmbeddingCollection.hint_next)."""
        shape = (ids.shape[0], ids.shape[1], self.dim)
        outs = getattr(self, "_outs", None)
        if outs is None or outs[0].shape != torch.Size(shape):
            outs = [torch.empty(shape, device=self.embedding.device) for _ in range(2)]
            self._outs, self._out_at = outs, 0
        if not ahead:
            self._out_at ^= 1
            return self._outs[self._out_at]
        return self._outs[self._out_at ^ 1]

    def dense_parameters(self):
        """What torch's optimizer updates: the head and the plain tables of fields too small for the sketch."""
        return list(self.head.parameters()) + list(self.plain.parameters())

    def train_step(self, dense, ids, labels, optimizer, lr, next_ids=None):
        """One step in the reference's order (dlrm_s_pytorch.py:1640-1658): forward, backward, insert_grad, SGD.
        `optimizer` holds dense_parameters(); the sketch-backed tables are updated by the fused backward.  Returns the loss.
        next_ids: the next step's ids when the data loader has them already (contiguous [F, B] int64 on the device, all
        fields sketch fields): that step's embedding forward is then queued behind this step's update before the host
        waits for the device's verdict on the insert (cafe_forward_hint_dev)."""
        optimizer.zero_grad(set_to_none=True)
        p = self.forward(dense, ids)
        loss = self.loss_fn(p, labels)
        loss.backward()
        if next_ids is not None and not self.plain_idx:
            self.embedding.hint_next(next_ids, self._lookup_out(next_ids, 1))
        self.embedding.backward_update(self._emb_grad, lr, per_field_calls=self.per_field_calls)
        self._emb_grad = None
        optimizer.step()
        return loss.detach()
