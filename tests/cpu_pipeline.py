"""The reference's per-step order on the C oracle + the numpy restatement of the tensor half:
test infrastructure shared by the GPU parity tests."""
import numpy as np


class CpuPipeline:
    """The reference's per-step order on the oracle + numpy tensor restatement."""

    def __init__(self, om, coll, thr, adjust, alpha):
        self.om = om
        self.sk = om.Oracle(coll.hotn, thr, adjust, alpha, 1)
        self.w_hot = coll.weight_high.cpu().numpy().copy()
        self.w_cold = [w.cpu().numpy().copy() for w in coll.weight_hash]
        self.offsets = coll.offsets
        self.hash_sizes = coll.hash_sizes

    def forward(self, ids):
        F, B = ids.shape
        self.gids = [(ids[f] + self.offsets[f]).astype(np.int64) for f in range(F)]
        self.que = [self.sk.query(self.gids[f]) for f in range(F)]
        return np.stack([self.om.ref_forward(self.que[f], self.w_hot, self.w_cold[f], self.hash_sizes[f])[0] for f in range(F)])

    def backward_update(self, ids, grad, scores, lr, per_field):
        F, B = ids.shape
        ins_all = []
        if per_field:
            for f in range(F):
                ins = self.sk.insert_val(self.gids[f], scores[f])
                qa = self.sk.query(self.gids[f])
                self.om.ref_migrate(ins, qa, ids[f], self.w_hot, self.w_cold[f], self.hash_sizes[f])
                ins_all.append(ins)
        else:
            ins = self.sk.insert_val(np.concatenate(self.gids), scores.reshape(-1)).reshape(F, B)
            for f in range(F):
                qa = self.sk.query(self.gids[f])
                self.om.ref_migrate(ins[f], qa, ids[f], self.w_hot, self.w_cold[f], self.hash_sizes[f])
                ins_all.append(ins[f])
        # SGD on both tables with the forward's routing; hot table shared by all fields
        acc_h = np.zeros(self.w_hot.shape, np.float64)
        for f in range(F):
            que = self.que[f].astype(np.int64)
            hot = que < 0
            np.add.at(acc_h, (-que[hot]) % self.w_hot.shape[0], grad[f][hot].astype(np.float64))
            acc_c = np.zeros(self.w_cold[f].shape, np.float64)
            np.add.at(acc_c, que[~hot] % self.hash_sizes[f], grad[f][~hot].astype(np.float64))
            self.w_cold[f] -= (lr * acc_c).astype(np.float32)
        self.w_hot -= (lr * acc_h).astype(np.float32)
        return np.stack(ins_all)
