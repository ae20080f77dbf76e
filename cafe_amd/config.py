"""Sizing rules and synthetic workloads of the CAFE hot path.

Everything here is host-side bookkeeping: table sizes exactly as the reference driver derives them
(dlrm_s_pytorch.py:1237-1242, 309-322) and seeded Zipf id streams shaped like the datasets the
reference's bench scripts name."""
import numpy as np

# Criteo-Kaggle categorical cardinalities (tricks/sketchtest.py:41-45); sum = 33 762 577
KAGGLE_COUNTS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
                 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def scaled_counts(total, base=KAGGLE_COUNTS):
    """26 fields with the Kaggle proportions scaled to `total` distinct ids (cfg3: ~880 M, Terabyte-shaped)."""
    s = float(sum(base))
    out = [max(3, int(round(c * total / s))) for c in base]
    return out


def avazu_like_counts():
    """22 fields, sum ~9.4 M (cfg5, Avazu-shaped; the reference ships no count file)."""
    return [240, 7, 7, 4737, 7745, 26, 8552, 559, 36, 2686408, 6729486, 8251, 5, 4, 2626, 8, 9, 435, 4, 68, 172, 60]


def hot_rows(counts, dim, compress_rate, hash_rate):
    """dlrm_s_pytorch.py:1237-1240: hotn = int(totn * cr * (1 - hash_rate) * (4m / (4m + 48)))."""
    totn = int(sum(counts))
    return int(totn * compress_rate * (1 - hash_rate) * (dim * 4 / (dim * 4 + 48)))


def cold_rows(counts, compress_rate, hash_rate):
    """dlrm_s_pytorch.py:1241 and :320: hash_size = round(cr * hash_rate * n + 0.51) per field."""
    hr = compress_rate * hash_rate
    return [int(round(hr * n + 0.51)) for n in counts]


def field_offsets(counts):
    """Global id offset of each field (dlrm_s_pytorch.py: tot_nums, the running sum of cardinalities)."""
    return np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))[:-1]]).astype(np.int64)


class Workload:
    """One named configuration: cardinalities, embedding dim, batch, derived table sizes."""

    def __init__(self, name, counts, dim, batch, compress_rate, hash_rate, threshold=500, alpha=1.0000005):
        self.name = name
        self.counts = [int(c) for c in counts]
        self.dim = int(dim)
        self.batch = int(batch)
        self.compress_rate = compress_rate
        self.hash_rate = hash_rate
        self.threshold = threshold
        self.alpha = alpha
        self.hotn = hot_rows(self.counts, dim, compress_rate, hash_rate)
        self.hash_sizes = cold_rows(self.counts, compress_rate, hash_rate)
        self.offsets = field_offsets(self.counts)
        self.num_fields = len(self.counts)

    @property
    def lookups_per_step(self):
        return self.num_fields * self.batch

    def bytes_per_lookup(self):
        """Algorithmic HBM bytes per lookup, SURVEY.md 8(d): 160 + 20 D."""
        return 160 + 20 * self.dim

    def make_ids(self, seed, steps=1, a=1.05):
        """[steps, num_fields, batch] int64 local ids: (zipf(a) - 1) mod n_k, field 0 never 0."""
        rng = np.random.default_rng(seed)
        out = np.empty((steps, self.num_fields, self.batch), dtype=np.int64)
        for s in range(steps):
            for f, n in enumerate(self.counts):
                loc = (rng.zipf(a, size=self.batch) - 1) % n if a > 1.0 else rng.integers(0, n, size=self.batch)
                if f == 0:
                    loc = np.maximum(loc, 1) if n > 1 else loc
                out[s, f] = loc
        return out


def workload(name, batch=None):
    """cfg1/cfg2: Kaggle-shaped D=16; cfg3: Terabyte-shaped D=128; cfg4: cfg3 tables at cr 0.1 / hash-rate 0.7;
    cfg5: Avazu-shaped D=16 (BASELINE.json configs[0..4])."""
    if name == "cfg1":
        return Workload(name, KAGGLE_COUNTS, 16, batch or 128, 0.001, 0.3)
    if name == "cfg2":
        return Workload(name, KAGGLE_COUNTS, 16, batch or 8192, 0.001, 0.3)
    if name == "cfg3":
        return Workload(name, scaled_counts(880_000_000), 128, batch or 8192, 0.001, 0.3)
    if name == "cfg4":
        return Workload(name, scaled_counts(880_000_000), 128, batch or 8192, 0.1, 0.7, threshold=20)
    if name == "cfg5":
        return Workload(name, avazu_like_counts(), 16, batch or 8192, 0.001, 0.3)
    raise ValueError(name)
