"""Reader and writer of the state file the reference's PinSAGE-generation sketch writes (TOIS_revision/PinSAGE/
sketch.cpp:333-402 save_state / load_state), and its mapping onto this package's state dict (HotSketch.state()).

Host logic only (numpy); the native checkpoint format of this package stays CAFESK01 (cafe_sketch_save).  Layout,
little endian, exactly the reference's fout.write sequence:

    int32 real_n, int32 num, float64 tot, float64 p, float32 t, float64 decay_importance, int32 batch_num,
    uint8 global_flip_bit, float32 Threshold                                                   (45 bytes, unpadded)
    n1 x Bucket  { int32 val[4]; float32 cnt[4]; int32 dic[4]; uint8 flip_bit; 3 padding bytes }          (52 bytes)
    n2 x Bucket2 { int32 val[4]; float32 cnt[4]; int32 dic[4]; float32 t[4]; uint8 flip_bit; 3 padding }  (68 bytes)
    int32 queue_size, queue_size x uint32 free slots in pop order
with n1 = max(int(lim * 0.9), 1), n2 = max(int(lim * 0.1), 1) (:53-56).  `p` is never initialised by the reference and
the padding bytes are whatever the allocator left: both are kept verbatim so that a file can be reproduced byte for
byte."""
import numpy as np

_HEAD = np.dtype([("real_n", "<i4"), ("num", "<i4"), ("tot", "<f8"), ("p_raw", "V8"), ("t", "<f4"),
                  ("decay_importance", "<f8"), ("batch_num", "<i4"), ("flip", "u1"), ("thr", "<f4")])
_B1 = np.dtype([("val", "<i4", 4), ("cnt", "<f4", 4), ("dic", "<i4", 4), ("flip", "u1"), ("pad", "V3")])
_B2 = np.dtype([("val", "<i4", 4), ("cnt", "<f4", 4), ("dic", "<i4", 4), ("t", "<f4", 4), ("flip", "u1"), ("pad", "V3")])
assert _HEAD.itemsize == 45 and _B1.itemsize == 52 and _B2.itemsize == 68


def table_sizes(lim):
    return max(int(lim * 0.9), 1), max(int(lim * 0.1), 1)


def read_pinsage_state(path, lim):
    """Parse a save_state file of a sketch created with `lim` slots.  Returns {"head", "main", "lru", "fifo"}: numpy
    structured arrays (padding and the uninitialised `p` included) and the free list in pop order."""
    n1, n2 = table_sizes(lim)
    raw = np.fromfile(path, dtype=np.uint8)
    need = _HEAD.itemsize + n1 * _B1.itemsize + n2 * _B2.itemsize + 4
    if raw.size < need:
        raise ValueError(f"{path}: {raw.size} bytes, a sketch of {lim} slots needs at least {need}")
    o = 0
    head = raw[o:o + _HEAD.itemsize].view(_HEAD)[0].copy(); o += _HEAD.itemsize
    main = raw[o:o + n1 * _B1.itemsize].view(_B1).copy(); o += n1 * _B1.itemsize
    lru = raw[o:o + n2 * _B2.itemsize].view(_B2).copy(); o += n2 * _B2.itemsize
    qn = int(raw[o:o + 4].view("<i4")[0]); o += 4
    if qn < 0 or raw.size != o + 4 * qn:
        raise ValueError(f"{path}: free list of {qn} entries does not match the file size {raw.size}")
    fifo = raw[o:o + 4 * qn].view("<u4").copy()
    return {"lim": int(lim), "head": head, "main": main, "lru": lru, "fifo": fifo}


def write_pinsage_state(path, st):
    """Inverse of read_pinsage_state: the same bytes."""
    with open(path, "wb") as f:
        f.write(np.asarray(st["head"]).tobytes())
        f.write(np.ascontiguousarray(st["main"]).tobytes())
        f.write(np.ascontiguousarray(st["lru"]).tobytes())
        f.write(np.int32(len(st["fifo"])).tobytes())
        f.write(np.ascontiguousarray(st["fifo"], dtype="<u4").tobytes())


def pinsage_query(st, ids):
    """What that file's sketch answers to batch_query (sketch.cpp: query -- 0 for id 0; -dic of a hot entry of the main
    bucket, else -dic of a hot entry of the LRU bucket, else the id)."""
    n1, n2 = len(st["main"]), len(st["lru"])
    out = np.empty(len(ids), np.int32)
    for k, x in enumerate(np.asarray(ids, np.uint32).tolist()):
        if x == 0:
            out[k] = 0
            continue
        q = x
        b = st["main"][x * 1000000007 % n1]
        hit = [i for i in range(4) if int(b["val"][i]) & 0xFFFFFFFF == x and b["dic"][i]]
        if hit:
            q = -int(b["dic"][hit[0]])
        else:
            b2 = st["lru"][x * 1000000007 % n2]
            hit = [i for i in range(4) if int(b2["val"][i]) & 0xFFFFFFFF == x and b2["dic"][i]]
            if hit:
                q = -int(b2["dic"][hit[0]])
        out[k] = np.int64(q).astype(np.int32) if q >= 0 else q
    return out


def to_state_dict(st, decay=None):
    """The parsed file as the state dict of HotSketch.load_state_dict / the oracle's load_state (timestamps, float in
    that generation, are integral values: the clock only ever counts)."""
    h = st["head"]
    return {
        "mval": st["main"]["val"].reshape(-1).astype(np.int32), "mcnt": st["main"]["cnt"].reshape(-1).astype(np.float32),
        "mdic": st["main"]["dic"].reshape(-1).astype(np.int32), "mflip": st["main"]["flip"].astype(np.uint8),
        "lval": st["lru"]["val"].reshape(-1).astype(np.int32), "lcnt": st["lru"]["cnt"].reshape(-1).astype(np.float32),
        "ldic": st["lru"]["dic"].reshape(-1).astype(np.int32), "lt": st["lru"]["t"].reshape(-1).astype(np.int32),
        "lflip": st["lru"]["flip"].astype(np.uint8), "fifo": st["fifo"].astype(np.uint32),
        "thr": np.float32(h["thr"]), "num": int(h["num"]), "real_n": int(h["real_n"]), "clk": int(h["t"]),
        "flip": int(h["flip"]), "batch_num": int(h["batch_num"]), "resets": 0,
        "decay": float(h["decay_importance"] if decay is None else decay),
        "n1": len(st["main"]), "n2": len(st["lru"]), "lim": st["lim"],
    }


def from_state_dict(sd, template=None):
    """A state dict as a save_state image.  `template` (a parsed file) supplies the bytes the reference never defines
    (`p`, `tot`, struct padding); zeros otherwise."""
    n1, n2 = int(sd["n1"]), int(sd["n2"])
    head = np.zeros((), _HEAD) if template is None else template["head"].copy()
    main = np.zeros(n1, _B1) if template is None else template["main"].copy()
    lru = np.zeros(n2, _B2) if template is None else template["lru"].copy()
    head["real_n"], head["num"], head["t"] = sd["real_n"], sd["num"], np.float32(sd["clk"])
    head["decay_importance"], head["batch_num"], head["flip"], head["thr"] = sd["decay"], sd["batch_num"], sd["flip"], sd["thr"]
    main["val"], main["cnt"], main["dic"] = (np.asarray(sd[k]).reshape(n1, 4) for k in ("mval", "mcnt", "mdic"))
    main["flip"] = np.asarray(sd["mflip"], np.uint8)
    lru["val"], lru["cnt"], lru["dic"] = (np.asarray(sd[k]).reshape(n2, 4) for k in ("lval", "lcnt", "ldic"))
    lru["t"] = np.asarray(sd["lt"]).reshape(n2, 4).astype(np.float32)
    lru["flip"] = np.asarray(sd["lflip"], np.uint8)
    return {"lim": int(sd["lim"]), "head": head, "main": main, "lru": lru, "fifo": np.asarray(sd["fifo"], np.uint32)}
