"""Binary embedding store replacing the JSON hand-off between
clip_inference.py and final_main.py.

The reference writes every image embedding as a JSON float list (clip_inference.py:235-271:
one `.cpu().numpy().tolist()` per sample, ~4 GB of text for CelebA) and reads it back with
`pd.read_json` of the whole file plus a DataFrame column lookup per item
(data/celeba_embeddings_reg.py:29,63-74).  Here one flat file holds

    magic "DBMMEMB1" | u64 header bytes | JSON header | pad to 4096 | arrays (64-B aligned)

with arrays `embedding` float32 [N, D], `y`, `confounder`, `group`, `split`, `y_pred` int64 [N]
and the file names as one utf-8 blob + offsets.  Loading is an `np.memmap` (no parse, no copy
until `.to(device)`).  `export_json` / `import_json` convert to and from the reference's exact
JSON schema so existing files and downstream notebooks keep working.
"""
import json
import os

import numpy as np

MAGIC = b"DBMMEMB1"
_INT_FIELDS = ("y", "confounder", "group", "split", "y_pred")
# JSON key names per dataset (clip_inference.py:238-257)
_KEYS = {"waterbirds": ("y", "place"), "celeba": ("blond", "male")}


def _align(n, a):
    return (n + a - 1) // a * a


def save(path, embeddings, y, confounder, group, split, y_pred, filenames, dataset="celeba"):
    emb = np.ascontiguousarray(np.asarray(embeddings, dtype=np.float32))
    n, d = emb.shape
    cols = {"y": y, "confounder": confounder, "group": group, "split": split, "y_pred": y_pred}
    arrays = {"embedding": emb}
    for k, v in cols.items():
        a = np.ascontiguousarray(np.asarray(v, dtype=np.int64)).reshape(-1)
        if a.shape[0] != n:
            raise ValueError(f"{k}: expected {n} entries, got {a.shape[0]}")
        arrays[k] = a
    if len(filenames) != n:
        raise ValueError("filenames: wrong length")
    blobs = [f.encode("utf-8") for f in filenames]
    arrays["name_offsets"] = np.cumsum([0] + [len(b) for b in blobs]).astype(np.int64)
    arrays["name_blob"] = np.frombuffer(b"".join(blobs), dtype=np.uint8) if blobs else np.zeros(0, np.uint8)
    header = {"version": 1, "dataset": dataset, "n": int(n), "dim": int(d), "arrays": {}}
    off = 0
    for k, a in arrays.items():
        off = _align(off, 64)
        header["arrays"][k] = {"dtype": str(a.dtype), "shape": list(a.shape), "offset": off}
        off += a.nbytes
    hj = json.dumps(header).encode("utf-8")
    data_start = _align(len(MAGIC) + 8 + len(hj), 4096)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(np.uint64(len(hj)).tobytes())
        f.write(hj)
        f.write(b"\0" * (data_start - f.tell()))
        for k, a in arrays.items():
            pos = data_start + header["arrays"][k]["offset"]
            f.write(b"\0" * (pos - f.tell()))
            f.write(a.tobytes())
    os.replace(tmp, path)
    return path


class Writer:
    """Incremental writer of a store file of KNOWN length (extract.Extractor: rows arrive batch by batch in file order).  The numeric
    arrays are memory-mapped at their final offsets and filled in place; the file names (unknown total size) go behind them at
    close(), then the header is written and the file is renamed into place.  Same format as save()."""

    def __init__(self, path, n, dim, dataset="celeba"):
        self.path, self.tmp, self.n, self.dim, self.dataset = path, path + ".tmp", int(n), int(dim), dataset
        self.header = {"version": 1, "dataset": dataset, "n": self.n, "dim": self.dim, "arrays": {}}
        off = 0
        spec = [("embedding", "float32", (self.n, self.dim))] + [(k, "int64", (self.n,)) for k in _INT_FIELDS] + \
            [("name_offsets", "int64", (self.n + 1,))]
        for k, dt, shape in spec:
            off = _align(off, 64)
            self.header["arrays"][k] = {"dtype": dt, "shape": list(shape), "offset": off}
            off += int(np.prod(shape)) * np.dtype(dt).itemsize
        self._blob_off = _align(off, 64)
        self._data_start = 4096                                # the header of a store is a few hundred bytes whatever n is
        with open(self.tmp, "wb") as f:
            f.truncate(self._data_start + self._blob_off)
        self._arr = {k: np.memmap(self.tmp, dtype=dt, mode="r+", offset=self._data_start + self.header["arrays"][k]["offset"], shape=shape)
                     for k, dt, shape in spec if int(np.prod(shape)) > 0}
        self._names, self._row = [], 0

    def append(self, embeddings, y, confounder, group, split, y_pred, filenames):
        b = len(filenames)
        if self._row + b > self.n:
            raise ValueError(f"store.Writer: {self._row + b} rows for a store of {self.n}")
        r = slice(self._row, self._row + b)
        self._arr["embedding"][r] = np.asarray(embeddings, dtype=np.float32).reshape(b, self.dim)
        for k, v in (("y", y), ("confounder", confounder), ("group", group), ("split", split), ("y_pred", y_pred)):
            self._arr[k][r] = np.asarray(v, dtype=np.int64).reshape(b)
        self._names += list(filenames)
        self._row += b

    def close(self, durable=False):
        """finish the file.  The arrays reach the page cache as they are appended; `durable=True` also waits for the disk (msync),
        which costs ~1 ms per MB on the GPU boxes' scratch disks -- the reference's json.dump does not sync either."""
        if self._row != self.n:
            raise ValueError(f"store.Writer: closed after {self._row} of {self.n} rows")
        blobs = [f.encode("utf-8") for f in self._names]
        if self.n:
            self._arr["name_offsets"][:] = np.cumsum([0] + [len(b) for b in blobs])
        if durable:
            for a in self._arr.values():
                a.flush()
        self._arr = {}
        blob = b"".join(blobs)
        self.header["arrays"]["name_blob"] = {"dtype": "uint8", "shape": [len(blob)], "offset": self._blob_off}
        hj = json.dumps(self.header).encode("utf-8")
        if len(MAGIC) + 8 + len(hj) > self._data_start:
            raise ValueError("store header too large")
        with open(self.tmp, "r+b") as f:
            f.write(MAGIC); f.write(np.uint64(len(hj)).tobytes()); f.write(hj)
            f.seek(self._data_start + self._blob_off)
            f.write(blob)
        os.replace(self.tmp, self.path)
        return self.path


class Store:
    """memory-mapped view of a store file"""

    def __init__(self, path):
        with open(path, "rb") as f:
            if f.read(8) != MAGIC:
                raise ValueError(f"{path}: not a dbmm embedding store")
            hl = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
            self.header = json.loads(f.read(hl).decode("utf-8"))
        self.path = path
        self.n, self.dim, self.dataset = self.header["n"], self.header["dim"], self.header["dataset"]
        start = _align(16 + hl, 4096)
        self._arr = {}
        for k, m in self.header["arrays"].items():
            count = int(np.prod(m["shape"])) if m["shape"] else 1
            if count == 0:
                self._arr[k] = np.zeros(m["shape"], dtype=m["dtype"])
            else:
                self._arr[k] = np.memmap(path, dtype=m["dtype"], mode="r", offset=start + m["offset"],
                                         shape=tuple(m["shape"]))

    def __len__(self):
        return self.n

    def __getattr__(self, k):
        arr = self.__dict__.get("_arr", {})
        if k in arr:
            return arr[k]
        raise AttributeError(k)

    @property
    def filenames(self):
        o, blob = self._arr["name_offsets"], self._arr["name_blob"].tobytes()
        return [blob[o[i]:o[i + 1]].decode("utf-8") for i in range(self.n)]

    def select(self, split):
        """row indices of one split (train 0 / val 1 / test 2), in file order"""
        return np.nonzero(np.asarray(self._arr["split"]) == split)[0]

    def table(self, rows=None, device="cuda"):
        """EmbeddingTable of (a subset of) the rows on `device` (one H2D copy)."""
        from .trainer import EmbeddingTable
        sel = slice(None) if rows is None else rows
        names = self.filenames
        return EmbeddingTable(np.asarray(self._arr["embedding"][sel]), np.asarray(self._arr["y"][sel]),
                              np.asarray(self._arr["confounder"][sel]), np.asarray(self._arr["y_pred"][sel]),
                              names if rows is None else [names[i] for i in rows], device=device)


def load(path):
    return Store(path)


def export_json(store, path):
    """the reference's image JSON (clip_inference.py:235-257, :263-271): per file name a dict with
    the keys in this order -- y|blond, place|male, group, split, image_embedding, y_pred -- ints as
    strings, the embedding as a float list."""
    ky, kc = _KEYS[store.dataset]
    out = {}
    names = store.filenames
    emb = store.embedding
    for i, name in enumerate(names):
        d = dict.fromkeys([ky, kc, "group", "split", "image_embedding", "y_pred"])
        d[ky] = str(int(store.y[i]))
        d["group"] = str(int(store.group[i]))
        d[kc] = str(int(store.confounder[i]))
        d["split"] = str(int(store.split[i]))
        d["image_embedding"] = np.asarray(emb[i]).tolist()
        d["y_pred"] = str(int(store.y_pred[i]))
        out[name] = d
    with open(path, "w") as f:
        json.dump(out, f)


def import_json(json_path, store_path, dataset=None):
    """convert a reference-format clip.json into a store file"""
    with open(json_path, "r") as f:
        d = json.load(f)
    names = list(d.keys())
    first = d[names[0]] if names else {}
    if dataset is None:
        dataset = "celeba" if "blond" in first else "waterbirds"
    ky, kc = _KEYS[dataset]
    emb = np.array([d[k]["image_embedding"] for k in names], dtype=np.float32).reshape(len(names), -1)
    col = lambda key: np.array([int(d[k][key]) for k in names], dtype=np.int64)
    return save(store_path, emb, col(ky), col(kc), col("group"), col("split"), col("y_pred"), names, dataset)
