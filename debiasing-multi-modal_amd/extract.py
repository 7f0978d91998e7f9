"""Stage-1 embedding extraction as a device pipeline (the writer side).

The reference's loop (/root/reference/clip_inference.py:188-271), per batch of 256 images: a blocking H2D copy (:205),
`encode_image` (:206), zero-shot logits / softmax / max (:207-216), minority flags on the host (:219-233), then -- per SAMPLE --
`.cpu().numpy().tolist()` of the embedding and four scalar `.cpu()` copies while a Python dict is filled (:235-257), and finally one
`json.dump` of everything (:263-271).  For CelebA that is ~1 M tiny D2H copies and ~4 GB of JSON text.

Here a batch is staged in pinned host memory and copied to the device on a side stream while the previous batch computes (double
buffered); the compute stream runs (device preprocessing of decoded uint8 images ->) `encode_image` -> fused normalise + logits +
argmax (`adapter.zeroshot_tail`) -> minority flags, packs embeddings + predictions + flags into ONE device buffer and issues ONE
asynchronous D2H copy per batch into a pinned result buffer; the host drains batch i - 1 into the memory-mapped store while batch i
runs.  The JSON files of the reference remain available through `store.export_json` (byte-identical text).

Inputs per batch are what the reference's DataLoader yields (clip_inference.py:203): `(image, (target, target_g, target_s,
target_split), file_name)` with `image` either the float32 [b, 3, R, R] tensor of the reference's PIL `preprocess`, or decoded RGB
uint8 [b, H, W, 3] of one geometry (CelebA: 218 x 178), which is preprocessed on the device (Pillow-exact, preprocess.py).
"""
import os
import time

import numpy as np
import torch

from . import adapter, preprocess, store, templates


@torch.no_grad()
def text_prompt_embeddings(model, dataset, tokenize):
    """clip_inference.py:57-84: every prompt encoded on its own (batch of one, `mean(dim=0)` of that single row), NOT normalised.
    Returns ({"class" | "spurious" | "group": {prompt: [D floats]}}, {"class" | ...: weights [D, C] on the model's device})."""
    dicts, weights = {}, {}
    dev = next(model.parameters()).device
    for which in ("class", "spurious", "group"):
        cols, d = [], {}
        for text in templates.prompts(dataset, which):
            e = model.encode_text(tokenize([text]).to(dev)).mean(dim=0)
            d[text] = e.clone().detach().cpu().numpy().tolist()
            cols.append(e)
        dicts[which], weights[which] = d, torch.stack(cols, dim=1)
    return dicts, weights


def row_key(dataset, file_name):
    """the key an image is stored under (clip_inference.py:237, :248): the last two path components for Waterbirds
    ("049.Black_footed_Albatross/x.jpg"), the base name for CelebA"""
    if dataset == "waterbirds":
        return "/".join(file_name.split("/")[-2:])
    return str(os.path.split(file_name)[-1])


class _Slot:
    """one of the two staging slots: pinned host input, device input, device result pack, pinned host result"""
    def __init__(self, b, in_shape, in_dtype, D, dev):
        self.h_in = torch.empty((b,) + in_shape, dtype=in_dtype).pin_memory()
        self.h_lab = torch.empty((b, 2), dtype=torch.int64).pin_memory()          # (target, target_s)
        self.d_in = torch.empty((b,) + in_shape, dtype=in_dtype, device=dev)
        self.d_lab = torch.empty((b, 2), dtype=torch.int64, device=dev)
        self.row_bytes = 4 * D + 24                                               # embedding fp32 [D] + pred, is_minor, is_minor_pred (int64)
        self.d_out = torch.empty(b * self.row_bytes, dtype=torch.uint8, device=dev)
        self.h_out = torch.empty(b * self.row_bytes, dtype=torch.uint8).pin_memory()
        self.copied = torch.cuda.Event()       # H2D of this slot's input finished (side stream)
        self.consumed = torch.cuda.Event()     # the compute stream no longer reads d_in / d_lab
        self.done = torch.cuda.Event()         # D2H of this slot's results finished (compute stream)
        self.meta = None                       # (rows, labels, keys) of the batch whose results are in flight
        self.staged = None                     # the same for the batch staged into h_in / d_in
        self.early = False


class Extractor:
    """clip_inference.py:188-271 for one model and dataset.  `run(batches, path, n_total)` writes the binary store and returns the
    (is_minor, is_minor_pred, pred) arrays the reference feeds to `classification_report` (:260)."""

    def __init__(self, model, zeroshot_weights, dataset, temperature=0.02, normalized=False, max_batch=1024):
        if dataset not in ("celeba", "waterbirds"):
            raise NotImplementedError(dataset)
        self.model, self.dataset, self.temperature, self.normalized = model, dataset, float(temperature), bool(normalized)
        self.dev = next(model.parameters()).device
        self.W = zeroshot_weights.to(self.dev).float().contiguous()               # [D, 2], used as given (not re-normalised, :69-71)
        self.D = int(self.W.shape[0])
        self.R = int(model.visual.input_resolution)
        self.max_batch = int(max_batch)
        self.side = torch.cuda.Stream(device=self.dev)
        self._slots = None
        # host seconds spent: staging into pinned memory, waiting for a batch's H2D copy (pinned sources only), enqueuing the device work,
        # waiting for the previous batch's results, writing rows into the store
        self.stats = {"batches": 0, "images": 0, "d2h_copies": 0, "h2d_copies": 0, "t_stage": 0.0, "t_h2d_wait": 0.0, "t_enqueue": 0.0,
                      "t_result_wait": 0.0, "t_store": 0.0}

    # ---- device work of one batch (compute stream) --------------------------------------------------------------------
    def _compute(self, slot, b):
        x = slot.d_in[:b]
        lab = slot.d_lab[:b].clone()                                              # (16 KB) so that the slot's inputs are dead early
        if x.dtype == torch.uint8:
            x = preprocess.preprocess_uniform(x, self.R)                          # Resize(BICUBIC) + CenterCrop + ToTensor + Normalize
            slot.consumed.record()                                                # the staged batch has been read: the slot may be refilled
            slot.early = True                                                     # while the encoder runs (H2D two batches ahead)
        else:
            slot.early = False
        f = self.model.encode_image(x)                                            # :206
        f32 = f.float().contiguous()
        logits, pred = adapter.zeroshot_tail(f32, self.W, self.temperature)       # :207-216 (argmax of softmax = argmax of logits)
        if self.normalized:                                                       # --normalized: the saved embedding is the unit vector
            f32 = f32 / f32.norm(dim=-1, keepdim=True) if f.dtype == torch.float32 else (f / f.norm(dim=-1, keepdim=True)).float()
        is_minor, is_minor_pred = adapter.minority_flags(self.dataset, lab[:, 0], lab[:, 1], pred)   # :219-233
        out = slot.d_out[:b * slot.row_bytes]
        nb = 4 * self.D * b
        out[:nb].view(torch.float32).view(b, self.D).copy_(f32)
        tail = out[nb:].view(torch.int64).view(3, b)
        tail[0].copy_(pred); tail[1].copy_(is_minor); tail[2].copy_(is_minor_pred)
        slot.h_out[:b * slot.row_bytes].copy_(out, non_blocking=True)             # the ONE D2H copy of this batch
        slot.done.record()
        self.stats["d2h_copies"] += 1

    def _drain(self, slot, writer, acc):
        b, targets, names = slot.meta
        t0 = time.perf_counter()
        slot.done.synchronize()
        t1 = time.perf_counter()
        self.stats["t_result_wait"] += t1 - t0
        raw = slot.h_out[:b * slot.row_bytes].numpy()
        emb = raw[:4 * self.D * b].view(np.float32).reshape(b, self.D)
        tail = raw[4 * self.D * b:].view(np.int64).reshape(3, b)
        target, target_g, target_s, target_split = targets
        writer.append(emb, target, target_s, target_g, target_split, tail[0], names)
        acc["pred"].append(tail[0].copy()); acc["is_minor"].append(tail[1].copy()); acc["is_minor_pred"].append(tail[2].copy())
        slot.meta = None
        self.stats["t_store"] += time.perf_counter() - t1

    def _stage(self, batch, k):
        """host side of batch k: into its slot's pinned buffers (or straight from the caller's pinned tensor) and onto the side stream"""
        image, labels, names = batch
        image = torch.as_tensor(image)
        b = int(image.shape[0])
        if b > self.max_batch:
            raise ValueError(f"batch of {b} images > max_batch {self.max_batch}")
        if self._slots is None or self._slots[0].h_in.shape[1:] != image.shape[1:] or self._slots[0].h_in.dtype != image.dtype:
            if image.dtype not in (torch.uint8, torch.float32, torch.float16):
                raise TypeError(f"images must be uint8 [b,H,W,3] or float [b,3,R,R], got {image.dtype}")
            if k:
                raise ValueError("all batches of one run must share geometry and dtype")
            torch.cuda.synchronize(self.dev)
            self._slots = [_Slot(self.max_batch, tuple(image.shape[1:]), image.dtype, self.D, self.dev) for _ in range(2)]
        slot = self._slots[k & 1]
        t0 = time.perf_counter()
        target, target_g, target_s, target_split = (np.asarray(t, dtype=np.int64).reshape(-1) for t in labels)
        # a batch that already lives in pinned memory (a decoder that writes there) is the DMA source itself
        direct = image.is_pinned() and image.is_contiguous()
        src = image if direct else slot.h_in[:b].copy_(image)
        slot.h_lab[:b, 0] = torch.from_numpy(target); slot.h_lab[:b, 1] = torch.from_numpy(target_s)
        with torch.cuda.stream(self.side):
            self.side.wait_event(slot.consumed)               # the batch staged here two batches ago has been read by its first kernel
            slot.d_in[:b].copy_(src, non_blocking=True)
            slot.d_lab[:b].copy_(slot.h_lab[:b], non_blocking=True)
            slot.copied.record(self.side)
        self.stats["h2d_copies"] += 2
        self.stats["t_stage"] += time.perf_counter() - t0
        slot.staged = (b, (target, target_g, target_s, target_split), [row_key(self.dataset, nm) for nm in names], direct)
        return slot

    def run(self, batches, path, n_total):
        """batches: iterable of (image, (target, target_g, target_s, target_split), file_names) with host tensors / arrays.
        Writes `path` (binary store, `n_total` rows) and returns dict(pred, is_minor, is_minor_pred) as int64 arrays.

        Submission order matters: the H2D copy of batch k + 1 is handed to the copy engine BEFORE the device work of batch k (which
        ends in that batch's D2H copy) is enqueued.  Submitted the other way round the H2D queues behind the D2H on the engine and
        starts only when batch k's encoder has finished (measured: 29 ms of host wait per batch, profiles/r04_extract_probe.log)."""
        writer = store.Writer(path, n_total, self.D, self.dataset)
        acc = {"pred": [], "is_minor": [], "is_minor_pred": []}
        main = torch.cuda.current_stream(self.dev)
        it = iter(batches)
        first = next(it, None)
        staged = self._stage(first, 0) if first is not None else None
        pending = None                                        # slot whose results are in flight
        k = 0
        while staged is not None:
            slot = staged
            b, targets, names, direct = slot.staged
            if direct:                                        # the caller may reuse its buffer once the next batch is requested
                t0 = time.perf_counter()
                slot.copied.synchronize()
                self.stats["t_h2d_wait"] += time.perf_counter() - t0
            nxt = next(it, None)
            staged = self._stage(nxt, k + 1) if nxt is not None else None          # H2D of batch k + 1: ahead of batch k's D2H
            t0 = time.perf_counter()
            main.wait_event(slot.copied)
            self._compute(slot, b)
            if not slot.early:
                slot.consumed.record(main)                    # float input: the encoder's first kernel reads it; released after the batch
            slot.meta = (b, targets, names)
            self.stats["t_enqueue"] += time.perf_counter() - t0
            if pending is not None:
                self._drain(pending, writer, acc)             # batch k - 1: its D2H finished while batch k was being issued
            pending = slot
            k += 1
            self.stats["batches"] += 1; self.stats["images"] += b
        if pending is not None:
            self._drain(pending, writer, acc)
        writer.close()
        return {k_: (np.concatenate(v) if v else np.zeros(0, np.int64)) for k_, v in acc.items()}
