"""Device-resident stage-2 data path: the adapter training /
validation loops of the reference (final_main.py:426-496, 571-653, 655-719) without its host
bottlenecks.

Reference loop per step: DataLoader workers do pandas column lookups per item
(data/celeba_embeddings_reg.py:63-84), `.cuda()` copies, `loss.item()`, and `update_dict` pulls
`g` to the host and loops over `np.unique` (final_main.py:383-391).  Here the whole embedding
table lives in HBM, a batch is one row-gather kernel, the step body is one C call
(`classifier.train_step`), the group counters stay on the device, and the host synchronises once
per epoch.  Batch composition follows the reference exactly: `dataloader_shuffle_order`
reproduces `DataLoader(shuffle=True)`'s index stream from the global torch RNG, and
`adapter.balance_val_indices` the per-epoch group balancing from the global numpy RNG.
"""
from functools import partial

import numpy as np
import torch

from . import adapter, ops

NEW_ORDER_FOR_PRINT = ["weighted_mean_acc", "worst_acc", "acc_0_0", "acc_0_1", "acc_1_0", "acc_1_1", "mean_acc"]


class EmbeddingTable:
    """[N, D] fp32 embeddings + int64 labels resident on one device.

    Mirrors the attributes the reference datasets expose (data/celeba_embeddings_reg.py:40-57):
    n_classes, n_groups, n_places, group_array, group_counts, group_ratio."""

    def __init__(self, embeddings, y, confounder, y_pred=None, filenames=None, device="cuda"):
        y_np, c_np, g_np = adapter.group_index(np.asarray(y), np.asarray(confounder))
        self.device = torch.device(device)
        self.embeddings = torch.as_tensor(embeddings, dtype=torch.float32).contiguous().to(self.device)
        self.targets = torch.from_numpy(y_np).to(self.device)
        self.targets_spurious = torch.from_numpy(c_np).to(self.device)
        self.targets_group = torch.from_numpy(g_np).to(self.device)
        self.y_pred = None if y_pred is None else torch.as_tensor(np.asarray(y_pred), dtype=torch.int64).to(self.device)
        self.filenames = filenames
        self.group_array = g_np
        self.n_classes, self.n_groups, self.n_places = 2, 4, 2
        self.group_counts = (torch.arange(self.n_groups).unsqueeze(1) == torch.from_numpy(g_np)).sum(1).float()
        self.group_ratio = self.group_counts / len(self)

    def __len__(self):
        return self.embeddings.shape[0]

    def labels(self, target):
        return {"class": self.targets, "group": self.targets_group, "spurious": self.targets_spurious}[target]

    def batch(self, idx, target="class"):
        """(embeddings[idx], labels[idx], groups[idx]) -- idx int64 on the table's device"""
        return ops.gather_rows(self.embeddings, idx), self.labels(target)[idx], self.targets_group[idx]


def dataloader_shuffle_order(n):
    """Index order of one `DataLoader(dataset, shuffle=True)` epoch, drawn from the global torch
    RNG like torch.utils.data does: one int64 for the iterator's base seed, one for the
    RandomSampler's generator seed, then randperm with that generator."""
    torch.empty((), dtype=torch.int64).random_()                       # _BaseDataLoaderIter._base_seed
    seed = int(torch.empty((), dtype=torch.int64).random_().item())   # RandomSampler.__iter__
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


def _epoch_batches(n, batch_size, shuffle, indices):
    order = dataloader_shuffle_order(n) if shuffle else torch.arange(n)
    if indices is not None:
        order = torch.as_tensor(np.asarray(indices), dtype=torch.int64)[order]
    return [order[i:i + batch_size] for i in range(0, n, batch_size)]


def _results(counts, n_places=2):
    """get_results (final_main.py:395-406) from integer (n, correct) counters."""
    meters = {}
    for g in range(counts.shape[0]):
        m = adapter.AverageMeter()
        n, corr = int(counts[g, 0]), int(counts[g, 1])
        if n:
            m.update(corr / n, n)
        meters[g] = m
    return adapter.get_results(meters, partial(adapter.get_y_p, n_places=n_places))


def train_epoch(table, classifier, optimizer, batch_size, target="class", use_group=False, indices=None,
                shuffle=True, lr_hook=None):
    """One epoch of train_one_epoch / train_reg_seq_one_epoch.  `indices` restricts the epoch to a
    subset (reg split, balanced indices); `lr_hook(step, n_steps)` runs before every step (the
    warm-up helpers).  Returns (loss average, accuracy, group accuracy dict) like the reference,
    computed from device-side accumulators with ONE host sync at the end."""
    classifier.train()
    n = len(table) if indices is None else len(indices)
    batches = _epoch_batches(n, batch_size, shuffle, indices)
    dev = table.device
    counts = torch.zeros((table.n_groups, 2), dtype=torch.int64, device=dev)
    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    for step, idx in enumerate(batches):
        idx = idx.to(dev, non_blocking=True)
        emb, labels, groups = table.batch(idx, target)
        if use_group:
            labels = groups
        if lr_hook is not None:
            lr_hook(step, len(batches))
        loss, logits, _ = classifier.train_step(emb, labels, optimizer, use_group)
        loss_sum += loss.double() * idx.numel()                      # losses.update(loss.item(), bsz)
        adapter.group_counts(logits, labels, groups, table.n_groups, counts)
    c = counts.cpu().numpy()
    total = int(c[:, 0].sum())
    res = _results(c, table.n_places)
    group_acc = {k: np.round(res[k], 4) for k in NEW_ORDER_FOR_PRINT[1:]}
    return loss_sum.item() / n, int(c[:, 1].sum()) / total, group_acc


@torch.no_grad()
def validate(table, classifier, batch_size, train_group_ratio, target="class", indices=None, spurious=False):
    """validate / validate_zs (final_main.py:655-803): eval-mode forward, CE, group accuracies and
    the train-ratio-weighted mean."""
    classifier.eval()
    n = len(table) if indices is None else len(indices)
    dev = table.device
    counts = torch.zeros((table.n_groups, 2), dtype=torch.int64, device=dev)
    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    for idx in _epoch_batches(n, batch_size, False, indices):
        idx = idx.to(dev, non_blocking=True)
        emb, labels, groups = table.batch(idx, target)
        _, logits, rows = classifier.loss(emb, labels, spurious=spurious)   # fused normalise + logits + CE kernel
        loss_sum += rows.double().sum()
        adapter.group_counts(logits, labels, groups, table.n_groups, counts)
    c = counts.cpu().numpy()
    res = _results(c, table.n_places)
    indiv = [res[f"acc_{g // table.n_places}_{g % table.n_places}"] for g in range(table.n_groups)]
    res["weighted_mean_acc"] = (np.array(indiv) * np.array(train_group_ratio)).sum()
    group_acc = {k: np.round(res[k], 4) for k in NEW_ORDER_FOR_PRINT}
    return loss_sum.item() / n, int(c[:, 1].sum()) / int(c[:, 0].sum()), group_acc
