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
    if shuffle:
        order = dataloader_shuffle_order(n)
    else:
        torch.empty((), dtype=torch.int64).random_()                   # every DataLoader iterator draws its base seed, shuffled or not:
        order = torch.arange(n)                                        # the global torch RNG advances like the reference's loops
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
                shuffle=True, lr_hook=None, stats=None):
    """One epoch of train_one_epoch / train_reg_seq_one_epoch.  `indices` restricts the epoch to a
    subset (reg split, balanced indices); `lr_hook(step, n_steps)` runs before every step (the
    warm-up helpers).  Returns (loss average, accuracy, group accuracy dict) like the reference,
    computed from device-side accumulators with ONE host sync at the end.  `stats` (a dict) receives
    the integer (n, correct) counters [G, 2] and the row order of the epoch."""
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
    if stats is not None:
        stats.update(counts=c, order=torch.cat(batches).numpy())
    return loss_sum.item() / n, int(c[:, 1].sum()) / total, group_acc


@torch.no_grad()
def validate(table, classifier, batch_size, train_group_ratio, target="class", indices=None, spurious=False, stats=None):
    """validate / validate_zs (final_main.py:655-803): eval-mode forward, CE, group accuracies and
    the train-ratio-weighted mean.  A LinearClassifier (tl_method linear_probing) is scored by validate_zs's
    own branch (:730-761): normalised raw embeddings @ column-normalised text / T -- pass `zs_text` via zeroshot()."""
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
    if stats is not None:
        stats.update(counts=c)
    res = _results(c, table.n_places)
    indiv = [res[f"acc_{g // table.n_places}_{g % table.n_places}"] for g in range(table.n_groups)]
    res["weighted_mean_acc"] = (np.array(indiv) * np.array(train_group_ratio)).sum()
    group_acc = {k: np.round(res[k], 4) for k in NEW_ORDER_FOR_PRINT}
    return loss_sum.item() / n, int(c[:, 1].sum()) / int(c[:, 0].sum()), group_acc


@torch.no_grad()
def validate_zs_linear_probing(table, text_embedding_dir, temperature, batch_size, train_group_ratio, target="class", stats=None):
    """validate_zs's `linear_probing` branch (final_main.py:730-761): no classifier -- raw embeddings, row-normalised, against the
    column-normalised prompt matrix / T (the CLIP zero-shot baseline), CE and group accuracies.  One fused launch per batch."""
    dev = table.device
    tn = ops.text_colnorm(adapter.get_text_embedding(text_embedding_dir).to(dev).float().contiguous())
    counts = torch.zeros((table.n_groups, 2), dtype=torch.int64, device=dev)
    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    n = len(table)
    for idx in _epoch_batches(n, batch_size, False, None):
        idx = idx.to(dev, non_blocking=True)
        emb, labels, groups = table.batch(idx, target)
        logits, rows, _, _, _ = ops.l2norm_sim_ce_fwd(emb, tn, temperature, labels=labels)
        loss_sum += rows.double().sum()
        adapter.group_counts(logits, labels, groups, table.n_groups, counts)
    c = counts.cpu().numpy()
    if stats is not None:
        stats.update(counts=c)
    res = _results(c, table.n_places)
    indiv = [res[f"acc_{g // table.n_places}_{g % table.n_places}"] for g in range(table.n_groups)]
    res["weighted_mean_acc"] = (np.array(indiv) * np.array(train_group_ratio)).sum()
    return loss_sum.item() / n, int(c[:, 1].sum()) / int(c[:, 0].sum()), {k: np.round(res[k], 4) for k in NEW_ORDER_FOR_PRINT}


def train_all_epochs(opt, train_table, val_table, test_table, input_dim=None, log=None):
    """The training schedule of the reference's driver (final_main.py:805-1046) for the adapter methods -- `adapter`,
    `adapter_reg_seq`, `adapter_reg_seq_alter`, with or without `--add_adapter`, `--balance_val`, `--continue_from_best` -- on
    device-resident tables, every step one fused C call:

      stage 1 (epoch <= epochs_feature_learning): train_one_epoch on the train split (:935), lr by adjust_learning_rate + warm-up;
      switch (epoch == efl + 1): restart from the best model so far (:941-943), MultipleAdapter over it with a fresh Adapter and
          set_optimizer_reg (fresh momentum) (:945-950);
      stage 2: train_reg_seq_one_epoch on the reg half of the validation split -- re-balanced per epoch by balance_val (:920-921)
          from the global numpy RNG, else shuffled like its DataLoader -- odd epochs on the class prompts, even epochs on the group
          prompts with group labels (`_alter`, :954-968), warm-up from the stage's own epoch count (:607);
      every epoch: validate on the other half of the validation split, keep a deepcopy of the best worst-group model (:1001-1008),
          validate on the test split (:1013-1016); finally zero-shot class / spurious scores of the best model (:1037-1045).

    `opt` = the namespace of the reference's parse_option (same field names).  Random streams are consumed like the reference does
    (global torch RNG: parameter initialisation and DataLoader orders; global numpy RNG: balance_val), so the same seeds give the same
    initial weights and batches.  Returns ((best train, best val, best test group-accuracy dicts), (zero-shot class, zero-shot
    spurious)) like the reference; `log` (a list) receives one record per train / validate pass."""
    from copy import deepcopy

    from . import optim as O
    if opt.tl_method not in ("adapter", "adapter_reg_seq", "adapter_reg_seq_alter"):
        raise ValueError(f"train_all_epochs covers the adapter methods, not tl_method={opt.tl_method!r}")
    two_stage = opt.tl_method != "adapter"
    dev = train_table.device
    D = input_dim or train_table.embeddings.shape[1]
    reg_idx = val_idx = None
    if two_stage:                                                         # load_*_embeddings: stratified 50/50 split of the val split
        reg_idx, val_idx = adapter.stratified_split_indices(val_table.group_array, 0.5)
    ratio = train_table.group_ratio.numpy()
    rec = (lambda **k: log.append(k)) if log is not None else (lambda **k: None)

    classifier = adapter.CustomCLIP(adapter.Adapter(D, opt.adapter_feat_dim), opt.text_embedding_dir, opt.text_spurious_embedding_dir,
                                    opt.text_group_embedding_dir, temperature=opt.zs_temperature)
    rec(kind="init", state={k: v.clone() for k, v in classifier.adapter.state_dict().items()})
    classifier = classifier.to(dev)
    optimizer = O.set_optimizer(opt, classifier)
    multiple_adapter = optimizer_reg = best_model = None
    best_acc, best_epoch = 0, 0
    train_accs, val_accs, test_accs = [], [], []
    # the reference evaluates in batches of batch_size_reg (as few as 4 rows: load_*_embeddings' bs_val); eval-mode scores do not
    # depend on the batch size, so evaluation runs in large batches -- one pass still advances the random stream once, like a loader
    bs_eval = max(opt.batch_size_reg if two_stage else opt.batch_size, 4096)
    efl = getattr(opt, "epochs_feature_learning", None) if two_stage else None
    for epoch in range(1, opt.epochs + 1):
        O.adjust_learning_rate(opt, optimizer, epoch)
        balanced = None
        if two_stage and opt.balance_val:                                # a fresh balanced subset every epoch, also in stage 1 (:920-921)
            balanced, bs_reg = adapter.balance_val_indices(val_table.group_array[reg_idx], val_table.n_groups, opt.batch_size_reg)
        st = {}
        stage2 = two_stage and epoch > efl
        if not stage2:
            hook = lambda i, n, e=epoch: O.warmup_learning_rate(opt, e, i, n, optimizer)
            loss, acc, gacc = train_epoch(train_table, classifier, optimizer, opt.batch_size, target=opt.train_target, lr_hook=hook, stats=st)
            rec(kind="train1", epoch=epoch, loss=loss, acc=acc, group_acc=gacc, **st)
        else:
            if epoch == efl + 1:
                if opt.continue_from_best:
                    classifier = deepcopy(best_model)
                if opt.add_adapter:
                    new_adapter = adapter.Adapter(D, opt.adapter_feat_dim)
                    rec(kind="init", state={k: v.clone() for k, v in new_adapter.state_dict().items()})
                    multiple_adapter = adapter.MultipleAdapter(classifier, new_adapter, init_near_identity=opt.init_near_identity,
                                                               ebd_weight=0.5).to(dev)
                    optimizer_reg = O.set_optimizer_reg(opt, multiple_adapter)
                else:
                    optimizer_reg = O.set_optimizer_reg(opt, classifier)
            O.adjust_learning_rate_reg(opt, optimizer_reg, epoch)
            model = multiple_adapter if opt.add_adapter else classifier
            if opt.tl_method == "adapter_reg_seq_alter":
                use_group = (epoch % 2) == 0
            else:
                use_group = not opt.use_cls_prompt_in_reg
            hook = lambda i, n, e=epoch: O.warmup_learning_rate_reg(opt, e - efl, i, n, optimizer_reg)
            if balanced is not None:                                     # DataLoader(balanced_subset, shuffle=False, batch_size=adjusted)
                rows, shuffle, bs = reg_idx[balanced], False, bs_reg
            else:                                                        # the reg loader itself: shuffle=True
                rows, shuffle, bs = reg_idx, True, opt.batch_size_reg
            loss, acc, gacc = train_epoch(val_table, model, optimizer_reg, bs, target=opt.train_target, use_group=use_group, indices=rows,
                                          shuffle=shuffle, lr_hook=hook, stats=st)
            rec(kind="train2", epoch=epoch, use_group=use_group, loss=loss, acc=acc, group_acc=gacc, **st)
        train_accs.append(gacc)
        model = multiple_adapter if (stage2 and opt.add_adapter) else classifier
        st = {}
        vloss, vacc, vg = validate(val_table, model, bs_eval, ratio, target=opt.train_target, indices=val_idx, stats=st)
        rec(kind="validate", epoch=epoch, split="val", loss=vloss, acc=vacc, group_acc=vg, **st)
        val_accs.append(vg)
        if vg["worst_acc"] > best_acc:
            best_acc, best_epoch, best_model = vg["worst_acc"], epoch, deepcopy(model)
        st = {}
        tloss, tacc, tg = validate(test_table, model, bs_eval, ratio, target="class", stats=st)
        rec(kind="validate", epoch=epoch, split="test", loss=tloss, acc=tacc, group_acc=tg, **st)
        test_accs.append(tg)
    st = {}
    zs = validate(test_table, best_model, bs_eval, ratio, target="class", stats=st)
    rec(kind="validate_zs", target="class", loss=zs[0], acc=zs[1], group_acc=zs[2], **st)
    st = {}
    zss = validate(test_table, best_model, bs_eval, ratio, target="spurious", spurious=True, stats=st)
    rec(kind="validate_zs", target="spurious", loss=zss[0], acc=zss[1], group_acc=zss[2], **st)
    rec(kind="final", best_epoch=best_epoch, best_model=best_model)
    return (train_accs[best_epoch - 1], val_accs[best_epoch - 1], test_accs[best_epoch - 1]), (zs[2], zss[2])
