"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's
debiasing-adapter step (final_main.py + demo/util.py) in plain torch-CPU fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned by tests/golden/adapter_*.npz / indices.npz, produced by importing the
reference's own final_main.py classes (oracle/make_golden.py).

State is carried in plain dicts with the reference's state_dict key names
("layers.0.weight", ... final_main.py:167-172) so fixtures are interchangeable.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def clone_state(sd):
    return {k: v.clone() for k, v in sd.items()}


def adapter_forward(sd, x, train, prefix="layers."):
    """Adapter.forward, final_main.py:165-174: Linear -> BatchNorm1d -> ReLU -> Linear.

    BatchNorm1d in train mode normalises with the *biased* batch variance and updates
    running_var with the *unbiased* one, momentum 0.1.  Mutates
    sd's running stats in place like the module does."""
    h = x @ sd[prefix + "0.weight"].t() + sd[prefix + "0.bias"]
    if train:
        mean = h.mean(dim=0)
        var = ((h - mean) ** 2).mean(dim=0)
        with torch.no_grad():
            n = h.shape[0]
            sd[prefix + "1.running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            sd[prefix + "1.running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * n / max(n - 1, 1))
            sd[prefix + "1.num_batches_tracked"].add_(1)
    else:
        mean, var = sd[prefix + "1.running_mean"], sd[prefix + "1.running_var"]
    hn = (h - mean) * torch.rsqrt(var + BN_EPS) * sd[prefix + "1.weight"] + sd[prefix + "1.bias"]
    return F.relu(hn) @ sd[prefix + "3.weight"].t() + sd[prefix + "3.bias"]


def _logits(image_features, text, temperature):
    """final_main.py:77-78: text normalised over dim 0 of [D, C]; f @ t / T."""
    t = text / text.norm(dim=0, keepdim=True)
    return image_features @ t / temperature


def custom_clip_logits(sd, x, text, temperature=0.01, train=True, prefix="adapter.layers."):
    """CustomCLIP.forward / forward_spurious, final_main.py:66-92 (the text matrix passed
    in selects class / group / spurious prompts)."""
    f = adapter_forward(sd, x, train, prefix)
    f = f / f.norm(dim=-1, keepdim=True)                      # no epsilon (final_main.py:68)
    return _logits(f, text, temperature)


def multiple_adapter_logits(sd, x, text, temperature=0.01, train=True, ebd_weight=0.5):
    """MultipleAdapter.forward, final_main.py:121-140: both adapters (the old one too runs
    in train mode under classifier.train(), Appendix B), blend of the two *normalised*
    features with old detached and NOT re-normalised."""
    fo = adapter_forward(sd, x, train, "old_cls.adapter.layers.")
    fo = fo / fo.norm(dim=-1, keepdim=True)
    fn = adapter_forward(sd, x, train, "new_adapter.layers.")
    fn = fn / fn.norm(dim=-1, keepdim=True)
    f = ebd_weight * fo.detach() + (1 - ebd_weight) * fn
    return _logits(f, text, temperature)


def sgd_step(params, grads, bufs, lr, momentum=0.9, weight_decay=5e-5):
    """torch.optim.SGD as configured in demo/util.py:118-136 (dampening 0, no nesterov):
    g += wd*w; buf = g (first step) or mu*buf + g; w -= lr*buf."""
    for k in params:
        g = grads[k] + weight_decay * params[k]
        if bufs.get(k) is None:
            bufs[k] = g.clone()
        else:
            bufs[k] = momentum * bufs[k] + g
        params[k] -= lr * bufs[k]


TRAINABLE_SUFFIXES = ("0.weight", "0.bias", "1.weight", "1.bias", "3.weight", "3.bias")


def train_step(sd, bufs, x, labels, text, lr, temperature=0.01, multiple=False,
               momentum=0.9, weight_decay=5e-5):
    """One body of final_main.py:455-466 / :610-623: forward on x.detach(), mean CE,
    backward, SGD.  With multiple=True only names without "old_cls" are stepped
    (demo/util.py:128).  Returns (loss, logits, grads)."""
    names = [k for k in sd if k.endswith(TRAINABLE_SUFFIXES) and not (multiple and "old_cls" in k)]
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    work = dict(sd); work.update(leaves)
    fwd = multiple_adapter_logits if multiple else custom_clip_logits
    logits = fwd(work, x.detach(), text, temperature, train=True)
    loss = F.cross_entropy(logits, labels)
    grads_t = torch.autograd.grad(loss, [leaves[k] for k in names])
    grads = dict(zip(names, grads_t))
    with torch.no_grad():
        params = {k: sd[k] for k in names}
        sgd_step(params, grads, bufs, lr, momentum, weight_decay)
    return loss.detach(), logits.detach(), grads


def per_group_loss(logits, labels, groups, n_groups=4):
    """Build-side addition: CE(reduction='none') segment-meaned by group."""
    l = F.cross_entropy(logits, labels, reduction="none")
    out = torch.zeros(n_groups)
    for g in range(n_groups):
        m = groups == g
        out[g] = l[m].mean() if m.any() else 0.0
    return out


# ---------------------------------------------------------------------------------------
# integer / index work (bit-exact rows a19-a22 of SURVEY section 8a)
# ---------------------------------------------------------------------------------------

def group_counts(logits, y, g, n_groups=4):
    """update_dict, final_main.py:383-391, as integer (count, correct) per group."""
    pred = torch.argmax(logits, dim=1)
    correct = pred == y
    out = np.zeros((n_groups, 2), dtype=np.int64)
    for gv in np.unique(g.cpu().numpy()):
        m = g == int(gv)
        out[int(gv), 0] = int(m.sum())
        out[int(gv), 1] = int(correct[m].sum())
    return out


class Meter:
    """AverageMeter, demo/util.py:18-33."""
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def results_from_meters(meters, n_places=2):
    """get_results + get_y_p, final_main.py:395-412."""
    res = {f"acc_{g // n_places}_{g % n_places}": meters[g].avg for g in meters}
    all_correct = sum(meters[g].sum for g in meters)
    all_total = sum(meters[g].count for g in meters)
    res["mean_acc"] = all_correct / all_total
    res["worst_acc"] = min(res.values())
    return res


def group_index(y, confounder):
    """data/celeba_embeddings_reg.py:34-38: y[y==-1]=0; group = 2*y + confounder."""
    y = np.array(y, dtype=np.int64).copy(); c = np.array(confounder, dtype=np.int64).copy()
    y[y == -1] = 0; c[c == -1] = 0
    return y, c, (y * 2 + c).astype(np.int64)


def balance_val_indices(group_array, n_groups, batch_size_reg):
    """balance_val, final_main.py:346-379 on a plain group array: per-group
    np.random.shuffle (global numpy RNG), truncate to the smallest group, interleave."""
    g_idx = [np.where(group_array == g)[0] for g in range(n_groups)]
    min_g = np.min([len(g) for g in g_idx])
    for i, g in enumerate(g_idx):
        np.random.shuffle(g)
        g_idx[i] = g[:min_g]
    balanced = np.array(list(zip(*g_idx))).reshape(-1)
    return balanced, (batch_size_reg if batch_size_reg <= len(balanced) else len(balanced))
