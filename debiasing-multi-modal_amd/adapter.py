"""Debiasing-adapter classes and step helpers with the reference's call signatures
(/root/reference/final_main.py:43-174, 383-424) over the MI355X kernels.

Module protocol kept: ordinary nn.Parameters, state-dict keys
`adapter.layers.{0,1,3}.*` / `old_cls.adapter.layers.*` / `new_adapter.layers.*`,
`.train()/.eval()`, `copy.deepcopy`, `loss.backward()` through torch.autograd.Function
wrappers whose forward/backward call the C ABI.  There is no torch fallback: CPU tensors
raise.

Two ways to take a training step:
  * drop-in: `logits = classifier(x.detach(), use_group)` then any torch criterion
    (final_main.py:455-466) -- the criterion's d(loss)/d(logits) is fed to the fused
    normalise+similarity backward kernel.
  * fused:   `loss, logits = classifier.loss(x, labels, use_group)` -- row L2-norm, image x
    text logits and mean cross-entropy in one kernel (and one backward kernel).
"""
import json
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops


class LinearClassifier(nn.Module):
    """Linear probing head (final_main.py:43-49) on the MFMA GEMM."""
    def __init__(self, input_dim, num_classes=2):
        super().__init__()
        self.fc = nn.Linear(input_dim, num_classes)

    def forward(self, features):
        return _LinearFn.apply(features, self.fc.weight, self.fc.bias)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        return ops.gemm(x, w.detach(), b.detach())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        # dW[n][k] = sum_b g[b][n] x[b][k]: the batch-major operand needs 16-B rows, so the
        # num_classes columns of g are zero-padded to a multiple of 4.
        n = w.shape[0]
        npad = (n + 3) // 4 * 4
        gp = torch.zeros((g.shape[0], npad), device=g.device)
        gp[:, :n] = g
        dw = ops.gemm(gp, x, trans_a=True, trans_w=True)[:n]
        db = ops.colsum(gp)[:n]
        dx = None
        if ctx.needs_input_grad[0]:
            wp = torch.zeros((npad, w.shape[1]), device=w.device); wp[:n] = w
            dx = ops.gemm(gp, wp, trans_w=True)
        return dx, dw, db


class _AdapterFn(torch.autograd.Function):
    """Adapter.forward as one fused op: Linear -> BatchNorm1d -> ReLU -> Linear."""

    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, w2, b2, bn, training):
        x = x.contiguous()
        if training and x.shape[0] < 2:
            raise ValueError("Expected more than 1 value per channel when training (BatchNorm1d)")
        z, h, mean, invstd, r = ops.adapter_fwd(x, w1.detach(), b1.detach(), gamma.detach(), beta.detach(),
                                                bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                w2.detach(), b2.detach(), training, bn.eps, bn.momentum)
        if not training:
            # eval-mode backward is not needed by the reference (validate() runs under no_grad)
            mean, invstd = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
        ctx.save_for_backward(x, h, mean, invstd, r, gamma, beta, w1, w2)
        ctx.training = training
        return z

    @staticmethod
    def backward(ctx, dz):
        x, h, mean, invstd, r, gamma, beta, w1, w2 = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("dbmm_amd Adapter: backward in eval mode is not implemented "
                               "(the reference never back-propagates through eval-mode BatchNorm)")
        dw1, db1, dgamma, dbeta, dw2, db2, dh = ops.adapter_bwd(x, dz.contiguous(), h, mean, invstd, r,
                                                                gamma.detach(), beta.detach(), w2.detach())
        dx = ops.gemm(dh, w1.detach(), trans_w=True) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dgamma, dbeta, dw2, db2, None, None


class Adapter(nn.Module):
    """final_main.py:160-174; `layers` keeps the reference's Sequential so the state-dict keys
    (`layers.0`, `layers.1`, `layers.3`) and deepcopy/load_state_dict behave identically."""
    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.BatchNorm1d(hidden_dim), nn.ReLU(),
                                    nn.Linear(hidden_dim, input_dim))

    def forward(self, features):
        l0, bn, l3 = self.layers[0], self.layers[1], self.layers[3]
        return _AdapterFn.apply(features, l0.weight, l0.bias, bn.weight, bn.bias, l3.weight, l3.bias, bn,
                                self.training)


class _SimFn(torch.autograd.Function):
    """logits = (w*norm(z_old) + (1-w)*norm(z)) @ colnorm(text) / T with z_old detached."""

    @staticmethod
    def forward(ctx, z, z_old, tn, temperature, ebd_weight):
        logits, _, _, _, inv_norm = ops.l2norm_sim_ce_fwd(z, tn, temperature, z_old=z_old, ebd_weight=ebd_weight)
        ctx.save_for_backward(z, inv_norm, tn)
        ctx.cfg = (temperature, ebd_weight, z_old is not None)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        z, inv_norm, tn = ctx.saved_tensors
        T, w, blended = ctx.cfg
        dz = ops.l2norm_sim_ce_bwd(z, inv_norm, tn, T, dlogits=dlogits.contiguous(), blended=blended, ebd_weight=w)
        return dz, None, None, None, None


class _SimCEFn(torch.autograd.Function):
    """Fused logits + mean cross-entropy; returns (loss, logits)."""

    @staticmethod
    def forward(ctx, z, z_old, tn, temperature, ebd_weight, labels):
        logits, loss_rows, loss_mean, _, inv_norm = ops.l2norm_sim_ce_fwd(z, tn, temperature, labels=labels,
                                                                          z_old=z_old, ebd_weight=ebd_weight)
        ctx.save_for_backward(z, inv_norm, tn, logits, labels)
        ctx.cfg = (temperature, ebd_weight, z_old is not None)
        ctx.mark_non_differentiable(logits, loss_rows)
        return loss_mean, logits, loss_rows

    @staticmethod
    def backward(ctx, gloss, _gl, _gr):
        z, inv_norm, tn, logits, labels = ctx.saved_tensors
        T, w, blended = ctx.cfg
        # gloss is a 0-dim device tensor (1.0 for loss.backward()); fold it in on device
        dz = ops.l2norm_sim_ce_bwd(z, inv_norm, tn, T, logits=logits, labels=labels, blended=blended, ebd_weight=w)
        return dz * gloss, None, None, None, None, None


def _step_key(new_ad, old_ad, optimizer):
    """every device address the fused step's argument block holds (15 of the trainable adapter incl. its six momentum buffers, 9 of the
    frozen one), or None while a momentum buffer does not exist yet"""
    key = []
    state = optimizer.state
    for ad, trainable in ((new_ad, True), (old_ad, False)):
        if ad is None:
            continue
        mods = ad.layers._modules
        l0, bn, l3 = mods["0"], mods["1"], mods["3"]
        ps = (l0._parameters["weight"], l0._parameters["bias"], bn._parameters["weight"], bn._parameters["bias"],
              l3._parameters["weight"], l3._parameters["bias"])
        bufs = bn._buffers
        key += [p.data_ptr() for p in ps]
        key += [bufs["running_mean"].data_ptr(), bufs["running_var"].data_ptr(), bufs["num_batches_tracked"].data_ptr()]
        if trainable:
            for p in ps:
                mb = state[p].get("momentum_buffer") if p in state else None
                if mb is None:
                    return None
                key.append(mb.data_ptr())
    key.append(id(optimizer))
    return tuple(key)


_text_cache = {}


def get_text_embedding(text_embedding_dir):
    """JSON {prompt: [D floats]} -> [D, C] (final_main.py:414-424; insertion order = columns).
    Parsed once per (path, mtime): the reference re-reads the group file every forward
    (final_main.py:72,131), which changes nothing numerically."""
    st = os.stat(text_embedding_dir)
    key = (os.path.abspath(text_embedding_dir), st.st_mtime_ns, st.st_size)
    if key not in _text_cache:
        with open(text_embedding_dir, "r") as f:
            d = json.load(f)
        _text_cache[key] = torch.stack([torch.tensor(v) for v in d.values()], dim=1)
    return _text_cache[key].clone()


class _TextBank:
    """device copies + column-normalised [C, D] forms of the prompt matrices."""
    def __init__(self):
        self._tn = {}

    def normalised(self, text):
        key = (text.data_ptr(), text._version, tuple(text.shape), str(text.device))
        tn = self._tn.get(key)
        if tn is None:
            tn = ops.text_colnorm(text.contiguous().float())
            self._tn = {key: tn} if len(self._tn) > 8 else {**self._tn, key: tn}
        return tn


class CustomCLIP(nn.Module):
    """final_main.py:53-92."""
    def __init__(self, adapter, text_embedding_dir, text_spurious_embedding_dir, text_group_embedding_dir,
                 temperature=0.01):
        super().__init__()
        self.text_embedding_dir = text_embedding_dir
        self.text_spurious_embedding_dir = text_spurious_embedding_dir
        self.text_group_embedding_dir = text_group_embedding_dir
        self.adapter = adapter
        self.temperature = temperature
        self.text_features = get_text_embedding(text_embedding_dir)
        self.n_cls = self.text_features.shape[0]       # = D, kept for compatibility (final_main.py:63)
        self.text_spurious_features = get_text_embedding(text_spurious_embedding_dir)
        self._bank = _TextBank()
        self._group = None

    def _apply(self, fn, *a, **k):
        # the reference keeps the text matrices as plain attributes moved with .cuda() at
        # construction; here they follow the module
        self.text_features = fn(self.text_features)
        self.text_spurious_features = fn(self.text_spurious_features)
        self._group = None
        return super()._apply(fn, *a, **k)

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_step_plan":                    # raw device addresses of THIS module's tensors: the copy builds its own
                continue
            new.__dict__[k] = _TextBank() if k == "_bank" else copy.deepcopy(v, memo)
        return new

    def __getstate__(self):                          # torch.save(module) / pickle: same rule as __deepcopy__
        state = dict(self.__dict__)
        state.pop("_step_plan", None)
        state["_bank"] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__["_bank"] = _TextBank()

    def _text(self, which, device):
        if which == "group":
            if self._group is None or self._group.device != device:
                self._group = get_text_embedding(self.text_group_embedding_dir).to(device)
            t = self._group
        elif which == "spurious":
            t = self.text_spurious_features
        else:
            t = self.text_features
        if t.device != device:
            t = t.to(device)
        return self._bank.normalised(t)

    def _features(self, features):
        return self.adapter(features), None

    def forward(self, features, use_group=False):
        z, z_old = self._features(features)
        tn = self._text("group" if use_group else "class", features.device)
        return _SimFn.apply(z, z_old, tn, self.temperature, getattr(self, "ebd_weight", 0.5))

    def forward_spurious(self, features):
        z, z_old = self._features(features)
        tn = self._text("spurious", features.device)
        return _SimFn.apply(z, z_old, tn, self.temperature, getattr(self, "ebd_weight", 0.5))

    def loss(self, features, labels, use_group=False, spurious=False):
        """fused step body: returns (mean CE, logits, per-row CE); `spurious`: against the spurious-attribute prompts
        (forward_spurious + criterion, final_main.py:764-766)."""
        z, z_old = self._features(features)
        tn = self._text("spurious" if spurious else "group" if use_group else "class", features.device)
        return _SimCEFn.apply(z, z_old, tn, self.temperature, getattr(self, "ebd_weight", 0.5), labels)

    def _step_adapters(self):
        """(trainable adapter, frozen old adapter or None)"""
        return self.adapter, None

    def train_step(self, features, labels, optimizer, use_group=False):
        """The whole step body of final_main.py:455-466 / :610-623 -- forward, mean CE, backward and
        the SGD-momentum update -- as ONE C call (~20 launches back to back, no autograd graph, no
        host round trip).  Uses the optimiser's lr / momentum / weight_decay and its
        `momentum_buffer` state, so it can be mixed freely with `loss.backward(); optimizer.step()`.
        Requires train mode.  Returns (mean CE, logits, per-row CE), all on the device."""
        if not self.training:
            raise RuntimeError("train_step needs classifier.train()")
        new_ad, old_ad = self._step_adapters()
        group = optimizer.param_groups[0]
        plan = self.__dict__.get("_step_plan")
        first = False
        # The argument block of the C call holds 24 raw pointers.  It is rebuilt whenever ANY tensor it points at was replaced or
        # moved (a parameter / buffer / momentum buffer re-assigned, `.data` swapped, the old adapter reloaded, another optimiser):
        # the key is the tuple of every tensor's data_ptr(), read through the modules' own dicts (~4 us; the step is launch-bound).
        key = _step_key(new_ad, old_ad, optimizer)
        if plan is None or plan["key"] != key or key is None:
            def pack(ad):
                l0, bn, l3 = ad.layers[0], ad.layers[1], ad.layers[3]
                return (l0.weight, l0.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                        l3.weight, l3.bias)
            new = pack(new_ad)
            trainable = [new[0], new[1], new[2], new[3], new[7], new[8]]
            owned = {id(p) for g in optimizer.param_groups for p in g["params"]}
            if len(optimizer.param_groups) != 1 or any(id(p) not in owned for p in trainable):
                raise RuntimeError("train_step: the optimiser must hold the adapter's six tensors in one param group")
            first = any("momentum_buffer" not in optimizer.state[p] for p in trainable)
            bufs = []
            for p in trainable:
                st = optimizer.state[p]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                bufs.append(st["momentum_buffer"])
            # plain integer addresses (no ctypes objects, no module / optimiser references): the plan never keeps anything alive
            # and never travels with the module (__deepcopy__ / __getstate__ drop it)
            plan = dict(key=_step_key(new_ad, old_ad, optimizer),
                        args=ops.adapter_step_args([t.data for t in new], bufs,
                                                   [t.data for t in pack(old_ad)] if old_ad is not None else None),
                        H=new[0].shape[0], with_old=old_ad is not None)
            self.__dict__["_step_plan"] = plan
        tn = self._text("group" if use_group else "class", features.device)
        with torch.no_grad():
            return ops.adapter_train_step(
                features.detach().contiguous(), labels.contiguous(), plan["args"], plan["H"], plan["with_old"],
                getattr(self, "ebd_weight", 0.5), tn, self.temperature, group["lr"], group.get("momentum", 0.0),
                group.get("weight_decay", 0.0), first)


class MultipleAdapter(CustomCLIP):
    """final_main.py:97-158.  Quirks kept: the old adapter also runs in
    train mode under `.train()`, its feature is detached, the blend is not re-normalised."""
    def __init__(self, old_cls, new_adapter, init_near_identity=True, ebd_weight=0.5):
        nn.Module.__init__(self)
        self.old_cls = old_cls
        self.text_embedding_dir = old_cls.text_embedding_dir
        self.text_spurious_embedding_dir = old_cls.text_spurious_embedding_dir
        self.text_group_embedding_dir = old_cls.text_group_embedding_dir
        self.text_features = get_text_embedding(self.text_embedding_dir)
        self.n_cls = self.text_features.shape[0]
        self.text_spurious_features = get_text_embedding(self.text_spurious_embedding_dir)
        self.new_adapter = new_adapter
        self.ebd_weight = ebd_weight
        if init_near_identity:
            print("Initialize paramters of [New adapter] from [Old adapter]")
            self.new_adapter.load_state_dict(self.old_cls.adapter.state_dict())
        self.temperature = old_cls.temperature
        self._bank = _TextBank()
        self._group = None

    def _features(self, features):
        with torch.no_grad():
            z_old = self.old_cls.adapter(features)     # detached branch; BN stats still update in train mode
        return self.new_adapter(features), z_old

    def _step_adapters(self):
        return self.new_adapter, self.old_cls.adapter


# ---------------------------------------------------------------------------------------
# metrics (final_main.py:383-412, demo/util.py:18-46)
# ---------------------------------------------------------------------------------------

class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def group_counts(logits, y, g, n_groups, counts=None):
    """device-side (n, correct) per group as int64 [G, 2]; accumulates into `counts`."""
    if counts is None:
        counts = torch.zeros((n_groups, 2), dtype=torch.int64, device=logits.device)
    return ops.group_count(logits.detach().contiguous().float(), y.contiguous(), g.to(logits.device).contiguous(), counts)


def update_dict(acc_groups, y, g, logits):
    """final_main.py:383-391 with the argmax / compare / per-group counting on the GPU and a
    single 64-byte D2H copy instead of 2 syncs per group."""
    counts = group_counts(logits, y, g, len(acc_groups)).cpu().numpy()
    for g_val in range(counts.shape[0]):
        n, corr = int(counts[g_val, 0]), int(counts[g_val, 1])
        if n:
            acc_groups[g_val].update(corr / n, n)


def get_y_p(g, n_places):
    return g // n_places, g % n_places


def get_results(acc_groups, get_yp_func):
    groups = acc_groups.keys()
    results = {f"acc_{get_yp_func(g)[0]}_{get_yp_func(g)[1]}": acc_groups[g].avg for g in groups}
    all_correct = sum(acc_groups[g].sum for g in groups)
    all_total = sum(acc_groups[g].count for g in groups)
    results.update({"mean_acc": all_correct / all_total})
    results.update({"worst_acc": min(results.values())})
    return results


def accuracy(output, target, batch_size=None):
    with torch.no_grad():
        return torch.sum(output.argmax(dim=1).eq(target)).item() / target.size(0)


def per_group_loss(loss_rows, g, n_groups=4):
    """mean CE per group (build-side addition) -> float32 [G] on device."""
    sums = torch.zeros((n_groups,), device=loss_rows.device, dtype=torch.float32)
    ops.group_loss_sum(loss_rows.contiguous(), g.contiguous(), sums)
    n = torch.bincount(g, minlength=n_groups).clamp_min(1).float()
    return sums / n


def group_index(y, confounder):
    """data/celeba_embeddings_reg.py:34-38: -1 -> 0 recode, group = 2*y + confounder (int64)."""
    y = np.array(y, dtype=np.int64).copy()
    c = np.array(confounder, dtype=np.int64).copy()
    y[y == -1] = 0
    c[c == -1] = 0
    return y, c, y * 2 + c


def balance_val_indices(group_array, n_groups, batch_size_reg):
    """Index generation of balance_val (final_main.py:346-379) on a plain group array; uses the
    global numpy RNG exactly like the reference so seeds reproduce its epochs."""
    g_idx = [np.where(group_array == g)[0] for g in range(n_groups)]
    min_g = np.min([len(g) for g in g_idx])
    for i, g in enumerate(g_idx):
        np.random.shuffle(g)
        g_idx[i] = g[:min_g]
    balanced = np.array(list(zip(*g_idx))).reshape(-1)
    return balanced, (batch_size_reg if batch_size_reg <= len(balanced) else len(balanced))


def stratified_split_indices(group_array, test_size=0.5):
    """Index arrays of stratified_split_dataset (data/celeba_embeddings_reg.py:95-102): the
    reference delegates to sklearn's train_test_split with random_state=42 stratified on the group
    array, and so does this (same dependency, same RNG stream -> identical indices)."""
    from sklearn.model_selection import train_test_split
    group_array = np.asarray(group_array)
    reg_idx, val_idx = train_test_split(np.arange(len(group_array)), test_size=test_size, random_state=42,
                                        stratify=group_array)
    return reg_idx, val_idx


def stratified_split_dataset(dataset, test_size=0.5):
    """data/celeba_embeddings_reg.py:95-107: two torch Subsets (reg, val) of a dataset that carries
    `.group_array`."""
    reg_idx, val_idx = stratified_split_indices(dataset.group_array, test_size)
    return torch.utils.data.Subset(dataset, reg_idx), torch.utils.data.Subset(dataset, val_idx)


def minority_flags(dataset, target, target_s, pred):
    """clip_inference.py:219-233."""
    if dataset == "waterbirds":
        is_minor_pred = (((target == 0) & (pred == 1)) | ((target == 1) & (pred == 0))).long()
        is_minor = (((target == 0) & (target_s == 1)) | ((target == 1) & (target_s == 0))).long()
    elif dataset == "celeba":
        is_minor_pred = ((target == 1) & (pred == 1)).long()
        is_minor = ((target == 1) & (target_s == 1)).long()
    else:
        raise NotImplementedError(dataset)
    return is_minor, is_minor_pred


def zeroshot_tail(image_features, zeroshot_weights, temperature=0.02):
    """clip_inference.py:207-216 fused: normalise rows, @ W / T, argmax (softmax is monotone,
    so `pred` equals torch.max(probs)); returns logits and int64 pred.  The input is left
    un-normalised like the `--normalized`-off branch that saves raw embeddings."""
    f = image_features.float().contiguous()
    W = zeroshot_weights.float().contiguous()
    tn = W.t().contiguous()                      # zero-shot weights are used as given (not re-normalised)
    logits, _, _, pred, _ = ops.l2norm_sim_ce_fwd(f, tn, temperature, want_pred=True)
    return logits, pred
