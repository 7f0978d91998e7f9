"""Data-parallel "embed + adapter step" (SURVEY.md section 8e).

The image batch shards across ranks (encode_image is per-image independent: frozen weights,
eval-mode BN).  The adapter step is NOT row-separable (BatchNorm1d batch statistics, the
batch-mean CE and the SGD update are defined over the global batch in the single-process
reference), so each rank all-gathers the per-rank embeddings [B_l, D] and labels over
RCCL/xGMI and then runs the identical full-batch adapter step redundantly: parameters stay
bit-identical on every rank by construction -- no gradient all-reduce, no SyncBN, and the
N-GPU result equals the 1-GPU result.  Rank-major gather order = the original row order.

Overlap (BASELINE configs[3], "overlap all-gather with adapter GEMM"): with `micro_batches = k` the local batch is encoded
in k row chunks; as soon as chunk j is encoded its all-gather is issued asynchronously from a side stream (RCCL runs it
on its own stream over xGMI) while chunk j + 1 encodes on the compute stream; the compute stream waits for the k
collectives once, in front of the adapter step.  Only the last chunk's gather (B_l / k rows) is exposed.
"""
import torch
import torch.distributed as dist

from . import adapter


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_rows(n_rows, world, rank):
    """[start, stop) of this rank's contiguous row block (equal blocks; n_rows % world == 0)."""
    if n_rows % world:
        raise ValueError(f"global batch {n_rows} is not divisible by world size {world}")
    per = n_rows // world
    return rank * per, (rank + 1) * per


def all_gather_rows(t, group=None, always=False):
    """concatenate equal-sized per-rank row blocks in rank order (one all-gather).  `always`: run the collective even in a world of
    one rank (tests/test_gpu_dist.py uses it to put RCCL itself on a one-GPU box)."""
    world, _ = _world(group)
    if world == 1 and not (always and dist.is_available() and dist.is_initialized()):
        return t
    t = t.contiguous()
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path only (several ranks sharing one GPU cannot use RCCL): stage through the host
        host = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(host, t.cpu(), group=group)
        return host.to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


class EmbedAdapterStep:
    """encode local images -> all-gather embeddings + (y, g) -> replicated adapter step.

    encode_fn:   images[B_l,3,R,R] -> embeddings[B_l,D]   (model.encode_image)
    classifier:  CustomCLIP / MultipleAdapter (must offer .loss(features, labels, use_group))
    optimizer:   dbmm_amd.optim.SGD over the classifier's trainable parameters
    """

    def __init__(self, encode_fn, classifier, optimizer, n_groups=4, group=None, fused=True, micro_batches=1, always_collective=False):
        self.encode_fn, self.classifier, self.optimizer = encode_fn, classifier, optimizer
        self.n_groups, self.group, self.fused = n_groups, group, fused
        self.always = bool(always_collective)          # a world of ONE rank still issues its collectives (RCCL on a one-GPU box)
        self.micro_batches = int(micro_batches)
        self.counts = None
        self._side = None

    def gather(self, emb_local, y_local, g_local):
        labels = torch.stack([y_local, g_local], dim=1)             # int64 [B_l, 2]: one message
        emb = all_gather_rows(emb_local, self.group, self.always)
        labels = all_gather_rows(labels, self.group, self.always)
        return emb, labels[:, 0].contiguous(), labels[:, 1].contiguous()

    def encode_gather_overlapped(self, images_local, y_local, g_local):
        """encode in `micro_batches` row chunks; chunk j's all-gather is in flight while chunk j + 1 encodes.
        Returns (emb [B, D], y [B], g [B]) in the original global row order (rank-major, chunks in order within a rank)."""
        world, _ = _world(self.group)
        k, Bl = self.micro_batches, images_local.shape[0]
        if Bl % k:
            raise ValueError(f"local batch {Bl} is not divisible by micro_batches {k}")
        m = Bl // k
        chunk = lambda j: images_local[j * m:(j + 1) * m].contiguous()
        if world == 1 and not (self.always and dist.is_available() and dist.is_initialized()):
            return torch.cat([self.encode_fn(chunk(j)) for j in range(k)]), y_local, g_local
        labels = torch.stack([y_local, g_local], dim=1)             # int64 [B_l, 2]: one message
        rccl = images_local.is_cuda and dist.get_backend(self.group) != "gloo"
        works, G = [], None
        if rccl:
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=images_local.device)
            side = self._side
        for j in range(k):
            e = self.encode_fn(chunk(j)).contiguous()
            if G is None:
                G = torch.empty((k, world * m) + tuple(e.shape[1:]), dtype=e.dtype, device=e.device)
                if rccl:
                    G.record_stream(side)
            if rccl:
                side.wait_stream(main)                              # chunk j is complete when the collective starts
                with torch.cuda.stream(side):
                    works.append(dist.all_gather_into_tensor(G[j], e, group=self.group, async_op=True))
                e.record_stream(side)
            else:                                                   # gloo rehearsal (ranks sharing one GPU / CPU tests): same data flow
                G[j].copy_(all_gather_rows(e, self.group, self.always))
        labels = all_gather_rows(labels, self.group, self.always)
        for w in works:
            w.wait()                                                # stream-level: the compute stream waits, the host does not
        if rccl:
            main.wait_stream(side)
        emb = G.view((k, world, m) + tuple(G.shape[2:])).transpose(0, 1).reshape((world * Bl,) + tuple(G.shape[2:]))
        return emb, labels[:, 0].contiguous(), labels[:, 1].contiguous()

    def step(self, images_local, y_local, g_local, use_group=False):
        if self.micro_batches > 1:
            emb, y, g = self.encode_gather_overlapped(images_local, y_local, g_local)
        else:
            emb_local = self.encode_fn(images_local)
            emb, y, g = self.gather(emb_local, y_local, g_local)
        if emb.dtype != torch.float32:                 # fp16 mode: the adapter consumes .float() embeddings, like the
            emb = emb.float()                          # reference's readers do (data/celeba_embeddings_reg.py:74)
        labels = g if use_group else y
        if self.fused and emb.is_cuda and hasattr(self.classifier, "train_step"):
            # the whole step body as one C call (bit-identical to the autograd path below,
            # tests/test_gpu_trainer.py); needs the optimiser to hold the adapter in one group
            loss, logits, loss_rows = self.classifier.train_step(emb.detach(), labels, self.optimizer, use_group)
        else:
            loss, logits, loss_rows = self.classifier.loss(emb.detach(), labels, use_group)
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
        # update_dict (final_main.py:473) without a host sync: counters stay on the device
        if not use_group:
            if self.counts is None:
                self.counts = torch.zeros((self.n_groups, 2), dtype=torch.int64, device=logits.device)
            adapter.group_counts(logits, y, g, self.n_groups, self.counts)
        return loss, logits, emb
