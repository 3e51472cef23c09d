"""Losses of the hot path on the engine's kernels.

* fused_cross_entropy  -- nn.CrossEntropyLoss(ignore_index=-1, reduction='mean') on [N,200] logits
                          (/root/reference/lib/train_test/pl_BaselineTrainer.py:94-99,350): one kernel computes
                          the loss and the gradient.
* ContrastiveLanguageLoss -- the CLIP text-anchor hinge loss
                          (/root/reference/lib/losses/ContrastiveLanguageLoss.py:97-194, feat_dist :73-95) written
                          on the dense MFMA contraction S = normalize(F) . normalize(T)^T plus index gathers, which
                          is mathematically the reference's [N,1+K,C] gather + bmm.
"""
import torch
import torch.nn as nn

from .me.core import get_backend


class _FusedCE(torch.autograd.Function):
    """forward: loss only; backward: the same kernel again writes d(logits) already scaled by the upstream gradient
    (one read of the logits instead of a write + read-modify-write of an [N,200] gradient tensor)."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        loss, _ = get_backend().cross_entropy(logits, labels, ignore_index, want_grad=False)
        ctx.save_for_backward(logits, labels)
        ctx.ignore_index = ignore_index
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        _, dlogits = get_backend().cross_entropy(logits, labels, ctx.ignore_index, grad_scale=g)
        return dlogits, None, None


def fused_cross_entropy(logits, labels, ignore_index=-1):
    """mean softmax cross-entropy over the non-ignored rows; logits may be bf16 or fp32."""
    backend = get_backend()
    if hasattr(backend, "cross_entropy"):
        return _FusedCE.apply(logits, labels, ignore_index)
    return torch.nn.functional.cross_entropy(logits.float(), labels, ignore_index=ignore_index)


class _ClipSimilarity(torch.autograd.Function):
    """S[n,a] = <f_n/|f_n|, t_a/|t_a|>; backward only w.r.t. the features (anchors are frozen CLIP embeddings;
    a learned projection of the anchors gets its gradient through `anchor_grad=True`)."""

    @staticmethod
    def forward(ctx, feats, anchors, anchor_grad):
        sim, inv = get_backend().clip_similarity(feats, anchors)
        ctx.anchor_grad = anchor_grad
        ctx.save_for_backward(feats, anchors, sim, inv)
        return sim

    @staticmethod
    def backward(ctx, gs):
        feats, anchors, sim, inv = ctx.saved_tensors
        gs = gs.float()
        tn = torch.nn.functional.normalize(anchors.float(), dim=1)
        # dS/df = (t^ - s f^) / |f|
        fh = feats.float() * inv[:, None]
        gf = (gs @ tn - (gs * sim).sum(1, keepdim=True) * fh) * inv[:, None]
        ga = None
        if ctx.anchor_grad:
            an = anchors.float().norm(dim=1, keepdim=True).clamp_min(1e-12)
            gt = gs.t() @ fh                                  # d/d t^
            ga = (gt - (gt * tn).sum(1, keepdim=True) * tn) / an
            ga = ga.to(anchors.dtype)
        return gf.to(feats.dtype), ga, None


def clip_similarity(feats, anchors):
    return _ClipSimilarity.apply(feats, anchors, anchors.requires_grad)


class ContrastiveLanguageLoss(nn.Module):
    """cos variant of the reference loss: per voxel one positive anchor (its class) and K negatives drawn
    uniformly from the other classes (clip_uniform_sampling=True, ContrastiveLanguageLoss.py:138-144):
        d = 1 - mean_j <f^, t^_j>;   loss = relu(d_pos - pos_thresh) + neg_weight * relu(neg_thresh - d_neg)
    ignored voxels contribute 0 but count in the mean (feat_dist zeroes them, :94).
    Negatives are sampled on the device (no host loop / joblib pool / np.random), or passed explicitly."""

    def __init__(self, num_labels=200, num_negative_samples=3, pos_thresh=0.0, neg_thresh=0.6, neg_weight=1.0,
                 ignore_label=-1, reduction="mean"):
        super().__init__()
        self.num_labels, self.K = num_labels, num_negative_samples
        self.pos_thresh, self.neg_thresh, self.neg_weight = pos_thresh, neg_thresh, neg_weight
        self.ignore_label, self.reduction = ignore_label, reduction

    def sample_negatives(self, labels, generator=None):
        n = labels.shape[0]
        r = torch.randint(0, self.num_labels - 1, (n, self.K), device=labels.device, generator=generator)
        return r + (r >= labels.clamp_min(0)[:, None]).long()     # uniform over the other num_labels-1 classes

    def forward(self, features, labels, anchor_feats, neg_indices=None, return_similarity=False):
        if features.dim() != 2:
            raise ValueError("`features` needs to be [n_points, feat_dim]")
        if anchor_feats.dim() == 3:                               # anchors with attributes: use the plain ones (:122-123)
            anchor_feats = anchor_feats[:, 0, :]
        labels = labels.long()
        sim = clip_similarity(features, anchor_feats)             # [N, num_labels] -- the MFMA contraction
        valid = labels != self.ignore_label
        lab = labels.clamp_min(0)
        if neg_indices is None:
            neg_indices = self.sample_negatives(labels)
        d_pos = 1.0 - sim.gather(1, lab[:, None]).squeeze(1)
        d_neg = 1.0 - sim.gather(1, neg_indices).mean(1)
        zero = torch.zeros((), dtype=sim.dtype, device=sim.device)
        d_pos = torch.where(valid, d_pos, zero)
        d_neg = torch.where(valid, d_neg, zero)
        pos_loss = torch.relu(d_pos - self.pos_thresh)
        neg_loss = torch.relu(self.neg_thresh - d_neg)
        if self.reduction == "mean":
            loss = pos_loss.mean() + neg_loss.mean() * self.neg_weight
        else:
            loss = pos_loss + neg_loss * self.neg_weight
        if return_similarity:
            return loss, pos_loss, neg_loss, sim
        return loss, pos_loss, neg_loss


def feature_sim_argmax(sim):
    """lib/losses/utils.py:80-103 (cosine branch) + argmax for the metrics: reuse the similarity matrix."""
    return sim.argmax(1)
