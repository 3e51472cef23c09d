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
        loss, _, inv_valid = get_backend().cross_entropy(logits, labels, ignore_index, want_grad=False)
        ctx.save_for_backward(logits, labels, inv_valid)
        ctx.ignore_index = ignore_index
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels, inv_valid = ctx.saved_tensors
        _, dlogits, _ = get_backend().cross_entropy(logits, labels, ctx.ignore_index, grad_scale=g, inv_valid=inv_valid)
        return dlogits, None, None


class _FusedCERows(torch.autograd.Function):
    """reduction='none': forward = the per-row losses, backward = the same kernel with the upstream per-row gradient as its
    row factor (lgs_ce_forward_backward_rows) -- no [N, C] softmax is kept between the two."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        ctx.save_for_backward(logits, labels)
        ctx.ignore_index = ignore_index
        return get_backend().cross_entropy_rows(logits, labels, ignore_index)

    @staticmethod
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        return get_backend().cross_entropy_rows(logits, labels, ctx.ignore_index, row_grad=g), None, None


def fused_cross_entropy(logits, labels, ignore_index=-1, reduction="mean"):
    """softmax cross-entropy; logits may be bf16 or fp32, any class count the kernel's half-wave holds (512 fp32 / 1024 bf16);
    wider heads go through torch's device op.
    reduction='mean': over the non-ignored rows (pl_BaselineTrainer.py:350 with balanced_category_sampling off);
    reduction='none': per-row losses [N], 0 for ignored rows -- what `self.criterion` returns when the fine-tune script's
    --balanced_category_sampling True is on (scripts/train_models.sh:37, pl_BaselineTrainer.py:94), the input of
    sample_categories_for_balancing."""
    if reduction not in ("mean", "none"):
        raise ValueError("fused_cross_entropy: reduction must be 'mean' or 'none'")
    backend = get_backend()
    if hasattr(backend, "cross_entropy") and logits.shape[1] <= (1024 if logits.dtype == torch.bfloat16 else 512):
        if reduction == "none":
            return _FusedCERows.apply(logits, labels, ignore_index)
        return _FusedCE.apply(logits, labels, ignore_index)
    return torch.nn.functional.cross_entropy(logits.float(), labels, ignore_index=ignore_index, reduction=reduction)


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


class _ClipLossFused(torch.autograd.Function):
    """(d_pos, d_neg, pred[, sim]) of the text-anchor loss in ONE pass over the features (lgs_clip_loss_forward: MFMA
    contraction with the gathers, the arg-max and 1/|f| in its epilogue -- the [N, A] similarity matrix is never written
    unless asked for); backward = lgs_clip_loss_backward, a streaming kernel that uses the 4-sparse upstream gradient."""

    @staticmethod
    def forward(ctx, feats, anchors, labels, neg, ignore_label, want_sim):
        be = get_backend()
        d_pos, d_neg, pred, saved, sim = be.clip_loss_forward(feats, anchors, labels, neg, ignore_label, want_sim)
        ctx.ignore_label = ignore_label
        ctx.anchors = anchors if anchors.requires_grad else None      # a learned projection of the anchors (clip_models.py:192-200)
        ctx.save_for_backward(*saved, d_pos, d_neg)
        ctx.mark_non_differentiable(pred)
        if sim is None:
            sim = d_pos.new_empty(0)
        ctx.mark_non_differentiable(sim)
        return d_pos, d_neg, pred, sim

    @staticmethod
    def backward(ctx, g_dpos, g_dneg, _gp, _gs):
        *saved, d_pos, d_neg = ctx.saved_tensors
        be = get_backend()
        gf = be.clip_loss_backward(tuple(saved), d_pos, d_neg, g_dpos, g_dneg, ctx.ignore_label) if ctx.needs_input_grad[0] else None
        ga = None
        if ctx.anchors is not None and ctx.needs_input_grad[1]:
            # d/dT^ = G^T F^ on the weight-gradient kernels (lgs_clip_loss_backward_anchors), then through t^ = t / |t|
            gt = be.clip_loss_backward_anchors(tuple(saved), g_dpos, g_dneg, ctx.ignore_label)
            tn = saved[1]
            an = ctx.anchors.detach().float().norm(dim=1, keepdim=True).clamp_min(1e-12)
            ga = ((gt - (gt * tn).sum(1, keepdim=True) * tn) / an).to(ctx.anchors.dtype)
        return gf, ga, None, None, None, None


def feature_sim(output_feats, anchor_feats):
    """lib/losses/utils.py:80-103, cosine branch: S[n, a] = <f^_n, t^_a> (attribute anchors: the plain ones, :83-84)."""
    if anchor_feats.dim() == 3:
        anchor_feats = anchor_feats[:, 0, :]
    return clip_similarity(output_feats.detach(), anchor_feats.detach())


class ContrastiveLanguageLoss(nn.Module):
    """cos variant of the reference loss: per voxel one positive anchor (its class) and K negatives drawn
    uniformly from the other classes (clip_uniform_sampling=True, ContrastiveLanguageLoss.py:138-144):
        d = 1 - mean_j <f^, t^_j>;   loss = relu(d_pos - pos_thresh) + neg_weight * relu(neg_thresh - d_neg)
    ignored voxels contribute 0 but count in the mean (feat_dist zeroes them, :94).
    Negatives are sampled on the device (no host loop / joblib pool / np.random), or passed explicitly."""

    def __init__(self, num_labels=200, num_negative_samples=3, pos_thresh=0.0, neg_thresh=0.6, neg_weight=1.0,
                 ignore_label=-1, reduction="mean", uniform_sampling=True, distance_type="cos"):
        super().__init__()
        if distance_type not in ("cos", "l1", "l2"):
            raise ValueError("representation_distance_type must be 'cos', 'l1' or 'l2' (ContrastiveLanguageLoss.py:79-92), got %r" % (distance_type,))
        self.distance_type = distance_type            # config.representation_distance_type (config.py:159; default 'cos')
        self.num_labels, self.K = num_labels, num_negative_samples
        self.pos_thresh, self.neg_thresh, self.neg_weight = pos_thresh, neg_thresh, neg_weight
        self.ignore_label, self.reduction = ignore_label, reduction
        self.uniform_sampling = uniform_sampling      # config.clip_uniform_sampling (ContrastiveLanguageLoss.py:138-141)

    @classmethod
    def from_config(cls, config, num_labels, reduction="mean", feature_dim=512):
        """the reference's constructor arguments (ContrastiveLanguageLoss.py:22) -> ReferenceContrastiveLanguageLoss"""
        return ReferenceContrastiveLanguageLoss(config, num_labels, reduction=reduction, feature_dim=feature_dim)

    def sample_negatives(self, labels, generator=None):
        """K negative classes per voxel, on the device, no host sync.
        uniform_sampling=True : uniform over all other classes (clip_candidates minus own, :139)
        uniform_sampling=False: uniform over the OTHER classes present in this batch (unique_targets minus own, :141)"""
        n = labels.shape[0]
        lab = labels.clamp_min(0)
        if self.uniform_sampling:
            r = torch.randint(0, self.num_labels - 1, (n, self.K), device=labels.device, generator=generator)
            return r + (r >= lab[:, None]).long()
        L = self.num_labels
        valid = labels != self.ignore_label
        present = torch.zeros(L + 1, dtype=torch.bool, device=labels.device)
        present[torch.where(valid, lab, torch.full_like(lab, L))] = True
        present = present[:L]
        rank = torch.cumsum(present.long(), 0) - 1                          # rank of a present class among the present ones
        n_present = present.sum()                                           # device scalar: never read on the host
        cls_of_rank = torch.zeros(L + 1, dtype=torch.long, device=labels.device)
        cls_of_rank[torch.where(present, rank, torch.full_like(rank, L))] = torch.arange(L, device=labels.device)
        u = torch.rand((n, self.K), device=labels.device, generator=generator)
        r = torch.minimum((u * (n_present - 1).clamp_min(1)).long(), (n_present - 2).clamp_min(0))
        own = rank[lab][:, None]
        idx = torch.minimum(r + (r >= own).long(), (n_present - 1).clamp_min(0))   # a single-class batch has no negatives: own class
        return cls_of_rank[idx]

    def _raw_distances(self, features, anchors, labels, neg_indices, ignore_label):
        """representation_distance_type 'l1' / 'l2' (ContrastiveLanguageLoss.py:79-86) on the raw, un-normalised vectors, without
        the reference's [N, 1+K, C] gathers (9.8 GB at 1.2 M voxels x 512-d):
          l2: mean_j sqrt(|f - t_j|^2 + 1e-7) with |f - t|^2 = |f|^2 - 2 f.t + |t|^2 from ONE dense [N, C] x [C, A] product;
          l1: mean_j sum_c (f_c - t_jc) -- the reference sums SIGNED differences (no abs; reproduced as written) = sum(f) - sum(t_j).
        Not the measured path (config.py:159 defaults to 'cos', which owns the fused kernels): plain torch device ops."""
        f, t = features.float(), anchors.float()
        # a label outside [0, A) is an ignored row (what the fused cos kernel and the CE kernel do), never an index: a
        # non-negative ignore_label such as 255 with 200 anchors must not reach the gathers
        valid = (labels != ignore_label) & (labels >= 0) & (labels < t.shape[0])
        lab = torch.where(valid, labels, torch.zeros_like(labels))
        if self.distance_type == "l1":
            fs, ts = f.sum(1), t.sum(1)
            d_pos = fs - ts[lab]
            d_neg = (fs[:, None] - ts[neg_indices]).mean(1)
        else:
            d2 = ((f * f).sum(1)[:, None] - 2.0 * (f @ t.t()) + (t * t).sum(1)[None, :]).clamp_min(0)
            d_pos = torch.sqrt(d2.gather(1, lab[:, None]).squeeze(1) + 1e-7)
            d_neg = torch.sqrt(d2.gather(1, neg_indices) + 1e-7).mean(1)
        zero = torch.zeros((), dtype=d_pos.dtype, device=d_pos.device)
        return torch.where(valid, d_pos, zero), torch.where(valid, d_neg, zero)

    def forward(self, features, labels, anchor_feats, neg_indices=None, return_similarity=False, return_pred=False, ignore_label=None):
        """-> (loss, pos_loss, neg_loss[, sim][, pred]); pred = argmax_a <f^, t^_a>, what the reference's trainer gets from
        feature_sim(...).argmax(1) (pl_RepresentationTrainer.py:237-238) -- here a by-product of the same pass.
        ignore_label: overrides self.ignore_label for this call (the [N, 2]-label adapter passes a sentinel that cannot collide
        with a flattened (category, attribute) index)."""
        if features.dim() != 2:
            raise ValueError("`features` needs to be [n_points, feat_dim]")
        if anchor_feats.dim() == 3:                               # anchors with attributes: use the plain ones (:122-123)
            anchor_feats = anchor_feats[:, 0, :]
        labels = labels.long()
        ign = self.ignore_label if ignore_label is None else ignore_label
        if neg_indices is None:
            neg_indices = self.sample_negatives(labels)
        if self.distance_type != "cos":
            if return_similarity or return_pred:
                raise ValueError("similarity / arg-max outputs exist for the 'cos' distance only")
            d_pos, d_neg = self._raw_distances(features, anchor_feats, labels, neg_indices, ign)
            return self._hinge(d_pos, d_neg)
        be = get_backend()
        fused = (hasattr(be, "clip_loss_forward") and features.is_cuda
                 and (not anchor_feats.requires_grad or hasattr(be, "clip_loss_backward_anchors"))
                 and anchor_feats.shape[0] % 4 == 0 and 4 <= anchor_feats.shape[0] <= be.CLIP_LOSS_MAX_ANCHORS
                 and 1 <= neg_indices.shape[1] <= 7)
        if fused:
            d_pos, d_neg, pred, sim = _ClipLossFused.apply(features, anchor_feats, labels, neg_indices, ign, bool(return_similarity))
        else:
            # > 224 anchors, more than 7 negatives, or the CPU oracle backend of the tests: dense similarity matrix + index
            # gathers (learned anchor projections take the fused path too: lgs_clip_loss_backward_anchors)
            sim = clip_similarity(features, anchor_feats)         # [N, num_labels] -- the MFMA contraction
            valid = (labels != ign) & (labels >= 0) & (labels < sim.shape[1])
            lab = torch.where(valid, labels, torch.zeros_like(labels))
            d_pos = 1.0 - sim.gather(1, lab[:, None]).squeeze(1)
            d_neg = 1.0 - sim.gather(1, neg_indices).mean(1)
            zero = torch.zeros((), dtype=sim.dtype, device=sim.device)
            d_pos = torch.where(valid, d_pos, zero)
            d_neg = torch.where(valid, d_neg, zero)
            pred = sim.detach().argmax(1) if return_pred else None
        out = self._hinge(d_pos, d_neg)
        if return_similarity:
            out = out + (sim,)
        if return_pred:
            out = out + (pred,)
        return out


    def _hinge(self, d_pos, d_neg):
        """ContrastiveLanguageLoss.py:185-192"""
        pos_loss = torch.relu(d_pos - self.pos_thresh)
        neg_loss = torch.relu(self.neg_thresh - d_neg)
        if self.reduction == "mean":
            loss = pos_loss.mean() + neg_loss.mean() * self.neg_weight
        else:
            loss = pos_loss + neg_loss * self.neg_weight
        return (loss, pos_loss, neg_loss)


class ReferenceContrastiveLanguageLoss(ContrastiveLanguageLoss):
    """Drop-in for the reference class, same constructor and call signature:
        /root/reference/lib/losses/ContrastiveLanguageLoss.py:22   __init__(config, num_labels, temperature, base_temperature,
                                                                           reduction, feature_dim)
        /root/reference/lib/losses/ContrastiveLanguageLoss.py:97   forward(features, labels, anchor_feats, preds=None)
                                                                   -> (loss, pos_loss, neg_loss)
    as built by pl_RepresentationTrainer.py:45 (`ContrastiveLanguageLoss(self.config, num_labels=..., reduction=...)`) and
    called at :216 (`criterion(soutput.F, target, anchor_feats=anchor_feats)`).  The one-line swap is the import at
    pl_RepresentationTrainer.py:8.  It reads the same config fields (ignore_label,
    num_negative_samples with -1 = all labels, contrast_pos_thresh / contrast_neg_thresh / contrast_neg_weight,
    clip_uniform_sampling, representation_distance_type) and keeps the attributes the trainer touches
    (`augment_categories`, :46; `confusion_hist` buffer).  The arithmetic is the engine's fused kernel
    (lgs_clip_loss_forward / lgs_clip_loss_backward); negatives are drawn on the device instead of the reference's
    per-class np.random.choice inside a joblib thread pool (whose draw order is not reproducible even with a seed).
    Category + attribute labels ([N, 2], :149-178) select the anchor (category, attribute) as the positive and plain
    category anchors as negatives, like the reference.  With clip_uniform_sampling=False the reference draws negatives from
    `unique_targets_np[unique_targets_np != ut[0]]`, where unique_targets_np is an array of (category, attribute) PAIRS (:152-153,:173):
    the comparison broadcasts over both columns and the "candidates" are a mix of category and attribute ids -- a reference bug;
    this class draws from the other CATEGORIES present in the batch, which is what the 1-D branch (:141) does. its latent augmentation (a pretrained AttributeFittingModel that is
    not part of the hot path) is not reproduced and raises if configured."""

    def __init__(self, config, num_labels, temperature=0.07, base_temperature=0.07, reduction="mean", feature_dim=512):
        k = int(getattr(config, "num_negative_samples", 3))
        if k <= -1:
            k = num_labels                                            # :35-38
        dist_type = getattr(config, "representation_distance_type", "cos")
        super().__init__(num_labels=num_labels, num_negative_samples=k,
                         pos_thresh=float(getattr(config, "contrast_pos_thresh", 0.0)),
                         neg_thresh=float(getattr(config, "contrast_neg_thresh", 0.6)),
                         neg_weight=float(getattr(config, "contrast_neg_weight", 1.0)),
                         ignore_label=int(getattr(config, "ignore_label", -1)), reduction=reduction,
                         uniform_sampling=bool(getattr(config, "clip_uniform_sampling", True)), distance_type=dist_type)
        self.config = config
        self.temperature, self.base_temperature, self.feature_dim = temperature, base_temperature, feature_dim
        self.num_negative_samples = k
        self.register_buffer("confusion_hist", torch.zeros((num_labels, num_labels)).long())
        self.augment_categories = torch.empty(0)
        # latent attribute augmentation (:46-60): eight learned linear maps ("A red ", ..., "A small "), weights from
        # config.scannet_path / config.projection_model_path when that file exists (else the module's initial weights, as in
        # the reference), eval mode
        import os
        import numpy as np
        from .projection_models import AttributeFittingModel
        self.attributes = np.array(['A red ', 'A green ', 'A blue ', 'A yellow ', 'A dark ', 'A bright ', 'A big ', 'A small '])
        self.augment_probability = float(getattr(config, "instance_augmentation_color_aug_prob", 0.0))
        self.projection_model = AttributeFittingModel(feature_dim, feature_dim, self.attributes.shape[0])
        model_path = "%s/%s" % (getattr(config, "scannet_path", ""), getattr(config, "projection_model_path", ""))
        if os.path.isfile(model_path):
            self.projection_model.load_state_dict(torch.load(model_path))
        self.projection_model.eval()

    def plan_latent_augmentation(self, generator=None, device="cpu"):
        """The reference decides per unique (category, attribute) target of the batch, inside a thread pool, with the host RNGs
        (`random.random() < p`, then `np.random.randint(0, 8)`, :62-70).  Here ONE independent draw per possible (category,
        attribute-slot) pair, on the device, no host loop: -> (augment? [L * A] bool, new attribute [L * A] int64) for A slots."""
        A = self.attributes.shape[0] + 1                                   # slot 0 = the raw category, 1..8 = attributes
        # draw where the generator lives (a CPU generator cannot feed a device op), then move: 2 x 1800 numbers
        gdev = generator.device if generator is not None else device
        u = torch.rand(self.num_labels * A, generator=generator, device=gdev).to(device)
        k = torch.randint(0, self.attributes.shape[0], (self.num_labels * A,), generator=generator, device=gdev).to(device)
        return u < self.augment_probability, k

    def latent_augmentation(self, features, labels, plan=None, generator=None):
        """ContrastiveLanguageLoss.py:62-70,160-165 for the whole batch at once: rows whose category is in `augment_categories`
        and whose (category, attribute) target drew "augment" are replaced IN PLACE (like the reference's
        `features[ut_inds, :] = aug_feats`) by attribute a's projection of themselves; their attribute label becomes a and the
        positive anchor slot a + 1 (:68, `current_aug + 1`).  -> (features, labels, positive attribute slot [N])"""
        cat, att = labels[:, 0].long(), labels[:, 1].long()
        A = self.attributes.shape[0] + 1
        dev = features.device
        if plan is None:
            plan = self.plan_latent_augmentation(generator, dev)
        on, new_attr = plan[0].to(dev), plan[1].to(dev)
        cats = self.augment_categories.to(dev).long()
        in_aug = torch.zeros(self.num_labels + 1, dtype=torch.bool, device=dev)
        if cats.numel():
            in_aug[cats.clamp(0, self.num_labels)] = True
        valid = (cat != self.ignore_label) & (cat >= 0) & (cat < self.num_labels) & (att >= 0) & (att < A)
        pair = torch.where(valid, cat * A + att, torch.zeros_like(cat))
        in_aug_row = valid & in_aug[torch.where(valid, cat, torch.full_like(cat, self.num_labels))]
        do = in_aug_row & on[pair]
        k = new_attr[pair]
        idx = do.nonzero().squeeze(1)                                      # (one host sync; the reference loops over classes)
        if self.projection_model.attr_linears[0].weight.device != dev:
            self.projection_model = self.projection_model.to(dev)
        if idx.numel():
            proj = self.projection_model.project(features[idx].float(), k[idx])
            features.index_copy_(0, idx, proj.to(features.dtype))
        labels[:, 1] = torch.where(do, k, att).to(labels.dtype)
        # a target of an augment category whose draw said "no" keeps its features and labels, but the reference then sets ut[1] =
        # attribute_id = 0 (:70,:163-165): its positive anchor is slot 0, the raw category, whatever attribute it carried
        return features, labels, torch.where(do, k + 1, torch.where(in_aug_row, torch.zeros_like(att), att))

    def forward(self, features, labels, anchor_feats, preds=None, neg_indices=None, aug_plan=None):
        if labels.dim() == 2:                                        # (category, attribute) targets, :149-181
            if anchor_feats.dim() != 3:
                raise ValueError("[N, 2] labels need [num_labels, num_attributes, C] anchors")
            A = anchor_feats.shape[1]
            cat, att = labels[:, 0].long(), labels[:, 1].long()
            if getattr(self.config, "instance_augmentation", None) == "latent":
                features, labels, att = self.latent_augmentation(features, labels, plan=aug_plan)
            if neg_indices is None:
                neg_indices = self.sample_negatives(cat)
            valid = cat != self.ignore_label
            # ignored rows become -1, which no flattened (category, attribute) index can equal -- a non-negative
            # config.ignore_label (e.g. 255) would otherwise silently drop the valid pair whose flat index is 255
            flat_lab = torch.where(valid, cat.clamp_min(0) * A + att, torch.full_like(cat, -1))
            out = super().forward(features, flat_lab, anchor_feats.reshape(-1, anchor_feats.shape[-1]), neg_indices=neg_indices * A,
                                  ignore_label=-1)
            return out[:3]
        return super().forward(features, labels, anchor_feats, neg_indices=neg_indices)[:3]


class _ClipCE(torch.autograd.Function):
    """cross-entropy over the cosine similarities to ALL anchors, one autograd node: forward = lgs_clip_similarity (the MFMA
    contraction normalize(F) . normalize(T)^T, [N, A] fp32) + lgs_ce_forward_backward (loss only); backward = the CE kernel again
    (d S, already scaled by the upstream gradient) pushed through dS/df = (t^ - s f^) / |f| (and dS/dT for learned anchors)."""

    @staticmethod
    def forward(ctx, feats, anchors, labels, ignore_index):
        be = get_backend()
        sim, inv = be.clip_similarity(feats, anchors)
        loss, _, inv_valid = be.cross_entropy(sim, labels, ignore_index, want_grad=False)
        ctx.save_for_backward(feats, anchors, sim, inv, labels, inv_valid)
        ctx.ignore_index = ignore_index
        return loss

    @staticmethod
    def backward(ctx, g):
        feats, anchors, sim, inv, labels, inv_valid = ctx.saved_tensors
        _, gs, _ = get_backend().cross_entropy(sim, labels, ctx.ignore_index, grad_scale=g, inv_valid=inv_valid)
        tn = torch.nn.functional.normalize(anchors.float(), dim=1)
        fh = feats.float() * inv[:, None]
        gf = ((gs @ tn - (gs * sim).sum(1, keepdim=True) * fh) * inv[:, None]).to(feats.dtype) if ctx.needs_input_grad[0] else None
        ga = None
        if ctx.needs_input_grad[1]:
            an = anchors.float().norm(dim=1, keepdim=True).clamp_min(1e-12)
            gt = gs.t() @ fh
            ga = ((gt - (gt * tn).sum(1, keepdim=True) * tn) / an).to(anchors.dtype)
        return gf, ga, None, None


def _ce_like_the_engine(scores, labels, ignore_index, reduction):
    """torch cross-entropy with the conventions of the engine's fused kernel, so that the result does not depend on which path
    the gate selects (advisor, round 5): a label outside [0, A) is an ignored row (nn.CrossEntropyLoss raises on it), and the mean
    over a batch with no counted row is 0 (nn.CrossEntropyLoss returns NaN)."""
    a = scores.shape[1]
    lab = torch.where((labels >= 0) & (labels < a) & (labels != ignore_index), labels, torch.full_like(labels, -100))
    if reduction != "mean":
        return torch.nn.functional.cross_entropy(scores, lab, ignore_index=-100, reduction=reduction)
    total = torch.nn.functional.cross_entropy(scores, lab, ignore_index=-100, reduction="sum")
    return total / (lab != -100).sum().clamp_min(1).to(total.dtype)


class ReferenceContrastiveLanguageCELoss(ReferenceContrastiveLanguageLoss):
    """Drop-in for /root/reference/lib/losses/ContrastiveLanguageLoss.py:196-237 (`embedding_loss_type=contrast_ce`,
    lib/train_test/pl_RepresentationTrainer.py:42-43): nn.CrossEntropyLoss(ignore_index, reduction) over the per-voxel
    "distances" to ALL num_labels anchors -- for 'cos' literally normalize(F) . normalize(T)^T in [N, num_labels] (no temperature:
    the reference never applies it), the one dense contraction of the hot path; for 'l2' sqrt(|f - t^|^2 + 1e-7) against the
    NORMALISED anchors (as written, :208-211,:230).  forward(features, labels, anchor_feats, preds=None) -> (loss, zeros(1), loss).
    Two deliberate conventions, the same on every path (fused kernel, dense fallback, l2): a label outside [0, num_labels) is an
    ignored row (the reference's nn.CrossEntropyLoss raises), and a batch without a counted row gives 0 (the reference: NaN)."""

    def __init__(self, config, num_labels, temperature=0.07, base_temperature=0.07, reduction="mean"):
        super().__init__(config, num_labels, temperature, base_temperature, reduction)

    def forward(self, features, labels, anchor_feats, preds=None):
        if features.dim() != 2:
            raise ValueError("`features` needs to be [n_points, feat_dim]")
        labels = labels.long()
        be = get_backend()
        if self.distance_type == "cos":
            if (self.reduction == "mean" and hasattr(be, "cross_entropy") and features.is_cuda
                    and anchor_feats.shape[0] <= 512 and features.dtype in (torch.float32, torch.bfloat16)):
                loss = _ClipCE.apply(features, anchor_feats, labels, self.ignore_label)
            else:
                out = clip_similarity(features, anchor_feats)
                loss = _ce_like_the_engine(out, labels, self.ignore_label, self.reduction)
        elif self.distance_type == "l2":
            f = features.float()
            t = torch.nn.functional.normalize(anchor_feats.float(), p=2, dim=1)
            d2 = ((f * f).sum(1)[:, None] - 2.0 * (f @ t.t()) + (t * t).sum(1)[None, :]).clamp_min(0)
            loss = _ce_like_the_engine(torch.sqrt(d2 + 1e-7), labels, self.ignore_label, self.reduction)
        else:
            raise ValueError("ContrastiveLanguageCELoss supports representation_distance_type 'cos' and 'l2' (:206-220)")
        return loss, torch.zeros(1), loss


def sample_categories_for_balancing(loss, targets, frequency_organized_cats, head_ratio, common_ratio, ignore_label=-1,
                                    generator=None, split="tensors"):
    """lib/losses/utils.py:13-77 on the device, no host loop / np.random.choice: per-point `loss` [N] is masked so
    that every HEAD class keeps round(head_ratio * count) of its points, every COMMON class round(common_ratio * count)
    (drawn without replacement), TAIL classes keep all; ratio <= 0 keeps everything (:41,:54).
    frequency_organized_cats: bool [num_labels, 3] (head, common, tail), lib/datasets/scannet.py:131-141.
    split="tensors" (the reference's return shape):
        -> (masked loss mean over ALL points, (head, common, tail per-point losses, detached), loss_items [N_valid, 3]);
        the three variable-length tensors and the row-filtered mask are boolean-index results, i.e. THREE + ONE device->host
        syncs for their sizes, exactly what the reference's own `loss[loss_items[:, 0]]` costs.
    split="stats" (the training step: no host sync at all):
        -> (masked loss mean, stats [3, 2] = (sum of the per-point losses, number of points) of head / common / tail -- what the
        trainer feeds its meters, `nanmean_t(split_losses[i])` = stats[i, 0] / stats[i, 1] and `.size(0)` = stats[i, 1]
        (pl_BaselineTrainer.py:353-355) --, loss_items [N, 3] over ALL rows, False on ignored ones)."""
    if split not in ("tensors", "stats"):
        raise ValueError("split must be 'tensors' or 'stats'")
    dev = loss.device
    foc = frequency_organized_cats.to(dev).bool()
    valid = targets != ignore_label
    lab = targets.clamp_min(0).long()
    L = foc.shape[0]
    group = torch.where(foc[:, 0], 0, torch.where(foc[:, 1], 1, 2)).to(dev)     # anything not head/common is kept like tail (:58-62)
    # per-class keep ratio, built from host scalars with fills only (a torch.tensor([...]) would be a blocking host->device copy)
    ratio = torch.ones(L, dtype=torch.float64, device=dev)
    ratio.masked_fill_(group == 0, head_ratio if head_ratio > 0 else 1.0).masked_fill_(group == 1, common_ratio if common_ratio > 0 else 1.0)
    if head_ratio <= 0 and common_ratio <= 0:
        # the configured default (config.py:281-282: both ratios -1; scripts/train_models.sh sets neither): every class keeps all
        # of its points (:45,:56,:60), nothing is drawn
        point_mask = valid
    else:
        # rank of every point inside its class by a random key = a uniform draw without replacement.  Class sizes and first
        # positions come from the sorted keys (binary searches for the 200 class boundaries), not from 1.2 M atomics on 200 counters
        u = torch.rand(loss.shape[0], device=dev, generator=generator)
        key = lab.double() + u.double()
        key = torch.where(valid, key, torch.full_like(key, float(L + 1)))
        skey, order = torch.sort(key)
        bounds = torch.searchsorted(skey, torch.arange(L + 1, device=dev, dtype=torch.float64))
        start, counts = bounds[:L], bounds[1:] - bounds[:L]                       # first sorted position / size of every class
        keep_n = torch.round(ratio * counts.double()).long()                     # python round == torch.round: half to even
        pos = torch.empty_like(order)
        pos[order] = torch.arange(order.shape[0], device=dev)
        rank_in_class = pos - start[lab]
        point_mask = valid & (rank_in_class < keep_n[lab])
    masked = loss * point_mask.to(loss.dtype)
    be = get_backend()
    if split == "stats" and loss.is_cuda and hasattr(be, "split_stats"):
        # one streaming pass for the three meters (lgs_split_stats); the [N, 3] membership mask is built only if somebody reads it
        stats = be.split_stats(loss, targets, group, ignore_label)
        return masked.mean(), stats, _LazyItems(valid, group, lab)
    pg = group[lab]
    loss_items = torch.stack([valid & (pg == 0), valid & (pg == 1), valid & (pg == 2)], 1)
    if split == "stats":
        ld = loss.detach().float()
        stats = torch.stack([torch.stack([(ld * loss_items[:, i]).sum() for i in range(3)]), loss_items.sum(0).float()], 1)   # [3, 2]
        return masked.mean(), stats, loss_items
    head, common, tail = (loss[loss_items[:, i]].detach() for i in range(3))
    return masked.mean(), (head, common, tail), loss_items[valid]


class _LazyItems:
    """loss_items [N, 3] of split='stats' on the device path, built when it is indexed / converted (the training step never does:
    its meters take the statistics; the reference's evaluation code indexes predictions with it, pl_BaselineTrainer.py:358-370)"""

    def __init__(self, valid, group, lab):
        self._parts, self._t = (valid, group, lab), None

    def tensor(self):
        if self._t is None:
            valid, group, lab = self._parts
            pg = group[lab]
            self._t = torch.stack([valid & (pg == 0), valid & (pg == 1), valid & (pg == 2)], 1)
            self._parts = None
        return self._t

    def __getitem__(self, idx):
        return self.tensor()[idx]

    def __getattr__(self, name):
        return getattr(self.tensor(), name)


def instance_offset_losses(pt_offsets, coords_xyz, centers, instance_ids, voxel_size):
    """downstream/insseg/lib/pl_Trainer.py:271-299 (PointGroup offset losses): L1 norm loss and direction loss of the
    predicted per-voxel offset to its instance centre, averaged over voxels with an instance (id != -1).
    pt_offsets [N,3] (any float dtype), coords_xyz [N,3] voxel coordinates, centers [N,3] in voxel units.
    -> (offset_norm_loss, offset_dir_loss)"""
    po = pt_offsets.float()
    gt = (centers.float() - coords_xyz.float()) * voxel_size
    valid = (instance_ids != -1).float()
    denom = valid.sum() + 1e-6
    norm_loss = ((po - gt).abs().sum(-1) * valid).sum() / denom
    gt_n = gt / (gt.norm(p=2, dim=1, keepdim=True) + 1e-8)
    po_n = po / (po.norm(p=2, dim=1, keepdim=True) + 1e-8)
    dir_loss = (-(gt_n * po_n).sum(-1) * valid).sum() / denom
    return norm_loss, dir_loss


def feature_sim_argmax(sim):
    """lib/losses/utils.py:80-103 (cosine branch) + argmax for the metrics: reuse the similarity matrix."""
    return sim.argmax(1)
