"""Occupancy losses of the FB-OCC head, written without host synchronisation.

Reference: mmdet3d/models/fbbev/modules/occ_loss_utils/
    focal_loss.py:13-54,191-286   CustomFocalLoss (radial weight 1 + r/r_max on the 200x200 BEV plane, class weights,
                                  sigmoid focal loss summed over classes, mean over the visible voxels, x100)
    semkitti.py:56-66             inverse_sigmoid (scalar while-loops)
    semkitti.py:78-107            geo_scal_loss
    semkitti.py:111-164           sem_scal_loss (python loop over classes with `if torch.sum(...) > 0` branches)
    semkitti.py:167-182           CE_ssc_loss
    lovasz_softmax.py:20-33,168-229  lovasz_softmax (classes='present': python loop, one sort per class)
    nusc_param.py:16-35           class frequencies

The reference evaluates every data-dependent branch on the host (`if torch.sum(x) > 0`, boolean-mask indexing,
`.nonzero()`): ~100 device->host round trips per training step.  Here every branch is a `torch.where` on device
scalars and every boolean-mask gather is a multiplication by the mask, so the whole loss is one asynchronous stream of
kernels (and graph-capturable).  Values equal the reference's up to fp32 summation order (masked sums run over all
voxels instead of the compacted ones).
"""
import numpy as np
import torch
import torch.nn.functional as F

# nusc_param.py:16-35
nusc_class_frequencies = np.array([944004, 1897170, 152386, 2391677, 16957802, 724139, 189027, 2074468, 413451, 2384460,
                                   5916653, 175883646, 4275424, 51393615, 61411620, 105975596, 116424404, 1892500630])


def class_weights(out_channel, balance=True):
    """occupancy_head.py:118-127."""
    if not balance:
        return torch.ones(out_channel) / out_channel
    freq = nusc_class_frequencies.copy()
    if out_channel == 19:
        w = torch.from_numpy(1 / np.log(freq[:out_channel] + 0.001))
        return torch.cat([torch.tensor([0]), w])
    if out_channel == 17:
        freq[0] += freq[-1]
    return torch.from_numpy(1 / np.log(freq[:out_channel] + 0.001))


def inverse_sigmoid(x):
    """semkitti.py:56-66 elementwise: the reference nudges a SCALAR with `while x >= 1-1e-5: x -= 1e-5` /
    `while x < 1e-5: x += 1e-5`; ratios live in [0, 1], where each loop runs at most twice -- four masked steps cover
    it with room and are no-ops afterwards."""
    x = x.to(torch.float32)
    for _ in range(4):
        x = torch.where(x >= 1 - 1e-5, x - 1e-5, x)
    for _ in range(4):
        x = torch.where(x < 1e-5, x + 1e-5, x)
    return -torch.log((1 / x) - 1)


def _bce_to_one(ratio):
    return F.binary_cross_entropy_with_logits(inverse_sigmoid(ratio), torch.ones_like(ratio, dtype=torch.float32),
                                              reduction='none')


def geo_scal_loss(pred, ssc_target, ignore_index=255, non_empty_idx=0):
    """semkitti.py:78-107."""
    pred = F.softmax(pred.float(), dim=1)
    empty_probs = pred[:, non_empty_idx]
    nonempty_probs = 1 - empty_probs
    mask = (ssc_target != ignore_index).float()
    nonempty_target = (ssc_target != non_empty_idx).float() * mask
    empty_target = (1 - (ssc_target != non_empty_idx).float()) * mask
    eps = 1e-5
    intersection = (nonempty_target * nonempty_probs).sum()
    precision = intersection / ((nonempty_probs * mask).sum() + eps)
    recall = intersection / (nonempty_target.sum() + eps)
    spec = (empty_target * empty_probs).sum() / (empty_target.sum() + eps)
    return _bce_to_one(precision) + _bce_to_one(recall) + _bce_to_one(spec)


def sem_scal_loss(pred_, ssc_target, ignore_index=255):
    """semkitti.py:111-164, all classes at once.  Class i contributes only if it occurs among the valid targets; its
    precision term only if sum(p_i) > 0; its specificity term only if some valid voxel is not of class i."""
    pred = F.softmax(pred_.float(), dim=1)
    n_classes = pred.shape[1]
    begin = 1 if n_classes == 19 else 0
    cls = torch.arange(begin, n_classes - 1, device=pred.device)
    mask = (ssc_target != ignore_index)
    p = pred[:, begin:n_classes - 1].transpose(0, 1).reshape(len(cls), -1)              # (K, N)
    maskf = mask.reshape(1, -1).float()
    tgt = ((ssc_target.reshape(1, -1) == cls[:, None]) & mask.reshape(1, -1)).float()   # completion_target, (K, N)
    p = p * maskf
    sum_t = tgt.sum(1)
    sum_p = p.sum(1)
    nominator = (p * tgt).sum(1)
    n_valid = maskf.sum()
    sum_not_t = n_valid - sum_t
    # sum((1-p)(1-t)) over the valid voxels = n_valid - sum_p - sum_t + nominator
    specificity = (n_valid - sum_p - sum_t + nominator) / (sum_not_t + 1e-5)
    zero = torch.zeros_like(sum_t)
    loss_class = torch.where(sum_p > 0, _bce_to_one(nominator / (sum_p + 1e-5)), zero)
    loss_class = loss_class + _bce_to_one(nominator / (sum_t + 1e-5))
    loss_class = loss_class + torch.where(sum_not_t > 0, _bce_to_one(specificity), zero)
    present = sum_t > 0
    return torch.where(present, loss_class, zero).sum() / present.float().sum()


def CE_ssc_loss(pred, target, class_weights=None, ignore_index=255):
    """semkitti.py:167-182."""
    return F.cross_entropy(pred.float(), target.long(), weight=class_weights, ignore_index=ignore_index, reduction='mean')


def lovasz_softmax(probas, labels, ignore=None):
    """lovasz_softmax.py:168-229 with classes='present', per_image=False: for every class c present among the valid
    labels, sort |fg_c - p_c| descending and dot with the Lovasz gradient (:20-33); mean over the present classes.
    Ignored voxels enter with error 0 and fg 0: they sort behind every positive error, leave the prefix sums of the
    others unchanged and contribute 0 to the dot product, so no compaction is needed.  One batched sort over (C, N)."""
    B, C = probas.shape[:2]
    p = probas.float().reshape(B, C, -1).permute(1, 0, 2).reshape(C, -1)                # (C, N), N in (b, voxel) order
    lab = labels.reshape(1, -1)
    valid = (lab != ignore) if ignore is not None else torch.ones_like(lab, dtype=torch.bool)
    fg = ((lab == torch.arange(C, device=p.device)[:, None]) & valid).float()           # (C, N)
    errors = (fg - p).abs() * valid.float()
    errors_sorted, perm = torch.sort(errors, dim=1, descending=True)
    fg_sorted = torch.gather(fg, 1, perm)
    gts = fg_sorted.sum(1, keepdim=True)
    # ONE prefix sum instead of the reference's two (lovasz_softmax.py:27-28): cumsum(1 - fg) == (1..N) - cumsum(fg), and
    # both are sums of 0/1 below 2^24 -- exact integers in fp32, so the values are bit-identical -- evaluated blockwise:
    # ATen's innermost-dim scan runs a few million-element rows at ~0.4 GB/s-per-row (6.5 ms per call at 200x200x16, B=4)
    csum = _cumsum_rows(fg_sorted)
    intersection = gts - csum
    union = gts + (torch.arange(1, fg_sorted.shape[1] + 1, device=p.device, dtype=csum.dtype)[None] - csum)
    jaccard = 1.0 - intersection / union
    grad = torch.cat([jaccard[:, :1], jaccard[:, 1:] - jaccard[:, :-1]], 1)
    losses = (errors_sorted * grad).sum(1)
    present = gts[:, 0] > 0
    return torch.where(present, losses, torch.zeros_like(losses)).sum() / present.float().sum().clamp(min=1)


def _cumsum_rows(x, block=4096):
    """cumsum along dim 1 of a (rows, N) tensor with long rows as a two-level scan: prefix sums inside blocks of `block`
    elements (many short rows: the fast case of the scan kernel) + the exclusive prefix of the block totals.  Same additions
    in another association: exact (hence identical) for integer-valued data below 2^24, fp32 rounding otherwise."""
    R, N = x.shape
    if N <= 4 * block:
        return x.cumsum(1)
    nb = (N + block - 1) // block
    pad = nb * block - N
    xb = (torch.nn.functional.pad(x, (0, pad)) if pad else x).reshape(R, nb, block)
    inner = xb.cumsum(2)
    carry = inner[:, :, -1].cumsum(1) - inner[:, :, -1]                  # exclusive prefix of the block totals
    return (inner + carry[:, :, None]).reshape(R, nb * block)[:, :N]


class CustomFocalLoss(torch.nn.Module):
    """focal_loss.py:191-286 (use_sigmoid, not activated).  On a GPU the reference calls mmcv's sigmoid_focal_loss op
    (external) with integer targets; its value is the textbook form the reference's own CPU branch spells out
    (py_sigmoid_focal_loss :13-54), which is what is evaluated here on every device."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=100.0, activated=False,
                 bev_hw=(200, 200)):
        super().__init__()
        assert use_sigmoid and not activated
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight
        H, W = bev_hw                                                                   # :225-230 (hard-coded 200x200)
        xy, yx = torch.meshgrid(torch.arange(H) - H / 2, torch.arange(W) - W / 2, indexing='ij')
        c = torch.stack([xy, yx], 2).norm(2, -1)
        self.register_buffer('c', c / c.max() + 1, persistent=False)

    def forward(self, pred, target, weight=None, avg_factor=None, ignore_index=255, reduction_override=None):
        B, H, W, D = target.shape
        num_classes = pred.size(1)
        vis = (target != ignore_index)
        logits = pred.permute(0, 2, 3, 4, 1).float()                                    # (B,H,W,D,C)
        onehot = F.one_hot(torch.where(vis, target, torch.zeros_like(target)).long(), num_classes).to(logits.dtype)
        prob = logits.sigmoid()
        pt = (1 - prob) * onehot + prob * (1 - onehot)
        focal_weight = (self.alpha * onehot + (1 - self.alpha) * (1 - onehot)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(logits, onehot, reduction='none') * focal_weight
        loss = loss * weight.to(loss.dtype)                                             # class weights  (:252)
        loss = loss.sum(-1) * self.c.to(loss.device)[None, :, :, None]                  # radial weight (:250-252)
        visf = vis.to(loss.dtype)
        return self.loss_weight * (loss * visf).sum() / visf.sum()                      # .sum(-1).mean() over visible
