"""Losses on the dense-head path (reference pcdet/utils/loss_utils.py): CenterNet focal + masked L1
(:264-385) and the anchor-head trio sigmoid-focal / weighted smooth-L1 / weighted CE (:9-206)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------ CenterHead losses
def neg_loss_cornernet(pred, gt, mask=None):
    """Penalty-reduced pixel-wise focal loss (alpha 2, beta 4).  pred, gt: [B, C, H, W]."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    pos_term = torch.log(pred) * torch.pow(1 - pred, 2) * pos
    neg_term = torch.log(1 - pred) * torch.pow(pred, 2) * torch.pow(1 - gt, 4) * neg
    if mask is not None:
        m = mask[:, None, :, :].float()
        pos_term, neg_term = pos_term * m, neg_term * m
        num_pos = (pos * m).sum()
    else:
        num_pos = pos.sum()
    pos_sum, neg_sum = pos_term.sum(), neg_term.sum()
    # branch-free form of `if num_pos == 0: -neg else: -(pos+neg)/num_pos` (pos_sum is 0 when
    # num_pos is 0), so no host sync is needed
    return -(pos_sum + neg_sum) / torch.clamp_min(num_pos, 1.0)


class FocalLossCenterNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.neg_loss = neg_loss_cornernet

    def forward(self, out, target, mask=None):
        return self.neg_loss(out, target, mask=mask)


def _gather_feat(feat, ind, mask=None):
    dim = feat.size(2)
    feat = feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim))
    if mask is not None:
        feat = feat[mask.unsqueeze(2).expand_as(feat)].view(-1, dim)
    return feat


def _transpose_and_gather_feat(feat, ind):
    """[B, C, H, W] sampled at flat positions ind [B, K] -> [B, K, C]"""
    b, c = feat.shape[:2]
    return _gather_feat(feat.permute(0, 2, 3, 1).reshape(b, -1, c), ind)


def _reg_loss(regr, gt_regr, mask):
    """Masked L1 per code dimension, normalised by the number of objects -> [dim]."""
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(gt_regr).float() * (~torch.isnan(gt_regr)).float()
    loss = torch.abs(regr * m - gt_regr * m)          # [B, K, dim]
    loss = loss.sum(dim=(0, 1))
    return loss / torch.clamp_min(num, min=1.0)


class RegLossCenterNet(nn.Module):
    def forward(self, output, mask, ind=None, target=None):
        pred = output if ind is None else _transpose_and_gather_feat(output, ind)
        return _reg_loss(pred, target, mask)


# ------------------------------------------------------------------ anchor-head losses
class SigmoidFocalClassificationLoss(nn.Module):
    def __init__(self, gamma=2.0, alpha=0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        return torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))

    def forward(self, input, target, weights):
        """input/target [B, A, C] (one-hot target), weights [B, A] -> [B, A, C]"""
        p = torch.sigmoid(input)
        alpha_w = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - p) + (1.0 - target) * p
        loss = alpha_w * torch.pow(pt, self.gamma) * self.sigmoid_cross_entropy_with_logits(input, target)
        if weights.dim() == 2 or (weights.dim() == 1 and target.dim() == 2):
            weights = weights.unsqueeze(-1)
        assert weights.dim() == loss.dim()
        return loss * weights


class WeightedSmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0 / 9.0, code_weights=None):
        super().__init__()
        self.beta = beta
        self.code_weights = None
        if code_weights is not None:
            self.register_buffer("_cw", torch.tensor(code_weights, dtype=torch.float32), persistent=False)
            self.code_weights = code_weights

    @staticmethod
    def smooth_l1_loss(diff, beta):
        if beta < 1e-5:
            return torch.abs(diff)
        n = torch.abs(diff)
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        target = torch.where(torch.isnan(target), input, target)  # ignore nan targets
        diff = input - target
        if self.code_weights is not None:
            if self._cw.device != diff.device:       # once: a per-call .to(device) of a host tensor is a synchronous copy
                self._cw = self._cw.to(diff.device)
            diff = diff * self._cw.view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


class WeightedCrossEntropyLoss(nn.Module):
    def forward(self, input, target, weights):
        """input [B, A, C] logits, target [B, A, C] one-hot, weights [B, A] -> [B, A]"""
        loss = F.cross_entropy(input.permute(0, 2, 1), target.argmax(dim=-1), reduction="none")
        return loss * weights
