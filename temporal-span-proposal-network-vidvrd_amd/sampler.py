"""Fixed-fraction positive / negative pair sampler — host mirror of the reference's
`BalancedPositiveNegativePairSampler` (lib/modeling/relpn/sampler.py:3-66).

The reference's PPN constructs it (relpn/ppn.py:20-23) and never calls it; it is kept for drop-in
parity of the module attribute (`ppn.fg_bg_sampler`).  Same draws as the reference under the same
torch seed: one `randperm` over the positives, then one over the negatives, per segment."""
import torch

__all__ = ["BalancedPositiveNegativePairSampler"]


class BalancedPositiveNegativePairSampler:
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction

    def __call__(self, matched_idxs):
        """matched_idxs: list of integer tensors (-1 ignored, 0 negative, >= 1 positive), one per segment.
        Returns (pos_masks, neg_masks): two lists of uint8 masks of the selected elements."""
        pos_masks, neg_masks = [], []
        want_pos = int(self.batch_size_per_image * self.positive_fraction)
        for labels in matched_idxs:
            picks = []
            quota = None
            for members in (torch.nonzero(labels >= 1).squeeze(1), torch.nonzero(labels == 0).squeeze(1)):
                # positives first: at most the positive quota; negatives fill the rest of the batch
                quota = min(members.numel(), want_pos if quota is None else self.batch_size_per_image - quota)
                order = torch.randperm(members.numel(), device=members.device)[:quota]
                picks.append(members[order])
            for chosen, out in zip(picks, (pos_masks, neg_masks)):
                mask = torch.zeros_like(labels, dtype=torch.uint8)
                mask[chosen] = 1
                out.append(mask)
        return pos_masks, neg_masks
