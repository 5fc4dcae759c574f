"""One distillation iteration's loss, composed exactly as the reference's training loop composes it
(/root/reference/train.py:144-172), on this package's decoders (gags_amd/decoders.py) and losses (gags_amd/losses.py).

    loss, terms = distillation_loss(feature_map, seg_map, img_embed, cnn_decoder, cnn_scale_decoder, iteration)

train.py, line by line:
    :149   scale_map = cnn_scale_decoder(feature_map.detach())          (the scale decoder never back-propagates into
                                                                          the rasterizer; it learns through the losses)
    :152   seg_map_trained = get_trained_seg(seg_map, scale_map)
    :153   feature_reionvar_loss = scale_region_regulation_loss(feature_map, seg_map_trained, mix_seg=True)
    :156   scale_CE_loss = scale_regulation_loss(scale_map)
    :159   feature_map = cnn_decoder(feature_map)                        (dataset.speedup: 16 -> 512 channels)
    :161-163  iteration < scale_balance_iteration:   Ll1 = l1_loss(feature_map * mask, gt * mask)
    :164-167  else:  Ll1 = Scale_balance_loss(l1_loss_map(feature_map * mask, gt * mask), seg_map_trained, mask, mix_seg=True)
    :169-172  iteration < scale_regulation_iteration:  loss = 1.0 * Ll1 + 0.001 * CE
              else:                                     loss = 1.0 * Ll1 + 0.002 * CE + 0.1 * regionvar
Defaults of the two thresholds are train.py:303-304 (1 and 15001).  The region-variance term is only evaluated where it
enters the loss (the reference computes it on every iteration and drops it before 15001; the loss and every gradient are
the same).  Pinned against the reference's own functions chained on one input by tests/golden/make_golden_iteration.py ->
tests/test_iteration_gpu.py; timed end to end (with the rasterizer in front) by tools/decoder_bench.py.
"""
import torch

from . import losses as L


def distillation_loss(feature_map, seg_map, img_embed, cnn_decoder, cnn_scale_decoder, iteration,
                      scale_balance_iteration=1, scale_regulation_iteration=15001, fused_head=None, speedup=True):
    """feature_map [C,H,W]: render(...)["render"] (requires grad); seg_map [4,h,w]: the view's SAM segment ids per level
    (ids into img_embed, -1 = none); img_embed [n_seg, 512]: the view's CLIP embeddings.  Returns (loss, terms) with
    terms = {"l1", "ce", "regionvar" (None before scale_regulation_iteration), "scale_map", "seg_map_trained"}.
    fused_head: None = use CNN_decoder.distill_l1 (head fused into the loss) whenever the decoder offers it."""
    scale_map = cnn_scale_decoder(feature_map.detach())                                  # train.py:149
    seg_map_trained = L.get_trained_seg(seg_map, scale_map)                              # :152
    late = iteration >= scale_regulation_iteration
    regionvar = None
    if late:                                                                             # :153
        # (the map travels on through the loss's node: its gradient and the decoder's are then summed by ONE kernel)
        regionvar, feature_map = L.scale_region_regulation_loss_tee(feature_map, seg_map_trained)
    ce = L.scale_regulation_loss(scale_map)                                              # :156
    if iteration < scale_balance_iteration:                                              # :161-163  L_distill
        pred = cnn_decoder(feature_map) if speedup else feature_map
        gt, mask = L.read_sam_clip_feature(img_embed, seg_map, scale_map)
        l1 = L.l1_loss(pred * mask, gt * mask)
    else:                                                                                # :164-167  L_r-distill
        if speedup and fused_head is not False and hasattr(cnn_decoder, "distill_l1"):
            l1_map, mask = cnn_decoder.distill_l1(feature_map, img_embed, seg_map, scale_map)   # :159 + :165-166 in one call
        else:
            pred = cnn_decoder(feature_map) if speedup else feature_map
            l1_map, mask = L.distill_l1_map(pred, img_embed, seg_map, scale_map)
        l1 = L.Scale_balance_loss(l1_map, seg_map_trained, mask.squeeze(0), mix_seg=True)
    # :169-172  loss = 1.0 * Ll1 + 0.001 * CE  |  1.0 * Ll1 + 0.002 * CE + 0.1 * regionvar, as one launch per added term
    # (torch.add's alpha) instead of a multiply per term and an add per pair; the same values within an ulp
    if not late:
        loss = torch.add(l1, ce, alpha=0.001)
    else:
        loss = torch.add(torch.add(l1, ce, alpha=0.002), regionvar, alpha=0.1)
    return loss, {"l1": l1, "ce": ce, "regionvar": regionvar, "scale_map": scale_map, "seg_map_trained": seg_map_trained}
