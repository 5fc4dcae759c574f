"""Query-time relevancy: the part of eval/openclip_encoder.py that runs per pixel.

`RelevancyHead` holds the unit text embeddings the reference's OpenCLIPNetwork computes with the CLIP text encoder
(eval/openclip_encoder.py:31-39,66-74; the encoder itself needs the CLIP weights and is out of scope) and mirrors
    get_relevancy(embed [P,512], positive_id) -> [P,2]          (:42-56)
    get_max_across(sem_map [L,h,w,512])       -> [L,n_phrases,h,w]   (:96-111)
on one HIP kernel that reads every pixel embedding once for all phrases (the reference re-reads the map once per
phrase and level through torch.mm + stack + softmax + gather).

`activate_maps` / `localize` are the rest of the per-view query path, evaluate_iou_loc.py:100-146 (`activate_stream`:
30x30 box mean -- cv2.filter2D on the host in the reference, one device->host->device round trip per phrase -- blend,
min-max normalise, clip, threshold, eval/utils.py:55-64 majority filter, IoU) and :163-176 (`lerf_localization`: box
mean, arg-max, hit test against the annotated boxes), for all phrases in one call on the GPU.  Saving heat maps /
composited images to disk (colormaps, mediapy) is out of scope."""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr


class RelevancyHead:
    def __init__(self, pos_embeds, neg_embeds):
        self.pos_embeds = pos_embeds.float().contiguous()
        self.neg_embeds = neg_embeds.float().contiguous()

    def set_positives(self, pos_embeds):
        self.pos_embeds = pos_embeds.float().contiguous()

    def _all(self, embed):
        if not embed.is_cuda:
            raise RuntimeError("gags_amd.relevancy: tensors must live on the GPU (there is no CPU path)")
        e = embed.float().contiguous()
        n_pix, c = e.shape
        out = torch.empty(self.pos_embeds.shape[0], n_pix, 2, device=e.device)
        check(_lib.load().gags_relevancy(n_pix, c, self.pos_embeds.shape[0], self.neg_embeds.shape[0], ptr(e),
                                         ptr(self.pos_embeds), ptr(self.neg_embeds), ptr(out),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gags_relevancy")
        return out

    @torch.no_grad()
    def get_relevancy(self, embed, positive_id):
        return self._all(embed)[positive_id]

    @torch.no_grad()
    def get_max_across(self, sem_map):
        n_levels, h, w, c = sem_map.shape
        probs = self._all(sem_map.reshape(-1, c))  # [n_phrases, L*h*w, 2]
        return probs[..., 0].reshape(-1, n_levels, h, w).permute(1, 0, 2, 3).contiguous()


@torch.no_grad()
def activate_maps(valid_map, thresh=0.5, box=30, smooth_scale=3):
    """evaluate_iou_loc.py:100-146 for every phrase of `valid_map` [n_phrases, h, w] (= get_max_across(sem_map).squeeze(0)).
    Returns a dict of device tensors: avg (box mean), heatmap (0.5 (avg + map), what the reference writes back into
    valid_map[k] and saves), output (normalised to [-1, 1], clipped to [0, 1]), mask_pred (output > thresh, uint8),
    mask (after the majority filter: the reference's final `mask_pred`), stats [n_phrases, 3] = (min, max of the heat
    map, max of avg)."""
    if not valid_map.is_cuda:
        raise RuntimeError("gags_amd.relevancy: tensors must live on the GPU (there is no CPU path)")
    v = valid_map.float().contiguous()
    k, h, w = v.shape
    lib = _lib.load()
    dev = v.device
    avg, heat, outp = (torch.empty_like(v) for _ in range(3))
    mask_pred = torch.empty(k, h, w, dtype=torch.uint8, device=dev)
    mask = torch.empty_like(mask_pred)
    stats = torch.empty(k, 3, device=dev)
    nb = lib.gags_relevancy_activate_scratch_bytes(k, h, w)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    check(lib.gags_relevancy_activate(k, h, w, ptr(v), float(thresh), int(box), int(smooth_scale), ptr(avg), ptr(heat),
                                      ptr(outp), ptr(mask_pred), ptr(mask), ptr(stats), ptr(scratch), nb,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gags_relevancy_activate")
    return {"avg": avg, "heatmap": heat, "output": outp, "mask_pred": mask_pred, "mask": mask, "stats": stats}


@torch.no_grad()
def mask_iou(mask, mask_gt):
    """evaluate_iou_loc.py:152-154: |pred & gt| / |pred | gt| per phrase; masks [n_phrases, h, w] (any integer / bool dtype)."""
    a, b = mask.bool(), mask_gt.bool()
    return (a & b).flatten(1).sum(1).double() / (a | b).flatten(1).sum(1).double()


@torch.no_grad()
def localize(valid_map, boxes=None, box=30):
    """evaluate_iou_loc.py:163-191 `lerf_localization` per phrase: score = max of the box mean, coords = every (x, y)
    that attains it; with `boxes` (a list, per phrase, of [n, 4] arrays x1, y1, x2, y2) also the hit flags the reference
    sums into acc_num.  Returns (scores [n_phrases], list of [m, 2] (x, y) tensors, hits or None)."""
    r = activate_maps(valid_map, box=box)
    avg, scores = r["avg"], r["stats"][:, 2]
    coords, hits = [], None if boxes is None else []
    for k in range(avg.shape[0]):
        yx = torch.nonzero(avg[k] == scores[k])
        xy = yx.flip(1)
        coords.append(xy)
        if boxes is not None:
            hit = False
            for bx in torch.as_tensor(boxes[k]).reshape(-1, 4).tolist():
                x0, x1, y0, y1 = min(bx[0], bx[2]), max(bx[0], bx[2]), min(bx[1], bx[3]), max(bx[1], bx[3])
                inside = (xy[:, 0] >= x0) & (xy[:, 0] <= x1) & (xy[:, 1] >= y0) & (xy[:, 1] <= y1)
                if bool(inside.any()):
                    hit = True
                    break
            hits.append(hit)
    return scores, coords, hits
