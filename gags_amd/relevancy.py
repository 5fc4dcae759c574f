"""Query-time relevancy: the part of eval/openclip_encoder.py that runs per pixel.

`RelevancyHead` holds the unit text embeddings the reference's OpenCLIPNetwork computes with the CLIP text encoder
(eval/openclip_encoder.py:31-39,66-74; the encoder itself needs the CLIP weights and is out of scope) and mirrors
    get_relevancy(embed [P,512], positive_id) -> [P,2]          (:42-56)
    get_max_across(sem_map [L,h,w,512])       -> [L,n_phrases,h,w]   (:96-111)
on one HIP kernel that reads every pixel embedding once for all phrases (the reference re-reads the map once per
phrase and level through torch.mm + stack + softmax + gather)."""
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
