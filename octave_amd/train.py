"""The training step of the hot path (a17: absent from the reference, whose OctaScribbleNet.forward
raises -- models/octa.py:59-60; defined in SURVEY.md 3.5):

    att, agg, _ = net.segmentor(x);  p = softmax(agg, 1)
    L_seg = WPCE(p, ys) [+ Dice(p, ys)] + kl_w * InterlayerDivergence([p, *att]) + adv_w * LSGen(D(att))
    L_seg.backward(); Adam(segmentor)
    L_d = LSDisc(D(real_pyramid), D([a.detach() for a in att])); L_d.backward(); Adam(discriminator)

Data parallel: one process per GPU, gradients live in two flat fp32 arenas (segmentor,
discriminator) that are all-reduced over RCCL (torch.distributed backend "nccl" on ROCm) and then
consumed by ONE fused Adam launch each.  BatchNorm statistics and the WPCE class weights are
per-replica, like DistributedDataParallel on the reference would be.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import functional as F_
from ._lib import lib
from .layers import defer_bn_counters, flush_bn_counters


class FlatArena:
    """Flat fp32 storage for the grad-bearing parameters of a module, their gradients and the Adam
    moments.  Parameters keep their logical shapes AND strides (channels-last conv weights stay so)."""

    def __init__(self, params: Sequence[nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            n = (n + 3) // 4 * 4            # 16-byte aligned slots
            offs.append(n)
            n += p.numel()
        self.numel = (n + 3) // 4 * 4
        self.p = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not dense:
                    p.data = p.data.contiguous()
                view = self.p.as_strided(p.shape, p.stride(), o)
                view.copy_(p.data)
                p.data = view
                p.grad = self.g.as_strided(p.shape, p.stride(), o)
        self.step_count = 0

    def zero_grad(self):
        self.g.zero_()

    def all_reduce(self, world: int):
        if world > 1:
            dist.all_reduce(self.g, op=dist.ReduceOp.SUM)

    def adam(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        self.step_count += 1
        lib().octa_adam_step(self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.numel, lr, betas[0], betas[1],
                             eps, weight_decay, self.step_count, grad_scale, torch.cuda.current_stream().cuda_stream)
        F_.bump_weight_epoch()


def mask_pyramid(mask: Tensor, levels: int = 5) -> List[Tensor]:
    """Real multi-scale pyramid for the discriminator: nearest down-sampling by 2**i (contract of
    discriminator/blocks.py:114-125; views, no copy)."""
    return [mask[:, :, ::2 ** i, ::2 ** i] for i in range(levels)]


class TrainStep:
    def __init__(self, net: nn.Module, lr: float = 1e-4, lr_disc: Optional[float] = None, betas=(0.9, 0.999), compute_dtype=torch.bfloat16,
                 adversarial: bool = True, use_dice: bool = True, kl_weight: float = 0.1, adv_weight: float = 0.1):
        self.net = net
        self.seg, self.disc = net.segmentor, getattr(net, "discriminator", None)
        self.adversarial = adversarial and self.disc is not None
        self.use_dice, self.kl_weight, self.adv_weight = use_dice, kl_weight, adv_weight
        self.lr, self.lr_disc, self.betas = lr, lr_disc or lr, betas
        self.seg.compute_dtype = compute_dtype
        if self.disc is not None:
            self.disc.compute_dtype = compute_dtype
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # linear_head_* never receive a gradient on this path: keep them out of the arena
        seg_params = [p for n, p in self.seg.named_parameters() if not n.startswith("linear_head_")]
        self.seg_arena = FlatArena(seg_params)
        self.disc_arena = FlatArena(list(self.disc.parameters())) if self.adversarial else None
        F_.set_grad_sink(True)
        defer_bn_counters(True)

    def __call__(self, x: Tensor, ys: Tensor, real_pyramid: Optional[Sequence[Tensor]] = None) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        inv_world = 1.0 / self.world
        # ---- segmentor (generator) step
        self.seg_arena.zero_grad()
        att, agg, _ = self.seg(x)
        l = F_.wpce_dice(agg, ys, from_logits=True)
        loss = l[0] + l[1] if self.use_dice else l[0]
        out["wpce"], out["dice"] = l[0].detach(), l[1].detach()
        if self.adversarial:
            self.disc_arena.zero_grad()
            p = F_.class_softmax(agg)
            kl = F_.interlayer_kl([p, *att], [1] * len(att))[0]
            g_adv = F_.lsgan_generator(self.disc(att))
            loss = loss + self.kl_weight * kl + self.adv_weight * g_adv
            out["kl"], out["g_adv"] = kl.detach(), g_adv.detach()
        loss.backward()
        self.seg_arena.all_reduce(self.world)
        self.seg_arena.adam(self.lr, self.betas, grad_scale=inv_world)
        out["loss_seg"] = loss.detach()
        # ---- discriminator step
        if self.adversarial:
            if real_pyramid is None:
                raise ValueError("the adversarial step needs the real mask pyramid")
            self.disc_arena.zero_grad()          # drop what the generator step left in D's gradients
            d_real = self.disc(real_pyramid)
            d_fake = self.disc([a.detach() for a in att])
            l_d = F_.lsgan_discriminator(d_real, d_fake)
            l_d.backward()
            self.disc_arena.all_reduce(self.world)
            self.disc_arena.adam(self.lr_disc, self.betas, grad_scale=inv_world)
            out["loss_disc"] = l_d.detach()
        flush_bn_counters()
        return out

    def close(self):
        """Leave the fused-training mode (per-parameter gradients, immediate BatchNorm counters)."""
        F_.set_grad_sink(False)
        defer_bn_counters(False)
