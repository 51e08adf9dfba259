"""RIFE `Model` wrapper (optimiser, DDP, update / inference, checkpoints) shared by the Flow-2D
and Flow-3D entry points.  Mirrors Flow-2D/model/RIFE.py:19-336 and Flow-3D/model/RIFE.py:18-275.
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch.nn.parallel import DistributedDataParallel as DDP
from torch.optim import AdamW

from . import ops
from .ifnet import IFNet


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("opticalflowscivis_amd needs a ROCm GPU: the HIP hot path has no CPU "
                           "fallback (the reference hard-codes torch.device('cuda') as well, "
                           "Flow-3D/model/RIFE.py:16)")
    return torch.device("cuda", torch.cuda.current_device())


class ModelBase:
    nd = None

    def __init__(self, local_rank=-1, arbitrary=False, device=None):
        if arbitrary:
            raise NotImplementedError("IFNet_m (arbitrary-timestep variant) is legacy upstream code "
                                      "not reached by train.py; out of scope (SURVEY §2 #18)")
        self.dev = torch.device(device) if device is not None else default_device()
        self.flownet = IFNet(self.nd)
        self.device()
        # large weight decay "may avoid NaN loss" (RIFE.py:28)
        # (on the GPU the update runs as ONE fused multi-tensor launch instead of torch's eight `_foreach_*` passes:
        # same AdamW arithmetic, 0.1 instead of 0.4 ms per step)
        self.optimG = AdamW(self.flownet.parameters(), lr=1e-6, weight_decay=1e-3, fused=self.dev.type == "cuda")
        if local_rank != -1:
            if self.dev.type == "cuda":
                # gradients live inside the all-reduce buckets (no per-step grad -> bucket copies); the
                # net has no buffers to broadcast
                self.flownet = DDP(self.flownet, device_ids=[local_rank], output_device=local_rank,
                                   gradient_as_bucket_view=True, broadcast_buffers=False)
            else:  # host-side tests of the sharding logic (gloo); the ops themselves need a GPU
                self.flownet = DDP(self.flownet)

    def train(self):
        self.flownet.train()

    def eval(self):
        self.flownet.eval()

    def device(self):
        self.flownet.to(self.dev)

    # ---- checkpoints (Flow-3D/model/RIFE.py:44-64, same in Flow-2D).  The reference's train.py always
    # DDP-wraps the net (local_rank defaults to 0), so its .pkl files hold a state_dict whose keys carry the
    # "module." prefix, and its load_model keeps ONLY keys containing "module.".  The on-disk format here is
    # therefore always the prefixed one -- also from an unwrapped single-GPU model -- so that either side
    # loads the other's files; the loader accepts both forms.
    def load_model(self, model_name, path, rank=0):
        if rank > 0:
            return
        sd = torch.load('{}/{}'.format(path, model_name), map_location=self.dev)
        wrapped = isinstance(self.flownet, DDP)
        fixed = {}
        for k, v in sd.items():
            has = k.startswith("module.")
            if wrapped and not has:
                k = "module." + k
            elif not wrapped and has:
                k = k[len("module."):]
            fixed[k] = v
        self.flownet.load_state_dict(fixed)

    def save_model(self, model_name, path, rank=0):
        if rank == 0:
            sd = self.flownet.state_dict()
            if not isinstance(self.flownet, DDP):
                sd = {"module." + k: v for k, v in sd.items()}
            torch.save(sd, '{}/{}'.format(path, model_name))

    def _set_lr(self, learning_rate):
        if getattr(self, "_in_capture", False):
            return  # the graph must not bake a learning-rate fill; step() sets it before each replay
        for g in self.optimG.param_groups:
            if torch.is_tensor(g['lr']):  # capturable optimiser: the rate lives in a device tensor
                g['lr'].fill_(float(learning_rate))
            else:
                g['lr'] = learning_rate

    def graphed_update(self, imgs, gt, **update_kwargs):
        """Capture one training step -- forward, losses, backward, AdamW -- for inputs of this shape into
        ONE HIP graph and return `step(imgs, gt, learning_rate) -> (pred, info)`, which copies the batch
        into the graph's static inputs, replays it and returns the graph's static outputs (valid until
        the next call).  No entry point of the hot path allocates outside the caching allocator,
        synchronises or copies from the host, so the whole step is capturable; what the graph saves is
        the ~660 launches' CPU cost and the gaps between short kernels.  Single-process models only
        (capturing through DDP's reducer needs its own warm-up protocol and is not attempted).
        Model2D: pass `dataset=...` through `update_kwargs`; its NaN / > 10 guard on the distillation loss
        (RIFE.py:295) runs on the device, so the 2-D step captures as well.

        Side effects on the optimiser, which outlive the returned `step` (graph capture needs device-resident
        optimiser scalars): every param group becomes `capturable=True`, its `'lr'` a 0-dim device tensor and
        the AdamW step counters device tensors.  Later eager `update()` calls keep working (they write the new
        rate into that tensor), but code that reads `param_group['lr']` as a Python float -- schedulers,
        loggers, `state_dict()` consumers -- now receives a tensor: use `float(group['lr'])`."""
        if self.dev.type != "cuda":
            raise RuntimeError("HIP graphs need a GPU")
        if isinstance(self.flownet, DDP):
            raise NotImplementedError("graph capture of the DDP-wrapped model is not supported")
        # the optimiser state must live on the device: step counters and the learning rate
        for grp in self.optimG.param_groups:
            grp['capturable'] = True
            if not torch.is_tensor(grp['lr']):
                grp['lr'] = torch.tensor(float(grp['lr']), device=self.dev)
        for st in self.optimG.state.values():
            if torch.is_tensor(st.get('step')) and not st['step'].is_cuda:
                st['step'] = st['step'].to(self.dev)
        s_imgs, s_gt = imgs.clone(), gt.clone()
        # one warm-up step off the capture stream (allocator pools, lazily created state), then restore:
        # graphed_update() itself leaves weights and optimiser state untouched
        weights = copy.deepcopy(self.flownet.state_dict())
        saved = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                 for p, st in self.optimG.state.items()}
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self.update(s_imgs, s_gt, learning_rate=0.0, training=True, **update_kwargs)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self.flownet.load_state_dict(weights)
        # the optimiser state now EXISTS (a state created during capture would be re-zeroed by every
        # replay): put the pre-warm-up values back into the same tensors
        with torch.no_grad():
            for p_, st in self.optimG.state.items():
                old_st = saved.get(p_)
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if old_st is not None and torch.is_tensor(old_st.get(k)):
                            v.copy_(old_st[k])
                        else:
                            v.zero_()
        self.optimG.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        self._in_capture = True
        try:
            with torch.cuda.graph(graph):
                out = self.update(s_imgs, s_gt, training=True, **update_kwargs)
        finally:
            self._in_capture = False

        def step(imgs, gt, learning_rate):
            s_imgs.copy_(imgs)
            s_gt.copy_(gt)
            self._set_lr(learning_rate)
            graph.replay()
            # the replay moved the weights without autograd's version counters noticing: a later EAGER step must not
            # trust the convolution weight slabs the graph's own re-layout launch left behind
            ops.invalidate_prepared_weights()
            return out
        step.graph = graph
        return step


class Model3D(ModelBase):
    """Flow-3D/model/RIFE.py:18-275."""
    nd = 3

    def __init__(self, local_rank=-1, arbitrary=False, device=None):
        super().__init__(local_rank, arbitrary, device)
        self.lap = LapLoss()  # constructed by the reference too (RIFE.py:30); see update(lap_loss=...)

    def inference(self, img0, img1, scale_list=(4, 2, 1), TTA=False, timestep=0.5):
        imgs = torch.cat((img0, img1), 1)
        with ops.prepared_weights():  # the convolution weights are re-laid-out once per call, not once per layer
            flow, mask, merged, _, _, _ = self.flownet(imgs, scale_list, timestep=timestep)
        if TTA:
            raise NotImplementedError("TTA is 'not implemented' in the reference too (RIFE.py:76)")
        return merged[2], flow, mask

    def update(self, imgs, gt, learning_rate=0, mul=1, training=True, flow_gt=None, lap_loss=False):
        """`lap_loss=True` swaps the two L1 terms for the Laplacian-pyramid loss the reference has commented
        out (RIFE.py:126, 133: `(self.lap(merged[2], gt)).mean()`), computed by the 3-D pyramid kernels; the
        default is the reference's active path."""
        # forward, backward and the optimiser step see ONE set of weights: their re-laid-out slabs are built by one
        # launch at the first convolution and dropped when the step ends (ops.prepared_weights)
        with ops.prepared_weights():
            return self._update(imgs, gt, learning_rate, training, lap_loss)

    def _update(self, imgs, gt, learning_rate, training, lap_loss):
        self._set_lr(learning_rate)
        if training:
            self.train()
        else:
            self.eval()
        gt = gt.contiguous()  # (a channel slice of the loader's [B,3,...] batch: one copy here, none in the consumers)
        flow, mask, merged, flow_teacher, merged_teacher, loss_distill = self.flownet(
            (imgs, gt), scale=[4, 2, 1])
        sp = tuple(min(a, b) for a, b in zip(imgs.shape[2:], mask.shape[2:]))
        gt = gt[(slice(None), slice(None)) + tuple(slice(0, s) for s in sp)]
        pair_loss = self.lap if lap_loss else ops.l1_loss
        loss_l1 = pair_loss(merged[2], gt)          # RIFE.py:132 (fused |a-b| + reduction)
        loss_tea = pair_loss(merged_teacher, gt)    # RIFE.py:134
        # RIFE.py:141-143 also sums an L1 norm of all parameters that never reaches loss_G
        # (lambda_reg = 0 and the term is commented out of :158); it is not computed here.
        loss_G = loss_l1 * 1 + loss_tea * 1 + loss_distill * 0.1  # RIFE.py:151-158
        if training:
            self.optimG.zero_grad()
            loss_G.backward()
            self.optimG.step()
        else:
            flow_teacher = flow[2]
            merged_teacher = merged[2]
        return merged[2], {
            'merged_tea': merged_teacher, 'mask': mask, 'mask_tea': mask, 'flow': flow[2],
            'flow_tea': flow_teacher, 'loss_l1': loss_l1, 'loss_tea': loss_tea,
            'loss_distill': loss_distill, 'loss_G': loss_G,
        }


class LapLoss(torch.nn.Module):
    """Flow-2D/model/laplacian.py:76-88: 5-level Laplacian-pyramid L1 loss, the dominant loss term of
    Flow-2D.  One pyramid of (input - target), two HIP launches per level
    (csrc/laplacian.hip); CPU tensors raise ValueError like every other op."""

    def __init__(self, max_levels=5, channels=1):
        super().__init__()
        self.max_levels, self.channels = max_levels, channels

    def forward(self, input, target):
        if input.dim() == 5:
            # Flow-3D/model/laplacian.py (dead code in the reference, CPU scipy round trip): the real 3-D
            # pyramid of csrc/laplacian3d.hip -- parity unpinned, see ops._LapLoss3D
            return ops.laploss3d(input, target, self.max_levels)
        return ops.laploss2d(input, target, self.max_levels)


_FLOW_GT_DATASETS = ("pipedcylinder2d", "cylinder2d", "FluidSimML2d", "rectangle2d", "lbs2d")


class Model2D(ModelBase):
    """Flow-2D/model/RIFE.py:19-336."""
    nd = 2

    def __init__(self, local_rank=-1, arbitrary=False, device=None):
        super().__init__(local_rank, arbitrary, device)
        self.lap = LapLoss()

    def inference(self, img0, img1, scale_list=(4, 2, 1), TTA=False, timestep=0.5):
        imgs = torch.cat((img0, img1), 1)
        flow, mask, merged, _, _, _ = self.flownet(imgs, scale_list, timestep=timestep)
        if not TTA:
            return merged, flow, mask  # all three frames (RIFE.py:75)
        flow2, mask2, merged2, _, _, _ = self.flownet(imgs.flip(2).flip(3), scale_list,
                                                      timestep=timestep)
        return (merged[2] + merged2[2].flip(2).flip(3)) / 2

    def update(self, imgs, gt, dataset, learning_rate=0, mul=1, training=True, flow_gt=None):
        self._set_lr(learning_rate)
        gt_flow = None
        if dataset in _FLOW_GT_DATASETS:  # these loaders pack (data, u, v) per frame (RIFE.py:90-105)
            gt_flow = gt[:, 0, 1:3]
            img0, img1 = imgs[:, 0, :1], imgs[:, 1, :1]
            imgs = torch.cat((img0, img1), 1)
            gt = gt[:, 0, :1]
        else:
            img0, img1 = imgs[:, :1], imgs[:, 1:2]
        if training:
            self.train()
        else:
            self.eval()
        flow, mask, merged, flow_teacher, merged_teacher, loss_distill = self.flownet(
            (imgs, gt), scale=[4, 2, 1])
        mask = mask[2]
        h, w = min(img0.shape[2], mask.shape[2]), min(img0.shape[3], mask.shape[3])
        gt = gt[:, :, :h, :w]
        loss_flow = torch.tensor(0.)
        if gt_flow is not None:  # RIFE.py:134-146
            gt_flow = gt_flow[:, :, :h, :w]
            loss_flow = sum(F.l1_loss(f[:, 2:4], gt_flow) + F.l1_loss(f[:, :2], -gt_flow)
                            for f in (flow[0], flow[1], flow[2], flow_teacher)) / 8.
        loss_l1 = self.lap(merged[2], gt).mean()          # RIFE.py:148
        loss_tea = self.lap(merged_teacher, gt).mean()    # RIFE.py:152
        # L1 "regulariser" over block2/block_tea read from state_dict(): detached, so it shifts
        # loss_G but contributes no gradient (RIFE.py:177-188)
        # (one multi-tensor L1-norm launch instead of ~90 abs + ~90 sum + ~90 add launches per step: the C2 step is
        # launch-bound; the value differs from the reference's sequential fp32 running sum by summation order only,
        # ~1e-7 relative on a term weighted 1e-6)
        with torch.no_grad():
            reg = [p for n, p in self.flownet.state_dict().items() if "block2" in n or "block_tea" in n]
            l1_reg = torch.stack(torch._foreach_norm(reg, 1)).sum()
        # hot path a11: two half-pixel-shifted backward warps of merged[2] + Charbonnier
        loss_photo = ops.rife2d_photometric(flow[2], merged[2], img0, img1)  # RIFE.py:274-279
        lambda_l1, lambda_tea, lambda_distill = 1, 1, 0.01  # RIFE.py:283-289
        lambda_reg, lambda_photo, lambda_flow = 1e-6, 1e-5, 0
        # RIFE.py:295-296 `if math.isnan(loss_distill) or loss_distill > 10: loss_distill = 0`, evaluated on the
        # device: same value, and a gradient of exactly 0 into the distillation term (torch.where selects, the
        # distillation backward kernels emit 0 for a zero cotangent even over non-finite flows) -- no host
        # synchronisation per step, so the whole step can be captured into a HIP graph (graphed_update)
        if torch.is_tensor(loss_distill):
            discard = torch.isnan(loss_distill) | (loss_distill > 10.)
            loss_distill = torch.where(discard, torch.zeros_like(loss_distill), loss_distill)
        elif math.isnan(loss_distill) or loss_distill > 10.:
            loss_distill = 0.
        if dataset in ("droplet2d", "vimeo2d"):
            loss_flow = torch.tensor(0.)
        loss_G = loss_l1 * lambda_l1 + loss_tea * lambda_tea + loss_distill * lambda_distill + \
            l1_reg * lambda_reg + loss_photo * lambda_photo + loss_flow * lambda_flow
        if training:
            self.optimG.zero_grad()
            loss_G.backward()
            self.optimG.step()
        else:
            flow_teacher = flow[2]
            merged_teacher = merged[2]
        return merged[2], {
            'merged_tea': merged_teacher, 'mask': mask, 'mask_tea': mask, 'flow': flow[2][:, :2],
            'flow_tea': flow_teacher, 'loss_l1': loss_l1 * lambda_l1, 'loss_tea': loss_tea * lambda_tea,
            'loss_distill': loss_distill * lambda_distill, 'l1_reg': l1_reg * lambda_reg,
            'loss_photo': loss_photo * lambda_photo, 'loss_flow': loss_flow * lambda_flow,
            'loss_G': loss_G,
        }
