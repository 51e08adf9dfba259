"""Drop-in for the hot-path members of UPFlow/utils/tools.py (`tools`): `torch_warp`,
`boundary_dilated_warp.warp_im`, `occ_check_model`, plus the small `abstract_config` /
`abstract_model` bases the network needs.  Everything else in the reference's 1 772-line tools.py
(KITTI I/O, meters, augmentation, ...) is out of scope."""
import torch
import torch.nn as nn

from ... import ops


class tools:
    class abstract_config:
        """UPFlow/utils/tools.py:80-114: `update(dict)` sets the attributes that already exist."""

        def update(self, data: dict):
            for k in [n for n in dir(self) if not n.startswith('_')]:
                if k in data:
                    setattr(self, k, data[k])

        def get_dict(self):
            return {n: getattr(self, n) for n in dir(self) if not n.startswith('_') and
                    not callable(getattr(self, n))}

    class abstract_model(nn.Module):
        """UPFlow/utils/tools.py:116-131: state_dict checkpoints, optional shape-filtered load."""

        def save_model(self, save_path):
            torch.save(self.state_dict(), save_path)

        def load_model(self, load_path, if_relax=False, if_print=True):
            sd = torch.load(load_path, map_location="cpu")
            if if_relax:
                own = self.state_dict()
                own.update({k: v for k, v in sd.items() if k in own and v.shape == own[k].shape})
                sd = own
            self.load_state_dict(sd)

    @classmethod
    def torch_warp(cls, x, flo):
        """UPFlow/utils/tools.py:1317-1361 (a6): zero-padded bilinear warp, no validity mask."""
        return ops.warp2d_pwc(x, flo, with_mask=False)

    class boundary_dilated_warp:
        @classmethod
        def warp_im(cls, I_nchw, flow_nchw, start_n211):
            """UPFlow/utils/tools.py:533-541 (a7).  `start` may be [1,2,1,1] (the reference passes
            zeros of that shape, upflow.py:503) or [B,2,1,1]."""
            B = I_nchw.shape[0]
            start = None
            if start_n211 is not None:
                start = start_n211.reshape(-1, 2).to(I_nchw.dtype)
                if start.shape[0] == 1 and B > 1:
                    start = start.expand(B, 2)
                start = start.contiguous()
            return ops.warp2d_dilated(I_nchw, flow_nchw, start)

    class occ_check_model:
        """Forward-backward consistency occlusion masks (UPFlow/utils/tools.py:543-719).
        `__call__` is ONE HIP launch (ops.occ_check2d): both flow warps, the
        magnitudes, the threshold test and the outgoing masks.  The per-step methods below keep
        the reference's names for callers that use them directly."""

        def __init__(self, occ_type='for_back_check', occ_alpha_1=1.0, occ_alpha_2=0.05,
                     sum_abs_or_squar=True, obj_out_all='all'):
            assert occ_type in ('for_back_check', 'forward_warp')
            assert obj_out_all in ('obj', 'out', 'all')
            self.occ_type, self.obj_out_all = occ_type, obj_out_all
            self.occ_alpha_1, self.occ_alpha_2 = occ_alpha_1, occ_alpha_2
            self.sum_abs_or_squar = True  # the reference forces True ("false is not OK", :559)

        def __call__(self, flow_f, flow_b, scale=1):
            if self.occ_type != 'for_back_check':
                raise ValueError('not implemented')  # as in the reference (:567)
            # one launch; CPU tensors raise ValueError like every other op (no CPU fallback)
            return ops.occ_check2d(flow_f, flow_b, self.occ_alpha_1, self.occ_alpha_2, scale,
                                   self.obj_out_all)

        def _forward_backward_occ_check(self, flow_fw, flow_bw, scale=1):
            def mag(x):  # length_sq_v0 (:596-601): sum_c |x_c|
                return torch.sum(torch.pow(x ** 2, 0.5), dim=1, keepdim=True)

            mag_sq = mag(flow_fw) + mag(flow_bw)
            flow_bw_warped = tools.torch_warp(flow_bw, flow_fw)
            flow_fw_warped = tools.torch_warp(flow_fw, flow_bw)
            occ_thresh = self.occ_alpha_1 * mag_sq + self.occ_alpha_2 / scale
            occ_fw = mag(flow_fw + flow_bw_warped) < occ_thresh  # 0 = occluded
            occ_bw = mag(flow_bw + flow_fw_warped) < occ_thresh
            return occ_fw.float(), occ_bw.float()

        @classmethod
        def torch_outgoing_occ_check(cls, flow):
            B, C, H, W = flow.shape
            xx = torch.arange(W, device=flow.device, dtype=flow.dtype).view(1, 1, 1, W)
            yy = torch.arange(H, device=flow.device, dtype=flow.dtype).view(1, 1, H, 1)
            px, py = xx + flow[:, 0:1], yy + flow[:, 1:2]
            inside = (px <= W - 1) & (px >= 0) & (py <= H - 1) & (py >= 0)
            return inside.float()

        @classmethod
        def torch_get_obj_occ_check(cls, occ_mask, out_occ):
            return ((occ_mask == 1) | (out_occ == 0)).float()
