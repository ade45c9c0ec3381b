"""FASTConvFormer -- mirror of unirec/model/sequential/fastconvformer.py:20-81: ConvFormer whose mixer is the spectral layer
``irfft(rfft(x) * rfft(pad(conv_weight)), norm='ortho')``, i.e. a circular convolution scaled by 1/sqrt(L); evaluated in the
time domain by the same HIP mixer kernels (fast = 1).  hidden_act is fixed to gelu, as in the reference.  Reference
state_dict names: encoder.{i}.filterlayer.conv_weight [1,K,d] (trained), ...filterlayer.zeros (constant buffer parameter) and the
unused ...filterlayer.conv.depthwise_conv.{weight,bias} (created but never called): all are kept so checkpoints interchange."""
import torch
import torch.nn as nn

from ..base.reco_abc import ParamHolder
from .convformer import ConvFormer


class FASTConvFormer(ConvFormer):
    FAST = True

    def _act(self):
        return "gelu"          # fastconvformer.py:71

    def _padding_mode(self):
        return 0               # the spectral layer is circular whatever the config says (fastconvformer.py:32-40)

    def _mixer_holder(self, v, o, d, K):
        fl = nn.Module()
        fl.register_parameter("conv_weight", v(o[0], (1, K, d)))
        with torch.no_grad():
            fl.conv_weight.copy_(torch.randn(1, K, d, device=self.device) * 0.02)            # fastconvformer.py:23
        fl.register_parameter("zeros", nn.Parameter(torch.zeros(1, self.max_seq_len - K, d, device=self.device), requires_grad=False))
        # the reference also builds a depth-wise Conv1d that forward() never uses; it exists in its state_dict
        fl.conv = nn.Module()
        fl.conv.depthwise_conv = ParamHolder(weight=nn.Parameter(torch.randn(d, 1, K, device=self.device) * 5e-3, requires_grad=False),
                                             bias=nn.Parameter(torch.randn(d, device=self.device) * 5e-3, requires_grad=False))
        return fl
