"""mmcv.cnn.ConvModule stand-in: Conv2d(bias) + optional ReLU with mmcv's attribute names (.conv/.activate)."""
import torch.nn as nn


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 norm_cfg=None, act_cfg=dict(type='ReLU'), **kwargs):
        super().__init__()
        assert norm_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.conv(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def kaiming_init(*a, **k):
    raise RuntimeError('mmcv stand-in')


def constant_init(*a, **k):
    raise RuntimeError('mmcv stand-in')
