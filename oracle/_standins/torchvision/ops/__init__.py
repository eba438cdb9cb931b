from .deform_conv import deform_conv2d
