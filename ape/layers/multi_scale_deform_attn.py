from ape_amd.layers.multi_scale_deform_attn import *  # noqa: F401,F403
from ape_amd.layers.multi_scale_deform_attn import MultiScaleDeformableAttention, multi_scale_deformable_attn_pytorch  # noqa: F401
