from ape_amd.modeling.ape_deta.deformable_transformer_vl import (DeformableDetrTransformerDecoderVL,  # noqa: F401
                                                                  DeformableDetrTransformerEncoderVL, DeformableDetrTransformerVL)
