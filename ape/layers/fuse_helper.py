from ape_amd.layers.fuse_helper import BiAttentionBlock, BiMultiHeadAttention  # noqa: F401
