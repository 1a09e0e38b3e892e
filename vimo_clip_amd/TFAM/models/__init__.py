from .AMO_CLIP import AMO_CLIP, AttentionLayer  # noqa: F401
