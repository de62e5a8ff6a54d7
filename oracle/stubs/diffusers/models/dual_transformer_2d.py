class DualTransformer2DModel:  # imported by unet_2d_blocks.py:22, never instantiated by SD 1.x/2.x
    def __init__(self, *a, **k):
        raise NotImplementedError
