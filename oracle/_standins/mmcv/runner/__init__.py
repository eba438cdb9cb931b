def load_checkpoint(*args, **kwargs):
    raise RuntimeError('mmcv stand-in: checkpoint loading is unavailable offline')
