import torch.utils.data as _d


class CocoDetection(_d.Dataset):
    """import-only stub."""


class VOCDetection(_d.Dataset):
    """import-only stub."""
