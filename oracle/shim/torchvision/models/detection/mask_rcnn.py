from oracle.tv042 import MaskRCNNHeads, MaskRCNNPredictor  # noqa
