from oracle.tv042 import TwoMLPHead, FastRCNNPredictor  # noqa
