from oracle.tv042 import KeypointRCNNHeads, KeypointRCNNPredictor  # noqa
