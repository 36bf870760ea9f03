from oracle.tv042 import AnchorGenerator, RPNHead, RegionProposalNetwork, concat_box_prediction_layers  # noqa
