from oracle.tv042 import FeaturePyramidNetwork, LastLevelMaxPool  # noqa
