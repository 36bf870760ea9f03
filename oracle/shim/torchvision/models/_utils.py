from oracle.tv042 import IntermediateLayerGetter  # noqa
