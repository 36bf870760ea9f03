from oracle.tv042 import BackboneWithFPN  # noqa
