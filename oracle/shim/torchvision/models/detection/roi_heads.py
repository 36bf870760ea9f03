from oracle.tv042 import RoIHeads  # noqa
