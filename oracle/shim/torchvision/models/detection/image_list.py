from oracle.tv042 import ImageList  # noqa
