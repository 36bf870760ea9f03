from oracle.tv042_det import BoxCoder  # noqa
