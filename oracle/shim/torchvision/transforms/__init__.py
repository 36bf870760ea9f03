from . import functional  # noqa
