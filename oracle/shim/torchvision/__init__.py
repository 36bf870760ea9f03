"""torchvision==0.4.2 subset for importing the reference (test infrastructure only)."""
__version__ = '0.4.2+oracle.shim'
from . import models, ops, datasets, transforms  # noqa
