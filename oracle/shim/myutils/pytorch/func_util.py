from oracle.myutils_r import get_optimizer, get_scheduler, get_loss  # noqa
