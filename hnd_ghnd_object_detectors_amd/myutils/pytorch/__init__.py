from . import func_util, module_util  # noqa: F401
