from . import func_util, module_util, tensor_util  # noqa
