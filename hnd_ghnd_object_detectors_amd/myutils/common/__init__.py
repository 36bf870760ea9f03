from . import file_util, yaml_util  # noqa: F401
