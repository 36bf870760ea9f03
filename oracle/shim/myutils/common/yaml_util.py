from oracle.myutils_r import load_yaml_file  # noqa
