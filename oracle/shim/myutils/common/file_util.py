from oracle.myutils_r import (check_if_exists, make_dirs, make_parent_dirs, get_binary_object_size,  # noqa
                              get_file_path_list)
