import os
import pickle
import sys


def check_if_exists(file_path):
    return file_path is not None and os.path.exists(file_path)


def make_dirs(dir_path):
    os.makedirs(dir_path, exist_ok=True)


def make_parent_dirs(file_path):
    parent = os.path.dirname(file_path)
    if parent:
        os.makedirs(parent, exist_ok=True)


def get_binary_object_size(x, unit_size=1024):
    return sys.getsizeof(pickle.dumps(x)) / unit_size
