"""YAML loading with the ``!join`` tag used by every reference config (e.g. config/ghnd/*.yaml:3,33,60)."""
import yaml


class _JoinLoader(yaml.FullLoader):
    pass


def _join(loader, node):
    return ''.join(str(part) for part in loader.construct_sequence(node))


_JoinLoader.add_constructor('!join', _join)


def load_yaml_file(file_path):
    with open(file_path, 'r') as fp:
        return yaml.load(fp, Loader=_JoinLoader)
