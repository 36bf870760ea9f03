from oracle.tv042 import load_state_dict_from_url  # noqa
