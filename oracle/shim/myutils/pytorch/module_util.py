from oracle.myutils_r import (get_module, freeze_module_params, unfreeze_module_params,  # noqa
                              get_updatable_param_names, count_params, get_components)
