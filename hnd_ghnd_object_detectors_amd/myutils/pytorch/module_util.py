"""get_module / freeze helpers (used at src/distillation/tool.py:28-29,55-56 and src/mimic_runner.py:34-35,132,136)."""


def _unwrap(module):
    return module.module if hasattr(module, 'module') and not hasattr(module, 'transform') else module


def get_module(root_module, module_path):
    module = _unwrap(root_module)
    for name in module_path.split('.'):
        module = getattr(module, name)
    return module


def freeze_module_params(module):
    for p in module.parameters():
        p.requires_grad = False


def unfreeze_module_params(module):
    for p in module.parameters():
        p.requires_grad = True


def get_updatable_param_names(module):
    return [name for name, p in module.named_parameters() if p.requires_grad]


def count_params(module):
    return sum(p.numel() for p in module.parameters())
