"""Factory of the neural-filter backbone (role of the reference's src/models/ext/__init__.py)."""
from .backbone import ExtBackboneWithFPN

_RETURN_LAYERS = {'layer1': 0, 'layer2': 1, 'layer3': 2, 'layer4': 3}
_TRAINABLE_STAGES = ('layer2', 'layer3', 'layer4')


def get_ext_fpn_backbone(base_backbone, ext_config, freeze_layers):
    """ResNet trunk + filter-aware layer getter + FPN; with ``freeze_layers`` only layer2-4 keep requires_grad."""
    if freeze_layers:
        frozen = [p for n, p in base_backbone.named_parameters() if not any(s in n for s in _TRAINABLE_STAGES)]
        for p in frozen:
            p.requires_grad_(False)
    width = base_backbone.inplanes // 8                      # 256 for ResNet-50
    channels = [width << i for i in range(4)]
    return ExtBackboneWithFPN(base_backbone, dict(_RETURN_LAYERS), channels, 256, ext_config)
