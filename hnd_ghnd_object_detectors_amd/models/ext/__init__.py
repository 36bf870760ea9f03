"""Neural-filter (ext) backbone factory (mirror of the reference's src/models/ext/__init__.py)."""
from .backbone import ExtBackboneWithFPN


def get_ext_fpn_backbone(base_backbone, ext_config, freeze_layers):
    if freeze_layers:
        for name, parameter in base_backbone.named_parameters():
            if not any(stage in name for stage in ('layer2', 'layer3', 'layer4')):
                parameter.requires_grad_(False)
    c2 = base_backbone.inplanes // 8
    return ExtBackboneWithFPN(base_backbone, {'layer1': 0, 'layer2': 1, 'layer3': 2, 'layer4': 3},
                              [c2, c2 * 2, c2 * 4, c2 * 8], 256, ext_config)
