"""Model factory (mirror of the reference's src/models/__init__.py:38-70) + re-exported checkpoint I/O."""
from ..structure.transformer import get_bottleneck_transformer
from .ckpt import load_ckpt, save_ckpt, unwrap  # noqa: F401
from .org import rcnn


def get_model(model_config, device, strict=True, bottleneck_transformer=None):
    """Build the detector named by ``model_config['name']`` from its YAML section, restore ``ckpt`` if the file
    exists, move it to ``device``."""
    name = model_config['name']
    if name not in rcnn.MODEL_CLASS_DICT:
        raise ValueError('model_name `{}` is not expected'.format(name))
    if bottleneck_transformer is None:
        codec_config = model_config.get('bottleneck_transformer')
        if codec_config is not None:
            bottleneck_transformer = get_bottleneck_transformer(codec_config)
    detector = rcnn.get_model(name, backbone_config=model_config['backbone'], strict=strict,
                              bottleneck_transformer=bottleneck_transformer, **model_config['params'])
    load_ckpt(model_config['ckpt'], model=detector, strict=strict)
    return detector.to(device)


def get_iou_types(model):
    kinds = {'segm': rcnn.MaskRCNN, 'keypoints': rcnn.KeypointRCNN}
    detector = unwrap(model)
    return ['bbox'] + [k for k, cls in kinds.items() if isinstance(detector, cls)]
