"""Bottleneck transformers (mirror of the reference's src/structure/transformer.py:131-174).

The reference applies them ONLY at evaluation time with ``-transform_bottleneck`` (src/models/mimic/base.py:54-57;
mimic_runner.py:90 disables them while distilling), so on this build they are configuration objects: constructing
them from the YAML works (configs load unchanged), applying them raises until the eval-time codec is built
(SURVEY.md section 8f, row f1).
"""


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


class DataLogger(object):
    def __init__(self, num_bits=8):
        self.num_bits4quant = num_bits


class _EvalOnly(object):
    def __init__(self, num_bits=8):
        self.num_bits = num_bits

    def __call__(self, z, target):
        raise NotImplementedError('%s: the eval-time bottleneck codec is not built yet (SURVEY.md 8f, row f1)'
                                  % type(self).__name__)


class Quantizer(_EvalOnly):
    pass


class Dequantizer(_EvalOnly):
    pass


TRANSFORMER_CLASS_DICT = {'quantizer': Quantizer, 'dequantizer': Dequantizer}


def get_bottleneck_transformer(transformer_config):
    components = []
    for name in transformer_config['order']:
        if name not in TRANSFORMER_CLASS_DICT:
            raise KeyError('transformer `{}` is not expected'.format(name))
        components.append(TRANSFORMER_CLASS_DICT[name](**transformer_config['components'][name]['params']))
    return Compose(components) if components else None
