"""Programmatic builders for the hnd/ghnd YAML schema (reference config/{hnd,ghnd}/*.yaml).

The reference's own YAML files load unchanged through myutils.common.yaml_util (``!join`` supported); this module
produces the same dictionaries for tests, bench.py and smoke(), and tools/gen_configs.py dumps them to config/.
"""
from collections import OrderedDict

MODELS = ('faster_rcnn', 'mask_rcnn', 'keypoint_rcnn')
FROZEN = ['backbone.body.layer2', 'backbone.body.layer3', 'backbone.body.layer4', 'backbone.fpn', 'rpn', 'roi_heads']


def _term(layer):
    path = 'backbone.body.' + layer
    return {'ts_modules': [path, path], 'criterion': {'type': 'MSELoss', 'params': {'reduction': 'sum'}},
            'factor': 1.0}


def make_config(model='faster_rcnn', method='ghnd', bch=3, batch_size=4, pretrained=True, min_size=None,
                max_size=None, ckpt_root='./resource/ckpt'):
    assert model in MODELS and method in ('hnd', 'ghnd')
    keypoint = model == 'keypoint_rcnn'
    dataset = 'coco2017'
    root = './resource/dataset/' + dataset
    ann = 'person_keypoints' if keypoint else 'instances'
    splits = {}
    for split, sub, rm in (('train', 'train2017', True), ('val', 'val2017', False), ('test', 'val2017', False)):
        splits[split] = {'images': '%s/%s' % (root, sub), 'annotations': '%s/annotations/%s_%s.json' % (root, ann, sub),
                         'remove_non_annotated_imgs': rm, 'jpeg_quality': None}
    params = {'num_classes': 2 if keypoint else 91, 'pretrained': pretrained}
    if keypoint:
        params['num_keypoints'] = 17
    if min_size is not None:
        params['min_size'] = min_size
    if max_size is not None:
        params['max_size'] = max_size
    t_exp = '%s-%s-backbone_resnet50' % (dataset, model)
    s_exp = '%s-%s-backbone_custom_resnet50_from_%s-backbone_resnet50-b%dch' % (dataset, model, model, bch)
    layers = ['layer1'] if method == 'hnd' else ['layer1', 'layer2', 'layer3', 'layer4']
    return {
        'dataset': {'name': dataset, 'root': root, 'num_workers': 4, 'aspect_ratio_group_factor': 3,
                    'splits': splits},
        'teacher_model': {'name': model,
                          'backbone': {'name': 'resnet50', 'params': {'pretrained': pretrained,
                                                                      'freeze_layers': True}},
                          'params': dict(params), 'experiment': t_exp,
                          'ckpt': '%s/org/%s.pt' % (ckpt_root, t_exp)},
        'student_model': {'name': model,
                          'backbone': {'name': 'custom_resnet50',
                                       'params': {'pretrained': pretrained, 'freeze_layers': False,
                                                  'layer1': {'name': 'Bottleneck4LargeResNet',
                                                             'bottleneck_channel': bch}}},
                          'bottleneck_transformer': {'order': ['quantizer', 'dequantizer'],
                                                     'components': {'quantizer': {'params': {'num_bits': 8}},
                                                                    'dequantizer': {'params': {'num_bits': 8}}}},
                          'params': dict(params), 'distill_backbone_only': True, 'frozen_modules': list(FROZEN),
                          'experiment': s_exp, 'ckpt': '%s/%s/%s.pt' % (ckpt_root, method, s_exp)},
        'train': {'num_epochs': 35 if keypoint else 20, 'batch_size': batch_size, 'log_freq': 1000,
                  'optimizer': {'type': 'Adam', 'params': {'lr': 0.001}},
                  'criterion': {'type': 'general', 'params': {'org_loss_factor': 0.0},
                                'terms': OrderedDict((l, _term(l)) for l in layers)},
                  'scheduler': {'type': 'MultiStepLR',
                                'params': {'milestones': [9, 27] if keypoint else [5, 15], 'gamma': 0.1}}},
        'test': {'batch_size': 1},
    }


def make_ext_config(bch=3, batch_size=2, pretrained=True, min_size=None, max_size=None, threshold=0.01,
                    ckpt_root='./resource/ckpt'):
    """the neural-filter schema of the reference's config/ext/keypoint_rcnn-backbone_ext_resnet50-b3ch.yaml"""
    base = make_config('keypoint_rcnn', 'ghnd', bch, batch_size, pretrained, min_size, max_size, ckpt_root)
    for split in base['dataset']['splits'].values():
        split['remove_non_annotated_imgs'] = False          # the filter needs the person-free images
    model = base['student_model']
    for key in ('distill_backbone_only', 'frozen_modules'):
        del model[key]
    model['backbone']['params']['freeze_layers'] = True
    model['backbone']['ext_config'] = {
        'backbone_frozen': True, 'threshold': threshold,
        'ckpt': '%s/ext/coco2017-keypoint_rcnn-backbone_ext_custom_resnet50-b%dch.pt' % (ckpt_root, bch)}
    return {'dataset': base['dataset'], 'model': model,
            'train': {'num_epochs': 30, 'batch_size': batch_size, 'log_freq': 10000,
                      'optimizer': {'type': 'SGD', 'params': {'lr': 0.001, 'momentum': 0.9, 'weight_decay': 0.0001}},
                      'scheduler': {'type': 'MultiStepLR', 'params': {'milestones': [15, 25], 'gamma': 0.1}}},
            'test': {'batch_size': 1}}


def make_org_config(model='faster_rcnn', pretrained=True, min_size=None, max_size=None, ckpt_root='./resource/ckpt'):
    """the schema of the reference's config/org/<model>-backbone_resnet50.yaml (the original detectors: what
    src/coco_runner.py trains / evaluates and the hnd/ghnd teachers are loaded from)"""
    base = make_config(model, 'ghnd', 3, 2, pretrained, min_size, max_size, ckpt_root)
    keypoint = model == 'keypoint_rcnn'
    return {'dataset': base['dataset'], 'model': base['teacher_model'],
            'train': {'num_epochs': 46 if keypoint else 26, 'batch_size': 2, 'log_freq': 1000,
                      'optimizer': {'type': 'SGD', 'params': {'lr': 0.0075, 'momentum': 0.9, 'weight_decay': 0.0001}},
                      'scheduler': {'type': 'MultiStepLR',
                                    'params': {'milestones': [36, 43] if keypoint else [16, 22], 'gamma': 0.1}}},
            'test': {'batch_size': 1}}
