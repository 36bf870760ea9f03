"""Evaluation of an original detector (role of the reference's src/coco_runner.py).

``python -m hnd_ghnd_object_detectors_amd.coco_runner --config config/org/faster_rcnn-backbone_resnet50.yaml``
builds ``config['model']`` (the detector the hnd/ghnd teachers are loaded from), reads the test split of the yaml and
runs ``main_util.evaluate`` (reference :131-132): the eval-mode detector on the HIP path -- RPN, RoI heads with their
mask / keypoint branches, NMS -- and the bbox (+ segm / keypoints) COCO statistics.

``-train`` (reference :28-96: SGD on the detector's own RPN / RoI losses) is NOT built: the detection losses, their
matchers / samplers and the backward through the heads are outside the distillation hot path; the
flag is accepted so that the reference's command lines parse, and raises.
"""
import argparse

import torch

from .models import get_model
from .myutils.common import yaml_util
from .utils import data_util, main_util


def get_argparser():
    argparser = argparse.ArgumentParser(description=__doc__)
    argparser.add_argument('--config', required=True, help='yaml config file')
    argparser.add_argument('--device', default='cuda', help='device')
    argparser.add_argument('--json', help='dictionary to overwrite config')
    argparser.add_argument('-train', action='store_true', help='train a model (not built: see the module docstring)')
    argparser.add_argument('-host_float_input', action='store_true',
                           help='ship float CHW images from the loader instead of decoded uint8 (reference behaviour)')
    # distributed parameters
    argparser.add_argument('--world_size', default=1, type=int, help='number of distributed processes')
    argparser.add_argument('--dist_url', default='env://', help='url used to set up distributed training')
    return argparser


def main(args):
    distributed, _ = main_util.init_distributed_mode(args.world_size, args.dist_url)
    config = yaml_util.load_yaml_file(args.config)
    if args.json is not None:
        main_util.overwrite_config(config, args.json)
    if args.train:
        raise NotImplementedError('coco_runner -train: the detectors\' own training losses are outside this build '
                                  '(only their evaluation is: the validation path of the distillation runner)')
    if not torch.cuda.is_available():
        raise RuntimeError('the HIP path needs an MI355X (no CPU fallback exists)')
    device = torch.device(args.device)
    print(args)
    print('Loading data')
    _, _, _, test_data_loader = data_util.get_coco_data_loaders(config['dataset'], config['train']['batch_size'],
                                                               distributed, decoded=not args.host_float_input)
    print('Creating model')
    model = get_model(config['model'], device)
    return main_util.evaluate(model, test_data_loader, device=device)


if __name__ == '__main__':
    main(get_argparser().parse_args())
