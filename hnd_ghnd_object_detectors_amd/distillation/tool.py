"""DistillationBox (role of the reference's src/distillation/tool.py).

Contract kept: for every criterion term the teacher / student modules named by ``ts_modules`` get a forward hook
that stashes their output in ``module.__dict__['distillation_box']`` (:19-20, :25-35); ``forward(images, targets)``
runs the teacher without targets, then the student with targets (Keypoint R-CNN: both with the same randomly
drawn ``fixed_sizes``, :45-48), collects ``{loss_name: ((teacher_path, out), (student_path, out))}`` (:53-58) and
returns ``criterion(output_dict, org_loss_dict)``.
MI355X specifics: OPT-IN (``HND_MERGE_TRUNK=1``; off by default -- measured 1.0-1.5 ms slower per step than two
independent chains on two streams, profiles/r04_trunk_sweep.txt), layers 2-4 and the feature pyramid of BOTH networks can
run as one pass over the concatenated batch while their (frozen) weights are bit-equal -- they are in every hnd / ghnd
config, the student being initialised from the teacher -- see ``engine.SharedTrunk``: the teacher's hooks then stash views
that are filled by the student's call, which is all the criterion needs.  By default the frozen teacher front (stem + layer1) is issued on a second HIP stream (it is independent of the student
forward, so the tails of one network's launches are filled by the other's); both models share one transformed
batch inside a transform scope.  Both feature pyramids -- executed as written, but read by nobody when
``org_loss_factor`` is 0 -- are issued on a third stream so they trail into the backward pass and fill the last-round
gaps of its dgrad launches (joined before the next forward): +0.8 % on a dedicated GPU (round 3, A/B on one box:
101.76 -> 100.94 ms).  It collapses (50x slower) when two PROCESSES time-share one GPU, so it is on by default only when
the local ranks do not outnumber the visible devices (``_defer_fpn_default``; ``HND_DEFER_FPN=0/1`` overrides).
"""
import os
import random

import torch
from torch import nn

from .. import engine as E
from ..models.ckpt import unwrap
from ..models.org.rcnn import KeypointRCNN
from ..myutils.pytorch import module_util
from .loss import get_loss

_SLOT = 'distillation_box'


def _defer_fpn_default():
    """HND_DEFER_FPN=0/1 decides when set.  Unset: on only when this process has its GPU to itself -- with more local
    ranks than visible devices (two processes time-sharing one GPU) the third stream makes the step ~50x slower
    (module docstring), so it stays off there and says so once."""
    env = os.environ.get('HND_DEFER_FPN')
    if env is not None:
        return env != '0'
    if not E.process_owns_device():
        if not _WARNED['defer_fpn']:
            _WARNED['defer_fpn'] = True
            print('DistillationBox: more local ranks than visible GPUs: feature pyramids and the head\'s weight gradients '
                  'stay on the main stream (HND_DEFER_FPN=1 / HND_WGRAD_STREAM=1 force the side streams)')
        return False
    return True


_WARNED = {'defer_fpn': False}


def _stash_output(module, inputs, output):
    module.__dict__[_SLOT]['output'] = output


class DistillationBox(nn.Module):
    def __init__(self, teacher_model, student_model, criterion_config):
        super().__init__()
        self.teacher_model, self.student_model = teacher_model, student_model
        self.target_module_pairs = []
        for loss_name, term in criterion_config['terms'].items():
            paths = tuple(term['ts_modules'])
            self.target_module_pairs.append(paths)
            for model, path, is_teacher in ((teacher_model, paths[0], True), (student_model, paths[1], False)):
                module = module_util.get_module(unwrap(model), path)
                module.__dict__[_SLOT] = {'loss_name': loss_name, 'path_from_root': path, 'is_teacher': is_teacher}
                module.register_forward_hook(_stash_output)
        self.criterion = get_loss(criterion_config)
        self.require_adjustment = isinstance(unwrap(student_model), KeypointRCNN)
        self.overlap_teacher = os.environ.get('HND_TEACHER_STREAM', '1') != '0'
        self._side_stream = None
        # the criterion ignores the models' own outputs (org_loss_factor 0): their FPNs may trail into the backward
        # ... unless a term sits on a pyramid map (hooks on backbone.fpn.*): then the pyramids are part of the loss
        fpn_terms = any('.fpn.' in p or p.startswith('fpn.') for pair in self.target_module_pairs for p in pair)
        self.defer_fpn = _defer_fpn_default() and getattr(self.criterion, 'org_loss_factor', 1) == 0 and not fpn_terms
        self._fpn_stream = None
        self._trunk = None          # engine.SharedTrunk, built lazily when both backbones qualify

    def _shared_trunk(self):
        """Opt-in (``HND_MERGE_TRUNK=1``; default off, see engine.MERGE_TRUNK): the SharedTrunk of this pair when layers
        2-4 + FPN of teacher and student are frozen and bit-equal (every hnd / ghnd config: the student is initialised from
        the teacher, src/models/org/rcnn.py:444-450), else None -- then each network runs its own pass.  The teacher's
        layer2-4 / FPN calls return views that are FILLED ONLY BY THE STUDENT'S CALL, so the teacher must stop at its
        backbone (``distill_backbone_only``): a teacher that went on into its RPN / RoI heads would read last step's
        features."""
        if not E.MERGE_TRUNK:
            return None
        teacher, student = unwrap(self.teacher_model), unwrap(self.student_model)
        if not getattr(teacher, 'distill_backbone_only', False):
            return None
        t_bb, s_bb = getattr(teacher, 'backbone', None), getattr(student, 'backbone', None)
        if teacher.training or t_bb is None or s_bb is None or not E.SharedTrunk.structure_ok(t_bb, s_bb):
            return None
        if self._trunk is None or self._trunk.backbones != (t_bb, s_bb) or self._trunk.LAYERS[0] != E.MERGE_FROM:
            self._trunk = E.SharedTrunk(t_bb, s_bb)
        return self._trunk if self._trunk.weights_equal() else None

    def _run_models(self, images, targets, extra):
        trunk = self._shared_trunk() if images[0].is_cuda else None
        if trunk is not None:
            trunk.begin()
        E.MERGE['trunk'] = trunk
        try:
            return self._run_models_inner(images, targets, extra)
        finally:
            E.MERGE['trunk'] = None

    def _run_models_inner(self, images, targets, extra):
        teacher = unwrap(self.teacher_model)
        overlap = self.overlap_teacher and images[0].is_cuda and not E.PROFILE['enabled']
        if not overlap:
            self.teacher_model(images, **extra)
            return self.student_model(images, targets, **extra)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=images[0].device)
        main = torch.cuda.current_stream()
        if self.defer_fpn:
            if self._fpn_stream is None:
                self._fpn_stream = torch.cuda.Stream(device=images[0].device)
            main.wait_stream(self._fpn_stream)       # last step's pyramids are done before their inputs are rewritten
            E.DEFER_FPN['stream'] = self._fpn_stream
        teacher.transform(images, None, extra.get('fixed_sizes'))   # shared batch, produced once on the main stream
        self._side_stream.wait_stream(main)
        with torch.cuda.stream(self._side_stream):
            self.teacher_model(images, **extra)
        try:
            student_out = self.student_model(images, targets, **extra)
        finally:
            E.DEFER_FPN['stream'] = None
        main.wait_stream(self._side_stream)
        return student_out

    def forward(self, images, targets):
        extra = {}
        if self.require_adjustment:
            sizes = unwrap(self.teacher_model).transform.min_size
            extra['fixed_sizes'] = [random.choice(sizes) for _ in images]
        E.transform_scope_begin()
        try:
            org_loss_dict = self._run_models(images, targets, extra)
        finally:
            E.transform_scope_end()
        output_dict = {}
        for teacher_path, student_path in self.target_module_pairs:
            t = module_util.get_module(unwrap(self.teacher_model), teacher_path).__dict__[_SLOT]
            s = module_util.get_module(unwrap(self.student_model), student_path).__dict__[_SLOT]
            for slot in (t, s):
                if 'output' not in slot:
                    raise NotImplementedError(
                        'the forward hook on `%s` never fired: that module executes fused inside its parent on the HIP '
                        'path and has no tensor of its own.  Hookable: backbone.body.layerN, backbone.body.layerN.K (a '
                        'Bottleneck), backbone.body.layer1.encoder / .decoder, backbone.fpn.layer_blocks.K'
                        % slot['path_from_root'])
            output_dict[t['loss_name']] = ((t['path_from_root'], t['output']), (s['path_from_root'], s['output']))
        return self.criterion(output_dict, org_loss_dict)
