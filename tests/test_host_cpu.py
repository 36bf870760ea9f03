"""CPU: host logic of the drop-in surface (no GPU compute): module tree / state-dict layout, trainable set,
YAML loading incl. the reference's own files, schedulers, synthetic loader, meters."""
import os
import random

import pytest
import torch

from oracle import hnd_oracle as O
from tests import model_util as MU

REF_CONFIG = '/root/reference/config'


def test_module_tree_matches_reference_state_dict_layout():
    cfg = MU.config_for(model='faster_rcnn', method='ghnd', bch=3)
    t_sd, s_sd = MU.oracle_states(3)
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, torch.device('cpu'))
    assert list(student.state_dict().keys()) == list(s_sd.keys())
    assert list(teacher.state_dict().keys()) == list(t_sd.keys())
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import module_util
    names = module_util.get_updatable_param_names(student)
    assert names == O.trainable_keys(s_sd) and len(names) == 25
    assert sum(p.numel() for p in student.parameters() if p.requires_grad) == 586566
    assert module_util.get_updatable_param_names(teacher) == []
    # dotted paths used by DistillationBox / mimic_runner resolve
    for path in ('backbone.body.layer1', 'backbone.body.layer4', 'backbone.fpn', 'rpn', 'roi_heads',
                 'backbone.body.layer1.encoder', 'backbone.body.layer1.decoder'):
        module_util.get_module(student, path)
    assert student.backbone.body.layer1.use_bottleneck_transformer is False
    assert student.backbone.body.layer1.bottleneck_transformer is not None


@pytest.mark.parametrize('model,extra', [('mask_rcnn', 'roi_heads.mask_head.mask_fcn1.weight'),
                                         ('keypoint_rcnn', 'roi_heads.keypoint_predictor.kps_score_lowres.weight')])
def test_mask_and_keypoint_heads_are_checkpoint_compatible(model, extra):
    cfg = MU.config_for(model=model, method='ghnd', bch=3)
    t_sd, s_sd = MU.oracle_states(4, model, num_classes=cfg['teacher_model']['params']['num_classes'])
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, torch.device('cpu'))
    assert extra in student.state_dict()
    if model == 'keypoint_rcnn':
        assert tuple(student.transform.min_size) == (640, 672, 704, 736, 768, 800)


def test_holders_refuse_eager_compute():
    cfg = MU.config_for()
    t_sd, s_sd = MU.oracle_states(5)
    _, student = MU.build_pair(cfg, t_sd, s_sd, torch.device('cpu'))
    with pytest.raises(RuntimeError, match='parameter holder'):
        student.backbone.body.conv1(torch.zeros(1, 3, 8, 8))
    with pytest.raises(NotImplementedError):        # training-mode RPN (proposal losses) is never run by hnd/ghnd
        student.rpn(None, None)
    with pytest.raises(NotImplementedError):        # mask / keypoint branches stay parameter holders
        from hnd_ghnd_object_detectors_amd import hipnn
        hipnn.KeypointRCNNPredictor(512, 17)(None)


def test_generated_yaml_equals_builder_and_reference_yaml_loads():
    from hnd_ghnd_object_detectors_amd.configs import make_config
    from hnd_ghnd_object_detectors_amd.myutils.common import yaml_util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mine = yaml_util.load_yaml_file(os.path.join(root, 'config/ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml'))
    built = make_config('faster_rcnn', 'ghnd', 3)
    assert mine['student_model'] == built['student_model']
    assert list(mine['train']['criterion']['terms']) == ['layer1', 'layer2', 'layer3', 'layer4']
    if os.path.isdir(REF_CONFIG):          # the reference's own files load unchanged (!join tag)
        for method, name in (('ghnd', 'faster_rcnn'), ('hnd', 'mask_rcnn'), ('ghnd', 'keypoint_rcnn')):
            ref = yaml_util.load_yaml_file('%s/%s/%s-backbone_resnet50-b3ch.yaml' % (REF_CONFIG, method, name))
            gen = make_config(name, method, 3)
            for section in ('teacher_model', 'student_model', 'train', 'test', 'dataset'):
                assert ref[section] == gen[section], (method, name, section)
        from hnd_ghnd_object_detectors_amd.configs import make_org_config
        for name in ('faster_rcnn', 'mask_rcnn', 'keypoint_rcnn'):       # config/org: what coco_runner evaluates
            ref = yaml_util.load_yaml_file('%s/org/%s-backbone_resnet50.yaml' % (REF_CONFIG, name))
            assert ref == make_org_config(name) == yaml_util.load_yaml_file(
                os.path.join(root, 'config/org/%s-backbone_resnet50.yaml' % name)), name


def test_json_override_and_warmup_schedule():
    from hnd_ghnd_object_detectors_amd.utils import main_util
    cfg = {'a': {'b': 1, 'c': 2}, 'd': 3}
    main_util.overwrite_config(cfg, '{"a": {"b": 5}, "e": {"f": 1}}')
    assert cfg == {'a': {'b': 5, 'c': 2}, 'd': 3, 'e': {'f': 1}}
    # lr_t = 1e-3 * (1e-3 * (1 - t/w) + t/w) for t < w   
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    sched = main_util.warmup_lr_scheduler(opt, 4, 1e-3)
    lrs = []
    for _ in range(6):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        sched.step()
    expect = [1e-3 * (1e-3 * (1 - t / 4) + t / 4) if t < 4 else 1e-3 for t in range(6)]
    assert all(abs(a - b) < 1e-12 for a, b in zip(lrs, expect))


def test_synthetic_loader_and_meters():
    from hnd_ghnd_object_detectors_amd.utils import data_util, misc_util
    ld = data_util.SyntheticDetectionLoader(2, 3, 16, 24, 'mask_rcnn', rank=1)
    batches = list(ld)
    assert len(ld) == 2 and len(batches) == 2 and len(batches[0][0]) == 3
    assert batches[0][0][0].shape == (3, 16, 24) and batches[0][1][0]['masks'].shape == (1, 16, 24)
    other = list(data_util.SyntheticDetectionLoader(2, 3, 16, 24, 'mask_rcnn', rank=0))
    assert not torch.equal(other[0][0][0], batches[0][0][0])           # ranks see different shards
    m = misc_util.MetricLogger()
    for v in (1.0, 3.0, 2.0):
        m.update(loss=torch.tensor(v))
    assert m.loss.median == 2.0 and abs(m.loss.global_avg - 2.0) < 1e-12


@pytest.mark.parametrize('workers', [0, 2])
def test_synthetic_loader_pool_hands_out_the_same_first_batches_and_then_rotates(workers):
    """on a host whose CPU share per rank cannot generate a batch per step the loader keeps a rotating pool
    of P pre-generated batches -- batches 0 .. P-1 hold the values the unpooled loader makes (so the pinned first-step losses
    still gate the run), batch k >= P is batch k % P again (the SAME tensors: nothing is generated), and the generation cost
    the step would have paid is accounted for separately"""
    from hnd_ghnd_object_detectors_amd.utils import data_util
    plain = list(data_util.SyntheticDetectionLoader(5, 2, 16, 24, 'faster_rcnn', seed=9, rank=3, workers=workers))
    pooled_ld = data_util.SyntheticDetectionLoader(5, 2, 16, 24, 'faster_rcnn', seed=9, rank=3, workers=workers, pool_batches=2)
    pooled = list(pooled_ld)
    assert len(pooled) == 5
    for k in range(5):
        for a, b in zip(pooled[k][0], plain[k % 2][0]):
            assert torch.equal(a, b), k
        assert all(torch.equal(pooled[k][1][i]['boxes'], plain[k][1][i]['boxes']) for i in range(2))
    assert all(pooled[2][0][i] is pooled[0][0][i] for i in range(2))          # the pool's own tensors, not copies
    plain_ld = data_util.SyntheticDetectionLoader(3, 2, 16, 24, 'faster_rcnn', seed=9, workers=workers)
    list(plain_ld)
    assert plain_ld.gen_batches == 3 and plain_ld.gen_cpu_s > 0 and pooled_ld.gen_batches == 0


def test_synthetic_loader_pool_slot_is_made_once_when_feeder_threads_race_for_it(monkeypatch):
    """with 2 feeder threads and a pool of 2, batch 2 asks for slot 0 while batch 0 may still be generating it: one thread
    makes the slot, the other waits (seen once under load: two copies of slot 0, the second replacing the first)"""
    import time
    from hnd_ghnd_object_detectors_amd.utils import data_util
    ld = data_util.SyntheticDetectionLoader(6, 2, 16, 24, 'faster_rcnn', seed=9, workers=3, pool_batches=2)
    made, real = [], ld._generate

    def slow(k, es):
        made.append(k)
        time.sleep(0.05)
        return real(k, es)
    monkeypatch.setattr(ld, '_generate', slow)
    got = list(ld)
    assert sorted(made) == [0, 1], made
    assert all(got[k][0][i] is got[k % 2][0][i] for k in range(6) for i in range(2))


def test_load_ckpt_tuple_arity_and_roundtrip(tmp_path):
    from hnd_ghnd_object_detectors_amd.models import load_ckpt, save_ckpt
    assert load_ckpt(str(tmp_path / 'missing.pt')) == (None, None)      # reference quirk: 2-tuple when missing
    cfg = MU.config_for()
    t_sd, s_sd = MU.oracle_states(6)
    _, student = MU.build_pair(cfg, t_sd, s_sd, torch.device('cpu'))
    opt = torch.optim.Adam([p for p in student.parameters() if p.requires_grad], lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [5, 15], 0.1)
    path = str(tmp_path / 'sub' / 'ckpt.pt')
    save_ckpt(student, opt, sch, 0.5, cfg, None, path)
    ck = torch.load(path, weights_only=False)
    assert sorted(ck) == ['args', 'best_value', 'config', 'lr_scheduler', 'model', 'optimizer']
    best, c2, a2 = load_ckpt(path, model=student, optimizer=opt, lr_scheduler=sch)
    assert best == 0.5 and c2['train']['batch_size'] == 4 and a2 is None


def test_decoded_input_pipeline_host_side_matches_reference_fixture():
    """ToTensor keeps uint8 (DecodedImage), RandomHorizontalFlip defers the image flip but flips the targets
    exactly like the reference; the deferred float image equals the oracle's ToTensor(+flip)."""
    from tests import golden_util as G
    from hnd_ghnd_object_detectors_amd.structure.transformer import DecodedImage, RandomHorizontalFlip, ToTensor
    from hnd_ghnd_object_detectors_amd.utils import data_util
    z = G.load_raw('tiny_input_pipeline')
    for u8, flip, tin, tout in G.pipeline_case(z):
        img, tgt = ToTensor()(u8.numpy(), {k: v.clone() for k, v in tin.items()})
        img, tgt = RandomHorizontalFlip(1.0 if flip else 0.0)(img, tgt)
        assert isinstance(img, DecodedImage) and img.hwc and img.flip == flip and img.data.dtype == torch.uint8
        assert tuple(img.shape) == (3, u8.shape[0], u8.shape[1])
        for k in tout:
            assert torch.equal(tgt[k], tout[k]), k
        ref = O.to_tensor_u8(u8)
        assert torch.equal(img.float_chw(), ref.flip(-1) if flip else ref)
        # float tensors still take the reference's eager route
        f_img, f_tgt = RandomHorizontalFlip(1.0)(ref.clone(), {k: v.clone() for k, v in tin.items()})
        o_img, o_tgt = O.horizontal_flip(ref, tin)
        assert torch.equal(f_img, o_img) and all(torch.equal(f_tgt[k], o_tgt[k]) for k in o_tgt)
    ld = data_util.SyntheticDetectionLoader(1, 4, 16, 24, 'keypoint_rcnn', decoded=True)
    images, targets = next(iter(ld))
    assert all(isinstance(im, DecodedImage) and tuple(im.shape) == (3, 16, 24) for im in images)
    for im, t in zip(images, targets):      # a flipped image carries mirrored boxes
        assert float(t['boxes'][0, 0]) == (24 - 0.5 * 24 if im.flip else 0.125 * 24)


def test_neural_filter_model_layout_labels_and_config():
    """the ext model keeps the reference's state-dict layout (oracle keys == reference keys, pinned by the
    tiny_ext_filter fixture generation), its label rule matches, and the generated YAML equals the schema."""
    from tests import golden_util as G
    from hnd_ghnd_object_detectors_amd.models.ext.backbone import check_if_valid_target
    s_sd, e_sd = MU.ext_states(5)
    cfg, model, ext = MU.build_ext_model(s_sd, e_sd, 'cpu', 64, 128)
    assert list(model.state_dict().keys())[:5] == list(s_sd.keys())[:5]
    assert set(model.state_dict().keys()) == set(s_sd) | set(e_sd)
    z, meta = G.load('tiny_ext_filter')
    import json
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    assert trainable == json.loads(str(z['param_names'])) == O.ext_trainable_keys({**s_sd, **e_sd})
    assert model.ext_training and model.backbone.body.ext_training and model.get_ext_classifier() is ext
    images, targets = G.ext_case_inputs(meta)
    cases = targets + [{}, {'boxes': torch.tensor([[1., 1., 5., 5.]])},
                       {'boxes': torch.tensor([[1., 1., 5., 0.5], [0., 0., 9., 9.]])}]
    for t in cases:
        assert check_if_valid_target(t) == O.valid_target(t)
    with pytest.raises(RuntimeError):           # parameter holders never compute eagerly
        ext.extractor[1](torch.zeros(1, 64, 8, 8))


def test_coco_format_loaders_without_pycocotools(tmp_path):
    """COCO-format folder -> datasets / aspect-ratio grouped loaders (reference data_util.py:18-48, coco_util.py):
    targets as the reference builds them, invalid images filtered, batches of one aspect-ratio bin, uint8 images."""
    from tests.coco_fixture import write_tiny_coco
    from hnd_ghnd_object_detectors_amd.structure.sampler import GroupedBatchSampler, create_aspect_ratio_groups
    from hnd_ghnd_object_detectors_amd.structure.transformer import DecodedImage
    from hnd_ghnd_object_detectors_amd.utils import coco_util, data_util
    img_dir, ann_file = write_tiny_coco(str(tmp_path))
    ds = coco_util.get_coco(img_dir, ann_file, data_util.get_transform(False), remove_non_annotated_imgs=False)
    assert len(ds) == 8 and ds.get_height_and_width(2) == (40, 80)
    img, t = ds[0]
    assert isinstance(img, DecodedImage) and tuple(img.shape) == (3, 48, 64) and img.data.dtype == torch.uint8
    assert t['boxes'].tolist() == [[8.0, 12.0, 40.0, 36.0]] and t['labels'].tolist() == [1]     # crowd anno dropped
    assert t['keypoints'].shape == (1, 17, 3) and t['image_id'].tolist() == [100]
    assert t['masks'].shape == (1, 48, 64) and int(t['masks'].sum()) == 32 * 24     # maskApi: w x h pixels exactly
    assert bool(t['masks'][0, 12:36, 8:40].all())
    assert t['area'].numel() == 1 and t['iscrowd'].tolist() == [0]
    _, t3 = ds[3]
    assert t3['boxes'].shape == (0, 4) and t3['masks'].shape == (0, 56, 56)                      # no annotations
    _, t6 = ds[6]
    assert t6['boxes'].shape == (1, 4)                  # a 0.5-wide box is kept by the converter ...
    kept = coco_util.remove_images_without_annotations(ds)
    assert [ds.ids[i] for i in kept.indices] == [100, 101, 102, 104, 107]   # ... but 103/105/106 are not valid images
    f_img, _ = coco_util.get_coco(img_dir, ann_file, data_util.get_transform(False, decoded=False), False)[1]
    assert f_img.dtype == torch.float32 and tuple(f_img.shape) == (3, 64, 48) and float(f_img.max()) <= 1.0
    groups = create_aspect_ratio_groups(ds, k=1)        # bins 0.5 | 1 | 2
    assert groups == [2, 1, 3, 2, 3, 2, 2, 1]
    sampler = torch.utils.data.SequentialSampler(ds)
    batches = list(GroupedBatchSampler(sampler, groups, 2))
    assert len(batches) == 4 and all(groups[a] == groups[b] for a, b in batches)
    assert batches[:3] == [[0, 3], [2, 4], [5, 6]] and batches[3] == [1, 7]
    cfg = {'num_workers': 0, 'aspect_ratio_group_factor': 1,
           'splits': {k: {'images': img_dir, 'annotations': ann_file, 'remove_non_annotated_imgs': k == 'train',
                          'jpeg_quality': None} for k in ('train', 'val', 'test')}}
    random.seed(0)
    torch.manual_seed(0)
    train_sampler, train_loader, val_loader, test_loader = data_util.get_coco_data_loaders(cfg, 2, False)
    assert len(train_loader) == 2 and len(val_loader) == 8
    for images, targets in train_loader:
        assert len(images) == 2 and all(isinstance(im, DecodedImage) for im in images)
        assert all(t['boxes'].shape[0] == 1 for t in targets)
        for im, t in zip(images, targets):              # RandomHorizontalFlip mirrored the box with the image
            w = im.shape[-1]
            x0 = float(t['boxes'][0, 0])
            assert abs(x0 - (w - 0.625 * w if im.flip else 0.125 * w)) < 1e-4
    with pytest.raises(FileNotFoundError):
        bad = {'num_workers': 0, 'aspect_ratio_group_factor': 1,
               'splits': {k: {'images': img_dir, 'annotations': '/nonexistent.json', 'remove_non_annotated_imgs': False,
                              'jpeg_quality': None} for k in ('train', 'val', 'test')}}
        data_util.get_coco_data_loaders(bad, 2, False)


def _relay_segments(T, kg8, G, lb):
    """Python mirror of the work split of csrc/conv_bstream.hip (bstream_kernel, relay mode): the units
    [U lb / G, U (lb + 1) / G) of the linear (tile, iteration) space as (tile, it0, it1, kind) segments in the order the
    workgroup processes them: head first, whole tiles, tail last."""
    U = T * kg8
    u0, u1 = U * lb // G, U * (lb + 1) // G
    tA, tB = u0 // kg8, u1 // kg8
    offA, offB = u0 - tA * kg8, u1 - tB * kg8
    first_full = tA + (1 if offA > 0 else 0)
    segs = []
    if offB > 0:
        segs.append((tB, 0, offB, 'head'))
    segs += [(t, 0, kg8, 'full') for t in range(first_full, tB)]
    if offA > 0:
        segs.append((tA, offA, kg8, 'tail'))
    return segs


def test_relay_work_split_covers_every_unit_once_and_pairs_heads_with_tails():
    """The B-streamed GEMM's relay: with at least one tile per workgroup every (tile, 128-k iteration) unit is computed
    exactly once, a workgroup holds at most one head and one tail (of DIFFERENT tiles), the tail of workgroup w + 1
    continues exactly where the head of w stopped, and the loads' balance is within one iteration."""
    import random
    rnd = random.Random(3)
    G = 256
    for _ in range(200):
        kg8 = rnd.choice([2, 4, 8, 9, 16, 18, 36])
        T = rnd.randint(G, 6 * G)
        seen = {}
        heads, tails, work = {}, {}, []
        for lb in range(G):
            segs = _relay_segments(T, kg8, G, lb)
            kinds = [s[3] for s in segs]
            assert kinds.count('head') <= 1 and kinds.count('tail') <= 1
            assert kinds == sorted(kinds, key=lambda k: ('head', 'full', 'tail').index(k))
            work.append(sum(s[2] - s[1] for s in segs))
            for tile, it0, it1, kind in segs:
                assert 0 <= tile < T and 0 <= it0 < it1 <= kg8
                for it in range(it0, it1):
                    assert (tile, it) not in seen
                    seen[(tile, it)] = lb
                if kind == 'head':
                    heads[lb] = (tile, it1)
                elif kind == 'tail':
                    tails[lb] = (tile, it0)
            if 'head' in kinds and 'tail' in kinds:
                assert segs[0][0] != segs[-1][0]
        assert len(seen) == T * kg8
        assert max(work) - min(work) <= 1 and min(work) >= kg8
        assert 0 not in tails and (G - 1) not in heads
        for lb, (tile, it1) in heads.items():
            assert tails.get(lb + 1) == (tile, it1), (lb, heads[lb], tails.get(lb + 1))
        assert len(tails) == len(heads)


def test_side_streams_only_when_the_process_owns_its_gpu(monkeypatch):
    """engine.process_owns_device / wgrad_stream_on and tool._defer_fpn_default: the optional side streams (pyramids,
    the head's weight gradients) default on only when there are no more local ranks than visible devices; the
    environment switches decide when set"""
    from hnd_ghnd_object_detectors_amd import engine as E
    from hnd_ghnd_object_detectors_amd.distillation import tool
    for k in ('LOCAL_WORLD_SIZE', 'WORLD_SIZE', 'HND_WGRAD_STREAM', 'HND_DEFER_FPN', 'HND_SHARED_DEVICE',
              'ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(k, raising=False)
    assert E.process_owns_device() and E.wgrad_stream_on() and tool._defer_fpn_default()
    # decide from what is actually shared
    monkeypatch.setenv('WORLD_SIZE', '16')                  # multi-node launcher without LOCAL_WORLD_SIZE: says nothing
    assert E.process_owns_device()
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    assert not E.process_owns_device()                      # 8 local ranks, one visible "device", list not narrowed
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '3')          # SLURM-style: one GPU per rank
    assert E.process_owns_device()
    monkeypatch.setenv('HND_SHARED_DEVICE', '1')            # bench.py --share_device
    assert not E.process_owns_device()
    for k in ('WORLD_SIZE', 'HIP_VISIBLE_DEVICES', 'HND_SHARED_DEVICE'):
        monkeypatch.delenv(k)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '4')             # (no GPU here: one "device")
    assert not E.process_owns_device() and not E.wgrad_stream_on() and not tool._defer_fpn_default()
    monkeypatch.setenv('HND_WGRAD_STREAM', '1')
    monkeypatch.setenv('HND_DEFER_FPN', '1')
    assert E.wgrad_stream_on() and tool._defer_fpn_default()
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '1')
    monkeypatch.setenv('HND_WGRAD_STREAM', '0')
    assert E.process_owns_device() and not E.wgrad_stream_on()


def test_step_loss_without_a_host_copy_behaves_like_a_tensor():
    """hip_loss.StepLoss: results of arithmetic on a loss (no early host copy of their own) fall back to Tensor.item();
    a loss with a copy returns the copied float after waiting for its event; autograd still reaches the function"""
    from hnd_ghnd_object_detectors_amd.distillation.hip_loss import StepLoss

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, v, p):
            return v.clone()

        @staticmethod
        def backward(ctx, g):
            return None, torch.full((3,), 2.0) * g

    waited = []
    p = torch.zeros(3, requires_grad=True)
    loss = Fn.apply(torch.tensor(2.5), p).as_subclass(StepLoss)
    loss._host = (torch.tensor(2.5), type('Ev', (), {'synchronize': lambda self: waited.append(1)})())
    assert isinstance(loss, StepLoss) and loss.requires_grad
    assert loss.item() == 2.5 and float(loss) == 2.5 and waited == [1, 1]
    doubled = loss * 2
    assert doubled.item() == 5.0 and len(waited) == 2       # (no copy of its own: plain Tensor.item)
    loss.backward()
    assert torch.equal(p.grad, torch.full((3,), 2.0))


def test_trace_and_picker_tools_on_synthetic_input(tmp_path):
    """tools/idle_gaps.py and tools/compare_pickers.py parse what rocprofv3 / bench.py --detail write"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    trace = tmp_path / 'trace.csv'
    rows = ['"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"']
    t = 0
    for step in range(5):
        for name, dur, gap in (('conv_a', 1000000, 0), ('conv_b', 2000000, 20000), ('adam_kernel', 10000, 0)):
            t += gap
            rows.append('"KERNEL_DISPATCH","%s",%d,%d' % (name, t, t + dur))
            t += dur
        t += 400000                                          # the host catches up between steps
    trace.write_text('\n'.join(rows) + '\n')
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'idle_gaps.py'), str(trace), '--steps', '2',
                          '--skip_last', '1'], capture_output=True, text=True, check=True).stdout
    assert 'window: 2 steps, 6 kernel launches' in out and 'idle 0.220' in out, out
    a, b = tmp_path / 'a.txt', tmp_path / 'b.txt'
    head = 'launch                             kernel           n        ms     GFLOP  TFLOP/s  shape\n'
    a.write_text(head + 'layer2.0.conv2.dgrad  bstream_128  3  0.817  70.46  86.23\nfpn.layer0  bres2_128  2  4.0  511.1  127.0\n')
    b.write_text(head + 'layer2.0.conv2.dgrad  igemm_128x128  4  0.780  79.27  101.5\nfpn.layer0  bres2_128  2  4.0  511.1  127.0\n')
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'compare_pickers.py'), str(a), 'tiled=' + str(b)],
                         capture_output=True, text=True, check=True).stdout
    assert '+ layer2.0.conv2.dgrad' in out and "'tiled': 0.037" in out, out
