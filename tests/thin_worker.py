"""Worker of tests/test_ops_gpu.py::test_thin_n_kernel_is_bit_identical_to_the_mfma_path: runs the cout <= 4 launches
and two whole distillation steps in THIS process' mode (HND_THIN_N=0: MFMA tiles, 1: vector-ALU kernel; the library reads
the switch once) and saves every result for a bit-for-bit comparison.
usage: HND_THIN_N=<0|1> python tests/thin_worker.py <out.pt>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    from hnd_ghnd_object_detectors_amd import ops
    from hnd_ghnd_object_detectors_amd.distillation.tool import DistillationBox
    from hnd_ghnd_object_detectors_amd.myutils.pytorch import func_util
    from tests import golden_util as G
    from tests import model_util as MU
    dev = torch.device('cuda:0')
    res = {'variants': []}
    g = torch.Generator().manual_seed(5)
    # (n, h, w, cin, cout, k, prologue, stats, mask/residual)
    cases = [(2, 33, 47, 64, 3, 2, True, True, False), (3, 20, 31, 64, 3, 2, False, False, True),
             (1, 9, 14, 128, 4, 1, True, True, True), (2, 16, 16, 64, 1, 3, False, True, False)]
    for ci, (n, h, w, cin, cout, k, pro, stats, extra) in enumerate(cases):
        pad = 1 if k == 3 else 0
        x = torch.randn(n, h, w, cin, generator=g).to(dev)
        wt = (torch.randn(cout, cin, k, k, generator=g) * 0.2).to(dev)
        oh, ow = ops.conv_out_size(h, k, 1, pad), ops.conv_out_size(w, k, 1, pad)
        ldc = 4
        y = torch.zeros(n, oh, ow, ldc, device=dev)
        kw = {}
        if pro:
            kw.update(pro_scale=(torch.rand(cin, generator=g) + 0.5).to(dev), pro_shift=torch.randn(cin, generator=g).to(dev),
                      pro_relu=True)
        if stats:
            kw['stats'] = torch.zeros(ops.stats_tiles(n * oh * ow) * 2 * cout, device=dev)
        if extra:
            kw.update(res1=torch.randn(n, oh, ow, ldc, generator=g).to(dev),
                      mask=(torch.rand(n, oh, ow, ldc, generator=g) - 0.3).to(dev), relu=True,
                      epi_scale=(torch.rand(cout, generator=g) + 0.5).to(dev), epi_shift=torch.randn(cout, generator=g).to(dev))
        l = ops.conv_forward(x, ops.pack_weights(wt), y, k, 1, pad, cout=cout, **kw)
        l.run()
        res['variants'].append(l.variant)
        res['fwd%d/y' % ci] = y.cpu()
        if stats:
            res['fwd%d/stats' % ci] = kw['stats'].cpu()
        # the data gradient of a cout_w -> 3 ... i.e. of a conv with 3 INPUT channels stored as 4: dy has cin channels here
        if k == 2:
            wd = (torch.randn(cin, cout, k, k, generator=g) * 0.2).to(dev)        # OIHW of the forward conv cout -> cin
            dy = torch.randn(n, h - 1, w - 1, cin, generator=g).to(dev)
            dx = torch.zeros(n, h, w, 4, device=dev)
            launches, _ = ops.conv_dgrad(dy, wd, dx, k, 1, 0)
            for ll in launches:
                ll.run()
            res['variants'].append(launches[0].variant)
            res['dgrad%d/dx' % ci] = dx.cpu()
    # two whole steps of the tiny GHND fixture (conv3 forward + BN statistics, conv4 data gradient inside)
    _, meta = G.load('tiny_ghnd_faster')
    cfg = MU.config_for(meta)
    t_sd, s_sd = MU.oracle_states(meta['seed'])
    teacher, student = MU.build_pair(cfg, t_sd, s_sd, dev)
    box = DistillationBox(teacher, student, cfg['train']['criterion'])
    opt = func_util.get_optimizer(student, 'Adam', {'lr': 1e-3})
    images, targets = G.case_inputs(meta)
    images = [im.to(dev) for im in images]
    targets = [{k2: v.to(dev) for k2, v in t.items()} for t in targets]
    losses = []
    for _ in range(2):
        opt.zero_grad()
        loss = box(images, targets)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    res['losses'] = losses
    res['params'] = {n2: p.detach().cpu().clone() for n2, p in student.named_parameters() if p.requires_grad}
    res['buffers'] = {n2: b.detach().cpu().clone() for n2, b in student.named_buffers() if 'layer1' in n2}
    torch.save(res, out)


if __name__ == '__main__':
    main()
