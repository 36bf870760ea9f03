"""CPU oracle for the HND/GHND distillation step.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from this package; the product path
(``hnd_ghnd_object_detectors_amd``) must never import it and fails loudly when
the HIP library is missing.

Contents
--------
``tv042``        restatement of the torchvision==0.4.2 pieces the reference path
                 leans on (``Pipfile:8``; third-party, absent from /root/reference).
``myutils_r``    restatement of the un-vendored ``myutils`` submodule symbols the
                 path touches (``.gitmodules:1-3``; unpinned master).
``hnd_oracle``   functional (state-dict driven) pure-torch CPU restatement of the
                 reference's distillation step; fp32 and fp64.
``shim/``        import shim (``torchvision``/``myutils``/``pycocotools``) that lets
                 ``/root/reference/src`` import UNMODIFIED in the build container;
                 used only by ``tests/golden/make_golden.py``.

Pinning status: the reference ships no tests, fixtures or golden vectors
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, run in the build container over ``shim/`` by
``tests/golden/make_golden.py`` and committed under ``tests/golden/*.npz``.
"""
