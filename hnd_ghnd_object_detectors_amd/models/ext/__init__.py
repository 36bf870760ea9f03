"""Neural-filter (ext) backbone: a different runner/workload in the reference (src/ext_runner.py);
outside this build's hot path."""


def get_ext_fpn_backbone(base_backbone, ext_config, freeze_layers):
    raise NotImplementedError('ext_config (neural filter) models are outside this build (SURVEY.md 8f-f2)')
