from oracle.pycoco_r import COCOeval, Params  # noqa
