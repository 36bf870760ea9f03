"""import-only stub: COCO evaluation is outside the hot path."""
