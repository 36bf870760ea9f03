"""import-only stub (BoxCoder etc. are outside the hot path)."""
