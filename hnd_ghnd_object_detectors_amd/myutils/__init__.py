"""Subset of the reference's un-vendored ``myutils`` submodule that the distillation path uses
(.gitmodules:1-3; call sites listed in SURVEY.md A.3), re-implemented for this package."""
