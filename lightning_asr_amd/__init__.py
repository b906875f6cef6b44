"""lightning_asr_amd — MI355X (gfx950) native QuartzNet-CTC training hot path.

The package directory is ``lightning_asr_amd`` (a valid Python identifier; the project name
"lightning-asr_amd" is not importable).  Hot-path compute lives in ``csrc/*.hip`` behind the C ABI
in ``include/lasr.h``; the Python here mirrors the reference's call surface and holds no math.
"""
__all__ = ["_lib", "ops", "engine"]
