"""Mirrors the reference's scheduler/ package: Novograd and CosineAnnealingWarmupRestarts."""
