"""Import-path mirror of scheduler/cosine_annearing_with_warmup.py (host scalar math, see ../schedule.py)."""
from ..schedule import CosineAnnealingWarmupRestarts  # noqa: F401
