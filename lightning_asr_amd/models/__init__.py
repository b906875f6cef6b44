"""Model surface of the reference's models/ package (MyModel2 in three files), backed by the native
plan in csrc/model.hip.  Import paths mirror the reference: ``from models.QuartNet import MyModel2``
becomes ``from lightning_asr_amd.models.QuartNet import MyModel2``."""
from ._base import MyModel2Base  # noqa: F401
