"""``MyModel2`` of the reference's models/QuartNetContextSE.py, on the native HIP plan (variant "context_se")."""
from ._base import MyModel2Base


class MyModel2(MyModel2Base):
    variant = "context_se"
