"""``MyModel2`` of the reference's models/QuartNetContext.py, on the native HIP plan (variant "context")."""
from ._base import MyModel2Base


class MyModel2(MyModel2Base):
    variant = "context"
