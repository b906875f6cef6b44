"""``MyModel2`` of the reference's models/QuartNet.py, on the native HIP plan (variant "plain")."""
from ._base import MyModel2Base


class MyModel2(MyModel2Base):
    variant = "plain"
