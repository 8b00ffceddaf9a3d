from .radam import RAdam

__all__ = ["RAdam"]
