"""Counterpart of ``dgl.nn.pytorch.softmax`` (reference models.py:5,153)."""
from .autograd import edge_softmax

__all__ = ["edge_softmax"]
