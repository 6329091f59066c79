from .base import QBase
from .function import FunLQ, FunLSQ, FunRootQ, FunUniformQ
from .layers import QConv2d, QLinear

__all__ = ["QBase", "QConv2d", "QLinear", "FunUniformQ", "FunLSQ", "FunRootQ", "FunLQ"]
