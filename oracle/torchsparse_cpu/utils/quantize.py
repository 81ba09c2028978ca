from ...ts_ref import sparse_quantize

__all__ = ['sparse_quantize']
