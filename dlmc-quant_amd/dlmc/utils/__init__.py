"""Model-conversion boundary (reference: dlmc/utils).  `merge_bn`, `tracker`, `count_operations` are
off the hot path and resolve to the reference's own files when its checkout is on sys.path."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
