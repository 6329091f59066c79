"""MI355X-native drop-in for DLMC-QUANT's fake-quantize path.

Import name `dlmc` on purpose: put `dlmc-quant_amd/` ahead of the reference checkout on
`sys.path` and the reference's trainers (`from dlmc.utils.quantize import quantize_model`,
`from dlmc.quantization.scalar.FSPTQuant import FSPTQBase`, ...) pick up the HIP path unchanged.
Sub-modules this overlay does not provide (merge_bn, tracker, count_operations - off the hot
path) still resolve to the reference's own files through `extend_path`.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
__version__ = "0.1.0"
