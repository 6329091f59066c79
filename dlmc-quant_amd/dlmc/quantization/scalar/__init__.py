"""Scalar (uniform) quantisation: HIP-backed primitives, observers and the quant wrappers."""
