"""chexpert_amd -- MI355X-native (gfx950) conv hot path for the CheXpert classifiers.

Product code: hand-written HIP kernels behind a C ABI (`include/chexpert_hip.h`,
`chexpert_amd/libchexpert_hip.so`) plus the Python host side that mirrors the reference's
`nn.Module` surface.  There is no CPU fallback: ops raise if the library is missing or a tensor is
not on the GPU.  (`oracle/` is test infrastructure and is never imported from here.)
"""
__version__ = "0.1.0"
