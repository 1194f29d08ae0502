"""haconvdr_amd — MI355X-native dense-retrieval hot path for HAConvDR.

Exact inner-product top-k search (the reference's faiss.IndexFlatIP use) and the
ANCE/RoBERTa encoder as hand-written gfx950 HIP kernels behind a C-ABI
(include/haconvdr.h, built to haconvdr_amd/csrc/libhaconvdr.so).  There is no CPU
fallback: every compute entry point raises if the HIP library is missing.
"""
__version__ = "0.6.0"
