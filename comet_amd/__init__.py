"""comet_amd — MI355X (gfx950) backend for the vector-search hot path of wizenheimer/comet.

The compute lives in ``libcomet_hip.so`` (hand-written HIP kernels behind the C ABI of
``include/comet_gpu.h``); this package is the thin host-side mirror of the reference's Go interfaces.
"""
from ._lib import CometError, ZeroVectorError, load, LIB_PATH  # noqa: F401
from .index import (  # noqa: F401
    COSINE, EUCLIDEAN, L2_SQUARED, MAX_AGGREGATION, MEAN_AGGREGATION, SUM_AGGREGATION, Context, FlatIndex, IVFIndex,
    IVFPQIndex, PQIndex, HNSWIndex, BM25SearchIndex, SegmentSet, TextResult, TextSearch, UnknownDistanceKind, VectorIndex, VectorResult, VectorSearch, aggregate, autocut, sanitize_k,
)
