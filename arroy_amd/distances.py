"""`arroy::distances` (src/lib.rs:145-150): the metric marker types of the `Distance` trait."""
from __future__ import annotations


class Distance:
    """Mirror of the static part of `trait Distance` (src/distance/mod.rs:40-124)."""
    metric: int = -1
    name: str = ""
    DEFAULT_OVERSAMPLING: int = 1  # mod.rs:41
    binary_quantized: bool = False

    @classmethod
    def header_size(cls) -> int:
        return 8 if cls.metric == 3 else 4

    @classmethod
    def vector_size(cls, dimensions: int) -> int:
        return ((dimensions + 63) // 64) * 8 if cls.binary_quantized else 4 * dimensions


class Euclidean(Distance):
    metric, name = 0, "euclidean"


class Manhattan(Distance):
    metric, name = 1, "manhattan"


class Cosine(Distance):
    metric, name = 2, "cosine"


class DotProduct(Distance):
    metric, name = 3, "dot-product"


class BinaryQuantizedEuclidean(Distance):
    metric, name = 4, "binary quantized euclidean"
    DEFAULT_OVERSAMPLING = 3  # src/distance/binary_quantized_euclidean.rs:37
    binary_quantized = True


class BinaryQuantizedManhattan(Distance):
    metric, name = 5, "binary quantized manhattan"
    DEFAULT_OVERSAMPLING = 3
    binary_quantized = True


class BinaryQuantizedCosine(Distance):
    metric, name = 6, "binary quantized cosine"
    DEFAULT_OVERSAMPLING = 3
    binary_quantized = True


ALL = [Euclidean, Manhattan, Cosine, DotProduct, BinaryQuantizedEuclidean, BinaryQuantizedManhattan,
       BinaryQuantizedCosine]
BY_METRIC = {d.metric: d for d in ALL}
