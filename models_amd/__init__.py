"""models_amd -- the MI355X-native hot path behind the Merlin Models ``mm`` surface.

``import models_amd as mm`` gives the names the reference exports at
merlin/models/tf/__init__.py:21-132 for the retrieval / ranking hot path.
"""
from .schema import ColumnSchema, Schema, Tags  # noqa: F401
from .core import (  # noqa: F401
    Block, ConcatFeatures, Filter, ParallelBlock, Parameter, SequentialBlock, StackFeatures, TabularBlock,
)
from .inputs import (  # noqa: F401
    Continuous, ContinuousFeatures, EmbeddingFeatures, EmbeddingOptions, EmbeddingTable, Embeddings, EmbeddingsBlock,
    InputBlock, InputBlockV2, Ragged, infer_embedding_dim,
)
from .blocks import (  # noqa: F401
    Activation, BatchNormalization, Cross, CrossBlock, DLRMBlock, DotProductInteraction, DotProductInteractionBlock, Dropout, MLPBlock,
    TwoTowerBlock, set_seed,
)
from .outputs import (  # noqa: F401
    MIN_FLOAT, BinaryOutput, BruteForce, ContrastiveOutput, DotProduct, Prediction, TopKOutput, TopKPrediction,
)
from .models import (  # noqa: F401
    DCNModel, DLRMModel, Encoder, Model, RankingModel, RetrievalModel, TopKEncoder, TwoTowerModel, TwoTowerModelV2,
)
from .loader import Loader  # noqa: F401
from .optim import LazyAdam  # noqa: F401
from .compat import (  # noqa: F401
    AvgPrecisionAt, BinaryClassificationTask, ItemRetrievalScorer, ItemRetrievalTask, L2Norm, LogitsTemperatureScaler, MRRAt,
    MultiOptimizer, NDCGAt, OptimizerBlocks, PrecisionAt, RecallAt, TopKMetricsAggregator, split_embeddings_on_size,
)
from .sampling import (  # noqa: F401
    CachedCrossBatchSampler, Candidate, CandidateSampler, FIFOQueue, InBatchSamplerV2, PopularityBasedSamplerV2,
    PopularityLogitsCorrection,
)

__version__ = "0.1.0"
