from .encoder import Encoder
from .decoder import Decoder
from .feature_retrieval import match_features

# The reference also exports `Discriminator` here; it is training-only and outside this package's
# scope.
__all__ = ["Encoder", "Decoder", "match_features"]
