"""dismember_amd — MI355X-native tree beam-search retrieval (TDM / JTM / OTM hot path).

Host-side mirror of the reference's facades over the C ABI in include/dismember_hip.h.
"""
from .engine import DismemberError, Engine  # noqa: F401
from .facade import OTM, TDM, DeepRetrieval  # noqa: F401
