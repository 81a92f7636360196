"""ntedit_amd -- MI355X-native implementation of ntEdit's k-mer Bloom-filter
membership + edit-search hot path."""
from ._lib import NtEditHipError, Params, Stats, LIB_PATH  # noqa: F401
from .polisher import Polisher, default_params, pack_batch, PRIMARY, SECONDARY  # noqa: F401
