"""auroralib.compression_amd -- MI355X-native batched LZ codec behind AuroraLib.Compression's
ICompressionAlgorithm surface.  See DESIGN.md / INTEGRATION.md."""
from . import _abi  # noqa: F401
