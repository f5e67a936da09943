"""zlib.es_amd -- MI355X-native engine behind the zlib.es compression API.

Layout: ``csrc/`` HIP kernels + C ABI (libzgpu.so), ``_native.py`` ctypes
binding, ``api.py`` the host-side mirror of the reference's classes
(RawDeflate / Deflate / Gzip / Zip / CRC32 / Adler32), ``js/`` the Node.js
N-API addon and the JavaScript shim with the same class surface.

The directory name contains a dot, so import it as ``zlib_es_amd`` (the loader
module at the repository root maps that name onto this directory).
"""
from . import _native  # noqa: F401
from ._native import (DYNAMIC, FIXED, NONE, Plan, ZgpuError, adler32_combine,  # noqa: F401
                      checksum, crc32_combine, deflate_batch, deflate_bound, set_search)
from .api import Adler32, CRC32, Deflate, Gzip, RawDeflate, Zip  # noqa: F401
from .shard import combine_checksums, compress_sharded, gzip_chunked, partition_by_bytes, zlib_chunked  # noqa: F401
