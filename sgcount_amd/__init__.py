"""sgcount_amd — MI355X-native drop-in for the count path of noamteyssier/sgcount.

Layout: csrc/ holds the gfx950 kernels and the C ABI (include/sgcount_hip.h);
host.py mirrors the reference's Library / Permuter / Counter / Offset interface on top of it.
Importing the package does not load the extension; the first use does, and raises if the
HIP library cannot be built or loaded (there is no CPU fallback).
"""
from . import _ffi, build  # noqa: F401
from .host import (Counter, DeviceLibrary, Library, Offset, Permuter, Record, initialize_reader,  # noqa: F401
                   pack_reads_host, parse_fastx, read_path)

__all__ = ["Counter", "DeviceLibrary", "Library", "Offset", "Permuter", "Record", "initialize_reader",
           "pack_reads_host", "parse_fastx", "read_path"]
