"""BGZF writer (bench / tests only): gzip members of <= 64 KiB that announce their own size in a 'BC' extra field — what
bgzip, htslib and Illumina's converters write, and what the host's TextFeeder inflates with several threads
(sgcount_amd/csrc/host/sgh.cpp run_bgzf).  Any gzip reader sees an ordinary multi-member stream.

    python -m sgcount_amd.bgzf SRC LO HI DST      compress bytes [LO, HI) of SRC into DST (no end-of-file marker)
"""
import struct
import sys
import zlib

BLOCK = 65280
EOF_MARKER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def member(chunk: bytes, level: int = 1) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(chunk) + co.flush()
    total = 12 + 6 + len(body) + 8
    if total > 65536:                      # incompressible data: store it
        co = zlib.compressobj(0, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        total = 12 + 6 + len(body) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", total - 1) + body +
            struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def bgzf_bytes(data: bytes, block: int = BLOCK, level: int = 1, eof_marker: bool = True) -> bytes:
    out = [member(data[i:i + block], level) for i in range(0, len(data), block)]
    if eof_marker:
        out.append(EOF_MARKER)
    return b"".join(out)


def compress_range(src: str, lo: int, hi: int, dst: str, level: int = 1):
    with open(src, "rb") as f, open(dst, "wb") as o:
        f.seek(lo)
        left = hi - lo
        while left > 0:
            chunk = f.read(min(BLOCK, left))
            if not chunk:
                break
            o.write(member(chunk, level))
            left -= len(chunk)


if __name__ == "__main__":
    compress_range(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
