"""Seed corpora of well-formed and malformed inputs for the host parsers (OBJ text, PNG streams) -- the inputs the reference fuzzes
(fuzz/target_mesh_parser.cpp, fuzz/target_image_io_read.cpp): truncated faces, huge indices, numbers spilling over line ends, bad PNG
chunks / checksums / filters / headers."""
import os
import random
import struct
import zlib


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def png(width, height, colour, rows, interlace=0, depth=8):
    header = struct.pack(">IIBBBBB", width, height, depth, colour, 0, 0, interlace)
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", header) + _chunk(b"IDAT", zlib.compress(rows)) + _chunk(b"IEND", b"")


def obj_inputs():
    texts = [
        b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
        b"v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3 4\nf 1/1/1 2/2/2 3/3/3\n",
        b"v 1e 2 -\nv 1 2 3\nf 99999999999 1 2\nf 1 2 3\nf -1 -2 -3\nf 0 0 0\n",
        b"v 0.25 1.5\nv 3 4 5\nf 3 4\n5\nf\nf 1\nv\n# comment\nvn 0 0 1\nvt 0 0\ng group\n",
        b"v " + b"9" * 400 + b" 1 2\nv 1e400 -1e400 nan\nv inf -inf 0x10\nf 1 2 3\n",
        b"f 1 2 3\n" * 50,
        b"v 0 0 0\n" * 3 + b"f 1 2 3",
        b"\r\nv 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\n",
        bytes(range(256)),
    ]
    # the fuzz target's first two bytes are cull_backface and smooth
    return [flags + t for t in texts for flags in (b"\x00\x00", b"\x01\x01", b"\x00\x01")]


def png_inputs():
    rnd = random.Random(1)
    channels = {0: 1, 2: 3, 4: 2, 6: 4}
    out = []
    for colour in (0, 2, 4, 6):
        for filt in range(5):
            w, h = 5, 4
            rows = b"".join(bytes([filt]) + bytes(rnd.randrange(256) for _ in range(w * channels[colour])) for _ in range(h))
            out.append(png(w, h, colour, rows))
    good = out[-1]
    bad_crc = bytearray(good)
    bad_crc[30] ^= 1
    out += [good[:20], good[:40], good[:-6], good.replace(b"IDAT", b"IDAX"), good[:33] + good[45:], good + b"junk", bytes(bad_crc),
            png(65535, 65535, 6, b"\0" * 10), png(0, 5, 6, b""), png(5, 4, 6, b"\x07" + b"\0" * 100), png(5, 4, 3, b"\0" * 30), png(5, 4, 6, b"\0" * 5),
            png(4, 4, 6, b"\0" * 200, interlace=1), png(4, 4, 6, b"\0" * 200, depth=16),
            b"\x89PNG\r\n\x1a\n" + _chunk(b"IDAT", zlib.compress(b"\0" * 30)) + _chunk(b"IEND", b""),
            b"\x89PNG\r\n\x1a\n" + struct.pack(">I", 0xFFFFFFFF) + b"IHDR", b"", b"\x89PNG\r\n\x1a\n"]
    return out


def write(directory, inputs):
    os.makedirs(directory, exist_ok=True)
    for i, data in enumerate(inputs):
        with open(os.path.join(directory, "seed%03d" % i), "wb") as f:
            f.write(data)
