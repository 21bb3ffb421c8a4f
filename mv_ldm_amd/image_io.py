"""Image output for the generation harness: the counterpart of `src/misc/image_io.py:42-73`
(`prep_image`: clip to [0,1], x255, truncate to uint8, HWC; `save_image`: create the parent directory, write).
The reference writes through PIL; here the PNG container is produced directly (zlib + CRC from the standard
library: an 8-bit RGB / RGBA / grey image is a filter byte + raw row per scanline, deflated)."""
from __future__ import annotations

import struct
import zlib
from pathlib import Path
from typing import Union

import numpy as np
import torch


def prep_image(image: torch.Tensor) -> np.ndarray:
    """[3|4, H, W] / [1, H, W] / [H, W] float in [0, 1] -> uint8 [H, W, C] (image_io.py:42-58: truncation, not rounding)"""
    if image.dim() == 2:
        image = image[None]
    if image.shape[0] == 1:
        image = image.expand(3, -1, -1)
    assert image.dim() == 3 and image.shape[0] in (3, 4), tuple(image.shape)
    image = (image.detach().float().clip(min=0, max=1) * 255).to(torch.uint8)
    return image.permute(1, 2, 0).contiguous().cpu().numpy()


def encode_png(pixels: np.ndarray) -> bytes:
    h, w, c = pixels.shape
    assert pixels.dtype == np.uint8 and c in (3, 4)
    color_type = 2 if c == 3 else 6

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    raw = np.concatenate([np.zeros((h, 1), np.uint8), pixels.reshape(h, w * c)], axis=1).tobytes()   # filter 0 per row
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def save_image(image: torch.Tensor, path: Union[Path, str]) -> None:
    """Save an image assumed to be in range 0-1 (image_io.py:61-73)."""
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    path.write_bytes(encode_png(prep_image(image)))


def decode_png(data: bytes) -> np.ndarray:
    """inverse of `encode_png` for its own output (filter-0 rows only); used by the tests and for round trips"""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF), "bad CRC"
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert depth == 8 and ctype in (2, 6)
            c = 3 if ctype == 2 else 4
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * c)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, c).copy()
